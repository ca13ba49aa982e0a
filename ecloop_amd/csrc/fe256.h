// fe256.h — secp256k1 base-field arithmetic for gfx950 (device; also compiles for the host for CPU-side tests).
//
// Replaces the reference's fe_modp_* family (lib/ecc.c:269-540) on the device.
//
// Representation: 9 limbs of 29 bits in u32 ("9x29", value = sum n[i] * 2^(29 i), top limb 24 bits), one lane = one
// element.  Why not the reference's 4 x u64 (or 8 x u32) saturated limbs: on gfx950 v_mad_u64_u32 (32x32+64 -> 64)
// issues at the same rate as any other VOP3 instruction (profiles/ubench_r01.txt) but it has no carry-in, and a
// carry chain costs an extra instruction plus wait states per link.  With unsaturated limbs a whole column of the
// schoolbook product accumulates in one 64-bit register with no carry handling (9 products of < 2^58 * m1*m2),
// additions and subtractions are 9 independent 32-bit ops (no carry chain, no reduction), and the reduction by
// 2^256 = 0x1000003D1 (mod p) is two small multiplies per column.  29 bits is the widest limb that still leaves
// room for lazy additions: 81 products per multiplication instead of 100 with 26-bit limbs.
//
// Magnitude discipline: a value has magnitude m if n[i] <= m * 2^29 (+ a few units) for i < 8 and n[8] <= m * 2^24.
// fe_mul accepts magnitudes with m1*m2 <= 7 (9 * 7 * 2^58 < 2^64), fe_sqr magnitude <= 2; both return magnitude 1;
// fe_add adds magnitudes; fe_neg(a, m) returns magnitude m+1; limbs must stay below 2^32, i.e. m <= 7.
// Only fe_normalize() gives the canonical residue in [0, p), and only canonical values are serialised for hashing,
// so results are bit-identical to the reference, which hashes canonical values only (DESIGN.md "Canonical form").
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
// host-only build of the same source (tests/test_devsrc_host.py compiles these headers with g++ and checks
// them against the oracle on the CPU; that covers the logic, the GPU tests cover the generated code)
#define __host__
#define __device__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#endif

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

#define FE_LIMBS 9
struct fe {
  u32 n[FE_LIMBS];
};

#define FE_FN __host__ __device__ __forceinline__
// Two code-generation pins for the device build (no effect on the value computed):
//  * FE_PIN64(x): the compiler otherwise re-associates "carry + sum of products" into "sum of products, then add
//    the carry" (one extra 64-bit add per column); pinning every accumulate makes each one a v_mad_u64_u32 that
//    continues the chain;
//  * fe_opaque(c): keeps a multiply by a power-of-two constant (R1 = 2^8) a single v_mad_u64_u32 instead of
//    mask + move + 64-bit shift + 64-bit add.
#if defined(__HIP_DEVICE_COMPILE__)
#define FE_PIN64(x) asm("" : "+v"(x))
FE_FN u32 fe_opaque(u32 c) {
  asm("" : "+s"(c));
  return c;
}
#else
#define FE_PIN64(x) ((void)0)
FE_FN u32 fe_opaque(u32 c) { return c; }
#endif
// Compiler-bug guard (ROCm 7.2 clang 22, gfx950): when BOTH factors of a 64-bit product are visibly masked to <= 24
// bits the AMDGPU backend first treats the multiply as a 24-bit one (which lets it drop the masks) and then
// selects v_mad_u64_u32 on the unmasked registers: the top limb of a field element squared as (c & 0xFFFFFF)^2
// came out as (low32(c))^2 whenever one multiplication fed another directly.  Reproduced and localised with an
// instruction-level emulation of the generated code; tests: diag ops 6-8 in tests/test_gpu_primitives.py.
// FE_HIDE24 makes the masked value opaque so the product is an ordinary 32 x 32 multiply of the masked register.
#if defined(__HIP_DEVICE_COMPILE__)
#define FE_HIDE24(x) asm("" : "+v"(x))
#else
#define FE_HIDE24(x) ((void)0)
#endif
#define FE_M 0x1FFFFFFFu
#define FE_TOP 0x00FFFFFFu /* limb 8: 24 bits */
#define FE_R0 0x7A20u      /* 2^261 = R1 * 2^29 + R0 (mod p): 32 * (2^32 + 977) */
#define FE_R1 0x100u

// p in 9x29 limbs
#define FE_P0 0x1FFFFC2Fu
#define FE_P1 0x1FFFFFF7u
#define FE_PM 0x1FFFFFFFu /* limbs 2..7 */
#define FE_P8 0x00FFFFFFu

FE_FN fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) r.n[i] = 0;
  return r;
}
FE_FN fe fe_one() {
  fe r = fe_zero();
  r.n[0] = 1;
  return r;
}

// 8 canonical little-endian u32 words (the reference's fe, lib/ecc.c:26) <-> limbs
FE_FN fe fe_from_words(const u32 w[8]) {
  fe r;
  r.n[0] = w[0] & FE_M;
  r.n[1] = (w[0] >> 29 | w[1] << 3) & FE_M;
  r.n[2] = (w[1] >> 26 | w[2] << 6) & FE_M;
  r.n[3] = (w[2] >> 23 | w[3] << 9) & FE_M;
  r.n[4] = (w[3] >> 20 | w[4] << 12) & FE_M;
  r.n[5] = (w[4] >> 17 | w[5] << 15) & FE_M;
  r.n[6] = (w[5] >> 14 | w[6] << 18) & FE_M;
  r.n[7] = (w[6] >> 11 | w[7] << 21) & FE_M;
  r.n[8] = w[7] >> 8;
  return r;
}
// a must be normalised
FE_FN void fe_to_words(u32 w[8], const fe& a) {
  w[0] = a.n[0] | a.n[1] << 29;
  w[1] = a.n[1] >> 3 | a.n[2] << 26;
  w[2] = a.n[2] >> 6 | a.n[3] << 23;
  w[3] = a.n[3] >> 9 | a.n[4] << 20;
  w[4] = a.n[4] >> 12 | a.n[5] << 17;
  w[5] = a.n[5] >> 15 | a.n[6] << 14;
  w[6] = a.n[6] >> 18 | a.n[7] << 11;
  w[7] = a.n[7] >> 21 | a.n[8] << 8;
}

// magnitude 1 result: the overflow above 2^256 folded in first (2^256 = 2^32 + 977: limb 1 gets x * 8), then one
// carry pass
FE_FN void fe_normalize_weak(fe& a) {
  u32 x = a.n[8] >> 24;
  a.n[8] &= FE_TOP;
  a.n[0] += x * 0x3D1u;
  a.n[1] += x << 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a.n[i + 1] += a.n[i] >> 29;
    a.n[i] &= FE_M;
  }
  FE_HIDE24(a.n[8]);
}
// 1 if a weakly normalised value (it is < 2p; n[8] may carry bit 24) is >= p
FE_FN u32 fe_weak_ge_p(const fe& a) {
  u32 m = a.n[2] & a.n[3] & a.n[4] & a.n[5] & a.n[6] & a.n[7];
  return (a.n[8] >> 24) |
         ((a.n[8] == FE_TOP) & (m == FE_M) & ((a.n[1] + 8u + ((a.n[0] + 0x3D1u) >> 29)) > FE_M));
}
// canonical residue in [0, p)
FE_FN void fe_normalize(fe& a) {
  fe_normalize_weak(a);
  // the value is in [p, 2p) only with probability ~2^-32 for field-like inputs: keep the final subtraction out of
  // the straight-line path (a wave takes the branch only if one of its lanes needs it)
  if (__builtin_expect(fe_weak_ge_p(a) != 0, 0)) {
    a.n[0] += 0x3D1u;
    a.n[1] += 8u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a.n[i + 1] += a.n[i] >> 29;
      a.n[i] &= FE_M;
    }
    a.n[8] &= FE_TOP;  // drops 2^256
    FE_HIDE24(a.n[8]);
  }
}
// parity of the canonical residue without producing it (p is odd: subtracting it flips the parity)
FE_FN u32 fe_parity(fe a) {
  fe_normalize_weak(a);
  return (a.n[0] ^ fe_weak_ge_p(a)) & 1u;
}
// is the value 0 mod p?  (any magnitude <= 7)
FE_FN bool fe_is_zero(fe a) {
  fe_normalize(a);
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) o |= a.n[i];
  return o == 0;
}

FE_FN fe fe_add(const fe& a, const fe& b) {  // magnitude ma + mb
  fe r;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) r.n[i] = a.n[i] + b.n[i];
  return r;
}
// -a for a of magnitude <= m; result magnitude m + 1   ((m+1) p - a, limb-wise, never underflows)
FE_FN fe fe_neg(const fe& a, u32 m) {
  fe r;
  const u32 k = m + 1;
  r.n[0] = FE_P0 * k - a.n[0];
  r.n[1] = FE_P1 * k - a.n[1];
#pragma unroll
  for (int i = 2; i < 8; ++i) r.n[i] = FE_PM * k - a.n[i];
  r.n[8] = FE_P8 * k - a.n[8];
  return r;
}
// a - b for b of magnitude 1; result magnitude ma + 2
FE_FN fe fe_sub(const fe& a, const fe& b) { return fe_add(a, fe_neg(b, 1)); }

// shared tail of fe_mul / fe_sqr: column 8, then the fold of everything above 2^256
FE_FN void fe_mul_tail(fe& r, u64 c, u64 d64, u32 t8) {
  // what is left in d sits at 2^(29*17) = 2^(29*8) * 2^261.  It is small: column 16 = a8*b8 + carry < 7 * 2^48 + 2^29,
  // so d < 2^23 - a 32-bit factor (left 64 bits wide, each product with it costs a second multiply instruction)
  u32 d = (u32)d64;
  FE_HIDE24(d);
  c += (u64)d * FE_R0 + t8;
  r.n[8] = (u32)c & FE_TOP;
  FE_HIDE24(r.n[8]);
  c >>= 24;
  c += (u64)d << 13;  // d * (R1 << 5)
  // c * 2^256 = c * (2^32 + 977): limbs 0 and 1, then a short carry
  u64 e = c * (FE_R0 >> 5) + r.n[0];
  r.n[0] = (u32)e & FE_M;
  e >>= 29;
  e += c * (FE_R1 >> 5) + r.n[1];
  r.n[1] = (u32)e & FE_M;
  e >>= 29;
  // the carry into limb 2 as an opaque 32-bit value: left visible, the backend keeps limb 2 as the untruncated 64-bit
  // sum and multiplies the NEXT product by it as a 64 x 32 bit value (one more v_mad_u64_u32 and two moves per use:
  // 9 uses per multiplication) - the same dropped-truncation family as the FE_HIDE24 bug, harmless only because the
  // carry is tiny
  u32 cy = (u32)e;
  FE_HIDE24(cy);
  r.n[2] += cy;
}

// lib/ecc.c:307-347. Inputs with m1*m2 <= 7, output magnitude 1.
// Columns 8..16 stream through d (their 29-bit digits u are folded down by 2^261 = R1*2^29 + R0), columns 0..7
// stream through c: every column is carried exactly once.
FE_FN fe fe_mul(const fe& a, const fe& b) {
  fe r;
  u64 c = 0, d = 0;
  const u32 R1 = fe_opaque(FE_R1);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_PIN64(d);
    d += (u64)a.n[i] * b.n[8 - i];
  }
  const u32 t8 = (u32)d & FE_M;
  d >>= 29;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int i = k + 1; i < 9; ++i) {
      FE_PIN64(d);
      d += (u64)a.n[i] * b.n[9 + k - i];
    }
    const u32 u = (u32)d & FE_M;
    d >>= 29;
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      FE_PIN64(c);
      c += (u64)a.n[i] * b.n[k - i];
    }
    FE_PIN64(c);
    c += (u64)u * FE_R0;
    r.n[k] = (u32)c & FE_M;
    c >>= 29;
    FE_PIN64(c);
    c += (u64)u * R1;
  }
  fe_mul_tail(r, c, d, t8);
  return r;
}
// lib/ecc.c:349-444. Same schedule with the symmetric products taken once against the doubled operand.
// Input magnitude <= 2 (the doubled limbs must stay below 2^32 and 9 * 4 * 2^58 below 2^64).
FE_FN fe fe_sqr(const fe& a) {
  fe r;
  u32 a2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a2[i] = a.n[i] * 2;
  u64 c = 0, d = 0;
  const u32 R1 = fe_opaque(FE_R1);
  // column k = sum_{i<j, i+j=k} a_i * 2a_j  (+ a_{k/2}^2)
#define FE_SQ_COL(acc, k)                                                    \
  _Pragma("unroll") for (int i = ((k) > 8 ? (k)-8 : 0); 2 * i < (k); ++i) {  \
    FE_PIN64(acc);                                                           \
    acc += (u64)a.n[i] * a2[(k)-i];                                          \
  }                                                                          \
  if (((k)&1) == 0) {                                                        \
    FE_PIN64(acc);                                                           \
    acc += (u64)a.n[(k) / 2] * a.n[(k) / 2];                                 \
  }
  FE_SQ_COL(d, 8)
  const u32 t8 = (u32)d & FE_M;
  d >>= 29;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    FE_SQ_COL(d, 9 + k)
    const u32 u = (u32)d & FE_M;
    d >>= 29;
    FE_SQ_COL(c, k)
    FE_PIN64(c);
    c += (u64)u * FE_R0;
    r.n[k] = (u32)c & FE_M;
    c >>= 29;
    FE_PIN64(c);
    c += (u64)u * R1;
  }
#undef FE_SQ_COL
  fe_mul_tail(r, c, d, t8);
  return r;
}

// Two independent products (two squarings) in one instruction stream.  On gfx950 a v_mad_u64_u32 whose accumulator was written by the
// instruction right before it costs a wait state (the compiler pads with s_nop: 50 per fe_mul, whose two accumulator chains have
// unequal lengths in most columns); other waves of the SIMD fill those slots at four waves per SIMD (fe_mul: 623 SIMD-clocks), not at
// two (713; csrc/tools/femul_bench.hip).  With the accumulations of two multiplications interleaved - four chains - there is always an
// independent multiply-add to issue: same instructions, no padding.  Used where a kernel runs at two waves per SIMD and has the
// independent pairs at hand (`mul`: ec.h, xyzz_madd_lazy).
FE_FN void fe_mul2(fe& r1, fe& r2, const fe& a1, const fe& b1, const fe& a2, const fe& b2) {
  fe o1, o2;  // the results may alias the operands: written at the end
  u64 c1 = 0, d1 = 0, c2 = 0, d2 = 0;
  const u32 R1 = fe_opaque(FE_R1);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    FE_PIN64(d1);
    d1 += (u64)a1.n[i] * b1.n[8 - i];
    FE_PIN64(d2);
    d2 += (u64)a2.n[i] * b2.n[8 - i];
  }
  const u32 t81 = (u32)d1 & FE_M, t82 = (u32)d2 & FE_M;
  d1 >>= 29, d2 >>= 29;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int i = k + 1; i < 9; ++i) {
      FE_PIN64(d1);
      d1 += (u64)a1.n[i] * b1.n[9 + k - i];
      FE_PIN64(d2);
      d2 += (u64)a2.n[i] * b2.n[9 + k - i];
    }
    const u32 u1 = (u32)d1 & FE_M, u2 = (u32)d2 & FE_M;
    d1 >>= 29, d2 >>= 29;
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      FE_PIN64(c1);
      c1 += (u64)a1.n[i] * b1.n[k - i];
      FE_PIN64(c2);
      c2 += (u64)a2.n[i] * b2.n[k - i];
    }
    FE_PIN64(c1);
    c1 += (u64)u1 * FE_R0;
    FE_PIN64(c2);
    c2 += (u64)u2 * FE_R0;
    o1.n[k] = (u32)c1 & FE_M, o2.n[k] = (u32)c2 & FE_M;
    c1 >>= 29, c2 >>= 29;
    FE_PIN64(c1);
    c1 += (u64)u1 * R1;
    FE_PIN64(c2);
    c2 += (u64)u2 * R1;
  }
  fe_mul_tail(o1, c1, d1, t81);
  fe_mul_tail(o2, c2, d2, t82);
  r1 = o1, r2 = o2;
}
FE_FN void fe_sqr2(fe& r1, fe& r2, const fe& a1, const fe& a2) {
  u32 x1[9], x2[9];  // doubled operands
#pragma unroll
  for (int i = 0; i < 9; ++i) x1[i] = a1.n[i] * 2, x2[i] = a2.n[i] * 2;
  fe o1, o2;  // the results may alias the operands: written at the end
  u64 c1 = 0, d1 = 0, c2 = 0, d2 = 0;
  const u32 R1 = fe_opaque(FE_R1);
#define FE_SQ2_COL(acc1, acc2, k)                                            \
  _Pragma("unroll") for (int i = ((k) > 8 ? (k)-8 : 0); 2 * i < (k); ++i) {  \
    FE_PIN64(acc1);                                                          \
    acc1 += (u64)a1.n[i] * x1[(k)-i];                                        \
    FE_PIN64(acc2);                                                          \
    acc2 += (u64)a2.n[i] * x2[(k)-i];                                        \
  }                                                                          \
  if (((k)&1) == 0) {                                                        \
    FE_PIN64(acc1);                                                          \
    acc1 += (u64)a1.n[(k) / 2] * a1.n[(k) / 2];                              \
    FE_PIN64(acc2);                                                          \
    acc2 += (u64)a2.n[(k) / 2] * a2.n[(k) / 2];                              \
  }
  FE_SQ2_COL(d1, d2, 8)
  const u32 t81 = (u32)d1 & FE_M, t82 = (u32)d2 & FE_M;
  d1 >>= 29, d2 >>= 29;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    FE_SQ2_COL(d1, d2, 9 + k)
    const u32 u1 = (u32)d1 & FE_M, u2 = (u32)d2 & FE_M;
    d1 >>= 29, d2 >>= 29;
    FE_SQ2_COL(c1, c2, k)
    FE_PIN64(c1);
    c1 += (u64)u1 * FE_R0;
    FE_PIN64(c2);
    c2 += (u64)u2 * FE_R0;
    o1.n[k] = (u32)c1 & FE_M, o2.n[k] = (u32)c2 & FE_M;
    c1 >>= 29, c2 >>= 29;
    FE_PIN64(c1);
    c1 += (u64)u1 * R1;
    FE_PIN64(c2);
    c2 += (u64)u2 * R1;
  }
#undef FE_SQ2_COL
  fe_mul_tail(o1, c1, d1, t81);
  fe_mul_tail(o2, c2, d2, t82);
  r1 = o1, r2 = o2;
}

__host__ __device__ __noinline__ inline fe fe_sqr_n(fe a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = fe_sqr(a);
  return a;
}
// a^(p-2): the 255 S + 15 M addition chain of lib/ecc.c:463-520 (x2,x3,x6,x9,x11,x22,x44,x88,x176,x220,x223).
// Input magnitude <= 2, output magnitude 1.
__host__ __device__ __noinline__ inline fe fe_inv_fermat(const fe& a) {
  fe x2 = fe_mul(fe_sqr(a), a);
  fe x3 = fe_mul(fe_sqr(x2), a);
  fe x6 = fe_mul(fe_sqr_n(x3, 3), x3);
  fe x9 = fe_mul(fe_sqr_n(x6, 3), x3);
  fe x11 = fe_mul(fe_sqr_n(x9, 2), x2);
  fe x22 = fe_mul(fe_sqr_n(x11, 11), x11);
  fe x44 = fe_mul(fe_sqr_n(x22, 22), x22);
  fe x88 = fe_mul(fe_sqr_n(x44, 44), x44);
  fe x176 = fe_mul(fe_sqr_n(x88, 88), x88);
  fe x220 = fe_mul(fe_sqr_n(x176, 44), x44);
  fe x223 = fe_mul(fe_sqr_n(x220, 3), x3);
  fe t = fe_mul(fe_sqr_n(x223, 23), x22);
  t = fe_mul(fe_sqr_n(t, 5), a);
  t = fe_mul(fe_sqr_n(t, 3), x2);
  return fe_mul(fe_sqr_n(t, 2), a);
}

// ---- the same inverse without the 255 squarings (round 4): Bernstein-Yang division steps ("safegcd", eprint 2019/266) ----------------
// fe_modp_inv's value (lib/ecc.c:463-520) by another algorithm: a GCD-style iteration on (f, g) = (p, a) whose every step looks at
// one bit of g, so 30 steps at a time run on the low words alone and are applied to the full-width values as one 2 x 2 matrix.
//   divstep (half-delta form, zeta = -(delta + 1/2), starts at -1):
//     g odd and zeta < 0:  (zeta, f, g) <- (-zeta - 2, g, (g - f) / 2)
//     otherwise:           (zeta, f, g) <- (zeta - 1,  f, (g + (g odd ? f : 0)) / 2)
//   590 steps bring g to 0 for any 256-bit input (the published bound for this form); 20 rounds x 30 = 600 are done, every lane
//   the same number: no divergence, no data-dependent branch.  d, e follow along with d a = f, e a = g (mod p), kept in (-2p, p);
//   at the end f = +-1 and the inverse is +-d.
// Cost on gfx950: ~720 one-clock-class VALU ops (and / xor / add / shift) per round for the steps + ~200 (72 v_mad_i64_i32) for
// the two matrix applications = ~18 500 instructions against ~34 200 (of which 23 000 v_mad_u64_u32) for the addition chain.
// Working form: 9 signed limbs of 30 bits (the matrix entries are <= 2^30 in magnitude: products and their sums fit 64 bits).
typedef int32_t i32;
typedef int64_t i64;
struct ds30 {
  i32 v[9];  // value = sum v[i] 2^(30 i); v[0..7] in [0, 2^30) between rounds, v[8] carries the sign
};
struct ds_mat {
  i32 u, v, q, r;  // 2^30 (f', g') = (u f + v g, q f + r g)
};
#define DS_M30 0x3FFFFFFF
#define DS_PINV30 0x2DDACACFu /* p^-1 mod 2^30 */
// 30 division steps on the low words; returns the new zeta.  The matrix rows follow the row operations of (f, g): (u, v) with f,
// (q, r) with g, doubled instead of halved.  Ten steps at a time both entries of a row share one register (u + v 2^16: every
// operation of a step is linear, and |u| + |v| <= 2^10 leaves the fields apart), the three 10-step matrices are multiplied out in
// 32 bits: 17 instructions per step + 42 per round instead of 24 per step.
#ifndef ECL_DS_PACKED
#define ECL_DS_PACKED 1 /* A/B: 0 = the four matrix entries in registers of their own, 30 steps in one go */
#endif
FE_FN i32 ds_divsteps30(i32 zeta, u32 f, u32 g, ds_mat& t) {
#if ECL_DS_PACKED
  i32 mu = 1, mv = 0, mq = 0, mr = 1;  // matrix of the steps done so far in this round
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    u32 uv = 1u, qr = 1u << 16;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      u32 c1 = (u32)(zeta >> 31);    // all ones: zeta < 0
      const u32 c2 = 0u - (g & 1u);  // all ones: g odd
      const u32 x = (f ^ c1) - c1, y = (uv ^ c1) - c1;  // -f, -(u, v) where zeta < 0
      g += x & c2, qr += y & c2;
      c1 &= c2;                      // swap
      zeta = (i32)((u32)zeta ^ c1) - 1;
      f += g & c1, uv += qr & c1;
      g >>= 1, uv <<= 1;
    }
    const i32 u = (i32)(uv << 16) >> 16, v = (i32)(uv - (u32)u) >> 16;
    const i32 q = (i32)(qr << 16) >> 16, r = (i32)(qr - (u32)q) >> 16;
    if (part == 0) mu = u, mv = v, mq = q, mr = r;
    else {
      const i32 nu = u * mu + v * mq, nv = u * mv + v * mr, nq = q * mu + r * mq, nr = q * mv + r * mr;
      mu = nu, mv = nv, mq = nq, mr = nr;
    }
  }
  t.u = mu, t.v = mv, t.q = mq, t.r = mr;
  return zeta;
#else
  u32 u = 1, v = 0, q = 0, r = 1;
#pragma unroll
  for (int i = 0; i < 30; ++i) {
    u32 c1 = (u32)(zeta >> 31);    // all ones: zeta < 0
    const u32 c2 = 0u - (g & 1u);  // all ones: g odd
    const u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // -f, -u, -v where zeta < 0
    g += x & c2, q += y & c2, r += z & c2;
    c1 &= c2;                      // swap
    zeta = (i32)((u32)zeta ^ c1) - 1;
    f += g & c1, u += q & c1, v += r & c1;
    g >>= 1, u <<= 1, v <<= 1;
  }
  t.u = (i32)u, t.v = (i32)v, t.q = (i32)q, t.r = (i32)r;
  return zeta;
#endif
}
// (f, g) <- t (f, g) / 2^30 (exact)
FE_FN void ds_update_fg(ds30& f, ds30& g, const ds_mat& t) {
  i64 cf = (i64)t.u * f.v[0] + (i64)t.v * g.v[0];
  i64 cg = (i64)t.q * f.v[0] + (i64)t.r * g.v[0];
  cf >>= 30, cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cf += (i64)t.u * f.v[i] + (i64)t.v * g.v[i];
    cg += (i64)t.q * f.v[i] + (i64)t.r * g.v[i];
    f.v[i - 1] = (i32)cf & DS_M30, g.v[i - 1] = (i32)cg & DS_M30;
    FE_HIDE24(f.v[i - 1]);
    FE_HIDE24(g.v[i - 1]);  // known non-negative, the products become unsigned x signed: three instructions instead of one v_mad_i64_i32
    cf >>= 30, cg >>= 30;
  }
  f.v[8] = (i32)cf, g.v[8] = (i32)cg;
}
// (d, e) <- t (d, e) / 2^30 (mod p): a multiple of p = 2^256 - 2^32 - 977 (signed limbs -977, -4, 0, ..., 0, 2^16) makes each sum
// divisible; negative d / e are taken as d + p, e + p first, which keeps the results in (-2p, p)
FE_FN void ds_update_de(ds30& d, ds30& e, const ds_mat& t) {
  const i32 sd = d.v[8] >> 31, se = e.v[8] >> 31;
  i32 md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  i64 cd = (i64)t.u * d.v[0] + (i64)t.v * e.v[0];
  i64 ce = (i64)t.q * d.v[0] + (i64)t.r * e.v[0];
  md -= (i32)((DS_PINV30 * (u32)cd + (u32)md) & DS_M30);
  me -= (i32)((DS_PINV30 * (u32)ce + (u32)me) & DS_M30);
  cd -= (i64)md * 977, ce -= (i64)me * 977;
  cd >>= 30, ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    cd += (i64)t.u * d.v[i] + (i64)t.v * e.v[i];
    ce += (i64)t.q * d.v[i] + (i64)t.r * e.v[i];
    if (i == 1) cd -= (i64)md * 4, ce -= (i64)me * 4;
    if (i == 8) cd += (i64)md * 65536, ce += (i64)me * 65536;
    d.v[i - 1] = (i32)cd & DS_M30, e.v[i - 1] = (i32)ce & DS_M30;
    FE_HIDE24(d.v[i - 1]);
    FE_HIDE24(e.v[i - 1]);
    cd >>= 30, ce >>= 30;
  }
  d.v[8] = (i32)cd, e.v[8] = (i32)ce;
}
// any magnitude <= 7 in, canonical residue out; 0 -> 0 like the addition chain
__host__ __device__ __noinline__ inline fe fe_inv_divsteps(fe a) {
  fe_normalize(a);
  ds30 f, g, d, e;
  f.v[0] = 0x3FFFFC2F, f.v[1] = 0x3FFFFFFB, f.v[8] = 0xFFFF;
#pragma unroll
  for (int i = 2; i < 8; ++i) f.v[i] = DS_M30;
#pragma unroll
  for (int i = 0; i < 8; ++i) g.v[i] = (i32)((a.n[i] >> i | a.n[i + 1] << (29 - i)) & DS_M30);
  g.v[8] = (i32)(a.n[8] >> 8);
#pragma unroll
  for (int i = 0; i < 9; ++i) d.v[i] = 0, e.v[i] = 0;
  e.v[0] = 1;
  i32 zeta = -1;
#pragma unroll 1
  for (int round = 0; round < 20; ++round) {
    ds_mat t;
    zeta = ds_divsteps30(zeta, (u32)f.v[0], (u32)g.v[0], t);
    ds_update_de(d, e, t);
    ds_update_fg(f, g, t);
  }
  // g = 0, f = +-1 (f = p, d = 0 if a was 0): the inverse is d * sign(f), brought from (-2p, p) to [0, p)
  const i32 neg = f.v[8] >> 31;
  i32 m = d.v[8] >> 31;
  d.v[0] += -977 & m, d.v[1] += -4 & m, d.v[8] += 65536 & m;
#pragma unroll
  for (int i = 0; i < 9; ++i) d.v[i] = (d.v[i] ^ neg) - neg;
#pragma unroll
  for (int i = 0; i < 8; ++i) d.v[i + 1] += d.v[i] >> 30, d.v[i] &= DS_M30;
  m = d.v[8] >> 31;
  d.v[0] += -977 & m, d.v[1] += -4 & m, d.v[8] += 65536 & m;
#pragma unroll
  for (int i = 0; i < 8; ++i) d.v[i + 1] += d.v[i] >> 30, d.v[i] &= DS_M30;
  fe r;
  r.n[0] = (u32)d.v[0] & FE_M;
#pragma unroll
  for (int i = 1; i < 9; ++i) r.n[i] = ((u32)d.v[i - 1] >> (30 - i) | (u32)d.v[i] << i) & FE_M;
  FE_HIDE24(r.n[8]);
  return r;
}
#ifndef ECL_FE_INV_DIVSTEPS
#define ECL_FE_INV_DIVSTEPS 1 /* A/B: 0 = the addition chain everywhere */
#endif
FE_FN fe fe_inv(const fe& a) {
#if ECL_FE_INV_DIVSTEPS
  return fe_inv_divsteps(a);
#else
  return fe_inv_fermat(a);
#endif
}

// secp256k1 constants as canonical little-endian u32 words
#define FE_BETA1_W                                                                                              \
  { 0x719501eeu, 0xc1396c28u, 0x12f58995u, 0x9cf04975u, 0xac3434e9u, 0x6e64479eu, 0x657c0710u, 0x7ae96a2bu }
#define FE_GX_W                                                                                                 \
  { 0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu }
#define FE_GY_W                                                                                                 \
  { 0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u, 0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u }
