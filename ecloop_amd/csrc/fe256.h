// fe256.h — secp256k1 base-field arithmetic for gfx950 (device only).
//
// Replaces the reference's fe_modp_* family (lib/ecc.c:269-540) on the device.  Representation: 8 x u32
// little-endian limbs held in VGPRs (one lane = one field element; the reference's 4 x u64 limbs are
// the same bits).  p = 2^256 - K with K = 2^32 + 977, so a 512-bit product folds as lo + hi*977 + (hi<<32).
//
// Contract (what makes the results bit-identical to the reference): every function returns the
// CANONICAL residue in [0, p) for canonical inputs.  The reference hashes only canonical values
// (DESIGN.md "Canonical form"), so any internally different-but-equivalent schedule gives the same bytes.
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

struct fe {
  u32 v[8];
};

#define FE_FN __host__ __device__ __forceinline__

// p = FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFF FFFFFFFE FFFFFC2F
#define FE_P0 0xFFFFFC2Fu
#define FE_P1 0xFFFFFFFEu
#define FE_K0 977u /* K = 2^32 + 977 */

FE_FN fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
FE_FN fe fe_one() {
  fe r = fe_zero();
  r.v[0] = 1;
  return r;
}
FE_FN bool fe_is_zero(const fe& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i];
  return o == 0;
}
FE_FN bool fe_eq(const fe& a, const fe& b) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = a + b over 2^256, returns carry (0/1)
FE_FN u32 fe_add_raw(fe& r, const fe& a, const fe& b) {
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (u64)a.v[i] + b.v[i];
    r.v[i] = (u32)c;
    c >>= 32;
  }
  return (u32)c;
}
// r = a - b over 2^256, returns borrow (0/1)
FE_FN u32 fe_sub_raw(fe& r, const fe& a, const fe& b) {
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 d = (u64)a.v[i] - b.v[i] - br;
    r.v[i] = (u32)d;
    br = (u32)(d >> 63);
  }
  return br;
}
// r += K*m (m = 0 or 1), i.e. subtract p modulo 2^256; returns carry
FE_FN u32 fe_add_k(fe& r, u32 m) {
  u64 c = (u64)r.v[0] + (FE_K0 & (0u - m));
  r.v[0] = (u32)c;
  c = (c >> 32) + r.v[1] + m;
  r.v[1] = (u32)c;
#pragma unroll
  for (int i = 2; i < 8; ++i) {
    c = (c >> 32) + r.v[i];
    r.v[i] = (u32)c;
  }
  return (u32)(c >> 32);
}
// r -= K*m (m = 0 or 1), i.e. add p modulo 2^256
FE_FN void fe_sub_k(fe& r, u32 m) {
  u64 d = (u64)r.v[0] - (FE_K0 & (0u - m));
  r.v[0] = (u32)d;
  u32 br = (u32)(d >> 63);
  d = (u64)r.v[1] - m - br;
  r.v[1] = (u32)d;
  br = (u32)(d >> 63);
#pragma unroll
  for (int i = 2; i < 8; ++i) {
    d = (u64)r.v[i] - br;
    r.v[i] = (u32)d;
    br = (u32)(d >> 63);
  }
}
// a >= p  (a < 2^256)
FE_FN bool fe_ge_p(const fe& a) {
  u32 hi = a.v[2] & a.v[3] & a.v[4] & a.v[5] & a.v[6] & a.v[7];
  return hi == 0xFFFFFFFFu && (a.v[1] == 0xFFFFFFFFu || (a.v[1] == FE_P1 && a.v[0] >= FE_P0));
}
// canonicalise a value in [0, 2^256)
FE_FN void fe_canon(fe& a) { fe_add_k(a, fe_ge_p(a) ? 1u : 0u); }

// a - b mod p, canonical for canonical inputs (lib/ecc.c:277-290)
FE_FN fe fe_sub(const fe& a, const fe& b) {
  fe r;
  u32 br = fe_sub_raw(r, a, b);
  fe_sub_k(r, br);
  return r;
}
// a + b mod p, canonical for canonical inputs (the reference's add, lib/ecc.c:292-305, only reduces on
// 2^256 overflow; the device version is used where the result must be canonical)
FE_FN fe fe_add(const fe& a, const fe& b) {
  fe r;
  u32 c = fe_add_raw(r, a, b);
  fe_add_k(r, c);  // wrapped: r + 2^256 == r + K (mod p); cannot carry again for canonical inputs
  fe_canon(r);
  return r;
}
// -a mod p for canonical nonzero a; neg(0) = 0 here (the reference returns p; never hashed)
FE_FN fe fe_neg(const fe& a) {
  fe z = fe_zero();
  return fe_sub(z, a);
}

// 512-bit product, operand scanning: t[i+j] += a[j]*b[i]
FE_FN void fe_mul_wide(u32 t[16], const fe& a, const fe& b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u32 carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      u64 acc = (u64)a.v[j] * b.v[i] + t[i + j] + carry;
      t[i + j] = (u32)acc;
      carry = (u32)(acc >> 32);
    }
    t[i + 8] = carry;
  }
}
// 512-bit square: off-diagonal products once, doubled, plus the diagonal
FE_FN void fe_sqr_wide(u32 t[16], const fe& a) {
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    u32 carry = 0;
#pragma unroll
    for (int j = i + 1; j < 8; ++j) {
      u64 acc = (u64)a.v[j] * a.v[i] + t[i + j] + carry;
      t[i + j] = (u32)acc;
      carry = (u32)(acc >> 32);
    }
    t[i + 8] = carry;
  }
  // double
  u32 top = 0;
#pragma unroll
  for (int i = 1; i < 16; ++i) {
    u32 nt = t[i] >> 31;
    t[i] = (t[i] << 1) | top;
    top = nt;
  }
  // add diagonal squares
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 sq = (u64)a.v[i] * a.v[i];
    c += (u64)t[2 * i] + (u32)sq;
    t[2 * i] = (u32)c;
    c >>= 32;
    c += (u64)t[2 * i + 1] + (u32)(sq >> 32);
    t[2 * i + 1] = (u32)c;
    c >>= 32;
  }
}
// fold a 512-bit value to the canonical residue: lo + hi*977 + (hi << 32), then the 9th word once more
FE_FN fe fe_reduce_wide(const u32 t[16]) {
  fe r;
  u64 acc = 0;
  u32 prev = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc += (u64)t[8 + i] * FE_K0 + t[i] + prev;
    r.v[i] = (u32)acc;
    acc >>= 32;
    prev = t[8 + i];
  }
  acc += prev;  // 9th word: < 2^34
  // second fold: acc * K = acc*977 + (acc << 32)
  u64 f0 = acc * FE_K0;  // < 2^44
  u64 c = (u64)r.v[0] + (u32)f0;
  r.v[0] = (u32)c;
  c = (c >> 32) + r.v[1] + (f0 >> 32) + (u32)acc;
  r.v[1] = (u32)c;
  c = (c >> 32) + r.v[2] + (acc >> 32);
  r.v[2] = (u32)c;
#pragma unroll
  for (int i = 3; i < 8; ++i) {
    c = (c >> 32) + r.v[i];
    r.v[i] = (u32)c;
  }
  // a carry out of 2^256 here means the true value is r + 2^256 == r + K (mod p); r is tiny then
  fe_add_k(r, (u32)(c >> 32));
  fe_canon(r);
  return r;
}
// lib/ecc.c:307-347
FE_FN fe fe_mul(const fe& a, const fe& b) {
  u32 t[16];
  fe_mul_wide(t, a, b);
  return fe_reduce_wide(t);
}
// lib/ecc.c:349-444
FE_FN fe fe_sqr(const fe& a) {
  u32 t[16];
  fe_sqr_wide(t, a);
  return fe_reduce_wide(t);
}

__host__ __device__ __noinline__ inline fe fe_sqr_n(fe a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = fe_sqr(a);
  return a;
}
// a^(p-2): the 255 S + 15 M addition chain of lib/ecc.c:463-520 (x2,x3,x6,x9,x11,x22,x44,x88,x176,x220,x223)
__host__ __device__ __noinline__ inline fe fe_inv(const fe& a) {
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

// secp256k1 constants as limb initialisers (little-endian u32 words)
#define FE_BETA1                                                                                                \
  { 0x719501eeu, 0xc1396c28u, 0x12f58995u, 0x9cf04975u, 0xac3434e9u, 0x6e64479eu, 0x657c0710u, 0x7ae96a2bu }
#define FE_GX                                                                                                   \
  { 0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu }
#define FE_GY                                                                                                   \
  { 0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u, 0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u }
