// add_kernel.h — the `add` hot loop on gfx950: batch affine point addition + hash160 + bloom probe, fused.
//
// Replaces batch_add + check_found_add of the reference (main.c:287-403) together with everything they call
// (fe_modp_grpinv lib/ecc.c:522-540, addr33/65_batch lib/addr.c:99-131, blf_has lib/utils.c:308-326).
//
// Same mathematics as the reference: a group of 2B consecutive keys around a centre point C is produced as
// C, C + G_i (i < B-1) and C - G_i (i < B) with G_i = (i+1)*stride*G taken from a precomputed table, using ONE
// field inversion for all B differences (Montgomery's trick); the next centre comes from one more addition.
// Re-designed for the GPU:
//   * one lane = one walk (its own centre), all lanes of the grid add the SAME table point in the same
//     iteration, so the table operand is wave-uniform (scalar loads / SGPR operands);
//   * the lane's chain of prefix products lives in HBM in a lane-interleaved layout (16 B per lane per
//     access -> 1 KiB coalesced per wave instruction), B x 32 B per lane: 288 GB of HBM makes a deep chain
//     cheap, so the inversion cost per key (270 mults / 2B keys) is small without any cross-lane exchange;
//   * the step to the next centre is folded into the same batched inversion as chain element 0, so there is
//     no separate full inversion per group (the reference pays one, main.c:400);
//   * walks are interleaved: lane g of T handles groups g, g+T, g+2T, ... so a contiguous follow-up call
//     continues from the centres left in HBM without any new scalar multiplication;
//   * hashing and the bloom probe run in the same kernel on the just-computed (x, y): points never go to HBM.
// Key order inside a group follows the reference (main.c:388-391): offset 0..B-1 = C - G_{B-1-j},
// offset B = C, offset B+1+i = C + G_i.
#pragma once
#include "bloom.h"
#include "hash160.h"

struct ecl_found_dev {
  u64 key_offset;
  u32 h160[5];
  u32 tag;  // byte 0 = endo, byte 1 = compressed
};

struct add_args {
  const u32* __restrict__ tab;  // [B][ECL_TAB_STRIDE]: limbs x[9], y[9] of (i+1)*stride*G (normalised 9x29), 2 words pad
  u32 jump[16];                 // x[8], y[8] of (T*2B*stride)*G
  uint4* __restrict__ cxy;      // lane centres as canonical words, planes {x.lo, x.hi, y.lo, y.hi} x T
  uint4* __restrict__ scratch;  // prefix products (9x29 limbs 0..7), [(k*2 + half) * T + lane]
  u32* __restrict__ scratch2;   // prefix products (limb 8), [k * T + lane]
  bloom_t bloom;
  ecl_found_dev* found;
  u32* counter;
  u32 cap;
  u32 B;      // table points per group (group = 2B keys)
  u32 T;      // lanes
  u32 nb;     // groups per lane in this launch
  u64 nkeys;  // keys with offset >= nkeys are not tested
};

// canonical words in two uint4 planes <-> fe
FE_FN fe fe_ld_words2(const uint4* p, size_t stride) {
  uint4 lo = p[0], hi = p[stride];
  const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return fe_from_words(w);
}
FE_FN void fe_st_words2(uint4* p, size_t stride, fe a) {  // normalises
  fe_normalize(a);
  u32 w[8];
  fe_to_words(w, a);
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[stride] = make_uint4(w[4], w[5], w[6], w[7]);
}
// raw limbs (any magnitude) in planes uint4, uint4, u32.  These are the prefix-product chain's accesses: every element is
// written once and read once, a whole group later - a stream with no reuse in any cache.  ECL_CHAIN_NT (A/B builds) marks
// the loads (bit 0) and / or the stores (bit 1) non-temporal (`nt`: the L1 is bypassed, the L2 / Infinity Cache line is
// first in line for eviction), so that the stream does not displace the bloom filter's lines.
#ifndef ECL_CHAIN_NT
#define ECL_CHAIN_NT 0
#endif
#if defined(__HIPCC__)
typedef u32 ecl_v4u __attribute__((ext_vector_type(4)));
#endif
FE_FN fe fe_ld_limbs(const uint4* p4, size_t stride4, const u32* p1) {
  fe r;
#if defined(__HIPCC__) && (ECL_CHAIN_NT & 1)
  const ecl_v4u a = __builtin_nontemporal_load((const ecl_v4u*)p4), b = __builtin_nontemporal_load((const ecl_v4u*)(p4 + stride4));
  r.n[8] = __builtin_nontemporal_load(p1);
#else
  const uint4 a = p4[0], b = p4[stride4];
  r.n[8] = p1[0];
#endif
  r.n[0] = a.x, r.n[1] = a.y, r.n[2] = a.z, r.n[3] = a.w;
  r.n[4] = b.x, r.n[5] = b.y, r.n[6] = b.z, r.n[7] = b.w;
  return r;
}
FE_FN void fe_st_limbs(uint4* p4, size_t stride4, u32* p1, const fe& a) {
#if defined(__HIPCC__) && (ECL_CHAIN_NT & 2)
  const ecl_v4u lo = {a.n[0], a.n[1], a.n[2], a.n[3]}, hi = {a.n[4], a.n[5], a.n[6], a.n[7]};
  __builtin_nontemporal_store(lo, (ecl_v4u*)p4);
  __builtin_nontemporal_store(hi, (ecl_v4u*)(p4 + stride4));
  __builtin_nontemporal_store(a.n[8], p1);
#else
  p4[0] = make_uint4(a.n[0], a.n[1], a.n[2], a.n[3]);
  p4[stride4] = make_uint4(a.n[4], a.n[5], a.n[6], a.n[7]);
  p1[0] = a.n[8];
#endif
}
// Table entries are read through the constant address space: the address is wave-uniform (kernel argument + loop
// counter), so these become scalar loads (s_load_dwordx*) into SGPRs and everything computed from them alone
// (negation, X-independent sums) runs on the scalar unit; the limbs are stored ready-made so there is nothing to
// convert.  The table is written by an earlier kernel and never by this one.
#define ECL_TAB_STRIDE 20u
typedef const __attribute__((address_space(4))) u32* ctab_ptr;
__device__ __forceinline__ fe fe_ld_tab(ctab_ptr p) {
  fe r;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) r.n[i] = p[i];
  return r;
}
// 8 canonical words at p (argument data) -> fe
FE_FN fe fe_ldw(const u32* p) {
  u32 w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = p[i];
  return fe_from_words(w);
}

__device__ __forceinline__ void found_push(const add_args& a, u64 off, const u32 h[5], u32 endo, u32 compressed) {
  u32 idx = atomicAdd(a.counter, 1u);
  if (idx < a.cap) {
    ecl_found_dev r;
    r.key_offset = off;
#pragma unroll
    for (int i = 0; i < 5; ++i) r.h160[i] = h[i];
    r.tag = endo | (compressed << 8);
    a.found[idx] = r;
  }
}

// ---- staged filter test ---------------------------------------------------------------------------------------
// Probe 0 (bloom.h) runs on every hash.  Its survivors (37 % at the .blf design density) are not finished on the
// spot - the live lanes would drag all 64 lanes of the wave through up to 19 more dependent probes (measured: 6 % of
// the kernel) - but parked in per-wave rings in LDS and finished 64 at a time (cand_queues below).  Wave-private:
// no barrier, no atomics; records: key offset, hash160, tag.
#define ECL_Q_SLOTS 128u
struct cand_queue {
  u32* mem;    // this wave's slice of LDS: 8 fields x ECL_Q_SLOTS words, field-major (conflict-free for lane-contiguous slots)
  u32 head;    // wave-uniform
  u32 count;   // wave-uniform, < 64 between calls
};

#ifndef ECL_TWO_LEVEL_QUEUE
#define ECL_TWO_LEVEL_QUEUE 1
#endif
struct cand_rec {
  u64 off;
  u32 h[5], tag;
};
__device__ __forceinline__ cand_rec cand_load(const cand_queue& q, u32 slot) {
  cand_rec r;
  r.off = (u64)q.mem[slot] | (u64)q.mem[ECL_Q_SLOTS + slot] << 32;
#pragma unroll
  for (int i = 0; i < 5; ++i) r.h[i] = q.mem[(2 + i) * ECL_Q_SLOTS + slot];
  r.tag = q.mem[7 * ECL_Q_SLOTS + slot];
  return r;
}
// append the records of the lanes with `pass` (ballot + mbcnt compaction); returns true when 64 or more are waiting.
// Must be reached by ALL lanes of the wave together (head / count are wave-uniform state).
__device__ __forceinline__ bool cand_append(cand_queue& q, bool pass, u64 off, const u32 h[5], u32 tag) {
  const u64 m = __builtin_amdgcn_ballot_w64(pass);
  if (m == 0) return false;
  if (pass) {
    const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
    const u32 slot = (q.head + q.count + below) & (ECL_Q_SLOTS - 1);
    q.mem[slot] = (u32)off;
    q.mem[ECL_Q_SLOTS + slot] = (u32)(off >> 32);
#pragma unroll
    for (int i = 0; i < 5; ++i) q.mem[(2 + i) * ECL_Q_SLOTS + slot] = h[i];
    q.mem[7 * ECL_Q_SLOTS + slot] = tag;
  }
  q.count += (u32)__builtin_popcountll(m);
  return q.count >= 64;
}
// take up to 64 records off the ring: lane i gets record i (valid for i < n)
__device__ __forceinline__ cand_rec cand_take(cand_queue& q, bool& valid) {
  const u32 lane = threadIdx.x & 63u;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const u32 n = q.count < 64 ? q.count : 64;
  valid = lane < n;
  cand_rec r = cand_load(q, (q.head + lane) & (ECL_Q_SLOTS - 1));
  q.head = (q.head + n) & (ECL_Q_SLOTS - 1);
  q.count -= n;
  return r;
}
// the two rings of a wave: A = survivors of probe 0 (37 % of all hashes at the .blf design density), B = survivors of
// the middle stage (probes 1-2: 14 % of A; one probe for multi-GB filters: 37 %).  A is drained 64 at a time without
// a loop; only B runs the remaining probes with the early-out loop, where the wave iterates until its slowest lane
// is done.  Per hash and wave: 68 VALU instructions for the whole filter test instead of 96 with one ring.
struct cand_queues {
  cand_queue a, b;
};
__device__ __forceinline__ void cand_finish(const add_args& a, cand_queue& qb) {  // up to 64 records of ring B
  bool valid;
  const cand_rec r = cand_take(qb, valid);
  const int from = ECL_TWO_LEVEL_QUEUE ? ECL_STAGE1_PROBES + (bloom_mid_two(a.bloom) ? 2 : 1) : ECL_STAGE1_PROBES;
  if (valid && bloom_probes_from(a.bloom, r.h, from))
    found_push(a, r.off, r.h, r.tag & 0xff, (r.tag >> 8) & 1);
}
__device__ __forceinline__ void cand_mid(const add_args& a, cand_queues& q) {  // up to 64 records of ring A -> ring B
  bool valid;
  const cand_rec r = cand_take(q.a, valid);
  const bool pass = valid && bloom_mid(a.bloom, r.h, bloom_mid_two(a.bloom));
  if (cand_append(q.b, pass, r.off, r.h, r.tag)) cand_finish(a, q.b);
}
__device__ __forceinline__ void cand_push(const add_args& a, cand_queues* q, bool pass, u64 off, const u32 h[5], u32 tag) {
#if ECL_TWO_LEVEL_QUEUE
  if (cand_append(q->a, pass, off, h, tag)) cand_mid(a, *q);
#else
  if (cand_append(q->b, pass, off, h, tag)) cand_finish(a, q->b);
#endif
}
__device__ __forceinline__ void cand_flush(const add_args& a, cand_queues& q) {  // end of the kernel: the remainders
#if ECL_TWO_LEVEL_QUEUE
  cand_mid(a, q);  // A holds < 64
#endif
  cand_finish(a, q.b);  // B holds < 128: at most two rounds
  cand_finish(a, q.b);
}
// Filter test of one hash; q == nullptr: no queue (`mul` kernel), everything in place.  With a queue the call must
// be reached by ALL lanes of the wave together: lanes whose key is outside the range come along with live = false.
// (Deferring the stage-1 test by one hash - loads in flight under the next hash160 - was measured: no gain, the
// other waves of the SIMD already cover the probe latency.)
__device__ __forceinline__ void filter_check(const add_args& a, cand_queues* q, bool live, u64 off, const u32 h[5], u32 endo,
                                             u32 compressed) {
  const bool pass = live && bloom_stage1(a.bloom, h);
  if (!q) {
    if (pass && bloom_stage2(a.bloom, h)) found_push(a, off, h, endo, compressed);
    return;
  }
  cand_push(a, q, pass, off, h, endo | (compressed << 8));
}

// hash every selected encoding / endomorphism image of the affine point (x, y) and probe the filter
// (check_found_add, main.c:287-347; endo images (x,-y) (bx,y) (bx,-y) (b2x,y) (b2x,-y), main.c:314-327).
// x: magnitude <= 4, y: magnitude <= 3.
template <bool A33, bool A65, bool ENDO>
__device__ __forceinline__ void check_point(const add_args& a, cand_queues* q, bool live, fe x, fe y, u64 off) {
  u32 xw[3][8], yw[2][8], par = 0;
  if (ENDO) {
    const u32 bw[8] = FE_BETA1_W;
    fe bx = fe_mul(x, fe_from_words(bw));  // magnitude 1
    fe b2x = fe_neg(fe_add(x, bx), 5);     // beta^2 = -1 - beta; magnitude 6
    fe_normalize(bx);
    fe_normalize(b2x);
    fe_to_words(xw[1], bx);
    fe_to_words(xw[2], b2x);
  }
  fe_normalize(x);
  fe_to_words(xw[0], x);
  if (A65) {
    fe_normalize(y);
    fe_to_words(yw[0], y);
    par = y.n[0] & 1u;
    if (ENDO) {
      fe ny = fe_neg(y, 1);  // y != 0 on the curve, so this is p - y after normalisation
      fe_normalize(ny);
      fe_to_words(yw[1], ny);
    }
  } else {
    par = fe_parity(y);
  }
  const int nvar = ENDO ? 6 : 1;
#pragma unroll 1
  for (int e = 0; e < nvar; ++e) {
    u32 xs[8], h[5];
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[i] = ENDO ? (e < 2 ? xw[0][i] : (e < 4 ? xw[1][i] : xw[2][i])) : xw[0][i];
    if (A33) {
      hash160_33(h, xs, (par ^ (u32)e) & 1u);  // parity(-y) = !parity(y): p is odd, y != 0
      filter_check(a, q, live, off, h, e, 1);
    }
    if (A65) {
      u32 ys[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) ys[i] = (ENDO && (e & 1)) ? yw[1][i] : yw[0][i];
      hash160_65(h, xs, ys);
      filter_check(a, q, live, off, h, e, 0);
    }
  }
}

// waves per SIMD the register allocator must leave room for (256-thread blocks: blocks per CU = this value)
// 1: load the next prefix product one iteration ahead (10 more live VGPRs across the hash), 0: load at use
#ifndef ECL_PREFETCH
#define ECL_PREFETCH 0  /* measured early in the round: 0 -> 9.70, 1 -> 9.35 Gkeys/s; with the final kernel both give the same rate */
#endif
#ifndef ECL_ADD_WAVES
#define ECL_ADD_WAVES 4  /* measured early in the round: 2 -> 8.38, 3 -> 8.99, 4 -> 7.97 Gkeys/s (addr33); final kernel: 2 is 2 % slower, 5 is 0.9 % and 6 is 4.8 % slower,
                           4 is 0.2-0.5 % faster than 3 except for -a cu -endo (0.4 % slower: it stays at 3); -a u -endo spills inside its per-point loop at 4
                           (17 scratch instructions per table point), so it takes 3 as well (round 6) */
#endif
// threads per workgroup of the add kernel.  Waves never talk to each other (no barrier, wave-private LDS rings), so the
// only thing the size decides is the granularity at which the dispatcher hands out work: 64 / 128 / 256 measured equal
// within 0.1 % in round 2 (tried and rejected); the lane count of a call is a multiple of 256
#ifndef ECL_ADD_BLOCK
#define ECL_ADD_BLOCK 256
#endif
template <bool A33, bool A65, bool ENDO>
__global__ void __launch_bounds__(ECL_ADD_BLOCK, (A65 && ENDO) ? 3 : ECL_ADD_WAVES) k_add(const add_args a) {
  __shared__ u32 q_mem[ECL_ADD_BLOCK / 64][2][8 * ECL_Q_SLOTS];  // two candidate rings per wave
  cand_queues q;
  q.a.mem = q_mem[threadIdx.x >> 6][0], q.a.head = 0, q.a.count = 0;
  q.b.mem = q_mem[threadIdx.x >> 6][1], q.b.head = 0, q.b.count = 0;
  const u32 g = blockIdx.x * (u32)ECL_ADD_BLOCK + threadIdx.x;
  const u32 T = a.T, B = a.B;
  if (g >= T) return;
  const size_t plane = T;
  // centre (X, Y): canonical in HBM, magnitude 1 in registers
  fe X = fe_ld_words2(a.cxy + g, plane), Y = fe_ld_words2(a.cxy + 2 * (size_t)T + g, plane);
  const fe Jx = fe_ldw(a.jump), Jy = fe_ldw(a.jump + 8);
  const ctab_ptr tab = (ctab_ptr)(uintptr_t)a.tab;
  uint4* scr4 = a.scratch + g;
  u32* scr2 = a.scratch2 + g;
  const size_t s4 = 2 * (size_t)T;  // one chain element = two uint4 planes + one u32 plane

#pragma unroll 1
  for (u32 b = 0; b < a.nb; ++b) {
    const u64 base = ((u64)b * T + g) * (2ull * B);
    // groups only grow: a wave leaves when none of its lanes has keys left (wave-uniform control flow keeps the
    // candidate queue state uniform; the lane count is sized to the range, so idle lanes are rare)
    if (__builtin_amdgcn_ballot_w64(base < a.nkeys) == 0) break;

    // ---- phase 1: prefix products of e_0 = Jx - X, e_k = Gx_{k-1} - X   (differences have magnitude 3)
    fe acc = fe_sub(Jx, X);
    fe_normalize_weak(acc);            // magnitude 1: the chain multiplies it by a magnitude-3 difference
    const bool dbl = fe_is_zero(acc);  // C == J: next centre is 2C (C == -J would be the scalar 0: excluded)
    if (dbl) acc = fe_one();
#pragma unroll 1
    for (u32 k = 1; k <= B; ++k) {
      fe_st_limbs(scr4 + (size_t)(k - 1) * s4, plane, scr2 + (size_t)(k - 1) * plane, acc);
      fe dx = fe_sub(fe_ld_tab(tab + (size_t)(k - 1) * ECL_TAB_STRIDE), X);
      acc = fe_mul(acc, dx);
    }
    // ---- phase 2: one inversion for the whole chain
    fe inv = fe_inv(acc);
    // ---- phase 3: walk the chain backwards, emit C +- G_i
#if ECL_PREFETCH
    fe pre = fe_ld_limbs(scr4 + (size_t)(B - 1) * s4, plane, scr2 + (size_t)(B - 1) * plane);
#endif
#pragma unroll 1
    for (u32 k = B; k >= 1; --k) {
      const u32 i = k - 1;
#if ECL_PREFETCH
      fe nxt = pre;
      if (k >= 2) nxt = fe_ld_limbs(scr4 + (size_t)(k - 2) * s4, plane, scr2 + (size_t)(k - 2) * plane);  // prefetch
#else
      const fe pre = fe_ld_limbs(scr4 + (size_t)i * s4, plane, scr2 + (size_t)i * plane);
#endif
      const fe gx = fe_ld_tab(tab + (size_t)i * ECL_TAB_STRIDE), gy = fe_ld_tab(tab + (size_t)i * ECL_TAB_STRIDE + FE_LIMBS);
      const fe dx = fe_sub(gx, X);
      const fe invk = fe_mul(inv, pre);  // 1 / (Gx_i - X)
      inv = fe_mul(inv, dx);
      const fe nxg = fe_neg(fe_add(X, gx), 2);  // -(X + Gx), magnitude 3
      const int nwhich = (k == 1) ? 3 : 2;
#pragma unroll 1
      for (int which = 0; which < nwhich; ++which) {
        fe px, py;
        u64 off;
        bool valid = true;  // wave-uniform
        if (which < 2) {
          // lambda = (+-Gy - Y) / (Gx - X); x3 = lambda^2 - X - Gx; y3 = lambda (X - x3) - Y   (main.c:379-386)
          // +-Gy - Y, magnitude 3.  The table side (Gy + 2p or 3p - Gy) is wave-uniform like `which`: selected on
          // the scalar unit, so the vector side is one subtraction per limb (written as a select of two vector
          // results the compiler emits both and nine v_cndmask)
          const fe c = which == 0 ? fe_add(gy, fe_neg(fe_zero(), 1)) : fe_neg(gy, 2);
          fe s;
#pragma unroll
          for (int l = 0; l < FE_LIMBS; ++l) s.n[l] = c.n[l] - Y.n[l];
          fe lam = fe_mul(s, invk);
          px = fe_add(fe_sqr(lam), nxg);                                   // magnitude 4
          py = fe_sub(fe_mul(lam, fe_add(X, fe_neg(px, 4))), Y);           // X - px: magnitude 6; py: magnitude 3
          off = base + (which == 0 ? B + 1 + i : B - 1 - i);  // scalar select, one 64-bit add
          valid = which == 1 || i + 1 < B;
        } else {
          px = X, py = Y, off = base + B;
          // the centre itself, once per B iterations: keep the copies of X and Y inside this branch (left alone, the
          // compiler copies them into px / py at the head of EVERY iteration and overwrites them: 18 moves per key)
#pragma unroll
          for (int l = 0; l < FE_LIMBS; ++l) {
            FE_HIDE24(px.n[l]);
            FE_HIDE24(py.n[l]);
          }
        }
        if (valid) check_point<A33, A65, ENDO>(a, &q, off < a.nkeys, px, py, off);
      }
#if ECL_PREFETCH
      pre = nxt;
#endif
    }
    // ---- next centre: C + J with 1/(Jx - X) = inv (or the tangent if C == J)
    fe lam;
    if (!dbl) {
      lam = fe_mul(fe_sub(Jy, Y), inv);
    } else {
      fe x2 = fe_sqr(X);
      lam = fe_mul(fe_add(fe_add(x2, x2), x2), fe_inv(fe_add(Y, Y)));
    }
    fe Xn = fe_add(fe_sqr(lam), fe_neg(fe_add(X, Jx), 2));           // magnitude 4
    fe Yn = fe_sub(fe_mul(lam, fe_add(X, fe_neg(Xn, 4))), Y);        // magnitude 3
    fe_normalize_weak(Xn);
    fe_normalize_weak(Yn);
    X = Xn, Y = Yn;
  }
  cand_flush(a, q);
  fe_st_words2(a.cxy + g, plane, X);
  fe_st_words2(a.cxy + 2 * (size_t)T + g, plane, Y);
}
