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
  const u32* __restrict__ tab;  // [B][16]: x[8], y[8] of (i+1)*stride*G, canonical affine
  u32 jump[16];                 // x[8], y[8] of (T*2B*stride)*G
  uint4* __restrict__ cxy;      // lane centres, planes {x.lo, x.hi, y.lo, y.hi} x T
  uint4* __restrict__ scratch;  // prefix products, [(k*2 + half) * T + lane]
  bloom_t bloom;
  ecl_found_dev* found;
  u32* counter;
  u32 cap;
  u32 B;      // table points per group (group = 2B keys)
  u32 T;      // lanes
  u32 nb;     // groups per lane in this launch
  u64 nkeys;  // keys with offset >= nkeys are not tested
};

FE_FN fe fe_ld2(const uint4* p, size_t stride) {
  uint4 lo = p[0], hi = p[stride];
  fe r;
  r.v[0] = lo.x, r.v[1] = lo.y, r.v[2] = lo.z, r.v[3] = lo.w;
  r.v[4] = hi.x, r.v[5] = hi.y, r.v[6] = hi.z, r.v[7] = hi.w;
  return r;
}
FE_FN void fe_st2(uint4* p, size_t stride, const fe& a) {
  p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
  p[stride] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
FE_FN fe fe_ldw(const u32* p) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = p[i];
  return r;
}
FE_FN fe fe_sel(bool c, const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
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

// hash every selected encoding / endomorphism image of the affine point (x, y) and probe the filter
// (check_found_add, main.c:287-347; endo images (x,-y) (bx,y) (bx,-y) (b2x,y) (b2x,-y), main.c:314-327)
template <bool A33, bool A65, bool ENDO>
__device__ __forceinline__ void check_point(const add_args& a, const fe& x, const fe& y, u64 off) {
  fe bx, b2x, ny;
  if (ENDO) {
    const fe beta = FE_BETA1;
    bx = fe_mul(x, beta);
    b2x = fe_neg(fe_add(x, bx));  // beta^2 = -1 - beta
  }
  if (ENDO && A65) ny = fe_neg(y);
  const int nvar = ENDO ? 6 : 1;
#pragma unroll 1
  for (int e = 0; e < nvar; ++e) {
    fe xs = x;
    if (ENDO) xs = e < 2 ? x : (e < 4 ? bx : b2x);
    u32 h[5];
    if (A33) {
      hash160_33(h, xs, (y.v[0] ^ (u32)e) & 1u);  // parity(-y) = !parity(y): p is odd, y != 0
      if (bloom_has(a.bloom, h)) found_push(a, off, h, e, 1);
    }
    if (A65) {
      fe ys = y;
      if (ENDO) ys = (e & 1) ? ny : y;
      hash160_65(h, xs, ys);
      if (bloom_has(a.bloom, h)) found_push(a, off, h, e, 0);
    }
  }
}

template <bool A33, bool A65, bool ENDO>
__global__ void __launch_bounds__(256) k_add(const add_args a) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  const u32 T = a.T, B = a.B;
  if (g >= T) return;
  const size_t plane = T;
  fe X = fe_ld2(a.cxy + g, plane), Y = fe_ld2(a.cxy + 2 * (size_t)T + g, plane);
  const fe Jx = fe_ldw(a.jump), Jy = fe_ldw(a.jump + 8);
  uint4* scr = a.scratch + g;
  const size_t sstep = 2 * (size_t)T;  // one chain element = two planes

#pragma unroll 1
  for (u32 b = 0; b < a.nb; ++b) {
    const u64 base = ((u64)b * T + g) * (2ull * B);
    if (base >= a.nkeys) break;  // groups only grow: nothing left for this lane

    // ---- phase 1: prefix products of e_0 = Jx - X, e_k = Gx_{k-1} - X
    fe acc = fe_sub(Jx, X);
    const bool dbl = fe_is_zero(acc);  // C == J: next centre is 2C (C == -J would be the scalar 0: excluded)
    if (dbl) acc = fe_one();
#pragma unroll 1
    for (u32 k = 1; k <= B; ++k) {
      fe_st2(scr + (size_t)(k - 1) * sstep, plane, acc);
      fe dx = fe_sub(fe_ldw(a.tab + (size_t)(k - 1) * 16), X);
      acc = fe_mul(acc, dx);
    }
    // ---- phase 2: one inversion for the whole chain
    fe inv = fe_inv(acc);
    // ---- phase 3: walk the chain backwards, emit C +- G_i
    fe pre = fe_ld2(scr + (size_t)(B - 1) * sstep, plane);
#pragma unroll 1
    for (u32 k = B; k >= 1; --k) {
      const u32 i = k - 1;
      fe nxt = pre;
      if (k >= 2) nxt = fe_ld2(scr + (size_t)(k - 2) * sstep, plane);  // prefetch for the next iteration
      const fe gx = fe_ldw(a.tab + (size_t)i * 16), gy = fe_ldw(a.tab + (size_t)i * 16 + 8);
      const fe dx = fe_sub(gx, X);
      const fe invk = fe_mul(inv, pre);  // 1 / (Gx_i - X)
      inv = fe_mul(inv, dx);
      const int nwhich = (k == 1) ? 3 : 2;
#pragma unroll 1
      for (int which = 0; which < nwhich; ++which) {
        fe px, py;
        u64 off;
        bool valid = true;
        if (which < 2) {
          // lambda = (+-Gy - Y) / (Gx - X); x3 = lambda^2 - X - Gx; y3 = lambda (X - x3) - Y   (main.c:379-386)
          fe s = which == 0 ? fe_sub(gy, Y) : fe_neg(fe_add(gy, Y));
          fe lam = fe_mul(s, invk);
          px = fe_sub(fe_sub(fe_sqr(lam), X), gx);
          py = fe_sub(fe_mul(lam, fe_sub(X, px)), Y);
          off = which == 0 ? base + B + 1 + i : base + (B - 1 - i);
          valid = which == 1 || i + 1 < B;
        } else {
          px = X, py = Y, off = base + B;
        }
        if (valid && off < a.nkeys) check_point<A33, A65, ENDO>(a, px, py, off);
      }
      pre = nxt;
    }
    // ---- next centre: C + J with 1/(Jx - X) = inv (or the tangent if C == J)
    fe lam;
    if (!dbl) {
      lam = fe_mul(fe_sub(Jy, Y), inv);
    } else {
      fe x2 = fe_sqr(X);
      lam = fe_mul(fe_add(fe_add(x2, x2), x2), fe_inv(fe_add(Y, Y)));
    }
    fe Xn = fe_sub(fe_sub(fe_sqr(lam), X), Jx);
    Y = fe_sub(fe_mul(lam, fe_sub(X, Xn)), Y);
    X = Xn;
  }
  fe_st2(a.cxy + g, plane, X);
  fe_st2(a.cxy + 2 * (size_t)T + g, plane, Y);
}
