// setup_kernels.h - kernels that position the walk: k*G by double-and-add, the walk table, the lane centres.
// (one translation unit: included by ecloop_hip.hip)
#pragma once
#include "add_kernel.h"
#include "ec.h"
// ------------------------------------------------------------------------------------------------ set-up kernels

// affine public keys of n scalars: out[i] = {x[8], y[8]} (canonical), ok[i] = 0 for infinity
__global__ void __launch_bounds__(64) k_mul_g(const u32* __restrict__ k, u32* __restrict__ out, u8* __restrict__ ok, u32 n) {
  u32 i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n) return;
  u32 kk[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) kk[w] = k[(size_t)i * 8 + w];
  fe x, y;
  int fin = ec_mul_g_affine(x, y, kk);
  u32 xw[8], yw[8];
  fe_to_words(xw, x), fe_to_words(yw, y);
#pragma unroll
  for (int w = 0; w < 8; ++w) out[(size_t)i * 16 + w] = xw[w], out[(size_t)i * 16 + 8 + w] = yw[w];
  if (ok) ok[i] = (u8)fin;
}

// canonical words {x[8], y[8]} -> the add kernel's table format {limbs x[9], limbs y[9], pad}
__global__ void k_tab_to_limbs(const u32* __restrict__ words, u32* __restrict__ tab, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x = fe_ldw(words + (size_t)i * 16), y = fe_ldw(words + (size_t)i * 16 + 8);
#pragma unroll
  for (int q = 0; q < FE_LIMBS; ++q) tab[(size_t)i * ECL_TAB_STRIDE + q] = x.n[q], tab[(size_t)i * ECL_TAB_STRIDE + FE_LIMBS + q] = y.n[q];
  tab[(size_t)i * ECL_TAB_STRIDE + 18] = 0, tab[(size_t)i * ECL_TAB_STRIDE + 19] = 0;
}

// lane centres C_g = C_0 + g*D from the ladder {2^j * D}: at most 32 mixed additions + one inversion per lane
__global__ void __launch_bounds__(256) k_init_centres(const u32* __restrict__ c0, const u32* __restrict__ ladder,
                                                       uint4* __restrict__ cxy, u32 T) {
  u32 g = blockIdx.x * 256u + threadIdx.x;
  if (g >= T) return;
  jac acc;
  acc.X = fe_ldw(c0), acc.Y = fe_ldw(c0 + 8), acc.Z = fe_one(), acc.inf = 0;
#pragma unroll 1
  for (int j = 0; j < 32; ++j) {
    if ((g >> j) == 0) break;
    if ((g >> j) & 1u) acc = jac_madd(acc, fe_ldw(ladder + j * 16), fe_ldw(ladder + j * 16 + 8));
  }
  fe x, y;
  jac_to_affine(x, y, acc);
  fe_st_words2(cxy + g, T, x);
  fe_st_words2(cxy + 2 * (size_t)T + g, T, y);
}

// The same centres with the inversion shared: one thread owns INIT_R consecutive lanes, walks them as Jacobian points
// (base from the ladder, then +D each), parks X, Y, Z and the running product of the Z's in `tmp` (the chain scratch
// of the add kernel, idle at this point; planes of T / INIT_R words), inverts the product once and unwinds
// (Montgomery's trick, as lib/ecc.c:522-540 does for the reference's batch).  44 multiplications per centre
// instead of ~410 (most of them the per-lane inversion): 1.6 ms -> 0.25 ms for 2^20 lanes.
#define INIT_R 16u
__global__ void __launch_bounds__(256) k_init_centres_batched(const u32* __restrict__ c0, const u32* __restrict__ ladder,
                                                               uint4* __restrict__ cxy, u32 T, u32* __restrict__ tmp) {
  const u32 nt = T / INIT_R, t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nt) return;
  const u32 g0 = t * INIT_R;
  jac acc;
  acc.X = fe_ldw(c0), acc.Y = fe_ldw(c0 + 8), acc.Z = fe_one(), acc.inf = 0;
#pragma unroll 1
  for (int j = 4; j < 32; ++j) {
    if ((g0 >> j) == 0) break;
    if ((g0 >> j) & 1u) acc = jac_madd(acc, fe_ldw(ladder + j * 16), fe_ldw(ladder + j * 16 + 8));
  }
  const fe dx = fe_ldw(ladder), dy = fe_ldw(ladder + 8);
  fe prod = fe_one();
#pragma unroll 1
  for (u32 r = 0; r < INIT_R; ++r) {
    if (r) acc = jac_madd(acc, dx, dy);
    const fe z = acc.inf ? fe_one() : acc.Z;  // infinity cannot occur for a scan the range check let through
    u32* p = tmp + (size_t)r * 36 * nt + t;
#pragma unroll
    for (int l = 0; l < FE_LIMBS; ++l) {
      p[(size_t)l * nt] = acc.X.n[l], p[(size_t)(9 + l) * nt] = acc.Y.n[l];
      p[(size_t)(18 + l) * nt] = z.n[l], p[(size_t)(27 + l) * nt] = prod.n[l];
    }
    prod = fe_mul(prod, z);
  }
  fe inv = fe_inv(prod);
#pragma unroll 1
  for (u32 r = INIT_R; r-- > 0;) {
    const u32* p = tmp + (size_t)r * 36 * nt + t;
    fe X, Y, Z, pre;
#pragma unroll
    for (int l = 0; l < FE_LIMBS; ++l) {
      X.n[l] = p[(size_t)l * nt], Y.n[l] = p[(size_t)(9 + l) * nt];
      Z.n[l] = p[(size_t)(18 + l) * nt], pre.n[l] = p[(size_t)(27 + l) * nt];
    }
    const fe zi = fe_mul(inv, pre);
    inv = fe_mul(inv, Z);
    const fe zi2 = fe_sqr(zi);
    const fe x = fe_mul(X, zi2), y = fe_mul(Y, fe_mul(zi2, zi));
    fe_st_words2(cxy + g0 + r, T, x);
    fe_st_words2(cxy + 2 * (size_t)T + g0 + r, T, y);
  }
}

// Round 5: the lane centres of a NON-contiguous call from a table that depends on the geometry only.  C_g = C_0 + g D = E + (g + 1) D with
// E = C_0 - D; the multiples (g + 1) D, g < T, are built once per (half group, lanes) by the kernel above (with C_0 = D) and kept in HBM in
// the centres' own planar format (64 bytes per lane: 134 MB for 2^21 lanes).  A call then needs one affine + affine addition per lane, the
// inversion of the x differences shared by the INITT_R lanes of a thread (lanes t, t + nt, ...: coalesced): a chain of ~21 000 instructions
// per thread instead of ~90 000 (ladder of up to 17 mixed additions, 15 more, parking, inversion, unwinding) - the set-up of a call went from
// 0.25-0.43 ms to ~0.1 ms, which is what a 2^21-key job of the reference pays when its jobs are not consecutive (several GPUs), a `rnd` window,
// and every step of the strong-scaling bench.  E = +-(g + 1) D (equal x) cannot be added this way: such a lane takes a unit in the chain and
// the complete formulas with an inversion of its own (E = -(g + 1) D would make the centre the point at infinity: excluded by the range check).
// Lanes per thread: 4 while the walk is small (the chain's latency is what a 2^17-lane call waits for), 8 from 2^19 lanes on (there the
// kernel is bound by its work, and half of that is the inversions).
__device__ __noinline__ void init_centre_complete(fe& x, fe& y, const fe& ex, const fe& ey, const fe& px, const fe& py) {
  jac acc;
  acc.X = ex, acc.Y = ey, acc.Z = fe_one(), acc.inf = 0;
  acc = jac_madd(acc, px, py);
  jac_to_affine(x, y, acc);
}
template <u32 INITT_R>
__global__ void __launch_bounds__(256) k_init_centres_table(const u32* __restrict__ e, const uint4* __restrict__ dtab, uint4* __restrict__ cxy, u32 T) {
  const u32 nt = T / INITT_R, t = blockIdx.x * 256u + threadIdx.x;  // T is a multiple of 256
  if (t >= nt) return;
  u32 ew[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) ew[w] = e[w];
  const fe ex = fe_from_words(ew), ey = fe_ldw(e + 8);
  fe d[INITT_R], pre[INITT_R];
  fe prod = fe_one();
  u32 same = 0;
#pragma unroll
  for (u32 r = 0; r < INITT_R; ++r) {
    const u32 g = r * nt + t;
    const uint4 lo = dtab[g], hi = dtab[(size_t)T + g];
    const u32 pw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    u32 diff = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) diff |= pw[w] ^ ew[w];
    same |= (diff == 0u ? 1u : 0u) << r;
    d[r] = diff ? fe_sub(fe_from_words(pw), ex) : fe_one();  // magnitude 3
    pre[r] = prod;
    prod = fe_mul(prod, d[r]);
  }
  fe inv = fe_inv(prod);
  // (written out for r = 7 ... 0: the compiler declines to unroll a loop of this size, and d[] / pre[] would be indexed through scratch)
#define INITT_STEP(r)                                                                                      \
  {                                                                                                        \
    const u32 g = (r)*nt + t;                                                                              \
    const fe di = fe_mul(inv, pre[r]);                                                                     \
    inv = fe_mul(inv, d[r]);                                                                               \
    const fe px = fe_ld_words2(dtab + g, T), py = fe_ld_words2(dtab + 2 * (size_t)T + g, T);               \
    const fe lam = fe_mul(fe_sub(py, ey), di);                                                             \
    fe x = fe_add(fe_sqr(lam), fe_neg(fe_add(ex, px), 2)); /* magnitude 4 */                               \
    fe_normalize_weak(x);                                                                                  \
    const fe y = fe_add(fe_mul(lam, fe_sub(ex, x)), fe_neg(ey, 1));                                        \
    fe_st_words2(cxy + g, T, x);                                                                           \
    fe_st_words2(cxy + 2 * (size_t)T + g, T, y);                                                           \
  }
  static_assert(INITT_R == 4u || INITT_R == 8u, "k_init_centres_table is written out for four or eight lanes per thread");
  if (INITT_R == 8u) { INITT_STEP(INITT_R - 1u) INITT_STEP(INITT_R - 2u) INITT_STEP(INITT_R - 3u) INITT_STEP(INITT_R - 4u) }
  INITT_STEP(3) INITT_STEP(2) INITT_STEP(1) INITT_STEP(0)
#undef INITT_STEP
  if (__builtin_expect(same != 0u, 0)) {  // the lanes whose table point has E's x: written again, by the complete formulas
#pragma unroll 1
    for (u32 r = 0; r < INITT_R; ++r) {
      if (!((same >> r) & 1u)) continue;
      const u32 g = r * nt + t;
      fe x, y;
      init_centre_complete(x, y, ex, ey, fe_ld_words2(dtab + g, T), fe_ld_words2(dtab + 2 * (size_t)T + g, T));
      fe_st_words2(cxy + g, T, x);
      fe_st_words2(cxy + 2 * (size_t)T + g, T, y);
    }
  }
}
