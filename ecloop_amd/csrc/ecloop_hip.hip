// ecloop_hip.hip — kernels' instantiation + the C ABI of include/ecloop_hip.h (host side, HIP runtime).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ecloop_hip.hip -o libecloop_hip.so
#include "../../include/ecloop_hip.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "add_kernel.h"
#include "ec.h"
#include "scalar_host.h"

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

// `mul` command body (main.c:530-534, 458-479): public key of each scalar by the fixed-base window method of
// ec_gtable_mul (lib/ecc.c:876-929: W = 14, 19 windows, table slot (2^14-1)*i + b-1 = b * 2^(14 i) * G), then
// ec_jacobi_grprdc (lib/ecc.c:695-707: ONE inversion for the whole batch), then hash + probe.
// <= 19 mixed additions of table points per scalar (64-byte gathers, the 19.9 MB table lives in L2 / Infinity Cache).
// The scalar 0 (mod n) yields no point (the reference emits garbage).
#define GT_W 14u
#define MUL_CHUNK (1u << 22)  /* scalars per staged chunk of ecl_hip_mul_batch (128 MB): 2^18 threads x MUL_R */
#define GT_WINDOWS 19u
#define GT_PER ((1u << GT_W) - 1u)
// k*G as a sum of table points, one per non-zero W-bit digit of k (LSB-first windows, ec_gtable_mul lib/ecc.c:907-929):
// slot PER*w + b-1 = b * 2^(W w) * G with PER = 2^W - 1, canonical x[8], y[8] words per slot.
template <u32 W, u32 NWIN>
__device__ __forceinline__ jac gtable_sum(const u32 kk[9], const u32* __restrict__ gtab) {
  constexpr u32 PER = (1u << W) - 1u;
  jac acc;
  acc.X = fe_zero(), acc.Y = fe_zero(), acc.Z = fe_one(), acc.inf = 1;
#pragma unroll 1
  for (u32 w = 0; w < NWIN; ++w) {
    const u32 bit = w * W, word = bit >> 5, sh = bit & 31;
    u32 lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // static indexing keeps kk[] in registers
      if (word == (u32)j) lo = kk[j], hi = kk[j + 1];
    }
    const u32 digit = (u32)((((u64)hi << 32 | lo) >> sh) & PER);
    if (!digit) continue;
    const u32* e = gtab + ((size_t)w * PER + digit - 1) * 16;
    acc = jac_madd(acc, fe_ldw(e), fe_ldw(e + 8));
  }
  return acc;
}
__device__ __forceinline__ jac gtable_mul(const u32 kk[9], const u32* __restrict__ gtab) { return gtable_sum<GT_W, GT_WINDOWS>(kk, gtab); }

// ---- `mul` has its own window tables, sized for HBM.  The reference's W = 14 (19 windows, 19.9 MB) is sized for a CPU's
// cache (lib/ecc.c:876, `bench-gtable` sweeps it); on a 288 GB part W = 22 costs 3.0 GB and turns 19 additions per
// scalar into 12 (one per non-zero digit).  Same method, same results; measured on 2^24-scalar calls, -a cu, device
// time: W = 14 725 M scalars/s, 16 811, 18 842, 20 878, 22 919, 24 (11.8 GB) 965 (profiles/r03_mul_w_sweep.txt).
// The width is a run-time property of the table (ecl_hip_set_mul_window; by default a context starts on W = 20, 872 MB,
// and moves to W = 22 once it has seen enough scalars to pay for the 50 ms build: ecl_hip_mul_batch).
// The rows are not built by millions of double-and-add ladders but the way the walk's lane centres are: row w is
// P_w, 2 P_w, 3 P_w, ... with P_w = 2^(W w) G - the points C0 + g D of k_init_centres_batched with C0 = D = P_w -
// 44 multiplications per entry, one inversion per 16 entries; the ladder points 2^j P_w of every row come from one
// k_mul_g launch.
struct wtab {
  const u32* p;  // slot per * w + b - 1 = b * 2^(W w) * G, canonical x[8], y[8]
  u32 W, nwin, per, top_per;  // bits per window, windows = ceil(256 / W), 2^W - 1, entries of the last row
};
__host__ __device__ inline wtab wtab_make(const u32* p, u32 W) {
  wtab t;
  t.p = p, t.W = W, t.nwin = (256u + W - 1u) / W, t.per = (1u << W) - 1u;
  t.top_per = (1u << (256u - W * (t.nwin - 1u))) - 1u;
  return t;
}
__host__ __device__ inline size_t wtab_slots(const wtab& t) { return (size_t)(t.nwin - 1u) * t.per + t.top_per; }
#ifndef ECL_WTAB_PREFETCH
#define ECL_WTAB_PREFETCH 1  /* 0: load a window's point when it is added (A/B) */
#endif
__device__ __forceinline__ u32 wtab_digit(const u32 kk[9], const wtab& t, u32 w) {
  const u32 bit = w * t.W, word = bit >> 5, sh = bit & 31;
  u32 lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {  // static indexing keeps kk[] in registers
    if (word == (u32)j) lo = kk[j], hi = kk[j + 1];
  }
  return (u32)((((u64)hi << 32 | lo) >> sh) & t.per);  // kk[8] = 0: the last window is as narrow as it is
}
// The point of window w + 1 is requested before the addition of window w's (64 bytes, 16 registers held across one
// mixed addition): the gathers come from HBM / Infinity Cache and the kernel runs at two waves per SIMD, too few to hide them.
// Measured on 2^24-scalar calls, 22-bit table, four processes each: 995-1001 M scalars/s with, 980-985 without.
__device__ __forceinline__ jac wtab_sum(const u32 kk[9], const wtab t) {
  jac acc;
  acc.X = fe_zero(), acc.Y = fe_zero(), acc.Z = fe_one(), acc.inf = 1;
#if ECL_WTAB_PREFETCH
  uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0;
  u32 dnext = wtab_digit(kk, t, 0);
  if (dnext) {
    const uint4* e = (const uint4*)(t.p + ((size_t)dnext - 1) * 16);
    n0 = e[0], n1 = e[1], n2 = e[2], n3 = e[3];
  }
#pragma unroll 1
  for (u32 w = 0; w < t.nwin; ++w) {
    const u32 digit = dnext;
    const uint4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    dnext = w + 1 < t.nwin ? wtab_digit(kk, t, w + 1) : 0u;
    if (dnext) {
      const uint4* e = (const uint4*)(t.p + ((size_t)(w + 1) * t.per + dnext - 1) * 16);
      n0 = e[0], n1 = e[1], n2 = e[2], n3 = e[3];
    }
    if (!digit) continue;
    const u32 xw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, yw[8] = {c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
    acc = jac_madd(acc, fe_from_words(xw), fe_from_words(yw));
  }
#else
#pragma unroll 1
  for (u32 w = 0; w < t.nwin; ++w) {
    const u32 digit = wtab_digit(kk, t, w);
    if (!digit) continue;
    const u32* e = t.p + ((size_t)w * t.per + digit - 1) * 16;
    acc = jac_madd(acc, fe_ldw(e), fe_ldw(e + 8));
  }
#endif
  return acc;
}
#ifndef ECL_MUL_MMADD
#define ECL_MUL_MMADD 1  /* A/B: 0 = the second point goes through the general mixed addition too (one code body less) */
#endif
// the complete sum out of line: the fallback of a scalar whose lazy sum ended with Z = 0 (never taken by a random scalar)
__device__ __noinline__ jac wtab_sum_complete(const u32 kk[9], const wtab t) { return wtab_sum(kk, t); }
// The same sum for k_mul_check's hot loop: lazy additions without exceptional cases (ec.h: jac_madd_lazy; the caller tests Z once
// at the end), the second point of a sum added to the first as affine + affine (4M + 2S instead of 8M + 3S).  State: npts = 0
// nothing yet, 1 = one table point held as it is (acc.X, acc.Y), >= 2 = Jacobian.  Returns with acc.inf = 1 for an all-zero scalar
// and acc.Z = 1 for a single point.
__device__ __forceinline__ jac wtab_sum_lazy(const u32 kk[9], const wtab t) {
  jac acc;
  acc.X = fe_zero(), acc.Y = fe_zero(), acc.Z = fe_one(), acc.inf = 1;
  u32 npts = 0;
  uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0;
  u32 dnext = wtab_digit(kk, t, 0);
  if (dnext) {
    const uint4* e = (const uint4*)(t.p + ((size_t)dnext - 1) * 16);
    n0 = e[0], n1 = e[1], n2 = e[2], n3 = e[3];
  }
#pragma unroll 1
  for (u32 w = 0; w < t.nwin; ++w) {
    const u32 digit = dnext;
    const uint4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    dnext = w + 1 < t.nwin ? wtab_digit(kk, t, w + 1) : 0u;
    if (dnext) {
      const uint4* e = (const uint4*)(t.p + ((size_t)(w + 1) * t.per + dnext - 1) * 16);
      n0 = e[0], n1 = e[1], n2 = e[2], n3 = e[3];
    }
    if (!digit) continue;
    const u32 xw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, yw[8] = {c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
    const fe qx = fe_from_words(xw), qy = fe_from_words(yw);
    if (npts == 0) acc.X = qx, acc.Y = qy, acc.inf = 0;
#if ECL_MUL_MMADD
    else if (npts == 1) acc = jac_mmadd_lazy(acc.X, acc.Y, qx, qy);
#endif
    else acc = jac_madd_lazy(acc, qx, qy);
    ++npts;
  }
  return acc;
}
// rows w0 + blockIdx.y of the table: out[w * per + g] = (g + 1) * P_w for g < count_w, P_w = ladder[w][0]; one thread owns 16
// consecutive entries (Jacobian, parked in `tmp`, one inversion for the 16 - the scheme of k_init_centres_batched)
__global__ void __launch_bounds__(256) k_gtable_rows(const u32* __restrict__ ladders, u32* __restrict__ table, u32* __restrict__ tmp_all, u32 nt,
                                                      u32 W, u32 w0) {
  const wtab tb = wtab_make(table, W);
  const u32 w = w0 + blockIdx.y, t = blockIdx.x * 256u + threadIdx.x;
  const u32 count = w == tb.nwin - 1u ? tb.top_per : tb.per;
  const u32 g0 = t * 16u;
  if (t >= nt || g0 >= count) return;
  const u32* ladder = ladders + (size_t)w * 32 * 16;
  u32* out = table + (size_t)w * tb.per * 16;
  u32* tmp = tmp_all + (size_t)blockIdx.y * 16 * 36 * nt;
  jac acc;
  acc.X = fe_ldw(ladder), acc.Y = fe_ldw(ladder + 8), acc.Z = fe_one(), acc.inf = 0;
#pragma unroll 1
  for (int j = 4; j < 32; ++j) {
    if ((g0 >> j) == 0) break;
    if ((g0 >> j) & 1u) acc = jac_madd(acc, fe_ldw(ladder + j * 16), fe_ldw(ladder + j * 16 + 8));
  }
  const fe dx = fe_ldw(ladder), dy = fe_ldw(ladder + 8);
  fe prod = fe_one();
#pragma unroll 1
  for (u32 r = 0; r < 16u; ++r) {
    if (r) acc = jac_madd(acc, dx, dy);
    const fe z = acc.inf ? fe_one() : acc.Z;
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
  for (u32 r = 16u; r-- > 0;) {
    const u32* p = tmp + (size_t)r * 36 * nt + t;
    fe X, Y, Z, pre;
#pragma unroll
    for (int l = 0; l < FE_LIMBS; ++l) {
      X.n[l] = p[(size_t)l * nt], Y.n[l] = p[(size_t)(9 + l) * nt];
      Z.n[l] = p[(size_t)(18 + l) * nt], pre.n[l] = p[(size_t)(27 + l) * nt];
    }
    const fe zi = fe_mul(inv, pre);
    inv = fe_mul(inv, Z);
    if (g0 + r >= count) continue;
    const fe zi2 = fe_sqr(zi);
    fe x = fe_mul(X, zi2), y = fe_mul(Y, fe_mul(zi2, zi));
    fe_normalize(x), fe_normalize(y);
    u32 xw[8], yw[8];
    fe_to_words(xw, x), fe_to_words(yw, y);
    uint4* o = (uint4*)(out + (size_t)(g0 + r) * 16);
    o[0] = make_uint4(xw[0], xw[1], xw[2], xw[3]), o[1] = make_uint4(xw[4], xw[5], xw[6], xw[7]);
    o[2] = make_uint4(yw[0], yw[1], yw[2], yw[3]), o[3] = make_uint4(yw[4], yw[5], yw[6], yw[7]);
  }
}
// copies chosen slots of the table out for the bring-up check against the double-and-add kernel
__global__ void k_gather_slots(const u32* __restrict__ table, const u64* __restrict__ slot, u32* __restrict__ out, u32 n) {
  const u32 i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n) return;
  const uint4* s = (const uint4*)(table + slot[i] * 16);
  uint4* o = (uint4*)(out + (size_t)i * 16);
  o[0] = s[0], o[1] = s[1], o[2] = s[2], o[3] = s[3];
}
// One thread owns MUL_R scalars (i = t, t + nt, ...: a wave reads 2 KiB of contiguous scalars per round): their window
// sums stay Jacobian and are parked in `tmp` (planes of nt words: X, Y, Z and the running product of the Z's, 144 bytes
// per scalar) until ONE inversion per thread turns them all affine (Montgomery's trick, as ec_jacobi_grprdc does for
// the reference's 2048-key job): 11 multiplications per non-zero digit + 17 + 7 per scalar instead of 209 + 270 + 3.
#define MUL_R 32u  /* at most (one bit of `infmask` each); short pieces take fewer per thread so that the chip still fills (ecl_hip_mul_batch) */
template <bool A33, bool A65>
__global__ void __launch_bounds__(256) k_mul_check(const u32* __restrict__ k, u32 n, u32 base, const wtab gtab, add_args a,
                                                   u32* __restrict__ tmp, u32 nt, u32 R) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nt) return;
  fe prod = fe_one();
  u32 infmask = 0;
#pragma unroll 1
  for (u32 r = 0; r < R; ++r) {
    const u32 i = r * nt + t;
    if (i >= n) break;
    u32 kk[9];
    const uint4 k0 = ((const uint4*)k)[(size_t)i * 2], k1 = ((const uint4*)k)[(size_t)i * 2 + 1];
    kk[0] = k0.x, kk[1] = k0.y, kk[2] = k0.z, kk[3] = k0.w, kk[4] = k1.x, kk[5] = k1.y, kk[6] = k1.z, kk[7] = k1.w, kk[8] = 0;
    jac acc = wtab_sum_lazy(kk, gtab);
    // an addition that met P = +-Q on the way (h = 0: only scalars that are 0 (mod n) or built around n) leaves Z = 0, and a zero in
    // the product chain would take the thread's other scalars with it: such a scalar goes through the complete formulas instead
    if (!acc.inf && __builtin_expect(fe_is_zero(acc.Z), 0)) acc = wtab_sum_complete(kk, gtab);
    const fe z = acc.inf ? fe_one() : acc.Z;
    infmask |= (acc.inf ? 1u : 0u) << r;
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
  for (u32 r = R; r-- > 0;) {
    const u32 i = r * nt + t;
    if (i >= n) continue;
    const u32* p = tmp + (size_t)r * 36 * nt + t;
    fe X, Y, Z, pre;
#pragma unroll
    for (int l = 0; l < FE_LIMBS; ++l) {
      X.n[l] = p[(size_t)l * nt], Y.n[l] = p[(size_t)(9 + l) * nt];
      Z.n[l] = p[(size_t)(18 + l) * nt], pre.n[l] = p[(size_t)(27 + l) * nt];
    }
    const fe zi = fe_mul(inv, pre);
    inv = fe_mul(inv, Z);
    if ((infmask >> r) & 1u) continue;
    const fe zi2 = fe_sqr(zi);
    const fe x = fe_mul(X, zi2), y = fe_mul(Y, fe_mul(zi2, zi));
    check_point<A33, A65, false>(a, nullptr, true, x, y, (u64)base + i);
  }
}
// `mul -raw` (main.c:505-527): the scalar of a line is the SHA-256 of its bytes.  One lane per line: the line's bytes are
// gathered from the text (any alignment: two aligned words and a funnel shift per message word), padded per FIPS 180-4 and
// compressed block by block; the digest, read as a big-endian 256-bit number, is written where k_mul_check expects the
// scalar (8 little-endian words).  lines[i] = start | length << 32, offsets into `text`; `text` carries 8 spare bytes.
__global__ void __launch_bounds__(256) k_raw_scalars(const u32* __restrict__ text, u32 text_bytes, const u64* __restrict__ lines, u32 n, u32* __restrict__ out,
                                                      u32* __restrict__ bad) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 ln = lines[i];
  const u32 start = (u32)ln;
  u32 L = (u32)(ln >> 32);
  if ((u64)start + L > text_bytes) *bad = 1, L = 0;  // a line outside the text: the call is refused (ECL_E_ARG), nothing is read there
  u32 st[8];
  sha256_init(st);
  const u32 nblk = (L + 9u + 63u) >> 6;
#pragma unroll 1
  for (u32 b = 0; b < nblk; ++b) {
    u32 w[16];
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
      const u32 pos = 64u * b + 4u * j;
      u32 v = 0;
      if (pos < L) {
        const u32 at = start + pos, sh = (at & 3u) * 8u;
        const u32 lo = text[at >> 2], hi = text[(at >> 2) + 1];
        const u32 raw = (u32)((((u64)hi << 32) | lo) >> sh);  // the four bytes at `at`, first byte lowest
        v = __builtin_bswap32(raw);
        const u32 have = L - pos;
        if (have < 4u) v &= ~(0xFFFFFFFFu >> (8u * have));
      }
      if (pos <= L && L - pos < 4u) v |= 0x80000000u >> (8u * (L - pos));
      w[j] = v;
    }
    if (b == nblk - 1u) w[14] = L >> 29, w[15] = L << 3;
    sha256_compress(st, w);
  }
  uint4* o = (uint4*)(out + (size_t)i * 8);
  o[0] = make_uint4(st[7], st[6], st[5], st[4]), o[1] = make_uint4(st[3], st[2], st[1], st[0]);
}
// k*G of ONE scalar (kernel argument) through the window table: the base centre of a non-contiguous add call.
// 19 mixed additions + one inversion (~0.1 ms) instead of the 256-step double-and-add of k_mul_g (~1.2 ms of latency).
struct scalar_arg { u32 w[8]; };
__global__ void __launch_bounds__(64) k_mul_window_one(scalar_arg s, const u32* __restrict__ gtab, u32* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u32 kk[9];
#pragma unroll
  for (int w = 0; w < 8; ++w) kk[w] = s.w[w];
  kk[8] = 0;
  fe x, y;
  jac_to_affine(x, y, gtable_mul(kk, gtab));  // never infinity: the range check excludes the scalar 0
  u32 xw[8], yw[8];
  fe_to_words(xw, x), fe_to_words(yw, y);
#pragma unroll
  for (int w = 0; w < 8; ++w) out[w] = xw[w], out[8 + w] = yw[w];
}

// pk_verify_hash (main.c:248-263) for the hits of a call: re-derive each reported private key's public key on a path
// that shares nothing with the walk (fixed-base window sum over the table that the double-and-add kernel built, own
// inversion per key) and hash it both ways.  One lane per key; ~0.15 ms whatever the count (the walk's hits are few).
__global__ void __launch_bounds__(64) k_verify(const u32* __restrict__ k, u32 n, const u32* __restrict__ gtab, u32* __restrict__ h33,
                                               u32* __restrict__ h65, u8* __restrict__ ok) {
  const u32 i = blockIdx.x * 64u + threadIdx.x;
  if (i >= n) return;
  u32 kk[9];
#pragma unroll
  for (int w = 0; w < 8; ++w) kk[w] = k[(size_t)i * 8 + w];
  kk[8] = 0;
  fe x, y;
  const int fin = jac_to_affine(x, y, gtable_mul(kk, gtab));
  u32 xw[8], yw[8], h[5];
  fe_to_words(xw, x), fe_to_words(yw, y);
  hash160_33(h, xw, yw[0] & 1u);
#pragma unroll
  for (int w = 0; w < 5; ++w) h33[(size_t)i * 5 + w] = h[w];
  hash160_65(h, xw, yw);
#pragma unroll
  for (int w = 0; w < 5; ++w) h65[(size_t)i * 5 + w] = h[w];
  ok[i] = (u8)fin;
}

// ------------------------------------------------------------------------------------------------ diagnostics
__global__ void k_diag_fe(int op, const u32* a, const u32* b, u32* r, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe x = fe_ldw(a + (size_t)i * 8), y = fe_ldw(b + (size_t)i * 8), z;
  switch (op) {
  case 0: z = fe_mul(x, y); break;
  case 1: z = fe_sqr(x); break;
  case 2: z = fe_inv(x); break;
  case 3: z = fe_sub(x, y); break;
  case 4: z = fe_add(x, y); break;
  case 5: z = fe_neg(x, 1); break;
  // chained operations (one result feeding the next with nothing in between): regression tests for the
  // dropped-mask miscompile described in fe256.h
  case 6: z = fe_sqr(fe_sqr(x)); break;
  case 7: z = fe_mul(fe_mul(x, y), y); break;
  default: z = fe_mul(fe_sqr(x), x); break;
  }
  fe_normalize(z);
  u32 zw[8];
  fe_to_words(zw, z);
#pragma unroll
  for (int w = 0; w < 8; ++w) r[(size_t)i * 8 + w] = zw[w];
}
__global__ void k_diag_hash(const u32* x, const u32* y, u32* h33, u32* h65, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 fx[8], fy[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) fx[w] = x[(size_t)i * 8 + w], fy[w] = y[(size_t)i * 8 + w];
  u32 h[5];
  hash160_33(h, fx, fy[0] & 1u);
#pragma unroll
  for (int w = 0; w < 5; ++w) h33[(size_t)i * 5 + w] = h[w];
  hash160_65(h, fx, fy);
#pragma unroll
  for (int w = 0; w < 5; ++w) h65[(size_t)i * 5 + w] = h[w];
}
__global__ void k_diag_bloom(bloom_t b, const u32* h160, u8* hit, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 h[5];
#pragma unroll
  for (int w = 0; w < 5; ++w) h[w] = h160[(size_t)i * 5 + w];
  hit[i] = bloom_has(b, h) ? 1 : 0;
}

// bloom_mod alone, for any filter size (no bit array needed): pins the reciprocal modulo of both width classes
__global__ void k_diag_bloom_mod(bloom_t b, const u64* x, u64* r, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = bloom_mod(b, x[i]);
}

__global__ void k_bloom_insert(bloom_t b, u64* bits, const u32* h160, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 h[5];
#pragma unroll
  for (int w = 0; w < 5; ++w) h[w] = h160[i * 5 + w];
  bloom_add(b, bits, h);
}

// ---- blf-gen's insert loop (utils.c:455-470) in bulk WITH its count: a hash is "new" iff at its turn (input order) at
// least one of its 20 bits is still clear.  The bits themselves do not depend on the order (ORs commute); the count
// does, so it is resolved per chunk of 2^20 hashes: every bit that is clear before the chunk and wanted by a hash of
// the chunk gets an OWNER - the smallest index wanting it - in an open-addressing table (key = bit position, value =
// index, one 64-bit word: atomicCAS claims a slot for a position, atomicMin keeps the smallest index); a hash is new
// iff it owns at least one bit.  Exactly the sequential answer, duplicates and colliding hashes included.
#define BLF_CHUNK_LOG2 20u
#define BLF_TAB_LOG2 26u  /* 2^26 slots for <= 20 * 2^20 wanted bits: load <= 0.32 */
#define BLF_EMPTY (~0ull)
__device__ __forceinline__ u64 blf_slot_hash(u64 p) {
  p *= 0x9E3779B97F4A7C15ull;
  return p >> (64 - BLF_TAB_LOG2);
}
__device__ __forceinline__ u64 blf_bitpos(const bloom_t& b, u64 idx) { return bloom_mod(b, idx >> 6) * 64 + (idx & 63); }
__global__ void k_blf_claim(bloom_t b, const u32* __restrict__ h160, u32 n, u64* __restrict__ tab) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 h[5];
#pragma unroll
  for (int w = 0; w < 5; ++w) h[w] = h160[(size_t)i * 5 + w];
  u64 a[5];
  bloom_words_of(a, h);
#pragma unroll
  for (int p = 0; p < 20; ++p) {  // unrolled: a[] must stay in registers (no runtime indexing)
    const u64 pos = blf_bitpos(b, bloom_index(a, p));
    if ((b.bits[pos >> 6] >> (pos & 63)) & 1) continue;  // set before this chunk: nobody's
    const u64 pack = pos << BLF_CHUNK_LOG2 | i;
    u64 slot = blf_slot_hash(pos);
    for (;;) {
      u64 cur = tab[slot];
      if (cur == BLF_EMPTY) {
        cur = atomicCAS((unsigned long long*)&tab[slot], BLF_EMPTY, pack);
        if (cur == BLF_EMPTY) break;
      }
      if ((cur >> BLF_CHUNK_LOG2) == pos) {
        atomicMin((unsigned long long*)&tab[slot], pack);
        break;
      }
      slot = (slot + 1) & ((1ull << BLF_TAB_LOG2) - 1);
    }
  }
}
__global__ void k_blf_count_and_set(bloom_t b, u64* __restrict__ bits, const u32* __restrict__ h160, u32 n, const u64* __restrict__ tab,
                                    unsigned long long* __restrict__ added) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  bool is_new = false;
  if (i < n) {
    u32 h[5];
#pragma unroll
    for (int w = 0; w < 5; ++w) h[w] = h160[(size_t)i * 5 + w];
    u64 a[5];
    bloom_words_of(a, h);
#pragma unroll
    for (int p = 0; p < 20; ++p) {
      const u64 pos = blf_bitpos(b, bloom_index(a, p));
      u64 slot = blf_slot_hash(pos);
      for (;;) {  // owner lookup: absent = the bit was set before the chunk
        const u64 cur = tab[slot];
        if (cur == BLF_EMPTY) break;
        if ((cur >> BLF_CHUNK_LOG2) == pos) {
          is_new |= (u32)(cur & ((1u << BLF_CHUNK_LOG2) - 1)) == i;
          break;
        }
        slot = (slot + 1) & ((1ull << BLF_TAB_LOG2) - 1);
      }
    }
  }
  const u64 m = __builtin_amdgcn_ballot_w64(is_new);
  if ((threadIdx.x & 63u) == 0 && m) atomicAdd(added, (unsigned long long)__builtin_popcountll(m));
  // the bits are set by a separate launch of k_bloom_insert AFTER this kernel: owners are looked up against the
  // filter state before the chunk
}

// ------------------------------------------------------------------------------------------------ context

struct ecl_hip {
  int dev = 0;
  u32 flags = 0, offs = 0;
  u32 B = 1024;       // table points per group
  u32 Tmax = 0;       // lanes walked concurrently (0 = derive from occupancy)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  // device buffers
  u32* d_tab = nullptr;  u32 tab_B = 0;        // table for (B, offs)
  u32* d_gtab = nullptr;                       // fixed-base window table for `mul` (19 x 16383 affine points)
  u32* d_aux = nullptr;                        // [0]=C0, [1]=jump, [2..33]=ladder : 34 points x 16 words
  u32* d_auxk = nullptr;                       // scalars for the above
  uint4* d_cxy = nullptr; size_t cxy_T = 0;
  uint4* d_scr = nullptr; u32* d_scr2 = nullptr; size_t scr_elems = 0;  // prefix-product chains
  u64* d_bloom = nullptr; u64 bloom_words = 0;
  // `mul`: scalars travel in chunks through two pinned staging buffers, copy engine and kernel overlapped
  u32* d_kbuf[2] = {nullptr, nullptr}; u32* pin_k[2] = {nullptr, nullptr}; u32 kbuf_cap = 0, pin_cap = 0;
  u32* d_multmp = nullptr;                     // parked Jacobian sums of one chunk (144 bytes per scalar)
  const u32* d_multab = nullptr; u32 multab_W = 0;  // `mul`'s window table in use: one per (device, width), shared by the contexts
  u32 mul_W_fixed = 0;                         // ecl_hip_set_mul_window: 0 = automatic
  uint64_t mul_seen = 0;                       // scalars this context has multiplied (never reset: the automatic width goes by it)
  bool mul_long_failed = false;                // the long table could not be allocated: do not try again
  void* d_ver = nullptr; u32 ver_cap = 0;      // staging of ecl_hip_verify
  u32* d_rawtext = nullptr; size_t rawtext_cap = 0; u64* d_rawlines = nullptr; u32 rawlines_cap = 0;  // `mul -raw`: text and line table of one call
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
  u32* d_list = nullptr; u64 list_n = 0;       // optional sorted hash list (exact confirm on the device)
  ecl_found_dev* d_found = nullptr; u32 found_cap = 0;
  u32* d_counter = nullptr;
  // walk state for contiguous continuation
  bool walk_valid = false;
  u32 walk_T = 0, walk_B = 0;
  bool B_auto = true;  // half group not fixed by the caller: short calls take a smaller one (see ecl_hip_add_range)
  u256 walk_next;  // scalar (mod n) the resident centres are positioned for
  u32 jump_host[16];
  u32 aux_B = 0, aux_T = 0;  // geometry the jump and the ladder in d_aux / jump_host were computed for
  // timing
  double kernel_ms = 0, setup_ms = 0, mul_ms = 0;
  uint64_t launches = 0, keys = 0, setups = 0, mul_calls = 0, mul_scalars = 0;
  hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr;
};

#define HIPCHK(h, call)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
      return ECL_E_HIP;                                                                    \
    }                                                                                      \
  } while (0)

template <typename T>
struct dbuf {
  T* p = nullptr;
  ~dbuf() { if (p) (void)hipFree(p); }
};

static void release_multable(struct ecl_hip* h);

static void words_of(u32 w[8], const u256& a) {
  for (int i = 0; i < 4; ++i) w[2 * i] = (u32)a.w[i], w[2 * i + 1] = (u32)(a.w[i] >> 32);
}

extern "C" {

int ecl_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* ecl_hip_strerror(int code) {
  switch (code) {
  case ECL_OK: return "ok";
  case ECL_E_ARG: return "bad argument";
  case ECL_E_HIP: return "HIP runtime error";
  case ECL_E_NODEV: return "no such GPU";
  case ECL_E_OVERFLOW: return "more hits than the output buffer holds";
  case ECL_E_NOBLOOM: return "no bloom filter set";
  case ECL_E_RANGE: return "range touches scalar 0 (mod n)";
  case ECL_E_SELFTEST: return "device self-test failed";
  default: return "unknown error";
  }
}
const char* ecl_hip_last_error(const ecl_hip* h) { return h ? h->err.c_str() : ""; }

int ecl_hip_open(ecl_hip** out, int device, uint32_t flags, uint32_t ord_offs) {
  if (!out || ord_offs > 255 || !(flags & (ECL_ADDR33 | ECL_ADDR65)) || (flags & ~7u)) return ECL_E_ARG;
  int n = ecl_hip_device_count();
  if (device < 0 || device >= n) return ECL_E_NODEV;
  ecl_hip* h = new ecl_hip();
  h->dev = device, h->flags = flags, h->offs = ord_offs;
  *out = h;
  HIPCHK(h, hipSetDevice(device));
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreate(&h->ev0));
  HIPCHK(h, hipEventCreate(&h->ev1));
  HIPCHK(h, hipEventCreate(&h->ev_s0));
  HIPCHK(h, hipEventCreate(&h->ev_s1));
  HIPCHK(h, hipMalloc(&h->d_aux, 34 * 16 * sizeof(u32)));
  HIPCHK(h, hipMalloc(&h->d_auxk, 34 * 8 * sizeof(u32)));
  HIPCHK(h, hipMalloc(&h->d_counter, 4 * sizeof(u32)));  // [0] records appended, [1] records confirmed by the list, [2] bad-input flag of mul_batch_raw
  // the self-test checks the CODE (known answers, walk kernel against the double-and-add kernel): once per process
  // for every (device, kernel selection) is enough - eight handles for eight shards of one scan do not repeat it
  static std::mutex mu;
  static std::set<u32> passed;
  const char* skip = getenv("ECL_HIP_SKIP_SELFTEST");
  const u32 key = (u32)device * 8u + flags;
  {
    std::lock_guard<std::mutex> lk(mu);
    if ((skip && skip[0] == '1') || passed.count(key)) return ECL_OK;
  }
  const int rc = ecl_hip_selftest(h);
  if (rc == ECL_OK) {
    std::lock_guard<std::mutex> lk(mu);
    passed.insert(key);
  }
  return rc;
}

void ecl_hip_close(ecl_hip* h) {
  if (!h) return;
  (void)hipSetDevice(h->dev);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  (void)hipFree(h->d_tab), (void)hipFree(h->d_gtab), (void)hipFree(h->d_aux), (void)hipFree(h->d_auxk), (void)hipFree(h->d_cxy);
  (void)hipFree(h->d_scr), (void)hipFree(h->d_scr2), (void)hipFree(h->d_bloom), (void)hipFree(h->d_list), (void)hipFree(h->d_found), (void)hipFree(h->d_counter);
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(h->d_kbuf[i]);
    if (h->pin_k[i]) (void)hipHostFree(h->pin_k[i]);
    if (h->ev_copied[i]) (void)hipEventDestroy(h->ev_copied[i]);
    if (h->ev_free[i]) (void)hipEventDestroy(h->ev_free[i]);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  (void)hipFree(h->d_multmp), (void)hipFree(h->d_ver), (void)hipFree(h->d_rawtext), (void)hipFree(h->d_rawlines);
  release_multable(h);
  if (h->ev_s0) (void)hipEventDestroy(h->ev_s0);
  if (h->ev_s1) (void)hipEventDestroy(h->ev_s1);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int ecl_hip_set_bloom(ecl_hip* h, const uint64_t* bits, uint64_t nwords) {
  if (!h || !bits || nwords == 0 || nwords >= (1ull << 58)) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->d_bloom) HIPCHK(h, hipFree(h->d_bloom));
  h->d_bloom = nullptr, h->bloom_words = 0;
  HIPCHK(h, hipMalloc(&h->d_bloom, nwords * sizeof(u64)));
  // on the handle's own stream: device threads upload in parallel, each over its own PCIe link; a buffer pinned with
  // ecl_hip_pin_host goes by DMA at link rate, a pageable one is staged by the runtime
  HIPCHK(h, hipMemcpyAsync(h->d_bloom, bits, nwords * sizeof(u64), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->bloom_words = nwords;
  return ECL_OK;
}

// Page-locking works on whole pages.  A small buffer shares its pages with whatever else the host allocator put there,
// and registering / unregistering such a page under other host buffers that the runtime copies from or to ended in GPU
// memory access faults on host heap addresses (tools/fuzz_mul_gpu.py, round 2: always a few calls after a 64-byte or
// 2 KB array had been pinned).  Buffers below 1 MiB are therefore left alone - they gain nothing from DMA anyway - and
// the pair of calls stays symmetric through a registry of what was really registered.
#define ECL_PIN_MIN_BYTES ((size_t)1 << 20)
static std::mutex g_pin_mu;
static std::set<const void*> g_pinned;
int ecl_hip_pin_host(const void* p, size_t bytes) {
  if (!p || !bytes) return ECL_E_ARG;
  if (bytes < ECL_PIN_MIN_BYTES) return ECL_OK;
  if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return ECL_E_HIP;
  }
  std::lock_guard<std::mutex> lk(g_pin_mu);
  g_pinned.insert(p);
  return ECL_OK;
}
void* ecl_hip_alloc_host(size_t bytes) {
  void* p = nullptr;
  if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
  return p;
}
void ecl_hip_free_host(void* p) {
  if (p) (void)hipHostFree(p);
}
int ecl_hip_unpin_host(const void* p) {
  if (!p) return ECL_E_ARG;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (!g_pinned.erase(p)) return ECL_OK;  // never registered (too small): nothing to undo
  }
  return hipHostUnregister(const_cast<void*>(p)) == hipSuccess ? ECL_OK : ECL_E_HIP;
}

int ecl_hip_set_list(ecl_hip* h, const uint32_t (*h160)[5], uint64_t n) {
  if (!h || (n && !h160)) return ECL_E_ARG;
  for (uint64_t i = 1; i < n; ++i) {  // strictly increasing in compare_160 order (addr.c:18-26)
    int c = 0;
    for (int k = 0; k < 5 && c == 0; ++k) c = h160[i - 1][k] < h160[i][k] ? -1 : (h160[i - 1][k] > h160[i][k] ? 1 : 0);
    if (c >= 0) {
      h->err = "ecl_hip_set_list: the list is not sorted and unique";
      return ECL_E_ARG;
    }
  }
  HIPCHK(h, hipSetDevice(h->dev));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->d_list) HIPCHK(h, hipFree(h->d_list));
  h->d_list = nullptr, h->list_n = 0;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipMalloc(&h->d_list, n * 20));
  HIPCHK(h, hipMemcpy(h->d_list, h160, n * 20, hipMemcpyHostToDevice));
  h->list_n = n;
  return ECL_OK;
}

// ---- load_filter's list preparation (main.c:96-131: qsort by compare_160, then one blf_add per entry) on the device ------
// 10^7 entries cost the host 13 s (qsort of 20-byte records + 2 * 10^8 scattered bit sets), 10^8 two minutes; here: five
// stable 32-bit radix passes over a permutation (least significant word first = compare_160's word-by-word order,
// addr.c:18-26), a gather, adjacent-duplicate flags + exclusive scan + scatter.  The bits are set by the bulk insert
// kernel into a filter of the reference's list-mode size (2 words per entry).
}  // extern "C"
__global__ void k_list_iota(u32* perm, u32 n) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) perm[i] = i;
}
__global__ void k_list_key(const u32* __restrict__ rec, const u32* __restrict__ perm, u32* __restrict__ key, u32 n, u32 word) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) key[i] = rec[(size_t)perm[i] * 5 + word];
}
__global__ void k_list_gather_flag(const u32* __restrict__ rec, const u32* __restrict__ perm, u32* __restrict__ out, u32* __restrict__ flag, u32 n) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u32* a = rec + (size_t)perm[i] * 5;
  bool first = i == 0;
  if (!first) {
    const u32* b = rec + (size_t)perm[i - 1] * 5;
    first = (a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]) | (a[3] != b[3]) | (a[4] != b[4]);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) out[(size_t)i * 5 + k] = a[k];
  flag[i] = first ? 1u : 0u;
}
__global__ void k_list_compact(const u32* __restrict__ in, const u32* __restrict__ flag, const u32* __restrict__ pos, u32* __restrict__ out, u32 n) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n || !flag[i]) return;
#pragma unroll
  for (int k = 0; k < 5; ++k) out[(size_t)pos[i] * 5 + k] = in[(size_t)i * 5 + k];
}
extern "C" {
int ecl_hip_sort_list(ecl_hip* h, uint32_t (*h160)[5], uint64_t n, uint64_t* kept) {
  if (!h || !kept || (n && !h160) || n >= (1ull << 31)) return ECL_E_ARG;
  *kept = 0;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  const u32 N = (u32)n;
  dbuf<u32> rec, out, key[2], perm[2], flag, pos;
  dbuf<u8> tmp;
  HIPCHK(h, hipMalloc(&rec.p, (size_t)N * 20));
  HIPCHK(h, hipMalloc(&out.p, (size_t)N * 20));
  for (int i = 0; i < 2; ++i) {
    HIPCHK(h, hipMalloc(&key[i].p, (size_t)N * 4));
    HIPCHK(h, hipMalloc(&perm[i].p, (size_t)N * 4));
  }
  HIPCHK(h, hipMalloc(&flag.p, (size_t)N * 4));
  HIPCHK(h, hipMalloc(&pos.p, (size_t)N * 4));
  size_t need_sort = 0, need_scan = 0;
  HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(nullptr, need_sort, key[0].p, key[1].p, perm[0].p, perm[1].p, (int)N, 0, 32, h->stream));
  HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(nullptr, need_scan, flag.p, pos.p, (int)N, h->stream));
  size_t need = need_sort > need_scan ? need_sort : need_scan;
  HIPCHK(h, hipMalloc(&tmp.p, need ? need : 16));
  HIPCHK(h, hipMemcpyAsync(rec.p, h160, (size_t)N * 20, hipMemcpyHostToDevice, h->stream));
  const dim3 grid((N + 255) / 256), blk(256);
  hipLaunchKernelGGL(k_list_iota, grid, blk, 0, h->stream, perm[0].p, N);
  int cur = 0;
  for (int word = 4; word >= 0; --word) {  // LSD: the last word first, every pass stable
    hipLaunchKernelGGL(k_list_key, grid, blk, 0, h->stream, rec.p, perm[cur].p, key[0].p, N, (u32)word);
    HIPCHK(h, hipcub::DeviceRadixSort::SortPairs(tmp.p, need, key[0].p, key[1].p, perm[cur].p, perm[cur ^ 1].p, (int)N, 0, 32, h->stream));
    cur ^= 1;
  }
  hipLaunchKernelGGL(k_list_gather_flag, grid, blk, 0, h->stream, rec.p, perm[cur].p, out.p, flag.p, N);
  HIPCHK(h, hipcub::DeviceScan::ExclusiveSum(tmp.p, need, flag.p, pos.p, (int)N, h->stream));
  hipLaunchKernelGGL(k_list_compact, grid, blk, 0, h->stream, out.p, flag.p, pos.p, rec.p, N);
  HIPCHK(h, hipGetLastError());
  u32 last_pos = 0, last_flag = 0;
  HIPCHK(h, hipMemcpyAsync(&last_pos, pos.p + (N - 1), 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&last_flag, flag.p + (N - 1), 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const u64 k = (u64)last_pos + last_flag;
  HIPCHK(h, hipMemcpy(h160, rec.p, (size_t)k * 20, hipMemcpyDeviceToHost));
  *kept = k;
  return ECL_OK;
}

int ecl_hip_set_geometry(ecl_hip* h, uint32_t half_group, uint32_t max_lanes) {
  if (!h) return ECL_E_ARG;
  if (half_group) {
    if (half_group < 2 || half_group > (1u << 16)) return ECL_E_ARG;
    h->B = half_group, h->B_auto = false;
  }
  if (max_lanes) h->Tmax = (max_lanes + 255u) & ~255u;
  h->walk_valid = false;
  return ECL_OK;
}

int ecl_hip_get_timing(ecl_hip* h, double* kernel_ms, uint64_t* launches, uint64_t* keys) {
  if (!h) return ECL_E_ARG;
  if (kernel_ms) *kernel_ms = h->kernel_ms;
  if (launches) *launches = h->launches;
  if (keys) *keys = h->keys;
  return ECL_OK;
}
int ecl_hip_reset_timing(ecl_hip* h) {
  if (!h) return ECL_E_ARG;
  h->kernel_ms = 0, h->launches = 0, h->keys = 0, h->setup_ms = 0, h->setups = 0;
  h->mul_ms = 0, h->mul_calls = 0, h->mul_scalars = 0;
  return ECL_OK;
}
int ecl_hip_get_setup_timing(ecl_hip* h, double* setup_ms, uint64_t* setups) {
  if (!h) return ECL_E_ARG;
  if (setup_ms) *setup_ms = h->setup_ms;
  if (setups) *setups = h->setups;
  return ECL_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ add path (host)

typedef void (*add_kernel_t)(const add_args);
static add_kernel_t pick_add_kernel(u32 flags) {
  bool a33 = flags & ECL_ADDR33, a65 = flags & ECL_ADDR65, endo = flags & ECL_ENDO;
  if (a33 && !a65) return endo ? k_add<true, false, true> : k_add<true, false, false>;
  if (!a33 && a65) return endo ? k_add<false, true, true> : k_add<false, true, false>;
  return endo ? k_add<true, true, true> : k_add<true, true, false>;
}

// ctx_check_hash's second step (main.c:212-216) for the records a search kernel left in `in`: bsearch over the sorted
// list (order of compare_160, addr.c:18-26: lexicographic on the five words); members are compacted into `out`.
// Its own tiny kernel after the search kernel, so the hot loop carries nothing for it (in the loop it cost 0.5 %).
__global__ void k_list_filter(const ecl_found_dev* in, const u32* counters, u32 in_cap, const u32* list, u64 list_n,
                              ecl_found_dev* out, u32* out_counter, u32 out_cap) {
  const u32 n_in = counters[0] < in_cap ? counters[0] : in_cap;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += gridDim.x * blockDim.x) {
    const ecl_found_dev r = in[i];
    u64 lo = 0, hi = list_n;
    bool hit = false;
    while (lo < hi) {
      const u64 mid = lo + ((hi - lo) >> 1);
      const u32* e = list + mid * 5;
      int c = 0;
      for (int k = 4; k >= 0; --k) c = e[k] < r.h160[k] ? -1 : (e[k] > r.h160[k] ? 1 : c);  // word 0 decides last
      if (c == 0) { hit = true; break; }
      if (c < 0) lo = mid + 1; else hi = mid;
    }
    if (hit) {
      const u32 idx = atomicAdd(out_counter, 1u);
      if (idx < out_cap) out[idx] = r;
    }
  }
}

// found records of one call: [0, raw_cap) written by the search kernel; in list mode the confirmed ones are
// compacted into [raw_cap, raw_cap + cap).  d_counter[0] = records pushed, d_counter[1] = records confirmed.
#define ECL_LIST_RAW_CAP (1u << 20)
static int ensure_found(ecl_hip* h, u32 cap) {
  if (cap <= h->found_cap) return ECL_OK;
  if (h->d_found) HIPCHK(h, hipFree(h->d_found));
  h->d_found = nullptr, h->found_cap = 0;
  HIPCHK(h, hipMalloc(&h->d_found, (size_t)cap * sizeof(ecl_found_dev)));
  h->found_cap = cap;
  return ECL_OK;
}

static u32 raw_cap_of(const ecl_hip* h, u32 cap) { return h->d_list ? (cap > ECL_LIST_RAW_CAP ? cap : ECL_LIST_RAW_CAP) : cap; }
// after the search kernel has been queued on h->stream: optional list confirm, then counters and records to the host
static int collect_found(ecl_hip* h, u32 cap, u32 rcap, ecl_found* out, u32* nout, bool keep_endo) {
  const bool lst = h->d_list != nullptr;
  if (lst) {
    const u32 blocks = (rcap + 255) / 256 < 1024 ? (rcap + 255) / 256 : 1024;
    hipLaunchKernelGGL(k_list_filter, dim3(blocks), dim3(256), 0, h->stream, h->d_found, h->d_counter, rcap, h->d_list, h->list_n,
                       h->d_found + rcap, h->d_counter + 1, cap);
    HIPCHK(h, hipGetLastError());
  }
  u32 cnts[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(cnts, h->d_counter, sizeof cnts, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const u32 cnt = lst ? cnts[1] : cnts[0];
  const u32 take = cnt < cap ? cnt : cap;
  if (take) {
    std::vector<ecl_found_dev> tmp(take);
    HIPCHK(h, hipMemcpy(tmp.data(), h->d_found + (lst ? rcap : 0), (size_t)take * sizeof(ecl_found_dev), hipMemcpyDeviceToHost));
    for (u32 i = 0; i < take; ++i) {
      out[i].key_offset = tmp[i].key_offset;
      memcpy(out[i].h160, tmp[i].h160, 20);
      out[i].endo = keep_endo ? (uint8_t)(tmp[i].tag & 0xff) : 0, out[i].compressed = (uint8_t)((tmp[i].tag >> 8) & 1);
      out[i].pad[0] = out[i].pad[1] = 0;
    }
  }
  *nout = cnt;
  if (lst && cnts[0] > rcap) {  // the bloom let more through than the staging area holds: some were never looked up
    h->err = "list mode: more bloom hits in one call than the device staging area holds";
    *nout = cnts[0];
    return ECL_E_OVERFLOW;
  }
  return cnt > cap ? ECL_E_OVERFLOW : ECL_OK;
}

static int default_lanes(ecl_hip* h) {
  if (h->Tmax) return ECL_OK;
  hipDeviceProp_t p;
  HIPCHK(h, hipGetDeviceProperties(&p, h->dev));
  int occ = 0;
  HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)pick_add_kernel(h->flags), ECL_ADD_BLOCK, 0));
  if (occ < 1) occ = 1;
  const u32 resident = (u32)p.multiProcessorCount * (u32)occ * (u32)ECL_ADD_BLOCK;
  // Oversubscribe: with exactly the resident number of lanes every wave of the chip is in the same phase at the same
  // time (prefix products, then the inversion chain, then the hash-heavy walk back); with several times more blocks
  // than slots the dispatcher staggers them and the phases overlap.  Measured on addr33: 196608 lanes (resident)
  // 10.9 Gkeys/s, 786432 11.5, 1048576 12.0, 2097152 12.1 (final kernel: 12.49 / 12.63 at 2^20 / 2^21, no more
  // beyond).  The chains cost lanes * B * 36 bytes of HBM (2^21 lanes, B = 1024: 77 GB of the 288), so the factor is
  // cut back if memory is short.
  u64 lanes = 1ull << 21;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
    while (lanes > resident && lanes * h->B * 36ull > free_b / 3) lanes /= 2;
  }
  if (lanes < resident) lanes = resident;
  h->Tmax = (u32)((lanes + 255) & ~255ull);
  return ECL_OK;
}

// table of (i+1)*stride*G, i < B (ctx_precompute_gpoints, main.c:219-246: only x,y of the positive half are stored)
static int ensure_table(ecl_hip* h) {
  if (h->d_tab && h->tab_B == h->B) return ECL_OK;
  if (h->d_tab) HIPCHK(h, hipFree(h->d_tab));
  h->d_tab = nullptr;
  const u32 B = h->B;
  std::vector<u32> ks((size_t)B * 8);
  u256 s = sc_pow2(h->offs), cur = s;
  for (u32 i = 0; i < B; ++i) {
    words_of(&ks[(size_t)i * 8], cur);
    cur = sc_add(cur, s);
  }
  u32 *d_k = nullptr, *d_w = nullptr;
  HIPCHK(h, hipMalloc(&d_k, ks.size() * sizeof(u32)));
  HIPCHK(h, hipMalloc(&d_w, (size_t)B * 16 * sizeof(u32)));
  HIPCHK(h, hipMalloc(&h->d_tab, (size_t)B * ECL_TAB_STRIDE * sizeof(u32)));
  HIPCHK(h, hipMemcpyAsync(d_k, ks.data(), ks.size() * sizeof(u32), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_mul_g, dim3((B + 63) / 64), dim3(64), 0, h->stream, d_k, d_w, (u8*)nullptr, B);
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_tab_to_limbs, dim3((B + 63) / 64), dim3(64), 0, h->stream, d_w, h->d_tab, B);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipFree(d_k));
  HIPCHK(h, hipFree(d_w));
  h->tab_B = B;
  return ECL_OK;
}

extern "C" int ecl_hip_get_geometry(ecl_hip* h, uint32_t* half_group, uint32_t* lanes) {
  if (!h) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc = default_lanes(h);
  if (rc != ECL_OK) return rc;
  if (half_group) *half_group = h->B;
  if (lanes) *lanes = h->Tmax;
  return ECL_OK;
}

// Geometry of one call.  The table for half group h->B holds the tables of all smaller ones as prefixes.  A call too
// short to give every one of the Tmax lanes a whole group takes a smaller half group (down to 256) instead of fewer
// lanes: the walk only reaches its rate when the chip is oversubscribed with blocks in different phases (2^29 keys:
// 11.5 Gkeys/s with 1024 x 2^18 lanes, 12.1 with 256 x 2^20; the price is a larger share of the inversion: 270 / 2B
// multiplications per key).  Contiguous calls of one size keep one geometry, so they still continue the resident walk.
static void call_geometry(const ecl_hip* h, u64 nkeys, u32& B, u32& nb, u32& T) {
  B = h->B;
  if (h->B_auto)
    while (B > 256 && nkeys < (u64)h->Tmax * 2 * B) B >>= 1;
  const u64 group = 2ull * B, ngroups = (nkeys + group - 1) / group;
  // nb groups per lane, then the smallest lane count (multiple of 256) that covers the range: no lane idles
  // through a mostly masked last group
  nb = (u32)((ngroups + h->Tmax - 1) / h->Tmax);
  T = (u32)(((ngroups + nb - 1) / nb + 255) & ~255ull);
  if (T > h->Tmax) T = h->Tmax;
}
// what call_geometry can represent: the group count must not wrap and groups per lane must fit 32 bits
// (a caller-set geometry of 2 x 256 lanes reaches that at 2^42 keys; the default one never does below 2^63)
static bool nkeys_ok(const ecl_hip* h, u64 nkeys) {
  if (nkeys > (1ull << 63)) return false;
  u32 B = h->B;
  if (h->B_auto)
    while (B > 256 && nkeys < (u64)h->Tmax * 2 * B) B >>= 1;
  const u64 ngroups = (nkeys + 2ull * B - 1) / (2ull * B);
  return (ngroups + h->Tmax - 1) / h->Tmax < (1ull << 32);
}
// device buffers of a call with that geometry: lane centres and prefix-product chains
static int ensure_walk_buffers(ecl_hip* h, u32 B, u32 T) {
  if (h->cxy_T < T) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_cxy) HIPCHK(h, hipFree(h->d_cxy));
    h->d_cxy = nullptr, h->cxy_T = 0, h->walk_valid = false;
    HIPCHK(h, hipMalloc(&h->d_cxy, (size_t)T * 4 * sizeof(uint4)));
    h->cxy_T = T;
  }
  const size_t need = (size_t)T * B * 2;
  if (h->scr_elems < need) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_scr) HIPCHK(h, hipFree(h->d_scr));
    if (h->d_scr2) HIPCHK(h, hipFree(h->d_scr2));
    h->d_scr = nullptr, h->d_scr2 = nullptr, h->scr_elems = 0;
    HIPCHK(h, hipMalloc(&h->d_scr, need * sizeof(uint4)));
    HIPCHK(h, hipMalloc(&h->d_scr2, (need / 2) * sizeof(u32)));
    h->scr_elems = need;
  }
  return ECL_OK;
}

static int ensure_gtable(ecl_hip* h);

extern "C" int ecl_hip_reserve(ecl_hip* h, uint64_t nkeys, uint32_t cap) {
  if (!h || nkeys == 0) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  if ((rc = default_lanes(h)) != ECL_OK) return rc;
  if ((rc = ensure_table(h)) != ECL_OK) return rc;
  if ((rc = ensure_gtable(h)) != ECL_OK) return rc;
  if (!nkeys_ok(h, nkeys)) return ECL_E_ARG;
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, rcap + (h->d_list ? cap : 0))) != ECL_OK) return rc;
  u32 B, nb, T;
  call_geometry(h, nkeys, B, nb, T);
  return ensure_walk_buffers(h, B, T);
}

extern "C" int ecl_hip_add_range(ecl_hip* h, const uint64_t start[4], uint64_t nkeys, ecl_found* out, uint32_t cap,
                                 uint32_t* nout) {
  if (!h || !start || (!out && cap) || !nout) return ECL_E_ARG;
  *nout = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (nkeys == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  if ((rc = default_lanes(h)) != ECL_OK) return rc;
  if (!nkeys_ok(h, nkeys)) {
    h->err = "nkeys too large for one call with this geometry";
    return ECL_E_ARG;
  }
  if ((rc = ensure_table(h)) != ECL_OK) return rc;
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, rcap + (h->d_list ? cap : 0))) != ECL_OK) return rc;

  u32 B, nb, T;
  call_geometry(h, nkeys, B, nb, T);
  const u64 group = 2ull * B;
  if ((rc = ensure_walk_buffers(h, B, T)) != ECL_OK) return rc;

  u256 k0 = sc_reduce(u256_from(start));
  const u256 s = sc_pow2(h->offs);
  {
    // The walk cannot represent the point at infinity: refuse a scan that contains the scalar 0 (mod n), like the
    // reference's range check (main.c:687-690).  j0 = offset of that key = -k0 / 2^offs (mod n).
    u256 j0 = sc_neg(k0);
    for (u32 i = 0; i < h->offs; ++i) j0 = sc_half(j0);
    const u64 walked = (u64)nb * T * group;
    if (!(j0.w[1] | j0.w[2] | j0.w[3]) && j0.w[0] < walked && j0.w[0] < nkeys + group) {
      h->err = "the scan contains the private key 0 (mod n)";
      return ECL_E_RANGE;
    }
  }
  bool cont = h->walk_valid && h->walk_T == T && h->walk_B == B && u256_eq(h->walk_next, k0);
  if (!cont) {
    if ((rc = ensure_gtable(h)) != ECL_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev_s0, h->stream));
    if (h->aux_B != B || h->aux_T != T) {
      // jump = (T*2B*s)*G and the ladder 2^j * (2B*s)*G depend on the geometry only: kept across calls
      h->aux_B = h->aux_T = 0;
      const u256 d = sc_mul_u64(s, group);
      std::vector<u32> ks(33 * 8);
      words_of(&ks[0], sc_mul_u64(d, T));
      u256 l = d;
      for (int j = 0; j < 32; ++j) {
        words_of(&ks[(size_t)(1 + j) * 8], l);
        l = sc_add(l, l);
      }
      HIPCHK(h, hipMemcpyAsync(h->d_auxk, ks.data(), ks.size() * sizeof(u32), hipMemcpyHostToDevice, h->stream));
      hipLaunchKernelGGL(k_mul_g, dim3(1), dim3(64), 0, h->stream, h->d_auxk, h->d_aux + 16, (u8*)nullptr, 33u);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipMemcpyAsync(h->jump_host, h->d_aux + 16, 16 * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      h->aux_B = B, h->aux_T = T;
    }
    // C0 = (k0 + B*s)*G through the window table (19 additions, ~0.1 ms), the scalar travelling as a kernel argument;
    // then the lane centres C0 + g*D.  Nothing here waits for the host: the search kernel queues right behind.
    scalar_arg c0;
    words_of(c0.w, sc_add(k0, sc_mul_u64(s, B)));
    hipLaunchKernelGGL(k_mul_window_one, dim3(1), dim3(64), 0, h->stream, c0, h->d_gtab, h->d_aux);
    HIPCHK(h, hipGetLastError());
    if (B >= 8)  // the chain scratch (T * B * 36 bytes) holds the 144 bytes per lane the batched set-up parks
      hipLaunchKernelGGL(k_init_centres_batched, dim3((T / INIT_R + 255) / 256), dim3(256), 0, h->stream, h->d_aux, h->d_aux + 32,
                         h->d_cxy, T, (u32*)h->d_scr);
    else
      hipLaunchKernelGGL(k_init_centres, dim3(T / 256), dim3(256), 0, h->stream, h->d_aux, h->d_aux + 32, h->d_cxy, T);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_s1, h->stream));
    h->walk_T = T, h->walk_B = B;
  }
  add_args a;
  a.tab = h->d_tab;
  memcpy(a.jump, h->jump_host, sizeof a.jump);
  a.cxy = h->d_cxy, a.scratch = h->d_scr, a.scratch2 = h->d_scr2;
  a.bloom = bloom_make(h->d_bloom, h->bloom_words);
  a.found = h->d_found, a.counter = h->d_counter, a.cap = rcap;
  a.B = B, a.T = T, a.nb = nb, a.nkeys = nkeys;
  HIPCHK(h, hipMemsetAsync(h->d_counter, 0, 2 * sizeof(u32), h->stream));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  hipLaunchKernelGGL(pick_add_kernel(h->flags), dim3(T / ECL_ADD_BLOCK), dim3(ECL_ADD_BLOCK), 0, h->stream, a);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  u32 cnt = 0;
  rc = collect_found(h, cap, rcap, out, &cnt, true);
  if (rc != ECL_OK && rc != ECL_E_OVERFLOW) return rc;
  float ms = 0;
  HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->kernel_ms += ms, h->launches += 1, h->keys += nkeys;
  if (!cont) {
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev_s0, h->ev_s1));
    h->setup_ms += ms, h->setups += 1;
  }

  // the centres now sit at the start of group nb*T + g: valid continuation only if the launch was exact
  const u64 walked = (u64)nb * T * group;
  h->walk_valid = walked == nkeys;
  if (h->walk_valid) h->walk_next = sc_add(k0, sc_mul_u64(s, walked));
  *nout = cnt;
  return rc;
}

// ec_gtable_init (lib/ecc.c:880-905) on the device: every slot is an independent double-and-add
static int ensure_gtable(ecl_hip* h) {
  if (h->d_gtab) return ECL_OK;
  const size_t slots = (size_t)GT_WINDOWS * GT_PER;
  std::vector<u32> ks(slots * 8);
  for (u32 w = 0; w < GT_WINDOWS; ++w) {
    u256 base = sc_pow2(w * GT_W), cur = base;
    for (u32 b = 1; b <= GT_PER; ++b) {
      words_of(&ks[((size_t)w * GT_PER + b - 1) * 8], cur);
      cur = sc_add(cur, base);
    }
  }
  u32* d_k = nullptr;
  HIPCHK(h, hipMalloc(&d_k, ks.size() * sizeof(u32)));
  HIPCHK(h, hipMalloc(&h->d_gtab, slots * 16 * sizeof(u32)));
  HIPCHK(h, hipMemcpy(d_k, ks.data(), ks.size() * sizeof(u32), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mul_g, dim3((unsigned)((slots + 63) / 64)), dim3(64), 0, h->stream, d_k, h->d_gtab, (u8*)nullptr, (u32)slots);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipFree(d_k));
  return ECL_OK;
}

// `mul`'s window tables (see k_gtable_rows): built once per (device, width) and process, shared by every context on that
// device that uses the width (the host program runs two per GPU), freed with the last of them.  Before a table is handed
// out, sample slots of every row - first, last, the low digits, the seams between threads, and a fixed pseudo-random set -
// are compared with the double-and-add kernel.
#define MUL_W_MIN 8u
#define MUL_W_MAX 26u  /* 10 rows x 2^26 points: 43 GB */
#define MUL_W_START 20u               /* 13 rows x 2^20 points, 872 MB: first call 48 ms against 41 ms at 14 bits and 47 at 18 */
#define MUL_W_LONG 22u                /* 12 rows x 2^22 points, 3.0 GB: ~50 ms */
#define MUL_LONG_AFTER (1ull << 30)   /* scalars a context has seen before it moves to MUL_W_LONG: at 955 vs 1006 M scalars/s the
                                         wider table gains 0.05 ns per scalar, so its build is paid back after 10^9 of them */
struct multab_t {
  u32* d = nullptr;
  int refs = 0;
  std::mutex mu;  // held while the table is built: a context that wants the same table waits for it, one that wants another width does not
};
static std::mutex g_multab_mu;  // guards the map only (its nodes stay where they are)
static std::map<std::pair<int, u32>, multab_t> g_multab;
static multab_t* multab_entry(int dev, u32 W) {
  std::lock_guard<std::mutex> lk(g_multab_mu);
  return &g_multab[{dev, W}];
}

static void release_multable(ecl_hip* h) {
  if (!h->d_multab) return;
  multab_t* t = multab_entry(h->dev, h->multab_W);
  std::lock_guard<std::mutex> lk(t->mu);
  if (--t->refs == 0) (void)hipFree(t->d), t->d = nullptr;
  h->d_multab = nullptr, h->multab_W = 0;
}

static int build_multable(ecl_hip* h, u32 W, u32** out) {
  dbuf<u32> lad_k, lad, tmp, tab, got, want, want_k;
  dbuf<u64> slots;
  const wtab tb = wtab_make(nullptr, W);
  // ladders: 2^j * 2^(W w) * G for j < the row's digit width
  std::vector<u32> ks((size_t)tb.nwin * 32 * 8, 0);
  for (u32 w = 0; w < tb.nwin; ++w)
    for (u32 j = 0; j < W && W * w + j < 256; ++j) words_of(&ks[((size_t)w * 32 + j) * 8], sc_pow2(W * w + j));
  const u32 nlad = tb.nwin * 32;
  HIPCHK(h, hipMalloc(&lad_k.p, ks.size() * sizeof(u32)));
  HIPCHK(h, hipMalloc(&lad.p, (size_t)nlad * 16 * sizeof(u32)));
  HIPCHK(h, hipMemcpyAsync(lad_k.p, ks.data(), ks.size() * sizeof(u32), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_mul_g, dim3((nlad + 63) / 64), dim3(64), 0, h->stream, lad_k.p, lad.p, (u8*)nullptr, nlad);
  HIPCHK(h, hipGetLastError());
  // rows: launches of ~2^18 threads (one thread per 16 entries), the parking space of one launch reused by the next
  const u32 nt = (tb.per + 15u) / 16u;
  u32 rows = (1u << 18) / nt;
  rows = rows < 1 ? 1 : (rows > tb.nwin ? tb.nwin : rows);
  HIPCHK(h, hipMalloc(&tmp.p, (size_t)rows * 16 * 36 * nt * sizeof(u32)));
  HIPCHK(h, hipMalloc(&tab.p, wtab_slots(tb) * 16 * sizeof(u32)));
  for (u32 w0 = 0; w0 < tb.nwin; w0 += rows) {
    const u32 ny = tb.nwin - w0 < rows ? tb.nwin - w0 : rows;
    hipLaunchKernelGGL(k_gtable_rows, dim3((nt + 255) / 256, ny), dim3(256), 0, h->stream, lad.p, tab.p, tmp.p, nt, W, w0);
  }
  HIPCHK(h, hipGetLastError());
  // the check
  const u32 PERW = 48;
  std::vector<u64> sl;
  std::vector<u32> wk;
  u64 z = 0xD1B54A32D192ED03ull;
  for (u32 w = 0; w < tb.nwin; ++w) {
    const u32 count = w == tb.nwin - 1u ? tb.top_per : tb.per;
    for (u32 i = 0; i < PERW; ++i) {
      z ^= z << 13, z ^= z >> 7, z ^= z << 17;
      const u32 b = i == 0 ? 1u : i == 1 ? count : i < 18 ? (i - 1u) : i < 34 ? (i - 17u) * 16u + (i & 1u) : (u32)(z % count) + 1u;
      const u32 digit = b > count ? count : b;
      sl.push_back((u64)w * tb.per + digit - 1);
      u32 kw[8];
      words_of(kw, sc_mul_u64(sc_pow2(W * w), digit));
      wk.insert(wk.end(), kw, kw + 8);
    }
  }
  const u32 ns = (u32)sl.size();
  HIPCHK(h, hipMalloc(&slots.p, (size_t)ns * 8));
  HIPCHK(h, hipMalloc(&got.p, (size_t)ns * 64));
  HIPCHK(h, hipMalloc(&want.p, (size_t)ns * 64));
  HIPCHK(h, hipMalloc(&want_k.p, (size_t)ns * 32));
  HIPCHK(h, hipMemcpyAsync(slots.p, sl.data(), (size_t)ns * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(want_k.p, wk.data(), (size_t)ns * 32, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_gather_slots, dim3((ns + 63) / 64), dim3(64), 0, h->stream, tab.p, slots.p, got.p, ns);
  hipLaunchKernelGGL(k_mul_g, dim3((ns + 63) / 64), dim3(64), 0, h->stream, want_k.p, want.p, (u8*)nullptr, ns);
  HIPCHK(h, hipGetLastError());
  std::vector<u32> a((size_t)ns * 16), b((size_t)ns * 16);
  HIPCHK(h, hipMemcpyAsync(a.data(), got.p, a.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(b.data(), want.p, b.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (a != b) {
    h->err = "mul window table disagrees with the double-and-add kernel";
    return ECL_E_SELFTEST;
  }
  *out = tab.p, tab.p = nullptr;
  return ECL_OK;
}

// the table of width W for this context.  The new table is acquired (built if nobody has it yet) BEFORE the one the context holds is
// given back: a failed switch - no room for the wider table - leaves the context with the table it had, nothing to rebuild.
static int ensure_multable(ecl_hip* h, u32 W) {
  if (h->d_multab && h->multab_W == W) return ECL_OK;
  multab_t* t = multab_entry(h->dev, W);
  {
    std::lock_guard<std::mutex> lk(t->mu);
    if (!t->d) {
      const int rc = build_multable(h, W, &t->d);
      if (rc != ECL_OK) return rc;
    }
    ++t->refs;
  }
  if (h->d_multab) {
    HIPCHK(h, hipStreamSynchronize(h->stream));  // kernels of earlier calls may still read the old table
    release_multable(h);
  }
  h->d_multab = t->d, h->multab_W = W;
  return ECL_OK;
}

extern "C" int ecl_hip_set_mul_window(ecl_hip* h, uint32_t bits) {
  if (!h || (bits != 0 && (bits < MUL_W_MIN || bits > MUL_W_MAX))) return ECL_E_ARG;
  h->mul_W_fixed = bits;
  return ECL_OK;
}
extern "C" int ecl_hip_get_mul_window(ecl_hip* h, uint32_t* bits) {
  if (!h || !bits) return ECL_E_ARG;
  *bits = h->multab_W;
  return ECL_OK;
}

// what a mul_batch of n scalars needs before its first copy: the window table of the width in force, the copy stream and
// its events, the device staging for one chunk (x2: the copy engine runs one chunk ahead of the kernel) and the parking
// space of one chunk - sized to the call, grown on demand
static int mul_setup(ecl_hip* h, u32 n, u32 W) {
  int rc;
  if ((rc = ensure_multable(h, W)) != ECL_OK) return rc;
  if (!h->copy_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_free[i], hipEventDisableTiming));
    }
  }
  u32 want = 1u << 16;
  while (want < MUL_CHUNK && want < n) want <<= 1;
  if (want > h->kbuf_cap) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    for (int i = 0; i < 2; ++i) {
      if (h->d_kbuf[i]) HIPCHK(h, hipFree(h->d_kbuf[i]));
      if (h->pin_k[i]) HIPCHK(h, hipHostFree(h->pin_k[i]));
      h->d_kbuf[i] = nullptr, h->pin_k[i] = nullptr;
    }
    h->pin_cap = 0;
    if (h->d_multmp) HIPCHK(h, hipFree(h->d_multmp));
    h->d_multmp = nullptr, h->kbuf_cap = 0;
    for (int i = 0; i < 2; ++i) HIPCHK(h, hipMalloc(&h->d_kbuf[i], (size_t)want * 32));
    // the kernel indexes the planes as r * 36 * nt + plane * nt + t with R * nt = m rounded up to a multiple of R:
    // up to R - 1 slots more than m, so the buffer carries MUL_R spare slots
    HIPCHK(h, hipMalloc(&h->d_multmp, ((size_t)want + MUL_R) * 36 * sizeof(u32)));
    h->kbuf_cap = want;
  }
  return ECL_OK;
}
// window width of the next call: the caller's, or the short table until this context has seen enough scalars to pay for the long one
static u32 mul_window_for(const ecl_hip* h, u32 n) {
  return h->mul_W_fixed ? h->mul_W_fixed : (h->mul_seen + n >= MUL_LONG_AFTER && !h->mul_long_failed ? MUL_W_LONG : MUL_W_START);
}
// mul_setup at the width in force; if the automatic choice was the long table and there is no room for it (3.6 GB while it is
// built), the context stays on the short one for good.  Shared by ecl_hip_mul_batch, ecl_hip_mul_batch_raw and ecl_hip_reserve_mul.
static int mul_setup_auto(ecl_hip* h, u32 n, u32* W_used) {
  u32 W = mul_window_for(h, n);
  int rc = mul_setup(h, n, W);
  if (rc == ECL_E_HIP && !h->mul_W_fixed && W == MUL_W_LONG) {
    (void)hipGetLastError();
    h->mul_long_failed = true, W = MUL_W_START;
    rc = mul_setup(h, n, W);
  }
  *W_used = W;
  return rc;
}

extern "C" int ecl_hip_reserve_mul(ecl_hip* h, uint32_t n, uint32_t cap) {
  if (!h || n == 0) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  if ((rc = ensure_found(h, raw_cap_of(h, cap ? cap : 1) + (h->d_list ? cap : 0))) != ECL_OK) return rc;
  u32 W;
  return mul_setup_auto(h, n, &W);
}

extern "C" int ecl_hip_mul_batch(ecl_hip* h, const uint64_t (*scalars)[4], uint32_t n, ecl_found* out, uint32_t cap,
                                 uint32_t* nout) {
  if (!h || (!scalars && n) || (!out && cap) || !nout) return ECL_E_ARG;
  *nout = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, rcap + (h->d_list ? cap : 0))) != ECL_OK) return rc;
  u32 W;
  if ((rc = mul_setup_auto(h, n, &W)) != ECL_OK) return rc;
  const wtab gtab = wtab_make(h->d_multab, W);
  h->mul_seen += n;
  // Scalars in page-locked host memory (ecl_hip_alloc_host / ecl_hip_pin_host) go to the device by DMA straight from the
  // caller's array; pageable ones are first copied into two pinned staging buffers - a single-threaded memcpy that caps
  // the call near 18 GB/s = 570 M scalars/s (measured), below what the kernel takes.
  bool direct = false;
  {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof attr);
    if ((size_t)n * 32 >= ECL_PIN_MIN_BYTES) {  // small batches are staged whatever their memory is
      if (hipPointerGetAttributes(&attr, scalars) == hipSuccess) direct = attr.type == hipMemoryTypeHost;
      else (void)hipGetLastError();
    }
  }
  if (!direct)
    for (int i = 0; i < 2; ++i)
      if (!h->pin_k[i] || h->pin_cap < h->kbuf_cap) {
        if (h->pin_k[i]) HIPCHK(h, hipHostFree(h->pin_k[i]));
        h->pin_k[i] = nullptr;
        HIPCHK(h, hipHostMalloc(&h->pin_k[i], (size_t)h->kbuf_cap * 32, hipHostMallocDefault));
        if (i == 1) h->pin_cap = h->kbuf_cap;
      }
  add_args a;
  memset(&a, 0, sizeof a);
  a.bloom = bloom_make(h->d_bloom, h->bloom_words);
  a.found = h->d_found, a.counter = h->d_counter, a.cap = rcap;
  HIPCHK(h, hipMemsetAsync(h->d_counter, 0, 2 * sizeof(u32), h->stream));
  const bool a33 = h->flags & ECL_ADDR33, a65 = h->flags & ECL_ADDR65;
  // Scalars are used as given (4 little-endian u64 = 8 u32 words): the window sum (wtab_sum_lazy, any width) is k*G for any
  // 256-bit k, which is (k mod n)*G; k = 0 (mod n) gives the point at infinity and is skipped.
  // A call is cut into pieces so that the copy engine runs one piece ahead of the kernel.  Round 3 used pieces of 2^20 scalars
  // (profiles/r03_mul_pieces.txt: 848 / 970 / 994 M scalars/s on calls of 2^22 / 2^24 / 2^26; 2^22-scalar pieces 622 / 894 / 1006);
  // fewer than 2^17 threads per kernel cost more than they save.  Now the
  // pieces grow: 2^18, 2^19, ... up to the staging size (2^22): the first copy is short, and the later pieces give a thread up to
  // 32 scalars to share its inversion (270 multiplications: 34 per scalar at 8 scalars per thread, 8 at 32)
  static const u32 first_log2 = getenv("ECL_HIP_MUL_FIRST") ? (u32)atoi(getenv("ECL_HIP_MUL_FIRST")) : 18u;   // tuning hooks (A/B runs)
  static const u32 grow_pct = getenv("ECL_HIP_MUL_GROW") ? (u32)atoi(getenv("ECL_HIP_MUL_GROW")) : 200u;
  static const u32 top_log2 = getenv("ECL_HIP_MUL_TOP") ? (u32)atoi(getenv("ECL_HIP_MUL_TOP")) : 22u;
  const u32 top = h->kbuf_cap < (1u << top_log2) ? h->kbuf_cap : 1u << top_log2;
  u32 lim = top < (1u << first_log2) ? top : 1u << first_log2;
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  for (u32 at = 0, c = 0, m = 0; at < n; at += m, ++c, lim = (u64)lim * grow_pct / 100 <= top ? (u32)((u64)lim * grow_pct / 100) & ~1023u : top) {
    const u32 b = c & 1;
    m = n - at < lim ? n - at : lim;
    if (c >= 2) HIPCHK(h, hipEventSynchronize(h->ev_free[b]));  // the kernel two chunks back is done with this pair
    const void* src = scalars[at];
    if (!direct) memcpy(h->pin_k[b], scalars[at], (size_t)m * 32), src = h->pin_k[b];
    HIPCHK(h, hipMemcpyAsync(h->d_kbuf[b], src, (size_t)m * 32, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipEventRecord(h->ev_copied[b], h->copy_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_copied[b], 0));
    // scalars per thread: as many as keep >= 2^17 threads in flight (two waves per SIMD hide the table gathers; the host
    // program keeps two contexts per GPU busy, which fills the other half), at most MUL_R
    static const u32 nt_target = getenv("ECL_HIP_MUL_NT") ? (u32)atoi(getenv("ECL_HIP_MUL_NT")) : 1u << 17;  // tuning hook (A/B runs)
    u32 R = m / nt_target;
    R = R < 1 ? 1 : (R > MUL_R ? MUL_R : R);
    const u32 nt = (m + R - 1) / R;
    dim3 grid((nt + 255) / 256), blk(256);
    if (a33 && a65) hipLaunchKernelGGL((k_mul_check<true, true>), grid, blk, 0, h->stream, h->d_kbuf[b], m, at, gtab, a, h->d_multmp, nt, R);
    else if (a33) hipLaunchKernelGGL((k_mul_check<true, false>), grid, blk, 0, h->stream, h->d_kbuf[b], m, at, gtab, a, h->d_multmp, nt, R);
    else hipLaunchKernelGGL((k_mul_check<false, true>), grid, blk, 0, h->stream, h->d_kbuf[b], m, at, gtab, a, h->d_multmp, nt, R);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_free[b], h->stream));
  }
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  u32 cnt = 0;
  rc = collect_found(h, cap, rcap, out, &cnt, false);
  *nout = cnt;
  if (rc == ECL_OK || rc == ECL_E_OVERFLOW) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));  // copies + kernels of this call, as the stream saw them
    h->mul_ms += ms, h->mul_calls += 1, h->mul_scalars += n;
  }
  return rc;
}

// `mul -raw`: lines of text in, SHA-256 on the device, then the `mul` body on the digests.  One chunk per call (the caller
// cuts: n <= 2^22 lines); text and line table cross PCIe on the copy stream, hashing and the window sums follow on the
// context's stream.  Two contexts per GPU overlap one call's copies with the other's kernels, as for ecl_hip_mul_batch.
extern "C" int ecl_hip_mul_batch_raw(ecl_hip* h, const uint8_t* text, uint32_t text_bytes, const uint64_t* lines, uint32_t n, ecl_found* out,
                                     uint32_t cap, uint32_t* nout) {
  if (!h || (!text && text_bytes) || (!lines && n) || (!out && cap) || !nout || n > MUL_CHUNK || text_bytes > 0xFFFFFFF0u) return ECL_E_ARG;
  *nout = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, rcap + (h->d_list ? cap : 0))) != ECL_OK) return rc;
  u32 W;
  if ((rc = mul_setup_auto(h, n, &W)) != ECL_OK) return rc;
  const wtab gtab = wtab_make(h->d_multab, W);
  h->mul_seen += n;
  const size_t text_words = ((size_t)text_bytes + 3) / 4 + 2;  // two spare words: the gather reads one word past the last byte
  if (text_words > h->rawtext_cap || n > h->rawlines_cap) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    if (text_words > h->rawtext_cap) {
      if (h->d_rawtext) HIPCHK(h, hipFree(h->d_rawtext));
      h->d_rawtext = nullptr, h->rawtext_cap = 0;
      size_t capw = (size_t)1 << 22;  // 16 MB of text
      while (capw < text_words) capw <<= 1;
      HIPCHK(h, hipMalloc(&h->d_rawtext, capw * 4));
      // (hipMemset is asynchronous to the host and runs on the legacy stream, which the context's non-blocking streams do not
      // wait for: the clearing goes on the copy stream, in front of the text that is copied there next)
      HIPCHK(h, hipMemsetAsync(h->d_rawtext, 0, capw * 4, h->copy_stream));
      h->rawtext_cap = capw;
    }
    if (n > h->rawlines_cap) {
      if (h->d_rawlines) HIPCHK(h, hipFree(h->d_rawlines));
      h->d_rawlines = nullptr, h->rawlines_cap = 0;
      u32 capl = 1u << 20;
      while (capl < n) capl <<= 1;
      HIPCHK(h, hipMalloc(&h->d_rawlines, (size_t)capl * 8));
      h->rawlines_cap = capl;
    }
  }
  add_args a;
  memset(&a, 0, sizeof a);
  a.bloom = bloom_make(h->d_bloom, h->bloom_words);
  a.found = h->d_found, a.counter = h->d_counter, a.cap = rcap;
  HIPCHK(h, hipMemsetAsync(h->d_counter, 0, 3 * sizeof(u32), h->stream));
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_rawtext, text, text_bytes, hipMemcpyHostToDevice, h->copy_stream));
  HIPCHK(h, hipMemcpyAsync(h->d_rawlines, lines, (size_t)n * 8, hipMemcpyHostToDevice, h->copy_stream));
  HIPCHK(h, hipEventRecord(h->ev_copied[0], h->copy_stream));
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_copied[0], 0));
  hipLaunchKernelGGL(k_raw_scalars, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_rawtext, text_bytes, h->d_rawlines, n, h->d_kbuf[0], h->d_counter + 2);
  u32 R = n >> 17;
  R = R < 1 ? 1 : (R > MUL_R ? MUL_R : R);
  const u32 nt = (n + R - 1) / R;
  dim3 grid((nt + 255) / 256), blk(256);
  const bool a33 = h->flags & ECL_ADDR33, a65 = h->flags & ECL_ADDR65;
  if (a33 && a65) hipLaunchKernelGGL((k_mul_check<true, true>), grid, blk, 0, h->stream, h->d_kbuf[0], n, 0u, gtab, a, h->d_multmp, nt, R);
  else if (a33) hipLaunchKernelGGL((k_mul_check<true, false>), grid, blk, 0, h->stream, h->d_kbuf[0], n, 0u, gtab, a, h->d_multmp, nt, R);
  else hipLaunchKernelGGL((k_mul_check<false, true>), grid, blk, 0, h->stream, h->d_kbuf[0], n, 0u, gtab, a, h->d_multmp, nt, R);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  u32 cnt = 0;
  rc = collect_found(h, cap, rcap, out, &cnt, false);
  *nout = cnt;
  u32 bad = 0;
  HIPCHK(h, hipMemcpy(&bad, h->d_counter + 2, sizeof bad, hipMemcpyDeviceToHost));
  if (bad) {
    h->err = "mul_batch_raw: a line of the table lies outside the text";
    *nout = 0;
    return ECL_E_ARG;
  }
  if (rc == ECL_OK || rc == ECL_E_OVERFLOW) {
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->mul_ms += ms, h->mul_calls += 1, h->mul_scalars += n;
  }
  return rc;
}

extern "C" int ecl_hip_verify(ecl_hip* h, const uint64_t (*k)[4], uint32_t n, uint32_t (*h33)[5], uint32_t (*h65)[5], uint8_t* ok) {
  if (!h || !k || !h33 || !h65 || !ok || n == 0 || n > (1u << 31)) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  if ((rc = ensure_gtable(h)) != ECL_OK) return rc;
  if (n > h->ver_cap) {  // grow-only device staging: scalars 32 B, two hashes 20 B each, flag
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_ver) HIPCHK(h, hipFree(h->d_ver));
    h->d_ver = nullptr, h->ver_cap = 0;
    u32 cap = 256;
    while (cap < n) cap <<= 1;
    HIPCHK(h, hipMalloc(&h->d_ver, (size_t)cap * 76));
    h->ver_cap = cap;
  }
  u8* base = (u8*)h->d_ver;
  u32* dk = (u32*)base;
  u32* d33 = (u32*)(base + (size_t)h->ver_cap * 32);
  u32* d65 = (u32*)(base + (size_t)h->ver_cap * 52);
  u8* dok = base + (size_t)h->ver_cap * 72;
  HIPCHK(h, hipMemcpyAsync(dk, k, (size_t)n * 32, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_verify, dim3((n + 63) / 64), dim3(64), 0, h->stream, dk, n, h->d_gtab, d33, d65, dok);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(h33, d33, (size_t)n * 20, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(h65, d65, (size_t)n * 20, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(ok, dok, n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return ECL_OK;
}

extern "C" int ecl_hip_get_mul_timing(ecl_hip* h, double* ms, uint64_t* calls, uint64_t* scalars) {
  if (!h) return ECL_E_ARG;
  if (ms) *ms = h->mul_ms;
  if (calls) *calls = h->mul_calls;
  if (scalars) *scalars = h->mul_scalars;
  return ECL_OK;
}

// ------------------------------------------------------------------------------------------------ diagnostics (host)

extern "C" int ecl_hip_diag_fe(ecl_hip* h, int op, const uint64_t (*a)[4], const uint64_t (*b)[4], uint64_t (*r)[4],
                               uint32_t n) {
  if (!h || !a || !r || n == 0 || op < 0 || op > 8) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  size_t bytes = (size_t)n * 32;
  dbuf<u32> da, db, dr;
  HIPCHK(h, hipMalloc(&da.p, bytes));
  HIPCHK(h, hipMalloc(&db.p, bytes));
  HIPCHK(h, hipMalloc(&dr.p, bytes));
  HIPCHK(h, hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));  // little-endian u64 limbs == u32 word pairs
  HIPCHK(h, hipMemcpy(db.p, b ? b : a, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_fe, dim3((n + 63) / 64), dim3(64), 0, h->stream, op, da.p, db.p, dr.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(r, dr.p, bytes, hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_diag_mulg(ecl_hip* h, const uint64_t (*k)[4], uint64_t (*x)[4], uint64_t (*y)[4], uint8_t* ok,
                                 uint32_t n) {
  if (!h || !k || !x || !y || n == 0) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u32> dk, dout;
  dbuf<u8> dok;
  HIPCHK(h, hipMalloc(&dk.p, (size_t)n * 32));
  HIPCHK(h, hipMalloc(&dout.p, (size_t)n * 64));
  HIPCHK(h, hipMalloc(&dok.p, n));
  HIPCHK(h, hipMemcpy(dk.p, k, (size_t)n * 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mul_g, dim3((n + 63) / 64), dim3(64), 0, h->stream, dk.p, dout.p, dok.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  std::vector<u32> o((size_t)n * 16);
  std::vector<u8> okv(n);
  HIPCHK(h, hipMemcpy(o.data(), dout.p, (size_t)n * 64, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(okv.data(), dok.p, n, hipMemcpyDeviceToHost));
  for (u32 i = 0; i < n; ++i) {
    memcpy(x[i], &o[(size_t)i * 16], 32);
    memcpy(y[i], &o[(size_t)i * 16 + 8], 32);
    if (ok) ok[i] = okv[i];
  }
  return ECL_OK;
}

extern "C" int ecl_hip_diag_hash160(ecl_hip* h, const uint64_t (*x)[4], const uint64_t (*y)[4], uint32_t (*h33)[5],
                                    uint32_t (*h65)[5], uint32_t n) {
  if (!h || !x || !y || !h33 || !h65 || n == 0) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u32> dx, dy, d33, d65;
  HIPCHK(h, hipMalloc(&dx.p, (size_t)n * 32));
  HIPCHK(h, hipMalloc(&dy.p, (size_t)n * 32));
  HIPCHK(h, hipMalloc(&d33.p, (size_t)n * 20));
  HIPCHK(h, hipMalloc(&d65.p, (size_t)n * 20));
  HIPCHK(h, hipMemcpy(dx.p, x, (size_t)n * 32, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dy.p, y, (size_t)n * 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_hash, dim3((n + 63) / 64), dim3(64), 0, h->stream, dx.p, dy.p, d33.p, d65.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h33, d33.p, (size_t)n * 20, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(h65, d65.p, (size_t)n * 20, hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_bloom_insert(ecl_hip* h, const uint32_t (*h160)[5], uint64_t n) {
  if (!h || (!h160 && n)) return ECL_E_ARG;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  const u64 chunk = 1ull << 24;  // 320 MB of hashes per upload
  dbuf<u32> dh;
  HIPCHK(h, hipMalloc(&dh.p, (size_t)(n < chunk ? n : chunk) * 20));
  for (u64 at = 0; at < n; at += chunk) {
    u64 m = n - at < chunk ? n - at : chunk;
    HIPCHK(h, hipMemcpy(dh.p, h160 + at, (size_t)m * 20, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bloom_insert, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream,
                       bloom_make(h->d_bloom, h->bloom_words), h->d_bloom, dh.p, m);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return ECL_OK;
}

extern "C" int ecl_hip_bloom_insert_count(ecl_hip* h, const uint32_t (*h160)[5], uint64_t n, uint64_t* added) {
  if (!h || (!h160 && n) || !added) return ECL_E_ARG;
  *added = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  if (h->bloom_words >= (1ull << (64 - BLF_CHUNK_LOG2 - 6))) return ECL_E_ARG;  // bit position must fit 44 bits (2 TB filter)
  HIPCHK(h, hipSetDevice(h->dev));
  const u64 chunk = 1ull << BLF_CHUNK_LOG2;
  dbuf<u32> dh;
  dbuf<u64> tab;
  dbuf<unsigned long long> cnt;
  HIPCHK(h, hipMalloc(&dh.p, (size_t)(n < chunk ? n : chunk) * 20));
  HIPCHK(h, hipMalloc(&tab.p, sizeof(u64) << BLF_TAB_LOG2));
  HIPCHK(h, hipMalloc(&cnt.p, sizeof(unsigned long long)));
  HIPCHK(h, hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), h->stream));
  const bloom_t b = bloom_make(h->d_bloom, h->bloom_words);
  for (u64 at = 0; at < n; at += chunk) {
    const u32 m = (u32)(n - at < chunk ? n - at : chunk);
    HIPCHK(h, hipMemcpyAsync(dh.p, h160 + at, (size_t)m * 20, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(tab.p, 0xFF, sizeof(u64) << BLF_TAB_LOG2, h->stream));
    hipLaunchKernelGGL(k_blf_claim, dim3((m + 255) / 256), dim3(256), 0, h->stream, b, dh.p, m, tab.p);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(k_blf_count_and_set, dim3((m + 255) / 256), dim3(256), 0, h->stream, b, h->d_bloom, dh.p, m, tab.p, cnt.p);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(k_bloom_insert, dim3((m + 255) / 256), dim3(256), 0, h->stream, b, h->d_bloom, dh.p, (u64)m);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));  // the host buffer slice is free again; next chunk sees these bits
  }
  unsigned long long c = 0;
  HIPCHK(h, hipMemcpy(&c, cnt.p, sizeof c, hipMemcpyDeviceToHost));
  *added = c;
  return ECL_OK;
}

extern "C" int ecl_hip_get_bloom(ecl_hip* h, uint64_t* bits, uint64_t nwords) {
  if (!h || !bits) return ECL_E_ARG;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (nwords != h->bloom_words) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(bits, h->d_bloom, nwords * sizeof(u64), hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_diag_bloom(ecl_hip* h, const uint32_t (*h160)[5], uint8_t* hit, uint32_t n) {
  if (!h || !h160 || !hit || n == 0) return ECL_E_ARG;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u32> dh;
  dbuf<u8> dhit;
  HIPCHK(h, hipMalloc(&dh.p, (size_t)n * 20));
  HIPCHK(h, hipMalloc(&dhit.p, n));
  HIPCHK(h, hipMemcpy(dh.p, h160, (size_t)n * 20, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_bloom, dim3((n + 63) / 64), dim3(64), 0, h->stream, bloom_make(h->d_bloom, h->bloom_words),
                     dh.p, dhit.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(hit, dhit.p, n, hipMemcpyDeviceToHost));
  return ECL_OK;
}

extern "C" int ecl_hip_diag_bloom_mod(ecl_hip* h, uint64_t nwords, const uint64_t* x, uint64_t* r, uint32_t n) {
  if (!h || !x || !r || n == 0 || nwords == 0 || nwords >= (1ull << 58)) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  dbuf<u64> dx, dr;
  HIPCHK(h, hipMalloc(&dx.p, (size_t)n * 8));
  HIPCHK(h, hipMalloc(&dr.p, (size_t)n * 8));
  HIPCHK(h, hipMemcpy(dx.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_diag_bloom_mod, dim3((n + 63) / 64), dim3(64), 0, h->stream, bloom_make(nullptr, nwords), dx.p, dr.p, n);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(r, dr.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return ECL_OK;
}

// ------------------------------------------------------------------------------------------------ self-test

extern "C" int ecl_hip_selftest(ecl_hip* h) {
  if (!h) return ECL_E_ARG;
  // (1) known answers: hash160 of k*G for k = 1, 2, 0xdc2a04 (compressed, uncompressed), public vectors
  static const uint64_t KS[3][4] = {{1, 0, 0, 0}, {2, 0, 0, 0}, {0xdc2a04, 0, 0, 0}};
  static const uint32_t KAT33[3][5] = {{0x751e76e8u, 0x199196d4u, 0x54941c45u, 0xd1b3a323u, 0xf1433bd6u},
                                       {112186475u, 3455918831u, 2494304810u, 2703172626u, 1151565516u},
                                       {156887041u, 569600746u, 330545875u, 1640062380u, 639147567u}};
  static const uint32_t KAT65[3][5] = {{0x91b24bf9u, 0xf5288532u, 0x960ac687u, 0xabb03512u, 0x7b1d28a5u},
                                       {3603490856u, 3253510587u, 2691031480u, 1042137763u, 1849195074u},
                                       {3514751675u, 162192179u, 1444810732u, 2475417333u, 3394525481u}};
  uint64_t x[3][4], y[3][4];
  uint8_t ok[3];
  uint32_t h33[3][5], h65[3][5];
  int rc = ecl_hip_diag_mulg(h, KS, x, y, ok, 3);
  if (rc == ECL_OK) rc = ecl_hip_diag_hash160(h, x, y, h33, h65, 3);
  if (rc != ECL_OK) return rc;
  if (memcmp(h33, KAT33, sizeof KAT33) != 0 || memcmp(h65, KAT65, sizeof KAT65) != 0 || !(ok[0] && ok[1] && ok[2])) {
    h->err = "known-answer test of k*G -> hash160 failed";
    return ECL_E_SELFTEST;
  }
  // (2) the walk kernel against the double-and-add kernel: 4096 consecutive keys through an all-ones filter
  const u32 N = 4096, saveB = h->B, saveT = h->Tmax;
  const bool saveAuto = h->B_auto;
  u64* save_bloom = h->d_bloom;
  const u64 save_words = h->bloom_words, save_list_n = h->list_n;
  u32* save_list = h->d_list;
  h->d_list = nullptr, h->list_n = 0;
  std::vector<u64> ones(64, ~0ull);
  h->d_bloom = nullptr, h->bloom_words = 0;
  h->B = 16, h->Tmax = 256, h->B_auto = false;
  const uint64_t start[4] = {0x0123456789abcdefull, 0x1f, 0, 0};
  const u32 per_key = ((h->flags & ECL_ADDR33) ? 1 : 0) + ((h->flags & ECL_ADDR65) ? 1 : 0);
  const u32 cap = N * per_key * ((h->flags & ECL_ENDO) ? 6 : 1);
  std::vector<ecl_found> recs(cap);
  u32 n = 0;
  rc = ecl_hip_set_bloom(h, ones.data(), ones.size());
  if (rc == ECL_OK) rc = ecl_hip_add_range(h, start, N, recs.data(), cap, &n);
  std::vector<uint64_t> ks((size_t)N * 4), xs((size_t)N * 4), ys((size_t)N * 4);
  std::vector<uint32_t> r33((size_t)N * 5), r65((size_t)N * 5);
  const u256 s = sc_pow2(h->offs);
  u256 cur = sc_reduce(u256_from(start));
  for (u32 i = 0; i < N; ++i) {
    memcpy(&ks[(size_t)i * 4], cur.w, 32);
    cur = sc_add(cur, s);
  }
  if (rc == ECL_OK) rc = ecl_hip_diag_mulg(h, (const uint64_t(*)[4])ks.data(), (uint64_t(*)[4])xs.data(), (uint64_t(*)[4])ys.data(), nullptr, N);
  if (rc == ECL_OK) rc = ecl_hip_diag_hash160(h, (const uint64_t(*)[4])xs.data(), (const uint64_t(*)[4])ys.data(),
                                              (uint32_t(*)[5])r33.data(), (uint32_t(*)[5])r65.data(), N);
  // restore the caller's state whatever happened
  if (h->d_bloom) (void)hipFree(h->d_bloom);
  h->d_bloom = save_bloom, h->bloom_words = save_words;
  h->d_list = save_list, h->list_n = save_list_n;
  h->B = saveB, h->Tmax = saveT, h->B_auto = saveAuto;
  if (h->d_tab) (void)hipFree(h->d_tab);
  h->d_tab = nullptr, h->tab_B = 0, h->walk_valid = false;
  h->kernel_ms = 0, h->launches = 0, h->keys = 0, h->setup_ms = 0, h->setups = 0;
  if (rc != ECL_OK) return rc;
  u32 seen = 0;
  bool good = n == cap;
  for (u32 i = 0; i < n && good; ++i) {
    const ecl_found& f = recs[i];
    if (f.key_offset >= N) { good = false; break; }
    if (f.endo != 0) continue;  // the endomorphism images are covered by the parity tests; here: the walk itself
    const uint32_t* want = f.compressed ? &r33[f.key_offset * 5] : &r65[f.key_offset * 5];
    good = memcmp(f.h160, want, 20) == 0;
    ++seen;
  }
  if (!good || seen != N * per_key) {
    h->err = "walk kernel disagrees with the double-and-add kernel";
    return ECL_E_SELFTEST;
  }
  // (3) the window-table sum (gtable_mul: ecl_hip_verify, the base centre of every non-contiguous walk, `mul`) against the
  // double-and-add kernel on full-width scalars, so that every one of the 19 windows carries a digit: the walk's base
  // centre and the verification of its hits share this function and the table, and a hit shares its high digits with
  // the base centre - (2) exercises only the low windows.  Scalars: a fixed xorshift stream, plus every digit at its
  // maximum (0x3fff in all windows) and a single top-window digit.
  {
    const u32 M = 48;
    std::vector<uint64_t> vk((size_t)M * 4), vx((size_t)M * 4), vy((size_t)M * 4);
    std::vector<uint32_t> w33((size_t)M * 5), w65((size_t)M * 5), g33((size_t)M * 5), g65((size_t)M * 5);
    std::vector<uint8_t> vok(M), gok(M);
    u64 z = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < vk.size(); ++i) {
      z ^= z << 13, z ^= z >> 7, z ^= z << 17;
      vk[i] = z;
    }
    for (int w = 0; w < 4; ++w) vk[w] = ~0ull;                 // all digits 0x3fff (the sum is (2^256 - 1) mod n times G)
    vk[4] = 0, vk[5] = 0, vk[6] = 0, vk[7] = 1ull << 60;       // window 18 only
    rc = ecl_hip_verify(h, (const uint64_t(*)[4])vk.data(), M, (uint32_t(*)[5])g33.data(), (uint32_t(*)[5])g65.data(), gok.data());
    if (rc == ECL_OK) rc = ecl_hip_diag_mulg(h, (const uint64_t(*)[4])vk.data(), (uint64_t(*)[4])vx.data(), (uint64_t(*)[4])vy.data(), vok.data(), M);
    if (rc == ECL_OK) rc = ecl_hip_diag_hash160(h, (const uint64_t(*)[4])vx.data(), (const uint64_t(*)[4])vy.data(),
                                                (uint32_t(*)[5])w33.data(), (uint32_t(*)[5])w65.data(), M);
    if (rc != ECL_OK) return rc;
    if (g33 != w33 || g65 != w65 || gok != vok) {
      h->err = "window-table scalar multiplication disagrees with the double-and-add kernel";
      return ECL_E_SELFTEST;
    }
  }
  return ECL_OK;
}
