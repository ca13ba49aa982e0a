// mul_kernels.h - the `mul` command on the device: window tables of any width, window sums, batched normalisation + hash160 + probe,
// SHA-256 of raw lines, the window-table forms of a single scalar multiplication (base centre of a walk, pk_verify_hash).
// (one translation unit: included by ecloop_hip.hip)
#pragma once
#include "add_kernel.h"
#include "ec.h"
// `mul` command body (main.c:530-534, 458-479): public key of each scalar by the fixed-base window method of
// ec_gtable_mul (lib/ecc.c:876-929: W = 14, 19 windows, table slot (2^14-1)*i + b-1 = b * 2^(14 i) * G), then
// ec_jacobi_grprdc (lib/ecc.c:695-707: ONE inversion for the whole batch), then hash + probe.
// <= 19 mixed additions of table points per scalar (64-byte gathers, the 19.9 MB table lives in L2 / Infinity Cache).
// The scalar 0 (mod n) yields no point (the reference emits garbage).
#define GT_W 14u
#ifndef MUL_NBUF
#define MUL_NBUF 4  /* staging buffers of ecl_hip_mul_batch: the copy engine runs up to MUL_NBUF - 1 pieces ahead of the kernel */
#endif
#define MUL_CHUNK (1u << 22)  /* scalars per staged chunk of ecl_hip_mul_batch (128 MB): 2^18 threads x MUL_R */
#define MUL_RAW_MAX (1u << 26) /* lines per ecl_hip_mul_batch_raw call (a 512 MB line table on the device) */
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
// cache (lib/ecc.c:876, `bench-gtable` sweeps it); on a 288 GB part W = 26 costs 19.6 GB and turns 19 additions per
// scalar into 10 (one per non-zero digit).  Same method, same results; measured on 2^26-scalar calls, -a cu (profiles/
// r04_mul_w_sweep.txt): W = 20 1048 M scalars/s, 22 1156, 24 1208-1232, 25 1221, 26 1285-1297.
// The width is a run-time property of the table (ecl_hip_set_mul_window; by default a context starts on W = 22, 1.5 GB,
// and moves to W = 26 once it has seen enough scalars to pay for the build: ecl_hip_mul_batch).
// The rows are not built by millions of double-and-add ladders but the way the walk's lane centres are: row w is
// P_w, 2 P_w, 3 P_w, ... with P_w = 2^(W w) G - the points C0 + g D of k_init_centres_batched with C0 = D = P_w -
// 44 multiplications per entry, one inversion per 16 entries; the ladder points 2^j P_w of every row come from one
// k_mul_g launch.
// Signed digits (round 4): a window's digit d in [0, 2^W) is taken as d - 2^W with a carry into the next window when it is above
// 2^(W-1), and a negative digit adds the NEGATED table point (same x, p - y): a row holds 2^(W-1) points instead of 2^W - 1, so a
// table of a given size is one bit wider - 26 bits (10 additions per scalar) in 19.6 GB, where the unsigned layout needed 43 GB.
// The last window is not recoded (its carry would have nowhere to go): its row holds 2^topbits points, topbits = 256 - W (nwin - 1).
struct wtab {
  const u32* p;  // slot stride * w + m - 1 = m * 2^(W w) * G, canonical x[8], y[8]
  u32 W, nwin, per, half, stride, top_cnt;  // bits per window, windows = ceil(256 / W), 2^W - 1, 2^(W-1), slots per row, slots of the last row
};
__host__ __device__ inline wtab wtab_make(const u32* p, u32 W) {
  wtab t;
  t.p = p, t.W = W, t.nwin = (256u + W - 1u) / W, t.per = (1u << W) - 1u;
  t.half = 1u << (W - 1u), t.stride = t.half;
  t.top_cnt = 1u << (256u - W * (t.nwin - 1u));
  return t;
}
__host__ __device__ inline size_t wtab_slots(const wtab& t) { return (size_t)(t.nwin - 1u) * t.stride + t.top_cnt; }
__host__ __device__ inline u32 wtab_row_count(const wtab& t, u32 w) { return w == t.nwin - 1u ? t.top_cnt : t.stride; }
// window w's digit with the carry of the window below: magnitude (0 = nothing to add, else slot m - 1 of row w), sign, carry out
__host__ __device__ __forceinline__ u32 wtab_recode(const wtab& t, u32 w, u32 raw, u32& carry, u32& neg) {
  const u32 v = raw + carry;  // 0 .. 2^W
  const bool n = w + 1u < t.nwin && v > t.half;
  carry = n ? 1u : 0u, neg = carry;
  return n ? t.per + 1u - v : v;
}
__device__ __forceinline__ u32 wtab_digit(const u32 kk[9], const wtab& t, u32 w) {
  const u32 bit = w * t.W, word = bit >> 5, sh = bit & 31;
  u32 lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {  // static indexing keeps kk[] in registers
    if (word == (u32)j) lo = kk[j], hi = kk[j + 1];
  }
  return (u32)((((u64)hi << 32 | lo) >> sh) & t.per);  // raw digit; kk[8] = 0: the last window is as narrow as it is
}
// The point of window w + 1 is requested before the addition of window w's (64 bytes, 16 registers held across one
// mixed addition): the gathers come from HBM / Infinity Cache and the kernel runs at two waves per SIMD, too few to hide them.
// Measured on 2^24-scalar calls, 22-bit table, four processes each: 995-1001 M scalars/s with, 980-985 without.
// (Variants that were built, measured and taken out again - the sum with the scalar in registers and per-lane states, the same work as
// two kernels, additions one operation at a time, filter tests in place: HISTORY.md and profiles/r04_mul_*.txt.)
// -y for a negative digit, magnitude 1 (the complete formulas and the first two points of a lazy sum take normalised operands)
__device__ __forceinline__ fe fe_cneg_weak(const fe& y, u32 neg) {
  const fe m = fe_neg(y, 1);
  fe r;
#pragma unroll
  for (int l = 0; l < FE_LIMBS; ++l) r.n[l] = neg ? m.n[l] : y.n[l];
  fe_normalize_weak(r);
  return r;
}
__device__ __forceinline__ jac wtab_sum(const u32 kk[9], const wtab t) {
  jac acc;
  acc.X = fe_zero(), acc.Y = fe_zero(), acc.Z = fe_one(), acc.inf = 1;
  u32 carry = 0;
#pragma unroll 1
  for (u32 w = 0; w < t.nwin; ++w) {
    u32 neg;
    const u32 m = wtab_recode(t, w, wtab_digit(kk, t, w), carry, neg);
    if (!m) continue;
    const u32* e = t.p + ((size_t)w * t.stride + m - 1) * 16;
    acc = jac_madd(acc, fe_ldw(e), fe_cneg_weak(fe_ldw(e + 8), neg));
  }
  return acc;
}
// the complete sum out of line: the fallback of a scalar whose lazy sum ended with Z = 0 (never taken by a random scalar)
__device__ __noinline__ jac wtab_sum_complete(const u32 kk[9], const wtab t) { return wtab_sum(kk, t); }
// rows w0 + blockIdx.y of the table: out[w * per + g] = (g + 1) * P_w for g < count_w, P_w = ladder[w][0]; one thread owns 16
// consecutive entries (Jacobian, parked in `tmp`, one inversion for the 16 - the scheme of k_init_centres_batched)
// (a launch covers threads t0 ... t0 + nt of each of its rows: wide rows - 2^24 threads at 29 bits - are built in slices so that the
// parking space stays at 2.4 GB instead of 38.6)
__global__ void __launch_bounds__(256) k_gtable_rows(const u32* __restrict__ ladders, u32* __restrict__ table, u32* __restrict__ tmp_all, u32 nt,
                                                      u32 W, u32 w0, u32 t0) {
  const wtab tb = wtab_make(table, W);
  const u32 w = w0 + blockIdx.y, tl = blockIdx.x * 256u + threadIdx.x;  // tl: the thread's place in this launch's parking planes
  const u32 count = wtab_row_count(tb, w);
  const u32 g0 = (t0 + tl) * 16u;
  if (tl >= nt || g0 >= count) return;
  const u32 t = tl;
  const u32* ladder = ladders + (size_t)w * 32 * 16;
  u32* out = table + (size_t)w * tb.stride * 16;
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
__device__ __forceinline__ xyzz wtab_sum_fast(const u32* __restrict__ kw, const wtab t, u32& bad);
#define MUL_R 32u  /* at most (one bit of `infmask` each); short pieces take fewer per thread so that the chip still fills (ecl_hip_mul_batch) */
#ifndef ECL_MUL_WAVES
#define ECL_MUL_WAVES 3  /* waves per SIMD the register allocator leaves room for (256-thread blocks: blocks per CU); the host side launches
                            65536 x ECL_MUL_WAVES threads per piece.  With the low-register window sum the kernel fits 168 VGPRs (one spill): three
                            waves hide the wait states between dependent multiply-adds better than two (profiles/r04_mul_fastsum.txt: 2^26-scalar
                            calls 1391 against 1370 M scalars/s, 2^24 equal); the sum with the scalar in registers needed 67 spills there */
#endif
template <bool A33, bool A65>
__global__ void __launch_bounds__(256, ECL_MUL_WAVES) k_mul_check(const u32* __restrict__ k, u32 n, u32 base, const wtab gtab, add_args a,
                                                   u32* __restrict__ tmp, u32 nt, u32 R) {
  __shared__ u32 q_mem[4][2][8 * ECL_Q_SLOTS];  // two candidate rings per wave (add_kernel.h)
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nt) return;  // nt is a multiple of 256: whole workgroups leave
  fe prod = fe_one();
  u32 infmask = 0;
  // parked per scalar: X * ZZZ, Y * ZZ, T = ZZ * ZZZ and the running product of the T's; x = X ZZZ / T, y = Y ZZ / T
#pragma unroll 1
  for (u32 r = 0; r < R; ++r) {
    const u32 i = r * nt + t;
    if (i >= n) break;
    u32 bad;
    xyzz acc = wtab_sum_fast(k + (size_t)i * 8, gtab, bad);
    acc.inf = 0;
    // a zero digit (stand-in point), or P = +-Q on the way (h = 0: only scalars that are 0 (mod n) or built around n) which leaves ZZ = 0 -
    // and a zero in the product chain would take the thread's other scalars with it: the complete sum, out of line
    if (__builtin_expect(bad || fe_is_zero(acc.ZZ), 0)) {
      u32 kk[9];
#pragma unroll
      for (int j = 0; j < 8; ++j) kk[j] = k[(size_t)i * 8 + j];
      kk[8] = 0;
      acc = xyzz_from_jac(wtab_sum_complete(kk, gtab));
    }
    infmask |= (acc.inf ? 1u : 0u) << r;
    fe tt, xs, ys, nprod;
    fe_mul_pair(tt, xs, acc.ZZ, acc.ZZZ, acc.X, acc.ZZZ);
    if (acc.inf) tt = fe_one();
    fe_mul_pair(ys, nprod, acc.Y, acc.ZZ, prod, tt);
    u32* p = tmp + (size_t)r * 36 * nt + t;
#pragma unroll
    for (int l = 0; l < FE_LIMBS; ++l) {
      p[(size_t)l * nt] = xs.n[l], p[(size_t)(9 + l) * nt] = ys.n[l];
      p[(size_t)(18 + l) * nt] = tt.n[l], p[(size_t)(27 + l) * nt] = prod.n[l];
    }
    prod = nprod;
  }
  fe inv = fe_inv(prod);
  // the filter test through the add kernel's two candidate rings per wave (add_kernel.h: survivors of probe 0 are parked in LDS and
  // finished 64 at a time): every lane of the wave walks all R rounds - a lane without a scalar (i >= n) or with the point at infinity
  // comes along with live = false and leaves the inversion chain alone - so that the rings' wave-uniform state stays uniform
  // (+1.7 % at the .blf design density against finishing every hash's test in place, profiles/r04_mul_rings.txt)
  cand_queues q;
  q.a.mem = q_mem[threadIdx.x >> 6][0], q.a.head = 0, q.a.count = 0;
  q.b.mem = q_mem[threadIdx.x >> 6][1], q.b.head = 0, q.b.count = 0;
#pragma unroll 1
  for (u32 r = R; r-- > 0;) {
    const u32 i = r * nt + t;
    const bool have = i < n;
    const u32* p = tmp + (size_t)r * 36 * nt + t;
    fe X, Y, T, pre;
#pragma unroll
    for (int l = 0; l < FE_LIMBS; ++l) {
      X.n[l] = have ? p[(size_t)l * nt] : 0u, Y.n[l] = have ? p[(size_t)(9 + l) * nt] : 0u;
      T.n[l] = have ? p[(size_t)(18 + l) * nt] : (l == 0 ? 1u : 0u), pre.n[l] = have ? p[(size_t)(27 + l) * nt] : 0u;
    }
    fe ti, ninv, x, y;
    fe_mul_pair(ti, ninv, inv, pre, inv, T);  // T = 1 for a lane without a scalar in this round
    inv = ninv;
    fe_mul_pair(x, y, X, ti, Y, ti);
    check_point<A33, A65, false>(a, &q, have && !((infmask >> r) & 1u), x, y, (u64)base + i);
  }
  cand_flush(a, q);
}
// ---- k_mul_check's window sum (round 4): written for a small register budget (three waves per SIMD) --------
// * the scalar stays in memory: a digit is one 8-byte load at the digit's word (L2 / L1 hits after the first window: a wave's scalars
//   are 2 KiB of contiguous memory) + a shift, not a 16-way select over eight registers, and is fetched one window ahead;
// * a table point's 16 words are requested after the two multiplications that consume the previous point (u2 = qx ZZ, s2 = qy ZZZ),
//   so the 18 limbs of one point and the 16 words of the next are never live together;
// * no per-lane control flow: windows 0 and 1 are added as affine + affine, every later one by the mixed addition; a zero digit
//   (2^-26 per window for a random scalar) takes slot 0 of its row instead and flags the scalar, which the caller then sends through
//   the complete sum (wtab_sum_complete) like one whose chain degenerated (ZZ = 0) - but the high windows in which NO lane of the
//   wave has a digit (small scalars) are cut off the loop (round 5; before, such input took the complete sum for every scalar).
__device__ __forceinline__ u32 wtab_digit_mem(const u32* __restrict__ kw, const wtab& t, u32 w) {
  u32 bit = w * t.W, word = bit >> 5, sh = bit & 31u;
  if (word > 6u) word = 6u, sh += 32u;  // the last words: shift further instead of reading past the scalar
  u64 v;
  __builtin_memcpy(&v, kw + word, 8);  // an 8-byte load from a 4-byte aligned address (odd `word`): one global_load_dwordx2 on gfx950
  return (u32)(v >> sh) & t.per;  // raw digit
}
#ifndef ECL_MUL_HOT_GATHERS
#define ECL_MUL_HOT_GATHERS 0  /* MEASUREMENT BUILD ONLY (wrong points): every gather lands in the first 256 slots of its row, i.e. in cache - the same
                                  instructions and loads without the table's HBM traffic (tools/ab_r05b.sh, tools/ab_r05c.sh) */
#endif
#if ECL_MUL_HOT_GATHERS && !defined(ECL_MEASUREMENT_BUILD_WRONG_RESULTS)
#error "ECL_MUL_HOT_GATHERS computes wrong points: measurement builds only, and they say so with -DECL_MEASUREMENT_BUILD_WRONG_RESULTS"
#endif
// How the table points reach the additions, and what was measured about it in round 5 (profiles/r05_mul_lds_gather.txt, commit be22f1e):
// the point of window w + 1 is requested in the middle of window w's addition and held in 16 VGPRs.  Three deeper forms were built - the
// point staged in LDS by global_load_lds_dwordx4 (no register held), the wave's scalars staged in LDS a round ahead (digits by ds_read
// instead of a global load four instructions before its use), the first two points of the next scalar requested during the last addition
// of this one - and all ran within 0.5 % of this form (1272-1279 M scalars/s on 2^24-scalar calls): at three waves per SIMD another wave
// issues while one waits.  The build with cache-resident gathers is 11-13 % faster, but through the CLOCK (2.26 against 2.06 GHz; wait
// share and clocks per instruction unchanged): the kernel runs against the board's power limit, and 0.9 TB/s of random 64-byte HBM
// reads take watts from the shader clock.  Fewer HBM bytes per scalar would help; hiding their latency does not.
__device__ __forceinline__ xyzz wtab_sum_fast(const u32* __restrict__ kw, const wtab t, u32& bad) {
  // Windows that hold a zero digit in EVERY lane of the wave are not walked at all: the loop ends at the highest window in which some
  // lane has something to add (round 5).  That is what small scalars need - puzzle-range or sequential keys: all of their high windows,
  // which the reference skips one by one (lib/ecc.c:913) and which used to send every such scalar through the complete sum - and it
  // costs the loop body nothing: the bound is wave-uniform, found from the scalars' bit lengths (k < 2^(j W - 1) leaves windows >= j
  // without a digit and without a carry into them).  A zero digit below that bound still takes the stand-in point and flags the scalar.
  u32 nw = t.nwin;
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const uint4 k0 = ((const uint4*)kw)[0], k1 = ((const uint4*)kw)[1];  // (the scalar's own cache line: the digit loads below hit it)
    const u32 ws[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    u32 bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) bits = ws[j] ? 32u * (u32)j + 32u - (u32)__builtin_clz(ws[j]) : bits;
    while (nw > 2u && __builtin_amdgcn_ballot_w64(bits >= (nw - 1u) * t.W) == 0ull) --nw;
  }
#endif
  u32 carry = 0, s0, s1, sn = 0;
  u32 d0 = wtab_recode(t, 0, wtab_digit_mem(kw, t, 0), carry, s0), d1 = wtab_recode(t, 1, wtab_digit_mem(kw, t, 1), carry, s1);
  u32 dn = nw > 2u ? wtab_recode(t, 2, wtab_digit_mem(kw, t, 2), carry, sn) : 1u;
  bad = (d0 == 0u) | (d1 == 0u);
  d0 = d0 ? d0 : 1u, d1 = d1 ? d1 : 1u;
  if (ECL_MUL_HOT_GATHERS) d0 = (d0 & 255u) + 1u, d1 = (d1 & 255u) + 1u;
  xyzz acc;
  {
    const uint4* e0 = (const uint4*)(t.p + ((size_t)d0 - 1) * 16);
    const uint4* e1 = (const uint4*)(t.p + ((size_t)t.stride + d1 - 1) * 16);
    const uint4 a0 = e0[0], a1 = e0[1], a2 = e0[2], a3 = e0[3], b0 = e1[0], b1 = e1[1], b2 = e1[2], b3 = e1[3];
    const u32 pxw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, pyw[8] = {a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
    const u32 qxw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, qyw[8] = {b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    acc = xyzz_mmadd_lazy(fe_from_words(pxw), fe_cneg_weak(fe_from_words(pyw), s0), fe_from_words(qxw), fe_cneg_weak(fe_from_words(qyw), s1));
  }
  uint4 n0 = make_uint4(0, 0, 0, 0), n1 = n0, n2 = n0, n3 = n0;
  if (nw > 2u) {
    bad |= dn == 0u;
    dn = dn ? dn : 1u;
    if (ECL_MUL_HOT_GATHERS) dn = (dn & 255u) + 1u;
    const uint4* e = (const uint4*)(t.p + ((size_t)2 * t.stride + dn - 1) * 16);
    n0 = e[0], n1 = e[1], n2 = e[2], n3 = e[3];
  }
#pragma unroll 1
  for (u32 w = 2; w < nw; ++w) {
    const u32 xw[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w}, yw[8] = {n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
    const bool more = w + 1u < nw;
    const u32 neg = sn;
    dn = more ? wtab_recode(t, w + 1u, wtab_digit_mem(kw, t, w + 1u), carry, sn) : 1u;
    fe u2, s2p;  // (the ten operations of an addition as five interleaved pairs: +0.5-1 %, profiles/r04_mul_pairs.txt)
    fe_mul_pair(u2, s2p, fe_from_words(xw), acc.ZZ, fe_from_words(yw), acc.ZZZ);
    const fe s2m = fe_neg(s2p, 1);
    if (more) {
      bad |= dn == 0u;
      dn = dn ? dn : 1u;
      if (ECL_MUL_HOT_GATHERS) dn = (dn & 255u) + 1u;
      const uint4* e = (const uint4*)(t.p + ((size_t)(w + 1u) * t.stride + dn - 1) * 16);
      n0 = e[0], n1 = e[1], n2 = e[2], n3 = e[3];
    }
    fe s2;
#pragma unroll
    for (int l = 0; l < FE_LIMBS; ++l) s2.n[l] = neg ? s2m.n[l] : s2p.n[l];
    fe h = fe_sub(u2, acc.X);
    fe_normalize_weak(h);
    fe rr = fe_add(s2, fe_neg(acc.Y, 3));
    fe_normalize_weak(rr);
    fe hh, rr2, hhh, v, t1, t2;
    fe_sqr_pair(hh, rr2, h, rr);
    fe_mul_pair(hhh, v, hh, h, acc.X, hh);
    fe X3 = fe_add(fe_add(rr2, fe_neg(hhh, 1)), fe_neg(fe_add(v, v), 2));
    fe_normalize_weak(X3);
    fe_mul_pair(t1, t2, rr, fe_sub(v, X3), acc.Y, hhh);
    acc.Y = fe_add(t1, fe_neg(t2, 1));
    acc.X = X3;
    fe_mul_pair(acc.ZZ, acc.ZZZ, acc.ZZ, hh, acc.ZZZ, hhh);
  }
  return acc;
}
// `mul -raw` (main.c:505-527): the scalar of a line is the SHA-256 of its bytes.  One lane per line: the line's bytes are
// gathered from the text (any alignment: two aligned words and a funnel shift per message word), padded per FIPS 180-4 and
// compressed block by block; the digest, read as a big-endian 256-bit number, is written where k_mul_check expects the
// scalar (8 little-endian words).  lines[i] = start | length << 32, offsets into `text`; `text` carries 8 spare bytes.
// text_have <= text_bytes: the part of the text that is on the device when this launch runs (the host sends the text along with the
// pieces of the table); a line beyond it raises bad[1] - the host then repeats the call with all of the text sent first.
__global__ void __launch_bounds__(256) k_raw_scalars(const u32* __restrict__ text, u32 text_bytes, u32 text_have, const u64* __restrict__ lines, u32 n,
                                                      u32* __restrict__ out, u32* __restrict__ bad) {
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const u64 ln = lines[i];
  const u32 start = (u32)ln;
  u32 L = (u32)(ln >> 32);
  if ((u64)start + L > text_bytes) bad[0] = 1, L = 0;  // a line outside the text: the call is refused (ECL_E_ARG), nothing is read there
  else if (start + L > text_have) bad[1] = 1, L = 0;    // not here yet
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
