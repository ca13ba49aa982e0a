// bloom.h — probe of the reference's bloom filter (lib/utils.c:274-326), bit-compatible with `.blf` files.
//
// 20 probe positions per hash160: five overlapping 64-bit words a1..a5 of the hash, for S in {24,28,36,40}
// and j = 1..5: idx = a_j << S | a_{j+1} >> S (a6 = a1); bit (idx mod 64) of word ((idx mod 64*size) / 64)
// = word ((idx >> 6) mod size).  `size` is arbitrary (not a power of two), so the modulo uses a host
// precomputed reciprocal (one 64x64 high multiply + one low multiply) instead of a 64-bit division.
// Probe order (S outer, j inner) and the early-out on the first zero bit follow lib/utils.c:308-326.
#pragma once
#include "fe256.h"

struct bloom_t {
  const u64* bits;
  u64 nwords;
  u64 recip;  // floor(2^64 / nwords) for nwords >= 2 (unused for nwords == 1)
};

FE_FN u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}
__host__ __device__ inline bloom_t bloom_make(const u64* bits, u64 nwords) {
  bloom_t b;
  b.bits = bits, b.nwords = nwords;
  b.recip = nwords >= 2 ? (u64)((((unsigned __int128)1) << 64) / nwords) : 0;
  return b;
}
// x mod nwords for x < 2^58.  q = floor(x * recip / 2^64) is floor(x / d) or one less (x < 2^58, so the error of the
// truncated reciprocal stays below 2^-6): one conditional subtraction finishes it.
FE_FN u64 bloom_mod(const bloom_t& b, u64 x) {
  if (b.nwords < (1ull << 31)) {
    // filters below 16 GB: the remainder and 2d fit 32 bits, so only the LOW word of q is needed
    const u32 d = (u32)b.nwords, xl = (u32)x, xh = (u32)(x >> 32), ml = (u32)b.recip, mh = (u32)(b.recip >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 t0 = __umulhi(xl, ml);
#else
    const u32 t0 = (u32)(((u64)xl * ml) >> 32);
#endif
    const u64 mid = (u64)xl * mh + ((u64)xh * ml + t0);  // mod 2^64: a lost carry only reaches bit 32 of q
    const u32 q = xh * mh + (u32)(mid >> 32);
    u32 r = xl - q * d;
    const u32 r2 = r - d;
    r = r2 < r ? r2 : r;  // r >= d  <=>  r - d does not wrap
    return d == 1 ? 0 : r;
  }
  u64 q = mulhi64(x, b.recip);
  u64 r = x - q * b.nwords;
  if (r >= b.nwords) r -= b.nwords;
  return r;
}
FE_FN u64 bloom_index(const u64 a[5], int probe) {
  const int S = probe < 5 ? 24 : probe < 10 ? 28 : probe < 15 ? 36 : 40;
  const int j = probe % 5;
  return a[j] << S | a[(j + 1) % 5] >> S;
}
// ECL_PROBE_NT (A/B builds): the probe's 8-byte load marked non-temporal (bypasses the per-CU L1, where a random probe
// into a filter of tens of MB never hits anyway)
#ifndef ECL_PROBE_NT
#define ECL_PROBE_NT 0
#endif
FE_FN u64 bloom_word(const bloom_t& b, u64 w) {
#if defined(__HIPCC__) && ECL_PROBE_NT
  return __builtin_nontemporal_load(b.bits + w);
#else
  return b.bits[w];
#endif
}
FE_FN bool bloom_bit(const bloom_t& b, u64 idx) { return (bloom_word(b, bloom_mod(b, idx >> 6)) >> (idx & 63)) & 1; }

// lib/utils.c:308-326 in stages.  Stage 1: probe 0 alone (at the `.blf` design density 0.375 it rejects 62 % of the
// hashes; ECL_STAGE1_PROBES = 2 would issue probes 0 and 1 together and reject 86 %).  Then the remaining probes, one
// at a time with the reference's early-out (bloom_stage2); bloom_has() = both.  The add kernel parks the survivors of
// stage 1 in per-wave rings and finishes them 64 at a time in two steps - bloom_mid (one or two probes, no loop),
// then bloom_probes_from (the early-out loop) - because inside the hot loop a few surviving lanes would keep the whole
// wave iterating (add_kernel.h: cand_queues).  Measured on MI355X, addr33 over 2^32 keys, 1 vs 2 probes in stage 1:
// 54 MB filter 11.95 vs 11.95 Gkeys/s, 5.9 GB filter 11.09 vs 10.69 - a multi-GB filter is bound by the number of
// random HBM sectors touched, and one probe first touches 1.59 per hash instead of 2.22.
FE_FN void bloom_words_of(u64 a[5], const u32 h[5]) {
  a[0] = (u64)h[0] << 32 | h[1];
  a[1] = (u64)h[2] << 32 | h[3];
  a[2] = (u64)h[4] << 32 | h[0];
  a[3] = (u64)h[1] << 32 | h[2];
  a[4] = (u64)h[3] << 32 | h[4];
}
#ifndef ECL_STAGE1_PROBES
#define ECL_STAGE1_PROBES 1
#endif
// (Issuing this probe as four instructions of 16 active lanes for multi-GB filters was built and rejected in round 4: -1.3 %, HISTORY.md.)
FE_FN bool bloom_stage1(const bloom_t& b, const u32 h[5]) {
  u64 a[5];
  bloom_words_of(a, h);
  bool p0 = bloom_bit(b, bloom_index(a, 0));
#if ECL_STAGE1_PROBES == 2
  bool p1 = bloom_bit(b, bloom_index(a, 1));
  return p0 && p1;
#else
  return p0;
#endif
}
// probes [from, 20) one at a time with the reference's early-out
FE_FN bool bloom_probes_from(const bloom_t& b, const u32 h[5], int from) {
  u64 a[5];
  bloom_words_of(a, h);
#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    const int S = s == 0 ? 24 : s == 1 ? 28 : s == 2 ? 36 : 40;
#pragma unroll
    for (int j = 0; j < 5; ++j) {  // unrolled: a[] must stay in registers (no runtime indexing)
      if (s * 5 + j < from) continue;
      u64 idx = a[j] << S | a[(j + 1) % 5] >> S;
      if (!bloom_bit(b, idx)) return false;
    }
  }
  return true;
}
FE_FN bool bloom_stage2(const bloom_t& b, const u32 h[5]) { return bloom_probes_from(b, h, ECL_STAGE1_PROBES); }
// the middle stage of the add kernel's two-level candidate queue: probe ECL_STAGE1_PROBES, and with `two` also the
// next one, issued together (independent loads, no loop)
FE_FN bool bloom_mid(const bloom_t& b, const u32 h[5], bool two) {
  u64 a[5];
  bloom_words_of(a, h);
  bool p = bloom_bit(b, bloom_index(a, ECL_STAGE1_PROBES));
  if (two) p = bloom_bit(b, bloom_index(a, ECL_STAGE1_PROBES + 1)) && p;
  return p;
}
// Filters that stay in the 256 MB Infinity Cache take two probes in the middle stage (fewer instructions: 5 % of the
// candidates reach the loop instead of 14 %); bigger ones take one (every probe is a random HBM sector + TLB miss, and
// one-at-a-time touches 1.59 sectors per hash instead of 1.79).
#ifdef ECL_MID_TWO_FORCE /* A/B builds: 0 / 1 fixes the choice whatever the filter size */
FE_FN bool bloom_mid_two(const bloom_t&) { return ECL_MID_TWO_FORCE != 0; }
#else
FE_FN bool bloom_mid_two(const bloom_t& b) { return b.nwords < (1ull << 24); }
#endif
FE_FN bool bloom_has(const bloom_t& b, const u32 h[5]) { return bloom_stage1(b, h) && bloom_stage2(b, h); }
#if defined(__HIPCC__)
// lib/utils.c:290-306 (blf_add) for one hash: 20 atomic ORs
__device__ __forceinline__ void bloom_add(const bloom_t& b, u64* bits, const u32 h[5]) {
  u64 a[5];
  a[0] = (u64)h[0] << 32 | h[1];
  a[1] = (u64)h[2] << 32 | h[3];
  a[2] = (u64)h[4] << 32 | h[0];
  a[3] = (u64)h[1] << 32 | h[2];
  a[4] = (u64)h[3] << 32 | h[4];
#pragma unroll
  for (int p = 0; p < 20; ++p) {
    u64 idx = bloom_index(a, p);
    atomicOr((unsigned long long*)&bits[bloom_mod(b, idx >> 6)], 1ull << (idx & 63));
  }
}
#endif
