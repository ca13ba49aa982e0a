// lookahead_host.cpp — the look-ahead logic of the library (csrc/abi_lookahead.h, verbatim) compiled for the host with g++ around a
// stand-in for the search kernel, so that the CPU test-suite can drive it without a GPU (tests/test_lookahead_host.py): call
// sequences, worker threads on several contexts and "devices", sweeps that overflow or fail.  Not part of the product library.
//
// The stand-in `add_core` reports a hit for key index i = scalar / 2^offs (low 64 bits) whenever mix(i, filter) falls under a density
// that may depend on the index (clustered hits); the test computes the same set in numpy.
#include "../../../include/ecloop_hip.h"

#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

typedef uint32_t u32;
typedef uint64_t u64;
#include "../scalar_host.h"
#include "../abi_lookahead_ctx.h"

#define HELD_MAX (1u << 18) /* records the stand-in device keeps per launch (the library: max(cap, 2^20)) */

struct ecl_hip {
  int dev = 0;
  u32 flags = 0, offs = 0;
  u64 bloom_words = 0, list_n = 0;
  u32 last_held = 0, last_total = 0;
  LA_CONTEXT_MEMBERS
  // the stand-in device
  u64 filter = 0, one_in = 4096, dense_from = ~0ull, dense_len = 0, dense_one_in = 1, poison = ~0ull;
  std::vector<ecl_found> kept;
  u64 launches = 0, launched_keys = 0, failed = 0;
};

static int hipGetLastError() { return 0; }

static inline u64 mix(u64 i, u64 f) {
  u64 z = i + f * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int add_core(ecl_hip* h, const u256& k0, uint64_t nkeys, ecl_found* out, uint32_t cap, uint32_t* nout) {
  *nout = 0;
  h->last_held = h->last_total = 0, h->last_from_host = false;
  u256 q = k0;
  for (u32 i = 0; i < h->offs; ++i) q = sc_half(q);  // exact for the test's scalars (multiples of the stride)
  const u64 i0 = q.w[0];
  h->launches += 1, h->launched_keys += nkeys;
  if (h->poison - i0 < nkeys) {  // "the scan contains the private key 0"
    h->failed += 1;
    return ECL_E_RANGE;
  }
  h->kept.clear();
  u64 cnt = 0;
  for (u64 j = 0; j < nkeys; ++j) {
    const u64 i = i0 + j, m = mix(i, h->filter);
    const u64 one_in = (i - h->dense_from < h->dense_len) ? h->dense_one_in : h->one_in;
    if (m % one_in) continue;
    if (cnt < HELD_MAX) {
      ecl_found f;
      memset(&f, 0, sizeof f);
      f.key_offset = j, f.compressed = (uint8_t)((m >> 40) & 1), f.endo = (uint8_t)((m >> 44) % 6);
      for (int w = 0; w < 5; ++w) f.h160[w] = (u32)mix(i, h->filter + 1 + w);
      h->kept.push_back(f);
    }
    ++cnt;
  }
  if (cnt > 0xFFFFFFFFull) cnt = 0xFFFFFFFFull;
  const u32 take = cnt < cap ? (u32)cnt : cap;
  const u32 held = cnt < HELD_MAX ? (u32)cnt : HELD_MAX;
  memcpy(out, h->kept.data(), (size_t)(take < held ? take : held) * sizeof(ecl_found));
  h->last_held = held, h->last_total = (u32)cnt;
  *nout = (u32)cnt;
  return cnt > cap ? ECL_E_OVERFLOW : ECL_OK;
}

static int la_fetch(ecl_hip* h, uint32_t first, ecl_found* out, uint32_t n, uint32_t* got);
extern "C" int ecl_hip_fetch_found(ecl_hip* h, uint32_t first, ecl_found* out, uint32_t n, uint32_t* got) {
  *got = 0;
  if (h->last_from_host) return la_fetch(h, first, out, n, got);
  if (first >= h->last_held || n == 0) return ECL_OK;
  const u32 take = h->last_held - first < n ? h->last_held - first : n;
  memcpy(out, h->kept.data() + first, (size_t)take * sizeof(ecl_found));
  *got = take;
  return ECL_OK;
}

#include "../abi_lookahead.h"

extern "C" {
ecl_hip* lh_open(int dev, u32 flags, u32 offs, u64 filter, u64 one_in) {
  ecl_hip* h = new ecl_hip();
  h->dev = dev, h->flags = flags, h->offs = offs, h->filter = filter, h->one_in = one_in ? one_in : 1;
  h->bloom_words = 1024, h->la_bloom_fp = filter, h->la_key_valid = true;
  h->la_max = 1ull << 22;
  return h;
}
void lh_close(ecl_hip* h) {
  la_leave(h);
  delete h;
}
void lh_cluster(ecl_hip* h, u64 from, u64 len, u64 one_in) { h->dense_from = from, h->dense_len = len, h->dense_one_in = one_in ? one_in : 1; }
void lh_poison(ecl_hip* h, u64 index) { h->poison = index; }
void lh_fix_geometry(ecl_hip* h) { h->geom_fixed = true, la_leave(h); }
int lh_add_range(ecl_hip* h, const uint64_t start[4], uint64_t nkeys, ecl_found* out, uint32_t cap, uint32_t* nout) {
  *nout = 0;
  if (nkeys == 0) return ECL_OK;
  return la_dispatch(h, sc_reduce(u256_from(start)), nkeys, out, cap, nout);
}
void lh_device_stats(ecl_hip* h, u64* launches, u64* launched_keys, u64* failed) { *launches = h->launches, *launched_keys = h->launched_keys, *failed = h->failed; }
}
