// aux_kernels.h - everything off the hot path: device primitives exposed for the parity tests, bulk bloom insert and blf-gen's exact
// count, the list preparation and the device-side list confirm.  (one translation unit: included by ecloop_hip.hip)
#pragma once
#include "add_kernel.h"
#include "ec.h"
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
  // the two inversions (fe256.h): division steps and the addition chain of lib/ecc.c:463-520; 11: unnormalised input, 1 / (4y - 2x)
  case 9: z = fe_inv_divsteps(x); break;
  case 10: z = fe_inv_fermat(x); break;
  case 11: z = fe_inv_divsteps(fe_add(fe_neg(fe_add(x, x), 2), fe_add(fe_add(y, y), fe_add(y, y)))); break;
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


// ---- list preparation (ecl_hip_sort_list): permutation, keys, duplicate flags, compaction ----------------------------------------
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

// fingerprint of n 64-bit words resident on the device (the look-ahead groups contexts by what their filters HOLD, abi_lookahead.h): the
// sum over all words of mix(word + index * odd constant) - order-free, so the grid adds it up with one atomic per wave; any differing word
// changes it but for a 2^-64 accident
__global__ void __launch_bounds__(256) k_fingerprint(const u64* __restrict__ w, u64 n, unsigned long long* __restrict__ out) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256u + threadIdx.x; i < n; i += (u64)gridDim.x * 256u) {
    u64 z = w[i] + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    acc += z ^ (z >> 31);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down((unsigned long long)acc, off, 64);
  if ((threadIdx.x & 63u) == 0) atomicAdd(out, (unsigned long long)acc);
}

// ---- the sort and the scan of the list preparation, written out (rounds 3-5 called hipCUB for them) ---------------------------------
// A stable LSD radix sort of (key, index) pairs, 8 bits per pass: `nt` threads own consecutive tiles of the input (thread t: elements
// [t L, (t + 1) L)); pass = count the digits of the tile into the thread's column of an LDS table, exclusive scan of all counts in
// digit-major / thread-minor order (= the position of each thread's first element of each digit), then every thread walks its tile
// again IN ORDER and puts each pair at its digit's running position.  Tiles are in index order and a thread keeps the order inside its
// tile, so equal digits keep their order: the five word passes of load_filter's qsort order (compare_160, addr.c:18-26) need that.
// One wave per workgroup (64 x 256 counters of 4 bytes = 64 KB of LDS: two workgroups per CU); a list is prepared once per run, 10^7
// entries take tens of milliseconds - nothing here is on the hot path.
#define RSORT_BLOCK 64u
__global__ void __launch_bounds__(RSORT_BLOCK) k_rsort_count(const u32* __restrict__ key, u32 n, u32 nt, u32 L, u32 shift, u32* __restrict__ counts) {
  __shared__ u32 cnt[256][RSORT_BLOCK];
  const u32 t = blockIdx.x * RSORT_BLOCK + threadIdx.x;
  for (u32 d = 0; d < 256; ++d) cnt[d][threadIdx.x] = 0;
  if (t < nt) {
    const u64 lo = (u64)t * L, hi = lo + L < n ? lo + L : n;
    for (u64 i = lo; i < hi; ++i) cnt[(key[i] >> shift) & 255u][threadIdx.x] += 1;
    for (u32 d = 0; d < 256; ++d) counts[(size_t)d * nt + t] = cnt[d][threadIdx.x];
  }
}
__global__ void __launch_bounds__(RSORT_BLOCK) k_rsort_scatter(const u32* __restrict__ key, const u32* __restrict__ val, u32 n, u32 nt, u32 L, u32 shift,
                                                               const u32* __restrict__ offs, u32* __restrict__ key_out, u32* __restrict__ val_out) {
  __shared__ u32 at[256][RSORT_BLOCK];
  const u32 t = blockIdx.x * RSORT_BLOCK + threadIdx.x;
  if (t >= nt) return;
  for (u32 d = 0; d < 256; ++d) at[d][threadIdx.x] = offs[(size_t)d * nt + t];
  const u64 lo = (u64)t * L, hi = lo + L < n ? lo + L : n;
  for (u64 i = lo; i < hi; ++i) {
    const u32 k = key[i], pos = at[(k >> shift) & 255u][threadIdx.x]++;
    key_out[pos] = k, val_out[pos] = val[i];
  }
}
// exclusive prefix sums of n 32-bit counts (totals below 2^32): a workgroup scans 2048 elements (8 per thread, Hillis-Steele over the
// threads' sums in LDS) and leaves its total in `sums`; the totals are scanned the same way one level up, and added back
#define SCAN_BLOCK 256u
#define SCAN_PER 8u
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(const u32* __restrict__ in, u32* __restrict__ out, u64 n, u32* __restrict__ sums) {
  __shared__ u32 part[SCAN_BLOCK];
  const u64 base = ((u64)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_PER;
  u32 v[SCAN_PER], mine = 0;
#pragma unroll
  for (u32 k = 0; k < SCAN_PER; ++k) v[k] = base + k < n ? in[base + k] : 0u, mine += v[k];
  part[threadIdx.x] = mine;
  __syncthreads();
  for (u32 step = 1; step < SCAN_BLOCK; step <<= 1) {
    const u32 add = threadIdx.x >= step ? part[threadIdx.x - step] : 0u;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  u32 run = part[threadIdx.x] - mine;  // exclusive prefix of this thread inside the workgroup
#pragma unroll
  for (u32 k = 0; k < SCAN_PER; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == SCAN_BLOCK - 1 && sums) sums[blockIdx.x] = part[SCAN_BLOCK - 1];
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_add(u32* __restrict__ out, u64 n, const u32* __restrict__ sums_scanned) {
  const u64 base = ((u64)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_PER;
  const u32 add = sums_scanned[blockIdx.x];
#pragma unroll
  for (u32 k = 0; k < SCAN_PER; ++k)
    if (base + k < n) out[base + k] += add;
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
