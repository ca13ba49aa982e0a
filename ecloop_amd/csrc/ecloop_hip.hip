// ecloop_hip.hip — kernels' instantiation + the C ABI of include/ecloop_hip.h (host side, HIP runtime).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ecloop_hip.hip -o libecloop_hip.so
#include "../../include/ecloop_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "add_kernel.h"
#include "ec.h"
#include "scalar_host.h"
#include "abi_lookahead_ctx.h"
#include "setup_kernels.h"
#include "mul_kernels.h"
#include "aux_kernels.h"

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
  uint4* d_ctab = nullptr; size_t ctab_cap = 0;  // (g + 1) * D, g < T, for the geometry of d_aux (k_init_centres_table)
  bool ctab_valid = false;
  uint4* d_scr = nullptr; u32* d_scr2 = nullptr; size_t scr_elems = 0;  // prefix-product chains
  u64* d_bloom = nullptr; u64 bloom_words = 0;
  // `mul`: scalars travel in pieces through MUL_NBUF device buffers (and as many pinned staging buffers for pageable callers), the copy
  // engine running up to MUL_NBUF - 1 pieces ahead of the kernel
  u32* d_kbuf[MUL_NBUF] = {}; u32* pin_k[MUL_NBUF] = {}; u32 kbuf_cap = 0, pin_cap = 0;
  u32* d_multmp[2] = {};                       // parked window sums of one piece (144 bytes per scalar), one space per compute stream
  const u32* d_multab = nullptr; u32 multab_W = 0;  // `mul`'s window table in use: one per (device, width), shared by the contexts
  u32 mul_W_fixed = 0;                         // ecl_hip_set_mul_window: 0 = automatic
  uint64_t mul_seen = 0;                       // scalars this context has multiplied (never reset: the automatic width goes by it)
  bool mul_long_failed = false;                // the long table could not be allocated: do not try again
  void* d_ver = nullptr; u32 ver_cap = 0;      // staging of ecl_hip_verify
  u32* d_rawtext = nullptr; size_t rawtext_cap = 0; u64* d_rawlines = nullptr; u32 rawlines_cap = 0;  // `mul -raw`: text and line table of one call
  hipStream_t copy_stream = nullptr, stream2 = nullptr;  // `mul`: the copy engine's stream; the second compute stream (pieces alternate)
  hipEvent_t ev_copied[MUL_NBUF] = {}, ev_free[MUL_NBUF] = {}, ev_fork = nullptr, ev_join = nullptr;
  hipStream_t prep_stream = nullptr; hipEvent_t ev_hashed[MUL_NBUF] = {};  // `mul -raw`: the lines are hashed on a stream of their own, ahead of the pieces
  u32* d_list = nullptr; u64 list_n = 0;       // optional sorted hash list (exact confirm on the device)
  ecl_found_dev* d_found = nullptr; u32 found_cap = 0;
  u32* d_counter = nullptr; u32* pin_counter = nullptr;  // (the counters' page-locked host copy: read back by the copy engine, no compute slot needed)
  // records of the last add_range / mul_batch call that are still on the device (ecl_hip_fetch_found): where they start in d_found,
  // how many the device holds, the call's total, and whether the endo byte is meaningful
  u32 last_at = 0, last_held = 0, last_total = 0; bool last_endo = false;
  // walk state for contiguous continuation
  bool walk_valid = false;
  u32 walk_T = 0, walk_B = 0;
  bool B_auto = true;  // half group not fixed by the caller: short calls take a smaller one (see ecl_hip_add_range)
  u256 walk_next;  // scalar (mod n) the resident centres are positioned for
  u32 jump_host[16];
  u32 aux_B = 0, aux_T = 0;  // geometry the jump and the ladder in d_aux / jump_host were computed for
  LA_CONTEXT_MEMBERS
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
static void la_leave(struct ecl_hip* h);
static void la_filter_changed(struct ecl_hip* h, const uint64_t* bits, uint64_t nwords);
static void la_list_changed(struct ecl_hip* h, const uint32_t (*h160)[5], uint64_t n);
static u64 la_default_max();

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
  h->dev = device, h->flags = flags, h->offs = ord_offs, h->la_max = la_default_max();
  *out = h;
  HIPCHK(h, hipSetDevice(device));
  HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreate(&h->ev0));
  HIPCHK(h, hipEventCreate(&h->ev1));
  HIPCHK(h, hipEventCreate(&h->ev_s0));
  HIPCHK(h, hipEventCreate(&h->ev_s1));
  HIPCHK(h, hipMalloc(&h->d_aux, 34 * 16 * sizeof(u32)));
  HIPCHK(h, hipMalloc(&h->d_auxk, 34 * 8 * sizeof(u32)));
  HIPCHK(h, hipMalloc(&h->d_counter, 4 * sizeof(u32)));  // [0] records appended, [1] records confirmed by the list, [2] [3] input flags of mul_batch_raw
  HIPCHK(h, hipHostMalloc((void**)&h->pin_counter, 4 * sizeof(u32), hipHostMallocDefault));
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
  la_leave(h);
  (void)hipSetDevice(h->dev);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  if (h->prep_stream) (void)hipStreamSynchronize(h->prep_stream);
  (void)hipFree(h->d_tab), (void)hipFree(h->d_gtab), (void)hipFree(h->d_aux), (void)hipFree(h->d_auxk), (void)hipFree(h->d_cxy), (void)hipFree(h->d_ctab);
  (void)hipFree(h->d_scr), (void)hipFree(h->d_scr2), (void)hipFree(h->d_bloom), (void)hipFree(h->d_list), (void)hipFree(h->d_found), (void)hipFree(h->d_counter);
  if (h->pin_counter) (void)hipHostFree(h->pin_counter);
  for (int i = 0; i < MUL_NBUF; ++i) {
    (void)hipFree(h->d_kbuf[i]);
    if (h->pin_k[i]) (void)hipHostFree(h->pin_k[i]);
    if (h->ev_copied[i]) (void)hipEventDestroy(h->ev_copied[i]);
    if (h->ev_free[i]) (void)hipEventDestroy(h->ev_free[i]);
    if (h->ev_hashed[i]) (void)hipEventDestroy(h->ev_hashed[i]);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->prep_stream) (void)hipStreamDestroy(h->prep_stream);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  (void)hipFree(h->d_multmp[0]), (void)hipFree(h->d_multmp[1]), (void)hipFree(h->d_ver), (void)hipFree(h->d_rawtext), (void)hipFree(h->d_rawlines);
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
  la_leave(h), h->la_key_valid = false;
  if (h->d_bloom) HIPCHK(h, hipFree(h->d_bloom));
  h->d_bloom = nullptr, h->bloom_words = 0;
  HIPCHK(h, hipMalloc(&h->d_bloom, nwords * sizeof(u64)));
  // on the handle's own stream: device threads upload in parallel, each over its own PCIe link; memory from ecl_hip_alloc_host goes
  // by DMA at link rate, a pageable buffer is staged by the runtime
  HIPCHK(h, hipMemcpyAsync(h->d_bloom, bits, nwords * sizeof(u64), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->bloom_words = nwords;
  la_filter_changed(h, bits, nwords);
  return ECL_OK;
}

// Page-locked host memory comes from the runtime (hipHostMalloc, next to the GPU's NUMA node), never from registering the caller's
// own memory in place: hipHostRegister / hipHostUnregister cycles on memory that the host allocator recycles end in GPU memory access
// faults inside the ROCm runtime (rounds 2 and 5; tools/repro_pin_fault.py, profiles/r05_pin_fault.txt), which is why the entry points
// that used to do that are gone.  Pageable buffers are copied by the runtime (filter) or staged through the library's own pinned
// buffers (scalar arrays of ecl_hip_mul_batch).
#define ECL_PIN_MIN_BYTES ((size_t)1 << 20)  /* ecl_hip_mul_batch: batches below this are staged whatever their memory is */
void* ecl_hip_alloc_host(size_t bytes) {
  void* p = nullptr;
  if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
  return p;
}
void ecl_hip_free_host(void* p) {
  if (p) (void)hipHostFree(p);
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
  la_leave(h);
  if (h->d_list) HIPCHK(h, hipFree(h->d_list));
  h->d_list = nullptr, h->list_n = 0;
  la_list_changed(h, h160, 0);
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipMalloc(&h->d_list, n * 20));
  HIPCHK(h, hipMemcpy(h->d_list, h160, n * 20, hipMemcpyHostToDevice));
  h->list_n = n;
  la_list_changed(h, h160, n);
  return ECL_OK;
}

// ---- load_filter's list preparation (main.c:96-131: qsort by compare_160, then one blf_add per entry) on the device ------
// 10^7 entries cost the host 13 s (qsort of 20-byte records + 2 * 10^8 scattered bit sets), 10^8 two minutes; here: five
// stable 32-bit radix sorts of a permutation (least significant word first = compare_160's word-by-word order,
// addr.c:18-26; each four 8-bit passes: aux_kernels.h), a gather, adjacent-duplicate flags + exclusive scan + scatter.  The bits are
// set by the bulk insert kernel into a filter of the reference's list-mode size (2 words per entry).
}  // extern "C"
// exclusive scan of n counts on the handle's stream; `tmp` holds the two upper levels (n / 2048 + n / 2048^2 + 2 words)
static u64 scan_tmp_words(u64 n) {
  const u64 per = (u64)SCAN_BLOCK * SCAN_PER, l1 = (n + per - 1) / per, l2 = (l1 + per - 1) / per;
  return l1 + l2 + 2;
}
static int scan_exclusive(ecl_hip* h, const u32* in, u32* out, u64 n, u32* tmp) {
  const u64 per = (u64)SCAN_BLOCK * SCAN_PER, nb = (n + per - 1) / per;
  if (nb <= 1) {
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, h->stream, in, out, n, (u32*)nullptr);
    HIPCHK(h, hipGetLastError());
    return ECL_OK;
  }
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, h->stream, in, out, n, tmp);
  HIPCHK(h, hipGetLastError());
  const int rc = scan_exclusive(h, tmp, tmp, nb, tmp + nb);  // the workgroups' totals, in place, one level up
  if (rc != ECL_OK) return rc;
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, h->stream, out, n, tmp);
  HIPCHK(h, hipGetLastError());
  return ECL_OK;
}
// stable sort of the n (key, value) pairs by key: four 8-bit passes, result back in key[0] / val[0]
static int rsort_pairs(ecl_hip* h, u32* key[2], u32* val[2], u32 n, u32 nt, u32 L, u32* counts, u32* scan_tmp) {
  const dim3 grid((nt + RSORT_BLOCK - 1) / RSORT_BLOCK), blk(RSORT_BLOCK);
  int cur = 0;
  for (u32 shift = 0; shift < 32; shift += 8) {
    hipLaunchKernelGGL(k_rsort_count, grid, blk, 0, h->stream, key[cur], n, nt, L, shift, counts);
    HIPCHK(h, hipGetLastError());
    const int rc = scan_exclusive(h, counts, counts, (u64)256 * nt, scan_tmp);
    if (rc != ECL_OK) return rc;
    hipLaunchKernelGGL(k_rsort_scatter, grid, blk, 0, h->stream, key[cur], val[cur], n, nt, L, shift, counts, key[cur ^ 1], val[cur ^ 1]);
    HIPCHK(h, hipGetLastError());
    cur ^= 1;
  }
  return ECL_OK;  // four passes: an even number of swaps
}
extern "C" {
int ecl_hip_sort_list(ecl_hip* h, uint32_t (*h160)[5], uint64_t n, uint64_t* kept) {
  if (!h || !kept || (n && !h160) || n >= (1ull << 31)) return ECL_E_ARG;
  *kept = 0;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  const u32 N = (u32)n;
  // threads of the sort: tiles of at least 256 elements, at most 65536 threads
  u32 nt = (N + 255u) / 256u;
  nt = nt < RSORT_BLOCK ? RSORT_BLOCK : nt > 65536u ? 65536u : nt;
  const u32 L = (N + nt - 1) / nt;
  dbuf<u32> rec, out, key[2], perm[2], flag, pos, counts, tmp;
  HIPCHK(h, hipMalloc(&rec.p, (size_t)N * 20));
  HIPCHK(h, hipMalloc(&out.p, (size_t)N * 20));
  for (int i = 0; i < 2; ++i) {
    HIPCHK(h, hipMalloc(&key[i].p, (size_t)N * 4));
    HIPCHK(h, hipMalloc(&perm[i].p, (size_t)N * 4));
  }
  HIPCHK(h, hipMalloc(&flag.p, (size_t)N * 4));
  HIPCHK(h, hipMalloc(&pos.p, (size_t)N * 4));
  HIPCHK(h, hipMalloc(&counts.p, (size_t)256 * nt * 4));
  const u64 tmp_words = scan_tmp_words((u64)256 * nt) > scan_tmp_words(N) ? scan_tmp_words((u64)256 * nt) : scan_tmp_words(N);
  HIPCHK(h, hipMalloc(&tmp.p, (size_t)tmp_words * 4));
  HIPCHK(h, hipMemcpyAsync(rec.p, h160, (size_t)N * 20, hipMemcpyHostToDevice, h->stream));
  const dim3 grid((N + 255) / 256), blk(256);
  hipLaunchKernelGGL(k_list_iota, grid, blk, 0, h->stream, perm[0].p, N);
  u32* kk[2] = {key[0].p, key[1].p};
  u32* pp[2] = {perm[0].p, perm[1].p};
  for (int word = 4; word >= 0; --word) {  // LSD: the last word first, every sort stable
    hipLaunchKernelGGL(k_list_key, grid, blk, 0, h->stream, rec.p, pp[0], kk[0], N, (u32)word);
    HIPCHK(h, hipGetLastError());
    const int rc = rsort_pairs(h, kk, pp, N, nt, L, counts.p, tmp.p);
    if (rc != ECL_OK) return rc;
  }
  hipLaunchKernelGGL(k_list_gather_flag, grid, blk, 0, h->stream, rec.p, pp[0], out.p, flag.p, N);
  HIPCHK(h, hipGetLastError());
  {
    const int rc = scan_exclusive(h, flag.p, pos.p, N, tmp.p);
    if (rc != ECL_OK) return rc;
  }
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
  if (half_group || max_lanes) h->geom_fixed = true, la_leave(h);
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

// found records of one call: [0, raw_cap) written by the search kernel; in list mode the confirmed ones are
// compacted into [raw_cap, 2 * raw_cap).  d_counter[0] = records pushed, d_counter[1] = records confirmed.
// raw_cap = max(cap, 2^20) whatever the caller's `cap` (32 MB of HBM, twice that in list mode): a call that reports
// ECL_E_OVERFLOW has kept the records that did not fit the caller's buffer, and ecl_hip_fetch_found hands them over
// without the search kernel running again.
#define ECL_RAW_CAP_MIN (1u << 20)
static int ensure_found(ecl_hip* h, u32 cap) {
  if (cap <= h->found_cap) return ECL_OK;
  h->last_held = h->last_total = 0;  // the buffer that held the last call's records goes away
  if (h->d_found) HIPCHK(h, hipFree(h->d_found));
  h->d_found = nullptr, h->found_cap = 0;
  HIPCHK(h, hipMalloc(&h->d_found, (size_t)cap * sizeof(ecl_found_dev)));
  h->found_cap = cap;
  return ECL_OK;
}

static u32 raw_cap_of(const ecl_hip*, u32 cap) { return cap > ECL_RAW_CAP_MIN ? cap : ECL_RAW_CAP_MIN; }
#define ECL_CAP_MAX (1u << 30)  /* records one call can be asked to deliver (32 GB of them); list mode allocates twice the count */
static u32 found_words_of(const ecl_hip* h, u32 rcap) { return h->d_list ? 2u * rcap : rcap; }  // records to allocate for a call (rcap <= 2^30)
static void found_to_host(ecl_found* out, const ecl_found_dev* tmp, u32 n, bool keep_endo) {
  for (u32 i = 0; i < n; ++i) {
    out[i].key_offset = tmp[i].key_offset;
    memcpy(out[i].h160, tmp[i].h160, 20);
    out[i].endo = keep_endo ? (uint8_t)(tmp[i].tag & 0xff) : 0, out[i].compressed = (uint8_t)((tmp[i].tag >> 8) & 1);
    out[i].pad[0] = out[i].pad[1] = 0;
  }
}
// after the search kernel has been queued on h->stream: optional list confirm, then counters and records to the host
static int collect_found(ecl_hip* h, u32 cap, u32 rcap, ecl_found* out, u32* nout, bool keep_endo) {
  const bool lst = h->d_list != nullptr;
  if (lst) {
    const u32 blocks = (rcap + 255) / 256 < 1024 ? (rcap + 255) / 256 : 1024;
    hipLaunchKernelGGL(k_list_filter, dim3(blocks), dim3(256), 0, h->stream, h->d_found, h->d_counter, rcap, h->d_list, h->list_n,
                       h->d_found + rcap, h->d_counter + 1, rcap);
    HIPCHK(h, hipGetLastError());
  }
  // One read-back and one wait per call, into page-locked memory: a copy into pageable memory goes through a staging kernel, and with
  // a second context's k_mul_check launches filling the chip every small kernel waits milliseconds for a slot (each such wait at the
  // end of a `mul` call is time this context has nothing in flight).  All four words: mul_batch_raw reads its flags from the same copy.
  u32* cnts = h->pin_counter;
  HIPCHK(h, hipMemcpyAsync(cnts, h->d_counter, 4 * sizeof(u32), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const u32 cnt = lst ? cnts[1] : cnts[0];
  const u32 take = cnt < cap ? cnt : cap;
  h->last_at = lst ? rcap : 0, h->last_held = cnt < rcap ? cnt : rcap, h->last_total = cnt, h->last_endo = keep_endo;
  if (take) {
    std::vector<ecl_found_dev> tmp(take);
    HIPCHK(h, hipMemcpy(tmp.data(), h->d_found + h->last_at, (size_t)take * sizeof(ecl_found_dev), hipMemcpyDeviceToHost));
    found_to_host(out, tmp.data(), take, keep_endo);
  }
  *nout = cnt;
  if (lst && cnts[0] > rcap) {  // the bloom let more through than the staging area holds: some were never looked up
    h->err = "list mode: more bloom hits in one call than the device staging area holds";
    *nout = cnts[0];
    h->last_held = 0, h->last_total = cnts[0];  // nothing to fetch: the confirmed set is incomplete
    return ECL_E_OVERFLOW;
  }
  return cnt > cap ? ECL_E_OVERFLOW : ECL_OK;
}

static int la_fetch(ecl_hip* h, uint32_t first, ecl_found* out, uint32_t n, uint32_t* got);
extern "C" int ecl_hip_fetch_found(ecl_hip* h, uint32_t first, ecl_found* out, uint32_t n, uint32_t* got) {
  if (!h || !got || (!out && n)) return ECL_E_ARG;
  *got = 0;
  if (h->last_from_host) return la_fetch(h, first, out, n, got);  // the last call was served from a look-ahead sweep: its records are on the host
  if (first >= h->last_held || n == 0) return ECL_OK;
  const u32 take = h->last_held - first < n ? h->last_held - first : n;
  HIPCHK(h, hipSetDevice(h->dev));
  std::vector<ecl_found_dev> tmp(take);
  HIPCHK(h, hipMemcpy(tmp.data(), h->d_found + h->last_at + first, (size_t)take * sizeof(ecl_found_dev), hipMemcpyDeviceToHost));
  found_to_host(out, tmp.data(), take, h->last_endo);
  *got = take;
  return ECL_OK;
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
  dbuf<u32> d_k, d_w;  // freed on every way out (a failed call used to leak them)
  HIPCHK(h, hipMalloc(&d_k.p, ks.size() * sizeof(u32)));
  HIPCHK(h, hipMalloc(&d_w.p, (size_t)B * 16 * sizeof(u32)));
  HIPCHK(h, hipMalloc(&h->d_tab, (size_t)B * ECL_TAB_STRIDE * sizeof(u32)));
  HIPCHK(h, hipMemcpyAsync(d_k.p, ks.data(), ks.size() * sizeof(u32), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_mul_g, dim3((B + 63) / 64), dim3(64), 0, h->stream, d_k.p, d_w.p, (u8*)nullptr, B);
  HIPCHK(h, hipGetLastError());
  hipLaunchKernelGGL(k_tab_to_limbs, dim3((B + 63) / 64), dim3(64), 0, h->stream, d_w.p, h->d_tab, B);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
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

// Geometry of one call.  The table for half group h->B holds the tables of all smaller ones as prefixes, so a call may walk any
// power-of-two half group up to h->B.  Two costs pull in opposite directions: a lane pays one inversion per group (the division
// steps cost about five keys' worth of work: + 5 / 2B per key), and a walk with few lanes leaves the chip empty or in lock-step -
// the kernel only reaches its rate when the slots are oversubscribed with blocks in different phases.  Both were measured
// (profiles/r05_short_calls.txt: calls of 2^21 ... 2^26 keys at half groups 8 ... 128; profiles/r04_short_calls.txt: 2^29 ... 2^32):
// the time per key relative to the long-call rate is, to a few per cent, lane_factor(lanes) x (1 + 5 / 2B) with
//   lanes        2^13  2^14  2^15  2^16  2^17  2^18  2^19  2^20  2^21
//   lane_factor  10.9   5.5  2.77  1.41  1.17  1.10  1.04  1.00  0.975
// and the automatic choice is the half group that minimises the product.  That gives 1024 x 2^21 lanes for 2^32 keys, 128 x 2^21 for
// 2^29 (the per-GPU shard of the named range on 8 GPUs; as round 4 found by sweeping), 32 x 2^18 for 2^24 and 8 x 2^17 for the
// reference's own MAX_JOB_SIZE of 2^21 keys (main.c:16) - 7.1 Gkeys/s where the round-4 floor of 128 (8192 lanes: 32 workgroups on
// 256 CUs) gave 1.1.  Contiguous calls of one size keep one geometry, so they still continue the resident walk.
#define ECL_B_FLOOR 8u  /* k_init_centres_batched parks 144 bytes per lane in the chain scratch (lanes * B * 36 bytes) */
static double lane_factor(double lanes) {
  static const double f[] = {10.9, 5.5, 2.77, 1.41, 1.17, 1.10, 1.04, 1.00, 0.975};  // 2^13 ... 2^21 lanes
  const double l = log2(lanes < 1 ? 1 : lanes);
  if (l <= 13) return f[0] * exp2(13 - l);  // below: the walk is one chain per lane, time goes with 1 / lanes
  if (l >= 21) return f[8];
  const int i = (int)l - 13;
  return f[i] + (f[i + 1] - f[i]) * (l - (13 + i));
}
static u32 auto_half_group(const ecl_hip* h, u64 nkeys) {
  u32 best = h->B;
  if (!h->B_auto) return best;
  double best_cost = 0;
  for (u32 B = h->B; B >= ECL_B_FLOOR; B >>= 1) {
    const u64 groups = (nkeys + 2ull * B - 1) / (2ull * B);
    const double lanes = groups < h->Tmax ? (double)groups : (double)h->Tmax;
    const double cost = lane_factor(lanes) * (1.0 + 5.0 / (2.0 * B));
    if (B == h->B || cost < best_cost * 0.995) best = B, best_cost = cost;  // a smaller half group has to win by more than noise
    if (groups >= h->Tmax) break;  // every lane already has a group of its own: halving further only adds inversions
  }
  return best;
}
static void call_geometry(const ecl_hip* h, u64 nkeys, u32& B, u32& nb, u32& T) {
  B = auto_half_group(h, nkeys);
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
  const u32 B = auto_half_group(h, nkeys);
  const u64 ngroups = (nkeys + 2ull * B - 1) / (2ull * B);
  return (ngroups + h->Tmax - 1) / h->Tmax < (1ull << 32);
}
extern "C" int ecl_hip_plan_geometry(ecl_hip* h, uint64_t nkeys, uint32_t* half_group, uint32_t* lanes, uint32_t* groups_per_lane) {
  if (!h || nkeys == 0) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc = default_lanes(h);
  if (rc != ECL_OK) return rc;
  if (!nkeys_ok(h, nkeys)) return ECL_E_ARG;
  u32 B, nb, T;
  call_geometry(h, nkeys, B, nb, T);
  if (half_group) *half_group = B;
  if (lanes) *lanes = T;
  if (groups_per_lane) *groups_per_lane = nb;
  return ECL_OK;
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
  if (h->ctab_cap < T) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_ctab) HIPCHK(h, hipFree(h->d_ctab));
    h->d_ctab = nullptr, h->ctab_cap = 0, h->ctab_valid = false, h->aux_B = h->aux_T = 0;
    HIPCHK(h, hipMalloc(&h->d_ctab, (size_t)T * 4 * sizeof(uint4)));
    h->ctab_cap = T;
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
  if (!h || nkeys == 0 || cap > ECL_CAP_MAX) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  if ((rc = default_lanes(h)) != ECL_OK) return rc;
  if ((rc = ensure_table(h)) != ECL_OK) return rc;
  if ((rc = ensure_gtable(h)) != ECL_OK) return rc;
  if (!nkeys_ok(h, nkeys)) return ECL_E_ARG;
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, found_words_of(h, rcap))) != ECL_OK) return rc;
  u32 B, nb, T;
  call_geometry(h, nkeys, B, nb, T);
  return ensure_walk_buffers(h, B, T);
}

// one launch of the search kernel over exactly the keys k0, k0 + s, ..., k0 + (nkeys - 1) s  (k0 reduced mod n): the body of
// ecl_hip_add_range; the look-ahead (abi_lookahead.h) calls it with a sweep instead of the caller's job
static int add_core(ecl_hip* h, const u256& k0, uint64_t nkeys, ecl_found* out, uint32_t cap, uint32_t* nout) {
  *nout = 0;
  int rc;
  if ((rc = default_lanes(h)) != ECL_OK) return rc;
  if (!nkeys_ok(h, nkeys)) {
    h->err = "nkeys too large for one call with this geometry";
    return ECL_E_ARG;
  }
  if ((rc = ensure_table(h)) != ECL_OK) return rc;
  h->last_held = h->last_total = 0, h->last_from_host = false;  // the records of the call before are about to be overwritten
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, found_words_of(h, rcap))) != ECL_OK) return rc;

  u32 B, nb, T;
  call_geometry(h, nkeys, B, nb, T);
  const u64 group = 2ull * B;
  if ((rc = ensure_walk_buffers(h, B, T)) != ECL_OK) return rc;

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
      // ... and so do the multiples (g + 1) * D, g < T (D = 2B * 2^offs * G = the first ladder point): the lane centres C_0 = D, D = D
      if (B >= 8)
        hipLaunchKernelGGL(k_init_centres_batched, dim3((T / INIT_R + 255) / 256), dim3(256), 0, h->stream, h->d_aux + 32, h->d_aux + 32,
                           h->d_ctab, T, (u32*)h->d_scr);
      else
        hipLaunchKernelGGL(k_init_centres, dim3(T / 256), dim3(256), 0, h->stream, h->d_aux + 32, h->d_aux + 32, h->d_ctab, T);
      HIPCHK(h, hipGetLastError());
      HIPCHK(h, hipStreamSynchronize(h->stream));
      h->aux_B = B, h->aux_T = T, h->ctab_valid = true;
    }
    // C0 = (k0 + B*s)*G through the window table (19 additions, ~0.1 ms), the scalar travelling as a kernel argument;
    // then the lane centres C0 + g*D.  Nothing here waits for the host: the search kernel queues right behind.
    // (round 5) ... or, unless E = C0 - D is the point at infinity, E through the window table and the centres E + (g + 1) D from the cached
    // multiples of D: one affine addition per lane (k_init_centres_table)
    const u256 c0s = sc_add(k0, sc_mul_u64(s, B)), es = sc_add(c0s, sc_neg(sc_mul_u64(s, group)));
    const bool from_table = h->ctab_valid && (es.w[0] | es.w[1] | es.w[2] | es.w[3]) != 0;
    scalar_arg c0;
    words_of(c0.w, from_table ? es : c0s);
    hipLaunchKernelGGL(k_mul_window_one, dim3(1), dim3(64), 0, h->stream, c0, h->d_gtab, h->d_aux);
    HIPCHK(h, hipGetLastError());
    if (from_table && T >= (1u << 19))
      hipLaunchKernelGGL(k_init_centres_table<8u>, dim3((T / 8u + 255) / 256), dim3(256), 0, h->stream, h->d_aux, h->d_ctab, h->d_cxy, T);
    else if (from_table)
      hipLaunchKernelGGL(k_init_centres_table<4u>, dim3((T / 4u + 255) / 256), dim3(256), 0, h->stream, h->d_aux, h->d_ctab, h->d_cxy, T);
    else if (B >= 8)  // the chain scratch (T * B * 36 bytes) holds the 144 bytes per lane the batched set-up parks
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

#include "abi_lookahead.h"

// fingerprint of what a filter / a list holds, computed over ALL of its words where they are resident (k_fingerprint, aux_kernels.h: a 6 GB
// filter takes 2 ms): contexts share sweeps only if flags, stride, sizes and these agree.  A failure here only switches the look-ahead off.
static bool la_device_fingerprint(ecl_hip* h, const void* d_words, u64 nwords64, u64* fp) {
  dbuf<unsigned long long> acc;
  if (hipMalloc(&acc.p, 8) != hipSuccess || hipMemsetAsync(acc.p, 0, 8, h->stream) != hipSuccess) return false;
  const u64 blocks = (nwords64 + 255) / 256;
  hipLaunchKernelGGL(k_fingerprint, dim3((unsigned)(blocks < 16384 ? (blocks ? blocks : 1) : 16384)), dim3(256), 0, h->stream, (const u64*)d_words, nwords64, acc.p);
  unsigned long long v = 0;
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&v, acc.p, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return false;
  *fp = v;
  return true;
}
static void la_filter_changed(ecl_hip* h, const uint64_t*, uint64_t nwords) { h->la_key_valid = la_device_fingerprint(h, h->d_bloom, nwords, &h->la_bloom_fp); }
static void la_list_changed(ecl_hip* h, const uint32_t (*list)[5], uint64_t n) {
  h->la_list_fp = 0;
  if (!n) return;
  // the list's 20-byte entries as 64-bit words (n * 5 / 2, a last odd 32-bit word left to the entry count in the key: lists are sorted and
  // unique, so two lists that agree in everything but that word differ in it alone - it is added from the host copy)
  u64 fp = 0;
  if (!la_device_fingerprint(h, h->d_list, n * 5 / 2, &fp)) h->la_key_valid = false;
  if ((n * 5) & 1) fp += 0x9E3779B97F4A7C15ull * ((u64)list[n - 1][4] + 1);
  h->la_list_fp = fp;
}


extern "C" int ecl_hip_add_range(ecl_hip* h, const uint64_t start[4], uint64_t nkeys, ecl_found* out, uint32_t cap,
                                 uint32_t* nout) {
  if (!h || !start || (!out && cap) || !nout || cap > ECL_CAP_MAX) return ECL_E_ARG;
  *nout = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (nkeys == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  return la_dispatch(h, sc_reduce(u256_from(start)), nkeys, out, cap, nout);
}

#include "abi_mul.h"
#include "abi_diag.h"
