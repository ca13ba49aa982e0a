// abi_mul.h - host side of ecl_hip_mul_batch / _raw / _verify: window tables (built, checked, shared per device), staging, the piece
// pipeline.  (one translation unit: included by ecloop_hip.hip after the context and the add path)
#pragma once
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
  dbuf<u32> d_k, tab;  // freed on every way out; the table is handed to the context only when it is complete
  HIPCHK(h, hipMalloc(&d_k.p, ks.size() * sizeof(u32)));
  HIPCHK(h, hipMalloc(&tab.p, slots * 16 * sizeof(u32)));
  HIPCHK(h, hipMemcpy(d_k.p, ks.data(), ks.size() * sizeof(u32), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_mul_g, dim3((unsigned)((slots + 63) / 64)), dim3(64), 0, h->stream, d_k.p, tab.p, (u8*)nullptr, (u32)slots);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->d_gtab = tab.p, tab.p = nullptr;
  return ECL_OK;
}

// `mul`'s window tables (see k_gtable_rows): built once per (device, width) and process, shared by every context on that
// device that uses the width (the host program runs two per GPU), freed with the last of them.  Before a table is handed
// out, sample slots of every row - first, last, the low digits, the seams between threads, and a fixed pseudo-random set -
// are compared with the double-and-add kernel.
#define MUL_W_MIN 8u
#define MUL_W_MAX 29u                 /* 8 rows x 2^28 points + 2^24: 138 GB, 9 additions per scalar - on request only (ecl_hip_set_mul_window): +2 % on
                                         2^24-scalar calls, +3.8 % on 2^26 over 26 bits (profiles/r05_mul_w29.txt), seconds to build */
#define MUL_W_START 22u               /* 11 rows x 2^21 points (signed digits), 1.5 GB */
#define MUL_W_LONG 26u                /* 9 rows x 2^25 points + 2^22, 19.6 GB: 10 additions per scalar instead of 12 */
#define MUL_LONG_AFTER (1ull << 30)   /* scalars a context has seen before it moves to MUL_W_LONG: the wider table gains ~0.1 ns per scalar,
                                         so its build is paid back after 10^9 of them */
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
  // (a row wider than 2^20 threads - 25 bits and up - is built in slices of 2^20 threads: 2.4 GB of parking space whatever the width)
  const u32 nt_row = ((tb.stride > tb.top_cnt ? tb.stride : tb.top_cnt) + 15u) / 16u;
  const u32 nt = nt_row < (1u << 20) ? nt_row : (1u << 20);
  u32 rows = (1u << 18) / nt;
  rows = rows < 1 ? 1 : (rows > tb.nwin ? tb.nwin : rows);
  HIPCHK(h, hipMalloc(&tmp.p, (size_t)rows * 16 * 36 * nt * sizeof(u32)));
  HIPCHK(h, hipMalloc(&tab.p, wtab_slots(tb) * 16 * sizeof(u32)));
  for (u32 w0 = 0; w0 < tb.nwin; w0 += rows) {
    const u32 ny = tb.nwin - w0 < rows ? tb.nwin - w0 : rows;
    for (u32 t0 = 0; t0 < nt_row; t0 += nt)
      hipLaunchKernelGGL(k_gtable_rows, dim3((nt + 255) / 256, ny), dim3(256), 0, h->stream, lad.p, tab.p, tmp.p, nt, W, w0, t0);
  }
  HIPCHK(h, hipGetLastError());
  // the check
  const u32 PERW = 48;
  std::vector<u64> sl;
  std::vector<u32> wk;
  u64 z = 0xD1B54A32D192ED03ull;
  for (u32 w = 0; w < tb.nwin; ++w) {
    const u32 count = wtab_row_count(tb, w);
    for (u32 i = 0; i < PERW; ++i) {
      z ^= z << 13, z ^= z >> 7, z ^= z << 17;
      const u32 b = i == 0 ? 1u : i == 1 ? count : i < 18 ? (i - 1u) : i < 34 ? (i - 17u) * 16u + (i & 1u) : (u32)(z % count) + 1u;
      const u32 digit = b > count ? count : b;
      sl.push_back((u64)w * tb.stride + digit - 1);
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
    if (h->stream2) HIPCHK(h, hipStreamSynchronize(h->stream2));  // ... on either compute stream (a call that failed between its pieces never joined them)
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

// what a mul_batch of n scalars needs before its first copy: the window table of the width in force, the copy stream, the second
// compute stream and the events, the device staging (MUL_NBUF buffers: the copy engine runs ahead of the kernels) and the parking
// space of two pieces in flight - sized to the call, grown on demand
static int mul_setup(ecl_hip* h, u32 n, u32 W) {
  int rc;
  if ((rc = ensure_multable(h, W)) != ECL_OK) return rc;
  if (!h->copy_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (int i = 0; i < MUL_NBUF; ++i) {
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
      HIPCHK(h, hipEventCreateWithFlags(&h->ev_free[i], hipEventDisableTiming));
    }
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  }
  u32 want = 1u << 16;
  while (want < MUL_CHUNK && want < n) want <<= 1;
  if (want > h->kbuf_cap) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream2));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    for (int i = 0; i < MUL_NBUF; ++i) {
      if (h->d_kbuf[i]) HIPCHK(h, hipFree(h->d_kbuf[i]));
      if (h->pin_k[i]) HIPCHK(h, hipHostFree(h->pin_k[i]));
      h->d_kbuf[i] = nullptr, h->pin_k[i] = nullptr;
    }
    h->pin_cap = 0;
    for (int i = 0; i < 2; ++i) {
      if (h->d_multmp[i]) HIPCHK(h, hipFree(h->d_multmp[i]));
      h->d_multmp[i] = nullptr;
    }
    h->kbuf_cap = 0;
    for (int i = 0; i < MUL_NBUF; ++i) HIPCHK(h, hipMalloc(&h->d_kbuf[i], (size_t)want * 32));
    // the kernel indexes the planes as r * 36 * nt + plane * nt + t with nt = ceil(m / R) rounded up to whole workgroups:
    // up to R * 256 slots more than m
    for (int i = 0; i < 2; ++i) HIPCHK(h, hipMalloc(&h->d_multmp[i], ((size_t)want + MUL_R * 256u) * 36 * sizeof(u32)));
    h->kbuf_cap = want;
  }
  return ECL_OK;
}
// (the tuning hooks of this file read the environment of whoever loaded the library: every value is clamped to what the piece loop
//  can run with - a zero or tiny thread count divided by zero / never advanced the loop, advisor r04)
static u32 env_u32(const char* name, u32 dflt, u32 lo, u32 hi) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const long v = atol(e);
  return v < (long)lo ? lo : (v > (long)hi ? hi : (u32)v);
}
// threads and scalars per thread of one k_mul_check launch over m scalars: as many scalars per thread (one shared inversion, at most
// MUL_R) as still keep 65536 x ECL_MUL_WAVES threads in flight - what the chip holds at once; a few blocks more would wait for a whole
// round - in whole workgroups, so that every row of scalars and every parking plane starts on a 1 KiB boundary
static u32 mul_nt_target() {
  static const u32 v = env_u32("ECL_HIP_MUL_NT", 65536u * ECL_MUL_WAVES, 1024u, 1u << 22) & ~255u;  // tuning hook (A/B runs)
  return v;
}
static void mul_geometry(u32 m, u32* R_out, u32* nt_out) {
  const u32 chains = mul_nt_target();
  u32 R = (m + chains - 1) / chains;
  R = R < 1 ? 1 : (R > MUL_R ? MUL_R : R);
  *R_out = R, *nt_out = ((m + R - 1) / R + 255u) / 256u * 256u;
}
// one piece on compute stream `lane` (0: the context's stream, 1: the second one), parking space `lane`
static void mul_launch_piece(ecl_hip* h, int lane, const u32* d_k, u32 m, u32 at, const wtab& gtab, const add_args& a) {
  u32 R, nt;
  const bool a33 = h->flags & ECL_ADDR33, a65 = h->flags & ECL_ADDR65;
  hipStream_t st = lane ? h->stream2 : h->stream;
  u32* tmp = h->d_multmp[lane];
  mul_geometry(m, &R, &nt);
  dim3 grid(nt / 256), blk(256);
  if (a33 && a65) hipLaunchKernelGGL((k_mul_check<true, true>), grid, blk, 0, st, d_k, m, at, gtab, a, tmp, nt, R);
  else if (a33) hipLaunchKernelGGL((k_mul_check<true, false>), grid, blk, 0, st, d_k, m, at, gtab, a, tmp, nt, R);
  else hipLaunchKernelGGL((k_mul_check<false, true>), grid, blk, 0, st, d_k, m, at, gtab, a, tmp, nt, R);
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

// The body of a `mul` call, shared by ecl_hip_mul_batch (scalars from the host) and ecl_hip_mul_batch_raw (scalars hashed on the device from
// lines of text): n scalars in pieces through the MUL_NBUF device buffers.  copy_in(b, at, m) puts piece [at, at + m)'s input on the copy
// stream; prepare(b, at, m, &ready) enqueues what else has to happen before the piece's window sums and names the event they wait for
// (-raw: SHA-256 of the lines into d_kbuf[b], on a stream of its own).  The caller has reset the counters on h->stream and collects the
// records after it.
template <class CopyIn, class Prepare> static int mul_pieces(ecl_hip* h, u32 n, const wtab& gtab, const add_args& a, CopyIn copy_in, Prepare prepare) {
  // A call is cut into pieces so that the copy engine runs ahead of the kernel (MUL_NBUF device buffers): pieces of 1, 2, 4, 8 and then 10
  // scalars per resident thread (12 for calls of 2^26 scalars and more), i.e. 196 608 x 1, 2, 4, 8, 10 (12) scalars at three waves per SIMD -
  // each copy is about as long as the kernel before it.  More scalars per thread share an inversion among more of them but park more sums
  // (144 bytes each: at 16 per thread a piece parks 453 MB, past the Infinity Cache) and lengthen the pipeline's fill and drain:
  // tools/ab_mul_topr.sh (profiles/r04_mul_sched.txt, three waves per SIMD): 2^24-scalar calls 1252 / 1277 / 1252 / 1199 M scalars/s at
  // 8 / 10 / 12 / 16 per thread, 2^25: 1313 / 1320 / 1319 / 1291, 2^26: 1354 / 1330 / 1346 / 1324, 2^27: 1368 / 1367 / 1378 / 1340.
  // Round 4's earlier measurements at two waves per SIMD (tools/mul_kernel_times.sh, profiles/r04_mul_split.txt): 8 per thread 0.888 ms per
  // piece = 1.18 G scalars/s (22-bit table), 16 per thread 1.763 ms = 1.19 G/s, 32 per thread lose; doubling first pieces 1222-1230 / 1261-1270
  // on 2^24 / 2^26-scalar calls against 1214 / 1253 for a 2^18-scalar piece followed at once by full ones; two staging buffers 1208 / 1249.
  // (in units of one scalar per chain = what the chip holds at once: 2^17 scalars at two waves per SIMD: 2^18, then 2^20 / 2^21)
  // Round 5: the pieces alternate between TWO compute streams (each with its own parking space).  A piece's 768 workgroups are all
  // resident at once and do the same work, but they do not end together; on one stream the next piece's first workgroup waits for the
  // last of this one.  On two streams the next piece's workgroups move into the slots as they fall free (ECL_HIP_MUL_STREAMS=1: one stream).
  static const u32 first_R = env_u32("ECL_HIP_MUL_FIRST_R", 1u, 1u, MUL_R);   // tuning hooks (A/B runs)
  static const u32 grow_pct = env_u32("ECL_HIP_MUL_GROW", 200u, 100u, 1600u);
  static const u32 top_R = env_u32("ECL_HIP_MUL_TOP_R", 0u, 0u, MUL_R);
  static const u32 nstreams = env_u32("ECL_HIP_MUL_STREAMS", 2u, 1u, 2u);
  const u64 unit = (u64)mul_nt_target();
  const u64 top_want = unit * (top_R ? top_R : (n >= (1u << 26) ? 12u : 10u));
  const u32 top = h->kbuf_cap < top_want ? h->kbuf_cap : (u32)top_want;
  u32 lim = top < unit * first_R ? top : (u32)(unit * first_R);
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));  // the second stream starts behind the counters' reset (and behind the call before)
  HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
  for (u32 at = 0, c = 0, m = 0; at < n; at += m, ++c, lim = (u64)lim * grow_pct / 100 <= top ? (u32)((u64)lim * grow_pct / 100) & ~1023u : top) {
    const u32 b = c % MUL_NBUF;
    const int lane = (int)(c % nstreams);
    hipStream_t st = lane ? h->stream2 : h->stream;
    m = n - at < lim ? n - at : lim;
    // no crumb at the end: what would be left after this piece is taken along if it is less than half a piece (a 2^24-scalar call used to
    // end on a 65 536-scalar launch - a third of the chip, one inversion per scalar - that took 0.10 ms, 0.8 % of the call)
    if (n - at - m < lim / 2 && n - at <= h->kbuf_cap && (u64)(n - at) <= unit * MUL_R) m = n - at;
    if (c >= MUL_NBUF) HIPCHK(h, hipEventSynchronize(h->ev_free[b]));  // the kernel MUL_NBUF pieces back is done with this buffer (and its staging twin)
    if (int rc = copy_in(b, at, m)) return rc;
    HIPCHK(h, hipEventRecord(h->ev_copied[b], h->copy_stream));
    hipEvent_t ready = h->ev_copied[b];
    if (int rc = prepare(b, at, m, &ready)) return rc;
    HIPCHK(h, hipStreamWaitEvent(st, ready, 0));
    mul_launch_piece(h, lane, h->d_kbuf[b], m, at, gtab, a);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev_free[b], st));
  }
  HIPCHK(h, hipEventRecord(h->ev_join, h->stream2));  // the context's stream goes on (list confirm, counters) when both are done
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
  return ECL_OK;
}

extern "C" int ecl_hip_reserve_mul(ecl_hip* h, uint32_t n, uint32_t cap) {
  if (!h || n == 0 || cap > ECL_CAP_MAX) return ECL_E_ARG;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  if ((rc = ensure_found(h, found_words_of(h, raw_cap_of(h, cap ? cap : 1)))) != ECL_OK) return rc;
  u32 W;
  return mul_setup_auto(h, n, &W);
}

extern "C" int ecl_hip_mul_batch(ecl_hip* h, const uint64_t (*scalars)[4], uint32_t n, ecl_found* out, uint32_t cap,
                                 uint32_t* nout) {
  if (!h || (!scalars && n) || (!out && cap) || !nout || cap > ECL_CAP_MAX) return ECL_E_ARG;
  *nout = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  h->last_held = h->last_total = 0, h->last_from_host = false;  // the records of the call before are about to be overwritten
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, found_words_of(h, rcap))) != ECL_OK) return rc;
  u32 W;
  if ((rc = mul_setup_auto(h, n, &W)) != ECL_OK) return rc;
  const wtab gtab = wtab_make(h->d_multab, W);
  h->mul_seen += n;
  // Scalars in page-locked host memory (ecl_hip_alloc_host) go to the device by DMA straight from the
  // caller's array; pageable ones are first copied into two pinned staging buffers - a single-threaded memcpy that caps
  // the call near 18 GB/s = 570 M scalars/s (measured), below what the kernel takes.
  bool direct = false;
  {
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof attr);
    if ((size_t)n * 32 >= ECL_PIN_MIN_BYTES) {  // small batches are staged whatever their memory is
      // page-locked from the first byte to the last (memory from ecl_hip_alloc_host, or registered by the caller itself)
      if (hipPointerGetAttributes(&attr, scalars) == hipSuccess) direct = attr.type == hipMemoryTypeHost;
      else (void)hipGetLastError();
      if (direct) {
        memset(&attr, 0, sizeof attr);
        if (hipPointerGetAttributes(&attr, (const char*)scalars + (size_t)n * 32 - 1) == hipSuccess) direct = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError(), direct = false;
      }
    }
  }
  if (!direct && (!h->pin_k[0] || h->pin_cap < h->kbuf_cap)) {
    for (int i = 0; i < MUL_NBUF; ++i) {
      if (h->pin_k[i]) HIPCHK(h, hipHostFree(h->pin_k[i]));
      h->pin_k[i] = nullptr;
    }
    h->pin_cap = 0;
    for (int i = 0; i < MUL_NBUF; ++i) HIPCHK(h, hipHostMalloc(&h->pin_k[i], (size_t)h->kbuf_cap * 32, hipHostMallocDefault));
    h->pin_cap = h->kbuf_cap;
  }
  add_args a;
  memset(&a, 0, sizeof a);
  a.bloom = bloom_make(h->d_bloom, h->bloom_words);
  a.found = h->d_found, a.counter = h->d_counter, a.cap = rcap;
  HIPCHK(h, hipMemsetAsync(h->d_counter, 0, 2 * sizeof(u32), h->stream));
  // Scalars are used as given (4 little-endian u64 = 8 u32 words): the window sum (wtab_sum_fast, any width) is k*G for any
  // 256-bit k, which is (k mod n)*G; k = 0 (mod n) gives the point at infinity and is skipped.
  const bool staged = !direct;
  rc = mul_pieces(
      h, n, gtab, a,
      [&](u32 b, u32 at, u32 m) -> int {  // piece `at`'s scalars onto the copy stream
        const void* src = scalars[at];
        if (staged) memcpy(h->pin_k[b], scalars[at], (size_t)m * 32), src = h->pin_k[b];
        HIPCHK(h, hipMemcpyAsync(h->d_kbuf[b], src, (size_t)m * 32, hipMemcpyHostToDevice, h->copy_stream));
        return ECL_OK;
      },
      [](u32, u32, u32, hipEvent_t*) -> int { return ECL_OK; });
  if (rc != ECL_OK) return rc;
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

// `mul -raw`: lines of text in, SHA-256 on the device, then the `mul` body on the digests (n <= 2^26 lines a call): the pieces of
// ecl_hip_mul_batch, with 8 bytes of table + the line instead of 32 bytes of scalar on the copy stream, and every piece's lines hashed
// into its staging buffer ahead of its window sums (details at the two lambdas below).  Until round 6 a call was ONE piece of at most
// 2^22 lines on one stream and the host program made ~2 M-line calls: 0.79-0.87 G lines/s over 2^30 pass phrases; as pieces with the
// hashing between the window sums of one stream and the whole text sent first, 2^24-line calls took 20.2 ms against 14.0 for 2^24
// scalars; now 14.5-15.1 ms (tools/raw_api_probe.py), 1.09-1.16 G lines/s through the host program (tools/rawprobe3_r06.sh).
extern "C" int ecl_hip_mul_batch_raw(ecl_hip* h, const uint8_t* text, uint32_t text_bytes, const uint64_t* lines, uint32_t n, ecl_found* out,
                                     uint32_t cap, uint32_t* nout) {
  if (!h || (!text && text_bytes) || (!lines && n) || (!out && cap) || !nout || n > MUL_RAW_MAX || text_bytes > 0xFFFFFFF0u || cap > ECL_CAP_MAX) return ECL_E_ARG;
  *nout = 0;
  if (!h->d_bloom) return ECL_E_NOBLOOM;
  if (n == 0) return ECL_OK;
  HIPCHK(h, hipSetDevice(h->dev));
  int rc;
  h->last_held = h->last_total = 0, h->last_from_host = false;
  const u32 rcap = raw_cap_of(h, cap ? cap : 1);
  if ((rc = ensure_found(h, found_words_of(h, rcap))) != ECL_OK) return rc;
  u32 W;
  if ((rc = mul_setup_auto(h, n, &W)) != ECL_OK) return rc;
  const wtab gtab = wtab_make(h->d_multab, W);
  h->mul_seen += n;
  const size_t text_words = ((size_t)text_bytes + 3) / 4 + 2;  // two spare words: the gather reads one word past the last byte
  if (text_words > h->rawtext_cap || n > h->rawlines_cap) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream2));
    HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    if (h->prep_stream) HIPCHK(h, hipStreamSynchronize(h->prep_stream));  // (a call that failed between its pieces)
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
  if (!h->prep_stream) {
    int lo = 0, hi = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));  // (numerically lower = higher priority)
    HIPCHK(h, hipStreamCreateWithPriority(&h->prep_stream, hipStreamNonBlocking, hi));  // (lowest priority instead: no difference measured)
    for (int i = 0; i < MUL_NBUF; ++i) HIPCHK(h, hipEventCreateWithFlags(&h->ev_hashed[i], hipEventDisableTiming));
  }
  add_args a;
  memset(&a, 0, sizeof a);
  a.bloom = bloom_make(h->d_bloom, h->bloom_words);
  a.found = h->d_found, a.counter = h->d_counter, a.cap = rcap;
  u64* d_lines = h->d_rawlines;
  u32* d_flags = h->d_counter + 2;  // [0] a line outside the text, [1] a line beyond the part of the text that was on the device when it was hashed
  u32* d_text = h->d_rawtext;
  // The text goes with the pieces: in front of piece c's share of the table, the text up to the end of that share's LAST line - for a
  // table in the order of the text (any caller that cut lines out of a buffer front to back) every line is then on the device when its
  // piece is hashed, and the first piece starts after its own bytes instead of after all of them (300 MB for 2^24 pass phrases: 5 ms of
  // a 20 ms call).  The kernel checks it line by line; a table in another order raises flag [1], and the call is run again with the
  // whole text sent first.  Page-locked text and table (ecl_hip_alloc_host) go by DMA from where they are.
  u32 flags[2] = {0, 0}, cnt = 0;
  for (int pass = 0; pass < 2; ++pass) {
    const bool in_order = pass == 0;
    HIPCHK(h, hipMemsetAsync(h->d_counter, 0, 4 * sizeof(u32), h->stream));
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));  // the hashing writes the flags: behind their reset
    HIPCHK(h, hipStreamWaitEvent(h->prep_stream, h->ev_fork, 0));
    u32 have = 0;  // bytes of the text on the copy stream so far
    if (!in_order) {
      HIPCHK(h, hipMemcpyAsync(d_text, text, text_bytes, hipMemcpyHostToDevice, h->copy_stream));
      have = text_bytes;
    }
    rc = mul_pieces(
        h, n, gtab, a,
        [&](u32, u32 at, u32 m) -> int {  // (the copy stream is in order: the piece's table is behind its text)
          const u64 last = lines[at + m - 1], end = (last & 0xFFFFFFFFull) + (last >> 32);
          const u32 want = at + m == n || end > text_bytes ? text_bytes : (u32)end;
          if (want > have) {
            HIPCHK(h, hipMemcpyAsync((uint8_t*)d_text + have, text + have, want - have, hipMemcpyHostToDevice, h->copy_stream));
            have = want;
          }
          HIPCHK(h, hipMemcpyAsync(d_lines + at, lines + at, (size_t)m * 8, hipMemcpyHostToDevice, h->copy_stream));
          return ECL_OK;
        },
        [&](u32 b, u32 at, u32 m, hipEvent_t* ready) -> int {
          // The hashing has a stream of its own, of higher priority than the compute streams, and runs up to MUL_NBUF - 1 pieces ahead like
          // the copies: on the piece's compute stream it would sit between two k_mul_check launches that each fill the chip.
          HIPCHK(h, hipStreamWaitEvent(h->prep_stream, h->ev_copied[b], 0));
          hipLaunchKernelGGL(k_raw_scalars, dim3((m + 255) / 256), dim3(256), 0, h->prep_stream, d_text, text_bytes, have, d_lines + at, m, h->d_kbuf[b], d_flags);
          HIPCHK(h, hipEventRecord(h->ev_hashed[b], h->prep_stream));
          *ready = h->ev_hashed[b];
          return ECL_OK;
        });
    if (rc != ECL_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    rc = collect_found(h, cap, rcap, out, &cnt, false);  // (the one wait of the call; the flags come back with the counters)
    if (rc != ECL_OK && rc != ECL_E_OVERFLOW) return rc;
    flags[0] = h->pin_counter[2], flags[1] = h->pin_counter[3];
    if (flags[0] || !flags[1]) break;
  }
  if (flags[0]) {
    h->err = "mul_batch_raw: a line of the table lies outside the text";
    h->last_held = h->last_total = 0;
    return ECL_E_ARG;
  }
  *nout = cnt;
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
