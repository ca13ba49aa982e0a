/* ecloop_gpu_binding.h - the reference-side binding of include/ecloop_hip.h: what a maintainer of vladkens/ecloop adds
 * to main.c so that its own host program (argument parsing, load_filter, scheduler, calc_priv, pk_verify_hash,
 * ctx_write_found, status line - all unchanged) runs its hot path on MI355X GPUs.
 *
 * This file is OUR code (no reference text).  It is textually included into a patched temporary copy of the
 * reference's main.c by oracle/ref_binding/build_ecloop_gpu.py, right before `batch_add` (main.c:349), i.e. after the
 * definitions it uses: ctx_t (+ the three members the patch adds), fe / h160_t, compare_160 (addr.c:18-26),
 * calc_priv (main.c:267), pk_verify_hash (main.c:248), ctx_write_found (main.c:182), GROUP_INV_SIZE (main.c:17).
 * The patch itself is six one-line edits (see the script); everything else lives here.
 *
 * Test infrastructure as far as this repository goes (it proves the drop-in claim of INTEGRATION.md by execution:
 * tests/test_gpu_ref_binding.py runs the resulting oracle/_ref/ecloop_gpu against the reference's golden outputs);
 * the product is the library it links. */
#include "ecloop_hip.h"
/* (the reference's u64 is unsigned long long, uint64_t is unsigned long here: same 64 bits, hence the pointer casts) */

#define GPU_MAX 16

static void gpu_die(ctx_t *ctx, int g, const char *what, int rc) {
  fprintf(stderr, "gpu %d: %s: %s (%s)\n", g, what, ecl_hip_strerror(rc), ctx->gpu[g] ? ecl_hip_last_error(ctx->gpu[g]) : "");
  exit(1);
}

/* once per run, after ord_offs has its final value (cmd_rnd lowers it, main.c:620) and before the workers start:
   one context per GPU, -t = number of contexts (one host thread per context, main.c:445-447 creates them as before) */
static void gpu_init(ctx_t *ctx) {
  if (ctx->gpu_count) return;
  const int real = ecl_hip_device_count();
  if (real <= 0) { fprintf(stderr, "no GPU visible\n"); exit(1); }
  /* contexts = GPUs, unless ECLOOP_GPU_CONTEXTS=N asks for N of them (context g on device g mod GPUs): several worker threads
     per GPU overlap one job's host side (hit records, calc_priv, pk_verify_hash, the sink) with the next job's kernel - and it
     is how a one-GPU box runs the -t N code path (gpu_claim, per-thread hit buffers) */
  int n = real;
  const char *want = getenv("ECLOOP_GPU_CONTEXTS");
  if (want && atoi(want) > 0) n = atoi(want);
  if (n > GPU_MAX) n = GPU_MAX;
  if ((size_t)n > ctx->threads_count) n = (int)ctx->threads_count;
  ctx->threads_count = (size_t)n; /* more host threads than contexts would only queue behind each other */
  const uint32_t flags = (ctx->check_addr33 ? ECL_ADDR33 : 0) | (ctx->check_addr65 ? ECL_ADDR65 : 0) | (ctx->use_endo ? ECL_ENDO : 0);
  for (int g = 0; g < n; ++g) {
    int rc = ecl_hip_open(&ctx->gpu[g], g % real, flags, ctx->cmd == CMD_MUL ? 0 : ctx->ord_offs);
    if (rc != ECL_OK) gpu_die(ctx, g, "ecl_hip_open", rc);
    rc = ecl_hip_set_bloom(ctx->gpu[g], (const uint64_t *)ctx->blf.bits, ctx->blf.size); /* the very words blf_has reads (utils.c:277-288) */
    if (rc != ECL_OK) gpu_die(ctx, g, "ecl_hip_set_bloom", rc);
  }
  ctx->gpu_count = n;
}

/* each worker thread of a cmd_add / cmd_rnd / cmd_mul run takes the next device */
static int gpu_claim(ctx_t *ctx) { return (int)(__atomic_fetch_add(&ctx->gpu_next, 1, __ATOMIC_RELAXED) % (unsigned)ctx->gpu_count); }

/* hit records of one call; grown to the count the library reports when a call overflows (ECL_E_OVERFLOW) */
typedef struct gpu_hits_t { ecl_found *rec; uint32_t cap; } gpu_hits_t;
static void gpu_hits_reserve(gpu_hits_t *h, uint32_t cap) { /* keeps what the buffer holds */
  if (cap <= h->cap) return;
  h->rec = realloc(h->rec, (size_t)cap * sizeof(ecl_found)), h->cap = cap;
  if (!h->rec) { fprintf(stderr, "out of memory for %u hit records\n", cap); exit(1); }
}

/* second half of ctx_check_hash (main.c:212-216): the device has done the bloom probe, the sorted list is confirmed here */
static bool gpu_confirm(const ctx_t *ctx, const uint32_t h160[5]) {
  if (ctx->to_find_hashes == NULL) return true;
  return bsearch(h160, ctx->to_find_hashes, ctx->to_find_count, sizeof(h160_t), compare_160) != NULL;
}

/* replaces  batch_add(ctx, pk, ctx->job_size)  in cmd_add_worker (main.c:430): the same keys - batch_add hashes whole
   groups, ceil(iterations / 2048) * 2048 of them (main.c:368,401) - and the same sink for every hit */
static void gpu_batch_add(ctx_t *ctx, int g, const fe pk, size_t iterations) {
  static __thread gpu_hits_t hits;
  const uint64_t nkeys = (iterations + GROUP_INV_SIZE - 1) / GROUP_INV_SIZE * GROUP_INV_SIZE;
  gpu_hits_reserve(&hits, 1u << 16);
  uint32_t n = 0;
  /* where this scan stops handing out jobs (cmd_add_worker leaves its loop at range_s >= range_e, main.c:420; cmd_rnd sets a new range_e
     per window before it starts the workers, main.c:606-617): lets the library look ahead over the 2^21-key jobs without passing the end */
  static int tell_end = -1; /* ECLOOP_GPU_NO_SCAN_END=1: measure the look-ahead without the hint (tools/bench_ref_binding.py) */
  if (tell_end < 0) tell_end = getenv("ECLOOP_GPU_NO_SCAN_END") == NULL;
  if (tell_end) (void)ecl_hip_set_scan_end(ctx->gpu[g], (const uint64_t *)ctx->range_e);
  int rc = ecl_hip_add_range(ctx->gpu[g], (const uint64_t *)pk, nkeys, hits.rec, hits.cap, &n);
  if (rc == ECL_E_OVERFLOW) { /* e.g. an all-ones filter: every hash is a hit.  The device kept the records that did not fit
                                 (up to max(cap, 2^20) per call): read them; only beyond that is the job run again */
    const uint32_t had = hits.cap;
    uint32_t got = 0;
    gpu_hits_reserve(&hits, n);
    rc = ecl_hip_fetch_found(ctx->gpu[g], had, hits.rec + had, n - had, &got);
    if (rc == ECL_OK && got != n - had) rc = ecl_hip_add_range(ctx->gpu[g], (const uint64_t *)pk, nkeys, hits.rec, hits.cap, &n);
  }
  if (rc != ECL_OK) gpu_die(ctx, g, "ecl_hip_add_range", rc);
  for (uint32_t i = 0; i < n; ++i) {
    const ecl_found *f = &hits.rec[i];
    if (!gpu_confirm(ctx, f->h160)) continue;
    fe ck;
    calc_priv(ck, pk, ctx->stride_k, f->key_offset, f->endo); /* main.c:267 */
    pk_verify_hash(ck, f->h160, f->compressed, f->endo);      /* main.c:248: the CPU re-derives every reported key */
    ctx_write_found(ctx, f->compressed ? "addr33" : "addr65", f->h160, ck);
  }
}

/* replaces  ec_gtable_mul x count + ec_jacobi_grprdc + check_found_mul  in cmd_mul_worker (main.c:531-534) */
static void gpu_mul_job(ctx_t *ctx, int g, const fe *pk, size_t count) {
  static __thread gpu_hits_t hits;
  gpu_hits_reserve(&hits, 2 * GROUP_INV_SIZE);
  uint32_t n = 0;
  int rc = ecl_hip_mul_batch(ctx->gpu[g], (const uint64_t(*)[4])pk, (uint32_t)count, hits.rec, hits.cap, &n);
  if (rc != ECL_OK) gpu_die(ctx, g, "ecl_hip_mul_batch", rc);
  for (uint32_t i = 0; i < n; ++i) {
    const ecl_found *f = &hits.rec[i];
    if (!gpu_confirm(ctx, f->h160)) continue;
    ctx_write_found(ctx, f->compressed ? "addr33" : "addr65", f->h160, pk[f->key_offset]);
  }
}
