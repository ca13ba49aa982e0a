/*
 * ecloop-hip — host program with ecloop's command-line surface (add / mul / rnd, blf-gen / blf-check,
 * -f -o -t -a -r -d -q -endo -seed -raw), driving MI355X GPUs through the C ABI of include/ecloop_hip.h.
 *
 * Plain C, links only libecloop_hip.so.  What lives here is what the reference keeps on the host side of the
 * boundary (citations into /root/reference): reading the filter (main.c:71-131), range / window
 * arguments (main.c:666-746), the job arithmetic of cmd_add (main.c:405-454), calc_priv (main.c:267-276), the
 * pk_verify_hash check of every hit (main.c:248-263; here one batched device call per scan chunk, ecl_hip_verify),
 * the found sink and the status line (main.c:134-203), cmd_mul's line reader (main.c:542-576), cmd_rnd's window
 * generator (main.c:580-662), blf-gen / blf-check (utils.c:400-529).  The program is organised around four objects of
 * its own - opts_t (the command line, parsed once), filter_t (bloom words + optional sorted list), report_t (found
 * sink, status line, pause state) and scan_t (one contiguous run of keys handed out to the device threads in chunks) -
 * not around the reference's ctx_t; formats, messages and counters are the reference's, byte for byte.
 * `-t N` selects the number of GPUs (one host thread per device context; default: all): a scan is range-partitioned,
 * no collective.  All curve and hash work of the search happens on the device; there is no CPU fallback.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <locale.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <poll.h>
#include <sys/stat.h>
#include <time.h>
#include <termios.h>
#include <unistd.h>

#include "ecloop_hip.h"

#define VERSION "0.5.0-hip"
#define GROUP_INV_SIZE 2048ull     /* main.c:17 */
#define MAX_JOB_SIZE (2ull << 20)  /* main.c:16 */
#define MAX_LINE_SIZE 1025         /* main.c:18 */
#define LAUNCH_KEYS (1ull << 32)   /* keys per device call on one GPU: one sweep of the default walk geometry */
#define MAX_GPUS 64

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
typedef struct { u64 w[4]; } sc; /* 256-bit scalar, little-endian limbs (the reference's fe) */

#include "cli_base.h"
#include "cli_filter.h"
#include "cli_report.h"
#include "cli_add.h"
#include "cli_mul.h"
#include "cli_rnd_blf.h"
#include "cli_keys.h"

/* ------------------------------------------------------------------------------------------- device bring-up */
/* Device contexts of a run: context g works on GPU (g mod shown) mod real, where `shown` is the -t count clamped to
   the visible GPUs and `real` the GPUs that exist.  `mul` opens TWO contexts per GPU - a batch is one synchronous
   ecl_hip_mul_batch call (scalars over PCIe, then the kernel), so the second context's copy runs under the first one's
   kernel and the other way round (290 -> 4xx M scalars/s from the same parsed stream, tools/bench_mul_cli.sh) - and the
   two contexts of a pair land on the SAME GPU, never on one the user did not ask for.  Returns the context count. */
static int context_devices(int cmd, int shown, int real, int dev_of[MAX_GPUS]) {
  int n = shown;
  if (cmd == CMD_MUL && 2 * n <= MAX_GPUS) n *= 2;
  for (int g = 0; g < n; ++g) dev_of[g] = (g % shown) % real;
  return n;
}
/* one context: open (self-test once per process), filter upload from the one host copy, optional list, walk
   buffers of the largest chunk this run will hand out; every step timed for ECLOOP_HIP_STATS */
typedef struct { run_t *run; int g, device; u32 flags; u64 reserve_keys; int rc; u64 t[5]; } bringup_t;
static void *bringup_thread(void *arg) {
  bringup_t *b = arg;
  run_t *run = b->run;
  ecl_hip **h = &run->dev[b->g];
  b->t[0] = us_now();
  int rc = ecl_hip_open(h, b->device, b->flags, run->cmd == CMD_MUL ? 0 : run->ord_offs);
  b->t[1] = us_now();
  if (rc == ECL_OK) rc = ecl_hip_set_bloom(*h, run->flt.words, run->flt.nwords);
  b->t[2] = us_now();
  if (rc == ECL_OK && run->flt.list) rc = ecl_hip_set_list(*h, (const uint32_t(*)[5])run->flt.list, run->flt.nlist);
  b->t[3] = us_now();
  if (rc == ECL_OK && b->reserve_keys) rc = ecl_hip_reserve(*h, b->reserve_keys, 4096);
  if (rc == ECL_OK && run->cmd == CMD_MUL && !run->parse_only) { /* window table, staging and record buffer of a usual batch (mul_flush's sizes) */
    u32 window = 0;
    const u32 n = (u32)mul_largest_batch(run, &window);
    if (window) rc = ecl_hip_set_mul_window(*h, window); /* the input's size is known (a file of 64-digit lines): the table that pays for it */
    if (rc == ECL_OK) rc = ecl_hip_reserve_mul(*h, n, n * 2 + 16 < MUL_HITS_CAP ? n * 2 + 16 : MUL_HITS_CAP);
  }
  b->t[4] = us_now();
  b->rc = rc;
  return NULL;
}
/* all contexts at once (every GPU over its own PCIe link), before the status clock starts; returns the seconds it took */
static double bring_up(run_t *run, int shown, int real) {
  const u64 t0 = ms_now();
  int dev_of[MAX_GPUS];
  run->ngpus = context_devices(run->cmd, shown, real, dev_of);
  u64 largest_call = 0;
  if (run->cmd != CMD_MUL) { /* keys of the largest device call: see scan_chunk() */
    sc keys;
    if (run->cmd == CMD_RND) keys = sc_u64(1ull << (run->ord_size < 21 ? 21 : run->ord_size > 62 ? 62 : run->ord_size));
    else {
      sc_subraw(&keys, &run->range_e, &run->range_s);
      for (u32 i = 0; i < run->ord_offs && i < 256; ++i) keys = sc_shr1(keys);
      keys = sc_add_u64_raw(keys, 4096);
    }
    bool fixed_shards;
    largest_call = scan_chunk(run, &keys, &fixed_shards);
    if (!(keys.w[1] | keys.w[2] | keys.w[3]) && keys.w[0] < largest_call) largest_call = keys.w[0];
  }
  const u32 flags = (run->a33 ? ECL_ADDR33 : 0) | (run->a65 ? ECL_ADDR65 : 0) | (run->endo ? ECL_ENDO : 0);
  pthread_t th[MAX_GPUS];
  bringup_t job[MAX_GPUS];
  for (int g = 0; g < run->ngpus; ++g) {
    job[g] = (bringup_t){run, g, dev_of[g], flags, largest_call, ECL_OK, {0}};
    pthread_create(&th[g], NULL, bringup_thread, &job[g]);
  }
  pthread_t pre;
  mul_prealloc_arg prea = {run, run->ngpus + 2};
  const bool prealloc = run->cmd == CMD_MUL && !run->parse_only && pthread_create(&pre, NULL, mul_prealloc, &prea) == 0;
  for (int g = 0; g < run->ngpus; ++g) pthread_join(th[g], NULL);
  if (prealloc) pthread_join(pre, NULL);
  for (int g = 0; g < run->ngpus; ++g)
    if (job[g].rc != ECL_OK) die_ecl(run, g, job[g].rc, "open");
  if (getenv("ECLOOP_HIP_STATS"))
    for (int g = 0; g < run->ngpus; ++g)
      printf("gpu %d bring-up: open %.1f ms, filter upload %.1f ms, list %.1f ms, reserve(%llu keys) %.1f ms\n", g,
             (job[g].t[1] - job[g].t[0]) / 1e3, (job[g].t[2] - job[g].t[1]) / 1e3, (job[g].t[3] - job[g].t[2]) / 1e3,
             (unsigned long long)largest_call, (job[g].t[4] - job[g].t[3]) / 1e3);
  return (ms_now() - t0) / 1000.0;
}

/* ------------------------------------------------------------------------------------------- main */
static void print_scalar_row(const char *name, const sc *v) {
  printf("%s: %016llx %016llx %016llx %016llx\n", name, (unsigned long long)v->w[3], (unsigned long long)v->w[2],
         (unsigned long long)v->w[1], (unsigned long long)v->w[0]);
}
int main(int argc, const char **argv) {
  setlocale(LC_NUMERIC, "");
  hexval_init();
#if defined(__x86_64__)
  have_ssse3 = __builtin_cpu_supports("ssse3");
  have_avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && !getenv("ECLOOP_HIP_NO_AVX512");
  have_avx2 = __builtin_cpu_supports("avx2") && !getenv("ECLOOP_HIP_NO_AVX2"); /* (the variable: tests run the SSSE3 form on a CPU that has both) */
#endif
  static run_t run;
  opts_t *o = &run.opt;
  opts_parse(o, argc, argv);
  const char *verb = argc > 1 ? argv[1] : "";
  /* commands that need no search context */
  if (!strcmp(verb, "blf-gen")) return cmd_blf_gen(o, argv[0]), 0;
  if (!strcmp(verb, "blf-check")) return cmd_blf_check(o, argc, argv), 0;
  if (!strcmp(verb, "parse")) { /* hidden: `mul`'s text front end alone (no GPU), for the parser tests */
    run.cmd = CMD_MUL, run.parse_only = true, run.ngpus = 1, run.bin = o->bin;
    report_init(&run.rep, NULL, true);
    cmd_mul(&run);
    return 0;
  }
  const bool plan_only = !strcmp(verb, "plan"); /* hidden: the job arithmetic of `add` / `rnd`, the context -> GPU map; no GPU */
  run.cmd = plan_only ? (o->rnd_jobs ? CMD_RND : CMD_ADD) /* `plan -rnd`: rnd's window rules (offset drawn or clamped) and full-size jobs */
            : !strcmp(verb, "add") ? CMD_ADD : !strcmp(verb, "mul") ? CMD_MUL : !strcmp(verb, "rnd") ? CMD_RND : CMD_NIL;
  if (run.cmd == CMD_NIL) {
    if (o->version) printf("ecloop-hip v%s\n", VERSION);
    else usage(argv[0]);
    return 0;
  }
  run.colour = isatty(fileno(stdout));
  if (o->seed) /* a seeded run draws from rand()'s stream (the reference free()s an argv pointer here and aborts, main.c:800-805) */
    run.seeded = true, seeded_start(o->seed);
  if (!plan_only) filter_open(&run.flt, o->filter);
  if (o->quiet && !o->outfile && !plan_only) { fprintf(stderr, "quiet mode chosen without output file\n"); exit(1); }
  run.a33 = o->addr ? strchr(o->addr, 'c') != NULL : true, run.a65 = o->addr && strchr(o->addr, 'u');
  if (!run.a33 && !run.a65) run.a33 = true; /* main.c:825-827 */
  run.endo = o->endo && run.cmd != CMD_MUL, run.bin = o->bin && run.cmd == CMD_MUL;
  report_init(&run.rep, o->outfile, o->quiet);
  range_from_option(o->range, &run.range_s, &run.range_e);
  window_from_option(&run);
  run.stride_k = sc_pow2(run.cmd == CMD_MUL ? 0 : run.ord_offs);

  if (plan_only) {
    run.ngpus = (int)opt_number(o->gpus, 1);
    if (o->visible) { /* the context -> GPU map of `-t N` on a box with that many GPUs */
      int real = (int)opt_number(o->visible, 1), shown = run.ngpus > real ? real : run.ngpus, dev_of[MAX_GPUS];
      int n = context_devices(o->as_mul ? CMD_MUL : CMD_ADD, shown, real, dev_of);
      printf("contexts %d gpus %d devices", n, shown);
      for (int g = 0; g < n; ++g) printf(" %d", dev_of[g]);
      printf("\n");
      return 0;
    }
    scan_t sn;
    scan_plan(&run, run.range_s, run.range_e, o->rnd_jobs, &sn);
    printf("stride_bits %u ", sc_bitlen(&run.stride_k) - 1);
    printf("ord_offs %u ord_size %u hashed %016llx%016llx%016llx%016llx status_total %llu chunk %llu\n", run.ord_offs, run.ord_size,
           (unsigned long long)sn.hashed.w[3], (unsigned long long)sn.hashed.w[2], (unsigned long long)sn.hashed.w[1],
           (unsigned long long)sn.hashed.w[0], (unsigned long long)sn.status_total, (unsigned long long)sn.chunk);
    return 0;
  }
  const int real = ecl_hip_device_count();
  int usable = real;
  /* test hook: ECLOOP_HIP_SHARE_GPU=N runs N device threads over the GPUs that exist, so the sharding / merging logic
     can be exercised on a one-GPU box */
  const char *share = getenv("ECLOOP_HIP_SHARE_GPU");
  if (real > 0 && share && atoi(share) > 0) usable = atoi(share);
  if (real <= 0) { fprintf(stderr, "no MI355X GPU visible (the search path has no CPU fallback)\n"); return 1; }
  u64 asked = opt_number(o->gpus, (u64)usable);
  int shown = (int)(asked < 1 ? 1 : asked > (u64)usable ? (u64)usable : asked);
  if (shown > MAX_GPUS) shown = MAX_GPUS;
  const double setup_s = bring_up(&run, shown, real);

  printf("gpus: %d ~ addr33: %d ~ addr65: %d ~ endo: %d | filter: ", shown, run.a33, run.a65, run.endo);
  if (run.flt.list) printf("list (%'llu)\n", (unsigned long long)run.flt.nlist);
  else printf("bloom\n");
  if (run.cmd == CMD_ADD) print_scalar_row("range_s", &run.range_s), print_scalar_row("range_e", &run.range_e);
  printf("setup: %.2fs (%d device context%s opened in parallel, %.0f MB filter uploaded, walk buffers reserved)\n", setup_s, run.ngpus,
         run.ngpus == 1 ? "" : "s", run.flt.nwords * 8 / 1e6);
  puts("----------------------------------------");
  fflush(stdout);
  signal(SIGINT, on_sigint);
  keys_listen(&run.rep);
  switch (run.cmd) {
  case CMD_ADD: cmd_add(&run); break;
  case CMD_MUL: cmd_mul(&run); break;
  default: cmd_rnd(&run); break;
  }
  for (int g = 0; g < run.ngpus; ++g) ecl_hip_close(run.dev[g]);
  return 0;
}
