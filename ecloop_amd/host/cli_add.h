/* cli_add.h - `add`: the scan of a range - job arithmetic of cmd_add, hand-out to the device threads, hits to the sink.
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
/* ------------------------------------------------------------------------------------------- add */
/* One scan = the contiguous run of keys  rs + i*stride, i < hashed  (what cmd_add's jobs hash, main.c:405-454).  The
   device threads pull chunks of it from a shared counter, like the reference's workers pull 2^21-key jobs
   (main.c:418-431): a GPU that sustains a few percent more clock simply takes more chunks, and a scan of any length
   (the default range 0x800:p included) streams through without its key count having to fit 64 bits. */
typedef struct {
  run_t *run;
  sc rs;             /* first scalar */
  sc hashed;         /* keys to hash (256-bit: `add` without -r walks ~2^256 / stride keys) */
  sc next;           /* keys handed out so far */
  u64 chunk;         /* keys per hand-out = per device call */
  bool fixed;        /* one contiguous shard per device thread (chunk g belongs to thread g) instead of the shared counter */
  u64 status_total;  /* what the status counter must have gained at the end (0: not representable, add as we go) */
  u64 status_given;
  int shards_left;   /* fixed shards not yet taken */
  u64 mult;          /* status units per key when status_total is 0 */
  pthread_mutex_t mu;
} scan_t;
typedef struct { scan_t *scan; int g; } scan_worker_t;

static sc sc_add_u64_raw(sc a, u64 v) {
  sc b = sc_u64(v), r;
  sc_addraw(&r, &a, &b);
  return r;
}
/* scalar of key number `off` (256-bit count): rs + off * stride (mod n); stride is a power of two */
static sc scan_scalar(const run_t *run, const sc *rs, const sc *off) {
  sc o = sc_reduce(*off); /* off < 2^256 < 2n */
  return sc_add(sc_reduce(*rs), sc_mul(run->stride_k, o));
}

static void *scan_worker(void *arg) {
  scan_worker_t *w = arg;
  scan_t *sn = w->scan;
  run_t *run = sn->run;
  u32 cap = 4096;
  ecl_found *buf = malloc(sizeof(ecl_found) * cap);
  if (!(sn->hashed.w[1] | sn->hashed.w[2] | sn->hashed.w[3])) { /* where the hand-out stops: the library may look ahead over small jobs up to here */
    sc end = scan_scalar(run, &sn->rs, &sn->hashed);
    if (sc_cmp(&end, &sn->rs) > 0) ecl_hip_set_scan_end(run->dev[w->g], end.w);
  }
  for (bool first = true;; first = false) {
    pthread_mutex_lock(&sn->mu);
    sc lo = sn->next, left;
    if (sn->fixed) { /* thread g's own shard: keys [g * chunk, (g + 1) * chunk) of the scan, one device call */
      lo = sc_u64(sn->chunk * (u64)w->g);
      if (!first || sc_cmp(&lo, &sn->hashed) >= 0) { pthread_mutex_unlock(&sn->mu); break; }
    } else if (sc_cmp(&lo, &sn->hashed) >= 0) { pthread_mutex_unlock(&sn->mu); break; }
    sc_subraw(&left, &sn->hashed, &lo);
    u64 n = (left.w[1] | left.w[2] | left.w[3]) || left.w[0] > sn->chunk ? sn->chunk : left.w[0];
    sc upto_key = sc_add_u64_raw(lo, n);
    if (sn->fixed) sn->shards_left--;
    else sn->next = upto_key;
    bool last = sn->fixed ? sn->shards_left == 0 : sc_cmp(&sn->next, &sn->hashed) >= 0;
    /* status counter: the reference adds job_size (x6 with endo) per job (main.c:431); spread over the chunks */
    u64 st;
    if (!sn->status_total) st = n * sn->mult;
    else if (last) st = sn->status_total - sn->status_given;
    else if (sn->fixed) st = (u64)((u128)sn->status_total * n / sn->hashed.w[0]); /* this shard's share; the last one rounds up */
    else {
      u128 done = (u128)sn->next.w[0]; /* status_total != 0 implies hashed < 2^63 */
      u64 upto = (u64)((u128)sn->status_total * done / sn->hashed.w[0]);
      st = upto - sn->status_given;
    }
    sn->status_given += st;
    pthread_mutex_unlock(&sn->mu);

    sc s = scan_scalar(run, &sn->rs, &lo);
    u32 cnt = 0;
    int rc;
    for (;;) {
      rc = ecl_hip_add_range(run->dev[w->g], s.w, n, buf, cap, &cnt);
      if (rc != ECL_E_OVERFLOW) break;
      /* dense filter: the device kept the records that did not fit (up to max(cap, 2^20) per call) - read them; only a call with
         more hits than that is run again with a buffer that fits */
      u32 had = cap, got = 0;
      cap = cnt, buf = realloc(buf, sizeof(ecl_found) * cap);
      if (!buf) { fprintf(stderr, "out of memory for %u hit records\n", cap); exit(1); }
      rc = ecl_hip_fetch_found(run->dev[w->g], had, buf + had, cnt - had, &got);
      if (rc != ECL_OK || got == cnt - had) break;
    }
    if (rc != ECL_OK) die_ecl(run, w->g, rc, "add_range");
    u32 kept = 0;
    sc *pks = cnt ? malloc(sizeof(sc) * cnt) : NULL;
    for (u32 i = 0; i < cnt; ++i) {
      if (!filter_confirms(&run->flt, buf[i].h160)) continue;
      pks[kept] = calc_priv(s, run->stride_k, buf[i].key_offset, buf[i].endo);
      buf[kept++] = buf[i];
    }
    verify_hits(run, w->g, pks, buf, kept);
    for (u32 i = 0; i < kept; ++i) report_hit(&run->rep, buf[i].compressed, buf[i].h160, &pks[i]);
    free(pks);
    report_progress(&run->rep, st);
  }
  free(buf);
  return NULL;
}

/* keys per hand-out.  One GPU: whole sweeps of the walk (2^32 keys at the default geometry), which continue on the
   device without re-initialisation.  Several GPUs, a scan of at most 2^33 keys (one 2^32-key range - the configuration the
   headline metric is quoted on -, a `rnd` window): ONE contiguous shard per GPU, a single device call each (*fixed) - a call of
   2^29 keys runs 2 % below a 2^30-key one and every call pays its re-positioning, so halving the shards to even out clocks that
   differ by a percent or two loses more than it wins.  Longer scans: the shared counter, at least two chunks per GPU so that uneven
   clocks even out, at least 2^27 keys (10 ms of kernel against ~0.4 ms of per-call set-up), at most 2^30. */
static u64 scan_chunk(const run_t *run, const sc *hashed, bool *fixed) {
  *fixed = false;
  { /* ECLOOP_HIP_JOB_KEYS=N (measurement): hand the scan out in jobs of N keys from the shared counter whatever its length - with
       N = 2097152 the reference's own scheduler (MAX_JOB_SIZE, main.c:16,418-431); the library's look-ahead is what keeps the rate */
    const char *e = getenv("ECLOOP_HIP_JOB_KEYS");
    const u64 j = e ? strtoull(e, NULL, 0) : 0;
    if (j >= GROUP_INV_SIZE) return j / GROUP_INV_SIZE * GROUP_INV_SIZE;
  }
  if (run->ngpus <= 1) return LAUNCH_KEYS;
  if (hashed->w[1] | hashed->w[2] | hashed->w[3]) return 1ull << 30;
  if (hashed->w[0] <= (1ull << 33) && !getenv("ECLOOP_HIP_SHARED_COUNTER")) {
    u64 c = (hashed->w[0] + (u64)run->ngpus - 1) / (u64)run->ngpus;
    *fixed = true;
    return (c + GROUP_INV_SIZE - 1) / GROUP_INV_SIZE * GROUP_INV_SIZE;
  }
  u64 c = (hashed->w[0] + 2 * (u64)run->ngpus - 1) / (2 * (u64)run->ngpus);
  c = (c + GROUP_INV_SIZE - 1) / GROUP_INV_SIZE * GROUP_INV_SIZE;
  if (c < (1ull << 27)) c = 1ull << 27;
  if (c > (1ull << 30)) c = 1ull << 30;
  return c;
}

/* The plan of one scan: cmd_add (main.c:437-454) over [range_s, range_e) hashes the contiguous run of `hashed` keys from
   range_s and adds `status_total` to the status counter (0: too long to count, added chunk by chunk). */
static void scan_plan(run_t *run, sc rs, sc re, bool full_jobs, scan_t *sn) {
  sc span;
  sc_subraw(&span, &re, &rs);
  /* cmd_rnd always uses MAX_JOB_SIZE jobs, even for a narrower window (main.c:624) */
  bool small = !full_jobs && !(span.w[1] | span.w[2] | span.w[3]) && span.w[0] < MAX_JOB_SIZE;
  u64 job = small ? span.w[0] : MAX_JOB_SIZE; /* main.c:442 */
  /* njobs = ceil(span / (job * stride)) (main.c:420-427): the counter steps by job*stride until it reaches range_e */
  sc njobs = {{0, 0, 0, 0}};
  if (small && run->ord_offs == 0) njobs = sc_u64(1);
  else if (!small) {
    unsigned sh = 21 + run->ord_offs; /* job * stride = 2^sh */
    if (sh >= 256) njobs = sc_u64(1);
    else {
      for (unsigned b = sh; b < 256; ++b)
        if ((span.w[b >> 6] >> (b & 63)) & 1) njobs.w[(b - sh) >> 6] |= 1ULL << ((b - sh) & 63);
      bool rem = false;
      for (unsigned b = 0; b < sh; ++b)
        if ((span.w[b >> 6] >> (b & 63)) & 1) rem = true;
      if (rem) njobs = sc_add_u64_raw(njobs, 1);
    }
  } else { /* a sub-2^21 job with a stride: step like the reference's counter (at most 2^21 / 2^offs + 1 steps) */
    sc inc = sc_mul(run->stride_k, sc_u64(job)), cur = rs;
    u64 n = 0;
    while (sc_cmp(&cur, &re) < 0 && n < (1u << 22)) {
      sc nx;
      n++;
      if (sc_addraw(&nx, &cur, &inc)) break;
      cur = nx;
    }
    njobs = sc_u64(n);
  }
  u64 per_job = (job + GROUP_INV_SIZE - 1) / GROUP_INV_SIZE * GROUP_INV_SIZE;
  memset(sn, 0, sizeof *sn);
  sn->run = run, sn->rs = rs, sn->mult = run->endo ? 6 : 1;
  if (!(njobs.w[1] | njobs.w[2] | njobs.w[3]) && njobs.w[0] < (1ull << 40)) {
    /* the usual case: hashed = (njobs-1)*job + ceil(job/2048)*2048 keys, status counter = njobs*job (x6 with endo) */
    sn->hashed = sc_u64((njobs.w[0] - 1) * job + per_job);
    sn->status_total = njobs.w[0] * job * sn->mult;
  } else {
    /* astronomically long (e.g. the default range): hashed = njobs * 2^21 as a 256-bit count; it will not finish,
       and the status counter advances by the keys of every chunk */
    sc h = njobs;
    for (int i = 0; i < 21; ++i) sc_addraw(&h, &h, &h); /* njobs < 2^235 here: no wrap */
    sn->hashed = h;
  }
  sn->chunk = scan_chunk(run, &sn->hashed, &sn->fixed);
  if (sn->fixed) /* shards that hold keys: a scan shorter than ngpus * 2048 keys leaves the last threads without one */
    sn->shards_left = (int)((sn->hashed.w[0] + sn->chunk - 1) / sn->chunk);
}

/* one scan, spread over the GPUs */
static void scan_range(run_t *run, sc rs, sc re, bool full_jobs) {
  scan_t sn;
  scan_plan(run, rs, re, full_jobs, &sn);
  pthread_mutex_init(&sn.mu, NULL);
  pthread_t th[MAX_GPUS];
  scan_worker_t ws[MAX_GPUS];
  for (int g = 0; g < run->ngpus; ++g) {
    ws[g] = (scan_worker_t){&sn, g};
    pthread_create(&th[g], NULL, scan_worker, &ws[g]);
  }
  for (int g = 0; g < run->ngpus; ++g) pthread_join(th[g], NULL);
  pthread_mutex_destroy(&sn.mu);
}

/* ECLOOP_HIP_STATS: where each device context's time went - calls of the search kernel, and what the non-contiguous ones paid
   for re-positioning the walk */
static void print_device_stats(run_t *run) {
  if (!getenv("ECLOOP_HIP_STATS")) return;
  for (int g = 0; g < run->ngpus; ++g) {
    double kernel_ms = 0, setup_ms = 0;
    u64 launches = 0, keys = 0, setups = 0;
    ecl_hip_get_timing(run->dev[g], &kernel_ms, &launches, &keys);
    ecl_hip_get_setup_timing(run->dev[g], &setup_ms, &setups);
    printf("gpu %d: %llu launches, %.3f ms in the search kernel, %llu set-ups, %.3f ms in set-up kernels (%.2f %%)\n", g,
           (unsigned long long)launches, kernel_ms, (unsigned long long)setups, setup_ms,
           kernel_ms > 0 ? 100.0 * setup_ms / (kernel_ms + setup_ms) : 0.0);
  }
}

static void cmd_add(run_t *run) {
  report_restart_clock(&run->rep);
  scan_range(run, run->range_s, run->range_e, false);
  print_device_stats(run);
  report_close(&run->rep);
}
