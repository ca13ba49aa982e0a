/* cli_extras.h - pause / resume keys; the reference's bench / bench-gtable / mult-verify commands.
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
/* ------------------------------------------------------------------------------------------- pause / resume keys */
/* 'p' parks the device threads at their next progress report, 'r' lets them go on (main.c:874-888; the reference's raw
   /dev/tty listener is utils.c:546-624).  Keys come from the controlling terminal in non-canonical mode, or from the
   path in ECLOOP_HIP_TTY (a FIFO works: containers without ptys); without either nothing is installed.  One detached
   thread polls the descriptor; the terminal's settings are put back at exit. */
static struct { int fd; bool is_terminal; struct termios saved; report_t *rep; } keys = {-1, false, {0}, NULL};
static void keys_restore(void) {
  if (keys.fd < 0) return;
  if (keys.is_terminal) tcsetattr(keys.fd, TCSANOW, &keys.saved);
  close(keys.fd), keys.fd = -1;
}
static void *keys_thread(void *unused) {
  (void)unused;
  struct pollfd p = {keys.fd, POLLIN, 0};
  for (char key; p.fd >= 0 && poll(&p, 1, 200) >= 0; p.fd = keys.fd)
    if ((p.revents & POLLIN) && read(p.fd, &key, 1) == 1 && (key == 'p' || key == 'r')) report_pause(keys.rep, key == 'p');
  return NULL;
}
static void keys_listen(report_t *rep) {
  const char *path = getenv("ECLOOP_HIP_TTY");
  keys.fd = open(path ? path : "/dev/tty", (path ? O_RDWR : O_RDONLY) | O_NONBLOCK);
  if (keys.fd < 0) return;
  keys.rep = rep;
  keys.is_terminal = tcgetattr(keys.fd, &keys.saved) == 0;
  if (!keys.is_terminal && !path) { close(keys.fd), keys.fd = -1; return; }
  atexit(keys_restore);
  if (keys.is_terminal) {
    struct termios t = keys.saved;
    t.c_lflag &= ~(tcflag_t)(ICANON | ECHO);
    tcsetattr(keys.fd, TCSANOW, &t);
  }
  pthread_t th;
  if (!pthread_create(&th, NULL, keys_thread, NULL)) pthread_detach(th);
}
static void on_sigint(int sig) { /* main.c:867-872: what was printed so far reaches its destination, then out */
  fflush(stderr), fflush(stdout);
  fputc('\n', stdout);
  exit(sig);
}

/* `bench` (the reference's `bench` / `bench-gtable`, lib/bench.c, time its CPU primitives): here the device paths,
   through the C ABI, with an empty filter: keys/s of the add walk per address / endo selection, scalars/s of mul. */
static int run_bench(const opts_t *o) {
  if (ecl_hip_device_count() <= 0) { fprintf(stderr, "no MI355X GPU visible (the search path has no CPU fallback)\n"); return 1; }
  u64 lg = opt_number(o->count, 31);
  if (lg < 20 || lg > 36) lg = 31;
  static const struct { const char *name; u32 flags; } cfg[] = {
      {"add -a c", ECL_ADDR33}, {"add -a u", ECL_ADDR65}, {"add -a cu", ECL_ADDR33 | ECL_ADDR65},
      {"add -a c -endo", ECL_ADDR33 | ECL_ENDO}, {"add -a cu -endo", ECL_ADDR33 | ECL_ADDR65 | ECL_ENDO}};
  u64 zeros[64] = {0};
  const u64 start[4] = {0x100000000ull, 0, 0, 0};
  ecl_found hit[16];
  for (size_t c = 0; c < sizeof cfg / sizeof cfg[0]; ++c) {
    ecl_hip *d = NULL;
    int rc = ecl_hip_open(&d, 0, cfg[c].flags, 0);
    if (rc == ECL_OK) rc = ecl_hip_set_bloom(d, zeros, 64);
    u64 n = 1ull << (lg - ((cfg[c].flags & ECL_ENDO) ? 2 : 0));
    u32 cnt = 0;
    if (rc == ECL_OK) rc = ecl_hip_add_range(d, start, n, hit, 16, &cnt); /* warm-up: table, centres, scratch */
    if (rc == ECL_OK) rc = ecl_hip_reset_timing(d);
    u64 t0 = ms_now();
    if (rc == ECL_OK) rc = ecl_hip_add_range(d, start, n, hit, 16, &cnt);
    u64 t1 = ms_now();
    double kms = 0;
    u64 launches = 0, keys = 0;
    if (rc == ECL_OK) rc = ecl_hip_get_timing(d, &kms, &launches, &keys);
    if (rc != ECL_OK) { fprintf(stderr, "[!] bench %s: %s (%s)\n", cfg[c].name, ecl_hip_strerror(rc), d ? ecl_hip_last_error(d) : ""); return 1; }
    int hashes = ((cfg[c].flags & ECL_ADDR33) ? 1 : 0) + ((cfg[c].flags & ECL_ADDR65) ? 1 : 0);
    if (cfg[c].flags & ECL_ENDO) hashes *= 6;
    printf("%-18s 2^%-2d keys: %9.2f Mkeys/s (kernel %9.2f) ~ %9.2f M hash160/s\n", cfg[c].name,
           (int)(lg - ((cfg[c].flags & ECL_ENDO) ? 2 : 0)), n / ((t1 - t0 ? t1 - t0 : 1) / 1000.0) / 1e6, keys / (kms / 1000.0) / 1e6,
           hashes * (keys / (kms / 1000.0)) / 1e6);
    fflush(stdout);
    ecl_hip_close(d);
  }
  { /* mul: 2^22 pseudo-random scalars, addr33 + addr65 */
    ecl_hip *d = NULL;
    u32 n = 1u << 22, cnt = 0;
    u64 (*ks)[4] = malloc((size_t)n * 32);
    u64 x = 0x9E3779B97F4A7C15ull;
    for (u32 i = 0; i < n; ++i)
      for (int j = 0; j < 4; ++j) x ^= x << 13, x ^= x >> 7, x ^= x << 17, ks[i][j] = x;
    int rc = ecl_hip_open(&d, 0, ECL_ADDR33 | ECL_ADDR65, 0);
    if (rc == ECL_OK) rc = ecl_hip_set_bloom(d, zeros, 64);
    if (rc == ECL_OK) rc = ecl_hip_mul_batch(d, ks, n, hit, 16, &cnt); /* warm-up: builds the window table */
    u64 t0 = ms_now();
    for (int r = 0; r < 4 && rc == ECL_OK; ++r) rc = ecl_hip_mul_batch(d, ks, n, hit, 16, &cnt);
    u64 t1 = ms_now();
    if (rc != ECL_OK) { fprintf(stderr, "[!] bench mul: %s\n", ecl_hip_strerror(rc)); return 1; }
    printf("%-18s 2^22 keys: %9.2f M it/s (scalars copied from host memory)\n", "mul -a cu", 4.0 * n / ((t1 - t0 ? t1 - t0 : 1) / 1000.0) / 1e6);
    free(ks);
    ecl_hip_close(d);
  }
  return 0;
}

/* `bench-gtable` (lib/bench.c:114-141: table build time, multiplications per second and memory for window widths 8..22):
   the same sweep over the device's window tables - here the width is a run-time property (ecl_hip_set_mul_window), so one
   process measures them all; same line format, "gen" = first batch minus a later one (table build + check), 2^22 scalars. */
static int run_bench_gtable(void) {
  if (ecl_hip_device_count() <= 0) { fprintf(stderr, "no MI355X GPU visible (the search path has no CPU fallback)\n"); return 1; }
  const u32 n = 1u << 22;
  u64 (*ks)[4] = ecl_hip_alloc_host((size_t)n * 32);
  if (!ks) { fprintf(stderr, "[!] bench-gtable: no page-locked memory\n"); return 1; }
  u64 x = 42;
  for (u32 i = 0; i < n; ++i)
    for (int j = 0; j < 4; ++j) x ^= x << 13, x ^= x >> 7, x ^= x << 17, ks[i][j] = x;
  u64 zeros[64] = {0};
  ecl_found hit[16];
  for (u32 w = 8; w <= 26; w += 2) {
    ecl_hip *d = NULL;
    u32 cnt = 0;
    int rc = ecl_hip_open(&d, 0, ECL_ADDR33, 0);
    if (rc == ECL_OK) rc = ecl_hip_set_bloom(d, zeros, 64);
    if (rc == ECL_OK) rc = ecl_hip_set_mul_window(d, w);
    const u64 t0 = us_now();
    if (rc == ECL_OK) rc = ecl_hip_mul_batch(d, ks, n, hit, 16, &cnt);
    const u64 t1 = us_now();
    const int reps = 8;
    for (int r = 0; r < reps && rc == ECL_OK; ++r) rc = ecl_hip_mul_batch(d, ks, n, hit, 16, &cnt);
    const u64 t2 = us_now();
    if (rc != ECL_OK) { fprintf(stderr, "[!] bench-gtable w=%u: %s (%s)\n", w, ecl_hip_strerror(rc), d ? ecl_hip_last_error(d) : ""); return 1; }
    const double mult = (double)(t2 - t1) / 1e6, one = mult / reps, gent = (double)(t1 - t0) / 1e6 - one;
    const u32 nwin = (256 + w - 1) / w;
    const double slots = (double)(nwin - 1) * (double)(1u << (w - 1)) + (double)(1u << (256 - w * (nwin - 1)));  // signed digits: 2^(w-1) per row
    printf("w=%02u: %.1fK it/s | gen: %5.2fs | mul: %5.2fs | mem: %8.1fMB\n", w, (double)n * reps / mult / 1000, gent > 0 ? gent : 0, mult,
           slots * 64 / 1024 / 1024);
    fflush(stdout);
    ecl_hip_close(d);
  }
  ecl_hip_free_host(ks);
  return 0;
}
/* `mult-verify` (lib/bench.c:143-166: ec_gtable_mul against ec_jacobi_mulrdc for the scalars 2 .. 16001, silent when they
   agree): both window-table paths of the device - ecl_hip_verify (the 14-bit table of the walk) and ecl_hip_mul_batch (its
   own table; every hash160 comes back through an all-ones filter) - against the double-and-add kernel. */
static int run_mult_verify(void) {
  if (ecl_hip_device_count() <= 0) { fprintf(stderr, "no MI355X GPU visible (the search path has no CPU fallback)\n"); return 1; }
  enum { N = 16000 };
  static u64 ks[N][4], px[N][4], py[N][4];
  static u32 want33[N][5], want65[N][5], got33[N][5], got65[N][5];
  static u8 ok[N], okv[N];
  static ecl_found hit[2 * N];
  for (int i = 0; i < N; ++i) ks[i][0] = (u64)i + 2, ks[i][1] = ks[i][2] = ks[i][3] = 0;
  u64 ones[64];
  memset(ones, 0xff, sizeof ones);
  ecl_hip *d = NULL;
  u32 cnt = 0;
  int rc = ecl_hip_open(&d, 0, ECL_ADDR33 | ECL_ADDR65, 0);
  if (rc == ECL_OK) rc = ecl_hip_set_bloom(d, ones, 64);
  if (rc == ECL_OK) rc = ecl_hip_diag_mulg(d, ks, px, py, ok, N);
  if (rc == ECL_OK) rc = ecl_hip_diag_hash160(d, px, py, want33, want65, N);
  if (rc == ECL_OK) rc = ecl_hip_verify(d, ks, N, got33, got65, okv);
  if (rc == ECL_OK) rc = ecl_hip_mul_batch(d, ks, N, hit, 2 * N, &cnt);
  if (rc != ECL_OK) { fprintf(stderr, "[!] mult-verify: %s (%s)\n", ecl_hip_strerror(rc), d ? ecl_hip_last_error(d) : ""); return 1; }
  int bad = -1;
  for (int i = 0; i < N && bad < 0; ++i)
    if (!ok[i] || !okv[i] || memcmp(got33[i], want33[i], 20) || memcmp(got65[i], want65[i], 20)) bad = i;
  if (bad < 0 && cnt != 2 * N) bad = 0;
  for (u32 i = 0; i < cnt && bad < 0; ++i) {
    const u64 k = hit[i].key_offset;
    if (k >= N || memcmp(hit[i].h160, hit[i].compressed ? want33[k] : want65[k], 20)) bad = (int)k;
  }
  ecl_hip_close(d);
  if (bad >= 0) {
    printf("invalid on %d\n", bad);
    return 1;
  }
  return 0;
}
