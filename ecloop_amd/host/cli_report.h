/* cli_report.h - report_t (found sink, status line, pause state) and run_t (one run of a search command).
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
/* ------------------------------------------------------------------------------------------- found sink + status line */
/* One object for everything the program reports while it runs: found keys (stdout unless -q, the -o file), the two
   counters behind the status line, the clock with the paused time taken out.  Formats are the reference's
   (ctx_write_found main.c:182-203, ctx_print_status main.c:134-144, ctx_update main.c:158-172), byte for byte; the device
   threads and the key listener share it through its mutex. */
typedef struct {
  pthread_mutex_t mu;
  FILE *file;      /* -o (appended to), or NULL */
  bool quiet;      /* -q: nothing on stdout */
  u64 found, checked;
  u64 t_start, t_progress, t_shown; /* ms: clock start, last progress report, last status print */
  u64 paused_ms, paused_since;
  volatile bool paused; /* read by the device threads without the mutex, like the reference's flag (main.c:153) */
  bool closed;
} report_t;

static void hex_of_words(char *dst, const u32 *w, int n) { /* 8 digits per word, most significant word first as given */
  for (int i = 0; i < n; ++i) sprintf(dst + 8 * i, "%08x", w[i]);
}
static void hex_of_scalar(char dst[65], const sc *k) {
  for (int i = 0; i < 4; ++i) sprintf(dst + 16 * i, "%016llx", (unsigned long long)k->w[3 - i]);
}
static void report_init(report_t *r, const char *outfile, bool quiet) {
  memset(r, 0, sizeof *r);
  pthread_mutex_init(&r->mu, NULL);
  r->quiet = quiet;
  if (outfile) r->file = fopen(outfile, "a");
  r->t_start = r->t_progress = ms_now();
  r->t_shown = r->t_start - 5000;
}
static void report_restart_clock(report_t *r) { r->t_start = ms_now(); } /* the commands start their clock after bring-up */
/* "<secs>s ~ <rate> Mkeys/s ~ <found> / <checked>" + the key hint; '\r' while running, '\n' once closed */
static void status_show_locked(report_t *r) {
  int64_t run_ms = (int64_t)(r->t_progress - r->t_start) - (int64_t)r->paused_ms;
  double secs = (run_ms < 1 ? 1 : run_ms) / 1000.0;
  const char *hint = r->closed ? "" : r->paused ? " ('r' \xe2\x80\x93 resume)" : " ('p' \xe2\x80\x93 pause)";
  erase_status_line();
  fprintf(stderr, "%.2fs ~ %.2f Mkeys/s ~ %'llu / %'llu%s%c", secs, r->checked / secs / 1000000, (unsigned long long)r->found,
          (unsigned long long)r->checked, hint, r->closed ? '\n' : '\r');
  fflush(stderr);
}
/* one found key: "addr33: <hash160> <- <key>" on stdout, "addr33\t<hash160>\t<key>" in the file; counts it */
static void report_hit(report_t *r, bool compressed, const u32 h160[5], const sc *key) {
  char hh[41], kk[65];
  hex_of_words(hh, h160, 5);
  hex_of_scalar(kk, key);
  const char *label = compressed ? "addr33" : "addr65";
  const struct { FILE *to; const char *fmt; } dest[2] = {{r->quiet ? NULL : stdout, "%s: %s <- %s\n"}, {r->file, "%s\t%s\t%s\n"}};
  pthread_mutex_lock(&r->mu);
  for (int d = 0; d < 2; ++d) {
    if (!dest[d].to) continue;
    if (dest[d].to == stdout) erase_status_line();
    fprintf(dest[d].to, dest[d].fmt, label, hh, kk);
    fflush(dest[d].to);
  }
  r->found++;
  status_show_locked(r);
  pthread_mutex_unlock(&r->mu);
}
/* `units` more keys checked (status units: the reference counts job_size per job, x6 with -endo, main.c:431); the line
   is redrawn at most every 100 ms; a paused run parks the caller here, between two device calls */
static void report_progress(report_t *r, u64 units) {
  u64 now = ms_now();
  pthread_mutex_lock(&r->mu);
  r->checked += units, r->t_progress = now;
  if (now - r->t_shown >= 100) r->t_shown = now, status_show_locked(r);
  pthread_mutex_unlock(&r->mu);
  while (r->paused) usleep(100000);
}
static void report_pause(report_t *r, bool on) { /* 'p' / 'r' (main.c:874-888): paused time does not count */
  pthread_mutex_lock(&r->mu);
  if (on != r->paused) {
    u64 now = ms_now();
    if (on) r->paused_since = now;
    else r->paused_ms += now - r->paused_since;
    r->paused = on;
    status_show_locked(r);
  }
  pthread_mutex_unlock(&r->mu);
}
static void report_close(report_t *r) { /* ctx_finish, main.c:174-180 */
  pthread_mutex_lock(&r->mu);
  r->closed = true, r->t_progress = ms_now();
  status_show_locked(r);
  if (r->file) fclose(r->file), r->file = NULL;
  pthread_mutex_unlock(&r->mu);
}

/* ------------------------------------------------------------------------------------------- one run of a search command */
enum { CMD_NIL, CMD_ADD, CMD_MUL, CMD_RND };
typedef struct run_t {
  int cmd;
  opts_t opt;
  filter_t flt;
  report_t rep;
  int ngpus; /* device contexts (threads); `mul` opens two per GPU */
  ecl_hip *dev[MAX_GPUS];
  bool a33, a65, endo, colour, bin, parse_only, seeded;
  sc range_s, range_e, stride_k;
  u32 ord_offs, ord_size;
} run_t;

static void die_ecl(run_t *run, int g, int rc, const char *what) {
  fprintf(stderr, "\n[!] %s: %s (%s)\n", what, ecl_hip_strerror(rc), run->dev[g] ? ecl_hip_last_error(run->dev[g]) : "");
  exit(1);
}
/* pk_verify_hash (main.c:248-263) for all hits of one device call at once: both hash160 values of every reported key are
   derived again on the device by the window-table sum (ecl_hip_verify: not the walk kernel; own inversion per key) and
   compared with what the walk reported; a mismatch is fatal, with the reference's diagnostics */
static void verify_hits(run_t *run, int g, const sc *keys, const ecl_found *hits, u32 n) {
  if (!n) return;
  u32 (*h33)[5] = malloc((size_t)n * 20), (*h65)[5] = malloc((size_t)n * 20);
  u8 *finite = malloc(n);
  int rc = ecl_hip_verify(run->dev[g], (const uint64_t(*)[4])keys, n, h33, h65, finite);
  if (rc != ECL_OK) die_ecl(run, g, rc, "verify");
  for (u32 i = 0; i < n; ++i) {
    const u32 *want = hits[i].compressed ? h33[i] : h65[i];
    if (finite[i] && !memcmp(want, hits[i].h160, 20)) continue;
    char kk[65], lh[41], rh[41];
    hex_of_scalar(kk, &keys[i]), hex_of_words(lh, hits[i].h160, 5), hex_of_words(rh, want, 5);
    fprintf(stderr, "[!] error: hash mismatch (compressed: %d endo: %d)\npk: %s\nlh: %s\nrh: %s\n", hits[i].compressed, hits[i].endo, kk, lh, rh);
    exit(1);
  }
  free(h33), free(h65), free(finite);
}
