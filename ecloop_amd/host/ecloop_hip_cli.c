/*
 * ecloop-hip — host program with ecloop's command-line surface (add / mul / rnd, blf-gen / blf-check,
 * -f -o -t -a -r -d -q -endo -seed -raw), driving MI355X GPUs through the C ABI of include/ecloop_hip.h.
 *
 * Plain C, links only libecloop_hip.so.  Everything here is what the reference keeps on the host side of the
 * boundary (citations into /root/reference): filter loading (main.c:71-131), range / offset
 * parsing (main.c:666-746), the job arithmetic of cmd_add (main.c:405-454), calc_priv (main.c:267-276),
 * the pk_verify_hash self-check (main.c:248-263, done by re-deriving the hit on the device with the independent
 * double-and-add kernel), the found sink and status line formats (main.c:134-203), cmd_mul's line reader
 * (main.c:542-576), cmd_rnd's window generator (main.c:580-662), blf-gen / blf-check (utils.c:400-529).
 * `-t N` selects the number of GPUs (one host thread per device; default: all): the scan is range-partitioned,
 * no collective.  All curve and hash work for the search itself happens on the device.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <locale.h>
#include <math.h>
#include <pthread.h>
#include <signal.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/select.h>
#include <sys/stat.h>
#include <sys/time.h>
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

/* ------------------------------------------------------------------------------------------- scalars mod n */
static const sc SC_N = {{0xbfd25e8cd0364141ULL, 0xbaaedce6af48a03bULL, 0xfffffffffffffffeULL, ~0ULL}};
static const sc SC_P = {{0xfffffffefffffc2fULL, ~0ULL, ~0ULL, ~0ULL}};
static const sc SC_LAMBDA = {{0xdf02967c1b23bd72ULL, 0x122e22ea20816678ULL, 0xa5261c028812645aULL, 0x5363ad4cc05c30e0ULL}};

static int sc_cmp(const sc *a, const sc *b) {
  for (int i = 3; i >= 0; --i)
    if (a->w[i] != b->w[i]) return a->w[i] > b->w[i] ? 1 : -1;
  return 0;
}
static u64 sc_addraw(sc *r, const sc *a, const sc *b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) c += (u128)a->w[i] + b->w[i], r->w[i] = (u64)c, c >>= 64;
  return (u64)c;
}
/* 256-bit logical shift right by one */
static sc sc_shr1(sc a) {
  for (int i = 0; i < 4; ++i) a.w[i] = (a.w[i] >> 1) | (i < 3 ? a.w[i + 1] << 63 : 0);
  return a;
}
static u64 sc_subraw(sc *r, const sc *a, const sc *b) {
  u64 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a->w[i] - b->w[i] - br;
    r->w[i] = (u64)d, br = (u64)(d >> 64) & 1;
  }
  return br;
}
static sc sc_u64(u64 v) { sc r = {{v, 0, 0, 0}}; return r; }
static bool sc_is_zero(const sc *a) { return !(a->w[0] | a->w[1] | a->w[2] | a->w[3]); }
static sc sc_add(sc a, sc b) { /* canonical inputs -> canonical sum */
  sc r;
  u64 c = sc_addraw(&r, &a, &b);
  if (c || sc_cmp(&r, &SC_N) >= 0) sc_subraw(&r, &r, &SC_N);
  return r;
}
static sc sc_neg(sc a) {
  sc r = {{0, 0, 0, 0}};
  if (!sc_is_zero(&a)) sc_subraw(&r, &SC_N, &a);
  return r;
}
static sc sc_mul(sc a, sc b) { /* double-and-add; per hit / per job only */
  sc r = {{0, 0, 0, 0}};
  for (int bit = 255; bit >= 0; --bit) {
    r = sc_add(r, r);
    if ((b.w[bit >> 6] >> (bit & 63)) & 1) r = sc_add(r, a);
  }
  return r;
}
static sc sc_reduce(sc a) {
  if (sc_cmp(&a, &SC_N) >= 0) sc_subraw(&a, &a, &SC_N);
  return a;
}
static sc sc_pow2(unsigned e) {
  sc r = sc_u64(1);
  for (unsigned i = 0; i < e; ++i) r = sc_add(r, r);
  return r;
}
static unsigned sc_bitlen(const sc *a) {
  for (int i = 3; i >= 0; --i)
    if (a->w[i]) return 64 * i + (64 - __builtin_clzll(a->w[i]));
  return 0;
}
/* fe_modn_from_hex (ecc.c:81-95,262-265): right to left, non-hex characters skipped, 64 digits at most */
static sc sc_from_hex(const char *hex) {
  sc r = {{0, 0, 0, 0}};
  int cnt = 0;
  for (long i = (long)strlen(hex) - 1; i >= 0 && cnt < 64; --i) {
    int c = tolower((unsigned char)hex[i]);
    u64 v;
    if (c >= '0' && c <= '9') v = c - '0';
    else if (c >= 'a' && c <= 'f') v = c - 'a' + 10;
    else continue;
    r.w[cnt / 16] |= v << (cnt * 4 % 64);
    cnt++;
  }
  return sc_reduce(r);
}
/* calc_priv (main.c:267-276) */
static sc calc_priv(sc start, sc stride, u64 off, int endo) {
  sc k = sc_add(sc_reduce(start), sc_mul(stride, sc_u64(off)));
  if (endo == 2 || endo == 3) k = sc_mul(k, SC_LAMBDA);
  if (endo == 4 || endo == 5) k = sc_mul(sc_mul(k, SC_LAMBDA), SC_LAMBDA);
  if (endo == 1 || endo == 3 || endo == 5) k = sc_neg(k);
  return k;
}

/* ------------------------------------------------------------------------------------------- small utilities */
static u64 tsnow(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return (u64)tv.tv_sec * 1000 + tv.tv_usec / 1000;
}
typedef struct { int argc; const char **argv; } args_t;
static bool args_bool(args_t *a, const char *name) {
  for (int i = 1; i < a->argc; ++i)
    if (strcmp(a->argv[i], name) == 0) return true;
  return false;
}
static const char *arg_str(args_t *a, const char *name) {
  for (int i = 1; i < a->argc - 1; ++i)
    if (strcmp(a->argv[i], name) == 0) return a->argv[i + 1];
  return NULL;
}
static u64 args_uint(args_t *a, const char *name, u64 def) {
  const char *s = arg_str(a, name);
  return s ? strtoull(s, NULL, 10) : def;
}
static void term_clear_line(void) { fputs("\033[2K\r", stderr); }

/* ------------------------------------------------------------------------------------------- bloom filter (host) */
#define BLF_MAGIC 0x45434246u
#define BLF_VERSION 1u
typedef struct { u64 size; u64 *bits; } blf_t;

static void blf_indices(u64 idx[20], const u32 h[5]) { /* utils.c:290-306 */
  u64 a[6] = {(u64)h[0] << 32 | h[1], (u64)h[2] << 32 | h[3], (u64)h[4] << 32 | h[0], (u64)h[1] << 32 | h[2],
              (u64)h[3] << 32 | h[4], 0};
  a[5] = a[0];
  static const int S[4] = {24, 28, 36, 40};
  for (int s = 0; s < 4; ++s)
    for (int j = 0; j < 5; ++j) idx[s * 5 + j] = a[j] << S[s] | a[j + 1] >> S[s];
}
static void blf_add(blf_t *b, const u32 h[5]) {
  u64 idx[20];
  blf_indices(idx, h);
  for (int i = 0; i < 20; ++i) b->bits[(idx[i] >> 6) % b->size] |= 1ULL << (idx[i] & 63);
}
static bool blf_has(const blf_t *b, const u32 h[5]) {
  u64 idx[20];
  blf_indices(idx, h);
  for (int i = 0; i < 20; ++i)
    if (!((b->bits[(idx[i] >> 6) % b->size] >> (idx[i] & 63)) & 1)) return false;
  return true;
}
static bool blf_save(const char *path, const blf_t *b) { /* utils.c:328-360 */
  FILE *f = fopen(path, "wb");
  if (!f) return false;
  u32 head[2] = {BLF_MAGIC, BLF_VERSION};
  bool ok = fwrite(head, 4, 2, f) == 2 && fwrite(&b->size, 8, 1, f) == 1 && fwrite(b->bits, 8, b->size, f) == b->size;
  fclose(f);
  return ok;
}
static bool blf_load(const char *path, blf_t *b) { /* utils.c:362-396 */
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "failed to open input file\n"); return false; }
  u32 head[2];
  u64 size;
  if (fread(head, 4, 2, f) != 2 || fread(&size, 8, 1, f) != 1) {
    fprintf(stderr, "failed to read bloom filter header\n");
    fclose(f);
    return false;
  }
  if (head[0] != BLF_MAGIC || head[1] != BLF_VERSION) {
    fprintf(stderr, "invalid bloom filter version; create a new filter with blf-gen command\n");
    fclose(f);
    return false;
  }
  u64 *bits = calloc(size ? size : 1, 8);
  if (fread(bits, 8, size, f) != size) {
    fprintf(stderr, "failed to read bloom filter bits\n");
    fclose(f);
    free(bits);
    return false;
  }
  fclose(f);
  b->size = size, b->bits = bits;
  return true;
}
static int cmp160(const void *a, const void *b) { /* addr.c:18-26 */
  const u32 *x = a, *y = b;
  for (int i = 0; i < 5; ++i)
    if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}
static bool parse_hash40(const char *s, u32 h[5]) {
  for (int i = 0; i < 40; ++i)
    if (!isxdigit((unsigned char)s[i])) return false;
  for (int j = 0; j < 5; ++j) {
    char t[9];
    memcpy(t, s + j * 8, 8), t[8] = 0;
    h[j] = (u32)strtoul(t, NULL, 16);
  }
  return true;
}

/* ------------------------------------------------------------------------------------------- context */
enum { CMD_NIL, CMD_ADD, CMD_MUL, CMD_RND };
typedef struct ctx_t {
  int cmd;
  pthread_mutex_t lock;
  int ngpus;
  ecl_hip *dev[MAX_GPUS];
  u64 k_checked, k_found;
  bool a33, a65, endo, quiet, use_color, raw_text, bin_input, parse_only, plan_only, has_seed, finished;
  FILE *outfile;
  u64 ts_started, ts_updated, ts_printed;
  volatile bool paused; /* 'p' / 'r' on the terminal (main.c:41-46,874-888) */
  u64 ts_paused_at, paused_time;
  u32 *list; /* sorted unique hashes (5 words each) or NULL in bloom-only mode (main.c:49-51) */
  u64 list_count;
  blf_t blf;
  sc range_s, range_e, stride_k;
  u32 ord_offs, ord_size;
} ctx_t;

static void die_ecl(ctx_t *ctx, int g, int rc, const char *what) {
  fprintf(stderr, "\n[!] %s: %s (%s)\n", what, ecl_hip_strerror(rc), ctx->dev[g] ? ecl_hip_last_error(ctx->dev[g]) : "");
  exit(1);
}

/* load_filter (main.c:71-131).  fgets into a 41-byte buffer reads 40-character chunks; only full chunks count.
   Chunks that are not clean hex are dropped (the reference parses garbage out of them). */
static void load_filter(ctx_t *ctx, const char *path) {
  if (!path) { fprintf(stderr, "missing filter file\n"); exit(1); }
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "failed to open filter file: %s\n", path); exit(1); }
  const char *ext = strrchr(path, '.');
  if (ext && strcmp(ext, ".blf") == 0) {
    fclose(f);
    if (!blf_load(path, &ctx->blf)) exit(1);
    return;
  }
  size_t cap = 32, n = 0;
  u32 *hs = malloc(cap * 20);
  char line[41];
  while (fgets(line, sizeof line, f)) {
    if (strlen(line) != 40) continue;
    if (n >= cap) cap *= 2, hs = realloc(hs, cap * 20);
    if (parse_hash40(line, hs + n * 5)) n++;
  }
  fclose(f);
  if (n == 0) { fprintf(stderr, "no hashes in filter file\n"); exit(1); }
  qsort(hs, n, 20, cmp160);
  size_t u = 0;
  for (size_t i = 1; i < n; ++i)
    if (memcmp(hs + u * 5, hs + i * 5, 20) != 0) memcpy(hs + (++u) * 5, hs + i * 5, 20);
  ctx->list = hs, ctx->list_count = u + 1;
  ctx->blf.size = ctx->list_count * 2;
  ctx->blf.bits = calloc(ctx->blf.size, 8);
  for (size_t i = 0; i < ctx->list_count; ++i) blf_add(&ctx->blf, hs + i * 5);
}

/* status line, main.c:134-144 */
static void ctx_print_unlocked(ctx_t *ctx) {
  const char *msg = ctx->finished ? "" : (ctx->paused ? " ('r' \xe2\x80\x93 resume)" : " ('p' \xe2\x80\x93 pause)");
  int64_t eff = (int64_t)(ctx->ts_updated - ctx->ts_started) - (int64_t)ctx->paused_time;
  double dt = (eff < 1 ? 1 : eff) / 1000.0;
  double it = ctx->k_checked / dt / 1000000;
  term_clear_line();
  fprintf(stderr, "%.2fs ~ %.2f Mkeys/s ~ %'llu / %'llu%s%c", dt, it, (unsigned long long)ctx->k_found,
          (unsigned long long)ctx->k_checked, msg, ctx->finished ? '\n' : '\r');
  fflush(stderr);
}
static void ctx_update(ctx_t *ctx, u64 k) { /* main.c:158-172 */
  u64 ts = tsnow();
  pthread_mutex_lock(&ctx->lock);
  ctx->k_checked += k, ctx->ts_updated = ts;
  if (ts - ctx->ts_printed >= 100) ctx->ts_printed = ts, ctx_print_unlocked(ctx);
  pthread_mutex_unlock(&ctx->lock);
  while (ctx->paused) usleep(100000); /* ctx_check_paused, main.c:152-156: the caller holds no device work here */
}
static void ctx_finish(ctx_t *ctx) { /* main.c:174-180 */
  pthread_mutex_lock(&ctx->lock);
  ctx->finished = true, ctx->ts_updated = tsnow();
  ctx_print_unlocked(ctx);
  if (ctx->outfile) fclose(ctx->outfile);
  pthread_mutex_unlock(&ctx->lock);
}
/* ctx_write_found, main.c:182-203 */
static void ctx_write_found(ctx_t *ctx, const char *label, const u32 h[5], sc pk) {
  pthread_mutex_lock(&ctx->lock);
  if (!ctx->quiet) {
    term_clear_line();
    printf("%s: %08x%08x%08x%08x%08x <- %016llx%016llx%016llx%016llx\n", label, h[0], h[1], h[2], h[3], h[4],
           (unsigned long long)pk.w[3], (unsigned long long)pk.w[2], (unsigned long long)pk.w[1], (unsigned long long)pk.w[0]);
    fflush(stdout);
  }
  if (ctx->outfile) {
    fprintf(ctx->outfile, "%s\t%08x%08x%08x%08x%08x\t%016llx%016llx%016llx%016llx\n", label, h[0], h[1], h[2], h[3], h[4],
            (unsigned long long)pk.w[3], (unsigned long long)pk.w[2], (unsigned long long)pk.w[1], (unsigned long long)pk.w[0]);
    fflush(ctx->outfile);
  }
  ctx->k_found += 1;
  ctx_print_unlocked(ctx);
  pthread_mutex_unlock(&ctx->lock);
}
/* second stage of ctx_check_hash (main.c:212-216) */
static bool list_confirm(const ctx_t *ctx, const u32 h[5]) {
  return !ctx->list || bsearch(h, ctx->list, ctx->list_count, 20, cmp160) != NULL;
}
/* pk_verify_hash (main.c:248-263) for all hits of one device call: re-derive them from their scalars on the device in
   one batch, on a path that shares no kernel with the walk (ecl_hip_verify: window-table sum, own inversion per key) */
static void pk_verify_hashes(ctx_t *ctx, int g, const sc *pks, const ecl_found *hits, u32 n) {
  if (!n) return;
  u64 (*k)[4] = malloc((size_t)n * 32);
  u32 (*h33)[5] = malloc((size_t)n * 20), (*h65)[5] = malloc((size_t)n * 20);
  u8 *ok = malloc(n);
  for (u32 i = 0; i < n; ++i) memcpy(k[i], pks[i].w, 32);
  int rc = ecl_hip_verify(ctx->dev[g], k, n, h33, h65, ok);
  if (rc != ECL_OK) die_ecl(ctx, g, rc, "verify");
  for (u32 i = 0; i < n; ++i) {
    const u32 *r = hits[i].compressed ? h33[i] : h65[i], *h = hits[i].h160;
    if (ok[i] && memcmp(r, h, 20) == 0) continue;
    fprintf(stderr, "[!] error: hash mismatch (compressed: %d endo: %d)\n", hits[i].compressed, hits[i].endo);
    fprintf(stderr, "pk: %016llx%016llx%016llx%016llx\n", (unsigned long long)pks[i].w[3], (unsigned long long)pks[i].w[2],
            (unsigned long long)pks[i].w[1], (unsigned long long)pks[i].w[0]);
    fprintf(stderr, "lh: %08x%08x%08x%08x%08x\n", h[0], h[1], h[2], h[3], h[4]);
    fprintf(stderr, "rh: %08x%08x%08x%08x%08x\n", r[0], r[1], r[2], r[3], r[4]);
    exit(1);
  }
  free(k), free(h33), free(h65), free(ok);
}

/* ------------------------------------------------------------------------------------------- add */
/* One scan = the contiguous run of keys  rs + i*stride, i < hashed  (what cmd_add's jobs hash, main.c:405-454).  The
   device threads pull chunks of it from a shared counter, like the reference's workers pull 2^21-key jobs
   (main.c:418-431): a GPU that sustains a few percent more clock simply takes more chunks, and a scan of any length
   (the default range 0x800:p included) streams through without its key count having to fit 64 bits. */
typedef struct {
  ctx_t *ctx;
  sc rs;             /* first scalar */
  sc hashed;         /* keys to hash (256-bit: `add` without -r walks ~2^256 / stride keys) */
  sc next;           /* keys handed out so far */
  u64 chunk;         /* keys per hand-out = per device call */
  u64 status_total;  /* what the status counter must have gained at the end (0: not representable, add as we go) */
  u64 status_given;
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
static sc scan_scalar(const ctx_t *ctx, const sc *rs, const sc *off) {
  sc o = sc_reduce(*off); /* off < 2^256 < 2n */
  return sc_add(sc_reduce(*rs), sc_mul(ctx->stride_k, o));
}

static void *scan_worker(void *arg) {
  scan_worker_t *w = arg;
  scan_t *sn = w->scan;
  ctx_t *ctx = sn->ctx;
  u32 cap = 4096;
  ecl_found *buf = malloc(sizeof(ecl_found) * cap);
  for (;;) {
    pthread_mutex_lock(&sn->mu);
    sc lo = sn->next, left;
    if (sc_cmp(&lo, &sn->hashed) >= 0) { pthread_mutex_unlock(&sn->mu); break; }
    sc_subraw(&left, &sn->hashed, &lo);
    u64 n = (left.w[1] | left.w[2] | left.w[3]) || left.w[0] > sn->chunk ? sn->chunk : left.w[0];
    sn->next = sc_add_u64_raw(lo, n);
    bool last = sc_cmp(&sn->next, &sn->hashed) >= 0;
    /* status counter: the reference adds job_size (x6 with endo) per job (main.c:431); spread over the chunks */
    u64 st;
    if (!sn->status_total) st = n * sn->mult;
    else if (last) st = sn->status_total - sn->status_given;
    else {
      u128 done = (u128)sn->next.w[0]; /* status_total != 0 implies hashed < 2^63 */
      u64 upto = (u64)((u128)sn->status_total * done / sn->hashed.w[0]);
      st = upto - sn->status_given;
    }
    sn->status_given += st;
    pthread_mutex_unlock(&sn->mu);

    sc s = scan_scalar(ctx, &sn->rs, &lo);
    u32 cnt = 0;
    int rc;
    for (;;) {
      rc = ecl_hip_add_range(ctx->dev[w->g], s.w, n, buf, cap, &cnt);
      if (rc != ECL_E_OVERFLOW) break;
      cap = cnt, buf = realloc(buf, sizeof(ecl_found) * cap); /* dense filter: rerun with a buffer that fits */
    }
    if (rc != ECL_OK) die_ecl(ctx, w->g, rc, "add_range");
    u32 kept = 0;
    sc *pks = cnt ? malloc(sizeof(sc) * cnt) : NULL;
    for (u32 i = 0; i < cnt; ++i) {
      if (!list_confirm(ctx, buf[i].h160)) continue;
      pks[kept] = calc_priv(s, ctx->stride_k, buf[i].key_offset, buf[i].endo);
      buf[kept++] = buf[i];
    }
    pk_verify_hashes(ctx, w->g, pks, buf, kept);
    for (u32 i = 0; i < kept; ++i) ctx_write_found(ctx, buf[i].compressed ? "addr33" : "addr65", buf[i].h160, pks[i]);
    free(pks);
    ctx_update(ctx, st);
  }
  free(buf);
  return NULL;
}

/* keys per hand-out.  One GPU: whole sweeps of the walk (2^32 keys at the default geometry), which continue on the
   device without re-initialisation.  Several GPUs: at least two chunks per GPU so that uneven clocks even out, at
   least 2^27 keys (10 ms of kernel against ~0.4 ms of per-call set-up), at most 2^30. */
static u64 scan_chunk(const ctx_t *ctx, const sc *hashed) {
  if (ctx->ngpus <= 1) return LAUNCH_KEYS;
  if (hashed->w[1] | hashed->w[2] | hashed->w[3]) return 1ull << 30;
  u64 c = (hashed->w[0] + 2 * (u64)ctx->ngpus - 1) / (2 * (u64)ctx->ngpus);
  c = (c + GROUP_INV_SIZE - 1) / GROUP_INV_SIZE * GROUP_INV_SIZE;
  if (c < (1ull << 27)) c = 1ull << 27;
  if (c > (1ull << 30)) c = 1ull << 30;
  return c;
}

/* The plan of one scan: cmd_add (main.c:437-454) over [range_s, range_e) hashes the contiguous run of `hashed` keys from
   range_s and adds `status_total` to the status counter (0: too long to count, added chunk by chunk). */
static void scan_plan(ctx_t *ctx, sc rs, sc re, bool full_jobs, scan_t *sn) {
  sc span;
  sc_subraw(&span, &re, &rs);
  /* cmd_rnd always uses MAX_JOB_SIZE jobs, even for a narrower window (main.c:624) */
  bool small = !full_jobs && !(span.w[1] | span.w[2] | span.w[3]) && span.w[0] < MAX_JOB_SIZE;
  u64 job = small ? span.w[0] : MAX_JOB_SIZE; /* main.c:442 */
  /* njobs = ceil(span / (job * stride)) (main.c:420-427): the counter steps by job*stride until it reaches range_e */
  sc njobs = {{0, 0, 0, 0}};
  if (small && ctx->ord_offs == 0) njobs = sc_u64(1);
  else if (!small) {
    unsigned sh = 21 + ctx->ord_offs; /* job * stride = 2^sh */
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
    sc inc = sc_mul(ctx->stride_k, sc_u64(job)), cur = rs;
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
  sn->ctx = ctx, sn->rs = rs, sn->mult = ctx->endo ? 6 : 1;
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
  sn->chunk = scan_chunk(ctx, &sn->hashed);
}

/* one scan, spread over the GPUs */
static void scan_range(ctx_t *ctx, sc rs, sc re, bool full_jobs) {
  scan_t sn;
  scan_plan(ctx, rs, re, full_jobs, &sn);
  pthread_mutex_init(&sn.mu, NULL);
  pthread_t th[MAX_GPUS];
  scan_worker_t ws[MAX_GPUS];
  for (int g = 0; g < ctx->ngpus; ++g) {
    ws[g] = (scan_worker_t){&sn, g};
    pthread_create(&th[g], NULL, scan_worker, &ws[g]);
  }
  for (int g = 0; g < ctx->ngpus; ++g) pthread_join(th[g], NULL);
  pthread_mutex_destroy(&sn.mu);
}

static void cmd_add(ctx_t *ctx) {
  ctx->ts_started = tsnow();
  scan_range(ctx, ctx->range_s, ctx->range_e, false);
  ctx_finish(ctx);
}

/* ------------------------------------------------------------------------------------------- mul */
/* host SHA-256 of a passphrase for `-raw` (main.c:505-527): input preparation, not the search path.  Block by block,
   nothing allocated per line. */
static void sha256_block(u32 st[8], const u8 *blk) {
  static const u32 K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
  u32 w[64], v[8];
  for (int i = 0; i < 16; ++i) w[i] = (u32)blk[4 * i] << 24 | (u32)blk[4 * i + 1] << 16 | (u32)blk[4 * i + 2] << 8 | blk[4 * i + 3];
  for (int i = 16; i < 64; ++i)
    w[i] = w[i - 16] + (ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
           (ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10));
  memcpy(v, st, 32);
  for (int i = 0; i < 64; ++i) {
    u32 t1 = v[7] + (ROR(v[4], 6) ^ ROR(v[4], 11) ^ ROR(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K[i] + w[i];
    u32 t2 = (ROR(v[0], 2) ^ ROR(v[0], 13) ^ ROR(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
    memmove(v + 1, v, 28);
    v[4] += t1, v[0] = t1 + t2;
  }
  for (int i = 0; i < 8; ++i) st[i] += v[i];
#undef ROR
}
static void sha256_stream(u32 st[8], const u8 *msg, size_t len) {
  static const u32 IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(st, IV, 32);
  size_t off = 0;
  for (; off + 64 <= len; off += 64) sha256_block(st, msg + off);
  u8 tail[128] = {0};
  size_t rem = len - off, total = rem + 9 <= 64 ? 64 : 128;
  memcpy(tail, msg + off, rem);
  tail[rem] = 0x80;
  for (int j = 0; j < 8; ++j) tail[total - 1 - j] = (u8)(((u64)len * 8) >> (8 * j));
  sha256_block(st, tail);
  if (total == 128) sha256_block(st, tail + 64);
}

static void mul_flush(ctx_t *ctx, int g, u64 (*ks)[4], u32 n) {
  if (!n) return;
  u32 cap = n * 2 + 16, cnt = 0;
  ecl_found *buf = malloc(sizeof(ecl_found) * cap);
  int rc = ecl_hip_mul_batch(ctx->dev[g], ks, n, buf, cap, &cnt);
  if (rc != ECL_OK) die_ecl(ctx, g, rc, "mul_batch");
  for (u32 i = 0; i < cnt; ++i) {
    if (!list_confirm(ctx, buf[i].h160)) continue;
    sc pk;
    memcpy(pk.w, ks[buf[i].key_offset], 32);
    ctx_write_found(ctx, buf[i].compressed ? "addr33" : "addr65", buf[i].h160, pk); /* no verify: main.c:469,474 */
  }
  free(buf);
  ctx_update(ctx, n);
}
/* cmd_mul (main.c:542-576): stdin lines -> scalars (hex, or SHA-256 of the text with -raw) -> device batches.
   The reference parses in its worker threads (main.c:503-527) and is bound by that; here the curve work is on the
   GPUs, so the text side is a three-stage pipeline that keeps every stage busy:
     reader thread   stdin -> 64 MB text chunks cut at a line end (ring of 3 buffers)
     parse pool      a chunk is cut into slices at line ends; every slice is parsed by one thread into its own scratch
                     (ONE pass; 64-digit lines - the normal input - decode 16 characters at a time with SSSE3), then
                     the slices are packed into one scalar array, order preserved
     device threads  one per GPU, each takes the next parsed array (`-t N` GPUs; the reference's worker queue,
                     main.c:556-571)
   `-bin` (not in the reference): stdin carries the scalars themselves, 32 bytes each (4 little-endian u64 = `fe`), for
   feeders that can produce more than text parsing can take.
   Difference kept small on purpose: a line longer than 1024 characters is one line here (the reference's fgets
   splits it, main.c:552). */
static signed char HEXVAL[256];
static void hexval_init(void) {
  memset(HEXVAL, -1, sizeof HEXVAL);
  for (int c = '0'; c <= '9'; ++c) HEXVAL[c] = (signed char)(c - '0');
  for (int c = 'a'; c <= 'f'; ++c) HEXVAL[c] = (signed char)(c - 'a' + 10), HEXVAL[c - 32] = (signed char)(c - 'a' + 10);
}
#if defined(__x86_64__)
#include <immintrin.h>
/* 16 hex characters (most significant first) -> one little-endian u64; false if any character is not a hex digit */
__attribute__((target("ssse3"))) static bool hex16_ssse3(const char *p, u64 *out) {
  const __m128i c = _mm_loadu_si128((const __m128i *)p);
  const __m128i lower = _mm_or_si128(c, _mm_set1_epi8(0x20));
  const __m128i isdig = _mm_and_si128(_mm_cmpgt_epi8(c, _mm_set1_epi8('0' - 1)), _mm_cmpgt_epi8(_mm_set1_epi8('9' + 1), c));
  const __m128i isalp = _mm_and_si128(_mm_cmpgt_epi8(lower, _mm_set1_epi8('a' - 1)), _mm_cmpgt_epi8(_mm_set1_epi8('f' + 1), lower));
  if (_mm_movemask_epi8(_mm_or_si128(isdig, isalp)) != 0xFFFF) return false;
  const __m128i nib = _mm_add_epi8(_mm_and_si128(c, _mm_set1_epi8(0x0F)), _mm_and_si128(isalp, _mm_set1_epi8(9)));
  const __m128i pair = _mm_maddubs_epi16(nib, _mm_set1_epi16(0x0110)); /* first digit * 16 + second digit */
  const __m128i bytes = _mm_packus_epi16(pair, pair);                   /* 8 bytes, most significant first */
  const __m128i rev = _mm_shuffle_epi8(bytes, _mm_set_epi8(-1, -1, -1, -1, -1, -1, -1, -1, 0, 1, 2, 3, 4, 5, 6, 7));
  *out = (u64)_mm_cvtsi128_si64(rev);
  return true;
}
static bool have_ssse3;
#endif
static sc line_to_scalar(const ctx_t *ctx, const char *p, size_t len) {
  sc k = {{0, 0, 0, 0}};
  if (!ctx->raw_text) { /* fe_modn_from_hex: right to left, non-hex skipped, 64 digits at most */
#if defined(__x86_64__)
    if (len == 64 && have_ssse3 && hex16_ssse3(p, &k.w[3]) && hex16_ssse3(p + 16, &k.w[2]) && hex16_ssse3(p + 32, &k.w[1]) &&
        hex16_ssse3(p + 48, &k.w[0]))
      return sc_reduce(k);
    k = (sc){{0, 0, 0, 0}};
#endif
    int cnt = 0;
    for (size_t i = len; i-- > 0 && cnt < 64;) {
      int v = HEXVAL[(u8)p[i]];
      if (v < 0) continue;
      k.w[cnt >> 4] |= (u64)v << ((cnt & 15) * 4);
      cnt++;
    }
    return sc_reduce(k);
  }
  u32 st[8];
  sha256_stream(st, (const u8 *)p, len);
  k.w[0] = (u64)st[6] << 32 | st[7], k.w[1] = (u64)st[4] << 32 | st[5];
  k.w[2] = (u64)st[2] << 32 | st[3], k.w[3] = (u64)st[0] << 32 | st[1];
  return k;
}
typedef struct {
  const ctx_t *ctx;
  const char *buf;
  size_t beg, end;   /* slice [beg, end): starts at a line start, ends after a '\n' (or at the chunk end) */
  u64 (*tmp)[4];     /* this thread's scratch, grown on demand */
  size_t tmp_cap, count;
  u64 (*dst)[4];     /* second phase: where the slice's scalars go in the chunk's array */
} parse_slice;
static void *parse_worker(void *arg) {
  parse_slice *s = arg;
  size_t n = 0, at = s->beg;
  while (at < s->end) {
    const char *nl = memchr(s->buf + at, '\n', s->end - at);
    size_t stop = nl ? (size_t)(nl - s->buf) : s->end, len = stop - at;
    if (len && s->buf[at + len - 1] == '\r') len--;
    if (len) {
      if (n >= s->tmp_cap) s->tmp_cap = s->tmp_cap ? s->tmp_cap * 2 : 1 << 16, s->tmp = realloc(s->tmp, s->tmp_cap * 32);
      sc k = line_to_scalar(s->ctx, s->buf + at, len);
      memcpy(s->tmp[n++], k.w, 32);
    }
    at = stop + 1;
  }
  s->count = n;
  return NULL;
}
static void *pack_worker(void *arg) {
  parse_slice *s = arg;
  memcpy(s->dst, s->tmp, s->count * 32);
  return NULL;
}

/* a pool of parse threads that lives as long as the command: run() executes fn(arg[i]) for i < n on the workers and
   returns when all are done (64 MB chunks come every few milliseconds; creating 2 x 32 threads for each cost a quarter
   of the front end's time) */
typedef struct {
  pthread_t th[32];
  int nth;
  void *(*fn)(void *);
  char *args;
  size_t stride;
  int n, next, done;
  u64 gen;
  bool quit;
  pthread_mutex_t mu;
  pthread_cond_t cv_work, cv_done;
} pool_t;
static void *pool_main(void *arg) {
  pool_t *p = arg;
  u64 seen = 0;
  pthread_mutex_lock(&p->mu);
  for (;;) {
    while (!p->quit && (p->gen == seen || p->next >= p->n)) {
      if (p->gen != seen && p->next >= p->n) seen = p->gen;
      pthread_cond_wait(&p->cv_work, &p->mu);
    }
    if (p->quit) break;
    int i = p->next++;
    void *(*fn)(void *) = p->fn;
    void *a = p->args + (size_t)i * p->stride;
    pthread_mutex_unlock(&p->mu);
    fn(a);
    pthread_mutex_lock(&p->mu);
    if (++p->done == p->n) pthread_cond_signal(&p->cv_done);
  }
  pthread_mutex_unlock(&p->mu);
  return NULL;
}
static void pool_init(pool_t *p, int nth) {
  memset(p, 0, sizeof *p);
  pthread_mutex_init(&p->mu, NULL), pthread_cond_init(&p->cv_work, NULL), pthread_cond_init(&p->cv_done, NULL);
  p->nth = nth;
  for (int i = 0; i < nth; ++i) pthread_create(&p->th[i], NULL, pool_main, p);
}
static void pool_run(pool_t *p, void *(*fn)(void *), void *args, size_t stride, int n) {
  if (n <= 0) return;
  pthread_mutex_lock(&p->mu);
  p->fn = fn, p->args = args, p->stride = stride, p->n = n, p->next = 0, p->done = 0, p->gen++;
  pthread_cond_broadcast(&p->cv_work);
  while (p->done < n) pthread_cond_wait(&p->cv_done, &p->mu);
  pthread_mutex_unlock(&p->mu);
}
static void pool_stop(pool_t *p) {
  pthread_mutex_lock(&p->mu);
  p->quit = true;
  pthread_cond_broadcast(&p->cv_work);
  pthread_mutex_unlock(&p->mu);
  for (int i = 0; i < p->nth; ++i) pthread_join(p->th[i], NULL);
}
typedef struct { void *dst; const void *src; size_t n; } copy_task;
static void *copy_worker(void *arg) {
  copy_task *t = arg;
  memcpy(t->dst, t->src, t->n);
  return NULL;
}

/* text chunks: reader thread -> parser */
#define MUL_TEXT_CHUNK ((size_t)64 << 20)
#define MUL_TEXT_RING 3
typedef struct { char *buf, *own; size_t len; } text_chunk; /* buf = own (a ring buffer) or a slice of the mapped input */
typedef struct {
  text_chunk ring[MUL_TEXT_RING];
  int head, tail, count; /* filled chunks: [tail, head) */
  bool eof, bin;
  pthread_mutex_t mu;
  pthread_cond_t cv;
} text_queue;
static void *mul_reader(void *arg) {
  text_queue *q = arg;
  /* a regular file on stdin is mapped: the parse threads read (and page in) their slices in parallel, nothing is copied */
  struct stat stt;
  off_t pos = lseek(0, 0, SEEK_CUR);
  if (pos >= 0 && fstat(0, &stt) == 0 && S_ISREG(stt.st_mode) && stt.st_size > pos) {
    size_t size = (size_t)stt.st_size;
    char *map = mmap(NULL, size, PROT_READ, MAP_PRIVATE, 0, 0);
    if (map != MAP_FAILED) {
      madvise(map, size, MADV_SEQUENTIAL);
      for (size_t at = (size_t)pos; at < size;) {
        size_t end = at + MUL_TEXT_CHUNK < size ? at + MUL_TEXT_CHUNK : size;
        if (end < size) {
          if (q->bin) end = at + (end - at) / 32 * 32;
          else {
            size_t e = end;
            while (e > at && map[e - 1] != '\n') e--;
            if (e > at) end = e;
          }
        }
        pthread_mutex_lock(&q->mu);
        while (q->count == MUL_TEXT_RING) pthread_cond_wait(&q->cv, &q->mu);
        text_chunk *c = &q->ring[q->head];
        c->buf = map + at, c->len = end - at;
        q->head = (q->head + 1) % MUL_TEXT_RING, q->count++;
        pthread_cond_broadcast(&q->cv);
        pthread_mutex_unlock(&q->mu);
        at = end;
      }
      pthread_mutex_lock(&q->mu);
      q->eof = true;
      pthread_cond_broadcast(&q->cv);
      pthread_mutex_unlock(&q->mu);
      return NULL; /* the mapping stays until exit: the last chunks are still being parsed */
    }
  }
  char *carry = malloc(MUL_TEXT_CHUNK);
  size_t have = 0;
  for (;;) {
    pthread_mutex_lock(&q->mu);
    while (q->count == MUL_TEXT_RING) pthread_cond_wait(&q->cv, &q->mu);
    text_chunk *c = &q->ring[q->head];
    pthread_mutex_unlock(&q->mu);
    c->buf = c->own;
    memcpy(c->buf, carry, have);
    size_t got;
    while (have < MUL_TEXT_CHUNK && (got = fread(c->buf + have, 1, MUL_TEXT_CHUNK - have, stdin)) > 0) have += got;
    bool eof = have < MUL_TEXT_CHUNK;
    size_t end = have;
    if (!eof) {
      if (q->bin) end = have / 32 * 32;
      else {
        while (end > 0 && c->buf[end - 1] != '\n') end--;
        if (end == 0) end = have; /* one line longer than the chunk: taken as it is */
      }
    }
    memcpy(carry, c->buf + end, have - end);
    c->len = end, have -= end;
    pthread_mutex_lock(&q->mu);
    if (end) q->head = (q->head + 1) % MUL_TEXT_RING, q->count++;
    if (eof) q->eof = true;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
    if (eof) break;
  }
  free(carry);
  return NULL;
}
/* parsed arrays: parser -> device threads */
#define MUL_MAX_ARRAYS (MAX_GPUS + 2)
typedef struct { u64 (*ks)[4]; size_t cap, n; bool pinned; } scalar_array;
/* scalar arrays live in page-locked memory so that the GPUs read them by DMA (no staging copy in ecl_hip_mul_batch) */
static void ks_free(const ctx_t *ctx, u64 (*ks)[4], bool pinned) {
  (void)ctx;
  if (pinned) ecl_hip_free_host(ks);
  else free(ks);
}
static void ks_grow(const ctx_t *ctx, scalar_array *ar, size_t n) {
  if (n <= ar->cap) return;
  ks_free(ctx, ar->ks, ar->pinned);
  size_t cap = n + n / 8 + 1024;
  ar->ks = ctx->parse_only ? NULL : ecl_hip_alloc_host(cap * 32);
  ar->pinned = ar->ks != NULL;
  if (!ar->ks) ar->ks = malloc(cap * 32);
  ar->cap = cap;
}
typedef struct {
  ctx_t *ctx;
  scalar_array arr[MUL_MAX_ARRAYS];
  int narr;
  int ready[MUL_MAX_ARRAYS], nready; /* indices waiting for a device */
  int idle[MUL_MAX_ARRAYS], nidle;   /* indices free for the parser */
  bool done;
  pthread_mutex_t mu;
  pthread_cond_t cv;
} scalar_queue;
typedef struct { scalar_queue *q; int g; } mul_dev_arg;
static void *mul_device_worker(void *arg) {
  mul_dev_arg *a = arg;
  scalar_queue *q = a->q;
  const size_t STEP = 1u << 22; /* scalars per device call */
  for (;;) {
    pthread_mutex_lock(&q->mu);
    while (!q->nready && !q->done) pthread_cond_wait(&q->cv, &q->mu);
    if (!q->nready) { pthread_mutex_unlock(&q->mu); break; }
    int i = q->ready[0];
    memmove(q->ready, q->ready + 1, sizeof(int) * --q->nready);
    pthread_mutex_unlock(&q->mu);
    scalar_array *ar = &q->arr[i];
    if (q->ctx->parse_only) { /* hidden `parse` command: the scalars as the device would get them, one per line */
      for (size_t k = 0; k < ar->n; ++k)
        printf("%016llx%016llx%016llx%016llx\n", (unsigned long long)ar->ks[k][3], (unsigned long long)ar->ks[k][2],
               (unsigned long long)ar->ks[k][1], (unsigned long long)ar->ks[k][0]);
    } else
      for (size_t at = 0; at < ar->n; at += STEP) mul_flush(q->ctx, a->g, ar->ks + at, (u32)(ar->n - at < STEP ? ar->n - at : STEP));
    pthread_mutex_lock(&q->mu);
    q->idle[q->nidle++] = i;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
  }
  return NULL;
}
static void cmd_mul(ctx_t *ctx) {
  ctx->ts_started = tsnow();
  hexval_init();
#if defined(__x86_64__)
  have_ssse3 = __builtin_cpu_supports("ssse3");
#endif
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  int P = (int)(ncpu < 1 ? 1 : ncpu > 32 ? 32 : ncpu);
  text_queue tq;
  memset(&tq, 0, sizeof tq);
  tq.bin = ctx->bin_input;
  pthread_mutex_init(&tq.mu, NULL), pthread_cond_init(&tq.cv, NULL);
  for (int i = 0; i < MUL_TEXT_RING; ++i) tq.ring[i].own = tq.ring[i].buf = malloc(MUL_TEXT_CHUNK);
  scalar_queue sq;
  memset(&sq, 0, sizeof sq);
  sq.ctx = ctx, sq.narr = ctx->ngpus + 2;
  pthread_mutex_init(&sq.mu, NULL), pthread_cond_init(&sq.cv, NULL);
  for (int i = 0; i < sq.narr; ++i) sq.idle[sq.nidle++] = i;
  pthread_t reader, devth[MAX_GPUS];
  mul_dev_arg dargs[MAX_GPUS];
  pthread_create(&reader, NULL, mul_reader, &tq);
  for (int g = 0; g < ctx->ngpus; ++g) dargs[g] = (mul_dev_arg){&sq, g}, pthread_create(&devth[g], NULL, mul_device_worker, &dargs[g]);
  parse_slice sl[32];
  memset(sl, 0, sizeof sl);
  pool_t pool;
  pool_init(&pool, P);
  for (;;) {
    pthread_mutex_lock(&tq.mu);
    while (!tq.count && !tq.eof) pthread_cond_wait(&tq.cv, &tq.mu);
    if (!tq.count) { pthread_mutex_unlock(&tq.mu); break; }
    text_chunk *c = &tq.ring[tq.tail];
    pthread_mutex_unlock(&tq.mu);
    /* an array for this chunk's scalars */
    pthread_mutex_lock(&sq.mu);
    while (!sq.nidle) pthread_cond_wait(&sq.cv, &sq.mu);
    int ai = sq.idle[--sq.nidle];
    pthread_mutex_unlock(&sq.mu);
    scalar_array *ar = &sq.arr[ai];
    if (ctx->bin_input) { /* the scalars as they are: into the page-locked array, P threads copying */
      ar->n = c->len / 32;
      ks_grow(ctx, ar, ar->n);
      copy_task ct[32];
      size_t per = (ar->n + (size_t)P - 1) / (size_t)P;
      int nc = 0;
      for (size_t at = 0; at < ar->n; at += per, ++nc)
        ct[nc] = (copy_task){ar->ks + at, c->buf + at * 32, (ar->n - at < per ? ar->n - at : per) * 32};
      pool_run(&pool, copy_worker, ct, sizeof ct[0], nc);
    } else {
      int ns = 0;
      size_t at = 0, end = c->len;
      for (int i = 0; i < P && at < end; ++i) { /* slices at line boundaries */
        size_t stop = i == P - 1 ? end : at + (end - at) / (size_t)(P - i);
        if (stop <= at) stop = at + 1;
        while (stop < end && c->buf[stop - 1] != '\n') stop++;
        sl[ns].ctx = ctx, sl[ns].buf = c->buf, sl[ns].beg = at, sl[ns].end = stop;
        at = stop, ns++;
      }
      pool_run(&pool, parse_worker, sl, sizeof sl[0], ns);
      size_t total = 0;
      for (int i = 0; i < ns; ++i) total += sl[i].count;
      ks_grow(ctx, ar, total);
      ar->n = total;
      size_t off = 0;
      for (int i = 0; i < ns; ++i) sl[i].dst = ar->ks + off, off += sl[i].count;
      pool_run(&pool, pack_worker, sl, sizeof sl[0], ns);
    }
    pthread_mutex_lock(&tq.mu); /* the text buffer goes back to the reader */
    tq.tail = (tq.tail + 1) % MUL_TEXT_RING, tq.count--;
    pthread_cond_broadcast(&tq.cv);
    pthread_mutex_unlock(&tq.mu);
    pthread_mutex_lock(&sq.mu);
    sq.ready[sq.nready++] = ai;
    pthread_cond_broadcast(&sq.cv);
    pthread_mutex_unlock(&sq.mu);
  }
  pthread_mutex_lock(&sq.mu);
  sq.done = true;
  pthread_cond_broadcast(&sq.cv);
  pthread_mutex_unlock(&sq.mu);
  pool_stop(&pool);
  pthread_join(reader, NULL);
  for (int g = 0; g < ctx->ngpus; ++g) pthread_join(devth[g], NULL);
  for (int i = 0; i < MUL_TEXT_RING; ++i) free(tq.ring[i].own);
  for (int i = 0; i < sq.narr; ++i) ks_free(ctx, sq.arr[i].ks, sq.arr[i].pinned);
  for (int i = 0; i < 32; ++i) free(sl[i].tmp);
  if (!ctx->parse_only) ctx_finish(ctx);
}

/* ------------------------------------------------------------------------------------------- rnd */
static u64 rand64(bool urandom) { /* utils.c:83-113 */
  u64 r = 0;
  if (urandom) {
    FILE *f = fopen("/dev/urandom", "rb");
    if (!f || fread(&r, 8, 1, f) != 1) { fprintf(stderr, "failed to read /dev/urandom\n"); exit(1); }
    fclose(f);
    return r;
  }
  return (u64)rand() << 32 | (u64)rand();
}
static sc sc_rand_range(const sc *a, const sc *b, bool urandom) { /* uniform-ish value in [a, b) (utils.c:115-153) */
  sc span, r;
  sc_subraw(&span, b, a);
  unsigned bits = sc_bitlen(&span);
  for (;;) {
    for (int i = 0; i < 4; ++i) r.w[i] = rand64(urandom);
    for (unsigned i = bits; i < 256; ++i) r.w[i >> 6] &= ~(1ULL << (i & 63));
    if (sc_cmp(&r, &span) < 0) break;
  }
  sc_addraw(&r, &r, a);
  return r;
}
static void print_range_mask(const sc *v, u32 bits_size, u32 offset, bool color) { /* main.c:593-617 */
  int mask_e = 255 - (int)offset, mask_s = mask_e - (int)bits_size + 1;
  for (int i = 0; i < 64; i++) {
    if (i % 16 == 0 && i != 0) putchar(' ');
    int bs = i * 4, be = bs + 3;
    u32 nib = (v->w[(255 - be) / 64] >> ((255 - be) % 64)) & 0xF;
    bool flag = (bs >= mask_s && bs <= mask_e) || (be >= mask_s && be <= mask_e);
    if (flag && color) fputs("\033[33m", stdout);
    putchar("0123456789abcdef"[nib]);
    if (flag && color) fputs("\033[0m", stdout);
  }
  putchar('\n');
}
/* cmd_rnd (main.c:619-662): random value in [A,B], bits offs..offs+size-1 cleared / set -> window, scanned like add */
static void cmd_rnd(ctx_t *ctx) {
  if (ctx->ord_offs > 255 - ctx->ord_size) ctx->ord_offs = 255 - ctx->ord_size;
  printf("[RANDOM MODE] offs: %d ~ bits: %d\n\n", ctx->ord_offs, ctx->ord_size);
  ctx->ts_started = tsnow();
  sc a = ctx->range_s, b = ctx->range_e;
  const char *mw = getenv("ECLOOP_HIP_RND_WINDOWS");
  u64 max_windows = mw ? strtoull(mw, NULL, 10) : 0, windows = 0;
  for (;;) {
    u64 last_c = ctx->k_checked, last_f = ctx->k_found, t0 = tsnow();
    sc s = sc_rand_range(&a, &b, !ctx->has_seed), e = s; /* gen_random_range, main.c:580-591 */
    for (u32 i = ctx->ord_offs; i < ctx->ord_offs + ctx->ord_size; ++i) s.w[i / 64] &= ~(1ULL << (i % 64)), e.w[i / 64] |= 1ULL << (i % 64);
    if (sc_cmp(&s, &a) <= 0) s = a;
    if (sc_cmp(&e, &b) >= 0) e = b;
    print_range_mask(&s, ctx->ord_size, ctx->ord_offs, ctx->use_color);
    print_range_mask(&e, ctx->ord_size, ctx->ord_offs, ctx->use_color);
    bool is_full = sc_cmp(&s, &a) == 0 && sc_cmp(&e, &b) == 0;
    if (sc_cmp(&s, &e) < 0) scan_range(ctx, s, e, true);
    u64 dc = ctx->k_checked - last_c, df = ctx->k_found - last_f;
    double dt = (tsnow() - t0 < 1 ? 1 : tsnow() - t0) / 1000.0;
    term_clear_line();
    printf("%'llu / %'llu ~ %.1fs\n\n", (unsigned long long)df, (unsigned long long)dc, dt);
    if (is_full) break;
    /* the reference loops until interrupted; ECLOOP_HIP_RND_WINDOWS=N (tests, timing runs) stops after N windows */
    if (max_windows && ++windows >= max_windows) break;
  }
  if (getenv("ECLOOP_HIP_STATS")) { /* where the device time of the windows went: search kernel vs per-window set-up */
    for (int g = 0; g < ctx->ngpus; ++g) {
      double kms = 0, sms = 0;
      u64 launches = 0, keys = 0, setups = 0;
      ecl_hip_get_timing(ctx->dev[g], &kms, &launches, &keys);
      ecl_hip_get_setup_timing(ctx->dev[g], &sms, &setups);
      printf("gpu %d: %llu launches, %.3f ms in the search kernel, %llu set-ups, %.3f ms in set-up kernels (%.2f %%)\n", g,
             (unsigned long long)launches, kms, (unsigned long long)setups, sms, kms > 0 ? 100.0 * sms / (kms + sms) : 0.0);
    }
  }
  ctx_finish(ctx);
}

/* ------------------------------------------------------------------------------------------- blf-gen / blf-check */
static void blf_gen(args_t *args) { /* utils.c:409-475 */
  u64 n = args_uint(args, "-n", 0);
  const char *path = arg_str(args, "-o");
  if (!n || !path) {
    fprintf(stderr, "Usage: %s blf-gen -n <count> -o <file>   (hex hash160 list on stdin)\n", args->argv[0]);
    exit(1);
  }
  u64 r = 1000000000ull;
  double p = 1.0 / (double)r;
  u64 m = (u64)(n * log(p) / log(1.0 / pow(2.0, log(2.0))));
  double mb = (double)m / 8 / 1024 / 1024;
  u64 size = (m + 63) / 64;
  blf_t blf = {0, NULL};
  if (access(path, F_OK) == 0) {
    printf("file %s already exists; loading...\n", path);
    if (!blf_load(path, &blf)) { fprintf(stderr, "[!] failed to load bloom filter: delete it or choose a different file\n"); exit(1); }
    if (blf.size != size) { fprintf(stderr, "[!] bloom filter size mismatch (%'llu != %'llu)\n", (unsigned long long)blf.size, (unsigned long long)size); exit(1); }
    printf("updating bloom filter...\n");
  } else {
    printf("creating bloom filter...\n");
    blf.size = size, blf.bits = calloc(size, 8);
  }
  printf("bloom filter params: n = %'llu | p = 1:%'llu | m = %'llu (%'.1f MB)\n", (unsigned long long)n, (unsigned long long)r, (unsigned long long)m, mb);
  u64 count = 0;
  char line[41];
  /* Filters sized for 2^16 entries and more are filled on the GPU when one is visible (`-host` keeps it here): the
     same 20 bits per hash by atomic ORs, and the same "new items" count as this loop gives in input order
     (ecl_hip_bloom_insert_count); the file written is byte-identical either way. */
  ecl_hip *dev = NULL;
  if (n >= (1u << 16) && !args_bool(args, "-host") && ecl_hip_device_count() > 0) {
    int rc = ecl_hip_open(&dev, 0, ECL_ADDR33, 0);
    if (rc == ECL_OK) rc = ecl_hip_set_bloom(dev, blf.bits, blf.size);
    if (rc != ECL_OK) { fprintf(stderr, "[!] GPU set-up failed: %s (%s)\n", ecl_hip_strerror(rc), dev ? ecl_hip_last_error(dev) : ""); exit(1); }
    printf("inserting on GPU 0\n");
  }
  if (dev) {
    const size_t BATCH = 1u << 22;
    u32 (*hs)[5] = malloc(BATCH * 20);
    size_t have = 0;
    bool more = true;
    while (more) {
      more = fgets(line, sizeof line, stdin) != NULL;
      if (more && strlen(line) == 40 && parse_hash40(line, hs[have])) have++;
      if (have == BATCH || (!more && have)) {
        u64 added = 0;
        int rc = ecl_hip_bloom_insert_count(dev, (const uint32_t(*)[5])hs, have, &added);
        if (rc != ECL_OK) { fprintf(stderr, "[!] GPU insert failed: %s (%s)\n", ecl_hip_strerror(rc), ecl_hip_last_error(dev)); exit(1); }
        count += added, have = 0;
      }
    }
    int rc = ecl_hip_get_bloom(dev, blf.bits, blf.size);
    if (rc != ECL_OK) { fprintf(stderr, "[!] reading the filter back failed: %s\n", ecl_hip_strerror(rc)); exit(1); }
    ecl_hip_close(dev);
    free(hs);
  } else
    while (fgets(line, sizeof line, stdin)) {
      u32 h[5];
      if (strlen(line) != 40 || !parse_hash40(line, h)) continue;
      if (blf_has(&blf, h)) continue;
      blf_add(&blf, h), count++;
    }
  printf("added %'llu new items; saving to %s\n", (unsigned long long)count, path);
  if (!blf_save(path, &blf)) { fprintf(stderr, "[!] failed to save bloom filter\n"); exit(1); }
}
static void blf_check(args_t *args) { /* utils.c:495-529 */
  const char *path = arg_str(args, "-f");
  blf_t blf = {0, NULL};
  if (!path || !blf_load(path, &blf)) { fprintf(stderr, "Usage: %s blf-check -f <file> <hash> [hash...]\n", args->argv[0]); exit(1); }
  bool any = false;
  for (int i = 1; i < args->argc; ++i) {
    u32 h[5];
    if (strlen(args->argv[i]) != 40 || !parse_hash40(args->argv[i], h)) continue;
    any = true;
    printf("%s %s\n", args->argv[i], blf_has(&blf, h) ? "FOUND" : "NOT FOUND");
  }
  if (any) return;
  char line[128];
  while (fgets(line, sizeof line, stdin)) {
    line[strcspn(line, "\r\n")] = 0;
    u32 h[5];
    if (strlen(line) != 40 || !parse_hash40(line, h)) continue;
    printf("%s %s\n", line, blf_has(&blf, h) ? "FOUND" : "NOT FOUND");
  }
}

/* ------------------------------------------------------------------------------------------- argument handling */
static void arg_search_range(args_t *args, sc *rs, sc *re) { /* main.c:666-701 */
  const char *raw = arg_str(args, "-r");
  if (!raw) { *rs = sc_u64(GROUP_INV_SIZE), *re = SC_P; return; }
  char *tmp = strdup(raw), *sep = strchr(tmp, ':');
  if (!sep) { fprintf(stderr, "invalid search range, use format: -r 8000:ffff\n"); exit(1); }
  *sep = 0;
  *rs = sc_from_hex(tmp), *re = sc_from_hex(sep + 1);
  free(tmp);
  sc lim = sc_u64(GROUP_INV_SIZE);
  if (sc_cmp(rs, &lim) <= 0) { fprintf(stderr, "invalid search range, start <= %#llx\n", (unsigned long long)GROUP_INV_SIZE); exit(1); }
  if (sc_cmp(re, &SC_P) > 0) { fprintf(stderr, "invalid search range, end > FE_P\n"); exit(1); }
  if (sc_cmp(rs, re) >= 0) { fprintf(stderr, "invalid search range, start >= end\n"); exit(1); }
}
static void load_offs_size(ctx_t *ctx, args_t *args) { /* main.c:703-746 */
  const u32 MIN_SIZE = 20, MAX_SIZE = 64;
  u32 range_bits = sc_bitlen(&ctx->range_e);
  u32 default_bits = range_bits < 32 ? (MIN_SIZE > range_bits ? MIN_SIZE : range_bits) : 32;
  u32 mx = MIN_SIZE > range_bits ? MIN_SIZE : range_bits;
  u32 max_offs = mx - default_bits > 1 ? mx - default_bits : 1;
  const char *raw = arg_str(args, "-d");
  if (!raw) {
    ctx->ord_offs = ctx->cmd == CMD_RND ? (u32)(rand64(!ctx->has_seed) % max_offs) : 0;
    ctx->ord_size = default_bits;
    return;
  }
  const char *sep = strchr(raw, ':');
  if (!sep) { fprintf(stderr, "invalid offset:size format, use format: -d 128:32\n"); exit(1); }
  u32 offs = (u32)atoi(raw), size = (u32)atoi(sep + 1);
  if (offs > 255) { fprintf(stderr, "invalid offset, max is 255\n"); exit(1); }
  if (size < MIN_SIZE || size > MAX_SIZE) { fprintf(stderr, "invalid size, min is %d and max is %d\n", MIN_SIZE, MAX_SIZE); exit(1); }
  ctx->ord_offs = offs < max_offs ? offs : max_offs;
  ctx->ord_size = size;
}
static void usage(const char *name) { /* main.c:750-772 */
  printf("Usage: %s <cmd> [-t <gpus>] [-f <file>] [-a <addr_type>] [-r <range>]\n", name);
  printf("v%s ~ MI355X build of the ecloop command set\n", VERSION);
  printf("\nCompute commands:\n");
  printf("  add             - search in given range with batch addition\n");
  printf("  mul             - search hex encoded private keys (from stdin)\n");
  printf("  rnd             - search random range of bits in given range\n");
  printf("\nCompute options:\n");
  printf("  -f <file>       - filter file to search (list of hashes or bloom fitler)\n");
  printf("  -o <file>       - output file to write found keys (default: stdout)\n");
  printf("  -t <gpus>       - number of GPUs to use (default: all)\n");
  printf("  -a <addr_type>  - address type to search: c - addr33, u - addr65 (default: c)\n");
  printf("  -r <range>      - search range in hex format (example: 8000:ffff, default all)\n");
  printf("  -d <offs:size>  - bit offset and size for search (example: 128:32, default: 0:32)\n");
  printf("  -q              - quiet mode (no output to stdout; -o required)\n");
  printf("  -endo           - use endomorphism (default: false)\n");
  printf("  -bin            - mul: stdin carries 32-byte little-endian scalars instead of hex lines\n");
  printf("\nOther commands:\n");
  printf("  blf-gen         - create bloom filter from list of hex-encoded hash160\n");
  printf("  blf-check       - check bloom filter for given hex-encoded hash160\n");
  printf("  bench           - run benchmark of the device paths (add per address type / endo, mul)\n\n");
}
/* pause / resume from the terminal (lib/utils.c:559-626, main.c:874-888): /dev/tty in non-canonical mode, one
   listener thread; 'p' stops the device threads at their next status update, 'r' lets them go on; paused time is
   taken out of the rate.  Without a controlling terminal (pipes, batch jobs) nothing is installed. */
static int tty_fd = -1;
static struct termios tty_orig;
static bool tty_is_term;
static void tty_cleanup(void) {
  if (tty_fd < 0) return;
  if (tty_is_term) tcsetattr(tty_fd, TCSANOW, &tty_orig);
  close(tty_fd), tty_fd = -1;
}
static void tty_key(ctx_t *ctx, char ch) {
  if (ch == 'p' && !ctx->paused) {
    ctx->ts_paused_at = tsnow(), ctx->paused = true;
    pthread_mutex_lock(&ctx->lock), ctx_print_unlocked(ctx), pthread_mutex_unlock(&ctx->lock);
  }
  if (ch == 'r' && ctx->paused) {
    ctx->paused_time += tsnow() - ctx->ts_paused_at, ctx->paused = false;
    pthread_mutex_lock(&ctx->lock), ctx_print_unlocked(ctx), pthread_mutex_unlock(&ctx->lock);
  }
}
static void *tty_listener(void *arg) {
  ctx_t *ctx = arg;
  for (;;) {
    int fd = tty_fd;
    if (fd < 0) break;
    fd_set fds;
    FD_ZERO(&fds);
    FD_SET(fd, &fds);
    struct timeval tv = {0, 200000};
    int r = select(fd + 1, &fds, NULL, NULL, &tv);
    if (r < 0) break;
    char ch;
    if (r > 0 && FD_ISSET(fd, &fds) && read(fd, &ch, 1) > 0) tty_key(ctx, ch);
  }
  return NULL;
}
static void tty_init(ctx_t *ctx) {
  const char *path = getenv("ECLOOP_HIP_TTY"); /* where the keys come from; a FIFO works too (containers without ptys) */
  tty_fd = open(path ? path : "/dev/tty", (path ? O_RDWR : O_RDONLY) | O_NONBLOCK);
  if (tty_fd < 0) return;
  atexit(tty_cleanup);
  tty_is_term = tcgetattr(tty_fd, &tty_orig) == 0;
  if (tty_is_term) {
    struct termios raw = tty_orig;
    raw.c_lflag &= ~(tcflag_t)(ICANON | ECHO);
    tcsetattr(tty_fd, TCSANOW, &raw);
  } else if (!path) {
    close(tty_fd), tty_fd = -1;
    return;
  }
  pthread_t th;
  if (pthread_create(&th, NULL, tty_listener, ctx) == 0) pthread_detach(th);
}

static void handle_sigint(int sig) {
  fflush(stderr), fflush(stdout);
  printf("\n");
  exit(sig);
}

/* `bench` (the reference's `bench` / `bench-gtable`, lib/bench.c, time its CPU primitives): here the device paths,
   through the C ABI, with an empty filter: keys/s of the add walk per address / endo selection, scalars/s of mul. */
static int run_bench(args_t *args) {
  if (ecl_hip_device_count() <= 0) { fprintf(stderr, "no MI355X GPU visible (the search path has no CPU fallback)\n"); return 1; }
  u64 lg = args_uint(args, "-n", 31);
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
    u64 t0 = tsnow();
    if (rc == ECL_OK) rc = ecl_hip_add_range(d, start, n, hit, 16, &cnt);
    u64 t1 = tsnow();
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
    u64 t0 = tsnow();
    for (int r = 0; r < 4 && rc == ECL_OK; ++r) rc = ecl_hip_mul_batch(d, ks, n, hit, 16, &cnt);
    u64 t1 = tsnow();
    if (rc != ECL_OK) { fprintf(stderr, "[!] bench mul: %s\n", ecl_hip_strerror(rc)); return 1; }
    printf("%-18s 2^22 keys: %9.2f M it/s (scalars copied from host memory)\n", "mul -a cu", 4.0 * n / ((t1 - t0 ? t1 - t0 : 1) / 1000.0) / 1e6);
    free(ks);
    ecl_hip_close(d);
  }
  return 0;
}

typedef struct { ctx_t *ctx; int g, device; u32 flags; u64 share; int rc; u64 t[5]; } open_job;
static u64 usnow(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return (u64)tv.tv_sec * 1000000 + tv.tv_usec;
}
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

static void *open_worker(void *arg) {
  open_job *j = arg;
  ctx_t *ctx = j->ctx;
  j->t[0] = usnow();
  int rc = ecl_hip_open(&ctx->dev[j->g], j->device, j->flags, ctx->cmd == CMD_MUL ? 0 : ctx->ord_offs);
  j->t[1] = usnow();
  if (rc == ECL_OK) rc = ecl_hip_set_bloom(ctx->dev[j->g], ctx->blf.bits, ctx->blf.size);
  j->t[2] = usnow();
  if (rc == ECL_OK && ctx->list) rc = ecl_hip_set_list(ctx->dev[j->g], (const uint32_t(*)[5])ctx->list, ctx->list_count);
  j->t[3] = usnow();
  if (rc == ECL_OK && j->share) rc = ecl_hip_reserve(ctx->dev[j->g], j->share, 4096);
  j->t[4] = usnow();
  j->rc = rc;
  return NULL;
}

int main(int argc, const char **argv) {
  setlocale(LC_NUMERIC, "");
  args_t args = {argc, argv};
  static ctx_t ctx;
  if (argc > 1) {
    if (!strcmp(argv[1], "blf-gen")) return blf_gen(&args), 0;
    if (!strcmp(argv[1], "blf-check")) return blf_check(&args), 0;
    if (!strcmp(argv[1], "bench")) return run_bench(&args);
    if (!strcmp(argv[1], "parse")) { /* hidden: `mul`'s text front end alone (no GPU), for the parser tests */
      ctx.cmd = CMD_MUL, ctx.parse_only = true, ctx.ngpus = 1;
      ctx.raw_text = args_bool(&args, "-raw"), ctx.bin_input = args_bool(&args, "-bin");
      pthread_mutex_init(&ctx.lock, NULL);
      cmd_mul(&ctx);
      return 0;
    }
    if (!strcmp(argv[1], "add")) ctx.cmd = CMD_ADD;
    if (!strcmp(argv[1], "plan")) ctx.cmd = CMD_ADD, ctx.plan_only = true;
    if (!strcmp(argv[1], "mul")) ctx.cmd = CMD_MUL;
    if (!strcmp(argv[1], "rnd")) ctx.cmd = CMD_RND;
  }
  if (ctx.cmd == CMD_NIL) {
    if (args_bool(&args, "-v")) printf("ecloop-hip v%s\n", VERSION);
    else usage(argv[0]);
    return 0;
  }
  ctx.use_color = isatty(fileno(stdout));
  const char *seed = arg_str(&args, "-seed");
  if (seed) {
    u32 s = 5381;
    for (const char *c = seed; *c; ++c) s = s * 33 + (u8)*c;
    ctx.has_seed = true, srand(s); /* the reference free()s an argv pointer here and aborts (main.c:800-805) */
  }
  if (!ctx.plan_only) load_filter(&ctx, arg_str(&args, "-f"));
  ctx.quiet = args_bool(&args, "-q");
  const char *outfile = arg_str(&args, "-o");
  if (outfile) ctx.outfile = fopen(outfile, "a");
  if (!outfile && ctx.quiet) { fprintf(stderr, "quiet mode chosen without output file\n"); return 1; }
  const char *addr = arg_str(&args, "-a");
  if (addr) ctx.a33 = strchr(addr, 'c') != NULL, ctx.a65 = strchr(addr, 'u') != NULL;
  if (!ctx.a33 && !ctx.a65) ctx.a33 = true;
  ctx.endo = args_bool(&args, "-endo") && ctx.cmd != CMD_MUL;
  ctx.raw_text = args_bool(&args, "-raw");
  ctx.bin_input = args_bool(&args, "-bin") && ctx.cmd == CMD_MUL;
  pthread_mutex_init(&ctx.lock, NULL);
  ctx.ts_started = ctx.ts_updated = tsnow();
  ctx.ts_printed = ctx.ts_started - 5000;
  arg_search_range(&args, &ctx.range_s, &ctx.range_e);
  load_offs_size(&ctx, &args);
  ctx.stride_k = sc_pow2(ctx.cmd == CMD_MUL ? 0 : ctx.ord_offs);

  if (ctx.plan_only) { /* hidden `plan`: the job arithmetic of `add` / `rnd` for -r / -d, no GPU (tests) */
    scan_t sn;
    ctx.ngpus = (int)args_uint(&args, "-t", 1);
    if (arg_str(&args, "-visible")) { /* the context -> GPU map of `-t N` on a box with that many GPUs */
      int real = (int)args_uint(&args, "-visible", 1), shown = ctx.ngpus > real ? real : ctx.ngpus, dev_of[MAX_GPUS];
      int n = context_devices(args_bool(&args, "-mul") ? CMD_MUL : CMD_ADD, shown, real, dev_of);
      printf("contexts %d gpus %d devices", n, shown);
      for (int g = 0; g < n; ++g) printf(" %d", dev_of[g]);
      printf("\n");
      return 0;
    }
    scan_plan(&ctx, ctx.range_s, ctx.range_e, args_bool(&args, "-rnd"), &sn);
    printf("ord_offs %u ord_size %u hashed %016llx%016llx%016llx%016llx status_total %llu chunk %llu\n", ctx.ord_offs, ctx.ord_size,
           (unsigned long long)sn.hashed.w[3], (unsigned long long)sn.hashed.w[2], (unsigned long long)sn.hashed.w[1],
           (unsigned long long)sn.hashed.w[0], (unsigned long long)sn.status_total, (unsigned long long)sn.chunk);
    return 0;
  }
  int have = ecl_hip_device_count(), real = have;
  /* test hook: ECLOOP_HIP_SHARE_GPU=N runs N device threads over the GPUs that exist (device g mod count), so the
     multi-GPU sharding / merging logic can be exercised on a one-GPU box */
  if (have > 0 && getenv("ECLOOP_HIP_SHARE_GPU")) have = atoi(getenv("ECLOOP_HIP_SHARE_GPU")) > 0 ? atoi(getenv("ECLOOP_HIP_SHARE_GPU")) : have;
  if (have <= 0) { fprintf(stderr, "no MI355X GPU visible (the search path has no CPU fallback)\n"); return 1; }
  u64 want = args_uint(&args, "-t", (u64)have);
  ctx.ngpus = (int)(want < 1 ? 1 : want > (u64)have ? (u64)have : want);
  if (ctx.ngpus > MAX_GPUS) ctx.ngpus = MAX_GPUS;
  int gpus_shown = ctx.ngpus, dev_of[MAX_GPUS];
  ctx.ngpus = context_devices(ctx.cmd, gpus_shown, real, dev_of);
  /* Device bring-up, all GPUs at once (one host thread each): context, filter upload from the one pinned host copy
     (every GPU over its own PCIe link), optional list, and the walk buffers of the chunks this scan will hand out -
     all before the clock of the status line starts; the time it took is printed in the banner. */
  u64 t_setup0 = tsnow();
  bool pinned = ctx.blf.size >= (8u << 20) && ecl_hip_pin_host(ctx.blf.bits, ctx.blf.size * 8) == ECL_OK;
  {
    pthread_t th[MAX_GPUS];
    open_job jobs[MAX_GPUS];
    u64 share = 0;
    if (ctx.cmd != CMD_MUL) {
      /* keys of the largest device call: see scan_chunk() */
      sc hashed;
      if (ctx.cmd == CMD_RND) hashed = sc_u64(1ull << (ctx.ord_size < 21 ? 21 : ctx.ord_size > 62 ? 62 : ctx.ord_size));
      else {
        sc_subraw(&hashed, &ctx.range_e, &ctx.range_s);
        for (u32 i = 0; i < ctx.ord_offs && i < 256; ++i) hashed = sc_shr1(hashed);
        hashed = sc_add_u64_raw(hashed, 4096);
      }
      share = scan_chunk(&ctx, &hashed);
      if (!(hashed.w[1] | hashed.w[2] | hashed.w[3]) && hashed.w[0] < share) share = hashed.w[0];
    }
    for (int g = 0; g < ctx.ngpus; ++g) {
      jobs[g] = (open_job){&ctx, g, dev_of[g], (ctx.a33 ? ECL_ADDR33 : 0) | (ctx.a65 ? ECL_ADDR65 : 0) | (ctx.endo ? ECL_ENDO : 0), share, ECL_OK};
      pthread_create(&th[g], NULL, open_worker, &jobs[g]);
    }
    for (int g = 0; g < ctx.ngpus; ++g) pthread_join(th[g], NULL);
    for (int g = 0; g < ctx.ngpus; ++g)
      if (jobs[g].rc != ECL_OK) die_ecl(&ctx, g, jobs[g].rc, "open");
    if (getenv("ECLOOP_HIP_STATS"))
      for (int g = 0; g < ctx.ngpus; ++g)
        printf("gpu %d bring-up: open %.1f ms, filter upload %.1f ms, list %.1f ms, reserve(%llu keys) %.1f ms\n", g,
               (jobs[g].t[1] - jobs[g].t[0]) / 1e3, (jobs[g].t[2] - jobs[g].t[1]) / 1e3, (jobs[g].t[3] - jobs[g].t[2]) / 1e3,
               (unsigned long long)share, (jobs[g].t[4] - jobs[g].t[3]) / 1e3);
  }
  if (pinned) ecl_hip_unpin_host(ctx.blf.bits);
  double setup_s = (tsnow() - t_setup0) / 1000.0;
  printf("gpus: %d ~ addr33: %d ~ addr65: %d ~ endo: %d | filter: ", gpus_shown, ctx.a33, ctx.a65, ctx.endo);
  if (ctx.list) printf("list (%'llu)\n", (unsigned long long)ctx.list_count);
  else printf("bloom\n");
  if (ctx.cmd == CMD_ADD) {
    printf("range_s: %016llx %016llx %016llx %016llx\n", (unsigned long long)ctx.range_s.w[3], (unsigned long long)ctx.range_s.w[2], (unsigned long long)ctx.range_s.w[1], (unsigned long long)ctx.range_s.w[0]);
    printf("range_e: %016llx %016llx %016llx %016llx\n", (unsigned long long)ctx.range_e.w[3], (unsigned long long)ctx.range_e.w[2], (unsigned long long)ctx.range_e.w[1], (unsigned long long)ctx.range_e.w[0]);
  }
  printf("setup: %.2fs (%d device context%s opened in parallel, %.0f MB filter uploaded, walk buffers reserved)\n", setup_s, ctx.ngpus,
         ctx.ngpus == 1 ? "" : "s", ctx.blf.size * 8 / 1e6);
  printf("----------------------------------------\n");
  fflush(stdout);
  signal(SIGINT, handle_sigint);
  tty_init(&ctx);
  if (ctx.cmd == CMD_ADD) cmd_add(&ctx);
  if (ctx.cmd == CMD_MUL) cmd_mul(&ctx);
  if (ctx.cmd == CMD_RND) cmd_rnd(&ctx);
  for (int g = 0; g < ctx.ngpus; ++g) ecl_hip_close(ctx.dev[g]);
  return 0;
}
