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
static u64 ms_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (u64)ts.tv_sec * 1000 + (u64)ts.tv_nsec / 1000000;
}
static u64 us_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (u64)ts.tv_sec * 1000000 + (u64)ts.tv_nsec / 1000;
}
static void erase_status_line(void) { fputs("\033[2K\r", stderr); }

/* hex digits: table for the general readers, 16 characters at a time where SSSE3 is there */
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
#endif
static bool have_ssse3; /* set once in main */

/* ------------------------------------------------------------------------------------------- command line */
/* Every option of every command, parsed in ONE pass over argv into this struct: a flag that takes a value consumes the
   next argument, anything else is left alone (`blf-check` reads hashes from the bare words).  Spelling and meaning of
   the reference's flags (main.c:794-862, utils.c:157-185), plus -bin / -host and the switches of the hidden test
   commands (plan: -rnd -mul -visible). */
typedef struct {
  const char *filter, *outfile, *range, *window, *seed, *addr, *gpus, *count, *visible;
  bool quiet, endo, raw, bin, version, host_only, rnd_jobs, as_mul;
} opts_t;
typedef struct { const char *flag; size_t at; bool takes_value; } optdef_t;
static const optdef_t OPTDEFS[] = {
    {"-f", offsetof(opts_t, filter), true},       {"-o", offsetof(opts_t, outfile), true},  {"-r", offsetof(opts_t, range), true},
    {"-d", offsetof(opts_t, window), true},       {"-seed", offsetof(opts_t, seed), true},  {"-a", offsetof(opts_t, addr), true},
    {"-t", offsetof(opts_t, gpus), true},         {"-n", offsetof(opts_t, count), true},    {"-visible", offsetof(opts_t, visible), true},
    {"-q", offsetof(opts_t, quiet), false},       {"-endo", offsetof(opts_t, endo), false}, {"-raw", offsetof(opts_t, raw), false},
    {"-bin", offsetof(opts_t, bin), false},       {"-v", offsetof(opts_t, version), false}, {"-host", offsetof(opts_t, host_only), false},
    {"-rnd", offsetof(opts_t, rnd_jobs), false},  {"-mul", offsetof(opts_t, as_mul), false},
};
static void opts_parse(opts_t *o, int argc, const char **argv) {
  memset(o, 0, sizeof *o);
  for (int i = 1; i < argc; ++i)
    for (size_t d = 0; d < sizeof OPTDEFS / sizeof OPTDEFS[0]; ++d) {
      if (strcmp(argv[i], OPTDEFS[d].flag) != 0) continue;
      char *field = (char *)o + OPTDEFS[d].at;
      if (!OPTDEFS[d].takes_value) *(bool *)field = true;
      else if (i + 1 < argc && !*(const char **)field) *(const char **)field = argv[++i];
      break;
    }
}
static u64 opt_number(const char *text, u64 fallback) { return text ? strtoull(text, NULL, 10) : fallback; }

/* ------------------------------------------------------------------------------------------- filter (host side) */
/* What -f names: the bloom words the GPUs probe, and - when the file was a hash list - the sorted list that confirms a
   bloom hit exactly (ctx->blf + ctx->to_find_hashes, main.c:48-51).  `.blf` files carry the words only. */
#define BLF_MAGIC 0x45434246u /* utils.c:274-275: 'ECBF', version 1, u64 word count, words */
#define BLF_VERSION 1u
typedef struct {
  u64 *words, nwords;
  u32 *list; /* nlist x 5 words, ascending, unique; NULL = bloom-only mode */
  u64 nlist;
} filter_t;

/* the 20 bit positions of a hash160 (utils.c:290-306): five overlapping 64-bit words, shifted by 24 / 28 / 36 / 40 */
static void bloom_positions(u64 pos[20], const u32 h[5]) {
  u64 a[6];
  for (int j = 0; j < 5; ++j) a[j] = (u64)h[(2 * j) % 5] << 32 | h[(2 * j + 1) % 5];
  a[5] = a[0];
  static const int SHIFT[4] = {24, 28, 36, 40};
  for (int p = 0; p < 20; ++p) pos[p] = a[p % 5] << SHIFT[p / 5] | a[p % 5 + 1] >> SHIFT[p / 5];
}
static void bloom_set(filter_t *f, const u32 h[5]) {
  u64 pos[20];
  bloom_positions(pos, h);
  for (int p = 0; p < 20; ++p) f->words[(pos[p] >> 6) % f->nwords] |= 1ULL << (pos[p] & 63);
}
static bool bloom_test(const filter_t *f, const u32 h[5]) {
  u64 pos[20];
  bloom_positions(pos, h);
  int p = 0;
  while (p < 20 && ((f->words[(pos[p] >> 6) % f->nwords] >> (pos[p] & 63)) & 1)) ++p;
  return p == 20;
}
static bool blf_write(const char *path, const filter_t *f) { /* utils.c:328-360 */
  FILE *out = fopen(path, "wb");
  if (!out) return false;
  struct { u32 magic, version; u64 nwords; } head = {BLF_MAGIC, BLF_VERSION, f->nwords};
  bool ok = fwrite(&head, sizeof head, 1, out) == 1 && fwrite(f->words, 8, f->nwords, out) == f->nwords;
  return fclose(out) == 0 && ok;
}
/* utils.c:362-396; NULL on success, else the reference's message for what went wrong */
static const char *blf_read(const char *path, filter_t *f) {
  FILE *in = fopen(path, "rb");
  if (!in) return "failed to open input file";
  struct { u32 magic, version; u64 nwords; } head;
  const char *why = NULL;
  u64 *words = NULL;
  if (fread(&head, sizeof head, 1, in) != 1) why = "failed to read bloom filter header";
  else if (head.magic != BLF_MAGIC || head.version != BLF_VERSION) why = "invalid bloom filter version; create a new filter with blf-gen command";
  else {
    words = calloc(head.nwords ? head.nwords : 1, 8);
    if (!words || fread(words, 8, head.nwords, in) != head.nwords) why = "failed to read bloom filter bits";
  }
  fclose(in);
  if (why) { free(words); return why; }
  f->words = words, f->nwords = head.nwords;
  return NULL;
}
static int order160(const void *a, const void *b) { /* compare_160, addr.c:18-26: word by word */
  const u32 *x = a, *y = b;
  int i = 0;
  while (i < 4 && x[i] == y[i]) ++i;
  return (x[i] > y[i]) - (x[i] < y[i]);
}
/* 40 hex digits -> 5 words; false if any character is not a hex digit */
static bool hash160_from_hex(const char *s, u32 h[5]) {
#if defined(__x86_64__)
  if (have_ssse3) { /* 16 + 16 characters, then the last 8 padded with zeros on the left */
    u64 a, b, c;
    char tail[16] = {'0', '0', '0', '0', '0', '0', '0', '0'};
    memcpy(tail + 8, s + 32, 8);
    if (!hex16_ssse3(s, &a) || !hex16_ssse3(s + 16, &b) || !hex16_ssse3(tail, &c)) return false;
    h[0] = (u32)(a >> 32), h[1] = (u32)a, h[2] = (u32)(b >> 32), h[3] = (u32)b, h[4] = (u32)c;
    return true;
  }
#endif
  for (int w = 0; w < 5; ++w) {
    u32 v = 0;
    for (int d = 0; d < 8; ++d) {
      int x = HEXVAL[(u8)s[w * 8 + d]];
      if (x < 0) return false;
      v = v << 4 | (u32)x;
    }
    h[w] = v;
  }
  return true;
}
/* Entries of a hash list, as the reference's reader sees them (main.c:96-110: fgets into a 41-byte buffer consumes a line
   in pieces of 40 characters, and every FULL piece is an entry).  Stated on the file image: cut at '\n', walk each line
   in steps of 40, keep the pieces that are 40 clean hex digits (the reference parses garbage out of the others - one
   phantom entry for the comment line of data/btc-bw-hash; dropped here).  `out` has room for len / 40 + 1
   entries (no piece is shorter than 40 characters). */
static size_t hashlist_entries(const char *text, size_t len, u32 *out) {
  size_t n = 0;
  for (size_t at = 0; at < len;) {
    const char *nl = memchr(text + at, '\n', len - at);
    size_t eol = nl ? (size_t)(nl - text) : len;
    for (size_t p = at; p + 40 <= eol; p += 40)
      if (hash160_from_hex(text + p, out + n * 5)) n++;
    at = eol + 1;
  }
  return n;
}
static char *slurp(FILE *in, size_t *len) {
  size_t cap = 1 << 16, n = 0, got;
  char *buf = malloc(cap);
  while ((got = fread(buf + n, 1, cap - n, in)) > 0)
    if ((n += got) == cap) buf = realloc(buf, cap *= 2);
  *len = n;
  return buf;
}
/* -f <file> (load_filter, main.c:71-131): `.blf` -> bloom-only mode; anything else -> hash list, sorted, duplicates
   removed, plus an in-memory bloom of two words per entry.  Errors end the program with the reference's messages. */
static void filter_open(filter_t *f, const char *path) {
  memset(f, 0, sizeof *f);
  if (!path) { fprintf(stderr, "missing filter file\n"); exit(1); }
  FILE *in = fopen(path, "rb");
  if (!in) { fprintf(stderr, "failed to open filter file: %s\n", path); exit(1); }
  const char *dot = strrchr(path, '.');
  if (dot && !strcmp(dot, ".blf")) {
    fclose(in);
    const char *why = blf_read(path, f);
    if (why) { fprintf(stderr, "%s\n", why); exit(1); }
    return;
  }
  const bool stats = getenv("ECLOOP_HIP_STATS") != NULL;
  u64 t0 = us_now();
  size_t len;
  char *text = slurp(in, &len);
  fclose(in);
  u64 t1 = us_now();
  u32 *hs = malloc((len / 40 + 1) * 20);
  size_t n = hashlist_entries(text, len, hs);
  if (!n) { fprintf(stderr, "no hashes in filter file\n"); exit(1); }
  free(text);
  hs = realloc(hs, n * 20);
  u64 t2 = us_now();
  if (stats) fprintf(stderr, "list: %zu entries; read %.1f ms, parse %.1f ms\n", n, (t1 - t0) / 1e3, (t2 - t1) / 1e3);
  /* long lists are sorted, made unique and turned into filter bits on GPU 0 (10^7 entries: 13 s here, qsort + 2 * 10^8
     scattered bit sets); short ones, or no GPU (the hidden CPU-only commands), on the host */
  if (n >= (1u << 16) && n < (1ull << 31) && !getenv("ECLOOP_HIP_LIST_ON_HOST") && ecl_hip_device_count() > 0) {
    ecl_hip *d = NULL;
    u64 kept = 0;
    int rc = ecl_hip_open(&d, 0, ECL_ADDR33, 0);
    if (rc == ECL_OK) rc = ecl_hip_sort_list(d, (uint32_t(*)[5])hs, n, &kept);
    if (rc == ECL_OK) {
      f->list = hs, f->nlist = kept;
      f->nwords = 2 * kept, f->words = calloc(f->nwords, 8);
      rc = ecl_hip_set_bloom(d, f->words, f->nwords);
    }
    if (rc == ECL_OK) rc = ecl_hip_bloom_insert(d, (const uint32_t(*)[5])hs, kept);
    if (rc == ECL_OK) rc = ecl_hip_get_bloom(d, f->words, f->nwords);
    if (rc != ECL_OK) { fprintf(stderr, "[!] preparing the hash list on the GPU failed: %s (%s)\n", ecl_hip_strerror(rc), d ? ecl_hip_last_error(d) : ""); exit(1); }
    ecl_hip_close(d);
    if (stats) fprintf(stderr, "list: sorted, %zu unique, filter bits set on GPU 0 in %.1f ms (context included)\n", (size_t)kept, (us_now() - t2) / 1e3);
    return;
  }
  qsort(hs, n, 20, order160);
  size_t kept = 1;
  for (size_t i = 1; i < n; ++i)
    if (order160(hs + (kept - 1) * 5, hs + i * 5)) memmove(hs + kept++ * 5, hs + i * 5, 20);
  f->list = hs, f->nlist = kept;
  f->nwords = 2 * kept, f->words = calloc(f->nwords, 8);
  for (size_t i = 0; i < kept; ++i) bloom_set(f, hs + i * 5);
  if (stats) fprintf(stderr, "list: sorted, %zu unique, filter bits set on the host in %.1f ms\n", kept, (us_now() - t2) / 1e3);
}
/* second stage of ctx_check_hash (main.c:212-216): the device reports bloom hits, the list decides */
static bool filter_confirms(const filter_t *f, const u32 h[5]) {
  return !f->list || bsearch(h, f->list, f->nlist, 20, order160) != NULL;
}

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
      cap = cnt, buf = realloc(buf, sizeof(ecl_found) * cap); /* dense filter: rerun with a buffer that fits */
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

/* ------------------------------------------------------------------------------------------- mul */
/* host SHA-256 of a passphrase for `-raw` (main.c:505-527): input preparation, not the search path.  Block by block,
   nothing allocated per line.  With the x86 SHA extensions (every EPYC, Xeons since Ice Lake) a block is 64 rounds in 32
   `sha256rnds2`; elsewhere the plain form with the eight working variables renamed instead of moved. */
static const u32 SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static void sha256_block_plain(u32 st[8], const u8 *blk) {
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
#define SHA_ROUND(a, b, c, d, e, f, g, h, i)                                                                      \
  do {                                                                                                            \
    u32 t1 = (h) + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + (((e) & (f)) ^ (~(e) & (g))) + SHA_K[i] + w[i];        \
    u32 t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + (((a) & (b)) ^ ((a) & (c)) ^ ((b) & (c)));                   \
    (d) += t1, (h) = t1 + t2;                                                                                     \
  } while (0)
  u32 w[64];
  for (int i = 0; i < 16; ++i) w[i] = (u32)blk[4 * i] << 24 | (u32)blk[4 * i + 1] << 16 | (u32)blk[4 * i + 2] << 8 | blk[4 * i + 3];
  for (int i = 16; i < 64; ++i)
    w[i] = w[i - 16] + (ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] +
           (ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10));
  u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; i += 8) {
    SHA_ROUND(a, b, c, d, e, f, g, h, i);
    SHA_ROUND(h, a, b, c, d, e, f, g, i + 1);
    SHA_ROUND(g, h, a, b, c, d, e, f, i + 2);
    SHA_ROUND(f, g, h, a, b, c, d, e, i + 3);
    SHA_ROUND(e, f, g, h, a, b, c, d, i + 4);
    SHA_ROUND(d, e, f, g, h, a, b, c, i + 5);
    SHA_ROUND(c, d, e, f, g, h, a, b, i + 6);
    SHA_ROUND(b, c, d, e, f, g, h, a, i + 7);
  }
  st[0] += a, st[1] += b, st[2] += c, st[3] += d, st[4] += e, st[5] += f, st[6] += g, st[7] += h;
#undef SHA_ROUND
#undef ROR
}
#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
__attribute__((target("sha,sse4.1,ssse3"))) static void sha256_block_ni(u32 st[8], const u8 *blk) {
  const __m128i swap = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL); /* big-endian words */
  __m128i t = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&st[0]), 0xB1);       /* c d a b */
  __m128i s1 = _mm_shuffle_epi32(_mm_loadu_si128((const __m128i *)&st[4]), 0x1B);      /* e f g h, reversed */
  __m128i s0 = _mm_alignr_epi8(t, s1, 8);                                              /* the unit's operand order: a b e f */
  s1 = _mm_blend_epi16(s1, t, 0xF0);                                                   /* c d g h */
  const __m128i keep0 = s0, keep1 = s1;
  __m128i m[4];
  for (int i = 0; i < 16; ++i) { /* four rounds per step */
    if (i < 4) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(blk + 16 * i)), swap);
    else {
      __m128i x = _mm_sha256msg1_epu32(m[i & 3], m[(i + 1) & 3]);                /* W[t-16] + s0(W[t-15]) */
      x = _mm_add_epi32(x, _mm_alignr_epi8(m[(i + 3) & 3], m[(i + 2) & 3], 4));  /* + W[t-7] */
      m[i & 3] = _mm_sha256msg2_epu32(x, m[(i + 3) & 3]);                        /* + s1(W[t-2]) */
    }
    __m128i wk = _mm_add_epi32(m[i & 3], _mm_loadu_si128((const __m128i *)&SHA_K[4 * i]));
    s1 = _mm_sha256rnds2_epu32(s1, s0, wk);
    s0 = _mm_sha256rnds2_epu32(s0, s1, _mm_shuffle_epi32(wk, 0x0E));
  }
  s0 = _mm_add_epi32(s0, keep0), s1 = _mm_add_epi32(s1, keep1);
  t = _mm_shuffle_epi32(s0, 0x1B);
  s1 = _mm_shuffle_epi32(s1, 0xB1);
  _mm_storeu_si128((__m128i *)&st[0], _mm_blend_epi16(t, s1, 0xF0));
  _mm_storeu_si128((__m128i *)&st[4], _mm_alignr_epi8(s1, t, 8));
}
static bool cpu_has_sha(void) {
  unsigned a, b, c, d;
  if (getenv("ECLOOP_HIP_NO_SHANI")) return false; /* tests: the plain form on a CPU that has the extension */
  return __get_cpuid_count(7, 0, &a, &b, &c, &d) && (b & (1u << 29)) && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
}
#else
static bool cpu_has_sha(void) { return false; }
#endif
static bool have_sha_ni; /* set once in cmd_mul */
static void sha256_block(u32 st[8], const u8 *blk) {
#if defined(__x86_64__)
  if (have_sha_ni) { sha256_block_ni(st, blk); return; }
#endif
  sha256_block_plain(st, blk);
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

static void mul_flush(run_t *run, int g, u64 (*ks)[4], u32 n) {
  if (!n) return;
  u32 cap = n * 2 + 16, cnt = 0;
  ecl_found *buf = malloc(sizeof(ecl_found) * cap);
  int rc = ecl_hip_mul_batch(run->dev[g], ks, n, buf, cap, &cnt);
  if (rc != ECL_OK) die_ecl(run, g, rc, "mul_batch");
  for (u32 i = 0; i < cnt; ++i) {
    if (!filter_confirms(&run->flt, buf[i].h160)) continue;
    sc pk;
    memcpy(pk.w, ks[buf[i].key_offset], 32);
    report_hit(&run->rep, buf[i].compressed, buf[i].h160, &pk); /* no verify: main.c:469,474 */
  }
  free(buf);
  report_progress(&run->rep, n);
}
/* -raw: lines [at, at + n) of a chunk, hashed on the device; a hit's private key is that line's SHA-256, recomputed here */
static void mul_flush_raw(run_t *run, int g, const u8 *text, size_t text_len, const u64 *lines, u32 n) {
  if (!n) return;
  u32 cap = n * 2 + 16, cnt = 0;
  ecl_found *buf = malloc(sizeof(ecl_found) * cap);
  int rc = ecl_hip_mul_batch_raw(run->dev[g], text, (u32)text_len, lines, n, buf, cap, &cnt);
  if (rc != ECL_OK) die_ecl(run, g, rc, "mul_batch_raw");
  for (u32 i = 0; i < cnt; ++i) {
    if (!filter_confirms(&run->flt, buf[i].h160)) continue;
    const u64 ln = lines[buf[i].key_offset];
    u32 st[8];
    sha256_stream(st, text + (u32)ln, (size_t)(ln >> 32));
    sc pk = {{(u64)st[6] << 32 | st[7], (u64)st[4] << 32 | st[5], (u64)st[2] << 32 | st[3], (u64)st[0] << 32 | st[1]}};
    report_hit(&run->rep, buf[i].compressed, buf[i].h160, &pk);
  }
  free(buf);
  report_progress(&run->rep, n);
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
   A line longer than 1024 characters is read in pieces of 1024, each an entry of its own, as the reference's
   fgets(line, 1025) does (main.c:548-552). */
/* fe_modn_from_hex (lib/ecc.c:81-95,262-265): right to left, characters that are not hex digits skipped, 64 digits at most */
static sc line_to_scalar(const char *p, size_t len) {
  sc k = {{0, 0, 0, 0}};
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
#define MUL_LINE_MAX 1024u /* main.c:18,548: MAX_LINE_SIZE - 1 characters per fgets */
typedef struct {
  const run_t *run;
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
    const size_t stop = nl ? (size_t)(nl - s->buf) : s->end;
    for (size_t q = at; q < stop; q += MUL_LINE_MAX) { /* the reference's fgets(line, 1025): a longer line is read in pieces */
      size_t len = stop - q < MUL_LINE_MAX ? stop - q : MUL_LINE_MAX;
      if (s->buf[q + len - 1] == '\r') len--;
      if (!len) continue;
      if (n >= s->tmp_cap) s->tmp_cap = s->tmp_cap ? s->tmp_cap * 2 : 1 << 16, s->tmp = realloc(s->tmp, s->tmp_cap * 32);
      sc k = line_to_scalar(s->buf + q, len);
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
/* The normal input - every line 64 hex digits and a newline - needs no line search, no scratch and no packing: record r of
   the chunk is at byte 65 r and its scalar goes to slot r of the chunk's array.  A slice that meets anything else (another
   length, a character that is not a hex digit, '\r') reports it and the chunk is parsed the general way. */
#define MUL_RECORD 65u
typedef struct { const char *buf; size_t first, last; u64 (*dst)[4]; bool ok; } fixed_slice; /* records [first, last) */
static void *parse_fixed_worker(void *arg) {
  fixed_slice *s = arg;
  s->ok = false;
#if defined(__x86_64__)
  for (size_t r = s->first; r < s->last; ++r) {
    const char *p = s->buf + r * MUL_RECORD;
    sc k;
    if (p[64] != '\n' || !hex16_ssse3(p, &k.w[3]) || !hex16_ssse3(p + 16, &k.w[2]) || !hex16_ssse3(p + 32, &k.w[1]) || !hex16_ssse3(p + 48, &k.w[0]))
      return NULL;
    k = sc_reduce(k);
    memcpy(s->dst[r], k.w, 32);
  }
  s->ok = true;
#endif
  return NULL;
}

/* -raw: nothing is parsed on the host - a slice's bytes go into the chunk's page-locked text buffer as they are, and its
   non-empty lines are listed (offset | length << 32, '\r' before the newline dropped); the GPU computes the SHA-256s */
typedef struct {
  const char *buf; u8 *text_dst;
  size_t beg, end;
  u64 *tmp; size_t tmp_cap, count;
  u64 *dst;
} raw_slice;
static void *raw_scan_worker(void *arg) {
  raw_slice *s = arg;
  memcpy(s->text_dst + s->beg, s->buf + s->beg, s->end - s->beg);
  size_t n = 0, at = s->beg;
  while (at < s->end) {
    const char *nl = memchr(s->buf + at, '\n', s->end - at);
    const size_t stop = nl ? (size_t)(nl - s->buf) : s->end;
    for (size_t q = at; q < stop; q += MUL_LINE_MAX) { /* pieces of 1024 characters, as the reference's fgets reads them */
      size_t len = stop - q < MUL_LINE_MAX ? stop - q : MUL_LINE_MAX;
      if (s->buf[q + len - 1] == '\r') len--;
      if (!len) continue;
      if (n >= s->tmp_cap) s->tmp_cap = s->tmp_cap ? s->tmp_cap * 2 : 1 << 16, s->tmp = realloc(s->tmp, s->tmp_cap * 8);
      s->tmp[n++] = (u64)q | (u64)len << 32;
    }
    at = stop + 1;
  }
  s->count = n;
  return NULL;
}
static void *raw_pack_worker(void *arg) {
  raw_slice *s = arg;
  memcpy(s->dst, s->tmp, s->count * 8);
  return NULL;
}

/* A pool of parse threads that lives as long as the command: run() executes fn(arg[i]) for i < n - task i on worker
   i mod nth - and returns when all are done.  A 64 MB chunk is ~1 ms of work for 32 threads and they come back to back,
   so the hand-over must cost microseconds: workers wait for the next generation number spinning (a few hundred
   microseconds at most, then they sleep on a condition variable until woken), finish by bumping one atomic counter.
   (Round 2's pool handed tasks out under a mutex and woke everybody through a condition variable: with 32 threads the
   hand-over cost as much as the parsing, with 64 it was slower than with 16.) */
#include <stdatomic.h>
#define MUL_POOL_MAX 128
typedef struct pool_t pool_t;
typedef struct { pool_t *pool; int idx; } pool_seat;
struct pool_t {
  pthread_t th[MUL_POOL_MAX];
  pool_seat seat[MUL_POOL_MAX];
  int nth;
  void *(*fn)(void *);
  char *args;
  size_t stride;
  int n;
  atomic_ullong gen;
  atomic_int done, sleepers;
  atomic_bool quit;
  pthread_mutex_t mu;
  pthread_cond_t cv;
};
static inline void cpu_relax(void) {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}
static void *pool_main(void *arg) {
  pool_seat *me = arg;
  pool_t *p = me->pool;
  unsigned long long seen = 0;
  for (;;) {
    int spins = 0;
    while (atomic_load(&p->gen) == seen && !atomic_load(&p->quit)) {
      if (++spins < 40000) { cpu_relax(); continue; }
      pthread_mutex_lock(&p->mu);
      atomic_fetch_add(&p->sleepers, 1);
      while (atomic_load(&p->gen) == seen && !atomic_load(&p->quit)) pthread_cond_wait(&p->cv, &p->mu);
      atomic_fetch_sub(&p->sleepers, 1);
      pthread_mutex_unlock(&p->mu);
    }
    if (atomic_load(&p->quit)) break;
    seen = atomic_load(&p->gen);
    for (int i = me->idx; i < p->n; i += p->nth) p->fn(p->args + (size_t)i * p->stride);
    atomic_fetch_add(&p->done, 1);
  }
  return NULL;
}
static void pool_wake(pool_t *p) {
  if (atomic_load(&p->sleepers) > 0) {
    pthread_mutex_lock(&p->mu);
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
  }
}
static void pool_init(pool_t *p, int nth) {
  memset(p, 0, sizeof *p);
  pthread_mutex_init(&p->mu, NULL), pthread_cond_init(&p->cv, NULL);
  p->nth = nth;
  for (int i = 0; i < nth; ++i) p->seat[i] = (pool_seat){p, i}, pthread_create(&p->th[i], NULL, pool_main, &p->seat[i]);
}
static void pool_run(pool_t *p, void *(*fn)(void *), void *args, size_t stride, int n) {
  if (n <= 0) return;
  p->fn = fn, p->args = args, p->stride = stride, p->n = n;
  atomic_store(&p->done, 0);
  atomic_fetch_add(&p->gen, 1); /* publishes the fields above */
  pool_wake(p);
  for (int spins = 0; atomic_load(&p->done) < p->nth; ++spins) {
    if (spins < 100000) cpu_relax();
    else sched_yield();
  }
}
static void pool_stop(pool_t *p) {
  atomic_store(&p->quit, true);
  pthread_mutex_lock(&p->mu);
  pthread_cond_broadcast(&p->cv);
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
#define MUL_TEXT_CHUNK ((size_t)64 << 20) /* hex lines and -bin: ~1 M / 2 M scalars per chunk */
#define MUL_RAW_CHUNK ((size_t)32 << 20)  /* -raw: pass phrases are a quarter as long as hex keys - ~2 M lines per chunk */
#define MUL_TEXT_RING 3
typedef struct { char *buf, *own; size_t len; } text_chunk; /* buf = own (a ring buffer) or a slice of the mapped input */
typedef struct {
  text_chunk ring[MUL_TEXT_RING];
  int head, tail, count; /* filled chunks: [tail, head) */
  bool eof, bin;
  size_t chunk; /* bytes per chunk */
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
        size_t end = at + q->chunk < size ? at + q->chunk : size;
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
  char *carry = malloc(q->chunk);
  size_t have = 0;
  for (;;) {
    pthread_mutex_lock(&q->mu);
    while (q->count == MUL_TEXT_RING) pthread_cond_wait(&q->cv, &q->mu);
    text_chunk *c = &q->ring[q->head];
    pthread_mutex_unlock(&q->mu);
    c->buf = c->own;
    memcpy(c->buf, carry, have);
    size_t got;
    while (have < q->chunk && (got = fread(c->buf + have, 1, q->chunk - have, stdin)) > 0) have += got;
    bool eof = have < q->chunk;
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
typedef struct {
  u64 (*ks)[4]; size_t cap, n; bool pinned; /* scalars (hex lines, -bin); n = entries of this chunk in either form */
  /* -raw: the chunk's text and its line table (offset | length << 32) instead - the GPU hashes (ecl_hip_mul_batch_raw) */
  u8 *text; size_t text_cap, text_len; u64 *lines; size_t lines_cap; bool text_pinned, lines_pinned;
} scalar_array;
/* scalar arrays live in page-locked memory so that the GPUs read them by DMA (no staging copy in ecl_hip_mul_batch) */
static void ks_free(const run_t *run, u64 (*ks)[4], bool pinned) {
  (void)run;
  if (pinned) ecl_hip_free_host(ks);
  else free(ks);
}
static void ks_grow(const run_t *run, scalar_array *ar, size_t n) {
  if (n <= ar->cap) return;
  ks_free(run, ar->ks, ar->pinned);
  size_t cap = n + n / 8 + 1024;
  ar->ks = run->parse_only ? NULL : ecl_hip_alloc_host(cap * 32);
  ar->pinned = ar->ks != NULL;
  if (!ar->ks) ar->ks = malloc(cap * 32);
  ar->cap = cap;
}
/* text and line table are page-locked independently (text_pinned / lines_pinned): one of them falling back to pageable
   memory leaves the other - and the bytes the scan workers already copied into it - alone */
static void raw_release(void *p, bool pinned) {
  if (!p) return;
  if (pinned) ecl_hip_free_host(p);
  else free(p);
}
static void raw_grow(const run_t *run, scalar_array *ar, size_t text_bytes, size_t nlines) {
  if (text_bytes > ar->text_cap) { /* only ever called for a chunk whose text has not been copied in yet */
    raw_release(ar->text, ar->text_pinned);
    ar->text_cap = text_bytes + text_bytes / 8 + 4096;
    ar->text = run->parse_only ? NULL : ecl_hip_alloc_host(ar->text_cap);
    ar->text_pinned = ar->text != NULL;
    if (!ar->text) ar->text = malloc(ar->text_cap);
  }
  if (nlines > ar->lines_cap) {
    raw_release(ar->lines, ar->lines_pinned);
    ar->lines_cap = nlines + nlines / 8 + 1024;
    ar->lines = run->parse_only ? NULL : ecl_hip_alloc_host(ar->lines_cap * 8);
    ar->lines_pinned = ar->lines != NULL;
    if (!ar->lines) ar->lines = malloc(ar->lines_cap * 8);
  }
}
/* The arrays of a run are allocated while the devices come up (bring_up starts mul_prealloc beside the device threads):
   page-locking costs 0.3 ms per MB - 45 ms for the four 33 MB arrays of a one-GPU text run, 90 ms with -bin - which the
   first chunks otherwise wait for one after the other. */
static scalar_array mul_ready_arrays[MUL_MAX_ARRAYS];
static int mul_ready_count;
typedef struct { const run_t *run; int narr; } mul_prealloc_arg;
static void *mul_prealloc(void *arg) {
  const mul_prealloc_arg *a = arg;
  const size_t per = a->run->bin ? MUL_TEXT_CHUNK / 32 : MUL_TEXT_CHUNK / MUL_RECORD + 1024;
  const bool raw = a->run->opt.raw && !a->run->bin;
  for (int i = 0; i < a->narr && i < MUL_MAX_ARRAYS; ++i) {
    scalar_array ar;
    memset(&ar, 0, sizeof ar);
    if (raw) raw_grow(a->run, &ar, MUL_RAW_CHUNK, MUL_RAW_CHUNK / 12);
    else ks_grow(a->run, &ar, per);
    mul_ready_arrays[i] = ar, mul_ready_count = i + 1;
  }
  return NULL;
}
typedef struct {
  run_t *run;
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
    if (q->run->parse_only) { /* hidden `parse` command: the scalars as the device would get them, one per line */
      static int quiet = -1; /* ECLOOP_HIP_PARSE_QUIET=1: the front end alone, nothing printed (timing) */
      if (quiet < 0) { const char *e = getenv("ECLOOP_HIP_PARSE_QUIET"); quiet = e && e[0] == '1'; }
      const bool raw = q->run->opt.raw && !q->run->bin;
      for (size_t k = 0; k < ar->n && !quiet; ++k) {
        if (raw) { /* what the device computes from the line table: the line's SHA-256 */
          u32 st[8];
          sha256_stream(st, ar->text + (u32)ar->lines[k], (size_t)(ar->lines[k] >> 32));
          printf("%08x%08x%08x%08x%08x%08x%08x%08x\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
        } else
          printf("%016llx%016llx%016llx%016llx\n", (unsigned long long)ar->ks[k][3], (unsigned long long)ar->ks[k][2],
                 (unsigned long long)ar->ks[k][1], (unsigned long long)ar->ks[k][0]);
      }
    } else if (q->run->opt.raw && !q->run->bin)
      for (size_t at = 0; at < ar->n; at += STEP)
        mul_flush_raw(q->run, a->g, ar->text, ar->text_len, ar->lines + at, (u32)(ar->n - at < STEP ? ar->n - at : STEP));
    else
      for (size_t at = 0; at < ar->n; at += STEP) mul_flush(q->run, a->g, ar->ks + at, (u32)(ar->n - at < STEP ? ar->n - at : STEP));
    pthread_mutex_lock(&q->mu);
    q->idle[q->nidle++] = i;
    pthread_cond_broadcast(&q->cv);
    pthread_mutex_unlock(&q->mu);
  }
  return NULL;
}
/* a text chunk made of fixed records only -> its array, in place; false: not such a chunk (the array's content is then undefined) */
static bool parse_fixed_chunk(const run_t *run, pool_t *pool, int P, const text_chunk *c, scalar_array *ar, u64 *t_grow, u64 *t_parse, u64 *t_mark) {
  if (run->opt.raw || !have_ssse3 || !c->len || c->len % MUL_RECORD) return false;
  const size_t nrec = c->len / MUL_RECORD, per = (nrec + (size_t)P - 1) / (size_t)P;
  ks_grow(run, ar, nrec);
  *t_grow += us_now() - *t_mark, *t_mark = us_now();
  fixed_slice fs[MUL_POOL_MAX];
  int nf = 0;
  for (size_t at = 0; at < nrec; at += per, ++nf) fs[nf] = (fixed_slice){c->buf, at, at + per < nrec ? at + per : nrec, ar->ks, false};
  pool_run(pool, parse_fixed_worker, fs, sizeof fs[0], nf);
  bool all = true;
  for (int i = 0; i < nf; ++i) all = all && fs[i].ok;
  *t_parse += us_now() - *t_mark, *t_mark = us_now();
  if (all) ar->n = nrec;
  return all;
}
static void cmd_mul(run_t *run) {
  report_restart_clock(&run->rep);
  have_sha_ni = cpu_has_sha();
  long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  /* pool size: the main thread and the reader keep a core each (the pool's workers spin).  Hex lines and -bin are bound by
     getting the input's pages mapped and read, which stops scaling at 16 threads on the 2 x 64-core box (text 2^27 lines:
     16 threads 636, 32 threads 378, 64 threads 275 M lines/s); with -raw the GPU hashes, the host only lists the lines */
  const int pool_cap = 16;
  int P = (int)(ncpu < 3 ? 1 : ncpu > pool_cap + 2 ? pool_cap : ncpu - 2);
  { const char *e = getenv("ECLOOP_HIP_PARSE_THREADS"); /* experiments */
    if (e && atoi(e) >= 1 && atoi(e) <= MUL_POOL_MAX) P = atoi(e); }
  text_queue tq;
  memset(&tq, 0, sizeof tq);
  tq.bin = run->bin, tq.chunk = run->opt.raw && !run->bin ? MUL_RAW_CHUNK : MUL_TEXT_CHUNK;
  pthread_mutex_init(&tq.mu, NULL), pthread_cond_init(&tq.cv, NULL);
  for (int i = 0; i < MUL_TEXT_RING; ++i) tq.ring[i].own = tq.ring[i].buf = malloc(tq.chunk);
  scalar_queue sq;
  memset(&sq, 0, sizeof sq);
  sq.run = run, sq.narr = run->ngpus + 2;
  pthread_mutex_init(&sq.mu, NULL), pthread_cond_init(&sq.cv, NULL);
  for (int i = 0; i < sq.narr; ++i) sq.idle[sq.nidle++] = i;
  for (int i = 0; i < mul_ready_count && i < sq.narr; ++i) sq.arr[i] = mul_ready_arrays[i]; /* allocated during bring-up */
  pthread_t reader, devth[MAX_GPUS];
  mul_dev_arg dargs[MAX_GPUS];
  pthread_create(&reader, NULL, mul_reader, &tq);
  for (int g = 0; g < run->ngpus; ++g) dargs[g] = (mul_dev_arg){&sq, g}, pthread_create(&devth[g], NULL, mul_device_worker, &dargs[g]);
  parse_slice sl[MUL_POOL_MAX];
  memset(sl, 0, sizeof sl);
  static raw_slice rs[MUL_POOL_MAX];
  memset(rs, 0, sizeof rs);
  pool_t pool;
  pool_init(&pool, P);
  u64 t_text = 0, t_array = 0, t_parse = 0, t_grow = 0, t_pack = 0, nchunks = 0, nfixed = 0, t_mark; /* us per stage (ECLOOP_HIP_STATS) */
  for (;;) {
    t_mark = us_now();
    pthread_mutex_lock(&tq.mu);
    while (!tq.count && !tq.eof) pthread_cond_wait(&tq.cv, &tq.mu);
    if (!tq.count) { pthread_mutex_unlock(&tq.mu); break; }
    text_chunk *c = &tq.ring[tq.tail];
    pthread_mutex_unlock(&tq.mu);
    t_text += us_now() - t_mark, t_mark = us_now(), nchunks++;
    /* an array for this chunk's scalars */
    pthread_mutex_lock(&sq.mu);
    while (!sq.nidle) pthread_cond_wait(&sq.cv, &sq.mu);
    int ai = sq.idle[--sq.nidle];
    pthread_mutex_unlock(&sq.mu);
    t_array += us_now() - t_mark, t_mark = us_now();
    scalar_array *ar = &sq.arr[ai];
    if (run->bin) { /* the scalars as they are: into the page-locked array, P threads copying */
      ar->n = c->len / 32;
      ks_grow(run, ar, ar->n);
      t_grow += us_now() - t_mark, t_mark = us_now();
      copy_task ct[MUL_POOL_MAX];
      size_t per = (ar->n + (size_t)P - 1) / (size_t)P;
      int nc = 0;
      for (size_t at = 0; at < ar->n; at += per, ++nc)
        ct[nc] = (copy_task){ar->ks + at, c->buf + at * 32, (ar->n - at < per ? ar->n - at : per) * 32};
      pool_run(&pool, copy_worker, ct, sizeof ct[0], nc);
      t_parse += us_now() - t_mark;
    } else if (run->opt.raw) { /* text and line table for the GPU */
      raw_grow(run, ar, c->len, 0);
      t_grow += us_now() - t_mark, t_mark = us_now();
      int ns = 0;
      size_t at = 0, end = c->len;
      for (int i = 0; i < P && at < end; ++i) {
        size_t stop = i == P - 1 ? end : at + (end - at) / (size_t)(P - i);
        if (stop <= at) stop = at + 1;
        while (stop < end && c->buf[stop - 1] != '\n') stop++;
        rs[ns].buf = c->buf, rs[ns].text_dst = ar->text, rs[ns].beg = at, rs[ns].end = stop;
        at = stop, ns++;
      }
      pool_run(&pool, raw_scan_worker, rs, sizeof rs[0], ns);
      t_parse += us_now() - t_mark, t_mark = us_now();
      size_t total = 0;
      for (int i = 0; i < ns; ++i) total += rs[i].count;
      raw_grow(run, ar, c->len, total);
      t_grow += us_now() - t_mark, t_mark = us_now();
      ar->n = total, ar->text_len = c->len;
      size_t off = 0;
      for (int i = 0; i < ns; ++i) rs[i].dst = ar->lines + off, off += rs[i].count;
      pool_run(&pool, raw_pack_worker, rs, sizeof rs[0], ns);
      t_pack += us_now() - t_mark;
    } else if (parse_fixed_chunk(run, &pool, P, c, ar, &t_grow, &t_parse, &t_mark)) {
      nfixed++; /* every line was 64 hex digits + newline: parsed in place */
    } else {
      int ns = 0;
      size_t at = 0, end = c->len;
      for (int i = 0; i < P && at < end; ++i) { /* slices at line boundaries */
        size_t stop = i == P - 1 ? end : at + (end - at) / (size_t)(P - i);
        if (stop <= at) stop = at + 1;
        while (stop < end && c->buf[stop - 1] != '\n') stop++;
        sl[ns].run = run, sl[ns].buf = c->buf, sl[ns].beg = at, sl[ns].end = stop;
        at = stop, ns++;
      }
      pool_run(&pool, parse_worker, sl, sizeof sl[0], ns);
      t_parse += us_now() - t_mark, t_mark = us_now();
      size_t total = 0;
      for (int i = 0; i < ns; ++i) total += sl[i].count;
      ks_grow(run, ar, total);
      t_grow += us_now() - t_mark, t_mark = us_now();
      ar->n = total;
      size_t off = 0;
      for (int i = 0; i < ns; ++i) sl[i].dst = ar->ks + off, off += sl[i].count;
      pool_run(&pool, pack_worker, sl, sizeof sl[0], ns);
      t_pack += us_now() - t_mark;
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
  for (int g = 0; g < run->ngpus; ++g) pthread_join(devth[g], NULL);
  for (int i = 0; i < MUL_TEXT_RING; ++i) free(tq.ring[i].own);
  for (int i = 0; i < sq.narr; ++i) {
    ks_free(run, sq.arr[i].ks, sq.arr[i].pinned);
    raw_release(sq.arr[i].text, sq.arr[i].text_pinned), raw_release(sq.arr[i].lines, sq.arr[i].lines_pinned);
  }
  for (int i = 0; i < MUL_POOL_MAX; ++i) free(sl[i].tmp), free(rs[i].tmp);
  if (!run->parse_only) report_close(&run->rep);
  if (getenv("ECLOOP_HIP_STATS")) /* where the front end's wall time went (the main thread drives one chunk at a time) */
    fprintf(stderr, "mul front end: %llu chunks (%llu of fixed 65-byte records), %d pool threads; ms waiting for text %.1f, waiting for a free array (devices behind) %.1f, "
            "parse / copy %.1f, array growth %.1f, pack %.1f\n", (unsigned long long)nchunks, (unsigned long long)nfixed, P, t_text / 1e3, t_array / 1e3, t_parse / 1e3,
            t_grow / 1e3, t_pack / 1e3);
}

/* ------------------------------------------------------------------------------------------- rnd */
/* 64 random bits: /dev/urandom, or - with -seed - pairs of rand() (utils.c:83-113) */
static u64 random_u64(bool seeded) {
  if (seeded) return (u64)rand() << 32 | (u64)rand();
  static FILE *pool;
  u64 v;
  if (!pool) pool = fopen("/dev/urandom", "rb");
  if (!pool || fread(&v, sizeof v, 1, pool) != 1) { fprintf(stderr, "failed to read /dev/urandom\n"); exit(1); }
  return v;
}
/* uniform value in [lo, hi], both inclusive (fe_rand_range, utils.c:115-153: draw as many bits as the span has, reject) */
static sc random_between(const sc *lo, const sc *hi, bool seeded) {
  sc span, v;
  sc_subraw(&span, hi, lo);
  span = sc_add_u64_raw(span, 1);
  unsigned bits = sc_bitlen(&span);
  do {
    for (int i = 0; i < 4; ++i) {
      unsigned keep = bits > 64u * i ? (bits - 64u * i >= 64 ? 64 : bits - 64u * i) : 0;
      v.w[i] = keep ? random_u64(seeded) & (keep == 64 ? ~0ULL : (1ULL << keep) - 1) : 0;
    }
  } while (bits && sc_cmp(&v, &span) >= 0);
  sc_addraw(&v, &v, lo);
  return v;
}
/* One window of `rnd` (gen_random_range, main.c:580-591): a random value of [A, B] with bits offs .. offs+size-1 cleared
   is the first key, the same value with those bits set the last; both clamped to [A, B]. */
typedef struct { sc first, last; } window_t;
static window_t window_draw(const sc *A, const sc *B, u32 offs, u32 size, bool seeded) {
  window_t w;
  w.first = w.last = random_between(A, B, seeded);
  for (u32 b = offs; b < offs + size; ++b) {
    const u64 bit = 1ULL << (b & 63);
    w.first.w[b >> 6] &= ~bit, w.last.w[b >> 6] |= bit;
  }
  if (sc_cmp(&w.first, A) < 0) w.first = *A;
  if (sc_cmp(&w.last, B) > 0) w.last = *B;
  return w;
}
/* a window bound as the reference prints it (print_range_mask, main.c:593-617): 64 hex digits in four groups, the digits
   that overlap the window's bit field in yellow on a terminal (digit i from the left holds bits 255-4i-3 .. 255-4i) */
static void window_print_bound(const sc *v, u32 offs, u32 size, bool colour) {
  char digits[65];
  hex_of_scalar(digits, v);
  const int top = 255 - (int)offs, bottom = top - (int)size + 1; /* in the reference's left-to-right bit numbering */
  for (int i = 0; i < 64; ++i) {
    const bool lit = colour && 4 * i + 3 >= bottom && 4 * i <= top;
    printf("%s%s%c%s", i && i % 16 == 0 ? " " : "", lit ? "\033[33m" : "", digits[i], lit ? "\033[0m" : "");
  }
  printf("\n");
}
/* cmd_rnd (main.c:619-662): window after window, each scanned like `add -r first:last -d offs:size` with full-size jobs;
   stops after the first window if that window is the whole range, otherwise runs until interrupted
   (ECLOOP_HIP_RND_WINDOWS=N, for tests and timing runs, stops after N windows). */
static void cmd_rnd(run_t *run) {
  report_t *rep = &run->rep; /* (the window was clamped to 255 bits where it was parsed: window_from_option) */
  printf("[RANDOM MODE] offs: %d ~ bits: %d\n\n", run->ord_offs, run->ord_size);
  report_restart_clock(rep);
  const sc A = run->range_s, B = run->range_e;
  const char *limit_text = getenv("ECLOOP_HIP_RND_WINDOWS");
  const u64 limit = limit_text ? strtoull(limit_text, NULL, 10) : 0;
  for (u64 done = 0;;) {
    const u64 found0 = rep->found, checked0 = rep->checked, t0 = ms_now();
    const window_t w = window_draw(&A, &B, run->ord_offs, run->ord_size, run->seeded);
    window_print_bound(&w.first, run->ord_offs, run->ord_size, run->colour);
    window_print_bound(&w.last, run->ord_offs, run->ord_size, run->colour);
    if (sc_cmp(&w.first, &w.last) < 0) scan_range(run, w.first, w.last, true);
    const u64 took = ms_now() - t0;
    erase_status_line();
    printf("%'llu / %'llu ~ %.1fs\n\n", (unsigned long long)(rep->found - found0), (unsigned long long)(rep->checked - checked0),
           (took ? took : 1) / 1000.0);
    const bool whole_range = !sc_cmp(&w.first, &A) && !sc_cmp(&w.last, &B);
    if (whole_range || (limit && ++done >= limit)) break;
  }
  print_device_stats(run);
  report_close(rep);
}

/* ------------------------------------------------------------------------------------------- blf-gen / blf-check */
/* hash160 lines of a text stream, a batch at a time (lines are read the way filter_open reads a list: 40-character
   pieces, clean hex only) */
/* hash lines of blf-gen / blf-check on stdin, a block at a time.  The reference reads with fgets into a 41-byte buffer
   (utils.c:451-466): a line is consumed in pieces of 40 characters and every full piece of 40 hex digits is an entry -
   hashlist_entries() on the block, which is cut at its last newline (the rest is carried into the next block). */
#define HASH_BLOCK ((size_t)64 << 20)
#define HASH_BLOCK_ENTRIES (HASH_BLOCK / 40 + 1)
typedef struct { FILE *in; char *buf; size_t have; bool eof; } hash_lines_t;
static size_t hash_lines_next(hash_lines_t *s, u32 (*out)[5]) { /* out: room for HASH_BLOCK_ENTRIES; 0 = end of input */
  if (!s->buf) s->buf = malloc(HASH_BLOCK);
  for (;;) {
    if (!s->eof) {
      size_t got = fread(s->buf + s->have, 1, HASH_BLOCK - s->have, s->in);
      s->have += got;
      if (s->have < HASH_BLOCK) s->eof = true;
    }
    if (!s->have) return 0;
    size_t end = s->have;
    if (!s->eof) {
      while (end > 0 && s->buf[end - 1] != '\n') end--;
      if (end == 0) end = s->have / 40 * 40; /* one line longer than the block: whole pieces now, the rest stays */
    }
    const size_t n = hashlist_entries(s->buf, end, (u32 *)out);
    memmove(s->buf, s->buf + end, s->have - end);
    s->have -= end;
    if (n || (s->eof && !s->have)) return n;
  }
}
/* blf-gen -n <count> -o <file> < hashes (utils.c:409-475): a filter sized for n entries at a false-positive rate of 1e-9,
   created or - if the file exists with that size - updated; prints how many of the hashes were new.  Filters for 2^16
   entries and more are filled on the GPU when one is visible (`-host` keeps it on the CPU): the same 20 bits per hash by
   atomic ORs, and the same "new items" count as the sequential loop gives in input order (ecl_hip_bloom_insert_count);
   the file written is byte-identical either way. */
static void cmd_blf_gen(const opts_t *o, const char *prog) {
  const u64 n = opt_number(o->count, 0);
  if (!n || !o->outfile) {
    fprintf(stderr, "Usage: %s blf-gen -n <count> -o <file>   (hex hash160 list on stdin)\n", prog);
    exit(1);
  }
  /* utils.c:421-427, the arithmetic kept operation for operation: its double rounding decides the file size */
  const u64 one_in = 1000000000ull;
  const double p = 1.0 / (double)one_in;
  const u64 m_bits = (u64)(n * log(p) / log(1.0 / pow(2.0, log(2.0))));
  filter_t f = {NULL, (m_bits + 63) / 64, NULL, 0};
  if (access(o->outfile, F_OK) == 0) {
    printf("file %s already exists; loading...\n", o->outfile);
    filter_t old = {0};
    if (blf_read(o->outfile, &old)) { fprintf(stderr, "[!] failed to load bloom filter: delete it or choose a different file\n"); exit(1); }
    if (old.nwords != f.nwords) { fprintf(stderr, "[!] bloom filter size mismatch (%'llu != %'llu)\n", (unsigned long long)old.nwords, (unsigned long long)f.nwords); exit(1); }
    f.words = old.words;
    printf("updating bloom filter...\n");
  } else {
    printf("creating bloom filter...\n");
    f.words = calloc(f.nwords, 8);
  }
  printf("bloom filter params: n = %'llu | p = 1:%'llu | m = %'llu (%'.1f MB)\n", (unsigned long long)n, (unsigned long long)one_in,
         (unsigned long long)m_bits, (double)m_bits / 8 / 1024 / 1024);
  hash_lines_t lines = {stdin, NULL, 0, false};
  u64 fresh = 0;
  ecl_hip *dev = NULL;
  if (n >= (1u << 16) && !o->host_only && ecl_hip_device_count() > 0) {
    int rc = ecl_hip_open(&dev, 0, ECL_ADDR33, 0);
    if (rc == ECL_OK) rc = ecl_hip_set_bloom(dev, f.words, f.nwords);
    if (rc != ECL_OK) { fprintf(stderr, "[!] GPU set-up failed: %s (%s)\n", ecl_hip_strerror(rc), dev ? ecl_hip_last_error(dev) : ""); exit(1); }
    printf("inserting on GPU 0\n");
  }
  u32 (*hs)[5] = malloc(HASH_BLOCK_ENTRIES * 20);
  for (size_t got; (got = hash_lines_next(&lines, hs)) > 0;) {
    if (dev) {
      u64 added = 0;
      int rc = ecl_hip_bloom_insert_count(dev, (const uint32_t(*)[5])hs, got, &added);
      if (rc != ECL_OK) { fprintf(stderr, "[!] GPU insert failed: %s (%s)\n", ecl_hip_strerror(rc), ecl_hip_last_error(dev)); exit(1); }
      fresh += added;
    } else
      for (size_t i = 0; i < got; ++i)
        if (!bloom_test(&f, hs[i])) bloom_set(&f, hs[i]), fresh++;
  }
  free(hs);
  if (dev) {
    int rc = ecl_hip_get_bloom(dev, f.words, f.nwords);
    if (rc != ECL_OK) { fprintf(stderr, "[!] reading the filter back failed: %s\n", ecl_hip_strerror(rc)); exit(1); }
    ecl_hip_close(dev);
  }
  printf("added %'llu new items; saving to %s\n", (unsigned long long)fresh, o->outfile);
  if (!blf_write(o->outfile, &f)) { fprintf(stderr, "[!] failed to save bloom filter\n"); exit(1); }
}
/* blf-check -f <file> [hash ...] (utils.c:495-529): the hashes named on the command line, or else those on stdin */
static void cmd_blf_check(const opts_t *o, int argc, const char **argv) {
  filter_t f = {0};
  if (!o->filter || blf_read(o->filter, &f)) { fprintf(stderr, "Usage: %s blf-check -f <file> <hash> [hash...]\n", argv[0]); exit(1); }
  u32 h[5];
  int named = 0;
  for (int i = 2; i < argc; ++i)
    if (strlen(argv[i]) == 40 && hash160_from_hex(argv[i], h)) printf("%s %s\n", argv[i], bloom_test(&f, h) ? "FOUND" : "NOT FOUND"), named++;
  if (named) return;
  char text[128];
  while (fgets(text, sizeof text, stdin)) {
    text[strcspn(text, "\r\n")] = 0;
    if (strlen(text) == 40 && hash160_from_hex(text, h)) printf("%s %s\n", text, bloom_test(&f, h) ? "FOUND" : "NOT FOUND");
  }
}

/* ------------------------------------------------------------------------------------------- range and window arguments */
/* -r A:B (arg_search_range, main.c:666-701): hex, A > 0x800, B <= p (p, not n), A < B; default 0x800 : p */
static void range_from_option(const char *text, sc *first, sc *last) {
  const sc floor = sc_u64(GROUP_INV_SIZE);
  *first = floor, *last = SC_P;
  if (!text) return;
  const char *colon = strchr(text, ':');
  if (!colon) { fprintf(stderr, "invalid search range, use format: -r 8000:ffff\n"); exit(1); }
  char *left = strndup(text, (size_t)(colon - text));
  *first = sc_from_hex(left), *last = sc_from_hex(colon + 1);
  free(left);
  const char *why = sc_cmp(first, &floor) <= 0 ? "start <= 0x800" : sc_cmp(last, &SC_P) > 0 ? "end > FE_P" : sc_cmp(first, last) >= 0 ? "start >= end" : NULL;
  if (why) { fprintf(stderr, "invalid search range, %s\n", why); exit(1); }
}
/* -d offs:size (load_offs_size, main.c:703-746).  size: 20..64, default min(32, max(20, bits of B)); offs: at most 255 and
   at most max(1, max(20, bits of B) - default size) - so that a window stays inside the range; `rnd` without -d draws
   the offset at random.  `rnd` also keeps offs + size within 255 bits (main.c:620) - here, before anything is derived from
   the offset (the stride 2^offs, the device contexts), as the reference does before ctx_precompute_gpoints (main.c:624). */
static void window_clamp_rnd(run_t *run) {
  if (run->cmd == CMD_RND && run->ord_offs + run->ord_size > 255) run->ord_offs = 255 - run->ord_size;
}
static void window_from_option(run_t *run) {
  const u32 lo_size = 20, hi_size = 64;
  const u32 span_bits = sc_bitlen(&run->range_e) > lo_size ? sc_bitlen(&run->range_e) : lo_size;
  const u32 usual = span_bits < 32 ? span_bits : 32;
  const u32 offs_cap = span_bits - usual > 1 ? span_bits - usual : 1;
  const char *text = run->opt.window;
  run->ord_offs = 0, run->ord_size = usual;
  if (!text) {
    if (run->cmd == CMD_RND) run->ord_offs = (u32)(random_u64(run->seeded) % offs_cap);
    window_clamp_rnd(run);
    return;
  }
  const char *colon = strchr(text, ':');
  if (!colon) { fprintf(stderr, "invalid offset:size format, use format: -d 128:32\n"); exit(1); }
  const u32 offs = (u32)atoi(text), size = (u32)atoi(colon + 1);
  if (offs > 255) { fprintf(stderr, "invalid offset, max is 255\n"); exit(1); }
  if (size < lo_size || size > hi_size) { fprintf(stderr, "invalid size, min is %d and max is %d\n", lo_size, hi_size); exit(1); }
  run->ord_offs = offs < offs_cap ? offs : offs_cap, run->ord_size = size;
  window_clamp_rnd(run);
}
static void usage(const char *prog) { /* the reference's help text (main.c:750-772) with this program's -t and extras */
  static const char *const TEXT[] = {
      "\nCompute commands:\n",
      "  add             - search in given range with batch addition\n",
      "  mul             - search hex encoded private keys (from stdin)\n",
      "  rnd             - search random range of bits in given range\n",
      "\nCompute options:\n",
      "  -f <file>       - filter file to search (list of hashes or bloom fitler)\n",
      "  -o <file>       - output file to write found keys (default: stdout)\n",
      "  -t <gpus>       - number of GPUs to use (default: all)\n",
      "  -a <addr_type>  - address type to search: c - addr33, u - addr65 (default: c)\n",
      "  -r <range>      - search range in hex format (example: 8000:ffff, default all)\n",
      "  -d <offs:size>  - bit offset and size for search (example: 128:32, default: 0:32)\n",
      "  -q              - quiet mode (no output to stdout; -o required)\n",
      "  -endo           - use endomorphism (default: false)\n",
      "  -raw            - mul: the private key is the SHA-256 of the line (hashed on the GPU)\n",
      "  -bin            - mul: stdin carries 32-byte little-endian scalars instead of hex lines\n",
      "\nOther commands:\n",
      "  blf-gen         - create bloom filter from list of hex-encoded hash160\n",
      "  blf-check       - check bloom filter for given hex-encoded hash160\n",
      "  bench           - run benchmark of the device paths (add per address type / endo, mul)\n",
      "  bench-gtable    - run benchmark of ecc multiplication (with different table size)\n",
      "  mult-verify     - check the window-table multiplication against double-and-add (2 .. 16001)\n\n"};
  printf("Usage: %s <cmd> [-t <gpus>] [-f <file>] [-a <addr_type>] [-r <range>]\nv%s ~ MI355X build of the ecloop command set\n", prog, VERSION);
  for (size_t i = 0; i < sizeof TEXT / sizeof TEXT[0]; ++i) fputs(TEXT[i], stdout);
}

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
  for (u32 w = 8; w <= 24; w += 2) {
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
    const double slots = (double)(nwin - 1) * (double)((1u << w) - 1) + (double)((1u << (256 - w * (nwin - 1))) - 1);
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
/* one context: open (self-test once per process), filter upload from the one pinned host copy, optional list, walk
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
    const u32 n = (u32)(run->bin ? MUL_TEXT_CHUNK / 32 : run->opt.raw ? MUL_RAW_CHUNK / 12 : MUL_TEXT_CHUNK / MUL_RECORD + 1024);
    rc = ecl_hip_reserve_mul(*h, n, n * 2 + 16);
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
  const bool pinned = run->flt.nwords >= (8u << 20) && ecl_hip_pin_host(run->flt.words, run->flt.nwords * 8) == ECL_OK;
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
  if (pinned) ecl_hip_unpin_host(run->flt.words);
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
#endif
  static run_t run;
  opts_t *o = &run.opt;
  opts_parse(o, argc, argv);
  const char *verb = argc > 1 ? argv[1] : "";
  /* commands that need no search context */
  if (!strcmp(verb, "blf-gen")) return cmd_blf_gen(o, argv[0]), 0;
  if (!strcmp(verb, "blf-check")) return cmd_blf_check(o, argc, argv), 0;
  if (!strcmp(verb, "bench")) return run_bench(o);
  if (!strcmp(verb, "bench-gtable")) return run_bench_gtable();
  if (!strcmp(verb, "mult-verify")) return run_mult_verify();
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
  if (o->seed) { /* a seeded run draws from rand() (the reference free()s an argv pointer here and aborts, main.c:800-805) */
    u32 s = 5381;
    for (const char *c = o->seed; *c; ++c) s = s * 33 + (u8)*c;
    run.seeded = true, srand(s);
  }
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
