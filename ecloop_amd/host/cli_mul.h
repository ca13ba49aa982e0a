/* cli_mul.h - `mul`: text front end (reader, parse pool, -raw line tables), batches fanned out to the device contexts.
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
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

/* hit records of one device call: the buffer holds MUL_HITS_CAP of them; a call that reports more (ECL_E_OVERFLOW: a dense filter) has
   kept the rest on the device, fetched here */
#define MUL_HITS_CAP (1u << 16)
static ecl_found *mul_collect(run_t *run, int g, int rc, ecl_found *buf, u32 cap, u32 cnt, const char *what) {
  if (rc == ECL_E_OVERFLOW) {
    u32 got = 0;
    buf = realloc(buf, sizeof(ecl_found) * cnt);
    rc = ecl_hip_fetch_found(run->dev[g], cap, buf + cap, cnt - cap, &got);
    if (rc == ECL_OK && got != cnt - cap) rc = ECL_E_OVERFLOW; /* more than the device keeps (2^20): the batch sizes below never get there */
  }
  if (rc != ECL_OK) die_ecl(run, g, rc, what);
  return buf;
}
static void mul_flush(run_t *run, int g, u64 (*ks)[4], u32 n) {
  if (!n) return;
  u32 cap = n * 2 + 16 < MUL_HITS_CAP ? n * 2 + 16 : MUL_HITS_CAP, cnt = 0;
  ecl_found *buf = malloc(sizeof(ecl_found) * cap);
  int rc = ecl_hip_mul_batch(run->dev[g], ks, n, buf, cap, &cnt);
  buf = mul_collect(run, g, rc, buf, cap, cnt, "mul_batch");
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
  u32 cap = n * 2 + 16 < MUL_HITS_CAP ? n * 2 + 16 : MUL_HITS_CAP, cnt = 0;
  ecl_found *buf = malloc(sizeof(ecl_found) * cap);
  int rc = ecl_hip_mul_batch_raw(run->dev[g], text, (u32)text_len, lines, n, buf, cap, &cnt);
  buf = mul_collect(run, g, rc, buf, cap, cnt, "mul_batch_raw");
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
  if (have_avx512 && len) {
    /* "0x" in front of at most 64 digits changes nothing: read right to left, the x is skipped and the 0 is a leading zero or the 65th digit */
    const bool prefixed = len >= 3 && len <= 66 && p[0] == '0' && (p[1] | 0x20) == 'x';
    const char *q = prefixed ? p + 2 : p;
    const size_t n = prefixed ? len - 2 : len;
    if (n <= 64 && hexline_avx512(q, n, k.w)) return n == 64 ? sc_reduce(k) : k; /* (fewer than 64 digits: below n) */
  }
  k = (sc){{0, 0, 0, 0}};
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

/* A pool of parse threads that lives as long as the command: run() executes fn(arg[i]) for i < n - each worker takes
   the next task from a shared counter - and returns when all are done.  A 64 MB chunk is ~1 ms of work for 32 threads and they come back to back,
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
  atomic_int done, sleepers, next;
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
    /* tasks are taken one at a time from a shared counter: a worker that shares its core with something else for a while (the device
       threads, the runtime's helpers) takes fewer slices instead of holding the whole batch up with a fixed share */
    for (int i; (i = atomic_fetch_add(&p->next, 1)) < p->n;) p->fn(p->args + (size_t)i * p->stride);
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
/* ECLOOP_HIP_PARSE_NODE=N (experiments): the pool's threads run on the CPUs of NUMA node N only (/sys/devices/system/node/nodeN/cpulist) */
static bool node_cpus(int node, cpu_set_t *set) {
  char path[96], text[4096];
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return false;
  const bool got = fgets(text, sizeof text, f) != NULL;
  fclose(f);
  if (!got) return false;
  CPU_ZERO(set);
  int any = 0;
  for (char *tok = strtok(text, ",\n"); tok; tok = strtok(NULL, ",\n")) {
    int a, b;
    const int k = sscanf(tok, "%d-%d", &a, &b);
    if (k < 1) continue;
    if (k == 1) b = a;
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET(c, set), any = 1;
  }
  return any;
}
static void pool_init(pool_t *p, int nth) {
  memset(p, 0, sizeof *p);
  pthread_mutex_init(&p->mu, NULL), pthread_cond_init(&p->cv, NULL);
  p->nth = nth;
  cpu_set_t set;
  const char *node = getenv("ECLOOP_HIP_PARSE_NODE");
  const bool bind = node && node_cpus(atoi(node), &set);
  for (int i = 0; i < nth; ++i) {
    p->seat[i] = (pool_seat){p, i}, pthread_create(&p->th[i], NULL, pool_main, &p->seat[i]);
    if (bind) pthread_setaffinity_np(p->th[i], sizeof set, &set);
  }
}
static void pool_run(pool_t *p, void *(*fn)(void *), void *args, size_t stride, int n) {
  if (n <= 0) return;
  p->fn = fn, p->args = args, p->stride = stride, p->n = n;
  atomic_store(&p->done, 0);
  atomic_store(&p->next, 0);
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
#define MUL_RAW_CHUNK ((size_t)32 << 20)  /* -raw from a pipe: pass phrases are a quarter as long as hex keys - ~2 M lines per chunk */
#define MUL_RAW_CHUNK_MAX ((size_t)256 << 20) /* -raw from a regular file: an eighth of the file up to this (~15 M lines = one device call) */
/* bytes per -raw chunk = per ecl_hip_mul_batch_raw call.  The device reaches its rate on calls of 2^24 scalars (a call is pipelined inside
   the library, and its first and last pieces run on a part-filled chip), so a file large enough is read in chunks of 256 MB; a small
   file keeps ~8 chunks so that its contexts still take turns; a pipe keeps 32 MB (what is read is what can be worked on) */
static size_t mul_raw_chunk(void) {
  static size_t chunk;
  if (!chunk) {
    struct stat st;
    const off_t pos = lseek(0, 0, SEEK_CUR);
    chunk = MUL_RAW_CHUNK;
    if (pos >= 0 && fstat(0, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > pos) {
      const size_t want = (size_t)(st.st_size - pos) / 8 & ~(((size_t)1 << 20) - 1);
      chunk = want < MUL_RAW_CHUNK ? MUL_RAW_CHUNK : want > MUL_RAW_CHUNK_MAX ? MUL_RAW_CHUNK_MAX : want;
    }
    const char *e = getenv("ECLOOP_HIP_MUL_RAW_CHUNK"); /* experiments: bytes */
    if (e && atol(e) >= 4096) chunk = (size_t)atol(e);
  }
  return chunk;
}
/* lines a regular file of -raw input holds, estimated from the line lengths of its first 256 KB (0: not a regular file); decides the
   window width during bring-up as st_size / 65 does for hex lines */
static size_t mul_raw_lines_estimate(void) {
  struct stat st;
  const off_t pos = lseek(0, 0, SEEK_CUR);
  if (pos < 0 || fstat(0, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= pos) return 0;
  const size_t look = 256 << 10; /* (every context's bring-up thread asks: a buffer of its own) */
  char *head = malloc(look);
  const ssize_t got = head ? pread(0, head, look, pos) : 0;
  size_t nl = 0;
  for (ssize_t i = 0; i < got; ++i) nl += head[i] == '\n';
  free(head);
  if (got <= 0) return 0;
  if (!nl) nl = 1;
  return (size_t)((double)(st.st_size - pos) * (double)nl / (double)got);
}
#define MUL_TEXT_RING 3
typedef struct { char *buf, *own; size_t len; } text_chunk; /* buf = own (a ring buffer) or a slice of the mapped input */
typedef struct {
  text_chunk ring[MUL_TEXT_RING];
  int head, tail, count; /* filled chunks: [tail, head) */
  bool eof, bin;
  size_t chunk; /* bytes per chunk */
  size_t limit; /* a seekable input only: stop at the first line end at or after this many bytes (0: read to the end) - cmd_mul then looks
                   again whether the batch path can go on from there */
  bool input_done; /* set by the reader: the input is exhausted (not just the stretch it was limited to) */
  char *map; size_t map_size; /* the mapping the chunks of this stretch point into; cmd_mul unmaps it when they are parsed */
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
      q->map = map, q->map_size = size;
      /* (populating each chunk's pages from this thread ahead of the parse threads - madvise MADV_POPULATE_READ - was measured in round 6 and
         is slower: 2^30 hex lines through this reader 565-595 against 657-671 M lines/s, -raw 974-989 against 1137-1142: the call holds the
         address-space lock the device threads' launches and copies need, like the munmap that cmd_mul now puts off) */
      size_t stop = size; /* where this stretch ends: the first line end at or after pos + limit (records of -bin: a multiple of 32) */
      if (q->limit && (size_t)pos + q->limit < size) {
        stop = (size_t)pos + q->limit;
        if (q->bin) stop = (size_t)pos + (q->limit + 31) / 32 * 32;
        else
          while (stop < size && map[stop - 1] != '\n') stop++;
        if (stop > size) stop = size;
      }
      for (size_t at = (size_t)pos; at < stop;) {
        size_t end = at + q->chunk < stop ? at + q->chunk : stop;
        if (end < stop) {
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
      if (lseek(0, (off_t)stop, SEEK_SET) < 0) { fprintf(stderr, "[!] lseek on the input failed\n"); exit(1); }
      pthread_mutex_lock(&q->mu);
      q->eof = true, q->input_done = stop >= size;
      pthread_cond_broadcast(&q->cv);
      pthread_mutex_unlock(&q->mu);
      return NULL; /* (the mapping stays until the chunks of this stretch are parsed: cmd_mul unmaps it) */
    }
  }
  q->input_done = true; /* whatever follows reads to the end of the input */
#ifdef F_SETPIPE_SZ
  { /* a pipe on stdin (`cat keys | ecloop-hip mul`): 1 MB instead of the default 64 KB in flight - fewer wake-ups of the writer.  More does not
       pay where it matters (tools/pipeprobe_r06.sh on the GPU box, 2^28 hex lines: 91-101 M lines/s at 1 MB, 98-101 at 16 MB, 100-103 at
       64 MB; -raw 330-410 M pass phrases/s at every size: `cat` is at its rate), though it did on the 8-core build container (2.1 / 2.4 /
       2.6 GB/s).  Above /proc/sys/fs/pipe-max-size the call needs CAP_SYS_RESOURCE. */
    const char *e = getenv("ECLOOP_HIP_PIPE_SZ"); /* experiments */
    long want = e && atol(e) >= 4096 ? atol(e) : 1l << 20;
    while (want >= (1l << 20) && fcntl(0, F_SETPIPE_SZ, (int)want) < 0) want >>= 2;
  }
#endif
  char *carry = malloc(q->chunk);
  size_t have = 0;
  for (;;) {
    pthread_mutex_lock(&q->mu);
    while (q->count == MUL_TEXT_RING) pthread_cond_wait(&q->cv, &q->mu);
    text_chunk *c = &q->ring[q->head];
    pthread_mutex_unlock(&q->mu);
    if (!c->own && !(c->own = malloc(q->chunk))) { fprintf(stderr, "out of memory for the text buffers\n"); exit(1); }
    c->buf = c->own;
    memcpy(c->buf, carry, have);
    for (ssize_t k; have < q->chunk && (k = read(0, c->buf + have, q->chunk - have)) != 0;) { /* (read, not fread: no second copy through stdio) */
      if (k < 0) { if (errno == EINTR) continue; break; }
      have += (size_t)k;
    }

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
static size_t mul_largest_batch(const run_t *run, u32 *window);
static size_t mul_batch_records(size_t total);
static size_t mul_fixed_file_records(const run_t *run, off_t *pos, size_t *rec);
static scalar_array mul_ready_arrays[MUL_MAX_ARRAYS];
static int mul_ready_count;
typedef struct { const run_t *run; int narr; } mul_prealloc_arg;
static void *mul_prealloc(void *arg) {
  const mul_prealloc_arg *a = arg;
  const size_t per = mul_largest_batch(a->run, NULL);
  const bool raw = a->run->opt.raw && !a->run->bin;
  /* a file of fewer batches than arrays gets only as many as it has batches (page-locking 512 MB takes 0.15 s) */
  off_t pos;
  size_t rec;
  const size_t total = mul_fixed_file_records(a->run, &pos, &rec), batch = mul_batch_records(total);
  int want = a->narr;
  if (total && (total + batch - 1) / batch < (size_t)want) want = (int)((total + batch - 1) / batch);
  for (int i = 0; i < want && i < MUL_MAX_ARRAYS; ++i) {
    scalar_array ar;
    memset(&ar, 0, sizeof ar);
    if (raw) raw_grow(a->run, &ar, mul_raw_chunk(), mul_raw_chunk() / 12);
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
typedef struct { scalar_queue *q; int g; u64 busy_us, wait_us, calls, scalars; } mul_dev_arg; /* (the last four: ECLOOP_HIP_STATS) */
static void *mul_device_worker(void *arg) {
  mul_dev_arg *a = arg;
  scalar_queue *q = a->q;
  const size_t STEP = 1u << 26, WHOLE = (size_t)1 << 26; /* lines per -raw call (the ABI's limit); scalars per call otherwise: the array as it is */
  for (;;) {
    u64 t_mark = us_now();
    pthread_mutex_lock(&q->mu);
    while (!q->nready && !q->done) pthread_cond_wait(&q->cv, &q->mu);
    if (!q->nready) { pthread_mutex_unlock(&q->mu); break; }
    int i = q->ready[0];
    memmove(q->ready, q->ready + 1, sizeof(int) * --q->nready);
    pthread_mutex_unlock(&q->mu);
    scalar_array *ar = &q->arr[i];
    a->wait_us += us_now() - t_mark, t_mark = us_now(), a->calls++, a->scalars += ar->n;
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
      for (size_t at = 0; at < ar->n; at += WHOLE) mul_flush(q->run, a->g, ar->ks + at, (u32)(ar->n - at < WHOLE ? ar->n - at : WHOLE));
    a->busy_us += us_now() - t_mark;
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
/* ---- the usual bulk input, straight from the file: a regular file on stdin whose lines are all 64 hex digits + newline ----
   Record r of the file is at byte 65 r, so nothing has to be searched, cut or packed: the input is taken in batches of 2^24 records
   (1.09 GB of text -> one page-locked array of 512 MB = one ecl_hip_mul_batch call, the size at which the device reaches its rate:
   2^20-scalar calls 0.73, 2^22 0.95, 2^24 1.27 G scalars/s), a batch in slices of 32768 records that the pool's threads take in turn:
   pread() of the slice into the thread's own 2 MB buffer (a copy out of the page cache without page faults - mapping the file instead
   made every thread fault its pages in one by one behind the process's one address-space lock: 0.64 G lines/s at 16 threads, less with
   more; ECLOOP_HIP_MUL_READ=mmap keeps that form for comparison), 16 characters at a time into slot r of the batch's array.  A record
   that is anything else (another length, '\r', a character that is no hex digit) ends this path at the batch before it: the rest of
   the input goes through the general reader below, which starts at the file offset this path leaves. */
#define MUL_BATCH_LOG2 24
#define MUL_SLICE_RECORDS_MAX 32768u
typedef struct { int fd; const char *map; off_t base; u64 (*dst)[4]; size_t rec; atomic_bool bad; } fixed_batch; /* rec: 65 (LF lines), 66 (CRLF), 67 / 68 (the same behind "0x"), 32 (-bin) */
static atomic_ullong fixed_read_us, fixed_decode_us; /* summed over the pool's threads (ECLOOP_HIP_STATS) */
typedef struct { fixed_batch *b; size_t first, last; } fixed_file_slice;
static bool pread_all(int fd, char *dst, size_t bytes, off_t at) {
  for (size_t have = 0; have < bytes;) {
    ssize_t got = pread(fd, dst + have, bytes - have, at + (off_t)have);
    if (got <= 0) return false; /* the file shrank under us */
    have += (size_t)got;
  }
  return true;
}
static void *fixed_file_worker(void *arg) {
#if defined(__x86_64__)
  fixed_file_slice *s = arg;
  fixed_batch *b = s->b;
  if (atomic_load(&b->bad)) return NULL;
  const size_t rec = b->rec, bytes = (s->last - s->first) * rec;
  const off_t at = b->base + (off_t)(s->first * rec);
  const u64 t0 = us_now();
  if (rec == 32) { /* -bin: the scalars as they are, straight into their slots */
    if (b->map) memcpy(b->dst[s->first], b->map + at, bytes);
    else if (!pread_all(b->fd, (char *)b->dst[s->first], bytes, at)) atomic_store(&b->bad, true);
    atomic_fetch_add(&fixed_read_us, us_now() - t0);
    return NULL;
  }
  const char *src;
  if (b->map) src = b->map + at;
  else {
    static __thread char *mine; /* this thread's text buffer, for the life of the command */
    if (!mine && !(mine = malloc((size_t)MUL_SLICE_RECORDS_MAX * 68))) { atomic_store(&b->bad, true); return NULL; }
    if (!pread_all(b->fd, mine, bytes, at)) { atomic_store(&b->bad, true); return NULL; }
    src = mine;
  }
  const bool wide = have_avx2, widest = have_avx512, crlf = rec == 66 || rec == 68, prefixed = rec >= 67;
  const u64 t1 = us_now();
  for (size_t r = 0; r < s->last - s->first; ++r) {
    const char *p = src + r * rec;
    u64 *dst = b->dst[s->first + r];
    sc k;
    bool ended = true;
    if (prefixed) ended = p[0] == '0' && (p[1] | 0x20) == 'x', p += 2; /* (fe_modn_from_hex reads right to left: behind 64 digits nothing counts) */
    ended = ended && (crlf ? p[64] == '\r' && p[65] == '\n' : p[64] == '\n');
    if (widest) { /* one load per record; the scalar goes straight to its slot, reduced there in the one case in 2^128 that needs it */
      if (ended && hex64_avx512(p, dst)) {
        if (dst[3] == ~0ull) memcpy(k.w, dst, 32), k = sc_reduce(k), memcpy(dst, k.w, 32);
        continue;
      }
      atomic_store(&b->bad, true);
      return NULL;
    }
    const bool ok = ended && (wide ? hex32_avx2(p, &k.w[3], &k.w[2]) && hex32_avx2(p + 32, &k.w[1], &k.w[0])
                                   : hex16_ssse3(p, &k.w[3]) && hex16_ssse3(p + 16, &k.w[2]) && hex16_ssse3(p + 32, &k.w[1]) && hex16_ssse3(p + 48, &k.w[0]));
    if (!ok) {
      atomic_store(&b->bad, true);
      return NULL;
    }
    k = sc_reduce(k);
    memcpy(dst, k.w, 32);
  }
  atomic_fetch_add(&fixed_read_us, t1 - t0), atomic_fetch_add(&fixed_decode_us, us_now() - t1);
#else
  (void)arg;
#endif
  return NULL;
}
/* stdin as such a file: the records it holds from the current offset and their length - 65 bytes (64 digits + LF), 66 (+ CR LF), or with
   -bin 32 (the scalars themselves); 0: not a regular file, -raw, no SSSE3, or the first line is not a 64-digit record */
static size_t mul_fixed_file_records(const run_t *run, off_t *pos, size_t *rec) {
  struct stat stt;
  char head[68], *first = head;
  *pos = lseek(0, 0, SEEK_CUR), *rec = MUL_RECORD;
  const char *how = getenv("ECLOOP_HIP_MUL_READ"); /* "chunks": the general reader only (tests compare the two) */
  if (how && !strcmp(how, "chunks")) return 0;
  if (run->opt.raw || *pos < 0 || fstat(0, &stt) != 0 || !S_ISREG(stt.st_mode) || stt.st_size < *pos + 66) return 0;
  if (run->bin) return *rec = 32, (size_t)(stt.st_size - *pos) / 32;
  if (!have_ssse3 || stt.st_size < *pos + 68 || pread(0, head, 68, *pos) != 68) return 0;
  size_t lead = 0;
  if (first[0] == '0' && (first[1] | 0x20) == 'x') first += 2, lead = 2; /* keys written as 0x + 64 digits */
  if (first[64] == '\r' && first[65] == '\n') *rec = lead + 66;
  else if (first[64] == '\n') *rec = lead + 65;
  else return 0;
  return (size_t)(stt.st_size - *pos) / *rec;
}
/* records per batch for a file of `total` records: 2^24 for large files (the device's rate needs calls that long); a smaller file is cut
   into about eight batches, 2^20 records at least, so that parsing one batch overlaps the device call of the one before and both
   contexts of the GPU get work (a 2^24-line file as ONE batch: parse 12 ms, then the device 15 ms, nothing overlapped) */
static size_t mul_batch_records(size_t total) {
  const char *e = getenv("ECLOOP_HIP_MUL_BATCH_LOG2"); /* experiments */
  if (e && atoi(e) >= 10 && atoi(e) <= 26) return (size_t)1 << atoi(e);
  size_t b = (size_t)1 << 20;
  while (b < ((size_t)1 << MUL_BATCH_LOG2) && b * 8 < total) b <<= 1;
  return b;
}

/* bytes per chunk of the general reader: 64 MB of hex lines (~1 M scalars per device call), 32 ... 256 MB of pass phrases with -raw (mul_raw_chunk).  (256 MB chunks
   for large files that are not all 64-digit records were measured in round 6: 0.70 against 0.78 G lines/s - the threads fault the mapping's
   pages in, and larger chunks only make them do it in lock-step.) */
static size_t mul_general_chunk(const run_t *run) { return run->opt.raw && !run->bin ? mul_raw_chunk() : MUL_TEXT_CHUNK; }
/* scalars of the largest array a run will hand to a device (bring-up sizes the device staging and the page-locked arrays by it), and
   the window width worth fixing up front when the input's size is known (st_size / 65 lines): the table is built during bring-up, and
   a wider one pays from a size on - 22 bits (1.5 GB, 40 ms; 1.11 G scalars/s on 2^24-scalar calls) below 2^28 lines, 24 bits (5.4 GB,
   +10 ms; 1.17) up to 2^31, 26 bits (19.6 GB, +90 ... 500 ms; 1.27) beyond (profiles/r04_mul_w_sweep.txt); 0 = leave it to the library
   (22, then 26 after 2^30 scalars: what a pipe gets) */
static size_t mul_largest_batch(const run_t *run, u32 *window) {
  off_t pos;
  size_t rec;
  size_t total = mul_fixed_file_records(run, &pos, &rec);
  if (!total && !run->opt.raw && !run->bin && (!getenv("ECLOOP_HIP_MUL_READ") || strcmp(getenv("ECLOOP_HIP_MUL_READ"), "chunks"))) {
    /* a regular file of hex lines that does not START with a record (a header line, a comment): the batch path will still take most of it
       (cmd_mul looks again after every stretch of the general reader), so the arrays and the window are sized for that */
    struct stat stt;
    if (pos >= 0 && fstat(0, &stt) == 0 && S_ISREG(stt.st_mode) && stt.st_size > pos) total = (size_t)(stt.st_size - pos) / MUL_RECORD;
  }
  const size_t batch = mul_batch_records(total);
  if (window) *window = !total ? 0 : total < ((size_t)1 << 28) ? 22 : total < ((size_t)1 << 31) ? 24 : 26;
  (void)rec;
  if (total) return total < batch ? total : batch;
  if (run->opt.raw && !run->bin) {
    const size_t est = mul_raw_lines_estimate();
    if (window && est) *window = est < ((size_t)1 << 28) ? 22 : est < ((size_t)1 << 31) ? 24 : 26;
    return mul_raw_chunk() / 12;
  }
  return run->bin ? MUL_TEXT_CHUNK / 32 : mul_general_chunk(run) / MUL_RECORD + 1024;
}
/* the fixed-record path over stdin from `pos`: batches -> arrays -> device threads; returns the records taken */
static size_t mul_fixed_file_run(run_t *run, pool_t *pool, int P, scalar_queue *sq, off_t pos, size_t total, size_t rec, u64 *t_array, u64 *t_grow, u64 *t_parse, u64 *nbatches) {
  const size_t batch = mul_batch_records(total);
  const char *how = getenv("ECLOOP_HIP_MUL_READ");
  const char *map = NULL;
  if (how && !strcmp(how, "mmap")) { /* comparison: the mapped form */
    struct stat stt;
    if (fstat(0, &stt) == 0) map = mmap(NULL, (size_t)stt.st_size, PROT_READ, MAP_PRIVATE, 0, 0);
    if (map == MAP_FAILED) map = NULL;
    else madvise((void *)map, (size_t)stt.st_size, MADV_SEQUENTIAL);
  }
  static fixed_file_slice fs[((size_t)1 << 26) / 4096 + 1];
  size_t slice = MUL_SLICE_RECORDS_MAX; /* records per slice (2 MB of text; measured: 32768 better than 8192 or 4096 - fewer hand-overs) */
  { const char *e = getenv("ECLOOP_HIP_MUL_SLICE"); if (e && atoi(e) >= 4096 && atoi(e) <= (int)MUL_SLICE_RECORDS_MAX) slice = (size_t)atoi(e); }
  size_t done = 0;
  (void)P;
  while (done < total) {
    const size_t nb = total - done < batch ? total - done : batch;
    u64 t_mark = us_now();
    pthread_mutex_lock(&sq->mu);
    while (!sq->nidle) pthread_cond_wait(&sq->cv, &sq->mu);
    const int ai = sq->idle[--sq->nidle];
    pthread_mutex_unlock(&sq->mu);
    *t_array += us_now() - t_mark, t_mark = us_now();
    scalar_array *ar = &sq->arr[ai];
    ks_grow(run, ar, nb);
    *t_grow += us_now() - t_mark, t_mark = us_now();
    fixed_batch b = {0, map, pos + (off_t)(done * rec), ar->ks, rec, false};
    int nf = 0;
    for (size_t at = 0; at < nb; at += slice, ++nf) fs[nf] = (fixed_file_slice){&b, at, at + slice < nb ? at + slice : nb};
    pool_run(pool, fixed_file_worker, fs, sizeof fs[0], nf);
    *t_parse += us_now() - t_mark;
    pthread_mutex_lock(&sq->mu);
    if (atomic_load(&b.bad)) { /* not such a batch after all: the array goes back, the general reader takes over from here */
      sq->idle[sq->nidle++] = ai;
      pthread_mutex_unlock(&sq->mu);
      break;
    }
    ar->n = nb;
    sq->ready[sq->nready++] = ai;
    pthread_cond_broadcast(&sq->cv);
    pthread_mutex_unlock(&sq->mu);
    done += nb, ++*nbatches;
  }
  if (lseek(0, pos + (off_t)(done * rec), SEEK_SET) < 0) { fprintf(stderr, "[!] lseek on the input failed\n"); exit(1); }
  return done;
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
  tq.bin = run->bin, tq.chunk = MUL_TEXT_CHUNK; /* (the general reader's chunk is decided after the batch path, from what is left: below) */
  pthread_mutex_init(&tq.mu, NULL), pthread_cond_init(&tq.cv, NULL);
  for (int i = 0; i < MUL_TEXT_RING; ++i) tq.ring[i].own = tq.ring[i].buf = NULL; /* the reader allocates them if it has to copy (a pipe) */
  scalar_queue sq;
  memset(&sq, 0, sizeof sq);
  sq.run = run, sq.narr = run->ngpus + 2;
  pthread_mutex_init(&sq.mu, NULL), pthread_cond_init(&sq.cv, NULL);
  for (int i = sq.narr; i-- > 0;) sq.idle[sq.nidle++] = i; /* (taken from the end: array 0 first - the ones allocated during bring-up) */
  for (int i = 0; i < mul_ready_count && i < sq.narr; ++i) sq.arr[i] = mul_ready_arrays[i]; /* allocated during bring-up */
  pthread_t reader, devth[MAX_GPUS];
  mul_dev_arg dargs[MAX_GPUS];
  for (int g = 0; g < run->ngpus; ++g) dargs[g] = (mul_dev_arg){&sq, g, 0, 0, 0, 0}, pthread_create(&devth[g], NULL, mul_device_worker, &dargs[g]);
  parse_slice sl[MUL_POOL_MAX];
  memset(sl, 0, sizeof sl);
  static raw_slice rs[MUL_POOL_MAX];
  memset(rs, 0, sizeof rs);
  pool_t pool;
  pool_init(&pool, P);
  u64 t_text = 0, t_array = 0, t_parse = 0, t_grow = 0, t_pack = 0, nchunks = 0, nfixed = 0, t_mark; /* us per stage (ECLOOP_HIP_STATS) */
  u64 nbatches = 0, nbatch_records = 0;
  /* The input is taken in stretches.  A regular file of 64-digit records goes batch by batch straight from the file; where that path meets a
     batch that holds anything else it stops, the general reader takes that batch's bytes (to the next line end), and the batch path is
     tried again from there - one odd line, or a commented header, costs one batch at the general reader's rate, not the file.  A file
     that does not start with a record is read a stretch at a time the general way, looking again after each; a pipe is read to its end. */
  size_t first_stretch = (size_t)1 << 20;
  { const char *e = getenv("ECLOOP_HIP_MUL_STRETCH"); /* tests */
    if (e && atol(e) >= 64) first_stretch = (size_t)atol(e); }
  int misses = 0;
  char *last_map = NULL;
  size_t last_map_size = 0;
  for (;;) {
    size_t limit = 0;
    {
      off_t pos;
      size_t rec;
      struct stat stt;
      const size_t total = mul_fixed_file_records(run, &pos, &rec);
      const bool file = pos >= 0 && fstat(0, &stt) == 0 && S_ISREG(stt.st_mode);
      if (total) {
        const size_t done = mul_fixed_file_run(run, &pool, P, &sq, pos, total, rec, &t_array, &t_grow, &t_parse, &nbatches);
        nbatch_records += done;
        if (pos + (off_t)(done * rec) >= stt.st_size) break; /* the whole file */
        if (done < total) { /* stopped at a batch that is not all records */
          const size_t batch = mul_batch_records(total);
          limit = (total - done < batch ? total - done : batch) * rec;
        }
        misses = 0;
      } else if (file && !run->opt.raw && !run->bin) {
        if (stt.st_size <= pos) break;
        /* a header costs 1 MB at the general reader's rate; a file with no records to find doubles the stretch each time, and after 8 looks
           (256 MB) the general reader keeps the rest (limit 0), which is what it did before there were stretches */
        limit = misses < 8 ? first_stretch << misses : 0;
        misses++;
      }
    }
    pthread_mutex_lock(&tq.mu);
    tq.head = tq.tail = tq.count = 0, tq.eof = tq.input_done = false, tq.limit = limit, tq.map = NULL, tq.map_size = 0;
    pthread_mutex_unlock(&tq.mu);
    tq.chunk = mul_general_chunk(run);
    pthread_create(&reader, NULL, mul_reader, &tq);
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
    pthread_join(reader, NULL);
    /* every chunk of the stretch has been parsed into an array.  The LAST mapping is left alone until the devices are done: munmap of 18 GB
       of touched pages holds the process's address-space lock for 0.2 s, and the device threads' launches and copies stood still behind it
       (rocprofv3 --hip-runtime-trace: one hipLaunchKernel of 220 ms and one hipMemcpyAsync of 217 ms, both threads at once - a fifth of
       a 2^30-line -raw run) */
    if (tq.input_done) { last_map = tq.map, last_map_size = tq.map_size; break; }
    if (tq.map) munmap(tq.map, tq.map_size);
  } /* stretches */
  pthread_mutex_lock(&sq.mu);
  sq.done = true;
  pthread_cond_broadcast(&sq.cv);
  pthread_mutex_unlock(&sq.mu);
  pool_stop(&pool);
  for (int g = 0; g < run->ngpus; ++g) pthread_join(devth[g], NULL);
  if (!run->parse_only) report_close(&run->rep); /* the search is over here: giving back page-locked arrays (0.1 ms per MB) is not part of it */
  if (last_map) munmap(last_map, last_map_size);
  for (int i = 0; i < MUL_TEXT_RING; ++i) free(tq.ring[i].own);
  for (int i = 0; i < sq.narr; ++i) { /* (page-locked arrays are left to the end of the process, which follows: unlocking 2 GB takes 0.2 s) */
    if (!sq.arr[i].pinned) ks_free(run, sq.arr[i].ks, false);
    if (!sq.arr[i].text_pinned) raw_release(sq.arr[i].text, false);
    if (!sq.arr[i].lines_pinned) raw_release(sq.arr[i].lines, false);
  }
  for (int i = 0; i < MUL_POOL_MAX; ++i) free(sl[i].tmp), free(rs[i].tmp);
  if (getenv("ECLOOP_HIP_STATS") && !run->parse_only)
    for (int g = 0; g < run->ngpus; ++g) {
      double ms = 0;
      uint64_t calls = 0, n = 0;
      ecl_hip_get_mul_timing(run->dev[g], &ms, &calls, &n);
      fprintf(stderr, "mul context %d: %llu arrays, %llu scalars, %.1f ms in device calls (%.1f ms by the library's events over %llu calls), %.1f ms waiting for parsed input\n", g,
              (unsigned long long)dargs[g].calls, (unsigned long long)dargs[g].scalars, dargs[g].busy_us / 1e3, ms, (unsigned long long)calls, dargs[g].wait_us / 1e3);
    }
  if (getenv("ECLOOP_HIP_STATS")) /* where the front end's wall time went (the main thread drives one chunk at a time) */
    fprintf(stderr, "mul front end: %llu batches of fixed records straight from the file (%llu lines), %llu chunks (%llu of fixed 65-byte records), %d pool threads; ms waiting for text %.1f, "
            "waiting for a free array (devices behind) %.1f, parse / copy %.1f (threads' sum: reading %.1f, decoding %.1f), array growth %.1f, pack %.1f\n", (unsigned long long)nbatches,
            (unsigned long long)nbatch_records, (unsigned long long)nchunks, (unsigned long long)nfixed, P, t_text / 1e3, t_array / 1e3, t_parse / 1e3,
            atomic_load(&fixed_read_us) / 1e3, atomic_load(&fixed_decode_us) / 1e3, t_grow / 1e3, t_pack / 1e3);
}
