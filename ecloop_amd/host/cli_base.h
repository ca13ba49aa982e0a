/* cli_base.h - 256-bit scalars mod n, small utilities, the command line (opts_t).
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
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
/* 32 hex characters -> two little-endian u64 (hi = characters 0..15, lo = 16..31); false if any character is not a hex digit */
__attribute__((target("avx2"))) static bool hex32_avx2(const char *p, u64 *hi, u64 *lo) {
  const __m256i c = _mm256_loadu_si256((const __m256i *)p);
  const __m256i lower = _mm256_or_si256(c, _mm256_set1_epi8(0x20));
  const __m256i isdig = _mm256_and_si256(_mm256_cmpgt_epi8(c, _mm256_set1_epi8('0' - 1)), _mm256_cmpgt_epi8(_mm256_set1_epi8('9' + 1), c));
  const __m256i isalp = _mm256_and_si256(_mm256_cmpgt_epi8(lower, _mm256_set1_epi8('a' - 1)), _mm256_cmpgt_epi8(_mm256_set1_epi8('f' + 1), lower));
  if ((u32)_mm256_movemask_epi8(_mm256_or_si256(isdig, isalp)) != 0xFFFFFFFFu) return false;
  const __m256i nib = _mm256_add_epi8(_mm256_and_si256(c, _mm256_set1_epi8(0x0F)), _mm256_and_si256(isalp, _mm256_set1_epi8(9)));
  const __m256i pair = _mm256_maddubs_epi16(nib, _mm256_set1_epi16(0x0110)); /* first digit * 16 + second digit, per 128-bit lane */
  const __m256i bytes = _mm256_packus_epi16(pair, pair);                      /* each lane: its 8 bytes, most significant first, twice */
  const __m256i rev = _mm256_shuffle_epi8(bytes, _mm256_setr_epi8(7, 6, 5, 4, 3, 2, 1, 0, -1, -1, -1, -1, -1, -1, -1, -1, 7, 6, 5, 4, 3, 2, 1, 0, -1, -1, -1,
                                                                  -1, -1, -1, -1, -1));
  *hi = (u64)_mm256_extract_epi64(rev, 0), *lo = (u64)_mm256_extract_epi64(rev, 2);
  return true;
}
/* a whole 64-digit record -> the scalar's four little-endian u64, stored at dst (32 bytes); false if any character is not a hex digit */
__attribute__((target("avx512f,avx512bw,avx512vl"))) static bool hex64_avx512(const char *p, u64 *dst) {
  const __m512i c = _mm512_loadu_si512((const void *)p);
  const __m512i lower = _mm512_or_si512(c, _mm512_set1_epi8(0x20));
  const __mmask64 isdig = _mm512_cmpgt_epi8_mask(c, _mm512_set1_epi8('0' - 1)) & _mm512_cmplt_epi8_mask(c, _mm512_set1_epi8('9' + 1));
  const __mmask64 isalp = _mm512_cmpgt_epi8_mask(lower, _mm512_set1_epi8('a' - 1)) & _mm512_cmplt_epi8_mask(lower, _mm512_set1_epi8('f' + 1));
  if ((isdig | isalp) != ~(__mmask64)0) return false;
  const __m512i low = _mm512_and_si512(c, _mm512_set1_epi8(0x0F));
  const __m512i nib = _mm512_mask_add_epi8(low, isalp, low, _mm512_set1_epi8(9));
  const __m512i pair = _mm512_maddubs_epi16(nib, _mm512_set1_epi16(0x0110)); /* first digit * 16 + second digit */
  const __m256i bytes = _mm512_cvtepi16_epi8(pair);                          /* 32 bytes, most significant first */
  const __m256i rev = _mm256_shuffle_epi8(bytes, _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4,
                                                                  3, 2, 1, 0));
  _mm256_storeu_si256((__m256i *)dst, _mm256_permute2x128_si256(rev, rev, 1)); /* all 32 bytes reversed: least significant limb first */
  return true;
}
/* a line of 1..64 hex digits (most significant first, any length: "1f", 40 digits, 64) -> the value's four little-endian u64 at dst; false
   if any character is not a hex digit (the caller's scalar loop then applies fe_modn_from_hex's skipping).  The line is loaded right-
   aligned - the 64 bytes that END at the line's end, the bytes in front of the line masked off and read as '0' (a masked load does not
   touch what it masks, so the line may start a buffer) - and decoded like a whole record. */
__attribute__((target("avx512f,avx512bw,avx512vl"))) static bool hexline_avx512(const char *p, size_t len, u64 *dst) {
  const __mmask64 m = ~(__mmask64)0 << (64 - len);
  const __m512i c = _mm512_mask_loadu_epi8(_mm512_set1_epi8('0'), m, (const void *)(p + len - 64));
  const __m512i lower = _mm512_or_si512(c, _mm512_set1_epi8(0x20));
  const __mmask64 isdig = _mm512_cmpgt_epi8_mask(c, _mm512_set1_epi8('0' - 1)) & _mm512_cmplt_epi8_mask(c, _mm512_set1_epi8('9' + 1));
  const __mmask64 isalp = _mm512_cmpgt_epi8_mask(lower, _mm512_set1_epi8('a' - 1)) & _mm512_cmplt_epi8_mask(lower, _mm512_set1_epi8('f' + 1));
  if ((isdig | isalp) != ~(__mmask64)0) return false;
  const __m512i low = _mm512_and_si512(c, _mm512_set1_epi8(0x0F));
  const __m512i nib = _mm512_mask_add_epi8(low, isalp, low, _mm512_set1_epi8(9));
  const __m512i pair = _mm512_maddubs_epi16(nib, _mm512_set1_epi16(0x0110));
  const __m256i bytes = _mm512_cvtepi16_epi8(pair);
  const __m256i rev = _mm256_shuffle_epi8(bytes, _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4,
                                                                  3, 2, 1, 0));
  _mm256_storeu_si256((__m256i *)dst, _mm256_permute2x128_si256(rev, rev, 1));
  return true;
}
#endif
static bool have_ssse3, have_avx2, have_avx512; /* set once in main */

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
