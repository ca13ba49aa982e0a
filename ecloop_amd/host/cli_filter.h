/* cli_filter.h - filter_t: .blf files and hash lists on the host side (load_filter, main.c:71-131).
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
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
