/* cli_rnd_blf.h - `rnd` (window generator), `blf-gen` / `blf-check`, range and window arguments, usage.
   Part of the one translation unit ecloop_hip_cli.c (included there, in this order). */
/* ------------------------------------------------------------------------------------------- rnd */
/* 64 random bits: /dev/urandom, or - with -seed - pairs of rand() (utils.c:83-113).  The seeded stream is glibc's rand() after
   srand(encode_seed(seed)) (main.c:800-805, utils.c:105-113) drawn from a generator state of this program's own (random_r): rand()'s
   state is process-wide, and the GPU runtime's threads draw from it too - with it a seeded run gave other windows every time
   (round 6: found when two runs with one seed were compared) */
static struct random_data seeded_state;
static char seeded_buf[128];
static void seeded_start(const char *seed) {
  u32 hsh = 0; /* encode_seed, utils.c:105-113 */
  for (const char *c = seed; *c; ++c) hsh = (hsh << 5) - hsh + (u8)*c;
  memset(&seeded_state, 0, sizeof seeded_state);
  initstate_r(hsh, seeded_buf, sizeof seeded_buf, &seeded_state); /* a 128-byte state = rand()'s own generator (TYPE_3), seeded like srand() */
}
static u64 random_u64(bool seeded) {
  if (seeded) {
    int32_t hi = 0, lo = 0;
    random_r(&seeded_state, &hi), random_r(&seeded_state, &lo); /* _prand64: (u64)rand() << 32 | (u64)rand(), left operand first */
    return (u64)(u32)hi << 32 | (u64)(u32)lo;
  }
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
      "  blf-check       - check bloom filter for given hex-encoded hash160\n\n"};
  printf("Usage: %s <cmd> [-t <gpus>] [-f <file>] [-a <addr_type>] [-r <range>]\nv%s ~ MI355X build of the ecloop command set\n", prog, VERSION);
  for (size_t i = 0; i < sizeof TEXT / sizeof TEXT[0]; ++i) fputs(TEXT[i], stdout);
}
