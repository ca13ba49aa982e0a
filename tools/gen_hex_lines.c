/* gen_hex_lines.c - measurement input for `mul` (BASELINE.json configs[4]): N lines of 64 hex digits (seeded splitmix64 stream, four words
   per line, most significant first) written to a file by T threads with pwrite.  bench.py / tools/bench_mul_cli.sh compile it on demand:
     gcc -O2 -pthread tools/gen_hex_lines.c -o /tmp/gen_hex_lines && /tmp/gen_hex_lines 268435456 7 /dev/shm/mul_in.txt 32
   Line i is a function of (seed, i) only, so any slice can be regenerated (tests compare the host program's parse with it). */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

static inline uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
typedef struct { int fd; uint64_t seed, first, last; int lead; } job_t; /* lead: 2 = every line starts with "0x" (67-byte records) */
static void *work(void *arg) {
  const job_t *j = arg;
  enum { BLOCK = 16384 };
  static const char hex[] = "0123456789abcdef";
  const size_t rec = 65 + (size_t)j->lead;
  char *buf = malloc((size_t)BLOCK * rec);
  for (uint64_t at = j->first; at < j->last; at += BLOCK) {
    const uint64_t n = j->last - at < BLOCK ? j->last - at : BLOCK;
    for (uint64_t r = 0; r < n; ++r) {
      char *p = buf + r * rec;
      if (j->lead) *p++ = '0', *p++ = 'x';
      for (int w = 0; w < 4; ++w) {
        const uint64_t v = mix(j->seed + ((at + r) * 4 + (uint64_t)w + 1) * 0x9E3779B97F4A7C15ull);
        for (int d = 0; d < 16; ++d) p[16 * w + d] = hex[(v >> (60 - 4 * d)) & 15];
      }
      p[64] = '\n';
    }
    size_t off = 0, bytes = (size_t)n * rec;
    while (off < bytes) {
      ssize_t k = pwrite(j->fd, buf + off, bytes - off, (off_t)(at * rec + off));
      if (k <= 0) { perror("pwrite"); exit(1); }
      off += (size_t)k;
    }
  }
  free(buf);
  return NULL;
}
int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s N seed out [threads] [0x]\n", argv[0]); return 2; }
  const uint64_t n = strtoull(argv[1], NULL, 0), seed = strtoull(argv[2], NULL, 0);
  int t = argc > 4 ? atoi(argv[4]) : 16;
  if (t < 1) t = 1;
  if (t > 256) t = 256;
  const int lead = argc > 5 && argv[5][0] == '0' ? 2 : 0;
  const int fd = open(argv[3], O_CREAT | O_TRUNC | O_WRONLY, 0644);
  if (fd < 0 || ftruncate(fd, (off_t)(n * (uint64_t)(65 + lead))) != 0) { perror(argv[3]); return 1; }
  pthread_t th[256];
  job_t job[256];
  for (int i = 0; i < t; ++i) {
    job[i] = (job_t){fd, seed, n * (uint64_t)i / (uint64_t)t, n * (uint64_t)(i + 1) / (uint64_t)t, lead};
    pthread_create(&th[i], NULL, work, &job[i]);
  }
  for (int i = 0; i < t; ++i) pthread_join(th[i], NULL);
  close(fd);
  return 0;
}
