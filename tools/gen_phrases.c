/* gen_phrases.c - measurement input for `mul -raw`: N pass-phrase-like lines of 8..24 characters [a-z0-9] (seeded splitmix64 stream),
   written to a file by T threads - every thread makes its own share of the lines in memory, the shares are written one after the other.
     gcc -O2 -pthread tools/gen_phrases.c -o /tmp/gen_phrases && /tmp/gen_phrases 1073741824 11 /dev/shm/mul_raw.txt 32 */
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
typedef struct { uint64_t seed, first, last; char *buf; size_t len; } job_t;
static void *work(void *arg) {
  job_t *j = arg;
  static const char abc[] = "abcdefghijklmnopqrstuvwxyz0123456789";
  char *p = j->buf = malloc((j->last - j->first) * 25 + 64);
  for (uint64_t i = j->first; i < j->last; ++i) {
    uint64_t v = mix(j->seed + (3 * i + 1) * 0x9E3779B97F4A7C15ull), w = mix(v + 1), x = mix(w + 2);
    const int n = 8 + (int)(v % 17);
    v >>= 8;
    for (int c = 0; c < n; ++c) {
      if (c == 10) v = w; else if (c == 20) v = x;
      *p++ = abc[v % 36], v /= 36;
    }
    *p++ = '\n';
  }
  j->len = (size_t)(p - j->buf);
  return NULL;
}
int main(int argc, char **argv) {
  if (argc < 4) return fprintf(stderr, "usage: gen_phrases N seed file [threads]\n"), 2;
  const uint64_t n = strtoull(argv[1], 0, 0), seed = strtoull(argv[2], 0, 0);
  int T = argc > 4 ? atoi(argv[4]) : 16;
  if (T < 1) T = 1; if (T > 256) T = 256;
  int fd = open(argv[3], O_CREAT | O_TRUNC | O_WRONLY, 0644);
  if (fd < 0) return perror(argv[3]), 1;
  pthread_t th[256];
  job_t jobs[256];
  size_t total = 0;
  /* in rounds of T shares of at most 2^22 lines each, so that memory stays bounded */
  for (uint64_t at = 0; at < n;) {
    int used = 0;
    for (; used < T && at < n; ++used) {
      const uint64_t take = n - at < (1u << 22) ? n - at : (1u << 22);
      jobs[used] = (job_t){seed, at, at + take, NULL, 0};
      at += take;
      pthread_create(&th[used], NULL, work, &jobs[used]);
    }
    for (int t = 0; t < used; ++t) {
      pthread_join(th[t], NULL);
      for (size_t off = 0; off < jobs[t].len;) {
        ssize_t w = write(fd, jobs[t].buf + off, jobs[t].len - off);
        if (w <= 0) return perror("write"), 1;
        off += (size_t)w;
      }
      total += jobs[t].len, free(jobs[t].buf);
    }
  }
  close(fd);
  printf("lines %llu bytes %zu\n", (unsigned long long)n, total);
  return 0;
}
