// scalar_host.h — 256-bit scalars modulo the group order n, host side (plain integer arithmetic).
// Covers what the reference's fe_modn_* (lib/ecc.c:166-265) is used for on the host side of the boundary:
// stepping range starts, stride multiples, and the endomorphism maps of calc_priv (main.c:267-276).
// All functions take and return canonical values in [0, n).
#pragma once
#include <stdint.h>
#include <string.h>

struct u256 {
  uint64_t w[4];
};

static const u256 SC_N = {{0xbfd25e8cd0364141ULL, 0xbaaedce6af48a03bULL, 0xfffffffffffffffeULL, 0xffffffffffffffffULL}};
static const u256 SC_LAMBDA = {{0xdf02967c1b23bd72ULL, 0x122e22ea20816678ULL, 0xa5261c028812645aULL, 0x5363ad4cc05c30e0ULL}};

static inline u256 u256_from(const uint64_t a[4]) {
  u256 r;
  memcpy(r.w, a, 32);
  return r;
}
static inline u256 u256_u64(uint64_t v) {
  u256 r = {{v, 0, 0, 0}};
  return r;
}
static inline bool u256_eq(const u256& a, const u256& b) { return memcmp(a.w, b.w, 32) == 0; }
static inline int u256_cmp(const u256& a, const u256& b) {
  for (int i = 3; i >= 0; --i)
    if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1;
  return 0;
}
static inline uint64_t u256_add(u256& r, const u256& a, const u256& b) {
  unsigned __int128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (unsigned __int128)a.w[i] + b.w[i];
    r.w[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
static inline uint64_t u256_sub(u256& r, const u256& a, const u256& b) {
  uint64_t br = 0;
  for (int i = 0; i < 4; ++i) {
    unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - br;
    r.w[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
// any 256-bit value -> [0, n)   (n > 2^255, so one subtraction suffices)
static inline u256 sc_reduce(u256 a) {
  if (u256_cmp(a, SC_N) >= 0) u256_sub(a, a, SC_N);
  return a;
}
static inline u256 sc_add(const u256& a, const u256& b) {
  u256 r;
  uint64_t c = u256_add(r, a, b);
  if (c || u256_cmp(r, SC_N) >= 0) u256_sub(r, r, SC_N);
  return r;
}
static inline u256 sc_neg(const u256& a) {
  u256 z = {{0, 0, 0, 0}}, r;
  if (u256_eq(a, z)) return z;
  u256_sub(r, SC_N, a);
  return r;
}
static inline u256 sc_mul(const u256& a, const u256& b) {  // double-and-add: a few hundred ns, host set-up only
  u256 r = {{0, 0, 0, 0}};
  for (int bit = 255; bit >= 0; --bit) {
    r = sc_add(r, r);
    if ((b.w[bit >> 6] >> (bit & 63)) & 1) r = sc_add(r, a);
  }
  return r;
}
static inline u256 sc_mul_u64(const u256& a, uint64_t m) { return sc_mul(a, u256_u64(m)); }
static inline u256 sc_pow2(unsigned e) {  // 2^e mod n, e <= 255
  u256 r = u256_u64(1);
  for (unsigned i = 0; i < e; ++i) r = sc_add(r, r);
  return r;
}
// a / 2 mod n (n is odd)
static inline u256 sc_half(u256 a) {
  uint64_t carry = 0;
  if (a.w[0] & 1) carry = u256_add(a, a, SC_N);
  for (int i = 0; i < 4; ++i) a.w[i] = (a.w[i] >> 1) | ((i < 3 ? a.w[i + 1] : carry) << 63);
  return a;
}
