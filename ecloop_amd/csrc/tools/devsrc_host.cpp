// devsrc_host.cpp — compiles the DEVICE headers (fe256.h, hash160.h, ...) for the host with g++ so the CPU
// test-suite can check the kernel source's logic against the oracle without a GPU
// (tests/test_devsrc_host.py).  Not part of the product library.
#include "../bloom.h"
#include "../ec.h"
#include "../hash160.h"
#include <string.h>

static void words(u32 w[8], const uint64_t a[4]) {
  for (int i = 0; i < 4; ++i) w[2 * i] = (u32)a[i], w[2 * i + 1] = (u32)(a[i] >> 32);
}
static fe ld(const uint64_t a[4]) {
  u32 w[8];
  words(w, a);
  return fe_from_words(w);
}
static void st(uint64_t r[4], fe a) {
  fe_normalize(a);
  u32 w[8];
  fe_to_words(w, a);
  for (int i = 0; i < 4; ++i) r[i] = (uint64_t)w[2 * i] | (uint64_t)w[2 * i + 1] << 32;
}
extern "C" {
void dh_fe_op(int op, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  fe x = ld(a), y = ld(b), z;
  switch (op) {
  case 0: z = fe_mul(x, y); break;
  case 1: z = fe_sqr(x); break;
  case 2: z = fe_inv(x); break;
  case 3: z = fe_sub(x, y); break;
  case 4: z = fe_add(x, y); break;
  case 5: z = fe_neg(x, 1); break;
  case 6: z = fe_mul(fe_add(fe_add(x, x), x), fe_add(y, y)); break;  // magnitudes 3 x 2
  case 7: z = fe_sqr(fe_add(x, y)); break;                           // magnitude 2
  case 8: {                                                          // magnitudes 1 x 7: the limit
    fe y7 = fe_neg(fe_neg(fe_neg(fe_neg(fe_neg(fe_neg(y, 1), 2), 3), 4), 5), 6);
    z = fe_mul(x, y7);
    break;
  }
  case 9: z = fe_mul(fe_neg(fe_add(x, x), 2), fe_add(y, y)); break;  // magnitudes 3 x 2 with a negated operand
  case 10: z = fe_inv_divsteps(x); break;
  case 11: z = fe_inv_fermat(x); break;
  case 12: z = fe_inv_divsteps(fe_add(fe_neg(fe_add(x, x), 2), fe_add(fe_add(y, y), fe_add(y, y)))); break;  // 1 / (4y - 2x), magnitude 7 in
  case 13: {  // the interleaved pair: (x y, y^2... ) with results aliasing operands; returns x*y + (x+y)^2's partner checked separately
    fe u = x, v = y;
    fe_mul2(u, v, u, y, v, x);  // u = x y, v = y x (aliased)
    z = fe_add(u, v);           // 2 x y
    break;
  }
  case 14: {
    fe u = x, v = y;
    fe_sqr2(u, v, u, v);        // aliased
    z = fe_add(u, v);           // x^2 + y^2
    break;
  }
  default: z = fe_zero();
  }
  st(r, z);
}
void dh_hash160(uint32_t h33[5], uint32_t h65[5], const uint64_t x[4], const uint64_t y[4]) {
  u32 fx[8], fy[8];
  words(fx, x), words(fy, y);
  hash160_33(h33, fx, fy[0] & 1);
  hash160_65(h65, fx, fy);
}
// k*G by the device's double-and-add, affine out; returns 0 if the result is the point at infinity
int dh_mulg(uint64_t x[4], uint64_t y[4], const uint64_t k[4]) {
  u32 kw[8];
  for (int i = 0; i < 4; ++i) kw[2 * i] = (u32)k[i], kw[2 * i + 1] = (u32)(k[i] >> 32);
  fe ax, ay;
  int ok = ec_mul_g_affine(ax, ay, kw);
  st(x, ax), st(y, ay);
  return ok;
}
// sum of n >= 1 affine points (x[i], y[i]) by the lazy XYZZ additions of `mul` (affine + affine first, then mixed), made affine the way
// k_mul_check does it (X ZZZ / T, Y ZZ / T with T = ZZ ZZZ); returns 0 when the chain degenerated (ZZ = 0)
int dh_xyzz_sum(uint64_t x[4], uint64_t y[4], const uint64_t* px, const uint64_t* py, int n) {
  xyzz acc;
  acc.X = ld(px), acc.Y = ld(py), acc.ZZ = fe_one(), acc.ZZZ = fe_one(), acc.inf = 0;
  for (int i = 1; i < n; ++i) {
    const fe qx = ld(px + 4 * i), qy = ld(py + 4 * i);
    acc = i == 1 ? xyzz_mmadd_lazy(acc.X, acc.Y, qx, qy) : xyzz_madd_lazy(acc, qx, qy);
  }
  if (fe_is_zero(acc.ZZ)) return 0;
  const fe ti = fe_inv(fe_mul(acc.ZZ, acc.ZZZ));
  st(x, fe_mul(fe_mul(acc.X, acc.ZZZ), ti)), st(y, fe_mul(fe_mul(acc.Y, acc.ZZ), ti));
  return 1;
}
int dh_parity(const uint64_t a[4], const uint64_t b[4]) { return (int)fe_parity(fe_sub(fe_add(ld(a), ld(a)), ld(b))); }
int dh_bloom_has(const uint64_t* bits, uint64_t nwords, const uint32_t h[5]) {
  bloom_t b = bloom_make(bits, nwords);
  return bloom_has(b, h) ? 1 : 0;
}
uint64_t dh_bloom_mod(uint64_t nwords, uint64_t x) { return bloom_mod(bloom_make(nullptr, nwords), x); }
}
