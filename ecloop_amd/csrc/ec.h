// ec.h — secp256k1 group operations for the set-up kernels (start points, table points, `mul` command).
//
// The reference's curve layer (lib/ecc.c:611-929) uses homogeneous projective formulas without any handling
// of the exceptional cases (it asserts, lib/ecc.c:666).  The hot loop never needs a general addition (see
// add_kernel.h: affine + affine with a batched inverse), so the general code here only serves the per-launch
// set-up and the `mul` path.  It is written for robustness: Jacobian coordinates, mixed addition with the
// doubling / inverse / infinity cases handled, results always returned as canonical affine coordinates, which
// is all that reaches the hash (an affine point is unique, so parity with lib/ecc.c is by value).
#pragma once
#include "fe256.h"

struct jac {
  fe X, Y, Z;
  u32 inf;  // 1 = point at infinity
};

// set-up code keeps every value at magnitude 1: add / subtract, then one weak normalisation
FE_FN fe fe_addn(const fe& a, const fe& b) {
  fe r = fe_add(a, b);
  fe_normalize_weak(r);
  return r;
}
FE_FN fe fe_subn(const fe& a, const fe& b) {
  fe r = fe_sub(a, b);
  fe_normalize_weak(r);
  return r;
}
FE_FN fe fe_dbl(const fe& a) { return fe_addn(a, a); }

// 2P, a = 0 curve: 2M + 5S ("dbl-2009-l")
FE_FN jac jac_dbl(const jac& p) {
  jac r;
  r.inf = p.inf;
  fe A = fe_sqr(p.X), B = fe_sqr(p.Y), C = fe_sqr(B);
  fe t = fe_sqr(fe_addn(p.X, B));
  fe D = fe_dbl(fe_subn(fe_subn(t, A), C));
  fe E = fe_addn(fe_dbl(A), A);
  fe F = fe_sqr(E);
  r.X = fe_subn(F, fe_dbl(D));
  fe C8 = fe_dbl(fe_dbl(fe_dbl(C)));
  r.Y = fe_subn(fe_mul(E, fe_subn(D, r.X)), C8);
  r.Z = fe_dbl(fe_mul(p.Y, p.Z));
  return r;
}

// P + (qx, qy) with Q affine and finite. Complete: handles P = inf, P = Q (doubling), P = -Q (infinity).
FE_FN jac jac_madd(const jac& p, const fe& qx, const fe& qy) {
  if (p.inf) {
    jac r;
    r.X = qx, r.Y = qy, r.Z = fe_one(), r.inf = 0;
    return r;
  }
  fe zz = fe_sqr(p.Z);
  fe u2 = fe_mul(qx, zz);
  fe s2 = fe_mul(qy, fe_mul(zz, p.Z));
  fe h = fe_subn(u2, p.X);
  fe rr = fe_subn(s2, p.Y);
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr)) return jac_dbl(p);
    jac r = p;
    r.inf = 1;
    return r;
  }
  fe hh = fe_sqr(h), hhh = fe_mul(hh, h), v = fe_mul(p.X, hh);
  jac r;
  r.inf = 0;
  r.X = fe_subn(fe_subn(fe_sqr(rr), hhh), fe_dbl(v));
  r.Y = fe_subn(fe_mul(rr, fe_subn(v, r.X)), fe_mul(p.Y, hhh));
  r.Z = fe_mul(p.Z, h);
  return r;
}

// ---- the window sums of `mul` (k_mul_check): mixed additions with lazy magnitudes and no exceptional cases -------
// A sum of table points b_w * 2^(W w) * G over distinct windows never meets P = +-Q on the way unless the scalar itself is
// 0 (mod n) or crafted around n - and then h = 0 makes ZZ3 = ZZ1 * h^2 = 0, which stays 0 through every later addition: the
// caller tests ZZ ONCE at the end and sends such a scalar through the complete formulas above (wtab_sum), instead of
// testing h in each of the additions.  Y is never normalised (it only ever enters a subtraction that is normalised anyway,
// or a multiplication); three weak normalisations per addition (h, r, X3) where jac_madd spends seven.
// Z^2 and Z^3 are carried instead of Z ("XYZZ": x = X / ZZ, y = Y / ZZZ): the squaring of Z that opens every Jacobian mixed
// addition goes away - 8M + 2S per table point instead of 8M + 3S - and the affine + affine start (4M + 2S) hands over hh, hhh
// as they are.  Magnitudes: X, ZZ, ZZZ in / out 1, Y in / out <= 3; h = 0 on the way leaves ZZ = ZZZ = 0 for good.
struct xyzz {
  fe X, Y, ZZ, ZZZ;
  u32 inf;
};
// neg != 0: P - Q, i.e. (qx, -qy) is added: the sign goes on s2 = qy ZZZ (one select per limb, inside the sum that is normalised anyway).
// The ten field operations of an addition come in five independent pairs; each pair runs as one interleaved instruction stream
// (fe256.h: fe_mul2 / fe_sqr2 - four accumulator chains instead of two, no wait states between dependent multiply-adds), which is worth
// 5 % of a multiplication at the two waves per SIMD the `mul` kernel runs at (csrc/tools/femul_bench.hip; ECL_MUL_PAIRS=0: one by one).
#ifndef ECL_MUL_PAIRS
#define ECL_MUL_PAIRS 1
#endif
FE_FN void fe_mul_pair(fe& r1, fe& r2, const fe& a1, const fe& b1, const fe& a2, const fe& b2) {
#if ECL_MUL_PAIRS
  fe_mul2(r1, r2, a1, b1, a2, b2);
#else
  r1 = fe_mul(a1, b1), r2 = fe_mul(a2, b2);
#endif
}
FE_FN void fe_sqr_pair(fe& r1, fe& r2, const fe& a1, const fe& a2) {
#if ECL_MUL_PAIRS
  fe_sqr2(r1, r2, a1, a2);
#else
  r1 = fe_sqr(a1), r2 = fe_sqr(a2);
#endif
}
FE_FN xyzz xyzz_madd_lazy(const xyzz& p, const fe& qx, const fe& qy, u32 neg = 0) {
  fe u2, s2p;
  fe_mul_pair(u2, s2p, qx, p.ZZ, qy, p.ZZZ);
  const fe s2m = fe_neg(s2p, 1);
  fe s2;
#pragma unroll
  for (int l = 0; l < FE_LIMBS; ++l) s2.n[l] = neg ? s2m.n[l] : s2p.n[l];
  fe h = fe_sub(u2, p.X);                   // magnitude 3
  fe_normalize_weak(h);
  fe rr = fe_add(s2, fe_neg(p.Y, 3));       // magnitude <= 6
  fe_normalize_weak(rr);
  fe hh, rr2, hhh, v, t1, t2;
  fe_sqr_pair(hh, rr2, h, rr);
  fe_mul_pair(hhh, v, hh, h, p.X, hh);
  xyzz r;
  r.inf = 0;
  r.X = fe_add(fe_add(rr2, fe_neg(hhh, 1)), fe_neg(fe_add(v, v), 2));  // magnitude 6
  fe_normalize_weak(r.X);
  fe_mul_pair(t1, t2, rr, fe_sub(v, r.X), p.Y, hhh);
  r.Y = fe_add(t1, fe_neg(t2, 1));                                     // magnitude 3
  fe_mul_pair(r.ZZ, r.ZZZ, p.ZZ, hh, p.ZZZ, hhh);
  return r;
}
FE_FN xyzz xyzz_mmadd_lazy(const fe& px, const fe& py, const fe& qx, const fe& qy) {
  fe h = fe_sub(qx, px), rr = fe_sub(qy, py);
  fe_normalize_weak(h);
  fe_normalize_weak(rr);
  fe hh, rr2, hhh, v, t1, t2;
  fe_sqr_pair(hh, rr2, h, rr);
  fe_mul_pair(hhh, v, hh, h, px, hh);
  xyzz r;
  r.inf = 0;
  r.X = fe_add(fe_add(rr2, fe_neg(hhh, 1)), fe_neg(fe_add(v, v), 2));
  fe_normalize_weak(r.X);
  fe_mul_pair(t1, t2, rr, fe_sub(v, r.X), py, hhh);
  r.Y = fe_add(t1, fe_neg(t2, 1));
  r.ZZ = hh, r.ZZZ = hhh;
  return r;
}
FE_FN xyzz xyzz_from_jac(const jac& p) {
  xyzz r;
  r.X = p.X, r.Y = p.Y, r.inf = p.inf;
  r.ZZ = fe_sqr(p.Z);
  r.ZZZ = fe_mul(r.ZZ, p.Z);
  return r;
}

// Jacobian -> canonical affine (x = X/Z^2, y = Y/Z^3); returns 0 for the point at infinity
FE_FN int jac_to_affine(fe& x, fe& y, const jac& p) {
  if (p.inf) {
    x = fe_zero(), y = fe_zero();
    return 0;
  }
  fe zi = fe_inv(p.Z), zi2 = fe_sqr(zi);
  x = fe_mul(p.X, zi2);
  y = fe_mul(p.Y, fe_mul(zi2, zi));
  fe_normalize(x);
  fe_normalize(y);
  return 1;
}

// k*G, MSB-first double-and-add (the reference's ec_jacobi_mulrdc(&G1, k), lib/ecc.c:821-853, by value)
__host__ __device__ __noinline__ inline int ec_mul_g_affine(fe& x, fe& y, const u32 k[8]) {
  const u32 gxw[8] = FE_GX_W, gyw[8] = FE_GY_W;
  const fe gx = fe_from_words(gxw), gy = fe_from_words(gyw);
  jac acc;
  acc.X = fe_zero(), acc.Y = fe_zero(), acc.Z = fe_one(), acc.inf = 1;
#pragma unroll 1
  for (int bit = 255; bit >= 0; --bit) {
    if (!acc.inf) acc = jac_dbl(acc);
    if ((k[bit >> 5] >> (bit & 31)) & 1u) acc = jac_madd(acc, gx, gy);
  }
  return jac_to_affine(x, y, acc);
}
