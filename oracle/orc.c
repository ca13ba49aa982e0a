/*
 * orc.c — CPU ORACLE (test infrastructure only; see orc.h for the rules and the parity status).
 *
 * Restates, in plain C, the algorithm of the reference hot path.  Citations are file:line into
 * /root/reference (vladkens/ecloop v0.5.0).  The code is written from the algorithm, not from the
 * reference text: generic limb loops instead of the reference's hand-unrolled carry chains, table
 * driven RIPEMD-160, one exact long-division mod n instead of the reference's Montgomery pair.
 * Where the reference has a behavioural quirk that changes WHICH keys get hashed or WHAT is
 * reported (non-reducing adds, job rounding, overrun of the range end, status counters), the quirk
 * is restated and marked QUIRK.
 */
#include "orc.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define GROUP 2048u            /* GROUP_INV_SIZE, main.c:17 */
#define HALF (GROUP / 2)
#define MAX_JOB (2u * 1024 * 1024) /* MAX_JOB_SIZE, main.c:16 */

/* ---------------------------------------------------------------- 256-bit helpers */

static const orc_fe P_ = {0xfffffffefffffc2fULL, ~0ULL, ~0ULL, ~0ULL};                               /* ecc.c:32 */
static const orc_fe N_ = {0xbfd25e8cd0364141ULL, 0xbaaedce6af48a03bULL, 0xfffffffffffffffeULL, ~0ULL}; /* ecc.c:33 */
static const u64 PK = 0x1000003D1ULL; /* 2^256 - p */

/* endomorphism constants lambda, lambda^2 (mod n), beta, beta^2 (mod p): ecc.c:36-39 */
static const orc_fe LAM1 = {0xdf02967c1b23bd72ULL, 0x122e22ea20816678ULL, 0xa5261c028812645aULL, 0x5363ad4cc05c30e0ULL};
static const orc_fe LAM2 = {0xe0cfc810b51283ceULL, 0xa880b9fc8ec739c2ULL, 0x5ad9e3fd77ed9ba4ULL, 0xac9c52b33fa3cf1fULL};
static const orc_fe BET1 = {0xc1396c28719501eeULL, 0x9cf0497512f58995ULL, 0x6e64479eac3434e9ULL, 0x7ae96a2b657c0710ULL};
static const orc_fe BET2 = {0x3ec693d68e6afa40ULL, 0x630fb68aed0a766aULL, 0x919bb86153cbcb16ULL, 0x851695d49a83f8efULL};

static const orc_pt GEN = {/* ecc.c:550-554 */
                           {0x59f2815b16f81798ULL, 0x029bfcdb2dce28d9ULL, 0x55a06295ce870b07ULL, 0x79be667ef9dcbbacULL},
                           {0x9c47d08ffb10d4b8ULL, 0xfd17b448a6855419ULL, 0x5da4fbfc0e1108a8ULL, 0x483ada7726a3c465ULL},
                           {1, 0, 0, 0}};

static inline void fe_cpy(orc_fe r, const orc_fe a) { memcpy(r, a, sizeof(orc_fe)); }
static inline void fe_set(orc_fe r, u64 v) { r[0] = v, r[1] = r[2] = r[3] = 0; }
static inline int fe_cmp(const orc_fe a, const orc_fe b) {
  for (int i = 3; i >= 0; --i)
    if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
  return 0;
}
/* r = a + b, returns carry out of 2^256 */
static inline u64 add256(orc_fe r, const orc_fe a, const orc_fe b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a[i] + b[i];
    r[i] = (u64)c;
    c >>= 64;
  }
  return (u64)c;
}
/* r = a - b, returns borrow */
static inline u64 sub256(orc_fe r, const orc_fe a, const orc_fe b) {
  u64 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - br;
    r[i] = (u64)d;
    br = (u64)(d >> 64) & 1;
  }
  return br;
}
static unsigned fe_bits(const orc_fe a) { /* ecc.c:52-57 */
  for (int i = 3; i >= 0; --i)
    if (a[i]) return 64 * i + (64 - __builtin_clzll(a[i]));
  return 0;
}

/* ---------------------------------------------------------------- field mod p */

/* ecc.c:292-305 — QUIRK: reduces only when the sum overflows 2^256, so the result may lie in [p, 2^256). */
void orc_fp_add(orc_fe r, const orc_fe a, const orc_fe b) {
  if (add256(r, a, b)) sub256(r, r, P_);
}
/* ecc.c:277-290 */
void orc_fp_sub(orc_fe r, const orc_fe a, const orc_fe b) {
  if (sub256(r, a, b)) add256(r, r, P_);
}
/* ecc.c:269-275 — p - a without reduction: neg(0) = p */
void orc_fp_neg(orc_fe r, const orc_fe a) { sub256(r, P_, a); }

/* ecc.c:307-347: 4x4 schoolbook -> 512 bit; fold the high half by 2^256 = 0x1000003D1 (mod p) into
   320 bit; fold the 5th limb (plus carry) once more; one conditional subtract -> canonical [0,p).
   QUIRK kept: the carry out of the second fold is discarded (cannot happen for reduced inputs). */
void orc_fp_mul(orc_fe r, const orc_fe a, const orc_fe b) {
  u64 w[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 t = (u128)a[j] * b[i] + w[i + j] + carry;
      w[i + j] = (u64)t;
      carry = (u64)(t >> 64);
    }
    w[i + 4] = carry;
  }
  /* hi * K -> 5 limbs */
  u64 f[5], carry = 0;
  for (int j = 0; j < 4; ++j) {
    u128 t = (u128)w[4 + j] * PK + carry;
    f[j] = (u64)t;
    carry = (u64)(t >> 64);
  }
  f[4] = carry;
  orc_fe lo = {w[0], w[1], w[2], w[3]}, fl = {f[0], f[1], f[2], f[3]};
  u64 c = add256(lo, lo, fl);
  u128 t2 = (u128)(f[4] + c) * PK;
  orc_fe add2 = {(u64)t2, (u64)(t2 >> 64), 0, 0};
  add256(r, lo, add2);
  if (fe_cmp(r, P_) >= 0) sub256(r, r, P_);
}
/* ecc.c:349-444 computes the same 512-bit square with 10 products and the same two folds. */
void orc_fp_sqr(orc_fe r, const orc_fe a) { orc_fp_mul(r, a, a); }

/* ecc.c:463-520: a^(p-2) by the 255 S + 15 M addition chain
   x2,x3,x6,x9,x11,x22,x44,x88,x176,x220,x223, then 23/5/3/2 squarings with x22,a,x2,a. */
static void sqr_n(orc_fe r, const orc_fe a, int n) {
  fe_cpy(r, a);
  while (n-- > 0) orc_fp_sqr(r, r);
}
void orc_fp_inv(orc_fe r, const orc_fe a) {
  orc_fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t;
  sqr_n(t, a, 1), orc_fp_mul(x2, t, a);
  sqr_n(t, x2, 1), orc_fp_mul(x3, t, a);
  sqr_n(t, x3, 3), orc_fp_mul(x6, t, x3);
  sqr_n(t, x6, 3), orc_fp_mul(x9, t, x3);
  sqr_n(t, x9, 2), orc_fp_mul(x11, t, x2);
  sqr_n(t, x11, 11), orc_fp_mul(x22, t, x11);
  sqr_n(t, x22, 22), orc_fp_mul(x44, t, x22);
  sqr_n(t, x44, 44), orc_fp_mul(x88, t, x44);
  sqr_n(t, x88, 88), orc_fp_mul(x176, t, x88);
  sqr_n(t, x176, 44), orc_fp_mul(x220, t, x44);
  sqr_n(t, x220, 3), orc_fp_mul(x223, t, x3);
  sqr_n(t, x223, 23), orc_fp_mul(t, t, x22);
  sqr_n(t, t, 5), orc_fp_mul(t, t, a);
  sqr_n(t, t, 3), orc_fp_mul(t, t, x2);
  sqr_n(t, t, 2), orc_fp_mul(r, t, a);
}

/* ecc.c:522-540: Montgomery's trick, in place. A zero input zeroes the whole batch (no check). */
void orc_fp_grpinv(orc_fe *r, uint32_t n) {
  if (n == 0) return;
  orc_fe *pre = (orc_fe *)malloc((size_t)n * sizeof(orc_fe));
  fe_cpy(pre[0], r[0]);
  for (u32 i = 1; i < n; ++i) orc_fp_mul(pre[i], pre[i - 1], r[i]);
  orc_fe acc, tmp;
  orc_fp_inv(acc, pre[n - 1]);
  for (u32 i = n - 1; i > 0; --i) {
    orc_fp_mul(tmp, acc, pre[i - 1]);
    orc_fp_mul(acc, r[i], acc);
    fe_cpy(r[i], tmp);
  }
  fe_cpy(r[0], acc);
  free(pre);
}

/* ---------------------------------------------------------------- scalars mod n */

/* ecc.c:174-187 — QUIRK: like the field add, reduces only on 2^256 overflow. */
void orc_sn_add(orc_fe r, const orc_fe a, const orc_fe b) {
  if (add256(r, a, b)) sub256(r, r, N_);
}
/* ecc.c:189-202 */
void orc_sn_sub(orc_fe r, const orc_fe a, const orc_fe b) {
  if (sub256(r, a, b)) add256(r, r, N_);
}
/* ecc.c:166-172 */
void orc_sn_neg(orc_fe r, const orc_fe a) { sub256(r, N_, a); }

/* ecc.c:205-253 computes a*b mod n with two Montgomery passes (x R^-1, then x R^2 R^-1) and a
   conditional subtract after each, i.e. the canonical product for canonical inputs.  Restated as the
   exact 512-bit product followed by binary long division by n. */
void orc_sn_mul(orc_fe r, const orc_fe a, const orc_fe b) {
  u64 w[8] = {0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 t = (u128)a[j] * b[i] + w[i + j] + carry;
      w[i + j] = (u64)t;
      carry = (u64)(t >> 64);
    }
    w[i + 4] = carry;
  }
  orc_fe rem = {0, 0, 0, 0};
  for (int bit = 511; bit >= 0; --bit) {
    u64 top = rem[3] >> 63;
    for (int i = 3; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
    rem[0] = (rem[0] << 1) | ((w[bit / 64] >> (bit % 64)) & 1);
    if (top || fe_cmp(rem, N_) >= 0) sub256(rem, rem, N_);
  }
  fe_cpy(r, rem);
}
/* ecc.c:255-260 */
void orc_sn_add_stride(orc_fe r, const orc_fe base, const orc_fe stride, uint64_t off) {
  orc_fe t;
  fe_set(t, off);
  orc_sn_mul(t, t, stride);
  orc_sn_add(r, t, base);
}
/* ecc.c:81-95,262-265: hex digits are consumed right-to-left, other characters are skipped.
   The reference has no bound check past 64 digits (ecc.c:92 writes out of bounds); the oracle stops there. */
void orc_sn_from_hex(orc_fe r, const char *hex) {
  fe_set(r, 0);
  int cnt = 0;
  for (long i = (long)strlen(hex) - 1; i >= 0 && cnt < 64; --i) {
    int c = hex[i];
    u64 v;
    if (c >= '0' && c <= '9') v = c - '0';
    else if (c >= 'a' && c <= 'f') v = c - 'a' + 10;
    else if (c >= 'A' && c <= 'F') v = c - 'A' + 10;
    else continue;
    r[cnt / 16] |= v << (cnt * 4 % 64);
    cnt++;
  }
  if (fe_cmp(r, N_) >= 0) orc_sn_sub(r, r, N_);
}

/* main.c:267-276 */
void orc_calc_priv(orc_fe pk, const orc_fe start, const orc_fe stride, uint64_t off, uint8_t endo) {
  orc_sn_add_stride(pk, start, stride, off);
  if (endo == 0) return;
  if (endo == 1) orc_sn_neg(pk, pk);
  if (endo == 2 || endo == 3) orc_sn_mul(pk, pk, LAM1);
  if (endo == 3) orc_sn_neg(pk, pk);
  if (endo == 4 || endo == 5) orc_sn_mul(pk, pk, LAM2);
  if (endo == 5) orc_sn_neg(pk, pk);
}

/* ---------------------------------------------------------------- curve, homogeneous projective (x=X/Z, y=Y/Z) */

/* ecc.c:611-646 (_ec_jacobi_dbl1): W=3X^2, S=YZ, B=XYS, H=W^2-8B, X'=2HS, Y'=W(4B-H)-8Y^2S^2, Z'=8S^3 */
void orc_pt_dbl(orc_pt *r, const orc_pt *p) {
  orc_fe w, s, b, h, t, y2, s2, b4, b8, ry;
  orc_fp_sqr(t, p->x);
  orc_fp_add(w, t, t);
  orc_fp_add(w, w, t);
  orc_fp_mul(s, p->y, p->z);
  orc_fp_mul(b, p->x, p->y);
  orc_fp_mul(b, b, s);
  orc_fp_add(b4, b, b);
  orc_fp_add(b4, b4, b4);
  orc_fp_add(b8, b4, b4);
  orc_fp_sqr(h, w);
  orc_fp_sub(h, h, b8);
  orc_fp_sqr(y2, p->y);
  orc_fp_sqr(s2, s);
  orc_fp_mul(r->x, h, s);
  orc_fp_add(r->x, r->x, r->x);
  orc_fp_sub(t, b4, h);
  orc_fp_mul(t, w, t);
  orc_fp_mul(ry, y2, s2);
  for (int i = 0; i < 3; ++i) orc_fp_add(ry, ry, ry);
  orc_fp_sub(r->y, t, ry);
  orc_fp_mul(r->z, s2, s);
  for (int i = 0; i < 3; ++i) orc_fp_add(r->z, r->z, r->z);
}

/* ecc.c:648-684 (_ec_jacobi_add1), 12M+2S; the equal-x case is not handled (the reference asserts). */
void orc_pt_add(orc_pt *r, const orc_pt *p, const orc_pt *q) {
  orc_fe u2, v2, u, v, w, a, vs, vc, t;
  orc_fp_mul(u2, p->y, q->z);
  orc_fp_mul(v2, p->x, q->z);
  orc_fp_mul(u, q->y, p->z);
  orc_fp_mul(v, q->x, p->z);
  orc_fp_mul(w, p->z, q->z);
  orc_fp_sub(u, u, u2);
  orc_fp_sub(v, v, v2);
  orc_fp_sqr(vs, v);
  orc_fp_mul(vc, vs, v);
  orc_fp_mul(vs, vs, v2);
  orc_fp_mul(r->z, vc, w);
  orc_fp_sqr(a, u);
  orc_fp_mul(a, a, w);
  orc_fp_add(t, vs, vs);
  orc_fp_sub(a, a, vc);
  orc_fp_sub(a, a, t);
  orc_fp_mul(r->x, v, a);
  orc_fp_sub(a, vs, a);
  orc_fp_mul(a, a, u);
  orc_fp_mul(u, vc, u2);
  orc_fp_sub(r->y, a, u);
}

/* ecc.c:686-693 */
void orc_pt_rdc(orc_pt *r, const orc_pt *a) {
  orc_fe zi;
  orc_fp_inv(zi, a->z);
  orc_fp_mul(r->x, a->x, zi);
  orc_fp_mul(r->y, a->y, zi);
  fe_set(r->z, 1);
}

/* ecc.c:695-707 */
void orc_pt_grprdc(orc_pt *r, uint64_t n) {
  orc_fe *zz = (orc_fe *)malloc(n * sizeof(orc_fe));
  for (u64 i = 0; i < n; ++i) fe_cpy(zz[i], r[i].z);
  orc_fp_grpinv(zz, (u32)n);
  for (u64 i = 0; i < n; ++i) {
    orc_fp_mul(r[i].x, r[i].x, zz[i]);
    orc_fp_mul(r[i].y, r[i].y, zz[i]);
    fe_set(r[i].z, 1);
  }
  free(zz);
}

/* ecc.c:821-843: LSB-first double-and-add; "accumulator empty" is encoded as x[0]==0 && y[0]==0. */
void orc_pt_mul(orc_pt *r, const orc_pt *p, const orc_fe k) {
  orc_pt t = *p, acc;
  fe_set(acc.x, 0), fe_set(acc.y, 0), fe_set(acc.z, 1);
  unsigned bits = fe_bits(k);
  for (unsigned i = 0; i < bits; ++i) {
    if (k[i / 64] >> (i % 64) & 1) {
      if (acc.x[0] == 0 && acc.y[0] == 0) acc = t;
      else orc_pt_add(&acc, &acc, &t);
    }
    orc_pt_dbl(&t, &t);
  }
  *r = acc;
}

void orc_pt_mulg_affine(orc_fe x, orc_fe y, const orc_fe k) { /* ecc.c:850-853 with p = G1 */
  orc_pt r;
  orc_pt_mul(&r, &GEN, k);
  orc_pt_rdc(&r, &r);
  fe_cpy(x, r.x), fe_cpy(y, r.y);
}

/* ecc.c:860-872: y^2 - x^3 == 7 */
int orc_pt_on_curve(const orc_pt *p) {
  orc_pt q;
  orc_pt_rdc(&q, p);
  orc_fe y2, x3;
  orc_fp_sqr(y2, q.y);
  orc_fp_sqr(x3, q.x);
  orc_fp_mul(x3, x3, q.x);
  orc_fp_sub(y2, y2, x3);
  return y2[0] == 7 && !y2[1] && !y2[2] && !y2[3];
}

/* ecc.c:876-905: fixed-base window table, W=14 -> 19 windows of 16383 affine points,
   slot (2^14-1)*i + (b-1) = b * 2^(14 i) * G. */
#define GT_W 14u
#define GT_N (1u << GT_W)
#define GT_D ((255u / GT_W) + 1u)
static orc_pt *g_gtable = NULL;
static pthread_mutex_t g_gtable_lock = PTHREAD_MUTEX_INITIALIZER;

size_t orc_gtable_init(void) {
  pthread_mutex_lock(&g_gtable_lock);
  size_t slots = (size_t)GT_N * GT_D - GT_D;
  if (!g_gtable) {
    orc_pt *tb = (orc_pt *)malloc(slots * sizeof(orc_pt));
    orc_pt base = GEN, run;
    for (u32 i = 0; i < GT_D; ++i) {
      size_t at = (size_t)(GT_N - 1) * i;
      tb[at] = base;
      run = base;
      for (u32 j = 1; j < GT_N - 1; ++j) {
        if (j == 1) orc_pt_dbl(&run, &run);
        else orc_pt_add(&run, &run, &base);
        tb[at + j] = run;
      }
      orc_pt_add(&base, &run, &base); /* (2^14-1)b + b = 2^14 b */
    }
    orc_pt_grprdc(tb, slots);
    g_gtable = tb;
  }
  pthread_mutex_unlock(&g_gtable_lock);
  return slots * sizeof(orc_pt);
}

/* ecc.c:907-929: LSB-first 14-bit digits, zero digits skipped, accumulator empty <=> q.x == 0.
   Result is projective. k=0 gives z=0 (no guard, as in the reference). */
void orc_gtable_mul(orc_pt *r, const orc_fe k) {
  if (!g_gtable) orc_gtable_init();
  orc_pt q;
  memset(&q, 0, sizeof q);
  orc_fe kk;
  fe_cpy(kk, k);
  for (u32 i = 0; i < GT_D; ++i) {
    u64 digit = kk[0] & (GT_N - 1);
    kk[0] = (kk[0] >> GT_W) | (kk[1] << (64 - GT_W));
    kk[1] = (kk[1] >> GT_W) | (kk[2] << (64 - GT_W));
    kk[2] = (kk[2] >> GT_W) | (kk[3] << (64 - GT_W));
    kk[3] >>= GT_W;
    if (!digit) continue;
    const orc_pt *e = &g_gtable[(size_t)(GT_N - 1) * i + digit - 1];
    if (!(q.x[0] | q.x[1] | q.x[2] | q.x[3])) q = *e;
    else orc_pt_add(&q, &q, e);
  }
  *r = q;
}

/* ---------------------------------------------------------------- SHA-256 (FIPS 180-4), as sha256.c:399-453 uses it */

static const u32 SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline u32 ror(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
static inline u32 rol(u32 x, int n) { return (x << n) | (x >> (32 - n)); }

/* sha256_final (sha256.c:399): ALWAYS starts from the IV (the incoming state is ignored) and
   compresses len/64 already-padded blocks; output = the 8 state words (big-endian word semantics). */
void orc_sha256_blocks(uint32_t st[8], const uint8_t *data, uint32_t len) {
  static const u32 IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(st, IV, sizeof IV);
  for (; len >= 64; len -= 64, data += 64) {
    u32 w[64], v[8];
    for (int i = 0; i < 16; ++i)
      w[i] = (u32)data[4 * i] << 24 | (u32)data[4 * i + 1] << 16 | (u32)data[4 * i + 2] << 8 | data[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      u32 s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
      u32 s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    memcpy(v, st, sizeof v);
    for (int i = 0; i < 64; ++i) {
      u32 t1 = v[7] + (ror(v[4], 6) ^ ror(v[4], 11) ^ ror(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + SHA_K[i] + w[i];
      u32 t2 = (ror(v[0], 2) ^ ror(v[0], 13) ^ ror(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
      memmove(v + 1, v, 7 * sizeof(u32));
      v[4] += t1;
      v[0] = t1 + t2;
    }
    for (int i = 0; i < 8; ++i) st[i] += v[i];
  }
}

/* ---------------------------------------------------------------- RIPEMD-160 single block (rmd160.c:46-130, rmd160s.c:122-336) */

static const u8 RL[80] = {0, 1, 2,  3,  4,  5,  6,  7, 8,  9,  10, 11, 12, 13, 14, 15, 7, 4,  13, 1, 10, 6,  15, 3,  12, 0, 9,
                          5, 2, 14, 11, 8,  3,  10, 14, 4, 9,  15, 8,  1,  2,  7,  0,  6, 13, 11, 5, 12, 1,  9,  11, 10, 0, 8,
                          12, 4, 13, 3, 7,  15, 14, 5,  6, 2,  4,  0,  5,  9,  7,  12, 2, 10, 14, 1, 3,  8,  11, 6,  15, 13};
static const u8 RR[80] = {5,  14, 7, 0, 9, 2,  11, 4,  13, 6,  15, 8,  1, 10, 3,  12, 6, 11, 3,  7,  0, 13, 5,  10, 14, 15, 8,
                          12, 4,  9, 1, 2, 15, 5,  1,  3,  7,  14, 6,  9, 11, 8,  12, 2, 10, 0,  4,  13, 8, 6,  4,  1,  3,  11,
                          15, 0,  5, 12, 2, 13, 9,  7,  10, 14, 12, 15, 10, 4, 1,  5,  8, 7,  6,  2,  13, 14, 0,  3,  9,  11};
static const u8 SL[80] = {11, 14, 15, 12, 5,  8,  7,  9,  11, 13, 14, 15, 6,  7,  9,  8,  7,  6,  8,  13, 11, 9,  7,  15, 7, 12, 15,
                          9,  11, 7,  13, 12, 11, 13, 6,  7,  14, 9,  13, 15, 14, 8,  13, 6,  5,  12, 7,  5,  11, 12, 14, 15, 14, 15,
                          9,  8,  9,  14, 5,  6,  8,  6,  5,  12, 9,  15, 5,  11, 6,  8,  13, 12, 5,  12, 13, 14, 11, 8,  5,  6};
static const u8 SR[80] = {8,  9,  9,  11, 13, 15, 15, 5,  7,  7,  8,  11, 14, 14, 12, 6,  9,  13, 15, 7,  12, 8,  9,  11, 7,  7, 12,
                          7,  6,  15, 13, 11, 9,  7,  15, 11, 8,  6,  6,  14, 12, 13, 5,  14, 13, 13, 7,  5,  15, 5,  8,  11, 14, 14,
                          6,  14, 6,  9,  12, 9,  12, 5,  15, 8,  8,  5,  12, 9,  12, 5,  14, 6,  8,  13, 6,  5,  15, 13, 11, 11};
static const u32 KL[5] = {0, 0x5a827999, 0x6ed9eba1, 0x8f1bbcdc, 0xa953fd4e};
static const u32 KR[5] = {0x50a28be6, 0x5c4dd124, 0x6d703ef3, 0x7a6d76e9, 0};
static inline u32 rmd_f(int j, u32 x, u32 y, u32 z) {
  switch (j) {
  case 0: return x ^ y ^ z;
  case 1: return (x & y) | (~x & z);
  case 2: return (x | ~y) ^ z;
  case 3: return (x & z) | (y & ~z);
  default: return x ^ (y | ~z);
  }
}
/* One compression from the standard IV over 16 LITTLE-ENDIAN message words; out = the five state
   words in RIPEMD's native (little-endian) sense.  Byte order adaptation is done by the callers. */
void orc_rmd160_block(uint32_t out[5], const uint32_t x[16]) {
  static const u32 IV[5] = {0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476, 0xc3d2e1f0};
  u32 l[5], r[5];
  memcpy(l, IV, sizeof IV), memcpy(r, IV, sizeof IV);
  for (int i = 0; i < 80; ++i) {
    int rd = i / 16;
    u32 t = rol(l[0] + rmd_f(rd, l[1], l[2], l[3]) + x[RL[i]] + KL[rd], SL[i]) + l[4];
    l[0] = l[4], l[4] = l[3], l[3] = rol(l[2], 10), l[2] = l[1], l[1] = t;
    t = rol(r[0] + rmd_f(4 - rd, r[1], r[2], r[3]) + x[RR[i]] + KR[rd], SR[i]) + r[4];
    r[0] = r[4], r[4] = r[3], r[3] = rol(r[2], 10), r[2] = r[1], r[1] = t;
  }
  out[0] = IV[1] + l[2] + r[3];
  out[1] = IV[2] + l[3] + r[4];
  out[2] = IV[3] + l[4] + r[0];
  out[3] = IV[4] + l[0] + r[1];
  out[4] = IV[0] + l[1] + r[2];
}

/* ---------------------------------------------------------------- hash160 of a public key (addr.c:33-131) */

static void put_be(u8 *dst, const orc_fe v) { /* 32 bytes big-endian (addr.c:37-40) */
  for (int i = 0; i < 4; ++i)
    for (int b = 0; b < 8; ++b) dst[i * 8 + b] = (u8)(v[3 - i] >> (56 - 8 * b));
}
/* sha state (big-endian words) -> RIPEMD message (addr.c:69-73 / 108-111) -> h160 words where word k
   holds digest bytes 4k..4k+3 big-endian (rmd160.c:129, rmd160s.c:334) */
static void rmd_of_sha(u32 h[5], const u32 sha[8]) {
  u32 m[16] = {0}, o[5];
  for (int i = 0; i < 8; ++i) m[i] = __builtin_bswap32(sha[i]);
  m[8] = 0x80;
  m[14] = 256;
  orc_rmd160_block(o, m);
  for (int i = 0; i < 5; ++i) h[i] = __builtin_bswap32(o[i]);
}
/* addr.c:33-45,75-84: 02|03 || X, SHA padding baked in: 0x80 at [33], bit length 0x0108 at [62..63] */
void orc_hash160_33(uint32_t h[5], const orc_fe x, const orc_fe y) {
  u8 msg[64] = {0};
  u32 st[8];
  msg[0] = (y[0] & 1) ? 0x03 : 0x02;
  put_be(msg + 1, x);
  msg[33] = 0x80, msg[62] = 0x01, msg[63] = 0x08;
  orc_sha256_blocks(st, msg, 64);
  rmd_of_sha(h, st);
}
/* addr.c:47-67,86-95: 04 || X || Y, 0x80 at [65], bit length 0x0208 at [126..127] */
void orc_hash160_65(uint32_t h[5], const orc_fe x, const orc_fe y) {
  u8 msg[128] = {0};
  u32 st[8];
  msg[0] = 0x04;
  put_be(msg + 1, x);
  put_be(msg + 33, y);
  msg[65] = 0x80, msg[126] = 0x02, msg[127] = 0x08;
  orc_sha256_blocks(st, msg, 128);
  rmd_of_sha(h, st);
}

/* ---------------------------------------------------------------- bloom filter (utils.c:274-326) */

/* 20 probe indices: five overlapping 64-bit words a1..a5 of the hash, for S in {24,28,36,40} and
   j=1..5: idx = a_j << S | a_{j+1} >> S (a6 = a1).  Order: S outer, j inner (matters for early-out only). */
static void blf_indices(u64 idx[20], const u32 h[5]) {
  u64 a[6];
  a[0] = (u64)h[0] << 32 | h[1];
  a[1] = (u64)h[2] << 32 | h[3];
  a[2] = (u64)h[4] << 32 | h[0];
  a[3] = (u64)h[1] << 32 | h[2];
  a[4] = (u64)h[3] << 32 | h[4];
  a[5] = a[0];
  static const int S[4] = {24, 28, 36, 40};
  for (int s = 0; s < 4; ++s)
    for (int j = 0; j < 5; ++j) idx[s * 5 + j] = a[j] << S[s] | a[j + 1] >> S[s];
}
/* utils.c:282-288: word = idx mod (size*64) / 64, bit = idx mod 64 */
void orc_blf_add(uint64_t *bits, uint64_t size, const uint32_t h[5]) {
  u64 idx[20];
  blf_indices(idx, h);
  for (int i = 0; i < 20; ++i) bits[idx[i] % (size * 64) / 64] |= 1ULL << (idx[i] % 64);
}
int orc_blf_has(const uint64_t *bits, uint64_t size, const uint32_t h[5]) {
  u64 idx[20];
  blf_indices(idx, h);
  for (int i = 0; i < 20; ++i)
    if (!(bits[idx[i] % (size * 64) / 64] >> (idx[i] % 64) & 1)) return 0;
  return 1;
}
/* bulk forms for the tests (plain loops over the two functions above).  orc_blf_gen_many is the insert loop of
   blf-gen (utils.c:455-470): a hash already reported by blf_has is skipped and not counted. */
void orc_blf_add_many(uint64_t *bits, uint64_t size, const uint32_t *h, uint64_t n) {
  for (u64 i = 0; i < n; ++i) orc_blf_add(bits, size, h + i * 5);
}
void orc_blf_has_many(const uint64_t *bits, uint64_t size, const uint32_t *h, uint64_t n, uint8_t *hit) {
  for (u64 i = 0; i < n; ++i) hit[i] = (uint8_t)orc_blf_has(bits, size, h + i * 5);
}
uint64_t orc_blf_gen_many(uint64_t *bits, uint64_t size, const uint32_t *h, uint64_t n) {
  u64 count = 0;
  for (u64 i = 0; i < n; ++i) {
    if (orc_blf_has(bits, size, h + i * 5)) continue;
    orc_blf_add(bits, size, h + i * 5), count++;
  }
  return count;
}
/* utils.c:421-427: m = n*ln(1e-9)/ln(1/2^ln2) bits, size = ceil(m/64) words */
uint64_t orc_blf_gen_size(uint64_t n) {
  double p = 1.0 / (double)1000000000ULL;
  u64 m = (u64)(n * log(p) / log(1.0 / pow(2.0, log(2.0))));
  return (m + 63) / 64;
}

/* ---------------------------------------------------------------- filter: bloom [+ sorted list] */

static int cmp160(const void *a, const void *b) { /* addr.c:18-26 */
  const u32 *x = (const u32 *)a, *y = (const u32 *)b;
  for (int i = 0; i < 5; ++i)
    if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}
/* main.c:113-130: sort, dedup, bloom of 2*count words */
void orc_filter_from_list(orc_filter *f, const uint32_t *hashes, uint64_t n) {
  u32 *l = (u32 *)malloc((n ? n : 1) * 20);
  memcpy(l, hashes, n * 20);
  qsort(l, n, 20, cmp160);
  u64 uniq = 0;
  for (u64 i = 1; i < n; ++i)
    if (memcmp(l + uniq * 5, l + i * 5, 20) != 0) memcpy(l + (++uniq) * 5, l + i * 5, 20);
  f->list = l;
  f->list_count = n ? uniq + 1 : 0;
  f->size = f->list_count * 2;
  f->bits = (u64 *)calloc(f->size ? f->size : 1, sizeof(u64));
  for (u64 i = 0; i < f->list_count; ++i) orc_blf_add(f->bits, f->size, l + i * 5);
}
void orc_filter_from_bloom(orc_filter *f, const uint64_t *bits, uint64_t size) {
  f->list = NULL, f->list_count = 0, f->size = size;
  f->bits = (u64 *)malloc(size * sizeof(u64));
  memcpy(f->bits, bits, size * sizeof(u64));
}
void orc_filter_free(orc_filter *f) {
  free(f->bits), free(f->list);
  memset(f, 0, sizeof *f);
}
/* main.c:205-217 */
int orc_filter_check(const orc_filter *f, const uint32_t h[5]) {
  if (!orc_blf_has(f->bits, f->size, h)) return 0;
  if (!f->list) return 1;
  return bsearch(h, f->list, f->list_count, 20, cmp160) != NULL;
}

/* ---------------------------------------------------------------- cmd add (main.c:219-454) */

typedef struct add_state {
  const orc_add_cfg *cfg;
  const orc_filter *flt;
  orc_fe stride_k;  /* 2^ord_offs */
  orc_pt stride_p;  /* (2048*stride)*G */
  orc_pt *gp;       /* [2048]: (i+1)*stride*G for i<1024, then the negatives */
  orc_fe range_s, range_e, initial_s;
  u64 job_size;
  pthread_mutex_t lock;
  orc_found *out;
  u64 cap, nout, checked, hashed;
  int err;
} add_state;

/* main.c:219-246 */
static void precompute_gpoints(add_state *st) {
  fe_set(st->stride_k, 1);
  unsigned offs = st->cfg->ord_offs;
  orc_fe s = {0, 0, 0, 0};
  s[offs / 64] = 1ULL << (offs % 64);
  fe_cpy(st->stride_k, s);
  orc_fe t, zero = {0, 0, 0, 0};
  orc_sn_add_stride(t, zero, st->stride_k, GROUP);
  orc_pt_mul(&st->stride_p, &GEN, t);
  orc_pt_rdc(&st->stride_p, &st->stride_p);
  orc_pt g1, g2;
  orc_pt_mul(&g1, &GEN, st->stride_k);
  orc_pt_rdc(&g1, &g1);
  orc_pt_dbl(&g2, &g1);
  orc_pt_rdc(&g2, &g2);
  st->gp = (orc_pt *)malloc(GROUP * sizeof(orc_pt));
  st->gp[0] = g1, st->gp[1] = g2;
  for (u32 i = 2; i < HALF; ++i) {
    orc_pt_add(&st->gp[i], &st->gp[i - 1], &g1);
    orc_pt_rdc(&st->gp[i], &st->gp[i]);
  }
  for (u32 i = 0; i < HALF; ++i) {
    st->gp[HALF + i] = st->gp[i];
    orc_fp_neg(st->gp[HALF + i].y, st->gp[i].y);
  }
}

static void emit(add_state *st, int compressed, const u32 h[5], const orc_fe pk, u8 endo) {
  pthread_mutex_lock(&st->lock);
  if (st->nout < st->cap) {
    orc_found *f = &st->out[st->nout];
    memcpy(f->h160, h, 20);
    f->compressed = (u8)compressed, f->endo = endo, f->pad[0] = f->pad[1] = 0;
    memcpy(f->pk, pk, 32);
  } else st->err = -1;
  st->nout++;
  pthread_mutex_unlock(&st->lock);
}

/* main.c:278-285 + 248-263 */
static void check_one(add_state *st, int compressed, const u32 h[5], const orc_fe start_pk, u64 off, u8 endo) {
  if (!orc_filter_check(st->flt, h)) return;
  orc_fe pk;
  orc_calc_priv(pk, start_pk, st->stride_k, off, endo);
  if (st->cfg->verify) {
    orc_fe x, y;
    u32 hv[5];
    orc_pt_mulg_affine(x, y, pk);
    compressed ? orc_hash160_33(hv, x, y) : orc_hash160_65(hv, x, y);
    if (memcmp(hv, h, 20) != 0) st->err = -2;
  }
  emit(st, compressed, h, pk, endo);
}

/* main.c:287-347. HASH_BATCH_SIZE only groups hashing work; the emission order it induces is:
   all endo=0 hashes in key order (33 before 65 per key), then for every key its endo 1..5 variants. */
static void check_group(add_state *st, const orc_fe start_pk, const orc_pt *bp) {
  const orc_add_cfg *c = st->cfg;
  u32 h[5];
  for (u32 i = 0; i < GROUP; ++i) {
    if (c->check33) orc_hash160_33(h, bp[i].x, bp[i].y), check_one(st, 1, h, start_pk, i, 0);
    if (c->check65) orc_hash160_65(h, bp[i].x, bp[i].y), check_one(st, 0, h, start_pk, i, 0);
  }
  if (!c->use_endo) return;
  for (u32 k = 0; k < GROUP; ++k) {
    /* (x,-y) (bx,y) (bx,-y) (b2x,y) (b2x,-y): main.c:314-327 */
    orc_fe ny, bx, b2x;
    orc_fp_neg(ny, bp[k].y);
    orc_fp_mul(bx, bp[k].x, BET1);
    orc_fp_mul(b2x, bp[k].x, BET2);
    const u64 *xs[5] = {bp[k].x, bx, bx, b2x, b2x};
    const u64 *ys[5] = {ny, bp[k].y, ny, bp[k].y, ny};
    for (u8 e = 0; e < 5; ++e) {
      if (c->check33) orc_hash160_33(h, xs[e], ys[e]), check_one(st, 1, h, start_pk, k, e + 1);
      if (c->check65) orc_hash160_65(h, xs[e], ys[e]), check_one(st, 0, h, start_pk, k, e + 1);
    }
  }
}

/* main.c:349-403 */
static void batch_add(add_state *st, const orc_fe pk, u64 iterations) {
  orc_pt *bp = (orc_pt *)malloc(GROUP * sizeof(orc_pt));
  orc_fe *dx = (orc_fe *)malloc(HALF * sizeof(orc_fe));
  orc_pt c; /* group centre */
  orc_fe ck, ss, lam, rx, ry, dd;
  orc_sn_add_stride(ss, pk, st->stride_k, HALF);
  orc_pt_mul(&c, &GEN, ss);
  orc_pt_rdc(&c, &c);
  fe_cpy(ck, pk);
  for (u64 done = 0; done < iterations; done += GROUP) {
    for (u32 i = 0; i < HALF; ++i) orc_fp_sub(dx[i], st->gp[i].x, c.x);
    orc_fp_grpinv(dx, HALF);
    bp[HALF] = c;
    for (int side = 0; side < 2; ++side) {
      u32 base = side == 0 ? 0 : HALF, cnt = side == 0 ? HALF - 1 : HALF;
      for (u32 i = 0; i < cnt; ++i) {
        const orc_pt *g = &st->gp[base + i];
        orc_fp_sub(ss, g->y, c.y);
        orc_fp_mul(lam, ss, dx[i]);
        orc_fp_sqr(rx, lam);
        orc_fp_sub(rx, rx, c.x);
        orc_fp_sub(rx, rx, g->x);
        orc_fp_sub(dd, c.x, rx);
        orc_fp_mul(dd, lam, dd);
        orc_fp_sub(ry, dd, c.y);
        u32 idx = side == 0 ? HALF + i + 1 : HALF - 1 - i;
        fe_cpy(bp[idx].x, rx), fe_cpy(bp[idx].y, ry), fe_set(bp[idx].z, 1);
      }
    }
    check_group(st, ck, bp);
    orc_sn_add_stride(ck, ck, st->stride_k, GROUP);
    orc_pt_add(&c, &c, &st->stride_p);
    orc_pt_rdc(&c, &c);
    pthread_mutex_lock(&st->lock);
    st->hashed += GROUP;
    pthread_mutex_unlock(&st->lock);
  }
  free(bp), free(dx);
}

/* main.c:405-435 */
static void *add_worker(void *arg) {
  add_state *st = (add_state *)arg;
  orc_fe inc, pk;
  fe_set(inc, st->job_size);
  orc_sn_mul(inc, inc, st->stride_k);
  for (;;) {
    pthread_mutex_lock(&st->lock);
    int wrapped = fe_cmp(st->range_s, st->initial_s) < 0;
    if (fe_cmp(st->range_s, st->range_e) >= 0 || wrapped || st->err) {
      pthread_mutex_unlock(&st->lock);
      break;
    }
    fe_cpy(pk, st->range_s);
    orc_sn_add(st->range_s, st->range_s, inc);
    pthread_mutex_unlock(&st->lock);
    batch_add(st, pk, st->job_size);
    pthread_mutex_lock(&st->lock);
    st->checked += st->cfg->use_endo ? st->job_size * 6 : st->job_size; /* QUIRK main.c:431 */
    pthread_mutex_unlock(&st->lock);
  }
  return NULL;
}

/* main.c:437-454 */
int orc_add_range(const orc_add_cfg *cfg, const orc_filter *flt, const orc_fe range_s, const orc_fe range_e,
                  orc_found *out, uint64_t cap, uint64_t *nout, uint64_t *checked, uint64_t *hashed) {
  add_state st;
  memset(&st, 0, sizeof st);
  st.cfg = cfg, st.flt = flt, st.out = out, st.cap = cap;
  pthread_mutex_init(&st.lock, NULL);
  fe_cpy(st.range_s, range_s), fe_cpy(st.range_e, range_e), fe_cpy(st.initial_s, range_s);
  precompute_gpoints(&st);
  orc_fe span;
  orc_sn_sub(span, range_e, range_s);
  st.job_size = (span[1] | span[2] | span[3]) == 0 && span[0] < MAX_JOB ? span[0] : MAX_JOB; /* main.c:442 */
  if (cfg->rnd_jobs) st.job_size = MAX_JOB; /* cmd_rnd runs the same workers with full-size jobs (main.c:624,645-651) */
  int nt = cfg->threads < 1 ? 1 : cfg->threads;
  if (nt == 1) add_worker(&st);
  else {
    pthread_t *th = (pthread_t *)malloc(nt * sizeof(pthread_t));
    for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, add_worker, &st);
    for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
    free(th);
  }
  free(st.gp);
  if (nout) *nout = st.nout;
  if (checked) *checked = st.checked;
  if (hashed) *hashed = st.hashed;
  return st.err;
}

/* ---------------------------------------------------------------- cmd mul body (main.c:458-479, 530-534) */

int orc_mul_batch(int check33, int check65, const orc_filter *flt, const orc_fe *pk, uint64_t n, orc_found *out,
                  uint64_t cap, uint64_t *nout) {
  orc_pt *cp = (orc_pt *)malloc((n ? n : 1) * sizeof(orc_pt));
  for (u64 i = 0; i < n; ++i) orc_gtable_mul(&cp[i], pk[i]);
  orc_pt_grprdc(cp, n);
  u64 cnt = 0;
  int err = 0;
  u32 h[5];
  for (u64 i = 0; i < n; ++i) {
    for (int pass = 0; pass < 2; ++pass) {
      int compressed = pass == 0;
      if (compressed ? !check33 : !check65) continue;
      compressed ? orc_hash160_33(h, cp[i].x, cp[i].y) : orc_hash160_65(h, cp[i].x, cp[i].y);
      if (!orc_filter_check(flt, h)) continue;
      if (cnt < cap) {
        orc_found *f = &out[cnt];
        memcpy(f->h160, h, 20);
        f->compressed = (u8)compressed, f->endo = 0, f->pad[0] = f->pad[1] = 0;
        memcpy(f->pk, pk[i], 32);
      } else err = -1;
      cnt++;
    }
  }
  free(cp);
  if (nout) *nout = cnt;
  return err;
}

/* ---------------------------------------------------------------- cmd mul at scale: what every input line's hash160 must be
   main.c:486-540 as a table: jobs of 2048 scalars (MAX_LINE_SIZE / GROUP, main.c:556-566), each one ec_gtable_mul per scalar
   (ecc.c:907-929) + ONE ec_jacobi_grprdc per job (ecc.c:695-707) + addr33 / addr65 (addr.c:99-131); worker threads pull jobs from a
   counter as cmd_mul_worker does from its queue (main.c:486-501).  A scalar that is 0 (mod n) has no point (z = 0; the reference lets it
   poison its job): ok[i] = 0, its slot is left out of the job's shared inversion, its hashes are zeroed. */
typedef struct mul_many_state {
  const orc_fe *pk;
  u64 n, next;
  u32 *h33, *h65;
  u8 *ok;
  pthread_mutex_t lock;
} mul_many_state;

static void *mul_many_worker(void *arg) {
  mul_many_state *st = (mul_many_state *)arg;
  orc_pt *cp = (orc_pt *)malloc(GROUP * sizeof(orc_pt));
  u8 *inf = (u8 *)malloc(GROUP);
  for (;;) {
    pthread_mutex_lock(&st->lock);
    u64 at = st->next;
    st->next += GROUP;
    pthread_mutex_unlock(&st->lock);
    if (at >= st->n) break;
    u64 cnt = st->n - at < GROUP ? st->n - at : GROUP;
    for (u64 i = 0; i < cnt; ++i) {
      orc_gtable_mul(&cp[i], st->pk[at + i]);
      inf[i] = !(cp[i].z[0] | cp[i].z[1] | cp[i].z[2] | cp[i].z[3]);
      if (inf[i]) fe_set(cp[i].z, 1); /* keep it out of the shared inversion */
    }
    orc_pt_grprdc(cp, cnt);
    for (u64 i = 0; i < cnt; ++i) {
      if (st->ok) st->ok[at + i] = !inf[i];
      if (st->h33) {
        if (inf[i]) memset(st->h33 + 5 * (at + i), 0, 20);
        else orc_hash160_33(st->h33 + 5 * (at + i), cp[i].x, cp[i].y);
      }
      if (st->h65) {
        if (inf[i]) memset(st->h65 + 5 * (at + i), 0, 20);
        else orc_hash160_65(st->h65 + 5 * (at + i), cp[i].x, cp[i].y);
      }
    }
  }
  free(cp), free(inf);
  return NULL;
}

void orc_mul_hash160_many(const orc_fe *pk, uint64_t n, uint32_t *h33, uint32_t *h65, uint8_t *ok, int threads) {
  orc_gtable_init();
  mul_many_state st;
  memset(&st, 0, sizeof st);
  st.pk = pk, st.n = n, st.h33 = h33, st.h65 = h65, st.ok = ok;
  pthread_mutex_init(&st.lock, NULL);
  int nt = threads < 1 ? 1 : threads;
  pthread_t *th = (pthread_t *)malloc(nt * sizeof(pthread_t));
  for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, mul_many_worker, &st);
  for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
  free(th);
  pthread_mutex_destroy(&st.lock);
}
