// hash160.h — SHA-256 -> RIPEMD-160 of a SEC1-serialised secp256k1 public key, one lane = one key.
//
// Device replacement for the reference's addr33/addr65 (+ _batch) pipe: lib/addr.c:33-131 (serialisation
// with the SHA padding baked in), lib/sha256.c:399-453 (compression from the IV over pre-padded blocks),
// lib/rmd160s.c:122-336 / lib/rmd160.c:46-130 (one RIPEMD-160 compression, input/output byte order).
// Everything is straight-line 32-bit VALU work: rotates are v_alignbit_b32, Ch/Maj/F2/F4 are v_bfi_b32.
// Output convention = the reference's h160_t: word k holds digest bytes 4k..4k+3 big-endian
// (lib/addr.c:16), i.e. "%08x" x5 prints the usual hex.
#pragma once
#include "fe256.h"

#define H_FN __host__ __device__ __forceinline__

H_FN u32 rotr32(u32 x, int n) { return (x >> (n & 31)) | (x << ((32 - n) & 31)); }  // -> v_alignbit_b32
H_FN u32 rotl32(u32 x, int n) { return (x << (n & 31)) | (x >> ((32 - n) & 31)); }
H_FN u32 bswap32(u32 x) { return __builtin_bswap32(x); }
// bitfield select: (m & a) | (~m & b)  -> v_bfi_b32
H_FN u32 bsel(u32 m, u32 a, u32 b) { return (a & m) | (b & ~m); }

// three-input xor: one v_bitop3_b32 on the device (the compiler prefers two v_xor_b32; ECL_XOR3_BITOP3=0 keeps that)
#ifndef ECL_XOR3_BITOP3
#define ECL_XOR3_BITOP3 1
#endif
// Boolean functions of three words as ONE v_bitop3_b32 (truth table = f(0xF0, 0xCC, 0xAA)).  Written with the
// builtin because the compiler otherwise splits Ch / Maj / select into disjoint AND terms that it folds into the
// additions ((e&f) + (~e&g)), which costs more instructions than it saves.
#define XOR3_C(a, b, c) ((a) ^ (b) ^ (c))
#define BITSEL_C(m, a, b) (((a) & (m)) | ((b) & ~(m)))
#define MAJ3_C(a, b, c) (((a) & (b)) | ((c) & ((a) | (b))))
#define ORN_XOR_C(x, y, z) (((x) | ~(y)) ^ (z))
#if defined(__HIP_DEVICE_COMPILE__) && ECL_XOR3_BITOP3
// The builtin is opaque to constant folding: with all-constant inputs (IV state in the first rounds, the constant
// words of the padded message) it would be evaluated at run time - hoisted out of the loops, but then its result
// occupies a VGPR for the whole kernel.  Constant inputs take the plain C form, which folds to a literal.
#define ECL_ALLCONST(a, b, c) (__builtin_constant_p(a) && __builtin_constant_p(b) && __builtin_constant_p(c))
H_FN u32 xor3_(u32 a, u32 b, u32 c) { return ECL_ALLCONST(a, b, c) ? XOR3_C(a, b, c) : __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
H_FN u32 bitsel_(u32 m, u32 a, u32 b) { return ECL_ALLCONST(m, a, b) ? BITSEL_C(m, a, b) : __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }
H_FN u32 maj3_(u32 a, u32 b, u32 c) { return ECL_ALLCONST(a, b, c) ? MAJ3_C(a, b, c) : __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
H_FN u32 orn_xor_(u32 x, u32 y, u32 z) { return ECL_ALLCONST(x, y, z) ? ORN_XOR_C(x, y, z) : __builtin_amdgcn_bitop3_b32(x, y, z, 0x59); }
#define XOR3(a, b, c) xor3_(a, b, c)
#define BITSEL(m, a, b) bitsel_(m, a, b)   /* (m & a) | (~m & b) */
#define MAJ3(a, b, c) maj3_(a, b, c)       /* majority */
#define ORN_XOR(x, y, z) orn_xor_(x, y, z) /* (x | ~y) ^ z */
#else
#define XOR3(a, b, c) XOR3_C(a, b, c)
#define BITSEL(m, a, b) BITSEL_C(m, a, b)
#define MAJ3(a, b, c) MAJ3_C(a, b, c)
#define ORN_XOR(x, y, z) ORN_XOR_C(x, y, z)
#endif

// ---------------------------------------------------------------- SHA-256
#define SHA_S0(x) XOR3(rotr32(x, 2), rotr32(x, 13), rotr32(x, 22))
#define SHA_S1(x) XOR3(rotr32(x, 6), rotr32(x, 11), rotr32(x, 25))
#define SHA_s0(x) XOR3(rotr32(x, 7), rotr32(x, 18), ((x) >> 3))
#define SHA_s1(x) XOR3(rotr32(x, 17), rotr32(x, 19), ((x) >> 10))
#define SHA_CH(e, f, g) BITSEL(e, f, g)
#define SHA_MAJ(a, b, c) MAJ3(a, b, c)

#define SHA_RND(a, b, c, d, e, f, g, h, k, w)            \
  {                                                      \
    u32 t1 = h + SHA_S1(e) + SHA_CH(e, f, g) + (k) + (w); \
    u32 t2 = SHA_S0(a) + SHA_MAJ(a, b, c);               \
    d += t1;                                             \
    h = t1 + t2;                                         \
  }
#define SHA_EXP(w, i) (w[(i) & 15] += SHA_s1(w[((i) - 2) & 15]) + w[((i) - 7) & 15] + SHA_s0(w[((i) - 15) & 15]))

static constexpr u32 SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// one compression: st += F(st, w); w is consumed (schedule expanded in place). Fully unrolled so that
// the round constants become literals and constant message words fold away.
H_FN void sha256_compress(u32 st[8], u32 w[16]) {
  u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i += 8) {
    if (i >= 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) SHA_EXP(w, i + j);
    }
    SHA_RND(a, b, c, d, e, f, g, h, SHA256_K[i + 0], w[(i + 0) & 15]);
    SHA_RND(h, a, b, c, d, e, f, g, SHA256_K[i + 1], w[(i + 1) & 15]);
    SHA_RND(g, h, a, b, c, d, e, f, SHA256_K[i + 2], w[(i + 2) & 15]);
    SHA_RND(f, g, h, a, b, c, d, e, SHA256_K[i + 3], w[(i + 3) & 15]);
    SHA_RND(e, f, g, h, a, b, c, d, SHA256_K[i + 4], w[(i + 4) & 15]);
    SHA_RND(d, e, f, g, h, a, b, c, SHA256_K[i + 5], w[(i + 5) & 15]);
    SHA_RND(c, d, e, f, g, h, a, b, SHA256_K[i + 6], w[(i + 6) & 15]);
    SHA_RND(b, c, d, e, f, g, h, a, SHA256_K[i + 7], w[(i + 7) & 15]);
  }
  st[0] += a, st[1] += b, st[2] += c, st[3] += d, st[4] += e, st[5] += f, st[6] += g, st[7] += h;
}
H_FN void sha256_init(u32 st[8]) {
  st[0] = 0x6a09e667, st[1] = 0xbb67ae85, st[2] = 0x3c6ef372, st[3] = 0xa54ff53a;
  st[4] = 0x510e527f, st[5] = 0x9b05688c, st[6] = 0x1f83d9ab, st[7] = 0x5be0cd19;
}

// ---------------------------------------------------------------- RIPEMD-160, one block from the IV
#define RMD_F1(x, y, z) XOR3(x, y, z)
#define RMD_F2(x, y, z) BITSEL(x, y, z)
#define RMD_F3(x, y, z) ORN_XOR(x, y, z)
#define RMD_F4(x, y, z) BITSEL(z, x, y)
#define RMD_F5(x, y, z) ORN_XOR(y, z, x)
#define RMD_STEP(a, b, c, d, e, fn, x, k, s)    \
  {                                             \
    a = rotl32(a + fn(b, c, d) + (x) + (k), s) + e; \
    c = rotl32(c, 10);                          \
  }
// five steps rotate the roles a,b,c,d,e -> e,a,b,c,d
#define RMD_5(A, B, C, D, E, fn, k, x0, s0, x1, s1, x2, s2, x3, s3, x4, s4) \
  RMD_STEP(A, B, C, D, E, fn, x0, k, s0)                                    \
  RMD_STEP(E, A, B, C, D, fn, x1, k, s1)                                    \
  RMD_STEP(D, E, A, B, C, fn, x2, k, s2)                                    \
  RMD_STEP(C, D, E, A, B, fn, x3, k, s3)                                    \
  RMD_STEP(B, C, D, E, A, fn, x4, k, s4)

// x: 16 little-endian message words. out: the five chaining words in RIPEMD's native sense.
H_FN void rmd160_compress_iv(u32 out[5], const u32 x[16]) {
  const u32 h0 = 0x67452301, h1 = 0xefcdab89, h2 = 0x98badcfe, h3 = 0x10325476, h4 = 0xc3d2e1f0;
  u32 a = h0, b = h1, c = h2, d = h3, e = h4;
  // left line
  RMD_5(a, b, c, d, e, RMD_F1, 0u, x[0], 11, x[1], 14, x[2], 15, x[3], 12, x[4], 5)
  RMD_5(a, b, c, d, e, RMD_F1, 0u, x[5], 8, x[6], 7, x[7], 9, x[8], 11, x[9], 13)
  RMD_5(a, b, c, d, e, RMD_F1, 0u, x[10], 14, x[11], 15, x[12], 6, x[13], 7, x[14], 9)
  // step 16 finishes round 1; the role rotation continues across round boundaries
  RMD_STEP(a, b, c, d, e, RMD_F1, x[15], 0u, 8)
  RMD_5(e, a, b, c, d, RMD_F2, 0x5a827999u, x[7], 7, x[4], 6, x[13], 8, x[1], 13, x[10], 11)
  RMD_5(e, a, b, c, d, RMD_F2, 0x5a827999u, x[6], 9, x[15], 7, x[3], 15, x[12], 7, x[0], 12)
  RMD_5(e, a, b, c, d, RMD_F2, 0x5a827999u, x[9], 15, x[5], 9, x[2], 11, x[14], 7, x[11], 13)
  RMD_STEP(e, a, b, c, d, RMD_F2, x[8], 0x5a827999u, 12)
  RMD_5(d, e, a, b, c, RMD_F3, 0x6ed9eba1u, x[3], 11, x[10], 13, x[14], 6, x[4], 7, x[9], 14)
  RMD_5(d, e, a, b, c, RMD_F3, 0x6ed9eba1u, x[15], 9, x[8], 13, x[1], 15, x[2], 14, x[7], 8)
  RMD_5(d, e, a, b, c, RMD_F3, 0x6ed9eba1u, x[0], 13, x[6], 6, x[13], 5, x[11], 12, x[5], 7)
  RMD_STEP(d, e, a, b, c, RMD_F3, x[12], 0x6ed9eba1u, 5)
  RMD_5(c, d, e, a, b, RMD_F4, 0x8f1bbcdcu, x[1], 11, x[9], 12, x[11], 14, x[10], 15, x[0], 14)
  RMD_5(c, d, e, a, b, RMD_F4, 0x8f1bbcdcu, x[8], 15, x[12], 9, x[4], 8, x[13], 9, x[3], 14)
  RMD_5(c, d, e, a, b, RMD_F4, 0x8f1bbcdcu, x[7], 5, x[15], 6, x[14], 8, x[5], 6, x[6], 5)
  RMD_STEP(c, d, e, a, b, RMD_F4, x[2], 0x8f1bbcdcu, 12)
  RMD_5(b, c, d, e, a, RMD_F5, 0xa953fd4eu, x[4], 9, x[0], 15, x[5], 5, x[9], 11, x[7], 6)
  RMD_5(b, c, d, e, a, RMD_F5, 0xa953fd4eu, x[12], 8, x[2], 13, x[10], 12, x[14], 5, x[1], 12)
  RMD_5(b, c, d, e, a, RMD_F5, 0xa953fd4eu, x[3], 13, x[8], 14, x[11], 11, x[6], 8, x[15], 5)
  RMD_STEP(b, c, d, e, a, RMD_F5, x[13], 0xa953fd4eu, 6)
  // after 80 steps the roles have rotated 80 mod 5 = 0 times: (a,b,c,d,e) are A..E again
  u32 al = a, bl = b, cl = c, dl = d, el = e;
  a = h0, b = h1, c = h2, d = h3, e = h4;
  // right line
  RMD_5(a, b, c, d, e, RMD_F5, 0x50a28be6u, x[5], 8, x[14], 9, x[7], 9, x[0], 11, x[9], 13)
  RMD_5(a, b, c, d, e, RMD_F5, 0x50a28be6u, x[2], 15, x[11], 15, x[4], 5, x[13], 7, x[6], 7)
  RMD_5(a, b, c, d, e, RMD_F5, 0x50a28be6u, x[15], 8, x[8], 11, x[1], 14, x[10], 14, x[3], 12)
  RMD_STEP(a, b, c, d, e, RMD_F5, x[12], 0x50a28be6u, 6)
  RMD_5(e, a, b, c, d, RMD_F4, 0x5c4dd124u, x[6], 9, x[11], 13, x[3], 15, x[7], 7, x[0], 12)
  RMD_5(e, a, b, c, d, RMD_F4, 0x5c4dd124u, x[13], 8, x[5], 9, x[10], 11, x[14], 7, x[15], 7)
  RMD_5(e, a, b, c, d, RMD_F4, 0x5c4dd124u, x[8], 12, x[12], 7, x[4], 6, x[9], 15, x[1], 13)
  RMD_STEP(e, a, b, c, d, RMD_F4, x[2], 0x5c4dd124u, 11)
  RMD_5(d, e, a, b, c, RMD_F3, 0x6d703ef3u, x[15], 9, x[5], 7, x[1], 15, x[3], 11, x[7], 8)
  RMD_5(d, e, a, b, c, RMD_F3, 0x6d703ef3u, x[14], 6, x[6], 6, x[9], 14, x[11], 12, x[8], 13)
  RMD_5(d, e, a, b, c, RMD_F3, 0x6d703ef3u, x[12], 5, x[2], 14, x[10], 13, x[0], 13, x[4], 7)
  RMD_STEP(d, e, a, b, c, RMD_F3, x[13], 0x6d703ef3u, 5)
  RMD_5(c, d, e, a, b, RMD_F2, 0x7a6d76e9u, x[8], 15, x[6], 5, x[4], 8, x[1], 11, x[3], 14)
  RMD_5(c, d, e, a, b, RMD_F2, 0x7a6d76e9u, x[11], 14, x[15], 6, x[0], 14, x[5], 6, x[12], 9)
  RMD_5(c, d, e, a, b, RMD_F2, 0x7a6d76e9u, x[2], 12, x[13], 9, x[9], 12, x[7], 5, x[10], 15)
  RMD_STEP(c, d, e, a, b, RMD_F2, x[14], 0x7a6d76e9u, 8)
  RMD_5(b, c, d, e, a, RMD_F1, 0u, x[12], 8, x[15], 5, x[10], 12, x[4], 9, x[1], 12)
  RMD_5(b, c, d, e, a, RMD_F1, 0u, x[5], 5, x[8], 14, x[7], 6, x[6], 8, x[2], 13)
  RMD_5(b, c, d, e, a, RMD_F1, 0u, x[13], 6, x[14], 5, x[0], 15, x[3], 13, x[9], 11)
  RMD_STEP(b, c, d, e, a, RMD_F1, x[11], 0u, 11)
  out[0] = h1 + cl + d;
  out[1] = h2 + dl + e;
  out[2] = h3 + el + a;
  out[3] = h4 + al + b;
  out[4] = h0 + bl + c;
}

// SHA-256 state (big-endian word semantics) -> h160 words (lib/addr.c:108-113 + lib/rmd160s.c:325-336)
H_FN void rmd160_of_sha(u32 h[5], const u32 st[8]) {
  u32 x[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = bswap32(st[i]);
  x[8] = 0x80u;
#pragma unroll
  for (int i = 9; i < 16; ++i) x[i] = 0;
  x[14] = 256u;
  u32 o[5];
  rmd160_compress_iv(o, x);
#pragma unroll
  for (int i = 0; i < 5; ++i) h[i] = bswap32(o[i]);
}

// hash160 of the compressed key: prefix = 0x02 | parity(y)  (lib/addr.c:33-45, 99-114)
// x, y: 8 canonical little-endian u32 words each (fe_to_words of a normalised element)
H_FN void hash160_33(u32 h[5], const u32 x[8], u32 y_parity) {
  u32 w[16], st[8];
  u32 prefix = 0x02u | (y_parity & 1u);
  w[0] = (prefix << 24) | (x[7] >> 8);
#pragma unroll
  for (int i = 1; i < 8; ++i) w[i] = (x[8 - i] << 24) | (x[7 - i] >> 8);
  w[8] = (x[0] << 24) | 0x00800000u;
#pragma unroll
  for (int i = 9; i < 15; ++i) w[i] = 0;
  w[15] = 33 * 8;
  sha256_init(st);
  sha256_compress(st, w);
  rmd160_of_sha(h, st);
}
// hash160 of the uncompressed key 04 || X || Y  (lib/addr.c:47-67, 116-131)
H_FN void hash160_65(u32 h[5], const u32 x[8], const u32 y[8]) {
  u32 w[16], st[8];
  w[0] = (0x04u << 24) | (x[7] >> 8);
#pragma unroll
  for (int i = 1; i < 8; ++i) w[i] = (x[8 - i] << 24) | (x[7 - i] >> 8);
  w[8] = (x[0] << 24) | (y[7] >> 8);
#pragma unroll
  for (int i = 1; i < 8; ++i) w[8 + i] = (y[8 - i] << 24) | (y[7 - i] >> 8);
  sha256_init(st);
  sha256_compress(st, w);
  w[0] = (y[0] << 24) | 0x00800000u;
#pragma unroll
  for (int i = 1; i < 15; ++i) w[i] = 0;
  w[15] = 65 * 8;
  sha256_compress(st, w);
  rmd160_of_sha(h, st);
}
