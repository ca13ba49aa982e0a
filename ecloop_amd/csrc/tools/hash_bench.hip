// hash_bench.hip — does a TWO-STREAM hash160 (two independent keys per lane, rounds interleaved so that the double-rate
// opcodes of both streams sit next to each other) issue faster on gfx950 than the one-stream code k_add runs today?
//
// Round-2 review item 3.  Background (profiles/ubench_r02.txt): add / sub / logic / v_bitop3 issue in ~2.4 clocks per
// wave64 instruction in LONG runs but cost ~4.2 like every other opcode when they stand alone between rotates and
// v_add3 - which is how they occur in SHA-256 / RIPEMD-160 (runs of 1-3).  Two streams double every run.
//
// What is timed: hash160 of the compressed key only (hash160.h, the code the add kernel inlines), the x words of the
// next key derived from the previous hash so nothing can be hoisted; between hashes every wave does a block of
// v_mad_u64_u32 work standing in for the curve arithmetic (about the add kernel's proportion: ~280 of ~3100
// instructions per key).  `drift`: the mad block's length varies per iteration with the same total for every wave -
// either in step for all waves (0) or shifted by the wave's index (1), so that the waves of a SIMD are in different
// phases like the waves of the real kernel (whose groups drift apart; DESIGN.md "oversubscribed").
//   variants: one      - hash160_33, one key per lane per iteration           (the shipped code)
//             two      - two keys per lane, two calls, the compiler interleaves as it likes
//             two_grp  - two keys per lane, SHA-256 rounds written for both streams with the rotates, the Boolean /
//                        add run and the v_add3 tail fenced into groups (sched_barrier), RIPEMD's four lines likewise;
//                        also with only the rotates fenced, and with no fence at all (interleaved source)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 hash_bench.hip -o hash_bench && ./hash_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "../hash160.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// grouping level G (template parameter of the two-stream code): 1 = fence after the rotates, after the Boolean / add run and
// after the sums; 2 = only after the rotates; 3 = no fence (source interleaved, the compiler orders)
#define FENCE_A() do { if (G <= 2) __builtin_amdgcn_sched_barrier(0); } while (0)
#define FENCE_B() do { if (G <= 1) __builtin_amdgcn_sched_barrier(0); } while (0)

// ---------------------------------------------------------------- two streams, grouped
// SHA-256 round for both streams: 12 rotates | 8 Boolean ops | the sums
#define S2_RND(a, b, c, d, e, f, g, h, k, wi)                                              \
  {                                                                                        \
    u32 r0[2], r1[2], r2[2], r3[2], r4[2], r5[2], s1[2], ch[2], s0[2], mj[2], t1[2];       \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                        \
      r0[s] = rotr32(e[s], 6), r1[s] = rotr32(e[s], 11), r2[s] = rotr32(e[s], 25);         \
      r3[s] = rotr32(a[s], 2), r4[s] = rotr32(a[s], 13), r5[s] = rotr32(a[s], 22);         \
    }                                                                                      \
    FENCE_A();                                                                               \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                        \
      s1[s] = XOR3(r0[s], r1[s], r2[s]), ch[s] = SHA_CH(e[s], f[s], g[s]);                 \
      s0[s] = XOR3(r3[s], r4[s], r5[s]), mj[s] = SHA_MAJ(a[s], b[s], c[s]);                \
    }                                                                                      \
    FENCE_B();                                                                               \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                        \
      t1[s] = h[s] + s1[s] + ch[s] + (k) + w[s][(wi) & 15];                                \
      d[s] += t1[s];                                                                       \
      h[s] = t1[s] + s0[s] + mj[s];                                                        \
    }                                                                                      \
    FENCE_B();                                                                               \
  }
// message schedule word i for both streams: 8 rotates | 4 shifts + 4 xor3 | sums
#define S2_EXP(i)                                                                          \
  {                                                                                        \
    u32 q0[2], q1[2], q2[2], q3[2];                                                        \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                        \
      q0[s] = rotr32(w[s][((i)-2) & 15], 17), q1[s] = rotr32(w[s][((i)-2) & 15], 19);      \
      q2[s] = rotr32(w[s][((i)-15) & 15], 7), q3[s] = rotr32(w[s][((i)-15) & 15], 18);     \
    }                                                                                      \
    FENCE_A();                                                                               \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                        \
      u32 x1 = XOR3(q0[s], q1[s], w[s][((i)-2) & 15] >> 10);                               \
      u32 x0 = XOR3(q2[s], q3[s], w[s][((i)-15) & 15] >> 3);                               \
      w[s][(i) & 15] += x1 + w[s][((i)-7) & 15] + x0;                                      \
    }                                                                                      \
    FENCE_B();                                                                               \
  }

template <int G>
__device__ __forceinline__ void sha256_compress_x2(u32 st[2][8], u32 w[2][16]) {
  u32 a[2], b[2], c[2], d[2], e[2], f[2], g[2], h[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) a[s] = st[s][0], b[s] = st[s][1], c[s] = st[s][2], d[s] = st[s][3], e[s] = st[s][4], f[s] = st[s][5], g[s] = st[s][6], h[s] = st[s][7];
#pragma unroll
  for (int i = 0; i < 64; i += 8) {
    if (i >= 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) S2_EXP(i + j);
    }
    S2_RND(a, b, c, d, e, f, g, h, SHA256_K[i + 0], i + 0);
    S2_RND(h, a, b, c, d, e, f, g, SHA256_K[i + 1], i + 1);
    S2_RND(g, h, a, b, c, d, e, f, SHA256_K[i + 2], i + 2);
    S2_RND(f, g, h, a, b, c, d, e, SHA256_K[i + 3], i + 3);
    S2_RND(e, f, g, h, a, b, c, d, SHA256_K[i + 4], i + 4);
    S2_RND(d, e, f, g, h, a, b, c, SHA256_K[i + 5], i + 5);
    S2_RND(c, d, e, f, g, h, a, b, SHA256_K[i + 6], i + 6);
    S2_RND(b, c, d, e, f, g, h, a, SHA256_K[i + 7], i + 7);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) st[s][0] += a[s], st[s][1] += b[s], st[s][2] += c[s], st[s][3] += d[s], st[s][4] += e[s], st[s][5] += f[s], st[s][6] += g[s], st[s][7] += h[s];
}

// RIPEMD-160: left and right line of both streams = four independent chains, stepped together.
// One step of a chain is f -> a+f+x(+k) -> rotl -> +e, and rotl(c, 10): the four chains' Boolean ops and adds are grouped.
struct rmd_line { u32 a, b, c, d, e; };
#define R4_STEP(L, A, B, C, D, E, fnL, xL, kL, sL, fnR, xR, kR, sR)                                          \
  {                                                                                                          \
    u32 fl[2], fr[2], tl[2], tr[2];                                                                          \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) fl[s] = fnL(L[s][0].B, L[s][0].C, L[s][0].D), fr[s] = fnR(L[s][1].B, L[s][1].C, L[s][1].D); \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) tl[s] = L[s][0].A + fl[s] + x[s][xL] + (kL), tr[s] = L[s][1].A + fr[s] + x[s][xR] + (kR); \
    FENCE_A();                                                                                                \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                          \
      tl[s] = rotl32(tl[s], sL), tr[s] = rotl32(tr[s], sR);                                                  \
      L[s][0].C = rotl32(L[s][0].C, 10), L[s][1].C = rotl32(L[s][1].C, 10);                                  \
    }                                                                                                        \
    FENCE_B();                                                                                                \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) L[s][0].A = tl[s] + L[s][0].E, L[s][1].A = tr[s] + L[s][1].E; \
  }
// message word order and rotations of the two lines (RIPEMD-160 specification; same tables as lib/rmd160.c:46-130)
template <int G>
__device__ __forceinline__ void rmd160_x2(u32 out[2][5], const u32 x[2][16]) {
  const u32 h0 = 0x67452301, h1 = 0xefcdab89, h2 = 0x98badcfe, h3 = 0x10325476, h4 = 0xc3d2e1f0;
  rmd_line L[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int l = 0; l < 2; ++l) L[s][l].a = h0, L[s][l].b = h1, L[s][l].c = h2, L[s][l].d = h3, L[s][l].e = h4;
  static constexpr int RL[80] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 7, 4, 13, 1, 10, 6, 15, 3, 12, 0, 9, 5, 2, 14, 11, 8,
                                 3, 10, 14, 4, 9, 15, 8, 1, 2, 7, 0, 6, 13, 11, 5, 12, 1, 9, 11, 10, 0, 8, 12, 4, 13, 3, 7, 15, 14, 5, 6, 2,
                                 4, 0, 5, 9, 7, 12, 2, 10, 14, 1, 3, 8, 11, 6, 15, 13};
  static constexpr int RR[80] = {5, 14, 7, 0, 9, 2, 11, 4, 13, 6, 15, 8, 1, 10, 3, 12, 6, 11, 3, 7, 0, 13, 5, 10, 14, 15, 8, 12, 4, 9, 1, 2,
                                 15, 5, 1, 3, 7, 14, 6, 9, 11, 8, 12, 2, 10, 0, 4, 13, 8, 6, 4, 1, 3, 11, 15, 0, 5, 12, 2, 13, 9, 7, 10, 14,
                                 12, 15, 10, 4, 1, 5, 8, 7, 6, 2, 13, 14, 0, 3, 9, 11};
  static constexpr int SL[80] = {11, 14, 15, 12, 5, 8, 7, 9, 11, 13, 14, 15, 6, 7, 9, 8, 7, 6, 8, 13, 11, 9, 7, 15, 7, 12, 15, 9, 11, 7, 13, 12,
                                 11, 13, 6, 7, 14, 9, 13, 15, 14, 8, 13, 6, 5, 12, 7, 5, 11, 12, 14, 15, 14, 15, 9, 8, 9, 14, 5, 6, 8, 6, 5, 12,
                                 9, 15, 5, 11, 6, 8, 13, 12, 5, 12, 13, 14, 11, 8, 5, 6};
  static constexpr int SR[80] = {8, 9, 9, 11, 13, 15, 15, 5, 7, 7, 8, 11, 14, 14, 12, 6, 9, 13, 15, 7, 12, 8, 9, 11, 7, 7, 12, 7, 6, 15, 13, 11,
                                 9, 7, 15, 11, 8, 6, 6, 14, 12, 13, 5, 14, 13, 13, 7, 5, 15, 5, 8, 11, 14, 14, 6, 14, 6, 9, 12, 9, 12, 5, 15, 8,
                                 8, 5, 12, 9, 12, 5, 14, 6, 8, 13, 6, 5, 15, 13, 11, 11};
#define R4_ROUND(base, fnL, kL, fnR, kR)                                                                        \
  _Pragma("unroll") for (int j = (base); j < (base) + 16; ++j) {                                                \
    switch (j % 5) {                                                                                            \
    case 0: R4_STEP(L, a, b, c, d, e, fnL, RL[j], kL, SL[j], fnR, RR[j], kR, SR[j]); break;                     \
    case 1: R4_STEP(L, e, a, b, c, d, fnL, RL[j], kL, SL[j], fnR, RR[j], kR, SR[j]); break;                     \
    case 2: R4_STEP(L, d, e, a, b, c, fnL, RL[j], kL, SL[j], fnR, RR[j], kR, SR[j]); break;                     \
    case 3: R4_STEP(L, c, d, e, a, b, fnL, RL[j], kL, SL[j], fnR, RR[j], kR, SR[j]); break;                     \
    default: R4_STEP(L, b, c, d, e, a, fnL, RL[j], kL, SL[j], fnR, RR[j], kR, SR[j]); break;                    \
    }                                                                                                           \
  }
  R4_ROUND(0, RMD_F1, 0u, RMD_F5, 0x50a28be6u)
  R4_ROUND(16, RMD_F2, 0x5a827999u, RMD_F4, 0x5c4dd124u)
  R4_ROUND(32, RMD_F3, 0x6ed9eba1u, RMD_F3, 0x6d703ef3u)
  R4_ROUND(48, RMD_F4, 0x8f1bbcdcu, RMD_F2, 0x7a6d76e9u)
  R4_ROUND(64, RMD_F5, 0xa953fd4eu, RMD_F1, 0u)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    out[s][0] = h1 + L[s][0].c + L[s][1].d;
    out[s][1] = h2 + L[s][0].d + L[s][1].e;
    out[s][2] = h3 + L[s][0].e + L[s][1].a;
    out[s][3] = h4 + L[s][0].a + L[s][1].b;
    out[s][4] = h0 + L[s][0].b + L[s][1].c;
  }
}

template <int G>
__device__ __forceinline__ void hash160_33_x2(u32 h[2][5], const u32 x[2][8], const u32 par[2]) {
  u32 w[2][16], st[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    w[s][0] = ((0x02u | (par[s] & 1u)) << 24) | (x[s][7] >> 8);
#pragma unroll
    for (int i = 1; i < 8; ++i) w[s][i] = (x[s][8 - i] << 24) | (x[s][7 - i] >> 8);
    w[s][8] = (x[s][0] << 24) | 0x00800000u;
#pragma unroll
    for (int i = 9; i < 15; ++i) w[s][i] = 0;
    w[s][15] = 33 * 8;
    sha256_init(st[s]);
  }
  sha256_compress_x2<G>(st, w);
  u32 m[2][16], o[2][5];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) m[s][i] = bswap32(st[s][i]);
    m[s][8] = 0x80u;
#pragma unroll
    for (int i = 9; i < 16; ++i) m[s][i] = 0;
    m[s][14] = 256u;
  }
  rmd160_x2<G>(o, m);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 5; ++i) h[s][i] = bswap32(o[s][i]);
}

// ---------------------------------------------------------------- the timed kernels
// stand-in for the curve arithmetic between hashes: n dependent-free v_mad_u64_u32 on 4 accumulators
__device__ __forceinline__ void mad_block(u64 acc[4], u32 m, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += (u64)(u32)acc[(j + 1) & 3] * m;
  }
}
#define MAD_UNIT 35  /* x 4 mads = 140 per unit; schedule {0,1,2,5} units averages 280 mads per hash */

template <int MODE, int OCC>
__global__ void __launch_bounds__(256, OCC) k_hash(u32* out, u32 seed, int iters, int drift) {
  const u32 wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  u32 x[2][8], par[2] = {seed & 1u, (seed >> 1) & 1u};
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) x[s][i] = (seed + threadIdx.x * 0x9E3779B9u + blockIdx.x) * (2 * i + 3 + 16 * s);
  u64 acc[4] = {seed, seed * 3ull, seed * 5ull, seed * 7ull};
  u32 sum = 0;
  static constexpr int SCHED[4] = {0, 1, 2, 5};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int units = SCHED[(it + (drift ? wave : 0)) & 3];
    if (MODE == 0) {
      u32 h[5];
#pragma unroll 1
      for (int s = 0; s < 2; ++s) {  // two keys, one after the other (the shipped `which` loop)
        mad_block(acc, x[0][1] | 1u, units * MAD_UNIT);
        hash160_33(h, x[0], par[0]);
#pragma unroll
        for (int i = 0; i < 5; ++i) x[0][i] ^= h[i], sum += h[i];
        x[0][5] += (u32)acc[0], par[0] ^= h[0];
      }
    } else {
      u32 h[2][5];
      mad_block(acc, x[0][1] | 1u, 2 * units * MAD_UNIT);
      if (MODE == 1) {
        hash160_33(h[0], x[0], par[0]);
        hash160_33(h[1], x[1], par[1]);
      } else {
        hash160_33_x2<MODE - 1>(h, x, par);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 5; ++i) x[s][i] ^= h[s][i], sum += h[s][i];
        x[s][5] += (u32)acc[s], par[s] ^= h[s][0];
      }
    }
  }
  if (sum == 0x12345678u && acc[0] == 1) out[0] = sum;
  if (iters < 0) out[threadIdx.x] = sum;
}

// correctness of the two-stream code against the one-stream code (same inputs)
__global__ void k_check(u32* out) {
  u32 x[2][8], par[2] = {threadIdx.x & 1u, (threadIdx.x >> 1) & 1u};
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) x[s][i] = (threadIdx.x * 0x9E3779B9u + 77u) * (2 * i + 3 + 16 * s) ^ (i << 29);
  u32 h2[2][5], h1[2][5];
  hash160_33_x2<1>(h2, x, par);
  hash160_33(h1[0], x[0], par[0]);
  hash160_33(h1[1], x[1], par[1]);
  u32 bad = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 5; ++i) bad |= h2[s][i] ^ h1[s][i];
  if (bad) atomicAdd(out, 1u);
}

typedef void (*kern_t)(u32*, u32, int, int);
int main(int argc, char** argv) {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  u32* out;
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMemset(out, 0, 4096));
  hipLaunchKernelGGL(k_check, dim3(4), dim3(256), 0, 0, out);
  u32 bad = 0;
  CHECK(hipMemcpy(&bad, out, 4, hipMemcpyDeviceToHost));
  printf("# %s, %d CUs; two-stream hash160 == one-stream hash160 on 1024 lanes x 2 keys: %s\n", p.gcnArchName, cus, bad ? "MISMATCH" : "yes");
  if (bad) return 1;
  const int iters = argc > 1 ? atoi(argv[1]) : 256;
  struct { const char* name; kern_t k; int occ; } es[] = {
      {"one stream (shipped), 4 waves/SIMD", k_hash<0, 4>, 4},
      {"one stream (shipped), 2 waves/SIMD", k_hash<0, 2>, 2},
      {"two streams, compiler order, 4 waves/SIMD", k_hash<1, 4>, 4},
      {"two streams, compiler order, 2 waves/SIMD", k_hash<1, 2>, 2},
      {"two streams, fenced groups, 4 waves/SIMD", k_hash<2, 4>, 4},
      {"two streams, fenced groups, 2 waves/SIMD", k_hash<2, 2>, 2},
      {"two streams, rotates fenced, 4 waves/SIMD", k_hash<3, 4>, 4},
      {"two streams, rotates fenced, 2 waves/SIMD", k_hash<3, 2>, 2},
      {"two streams, interleaved source, 4 waves/SIMD", k_hash<4, 4>, 4},
      {"two streams, interleaved source, 2 waves/SIMD", k_hash<4, 2>, 2},
      {"one stream (shipped), 1 wave/SIMD", k_hash<0, 1>, 1},
      {"two streams, fenced groups, 1 wave/SIMD", k_hash<2, 1>, 1},
      {"two streams, rotates fenced, 1 wave/SIMD", k_hash<3, 1>, 1},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-46s %8s %14s %14s\n", "variant", "drift", "G hash160/s", "vs shipped");
  double base[2] = {0, 0};
  for (int drift = 0; drift < 2; ++drift)
    for (auto& e : es) {
      const int blocks = cus * e.occ * 4;  // 4 rounds of resident blocks: the tail is short against the run
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, 8, drift);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters, drift);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double hashes = 2.0 * iters * blocks * 256.0;
      const double rate = hashes / (ms * 1e-3) / 1e9;
      if (base[drift] == 0) base[drift] = rate;
      printf("%-46s %8s %14.3f %13.1f%%\n", e.name, drift ? "shifted" : "in step", rate, 100.0 * (rate / base[drift] - 1.0));
    }
  return 0;
}
