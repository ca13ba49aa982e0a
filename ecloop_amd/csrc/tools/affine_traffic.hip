// affine_traffic.hip - upper bound for a batched-affine `mul` (round-3 review, item 3), memory side only.  (round 4)
//
// The proposal: keep the window sums affine, park R accumulators per thread in HBM and share one inversion per addition step
// (5 M + 1 S + 270 / R per addition instead of the Jacobian 8 M + 3 S).  The cheapest data flow found for it (`mul`:
// binary tree over the 12 table points of a scalar - 6 + 3 + 1 + 1 additions, four inversions per thread instead of eleven -,
// backward pass of one level fused with the forward pass of the next, 36-byte limb form) still moves per scalar:
//   level 1 forward : 32 B scalar, 12 table points gathered (64 B each, random), 6 prefix products written (36 B each)
//   level 1 backward: scalar, 6 prefix products read, the 12 table points gathered AGAIN (x and y are needed now), 6 sums written
//                     (72 B each), 3 prefix products of level 2 written
//   level 2 backward: 3 prefix products read, 6 sums read, 3 sums written, 1 prefix product written
//   level 3 backward: 1 + 1 prefix products, 2 sums read, 1 sum written, 1 x coordinate read
//   level 4 backward: 1 prefix product, 2 sums read -> hash
// = 24 random 64-byte gathers + ~1.1 KB written + ~1.3 KB read in streams, against 12 gathers + 288 B for the Jacobian kernel.
// This program replays exactly those accesses (no arithmetic: XOR into a checksum) with the thread / scalar mapping the kernel
// would have (thread t owns scalars r * nt + t, planes strided by nt) and reports scalars per second.  Whatever the field
// arithmetic costs, a batched-affine `mul` cannot run faster than this.
// Build: hipcc --offload-arch=gfx950 -O3 affine_traffic.hip -o affine_traffic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32;
typedef unsigned long long u64;

__device__ __forceinline__ u32 mix(u32 a, u32 b) {
  u32 x = a * 0x9E3779B1u ^ b * 0x85EBCA77u;
  x ^= x >> 15, x *= 0x2C1B3C6Du, x ^= x >> 12;
  return x;
}
__device__ __forceinline__ u32 gather(const uint4* tab, u64 slots, u32 i, u32 w) {  // one 64-byte table point
  const uint4* e = tab + ((u64)mix(i, w) * slots >> 32) * 4;
  const uint4 a = e[0], b = e[1], c = e[2], d = e[3];
  return a.x ^ b.y ^ c.z ^ d.w;
}
// `words` 4-byte words per scalar in plane layout: word j of scalar (r, t) at (r * words + j) * nt + t
__device__ __forceinline__ void put(u32* buf, u32 nt, u32 t, u32 r, u32 words, u32 v) {
  for (u32 j = 0; j < words; ++j) buf[((size_t)r * words + j) * nt + t] = v + j;
}
__device__ __forceinline__ u32 get(const u32* buf, u32 nt, u32 t, u32 r, u32 words) {
  u32 v = 0;
  for (u32 j = 0; j < words; ++j) v ^= buf[((size_t)r * words + j) * nt + t];
  return v;
}
// chain: 9 words per prefix product; sums: 18 words per point
__global__ void __launch_bounds__(256) k_replay(const uint4* tab, u64 slots, const uint4* scal, u32 nt, u32 R, u32* chain, u32* s1, u32* s2, u32* s3, u32* out) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nt) return;
  u32 acc = 0;
  for (u32 r = 0; r < R; ++r) {  // level 1 forward
    const u32 i = r * nt + t;
    acc ^= scal[(size_t)i * 2].x ^ scal[(size_t)i * 2 + 1].y;
    for (u32 w = 0; w < 12; ++w) acc ^= gather(tab, slots, i, w);
    put(chain, nt, t, r, 54, acc);
  }
  for (u32 r = R; r-- > 0;) {  // level 1 backward + level 2 forward
    const u32 i = r * nt + t;
    acc ^= scal[(size_t)i * 2].x ^ get(chain, nt, t, r, 54);
    for (u32 w = 0; w < 12; ++w) acc ^= gather(tab, slots, i, w);
    put(s1, nt, t, r, 108, acc);
    put(chain, nt, t, r, 27, acc);  // (the real kernel writes these into a second chain buffer; the bytes are the same)
  }
  for (u32 r = 0; r < R; ++r) {  // level 2 backward + level 3 forward
    acc ^= get(chain, nt, t, r, 27) ^ get(s1, nt, t, r, 108);
    put(s2, nt, t, r, 54, acc);
    put(chain, nt, t, r, 9, acc);
  }
  for (u32 r = R; r-- > 0;) {  // level 3 backward + level 4 forward
    acc ^= get(chain, nt, t, r, 9) ^ get(s2, nt, t, r, 36) ^ get(s2, nt, t, r, 9);
    put(s3, nt, t, r, 18, acc);
    put(chain, nt, t, r, 9, acc);
  }
  for (u32 r = 0; r < R; ++r) acc ^= get(chain, nt, t, r, 9) ^ get(s3, nt, t, r, 18) ^ get(s2, nt, t, r, 18);  // level 4 backward
  if (acc == 0x1234567u) *out = acc;
}
// the Jacobian kernel's traffic for comparison: 12 gathers + 36 words parked and read back
__global__ void __launch_bounds__(256) k_replay_jacobian(const uint4* tab, u64 slots, const uint4* scal, u32 nt, u32 R, u32* park, u32* out) {
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nt) return;
  u32 acc = 0;
  for (u32 r = 0; r < R; ++r) {
    const u32 i = r * nt + t;
    acc ^= scal[(size_t)i * 2].x ^ scal[(size_t)i * 2 + 1].y;
    for (u32 w = 0; w < 12; ++w) acc ^= gather(tab, slots, i, w);
    put(park, nt, t, r, 36, acc);
  }
  for (u32 r = R; r-- > 0;) acc ^= get(park, nt, t, r, 36);
  if (acc == 0x1234567u) *out = acc;
}

int main() {
  const u64 slots = 12ull * ((1ull << 22) - 1);  // the 22-bit window table: 3.2 GB
  uint4 *tab, *scal;
  u32 *chain, *s1, *s2, *s3, *out;
  const u32 nmax = 1u << 24;
  CHK(hipMalloc(&tab, slots * 64));
  CHK(hipMalloc(&scal, (size_t)nmax * 32));
  CHK(hipMalloc(&chain, (size_t)nmax * 54 * 4));
  CHK(hipMalloc(&s1, (size_t)nmax * 108 * 4));
  CHK(hipMalloc(&s2, (size_t)nmax * 54 * 4));
  CHK(hipMalloc(&s3, (size_t)nmax * 18 * 4));
  CHK(hipMalloc(&out, 4));
  CHK(hipMemset(tab, 1, slots * 64));
  CHK(hipMemset(scal, 2, (size_t)nmax * 32));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (u32 nt : {1u << 17, 1u << 18, 1u << 19}) {
    for (u32 R : {16u, 32u, 64u}) {
      if ((u64)nt * R > nmax) continue;
      for (int which = 0; which < 2; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CHK(hipEventRecord(e0, 0));
          if (which == 0) hipLaunchKernelGGL(k_replay, dim3(nt / 256), dim3(256), 0, 0, tab, slots, scal, nt, R, chain, s1, s2, s3, out);
          else hipLaunchKernelGGL(k_replay_jacobian, dim3(nt / 256), dim3(256), 0, 0, tab, slots, scal, nt, R, chain, out);
          CHK(hipEventRecord(e1, 0));
          CHK(hipEventSynchronize(e1));
          float ms;
          CHK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        const double n = (double)nt * R;
        printf("%-44s %7u threads x %2u scalars: %8.3f ms = %7.1f M scalars/s\n", which ? "Jacobian kernel's accesses (12 gathers + park)" : "batched-affine tree's accesses (memory only)", nt, R, best, n / best / 1e3);
      }
    }
  }
  return 0;
}
