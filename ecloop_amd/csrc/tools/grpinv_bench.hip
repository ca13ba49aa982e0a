// grpinv_bench.hip — the batch-inverse product tree staged in LDS (BASELINE north_star), built and timed against
// what the add kernel does instead (one inversion chain per lane, prefix products parked in HBM).
//
// In k_add every lane inverts the product of ITS OWN B = 1024 differences: 3 multiplications per element for
// Montgomery's trick + one 255 S + 15 M inversion per lane per group = 270 / 2048 = 0.13 multiplications per key.
// A product tree over the 256 lanes of a workgroup would share that inversion: the lanes' chain products go up a
// binary tree in LDS (8 levels), ONE wave inverts the root, the inverse comes down (2 multiplications per node).
// This program measures exactly that step, in isolation and at the add kernel's occupancy (4 workgroups of 256 per CU):
//   per_lane : every lane runs fe_inv on its own value                      (what k_add does once per group)
//   lds_tree : 256 values -> tree in LDS -> one fe_inv by wave 0 -> down     (what the north_star sketches)
// and checks that both give the same inverses.  Result (profiles/r02_grpinv_lds_tree.txt) and what it means for the
// kernel: DESIGN.md §3.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 grpinv_bench.hip -o grpinv_bench && ./grpinv_bench
#include "../fe256.h"

#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define ITER 64

__device__ __forceinline__ fe seed_fe(u32 s) {
  fe x;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) x.n[i] = (s * 2654435761u + 0x9E3779B9u * (i + 1)) & (i == 8 ? FE_TOP : FE_M);
  x.n[0] |= 1;
  return x;
}
__device__ __forceinline__ void store_words(u32* out, fe x) {
  fe_normalize(x);
  u32 w[8];
  fe_to_words(w, x);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = w[i];
}

__global__ void __launch_bounds__(256, 4) k_inv_per_lane(u32* out, u32 seed, int iters) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  fe x = seed_fe(seed ^ g);
  const fe c = seed_fe(g + 77);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    fe r = fe_inv(x);
    x = fe_add(r, c);
    fe_normalize_weak(x);
  }
  store_words(out + (size_t)g * 8, x);
}

struct lds_fe { u32 n[FE_LIMBS]; };
__device__ __forceinline__ fe ld(const lds_fe* p) {
  fe r;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) r.n[i] = p->n[i];
  return r;
}
__device__ __forceinline__ void st(lds_fe* p, const fe& a) {
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) p->n[i] = a.n[i];
}

__global__ void __launch_bounds__(256, 4) k_inv_lds_tree(u32* out, u32 seed, int iters) {
  __shared__ lds_fe node[512];  // heap: node[1] = root, children of i are 2i and 2i+1, leaves 256..511
  __shared__ lds_fe invn[512];
  const u32 lane = threadIdx.x, g = blockIdx.x * 256u + lane;
  fe x = seed_fe(seed ^ g);
  const fe c = seed_fe(g + 77);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    st(&node[256 + lane], x);
    __syncthreads();
#pragma unroll 1
    for (u32 s = 128; s >= 1; s >>= 1) {  // up: 255 multiplications, 8 barriers
      if (lane < s) st(&node[s + lane], fe_mul(ld(&node[2 * (s + lane)]), ld(&node[2 * (s + lane) + 1])));
      __syncthreads();
    }
    if (lane < 64) {  // one wave inverts the root (all its lanes the same value: one inversion's worth of issue slots)
      fe r = fe_inv(ld(&node[1]));
      if (lane == 0) st(&invn[1], r);
    }
    __syncthreads();
#pragma unroll 1
    for (u32 s = 1; s <= 128; s <<= 1) {  // down: 510 multiplications, 8 barriers
      if (lane < s) {
        const u32 i = s + lane;
        const fe I = ld(&invn[i]);
        st(&invn[2 * i], fe_mul(I, ld(&node[2 * i + 1])));
        st(&invn[2 * i + 1], fe_mul(I, ld(&node[2 * i])));
      }
      __syncthreads();
    }
    x = fe_add(ld(&invn[256 + lane]), c);
    fe_normalize_weak(x);
    __syncthreads();
  }
  store_words(out + (size_t)g * 8, x);
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 4 * 8;  // 8 rounds of the resident 4 workgroups per CU
  const size_t n = (size_t)blocks * 256;
  u32 *a, *b;
  CHECK(hipMalloc(&a, n * 32));
  CHECK(hipMalloc(&b, n * 32));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  for (int rep = 0; rep < 2; ++rep) {  // first pass warms up
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_inv_per_lane, dim3(blocks), dim3(256), 0, 0, a, 12345u, ITER);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms[0], e0, e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_inv_lds_tree, dim3(blocks), dim3(256), 0, 0, b, 12345u, ITER);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms[1], e0, e1));
  }
  std::vector<u32> ha(n * 8), hb(n * 8);
  CHECK(hipMemcpy(ha.data(), a, n * 32, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hb.data(), b, n * 32, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < n * 8; ++i) diff += ha[i] != hb[i];
  const double inv = (double)n * ITER;
  printf("# %s, %d CUs: %zu lanes x %d inversions each, 4 workgroups of 256 per CU\n", p.gcnArchName, p.multiProcessorCount, n, ITER);
  printf("per_lane : %8.3f ms  %7.2f G inversions/s\n", ms[0], inv / ms[0] / 1e6);
  printf("lds_tree : %8.3f ms  %7.2f G inversions/s   (%.2fx)\n", ms[1], inv / ms[1] / 1e6, ms[0] / ms[1]);
  printf("results identical: %s (%zu differing words)\n", diff == 0 ? "yes" : "NO", diff);
  // what the difference is worth in the add kernel: one inversion per lane per 2B = 2048 keys = 2^21 per 2^32-key launch
  const double share0 = ms[0] * (double)(1u << 21) / inv, share1 = ms[1] * (double)(1u << 21) / inv;
  printf("k_add, 2^32 keys = 2^21 lane-inversions: per-lane %.3f ms of GPU time, LDS tree %.3f ms -> the tree would save %.3f ms per launch\n", share0, share1,
         share0 - share1);
  return 0;
}
