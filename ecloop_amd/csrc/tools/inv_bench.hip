// inv_bench.hip — the two field inversions of fe256.h timed against each other on the device:
//   fermat   : a^(p-2), the 255 S + 15 M addition chain of lib/ecc.c:463-520 (what every kernel used until round 4)
//   divsteps : 600 Bernstein-Yang division steps in 20 rounds of 30 (fe_inv_divsteps)
// every lane inverts its own value, ITER times in a dependent chain, at 1 / 2 / 4 workgroups of 256 per CU (the occupancies of the
// set-up, `mul` and add kernels); results are compared word by word.  Result: profiles/r04_inv_bench.txt, DESIGN.md §3.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 inv_bench.hip -o inv_bench && ./inv_bench
#include "../fe256.h"

#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define ITER 32

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
template <int ALG, int WG>
__global__ void __launch_bounds__(256, WG) k_inv(u32* out, u32 seed, int iters) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  fe x = seed_fe(seed ^ g);
  const fe c = seed_fe(g + 77);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    fe r = ALG ? fe_inv_divsteps(x) : fe_inv_fermat(x);
    x = fe_add(r, c);
    fe_normalize_weak(x);
  }
  store_words(out + (size_t)g * 8, x);
}

template <int WG>
static int run(int cus, u32* a, u32* b, size_t cap) {
  const int blocks = cus * WG * 4;  // 4 rounds of the resident workgroups
  const size_t n = (size_t)blocks * 256;
  if (n > cap) return 1;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_inv<0, WG>), dim3(blocks), dim3(256), 0, 0, a, 12345u, ITER);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms[0], e0, e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_inv<1, WG>), dim3(blocks), dim3(256), 0, 0, b, 12345u, ITER);
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
  // SIMD-clocks per inversion and wave at 2.4 GHz: time x clock x SIMDs / wave-inversions
  const double waves = inv / 64.0, simds = cus * 4.0;
  printf("%d workgroups/CU: fermat %8.3f ms %6.2f G inv/s (%6.0f SIMD-clocks per wave-inversion)   divsteps %8.3f ms %6.2f G inv/s (%6.0f)   %.2fx   identical: %s\n", WG,
         ms[0], inv / ms[0] / 1e6, ms[0] * 1e-3 * 2.4e9 * simds / waves, ms[1], inv / ms[1] / 1e6, ms[1] * 1e-3 * 2.4e9 * simds / waves, ms[0] / ms[1],
         diff == 0 ? "yes" : "NO");
  return diff != 0;
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const size_t cap = (size_t)p.multiProcessorCount * 4 * 4 * 256;
  u32 *a, *b;
  CHECK(hipMalloc(&a, cap * 32));
  CHECK(hipMalloc(&b, cap * 32));
  printf("# %s, %d CUs, %d dependent inversions per lane\n", p.gcnArchName, p.multiProcessorCount, ITER);
  int bad = run<1>(p.multiProcessorCount, a, b, cap);
  bad |= run<2>(p.multiProcessorCount, a, b, cap);
  bad |= run<4>(p.multiProcessorCount, a, b, cap);
  return bad;
}
