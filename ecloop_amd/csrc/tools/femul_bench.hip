// femul_bench.hip — issue cost of the field multiplication / squaring of fe256.h as the compiler emits them: SIMD-clocks per
// fe_mul and fe_sqr (and per VALU instruction of them) at 1 / 2 / 4 workgroups of 256 per CU, as one dependent chain per lane and as
// two independent chains per lane.  The `mul` kernels run at ~5 clocks per instruction on a stream that is half v_mad_u64_u32
// (4.6 clocks in the dependency-free microbenchmark, profiles/ubench_r04.txt); this program says how much of the difference is the
// multiplication's own instruction stream (accumulator chains, register banks) and how much the kernel around it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 femul_bench.hip -o femul_bench && ./femul_bench
#include "../fe256.h"

#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define ITER 2048

__device__ __forceinline__ fe seed_fe(u32 s) {
  fe x;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) x.n[i] = (s * 2654435761u + 0x9E3779B9u * (i + 1)) & (i == 8 ? FE_TOP : FE_M);
  x.n[0] |= 1;
  return x;
}
// MODE 0: x = x * y (one chain); 1: x = x * y, z = z * y (two chains); 2: x = x^2; 3: x = x^2, z = z^2; 4 / 5: the two chains through fe_mul2 / fe_sqr2
template <int MODE, int WG>
__global__ void __launch_bounds__(256, WG) k_mul(u32* out, u32 seed, int iters) {
  const u32 g = blockIdx.x * 256u + threadIdx.x;
  fe x = seed_fe(seed ^ g), z = seed_fe(seed + g);
  const fe y = seed_fe(g + 77);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) x = fe_mul(x, y);
    if (MODE == 1) x = fe_mul(x, y), z = fe_mul(z, y);
    if (MODE == 2) x = fe_sqr(x);
    if (MODE == 3) x = fe_sqr(x), z = fe_sqr(z);
    if (MODE == 4) fe_mul2(x, z, x, y, z, y);
    if (MODE == 5) fe_sqr2(x, z, x, z);
  }
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < FE_LIMBS; ++i) acc ^= x.n[i] ^ z.n[i];
  out[g] = acc;
}

template <int MODE, int WG>
static int run(int cus, u32* out, const char* what, int per_iter, int instr) {
  const int blocks = cus * WG * 2;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mul<MODE, WG>), dim3(blocks), dim3(256), 0, 0, out, 12345u, ITER);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double wave_ops = (double)blocks * 4 * ITER * per_iter, simds = cus * 4.0;
  const double clocks = ms * 1e-3 * 2.4e9 * simds / wave_ops;  // SIMD-clocks per operation and wave at 2.4 GHz
  printf("%d workgroups/CU  %-26s %8.3f ms  %7.1f SIMD-clocks per operation  (%.2f per instruction at %d)\n", WG, what, ms, clocks, clocks / instr, instr);
  return 0;
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  u32* out;
  CHECK(hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4));
  printf("# %s, %d CUs, %d dependent operations per lane; instruction counts: fe_mul 153, fe_sqr 125 (tools/isa_mix.py)\n", p.gcnArchName, p.multiProcessorCount, ITER);
  const int c = p.multiProcessorCount;
  run<0, 1>(c, out, "fe_mul, one chain", 1, 153), run<1, 1>(c, out, "fe_mul, two chains", 2, 153), run<2, 1>(c, out, "fe_sqr, one chain", 1, 125), run<3, 1>(c, out, "fe_sqr, two chains", 2, 125);
  run<4, 1>(c, out, "fe_mul2 (interleaved pair)", 2, 153), run<5, 1>(c, out, "fe_sqr2 (interleaved pair)", 2, 125);
  run<0, 2>(c, out, "fe_mul, one chain", 1, 153), run<1, 2>(c, out, "fe_mul, two chains", 2, 153), run<2, 2>(c, out, "fe_sqr, one chain", 1, 125), run<3, 2>(c, out, "fe_sqr, two chains", 2, 125);
  run<4, 2>(c, out, "fe_mul2 (interleaved pair)", 2, 153), run<5, 2>(c, out, "fe_sqr2 (interleaved pair)", 2, 125);
  run<0, 4>(c, out, "fe_mul, one chain", 1, 153), run<1, 4>(c, out, "fe_mul, two chains", 2, 153), run<2, 4>(c, out, "fe_sqr, one chain", 1, 125), run<3, 4>(c, out, "fe_sqr, two chains", 2, 125);
  run<4, 4>(c, out, "fe_mul2 (interleaved pair)", 2, 153), run<5, 4>(c, out, "fe_sqr2 (interleaved pair)", 2, 125);
  return 0;
}
