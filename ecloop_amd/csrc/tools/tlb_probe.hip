// tlb_probe.hip — address translation under the bloom probe's access pattern (one random 8-byte word per lane) on gfx950.
// profiles/r03_pmc_filter_compare.txt: with a 5.9 GB filter 70 % of the add kernel's probes miss the per-CU translation
// cache (TCP_UTCL1_TRANSLATION_MISS x240 against a 54 MB filter) and the address pipeline stalls behind them.  This tool
// asks what decides the miss rate: the array size (reach of the UTCL1), and how the array was allocated -
//   malloc   hipMalloc(size)
//   aligned  hipMalloc(size + 1 GB), array at the 1 GB-aligned address inside
//   (a third variant through hipMemAddressReserve / hipMemCreate / hipMemMap was dropped: the reservation ignored the
//    requested 1 GB alignment and mapping 2 GB ended in a GPU memory access fault on this ROCm 7.2 stack)
// - larger physically contiguous, naturally aligned fragments let one translation entry cover more of the array.
// Run plain for rates, and under `rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum`
// for the miss counts (one dispatch per line of output, in order).
//   hipcc --offload-arch=gfx950 -O3 tlb_probe.hip -o tlb_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void random8_rd(const uint64_t* __restrict__ p, uint64_t nwords, uint32_t per, uint32_t* out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = 0;
  for (uint32_t k = 0; k < per; ++k) acc ^= p[mix(t * per + k + 0x9E3779B97F4A7C15ull) % nwords];
  if (acc == 0x12345678u) out[0] = (uint32_t)acc;
}
int main() {
  CHECK(hipSetDevice(0));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  uint32_t* out;
  CHECK(hipMalloc(&out, 64));
  const size_t GB = (size_t)1 << 30;
  const size_t sizes[] = {(size_t)54 << 20, (size_t)256 << 20, GB, 2 * GB, (size_t)5900 << 20, 16 * GB};
  printf("%-8s %10s %18s %14s\n", "alloc", "MB", "address mod 1 GB", "G probes/s");
  for (int mode = 0; mode < 2; ++mode)
    for (size_t bytes : sizes) {
      uint64_t* p = nullptr;
      void* base = nullptr;
      if (mode == 0) {
        CHECK(hipMalloc(&base, bytes));
        p = (uint64_t*)base;
      } else if (mode == 1) {
        CHECK(hipMalloc(&base, bytes + GB));
        p = (uint64_t*)(((uintptr_t)base + GB - 1) & ~(uintptr_t)(GB - 1));
      }
      CHECK(hipMemset(p, 0x5a, bytes));
      const uint64_t nwords = bytes / 8;
      const uint32_t per = 8;
      const unsigned blocks = 1u << 17;  // 2^25 lanes x 8 probes = 2^28 probes
      hipLaunchKernelGGL(random8_rd, dim3(blocks), dim3(256), 0, 0, p, nwords, per, out);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(random8_rd, dim3(blocks), dim3(256), 0, 0, p, nwords, per, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-8s %10zu %18zu %14.2f\n", mode == 0 ? "malloc" : "aligned", bytes >> 20, (size_t)((uintptr_t)p & (GB - 1)),
             (double)blocks * 256 * per / ms / 1e6);
      fflush(stdout);
      CHECK(hipFree(base));
    }
  return 0;
}
