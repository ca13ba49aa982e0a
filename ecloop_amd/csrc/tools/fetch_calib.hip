// fetch_calib.hip — known-byte-count kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the two
// access patterns of the add kernel (MI355X_MICROARCH.md §HBM: "calibrate on a known byte count in your own access
// pattern before trusting an absolute"):
//   stream16_rd / stream16_wr : 16 bytes per lane, coalesced (1 KiB per wave instruction) - the prefix-product chain
//   random8_*                 : one 8-byte word per lane at a pseudo-random index - the bloom probe; over a 54 MB array
//                               (Infinity-Cache resident, the headline filter) and over a 5.9 GB array (configs[2])
// Run it under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and again with WRITE_SIZE); it prints the algorithmic
// bytes of every launch as "CALIB <kernel> <requests> <bytes>"; tools/make_roofline_profile.py joins the two.
//   hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void stream16_rd(const uint4* __restrict__ p, size_t n, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void stream16_wr(uint4* __restrict__ p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_uint4(seed, (uint32_t)i, seed ^ (uint32_t)i, 7u);
}
__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// `per` independent random 8-byte reads per lane over nwords words
__global__ void random8_rd(const uint64_t* __restrict__ p, uint64_t nwords, uint32_t per, uint32_t* out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t acc = 0;
  for (uint32_t k = 0; k < per; ++k) acc ^= p[mix(t * per + k + 0x9E3779B97F4A7C15ull) % nwords];
  if (acc == 0x12345678u) out[0] = (uint32_t)acc;
}

static hipEvent_t e0, e1;
#define TIMED(name, units, call)                                                           \
  do {                                                                                     \
    CHECK(hipEventRecord(e0));                                                             \
    call;                                                                                  \
    CHECK(hipEventRecord(e1));                                                             \
    CHECK(hipEventSynchronize(e1));                                                        \
    float ms_;                                                                             \
    CHECK(hipEventElapsedTime(&ms_, e0, e1));                                              \
    printf("TIME %s %.3f ms  %.2f G requests/s  %.1f GB/s algorithmic\n", name, ms_, (units) / ms_ / 1e6, (units) * bytes_per_ / ms_ / 1e6); \
  } while (0)

int main() {
  CHECK(hipSetDevice(0));
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double bytes_per_ = 16;
  uint32_t* out;
  CHECK(hipMalloc(&out, 64));
  const size_t big = (size_t)5900 << 20, small = (size_t)54 << 20;  // bytes
  uint64_t* a;
  CHECK(hipMalloc(&a, big));
  CHECK(hipMemset(a, 0x5a, big));
  CHECK(hipDeviceSynchronize());
  // streaming read / write of the whole 5.9 GB array (far beyond the 256 MB Infinity Cache)
  TIMED("stream16_rd", (double)(big / 16), hipLaunchKernelGGL(stream16_rd, dim3(256 * 32), dim3(256), 0, 0, (const uint4*)a, big / 16, out));
  printf("CALIB stream16_rd %zu %zu\n", big / 16, big);
  TIMED("stream16_wr", (double)(big / 16), hipLaunchKernelGGL(stream16_wr, dim3(256 * 32), dim3(256), 0, 0, (uint4*)a, big / 16, 3u));
  printf("CALIB stream16_wr %zu %zu\n", big / 16, big);
  // random 8-byte probes: 2^28 of them, over 54 MB and over 5.9 GB.  Two launches each (the first warms the caches
  // for the small array; both are reported, the profile summary lists them in launch order)
  const uint32_t per = 16;
  const uint64_t lanes = 1ull << 24;
  bytes_per_ = 8;
  for (int rep = 0; rep < 2; ++rep) {
    TIMED("random8_rd_54MB", (double)(lanes * per), hipLaunchKernelGGL(random8_rd, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, a, (uint64_t)(small / 8), per, out));
    printf("CALIB random8_rd_54MB %llu %llu\n", (unsigned long long)(lanes * per), (unsigned long long)(lanes * per * 8));
  }
  for (int rep = 0; rep < 2; ++rep) {
    TIMED("random8_rd_5900MB", (double)(lanes * per), hipLaunchKernelGGL(random8_rd, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, a, (uint64_t)(big / 8), per, out));
    printf("CALIB random8_rd_5900MB %llu %llu\n", (unsigned long long)(lanes * per), (unsigned long long)(lanes * per * 8));
  }
  CHECK(hipFree(a));
  return 0;
}
