// h2d_probe.hip - how fast do 32-byte scalars get from host memory to the kernels of `mul`?  (round 4)
//   (1) hipMemcpyAsync from page-locked memory (hipHostMalloc / hipHostRegister), one copy stream and two;
//   (2) a kernel reading the page-locked array in place over PCIe (zero-copy), 32 bytes per lane, coalesced.
// Build: hipcc --offload-arch=gfx950 -O3 h2d_probe.hip -o h2d_probe;  run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_read(const uint4* __restrict__ src, size_t n16, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i * 2 + 1 < n16; i += stride) {  // one 32-byte scalar per lane per round, like k_mul_check
    const uint4 a = src[i * 2], b = src[i * 2 + 1];
    acc += a.x ^ a.w ^ b.y ^ b.z;
  }
  if (acc == 0x12345678u) *out = acc;
}

static double copy_rate(void* dst, const void* src, size_t bytes, int streams, hipStream_t* st, int reps) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0, st[0]));
  for (int r = 0; r < reps; ++r)
    for (int s = 0; s < streams; ++s) {
      size_t part = bytes / streams;
      CHK(hipMemcpyAsync((char*)dst + s * part, (const char*)src + s * part, part, hipMemcpyHostToDevice, st[s]));
    }
  for (int s = 1; s < streams; ++s) { hipEvent_t d; CHK(hipEventCreate(&d)); CHK(hipEventRecord(d, st[s])); CHK(hipStreamWaitEvent(st[0], d, 0)); }
  CHK(hipEventRecord(e1, st[0]));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  return (double)bytes * reps / (ms * 1e-3) / 1e9;
}

int main() {
  hipStream_t st[4];
  for (int i = 0; i < 4; ++i) CHK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  const size_t sizes[] = {(size_t)32 << 20, (size_t)128 << 20, (size_t)512 << 20};
  void *dev = nullptr, *pin = nullptr, *reg = nullptr;
  const size_t big = sizes[2];
  CHK(hipMalloc(&dev, big));
  CHK(hipHostMalloc(&pin, big, hipHostMallocDefault));
  reg = aligned_alloc(4096, big);
  memset(pin, 1, big); memset(reg, 2, big);
  CHK(hipHostRegister(reg, big, hipHostRegisterDefault));
  unsigned* out = nullptr;
  CHK(hipMalloc(&out, 4));
  for (size_t bytes : sizes) {
    printf("H2D %4zu MB  hipHostMalloc: 1 stream %6.1f GB/s, 2 streams %6.1f, 4 streams %6.1f | hipHostRegister: 1 stream %6.1f, 2 streams %6.1f\n", bytes >> 20,
           copy_rate(dev, pin, bytes, 1, st, 4), copy_rate(dev, pin, bytes, 2, st, 4), copy_rate(dev, pin, bytes, 4, st, 4),
           copy_rate(dev, reg, bytes, 1, st, 4), copy_rate(dev, reg, bytes, 2, st, 4));
  }
  void *dpin = nullptr, *dreg = nullptr;
  CHK(hipHostGetDevicePointer(&dpin, pin, 0));
  CHK(hipHostGetDevicePointer(&dreg, reg, 0));
  for (int threads : {1 << 16, 1 << 17, 1 << 18, 1 << 20}) {
    for (int which = 0; which < 2; ++which) {
      hipEvent_t e0, e1;
      CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
      CHK(hipEventRecord(e0, st[0]));
      hipLaunchKernelGGL(k_read, dim3(threads / 256), dim3(256), 0, st[0], (const uint4*)(which ? dreg : dpin), big / 16, out);
      CHK(hipEventRecord(e1, st[0]));
      CHK(hipEventSynchronize(e1));
      float ms = 0;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      printf("zero-copy read of 512 MB %s, %7d threads: %6.1f GB/s = %6.1f M scalars/s\n", which ? "hipHostRegister'ed" : "hipHostMalloc'ed   ", threads,
             big / (ms * 1e-3) / 1e9, big / 32 / (ms * 1e-3) / 1e6);
    }
  }
  return 0;
}
