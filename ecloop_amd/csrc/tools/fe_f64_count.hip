// fe_f64_count.hip - round-2 review item 4: would v_fma_f64 make the field multiplication cheaper?  Compile-only experiment:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c fe_f64_count.hip -save-temps ; count the VALU instructions of k_f64_product
// (result and verdict: profiles/r03_fe_f64_count.txt, DESIGN.md "tried and rejected")
// product stage of a 256-bit multiplication on v_fma_f64: 5 x 52-bit limbs held as doubles (exact integers < 2^52).
// Per partial product: hi = fma(a, b, M) rounds a*b to a multiple of 2^52 (M = 1.5 * 2^104 fixes the exponent),
// lo = fma(a, b, -(hi - M)) is the exact remainder (|lo| <= 2^51).  Five lows and four highs meet in a column, so the
// column does not fit a double's 53 bits: the parts are summed as INTEGERS - hi and lo are offset so that every value
// of a kind has the same exponent field and the raw bit patterns add (Emmart / Luitjens / Weems / Woolley, "Optimizing
// modular multiplication for NVIDIA's Maxwell GPUs" and the DPF follow-up).  Columns out: 10 x int64.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint64_t u64;
__device__ __forceinline__ u64 bits(double x) { return __double_as_longlong(x); }
__global__ void k_f64_product(const double* __restrict__ a_, const double* __restrict__ b_, u64* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double a[5], b[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) a[i] = a_[t * 5 + i], b[i] = b_[t * 5 + i];
  const double M = 30423614405477505635920876929024.0;  /* 1.5 * 2^104 */
  const double M2 = 6755399441055744.0;                  /* 1.5 * 2^52  */
  u64 col[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) col[k] = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double hi = __fma_rn(a[i], b[j], M);       // M + round(a*b / 2^52) * 2^52
      const double hs = M - hi;                        // -(high part), exact
      const double lo = __fma_rn(a[i], b[j], hs + M2); // M2 + low part (signed, |.| <= 2^51), exact
      col[i + j + 1] += bits(hi);                      // exponent fields are constant: subtracted once per column below
      col[i + j] += bits(lo);
    }
#pragma unroll
  for (int k = 0; k < 10; ++k) out[t * 10 + k] = col[k];
}
