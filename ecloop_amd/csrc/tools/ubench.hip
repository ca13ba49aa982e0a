// ubench.hip — instruction issue-rate microbenchmark for gfx950 (MI355X).
// Calibrates the integer-VALU roofline that bench.py / DESIGN.md price the key-search kernel against:
// the hot path is 32x32->64 multiply-add (v_mad_u64_u32), carry adds, and the rotate/xor/select mix of
// SHA-256 / RIPEMD-160, none of which has a published rate for CDNA4.
//   hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench && ./ubench
// Output: one line per instruction: G wave-instr/s over the chip, and cycles per wave-instruction per SIMD
// (using the device's reported clock), 8 independent dependency chains per lane, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ITERS 4096

// ---- 32-bit in/out, two sources: d = op(d, a)
#define K32_2(NAME, OPSTR)                                                                      \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                          \
    uint32_t r[8], a = seed ^ threadIdx.x;                                                      \
    for (int i = 0; i < 8; ++i) r[i] = a * (i + 3);                                             \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                           \
          asm volatile(OPSTR " %0, %0, %1" : "+v"(r[i]) : "v"(a));                              \
      }                                                                                         \
    }                                                                                           \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                      \
    if (s == 0x12345) out[0] = s;                                                               \
  }
// ---- 32-bit, three sources: d = op(d, a, b)
#define K32_3(NAME, OPSTR)                                                                      \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                          \
    uint32_t r[8], a = seed ^ threadIdx.x, b = a * 7 + 1;                                       \
    for (int i = 0; i < 8; ++i) r[i] = a * (i + 3);                                             \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                           \
          asm volatile(OPSTR " %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));                  \
      }                                                                                         \
    }                                                                                           \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                      \
    if (s == 0x12345) out[0] = s;                                                               \
  }

K32_2(k_add_u32, "v_add_u32")
K32_2(k_xor_b32, "v_xor_b32")
K32_2(k_add_u32_e64, "v_add_u32_e64")
K32_2(k_and_b32, "v_and_b32")
K32_2(k_lshlrev_b32, "v_lshlrev_b32")
K32_2(k_sub_u32, "v_sub_u32")
K32_2(k_mul_lo_u32, "v_mul_lo_u32")
K32_2(k_mul_hi_u32, "v_mul_hi_u32")
K32_2(k_mul_u32_u24, "v_mul_u32_u24")
K32_3(k_add3_u32, "v_add3_u32")
K32_3(k_bfi_b32, "v_bfi_b32")
K32_3(k_alignbit_b32, "v_alignbit_b32")
K32_3(k_mad_u32_u24, "v_mad_u32_u24")
K32_3(k_and_or_b32, "v_and_or_b32")
K32_3(k_perm_b32, "v_perm_b32")
K32_3(k_lshl_add_u32, "v_lshl_add_u32")
K32_3(k_mad_i32_i24, "v_mad_i32_i24")

// ---- round 2: is there a 2-cycle wave64 issue (MI355X_MICROARCH.md: "v_fma_f32 (wave64) 2 cyc") and for which opcodes?
// Same 8 independent chains per lane, but every chain has its OWN source registers (spread over the register banks),
// and the float forms in their VOP2 (e32) encodings next to the VOP3 one measured in round 1.
#define K32_2D(NAME, OPSTR)                                                                     \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                          \
    uint32_t r[8], a[8];                                                                        \
    for (int i = 0; i < 8; ++i) a[i] = (seed ^ threadIdx.x) * (2 * i + 5), r[i] = a[i] * (i + 3); \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                           \
          asm volatile(OPSTR " %0, %0, %1" : "+v"(r[i]) : "v"(a[i]));                           \
      }                                                                                         \
    }                                                                                           \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                      \
    if (s == 0x12345) out[0] = s;                                                               \
  }
#define K32_3D(NAME, OPSTR, TAIL)                                                               \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                          \
    uint32_t r[8], a[8], b[8];                                                                  \
    for (int i = 0; i < 8; ++i) a[i] = (seed ^ threadIdx.x) * (2 * i + 5), b[i] = a[i] * 7 + 1, r[i] = a[i] * (i + 3); \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                           \
          asm volatile(OPSTR " %0, %0, %1, %2" TAIL : "+v"(r[i]) : "v"(a[i]), "v"(b[i]));       \
      }                                                                                         \
    }                                                                                           \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                      \
    if (s == 0x12345) out[0] = s;                                                               \
  }
// one source: d = op(a) into 8 destinations (VOP1)
#define K32_1(NAME, OPSTR)                                                                      \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                          \
    uint32_t r[8], a[8];                                                                        \
    for (int i = 0; i < 8; ++i) a[i] = (seed ^ threadIdx.x) * (2 * i + 5), r[i] = 0;            \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                           \
          asm volatile(OPSTR " %0, %1" : "+v"(r[i]) : "v"(a[i]));                               \
      }                                                                                         \
    }                                                                                           \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                      \
    if (s == 0x12345) out[0] = s;                                                               \
  }
K32_2D(k_fmac_f32_e32, "v_fmac_f32_e32")
K32_2D(k_mul_f32_e32, "v_mul_f32_e32")
K32_2D(k_add_f32_e32, "v_add_f32_e32")
K32_2D(k_add_u32_d, "v_add_u32_e32")
K32_2D(k_or_b32_d, "v_or_b32_e32")
K32_2D(k_lshrrev_b32_d, "v_lshrrev_b32_e32")
K32_2D(k_lshlrev_b32_d, "v_lshlrev_b32_e32")
K32_2(k_lshrrev_b32_s, "v_lshrrev_b32")
K32_2D(k_mul_u32_u24_d, "v_mul_u32_u24_e32")
K32_2D(k_mul_lo_u32_d, "v_mul_lo_u32")
K32_2D(k_ashrrev_d, "v_ashrrev_i32_e32")
K32_2D(k_max_u32_d, "v_max_u32_e32")
K32_2D(k_min_u32_d, "v_min_u32_e32")
K32_2D(k_cndmask_d, "v_cndmask_b32_e32")
K32_2D(k_mul_lo_u16_d, "v_mul_lo_u16_e32")
K32_2D(k_pk_mul_lo_u16_d, "v_pk_mul_lo_u16")
K32_1(k_mov_b32, "v_mov_b32_e32")
K32_1(k_not_b32, "v_not_b32_e32")
K32_1(k_bfrev_b32, "v_bfrev_b32_e32")
K32_3D(k_fma_f32_d, "v_fma_f32", "")
K32_3D(k_bitop3_d, "v_bitop3_b32", " bitop3:0x96")
K32_3D(k_bfe_u32_d, "v_bfe_u32", "")
K32_3D(k_xad_u32_d, "v_xad_u32", "")
K32_3D(k_or3_b32_d, "v_or3_b32", "")
K32_3D(k_add_lshl_u32_d, "v_add_lshl_u32", "")
K32_3D(k_alignbit_d, "v_alignbit_b32", "")
K32_3D(k_add3_d, "v_add3_u32", "")
K32_3D(k_mad_u32_u16_d, "v_mad_u32_u16", "")
K32_3D(k_dot4_u32_u8_d, "v_dot4_u32_u8", "")

// carry chain: v_add_co_u32 + v_addc_co_u32 pairs (VOP2, implicit vcc)
__global__ void k_addc_pair(uint32_t* out, uint32_t seed) {
  uint32_t r[8], a = seed ^ threadIdx.x;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; i += 2)
        asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(a) : "vcc");
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];
  if (s == 0x12345) out[0] = s;
}

// 64-bit destination ops
#define K64(NAME, ASM_BODY)                                                                     \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                          \
    uint64_t r[8]; uint32_t a = seed ^ threadIdx.x, b = a * 7 + 1;                              \
    for (int i = 0; i < 8; ++i) r[i] = (uint64_t)a * (i + 3);                                   \
    for (int it = 0; it < ITERS; ++it) {                                                        \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) { ASM_BODY; }                             \
      }                                                                                         \
    }                                                                                           \
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                      \
    if (s == 0x12345) out[0] = (uint32_t)s;                                                     \
  }
K64(k_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"))
K64(k_mad_u64_u32_sgpr, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "s"(seed) : "vcc"))
K64(k_mad_u64_u32_addc, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc" : "+v"(r[i]), "+v"(b) : "v"(a), "v"(seed) : "vcc"))
K64(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7])))
K64(k_lshrrev_b64, asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(r[i])))
K64(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7])))
K64(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 7])))
K64(k_add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7])))
K64(k_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 7])))
K64(k_pk_add_u16, asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(*(uint32_t*)&r[i]) : "v"(a)))
K32_3(k_fma_f32, "v_fma_f32")

// v_mad_u64_u32 with distinct multiplicand registers per chain and random data (as in a field multiplication)
__global__ void k_mad_u64_distinct(uint32_t* out, uint32_t seed) {
  uint64_t r[8]; uint32_t a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = (seed ^ threadIdx.x) * 2654435761u * (i + 1) + 0x9E3779B9u; b[i] = a[i] * 40503u + 0x7F4A7C15u; r[i] = (uint64_t)a[i] * b[i]; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a[i]), "v"(b[(i + u) & 7]) : "vcc");
    }
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];
  if (s == 0x12345) out[0] = (uint32_t)s;
}
// column accumulation as the compiler emits it for 10x26: acc = a*b + acc with one accumulator per 5 products
__global__ void k_mad_u64_column(uint32_t* out, uint32_t seed) {
  uint64_t r[2] = {seed, seed * 3ull}; uint32_t a[10], b[10];
  for (int i = 0; i < 10; ++i) { a[i] = ((seed ^ threadIdx.x) * 2654435761u * (i + 1)) & 0x3FFFFFF; b[i] = (a[i] * 40503u + 0x7F4A7C15u) & 0x3FFFFFF; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[0]) : "v"(a[u % 10]), "v"(b[(u * 3) % 10]) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[1]) : "v"(a[(u + 5) % 10]), "v"(b[(u * 7) % 10]) : "vcc");
    }
  }
  if ((r[0] ^ r[1]) == 0x12345) out[0] = (uint32_t)r[0];
}

// dependent chain latency: one accumulator
__global__ void k_mad_u64_dep(uint32_t* out, uint32_t seed) {
  uint64_t r = seed ^ threadIdx.x; uint32_t a = seed, b = a * 7 + 1;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(a), "v"(b) : "vcc");
  }
  if (r == 0x12345) out[0] = (uint32_t)r;
}
__global__ void k_add_dep(uint32_t* out, uint32_t seed) {
  uint32_t r = seed ^ threadIdx.x, a = seed;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(a));
  }
  if (r == 0x12345) out[0] = r;
}

// mixed streams: do double-rate ops keep their rate next to 4-cycle ops, and does their order matter?
// per iteration 32 instructions: 16 v_add_u32 + 16 v_alignbit_b32 on independent registers
__global__ void k_mix_alternating(uint32_t* out, uint32_t seed) {
  uint32_t r[8], q[8], a = seed ^ threadIdx.x, b = a * 7 + 1;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3), q[i] = b * (i + 5);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(a), "v"(b));
      }
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i] ^ q[i];
  if (s == 0x12345) out[0] = s;
}
__global__ void k_mix_grouped(uint32_t* out, uint32_t seed) {
  uint32_t r[8], q[8], a = seed ^ threadIdx.x, b = a * 7 + 1;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3), q[i] = b * (i + 5);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(a), "v"(b));
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i] ^ q[i];
  if (s == 0x12345) out[0] = s;
}
// 8 adds per 24 alignbits (the hash's ratio), alternating 1:3
__global__ void k_mix_1to3(uint32_t* out, uint32_t seed) {
  uint32_t r[8], q[8], a = seed ^ threadIdx.x, b = a * 7 + 1;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3), q[i] = b * (i + 5);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(a), "v"(b));
      asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[(i + 3) & 7]) : "v"(a), "v"(b));
      asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[(i + 5) & 7]) : "v"(a), "v"(b));
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i] ^ q[i];
  if (s == 0x12345) out[0] = s;
}

__global__ void k_mix_pairs(uint32_t* out, uint32_t seed) {
  uint32_t r[8], q[8], a = seed ^ threadIdx.x, b = a * 7 + 1;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3), q[i] = b * (i + 5);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i + 1]) : "v"(a));
        asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(a), "v"(b));
        asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[i + 1]) : "v"(a), "v"(b));
      }
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i] ^ q[i];
  if (s == 0x12345) out[0] = s;
}
// 2 adds per 12 four-cycle ops (one RIPEMD step pair): 2 of 14
__global__ void k_mix_pair_in_14(uint32_t* out, uint32_t seed) {
  uint32_t r[8], q[8], a = seed ^ threadIdx.x, b = a * 7 + 1;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3), q[i] = b * (i + 5);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[u]) : "v"(a));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[u + 2]) : "v"(a));
#pragma unroll
      for (int i = 0; i < 14; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[i & 7]) : "v"(a), "v"(b));
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i] ^ q[i];
  if (s == 0x12345) out[0] = s;
}

// ---- calibration in REAL shader cycles: every wave brackets its loop with s_memtime (shader clock) and
// s_memrealtime (constant 100 MHz); out[2..] collects the maxima, main() prints the sustained clock and the
// SIMD-cycles per wave-instruction that follow from it (independent of the nominal clock the table above assumes).
#define CAL_BEGIN uint64_t c0_ = __builtin_readcyclecounter(), w0_ = wall_clock64();
#define CAL_END(out)                                                                         \
  {                                                                                          \
    uint64_t c1_ = __builtin_readcyclecounter(), w1_ = wall_clock64();                       \
    if ((threadIdx.x & 63) == 0) {                                                           \
      atomicMax((unsigned long long*)(out) + 1, (unsigned long long)(c1_ - c0_));            \
      atomicMax((unsigned long long*)(out) + 2, (unsigned long long)(w1_ - w0_));            \
    }                                                                                        \
  }
#define KCAL(NAME, BODY)                                                     \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                       \
    uint32_t r[8], q[8], a = seed ^ threadIdx.x, b = a * 7 + 1;              \
    uint64_t m[2] = {a, b};                                                  \
    for (int i = 0; i < 8; ++i) r[i] = a * (i + 3), q[i] = b * (i + 5);      \
    CAL_BEGIN                                                                \
    for (int it = 0; it < ITERS; ++it) { BODY }                              \
    CAL_END(out)                                                             \
    uint32_t s = (uint32_t)(m[0] ^ m[1]);                                    \
    for (int i = 0; i < 8; ++i) s ^= r[i] ^ q[i];                            \
    if (s == 0x12345) out[0] = s;                                            \
  }
#define I_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[(i) & 7]) : "v"(a));
#define I_ROT(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(q[(i) & 7]) : "v"(a), "v"(b));
#define I_MAD(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[(i) & 1]) : "v"(a), "v"(b) : "vcc");
#define I_BOP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(q[(i) & 7]) : "v"(a), "v"(b));
#define I_AD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[(i) & 7]) : "v"(a), "v"(b));
#define R4(M, i) M(i) M(i + 1) M(i + 2) M(i + 3)
#define R16(M) R4(M, 0) R4(M, 4) R4(M, 8) R4(M, 12)
#define R32(M) R16(M) R16(M)
KCAL(k_cal_rot, R32(I_ROT))
KCAL(k_cal_add, R32(I_ADD))
KCAL(k_cal_bop, R32(I_BOP))
KCAL(k_cal_ad3, R32(I_AD3))
KCAL(k_cal_hashmix, R4(I_ROT, 0) I_BOP(1) I_AD3(0) I_ROT(2) I_ROT(3) I_ROT(4) I_BOP(5) I_BOP(6) I_AD3(1) I_ADD(2) I_AD3(3) I_ROT(5) I_ROT(6) I_BOP(7) I_AD3(4) I_ROT(0) I_ROT(1) I_ROT(2) I_BOP(3) I_AD3(5) I_ADD(6) I_ROT(4) I_BOP(5) I_AD3(7) I_ROT(6) I_ADD(0) I_ROT(7) I_BOP(0) I_AD3(1))
KCAL(k_cal_mad, R32(I_MAD))
KCAL(k_cal_alt, R4(I_ADD, 0) R4(I_ROT, 0) R4(I_ADD, 4) R4(I_ROT, 4) R4(I_ADD, 0) R4(I_ROT, 0) R4(I_ADD, 4) R4(I_ROT, 4))        /* runs of 4 */
KCAL(k_cal_alt1, I_ADD(0) I_ROT(0) I_ADD(1) I_ROT(1) I_ADD(2) I_ROT(2) I_ADD(3) I_ROT(3) I_ADD(4) I_ROT(4) I_ADD(5) I_ROT(5) I_ADD(6) I_ROT(6) I_ADD(7) I_ROT(7) \
                 I_ADD(0) I_ROT(0) I_ADD(1) I_ROT(1) I_ADD(2) I_ROT(2) I_ADD(3) I_ROT(3) I_ADD(4) I_ROT(4) I_ADD(5) I_ROT(5) I_ADD(6) I_ROT(6) I_ADD(7) I_ROT(7))  /* runs of 1 */
KCAL(k_cal_run16, R16(I_ADD) R16(I_ROT))                                                                                           /* runs of 16 */
KCAL(k_cal_hashmix_g10, R4(I_ROT, 0) R4(I_ROT, 4) R4(I_ROT, 0) I_ROT(4) R4(I_AD3, 0) R4(I_AD3, 4) I_AD3(0) R4(I_BOP, 0) I_BOP(4) I_BOP(5) I_BOP(6) I_ADD(0) I_ADD(1) I_ADD(2))
KCAL(k_cal_hashmix_g5, R4(I_ROT, 0) I_ROT(4) I_ROT(5) R4(I_AD3, 0) I_AD3(4) R4(I_BOP, 0) I_ADD(0) R4(I_ROT, 0) I_ROT(6) I_ROT(7) I_ROT(1) R4(I_AD3, 4) I_BOP(4) I_BOP(5) I_BOP(6) I_ADD(1) I_ADD(2))
KCAL(k_cal_madrot, I_MAD(0) I_ROT(0) I_MAD(1) I_ROT(1) I_MAD(0) I_ROT(2) I_MAD(1) I_ROT(3) I_MAD(0) I_ROT(4) I_MAD(1) I_ROT(5) I_MAD(0) I_ROT(6) I_MAD(1) I_ROT(7) \
                   I_MAD(0) I_ROT(0) I_MAD(1) I_ROT(1) I_MAD(0) I_ROT(2) I_MAD(1) I_ROT(3) I_MAD(0) I_ROT(4) I_MAD(1) I_ROT(5) I_MAD(0) I_ROT(6) I_MAD(1) I_ROT(7))

// do the double-rate runs of one wave survive next to OTHER waves of the SIMD issuing 4-clock instructions?  Waves with an
// odd index on their SIMD run the add stream, the even ones the alignbit stream (blocks of 256 = 4 waves = one per SIMD,
// so block parity decides).  If the two pipes did not interfere the mean would be (2.4 + 4.2) / 2.
__global__ void k_mix_by_wave(uint32_t* out, uint32_t seed) {
  uint32_t r[8], a = seed ^ threadIdx.x, b = a * 7 + 1;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3);
  if (blockIdx.x & 1) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      }
    }
  } else {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      }
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];
  if (s == 0x12345) out[0] = s;
}
// the same with the add waves at raised priority (s_setprio 3)
__global__ void k_mix_by_wave_prio(uint32_t* out, uint32_t seed) {
  uint32_t r[8], a = seed ^ threadIdx.x, b = a * 7 + 1;
  for (int i = 0; i < 8; ++i) r[i] = a * (i + 3);
  if (blockIdx.x & 1) {
    __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      }
    }
  } else {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      }
    }
  }
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];
  if (s == 0x12345) out[0] = s;
}

// round 3: the multiplier against the other opcode classes, by wave: the waves of odd blocks run v_mad_u64_u32, the even ones
// another stream.  If 64-bit multiply-adds and plain ALU operations went through units that overlap, the mean per
// instruction would fall below the mean of the two streams run alone (4.6 and 4.2 / 2.4).
#define MIX_MAD_WITH(NAME, OTHER_ASM)                                                            \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                          \
    uint32_t r[8], a = seed ^ threadIdx.x, b = a * 7 + 1;                                       \
    uint64_t q[8];                                                                              \
    for (int i = 0; i < 8; ++i) r[i] = a * (i + 3), q[i] = (uint64_t)r[i] * 0x9E3779B1u;        \
    if (blockIdx.x & 1) {                                                                       \
      for (int it = 0; it < ITERS; ++it) {                                                      \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
          _Pragma("unroll") for (int i = 0; i < 8; ++i)                                         \
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a), "v"(b) : "vcc"); \
        }                                                                                       \
      }                                                                                         \
    } else {                                                                                    \
      for (int it = 0; it < ITERS; ++it) {                                                      \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) OTHER_ASM;                              \
        }                                                                                       \
      }                                                                                         \
    }                                                                                           \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= r[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32); \
    if (s == 0x12345) out[0] = s;                                                               \
  }
MIX_MAD_WITH(k_mix_mad_alignbit, asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))
MIX_MAD_WITH(k_mix_mad_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a)))
MIX_MAD_WITH(k_mix_mad_add3, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)))

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; double per_iter; };

int main(int argc, char** argv) {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, dev));
  int cus = p.multiProcessorCount; double clk = p.clockRate * 1e3;  // Hz
  printf("# device %s, %d CUs, clock %.0f MHz, wave %d\n", p.gcnArchName, cus, clk / 1e6, p.warpSize);
  uint32_t* out; CHECK(hipMalloc(&out, 4096));
  Entry es[] = {
      {"v_add_u32 (e32)", k_add_u32, 32}, {"v_xor_b32 (e32)", k_xor_b32, 32}, {"v_and_b32 (e32)", k_and_b32, 32},
      {"v_lshlrev_b32 (e32)", k_lshlrev_b32, 32}, {"v_sub_u32 (e32)", k_sub_u32, 32}, {"v_add_u32_e64", k_add_u32_e64, 32},
      {"v_add3_u32", k_add3_u32, 32},
      {"v_bfi_b32", k_bfi_b32, 32}, {"v_alignbit_b32", k_alignbit_b32, 32}, {"v_and_or_b32", k_and_or_b32, 32},
      {"v_perm_b32", k_perm_b32, 32}, {"v_lshl_add_u32", k_lshl_add_u32, 32},
      {"v_add_co+v_addc_co (per instr)", k_addc_pair, 32},
      {"v_mul_lo_u32", k_mul_lo_u32, 32}, {"v_mul_hi_u32", k_mul_hi_u32, 32}, {"v_mul_u32_u24", k_mul_u32_u24, 32},
      {"v_mad_u32_u24", k_mad_u32_u24, 32}, {"v_mad_i32_i24", k_mad_i32_i24, 32},
      {"v_mad_u64_u32", k_mad_u64_u32, 32}, {"v_mad_u64_u32 (sgpr src)", k_mad_u64_u32_sgpr, 32},
      {"v_mad_u64_u32+v_addc (per pair)", k_mad_u64_u32_addc, 32},
      {"v_mad_u64_u32 distinct regs/random", k_mad_u64_distinct, 32}, {"v_mad_u64_u32 2 column accumulators", k_mad_u64_column, 32},
      {"v_lshl_add_u64", k_lshl_add_u64, 32}, {"v_lshrrev_b64", k_lshrrev_b64, 32},
      {"v_fma_f32", k_fma_f32, 32}, {"v_pk_fma_f32", k_pk_fma_f32, 32}, {"v_pk_add_u16", k_pk_add_u16, 32},
      {"v_add_f64", k_add_f64, 32}, {"v_mul_f64", k_mul_f64, 32}, {"v_fma_f64", k_fma_f64, 32},
      {"v_mad_u64_u32 dependent chain", k_mad_u64_dep, 32}, {"v_add_u32 dependent chain", k_add_dep, 32},
      {"mix 16 add + 16 alignbit, alternating", k_mix_alternating, 32}, {"mix 16 add + 16 alignbit, grouped by 8", k_mix_grouped, 32},
      {"mix 8 add + 24 alignbit (1:3)", k_mix_1to3, 32},
      {"half the waves add, half alignbit", k_mix_by_wave, 32}, {"  same, add waves at s_setprio 3", k_mix_by_wave_prio, 32},
      // round 2: distinct source registers per chain; VOP2 float forms; more opcodes of the hash / limb arithmetic
      {"v_fmac_f32_e32  (distinct regs)", k_fmac_f32_e32, 32}, {"v_mul_f32_e32   (distinct regs)", k_mul_f32_e32, 32},
      {"v_add_f32_e32   (distinct regs)", k_add_f32_e32, 32}, {"v_fma_f32 VOP3  (distinct regs)", k_fma_f32_d, 32},
      {"v_add_u32_e32   (distinct regs)", k_add_u32_d, 32}, {"v_or_b32_e32    (distinct regs)", k_or_b32_d, 32},
      {"v_mov_b32_e32", k_mov_b32, 32}, {"v_not_b32_e32", k_not_b32, 32}, {"v_bfrev_b32_e32", k_bfrev_b32, 32},
      {"v_cndmask_b32_e32 (vcc)", k_cndmask_d, 32}, {"v_min_u32_e32", k_min_u32_d, 32}, {"v_lshrrev_b32_e32 (distinct regs)", k_lshrrev_b32_d, 32}, {"v_lshlrev_b32_e32 (distinct regs)", k_lshlrev_b32_d, 32},
      {"v_lshrrev_b32 (shared operand)", k_lshrrev_b32_s, 32}, {"v_ashrrev_i32_e32 (distinct regs)", k_ashrrev_d, 32},
      {"v_mul_u32_u24_e32 (distinct regs)", k_mul_u32_u24_d, 32}, {"v_mul_lo_u32 (distinct regs)", k_mul_lo_u32_d, 32}, {"v_max_u32_e32", k_max_u32_d, 32},
      {"v_mul_lo_u16_e32", k_mul_lo_u16_d, 32}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16_d, 32},
      {"v_bitop3_b32    (distinct regs)", k_bitop3_d, 32}, {"v_alignbit_b32  (distinct regs)", k_alignbit_d, 32},
      {"v_add3_u32      (distinct regs)", k_add3_d, 32}, {"v_bfe_u32", k_bfe_u32_d, 32}, {"v_xad_u32", k_xad_u32_d, 32},
      {"v_or3_b32", k_or3_b32_d, 32}, {"v_add_lshl_u32", k_add_lshl_u32_d, 32}, {"v_mad_u32_u16", k_mad_u32_u16_d, 32},
      {"v_dot4_u32_u8", k_dot4_u32_u8_d, 32},
      {"mix 16 add + 16 alignbit, in pairs", k_mix_pairs, 32}, {"mix 4 add (2 pairs) + 28 alignbit", k_mix_pair_in_14, 32},
      // round 3: is the 64-bit multiplier a unit of its own?
      {"half the waves mad_u64, half alignbit", k_mix_mad_alignbit, 32}, {"half the waves mad_u64, half add", k_mix_mad_add, 32},
      {"half the waves mad_u64, half add3", k_mix_mad_add3, 32},
  };
  int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
  int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-36s %14s %16s\n", "instruction", "Gwave-instr/s", "cyc/instr/SIMD");
  for (auto& e : es) {
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);  // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double instrs = 3.0 * blocks * 4.0 * ITERS * e.per_iter;  // wave-instructions
    double rate = instrs / (ms * 1e-3);
    double cyc = (cus * 4.0 * clk) / rate;  // SIMD-cycles per wave-instr
    printf("%-36s %14.1f %16.2f\n", e.name, rate / 1e9, cyc);
  }
  printf("\n# calibration in real shader cycles (s_memtime / s_memrealtime inside the kernels)\n");
  printf("%-44s %12s %20s\n", "stream (32 instr per iteration)", "clock GHz", "real cyc/instr/SIMD");
  struct { const char* name; kern_t k; } cs[] = {
      {"v_alignbit_b32 only", k_cal_rot}, {"v_add_u32 only", k_cal_add}, {"v_bitop3_b32 only", k_cal_bop}, {"v_add3_u32 only", k_cal_ad3},
      {"hash-like mix (rot/bitop3/add3, 3 single adds)", k_cal_hashmix},
      {"same multiset, the 10 fast ops in one run", k_cal_hashmix_g10}, {"same multiset, fast ops in two runs of 5", k_cal_hashmix_g5}, {"v_mad_u64_u32 only", k_cal_mad},
      {"add / alignbit, runs of 1", k_cal_alt1}, {"add / alignbit, runs of 4", k_cal_alt}, {"add / alignbit, runs of 16", k_cal_run16},
      {"mad / alignbit alternating", k_cal_madrot}};
  for (auto& c : cs) {
    hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[4];
    CHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    double cyc = (double)h[1], ghz = h[2] ? cyc / ((double)h[2] / 100e6) / 1e9 : 0;
    printf("%-44s %12.3f %20.2f\n", c.name, ghz, cyc / ((double)ITERS * 32 * waves_per_simd));
  }
  return 0;
}
