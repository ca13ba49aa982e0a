#!/bin/bash
# round 4: `mul` A/B - waves per SIMD of k_mul_check (launch bounds 2 / 3 / 4 = 256 / 168 / 128 VGPRs; 65536 threads per wave-per-SIMD),
# with (w3, w4) and without (w3n, w4n) the affine + affine first addition
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04j
{
echo "## 2^24-scalar calls, 12 steps"; STEPS=12 bash tools/ab_mul.sh "22" build_ab/r04_mul_w3.so build_ab/r04_mul_w4.so build_ab/r04_mul_w3n.so build_ab/r04_mul_w4n.so build_ab/r04_base.so
echo "## 2^26-scalar calls"; LOG2=26 STEPS=4 bash tools/ab_mul.sh "22" build_ab/r04_mul_w3.so build_ab/r04_mul_w4.so build_ab/r04_mul_w3n.so build_ab/r04_mul_w4n.so build_ab/r04_base.so
} 2>&1 | tee gpurun_out/r04j/mul_waves.txt
