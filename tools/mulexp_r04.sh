#!/bin/bash
# round 4: `mul` A/B - lazy additions with / without the affine + affine first addition, 2 / 3 waves per SIMD, window widths
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04f
{
echo "## default schedule, 2^24, 12 steps"; STEPS=12 bash tools/ab_mul.sh "22 24" build_ab/r04_mul_nommadd.so build_ab/r04_base.so
echo "## default schedule, 2^26"; LOG2=26 STEPS=4 bash tools/ab_mul.sh "22 24" build_ab/r04_mul_nommadd.so build_ab/r04_base.so
echo "## 3 waves per SIMD (196608 threads), 2^24 and 2^26"; ECL_HIP_MUL_NT=196608 STEPS=12 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
ECL_HIP_MUL_NT=196608 LOG2=26 STEPS=4 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
echo "## pieces of 2^21 from the second on, 2^24"; ECL_HIP_MUL_TOP=21 STEPS=12 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
} 2>&1 | tee gpurun_out/r04f/mul_ab.txt
bash tools/pmc_filter_ab.sh build_ab/r04_quarter.so build_ab/r04_waves2.so 2>&1 | tee gpurun_out/r04f/pmc_filter_ab.txt
