#!/bin/bash
# round 6, the one A/B of k_mul_check the review asked for: the four values a scalar parks for the batched inversion as canonical
# 8 x 32-bit words (build_ab/park_words.so: -DECL_MUL_PARK_WORDS=1, 256 bytes of parking traffic per scalar, + 4 normalisations) against
# the shipped 9 x 29-bit limbs (288 bytes), alternating, 26-bit table, design-density filter.   -> gpurun_out/r06_mul_park_words.txt
cd "$(dirname "$0")/.."
{
echo "# tools/ab_r06_park.sh: device rate of the last 3 of 5 calls, M scalars/s (tools/bench_mul.py <log2> 5 26 design)"
for rep in 1 2 3; do
  for lib in shipped build_ab/park_words.so; do
    path=$PWD/$lib; [ "$lib" = shipped ] && path=$PWD/ecloop_amd/libecloop_hip.so
    for L in 24 26; do
      echo "== $lib  2^$L scalars"
      ECLOOP_HIP_LIB=$path python tools/bench_mul.py $L 5 26 design | tail -3
    done
  done
done
} > gpurun_out/r06_mul_park_words.txt 2>&1
cat gpurun_out/r06_mul_park_words.txt
