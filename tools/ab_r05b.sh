#!/bin/bash
# round 5, GPU session 3: (1) the bound on what ANY deeper prefetch of `mul`'s table points could gain - the shipped library against a
# measurement build whose gathers all land in cache (-DECL_MUL_HOT_GATHERS=1: same instructions, same loads, wrong points), alternating;
# (2) short add_range calls with the automatic geometry; (3) the reference-side binding's throughput.   -> gpurun_out/s3_*.txt
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
{
for rep in 1 2 3; do
  for lib in shipped build_ab/hot_gathers.so; do
    path=$PWD/$lib; [ "$lib" = shipped ] && path=$PWD/ecloop_amd/libecloop_hip.so
    for L in 24 26; do
      echo "== $lib  2^$L scalars"
      ECLOOP_HIP_LIB=$path python tools/bench_mul.py $L 5 26 design | tail -3
    done
  done
done
} > $O/s3_hot_gathers.txt 2>&1
cat $O/s3_hot_gathers.txt
python tools/sweep_short_calls.py 21,22,23,24,25,26,28 0 > $O/s3_short_calls_auto.txt 2>&1
cat $O/s3_short_calls_auto.txt
python tools/bench_ref_binding.py > $O/s3_ref_binding.log 2>&1
cat $O/r05_ref_binding.txt
