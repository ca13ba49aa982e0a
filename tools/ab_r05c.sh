#!/bin/bash
# round 5, GPU session 4: `mul` with the next window's table point staged in LDS (shipped) against the register form of round 4
# (build_ab/reg_gather.so, -DECL_MUL_LDS_GATHER=0) and against the cache-resident bound (build_ab/hot_gathers.so), alternating;
# then the stall counters of the shipped kernel.   -> gpurun_out/s4_*.txt
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
{
for rep in 1 2 3; do
  for lib in shipped build_ab/digits_lds.so build_ab/reg_gather.so build_ab/hot_gathers.so; do
    path=$PWD/$lib; [ "$lib" = shipped ] && path=$PWD/ecloop_amd/libecloop_hip.so
    for L in 24 26; do
      echo "== $lib  2^$L scalars"
      ECLOOP_HIP_LIB=$path python tools/bench_mul.py $L 5 26 design | tail -3
    done
  done
done
for w in 22 24; do
  echo "== shipped, W = $w, 2^24"
  python tools/bench_mul.py 24 5 $w design | tail -2
done
} > $O/s4_lds_gather.txt 2>&1
cat $O/s4_lds_gather.txt
python tools/mul_stall_profile.py --windows 26 --passes 4 --label lds > $O/s4_stall.log 2>&1
grep -A16 "^derived" $O/r05_mul_stall_lds.txt
# what is left to wait for when the gathers are cache-resident (the measurement build under the same counters), and the clock it runs at
ECLOOP_HIP_LIB=$PWD/build_ab/hot_gathers.so python tools/mul_stall_profile.py --windows 26 --passes 4 --label hot > $O/s4_stall_hot.log 2>&1
grep -A12 "^derived" $O/r05_mul_stall_hot.txt
