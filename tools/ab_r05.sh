#!/bin/bash
# round 5, GPU session 2: short-call geometry sweep, per-dispatch view of the mul pieces, one against two compute streams for `mul`,
# small-scalar legs, the reference-side binding's throughput.  bash tools/ab_r05.sh  -> gpurun_out/s2_*.txt
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python tools/sweep_short_calls.py > $O/s2_short_calls.txt 2>&1
tail -32 $O/s2_short_calls.txt
python tools/mul_stall_profile.py --windows 26 --passes 1 --dump --label dump > $O/s2_dump.log 2>&1
cat $O/r05_mul_stall_dump.txt | cut -c1-150
{
for rep in 1 2; do
for st in 1 2; do
  for L in 24 26; do
    echo "== ECL_HIP_MUL_STREAMS=$st  2^$L scalars"
    ECL_HIP_MUL_STREAMS=$st python tools/bench_mul.py $L 6 26 design | tail -4
  done
done
done
for topr in 8 10 12 16; do
  echo "== two streams, top R $topr, 2^24"
  ECL_HIP_MUL_TOP_R=$topr python tools/bench_mul.py 24 6 26 design | tail -3
done
for fr in 2 4; do
  echo "== two streams, first R $fr, 2^24"
  ECL_HIP_MUL_FIRST_R=$fr python tools/bench_mul.py 24 6 26 design | tail -3
done
echo "== small scalars (< 2^66), 2^24"
python tools/bench_mul.py 24 4 26 design small | tail -2
echo "== consecutive scalars from 2^65, 2^24"
python tools/bench_mul.py 24 4 26 design seq | tail -2
echo "== small scalars (< 2^66), 2^24, W = 22"
python tools/bench_mul.py 24 4 22 design small | tail -2
} > $O/s2_mul_ab.txt 2>&1
cat $O/s2_mul_ab.txt
python tools/bench_ref_binding.py > $O/s2_ref_binding.log 2>&1
cat $O/r05_ref_binding.txt
