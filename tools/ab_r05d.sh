#!/bin/bash
# round 5: one against two compute streams for `mul` through the C host program (2^26 hex lines / -bin on stdin: 1-2 M-scalar batches on
# two contexts per GPU) and through the API at the batch sizes the host program uses.   -> gpurun_out/s7_mul_cli_streams.txt
export TMPDIR=/tmp
O=gpurun_out
{
for rep in 1 2; do
for st in 1 2; do
  echo "=== ECL_HIP_MUL_STREAMS=$st  ecloop-hip mul, 2^26 scalars"
  ECL_HIP_MUL_STREAMS=$st bash tools/bench_mul_cli.sh 67108864 2>&1 | grep -E "run [12]"
done
done
for st in 1 2; do
  for L in 20 21 22; do
    echo "=== ECL_HIP_MUL_STREAMS=$st  API calls of 2^$L scalars, automatic window (22 bits)"
    ECL_HIP_MUL_STREAMS=$st python tools/bench_mul.py $L 8 0 design | tail -3
  done
done
} > $O/s7_mul_cli_streams.txt 2>&1
cat $O/s7_mul_cli_streams.txt
