#!/bin/bash
# round 4: inversion by division steps against the addition chain - microbenchmark, then A/B on the `mul` bench and on the add bench
# (2^32-key and 2^29-key calls), builds alternating so that a drifting clock hits all of them.  Run on the GPU box:
#   bash tools/ab_inv_r04.sh build_ab/r04_xyzz_fermat.so [build_ab/r04_jac.so ...]
cd "$(dirname "$0")/.."
echo "## inv_bench"; ecloop_amd/csrc/tools/inv_bench
for rep in 1 2 3; do
  echo "## mul, round $rep"
  STEPS=8 LOG2=24 tools/ab_mul.sh 22 "$@"
  STEPS=4 LOG2=26 tools/ab_mul.sh 22 "$@"
done
for rep in 1 2; do
  echo "## add 2^32 keys, round $rep"
  STEPS=4 tools/ab_bench.sh 10000000 "$1"
  echo "## add 2^29-key calls, round $rep"
  for lib in shipped "$1"; do
    path=$PWD/$lib; [ "$lib" = shipped ] && path=$PWD/ecloop_amd/libecloop_hip.so
    ECLOOP_HIP_LIB=$path python3 bench.py --no-cpu --no-secondary --keys-log2 29 --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1])
print('%-34s 2^29 keys %9.1f Mkeys/s whole step  kernel %9.1f  %8.3f ms/launch' % ('$lib', r['value'], r['roofline']['kernel_mkeys_s'], r['roofline']['ms_per_launch']))"
  done
done
