#!/bin/bash
# A/B of add-kernel builds on the headline bench: tools/ab_bench.sh "<filter-n list>" lib1.so lib2.so ...
# (libraries built with a -D switch into build_ab/, selected through ECLOOP_HIP_LIB; "shipped" = the in-tree library).
# Prints Mkeys/s and the kernel's ms per 2^32-key launch for every (library, filter size).  Run on the GPU box.
cd "$(dirname "$0")/.."
FILTERS=${1:-10000000}; shift
for lib in shipped "$@"; do
  path=$PWD/$lib; [ "$lib" = shipped ] && path=$PWD/ecloop_amd/libecloop_hip.so
  for n in $FILTERS; do
    ECLOOP_HIP_LIB=$path python3 bench.py --no-cpu --no-secondary --steps ${STEPS:-4} --warmup 1 --filter-n $n 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1])
print('%-34s filter-n %-11s %9.1f Mkeys/s  %8.3f ms/launch' % ('$lib', '$n', r['value'], r['roofline']['ms_per_launch']))"
  done
done
