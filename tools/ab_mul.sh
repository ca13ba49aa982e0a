#!/bin/bash
# A/B of `mul` builds / window widths on bench.py --cmd mul (2^24 scalars per call, -a cu, page-locked, empty filter):
#   tools/ab_mul.sh "<window list>" lib1.so lib2.so ...     ("shipped" = the in-tree library is always run)
cd "$(dirname "$0")/.."
WIDTHS=${1:-22}; shift
for lib in shipped "$@"; do
  path=$PWD/$lib; [ "$lib" = shipped ] && path=$PWD/ecloop_amd/libecloop_hip.so
  for w in $WIDTHS; do
    ECLOOP_HIP_LIB=$path python3 bench.py --cmd mul --steps ${STEPS:-6} --warmup 2 --mul-window $w --mul-log2 ${LOG2:-24} 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1])
print('%-28s W=%-3s 2^${LOG2:-24} %9.1f Mscalars/s whole-call  %9.1f device  first call %7.1f ms' % ('$lib', '$w', r['value'], r['roofline']['device_mscalars_s'], r['config']['first_call_ms_incl_table_build']))"
  done
done
