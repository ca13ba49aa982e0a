#!/bin/bash
# `mul` piece schedules (first piece / growth / top, in units of one scalar per resident thread) and staging-buffer counts on bench.py --cmd mul:
#   bash tools/ab_mul_sched.sh [lib-with-another-MUL_NBUF.so]
cd "$(dirname "$0")/.."
run() {  # label, lib, env...
  label=$1; lib=$2; shift 2
  for L in ${LOGS:-24 26}; do
    st=12; [ $L -ge 26 ] && st=4
    env "$@" ECLOOP_HIP_LIB=$lib python3 bench.py --cmd mul --steps $st --warmup 2 --mul-log2 $L 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1])
print('%-44s 2^$L %9.1f Mscalars/s whole-call  %9.1f device' % ('$label', r['value'], r['roofline']['device_mscalars_s']))"
  done
}
S=$PWD/ecloop_amd/libecloop_hip.so
for rep in 1 2; do
  [ -n "$1" ] && run "other NBUF, first 2 x16 top 8/16" $PWD/$1 A=1
  run "shipped default" $S A=1
  run "first 2 x16 top 8/16 (round 3)" $S ECL_HIP_MUL_FIRST_R=2 ECL_HIP_MUL_GROW=1600
  run "first 1 x2 top 8/16" $S ECL_HIP_MUL_FIRST_R=1 ECL_HIP_MUL_GROW=200
  run "first 2 x2 top 8/16" $S ECL_HIP_MUL_FIRST_R=2 ECL_HIP_MUL_GROW=200
  run "first 1 x2 top 16" $S ECL_HIP_MUL_FIRST_R=1 ECL_HIP_MUL_GROW=200 ECL_HIP_MUL_TOP_R=16
  run "first 2 x4 top 16" $S ECL_HIP_MUL_FIRST_R=2 ECL_HIP_MUL_GROW=400 ECL_HIP_MUL_TOP_R=16
  run "first 4 x2 top 16" $S ECL_HIP_MUL_FIRST_R=4 ECL_HIP_MUL_GROW=200 ECL_HIP_MUL_TOP_R=16
done
