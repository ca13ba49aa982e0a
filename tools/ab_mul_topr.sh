#!/bin/bash
# top piece size of ecl_hip_mul_batch (scalars per resident thread; 0 = the library's choice) by call size:  bash tools/ab_mul_topr.sh "24 25" "0 8 10 12 16"
cd "$(dirname "$0")/.."
for rep in 1 2; do
for L in ${1:-24 25}; do
for R in ${2:-0 8 10 12 16}; do
  st=12; [ $L -ge 26 ] && st=4
  env ECL_HIP_MUL_TOP_R=$R python3 bench.py --cmd mul --steps $st --warmup 2 --mul-log2 $L 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readlines()[-1])
print('top_R=%-3s 2^$L %9.1f Mscalars/s whole-call  %9.1f device' % ('$R', r['value'], r['roofline']['device_mscalars_s']))"
done; done; done
