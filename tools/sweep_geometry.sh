#!/bin/bash
# kernel rate of short calls (the per-GPU shard of a 2^32-key window on 8 GPUs, a `rnd -d x:29` window) by walk geometry
#   bash tools/sweep_geometry.sh [keys_log2=29]
L=${1:-29}
for geo in "0 0" "128 0" "128 1048576" "128 2097152" "256 0" "256 1048576" "256 524288" "512 0" "512 524288" "512 262144" "1024 0" "64 0"; do
  set -- $geo
  python bench.py --keys-log2 $L --steps 10 --warmup 2 --no-cpu --no-secondary --half-group $1 --lanes $2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.readline()); f=r['roofline']
print('half_group %5s lanes %8s: %9.1f Mkeys/s whole step, kernel %9.1f Mkeys/s, %.3f ms per launch, set-up %.3f ms' % ('$1','$2', r['value'], f['kernel_mkeys_s'], f['ms_per_launch'], r['config']['setup_ms_per_step_on_device']))"
done
