#!/bin/bash
# round 6 probe: `ecloop-hip mul -raw` over 2^30 pass phrases by the number of hardware queues the runtime spreads the streams over
ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip; N=$((1 << 30))
gcc -O2 -pthread "$ROOT/tools/gen_phrases.c" -o /tmp/gen_phrases && /tmp/gen_phrases $N 11 /dev/shm/mul_raw.txt 32
$CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt >/dev/null 2>&1
for q in default 2 4 8 16; do
  for rep in 1 2 3; do
    if [ $q = default ]; then ECLOOP_HIP_STATS=1 $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null
    else GPU_MAX_HW_QUEUES=$q ECLOOP_HIP_STATS=1 $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null; fi
    echo "queues $q run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
  done
  tr '\r' '\n' < /tmp/e.txt | grep -E "mul context" | cut -c1-200 | sed "s/^/      /"
done
rm -f /dev/shm/mul_raw.txt
