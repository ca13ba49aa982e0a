#!/bin/bash
# round 6 probe: `ecloop-hip mul -raw` over 2^30 pass phrases: priority of the hashing stream, and the same run under rocprofv3
ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip; N=$((1 << 30))
cd /tmp && export TMPDIR=/tmp
gcc -O2 -pthread "$ROOT/tools/gen_phrases.c" -o /tmp/gen_phrases && /tmp/gen_phrases $N 11 /dev/shm/mul_raw.txt 32
$CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt >/dev/null 2>&1
for v in "X=1" "ECL_HIP_MUL_STREAMS=1" "HSA_ENABLE_INTERRUPT=0"; do
  for rep in 1 2 3; do
    env $v ECLOOP_HIP_STATS=1 $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null
    echo "$v run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
  done
  tr '\r' '\n' < /tmp/e.txt | grep -E "mul context" | cut -c1-200 | sed "s/^/      /"
done
for rep in 1 2; do
  rocprofv3 --kernel-trace -d /tmp/rp -o x -- $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null
  echo "under rocprofv3 --kernel-trace run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
done
rm -f /dev/shm/mul_raw.txt
