#!/bin/bash
# round 6 probe: `ecloop-hip mul -raw` over 2^L pass phrases (tools/gen_phrases.c) by -raw chunk size; and the GPU tests that touch -raw
L=${1:-30}; N=$((1 << L)); ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip
gcc -O2 -pthread "$ROOT/tools/gen_phrases.c" -o /tmp/gen_phrases && /tmp/gen_phrases $N 11 /dev/shm/mul_raw.txt 32
$CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt >/dev/null 2>&1
for chunk in 0 33554432 67108864 134217728; do
  for rep in 1 2 3; do
    t0=$(date +%s.%N)
    if [ $chunk = 0 ]; then ECLOOP_HIP_STATS=1 $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null
    else ECLOOP_HIP_MUL_RAW_CHUNK=$chunk ECLOOP_HIP_STATS=1 $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null; fi
    t1=$(date +%s.%N)
    echo "raw chunk $chunk 2^$L lines run $rep: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
    [ $rep = 3 ] && tr '\r' '\n' < /tmp/e.txt | grep -E "front end|mul context" | cut -c1-330 | sed "s/^/      /"
  done
done
rm -f /dev/shm/mul_raw.txt
