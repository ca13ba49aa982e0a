#!/bin/bash
# round 6 probe: `ecloop-hip mul -raw` over 2^L pass phrases by parse threads / contexts (is the device call slowed by the host's own memory traffic?)
L=${1:-30}; N=$((1 << L)); ROOT=$(cd "$(dirname "$0")/.." && pwd); CLI=$ROOT/ecloop_amd/host/ecloop-hip
gcc -O2 -pthread "$ROOT/tools/gen_phrases.c" -o /tmp/gen_phrases && /tmp/gen_phrases $N 11 /dev/shm/mul_raw.txt 32
$CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt >/dev/null 2>&1
run() { # label, env..., then -- extra args
  label=$1; shift
  for rep in 1 2; do
    env "$@" ECLOOP_HIP_STATS=1 $CLI mul -raw -f $ROOT/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt $EXTRA < /dev/shm/mul_raw.txt 2>/tmp/e.txt >/dev/null
    echo "$label run $rep | status: $(tr '\r' '\n' < /tmp/e.txt | grep Mkeys | tail -1)"
  done
  tr '\r' '\n' < /tmp/e.txt | grep -E "front end|mul context" | cut -c1-330 | sed "s/^/      /"
}
EXTRA=""
run "default" X=1
run "parse threads 8" ECLOOP_HIP_PARSE_THREADS=8
run "parse threads 4" ECLOOP_HIP_PARSE_THREADS=4
run "parse threads 32" ECLOOP_HIP_PARSE_THREADS=32
EXTRA="-t 3"; run "-t 3" X=1
EXTRA="-t 4"; run "-t 4" X=1
EXTRA="-t 1"; run "-t 1" X=1
rm -f /dev/shm/mul_raw.txt
