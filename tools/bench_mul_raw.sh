#!/bin/bash
# `mul -raw` end to end through the C host program: N seeded pass-phrase-like lines (8..24 characters) on stdin; every
# line's SHA-256 is the scalar (main.c:503-527).   bash tools/bench_mul_raw.sh [N=33554432]
N=${1:-33554432}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
gcc -O2 -pthread "$ROOT/tools/gen_phrases.c" -o /tmp/gen_phrases && /tmp/gen_phrases $N 11 /dev/shm/mul_raw.txt 32
"$ROOT/ecloop_amd/host/ecloop-hip" mul -raw -f "$ROOT/tests/golden/btc-bw-hash" -a cu -q -o /tmp/mul_out.txt < /dev/shm/mul_raw.txt >/dev/null 2>&1
for rep in 1 2 3; do
  t0=$(date +%s.%N)
  ECLOOP_HIP_STATS=1 "$ROOT/ecloop_amd/host/ecloop-hip" mul -raw -f "$ROOT/tests/golden/btc-bw-hash" -a cu -q -o /tmp/mul_out.txt < /dev/shm/mul_raw.txt 2>/tmp/mul_err.txt >/dev/null
  t1=$(date +%s.%N)
  st=$(tr '\r' '\n' < /tmp/mul_err.txt | grep Mkeys | tail -1)
  tr "\r" "\n" < /tmp/mul_err.txt | grep -E "front end|mul context" | sed "s/^/      /"
  echo "raw run $rep, $N lines: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s | status line: $st"
done
rm -f /dev/shm/mul_raw.txt
