#!/bin/bash
# `mul -raw` end to end through the C host program: N seeded pass-phrase-like lines (8..24 characters) on stdin; every
# line's SHA-256 is the scalar (main.c:503-527).   bash tools/bench_mul_raw.sh [N=33554432]
N=${1:-33554432}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 - "$N" <<'PY'
import sys
import numpy as np
n = int(sys.argv[1])
rng = np.random.default_rng(11)
ln = rng.integers(8, 25, n)                      # characters per line
tot = int(ln.sum()) + n
alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)
buf = alphabet[rng.integers(0, len(alphabet), tot)]
ends = np.cumsum(ln + 1) - 1
buf[ends] = 10
buf.tofile("/tmp/mul_raw.txt")
print("lines", n, "bytes", tot)
PY
for rep in 1 2; do
  t0=$(date +%s.%N)
  ECLOOP_HIP_STATS=1 "$ROOT/ecloop_amd/host/ecloop-hip" mul -raw -f "$ROOT/tests/golden/btc-bw-hash" -a cu -q -o /tmp/mul_out.txt < /tmp/mul_raw.txt 2>/tmp/mul_err.txt >/dev/null
  t1=$(date +%s.%N)
  st=$(tr '\r' '\n' < /tmp/mul_err.txt | grep Mkeys | tail -1)
  tr "\r" "\n" < /tmp/mul_err.txt | grep -E "front end" | sed "s/^/      /"
  echo "raw run $rep, $N lines: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s | status line: $st"
done
