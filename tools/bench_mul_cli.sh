#!/bin/bash
# `mul` end to end through the C host program: N seeded 64-hex-digit scalars on stdin (BASELINE.json configs[4]).
N=${1:-4194304}
python3 - "$N" > /tmp/mul_in.txt <<'PY'
import sys, random
r = random.Random(7)
n = int(sys.argv[1])
sys.stdout.write("".join("%064x\n" % r.getrandbits(256) for _ in range(n)))
PY
ROOT=$(cd "$(dirname "$0")/.." && pwd)
time "$ROOT/ecloop_amd/host/ecloop-hip" mul -f "$ROOT/tests/golden/btc-bw-hash" -a cu -q -o /tmp/mul_out.txt < /tmp/mul_in.txt 2>&1 | tr '\r' '\n' | tail -1
