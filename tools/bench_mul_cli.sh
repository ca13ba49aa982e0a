#!/bin/bash
# `mul` end to end through the C host program (BASELINE.json configs[4]): N seeded 256-bit scalars on stdin, once as
# 64-hex-digit lines (the reference's input format) and once as 32-byte little-endian scalars (`-bin`).
#   bash tools/bench_mul_cli.sh [N=16777216]
N=${1:-16777216}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 - "$N" <<'PY'
import sys
import numpy as np
n = int(sys.argv[1])
b = np.frombuffer(np.random.default_rng(7).bytes(n * 32), dtype=np.uint8).reshape(n, 32)
hexd = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)
t = np.empty((n, 65), dtype=np.uint8)
t[:, 0:64:2], t[:, 1:64:2], t[:, 64] = hexd[b >> 4], hexd[b & 15], 10
t.tofile("/tmp/mul_in.txt")
b[:, ::-1].copy().tofile("/tmp/mul_in.bin")   # big-endian hex digits -> little-endian limbs
PY
for mode in txt bin; do
  flag=""; [ $mode = bin ] && flag="-bin"
  for rep in 1 2; do
    t0=$(date +%s.%N)
    ECLOOP_HIP_STATS=1 "$ROOT/ecloop_amd/host/ecloop-hip" mul -f "$ROOT/tests/golden/btc-bw-hash" -a cu $flag -q -o /tmp/mul_out.txt < /tmp/mul_in.$mode 2>/tmp/mul_err.txt >/dev/null
    t1=$(date +%s.%N)
    st=$(tr '\r' '\n' < /tmp/mul_err.txt | grep Mkeys | tail -1)
    tr "\r" "\n" < /tmp/mul_err.txt | grep -E "front end|setup|mul:" | sed "s/^/      /"
    echo "$mode run $rep, $N scalars: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s (process start-up and GPU bring-up included) | status line: $st"
  done
done
