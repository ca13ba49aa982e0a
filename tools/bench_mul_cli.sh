#!/bin/bash
# `mul` end to end through the C host program (BASELINE.json configs[4]): N seeded 64-hex-digit lines (the reference's input format,
# tools/gen_hex_lines.c) in a file on tmpfs, fed (a) as a regular file on stdin, (b) through a pipe (`cat file | ecloop-hip mul`, the
# reference's usual form, main.c:542-576) and (c) as 32-byte scalars (`-bin`, regular file); rates by the status line (clock starts after
# bring-up, like the reference's) and by the wall clock of the whole process.
#   bash tools/bench_mul_cli.sh [LOG2_N=30] [REPS=3] [DIR=/dev/shm]      (2^30 lines = 70 GB of text: a timed window of about a second)
L=${1:-30}; REPS=${2:-3}; DIR=${3:-/dev/shm}
N=$((1 << L))
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CLI="$ROOT/ecloop_amd/host/ecloop-hip"
gcc -O2 -pthread "$ROOT/tools/gen_hex_lines.c" -o /tmp/gen_hex_lines || exit 1
t0=$(date +%s.%N); /tmp/gen_hex_lines $N 7 $DIR/mul_in.txt 64; t1=$(date +%s.%N)
echo "# tools/bench_mul_cli.sh $L $REPS $DIR: 2^$L lines ($(du -h $DIR/mul_in.txt | cut -f1)) generated in $(python3 -c "print('%.1f' % ($t1 - $t0))") s; host: $(nproc) hardware threads, $(free -g | awk '/Mem:/{print $2}') GB RAM"
run() { # label, stdin form, extra env / flags
  local label=$1 form=$2; shift 2
  for rep in $(seq 1 $REPS); do
    t0=$(date +%s.%N)
    if [ $form = pipe ]; then cat $DIR/mul_in.txt | env ECLOOP_HIP_STATS=1 "$@" "$CLI" mul -f "$ROOT/tests/golden/btc-bw-hash" -a cu -q -o /tmp/mul_out.txt 2>/tmp/mul_err.txt >/dev/null
    else env ECLOOP_HIP_STATS=1 "$@" "$CLI" mul -f "$ROOT/tests/golden/btc-bw-hash" -a cu -q -o /tmp/mul_out.txt < $DIR/mul_in.txt 2>/tmp/mul_err.txt >/dev/null; fi
    t1=$(date +%s.%N)
    st=$(tr '\r' '\n' < /tmp/mul_err.txt | grep Mkeys | tail -1)
    [ $rep = 1 ] && tr "\r" "\n" < /tmp/mul_err.txt | grep -E "front end|mul context" | sed "s/^/      /"
    echo "$label run $rep: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s | status line: $st"
  done
}
echo "# (one untimed pass first: the first read of freshly written tmpfs pages runs at a fifth of the rate of the following ones)"
env "$CLI" mul -f "$ROOT/tests/golden/btc-bw-hash" -a cu -q -o /tmp/mul_out.txt < $DIR/mul_in.txt >/dev/null 2>&1
run "file  (pread, default)         " file
run "pipe  (cat file | ecloop-hip)   " pipe
run "file  (mmap form, round 5)      " file ECLOOP_HIP_MUL_READ=mmap
run "file  (general reader, 64 MB)   " file ECLOOP_HIP_MUL_READ=chunks
run "file  (AVX2 decoder)            " file ECLOOP_HIP_NO_AVX512=1
for T in 8 24 32; do run "file  (pread, $T parse threads)  " file ECLOOP_HIP_PARSE_THREADS=$T; done
# the same scalars as 32-byte little-endian records (`-bin`, not in the reference: for feeders that can produce more than text parsing takes)
python3 - $DIR/mul_in.txt $DIR/mul_in.bin <<'PY'
import os
import sys
import numpy as np
src, dst = sys.argv[1], sys.argv[2]
n = min(1 << 28, os.path.getsize(src) // 65)
t = np.fromfile(src, dtype=np.uint8, count=n * 65).reshape(n, 65)[:, :64]
v = np.where(t >= 97, t - 87, t - 48).astype(np.uint8)
b = (v[:, 0::2] << 4 | v[:, 1::2])[:, ::-1]   # big-endian digits -> little-endian limbs
np.ascontiguousarray(b).tofile(dst)
PY
for rep in $(seq 1 $REPS); do
  t0=$(date +%s.%N); env ECLOOP_HIP_STATS=1 "$CLI" mul -f "$ROOT/tests/golden/btc-bw-hash" -a cu -bin -q -o /tmp/mul_out.txt < $DIR/mul_in.bin 2>/tmp/mul_err.txt >/dev/null; t1=$(date +%s.%N)
  echo "file  (-bin, the first 2^28 scalars) run $rep: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s | status line: $(tr '\r' '\n' < /tmp/mul_err.txt | grep Mkeys | tail -1)"
done
rm -f $DIR/mul_in.bin
echo "# the front end alone (hidden command \`parse\`, nothing printed, no GPU):"
for T in 8 16 32; do
  for m in pread mmap; do
    t0=$(date +%s.%N); ECLOOP_HIP_PARSE_QUIET=1 ECLOOP_HIP_MUL_READ=$m ECLOOP_HIP_PARSE_THREADS=$T "$CLI" parse < $DIR/mul_in.txt >/dev/null 2>&1; t1=$(date +%s.%N)
    echo "parse only, $m, $T threads: $(python3 -c "print('%.0f M lines/s' % ($N / ($t1 - $t0) / 1e6))")"
  done
done
rm -f $DIR/mul_in.txt
