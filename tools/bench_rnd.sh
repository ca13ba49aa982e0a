#!/bin/bash
# `rnd` window loop on one GPU (BASELINE configs[3] shape): N random windows of 2^SIZE keys at stride 2^128 over a
# 168-bit range, synthetic 56 MB filter at the .blf design density.  Prints the host program's per-window lines and, at the end, how the device
# time splits between the search kernel and the per-window set-up (ECLOOP_HIP_STATS).
#   bash tools/bench_rnd.sh [SIZE=29] [WINDOWS=40]
SIZE=${1:-29}; N=${2:-40}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python3 - <<PY
import sys
sys.path.insert(0, "$ROOT/tests")
from synth import synth_bloom_words, write_blf
write_blf("/tmp/rnd_bench.blf", synth_bloom_words(7000003, 23, "a&(b|c)"))  # 56 MB at the design density: ~13 false positives per 2^32 keys
PY
t0=$(date +%s.%N)
ECLOOP_HIP_RND_WINDOWS=$N ECLOOP_HIP_STATS=1 "$ROOT/ecloop_amd/host/ecloop-hip" rnd -f /tmp/rnd_bench.blf \
  -r 8000000000000000000000000000000001234567:ffffffffffffffffffffffffffffffffff89abcdef -d 128:$SIZE -seed bench -t 1 -q -o /tmp/rnd_bench_out.txt 2>/tmp/rnd_bench.err | grep -v '^[0-9a-f ]\{67\}$' | tail -12
t1=$(date +%s.%N)
python3 -c "print('wall %.2f s for $N windows of 2^$SIZE keys: %.2f windows/s, %.1f Mkeys/s end to end (process start-up included)' % ($t1-$t0, $N/($t1-$t0), $N*2**$SIZE/($t1-$t0)/1e6))"
tr '\r' '\n' < /tmp/rnd_bench.err | tail -1
