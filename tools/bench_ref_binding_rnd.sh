#!/bin/bash
# the reference's own `rnd` (cmd_rnd, main.c:619-662: random 2^32-key windows at stride 2^128, its workers pulling 2^21-key jobs) bound to
# the library, for a few seconds: keys per second per window by the reference's own per-window summary line.  With and without the look-ahead.
#   bash tools/bench_ref_binding_rnd.sh [seconds=12]     -> gpurun_out/r06_ref_binding_rnd.txt
S=${1:-12}; ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from synth import synth_bloom_words, write_blf
write_blf("/tmp/rnd_bench.blf", synth_bloom_words(7000003, 23, "a&(b|c)"))   # 56 MB at the design density
PY
{
echo "# tools/bench_ref_binding_rnd.sh $S: oracle/_ref/ecloop_gpu rnd -d 128:32 (MAX_JOB_SIZE unchanged), 56 MB .blf, one MI355X; per-window lines of the reference"
for la in 30 0; do
  echo "== ECL_HIP_LOOKAHEAD_LOG2=$la"
  ECL_HIP_LOOKAHEAD_LOG2=$la timeout $S stdbuf -oL oracle/_ref/ecloop_gpu rnd -f /tmp/rnd_bench.blf -r 800000000000000000000000000000000001234567:fffffffffffffffffffffffffffffffffff89abcdf -d 128:32 -t 1 -q -o /tmp/rnd_out.txt 2>/dev/null | python3 -u -c "
import sys, time
stamps = []
for l in sys.stdin:
    if ' / ' in l and '~' in l:
        stamps.append((time.time(), int(l.replace(',', '').split()[2])))
if len(stamps) > 2:
    keys = sum(k for _, k in stamps[1:])
    print('%d windows of %d keys in %.2f s (first to last summary line): %.0f Mkeys/s' % (len(stamps) - 1, stamps[0][1], stamps[-1][0] - stamps[0][0], keys / (stamps[-1][0] - stamps[0][0]) / 1e6))
else:
    print('too few windows', len(stamps))"
done
} | tee gpurun_out/r06_ref_binding_rnd.txt
