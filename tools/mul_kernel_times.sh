#!/bin/bash
# kernel-only time of k_mul_check per scalar by piece size (scalars per thread 8 / 16 / 32): rocprofv3 --kernel-trace of
# bench.py --cmd mul on 2^26-scalar calls with the piece size fixed through ECL_HIP_MUL_TOP.   bash tools/mul_kernel_times.sh [lib.so]
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
LIB=${1:-$R/ecloop_amd/libecloop_hip.so}
cd /tmp
for top in 20 21 22; do
  rm -rf /tmp/mkt
  ECLOOP_HIP_LIB=$LIB ECL_HIP_MUL_TOP=$top ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/mkt -o t -- \
      python "$R/bench.py" --cmd mul --mul-log2 26 --steps 2 --warmup 1 > /tmp/mkt.log 2>&1
  python - "$top" <<'PY'
import csv, glob, sys, collections
top = int(sys.argv[1])
d = collections.defaultdict(list)
for f in glob.glob("/tmp/mkt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mul_check" in r["Kernel_Name"]:
            d[int(r["Grid_Size"]) if "Grid_Size" in r else 0].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for g in sorted(d):
    v = sorted(d[g])
    print("pieces of 2^%d: grid %8d threads x %4d launches: median %.3f ms  (min %.3f, max %.3f)" % (top, g, len(v), v[len(v) // 2] / 1e6, v[0] / 1e6, v[-1] / 1e6))
PY
  grep -o '"value": [0-9.]*' /tmp/mkt.log | head -1
done
