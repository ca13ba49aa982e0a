#!/bin/bash
# kernel-only time of the `mul` kernels per piece (scalars per chain 8 / 16; one-kernel builds: k_mul_check, two-kernel builds: k_mul_sum +
# k_mul_finish): rocprofv3 --kernel-trace of bench.py --cmd mul on 2^26-scalar calls with the piece size fixed through ECL_HIP_MUL_TOP_R.
#   bash tools/mul_kernel_times.sh [lib.so ...]        (W=22; WIDTH=24 for another table)
export TMPDIR=/tmp
R=$(cd "$(dirname "$0")/.." && pwd)
[ $# -eq 0 ] && set -- "$R/ecloop_amd/libecloop_hip.so"
cd /tmp
for LIB in "$@"; do
  case "$LIB" in /*) ;; *) LIB=$R/$LIB;; esac
  for top in ${TOPS:-8 16}; do
    rm -rf /tmp/mkt
    ECLOOP_HIP_LIB=$LIB ECL_HIP_MUL_TOP_R=$top ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/mkt -o t -- \
        python "$R/bench.py" --cmd mul --mul-log2 26 --mul-window ${WIDTH:-22} --steps 2 --warmup 1 > /tmp/mkt.log 2>&1
    python - "$top" "$(basename $LIB)" <<'PY'
import csv, glob, sys, collections, json
top, lib = int(sys.argv[1]), sys.argv[2]
d = collections.defaultdict(list)
for f in glob.glob("/tmp/mkt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("k_mul_check", "k_mul_sum", "k_mul_finish"):
            if k in n:
                d[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
try:
    line = json.loads(open("/tmp/mkt.log").read().strip().splitlines()[-1])
    whole = line["value"]
except Exception:
    whole = float("nan")
tot = 0.0
parts = []
for k, v in sorted(d.items()):
    v = sorted(v)
    med = v[len(v) // 2] / 1e6
    tot += med
    parts.append("%s median %.3f ms x %d" % (k, med, len(v)))
print("%-22s R=%-2d  %s  | kernels per full piece %.3f ms | whole-call under the tracer %.1f M scalars/s" % (lib, top, "; ".join(parts), tot, whole))
PY
  done
done
