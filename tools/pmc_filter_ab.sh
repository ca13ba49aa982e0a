#!/bin/bash
# TCP miss-path counters of add-kernel A/B builds against the 5.9 GB filter (round 4: quarter-wave stage-1 probe, 2 / 3 waves per
# SIMD): one 2^32-key launch per pass.   bash tools/pmc_filter_ab.sh lib1.so lib2.so ...   ("shipped" is always run)
set -u
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/pmc_filter_ab
rm -rf "$O"; mkdir -p "$O"
cd /tmp
for lib in shipped "$@"; do
  path=$R/$lib; [ "$lib" = shipped ] && path=$R/ecloop_amd/libecloop_hip.so
  for set in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    ECLOOP_HIP_LIB=$path ECL_HIP_SKIP_SELFTEST=1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/p" -o p -- \
        python "$R/bench.py" --steps 1 --warmup 0 --no-cpu --no-secondary --filter-n 1100000000 > "$O/p.log" 2>&1
    python - "$O/p" "$lib" <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_add" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
ns = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
      for r in csv.DictReader(open(f)) if "k_add" in r["Kernel_Name"]]
line = "%-26s launch %.1f ms" % (sys.argv[2], max(ns) / 1e6 if ns else 0)
if "TCP_PENDING_STALL_CYCLES_sum" in acc and acc.get("TCP_GATE_EN1_sum"):
    line += "  TCP_PENDING_STALL %.1f %% of TCP cycles, L1->L2 read latency %.0f cycles" % (
        100 * acc["TCP_PENDING_STALL_CYCLES_sum"] / acc["TCP_GATE_EN1_sum"], acc["TCP_TCC_READ_REQ_LATENCY_sum"] / max(acc["TCP_TCC_READ_REQ_sum"], 1))
if "SQ_WAIT_INST_ANY" in acc and acc.get("SQ_WAVE_CYCLES"):
    line += "  waves waiting %.1f %% of wave cycles" % (100 * acc["SQ_WAIT_INST_ANY"] / acc["SQ_WAVE_CYCLES"])
print(line)
PY
    rm -rf "$O/p"
  done
done
