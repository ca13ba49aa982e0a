#!/bin/bash
# Collects the measured evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh [round-tag, default r02]
# -> gpurun_out/prof_<tag>/{bench.json, stats.txt, pmc.txt, calib.txt, ubench8.txt, ubench4.txt, <tag>_roofline.json}
# Counter passes are separate runs with --kernel-trace only (never --pmc together with sys/hip/hsa tracing), one
# 2^32-key launch each (ECL_HIP_SKIP_SELFTEST=1: no 4096-key self-test launch in the counters).
# Copy what is to be kept into profiles/ (tracked); bench.py reads profiles/<tag>_roofline.json.
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
T=ecloop_amd/csrc/tools
[ -x $T/ubench ] || $HIPCC --offload-arch=gfx950 -O3 $T/ubench.hip -o $T/ubench
[ -x $T/fetch_calib ] || $HIPCC --offload-arch=gfx950 -O3 $T/fetch_calib.hip -o $T/fetch_calib
$T/ubench 8 > "$O/ubench8.txt" 2>&1
$T/ubench 4 > "$O/ubench4.txt" 2>&1

cd /tmp
# per-kernel time of the default bench command
# (ECL_HIP_SKIP_SELFTEST=1: without the 4096-key self-test launch every k_add launch in the trace is a 2^32-key one, so the
#  kernel's average duration in the summary is directly comparable with bench.py's roofline.ms_per_launch)
ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --stats -d "$O/stats" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu > "$O/stats.log" 2>&1
db=$(find "$O/stats" -name '*.db' | head -1)
[ -n "$db" ] && python "$R/tools/rocprof_summary.py" "$db" > "$O/stats.txt"
rm -rf "$O/stats"

summ() {  # counter csv (+ kernel trace csv) of one pass -> "COUNTER sum dispatches" lines for kernels matching $2
python - "$1" "$2" <<'PY'
import csv, glob, os, sys, collections
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        acc[key] += float(r["Counter_Value"]); n[key] += 1
for (k, c) in sorted(acc): print("PMC %s %s %.0f %d" % (k.replace(" ", ""), c, acc[(k, c)], n[(k, c)]))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        print("TRACE %s %d" % (r["Kernel_Name"].split("(")[0][:60].replace(" ", ""), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
PY
}

: > "$O/pmc.txt"
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "VALUUtilization"; do
  i=$((i+1))
  ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/pmc$i" -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu > "$O/pmc$i.log" 2>&1
  echo "# pass $i: --pmc $set" >> "$O/pmc.txt"
  summ "$O/pmc$i" k_add >> "$O/pmc.txt"
  rm -rf "$O/pmc$i"
done

# FETCH_SIZE / WRITE_SIZE against known byte counts in the kernel's two access patterns
: > "$O/calib.txt"
for set in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/cal_$set" -o c -- "$R/$T/fetch_calib" > "$O/cal_$set.log" 2>&1
  echo "# --pmc $set" >> "$O/calib.txt"
  grep -E '^(CALIB|TIME)' "$O/cal_$set.log" >> "$O/calib.txt"
  python - "$O/cal_$set" >> "$O/calib.txt" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
    for r in rows:
        print("PMCROW %s %s %s" % (r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"]))
PY
  rm -rf "$O/cal_$set"
done

# the `mul` kernel: VALU instructions per scalar (bench.py --cmd mul prices its roofline with it)
: > "$O/pmc_mul.txt"
for set in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "VALUBusy"; do
  ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/pmcm" -o p -- python "$R/bench.py" --cmd mul --steps 1 --warmup 1 > "$O/pmcm.log" 2>&1
  echo "# --pmc $set   (bench.py --cmd mul --steps 1 --warmup 1: 3 x 2^24 scalars, -a cu, 22-bit window table)" >> "$O/pmc_mul.txt"
  summ "$O/pmcm" k_mul_check >> "$O/pmc_mul.txt"
  rm -rf "$O/pmcm"
done

cd "$R"
python tools/make_roofline_profile.py "$O" "$TAG" > "$O/${TAG}_roofline.json" 2> "$O/make_profile.err"
python - "$O/pmc_mul.txt" "$TAG" > "$O/${TAG}_mul.json" <<'PY'
import json, sys
pmc, ns, passes = {}, [], 0
for line in open(sys.argv[1]):
    f = line.split()
    if line.startswith("# --pmc"): passes += 1
    if f and f[0] == "PMC": pmc[f[2]] = (float(f[3]), int(f[4]))
    if f and f[0] == "TRACE": ns.append(int(f[2]))
scalars = 3 * (1 << 24)  # the call that builds the table + warm-up + one step
out = {"tag": sys.argv[2], "kernel": "k_mul_check<addr33,addr65>", "workload": "bench.py --cmd mul: 2^24 scalars per step, -a cu, pieces of 2^20 scalars, 8 per thread, window table W = 22 (12 additions per scalar)",
       "pmc": {k: v[0] for k, v in pmc.items()}, "dispatches": {k: v[1] for k, v in pmc.items()}, "derived": {}}
if "SQ_INSTS_VALU" in pmc:
    out["derived"]["valu_lane_ops_per_scalar"] = pmc["SQ_INSTS_VALU"][0] * 64 / scalars
if "FETCH_SIZE" in pmc:
    out["derived"]["fetch_bytes_per_scalar_reported"] = pmc["FETCH_SIZE"][0] * 1024 / scalars
if "VALUBusy" in pmc:
    out["derived"]["valu_busy_pct"] = pmc["VALUBusy"][0] / max(pmc["VALUBusy"][1], 1)
if ns and passes:  # a call is cut into chunks of unequal size (the first is a quarter): total kernel time over total scalars
    out["derived"]["kernel_ms_per_2^22_scalars"] = sum(ns) / (passes * scalars) * (1 << 22) / 1e6
    out["derived"]["kernel_mscalars_s"] = passes * scalars / sum(ns) * 1e3
print(json.dumps(out, indent=1))
PY
cp "$O/${TAG}_mul.json" "profiles/${TAG}_mul.json"
# the final bench line, priced with the profile just taken
cp "$O/${TAG}_roofline.json" "profiles/${TAG}_roofline.json"
python bench.py > "$O/bench.json" 2> "$O/bench.err"
cat "$O/bench.json"; head -8 "$O/stats.txt"; cat "$O/pmc.txt" "$O/calib.txt"; cat "$O/make_profile.err"
