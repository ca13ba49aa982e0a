#!/bin/bash
# Collects the measured evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh [round-tag, default r02]
# -> gpurun_out/prof_<tag>/{bench.json, stats.txt, pmc.txt, calib.txt, ubench8.txt, ubench4.txt, <tag>_roofline.json}
# Counter passes are separate runs with --kernel-trace only (never --pmc together with sys/hip/hsa tracing), one
# 2^32-key launch each (ECL_HIP_SKIP_SELFTEST=1: no 4096-key self-test launch in the counters).
# Copy what is to be kept into profiles/ (tracked); bench.py reads profiles/<tag>_roofline.json.
#   PARTS="ubench headline calib mul cu_endo bench binding mulcli parity" (default: all) selects what is collected; a part that is left out keeps
#   whatever profiles/ already holds for it
set -u
TAG=${1:-r06}
PARTS=${PARTS:-ubench headline calib mul cu_endo bench binding mulcli parity}
want() { [[ " $PARTS " == *" $1 "* ]]; }
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
T=ecloop_amd/csrc/tools
[ -x $T/ubench ] || $HIPCC --offload-arch=gfx950 -O3 $T/ubench.hip -o $T/ubench
[ -x $T/fetch_calib ] || $HIPCC --offload-arch=gfx950 -O3 $T/fetch_calib.hip -o $T/fetch_calib
if want ubench; then
$T/ubench 8 > "$O/ubench8.txt" 2>&1
$T/ubench 4 > "$O/ubench4.txt" 2>&1
fi

cd /tmp
# per-kernel time of the default bench command
# (ECL_HIP_SKIP_SELFTEST=1: without the 4096-key self-test launch every k_add launch in the trace is a 2^32-key one, so the
#  kernel's average duration in the summary is directly comparable with bench.py's roofline.ms_per_launch)
if want headline; then
ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --stats -d "$O/stats" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-secondary > "$O/stats.log" 2>&1
db=$(find "$O/stats" -name '*.db' | head -1)
[ -n "$db" ] && python "$R/tools/rocprof_summary.py" "$db" > "$O/stats.txt"
rm -rf "$O/stats"
fi

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
if os.environ.get("SUMM_PER_DISPATCH"):  # one row per dispatch as well (time-weighted ratios over the full-size pieces of `mul`)
    per = collections.defaultdict(dict); ns = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                i = int(r["Dispatch_Id"]); per[i][r["Counter_Name"]] = per[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]: ns[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for i in sorted(per):
        for c, v in sorted(per[i].items()): print("DISPATCH %d %d %s %.0f" % (i, ns.get(i, 0), c, v))
PY
}

if want headline; then
: > "$O/pmc.txt"
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "VALUUtilization"; do
  i=$((i+1))
  ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/pmc$i" -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu --no-secondary > "$O/pmc$i.log" 2>&1
  echo "# pass $i: --pmc $set" >> "$O/pmc.txt"
  summ "$O/pmc$i" k_add >> "$O/pmc.txt"
  rm -rf "$O/pmc$i"
done
fi

# FETCH_SIZE / WRITE_SIZE against known byte counts in the kernel's two access patterns
if want calib; then
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
fi

# the `mul` kernels (configs[4]): VALU instructions and HBM bytes per scalar (bench.py prices `secondary.cfg4.api.roofline` and
# `--cmd mul` with it).  bench.py --cmd mul --steps 1 --warmup 1 = 3 calls of 2^24 scalars (table build call, warm-up, one step).
if want mul; then
: > "$O/pmc_mul.txt"
for set in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "VALUBusy"; do
  ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/pmcm" -o p -- python "$R/bench.py" --cmd mul --steps 1 --warmup 1 > "$O/pmcm.log" 2>&1
  echo "# --pmc $set   (bench.py --cmd mul --steps 1 --warmup 1: 3 x 2^24 scalars, -a cu, 26-bit window table)" >> "$O/pmc_mul.txt"
  SUMM_PER_DISPATCH=1 summ "$O/pmcm" k_mul >> "$O/pmc_mul.txt"
  rm -rf "$O/pmcm"
done
# where the kernel's issue slots go: wait / TCP / translation counters per dispatch, full-size pieces only, W = 22 and 26
( cd "$R" && python tools/mul_stall_profile.py --tag "$TAG" > "$O/mul_stall.log" 2>&1 && cp "gpurun_out/${TAG}_mul_stall.txt" "profiles/${TAG}_mul_stall.txt" )
fi

# the -a cu -endo kernel (configs[2]'s per-GPU shape): one 2^30-key launch against the 5.9 GB filter
if want cu_endo; then
: > "$O/pmc_cu_endo.txt"
for set in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "TCP_PENDING_STALL_CYCLES TCP_GATE_EN1"; do
  ECL_HIP_SKIP_SELFTEST=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/pmce" -o p -- python "$R/bench.py" --addr cu --endo --filter-n 1100000000 --keys-log2 30 --steps 1 --warmup 0 --no-cpu > "$O/pmce.log" 2>&1
  echo "# --pmc $set   (bench.py --addr cu --endo --filter-n 1100000000 --keys-log2 30 --steps 1 --warmup 0: one 2^30-key launch)" >> "$O/pmc_cu_endo.txt"
  summ "$O/pmce" k_add >> "$O/pmc_cu_endo.txt"
  rm -rf "$O/pmce"
done
fi

cd "$R"
if want headline; then
  [ -f "$O/calib.txt" ] || cp "profiles/$(ls profiles | grep fetch_calibration | tail -1)" "$O/calib.txt" 2>/dev/null
  python tools/make_roofline_profile.py "$O" "$TAG" > "$O/${TAG}_roofline.json" 2> "$O/make_profile.err"
  cp "$O/${TAG}_roofline.json" "profiles/${TAG}_roofline.json"
  cp "$O/stats.txt" "profiles/${TAG}_kernel_stats.txt"; cp "$O/pmc.txt" "profiles/${TAG}_pmc.txt"
fi
python tools/make_secondary_profiles.py "$O" "$TAG" 2>> "$O/make_profile.err"
for k in mul cu_endo; do
  [ -f "$O/${TAG}_roofline_$k.json" ] && cp "$O/${TAG}_roofline_$k.json" profiles/ && cp "$O/pmc_$k.txt" "profiles/${TAG}_pmc_$k.txt"
done
want calib && cp "$O/calib.txt" "profiles/${TAG}_fetch_calibration.txt"
want ubench && cat "$O/ubench8.txt" "$O/ubench4.txt" > "profiles/ubench_${TAG}.txt"
# the final bench line, priced with the profiles just taken
if want bench; then
  python bench.py > "$O/bench.json" 2> "$O/bench.err"
  cp "$O/bench.json" "profiles/${TAG}_bench.json"
  cat "$O/bench.json"
fi
# the reference's own host program bound to the library (unchanged MAX_JOB_SIZE: the look-ahead), and `mul` through the C host program
if want binding; then
  python tools/bench_ref_binding.py --tag "$TAG" > "$O/ref_binding.log" 2>&1; echo "bench_ref_binding exit code $?"
  cp "gpurun_out/${TAG}_ref_binding.txt" "profiles/${TAG}_ref_binding.txt" && cat "profiles/${TAG}_ref_binding.txt"
fi
if want mulcli; then
  bash tools/bench_mul_cli.sh 30 3 > "$O/mul_cli.txt" 2>&1
  { echo "# tools/bench_mul_raw.sh 1073741824: mul -raw over 2^30 pass phrases of 8..24 characters (tools/gen_phrases.c), one untimed pass first"
    bash tools/bench_mul_raw.sh 1073741824; } >> "$O/mul_cli.txt" 2>&1
  cp "$O/mul_cli.txt" "profiles/${TAG}_mul_cli.txt"; grep -E "^#|run [0-9]|parse only" "profiles/${TAG}_mul_cli.txt" | cut -c1-200
fi
# bit-exact found lists against the reference binary on the final build: the whole 2^32-key range of the headline config (54 MB filter)
# and -a cu -endo over 2^28 keys against the 5.9 GB filter (two runs of tools/full_range_parity.py), both binaries reading the same .blf
if want parity; then
  { echo "# tools/collect_profiles.sh $TAG, last step: found lists of the HIP path and of the unmodified reference binary (oracle/_ref/ecloop_sane, built from"
    echo "# /root/reference/main.c by oracle/Makefile) on identical private-key ranges, same .blf file.  source_sha256 $(python -c 'from ecloop_amd.build import source_sha256; print(source_sha256())')"
    python tools/full_range_parity.py --endo-log2 24 2> "$O/parity1.err"; echo "exit code $?"
    echo
    python tools/full_range_parity.py --filter-n 1100000000 --main-log2 28 --endo-log2 28 2> "$O/parity2.err"; echo "exit code $?"
  } > "$O/full_range_parity.txt"
  cp "$O/full_range_parity.txt" "profiles/${TAG}_full_range_parity.txt"
  grep -E "^(==|HIP|reference|bound ref|identical|exit)" "$O/full_range_parity.txt"
fi
[ -f "$O/stats.txt" ] && head -8 "$O/stats.txt"
cat "$O"/pmc*.txt 2>/dev/null | grep -v TRACE | head -80; cat "$O/make_profile.err"
