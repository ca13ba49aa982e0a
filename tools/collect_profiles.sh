#!/bin/bash
# Collects the evidence under profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/{bench.json, stats.txt, pmc_*.csv}
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa tracing together with --pmc).
set -u
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
python bench.py > "$O/bench.json" 2> "$O/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$O/stats" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu > "$O/stats.log" 2>&1
db=$(find "$O/stats" -name '*.db' | head -1)
[ -n "$db" ] && python "$R/tools/rocprof_summary.py" "$db" > "$O/stats.txt"
i=0
for set in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE" "VALUBusy" "VALUUtilization"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/pmc$i" -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu > "$O/pmc$i.log" 2>&1
  f=$(find "$O/pmc$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" > "$O/pmc$i.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "k_add" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print("%-24s %d   (dispatches %d)" % (k, acc[k], n[k])) if acc[k] > 1000 else print("%-24s %.3f   (sum over %d dispatches)" % (k, acc[k], n[k]))
PY
  rm -rf "$O/pmc$i"
done
rm -rf "$O/stats"
cd "$R"; cat "$O"/bench.json; cat "$O"/stats.txt | head -8; cat "$O"/pmc*.txt
