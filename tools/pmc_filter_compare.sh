#!/bin/bash
# Why does a multi-GB bloom filter cost the add kernel 6-10 %?  The same 2^32-key launch under the same counters with the
# 54 MB filter (Infinity-Cache resident) and the 5.9 GB one (HBM): wave stall cycles, VMEM issue back-pressure, address
# translation (UTCL1 / UTCL2), L2 hit rate, read latency.   bash tools/pmc_filter_compare.sh  -> gpurun_out/pmc_filter/
# Counter passes are separate runs with --kernel-trace only (never --pmc together with sys/hip/hsa tracing).
set -u
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/pmc_filter
rm -rf "$O"; mkdir -p "$O"
cd /tmp
summ() {
python - "$1" <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_add" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in sorted(acc): print("PMC %s %.0f %d" % (c, acc[c], n[c]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_add" in r["Kernel_Name"]: print("TRACE_ns %d" % (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
PY
}
for n in 10000000 1100000000; do
  out="$O/filter_$n.txt"; : > "$out"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU" \
             "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum" \
             "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    ECL_HIP_SKIP_SELFTEST=1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$O/p$i" -o p -- \
        python "$R/bench.py" --steps 1 --warmup 0 --no-cpu --no-secondary --filter-n $n > "$O/p$i.log" 2>&1
    echo "# pass $i: --pmc $set" >> "$out"
    summ "$O/p$i" >> "$out"
    grep -m1 '"value"' "$O/p$i.log" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('BENCH_mkeys', r['value'])" >> "$out" 2>/dev/null
    rm -rf "$O/p$i"
  done
  echo "==== filter_n=$n"; cat "$out"
done
