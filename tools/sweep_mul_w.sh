#!/bin/bash
# `mul` throughput against the window width of its table (ecl_hip_set_mul_window): ceil(256/W) rows x 2^(W-1) points (signed digits since round 4);
# W = 14 is the reference's CPU-cache-sized table.  Run on the GPU box.   tools/sweep_mul_w.sh [log2 scalars per call]
cd "$(dirname "$0")/.."
for w in 14 16 18 20 22 24 0; do
  echo "== window $w bits"
  python3 tools/bench_mul.py ${1:-24} 4 $w 2>&1 | tail -4
done
