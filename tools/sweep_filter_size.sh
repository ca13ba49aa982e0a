#!/bin/bash
# add kernel rate against the filter size (54 MB ... 5.9 GB): where the per-CU translation cache (UTCL1) stops covering
# the filter (profiles/r03_tlb_reach.txt).   bash tools/sweep_filter_size.sh  (on the GPU box)
for n in 10000000 100000000 370000000 500000000 740000000 1100000000; do
  python bench.py --steps 4 --warmup 2 --no-cpu --no-secondary --filter-n $n 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('filter', r['config']['workload'].split('bloom (')[1].split(')')[0], '|', r['value'], 'Mkeys/s |', r['roofline']['ms_per_launch'], 'ms per 2^32-key launch')"
done
