#!/bin/bash
# configs[2]'s shape on one GPU: bench.py with -a cu -endo, the 54 MB and the 5.9 GB filter, and addr33 with the 5.9 GB filter
cd "$(dirname "$0")/.."
for args in "--addr cu --endo" "--addr cu --endo --filter-n 1100000000" "--filter-n 1100000000"; do
  python3 bench.py --no-cpu --no-secondary --steps 3 $args 2>/dev/null | python3 -c "
import json, sys
r = json.loads(sys.stdin.readlines()[-1])
print('bench.py $args | %.2f %s | kernel %.3f ms per launch of %d keys' % (r['value'], r['unit'], r['roofline']['ms_per_launch'], r['roofline']['keys_per_launch']))"
done
