#!/bin/bash
# round 5, review lever 1(c): the bound on a ramp-free / zero-copy first piece of ecl_hip_mul_batch - the shipped library against a measurement build
# whose first piece is already on the device (tools/build_first_resident_variant.py), alternating.   -> gpurun_out/s14_first_resident.txt
export TMPDIR=/tmp
O=gpurun_out
{
for rep in 1 2 3 4; do
  for lib in shipped build_ab/first_resident.so; do
    path=$PWD/$lib; [ "$lib" = shipped ] && path=$PWD/ecloop_amd/libecloop_hip.so
    for L in 22 24 26; do
      echo "== $lib  2^$L scalars"
      ECLOOP_HIP_LIB=$path python tools/bench_mul.py $L 6 26 design | tail -4
    done
  done
done
} > $O/s14_first_resident.txt 2>&1
python - <<'PY'
import re,collections
cur=None; acc=collections.defaultdict(list)
for l in open('gpurun_out/s14_first_resident.txt'):
    if l.startswith("=="): cur=l.strip()
    m=re.search(r"device [\d.]+ ms -> ([\d.]+) M/s", l)
    if m and cur: acc[cur].append(float(m.group(1)))
for k,v in acc.items(): print("%-50s n=%d mean %.1f min %.1f max %.1f M scalars/s" % (k, len(v), sum(v)/len(v), min(v), max(v)))
PY
