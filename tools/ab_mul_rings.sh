cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in ecloop_amd/libecloop_hip.so build_ab/r04_norings.so; do
  for kind in design empty list; do
    echo -n "$(basename $lib) $kind: "; ECLOOP_HIP_LIB=$PWD/$lib python tools/bench_mul.py 24 6 26 $kind 2>&1 | tail -1
  done
done
done
