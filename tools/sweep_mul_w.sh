#!/bin/bash
# A/B of `mul`'s window width: build_ab/libecl_mulw<W>.so = the library built with -DMUL_W=<W>u (table of
# ceil(256/W) rows x (2^W - 1) points; W = 14 is the reference's CPU-cache-sized table).  Run on the GPU box.
cd "$(dirname "$0")/.."
for w in 14 16 18 20 22 24; do
  def=22
  lib=build_ab/libecl_mulw$w.so
  [ "$w" = "$def" ] && lib=ecloop_amd/libecloop_hip.so
  [ -f "$lib" ] || continue
  echo "== MUL_W=$w ($lib)"
  ECLOOP_HIP_LIB=$PWD/$lib python3 tools/bench_mul.py ${1:-24} 4 2>&1 | tail -4
done
