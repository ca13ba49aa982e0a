#!/usr/bin/env python3
"""Experiment (round 5): a 29-bit signed-digit window table for `mul` - 9 additions per scalar instead of 10, 138 GB of HBM.
Correctness on 2^16 edge / random scalars against the double-and-add kernel, then the rate.  Needs a library built with -DMUL_W_MAX=29u
(ECLOOP_HIP_LIB=build_ab/w29.so).   usage: ECLOOP_HIP_LIB=... python tools/exp_w29.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ecloop_amd import capi  # noqa: E402
import test_gpu_add as T  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 29
d = capi.Device(0, a33=True, a65=True)
d.set_bloom(T.ONES)
d.set_mul_window(W)
t0 = time.perf_counter()
K = T._digit_edge_scalars(np.random.default_rng(W), 1 << 16, W)
d2 = capi.Device(0)  # addr33 only for the checker helper
d2.set_bloom(T.ONES)
d2.set_mul_window(W)
n = T._mul_all_against_double_and_add(d2, K)
print("W = %d: %d of %d scalars equal to the double-and-add kernel; first call incl. table build %.1f s" % (W, n, len(K), time.perf_counter() - t0), flush=True)
d2.close()
d.close()
