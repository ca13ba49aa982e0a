#!/usr/bin/env python3
"""`mul` path throughput (BASELINE.json configs[4]): seeded 256-bit scalars through ecl_hip_mul_batch from a page-locked
array (host -> device copy of the scalars included), addr33 + addr65, list filter of the brainwallet hashes.  Prints the
wall rate of each call and the device-side rate (HIP events over the copies + kernels of the call); the first call
includes building the window table.  usage: bench_mul.py [log2 n] [calls] [window bits, 0 = automatic] [list|design|empty] [random|small|seq]
(filter: the brainwallet list's 128-bits-per-entry filter, a 56 MB synthetic .blf at the design density 0.375, or no bit set;
scalars: random 255-bit ones, random ones below 2^66 (a puzzle range: every high window is empty), or consecutive ones from 2^65)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecloop_amd import capi  # noqa: E402
from ecloop_amd.engine import load_filter  # noqa: E402

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
window = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # 0 = the library's automatic choice
kind = sys.argv[4] if len(sys.argv) > 4 else "list"
scal = sys.argv[5] if len(sys.argv) > 5 else "random"
if kind == "design":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import synth_bloom_words
    words = synth_bloom_words(7000003, 23, "a&(b|c)")
elif kind == "empty":
    words = np.zeros(64, dtype=np.uint64)
else:
    words = load_filter(os.path.join(ROOT, "tests", "golden", "btc-bw-hash")).words
d = capi.Device(0, a33=True, a65=True)
d.set_bloom(words)
d.set_mul_window(window)
rng = np.random.RandomState(1)
ptr = d.lib.ecl_hip_alloc_host(n * 32)
assert ptr
K = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n, 4))
K[:] = rng.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
if scal == "small":
    K[:, 1] &= np.uint64(3)
    K[:, 2:] = 0
elif scal == "seq":
    K[:, 0] = np.arange(n, dtype=np.uint64)
    K[:, 1] = 2
    K[:, 2:] = 0
out = np.zeros(4096, dtype=capi.FOUND_DTYPE)
cnt = C.c_uint32()
prev = 0.0
for it in range(calls):
    t0 = time.perf_counter()
    rc = d.lib.ecl_hip_mul_batch(d.h, ptr, n, out.ctypes.data, 4096, C.byref(cnt))
    dt = time.perf_counter() - t0
    assert rc == 0
    ms = d.mul_timing()[0]
    print(f"mul_batch: {n} scalars, a33+a65: wall {dt*1e3:.1f} ms -> {n/dt/1e6:.1f} M/s; device {ms-prev:.2f} ms -> {n/(ms-prev)/1e3:.1f} M/s (hits {cnt.value}, window {d.mul_window()} bits, {scal} scalars)")
    prev = ms
d.lib.ecl_hip_free_host(ptr)
