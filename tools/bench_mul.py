#!/usr/bin/env python3
"""`mul` path throughput (BASELINE.json configs[4]): 2^22 seeded 256-bit scalars per batch through ecl_hip_mul_batch
(host -> device copy of the scalars included), addr33 + addr65, list filter of the brainwallet hashes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecloop_amd import capi  # noqa: E402
from ecloop_amd.engine import load_filter  # noqa: E402

n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
flt = load_filter(os.path.join(ROOT, "tests", "golden", "btc-bw-hash"))
d = capi.Device(0, a33=True, a65=True)
d.set_bloom(flt.words)
rng = np.random.RandomState(1)
K = rng.randint(0, 1 << 63, size=(n, 4), dtype=np.int64).astype(np.uint64)
out = np.zeros(4096, dtype=capi.FOUND_DTYPE)
cnt = capi.C.c_uint32()
for it in range(3):
    t0 = time.perf_counter()
    rc = d.lib.ecl_hip_mul_batch(d.h, K.ctypes.data, n, out.ctypes.data, 4096, capi.C.byref(cnt))
    dt = time.perf_counter() - t0
    assert rc == 0
    print(f"mul_batch: {n} scalars, a33+a65: {dt*1e3:.1f} ms -> {n/dt/1e6:.1f} Mkeys/s (hits {cnt.value})")
