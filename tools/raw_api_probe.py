#!/usr/bin/env python3
"""ecl_hip_mul_batch_raw against ecl_hip_mul_batch at the same call size (2^24), page-locked inputs, one context and two contexts on
two threads: where `mul -raw` stands against hex scalars at the device, without the host program's text side.
usage: python tools/raw_api_probe.py [log2_lines=24] [calls=6] [kinds=hex,raw] [contexts=1,2]"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ecloop_amd import capi  # noqa: E402
from ecloop_amd import Device  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 24
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << L
lib = capi.load()
rng = np.random.default_rng(5)
ln = rng.integers(8, 25, n).astype(np.uint64)
tot = int(ln.sum()) + n
words = (rng.integers(0, 1 << 32, 1 << 21, dtype=np.uint32) & rng.integers(0, 1 << 32, 1 << 21, dtype=np.uint32) & rng.integers(0, 1 << 32, 1 << 21, dtype=np.uint32))


def pinned(nbytes):
    p = lib.ecl_hip_alloc_host(nbytes)
    assert p
    return p, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


def make_inputs():
    pt, text = pinned(tot + 16)
    text[:tot] = rng.integers(97, 123, tot, dtype=np.uint8)
    ends = np.cumsum(ln + 1) - 1
    text[ends] = 10
    starts = np.concatenate([[0], ends[:-1] + 1]).astype(np.uint64)
    pl, lines8 = pinned(n * 8)
    lines = lines8.view(np.uint64)
    lines[:] = starts | (ln << np.uint64(32))
    pk, ks8 = pinned(n * 32)
    ks8[:] = rng.integers(0, 256, n * 32, dtype=np.uint8)
    return pt, pl, pk


def run(kind, nctx):
    devs = [Device(0, a33=True, a65=True) for _ in range(nctx)]
    ins = [make_inputs() for _ in range(nctx)]
    for d in devs:
        d.set_bloom(words)
        d.set_mul_window(24)

    def call(i):
        d, (pt, pl, pk) = devs[i], ins[i]
        cnt = C.c_uint32(0)
        buf = (C.c_uint8 * (64 * 65536))()
        if kind == "raw":
            rc = lib.ecl_hip_mul_batch_raw(d.h, pt, tot, pl, n, buf, 65536, C.byref(cnt))
        else:
            rc = lib.ecl_hip_mul_batch(d.h, pk, n, buf, 65536, C.byref(cnt))
        assert rc == 0, rc
        return rc

    for i in range(nctx):
        call(i)  # tables, buffers
    t0 = time.time()
    ts = [threading.Thread(target=lambda i=i: [call(i) for _ in range(CALLS)]) for i in range(nctx)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.time() - t0
    rate = nctx * CALLS * n / dt / 1e6
    print("%-4s %d context(s): %d calls of 2^%d -> %.1f M/s (%.2f ms per call)" % (kind, nctx, nctx * CALLS, L, rate, dt / CALLS * 1e3), flush=True)
    [d.close() for d in devs]


KINDS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["hex", "raw"]
NCTX = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2]
for kind in KINDS:
    for nctx in NCTX:
        run(kind, nctx)
