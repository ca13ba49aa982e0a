#!/usr/bin/env python3
"""Where the bring-up time of one context goes (GPU box): open (+ self-test), filter upload pageable / pinned,
reserve (window table, walk table, chains) at several geometries, first and second call of a 2^32-key scan.
usage: python tools/bringup_timing.py [filter MB, default 54]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecloop_amd import Device, capi  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 54
words = np.random.default_rng(1).integers(0, 1 << 62, mb * 131072, dtype=np.int64).astype(np.uint64)


def t(label, f):
    t0 = time.perf_counter()
    r = f()
    print("%-58s %8.1f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
    return r


for geo in [(0, 0), (256, 0), (256, 1 << 20), (1024, 1 << 20)]:
    print("== geometry half_group=%d lanes=%d (0 = default 1024 / 2^21), filter %d MB" % (geo[0], geo[1], mb))
    d = t("open (first handle of the process runs the self-test)", lambda: Device(0))
    if geo != (0, 0):
        d.set_geometry(*geo)
    t("set_bloom, pageable host memory", lambda: d.set_bloom(words))
    import ctypes as C
    ptr = capi.load().ecl_hip_alloc_host(words.nbytes)
    pl = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=words.shape)
    pl[:] = words
    t("set_bloom, page-locked host memory (ecl_hip_alloc_host)", lambda: d.set_bloom(pl))
    del pl
    capi.load().ecl_hip_free_host(ptr)
    t("reserve(2^32 keys): window table, walk table, buffers", lambda: d.reserve(1 << 32))
    t("add_range 2^32 keys, first call", lambda: d.add_range(0x100000000, 1 << 32, cap=1 << 16))
    ms, launches, keys = d.timing()
    print("%-58s %8.1f ms" % ("   of which search kernel (HIP events)", ms))
    d.reset_timing()
    t("add_range 2^32 keys, second call (same range)", lambda: d.add_range(0x100000000, 1 << 32, cap=1 << 16))
    ms, launches, keys = d.timing()
    print("%-58s %8.1f ms" % ("   of which search kernel (HIP events)", ms))
    t("close", d.close)
