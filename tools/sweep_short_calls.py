#!/usr/bin/env python3
"""Short ecl_hip_add_range calls (the reference's MAX_JOB_SIZE 2^21 keys, main.c:16, up to 2^26) by half group: contiguous calls that
continue the resident walk, wall-clock rate over the calls and HIP-event time of the kernel alone.
usage: python tools/sweep_short_calls.py [log2 sizes, default 21,22,23,24,26] [half groups, default 0,128,64,32,16,8]  (0 = automatic)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecloop_amd import capi  # noqa: E402

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "21,22,23,24,26").split(",")]
halves = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,128,64,32,16,8").split(",")]
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import synth_bloom_words  # noqa: E402
words = synth_bloom_words(7000003, 23, "a&(b|c)")  # 56 MB at the design density
print("# tools/sweep_short_calls.py: contiguous add_range calls, addr33, 56 MB filter at the design density; rate by the wall clock over the calls")
for L in sizes:
    n = 1 << L
    calls = max(8, min(400, (1 << 30) >> L))
    for hg in halves:
        d = capi.Device(0)
        d.set_lookahead(0)  # this tool measures the launches themselves (the geometry model), not the look-ahead over them
        d.set_bloom(words)
        if hg:
            d.set_geometry(hg, 0)
        start = 0x100000000
        for _ in range(3):  # warm-up: table, buffers, first positioning
            d.add_range(start, n, cap=4096)
            start += n
        d.reset_timing()
        t0 = time.perf_counter()
        for _ in range(calls):
            d.add_range(start, n, cap=4096)
            start += n
        dt = time.perf_counter() - t0
        kms, launches, keys = d.timing()
        sms, setups = d.setup_timing()
        print("2^%d keys x %3d calls, half_group %4s: %8.1f Mkeys/s by the wall clock, kernel alone %8.1f Mkeys/s (%.3f ms per call), %d re-positionings"
              % (L, calls, hg or "auto", n * calls / dt / 1e6, keys / kms / 1e3, kms / launches, setups), flush=True)
        d.close()
