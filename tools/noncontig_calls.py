#!/usr/bin/env python3
"""ecl_hip_add_range calls that do NOT continue the resident walk (every call starts 2^40 keys further on): what a job of the reference costs
when its jobs are not consecutive on a GPU (several worker threads / GPUs pulling from one counter, main.c:418-431), a `rnd` window, a step of
the strong-scaling bench.  Prints the wall rate, the kernel time and the set-up time on the device per call.
usage: python tools/noncontig_calls.py   -> profiles/r05_noncontig_calls.txt"""
import os, sys, time
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from ecloop_amd import capi
from synth import synth_bloom_words
words = synth_bloom_words(7000003, 23, "a&(b|c)")
for L in (21, 24):
    n = 1 << L
    d = capi.Device(0); d.set_lookahead(0); d.set_bloom(words)  # (the launches themselves, not the look-ahead)
    start = 0x100000000
    for _ in range(3):
        d.add_range(start, n, cap=4096); start += n + (1 << 40)
    d.reset_timing()
    calls = 300
    t0 = time.perf_counter()
    for _ in range(calls):
        d.add_range(start, n, cap=4096); start += n + (1 << 40)
    dt = time.perf_counter() - t0
    kms, launches, keys = d.timing(); sms, setups = d.setup_timing()
    print("2^%d keys, %d NON-contiguous calls: %.1f Mkeys/s wall (%.3f ms per call), kernel %.3f ms, set-up on device %.3f ms per call (%d set-ups), geometry %s" % (L, calls, n*calls/dt/1e6, dt/calls*1e3, kms/launches, sms/max(setups,1), setups, d.plan_geometry(n)))
    d.close()
