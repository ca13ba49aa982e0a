#!/usr/bin/env python3
"""Kernel rate of `add` addr33 as a function of the bloom fill (cost of the probe loop): zeros, list-mode density, .blf density."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecloop_amd import capi
n = 1 << 20
rng = np.random.RandomState(3)
def words(density):
    if density == 0: return np.zeros(n, dtype=np.uint64)
    bits = rng.random_sample((n, 64)) < density
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(-1)
for dens in (0.0, 0.145, 0.375, 0.6):
    d = capi.Device(0)
    d.set_bloom(words(dens))
    b, lanes = d.geometry()
    nk = lanes * 2 * b * 2
    d.add_range(0x100000000, nk, cap=1 << 20)
    d.reset_timing()
    recs, tot = d.add_range(0x100000000 + nk, nk, cap=1 << 20)
    ms, launches, keys = d.timing()
    print(f"fill {dens:.3f}: {keys/ms/1e3:.1f} Mkeys/s kernel, hits {tot}")
    d.close()
