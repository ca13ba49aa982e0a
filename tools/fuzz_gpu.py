#!/usr/bin/env python3
"""Differential fuzz of the add path on the GPU box: random start scalars (any size up to 2^255), key counts, launch
geometries (half group B, lanes T), address / endo selections, bloom densities and nwords (odd sizes), split into 1-3
contiguous calls (continuation of the walk state) - every found list must equal the oracle's on the same keys.
usage: python tools/fuzz_gpu.py [seconds=150] [seed=1]      -> gpurun_out/fuzz.txt"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from ecloop_amd import Device  # noqa: E402
from synth import synth_bloom_words  # noqa: E402

LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72


def privkey(start, off, endo, offs=0):
    k = (start + (off << offs)) % orc.N
    if endo in (2, 3):
        k = k * LAM % orc.N
    if endo in (4, 5):
        k = k * LAM % orc.N * LAM % orc.N
    if endo in (1, 3, 5):
        k = (-k) % orc.N
    return k


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t_end, trials, keys, hits, log = time.time() + budget, 0, 0, 0, []
    while time.time() < t_end:
        a33, a65 = rnd.choice([(True, False), (False, True), (True, True)])
        endo = rnd.random() < 0.4
        bits = rnd.choice([24, 40, 64, 128, 200, 255])
        start = rnd.getrandbits(bits) | (1 << (bits - 1)) | 0x1000
        nkeys = rnd.choice([1, 7, 2047, 2048, 2049, rnd.randrange(1, 5000), rnd.randrange(5000, 120000)])
        B = rnd.choice([2, 3, 8, 16, 33, 64, 100, 255, 256, 1024, 2048])
        T = rnd.choice([256, 512, 768, 1024, 4096])
        nw = rnd.choice([1, 2, 63, 64, 1000, 4097, 65536 + 3])
        mode = rnd.choice(["a", "a|(b&c)", "a|b"])
        words = synth_bloom_words(nw, rnd.randrange(1 << 30), mode)
        if rnd.random() < 0.15:
            words = np.full(nw, 0xFFFFFFFFFFFFFFFF, np.uint64)
            nkeys = min(nkeys, 3000)
        offs = 0
        if rnd.random() < 0.2:  # strided scan (-d offs:size): whole 2048-key groups, which is what the oracle can dump
            offs = rnd.choice([1, 5, 29, 64, 128, 200])
            nkeys = 2048 * rnd.choice([1, 2, 3])
            start = (rnd.getrandbits(255 - 12) | 0x1000) % (orc.N >> 12)
        cuts = sorted(rnd.sample(range(1, nkeys), min(rnd.choice([0, 1, 2]), max(nkeys - 1, 0)))) if nkeys > 2 else []
        if offs:
            cuts = []
        if cuts and rnd.random() < 0.5:  # cuts on sweep boundaries continue the resident walk, others re-initialise
            cuts = sorted({min(nkeys - 1, max(1, c // (2 * B * T) * (2 * B * T) or 2 * B)) for c in cuts})
        auto_geo = rnd.random() < 0.15  # library default: 2^21 lanes, half group chosen per call
        desc = dict(auto_geo=auto_geo, offs=offs, a33=a33, a65=a65, endo=endo, start=hex(start), nkeys=nkeys, B=B, T=T, nwords=nw, mode=mode, cuts=cuts)
        d = Device(0, a33=a33, a65=a65, endo=endo, ord_offs=offs)
        got = []
        try:
            if not auto_geo:
                d.set_geometry(B, T)
            d.set_bloom(words)
            at = 0
            for c in cuts + [nkeys]:
                cap = 4096
                while True:
                    recs, n = d.add_range((start + (at << offs)) % orc.N, c - at, cap=cap)
                    if n <= cap:
                        break
                    cap = n
                got += ["%s\t%s\t%064x" % ("addr33" if r["compressed"] else "addr65", orc.hex160(r["h160"]),
                                            privkey(start + (at << offs), int(r["key_offset"]), int(r["endo"]), offs)) for r in recs]
                at = c
        finally:
            d.close()
        flt = orc.OrcFilter(bloom_words=words)
        want = []
        stride = 1 << offs
        if offs == 0:
            batches = [orc.add_range(flt, start, start + nkeys, a33=a33, a65=a65, endo=endo, verify=False, threads=16, cap=1 << 22)]
        else:  # a one-key-wide range makes the oracle hash exactly one 2048-key group at this stride (main.c:442)
            batches = [orc.add_range(flt, (start + g * 2048 * stride) % orc.N, (start + g * 2048 * stride) % orc.N + 1, a33=a33, a65=a65,
                                     endo=endo, offs=offs, verify=False, cap=1 << 18) for g in range(nkeys // 2048)]
        for rc, out, n, _, _ in batches:
            assert rc == 0, rc
            for i in range(n):
                r = out[i]
                k = orc.val(r.pk)
                base = k
                if r.endo in (1, 3, 5):
                    base = (-base) % orc.N
                if r.endo in (2, 3):
                    base = base * pow(LAM, -1, orc.N) % orc.N
                if r.endo in (4, 5):
                    base = base * pow(LAM, -2, orc.N) % orc.N
                if ((base - start) % orc.N) >> offs < nkeys:
                    want.append("%s\t%s\t%064x" % ("addr33" if r.compressed else "addr65", orc.hex160(r.h160), k))
        if sorted(got) != sorted(want):
            msg = "MISMATCH %r: gpu %d lines, oracle %d lines" % (desc, len(got), len(want))
            print(msg)
            log.append(msg)
            break
        trials, keys, hits = trials + 1, keys + nkeys, hits + len(got)
    rep = ["# tools/fuzz_gpu.py %s %s: %d trials, %d keys, %d compared hits, %s" % (budget, sys.argv[2] if len(sys.argv) > 2 else 1, trials, keys, hits,
                                                                                   "ALL EQUAL to the oracle" if not log else "FAILED")] + log
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "fuzz.txt"), "w").write("\n".join(rep) + "\n")
    print("\n".join(rep))
    sys.exit(1 if log else 0)


if __name__ == "__main__":
    main()
