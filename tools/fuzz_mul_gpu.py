#!/usr/bin/env python3
"""Differential fuzz of the `mul` path on the GPU box: random batch sizes (1 .. 2^22+, so that 1, 2, 4, 8 and 16 scalars
per thread and several staged chunks all occur), random 256-bit scalars with zeros / n / small values mixed in,
address selections, filters of several sizes and densities, pageable and page-locked scalar arrays, the window width of the
table fixed at random (8 .. 26 bits, now and then 27 .. 29 where the HBM is free) or automatic; every third trial feeds text lines to ecl_hip_mul_batch_raw (`mul -raw`:
SHA-256 of the line on the device; lengths 0 .. 300, any alignment) with hashlib's digests as the scalars of the yardstick.
Every hit set of ecl_hip_mul_batch / _raw (window table + ONE inversion per thread) must equal the ORACLE's for the same scalars:
orc.mul_hash160_many (cmd_mul's jobs restated: ec_gtable_mul, one grprdc per 2048 scalars, addr33 / addr65; every scalar of every
trial, threaded over the host's cores) -> the oracle's blf_has.  No device kernel is the yardstick of another.
usage: python tools/fuzz_mul_gpu.py [seconds=120] [seed=1]      -> gpurun_out/fuzz_mul.txt"""
import ctypes as C
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from ecloop_amd import Device, capi  # noqa: E402
from synth import synth_bloom_words  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd, rng = random.Random(seed), np.random.default_rng(seed)
    L = orc.lib()
    L.orc_blf_has_many.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    N_LIMBS = np.array([(orc.N >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)], dtype=np.uint64)
    t_end, trials, scalars, hits = time.time() + budget, 0, 0, 0
    while time.time() < t_end:
        a33, a65 = rnd.choice([(True, False), (False, True), (True, True)])
        n = rnd.choice([1, 2, 63, 64, 65, 1000, rnd.randrange(1, 1 << 17), rnd.randrange(1 << 17, 1 << 19), rnd.randrange(1 << 19, 1 << 21),
                        (1 << 22) + rnd.randrange(1, 1 << 18)])
        K = rng.integers(0, 1 << 63, (n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, (n, 4), dtype=np.int64).astype(np.uint64)
        for _ in range(min(n, 6)):  # edge scalars at random places: 0, n (both infinity), 1, n-1, 2^14-1, a value above n
            i = rnd.randrange(n)
            K[i] = rnd.choice([np.zeros(4, np.uint64), N_LIMBS, np.array([1, 0, 0, 0], np.uint64), N_LIMBS - np.array([1, 0, 0, 0], np.uint64),
                               np.array([(1 << 14) - 1, 0, 0, 0], np.uint64), np.full(4, 0xFFFFFFFFFFFFFFFF, np.uint64),
                               np.full(4, 0x8000000000000000, np.uint64), np.full(4, 0x8000000080000000, np.uint64), np.full(4, 0x7FFFFFFFFFFFFFFF, np.uint64)])
        nw = rnd.choice([1, 64, 4099, 65539, (1 << 20) + 7])
        mode = rnd.choice(["a|b", "a|(b&c)", "a", "ones"])
        words = np.full(nw, 0xFFFFFFFFFFFFFFFF, np.uint64) if mode == "ones" else synth_bloom_words(nw, rnd.randrange(1 << 30), mode)
        if mode == "ones" and n > (1 << 20):
            words = synth_bloom_words(nw, 7, "a|b")  # keep the record count of the big batches moderate
        pinned = rnd.random() < 0.5
        host_ptr = None  # pinned: the scalars in page-locked memory from ecl_hip_alloc_host (read by DMA), else a pageable numpy array (staged)
        window = rnd.choice([0, 0, rnd.randrange(8, 25), rnd.randrange(8, 27), rnd.randrange(8, 27), rnd.randrange(8, 27)])
        if rnd.random() < 0.04:
            window = rnd.randrange(27, 30)  # 27: 36 GB, 28: 71 GB, 29: 138 GB of table, seconds to build
        raw = trials % 3 == 2
        if raw:  # the scalars ARE the SHA-256 digests of random lines
            import hashlib
            n = min(n, 1 << 18)
            lens = rng.integers(0, rnd.choice([20, 70, 300]), n)
            blob = rng.integers(0, 256, int(lens.sum()) + 8, dtype=np.uint8).tobytes()
            starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.uint64)
            table = starts | (lens.astype(np.uint64) << np.uint64(32))
            K = np.zeros((n, 4), dtype=np.uint64)
            for i in range(n):
                v = int.from_bytes(hashlib.sha256(blob[int(starts[i]): int(starts[i]) + int(lens[i])]).digest(), "big")
                K[i] = [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
            text = np.frombuffer(blob, dtype=np.uint8)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)  # a run that dies (GPU fault) leaves the trial it was in
        open(os.path.join(ROOT, "gpurun_out", "fuzz_mul_last_trial.txt"), "w").write("seed %d trial %d %r\n" % (
            seed, trials, dict(n=n, a33=a33, a65=a65, nw=nw, mode=mode, pinned=pinned, window=window, raw=raw)))
        if os.environ.get("FUZZ_VERBOSE"):
            print("trial", trials, dict(n=n, a33=a33, a65=a65, nw=nw, mode=mode, pinned=pinned, window=window, raw=raw), flush=True)
        d = Device(0, a33=a33, a65=a65)
        try:
            d.set_bloom(words)
            d.set_mul_window(window)
            if pinned and not raw:
                host_ptr = d.lib.ecl_hip_alloc_host(K.nbytes)
                assert host_ptr
                Kp = np.ctypeslib.as_array(C.cast(host_ptr, C.POINTER(C.c_uint64)), shape=K.shape)
                Kp[:] = K
                K = Kp
            cap = 2 * n + 16
            out = np.zeros(cap, dtype=capi.FOUND_DTYPE)
            cnt = C.c_uint32()
            if raw:
                rc = d.lib.ecl_hip_mul_batch_raw(d.h, text.ctypes.data, int(lens.sum()), table.ctypes.data, n, out.ctypes.data, cap, C.byref(cnt))
            else:
                rc = d.lib.ecl_hip_mul_batch(d.h, K.ctypes.data, n, out.ctypes.data, cap, C.byref(cnt))
            assert rc == 0, rc
        finally:
            if host_ptr:
                K = np.array(K)  # the checks below read the scalars after the page-locked copy is gone
                d.lib.ecl_hip_free_host(host_ptr)
            d.close()
        h33, h65, ok = orc.mul_hash160_many(K, a33, a65)
        want = set()
        for comp, hh, on in ((1, h33, a33), (0, h65, a65)):
            if not on:
                continue
            hit = np.zeros(n, np.uint8)
            hh = np.ascontiguousarray(hh)
            L.orc_blf_has_many(words.ctypes.data, nw, hh.ctypes.data, n, hit.ctypes.data)
            for i in np.nonzero(hit & ok)[0]:
                want.add((int(i), comp, tuple(int(v) for v in hh[i])))
        got = {(int(r["key_offset"]), int(r["compressed"]), tuple(int(v) for v in r["h160"])) for r in out[: cnt.value]}
        if got != want or cnt.value != len(want):
            print("MISMATCH", dict(n=n, a33=a33, a65=a65, nw=nw, mode=mode, pinned=pinned, window=window, raw=raw, got=len(got), want=len(want), seed=seed, trial=trials))
            sys.exit(1)
        trials, scalars, hits = trials + 1, scalars + n, hits + len(want)
    line = "# tools/fuzz_mul_gpu.py %s %d: %d trials, %d scalars, %d compared hits, ALL EQUAL to the oracle (orc.mul_hash160_many + blf_has on every scalar)" % (
        budget, seed, trials, scalars, hits)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "fuzz_mul.txt"), "w").write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
