"""The look-ahead of the library (ecloop_amd/csrc/abi_lookahead.h: a caller's small contiguous jobs answered from large sweeps) on the
CPU: the header is compiled verbatim around a stand-in search kernel whose hits are a known function of the key index
(csrc/tools/lookahead_host.cpp), and every call's records are compared with that function evaluated in numpy.  What the -m gpu tests
(tests/test_gpu_lookahead.py) cannot reach is here: several "devices", sweeps that overflow the record store, sweeps that fail, long random
call sequences, worker threads racing for jobs."""
import ctypes as C
import os
import random
import subprocess
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOUND = np.dtype([("key_offset", "<u8"), ("h160", "<u4", (5,)), ("endo", "u1"), ("compressed", "u1"), ("pad", "u1", (2,))])
E_OVERFLOW, E_RANGE = -4, -6
M64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def L(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("lahost") / "liblahost.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so,
                    os.path.join(ROOT, "ecloop_amd", "csrc", "tools", "lookahead_host.cpp")], check=True)
    lib = C.CDLL(so)
    lib.lh_open.restype = C.c_void_p
    lib.lh_open.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64]
    lib.lh_close.argtypes = [C.c_void_p]
    lib.lh_cluster.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    lib.lh_poison.argtypes = [C.c_void_p, C.c_uint64]
    lib.lh_fix_geometry.argtypes = [C.c_void_p]
    lib.lh_add_range.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.lh_device_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
    lib.ecl_hip_set_lookahead.argtypes = [C.c_void_p, C.c_uint64]
    lib.ecl_hip_set_scan_end.argtypes = [C.c_void_p, C.c_void_p]
    lib.ecl_hip_get_lookahead_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4
    lib.ecl_hip_fetch_found.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    return lib


def mix(i, f):
    with np.errstate(over="ignore"):
        z = i + np.uint64((f * 0x9E3779B97F4A7C15) & M64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


class Ctx:
    """one stand-in context; `expected` evaluates the stand-in kernel's hit function for a call"""

    def __init__(self, L, dev=0, flags=1, offs=0, filt=7, one_in=997):
        self.L, self.offs, self.filt, self.one_in = L, offs, filt, one_in
        self.cluster = None
        self.h = L.lh_open(dev, flags, offs, filt, one_in)

    def close(self):
        self.L.lh_close(self.h)

    def limbs(self, v):
        return (C.c_uint64 * 4)(*[(v >> (64 * i)) & M64 for i in range(4)])

    def set_cluster(self, lo, n, one_in):
        self.cluster = (lo, n, one_in)
        self.L.lh_cluster(self.h, lo, n, one_in)

    def set_end(self, end):
        self.L.ecl_hip_set_scan_end(self.h, self.limbs(end) if end is not None else None)

    def add(self, start, n, cap=1 << 16):
        out = np.zeros(cap, dtype=FOUND)
        cnt = C.c_uint32()
        rc = self.L.lh_add_range(self.h, self.limbs(start), n, out.ctypes.data, cap, C.byref(cnt))
        return rc, out[: min(cap, cnt.value)], cnt.value

    def fetch(self, first, n):
        out = np.zeros(max(n, 1), dtype=FOUND)
        got = C.c_uint32()
        assert self.L.ecl_hip_fetch_found(self.h, first, out.ctypes.data, n, C.byref(got)) == 0
        return out[: got.value]

    def expected(self, start, n):
        i0 = start >> self.offs
        i = np.arange(i0, i0 + n, dtype=np.uint64)
        m = mix(i, self.filt)
        one = np.full(n, self.one_in, dtype=np.uint64)
        if self.cluster:
            lo, ln, d = self.cluster
            one[(i >= np.uint64(lo)) & (i < np.uint64(lo + ln))] = d
        at = np.nonzero(m % one == 0)[0]
        recs = np.zeros(len(at), dtype=FOUND)
        recs["key_offset"] = at
        recs["compressed"] = (m[at] >> np.uint64(40)) & np.uint64(1)
        recs["endo"] = (m[at] >> np.uint64(44)) % np.uint64(6)
        for w in range(5):
            recs["h160"][:, w] = mix(i[at], self.filt + 1 + w) & np.uint64(0xFFFFFFFF)
        return recs

    def stats(self):
        v = [C.c_uint64() for _ in range(4)]
        self.L.ecl_hip_get_lookahead_stats(self.h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def device(self):
        v = [C.c_uint64() for _ in range(3)]
        self.L.lh_device_stats(self.h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)


def same(a, b):
    k = lambda r: sorted(zip(r["key_offset"].tolist(), r["endo"].tolist(), r["compressed"].tolist(), map(tuple, r["h160"].tolist())))
    return k(a) == k(b)


@pytest.mark.parametrize("offs", [0, 64, 200])
@pytest.mark.parametrize("hint", [False, True])
def test_every_call_of_a_long_scan_gets_its_own_records(L, offs, hint):
    """3000 contiguous jobs of 4096 keys (strides 1, 2^64 and 2^200: the two forms of la_offset): answered from sweeps of up to 2^22 keys,
    each call's records are the hit function over its own keys; the launches together cover every key at most 1.5 times without the
    hint and exactly once with it"""
    c = Ctx(L, offs=offs, one_in=613)
    try:
        A, n, jobs = (0x1234567 << offs), 4096, 3000
        if hint:
            c.set_end(A + ((jobs * n - 7) << offs))
        for j in range(jobs):
            rc, got, cnt = c.add(A + ((j * n) << offs), n)
            assert rc == 0 and cnt == len(got) and same(got, c.expected(A + ((j * n) << offs), n)), j
        sweeps, swept, served, served_keys = c.stats()
        launches, launched, _ = c.device()
        assert served_keys == served * n and served > jobs * 0.9 and sweeps >= 3
        assert launched == jobs * n if hint else launched <= jobs * n * 3 // 2
    finally:
        c.close()


def test_random_call_sequences_equal_the_hit_function(L):
    """streaks of contiguous jobs of random sizes, jumps forwards and backwards, jobs inside finished sweeps, jobs straddling their ends,
    a changing scan end, sizes above a quarter of the sweep limit (never looked ahead for): whatever is answered from a sweep equals the
    call's own hits"""
    rnd = random.Random(5)
    c = Ctx(L, one_in=251)
    try:
        pos = 1 << 33
        served_before = 0
        for trial in range(400):
            kind = rnd.random()
            n = rnd.choice([1, 100, 2048, 4096, 5000, 1 << 14, 1 << 16])
            if kind < 0.15:
                pos = rnd.randrange(1 << 30, 1 << 40)
            if kind < 0.3:
                c.set_end(pos + n * rnd.randrange(1, 300) - rnd.randrange(0, n) if rnd.random() < 0.7 else None)
            for _ in range(rnd.randrange(1, 40)):
                r = rnd.random()
                if r < 0.05:
                    s, m = pos - rnd.randrange(0, 20) * n, n            # back into what was just covered
                elif r < 0.08:
                    s, m = pos + rnd.randrange(1, 5) * n, n             # skips ahead
                elif r < 0.1:
                    s, m = pos, (1 << 21) + rnd.randrange(0, 1000)      # too large to be looked ahead for
                else:
                    s, m = pos, n
                    pos += n
                rc, got, cnt = c.add(s, m, cap=1 << 15)
                want = c.expected(s, m)
                assert rc == 0 and cnt == len(want) and same(got, want), (trial, hex(s), m)
        assert c.stats()[2] > 2000 > served_before
    finally:
        c.close()


@pytest.mark.parametrize("threads,devices", [(2, 1), (4, 2), (8, 8), (6, 1)])
def test_worker_threads_pulling_jobs_from_one_counter(L, threads, devices):
    """the reference's scheduler (main.c:405-435): N worker threads, one context each, jobs of equal size handed out from a mutex-guarded
    counter.  Contexts on `devices` stand-in devices with one filter = one group: every job's records are its own, wherever the sweep
    that covered it ran; with the end known no key is launched twice (a few jobs may be launched singly: those whose worker called
    before the first sweep, or late); contexts on one device elect one sweeper"""
    n, jobs, A = 2048, 6000, 0x5000_0000
    ctxs = [Ctx(L, dev=t % devices, one_in=401) for t in range(threads)]
    results, errors, lock, state = {}, [], threading.Lock(), {"next": 0}

    def worker(c):
        try:
            c.set_end(A + jobs * n)
            while True:
                with lock:
                    j = state["next"]
                    state["next"] += 1
                if j >= jobs:
                    return
                rc, got, cnt = c.add(A + j * n, n)
                assert rc == 0 and cnt == len(got)
                results[j] = got
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    try:
        ts = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors[0]
        for j in range(jobs):
            assert same(results[j], ctxs[0].expected(A + j * n, n)), j
        st = [c.stats() for c in ctxs]
        dv = [c.device() for c in ctxs]
        # no key launched twice (the hint bounds the sweeps; single launches: jobs before the first sweep, late ones, and - on a shared device -
        # jobs that reached the front while the device's one sweeping context was not the caller: how many depends on thread timing)
        assert sum(d[1] for d in dv) <= (jobs + 8 * threads) * n and sum(s[2] for s in st) >= 1  # (how many: a matter of scheduling - typically > 90 %)
        for dev in range(devices):  # one sweeper per device
            assert sum(1 for c, s in zip(ctxs, st) if c is not None and s[0] > 0 and ctxs.index(c) % devices == dev) <= 1
    finally:
        [c.close() for c in ctxs]


def test_contexts_with_other_flags_stride_or_filter_are_other_groups(L):
    a, b, c, d = Ctx(L, filt=1), Ctx(L, filt=2), Ctx(L, filt=1, flags=3), Ctx(L, filt=1)
    try:
        n, A = 4096, 1 << 32
        for ctx in (a, b, c, d):
            ctx.set_end(A + 64 * n)
        for j in range(40):
            for ctx in (a, b, c):
                rc, got, cnt = ctx.add(A + j * n, n)
                assert rc == 0 and same(got, ctx.expected(A + j * n, n))
        # d shares a's filter, flags and stride: a's sweep answers it without a launch of its own
        rc, got, cnt = d.add(A + 50 * n, n)
        assert rc == 0 and same(got, d.expected(A + 50 * n, n)) and d.device()[0] == 0 and d.stats()[2] == 1
        assert all(x.device()[0] == 2 for x in (a, b, c))  # each: the first job + one sweep
    finally:
        [x.close() for x in (a, b, c, d)]


def test_a_sweep_with_more_hits_than_can_be_kept_is_dropped_once(L):
    """hits clustered where the first jobs did not look (density unknown to the plan): the sweep comes back with more records than the
    device keeps -> dropped, the pattern goes on with launches of its own jobs and is not swept again (no repeated waste); when the
    pattern changes, look-ahead resumes.  A moderately dense cluster (more than the first copy holds, fewer than the device keeps) is
    read in two parts."""
    c = Ctx(L, one_in=100003)
    try:
        n, A = 4096, 1 << 34
        c.set_end(A + 4000 * n)
        c.set_cluster(A + 100 * n, 1 << 20, 2)          # 2^19 hits in 2^20 keys: more than HELD_MAX = 2^18
        for j in range(400):
            rc, got, cnt = c.add(A + j * n, n, cap=4096)
            want = c.expected(A + j * n, n)
            assert cnt == len(want) and rc == (E_OVERFLOW if cnt > 4096 else 0), j
            if rc == 0:
                assert same(got, want), j
            else:
                rest = c.fetch(4096, cnt - 4096)
                assert same(np.concatenate([got, rest]), want), j
        sweeps, swept, served, _ = c.stats()
        launches, launched, _ = c.device()
        assert sweeps == 0 and served == 0 and launches == 401 and launched <= (400 + 1024) * n  # one sweep attempted, dropped
        # a new pattern elsewhere: swept again
        B = 1 << 36
        c.set_end(B + 100 * n)
        for j in range(100):
            rc, got, cnt = c.add(B + j * n, n)
            assert rc == 0 and same(got, c.expected(B + j * n, n))
        assert c.stats()[0] == 1 and c.stats()[2] == 99
        # 200 000 hits in one sweep: above the 2^17 records copied with the launch, below what the device keeps
        D = 1 << 38
        c.set_cluster(D + 10 * n, 400000, 2)
        c.set_end(D + 1000 * n)
        for j in range(120):
            rc, got, cnt = c.add(D + j * n, n, cap=4096)
            want = c.expected(D + j * n, n)
            assert cnt == len(want), j
            got = got if rc == 0 else np.concatenate([got, c.fetch(4096, cnt - 4096)])
            assert same(got, want), j
        assert c.stats()[0] == 2 and c.stats()[2] == 99 + 119
    finally:
        c.close()


def test_a_sweep_that_fails_leaves_the_jobs_to_their_own_launches(L):
    """a scan that runs into a key the walk cannot represent (ECL_E_RANGE: the scalar 0): the sweep fails, the jobs before that key are
    launched one by one and succeed, the job that contains it fails with the launch's own error"""
    c = Ctx(L, one_in=503)
    try:
        n, A = 4096, 1 << 35
        L.lh_poison(c.h, A + 50 * n + 17)
        c.set_end(A + 200 * n)
        for j in range(50):
            rc, got, cnt = c.add(A + j * n, n)
            assert rc == 0 and same(got, c.expected(A + j * n, n)), j
        rc, got, cnt = c.add(A + 50 * n, n)
        assert rc == E_RANGE and cnt == 0
        for j in range(51, 60):
            rc, got, cnt = c.add(A + j * n, n)
            assert rc == 0 and same(got, c.expected(A + j * n, n)), j
        assert c.stats()[0] == 0 and c.device()[2] == 2  # the sweep and the job itself
    finally:
        c.close()


def test_switched_off_by_the_caller_or_by_a_fixed_geometry(L):
    for how in ("off", "geometry"):
        c = Ctx(L)
        try:
            if how == "off":
                assert L.ecl_hip_set_lookahead(c.h, 0) == 0
            else:
                L.lh_fix_geometry(c.h)
            c.set_end(1 << 40)
            for j in range(50):
                rc, got, cnt = c.add((1 << 32) + j * 4096, 4096)
                assert rc == 0 and same(got, c.expected((1 << 32) + j * 4096, 4096))
            assert c.stats() == (0, 0, 0, 0) and c.device()[:2] == (50, 50 * 4096)
        finally:
            c.close()
    c = Ctx(L)
    try:
        assert L.ecl_hip_set_lookahead(c.h, 1000) == -1 and L.ecl_hip_set_lookahead(c.h, 1 << 33) == -1 and L.ecl_hip_set_lookahead(c.h, 1 << 24) == 0
    finally:
        c.close()


@pytest.mark.parametrize("contexts", [2, 8])
def test_one_shard_per_context_stays_one_launch_per_context(L, contexts):
    """a host that knows its GPUs hands every context ONE contiguous shard of the scan (the host program's `-t N` on scans up to 2^33 keys):
    N equal jobs, each starting where the one before ended - the reference's pattern to the letter, but sweeping it would put all shards
    on one GPU while the others wait.  A sweep must replace at least 4 jobs per context of the group: here every context launches its
    own shard, in parallel, and nothing is swept."""
    n, A = 1 << 18, 0x9000_0000
    ctxs = [Ctx(L, dev=t, one_in=733) for t in range(contexts)]
    results, errors = {}, []

    def worker(t):
        try:
            ctxs[t].set_end(A + contexts * n)
            rc, got, cnt = ctxs[t].add(A + t * n, n)
            assert rc == 0
            results[t] = got
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    try:
        for c in ctxs:  # all contexts are members of the group before the first job arrives
            c.add(1 << 50, 64)
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(contexts)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors[0]
        for t in range(contexts):
            assert same(results[t], ctxs[t].expected(A + t * n, n))
            assert ctxs[t].stats() == (0, 0, 0, 0) and ctxs[t].device()[:2] == (2, n + 64), (t, ctxs[t].stats(), ctxs[t].device())
    finally:
        [c.close() for c in ctxs]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_sequences_from_worker_threads_on_several_devices(L, seed):
    """the differential fuzz of tools/fuzz_lookahead_gpu.py in the shape hardware here cannot give it: 2 .. 8 worker threads whose contexts sit
    on 1 .. 8 stand-in devices pull a sequence of calls from one list - mostly the next job of a scan, now and then a jump, a job from
    behind, one that straddles two jobs, another size, a second scan interleaved - with and without the scan's end told; every call's
    records are the hit function over its own keys, and no key is launched more than twice (sweeps never overlap; only odd calls that
    cut across a sweep can repeat keys)"""
    rnd = random.Random(seed)
    for trial in range(6):
        threads = rnd.choice([2, 3, 4, 8])
        devices = rnd.choice([1, 2, threads])
        n = rnd.choice([512, 2048, 4096])
        offs = rnd.choice([0, 0, 64])
        one_in = rnd.choice([97, 401, 5003])
        ctxs = [Ctx(L, dev=t % devices, offs=offs, filt=100 + seed, one_in=one_in) for t in range(threads)]
        A, B = rnd.randrange(1 << 30, 1 << 40), rnd.randrange(1 << 41, 1 << 42)
        seq, pa, pb = [], 0, 0
        for _ in range(rnd.choice([300, 1500, 4000])):
            r = rnd.random()
            if r < 0.02:
                pa += rnd.randrange(1, 9) * n
            elif r < 0.03 and pa > 8 * n:
                seq.append((A + pa - rnd.randrange(1, 8) * n, n))
            elif r < 0.04:
                seq.append((A + pa + n // 2, n))
            elif r < 0.05:
                seq.append((A + pa, rnd.choice([1, 77, n // 2, 3 * n])))
            elif r < 0.10:
                seq.append((B + pb, n))  # a second scan elsewhere, interleaved now and then
                pb += n
                continue
            seq.append((A + pa, n))
            pa += n
        hint = rnd.random() < 0.5
        results, errors, lock, state = {}, [], threading.Lock(), {"next": 0}

        def worker(c):
            try:
                c.set_end((A + pa) << offs if hint else None)
                while True:
                    with lock:
                        i = state["next"]
                        state["next"] += 1
                    if i >= len(seq):
                        return
                    s, m = seq[i]
                    rc, got, cnt = c.add(s << offs, m, cap=1 << 15)
                    assert rc == 0 and cnt == len(got)
                    results[i] = got
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        try:
            ts = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
            [t.start() for t in ts]
            [t.join() for t in ts]
            assert not errors, errors[0]
            for i, (s, m) in enumerate(seq):
                assert same(results[i], ctxs[0].expected(s << offs, m)), (trial, i, hex(s), m)
            asked = sum(m for _, m in seq)
            assert sum(c.device()[1] for c in ctxs) <= 2 * asked + (1 << 22) * threads, trial
        finally:
            [c.close() for c in ctxs]
