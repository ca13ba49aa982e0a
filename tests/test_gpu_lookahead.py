"""Look-ahead over small contiguous jobs (ecloop_amd/csrc/abi_lookahead.h): a caller that hands the library the reference's 2^21-key
jobs one after the other (cmd_add_worker, main.c:405-435) is answered from large sweeps - and must receive, call by call, exactly the
records a launch of its own would have reported.  The yardstick of every test is the same sequence of calls on a context with the
look-ahead switched off (plain launches: the path the golden-dump tests of test_gpu_add.py pin to the reference), plus the oracle on
a sample of the hits."""
import ctypes as C
import threading

import numpy as np
import pytest

import orc
from synth import synth_bloom_words

pytestmark = pytest.mark.gpu
JOB = 1 << 21  # MAX_JOB_SIZE, main.c:16


def key(recs):
    return sorted((int(r["key_offset"]), int(r["endo"]), int(r["compressed"]), tuple(int(v) for v in r["h160"])) for r in recs)


def plain_device(words, **kw):
    from ecloop_amd import Device
    d = Device(0, **kw)
    d.set_lookahead(0)
    d.set_bloom(words)
    return d


def ahead_device(words, max_keys=None, **kw):
    from ecloop_amd import Device
    d = Device(0, **kw)
    if max_keys is not None:
        d.set_lookahead(max_keys)
    d.set_bloom(words)
    return d


@pytest.mark.parametrize("hint", [False, True])
def test_contiguous_jobs_get_the_records_of_their_own_launch(hint):
    """96 jobs of 2^21 keys, each starting where the one before ended: with the scan's end known the second job already runs a sweep to the
    end (never past the job that contains it); without, the sweeps grow with what the pattern has consumed.  Every call: same records as
    its own launch on the plain context.  The swept keys never exceed the scan, and most calls never launched."""
    words = synth_bloom_words(1 << 16, 5, "a|(b&c)")  # 0.625^20: a hit every ~12000 keys, ~170 per job
    A, jobs = 0x1_0000_0000, 96
    end = A + (jobs - 1) * JOB + 12345  # inside the last job: a worker stops when range_s >= range_e (main.c:420)
    p, a = plain_device(words), ahead_device(words)
    try:
        if hint:
            a.set_scan_end(end)
        for j in range(jobs):
            want, nw = p.add_range(A + j * JOB, JOB, cap=4096)
            got, ng = a.add_range(A + j * JOB, JOB, cap=4096)
            assert ng == nw == len(want) and key(got) == key(want), j
        sweeps, swept, served_calls, served_keys = a.lookahead_stats()
        assert sweeps >= 1 and served_keys == served_calls * JOB
        if hint:
            assert sweeps == 1 and swept == (jobs - 1) * JOB and served_calls == jobs - 1
        else:
            assert swept <= jobs * JOB * 3 // 2 and served_calls >= jobs // 2  # (at most half of what was consumed is looked ahead)
        # the oracle on a sample of what the look-ahead delivered: hash160 of the key at that offset, and that the filter passes it
        flt = orc.OrcFilter(bloom_words=words)
        got, _ = a.add_range(A + 40 * JOB, JOB, cap=4096)  # (answered from nothing prepared: a job from before the front)
        ks = [A + 40 * JOB + int(r["key_offset"]) for r in got[:64]]
        h33, _, ok = orc.mul_hash160_many(np.array([[(k >> (64 * i)) & orc.MASK64 for i in range(4)] for k in ks], dtype=np.uint64))
        assert ok.all() and [tuple(h) for h in h33] == [tuple(r["h160"]) for r in got[:64]] and all(flt.check([int(v) for v in h]) for h in h33)
    finally:
        p.close(), a.close()


def test_end_of_range_is_never_passed_and_the_stride_is_honoured():
    """`-d 64:..`-style scan (stride 2^64, `-a cu -endo`: 12 hashes per key) whose end lies 10.5 jobs ahead: sweeps cover the 11 jobs
    that start before the end and not one key more; records identical to plain launches, endo and address form included."""
    words = synth_bloom_words(1 << 14, 9, "a|(b&c)")
    offs, job = 64, 1 << 18
    A = (0x3 << 200) | (0x1234 << 64) | 0x99
    kw = dict(a33=True, a65=True, endo=True, ord_offs=offs)
    p, a = plain_device(words, **kw), ahead_device(words, max_keys=1 << 22, **kw)  # sweeps of at most 16 jobs
    try:
        a.set_scan_end(A + ((10 * job + job // 2) << offs))
        for j in range(11):
            want, nw = p.add_range(A + ((j * job) << offs), job, cap=1 << 14)
            got, ng = a.add_range(A + ((j * job) << offs), job, cap=1 << 14)
            assert ng == nw and key(got) == key(want) and nw > 100, j
        sweeps, swept, served_calls, _ = a.lookahead_stats()
        assert (sweeps, swept, served_calls) == (1, 10 * job, 10)
    finally:
        p.close(), a.close()


def test_jobs_that_do_not_continue_fall_back_to_their_own_launch():
    """jobs that jump (another window of `rnd`, a scheduler that hands a context every other job, sizes that change): nothing is swept
    beyond what a pattern justified, every call has its own launch's records; a job INSIDE a finished sweep is answered from it even
    out of order, one that straddles its end is launched."""
    words = synth_bloom_words(1 << 16, 6, "a|(b&c)")
    A = 0x2_0000_0000
    p, a = plain_device(words), ahead_device(words)
    try:
        a.set_scan_end(A + 64 * JOB)
        calls = [(A + j * JOB, JOB) for j in (0, 2, 4, 6, 9, 13)]                        # strided: never two in a row
        calls += [(A + (1 << 40) + j * JOB, JOB) for j in range(3)]                       # a pattern elsewhere: hint lies behind -> no sweep
        calls += [(A + j * JOB, JOB) for j in (20, 21)]                                   # back: second one sweeps 21 .. 63
        calls += [(A + j * JOB, JOB) for j in (40, 30, 63, 22)]                           # inside the sweep, any order
        calls += [(A + 63 * JOB + JOB // 2, JOB), (A + 25 * JOB + 5, 1000), (A + 64 * JOB, JOB), (A + 26 * JOB, JOB // 2)]  # straddling / odd sizes
        for i, (s, n) in enumerate(calls):
            want, nw = p.add_range(s, n, cap=4096)
            got, ng = a.add_range(s, n, cap=4096)
            assert ng == nw and key(got) == key(want), (i, hex(s), n)
        sweeps, swept, served_calls, served_keys = a.lookahead_stats()
        assert sweeps == 1 and swept == 43 * JOB and served_calls >= 6
    finally:
        p.close(), a.close()


def test_a_call_with_too_small_a_buffer_overflows_and_fetches_from_the_sweep():
    """a job answered from a sweep whose own records exceed the caller's `cap`: ECL_E_OVERFLOW with the total, the first `cap` records
    delivered, the rest through ecl_hip_fetch_found - as after a launch (include/ecloop_hip.h)."""
    from ecloop_amd import capi
    words = synth_bloom_words(1 << 16, 7, "a|(b&c)")
    A = 0x3_0000_0000
    p, a = plain_device(words), ahead_device(words)
    try:
        a.set_scan_end(A + 8 * JOB)
        for j in range(8):
            want, nw = p.add_range(A + j * JOB, JOB, cap=4096)
            out = np.zeros(50, dtype=capi.FOUND_DTYPE)
            n = C.c_uint32()
            s = capi.limbs(A + j * JOB)
            rc = a.lib.ecl_hip_add_range(a.h, s.ctypes.data, JOB, out.ctypes.data, 50, C.byref(n))
            assert rc == capi.E_OVERFLOW and n.value == nw > 50
            rest = a.fetch_found(50, n.value - 50)
            assert len(rest) == n.value - 50 and key(np.concatenate([out, rest])) == key(want)
            assert len(a.fetch_found(n.value, 10)) == 0
        assert a.lookahead_stats()[2] == 7
    finally:
        p.close(), a.close()


def test_a_dense_filter_is_swept_in_small_pieces_or_not_at_all():
    """hit density decides how far a sweep may go (about 2^16 records): a filter that passes one key in ~300 is looked ahead by a few jobs
    at a time, the all-ones filter (every key a hit: the dump tests) not at all - records equal to plain launches either way."""
    A = 0x5_0000_0000
    for words, job, expect_sweeps in ((synth_bloom_words(1 << 16, 8, "a|b"), 1 << 20, True), (np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64), 1 << 16, False)):
        p, a = plain_device(words), ahead_device(words)
        try:
            a.set_scan_end(A + 4096 * job)
            for j in range(40):
                want, nw = p.add_range(A + j * job, job, cap=1 << 16)
                got, ng = a.add_range(A + j * job, job, cap=1 << 16)
                assert ng == nw and key(got) == key(want), j
            sweeps, swept, served_calls, _ = a.lookahead_stats()
            assert (sweeps > 2 and served_calls > 20 and swept / sweeps <= (1 << 25)) if expect_sweeps else (sweeps == 0 and served_calls == 0)
        finally:
            p.close(), a.close()


@pytest.mark.parametrize("contexts", [2, 4])
def test_worker_threads_on_several_contexts_share_the_sweeps(contexts):
    """the reference's -t N: N worker threads, each with its own context, pull 2^21-key jobs from one mutex-guarded counter
    (main.c:418-431).  The contexts share a filter, so they form one group: a job handed to one context is answered from the sweep another
    one ran.  Every job's records equal its own plain launch; the sweeps together cover each key at most once."""
    words = synth_bloom_words(1 << 16, 11, "a|(b&c)")
    A, jobs = 0x7_0000_0000, 200
    p = plain_device(words)
    devs = [ahead_device(words) for _ in range(contexts)]
    lock, state, results, errors = threading.Lock(), {"next": 0}, {}, []

    def worker(d):
        try:
            d.set_scan_end(A + jobs * JOB)
            while True:
                with lock:
                    j = state["next"]
                    state["next"] += 1
                if j >= jobs:
                    return
                got, n = d.add_range(A + j * JOB, JOB, cap=4096)
                assert n == len(got)
                results[j] = key(got)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    try:
        ts = [threading.Thread(target=worker, args=(d,)) for d in devs]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors
        for j in range(jobs):
            want, _ = p.add_range(A + j * JOB, JOB, cap=4096)
            assert results[j] == key(want), j
        stats = [d.lookahead_stats() for d in devs]
        assert sum(s[1] for s in stats) <= jobs * JOB and sum(s[2] for s in stats) >= jobs // 4  # (typically > 95 %: how the threads interleave decides)
    finally:
        p.close()
        [d.close() for d in devs]


def test_contexts_with_different_filters_do_not_share():
    """contexts whose filters differ walk the same jobs alternately: each must see its own filter's hits and run its own sweeps (groups are
    keyed by a fingerprint over ALL filter words as they are resident on the device, the flags and the stride) - a filter with every other
    word changed, and one that differs from the first in a single bit"""
    w1 = synth_bloom_words(1 << 16, 12, "a|(b&c)")
    w2 = w1.copy()
    w2[::2] |= np.uint64(0x00FF00FF00FF00FF)
    w3 = w1.copy()
    w3[33333] ^= np.uint64(1 << 17)
    A = 0x9_0000_0000
    p1, p2, p3, a1, a2, a3 = plain_device(w1), plain_device(w2), plain_device(w3), ahead_device(w1), ahead_device(w2), ahead_device(w3)
    try:
        for j in range(12):
            for p, a in ((p1, a1), (p2, a2), (p3, a3)):
                want, nw = p.add_range(A + j * JOB, JOB, cap=1 << 14)
                got, ng = a.add_range(A + j * JOB, JOB, cap=1 << 14)
                assert ng == nw and key(got) == key(want), j
        # every context swept for itself (a shared group would have answered the later contexts from the first one's sweeps)
        assert all(a.lookahead_stats()[0] >= 1 and a.lookahead_stats()[2] >= 4 for a in (a1, a2, a3))
        # ... while a fourth context with the first one's filter - uploaded from another host copy - does join its group
        a4 = ahead_device(w1.copy())
        try:
            a1.set_scan_end(A + 64 * JOB), a4.set_scan_end(A + 64 * JOB)
            for j in range(12, 40):
                got, ng = (a1 if j % 2 else a4).add_range(A + j * JOB, JOB, cap=1 << 14)
                want, nw = p1.add_range(A + j * JOB, JOB, cap=1 << 14)
                assert ng == nw and key(got) == key(want), j
            assert a4.lookahead_stats()[2] >= 10 and a4.lookahead_stats()[0] + a1.lookahead_stats()[0] <= 4
        finally:
            a4.close()
    finally:
        [d.close() for d in (p1, p2, p3, a1, a2, a3)]
