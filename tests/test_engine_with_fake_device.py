"""Host logic of engine.KeySearch (launch sizing, overflow retry, list confirm, calc_priv, the verify step, cmd_add's
job arithmetic and sharding) exercised on the CPU with the GPU replaced by a stand-in built on the oracle.
Only the host logic is under test here; the real device is covered by the -m gpu tests."""
import json
import os

import numpy as np
import pytest

import orc
from ecloop_amd import capi, engine

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]


class FakeDevice:
    """Same surface as capi.Device; key ranges are answered by the oracle through an all-ones-free path: every key of
    the requested range is hashed by the oracle's own add_range with the filter that was set."""
    calls = []

    def __init__(self, device=0, a33=True, a65=False, endo=False, ord_offs=0):
        self.a33, self.a65, self.endo, self.offs = a33, a65, endo, ord_offs
        self.words = None
        self.list = None
        self.lanes, self.half = 512, 64

    def close(self):
        pass

    def set_bloom(self, words):
        self.words = np.array(words, dtype=np.uint64)

    def set_list(self, hashes):
        self.list = None if hashes is None else {tuple(int(w) for w in h) for h in hashes}

    def reserve(self, nkeys, cap=4096):
        pass

    def set_geometry(self, half_group=0, max_lanes=0):
        self.half = half_group or self.half
        self.lanes = max_lanes or self.lanes

    def geometry(self):
        return self.half, self.lanes

    def add_range(self, start, nkeys, cap=4096):
        FakeDevice.calls.append((start, nkeys, cap))
        stride = 1 << self.offs
        flt = orc.OrcFilter(bloom_words=self.words)
        recs = []
        lam = engine.LAMBDA
        if stride == 1:
            # the oracle hashes whole 2048-key groups: ask for the covering range and keep the keys inside
            batches = [orc.add_range(flt, start, start + nkeys, a33=self.a33, a65=self.a65, endo=self.endo, verify=False,
                                     threads=4, cap=1 << 18)]
        else:
            # with a stride, a one-key-wide range makes the oracle hash exactly one 2048-key group (main.c:442)
            batches = [orc.add_range(flt, (start + g * 2048 * stride) % orc.N, (start + g * 2048 * stride) % orc.N + 1, a33=self.a33,
                                     a65=self.a65, endo=self.endo, offs=self.offs, verify=False, cap=1 << 16)
                       for g in range((nkeys + 2047) // 2048)]
        for rc, out, n, _, hashed in batches:
          assert rc == 0
          for i in range(n):
            r = out[i]
            k = orc.val(r.pk)
            if r.endo in (1, 3, 5):
                k = (-k) % orc.N
            if r.endo in (2, 3):
                k = k * pow(lam, -1, orc.N) % orc.N
            if r.endo in (4, 5):
                k = k * pow(lam, -2, orc.N) % orc.N
            off = ((k - start) % orc.N) >> self.offs
            if off < nkeys and (self.list is None or tuple(int(w) for w in r.h160) in self.list):
                recs.append((off, [int(w) for w in r.h160], r.endo, r.compressed))
        arr = np.zeros(min(len(recs), cap), dtype=capi.FOUND_DTYPE)
        for j, (off, h, e, c) in enumerate(recs[:cap]):
            arr[j]["key_offset"], arr[j]["h160"], arr[j]["endo"], arr[j]["compressed"] = off, h, e, c
        return arr, len(recs)

    def diag_mulg(self, ks):
        pts = [orc.point_of(k) for k in ks]
        return [p[0] for p in pts], [p[1] for p in pts], np.ones(len(ks), dtype=np.uint8)

    def verify(self, ks):
        xs, ys, ok = self.diag_mulg(ks)
        h33, h65 = self.diag_hash160(xs, ys)
        return h33, h65, ok

    def diag_hash160(self, xs, ys):
        return (np.array([orc.hash160(x, y, True) for x, y in zip(xs, ys)], dtype=np.uint32).reshape(-1, 5),
                np.array([orc.hash160(x, y, False) for x, y in zip(xs, ys)], dtype=np.uint32).reshape(-1, 5))


@pytest.fixture
def fake(monkeypatch):
    FakeDevice.calls = []
    monkeypatch.setattr(engine, "Device", FakeDevice)
    return FakeDevice


def test_cmd_add_known_answers_and_counters(fake):
    flt = engine.load_filter(os.path.join(GOLD, "btc-puzzles-hash"))
    ks = engine.KeySearch(flt, launch_keys=1 << 14)
    ks.cmd_add(0x8000, 0xFFFF)
    g = G["ci_smoke_8000_ffff"]
    assert [r.line() for r in ks.found] == g["lines"] and (ks.k_found, ks.k_checked) == (g["status_found"], g["status_checked"])
    # 32768 keys hashed (QUIRK: one 32767-key job rounded up), launches are whole sweeps of lanes*2*half_group keys
    assert sum(c[1] for c in fake.calls) == 32768 and all(c[1] % (512 * 2 * 64) == 0 or c is fake.calls[-1] for c in fake.calls)


def test_sharded_cmd_add_unions_to_the_single_rank_result(fake):
    flt = engine.load_filter(os.path.join(GOLD, "btc-puzzles-hash"))
    whole = engine.KeySearch(flt)
    whole.cmd_add(0x8000, 0xFFFFF)
    parts = []
    for rank in range(3):
        ks = engine.KeySearch(flt)
        ks.cmd_add(0x8000, 0xFFFFF, rank=rank, world=3)
        parts += [r.line() for r in ks.found]
    assert sorted(parts) == sorted(r.line() for r in whole.found) and len(parts) == 5


def test_overflow_retry_and_endo_private_keys(fake):
    """dense filter: the first call overflows its buffer, the retry gets everything; endo private keys via calc_priv"""
    g = G["dense_fp_cu_endo"]
    from synth import synth_bloom_words
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    ks = engine.KeySearch(engine.Filter(words), a33=True, a65=True, endo=True)
    ks.add_keys(0x8000, 2048, cap=8)
    assert orc.digest([r.line() for r in ks.found]) == g["sha256_sorted"]
    assert len(fake.calls) == 2 and fake.calls[1][2] == g["count"]


def test_verify_catches_a_wrong_hit(fake, monkeypatch):
    flt = engine.load_filter(os.path.join(GOLD, "btc-puzzles-hash"))
    ks = engine.KeySearch(flt)
    orig = FakeDevice.diag_hash160

    def corrupt(self, xs, ys):
        a, b = orig(self, xs, ys)
        a[0][0] ^= 1
        return a, b

    monkeypatch.setattr(FakeDevice, "diag_hash160", corrupt)
    with pytest.raises(engine.EclError, match="hash mismatch"):
        ks.cmd_add(0x8000, 0xFFFF)


def test_strided_window(fake):
    a = (1 << 164) + 0x12345
    ks = engine.KeySearch(engine.Filter(np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64)), ord_offs=128)
    ks.cmd_add(a, a + 1)
    g = G["dump33_stride128"]
    assert orc.digest([r.line() for r in ks.found]) == g["sha256_sorted"] and ks.k_checked == g["status_checked"]
