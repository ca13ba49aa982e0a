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


from fake_device import FakeDevice  # noqa: E402


@pytest.fixture
def fake(monkeypatch):
    FakeDevice.calls = []
    FakeDevice.fetches = []
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
    # the records beyond the caller's buffer are read from what the device kept: one launch, one fetch
    assert len(fake.calls) == 1 and fake.fetches == [(8, g["count"] - 8)]


def test_overflow_beyond_what_the_device_keeps_reruns_the_launch(fake, monkeypatch):
    """more hits than the device keeps of one call (the library: max(cap, 2^20)): the fetch comes back short and the launch is repeated"""
    g = G["dense_fp_cu_endo"]
    from synth import synth_bloom_words
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    monkeypatch.setattr(FakeDevice, "keep_min", 16)
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
