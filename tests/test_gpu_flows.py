"""The reference's known-answer flows end to end on the GPU through the host mirror (engine.KeySearch):
config 1, `make add` (9 keys), the endo/list flow, `make mul` (1080 keys), and a default-geometry scan whose
false-positive list is compared with the oracle."""
import json
import os

import numpy as np
import pytest

import orc
from synth import synth_bloom_words

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]


def run_add(name, rs, re_, flt, **kw):
    from ecloop_amd.engine import KeySearch
    ks = KeySearch(flt, device=0, **kw)
    try:
        ks.cmd_add(rs, re_)
        lines = sorted(r.line() for r in ks.found)
        g = G[name]
        assert ks.k_found == g["status_found"] and ks.k_checked == g["status_checked"]
        assert orc.digest(lines) == g["sha256_sorted"]
        return lines
    finally:
        ks.close()


def test_config1_and_make_add_and_ci_smoke():
    from ecloop_amd.engine import load_filter
    flt = load_filter(os.path.join(GOLD, "btc-puzzles-hash"))
    lines = run_add("cfg1_list_800000_ffffff", 0x800000, 0xFFFFFF, flt)
    assert lines == G["cfg1_list_800000_ffffff"]["lines"] and lines[0].endswith("dc2a04")
    lines = run_add("make_add_8000_ffffff", 0x8000, 0xFFFFFF, flt)
    assert lines == sorted(G["make_add_8000_ffffff"]["lines"]) and len(lines) == 9
    run_add("ci_smoke_8000_ffff", 0x8000, 0xFFFF, flt)


def test_endo_cu_list_flow():
    from ecloop_amd.engine import load_filter
    flt = load_filter(os.path.join(GOLD, "btc-puzzles-hash"))
    lines = run_add("endo_cu_list_8000_fffff", 0x8000, 0xFFFFF, flt, a65=True, endo=True)
    assert lines == sorted(G["endo_cu_list_8000_fffff"]["lines"])


def test_strided_range_through_cmd_add():
    from ecloop_amd.engine import Filter
    a = (1 << 164) + 0x12345
    run_add("dump33_stride128", a, a + 1, Filter(np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64)), ord_offs=128)


def test_make_mul_brainwallet_flow():
    from ecloop_amd.engine import KeySearch, load_filter, scalar_from_hex
    flt = load_filter(os.path.join(GOLD, "btc-bw-hash"))
    scalars = [scalar_from_hex(l.strip()) for l in open(os.path.join(GOLD, "btc-bw-priv")) if l.strip()]
    ks = KeySearch(flt, device=0, a33=True, a65=True)
    try:
        ks.cmd_mul(scalars)
        lines = sorted(r.line() for r in ks.found)
    finally:
        ks.close()
    g = G["make_mul_bw"]
    assert len(lines) == 1080 == g["count"] and lines[:64] == g["head"] and orc.digest(lines) == g["sha256_sorted"]


def test_default_geometry_large_scan_false_positives_match_oracle():
    """2^26 keys at the default launch geometry (B = 1024, all lanes): every bloom hit must equal the oracle's."""
    from ecloop_amd.engine import Filter, KeySearch
    words = synth_bloom_words(54321, seed=3, mode="a|(b&c)")
    start, nkeys = 0x100000000, 1 << 26
    ks = KeySearch(Filter(words), device=0, verify=True, launch_keys=1 << 25)
    try:
        ks.add_keys(start, nkeys, cap=1 << 16)
        got = sorted(r.line() for r in ks.found)
    finally:
        ks.close()
    rc, out, n, _, hashed = orc.add_range(orc.OrcFilter(bloom_words=words), start, start + nkeys, verify=False, threads=64, cap=1 << 16)
    assert rc == 0 and hashed == nkeys
    assert got == sorted(orc.found_lines(out, n)) and len(got) > 1000


def test_cu_endo_large_scan_false_positives_match_oracle():
    """BASELINE configs[2] in small: `-a cu -endo` (12 hashes per key) over 2^20 keys at the default geometry, every
    bloom hit equal to the oracle's (private keys through the endomorphism maps included)."""
    from ecloop_amd.engine import Filter, KeySearch
    words = synth_bloom_words(77777, seed=9, mode="a|(b&c)")
    start, nkeys = 0x200000000, 1 << 20
    ks = KeySearch(Filter(words), device=0, a33=True, a65=True, endo=True, verify=True)
    try:
        ks.add_keys(start, nkeys, cap=1 << 16)
        got = sorted(r.line() for r in ks.found)
    finally:
        ks.close()
    rc, out, n, _, hashed = orc.add_range(orc.OrcFilter(bloom_words=words), start, start + nkeys, a65=True, endo=True, verify=False,
                                          threads=64, cap=1 << 16)
    assert rc == 0 and hashed == nkeys
    assert got == sorted(orc.found_lines(out, n)) and len(got) > 500
