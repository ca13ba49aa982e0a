"""GPU parity of the bulk filter builder: k_bloom_insert behind ecl_hip_bloom_insert = blf_add
(lib/utils.c:290-306) for many hashes at once.  The bit array must equal the reference's byte for byte."""
import ctypes as C
import hashlib
import json
import os
import struct

import numpy as np
import pytest

import orc
from synth import splitmix64

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]


def test_blf_gen_on_device_matches_reference_file():
    """engine.blf_gen (device insert) over data/btc-puzzles-hash with -n 32768 -> the reference's .blf bytes, and a
    second pass over the same list (the `make blf` update flow, Makefile:35-44) leaves them unchanged"""
    from ecloop_amd.engine import blf_gen
    g = G["blf_gen_puzzles_32768"]
    hs = np.array([h for h in orc.parse_hash_list(os.path.join(GOLD, "btc-puzzles-hash")) if h], dtype=np.uint32)
    words = blf_gen(hs, 32768)
    raw = struct.pack("<IIQ", 0x45434246, 1, len(words)) + words.astype("<u8").tobytes()
    assert len(words) == g["size_words"] and len(raw) == g["bytes"] and raw[:16].hex() == g["header_hex"]
    assert hashlib.sha256(raw).hexdigest() == g["sha256"]
    again = blf_gen(hs, 32768, existing=words)
    assert np.array_equal(again, words)


@pytest.mark.parametrize("nwords", [(1 << 20) + 7, 65539, 1])
def test_bulk_insert_random_hashes_bit_for_bit(nwords):
    """10^6 seeded hashes inserted on the device vs orc_blf_add one by one: identical words (atomic ORs commute);
    then every inserted hash is found and the filter survives a second, redundant insert"""
    from ecloop_amd import Device
    n = 1_000_000
    h = np.ascontiguousarray(splitmix64(n * 3, 4242).view(np.uint32).reshape(n, 6)[:, :5])
    d = Device(0)
    try:
        d.set_bloom(np.zeros(nwords, np.uint64))
        d.bloom_insert(h[: n // 2])
        d.bloom_insert(h[n // 2 :])  # two calls: the resident filter accumulates
        got = d.get_bloom(nwords)
        assert d.diag_bloom(h[:4096]).all()
        d.bloom_insert(h[:1000])
        assert np.array_equal(d.get_bloom(nwords), got)
    finally:
        d.close()
    want = np.zeros(nwords, np.uint64)
    L = orc.lib()
    L.orc_blf_add_many.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.orc_blf_add_many(want.ctypes.data, nwords, h.ctypes.data, n)
    assert np.array_equal(got, want)


def test_insert_count_is_the_sequential_count():
    """ecl_hip_bloom_insert_count = blf-gen's loop (utils.c:455-470): `if (blf_has) continue; blf_add; count++` in input
    order.  Inputs with exact duplicates (inside one 2^20-hash chunk and across chunks), with hashes already present,
    and with a filter so small that distinct hashes collide on every bit: count and bits must equal the oracle's."""
    from ecloop_amd import Device
    L = orc.lib()
    L.orc_blf_gen_many.restype = C.c_uint64
    L.orc_blf_gen_many.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    rng = np.random.default_rng(77)
    base = np.ascontiguousarray(splitmix64(3 * 1_500_000, 31).view(np.uint32).reshape(-1, 6)[:, :5])
    cases = []
    dup = np.concatenate([base[:600_000], base[:300_000], base[1_200_000:], base[100_000:200_000]])  # duplicates, 2 chunks
    cases.append(((1 << 20) + 7, rng.permutation(dup)))
    cases.append((64, base[:50_000]))         # 4096 bits: saturates, almost nothing is new after the first few hundred
    cases.append((1, base[:1000]))            # 64 bits
    cases.append((65539, np.repeat(base[:1000], 5, axis=0)))  # every hash five times in a row
    for nwords, h in cases:
        h = np.ascontiguousarray(h)
        want_bits = np.zeros(nwords, np.uint64)
        want_bits[: min(3, nwords)] = np.uint64(0x0123456789ABCDEF)  # a filter that is not empty to begin with
        start = want_bits.copy()
        want = L.orc_blf_gen_many(want_bits.ctypes.data, nwords, h.ctypes.data, len(h))
        d = Device(0)
        try:
            d.set_bloom(start)
            got = d.bloom_insert_count(h)
            bits = d.get_bloom(nwords)
            assert d.bloom_insert_count(h[:5000]) == 0  # everything is in now
        finally:
            d.close()
        assert got == want and np.array_equal(bits, want_bits), (nwords, got, want)


def test_cli_blf_gen_uses_the_device_above_65536_entries(tmp_path):
    """`ecloop-hip blf-gen -n 100000` fills the filter on the GPU: file and "added N new items" equal the host path's"""
    import subprocess
    from ecloop_amd.build import build_host_cli, build_library
    build_library()
    cli = build_host_cli()
    h = np.ascontiguousarray(splitmix64(3 * 90_000, 5).view(np.uint32).reshape(-1, 6)[:, :5])
    h = np.concatenate([h, h[:1234]])
    text = "".join("%08x%08x%08x%08x%08x\n" % tuple(int(v) for v in row) for row in h).encode()
    outs = {}
    for mode, extra in (("gpu", []), ("host", ["-host"])):
        f = str(tmp_path / (mode + ".blf"))
        pr = subprocess.run([cli, "blf-gen", "-n", "100000", "-o", f] + extra, input=text, stdout=subprocess.PIPE, check=True)
        outs[mode] = (open(f, "rb").read(), pr.stdout.decode())
    assert outs["gpu"][0] == outs["host"][0]
    assert "inserting on GPU 0" in outs["gpu"][1] and "inserting on GPU" not in outs["host"][1]
    added = lambda t: [l for l in t.splitlines() if l.startswith("added")][0].split(";")[0]
    assert added(outs["gpu"][1]) == added(outs["host"][1]) and "90" in added(outs["gpu"][1])  # 90 000 new, 1234 duplicates
