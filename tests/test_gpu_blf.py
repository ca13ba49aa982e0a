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
