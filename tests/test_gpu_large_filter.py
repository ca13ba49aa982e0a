"""GPU parity of the filter probe on LARGE filters (BASELINE configs[2]: `-a cu -endo` with a multi-GB .blf).

Filters of 2^24 words (128 MB) and more select different device code: the middle stage of the candidate queue
takes one probe instead of two (bloom.h: bloom_mid_two) and the early-out loop resumes at probe 2 (add_kernel.h:
cand_finish); filters of 2^31 words and more take the 64-bit branch of the reciprocal modulo (bloom.h: bloom_mod).
Compared with the oracle (blf_has, lib/utils.c:308-326): the complete hit set, false positives included — the filters
are synthetic with a bit density of 0.625 so that thousands of false positives pin probe order and modulus."""
import numpy as np
import pytest

import orc
from synth import splitmix64, synth_bloom_words
from test_devsrc_host import MOD_SIZES, mod_inputs
from test_gpu_add import lines_of

pytestmark = pytest.mark.gpu
NW = (1 << 25) + 3  # 256 MB, not a power of two


@pytest.fixture(scope="module")
def big_words():
    return synth_bloom_words(NW, 41, "a|(b&c)")


@pytest.mark.parametrize("nw", MOD_SIZES)
def test_bloom_mod_on_device(nw):
    from ecloop_amd import Device
    d = Device(0)
    try:
        xs = mod_inputs(nw)
        got = d.diag_bloom_mod(nw, xs)
        assert [int(v) for v in got] == [x % nw for x in xs]
    finally:
        d.close()


def test_probe_hit_set_on_256mb_filter(big_words):
    """blf_has for 2^20 random hashes: the device's hit vector equals the oracle's, bit for bit"""
    import ctypes as C
    from ecloop_amd import Device
    n = 1 << 20
    h = np.ascontiguousarray(splitmix64(n * 3, 99).view(np.uint32).reshape(n, 6)[:, :5])
    d = Device(0)
    try:
        d.set_bloom(big_words)
        got = d.diag_bloom(h)
    finally:
        d.close()
    L = orc.lib()
    L.orc_blf_has_many.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    want = np.zeros(n, np.uint8)
    L.orc_blf_has_many(big_words.ctypes.data, NW, h.ctypes.data, n, want.ctypes.data)
    assert np.array_equal(got, want) and 20 < int(want.sum()) < 1 << 10  # 0.625^20 * 2^20 = 87 expected


def test_add_addr33_large_filter_matches_oracle(big_words):
    """add, addr33, 2^24 keys at the DEFAULT geometry against the 256 MB filter: found list == oracle (all of it
    false positives: ~1400 at this density)"""
    from ecloop_amd import Device
    start, nkeys = 0x100000000, 1 << 24
    d = Device(0)
    try:
        d.set_bloom(big_words)
        recs, n = d.add_range(start, nkeys, cap=1 << 16)
        assert n == len(recs)
    finally:
        d.close()
    rc, out, cnt, _, hashed = orc.add_range(orc.OrcFilter(bloom_words=big_words), start, start + nkeys, verify=False,
                                            threads=64, cap=1 << 16)
    assert rc == 0 and hashed == nkeys and cnt > 500
    assert lines_of(recs, start) == sorted(orc.found_lines(out, cnt))


def test_add_cu_endo_large_filter_matches_oracle(big_words):
    """add -a cu -endo (12 hashes per key), 2^20 keys, default geometry, same filter"""
    from ecloop_amd import Device
    start, nkeys = 0x100000000, 1 << 20
    d = Device(0, a33=True, a65=True, endo=True)
    try:
        d.set_bloom(big_words)
        recs, n = d.add_range(start, nkeys, cap=1 << 16)
        assert n == len(recs)
    finally:
        d.close()
    rc, out, cnt, _, hashed = orc.add_range(orc.OrcFilter(bloom_words=big_words), start, start + nkeys, a65=True,
                                            endo=True, verify=False, threads=64, cap=1 << 16)
    assert rc == 0 and hashed == nkeys and cnt > 500
    assert lines_of(recs, start) == sorted(orc.found_lines(out, cnt))
