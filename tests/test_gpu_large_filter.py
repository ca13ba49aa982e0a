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


@pytest.mark.parametrize("nw,label", [(750_000_003, "6.0 GB: configs[2]'s filter size, 32-bit reciprocal modulo"),
                                      ((1 << 31) + 5, "17.2 GB: >= 2^31 words, the 64-bit branch of bloom_mod")])
def test_add_against_multi_gigabyte_filters_at_their_real_size(nw, label):
    """configs[2] at its real filter size, on the driver's box: a synthetic filter of `nw` words (bit density 0.625,
    generated on the device, one copy kept on the host for the oracle), `add` addr33 over 2^24 keys and `-a cu -endo`
    over 2^20 keys through k_add at the default geometry; the complete found lists (all false positives, ~1400 and
    ~1000) equal the oracle's blf_has on the same words.  The 17 GB case is the only place k_add itself takes the
    64-bit reciprocal modulo (bloom.h: bloom_mod; until round 3 reached through the diag kernel only)."""
    import torch
    from ecloop_amd import Device
    g = torch.Generator(device="cuda:0").manual_seed(nw % 1000003)
    words = np.empty(nw, dtype=np.uint64)
    chunk = 1 << 27
    for at in range(0, nw, chunk):
        m = min(chunk, nw - at)
        a, b, c = (torch.randint(-(1 << 63), (1 << 63) - 1, (m,), dtype=torch.int64, device="cuda:0", generator=g) for _ in range(3))
        words[at : at + m] = (a | (b & c)).cpu().numpy().view(np.uint64)
    del a, b, c
    torch.cuda.empty_cache()
    flt = orc.OrcFilter(bloom_words=words, borrow=True)
    start = 0x100000000
    for kw, okw, nkeys in (({}, {}, 1 << 24), ({"a33": True, "a65": True, "endo": True}, {"a65": True, "endo": True}, 1 << 20)):
        d = Device(0, **kw)
        try:
            d.set_bloom(words)
            recs, n = d.add_range(start, nkeys, cap=1 << 16)
            assert n == len(recs)
        finally:
            d.close()
        rc, out, cnt, _, hashed = orc.add_range(flt, start, start + nkeys, verify=False, threads=64, cap=1 << 16, **okw)
        assert rc == 0 and hashed == nkeys and cnt > 500, (label, cnt)
        assert lines_of(recs, start) == sorted(orc.found_lines(out, cnt)), label
