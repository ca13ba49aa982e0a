"""GPU parity of the device primitives (through the C ABI's diagnostic entry points) against the CPU oracle."""
import random

import numpy as np
import pytest

import orc
from synth import synth_bloom_words

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from ecloop_amd import Device
    d = Device(0, a33=True, a65=True)
    yield d
    d.close()


def test_field_ops(dev):
    rnd = random.Random(11)
    P = orc.P
    edge = [0, 1, 2, P - 1, P - 2, 0x1000003D1, 1 << 255, P - 0x1000003D1, (1 << 32) - 1, (1 << 224) - 1,
            (1 << 256) - 1 - 0x1000003D1]
    edge = [e % P for e in edge]
    a = edge * len(edge) + [rnd.randrange(P) for _ in range(4000)]
    b = [y for y in edge for _ in edge] + [rnd.randrange(P) for _ in range(4000)]
    assert dev.diag_fe(0, a, b) == [x * y % P for x, y in zip(a, b)]
    assert dev.diag_fe(1, a) == [x * x % P for x in a]
    assert dev.diag_fe(3, a, b) == [(x - y) % P for x, y in zip(a, b)]
    assert dev.diag_fe(4, a, b) == [(x + y) % P for x, y in zip(a, b)]
    assert dev.diag_fe(5, a) == [(-x) % P for x in a]
    # chained operations: a result feeding the next multiplication directly (compiler-bug regression, see fe256.h)
    assert dev.diag_fe(6, a) == [pow(x, 4, P) for x in a]
    assert dev.diag_fe(7, a, b) == [x * y % P * y % P for x, y in zip(a, b)]
    assert dev.diag_fe(8, a) == [pow(x, 3, P) for x in a]
    nz = [x for x in a if x][:1500]
    assert dev.diag_fe(2, nz) == [pow(x, P - 2, P) for x in nz]


def test_inverse_by_division_steps_and_by_the_addition_chain(dev):
    """fe_inv_divsteps (what fe_inv is since round 4: 600 division steps) and fe_inv_fermat (lib/ecc.c:463-520) on the device against
    pow(a, p - 2, p): edge values (0 included: both give 0), inputs that make whole rounds pure halvings or pure swaps, 20 000
    random values; unnormalised input of magnitude 7."""
    rnd = random.Random(13)
    P = orc.P
    edge = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, 0x1000003D1, P - 0x1000003D1, 1 << 255, (1 << 255) - 1, 1 << 30, 1 << 60, 1 << 240, (1 << 30) - 1,
            (1 << 240) + 1, ((1 << 256) - 1) % P, (1 << 232) - 1, P - (1 << 29), 0x3FFFFFFF << 30, 1 << 29, 1 << 58, 3 << 90, (1 << 200) - (1 << 100)]
    edge += [pow(2, -k, P) for k in (1, 30, 31, 60, 300, 590, 600)] + [pow(3, k, P) for k in (100, 200, 255)]
    a = edge + [rnd.randrange(P) for _ in range(20000)]
    want = [pow(x, P - 2, P) for x in a]
    assert dev.diag_fe(9, a) == want
    assert dev.diag_fe(10, a[:4000]) == want[:4000]
    b = [rnd.randrange(P) for _ in a[:4000]]
    assert dev.diag_fe(11, a[:4000], b) == [pow((4 * y - 2 * x) % P, P - 2, P) for x, y in zip(a, b)]


def test_scalar_mul_and_hash160(dev):
    rnd = random.Random(12)
    ks = [1, 2, 3, 0xDC2A04, orc.N - 1, (orc.N + 1) // 2, 1 << 255, 0, orc.N] + [rnd.randrange(1, orc.N) for _ in range(120)]
    xs, ys, ok = dev.diag_mulg(ks)
    for k, x, y, o in zip(ks, xs, ys, ok):
        if k % orc.N == 0:
            assert o == 0
        else:
            assert o == 1 and (x, y) == orc.point_of(k), hex(k)
    good = [(x, y) for k, x, y in zip(ks, xs, ys) if k % orc.N]
    h33, h65 = dev.diag_hash160([g[0] for g in good], [g[1] for g in good])
    for (x, y), a, b in zip(good, h33, h65):
        assert list(a) == orc.hash160(x, y, True)
        assert list(b) == orc.hash160(x, y, False)
    # public KATs for G
    h33, h65 = dev.diag_hash160([xs[0]], [ys[0]])
    assert orc.hex160(h33[0]) == "751e76e8199196d454941c45d1b3a323f1433bd6"
    assert orc.hex160(h65[0]) == "91b24bf9f5288532960ac687abb035127b1d28a5"


@pytest.mark.parametrize("nwords,mode", [(12345, "a|(b&c)"), (1, "ones"), (2, "a|b"), (4099, "a|b"), (1 << 20, "a|b")])
def test_bloom_probe(dev, nwords, mode):
    w = np.full(1, 0xFFFFFFFFFFFFFFFF, np.uint64) if mode == "ones" else synth_bloom_words(nwords, 5, mode)
    dev.set_bloom(w)
    rnd = np.random.RandomState(7)
    hs = rnd.randint(0, 1 << 32, size=(50000, 5), dtype=np.uint64).astype(np.uint32)
    # members: add 200 hashes to a copy on the CPU, all must hit
    w2 = w.copy()
    p2 = w2.ctypes.data_as(orc.C.POINTER(orc.C.c_uint64))
    for h in hs[:200]:
        orc.lib().orc_blf_add(p2, orc.C.c_uint64(len(w2)), orc.H160(*[int(v) for v in h]))
    dev.set_bloom(w2)
    hit = dev.diag_bloom(hs)
    exp = np.array([orc.lib().orc_blf_has(p2, orc.C.c_uint64(len(w2)), orc.H160(*[int(v) for v in h])) for h in hs], dtype=np.uint8)
    assert hit[:200].all()
    assert (hit == exp).all()


def test_selftest_passes_and_leaves_state_alone(dev):
    """ecl_hip_open already ran it; run it again with a filter set and make sure the filter survives"""
    w = synth_bloom_words(4099, 5, "a|b")
    dev.set_bloom(w)
    dev.selftest()
    assert (dev.get_bloom(len(w)) == w).all()


def test_c_abi_error_codes():
    """error behaviour of the boundary (include/ecloop_hip.h): plain int codes, nothing printed, nothing exits"""
    import ctypes as C
    from ecloop_amd import capi
    lib = capi.load()
    P = C.c_void_p
    h = P()
    assert lib.ecl_hip_open(C.byref(h), 99, capi.ADDR33, 0) == -3 and not h          # ECL_E_NODEV
    assert lib.ecl_hip_open(C.byref(h), 0, 0, 0) == -1                                # no address type: ECL_E_ARG
    assert lib.ecl_hip_open(C.byref(h), 0, capi.ADDR33, 256) == -1                    # ord_offs > 255
    assert lib.ecl_hip_open(C.byref(h), 0, capi.ADDR33 | 8, 0) == -1                  # unknown flag
    assert lib.ecl_hip_open(None, 0, capi.ADDR33, 0) == -1
    assert lib.ecl_hip_open(C.byref(h), 0, capi.ADDR33, 0) == 0 and h
    try:
        start = (C.c_uint64 * 4)(0x8000, 0, 0, 0)
        out = np.zeros(16, dtype=capi.FOUND_DTYPE)
        n = C.c_uint32(7)
        assert lib.ecl_hip_add_range(h, start, 2048, out.ctypes.data, 16, C.byref(n)) == -5 and n.value == 0   # ECL_E_NOBLOOM
        assert lib.ecl_hip_mul_batch(h, start, 1, out.ctypes.data, 16, C.byref(n)) == -5
        words = np.full(4, 0xFFFFFFFFFFFFFFFF, np.uint64)
        assert lib.ecl_hip_set_bloom(h, None, 4) == -1 and lib.ecl_hip_set_bloom(h, words.ctypes.data, 0) == -1
        assert lib.ecl_hip_set_bloom(h, words.ctypes.data, 4) == 0
        assert lib.ecl_hip_add_range(h, None, 2048, out.ctypes.data, 16, C.byref(n)) == -1
        assert lib.ecl_hip_add_range(h, start, 2048, None, 16, C.byref(n)) == -1
        assert lib.ecl_hip_add_range(h, start, 2048, out.ctypes.data, 16, None) == -1
        assert lib.ecl_hip_add_range(h, start, 0, out.ctypes.data, 16, C.byref(n)) == 0 and n.value == 0      # empty range
        assert lib.ecl_hip_add_range(h, start, 2048, out.ctypes.data, 16, C.byref(n)) == -4 and n.value == 2048  # ECL_E_OVERFLOW
        assert sorted(int(k) for k in out["key_offset"]) == sorted(set(int(k) for k in out["key_offset"])) and all(out["key_offset"] < 2048)
        assert lib.ecl_hip_add_range(h, start, 2048, None, 0, C.byref(n)) == -4 and n.value == 2048            # count only
        zero = (C.c_uint64 * 4)(0, 0, 0, 0)
        assert lib.ecl_hip_add_range(h, zero, 2048, out.ctypes.data, 16, C.byref(n)) == -6                     # ECL_E_RANGE
        assert b"private key 0" in lib.ecl_hip_last_error(h)
        assert lib.ecl_hip_set_geometry(h, 1, 0) == -1 and lib.ecl_hip_set_geometry(h, 1 << 17, 0) == -1
        # key counts the walk geometry cannot represent are refused, not walked into a division by zero
        for nk in (2**64 - 1, 2**63 + 1):
            assert lib.ecl_hip_add_range(h, start, nk, out.ctypes.data, 16, C.byref(n)) == -1
            assert lib.ecl_hip_reserve(h, nk, 16) == -1
        assert lib.ecl_hip_set_geometry(h, 2, 256) == 0
        assert lib.ecl_hip_add_range(h, start, 1 << 42, out.ctypes.data, 16, C.byref(n)) == -1
        assert lib.ecl_hip_reserve(h, 1 << 42, 16) == -1
        assert lib.ecl_hip_add_range(h, start, 4096, None, 0, C.byref(n)) == -4 and n.value == 4096
        for rc in range(-7, 1):
            assert lib.ecl_hip_strerror(rc)
        # round-3 entry points
        bits = C.c_uint32(9)
        assert lib.ecl_hip_set_mul_window(h, 7) == -1 and lib.ecl_hip_set_mul_window(h, 30) == -1 and lib.ecl_hip_set_mul_window(None, 18) == -1
        assert lib.ecl_hip_get_mul_window(h, C.byref(bits)) == 0 and bits.value == 0 and lib.ecl_hip_get_mul_window(h, None) == -1
        assert lib.ecl_hip_reserve_mul(h, 0, 16) == -1 and lib.ecl_hip_reserve_mul(None, 16, 16) == -1
        text = np.frombuffer(b"abcdef", dtype=np.uint8)
        table = np.array([0 | (6 << 32)], dtype=np.uint64)
        assert lib.ecl_hip_mul_batch_raw(h, text.ctypes.data, 6, table.ctypes.data, (1 << 26) + 1, out.ctypes.data, 16, C.byref(n)) == -1
        assert lib.ecl_hip_mul_batch_raw(h, None, 6, table.ctypes.data, 1, out.ctypes.data, 16, C.byref(n)) == -1
        assert lib.ecl_hip_mul_batch_raw(h, text.ctypes.data, 6, None, 1, out.ctypes.data, 16, C.byref(n)) == -1
        assert lib.ecl_hip_mul_batch_raw(h, text.ctypes.data, 6, table.ctypes.data, 0, out.ctypes.data, 16, C.byref(n)) == 0 and n.value == 0
        assert lib.ecl_hip_mul_batch_raw(h, text.ctypes.data, 6, table.ctypes.data, 1, out.ctypes.data, 16, C.byref(n)) == 0 and n.value == 1  # all-ones filter
        kept = C.c_uint64(5)
        hs = np.zeros((4, 5), dtype=np.uint32)
        assert lib.ecl_hip_sort_list(h, hs.ctypes.data, 1 << 31, C.byref(kept)) == -1 and lib.ecl_hip_sort_list(h, None, 4, C.byref(kept)) == -1
        assert lib.ecl_hip_sort_list(h, hs.ctypes.data, 4, None) == -1
        assert lib.ecl_hip_sort_list(h, hs.ctypes.data, 0, C.byref(kept)) == 0 and kept.value == 0
        assert lib.ecl_hip_sort_list(h, hs.ctypes.data, 4, C.byref(kept)) == 0 and kept.value == 1
        # round 6: record capacities above 2^30 are refused (list mode allocates twice the count: 2^31 and more would wrap 32 bits)
        big = (1 << 30) + 1
        assert lib.ecl_hip_add_range(h, start, 2048, out.ctypes.data, big, C.byref(n)) == -1 and lib.ecl_hip_reserve(h, 2048, big) == -1
        assert lib.ecl_hip_mul_batch(h, start, 1, out.ctypes.data, big, C.byref(n)) == -1 and lib.ecl_hip_reserve_mul(h, 16, big) == -1
        # ... a reserve call leaves the records of the last call where ecl_hip_fetch_found finds them
        got = C.c_uint32()
        assert lib.ecl_hip_add_range(h, start, 4096, out.ctypes.data, 16, C.byref(n)) == -4 and n.value == 4096
        assert lib.ecl_hip_reserve(h, 4096, 16) == 0 and lib.ecl_hip_reserve_mul(h, 16, 16) == 0
        rest = np.zeros(4096, dtype=capi.FOUND_DTYPE)
        assert lib.ecl_hip_fetch_found(h, 16, rest.ctypes.data, 4080, C.byref(got)) == 0 and got.value == 4080
        assert sorted(int(k) for k in np.concatenate([out["key_offset"], rest["key_offset"][:4080]])) == list(range(4096))
        # look-ahead switches
        assert lib.ecl_hip_set_lookahead(h, 1000) == -1 and lib.ecl_hip_set_lookahead(h, 1 << 33) == -1 and lib.ecl_hip_set_lookahead(None, 0) == -1
        assert lib.ecl_hip_set_lookahead(h, 0) == 0 and lib.ecl_hip_set_lookahead(h, 1 << 26) == 0 and lib.ecl_hip_set_scan_end(h, None) == 0
        assert lib.ecl_hip_set_scan_end(None, None) == -1 and lib.ecl_hip_get_lookahead_stats(None, None, None, None, None) == -1
        assert lib.ecl_hip_get_lookahead_stats(h, None, None, None, None) == 0
    finally:
        lib.ecl_hip_close(h)
    lib.ecl_hip_close(None)  # no-op


def test_verify_path_equals_the_oracle_and_the_double_and_add_path():
    """ecl_hip_verify (window-table sum, used as pk_verify_hash for the hits of a call) against the ORACLE for every scalar
    (orc.mul_hash160_many) and against the device's double-and-add kernel + hash kernel (which therefore is pinned to the oracle on the
    same 3000 scalars): random scalars, window edge cases, values >= n, and k = 0 (mod n) flagged"""
    import orc
    from ecloop_amd import Device
    rng = np.random.default_rng(21)
    ks = [int.from_bytes(rng.bytes(32), "big") for _ in range(3000)] + [1, 2, 3, orc.N - 1, orc.N + 5, (1 << 256) - 1, (1 << 14) - 1, 1 << 14,
                                                                         1 << 252, 0xDC2A04, 0, orc.N]
    K = np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for k in ks], dtype=np.uint64)
    w33, w65, wok = orc.mul_hash160_many(K, True, True)
    d = Device(0)
    try:
        h33, h65, ok = d.verify(ks)
        xs, ys, ok2 = d.diag_mulg([k % orc.N for k in ks])
        g33, g65 = d.diag_hash160(xs, ys)
        assert list(ok) == list(ok2) == list(wok) and list(ok[-2:]) == [0, 0] and all(ok[:-2])
        assert np.array_equal(h33[:-2], w33[:-2]) and np.array_equal(h65[:-2], w65[:-2])
        assert np.array_equal(g33[:-2], w33[:-2]) and np.array_equal(g65[:-2], w65[:-2])
        one = d.verify([0xDC2A04])  # a single key, as after a scan with one hit
        assert list(one[0][0]) == orc.hash160(*orc.point_of(0xDC2A04), True)
    finally:
        d.close()
