"""GPU parity of the `add` hot path through the C ABI: per-key dumps via the all-ones bloom (every hashed key is a
hit) against the reference's golden dumps and against the oracle, plus false-positive lists on synthetic blooms."""
import json
import os

import numpy as np
import pytest

import orc
from synth import synth_bloom_words

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]
ONES = np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64)
LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72


def privkey(start, off, offs, endo):
    """calc_priv (main.c:267-276) in python ints"""
    k = (start + (off << offs)) % orc.N
    if endo in (2, 3):
        k = k * LAM % orc.N
    if endo in (4, 5):
        k = k * LAM % orc.N * LAM % orc.N
    if endo in (1, 3, 5):
        k = (-k) % orc.N
    return k


def lines_of(recs, start, offs=0):
    return sorted("%s\t%s\t%064x" % ("addr33" if r["compressed"] else "addr65", orc.hex160(r["h160"]),
                                      privkey(start, int(r["key_offset"]), offs, int(r["endo"]))) for r in recs)


def dev_dump(start, nkeys, a33=True, a65=False, endo=False, offs=0, bloom=ONES, geometry=None, cap=None):
    from ecloop_amd import Device
    d = Device(0, a33=a33, a65=a65, endo=endo, ord_offs=offs)
    try:
        if geometry:
            d.set_geometry(*geometry)
        d.set_bloom(bloom)
        cap = cap or nkeys * (2 if a33 and a65 else 1) * (6 if endo else 1)
        recs, n = d.add_range(start, nkeys, cap=cap)
        assert n == len(recs)
        return recs
    finally:
        d.close()


@pytest.mark.parametrize("geometry", [None, (64, 256), (16, 1024), (1024, 256), (33, 512)])
def test_dump33_matches_reference_golden(geometry):
    recs = dev_dump(0x8000, 2048, geometry=geometry)
    lines = lines_of(recs, 0x8000)
    g = G["dump33_8000_87ff"]
    assert len(lines) == 2048 and orc.digest(lines) == g["sha256_sorted"]
    z = np.load(os.path.join(GOLD, g["npz"]))
    got = {int(r["key_offset"]): list(r["h160"]) for r in recs}
    for h, pk in zip(z["h160"], z["pk"]):
        assert got[int(pk[0]) - 0x8000] == list(h)


def test_dump65_and_cu_endo_match_reference_golden():
    lines = lines_of(dev_dump(0x8000, 2048, a33=False, a65=True), 0x8000)
    assert orc.digest(lines) == G["dump65_8000_87ff"]["sha256_sorted"]
    g = G["dump_cu_endo_8000_87ff"]
    lines = lines_of(dev_dump(0x8000, 2048, a33=True, a65=True, endo=True, geometry=(128, 256)), 0x8000)
    assert len(lines) == g["count"] == 24576 and lines[:64] == g["head"] and orc.digest(lines) == g["sha256_sorted"]


def test_strided_and_overrun_dumps():
    a = (1 << 164) + 0x12345
    lines = lines_of(dev_dump(a, 2048, offs=128), a, 128)
    assert orc.digest(lines) == G["dump33_stride128"]["sha256_sorted"]
    lines = lines_of(dev_dump(0x9000, 4096, geometry=(256, 256)), 0x9000)
    assert orc.digest(lines) == G["dump33_overrun_9000_9801"]["sha256_sorted"]


def test_ragged_sizes_against_oracle():
    """nkeys that are not multiples of the group, tiny ranges, one key: exactly nkeys keys are tested"""
    flt = orc.OrcFilter(bloom_words=ONES)
    rc, out, n, _, hashed = orc.add_range(flt, 0x123456, 0x123456 + 2048 * 3, threads=4, cap=1 << 14)
    ref = {int(l.split("\t")[2], 16): l for l in orc.found_lines(out, n)}
    for nkeys, geo in [(1, (8, 256)), (2, (8, 256)), (15, (8, 256)), (16, (8, 256)), (17, (8, 256)), (1000, (8, 256)),
                       (4097, (8, 256)), (6144, (64, 256)), (6143, (1024, 256)), (5000, (100, 512))]:
        lines = lines_of(dev_dump(0x123456, nkeys, geometry=geo), 0x123456)
        assert lines == sorted(ref[0x123456 + i] for i in range(nkeys)), (nkeys, geo)


def test_centre_equals_jump_point_doubling_path():
    """start chosen so that one lane's centre coincides with the jump point T*2B*G (tangent fallback)"""
    B, T = 8, 256
    start = (2 * (T - 3) - 1) * B  # centre of group m=3... start + B + m*2B == T*2B  -> m = 2
    assert (start + B) % (2 * B) == 0
    nkeys = 2 * B * T * 2  # two groups per lane so the jump is exercised
    flt = orc.OrcFilter(bloom_words=ONES)
    lines = lines_of(dev_dump(start, nkeys, geometry=(B, T)), start)
    xs = {}
    for l in lines[:: max(1, len(lines) // 300)] + lines[-64:]:
        k = int(l.split("\t")[2], 16)
        x, y = orc.point_of(k)
        assert l.split("\t")[1] == orc.hex160(orc.hash160(x, y, True)), hex(k)
    assert len(lines) == nkeys


def test_lane_centres_from_the_table_of_multiples_and_its_exceptions():
    """a non-contiguous call positions its lanes as E + (g + 1) D from the cached multiples of D = 2B G, E = C0 - D (k_init_centres_table).
    With half group 8 (D = 16 G): start = 16 g + 24 makes E equal to the table point of lane g (same x: that lane takes the complete formulas,
    here a doubling), start = 8 makes E the point at infinity (the call falls back to the ladder kernel); then a larger geometry, a second call
    on the same context (table cached), a strided context, and the automatic geometry from far-apart starts.  Every key against the oracle."""
    from ecloop_amd import Device
    B, T = 8, 256
    nkeys = 2 * B * T
    for start in (16 * 5 + 24, 16 * 255 + 24, 24, 8, 9, 1000003):
        lines = lines_of(dev_dump(start, nkeys, geometry=(B, T)), start)
        assert len(lines) == nkeys
        assert sorted(int(l.split("\t")[2], 16) for l in lines) == list(range(start, start + nkeys))
        for l in lines:
            k = int(l.split("\t")[2], 16)
            x, y = orc.point_of(k)
            assert start <= k < start + nkeys and l.split("\t")[1] == orc.hex160(orc.hash160(x, y, True)), (start, hex(k))
    flt = orc.OrcFilter(bloom_words=ONES)
    d = Device(0)
    try:
        d.set_bloom(ONES)
        d.set_geometry(64, 512)
        for start in (0x9000, 0x51234567, 0x9000, (1 << 200) + 77):  # non-contiguous: each call re-positions from the cached table
            recs, n = d.add_range(start, 2048 * 4, cap=2048 * 4)
            rc, out, cnt, _, hashed = orc.add_range(flt, start, start + 2048 * 4, verify=False, threads=4, cap=1 << 14)
            assert n == cnt == hashed and sorted(lines_of(recs, start)) == sorted(orc.found_lines(out, cnt)), hex(start)
        assert d.setup_timing()[1] == 4
    finally:
        d.close()
    d = Device(0, ord_offs=13)  # stride 2^13: D = 2B * 2^13 G
    try:
        d.set_bloom(ONES)
        d.set_geometry(16, 256)
        for start in (0x777777, (1 << 150) + 5):
            recs, n = d.add_range(start, 2048 * 2, cap=4096)
            want = []
            for g in range(2):  # the oracle dumps one 2048-key group of a strided scan per one-key range (main.c:442)
                s0 = (start + g * 2048 * (1 << 13)) % orc.N
                rc, out, cnt, _, _ = orc.add_range(flt, s0, s0 + 1, offs=13, verify=False, cap=4096)
                want += orc.found_lines(out, cnt)
            assert n == 4096 and sorted(lines_of(recs, start, 13)) == sorted(want)
    finally:
        d.close()


def test_sparse_bloom_false_positives_and_continuation():
    """same false-positive list as the reference over two reference jobs (2^22 keys);
    done as two contiguous calls so the second one continues from the walk state left in HBM."""
    from ecloop_amd import Device
    g = G["sparse_fp33_two_jobs"]
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    d = Device(0)
    try:
        d.set_geometry(256, 4096)
        d.set_bloom(words)
        r1, n1 = d.add_range(0x8000, 1 << 21, cap=4096)
        r2, n2 = d.add_range(0x8000 + (1 << 21), 1 << 21, cap=4096)
        ms, launches, keys = d.timing()
        assert launches == 2 and keys == 1 << 22 and ms > 0
    finally:
        d.close()
    lines = sorted(lines_of(r1, 0x8000) + lines_of(r2, 0x8000 + (1 << 21)))
    assert len(lines) == g["count"] and orc.digest(lines) == g["sha256_sorted"]


def test_planned_geometry_follows_the_cost_model():
    """ecl_hip_plan_geometry: the half group of a call by its size (ecloop_hip.hip: auto_half_group, fitted to profiles/r05_short_calls.txt
    and r04_short_calls.txt), lanes a multiple of 256 that cover the range with equal work, and a caller-set geometry taken as it is"""
    from ecloop_amd import Device
    d = Device(0)
    try:
        want = {11: 8, 16: 8, 21: 8, 22: 16, 23: 32, 24: 32, 25: 32, 26: 64, 27: 64, 28: 64, 29: 128, 30: 256, 31: 512, 32: 1024, 36: 1024}
        for lg, b in want.items():
            hb, lanes, nb = d.plan_geometry(1 << lg)
            assert hb == b and lanes % 256 == 0 and lanes * nb * 2 * hb >= 1 << lg, (lg, hb, lanes, nb)
            assert lanes == min(max((1 << lg) // (2 * hb), 256), 1 << 21) and (nb == 1 or lg > 32), (lg, hb, lanes, nb)
        assert d.plan_geometry((1 << 21) + 5)[0] == 8 and d.plan_geometry(3 << 22)[0] in (16, 32)
        d.set_geometry(64, 4096)
        assert d.plan_geometry(1 << 21) == (64, 4096, 4) and d.plan_geometry(1 << 32)[:2] == (64, 4096)
    finally:
        d.close()


def test_automatic_geometry_of_short_calls():
    """no geometry set: the library picks the half group by its cost model (ecloop_hip.hip: auto_half_group - 8 for the reference's 2^21-key
    job, 16 / 32 for 2^22 ... 2^25, up to 1024 for 2^32).  The reference's golden false-positive set over two 2^21-key jobs as two
    contiguous calls (the second continues the resident walk: no second set-up), then call sizes from one group to 2^22 keys against
    the oracle, contiguous continuation across a change of size included."""
    from ecloop_amd import Device
    g = G["sparse_fp33_two_jobs"]
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    d = Device(0)
    try:
        d.set_bloom(words)
        r1, n1 = d.add_range(0x8000, 1 << 21, cap=4096)
        r2, n2 = d.add_range(0x8000 + (1 << 21), 1 << 21, cap=4096)
        assert d.timing()[1] == 2 and d.setup_timing()[1] == 1  # two launches, one positioning
        lines = sorted(lines_of(r1, 0x8000) + lines_of(r2, 0x8000 + (1 << 21)))
        assert len(lines) == g["count"] and orc.digest(lines) == g["sha256_sorted"]
        flt = orc.OrcFilter(bloom_words=words)
        at = 0x123456789A
        for nkeys in (2048, 4096, 1 << 14, 3 << 15, 1 << 17, 5 << 17, 1 << 20, 1 << 22, 1 << 19):
            recs, n = d.add_range(at, nkeys, cap=1 << 16)
            rc, out, cnt, _, hashed = orc.add_range(flt, at, at + nkeys, verify=False, threads=8, cap=1 << 16)
            assert rc == 0 and hashed == nkeys and n == cnt
            assert sorted(lines_of(recs, at)) == sorted(orc.found_lines(out, cnt)), nkeys
            at += nkeys
    finally:
        d.close()


def test_dense_bloom_cu_endo_false_positives():
    g = G["dense_fp_cu_endo"]
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    recs = dev_dump(0x8000, 2048, a33=True, a65=True, endo=True, bloom=words, cap=4096)
    lines = lines_of(recs, 0x8000)
    assert len(lines) == g["count"] and orc.digest(lines) == g["sha256_sorted"]


def test_overflow_is_reported():
    from ecloop_amd import Device
    d = Device(0)
    try:
        d.set_geometry(64, 256)
        d.set_bloom(ONES)
        d.reset_timing()
        recs, n = d.add_range(0x8000, 4096, cap=100)
        assert n == 4096 and len(recs) == 100
        # the other 3996 records are still on the device (it keeps max(cap, 2^20) per call): fetched, not recomputed
        rest = d.fetch_found(100, n - 100)
        assert len(rest) == 3996 and d.timing()[1] == 1  # one search-kernel launch so far
        tail = d.fetch_found(4000, 1000)  # a request past the end is cut, one past it is empty
        assert len(tail) == 96 and len(d.fetch_found(4096, 5)) == 0
        whole, n2 = d.add_range(0x8000, 4096, cap=4096)
        assert n2 == 4096
        key = lambda a: sorted((int(r["key_offset"]), int(r["compressed"]), tuple(int(w) for w in r["h160"])) for r in a)
        assert key(np.concatenate([recs, rest])) == key(whole) and key(tail) == key(rest[-96:])
        assert len(d.fetch_found(0, 10)) == 10  # the latest call's records (it fitted: nothing was lost, they are still readable)
    finally:
        d.close()


def test_overflow_in_list_mode_fetches_the_confirmed_records():
    """device-side list confirm + a caller buffer that is too small: the confirmed records beyond it are fetched too"""
    from ecloop_amd import Device
    d = Device(0)
    try:
        d.set_geometry(64, 256)
        d.set_bloom(ONES)
        whole, n = d.add_range(0x8000, 4096, cap=4096)
        hs = np.array(sorted({tuple(int(w) for w in r["h160"]) for r in whole[::3]}), dtype=np.uint32)
        d.set_list(hs)
        d.reset_timing()
        recs, n = d.add_range(0x8000, 4096, cap=50)
        assert n == len(hs) and len(recs) == 50
        rest = d.fetch_found(50, n - 50)
        assert len(rest) == n - 50 and d.timing()[1] == 1
        got = sorted(tuple(int(w) for w in r["h160"]) for r in np.concatenate([recs, rest]))
        assert got == [tuple(int(w) for w in h) for h in hs]
    finally:
        d.close()


def test_mul_batch_dump_matches_reference_golden():
    from ecloop_amd import Device
    ks = [orc.sn_from_hex(l.strip()) for l in open(os.path.join(GOLD, "mul_scalars.txt"))]
    d = Device(0, a33=True, a65=True)
    try:
        d.set_bloom(ONES)
        recs, n = d.mul_batch(ks, cap=2048)
    finally:
        d.close()
    lines = sorted("%s\t%s\t%064x" % ("addr33" if r["compressed"] else "addr65", orc.hex160(r["h160"]), ks[int(r["key_offset"])])
                   for r in recs)
    g = G["mul_dump_cu"]
    assert n == g["count"] and orc.digest(lines) == g["sha256_sorted"]


def test_scan_through_scalar_zero_is_refused():
    """the reference cannot walk through the point at infinity either (-r must start above 0x800, main.c:687-690)"""
    from ecloop_amd import Device, EclError
    for offs, start in [(0, orc.N - 1000), (0, 0), (7, orc.N - (1000 << 7))]:
        d = Device(0, ord_offs=offs)
        try:
            d.set_bloom(ONES)
            with pytest.raises(EclError, match="private key 0"):
                d.add_range(start, 4096, cap=16)
            recs, n = d.add_range((start + (5000 << offs)) % orc.N, 2048, cap=4096)  # just past it: fine
            assert n == 2048
        finally:
            d.close()


def test_device_side_list_confirm():
    """ecl_hip_set_list = ctx_check_hash's bsearch (main.c:212-216) on the device: with the all-ones bloom every hash
    is a bloom hit, and only list members come back - for add (with endo images) and for mul."""
    from ecloop_amd import Device
    from ecloop_amd.capi import EclError
    start, n = 0x8000, 4096
    full = dev_dump(start, n, a33=True, a65=True, endo=True)
    assert len(full) == n * 12
    rng = np.random.default_rng(5)
    pick = rng.choice(len(full), 37, replace=False)
    want = np.unique(np.array([full[i]["h160"] for i in pick], dtype=np.uint32), axis=0)
    decoys = rng.integers(0, 2**32, (1000, 5), dtype=np.uint64).astype(np.uint32)
    lst = np.unique(np.concatenate([want, decoys]), axis=0)  # np.unique sorts rows lexicographically = compare_160 order
    d = Device(0, a33=True, a65=True, endo=True)
    try:
        d.set_bloom(ONES)
        d.set_list(lst)
        recs, cnt = d.add_range(start, n, cap=4096)
        assert cnt == len(recs) == len(want)
        assert sorted(tuple(r["h160"]) for r in recs) == sorted(tuple(w) for w in want)
        by_h = {tuple(r["h160"]): r for r in full}
        for r in recs:  # same offsets / tags as the unfiltered dump
            f = by_h[tuple(r["h160"])]
            assert (r["key_offset"], r["endo"], r["compressed"]) == (f["key_offset"], f["endo"], f["compressed"])
        with pytest.raises(EclError):
            d.set_list(lst[::-1])  # not sorted
        with pytest.raises(EclError):
            d.set_list(np.concatenate([lst[:3], lst[2:5]]))  # duplicate
        recs, cnt = d.add_range(start, 64, cap=4096)  # a refused list leaves the resident one in place
        assert cnt == sum(1 for r in full if r["key_offset"] < 64 and tuple(r["h160"]) in {tuple(w) for w in want})
        d.set_list(None)
        recs, cnt = d.add_range(start, 64, cap=4096)
        assert cnt == 64 * 12
    finally:
        d.close()
    # mul path: the brainwallet list and keys; all-ones bloom, so the list alone decides -> the reference's 1080 hits
    from ecloop_amd.engine import load_filter, scalar_from_hex
    flt = load_filter(os.path.join(GOLD, "btc-bw-hash"))
    ks = [scalar_from_hex(l.strip()) for l in open(os.path.join(GOLD, "btc-bw-priv")) if l.strip()]
    d = Device(0, a33=True, a65=True)
    try:
        d.set_bloom(ONES)
        recs, cnt = d.mul_batch(ks, cap=8192)
        assert cnt == 2 * len(ks)
        d.set_list(flt.hashes)
        recs, cnt = d.mul_batch(ks, cap=8192)
        assert cnt == len(recs) == G["make_mul_bw"]["count"]
        lines = sorted("%s\t%s\t%064x" % ("addr33" if r["compressed"] else "addr65", orc.hex160(r["h160"]), ks[int(r["key_offset"])])
                       for r in recs)
        assert orc.digest(lines) == G["make_mul_bw"]["sha256_sorted"]
    finally:
        d.close()


def test_mult_verify_gtable_against_the_oracle():
    """the reference's hidden `mult-verify` (lib/bench.c:143-166: fixed-base window multiplication == double-and-add for
    k = 2..16001), with the ORACLE as the yardstick of both device paths: ecl_hip_mul_batch's hash160s (window table, one inversion per
    thread) and the double-and-add kernel's must equal orc.mul_hash160_many's (ec_gtable_mul + grprdc + addr33/addr65 restated,
    lib/ecc.c:907-929, main.c:458-479) for every scalar; plus scalars that exercise every window, >= n, and 0."""
    from ecloop_amd import Device
    ks = list(range(2, 16002))
    rng = np.random.default_rng(11)
    ks += [int.from_bytes(rng.bytes(32), "big") for _ in range(4096)]            # any 256-bit value, some >= n
    ks += [orc.N - 1, orc.N + 1, orc.N + 12345, (1 << 256) - 1, 1 << 255, (1 << 14) - 1, 1 << 14, 1 << 252]
    ks += [sum(((1 << 14) - 1) << (14 * w) for w in range(0, 19, 2)) % (1 << 256)]
    K = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in ks], dtype=np.uint64)
    w33, w65, wok = orc.mul_hash160_many(K, True, True)
    assert wok.all()
    d = Device(0, a33=True, a65=True)
    try:
        d.set_bloom(ONES)
        recs, cnt = d.mul_batch(ks + [0, orc.N], cap=2 * len(ks) + 8)           # k = 0 (mod n): infinity, skipped
        assert cnt == len(recs) == 2 * len(ks)
        xs, ys, ok = d.diag_mulg([k % orc.N for k in ks])
        h33, h65 = d.diag_hash160(xs, ys)
        assert all(ok)
    finally:
        d.close()
    got33 = {int(r["key_offset"]): tuple(r["h160"]) for r in recs if r["compressed"]}
    got65 = {int(r["key_offset"]): tuple(r["h160"]) for r in recs if not r["compressed"]}
    assert len(got33) == len(got65) == len(ks)
    for i in range(len(ks)):
        assert got33[i] == tuple(w33[i]) and got65[i] == tuple(w65[i]), hex(ks[i])
    assert np.array_equal(np.array(h33, dtype=np.uint32), w33) and np.array_equal(np.array(h65, dtype=np.uint32), w65)  # the double-and-add kernel too


@pytest.mark.parametrize("n,W", [((1 << 18) + 77, 0), ((1 << 20) + 4099, 26), ((1 << 22) + (1 << 19) + 5, 0), ((1 << 23) + 12345, 26)])
def test_mul_batched_inversion_every_scalar_against_the_oracle(n, W):
    """ecl_hip_mul_batch at sizes whose pieces (2^18, 2^19, ... 2^22 scalars, then the rest) give a thread 2, 4, 8, 16 and - the
    largest size - 32 scalars (one shared inversion per thread, lib/ecc.c:695-707), over several staged pieces, on the automatic
    (22-bit) and the 26-bit table: with the all-ones filter every scalar comes back once, and every hash160 must equal the ORACLE's for
    the same scalar (orc.mul_hash160_many: cmd_mul's 2048-scalar jobs restated, main.c:458-540); scalars that are 0 (mod n) inside a
    batch are skipped without disturbing their neighbours' shared inversion."""
    import ctypes as C
    from ecloop_amd import Device, capi
    rng = np.random.default_rng(n)
    K = rng.integers(0, 1 << 63, (n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, (n, 4), dtype=np.int64).astype(np.uint64)
    zero_at = [5, n // 3, n - 1]
    for i in zero_at:
        K[i] = 0
    K[7] = np.array([(orc.N >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)], dtype=np.uint64)  # n itself: also infinity
    zero_at.append(7)
    d = Device(0)
    try:
        d.set_bloom(ONES)
        d.set_mul_window(W)
        out = np.zeros(n, dtype=capi.FOUND_DTYPE)
        cnt = C.c_uint32()
        rc = d.lib.ecl_hip_mul_batch(d.h, K.ctypes.data, n, out.ctypes.data, n, C.byref(cnt))
        assert rc == 0 and cnt.value == n - len(zero_at) and d.mul_window() == (W or 22)
    finally:
        d.close()
    h33, _, ok = orc.mul_hash160_many(K, True, False)
    recs = out[: cnt.value]
    order = np.argsort(recs["key_offset"])
    offs = recs["key_offset"][order]
    want = np.setdiff1d(np.arange(n, dtype=np.uint64), np.array(zero_at, dtype=np.uint64))
    assert np.array_equal(offs, want) and list(ok[zero_at]) == [0] * len(zero_at) and ok.sum() == n - len(zero_at) and recs["compressed"].all()
    assert np.array_equal(recs["h160"][order], h33[want.astype(np.int64)])


def _mul_all_against_the_oracle(d, K):
    """every scalar of K through ecl_hip_mul_batch (all-ones filter) == the oracle's hash160 of that scalar (orc.mul_hash160_many:
    ec_gtable_mul + one grprdc per 2048-scalar job + addr33/addr65, lib/ecc.c:907-929, main.c:458-479); -> hits.
    A context that checks both address forms is compared on both."""
    import ctypes as C
    from ecloop_amd import capi
    n = len(K)
    both = bool(getattr(d, "a65", False))
    out = np.zeros((2 if both else 1) * n, dtype=capi.FOUND_DTYPE)
    cnt = C.c_uint32()
    assert d.lib.ecl_hip_mul_batch(d.h, K.ctypes.data, n, out.ctypes.data, len(out), C.byref(cnt)) == 0
    h33, h65, ok = orc.mul_hash160_many(K, True, both)
    want = np.nonzero(ok)[0]
    recs = out[: cnt.value]
    for comp, hh in ((1, h33), (0, h65)) if both else ((1, h33),):
        part = recs[recs["compressed"] == comp]
        order = np.argsort(part["key_offset"])
        assert np.array_equal(part["key_offset"][order], want.astype(np.uint64))
        assert np.array_equal(part["h160"][order], hh[want])
    return cnt.value // (2 if both else 1)


def _digit_edge_scalars(rng, n, W):
    """random 256-bit scalars plus the digit patterns a W-bit window table can get wrong: every digit at its maximum,
    a single digit per window (1 and the maximum, the last window's narrower maximum included), 2^256 - 1, n - 1; and what the
    signed recoding turns on: a digit of exactly 2^(W-1) (the largest that stays positive), 2^(W-1) + 1 (the first that becomes
    negative with a carry), those in every window at once, a carry that runs through windows of all ones into the last row"""
    K = rng.integers(0, 1 << 63, (n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, (n, 4), dtype=np.int64).astype(np.uint64)
    special = [(1 << 256) - 1, orc.N - 1, orc.N + 1, 1, 2]
    nwin = (256 + W - 1) // W
    for w in range(nwin):
        width = min(W, 256 - W * w)
        special += [1 << (W * w), ((1 << width) - 1) << (W * w), (((1 << width) - 1) << (W * w)) | 1]
        if width == W:
            half = 1 << (W - 1)
            special += [half << (W * w), (half + 1) << (W * w), ((half + 1) << (W * w)) | (((1 << 256) - 1) >> (W * (w + 1)) << (W * (w + 1)))]
    M256 = (1 << 256) - 1
    special += [sum((1 << (W - 1)) << (W * w) for w in range(nwin)) & M256, sum(((1 << (W - 1)) + 1) << (W * w) for w in range(nwin)) & M256,
                sum(((1 << (W - 1)) - 1) << (W * w) for w in range(nwin)) & M256]
    for i, v in enumerate(special):
        K[i] = [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
    return K


@pytest.mark.parametrize("W", [8, 13, 14, 16, 18, 20, 22, 24, 26, 27, 29])
def test_mul_every_window_width_against_the_oracle(W):
    """the window width of `mul`'s table is a run-time choice (ecl_hip_set_mul_window; the reference's is the compile-time
    _GTABLE_W = 14, lib/ecc.c:876): results must not depend on it.  Widths that divide 256 and widths that leave a
    narrower last window, the table built by k_gtable_rows in one launch (small widths) and row group by row group."""
    from ecloop_amd import Device
    n = 1 << 16
    K = _digit_edge_scalars(np.random.default_rng(W), n, W)
    d = Device(0)
    try:
        d.set_bloom(ONES)
        d.set_mul_window(W)
        assert _mul_all_against_the_oracle(d, K) == n and d.mul_window() == W
        with pytest.raises(Exception):
            d.set_mul_window(30)
        with pytest.raises(Exception):
            d.set_mul_window(7)
    finally:
        d.close()


@pytest.mark.parametrize("W", [9, 16, 22, 26])
def test_mul_short_scalars_cut_the_window_loop(W):
    """wtab_sum_fast ends its loop at the highest window in which some lane of the WAVE has a digit (round 5: small scalars no longer take
    the complete sum one by one).  The bound comes from the scalars' bit lengths, and the signed recoding can carry one window further
    than the bits reach, so: whole waves of scalars of every bit length 1..256 (thread t of a short batch owns scalar t, a wave = 64
    consecutive ones), the values around every 2^(jW-1) (where the carry into window j starts) and 2^(jW), waves that mix one long
    scalar among short ones, consecutive keys from a puzzle range, and 0.  Every scalar against the oracle."""
    from ecloop_amd import Device
    rng = np.random.default_rng(1000 + W)
    ks = []
    for bits in range(1, 257):  # one wave per bit length: every lane below 2^bits, lane 0 exactly bits long
        wave = [int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62) | (int(rng.integers(0, 1 << 62)) << 124) | (int(rng.integers(0, 1 << 62)) << 186) |
                (int(rng.integers(0, 1 << 8)) << 248) for _ in range(64)]
        wave = [(v & ((1 << bits) - 1)) or 1 for v in wave]
        wave[0] |= 1 << (bits - 1)
        ks += wave
    nwin = (256 + W - 1) // W
    for j in range(1, nwin):  # the carry boundary of every window, a whole wave each (so that the bound is the lanes' own)
        for base in ((1 << (j * W - 1)), (1 << (j * W))):
            ks += [max(1, base + d) for d in range(-32, 32)]
    for j in range(2, nwin):  # one long scalar in a wave of short ones, in lane 0, 17 and 63
        for lane in (0, 17, 63):
            wave = [int(rng.integers(1, 1 << 40)) for _ in range(64)]
            wave[lane] = (1 << (j * W + 3)) | int(rng.integers(0, 1 << 60))
            ks += wave
    ks += [(1 << 65) + i for i in range(256)] + [0] * 64 + [1] * 64
    K = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in ks], dtype=np.uint64)
    d = Device(0)
    try:
        d.set_bloom(ONES)
        d.set_mul_window(W)
        assert _mul_all_against_the_oracle(d, K) == len(ks) - 64  # the 64 zeros have no point
        # ... and as a long batch, where a thread owns several scalars (scalar i = r * threads + t)
        big = np.concatenate([K] * ((1 << 18) // len(K) + 1))
        assert _mul_all_against_the_oracle(d, big) == len(big) - 64 * ((1 << 18) // len(K) + 1)
    finally:
        d.close()


def test_mul_moves_to_the_long_table_after_2_pow_30_scalars_with_identical_results():
    """automatic width: 22 bits until the context has seen 2^30 scalars, 26 bits from then on (the call that crosses the
    line already runs on the long table); two contexts on the device share each table; same hits before and after"""
    import ctypes as C
    from ecloop_amd import Device, capi
    n = 1 << 22
    K = _digit_edge_scalars(np.random.default_rng(29), n, 22)
    r = np.random.default_rng(7).integers(0, 1 << 63, (2, 1 << 10), dtype=np.int64).astype(np.uint64)
    flt = (r[0] | r[1]) << np.uint64(1) | np.uint64(1)  # density ~0.75: 0.75^20 of the hashes pass, ~10^4 hits per call
    d, d2 = Device(0), Device(0)
    try:
        hits = []
        for dev in (d, d2):
            dev.set_bloom(flt)
        out = np.zeros(1 << 16, dtype=capi.FOUND_DTYPE)
        cnt = C.c_uint32()
        for call in range(259):  # 258 * 2^22 > 2^30
            dev = d if call != 64 else d2  # the second context stays on the short table
            assert dev.lib.ecl_hip_mul_batch(dev.h, K.ctypes.data, n, out.ctypes.data, len(out), C.byref(cnt)) == 0
            assert dev.mul_window() == (26 if dev is d and call >= 256 else 22)  # d's 256th call (call 64 went to d2) completes 2^30
            if call in (0, 64, 254, 255, 256, 258):
                r = out[: cnt.value]
                hits.append(sorted(zip(r["key_offset"].tolist(), map(tuple, r["h160"].tolist()))))
        assert len(hits[0]) > 100 and all(h == hits[0] for h in hits)
    finally:
        d.close()
        d2.close()


@pytest.mark.parametrize("n", [(1 << 19) - 1, (1 << 20) - 3, (1 << 20) - 1, (1 << 21) - 1])
def test_mul_batch_sizes_just_below_a_power_of_two(n):
    """ecl_hip_mul_batch with n one to three below 2^19 / 2^20 / 2^21 on a FRESH context: the scalars-per-thread count R
    (3, 7, 7, 15) does not divide n, so R * ceil(n / R) exceeds the power-of-two capacity the parking buffer used to be
    sized for and the last plane landed outside it (round-2 advisor finding).  Every scalar must come back once with
    the oracle's hash."""
    from ecloop_amd import Device
    rng = np.random.default_rng(n)
    K = rng.integers(1, 1 << 62, (n, 4), dtype=np.int64).astype(np.uint64)
    d = Device(0)
    try:
        d.set_bloom(ONES)
        assert _mul_all_against_the_oracle(d, K) == n
    finally:
        d.close()


def test_mul_both_address_forms_on_the_long_table_against_the_oracle():
    """`mul -a cu` (BASELINE configs[4]'s selection) on the 26-bit table the host program moves to: 2^20 random scalars + the digit-edge
    set, addr33 AND addr65 of every scalar against the oracle."""
    from ecloop_amd import Device
    n = (1 << 20) + 333
    K = _digit_edge_scalars(np.random.default_rng(2026), n, 26)
    d = Device(0, a33=True, a65=True)
    try:
        d.set_bloom(ONES)
        d.set_mul_window(26)
        assert _mul_all_against_the_oracle(d, K) == n
    finally:
        d.close()


def test_small_pageable_and_page_locked_scalar_arrays():
    """ecl_hip_mul_batch from small pageable arrays that the host allocator recycles between calls (the pattern that met GPU memory
    faults in round 2 when such arrays were page-locked in place - the library no longer offers that, tools/repro_pin_fault.py), from a
    large pageable array (staged) and from ecl_hip_alloc_host memory (read by DMA): same records whichever way the scalars travel."""
    import ctypes as C
    from ecloop_amd import Device, capi
    rng = np.random.default_rng(3)
    d = Device(0, a33=True, a65=True)
    try:
        d.set_bloom(ONES)
        big = rng.integers(1, 1 << 62, (1 << 17, 4), dtype=np.int64).astype(np.uint64)
        out = np.zeros(2 * len(big), dtype=capi.FOUND_DTYPE)
        cnt = C.c_uint32()
        assert d.lib.ecl_hip_mul_batch(d.h, big.ctypes.data, len(big), out.ctypes.data, len(out), C.byref(cnt)) == 0
        ref = np.sort(out[: cnt.value], order=["key_offset", "compressed"])
        assert cnt.value == 2 * len(big)
        for it in range(150):
            small = rng.integers(1, 1 << 62, (rng.integers(1, 70), 4), dtype=np.int64).astype(np.uint64)
            o = np.zeros(2 * len(small), dtype=capi.FOUND_DTYPE)
            assert d.lib.ecl_hip_mul_batch(d.h, small.ctypes.data, len(small), o.ctypes.data, len(o), C.byref(cnt)) == 0 and cnt.value == 2 * len(small)
            if it % 30 == 0:
                assert d.lib.ecl_hip_mul_batch(d.h, big.ctypes.data, len(big), out.ctypes.data, len(out), C.byref(cnt)) == 0
                assert np.array_equal(np.sort(out[: cnt.value], order=["key_offset", "compressed"]), ref)
        ptr = d.lib.ecl_hip_alloc_host(big.nbytes)  # page-locked by the runtime: the DMA path
        assert ptr
        pl = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=big.shape)
        pl[:] = big
        assert d.lib.ecl_hip_mul_batch(d.h, ptr, len(big), out.ctypes.data, len(out), C.byref(cnt)) == 0
        assert np.array_equal(np.sort(out[: cnt.value], order=["key_offset", "compressed"]), ref)
        d.lib.ecl_hip_free_host(ptr)
    finally:
        d.close()


def test_mul_batch_raw_hashes_lines_on_the_device():
    """`mul -raw` (main.c:505-527): the scalar of a line is its SHA-256.  ecl_hip_mul_batch_raw hashes on the device: lines
    of every length around the padding boundaries (0, 1, 55, 56, 63, 64, 119, 120, ... several blocks), at every byte
    alignment, must give the hits that ecl_hip_mul_batch gives for hashlib's digests; a line table that points outside the
    text is refused."""
    import ctypes as C
    import hashlib
    import random
    from ecloop_amd import Device, capi
    r = random.Random(3)
    lens = list(range(0, 200)) + [r.randrange(0, 40) for _ in range(30000)] + [255, 256, 257, 1000, 1024, 4000]
    lines = [bytes(r.getrandbits(8) for _ in range(n)) for n in lens]
    want_k = [int.from_bytes(hashlib.sha256(l).digest(), "big") for l in lines]
    d = Device(0, a33=True, a65=True)
    try:
        d.set_bloom(ONES)
        got, n1 = d.mul_batch_raw(lines, cap=2 * len(lines))
        ref, n2 = d.mul_batch(want_k, cap=2 * len(lines))
        assert n1 == n2 == 2 * len(lines)
        key = lambda a: sorted(zip(a["key_offset"].tolist(), a["compressed"].tolist(), map(tuple, a["h160"].tolist())))
        assert key(got) == key(ref)
        # the empty string and "abc": public known answers through the oracle
        g, _ = d.mul_batch_raw([b"", b"abc"], cap=8)
        for rec in g:
            k = int.from_bytes(hashlib.sha256([b"", b"abc"][int(rec["key_offset"])]).digest(), "big") % orc.N
            assert list(rec["h160"]) == orc.hash160(*orc.point_of(k), bool(rec["compressed"]))
        text = np.frombuffer(b"0123456789", dtype=np.uint8)
        table = np.array([0 | (4 << 32), 8 | (3 << 32)], dtype=np.uint64)  # the second line ends one byte after the text
        out = np.zeros(8, dtype=capi.FOUND_DTYPE)
        cnt = C.c_uint32()
        assert d.lib.ecl_hip_mul_batch_raw(d.h, text.ctypes.data, 10, table.ctypes.data, 2, out.ctypes.data, 8, C.byref(cnt)) == -1  # ECL_E_ARG
    finally:
        d.close()


def test_mul_batch_raw_in_pieces_table_in_any_order_against_the_oracle():
    """a raw call of several pieces (the text travels with the table's pieces, the lines are hashed on a stream of their own ahead of the
    window sums): 700 001 lines of 1..40 bytes with the table in the text's order (every line on the device when its piece is hashed),
    reversed and shuffled (flag [1] of k_raw_scalars: the library repeats the call with the whole text first) - every line's two hashes
    against the ORACLE's for the SHA-256 of that line (orc.mul_hash160_many), the all-ones filter letting every line through; then the
    same through a filter that lets 3 in 1000 through, against the oracle's blf_has on those hashes"""
    import ctypes as C
    import hashlib
    from ecloop_amd import Device, capi
    rng = np.random.default_rng(2026)
    n = 700_001
    ln = rng.integers(1, 41, n).astype(np.uint64)
    starts = np.concatenate([[0], np.cumsum(ln + 1)[:-1]]).astype(np.uint64)
    tot = int(ln.sum()) + n
    text = rng.integers(32, 127, tot, dtype=np.uint8)
    text[(starts + ln).astype(np.int64)] = 10
    blob = text.tobytes()
    ks = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        dg = hashlib.sha256(blob[int(starts[i]): int(starts[i] + ln[i])]).digest()
        ks[i] = np.frombuffer(dg, dtype=">u8")[::-1]
    h33, h65, ok = orc.mul_hash160_many(ks, a33=True, a65=True)
    assert ok.all()
    in_order = starts | (ln << np.uint64(32))
    words = synth_bloom_words(4099, 5, "a|b")  # bit density 0.75: 20 probes let 3 in 1000 through
    flt = orc.OrcFilter(bloom_words=words)
    passes = sorted((i, c) for c, hh in ((1, h33), (0, h65)) for i, row in enumerate(hh.tolist()) if flt.check(row))
    d = Device(0, a33=True, a65=True)
    try:
        out = np.zeros(2 * n, dtype=capi.FOUND_DTYPE)
        cnt = C.c_uint32()
        for name, perm in (("text order", np.arange(n)), ("reversed", np.arange(n)[::-1].copy()), ("shuffled", rng.permutation(n))):
            table = np.ascontiguousarray(in_order[perm])
            d.set_bloom(ONES)
            assert d.lib.ecl_hip_mul_batch_raw(d.h, text.ctypes.data, tot, table.ctypes.data, n, out.ctypes.data, 2 * n, C.byref(cnt)) == 0, name
            assert cnt.value == 2 * n, name
            got = out[: 2 * n]
            line = perm[got["key_offset"].astype(np.int64)]
            comp = got["compressed"].astype(bool)
            assert np.array_equal(got["h160"][comp], h33[line[comp]]) and np.array_equal(got["h160"][~comp], h65[line[~comp]]), name
            assert comp.sum() == n and len(set(line[comp].tolist())) == n, name
            d.set_bloom(words)
            rc = d.lib.ecl_hip_mul_batch_raw(d.h, text.ctypes.data, tot, table.ctypes.data, n, out.ctypes.data, 2 * n, C.byref(cnt))
            assert rc == 0, name
            got = out[: cnt.value]
            seen = sorted((int(perm[int(r["key_offset"])]), int(r["compressed"])) for r in got)
            assert seen == passes and len(seen) > 1000, name
    finally:
        d.close()


def test_sort_list_on_the_device_equals_qsort_by_compare_160():
    """ecl_hip_sort_list: load_filter's qsort by compare_160 (main.c:112, addr.c:18-26: word by word, unsigned) + duplicate
    removal, on the device; entries that differ only in the last / only in the first word, runs of duplicates, and the
    filter bits of the result equal to the host's blf_add loop"""
    from ecloop_amd import Device
    rng = np.random.default_rng(160)
    n = 300_000
    a = rng.integers(0, 1 << 32, (n, 5), dtype=np.uint64).astype(np.uint32)
    a[1000:2000] = a[0:1000]                    # duplicates far apart
    a[5000:5100] = a[5000]                      # a run of one value
    a[6000:7000, :4] = a[6000, :4]              # equal but for the last word
    a[8000:9000, 1:] = a[8000, 1:]              # equal but for the first word
    a[9000] = 0
    a[9001] = 0xFFFFFFFF
    d = Device(0)
    try:
        got = d.sort_list(a)
        order = np.lexsort(tuple(a[:, k] for k in (4, 3, 2, 1, 0)))
        s = a[order]
        keep = np.ones(n, dtype=bool)
        keep[1:] = (s[1:] != s[:-1]).any(axis=1)
        want = s[keep]
        assert got.shape == want.shape and np.array_equal(got, want)
        d.set_list(got)  # the ABI's own check of "sorted and unique" accepts it
        words = np.zeros(2 * len(got), dtype=np.uint64)
        d.set_bloom(words)
        d.bloom_insert(got)
        bits = d.get_bloom(len(words))
        flt = orc.OrcFilter(hashes=got)  # the oracle's load_filter: 2 words per entry, blf_add of every entry
        assert flt.f.size == len(words) and np.array_equal(np.ctypeslib.as_array(flt.f.bits, shape=(flt.f.size,)), bits)
    finally:
        d.close()
    d = Device(0)
    try:
        assert len(d.sort_list(a[:1])) == 1 and len(d.sort_list(np.repeat(a[:1], 1000, axis=0))) == 1
    finally:
        d.close()


@pytest.mark.parametrize("n", [2, 63, 64, 65, 255, 257, 16384, 16385, 1_000_003, 17_000_001])
def test_sort_list_sizes_around_the_sorts_tiles_and_scan_levels(n):
    """the hand-written radix sort + scan behind ecl_hip_sort_list (aux_kernels.h) at sizes around its structure: fewer elements than
    threads, one more than a whole number of 256-element tiles, the switch to 65536 threads with longer tiles (17 M), two and three scan
    levels; keys with few distinct values per word (long runs of equal digits: stability) mixed with random ones; against numpy's lexsort"""
    from ecloop_amd import Device
    rng = np.random.default_rng(n)
    a = rng.integers(0, 1 << 32, (n, 5), dtype=np.uint64).astype(np.uint32)
    a[: n // 2, 0] = rng.integers(0, 3, n // 2)          # the most significant word nearly constant in one half
    a[n // 3 : 2 * n // 3, 4] &= np.uint32(0xFF00)      # low word with two zero bytes
    if n > 1000:
        a[n // 5 : n // 5 + n // 10] = a[: n // 10]      # a tenth of the list duplicated
    d = Device(0)
    try:
        got = d.sort_list(a)
    finally:
        d.close()
    order = np.lexsort(tuple(a[:, k] for k in (4, 3, 2, 1, 0)))
    s_ = a[order]
    keep = np.ones(n, dtype=bool)
    keep[1:] = (s_[1:] != s_[:-1]).any(axis=1)
    assert np.array_equal(got, s_[keep])


def test_mul_batch_raw_first_call_of_many_fresh_contexts():
    """the first raw call of a context allocates and clears its text buffer; the clearing once ran on the legacy stream and
    could land on text already copied (lines hashed wrong, now and then).  Forty fresh contexts, a text of several MB each,
    first call compared with the scalars' own path."""
    import hashlib
    from ecloop_amd import Device
    rng = np.random.default_rng(77)
    blob = rng.integers(97, 123, 6_000_000, dtype=np.uint8).tobytes()
    lines = [blob[i: i + 60] for i in range(0, len(blob), 60)][:100_000]
    ks = [int.from_bytes(hashlib.sha256(l).digest(), "big") for l in lines[::997]]
    flt = np.zeros(64, dtype=np.uint64)
    for trial in range(40):
        d = Device(0)
        try:
            d.set_bloom(ONES)
            got, n = d.mul_batch_raw(lines, cap=len(lines))
            ref, m = d.mul_batch(ks, cap=len(ks))
            assert n == len(lines) and m == len(ks)
            by_off = {int(r["key_offset"]): tuple(r["h160"]) for r in got}
            assert [by_off[i * 997] for i in range(len(ks))] == [tuple(r["h160"]) for r in sorted(ref, key=lambda r: r["key_offset"])], trial
        finally:
            d.close()


def test_full_size_range_equals_its_parts_and_the_oracle_on_samples():
    """size-independent properties at BASELINE.json's full size (2^32 keys, addr33): the found set of ONE 2^32-key call equals
    the union of four 2^30-key calls and of 2^29-key calls on a second context with another walk geometry (a GPU's shard
    on 8 GPUs); the hit count is what the filter's density predicts; every hit is confirmed by the oracle's blf_has on the
    hash the ORACLE derives for that key (orc.mul_hash160_many)"""
    from ecloop_amd import Device
    words = synth_bloom_words(1 << 20, 99, "a")  # density 0.5: 2^32 * 2^-20 = 4096 expected hits
    A = 0x100000000
    d, e = Device(0), Device(0)
    try:
        d.set_bloom(words), e.set_bloom(words)
        whole, n = d.add_range(A, 1 << 32, cap=1 << 14)
        assert n == len(whole) and 3700 < n < 4500
        key = lambda recs, base: {(base + int(r["key_offset"]), tuple(int(v) for v in r["h160"])) for r in recs}
        want = key(whole, A)
        parts = set()
        for q in range(4):
            recs, m = d.add_range(A + (q << 30), 1 << 30, cap=1 << 13)
            parts |= key(recs, A + (q << 30))
        assert parts == want
        e.set_geometry(256, 1 << 19)
        eighths = set()
        for q in range(8):
            recs, m = e.add_range(A + (q << 29), 1 << 29, cap=1 << 13)
            eighths |= key(recs, A + (q << 29))
        assert eighths == want
        ks = sorted(k for k, _ in want)
        h33, _, ok = orc.mul_hash160_many(np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for k in ks], dtype=np.uint64))
        assert all(ok) and {(k, tuple(int(v) for v in h)) for k, h in zip(ks, h33)} == want  # every hit's hash160 is the oracle's for its key
        flt = orc.OrcFilter(bloom_words=words)
        assert all(flt.check([int(v) for v in h]) for h in h33)
    finally:
        d.close(), e.close()
