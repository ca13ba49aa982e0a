"""Pins the CPU oracle (oracle/orc.c) to the golden vectors produced by the UNMODIFIED reference
(tests/golden/make_golden.py) and to the reference's known-answer flows. CPU only."""
import json
import os
import random

import numpy as np
import pytest

import orc
from synth import read_blf, synth_bloom_words, write_blf

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]
ONES = np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64)


def puzzles():
    return orc.OrcFilter(hashes=[h for h in orc.parse_hash_list(os.path.join(GOLD, "btc-puzzles-hash")) if h])


def run_case(name, flt, rs, re_, **kw):
    rc, out, n, checked, hashed = orc.add_range(flt, rs, re_, cap=1 << 17, **kw)
    assert rc == 0
    g = G[name]
    lines = orc.found_lines(out, n)
    assert n == g["count"] == g["status_found"]
    assert checked == g["status_checked"]
    assert orc.digest(lines) == g["sha256_sorted"]
    return sorted(lines), hashed


def test_field_ops_against_python_ints():
    rnd = random.Random(5)
    L = orc.lib()
    edge = [0, 1, 2, orc.P - 1, orc.P - 2, 0x1000003D1, (1 << 256) - 1 - 0x1000003D1, 1 << 255]
    vals = edge + [rnd.randrange(orc.P) for _ in range(300)]
    for a in vals[:40]:
        for b in vals[:40]:
            r = orc.FE()
            L.orc_fp_mul(r, orc.fe(a), orc.fe(b))
            assert orc.val(r) == a * b % orc.P
            L.orc_fp_sub(r, orc.fe(a % orc.P), orc.fe(b % orc.P))
            assert orc.val(r) == (a % orc.P - b % orc.P) % orc.P
    for a in vals[8:]:
        r = orc.FE()
        L.orc_fp_inv(r, orc.fe(a))
        assert orc.val(r) == pow(a, orc.P - 2, orc.P)
        L.orc_fp_sqr(r, orc.fe(a))
        assert orc.val(r) == a * a % orc.P
    arr = (orc.FE * 257)(*[orc.fe(v) for v in vals[8 : 8 + 257]])
    L.orc_fp_grpinv(arr, 257)
    for i, v in enumerate(vals[8 : 8 + 257]):
        assert orc.val(arr[i]) == pow(v, orc.P - 2, orc.P)


def test_scalar_ops_and_hex():
    rnd = random.Random(6)
    L = orc.lib()
    for _ in range(200):
        a, b = rnd.randrange(orc.N), rnd.randrange(orc.N)
        r = orc.FE()
        L.orc_sn_mul(r, orc.fe(a), orc.fe(b))
        assert orc.val(r) == a * b % orc.N
        L.orc_sn_sub(r, orc.fe(a), orc.fe(b))
        assert orc.val(r) == (a - b) % orc.N
    # QUIRK ecc.c:174-187: add reduces only on 2^256 overflow
    r = orc.FE()
    L.orc_sn_add(r, orc.fe(orc.N - 1), orc.fe(5))
    assert orc.val(r) == orc.N + 4
    L.orc_sn_add(r, orc.fe((1 << 256) - 1), orc.fe(5))
    assert orc.val(r) == ((1 << 256) + 4 - orc.N)
    assert orc.sn_from_hex("dc2a04") == 0xDC2A04
    assert orc.sn_from_hex("0x00ff zz 01") == 0xFF01  # non-hex skipped, right-to-left
    assert orc.sn_from_hex("%x" % (orc.N + 7)) == 7


def test_generator_multiples_on_curve_and_kats():
    gx, gy = orc.point_of(1)
    assert orc.hex160(orc.hash160(gx, gy, True)) == "751e76e8199196d454941c45d1b3a323f1433bd6"
    assert orc.hex160(orc.hash160(gx, gy, False)) == "91b24bf9f5288532960ac687abb035127b1d28a5"
    assert sorted(G["mul_G"]["lines"]) == sorted([
        "addr33\t751e76e8199196d454941c45d1b3a323f1433bd6\t%064x" % 1,
        "addr65\t91b24bf9f5288532960ac687abb035127b1d28a5\t%064x" % 1])
    for k in (2, 3, 0xDC2A04, orc.N - 1, 1 << 200):
        x, y = orc.point_of(k)
        assert (y * y - x * x * x - 7) % orc.P == 0
    x1, y1 = orc.point_of(orc.N - 1)
    assert x1 == gx and y1 == orc.P - gy


def test_gtable_mul_equals_double_and_add():
    """the reference's hidden `mult-verify` (bench.c:143-166), on a sample"""
    L = orc.lib()
    L.orc_gtable_init()
    rnd = random.Random(9)
    ks = list(range(2, 40)) + [0x3FFF, 0x4000, 0x4001, orc.N - 1] + [rnd.randrange(1, orc.N) for _ in range(40)]
    for k in ks:
        p = orc.Pt()
        L.orc_gtable_mul(orc.C.byref(p), orc.fe(k))
        L.orc_pt_rdc(orc.C.byref(p), orc.C.byref(p))
        assert (orc.val(p.x), orc.val(p.y)) == orc.point_of(k)


def test_bloom_roundtrip_and_blf_gen_bytes(tmp_path):
    """blf-gen over the puzzles list reproduces the reference's file byte for byte."""
    import hashlib
    import struct
    g = G["blf_gen_puzzles_32768"]
    size = orc.lib().orc_blf_gen_size(32768)
    assert size == g["size_words"]
    bits = np.zeros(size, np.uint64)
    hs = [h for h in orc.parse_hash_list(os.path.join(GOLD, "btc-puzzles-hash")) if h]
    p = bits.ctypes.data_as(orc.C.POINTER(orc.C.c_uint64))
    for h in hs:
        if not orc.lib().orc_blf_has(p, orc.C.c_uint64(size), orc.H160(*h)):
            orc.lib().orc_blf_add(p, orc.C.c_uint64(size), orc.H160(*h))
    raw = struct.pack("<IIQ", 0x45434246, 1, size) + bits.tobytes()
    assert len(raw) == g["bytes"] and raw[:16].hex() == g["header_hex"]
    assert hashlib.sha256(raw).hexdigest() == g["sha256"]
    path = str(tmp_path / "x.blf")
    write_blf(path, bits)
    assert (read_blf(path) == bits).all()
    for h in hs:
        assert orc.lib().orc_blf_has(p, orc.C.c_uint64(size), orc.H160(*h))


def test_list_parse_quirk_counts():
    """main.c:96-98: the comment line of btc-bw-hash yields one extra (garbage) entry -> banner says list (1081)."""
    ent = orc.parse_hash_list(os.path.join(GOLD, "btc-bw-hash"))
    assert len(ent) == 1081 and sum(e is None for e in ent) == 1
    assert len(orc.parse_hash_list(os.path.join(GOLD, "btc-puzzles-hash"))) == 160


def test_ci_smoke_and_cfg1_known_answers():
    lines, hashed = run_case("ci_smoke_8000_ffff", puzzles(), 0x8000, 0xFFFF)
    assert lines == G["ci_smoke_8000_ffff"]["lines"] and lines[0].endswith("c936")
    assert hashed == 32768  # QUIRK: 32767-key job is rounded up to 16 groups


@pytest.mark.timeout(600)
def test_cfg1_and_make_add():
    lines, hashed = run_case("cfg1_list_800000_ffffff", puzzles(), 0x800000, 0xFFFFFF, threads=8)
    assert lines == G["cfg1_list_800000_ffffff"]["lines"] and lines[0].endswith("dc2a04") and hashed == 8388608
    lines, hashed = run_case("make_add_8000_ffffff", puzzles(), 0x8000, 0xFFFFFF, threads=8)
    assert lines == sorted(G["make_add_8000_ffffff"]["lines"]) and len(lines) == 9
    assert hashed == 16777216  # scans 0x8000..0x1007fff: overruns the range end (QUIRK main.c:420-431)


def test_endo_list():
    lines, _ = run_case("endo_cu_list_8000_fffff", puzzles(), 0x8000, 0xFFFFF, a65=True, endo=True, threads=8)
    assert lines == sorted(G["endo_cu_list_8000_fffff"]["lines"])


@pytest.mark.parametrize("name,kw", [
    ("dump33_8000_87ff", dict()),
    ("dump65_8000_87ff", dict(a33=False, a65=True)),
])
def test_all_ones_dumps(name, kw):
    flt = orc.OrcFilter(bloom_words=ONES)
    lines, hashed = run_case(name, flt, 0x8000, 0x87FF, **kw)
    z = np.load(os.path.join(GOLD, G[name]["npz"]))
    assert hashed == 2048 and len(lines) == 2048
    # every record of the reference dump, field by field
    got = {l.split("\t")[2]: l for l in lines}
    for c, h, pk in zip(z["compressed"], z["h160"], z["pk"]):
        k = sum(int(pk[i]) << (64 * i) for i in range(4))
        assert got["%064x" % k] == "%s\t%s\t%064x" % ("addr33" if c else "addr65", orc.hex160(h), k)


def test_all_ones_dump_cu_endo():
    flt = orc.OrcFilter(bloom_words=ONES)
    lines, _ = run_case("dump_cu_endo_8000_87ff", flt, 0x8000, 0x87FF, a65=True, endo=True)
    assert lines[:64] == G["dump_cu_endo_8000_87ff"]["head"]
    assert len({l.split("\t")[2] for l in lines}) == 12288


def test_strided_dump_and_overrun():
    flt = orc.OrcFilter(bloom_words=ONES)
    a = (1 << 164) + 0x12345
    lines, hashed = run_case("dump33_stride128", flt, a, a + 1, offs=128)
    keys = sorted(int(l.split("\t")[2], 16) for l in lines)
    assert keys == [a + (i << 128) for i in range(2048)]
    lines, hashed = run_case("dump33_overrun_9000_9801", flt, 0x9000, 0x9801)
    assert hashed == 4096  # 2049-key job -> 2 groups


def test_sparse_bloom_false_positives_two_jobs():
    g = G["sparse_fp33_two_jobs"]
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    lines, hashed = run_case("sparse_fp33_two_jobs", orc.OrcFilter(bloom_words=words), 0x8000, 0x208800, threads=8)
    assert hashed == 2 * (1 << 21)
    z = np.load(os.path.join(GOLD, g["npz"]))
    assert len(z["pk"]) == len(lines)


def test_dense_bloom_cu_endo_false_positives():
    g = G["dense_fp_cu_endo"]
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    run_case("dense_fp_cu_endo", orc.OrcFilter(bloom_words=words), 0x8000, 0x87FF, a65=True, endo=True)


def _mask(line):
    return int(line.replace(" ", ""), 16)


@pytest.mark.parametrize("name,kw", [("rnd_d0_20_overscan", {}), ("rnd_d0_22_cu_endo", {"a65": True, "endo": True})])
def test_rnd_single_window_runs_of_the_reference(name, kw):
    """`rnd -d 0:N` on a range one window wide: the reference's only deterministic rnd run (is_full, main.c:643,658).
    The oracle's cmd_add workers with cmd_rnd's full-size jobs (main.c:624) over the printed bounds give the reference's
    found list and its `found / checked` summary."""
    g = G[name]
    s, e = _mask(g["mask_s"]), _mask(g["mask_e"])
    lo, hi = (int(x, 16) for x in g["args"][g["args"].index("-r") + 1].split(":"))
    assert (s, e) == (lo, hi) and g["header"].startswith("[RANDOM MODE] offs: 0 ~ bits: ")
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    rc, out, n, checked, hashed = orc.add_range(orc.OrcFilter(bloom_words=words), s, e, rnd=True, threads=8, cap=1 << 16, **kw)
    lines = sorted(orc.found_lines(out, n))
    assert rc == 0 and (n, checked) == (g["window_found"], g["window_checked"]) and n == g["count"]
    assert orc.digest(lines) == g["sha256_sorted"] and lines[:16] == g["head"]


def test_rnd_windows_of_the_reference_at_an_offset():
    """four windows the reference drew at `-d 128:21` on a 168-bit range (stopped by SIGINT): for each, the oracle over
    the PRINTED bounds with stride 2^128 reproduces the found lines the reference printed between the masks and the
    window's summary; and the bounds have the shape gen_random_range gives them (main.c:580-591)."""
    g = G["rnd_windows_d128_21"]
    lo, hi = (int(x, 16) for x in g["args"][g["args"].index("-r") + 1].split(":"))
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    flt = orc.OrcFilter(bloom_words=words)
    field = ((1 << 21) - 1) << 128
    assert g["header"] == "[RANDOM MODE] offs: 128 ~ bits: 21" and len(g["windows"]) >= 3
    for w in g["windows"]:
        s, e = _mask(w["mask_s"]), _mask(w["mask_e"])
        assert lo <= s < e <= hi and s & field == 0 and e == s | field
        rc, out, n, checked, hashed = orc.add_range(flt, s, e, offs=128, rnd=True, threads=8)
        assert rc == 0 and (n, checked, hashed) == (w["found"], w["checked"], 1 << 21)
        printed = sorted("%s: %s <- %s" % tuple(l.split("\t")) for l in orc.found_lines(out, n))  # stdout format, main.c:187-189
        assert orc.digest(printed) == w["stdout_lines_sha256"] and printed[:4] == w["stdout_lines_head"]


def test_mul_flows():
    """`make mul` = 1080 keys; seeded scalar dump through the all-ones bloom."""
    bw = orc.OrcFilter(hashes=[h for h in orc.parse_hash_list(os.path.join(GOLD, "btc-bw-hash")) if h])
    ks = [orc.sn_from_hex(l.strip()) for l in open(os.path.join(GOLD, "btc-bw-priv")) if l.strip()]
    rc, out, n = orc.mul_batch(bw, ks, a33=True, a65=True)
    lines = orc.found_lines(out, n)
    g = G["make_mul_bw"]
    assert rc == 0 and n == 1080 == g["count"] and orc.digest(lines) == g["sha256_sorted"]
    assert sorted(lines)[:64] == g["head"]
    ks = [orc.sn_from_hex(l.strip()) for l in open(os.path.join(GOLD, "mul_scalars.txt"))]
    rc, out, n = orc.mul_batch(orc.OrcFilter(bloom_words=ONES), ks, a33=True, a65=True)
    g = G["mul_dump_cu"]
    assert rc == 0 and n == g["count"] and orc.digest(orc.found_lines(out, n)) == g["sha256_sorted"]
    # the table form the GPU parity tests use at scale (orc.mul_hash160_many: same jobs, threaded, hashes in input order), pinned to the
    # same reference dump: its lines for these scalars have the dump's digest
    K = np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for k in ks], dtype=np.uint64)
    h33, h65, ok = orc.mul_hash160_many(K, True, True, threads=3)
    assert ok.all()
    lines = ["addr33\t%s\t%064x" % (orc.hex160(h), k) for h, k in zip(h33, ks)] + ["addr65\t%s\t%064x" % (orc.hex160(h), k) for h, k in zip(h65, ks)]
    assert orc.digest(lines) == g["sha256_sorted"]


def test_mul_hash160_many_on_edge_scalars_and_job_boundaries():
    """orc.mul_hash160_many against the oracle's double-and-add (orc.point_of, lib/ecc.c:821-853) + addr33/addr65: scalars 0 and n
    (no point: ok = 0, hashes zeroed, neighbours in the same 2048-scalar job unaffected), values >= n, single digits, a count that
    is not a multiple of the job size, any thread count"""
    rnd = random.Random(77)
    ks = [rnd.getrandbits(256) for _ in range(2048 + 300)]
    ks[0], ks[1], ks[2047], ks[2048], ks[-1] = 0, orc.N, (1 << 256) - 1, orc.N + 1, 1
    ks[5:9] = [1 << 14, (1 << 14) - 1, 1 << 252, orc.N - 1]
    K = np.array([[(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for k in ks], dtype=np.uint64)
    outs = [orc.mul_hash160_many(K, True, True, threads=t) for t in (1, 5)]
    for h33, h65, ok in outs:
        assert list(np.nonzero(ok == 0)[0]) == [0, 1] and not h33[:2].any() and not h65[:2].any()
        assert np.array_equal(h33, outs[0][0]) and np.array_equal(h65, outs[0][1])
    h33, h65, ok = outs[0]
    for i in list(range(2, 12)) + [2046, 2047, 2048, 2049, len(ks) - 1] + [rnd.randrange(len(ks)) for _ in range(40)]:
        x, y = orc.point_of(ks[i] % orc.N)
        assert list(h33[i]) == orc.hash160(x, y, True) and list(h65[i]) == orc.hash160(x, y, False), i
