"""CPU tests of the host side: the C ABI library loads and exports every declared symbol, the engine's filter /
job / scalar logic agrees with the oracle and with the reference's golden status counters."""
import json
import os
import re

import numpy as np
import pytest

import orc
from synth import write_blf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]


@pytest.fixture(scope="module")
def built():
    from ecloop_amd.build import build_library
    return build_library()


def test_c_abi_loads_and_exports_every_declared_symbol(built):
    from ecloop_amd import capi
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "ecloop_hip.h")).read()
    declared = set(re.findall(r"\b(ecl_hip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ecl_hip_strerror(-4).decode().startswith("more hits")
    assert capi.C.sizeof(capi.Found) == 32


def test_no_gpu_means_loud_failure_not_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ecloop_amd import Device, EclError
    with pytest.raises(EclError):
        Device(0)


def test_job_plan_matches_reference_status_counters():
    from ecloop_amd.engine import job_plan
    for name, rs, re_, stride, hashed in [
        ("cfg1_list_800000_ffffff", 0x800000, 0xFFFFFF, 1, 8388608),
        ("make_add_8000_ffffff", 0x8000, 0xFFFFFF, 1, 16777216),
        ("ci_smoke_8000_ffff", 0x8000, 0xFFFF, 1, 32768),
        ("dump33_8000_87ff", 0x8000, 0x87FF, 1, 2048),
        ("dump33_overrun_9000_9801", 0x9000, 0x9801, 1, 4096),
        ("sparse_fp33_two_jobs", 0x8000, 0x208800, 1, 1 << 22),
        ("dump33_stride128", (1 << 164) + 0x12345, (1 << 164) + 0x12346, 1 << 128, 2048),
    ]:
        job, njobs, h = job_plan(rs, re_, stride)
        assert njobs * job == G[name]["status_checked"], name
        assert h == hashed, name
    job, njobs, h = job_plan(0x8000, 0xFFFFF)
    assert njobs * job * 6 == G["endo_cu_list_8000_fffff"]["status_checked"]


def test_shards_tile_the_scan_exactly():
    from ecloop_amd.engine import shard
    for total in (2048, 4096, 1 << 22, (1 << 22) + 2048, 1 << 32, 3 * 2048):
        for world in (1, 2, 3, 4, 8):
            at = 0
            for r in range(world):
                lo, cnt = shard(total, r, world)
                assert lo == at and lo % 2048 == 0
                at += cnt
            assert at == total


def test_calc_priv_and_hex_parsing_match_oracle():
    from ecloop_amd.engine import calc_priv, scalar_from_hex
    import random
    rnd = random.Random(2)
    for _ in range(50):
        start, off, offs, endo = rnd.randrange(1, orc.N), rnd.randrange(1 << 40), rnd.choice([0, 1, 64, 128, 200]), rnd.randrange(6)
        pk = orc.FE()
        orc.lib().orc_calc_priv(pk, orc.fe(start), orc.fe(1 << offs), off, endo)
        assert calc_priv(start, 1 << offs, off, endo) == orc.val(pk) % orc.N
    for s in ("dc2a04", "0x00ff zz 01", "%x" % (orc.N + 7), "", "F" * 70):
        assert scalar_from_hex(s) == orc.sn_from_hex(s)


def test_load_filter_list_and_blf(tmp_path):
    from ecloop_amd.engine import blf_load, blf_save, blf_size_words, load_filter, parse_hash_list
    f = load_filter(os.path.join(GOLD, "btc-puzzles-hash"))
    o = orc.OrcFilter(hashes=[h for h in orc.parse_hash_list(os.path.join(GOLD, "btc-puzzles-hash")) if h])
    assert f.count == 160 and (f.words == o.bloom_words()).all()
    assert f.confirm(f.hashes[3]) and not f.confirm([1, 2, 3, 4, 5])
    assert len(parse_hash_list(os.path.join(GOLD, "btc-bw-hash"))) == 1080  # comment chunk dropped (see docstring)
    assert blf_size_words(32768) == G["blf_gen_puzzles_32768"]["size_words"] == orc.lib().orc_blf_gen_size(32768)
    p = str(tmp_path / "x.blf")
    blf_save(p, f.words)
    assert (blf_load(p) == f.words).all()
    g = load_filter(p)
    assert g.hashes is None and g.confirm([1, 2, 3, 4, 5])
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.blf"), "wb").write(b"\0" * 16)
        load_filter(str(tmp_path / "bad.blf"))


def test_list_confirm_binary_search_edge_cases():
    from ecloop_amd.engine import Filter
    hs = np.unique(np.array([[1, 2, 3, 4, 0], [1, 2, 3, 4, 1], [0, 0, 0, 0, 0], [0xFFFFFFFF] * 5, [1, 2, 3, 0x04000000, 0],
                             [0, 0, 0, 0, 0x100]], dtype=np.uint32), axis=0)
    f = Filter(np.zeros(4, dtype=np.uint64), hs)
    for h in hs:
        assert f.confirm(h)
    for h in ([1, 2, 3, 4, 2], [0, 0, 0, 0, 1], [1, 2, 3, 5, 0], [0xFFFFFFFF] * 4 + [0xFFFFFFFE], [1, 2, 3, 4, 0x01000000]):
        assert not f.confirm(h)


def test_blf_gen_bytes_on_host(tmp_path):
    """host-side blf_add reproduces the reference's blf-gen file byte for byte"""
    import hashlib
    import struct
    from ecloop_amd.engine import blf_add_host, blf_size_words, parse_hash_list
    g = G["blf_gen_puzzles_32768"]
    words = np.zeros(blf_size_words(32768), dtype=np.uint64)
    blf_add_host(words, np.array(parse_hash_list(os.path.join(GOLD, "btc-puzzles-hash")), dtype=np.uint32))
    raw = struct.pack("<IIQ", 0x45434246, 1, len(words)) + words.tobytes()
    assert hashlib.sha256(raw).hexdigest() == g["sha256"]


def test_parse_range_errors():
    from ecloop_amd.engine import P, parse_range
    assert parse_range(None) == (2048, P)
    assert parse_range("8000:ffff") == (0x8000, 0xFFFF)
    for bad in ("800:ffff", "8000", "ffff:8000"):  # (an end above p cannot occur: -r values are reduced mod n first)
        with pytest.raises(ValueError):
            parse_range(bad)


def test_committed_bench_line_has_the_contract_fields():
    """the newest profiles/rNN_bench.json is a bench.py line from the GPU box: the driver's contract fields, the roofline
    and the cpu_baseline objects must all be there, and bench.py must still print the same field names."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")))
    line = json.loads(open(files[-1]).readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"].startswith("Mkeys/sec (add, addr33)") and line["unit"] == "Mkeys/s" and line["higher_is_better"] is True
    assert line["scaling"] in ("weak", "strong") and line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 2e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["cpu_baseline"]["kind"] in ("reference", "port")
    assert abs(line["value"] - line["config"]["keys_per_gpu_per_step"] * line["n_gpus"] / (line["ms_per_step"] * 1e3)) / line["value"] < 1e-3
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ('"ms_per_step"', '"higher_is_better"', '"vs_baseline"', '"roofline"', '"cpu_baseline"', '"traffic"', '"frac"', '"cores"', '"sample"'):
        assert k in src, k
    if files[-1].endswith("r01_bench.json"):
        return
    # from round 2 on the roofline names the tracked profile its per-key numbers come from, and that file exists
    prof = line["roofline"]["profile"]
    assert os.path.exists(os.path.join(ROOT, prof["file"])) and prof["matches_build"] is True
    assert line["roofline"]["peak"] == 78.64 and 0 < line["roofline"]["issue_cycles"]["frac"] <= 1.0
