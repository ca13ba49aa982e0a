"""The tracked PMC profile should belong to the kernel that is being built: bench.py prices its roofline with
profiles/r<NN>_roofline.json (VALU instructions per key, HBM bytes, clock - measured on the GPU box for ONE build).
The check is static: the instruction mix of k_add<addr33> in the assembly of the freshly built library
(tools/isa_mix.py) against the fingerprint stored in the profile, 1 % tolerance per field.  A drift is reported as a
WARNING here (and as `profile.matches_build: false` + a STALE note in bench.py's line): an ordinary kernel change must
not turn the CPU suite red until somebody has been to the GPU box - tools/collect_profiles.sh is the remedy.  With
ECL_REQUIRE_FRESH_PROFILES=1 (collection / release runs) the drift is a failure, and on the GPU box
tests/test_gpu_bench.py fails when the line bench.py prints says `matches_build: false`.  What
stays a hard failure is structural: the loop nest the per-key estimate relies on, and no scratch (spill) traffic in
the per-key loops.  Needs hipcc (or the assembly a previous build kept); skipped without either."""
import glob
import json
import os
import shutil
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def newest_profile():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_roofline.json")))
    assert files, "no profiles/rNN_roofline.json: run tools/collect_profiles.sh on the GPU box"
    return json.load(open(files[-1])), os.path.basename(files[-1])


def test_static_mix_of_the_built_kernel_matches_the_profile():
    from ecloop_amd.build import ASM, build_library
    import isa_mix
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        build_library()  # no-op when current; always leaves the assembly of the shipped code object beside the library
    if not os.path.exists(ASM):
        pytest.skip("no hipcc and no kept assembly: nothing to analyse")
    a = isa_mix.analyse(ASM)
    prof, name = newest_profile()
    fp, want = a["fingerprint"], prof["fingerprint"]
    stale = [f"{k}: built {fp.get(k)} vs {v}" for k, v in want.items() if k in fp and abs(fp[k] - v) > max(0.01 * v, 1)]
    if stale:
        msg = f"{name} was collected on another build of k_add ({'; '.join(stale)}): re-run tools/collect_profiles.sh"
        # collection / release runs (tools/collect_profiles.sh, the end-of-round check) set ECL_REQUIRE_FRESH_PROFILES=1:
        # there a profile of another build is a failure; tests/test_gpu_bench.py asserts the same on the GPU box
        # (`roofline.profile.matches_build` of the line bench.py prints)
        assert os.environ.get("ECL_REQUIRE_FRESH_PROFILES") != "1", msg
        warnings.warn(msg)
    # the loop nest the estimate relies on is the one add_kernel.h describes
    assert a["which_loop"]["valu"] > 2500 and a["table_loop"]["mad64"] >= 162 and a["prefix_loop"]["mad64"] >= 81
    # spill traffic stays out of the per-key loops: scratch instructions only in the once-per-group launch loop
    assert a["which_loop"]["scratch"] == 0 and a["table_loop"]["scratch"] == 0 and a["prefix_loop"]["scratch"] == 0
    est = a["per_key_static"]["valu"]
    pmc = prof["derived"]["valu_lane_ops_per_key"]
    if not stale:
        assert 0.97 < est / pmc < 1.10, (est, pmc)  # static upper estimate vs the PMC count of the profiled build


def test_profile_is_self_consistent():
    prof, name = newest_profile()
    d, t, c = prof["derived"], prof["traffic"], prof["corrections"]
    assert 2500 < d["valu_lane_ops_per_key"] < 4000 and 1.8 < d["clock_ghz"] < 2.5 and 3.0 < d["simd_cycles_per_valu_instr"] < 5.0
    # cycles per instruction = clock x time x SIMDs / instructions
    cyc = d["clock_ghz"] * 1e9 * prof["profiled_launch_ms"] * 1e-3 * 1024 / prof["pmc"]["SQ_INSTS_VALU"]
    assert abs(cyc - d["simd_cycles_per_valu_instr"]) < 1e-6
    # streaming calibration: FETCH_SIZE reports half of a coalesced 16-byte-per-lane read, WRITE_SIZE all of a write
    assert abs(c["fetch_stream16_reported_over_actual"] - 0.5) < 0.02 and abs(c["write_stream16_reported_over_actual"] - 1.0) < 0.02
    assert 56 < c["fetch_reported_bytes_per_random8_rd_5900MB"] < 70  # one 64-byte request per random 8-byte probe
    assert abs(t["bytes_per_key_corrected"] - (t["chain_bytes_per_key_each_way"] + t["probe_fetch_bytes_per_key_reported"] + t["write_bytes_per_key_corrected"])) < 1e-6


def test_static_mix_of_the_mul_kernel_matches_its_profile():
    """the same for k_mul_check and profiles/rNN_roofline_mul.json (bench.py prices `secondary.cfg4.api.roofline` and its `mix_ceiling` with
    it): the kernel's fingerprint (tools/isa_mix.py: analyse_mul) against the one stored with the counters; structural: the window loop is
    free of scratch traffic and holds the 918 multiply-adds of one XYZZ addition (8 M + 2 S), and the static per-scalar estimate agrees with
    the PMC count of the profiled build"""
    from ecloop_amd.build import ASM, build_library
    import isa_mix
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        build_library()
    if not os.path.exists(ASM):
        pytest.skip("no hipcc and no kept assembly: nothing to analyse")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_roofline_mul.json")))
    assert files, "no profiles/rNN_roofline_mul.json: run tools/collect_profiles.sh on the GPU box"
    prof, name = json.load(open(files[-1])), os.path.basename(files[-1])
    m = isa_mix.analyse_mul(ASM)
    fp = m["fingerprint"]
    assert fp["window_loop_scratch"] == 0 and 880 <= fp["window_loop_mad64"] <= 960 and fp["walk_back_loop_valu"] > 6000, fp
    want = (prof.get("static_mix") or {}).get("fingerprint")
    if not want:
        pytest.skip(f"{name} predates the stored class mix")
    stale = [f"{k}: built {fp.get(k)} vs {v}" for k, v in want.items() if k in fp and abs(fp[k] - v) > max(0.01 * v, 1)]
    if stale:
        msg = f"{name} was collected on another build of k_mul_check ({'; '.join(stale)}): re-run tools/collect_profiles.sh"
        assert os.environ.get("ECL_REQUIRE_FRESH_PROFILES") != "1", msg
        warnings.warn(msg)
    else:
        est, pmc = m["per_scalar_static"]["valu"], prof["derived"]["valu_lane_ops_per_scalar"]
        assert 0.90 < est / pmc < 1.05, (est, pmc)  # the static estimate leaves the inversion's share out and counts rarely taken ring blocks


def test_every_instantiation_is_fingerprinted_and_keeps_scratch_out_of_its_loops():
    """all six k_add instantiations (-a c / u / cu, each with and without -endo) and the three of k_mul_check, not only the headline
    kernel: (1) structural - no scratch (spill) instruction inside a per-key or per-table-point loop (k_add: prefix-product, table and
    `which` loops; k_mul_check: window loop, per-scalar sum loop outside the call of the out-of-line complete sum, walk-back loop);
    what remains sits in the once-per-group launch loop, or passes arguments to the rarely taken complete sum; (2) the tracked
    profiles/rNN_static_mix.json (tools/isa_mix.py --all: fingerprints, registers, spills) describes the library that is being built -
    a drift warns, and fails under ECL_REQUIRE_FRESH_PROFILES=1, like the PMC profiles above"""
    from ecloop_amd.build import ASM, build_library
    import isa_mix
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        build_library()
    if not os.path.exists(ASM):
        pytest.skip("no hipcc and no kept assembly: nothing to analyse")
    now = isa_mix.analyse_all(ASM)
    assert set(now) == set(isa_mix.ADD_KERNELS) | set(isa_mix.MUL_KERNELS)
    for label, a in now.items():
        loops = a["scratch_in_loops"]
        inner = {k: v for k, v in loops.items() if k not in ("launch", "sum_at_the_call_of_the_complete_sum")}
        assert all(v == 0 for v in inner.values()), (label, loops)
        assert a["registers"]["vgpr_count"] in (128, 168), (label, a["registers"])  # 4 / 3 waves per SIMD
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_static_mix.json")))
    assert files, "no profiles/rNN_static_mix.json: python tools/isa_mix.py --all > profiles/rNN_static_mix.json"
    kept = json.load(open(files[-1]))
    stale = []
    for label, a in now.items():
        want = kept.get(label, {}).get("fingerprint", {})
        stale += [f"{label} {k}: built {a['fingerprint'].get(k)} vs {v}" for k, v in want.items() if abs(a["fingerprint"].get(k, 0) - v) > max(0.01 * v, 1)]
        if not want:
            stale.append(f"{label}: not in {os.path.basename(files[-1])}")
    if stale:
        msg = f"{os.path.basename(files[-1])} describes another build ({'; '.join(stale[:6])}): python tools/isa_mix.py --all > profiles/rNN_static_mix.json"
        assert os.environ.get("ECL_REQUIRE_FRESH_PROFILES") != "1", msg
        warnings.warn(msg)
