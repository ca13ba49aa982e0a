"""The drop-in claim, executed: oracle/_ref/ecloop_gpu is the REFERENCE's own host program (vladkens/ecloop main.c with
the six one-line edits of oracle/ref_binding/build_ecloop_gpu.py) linked against ecloop_amd/libecloop_hip.so.  Its
argument parsing, load_filter, job scheduler, calc_priv, CPU pk_verify_hash of every hit, found sink and status line are
the reference's; batch_add (main.c:430) and the ec_gtable_mul / grprdc / check_found_mul block (main.c:531-534) are
ecl_hip_add_range / ecl_hip_mul_batch.  Every flow below must reproduce what the unmodified reference printed when
tests/golden/make_golden.py ran it on the CPU (golden.json): found lines, status counters, rnd masks."""
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

from synth import synth_bloom_words, write_blf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]
BIN = os.path.join(ROOT, "oracle", "_ref", "ecloop_gpu")


@pytest.fixture(scope="module")
def ecloop_gpu():
    if os.path.exists("/root/reference/main.c"):  # the build container; on the GPU box the prebuilt binary travels with the tree
        subprocess.run(["python3", os.path.join(ROOT, "oracle", "ref_binding", "build_ecloop_gpu.py")], check=True, stdout=subprocess.DEVNULL)
    assert os.path.exists(BIN), "oracle/_ref/ecloop_gpu is missing: run __graft_entry__.build() where /root/reference exists"
    return BIN


def run(binary, args, tmp_path, name, stdin_path=None, quiet=True, env=None):
    out = str(tmp_path / (name + ".txt"))
    cmd = [binary] + args + (["-q", "-o", out] if quiet else ["-o", out])
    pr = subprocess.run(cmd, stdin=open(stdin_path, "rb") if stdin_path else subprocess.DEVNULL, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, **env) if env else None)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-2000:]
    status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
    lines = sorted(l.rstrip("\n") for l in open(out)) if os.path.exists(out) else []
    return lines, status, pr.stdout.decode(errors="replace")


def counts(status):
    found, checked = status.split("~")[-1].split("/")
    clean = lambda s: int("".join(c for c in s if c.isdigit()))
    return clean(found), clean(checked)


def digest(lines):
    return hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest()


def blf_for(g, tmp_path, name):
    p = str(tmp_path / (name + ".blf"))
    b = g.get("bloom")
    write_blf(p, synth_bloom_words(b["words"], b["seed"], b["mode"]) if b else np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    return p


def test_list_mode_known_answers(ecloop_gpu, tmp_path):
    """`make add` (Makefile:26), the CI smoke, configs[0] and the -a cu -endo list run: bloom on the GPU, sorted-list
    bsearch, calc_priv, pk_verify_hash and the sink on the host - the reference's own code for all of those"""
    puz = os.path.join(GOLD, "btc-puzzles-hash")
    for name, rng, extra in [("make_add_8000_ffffff", "8000:ffffff", []), ("ci_smoke_8000_ffff", "8000:ffff", []),
                             ("cfg1_list_800000_ffffff", "800000:ffffff", []), ("endo_cu_list_8000_fffff", "8000:fffff", ["-a", "cu", "-endo"])]:
        lines, status, banner = run(ecloop_gpu, ["add", "-f", puz, "-r", rng, "-t", "1"] + extra, tmp_path, name)
        g = G[name]
        assert lines == sorted(g["lines"]) and counts(status) == (g["status_found"], g["status_checked"]), name
        assert "filter: list (160)" in banner


def test_all_ones_dumps_every_hash_of_every_key(ecloop_gpu, tmp_path):
    """an all-ones .blf makes every hashed key a found line: 2048-key groups (incl. the overrun past range_e and the
    2^128 stride of `-d 128:32`), both encodings, and the twelve hashes per key of -a cu -endo (24 576 lines whose
    private keys went through the reference's calc_priv and whose hashes its CPU pk_verify_hash re-derived)"""
    ones = blf_for({}, tmp_path, "ones")
    for name in ("dump33_8000_87ff", "dump65_8000_87ff", "dump_cu_endo_8000_87ff", "dump33_overrun_9000_9801", "dump33_stride128"):
        g = G[name]
        args = list(g["args"])
        args[args.index("-f") + 1] = ones
        lines, status, _ = run(ecloop_gpu, args, tmp_path, name)
        assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"], name
        assert counts(status) == (g["status_found"], g["status_checked"]), name


def test_false_positive_sets_of_synthetic_filters(ecloop_gpu, tmp_path):
    """bloom-only mode over two 2^21-key jobs (the second call continues the walk the first left on the device) and a
    dense filter under -a cu -endo: the found sets are the false positives, decided by blf_has's probe order on the device"""
    for name in ("sparse_fp33_two_jobs", "dense_fp_cu_endo"):
        g = G[name]
        args = list(g["args"])
        args[args.index("-f") + 1] = blf_for(g, tmp_path, name)
        lines, status, _ = run(ecloop_gpu, args, tmp_path, name)
        assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"], name
        assert counts(status) == (g["status_found"], g["status_checked"]), name


def test_make_mul_and_the_mul_fixtures(ecloop_gpu, tmp_path):
    """`make mul` (Makefile:29): 1080 brain-wallet keys, -a cu; k = 1 through an all-ones filter; 256 seeded scalars"""
    g = G["make_mul_bw"]
    lines, status, _ = run(ecloop_gpu, ["mul", "-f", os.path.join(GOLD, "btc-bw-hash"), "-t", "1", "-a", "cu"], tmp_path, "mul",
                           stdin_path=os.path.join(GOLD, "btc-bw-priv"))
    assert len(lines) == 1080 and digest(lines) == g["sha256_sorted"] and counts(status) == (1080, 1080)
    ones = blf_for({}, tmp_path, "ones")
    g = G["mul_dump_cu"]
    lines, status, _ = run(ecloop_gpu, ["mul", "-f", ones, "-t", "1", "-a", "cu"], tmp_path, "muldump", stdin_path=os.path.join(GOLD, "mul_scalars.txt"))
    assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"] and counts(status) == (g["status_found"], g["status_checked"])
    for name in ("mul_long_lines_hex", "mul_long_lines_raw"):  # the reference's own reader and -raw SHA-256 feed the device
        g = G[name]
        extra = ["-raw"] if name.endswith("raw") else []
        lines, status, _ = run(ecloop_gpu, ["mul", "-f", ones, "-t", "1"] + extra, tmp_path, name, stdin_path=os.path.join(GOLD, name + ".txt"))
        assert lines == sorted(g["lines"]) and list(counts(status)) == g["status"], name


def test_rnd_single_window_runs(ecloop_gpu, tmp_path):
    """cmd_rnd (main.c:619-662) over a range exactly one window wide: deterministic masks, one summary line, exits"""
    for name in ("rnd_d0_20_overscan", "rnd_d0_22_cu_endo"):
        g = G[name]
        args = list(g["args"])
        args[args.index("-f") + 1] = blf_for(g, tmp_path, name)
        lines, _, text = run(ecloop_gpu, args, tmp_path, name)
        assert g["header"] in text and g["mask_s"] in text and g["mask_e"] in text
        m = re.search(r"^([\d,]+) / ([\d,]+) ~ [\d.]+s$", text, re.M)
        assert m and (int(m.group(1).replace(",", "")), int(m.group(2).replace(",", ""))) == (g["window_found"], g["window_checked"])
        assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"] and lines[:16] == g["head"], name


def with_threads(args, t):
    args = list(args)
    if "-t" in args:
        args[args.index("-t") + 1] = str(t)
    else:
        args += ["-t", str(t)]
    return args


def test_four_worker_threads_on_four_contexts(ecloop_gpu, tmp_path):
    """`-t 4`: the reference's scheduler hands its 2^21-key jobs to four worker threads (main.c:405-435, 445-447), each bound to its own
    context by gpu_claim, each with its own hit buffer.  ECLOOP_GPU_CONTEXTS=4 opens four contexts whatever the number of GPUs (context
    g on device g mod GPUs), so the path runs on a one-GPU box too.  Same found sets and status counters as the single-thread goldens:
    list mode over 8 jobs, the sparse filter's false positives over two jobs, the 24 576-line -a cu -endo dump, `make mul`, an rnd
    window; then a 32-job scan against its own -t 1 run."""
    env = {"ECLOOP_GPU_CONTEXTS": "4"}
    puz = os.path.join(GOLD, "btc-puzzles-hash")
    g = G["make_add_8000_ffffff"]
    lines, status, banner = run(ecloop_gpu, ["add", "-f", puz, "-r", "8000:ffffff", "-t", "4"], tmp_path, "t4_list", env=env)
    assert lines == sorted(g["lines"]) and counts(status) == (g["status_found"], g["status_checked"])
    assert "threads: 4 ~" in banner, banner  # main.c:849 prints the -t it was given; gpu_init keeps it (4 contexts)
    ones = blf_for({}, tmp_path, "ones")
    for name in ("sparse_fp33_two_jobs", "dump_cu_endo_8000_87ff", "dense_fp_cu_endo"):
        g = G[name]
        args = with_threads(g["args"], 4)
        args[args.index("-f") + 1] = blf_for(g, tmp_path, name) if g.get("bloom") else ones
        lines, status, _ = run(ecloop_gpu, args, tmp_path, "t4_" + name, env=env)
        assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"], name
        assert counts(status) == (g["status_found"], g["status_checked"]), name
    g = G["make_mul_bw"]
    lines, status, _ = run(ecloop_gpu, ["mul", "-f", os.path.join(GOLD, "btc-bw-hash"), "-t", "4", "-a", "cu"], tmp_path, "t4_mul",
                           stdin_path=os.path.join(GOLD, "btc-bw-priv"), env=env)
    assert len(lines) == 1080 and digest(lines) == g["sha256_sorted"] and counts(status) == (1080, 1080)
    g = G["rnd_d0_22_cu_endo"]
    args = with_threads(g["args"], 4)
    args[args.index("-f") + 1] = blf_for(g, tmp_path, "t4_rnd")
    lines, _, text = run(ecloop_gpu, args, tmp_path, "t4_rnd", env=env)
    assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"]
    # a long scan: 2^26 keys = 32 jobs over the four contexts, bloom-only with the sparse filter; the found set equals the -t 1 run's
    g = G["sparse_fp33_two_jobs"]
    blf = blf_for(g, tmp_path, "t4_long")
    one, s1, _ = run(ecloop_gpu, ["add", "-f", blf, "-r", "100000000:103ffffff", "-t", "1"], tmp_path, "t1_long")
    four, s4, _ = run(ecloop_gpu, ["add", "-f", blf, "-r", "100000000:103ffffff", "-t", "4"], tmp_path, "t4_long", env=env)
    assert one == four and len(one) > 0 and counts(s1) == counts(s4) == (len(one), 1 << 26)


def test_a_job_with_more_hits_than_the_hit_buffer(ecloop_gpu, tmp_path):
    """all-ones filter, one 2^17-key job: 131 072 hits against the binding's 2^16-record buffer - the call reports ECL_E_OVERFLOW and the
    other half is fetched from the device (ecl_hip_fetch_found), the kernel is not run again; every line equals the oracle's"""
    import orc
    ones = blf_for({}, tmp_path, "ones")
    lines, status, _ = run(ecloop_gpu, ["add", "-f", ones, "-r", "8000:27fff", "-t", "1"], tmp_path, "overflow")
    rc, out, n, checked, hashed = orc.add_range(orc.OrcFilter(bloom_words=np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64)), 0x8000, 0x27fff,
                                                verify=False, threads=4, cap=1 << 18)
    assert rc == 0 and n == hashed == 1 << 17
    assert lines == sorted(orc.found_lines(out, n)) and counts(status) == (1 << 17, checked)
