"""The C host program (ecloop_amd/host/ecloop-hip): blf-gen / blf-check on the CPU, the search commands on the GPU,
against the reference's golden outputs (same flags, same output formats, same status counters)."""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

from synth import synth_bloom_words, write_blf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
G = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]


@pytest.fixture(scope="module")
def cli():
    from ecloop_amd.build import build_host_cli, build_library
    build_library()
    return build_host_cli()


def run(cli, args, stdin_path=None, out=None, env=None):
    cmd = [cli] + args + (["-q", "-o", out] if out else [])
    pr = subprocess.run(cmd, stdin=open(stdin_path, "rb") if stdin_path else subprocess.DEVNULL, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=600, env=env)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-2000:]
    status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
    lines = sorted(l.rstrip("\n") for l in open(out)) if out and os.path.exists(out) else []
    return lines, status, pr.stdout.decode(errors="replace")


def counts(status):
    found, checked = status.split("~")[-1].split("/")
    clean = lambda s: int("".join(c for c in s if c.isdigit()))
    return clean(found), clean(checked)


def digest(lines):
    return hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest()


def test_blf_gen_and_check_byte_exact(cli, tmp_path):
    """`make blf` flow (Makefile:35-44): create, then update in place; file identical to the reference's."""
    out = str(tmp_path / "p.blf")
    g = G["blf_gen_puzzles_32768"]
    for expect in ("creating bloom filter", "updating bloom filter"):
        pr = subprocess.run([cli, "blf-gen", "-n", "32768", "-o", out], stdin=open(os.path.join(GOLD, "btc-puzzles-hash"), "rb"),
                            stdout=subprocess.PIPE, check=True)
        assert expect in pr.stdout.decode()
        raw = open(out, "rb").read()
        assert len(raw) == g["bytes"] and hashlib.sha256(raw).hexdigest() == g["sha256"]
    assert "added 0 new items" in pr.stdout.decode()
    h = open(os.path.join(GOLD, "btc-puzzles-hash")).readline().strip()
    pr = subprocess.run([cli, "blf-check", "-f", out, h, "00" * 20], stdout=subprocess.PIPE, check=True)
    assert pr.stdout.decode().splitlines() == [h + " FOUND", "00" * 20 + " NOT FOUND"]


def test_odd_list_layout_read_like_the_reference_host_side():
    """the 48 hash160s the reference finds through tests/golden/odd-list.txt (one, two, three to a line, upper case, CRLF,
    blank lines, short comments, no final newline: main.c:96-110) are all entries of the list the Python mirror loads"""
    from ecloop_amd.engine import load_filter
    flt = load_filter(os.path.join(GOLD, "odd-list.txt"))
    have = {"".join("%08x" % int(w) for w in e) for e in flt.hashes}
    want = {l.split("\t")[1] for l in G["odd_list_8000_87ff"]["lines"]}
    assert len(want) == 48 and want <= have


@pytest.mark.gpu
def test_odd_list_layout_found_lines_equal_the_references(cli, tmp_path):
    lines, status, _ = run(cli, ["add", "-f", os.path.join(GOLD, "odd-list.txt"), "-r", "8000:87ff", "-t", "1"], out=str(tmp_path / "o.txt"))
    g = G["odd_list_8000_87ff"]
    assert lines == sorted(g["lines"]) and list(counts(status)) == g["status"]


def test_usage_and_version(cli):
    assert "ecloop-hip v" in subprocess.run([cli, "-v"], stdout=subprocess.PIPE).stdout.decode()
    assert "blf-gen" in subprocess.run([cli], stdout=subprocess.PIPE).stdout.decode()
    pr = subprocess.run([cli, "add", "-f", os.path.join(GOLD, "btc-puzzles-hash"), "-r", "800:ffff"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE)
    assert pr.returncode == 1 and b"invalid search range" in pr.stderr


@pytest.mark.gpu
def test_add_known_answers(cli, tmp_path):
    puz = os.path.join(GOLD, "btc-puzzles-hash")
    for name, rng, extra in [("cfg1_list_800000_ffffff", "800000:ffffff", []), ("make_add_8000_ffffff", "8000:ffffff", []),
                             ("ci_smoke_8000_ffff", "8000:ffff", []), ("endo_cu_list_8000_fffff", "8000:fffff", ["-a", "cu", "-endo"])]:
        lines, status, _ = run(cli, ["add", "-f", puz, "-r", rng, "-t", "1"] + extra, out=str(tmp_path / (name + ".txt")))
        g = G[name]
        assert lines == sorted(g["lines"])
        assert counts(status) == (g["status_found"], g["status_checked"])


@pytest.mark.gpu
@pytest.mark.parametrize("ngpu,counter", [(2, False), (3, False), (8, False), (3, True)])
def test_add_sharded_over_device_threads(cli, tmp_path, ngpu, counter):
    """`-t N` = N device threads.  A scan of at most 2^33 keys is cut into N contiguous shards, ONE device call per
    thread (the per-context `launches` of ECLOOP_HIP_STATS); longer scans - here forced with ECLOOP_HIP_SHARED_COUNTER - pull
    chunks from the shared counter like the reference's workers pull jobs (main.c:418-431).  On a one-GPU box the test hook
    ECLOOP_HIP_SHARE_GPU lets the N threads share the device: found lists and status counters must not depend on N."""
    env = dict(os.environ, ECLOOP_HIP_SHARE_GPU=str(ngpu), ECLOOP_HIP_STATS="1")
    if counter:
        env["ECLOOP_HIP_SHARED_COUNTER"] = "1"
    puz = os.path.join(GOLD, "btc-puzzles-hash")
    ones = str(tmp_path / "ones.blf")
    write_blf(ones, np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    for name, args, hashed in [("make_add_8000_ffffff", ["-f", puz, "-r", "8000:ffffff"], 1 << 24),
                               ("endo_cu_list_8000_fffff", ["-f", puz, "-r", "8000:fffff", "-a", "cu", "-endo"], 1015808),
                               ("dump33_overrun_9000_9801", ["-f", ones, "-r", "9000:9801"], 4096),
                               ("dump_cu_endo_8000_87ff", ["-f", ones, "-r", "8000:87ff", "-a", "cu", "-endo"], 2048)]:
        lines, status, stdout = run(cli, ["add", "-t", str(ngpu)] + args, out=str(tmp_path / (name + ".txt")), env=env)
        g = G[name]
        assert "gpus: %d " % ngpu in stdout
        assert len(lines) == g.get("count", len(g.get("lines", []))) and digest(lines) == g["sha256_sorted"], name
        assert counts(status) == (g["status_found"], g["status_checked"]), name
        calls = [int(m) for m in re.findall(r"^gpu \d+: (\d+) launches", stdout, re.M)]
        assert len(calls) == ngpu and sum(calls) >= 1
        if not counter:  # static shards: whole 2048-key groups, ceil(hashed / N) per thread, one call each
            per = -(-(-(-hashed // ngpu)) // 2048) * 2048
            nshards = -(-hashed // per)
            active = sorted(calls, reverse=True)[:nshards]
            assert sorted(calls, reverse=True)[nshards:] == [0] * (ngpu - nshards), (name, calls)
            # one call per shard; with the all-ones filter a shard of more than 4096 hits overflows the first buffer and is run once more
            assert all(c == 1 for c in active) if "ones" not in args[1] else all(c in (1, 2) for c in active), (name, calls)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
def test_add_handed_out_in_the_references_job_size(cli, tmp_path, threads):
    """ECLOOP_HIP_JOB_KEYS=2097152: the host program hands its scan out the way the reference's scheduler does - 2^21-key jobs from the
    shared counter (MAX_JOB_SIZE, main.c:16,418-431), one ecl_hip_add_range call each - with one worker thread and with three on three
    contexts of the GPU.  The library answers those jobs from look-ahead sweeps (few launches instead of one per job); found lines and
    status counters equal the default hand-out's, and those of the same run with the look-ahead off, for the reference's golden runs
    and for a 2^29-key scan through a synthetic filter, `-a cu -endo` included."""
    flt = str(tmp_path / "f.blf")
    write_blf(flt, synth_bloom_words(1 << 16, 21, "a|(b&c)"))
    puz = os.path.join(GOLD, "btc-puzzles-hash")
    cases = [["-f", puz, "-r", "8000:ffffff"], ["-f", flt, "-r", "100000000:11fffffff"], ["-f", flt, "-r", "100000000:101ffffff", "-a", "cu", "-endo"],
             ["-f", flt, "-r", "%x:%x" % (1 << 160, (1 << 160) + (((1 << 26) - 1) << 64)), "-d", "64:32"]]
    for k, args in enumerate(cases):
        base = run(cli, ["add", "-t", "1"] + args, out=str(tmp_path / ("base%d.txt" % k)))
        outs = []
        for e, extra in enumerate(({}, {"ECL_HIP_LOOKAHEAD_LOG2": "0"})):
            env = dict(os.environ, ECLOOP_HIP_JOB_KEYS=str(1 << 21), ECLOOP_HIP_SHARE_GPU=str(threads), ECLOOP_HIP_STATS="1", **extra)
            lines, status, stdout = run(cli, ["add", "-t", str(threads)] + args, out=str(tmp_path / ("job%d_%d.txt" % (k, e))), env=env)
            assert lines == base[0] and counts(status) == counts(base[1]), (k, extra)
            outs.append(sum(int(m) for m in re.findall(r"^gpu \d+: (\d+) launches", stdout, re.M)))
        if k == 1:  # 2^29 keys = 256 jobs: a handful of launches with the look-ahead, one per job without
            assert outs[0] <= 8 + 2 * threads and outs[1] == 256, outs
        assert len(base[0]) > 0


@pytest.mark.gpu
def test_mul_from_a_file_in_batches_equals_the_general_reader_the_pipe_and_the_oracle(cli, tmp_path):
    """`mul -a cu` over 2^21 + 777 lines of 64 hex digits through a filter that passes one hash in ~300: the batch path straight from the file
    (several batches on two contexts, AVX-512 / AVX2 decoders), the general reader on the same file, the same bytes through a pipe, CR LF
    records and `-bin` - one and the same found file, equal to the ORACLE's for every line (orc.mul_hash160_many + blf_has)."""
    import orc
    n = (1 << 21) + 777
    rng = np.random.default_rng(77)
    b = np.frombuffer(rng.bytes(n * 32), dtype=np.uint8).reshape(n, 32).copy()
    b[5] = 0  # the scalar 0: no point, skipped
    hexd = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)
    t = np.empty((n, 65), dtype=np.uint8)
    t[:, 0:64:2], t[:, 1:64:2], t[:, 64] = hexd[b >> 4], hexd[b & 15], 10
    src = str(tmp_path / "in.txt")
    t.tofile(src)
    t2 = np.empty((n, 66), dtype=np.uint8)
    t2[:, :64], t2[:, 64], t2[:, 65] = t[:, :64], 13, 10
    crlf = str(tmp_path / "in_crlf.txt")
    t2.tofile(crlf)
    binf = str(tmp_path / "in.bin")
    np.ascontiguousarray(b[:, ::-1]).tofile(binf)
    words = synth_bloom_words(65539, 41, "a|b")
    flt = str(tmp_path / "f.blf")
    write_blf(flt, words)
    K = np.ascontiguousarray(b[:, ::-1]).view("<u8").reshape(n, 4)
    h33, h65, ok = orc.mul_hash160_many(K, True, True)
    L = orc.lib()
    L.orc_blf_has_many.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    want = []
    Nn = orc.N
    for label, hh in (("addr33", h33), ("addr65", h65)):
        hit = np.zeros(n, np.uint8)
        L.orc_blf_has_many(words.ctypes.data, len(words), np.ascontiguousarray(hh).ctypes.data, n, hit.ctypes.data)
        for i in np.nonzero(hit & ok)[0]:
            want.append("%s\t%s\t%064x" % (label, orc.hex160(hh[i]), int.from_bytes(bytes(b[i]), "big") % Nn))
    want.sort()
    assert len(want) > 5000
    base = ["mul", "-f", flt, "-a", "cu"]
    runs = {"file": (base, src, {"ECLOOP_HIP_MUL_BATCH_LOG2": "19"}), "file-avx2": (base, src, {"ECLOOP_HIP_NO_AVX512": "1"}),
            "chunks": (base, src, {"ECLOOP_HIP_MUL_READ": "chunks"}), "crlf": (base, crlf, {}), "bin": (base + ["-bin"], binf, {})}
    for name, (args, path, extra) in runs.items():
        lines, status, _ = run(cli, args, stdin_path=path, out=str(tmp_path / (name + ".txt")), env=dict(os.environ, **extra))
        assert lines == want and counts(status) == (len(want), n), name
    pr = subprocess.Popen(["cat", src], stdout=subprocess.PIPE)
    out = str(tmp_path / "pipe.txt")
    got = subprocess.run([cli] + base + ["-q", "-o", out], stdin=pr.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    pr.wait()
    assert got.returncode == 0 and sorted(l.rstrip("\n") for l in open(out)) == want


@pytest.mark.gpu
def test_rnd_windows_handed_out_in_the_references_job_size(cli, tmp_path):
    """`rnd -d 128:28 -seed ..` over three windows with the scan handed out in 2^21-key jobs (ECLOOP_HIP_JOB_KEYS: the reference's scheduler;
    every window is a new scan with a new end, which the host program tells the library): same masks, found lines and per-window summaries
    as the default hand-out, with the look-ahead and without; few launches per window with it, 128 without."""
    flt = str(tmp_path / "f.blf")
    write_blf(flt, synth_bloom_words(1 << 16, 31, "a|(b&c)"))
    lo, hi = (1 << 167) + 0x1234567, (1 << 168) - 0x7654321
    args = ["rnd", "-f", flt, "-r", f"{lo:x}:{hi:x}", "-d", "128:28", "-seed", "jobs", "-t", "1"]
    outs = []
    for k, extra in enumerate(({}, {"ECLOOP_HIP_JOB_KEYS": str(1 << 21)}, {"ECLOOP_HIP_JOB_KEYS": str(1 << 21), "ECL_HIP_LOOKAHEAD_LOG2": "0"})):
        env = dict(os.environ, ECLOOP_HIP_RND_WINDOWS="3", ECLOOP_HIP_STATS="1", **extra)
        lines, status, stdout = run(cli, args, out=str(tmp_path / ("rnd%d.txt" % k)), env=env)
        masks = [l for l in stdout.splitlines() if re.fullmatch(r"[0-9a-f ]{67}", l)]
        launches = sum(int(m) for m in re.findall(r"^gpu \d+: (\d+) launches", stdout, re.M))
        outs.append((lines, counts(status), masks, launches))
    assert outs[0][:3] == outs[1][:3] == outs[2][:3] and len(outs[0][2]) == 6 and len(outs[0][0]) > 10
    assert outs[0][3] == 3 and outs[1][3] <= 12 and outs[2][3] == 3 * 128, [o[3] for o in outs]


@pytest.mark.gpu
def test_add_dumps_and_stride(cli, tmp_path):
    ones = str(tmp_path / "ones.blf")
    write_blf(ones, np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    a = (1 << 164) + 0x12345
    for name, args in [("dump33_8000_87ff", ["-r", "8000:87ff"]), ("dump65_8000_87ff", ["-r", "8000:87ff", "-a", "u"]),
                       ("dump_cu_endo_8000_87ff", ["-r", "8000:87ff", "-a", "cu", "-endo"]),
                       ("dump33_overrun_9000_9801", ["-r", "9000:9801"]),
                       ("dump33_stride128", ["-r", f"{a:x}:{a + 1:x}", "-d", "128:32"])]:
        lines, status, _ = run(cli, ["add", "-f", ones, "-t", "1"] + args, out=str(tmp_path / (name + ".txt")))
        g = G[name]
        assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"], name
        assert counts(status) == (g["status_found"], g["status_checked"]), name


@pytest.mark.gpu
def test_sparse_bloom_two_jobs(cli, tmp_path):
    g = G["sparse_fp33_two_jobs"]
    blf = str(tmp_path / "s.blf")
    write_blf(blf, synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"]))
    lines, status, _ = run(cli, ["add", "-f", blf, "-r", "8000:208800"], out=str(tmp_path / "o.txt"))
    assert digest(lines) == g["sha256_sorted"] and counts(status) == (g["status_found"], g["status_checked"])


@pytest.mark.gpu
def test_long_hash_list_prepared_on_the_device(cli, tmp_path):
    """a list of >= 2^16 entries is sorted, made unique and turned into filter bits on the GPU (ecl_hip_sort_list); the run
    must find what the host-prepared list finds (ECLOOP_HIP_LIST_ON_HOST=1), duplicates and all"""
    import random
    import orc
    r = random.Random(16)
    want = [orc.hash160(*orc.point_of(k), True) for k in (0x8123, 0x9abc, 0xfff0)]
    entries = ["%040x" % r.getrandbits(160) for _ in range(70000)] + ["".join("%08x" % w for w in h) for h in want]
    entries += entries[:500]  # duplicates
    r.shuffle(entries)
    lst = tmp_path / "big.txt"
    lst.write_text("\n".join(entries) + "\n")
    a, sa, _ = run(cli, ["add", "-f", str(lst), "-r", "8000:ffff"], out=str(tmp_path / "a.txt"))
    b, sb, _ = run(cli, ["add", "-f", str(lst), "-r", "8000:ffff"], out=str(tmp_path / "b.txt"), env=dict(os.environ, ECLOOP_HIP_LIST_ON_HOST="1"))
    assert sorted(a) == sorted(b) and len(a) == 3 and counts(sa) == counts(sb)
    assert sorted(l.split("\t")[2][-4:] for l in a) == ["8123", "9abc", "fff0"]


@pytest.mark.gpu
def test_mul_flows(cli, tmp_path):
    lines, status, _ = run(cli, ["mul", "-f", os.path.join(GOLD, "btc-bw-hash"), "-a", "cu"], stdin_path=os.path.join(GOLD, "btc-bw-priv"),
                           out=str(tmp_path / "m.txt"))
    g = G["make_mul_bw"]
    assert len(lines) == 1080 and digest(lines) == g["sha256_sorted"] and counts(status) == (1080, 1080)
    ones = str(tmp_path / "ones.blf")
    write_blf(ones, np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    lines, _, _ = run(cli, ["mul", "-f", ones, "-a", "cu"], stdin_path=os.path.join(GOLD, "mul_scalars.txt"), out=str(tmp_path / "d.txt"))
    assert digest(lines) == G["mul_dump_cu"]["sha256_sorted"]
    # -raw: scalar = SHA-256(line); "abc" is a public SHA-256 test vector
    raw = tmp_path / "raw.txt"
    raw.write_text("abc\n")
    lines, _, _ = run(cli, ["mul", "-f", ones, "-raw"], stdin_path=str(raw), out=str(tmp_path / "r.txt"))
    assert lines[0].split("\t")[2] == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"


@pytest.mark.gpu
def test_rnd_window_is_an_add_over_the_printed_bounds(cli, tmp_path):
    """a random window equals `add -r range_s:range_e -d offs:size` on the printed bounds."""
    ones = str(tmp_path / "ones.blf")
    write_blf(ones, synth_bloom_words(4099, seed=11, mode="a|b"))
    out = str(tmp_path / "rnd.txt")
    # range = one window wide, so the first window is the full range and rnd stops after it (main.c:643,658)
    lo, hi = 1 << 40, (1 << 40) + (1 << 20) - 1
    pr = subprocess.run([cli, "rnd", "-f", ones, "-r", f"{lo:x}:{hi:x}", "-d", "0:20", "-q", "-o", out], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=300)
    assert pr.returncode == 0, pr.stderr.decode()[-1000:]
    text = pr.stdout.decode()
    assert "[RANDOM MODE] offs: 0 ~ bits: 20" in text
    masks = [l.replace(" ", "") for l in text.splitlines() if re.fullmatch(r"[0-9a-f ]{67}", l)]
    assert int(masks[0], 16) == lo and int(masks[1], 16) == hi
    rnd_lines = sorted(l.rstrip("\n") for l in open(out))
    add_out = str(tmp_path / "add.txt")
    add_lines, _, _ = run(cli, ["add", "-f", ones, "-r", f"{lo:x}:{hi:x}"], out=add_out)
    # rnd scans whole 2^21-key jobs (main.c:624): the add over the same bounds is a subset; same keys inside the window
    inside = [l for l in rnd_lines if lo <= int(l.split("\t")[2], 16) <= hi + 2048]
    assert set(add_lines) <= set(rnd_lines) and len(add_lines) > 100
    assert all(int(l.split("\t")[2], 16) < lo + (1 << 21) for l in rnd_lines)


@pytest.mark.gpu
@pytest.mark.timeout(1500)
@pytest.mark.parametrize("size,mode,nwin", [(20, "a|b", 1), (32, "a", 2)])
def test_rnd_at_configs3_shape_against_the_oracle_on_the_printed_masks(cli, tmp_path, size, mode, nwin):
    """BASELINE configs[3]: `rnd -d 128:32` on a 168-bit range.  Every window the generator prints (two 64-digit masks,
    main.c:593-617) must hash exactly the keys the ORACLE's cmd_add workers hash on those bounds at stride 2^128 with
    cmd_rnd's full-size jobs (orc.add_range(..., offs=128, rnd=True): the restatement of main.c:405-454,619-662 that
    tests/test_oracle_golden.py pins to the windows the reference itself drew).  A three-quarters-dense filter (~6600 hits) for the
    2^21-key window, which the oracle covers whole (the all-ones filter of round 3 - two million found lines - spent three minutes in
    this test's own line handling); for the named config's real window size a half-dense synthetic filter (~4096 false positives per
    2^32 keys): the oracle covers the first 2^28 and the last 2^27 keys of the first printed window and the first 2^27 keys of
    the second (slices of the printed windows: the oracle runs at ~15 M keys/s on the box's host cores)."""
    import orc
    words = np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64) if mode == "ones" else synth_bloom_words(70001, seed=23, mode=mode)
    blf = str(tmp_path / "f.blf")
    write_blf(blf, words)
    lo, hi = (1 << 167) + 0x1234567, (1 << 168) - 0x7654321
    out = str(tmp_path / "rnd.txt")
    env = dict(os.environ, ECLOOP_HIP_RND_WINDOWS=str(nwin), ECLOOP_HIP_STATS="1")
    pr = subprocess.run([cli, "rnd", "-f", blf, "-r", f"{lo:x}:{hi:x}", "-d", f"128:{size}", "-seed", "graft", "-q", "-o", out],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert pr.returncode == 0, pr.stderr.decode()[-1000:]
    text = pr.stdout.decode()
    assert f"[RANDOM MODE] offs: 128 ~ bits: {size}" in text
    masks = [int(l.replace(" ", ""), 16) for l in text.splitlines() if re.fullmatch(r"[0-9a-f ]{67}", l)]
    assert len(masks) == 2 * nwin
    _, wins = rnd_windows(text)
    rnd_lines = sorted(l.rstrip("\n") for l in open(out))
    flt = orc.OrcFilter(bloom_words=words)
    threads = min(os.cpu_count() or 8, 128)
    span = max(1 << size, 1 << 21)  # keys a window hashes: cmd_rnd runs whole 2^21-key jobs (main.c:624)
    covered = 0
    for w in range(nwin):
        s, e = masks[2 * w], masks[2 * w + 1]
        field = ((1 << size) - 1) << 128
        assert lo <= s < e <= hi
        if s != lo and e != hi:  # not clipped to the range (main.c:586-589): bits offs..offs+size-1 cleared / set
            assert s & field == 0 and e & field == field and s | field == e
        mine = [l for l in rnd_lines if (int(l.split("\t")[2], 16) - s) % (1 << 128) == 0 and 0 <= (int(l.split("\t")[2], 16) - s) >> 128 < span]
        key = lambda l: (int(l.split("\t")[2], 16) - s) >> 128  # position of a found key inside its window
        if size <= 21:
            rc, o, n, checked, hashed = orc.add_range(flt, s, e, offs=128, rnd=True, threads=threads, cap=1 << 22)
            assert rc == 0 and (checked, hashed) == (span, span)
            assert wins[w][3:] == (n, checked)  # the window's `found / checked` summary line
            assert mine == sorted(orc.found_lines(o, n))
            covered += len(mine)
        else:
            # slices of the window through the oracle (its bounds are inclusive like -r; whole 2^21-key jobs): the first window's
            # first 2^28 keys and last 2^27, the second window's first 2^27 - the oracle hashes ~15 M keys/s on the box's host
            # cores (the whole 2^32-key window took 5 minutes when this test did that)
            for first, count in ([(0, 1 << 28), (span - (1 << 27), 1 << 27)] if w == 0 else [(0, 1 << 27)]):
                a = s + (first << 128)
                rc, o, n, checked, hashed = orc.add_range(flt, a, a + ((count - 1) << 128), offs=128, rnd=True, threads=threads, cap=1 << 20)
                assert rc == 0 and hashed == count
                part = [l for l in mine if first <= key(l) < first + count]
                assert part == sorted(orc.found_lines(o, n)) and len(part) > 60
                covered += len(part)
            assert wins[w][4] == span and wins[w][3] == len(mine)
    assert covered > (3000 if size <= 21 else 300)
    assert "set-ups" in text


def rnd_windows(text):
    """stdout of `rnd` -> header line, [(mask_s line, mask_e line, [found lines printed in the window], found, checked)]"""
    body = text[text.index("[RANDOM MODE]"):]
    blocks = body.split("\n\n")
    wins = []
    for b in blocks[1:]:
        rows = [r for r in b.split("\n") if r]
        if len(rows) < 3 or not re.fullmatch(r"[0-9a-f ]{67}", rows[0]):
            continue
        m = re.fullmatch(r"([\d,]+) / ([\d,]+) ~ [\d.]+s", rows[-1])
        assert m, rows[-1]
        wins.append((rows[0], rows[1], sorted(rows[2:-1]), int(m.group(1).replace(",", "")), int(m.group(2).replace(",", ""))))
    return blocks[0], wins


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["rnd_d0_20_overscan", "rnd_d0_22_cu_endo"])
def test_rnd_reproduces_the_references_single_window_runs(cli, tmp_path, name):
    """the reference's deterministic `rnd` runs (`-d 0:N` on a range one window wide: it recognises the full range and
    exits after one window, main.c:643,658), captured from the reference binary by tests/golden/make_golden.py:
    header, both printed masks, the window's `found / checked` summary and the found list must be the reference's."""
    g = G[name]
    blf = str(tmp_path / "f.blf")
    write_blf(blf, synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"]))
    out = str(tmp_path / "rnd.txt")
    args = [a for a in g["args"] if a not in ("-t", "1")]
    args[args.index("-f") + 1] = blf
    pr = subprocess.run([cli] + args + ["-q", "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert pr.returncode == 0, pr.stderr.decode()[-1000:]
    header, wins = rnd_windows(pr.stdout.decode())
    assert header == g["header"] and len(wins) == 1
    ms, me, _, found, checked = wins[0]
    assert (ms, me) == (g["mask_s"], g["mask_e"]) and (found, checked) == (g["window_found"], g["window_checked"])
    lines = sorted(l.rstrip("\n") for l in open(out))
    assert len(lines) == g["count"] and digest(lines) == g["sha256_sorted"] and lines[:16] == g["head"]


@pytest.mark.gpu
def test_rnd_windows_at_an_offset_against_the_oracle(cli, tmp_path):
    """`rnd -d 128:21` on the 168-bit range of the reference fixture `rnd_windows_d128_21` (same filter): the windows
    are random, so each one the program prints is checked against the ORACLE's cmd_add workers over the printed bounds
    (stride 2^128, cmd_rnd's full-size jobs) - the found lines printed between the masks and the summary line - the way
    tests/test_oracle_golden.py checks the oracle against the windows the reference drew; and the bounds must have the
    shape gen_random_range gives them (main.c:580-591)."""
    import orc
    g = G["rnd_windows_d128_21"]
    words = synth_bloom_words(g["bloom"]["words"], g["bloom"]["seed"], g["bloom"]["mode"])
    blf = str(tmp_path / "f.blf")
    write_blf(blf, words)
    args = [a for a in g["args"] if a not in ("-t", "1")]
    args[args.index("-f") + 1] = blf
    lo, hi = (int(x, 16) for x in args[args.index("-r") + 1].split(":"))
    env = dict(os.environ, ECLOOP_HIP_RND_WINDOWS="3")
    pr = subprocess.run([cli] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert pr.returncode == 0, pr.stderr.decode()[-1000:]
    header, wins = rnd_windows(pr.stdout.decode())
    assert header == g["header"] and len(wins) == 3
    flt = orc.OrcFilter(bloom_words=words)
    field = ((1 << 21) - 1) << 128
    seen = set()
    for ms, me, printed, found, checked in wins:
        assert re.fullmatch(r"([0-9a-f]{16} ){3}[0-9a-f]{16}", ms) and re.fullmatch(r"([0-9a-f]{16} ){3}[0-9a-f]{16}", me)
        s, e = int(ms.replace(" ", ""), 16), int(me.replace(" ", ""), 16)
        assert lo <= s < e <= hi and s & field == 0 and e == s | field
        rc, out, n, want_checked, hashed = orc.add_range(flt, s, e, offs=128, rnd=True, threads=8)
        want = sorted("%s: %s <- %s" % tuple(l.split("\t")) for l in orc.found_lines(out, n))
        assert rc == 0 and (found, checked) == (n, want_checked) and printed == want and n > 100
        seen.add(s)
    assert len(seen) == 3  # three different draws


@pytest.mark.gpu
def test_pause_resume_keys(cli, tmp_path):
    """'p' / 'r' (main.c:874-888, lib/utils.c:559-626): the scan stops at the next status update, the status line offers
    the other key, the counter stands still, and the paused time is not in the reported time.  The keys are fed through
    a FIFO named by ECLOOP_HIP_TTY (the GPU boxes have no pty devices; with a terminal the program reads /dev/tty)."""
    import select
    import time
    out, keys = str(tmp_path / "o.txt"), str(tmp_path / "keys")
    os.mkfifo(keys)
    t_start = time.time()
    pr = subprocess.Popen([cli, "add", "-f", os.path.join(GOLD, "btc-puzzles-hash"), "-r", "1000000000:2fffffffff", "-q", "-o", out],
                          stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                          env=dict(os.environ, ECLOOP_HIP_TTY=keys))
    kfd = os.open(keys, os.O_WRONLY)  # the program holds the FIFO open O_RDWR, so this does not block for long
    fd = pr.stderr.fileno()
    buf = b""

    def pump(seconds):
        nonlocal buf
        end = time.time() + seconds
        while time.time() < end:
            r, _, _ = select.select([fd], [], [], 0.05)
            if r:
                chunk = os.read(fd, 65536)
                if not chunk:
                    return False
                buf += chunk
        return True

    def last_status():
        st = [s for s in buf.decode(errors="replace").replace("\x1b[2K", "\r").split("\r") if "Mkeys/s" in s]
        return counts(st[-1].split("(")[0])[1], st[-1]

    try:
        deadline = time.time() + 120
        while b"pause)" not in buf and time.time() < deadline:
            assert pump(0.2), buf[-500:]
        assert b"('p' \xe2\x80\x93 pause)" in buf
        os.write(kfd, b"p")
        assert pump(1.0)
        c1, line1 = last_status()
        assert "resume" in line1, line1
        assert pump(1.0)
        c2, line2 = last_status()
        assert c2 == c1 and "resume" in line2  # nothing advances while paused
        os.write(kfd, b"r")
        while pump(0.2):
            pass
    finally:
        os.close(kfd)
        rc = pr.wait(timeout=120)
    assert rc == 0
    final = [s for s in buf.decode(errors="replace").replace("\x1b[2K", "\r").split("\r") if "Mkeys/s" in s][-1]
    found, checked = counts(final.split("\n")[0])
    assert checked == 0x2000000000 and found == len(open(out).readlines()) == 2  # puzzles 37 and 38
    assert "pause" not in final and "resume" not in final
    secs = float(final.split("s ~")[0])
    assert secs < time.time() - t_start - 1.5  # the two paused seconds are not in the reported time


# ---- `mul` text front end (no GPU): the hidden `parse` command prints the scalars the device threads would get


def _parse(cli, data, *flags):
    pr = subprocess.run([cli, "parse", *flags], input=data, stdout=subprocess.PIPE, check=True, timeout=300)
    return pr.stdout.decode().split()


def test_mul_parser_hex_lines_match_fe_modn_from_hex(cli):
    """fe_modn_from_hex (lib/ecc.c:81-95,262-265) on every kind of line: the 64-digit fast path (SSSE3), upper case,
    short / over-long values, prefixes and junk characters (skipped), values >= n (reduced), CRLF, empty lines"""
    import random
    import orc
    r = random.Random(5)
    lines = []
    for _ in range(30000):
        t = r.random()
        if t < 0.6:
            l = "%064x" % r.getrandbits(256)
        elif t < 0.7:
            l = ("%064x" % r.getrandbits(256)).upper()
        elif t < 0.75:
            l = "%x" % r.getrandbits(r.randrange(1, 300))
        elif t < 0.8:
            l = "0x" + "%064x" % r.getrandbits(256)
        elif t < 0.85:
            l = "zz%062xgg" % r.getrandbits(200)
        elif t < 0.88:
            l = "f" * 64
        elif t < 0.92:
            l = "%064x" % (orc.N + r.randrange(-3, 3))
        elif t < 0.95:
            l = ""
        else:
            l = "".join(r.choice("0123456789abcdefXYZ -") for _ in range(r.randrange(1, 100)))
        lines.append(l)
    want = ["%064x" % orc.sn_from_hex(l) for l in lines if l]
    assert _parse(cli, ("\n".join(lines) + "\n").encode()) == want
    assert _parse(cli, ("\r\n".join(lines)).encode()) == want  # CRLF, no newline at the end


def test_mul_parser_fixed_record_path_and_its_fallback(cli):
    """a chunk made only of 64-hex-digit lines is parsed in place (record r at byte 65 r); ONE line of the same length
    with a character that is not a hex digit, or a missing final newline, sends the chunk through the general parser -
    same scalars either way (fe_modn_from_hex skips the junk character, lib/ecc.c:81-95)"""
    import random
    import orc
    r = random.Random(11)
    lines = [("%064x" % r.getrandbits(256)) if i % 3 else ("%064X" % r.getrandbits(256)) for i in range(40000)]
    lines[7] = "%064x" % (orc.N + 1)
    want = ["%064x" % orc.sn_from_hex(l) for l in lines]
    assert _parse(cli, ("\n".join(lines) + "\n").encode()) == want       # fixed path
    assert _parse(cli, ("\n".join(lines)).encode()) == want              # no final newline: general path
    bad = list(lines)
    bad[23456] = bad[23456][:10] + "z" + bad[23456][11:]                  # still 64 characters
    bad[39999] = " " + bad[39999][1:]
    assert _parse(cli, ("\n".join(bad) + "\n").encode()) == ["%064x" % orc.sn_from_hex(l) for l in bad]
    crlf = ("\r\n".join(lines[:1000]) + "\r\n").encode()                 # 66-byte records
    assert _parse(cli, crlf) == want[:1000]


def test_mul_lines_longer_than_1024_characters_are_read_in_pieces(cli):
    """cmd_mul reads with fgets(line, 1025) (main.c:18,548-552): a longer line arrives in pieces of 1024 characters, each an
    entry of its own (a '\\r' at the end of a piece is dropped, empty pieces are skipped) - hex and -raw"""
    import random
    import orc
    r = random.Random(1024)
    def pieces(line):
        out = []
        for q in range(0, len(line), 1024):
            p = line[q:q + 1024]
            if p.endswith("\r"):
                p = p[:-1]
            if p:
                out.append(p)
        return out
    hexl = ["%064x" % r.getrandbits(256), "".join(r.choice("0123456789abcdef") for _ in range(2500)),
            "".join(r.choice("0123456789abcdef") for _ in range(1024)), "".join(r.choice("0123456789abcdef") for _ in range(1025)),
            "".join(r.choice("0123456789abcdef") for _ in range(1023)) + "\r" + "ab", "%064x" % 5]
    want = ["%064x" % orc.sn_from_hex(p) for l in hexl for p in pieces(l)]
    assert _parse(cli, ("\n".join(hexl) + "\n").encode()) == want
    rawl = ["x" * 3000, "abc", "y" * 1024, "z" * 2047 + "\r"]
    want = [hashlib.sha256(p.encode()).hexdigest() for l in rawl for p in pieces(l)]
    assert _parse(cli, ("\n".join(rawl) + "\n").encode(), "-raw") == want


def test_mul_long_lines_against_the_references_reader(cli):
    """the same, pinned to the reference: its found lines (all-ones filter) for tests/golden/mul_long_lines_{hex,raw}.txt
    carry the private key of every piece it read; the front end must produce exactly those scalars"""
    for name, flag in (("mul_long_lines_hex", ()), ("mul_long_lines_raw", ("-raw",))):
        data = open(os.path.join(GOLD, name + ".txt"), "rb").read()
        want = sorted(l.split("\t")[2] for l in G[name]["lines"])
        assert sorted(_parse(cli, data, *flag)) == want and len(want) == G[name]["count"] >= 11


@pytest.mark.gpu
def test_mul_long_lines_found_lines_equal_the_references(cli, tmp_path):
    ones = str(tmp_path / "ones.blf")
    write_blf(ones, np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    for name, flag in (("mul_long_lines_hex", []), ("mul_long_lines_raw", ["-raw"])):
        lines, status, _ = run(cli, ["mul", "-f", ones] + flag, stdin_path=os.path.join(GOLD, name + ".txt"), out=str(tmp_path / (name + ".txt")))
        assert lines == sorted(G[name]["lines"]) and list(counts(status)) == G[name]["status"]


def test_mul_parser_raw_and_bin_and_chunk_boundaries(cli):
    """-raw = SHA-256 of the line (main.c:505-527) for lengths around the padding boundaries; -bin passes 32-byte
    little-endian scalars through; an input of several 64 MB chunks keeps every line, in order"""
    import random
    r = random.Random(9)
    words = ["".join(r.choice("abcdefghijklmnopqrstuvwxyz0123456789 !") for _ in range(n)) for n in
             list(range(1, 200)) + [r.randrange(1, 1000) for _ in range(300)]]
    got = _parse(cli, ("\n".join(words) + "\n").encode(), "-raw")
    assert got == [hashlib.sha256(w.encode()).hexdigest() for w in words]
    ks = [r.getrandbits(256) for _ in range(5000)]
    raw = b"".join(k.to_bytes(32, "little") for k in ks)
    assert _parse(cli, raw, "-bin") == ["%064x" % k for k in ks]
    n = 2_300_000  # 65 bytes per line -> 150 MB: three text chunks
    blob = np.frombuffer(np.random.default_rng(3).bytes(n * 32), dtype=np.uint8)
    hexd = np.frombuffer(b"0123456789abcdef", dtype=np.uint8)
    text = np.empty((n, 65), dtype=np.uint8)
    b = blob.reshape(n, 32)
    text[:, 0:64:2], text[:, 1:64:2], text[:, 64] = hexd[b >> 4], hexd[b & 15], 10
    out = _parse(cli, text.tobytes())
    assert len(out) == n
    # the same input as a regular file on stdin takes the mmap path (slices of the mapping, no reader copies)
    import tempfile
    with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".txt") as f:
        f.write(text.tobytes())
        f.flush()
        pr = subprocess.run([cli, "parse"], stdin=open(f.name, "rb"), stdout=subprocess.PIPE, check=True, timeout=300)
    assert pr.stdout.decode().split() == out
    with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".bin") as f:
        f.write(raw)
        f.flush()
        pr = subprocess.run([cli, "parse", "-bin"], stdin=open(f.name, "rb"), stdout=subprocess.PIPE, check=True, timeout=300)
    assert pr.stdout.decode().split() == ["%064x" % k for k in ks]
    N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    for i in list(range(0, n, 9973)) + [n - 1, 1032444, 1032445, 1032446]:
        v = int.from_bytes(b[i].tobytes(), "big")
        assert int(out[i], 16) == (v - N if v >= N else v), i


def test_mul_parser_raw_file_in_chunks_of_a_forced_size(cli, tmp_path):
    """`mul -raw` from a regular file (cli_mul.h: the mapped file in chunks of mul_raw_chunk() bytes cut at line ends, P threads list the
    lines, a second pass packs the table): 200 000 pass phrases of 0..60 characters, an empty line and a CR LF line among them, the last
    line without newline, with the chunk size forced to 64 KB / 1 MB (dozens of chunks, lines that straddle the cut) - the digests the
    device would compute from the line table (hidden `parse` command) against hashlib; the same bytes through a pipe give the same"""
    import hashlib
    import random
    r = random.Random(21)
    abc = "abcdefghijklmnopqrstuvwxyz0123456789 -_"
    lines = ["".join(r.choice(abc) for _ in range(r.randrange(0, 61))) for _ in range(200_000)]
    lines[5], lines[77] = "", "with carriage return\r"
    blob = ("\n".join(lines)).encode()  # (no newline at the end)
    want = [hashlib.sha256(l.rstrip("\r").encode()).hexdigest() for l in lines if l.rstrip("\r")]
    src = tmp_path / "phrases.txt"
    src.write_bytes(blob)
    for chunk in ("65536", "1048576", None):
        env = dict(os.environ, ECLOOP_HIP_STATS="1")
        if chunk:
            env["ECLOOP_HIP_MUL_RAW_CHUNK"] = chunk
        pr = subprocess.run([cli, "parse", "-raw"], stdin=open(src, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, timeout=300, env=env)
        assert pr.stdout.decode().split() == want, chunk
        m = re.search(r"(\d+) chunks", pr.stderr.decode())
        assert m and (int(m.group(1)) >= len(blob) // int(chunk) if chunk else int(m.group(1)) == 1), pr.stderr.decode()[-300:]
    assert _parse(cli, blob, "-raw") == want


@pytest.mark.parametrize("decoder", ["avx512", "avx2", "ssse3"])
def test_mul_parser_file_of_64_digit_lines_in_batches(cli, tmp_path, decoder):
    """a regular file of 64-digit lines on stdin is taken in batches straight from the file (cli_mul.h: pread slices, one decoder per
    instruction set: a whole record per AVX-512 load, two AVX2 halves, four SSSE3 quarters) - against fe_modn_from_hex in python ints and
    against the general reader on the same bytes: random values, upper case, values >= n (reduced; top limb all ones is the AVX-512 form's
    slow case), a batch count that does not divide the file; then the same file with a record that is no such line in the middle (the
    path stops at the batch before it, the general reader takes that batch's bytes, the batch path goes on after it) and with a short line +
    a last line without newline."""
    import random
    r = random.Random(9)
    N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    vals = [r.getrandbits(256) for _ in range(50000)]
    vals[7], vals[8], vals[9], vals[10], vals[11] = N, N - 1, (1 << 256) - 1, N + 12345, ((1 << 64) - 1) << 192
    lines = [("%064x" % v).upper() if i % 5 == 3 else "%064x" % v for i, v in enumerate(vals)]
    env = dict(os.environ, ECLOOP_HIP_MUL_BATCH_LOG2="13", ECLOOP_HIP_MUL_SLICE="4096")
    if decoder != "avx512":
        env["ECLOOP_HIP_NO_AVX512"] = "1"
    if decoder == "ssse3":
        env["ECLOOP_HIP_NO_AVX2"] = "1"

    def parse(path, **more):
        pr = subprocess.run([cli, "parse"], stdin=open(path, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, timeout=300,
                            env=dict(env, ECLOOP_HIP_STATS="1", **more))
        return pr.stdout.decode().split(), pr.stderr.decode()

    clean = tmp_path / "clean.txt"
    clean.write_text("".join(l + "\n" for l in lines))
    got, stats = parse(str(clean))
    assert got == ["%064x" % (v % N) for v in vals]
    assert "7 batches of fixed records straight from the file (50000 lines)" in stats  # 6 x 8192 + 848
    assert parse(str(clean), ECLOOP_HIP_MUL_READ="chunks")[0] == got and parse(str(clean), ECLOOP_HIP_MUL_READ="mmap")[0] == got
    odd = list(lines)
    odd[30000] = "0x" + odd[30000][:40]            # a 42-character line in the fourth batch
    odd[30001] = odd[30001][:63] + "g"             # 64 characters, one of them no hex digit (skipped by fe_modn_from_hex)
    bad = tmp_path / "bad.txt"
    bad.write_text("".join(l + "\n" for l in odd) + "abc\n" + lines[0])  # ... a short line and a last line without newline
    got, stats = parse(str(bad))
    want, _ = parse(str(bad), ECLOOP_HIP_MUL_READ="chunks")
    assert got == want and len(got) == 50002 and got[-1] == "%064x" % (vals[0] % N) and got[-2] == "%064x" % 0xABC
    assert got[30000] == "%064x" % int(odd[30000][2:], 16) and got[30001] == "%064x" % int(odd[30001][:63], 16)
    # batches 0-2 (24576 lines), the fourth batch's bytes through the general reader (it holds the two odd lines), then the batch path again
    # on the new alignment: 2 more batches before the one that meets the short line and the unterminated last one
    assert "5 batches of fixed records straight from the file (40960 lines)" in stats
    # a file that does not START with a record (a header line): the general reader takes the first stretch, the batch path the rest
    head = tmp_path / "head.txt"
    head.write_text("0x1f\n# keys\n" + "".join(l + "\n" for l in lines))
    got, stats = parse(str(head), ECLOOP_HIP_MUL_STRETCH="4096")
    want, _ = parse(str(head), ECLOOP_HIP_MUL_READ="chunks")
    m = re.search(r"(\d+) batches of fixed records straight from the file \((\d+) lines\)", stats)
    assert got == want and got[0] == "%064x" % 0x1F and got[2:] == ["%064x" % (v % N) for v in vals] and m and int(m.group(2)) >= 40000, stats


def test_mul_parser_crlf_files_and_binary_files_in_batches(cli, tmp_path):
    """the batch path over a regular file also takes 66-byte records (64 digits + CR LF: key lists written on Windows) and, with -bin,
    the 32-byte scalars themselves; both against the general reader on the same bytes, a tail that is no whole record included"""
    import random
    r = random.Random(10)
    N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    vals = [r.getrandbits(256) for _ in range(30000)]
    env = dict(os.environ, ECLOOP_HIP_MUL_BATCH_LOG2="12", ECLOOP_HIP_MUL_SLICE="4096", ECLOOP_HIP_STATS="1")

    def parse(path, *flags, **more):
        pr = subprocess.run([cli, "parse", *flags], stdin=open(path, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, timeout=300, env=dict(env, **more))
        return pr.stdout.decode().split(), pr.stderr.decode()

    crlf = tmp_path / "crlf.txt"
    crlf.write_bytes(b"".join(b"%064x\r\n" % v for v in vals) + b"1f\r\n")
    got, stats = parse(str(crlf))
    assert got == ["%064x" % (v % N) for v in vals] + ["%064x" % 0x1F] and "8 batches of fixed records straight from the file (30000 lines)" in stats
    assert parse(str(crlf), ECLOOP_HIP_MUL_READ="chunks")[0] == got
    mixed = tmp_path / "mixed.txt"  # LF records, then CRLF ones: the path ends where the record length changes
    mixed.write_bytes(b"".join(b"%064x\n" % v for v in vals[:10000]) + b"".join(b"%064x\r\n" % v for v in vals[10000:]))
    got, stats = parse(str(mixed))
    # 2 batches of LF records, the batch with the change of record length through the general reader, then CR LF records in batches again
    m = re.search(r"(\d+) batches of fixed records straight from the file \((\d+) lines\)", stats)
    assert got == ["%064x" % (v % N) for v in vals] and m and int(m.group(1)) >= 6 and int(m.group(2)) >= 24000, stats
    # keys written as 0x + 64 digits (67-byte records, 68 with CR LF): the same path, two bytes further in; an upper-case X among them,
    # and one line with the prefix missing stops the batch it is in
    for name, end, batches in (("x.txt", b"\n", 8), ("xcrlf.txt", b"\r\n", 8)):
        f = tmp_path / name
        f.write_bytes(b"".join((b"0X" if i % 11 == 3 else b"0x") + b"%064x" % v + end for i, v in enumerate(vals)))
        got, stats = parse(str(f))
        assert got == ["%064x" % (v % N) for v in vals] and "%d batches of fixed records straight from the file (30000 lines)" % batches in stats, stats
        assert parse(str(f), ECLOOP_HIP_MUL_READ="chunks")[0] == got
    f = tmp_path / "xodd.txt"
    f.write_bytes(b"".join((b"" if i == 20000 else b"0x") + b"%064x\n" % v for i, v in enumerate(vals)))
    got, stats = parse(str(f))
    m = re.search(r"(\d+) batches of fixed records straight from the file \((\d+) lines\)", stats)
    assert got == ["%064x" % (v % N) for v in vals] and m and 20000 <= int(m.group(2)) < 30000, stats
    raw = tmp_path / "k.bin"
    raw.write_bytes(b"".join(v.to_bytes(32, "little") for v in vals) + b"\x01\x02\x03")
    got, stats = parse(str(raw), "-bin")
    assert got == ["%064x" % v for v in vals] and "8 batches of fixed records straight from the file (30000 lines)" in stats
    assert parse(str(raw), "-bin", ECLOOP_HIP_MUL_READ="chunks")[0] == got and parse(str(raw), "-bin", ECLOOP_HIP_MUL_READ="mmap")[0] == got


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["add"], ["rnd", "-d", "64:64", "-seed", "w"]])
def test_unbounded_scans_stream_instead_of_being_refused(cli, tmp_path, args):
    """`add` without -r walks 0x800:p in 2^21-key jobs for ever (main.c:405-454, 668-672) and `rnd -d x:64` scans
    2^64-key windows: neither key count fits 64 bits.  The host program hands the scan out chunk by chunk from a
    256-bit counter: it must start, keep counting, and stop cleanly on SIGINT."""
    import signal
    import time
    out = str(tmp_path / "o.txt")
    pr = subprocess.Popen([cli] + args + ["-f", os.path.join(GOLD, "btc-puzzles-hash"), "-q", "-o", out], stdin=subprocess.DEVNULL,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    time.sleep(6)
    assert pr.poll() is None, pr.stderr.read().decode(errors="replace")[-500:]
    pr.send_signal(signal.SIGINT)
    so, se = pr.communicate(timeout=60)
    st = [s for s in se.decode(errors="replace").replace("\x1b[2K", "\r").split("\r") if "Mkeys/s" in s]
    assert st and counts(st[-1].split("(")[0])[1] >= 1 << 33, st[-3:]
    if args[0] == "add":  # the first keys of the default range hold four puzzle keys below 0x10000... none below 0x800: just a sanity check of the banner
        assert b"range_s: 0000000000000000 0000000000000000 0000000000000000 0000000000000800" in so


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["text", "bin"])
def test_mul_fans_out_over_device_threads(cli, tmp_path, mode):
    """`mul -t 4` (four device threads on the one GPU, ECLOOP_HIP_SHARE_GPU): an input of several 64 MB chunks - the
    brainwallet keys of `make mul` twice, far apart, in 2.3 M filler lines - is parsed chunk by chunk and every chunk
    goes to whichever device thread is idle (main.c:556-571).  Every key is found exactly as often as it occurs and the
    counter equals the line count; the same through `-bin` (32-byte little-endian scalars)."""
    from collections import Counter
    bw = [l.strip() for l in open(os.path.join(GOLD, "btc-bw-priv")) if l.strip()]
    rng = np.random.default_rng(12)
    fill = ["%064x" % int.from_bytes(rng.bytes(32), "big") for _ in range(1000)]
    n_fill = 1_150_000
    blocks = [fill * (n_fill // 1000), bw, fill * (n_fill // 1000), bw]
    lines_in = [l for b in blocks for l in b]
    src = tmp_path / ("in." + mode)
    if mode == "text":
        src.write_text("\n".join(lines_in) + "\n")
        extra = []
    else:
        N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
        src.write_bytes(b"".join((int(l, 16) % N).to_bytes(32, "little") for l in lines_in))
        extra = ["-bin"]
    env = dict(os.environ, ECLOOP_HIP_SHARE_GPU="4")
    lines, status, stdout = run(cli, ["mul", "-f", os.path.join(GOLD, "btc-bw-hash"), "-a", "cu", "-t", "4"] + extra, stdin_path=str(src),
                                out=str(tmp_path / "m.txt"), env=env)
    assert "gpus: 4 " in stdout
    c = Counter(lines)
    assert len(c) == 1080 and set(c.values()) == {2} and digest(sorted(c)) == G["make_mul_bw"]["sha256_sorted"]
    assert counts(status) == (2160, len(lines_in))


@pytest.mark.gpu
def test_mul_raw_fans_out_over_device_threads(cli, tmp_path):
    """`mul -raw -t 4`: pass phrases in several 32 MB chunks, hashed on the device (ecl_hip_mul_batch_raw) by whichever of
    the four device threads is idle; 200 target phrases, each twice and far apart, must be found exactly twice with the
    private key SHA-256(phrase) (main.c:505-527) and nothing else; the counter equals the number of non-empty lines"""
    import orc
    from collections import Counter
    targets = ["correct horse battery staple %d" % i for i in range(200)]
    keys = [int.from_bytes(hashlib.sha256(t.encode()).digest(), "big") for t in targets]
    hs = [orc.hash160(*orc.point_of(k % orc.N), True) for k in keys]
    lst = tmp_path / "targets.txt"
    lst.write_text("".join("".join("%08x" % w for w in h) + "\n" for h in hs))
    fill = ["filler phrase %07d" % i for i in range(2_400_000)]  # 21 bytes a line: 50 MB a block
    src = tmp_path / "phrases.txt"
    with open(src, "w") as f:
        for block in (fill, targets, [""], fill, targets):
            f.write("\n".join(block) + "\n")
    env = dict(os.environ, ECLOOP_HIP_SHARE_GPU="4")
    lines, status, stdout = run(cli, ["mul", "-raw", "-f", str(lst), "-t", "4"], stdin_path=str(src), out=str(tmp_path / "r.txt"), env=env)
    assert "gpus: 4 " in stdout
    c = Counter(lines)
    want = {"addr33\t%s\t%064x" % ("".join("%08x" % w for w in h), k) for h, k in zip(hs, keys)}
    assert set(c) == want and set(c.values()) == {2}
    assert counts(status) == (400, 2 * len(fill) + 2 * len(targets))


def test_scan_plan_matches_the_oracles_job_loop(cli):
    """the hidden `plan` command prints what scan_plan() derives from -r / -d: keys hashed and status counter of
    cmd_add (main.c:405-454).  Compared with the oracle, which RUNS the reference's job loop (counter stepping by
    job_size * stride until range_e, each job hashing ceil(job_size / 2048) * 2048 keys): sub-job ranges, ranges of a few
    jobs, strides, ranges that end one scalar after a job boundary."""
    import random
    import orc
    r = random.Random(77)
    zero = orc.OrcFilter(bloom_words=np.zeros(8, np.uint64))
    cases = [(0x8000, 0xFFFF, 0), (0x8000, 0x8001, 0), (0x801, 0x802, 0), (0x9000, 0x9801, 0), (0x800000, 0xFFFFFF, 0),
             (0x8000, 0x8000 + (1 << 21), 0), (0x8000, 0x8001 + (1 << 21), 0), (0x8000, 0x7FFF + (1 << 21), 0)]
    for _ in range(14):
        offs = r.choice([0, 0, 1, 3, 7, 40, 128, 200])
        a = r.randrange(1 << (offs + 33), 1 << (offs + 40))
        keys = r.choice([1, 2, 100, 2047, 2048, 2049, 5000, 70000])
        b = a + keys * (1 << offs) + (r.randrange(1 << offs) if offs and r.random() < 0.5 else 0)
        cases.append((a, b, offs))
    for a, b, offs in cases:
        args = ["plan", "-r", f"{a:x}:{b:x}"] + (["-d", f"{offs}:32"] if offs else [])
        out = subprocess.run([cli] + args, stdout=subprocess.PIPE, check=True).stdout.decode().split()
        got = dict(zip(out[::2], out[1::2]))
        eff = int(got["ord_offs"])
        bits = max(20, b.bit_length())
        assert eff == min(offs, max(1, bits - (bits if bits < 32 else 32))) if offs else eff == 0  # load_offs_size (main.c:703-746)
        rc, _, n, checked, hashed = orc.add_range(zero, a, b, offs=eff, verify=False, threads=4)
        assert rc == 0 and n == 0
        assert int(got["hashed"], 16) == hashed and int(got["status_total"]) == checked, (hex(a), hex(b), offs, got, checked, hashed)
    # -endo multiplies the counter by 6 (main.c:431); the default range is too long to count and is streamed
    out = subprocess.run([cli, "plan", "-r", "8000:ffffff", "-endo"], stdout=subprocess.PIPE, check=True).stdout.decode().split()
    assert out[out.index("status_total") + 1] == str(6 * 16777216)
    out = subprocess.run([cli, "plan"], stdout=subprocess.PIPE, check=True).stdout.decode().split()
    assert out[out.index("status_total") + 1] == "0" and int(out[out.index("hashed") + 1], 16) > 1 << 255


def test_rnd_window_wider_than_255_bits_moves_the_stride_with_the_offset(cli):
    """`rnd -d 200:64` on a 256-bit range: load_offs_size lets the offset through (max_offs = 224), cmd_rnd lowers it to
    255 - 64 = 191 (main.c:620) BEFORE ctx_precompute_gpoints derives stride_k = 2^191 from it (main.c:222,624).  The stride
    the walk uses and the offset of the window mask must be the same number (round-3 advisor finding: the clamp came after
    the stride had been fixed at 2^200, so 511 of 512 keys of the window were never visited); `add` is not clamped."""
    def plan(*a):
        out = subprocess.run([cli, "plan"] + list(a), stdout=subprocess.PIPE, check=True).stdout.decode().split()
        return dict(zip(out[::2], out[1::2]))
    top = "%x:%x" % (1 << 255, (1 << 256) - (1 << 130))  # below n
    for offs, size, want in ((200, 64, 191), (224, 64, 191), (224, 40, 215), (100, 64, 100), (192, 63, 192), (193, 63, 192)):
        got = plan("-rnd", "-r", top, "-d", f"{offs}:{size}")
        assert (int(got["ord_offs"]), int(got["stride_bits"]), int(got["ord_size"])) == (want, want, size), (offs, size, got)
    got = plan("-r", top, "-d", "224:64")
    assert (int(got["ord_offs"]), int(got["stride_bits"])) == (224, 224)


def test_context_to_gpu_map(cli):
    """which GPU each device context opens (hidden `plan -visible R [-mul]`, no GPU needed): `add` / `rnd` one context per
    GPU of `-t`; `mul` two contexts per GPU, both on the SAME GPU and never on one beyond `-t` (round-2 advisor finding:
    `-t 1` on an 8-GPU box put the second context on GPU 1)."""
    def devmap(*a):
        out = subprocess.run([cli, "plan", "-r", "8000:ffff"] + list(a), stdout=subprocess.PIPE, check=True).stdout.decode().split()
        return int(out[1]), int(out[3]), [int(x) for x in out[5:]]
    assert devmap("-t", "1", "-visible", "8", "-mul") == (2, 1, [0, 0])
    assert devmap("-t", "3", "-visible", "8", "-mul") == (6, 3, [0, 1, 2, 0, 1, 2])
    assert devmap("-t", "8", "-visible", "8", "-mul") == (16, 8, list(range(8)) * 2)
    assert devmap("-t", "8", "-visible", "8") == (8, 8, list(range(8)))
    assert devmap("-t", "8", "-visible", "2") == (2, 2, [0, 1])
    assert devmap("-t", "4", "-visible", "1", "-mul") == (2, 1, [0, 0])
