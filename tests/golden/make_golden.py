#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE UNMODIFIED REFERENCE.

    make -C oracle ref            # builds oracle/_ref/ecloop_* from /root/reference (sources stay there)
    python tests/golden/make_golden.py

Only inputs and outputs are stored (no reference text).  Needs /root/reference/data for the two
reference-owned input lists; everything else is synthesised here.  Output: tests/golden/golden.json
(+ .npz files for the per-key dumps).  The all-ones-.blf trick turns the reference into a
per-key hash160 dump: a bloom filter whose every bit is set reports every hashed key as "found".
"""
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import synth_bloom_words, write_blf  # noqa: E402  (shared with the tests: same inputs)

REF_DATA = "/root/reference/data"


def ref_bin():
    for name in ("ecloop_sane", "ecloop_avx2", "ecloop_native"):
        p = os.path.join(ROOT, "oracle", "_ref", name)
        if os.path.exists(p):
            return p
    raise SystemExit("build the reference first: make -C oracle ref")


def run_ref(args, stdin=None):
    """Run the reference with -t 1 -q -o <tmp>; return (sorted found lines, final status line)."""
    with tempfile.NamedTemporaryFile("r", suffix=".txt", delete=False) as out:
        path = out.name
    try:
        cmd = [ref_bin()] + args + ["-q", "-o", path]
        pr = subprocess.run(cmd, stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        with open(path) as f:
            lines = [l.rstrip("\n") for l in f if l.strip()]
        status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
        return sorted(lines), status
    finally:
        os.unlink(path)


def digest(lines):
    return hashlib.sha256(("\n".join(lines) + "\n").encode()).hexdigest()


def parse(lines):
    """found lines -> (compressed u8[N], h160 u32[N,5], pk u64[N,4] little-endian limbs)"""
    n = len(lines)
    comp = np.zeros(n, np.uint8)
    h = np.zeros((n, 5), np.uint32)
    pk = np.zeros((n, 4), np.uint64)
    for i, l in enumerate(lines):
        label, hh, kk = l.split("\t")
        comp[i] = 1 if label == "addr33" else 0
        h[i] = [int(hh[8 * j : 8 * j + 8], 16) for j in range(5)]
        v = int(kk, 16)
        pk[i] = [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
    return comp, h, pk


def status_counts(status):
    # "3.25s ~ 2.58 Mkeys/s ~ 1 / 8,388,608"
    tail = status.split("~")[-1]
    found, checked = tail.split("/")
    clean = lambda s: int("".join(ch for ch in s if ch.isdigit()))
    return clean(found), clean(checked)


def rnd_cases(g, tmp):
    """`rnd` (main.c:580-662) pinned to the reference.  Two kinds of fixture:
    * a range exactly ONE window wide at offset 0 (`-d 0:N`, range = [H + 0, H + 2^N - 1]): every random draw clears /
      sets to the same bounds, the reference recognises the full range (`is_full`, main.c:643,658) and exits after one
      window - masks, summary line and found list are all deterministic.  (At an offset > 0 the bits BELOW the window
      are random too, so no range makes such a run deterministic.)
    * windows at `-d 128:21` on a 168-bit range: the reference loops for ever, so it is stopped with SIGINT (its handler
      flushes and exits, main.c:867-872) after a few seconds and the COMPLETE windows are kept: both printed masks,
      the found lines printed between them and the `found / checked` summary, window by window."""
    import re
    sp = os.path.join(tmp, "sparse_rnd.blf")
    write_blf(sp, synth_bloom_words(12345, seed=7, mode="a|(b&c)"))
    bloom = {"words": 12345, "seed": 7, "mode": "a|(b&c)"}

    def one_window(name, extra):
        out = os.path.join(tmp, name + ".txt")
        if os.path.exists(out):
            os.unlink(out)
        args = ["rnd", "-f", sp, "-t", "1"] + extra
        pr = subprocess.run(["timeout", "120", ref_bin()] + args + ["-q", "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        text = pr.stdout.decode()
        body = text[text.index("[RANDOM MODE]"):].split("\n")
        summary = re.fullmatch(r"([\d,]+) / ([\d,]+) ~ [\d.]+s", body[4])
        lines = sorted(l.rstrip("\n") for l in open(out))
        g["cases"][name] = {"args": args, "bloom": bloom, "header": body[0], "mask_s": body[2], "mask_e": body[3],
                            "window_found": int(summary.group(1).replace(",", "")), "window_checked": int(summary.group(2).replace(",", "")),
                            "count": len(lines), "sha256_sorted": digest(lines), "head": lines[:16]}
        print(f"{name}: {body[2]} .. {body[3]}: {body[4]}")

    one_window("rnd_d0_20_overscan", ["-r", "100000:1fffff", "-d", "0:20"])            # 2^20-key window, 2^21-key job (over-scan)
    one_window("rnd_d0_22_cu_endo", ["-r", "400000:7fffff", "-d", "0:22", "-a", "cu", "-endo"])  # two jobs, 12 hashes per key
    # windows at an offset: stop the endless loop after a few seconds, keep the complete windows
    lo, hi = 1 << 167, (1 << 168) - 1
    args = ["rnd", "-f", sp, "-t", "1", "-r", f"{lo:x}:{hi:x}", "-d", "128:21"]
    pr = subprocess.run(["timeout", "-s", "INT", "8", ref_bin()] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = pr.stdout.decode()
    blocks = text[text.index("[RANDOM MODE]"):].split("\n\n")[1:]
    wins = []
    for b in blocks:
        rows = b.split("\n")
        m = re.fullmatch(r"([\d,]+) / ([\d,]+) ~ [\d.]+s", rows[-1]) if len(rows) >= 3 else None
        if not m:
            continue  # the window that was running when the signal arrived
        found = sorted(rows[2:-1])
        assert all(re.fullmatch(r"addr33: [0-9a-f]{40} <- [0-9a-f]{64}", f) for f in found), found[:2]
        wins.append({"mask_s": rows[0], "mask_e": rows[1], "found": int(m.group(1).replace(",", "")), "checked": int(m.group(2).replace(",", "")),
                     "stdout_lines_sha256": digest(found), "stdout_lines_head": found[:4]})
        if len(wins) == 4:
            break
    assert len(wins) >= 3, text[-2000:]
    g["cases"]["rnd_windows_d128_21"] = {"args": args, "bloom": bloom, "header": text[text.index("[RANDOM MODE]"):].split("\n")[0], "windows": wins}
    print(f"rnd_windows_d128_21: {len(wins)} complete windows, found {[w['found'] for w in wins]}")


def long_line_cases(g, tmp):
    """cmd_mul's reader: fgets(line, 1025) (main.c:18,548-552) hands a line longer than 1024 characters over in pieces.
    The input is a committed data file (tests/golden/mul_long_lines.txt); the reference's found lines through the all-ones
    filter, -raw and hex, are the fixture."""
    import random
    ones = os.path.join(tmp, "ones.blf")
    write_blf(ones, np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64))
    r = random.Random(1024)
    hx = lambda n: "".join(r.choice("0123456789abcdef") for _ in range(n))
    hexl = [hx(64), hx(2500), hx(1024), hx(1025), hx(1023) + "\r" + "ab", "abc", "%064x" % 5]
    rawl = hexl + ["x" * 3000, "z" * 2047 + "\r"]  # (as hex these would be the scalar 0, which ruins the reference's whole batch)
    for name, extra, lines in (("mul_long_lines_raw", ["-raw"], rawl), ("mul_long_lines_hex", [], hexl)):
        path = os.path.join(HERE, name + ".txt")
        open(path, "w", newline="").write("\n".join(lines) + "\n")
        found, status = run_ref(["mul", "-f", ones, "-t", "1"] + extra, open(path, "rb"))
        g["cases"][name] = {"args": ["mul", "-f", "<all-ones .blf>", "-t", "1"] + extra, "stdin": "tests/golden/" + name + ".txt",
                            "count": len(found), "lines": found, "status": status_counts(status)}
        print(f"{name}: {len(found)} lines, status {status_counts(status)}")


def odd_list_case(g, tmp):
    """load_filter's list reader (main.c:96-110: fgets into a 41-byte buffer, every full 40-character piece an entry) on a
    well-formed but oddly laid out list: the hash160s of 48 keys of 0x8000..0x87ff (taken from the reference's own dump of
    that range) one, two and three to a line, upper case, CRLF, between blank lines, short comments and short junk.  The
    list is a committed data file; the reference's found lines for `add -r 8000:87ff` are the fixture."""
    import random
    d = np.load(os.path.join(HERE, "dump33_8000_87ff.npz"))
    r = random.Random(4040)
    pick = sorted(r.sample(range(len(d["h160"])), 48))
    hexes = ["".join("%08x" % w for w in d["h160"][i]) for i in pick]
    filler = lambda: "".join(r.choice("0123456789abcdef") for _ in range(40))
    out, i = [], 0
    while i < len(hexes):
        kind = r.randrange(7)
        if kind == 0:
            out.append(hexes[i]); i += 1
        elif kind == 1 and i + 2 <= len(hexes):
            out.append(hexes[i] + hexes[i + 1]); i += 2
        elif kind == 2 and i + 3 <= len(hexes):
            out.append(hexes[i] + filler() + hexes[i + 1] + hexes[i + 2] + "abc"); i += 3      # 160 characters + a short tail
        elif kind == 3:
            out.append(hexes[i].upper() + "\r"); i += 1
        elif kind == 4:
            out += ["", "# short comment", filler()]
        elif kind == 5:
            out.append(hexes[i]); out.append(hexes[i]); i += 1                                  # duplicate
        else:
            out.append("deadbeef")                                                              # too short to be an entry
    path = os.path.join(HERE, "odd-list.txt")
    open(path, "w", newline="").write("\n".join(out))                                           # no newline at the end
    found, status = run_ref(["add", "-f", path, "-r", "8000:87ff", "-t", "1"])
    g["cases"]["odd_list_8000_87ff"] = {"args": ["add", "-f", "tests/golden/odd-list.txt", "-r", "8000:87ff", "-t", "1"], "count": len(found),
                                        "lines": found, "status": status_counts(status)}
    print(f"odd_list_8000_87ff: {len(found)} lines, status {status_counts(status)}")


def main():
    if "--only-odd-list" in sys.argv:
        g = json.load(open(os.path.join(HERE, "golden.json")))
        odd_list_case(g, tempfile.mkdtemp())
        json.dump(g, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
        return
    if "--only-long-lines" in sys.argv:
        g = json.load(open(os.path.join(HERE, "golden.json")))
        long_line_cases(g, tempfile.mkdtemp())
        json.dump(g, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
        return
    if "--only-rnd" in sys.argv:  # add / refresh the rnd cases without touching the other fixtures
        g = json.load(open(os.path.join(HERE, "golden.json")))
        rnd_cases(g, tempfile.mkdtemp())
        json.dump(g, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
        return
    g = {"reference": "vladkens/ecloop v0.5.0, built by oracle/Makefile `ref`", "cases": {}}
    tmp = tempfile.mkdtemp()
    ones = os.path.join(tmp, "ones.blf")
    write_blf(ones, np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64))

    def case(name, args, store="npz", stdin_path=None):
        stdin = open(stdin_path, "rb") if stdin_path else None
        lines, status = run_ref(args, stdin)
        found, checked = status_counts(status)
        entry = {"args": args, "count": len(lines), "status_found": found, "status_checked": checked,
                 "sha256_sorted": digest(lines)}
        if stdin_path:
            entry["stdin"] = stdin_path
        if store == "npz":
            comp, h, pk = parse(lines)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), compressed=comp, h160=h, pk=pk)
            entry["npz"] = name + ".npz"
        elif store == "lines":
            entry["lines"] = lines
        elif store == "head":
            entry["head"] = lines[:64]
        g["cases"][name] = entry
        print(f"{name}: {len(lines)} lines, status {found} / {checked}")

    puzzles = os.path.join(REF_DATA, "btc-puzzles-hash")
    # --- known-answer flows of the reference
    case("cfg1_list_800000_ffffff", ["add", "-f", puzzles, "-r", "800000:ffffff", "-t", "1"], "lines")
    case("make_add_8000_ffffff", ["add", "-f", puzzles, "-r", "8000:ffffff", "-t", "1"], "lines")
    case("ci_smoke_8000_ffff", ["add", "-f", puzzles, "-r", "8000:ffff", "-t", "1"], "lines")
    case("endo_cu_list_8000_fffff", ["add", "-f", puzzles, "-r", "8000:fffff", "-t", "1", "-a", "cu", "-endo"], "lines")
    case("make_mul_bw", ["mul", "-f", os.path.join(REF_DATA, "btc-bw-hash"), "-t", "1", "-a", "cu"], "head",
         stdin_path=os.path.join(REF_DATA, "btc-bw-priv"))
    # --- per-key dumps through the all-ones bloom
    case("dump33_8000_87ff", ["add", "-f", ones, "-r", "8000:87ff", "-t", "1"], "npz")
    case("dump65_8000_87ff", ["add", "-f", ones, "-r", "8000:87ff", "-t", "1", "-a", "u"], "npz")
    case("dump_cu_endo_8000_87ff", ["add", "-f", ones, "-r", "8000:87ff", "-t", "1", "-a", "cu", "-endo"], "head")
    # strided: one key wide range, 2048 keys spaced 2^128; range end is 165 bits so ord_offs is not clamped
    a = 1 << 164
    case("dump33_stride128", ["add", "-f", ones, "-r", f"{a + 0x12345:x}:{a + 0x12346:x}", "-d", "128:32", "-t", "1"], "npz")
    # range that is not a multiple of 2048 and overruns its end (QUIRK main.c:442 + 368)
    case("dump33_overrun_9000_9801", ["add", "-f", ones, "-r", "9000:9801", "-t", "1"], "npz")
    # mul KAT: stdin "1" -> hash160 of G, both encodings
    one = os.path.join(tmp, "one.txt")
    open(one, "w").write("1\n")
    case("mul_G", ["mul", "-f", ones, "-t", "1", "-a", "cu"], "lines", stdin_path=one)
    # mul over seeded scalars of assorted sizes (all-ones dump)
    import random
    rnd = random.Random(20250929)
    n_ = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    ks = [2, 3, 0x3FFF, 0x4000, 0x4001, n_ - 1, n_ - 2, (1 << 255) + 12345]
    ks += [rnd.randrange(1, 1 << rnd.choice([8, 14, 28, 64, 128, 200, 256])) % n_ or 5 for _ in range(248)]
    mul_in = os.path.join(HERE, "mul_scalars.txt")
    open(mul_in, "w").write("".join(f"{k:x}\n" for k in ks))
    case("mul_dump_cu", ["mul", "-f", ones, "-t", "1", "-a", "cu"], "npz", stdin_path=mul_in)
    g["cases"]["mul_dump_cu"]["stdin"] = "tests/golden/mul_scalars.txt"
    # --- sparse synthetic bloom: false positives pin the probe order / modulus (non power-of-two size)
    sp = os.path.join(tmp, "sparse.blf")
    write_blf(sp, synth_bloom_words(12345, seed=7, mode="a|(b&c)"))
    case("sparse_fp33_two_jobs", ["add", "-f", sp, "-r", "8000:208800", "-t", "1"], "npz")
    g["cases"]["sparse_fp33_two_jobs"]["bloom"] = {"words": 12345, "seed": 7, "mode": "a|(b&c)"}
    dn = os.path.join(tmp, "dense.blf")
    write_blf(dn, synth_bloom_words(4099, seed=11, mode="a|b"))
    case("dense_fp_cu_endo", ["add", "-f", dn, "-r", "8000:87ff", "-t", "1", "-a", "cu", "-endo"], "npz")
    g["cases"]["dense_fp_cu_endo"]["bloom"] = {"words": 4099, "seed": 11, "mode": "a|b"}
    # --- blf-gen bytes from the puzzles list (comment-free input)
    blf_out = os.path.join(tmp, "puz.blf")
    subprocess.run([ref_bin(), "blf-gen", "-n", "32768", "-o", blf_out], stdin=open(puzzles, "rb"),
                   stdout=subprocess.DEVNULL, check=True)
    raw = open(blf_out, "rb").read()
    g["cases"]["blf_gen_puzzles_32768"] = {"bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest(),
                                           "header_hex": raw[:16].hex(), "size_words": struct.unpack("<Q", raw[8:16])[0]}
    rnd_cases(g, tmp)
    long_line_cases(g, tmp)
    odd_list_case(g, tmp)
    # inputs owned by the reference's data/ directory: stored as data fixtures for the GPU box
    for name in ("btc-puzzles-hash", "btc-bw-hash", "btc-bw-priv"):
        dst = os.path.join(HERE, name)
        open(dst, "wb").write(open(os.path.join(REF_DATA, name), "rb").read())
    json.dump(g, open(os.path.join(HERE, "golden.json"), "w"), indent=1)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
