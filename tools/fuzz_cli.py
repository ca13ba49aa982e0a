#!/usr/bin/env python3
"""Differential fuzz of the C host program on the GPU box: random `add` command lines (-r, -d, -a, -endo, -t N device
threads on the one GPU) - the found file and the status counters must equal what the oracle's restatement of cmd_add
gives for the same range (job loop, overrun, strides, endo multiplier).  Complements tools/fuzz_gpu.py (library calls)
and tests/test_cli.py::test_scan_plan_matches_the_oracles_job_loop (arithmetic only).
usage: python tools/fuzz_cli.py [seconds=150] [seed=1]      -> gpurun_out/fuzz_cli.txt"""
import os
import random
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from ecloop_amd.build import build_host_cli  # noqa: E402
from synth import synth_bloom_words, write_blf  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    r = random.Random(seed)
    cli = build_host_cli()
    tmp = tempfile.mkdtemp(prefix="eclfuzzcli")
    ones_w = np.full(64, 0xFFFFFFFFFFFFFFFF, np.uint64)
    half_w = synth_bloom_words(70001, 23, "a")
    dense_w = synth_bloom_words(4099, 11, "a|b")
    files = {}
    for name, w in (("ones", ones_w), ("half", half_w), ("dense", dense_w)):
        files[name] = (os.path.join(tmp, name + ".blf"), orc.OrcFilter(bloom_words=w))
        write_blf(files[name][0], w)
    out = os.path.join(tmp, "found.txt")
    t_end, trials, lines_total = time.time() + budget, 0, 0
    while time.time() < t_end:
        offs = r.choice([0, 0, 0, 1, 5, 64, 128, 200])
        a33, a65 = r.choice([(True, False), (False, True), (True, True)])
        endo = r.random() < 0.4
        nthreads = r.choice([1, 1, 2, 3, 8])
        if offs:  # every job of a strided scan hashes 2^21 keys: sparse filter, one hash per key
            fname, endo = "half", False
            a33, a65 = (True, False) if a33 else (False, True)
            keys = r.choice([1, 2047, 5000, (1 << 21) - 1, (1 << 21) + 1])
            a = r.randrange(1 << (offs + 33), 1 << min(offs + 60, 250))
            b = a + keys * (1 << offs) + (r.randrange(1 << offs) if r.random() < 0.5 else 0)
        else:
            fname = r.choice(["ones", "dense", "half"])
            keys = r.choice([1, 2, 2047, 2048, 2049, 6000, 20000]) if fname == "ones" else r.choice([1, 5000, 1 << 20, (1 << 21) + 5, 3 * (1 << 21) + 77])
            a = r.choice([0x801, 0x8000, r.randrange(1 << 33, 1 << 200)])
            b = a + keys
        path, flt = files[fname]
        args = ["add", "-f", path, "-r", f"{a:x}:{b:x}", "-a", ("c" if a33 else "") + ("u" if a65 else ""), "-t", str(nthreads), "-q", "-o", out]
        if offs:
            args += ["-d", f"{offs}:32"]
        if endo:
            args.append("-endo")
        if os.path.exists(out):
            os.unlink(out)
        env = dict(os.environ, ECLOOP_HIP_SHARE_GPU=str(nthreads)) if nthreads > 1 else None
        pr = subprocess.run([cli] + args, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1]
        got = sorted(l.rstrip("\n") for l in open(out)) if os.path.exists(out) else []
        bits = max(20, b.bit_length())
        eff = min(offs, max(1, bits - min(bits, 32))) if offs else 0
        rc, recs, n, checked, hashed = orc.add_range(flt, a, b, a33=a33, a65=a65, endo=endo, offs=eff, threads=8, verify=False, cap=1 << 19)
        want = sorted(orc.found_lines(recs, n))
        clean = lambda s: int("".join(c for c in s if c.isdigit()) or 0)
        sf, sc = (clean(x) for x in status.split("~")[-1].split("/")) if "/" in status else (-1, -1)
        if pr.returncode != 0 or rc != 0 or got != want or (sf, sc) != (n, checked):
            print("MISMATCH", " ".join(args), "rc", pr.returncode, "lines", len(got), len(want), "status", (sf, sc), "oracle", (n, checked))
            print(pr.stderr.decode(errors="replace")[-400:])
            sys.exit(1)
        trials, lines_total = trials + 1, lines_total + len(got)
    line = "# tools/fuzz_cli.py %s %d: %d command lines, %d found lines and every status counter EQUAL to the oracle's cmd_add" % (budget, seed, trials, lines_total)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "fuzz_cli.txt"), "w").write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
