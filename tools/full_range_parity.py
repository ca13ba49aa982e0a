#!/usr/bin/env python3
"""One-off evidence run (GPU box): the found lists of the HIP path and of the REFERENCE binary (oracle/_ref, built in
the dev container by oracle/Makefile) over the FULL 2^32-key range of the headline config, same .blf, plus an
`-a cu -endo` leg over 2^28 keys.  Writes gpurun_out/full_range_parity.txt (copied to profiles/ by hand).
usage: python tools/full_range_parity.py [--endo-log2 28]"""
import argparse
import hashlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ecloop_amd import Device  # noqa: E402
from ecloop_amd.build import build_host_cli  # noqa: E402
from ecloop_amd.engine import blf_save  # noqa: E402


def run(cmd, out):
    if os.path.exists(out):
        os.unlink(out)
    t0 = time.time()
    pr = subprocess.run(cmd + ["-q", "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL)
    dt = time.time() - t0
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-1000:]
    status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
    lines = sorted(l.rstrip("\n") for l in open(out)) if os.path.exists(out) else []
    return lines, status, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--endo-log2", type=int, default=28)
    ap.add_argument("--filter-n", type=int, default=bench.FILTER_N, help="1100000000: the ~6 GB filter of configs[2] (random fill at the design density)")
    ap.add_argument("--main-log2", type=int, default=32)
    a = ap.parse_args()
    ref = os.path.join(ROOT, "oracle", "_ref", "ecloop_sane")
    cli = build_host_cli()
    tmp = tempfile.mkdtemp(prefix="eclparity")
    import atexit, shutil
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    blf = os.path.join(tmp, "bench.blf")
    if a.filter_n > 50_000_000:  # the big filter is generated with torch on the GPU: torch's HIP runtime has to come up first
        import torch
        torch.cuda.init()
    d = Device(0)
    size, offs, _ = bench.build_filter(d, bench.RANGE_A, 1 << 32, a.filter_n)
    blf_save(blf, d.get_bloom(size))
    d.close()
    rep = ["# tools/full_range_parity.py: HIP path vs the reference binary, same .blf file (%d words = %.0f MB, %d entries + %d planted keys)" % (size, size * 8 / 1e6, a.filter_n, len(offs))]
    threads = str(min(os.cpu_count() or 1, 64))
    legs = [("add addr33, 2^%d keys" % a.main_log2, [], a.main_log2), ("add -a cu -endo, 2^%d keys" % a.endo_log2, ["-a", "cu", "-endo"], a.endo_log2)]
    ok = True
    for name, extra, lg in legs:
        rng = "%x:%x" % (bench.RANGE_A, bench.RANGE_A + (1 << lg) - 1)
        g, gs, gt = run([cli, "add", "-f", blf, "-r", rng] + extra, os.path.join(tmp, "gpu.txt"))
        r, rs, rt = run([ref, "add", "-f", blf, "-r", rng, "-t", threads] + extra, os.path.join(tmp, "ref.txt"))
        same = g == r
        ok &= same
        bound = os.path.join(ROOT, "oracle", "_ref", "ecloop_gpu")  # the reference's own host program on the library, MAX_JOB_SIZE unchanged
        b = bs = bt = None
        if os.path.exists(bound):
            b, bs, bt = run([bound, "add", "-f", blf, "-r", rng, "-t", "1"] + extra, os.path.join(tmp, "bound.txt"))
            ok &= b == r
        rep += ["", "== %s   -r %s" % (name, rng),
                "HIP       : %d lines, sha256(sorted) %s, wall %.1f s, status: %s" % (len(g), hashlib.sha256("\n".join(g).encode()).hexdigest()[:16], gt, gs),
                "reference : %d lines, sha256(sorted) %s, wall %.1f s (-t %s), status: %s" % (len(r), hashlib.sha256("\n".join(r).encode()).hexdigest()[:16], rt, threads, rs),
                "identical : %s" % same]
        if b is not None:
            rep += ["bound ref : %d lines, sha256(sorted) %s, wall %.1f s (the reference's main.c on the library, 2^21-key jobs, -t 1), status: %s" % (
                len(b), hashlib.sha256("\n".join(b).encode()).hexdigest()[:16], bt, bs), "identical : %s" % (b == r)]
        if not same:
            rep += ["only HIP: %s" % sorted(set(g) - set(r))[:5], "only reference: %s" % sorted(set(r) - set(g))[:5]]
        if not extra:
            rep += ["lines:"] + ["  " + l for l in g]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = "full_range_parity.txt" if a.filter_n == bench.FILTER_N else "full_range_parity_%dM.txt" % (a.filter_n // 1000000)
    open(os.path.join(ROOT, "gpurun_out", name), "w").write("\n".join(rep) + "\n")
    print("\n".join(rep))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
