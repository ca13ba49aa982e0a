#!/usr/bin/env python3
"""Throughput of the REFERENCE's host program bound to the library (oracle/_ref/ecloop_gpu: vladkens/ecloop main.c + the six one-line
edits of oracle/ref_binding/build_ecloop_gpu.py), by the reference's own status line (main.c:134-172):

  add -r 100000000:4ffffffff   (2^34 keys, the bench's 54 MB .blf)
    - MAX_JOB_SIZE unchanged (2^21 keys per ecl_hip_add_range call, main.c:16), -t 1 and -t N on N contexts (ECLOOP_GPU_CONTEXTS)
    - one #define changed: MAX_JOB_SIZE 2^30 (oracle/_ref/ecloop_gpu_j30, built with --job-log2 30), -t 1 and -t 2
  and the found sets of all runs against the C host program of this repository on the same range.
-> gpurun_out/<tag>_ref_binding.txt (copy to profiles/).  usage: python tools/bench_ref_binding.py [--log2 34] [--tag r05]"""
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
from ecloop_amd.build import build_host_cli, source_sha256  # noqa: E402
from ecloop_amd.engine import blf_save  # noqa: E402


def run(cmd, out, env=None):
    if os.path.exists(out):
        os.unlink(out)
    t0 = time.time()
    pr = subprocess.run(cmd + ["-q", "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, stdin=subprocess.DEVNULL,
                        env=dict(os.environ, **(env or {})), timeout=900)
    dt = time.time() - t0
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-1000:]
    status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
    lines = sorted(l.rstrip("\n") for l in open(out)) if os.path.exists(out) else []
    return lines, status, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=34)
    ap.add_argument("--tag", default="r06")
    a = ap.parse_args()
    ref = os.path.join(ROOT, "oracle", "_ref")
    cli = build_host_cli()
    tmp = tempfile.mkdtemp(prefix="eclbind")
    import atexit
    import shutil
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    blf = os.path.join(tmp, "bench.blf")
    d = Device(0)
    size, offs, _ = bench.build_filter(d, bench.RANGE_A, 1 << 32, bench.FILTER_N)
    blf_save(blf, d.get_bloom(size))
    d.close()
    rng = "%x:%x" % (bench.RANGE_A, bench.RANGE_A + (1 << a.log2) - 1)
    rep = ["# tools/bench_ref_binding.py: the reference's host program on the library, add -r %s (2^%d keys), .blf of %d entries (%.0f MB), rates by the reference's status line"
           % (rng, a.log2, bench.FILTER_N, size * 8 / 1e6), "# source_sha256 %s" % source_sha256()]
    want, ws, wt = run([cli, "add", "-f", blf, "-r", rng], os.path.join(tmp, "cli.txt"))
    rep.append("%-78s: %3d lines %s  wall %5.1f s  status: %s" % ("ecloop-hip (this repository's C host program)", len(want), hashlib.sha256("\n".join(want).encode()).hexdigest()[:12], wt, ws))
    ok = True
    legs = [("ecloop_gpu", "MAX_JOB_SIZE 2^21 (unchanged), -t 1", ["-t", "1"], {}),
            ("ecloop_gpu", "2^21, -t 1, scan end not told (ECLOOP_GPU_NO_SCAN_END=1)", ["-t", "1"], {"ECLOOP_GPU_NO_SCAN_END": "1"}),
            ("ecloop_gpu", "2^21, -t 1, sweeps of 2^32 keys (ECL_HIP_LOOKAHEAD_LOG2=32)", ["-t", "1"], {"ECL_HIP_LOOKAHEAD_LOG2": "32"}),
            ("ecloop_gpu", "2^21, -t 1, look-ahead OFF (ECL_HIP_LOOKAHEAD_LOG2=0: round 5's library)", ["-t", "1"], {"ECL_HIP_LOOKAHEAD_LOG2": "0"}),
            ("ecloop_gpu", "2^21, -t 2, ECLOOP_GPU_CONTEXTS=2", ["-t", "2"], {"ECLOOP_GPU_CONTEXTS": "2"}),
            ("ecloop_gpu", "2^21, -t 4, ECLOOP_GPU_CONTEXTS=4", ["-t", "4"], {"ECLOOP_GPU_CONTEXTS": "4"}),
            ("ecloop_gpu", "2^21, -t 8, ECLOOP_GPU_CONTEXTS=8", ["-t", "8"], {"ECLOOP_GPU_CONTEXTS": "8"}),
            ("ecloop_gpu", "2^21, -t 8, ECLOOP_GPU_CONTEXTS=8, look-ahead OFF", ["-t", "8"], {"ECLOOP_GPU_CONTEXTS": "8", "ECL_HIP_LOOKAHEAD_LOG2": "0"}),
            ("ecloop_gpu_j30", "MAX_JOB_SIZE 2^30 (one #define), -t 1", ["-t", "1"], {})]
    for binary, what, extra, env in legs:
        path = os.path.join(ref, binary)
        if not os.path.exists(path):
            rep.append("%-78s: binary missing (built by __graft_entry__.build() where /root/reference exists)" % (binary + ", " + what))
            ok = False
            continue
        lines, status, dt = run([path, "add", "-f", blf, "-r", rng] + extra, os.path.join(tmp, "ref.txt"), env)
        same = lines == want
        ok &= same
        rep.append("%-78s: %3d lines %s  wall %5.1f s  status: %s  found set %s" % (binary + ", " + what, len(lines), hashlib.sha256("\n".join(lines).encode()).hexdigest()[:12], dt, status,
                                                                                   "identical" if same else "DIFFERS"))
    # `mul`: the reference's reader (fgets, 2048-line jobs, main.c:542-576) and workers (hex parse, main.c:503-527) unchanged, each job ONE
    # ecl_hip_mul_batch call of 2048 scalars: bounded by the reference's own line reader and by the latency of such small calls, stated for
    # completeness (the library's rate needs calls of 2^20 scalars and more: ecloop-hip's front end, profiles/*_mul_cli.txt)
    gen = os.path.join(tmp, "gen_hex_lines")
    subprocess.run(["gcc", "-O2", "-pthread", os.path.join(ROOT, "tools", "gen_hex_lines.c"), "-o", gen], check=True)
    lines_log2 = 24
    src = os.path.join(tmp, "mul_in.txt")
    subprocess.run([gen, str(1 << lines_log2), "7", src, "16"], check=True)

    def run_mul(cmd, env=None):
        out = os.path.join(tmp, "mul_out.txt")
        if os.path.exists(out):
            os.unlink(out)
        t0 = time.time()
        pr = subprocess.run(cmd + ["-q", "-o", out], stdin=open(src, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})), timeout=900)
        dt = time.time() - t0
        assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-1000:]
        status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
        return sorted(l.rstrip("\n") for l in open(out)) if os.path.exists(out) else [], status, dt

    rep.append("# mul -a cu, 2^%d lines of 64 hex digits from a file on stdin, same .blf (rates by each program's status line)" % lines_log2)
    mwant, ms, mt = run_mul([cli, "mul", "-f", blf, "-a", "cu"])
    rep.append("%-78s: %3d lines  wall %5.1f s  status: %s" % ("ecloop-hip mul (batches of records straight from the file)", len(mwant), mt, ms))
    for t in (1, 4, 8):
        path = os.path.join(ref, "ecloop_gpu")
        if os.path.exists(path):
            lines, status, dt = run_mul([path, "mul", "-f", blf, "-a", "cu", "-t", str(t)], {"ECLOOP_GPU_CONTEXTS": str(min(t, 8))})
            # (the reference's tail batch emits stale slots beyond the last line, main.c:467: compare the sets of real keys' lines)
            same = set(lines) >= set(mwant) and len(set(lines) - set(mwant)) <= 16
            ok &= same
            rep.append("%-78s: %3d lines  wall %5.1f s  status: %s  found set %s" % ("ecloop_gpu mul, reference reader + 2048-line jobs, -t %d on %d contexts" % (t, min(t, 8)),
                                                                                   len(lines), dt, status, "identical" if same else "DIFFERS"))
    cpu = os.path.join(ref, "ecloop_native")
    if os.path.exists(cpu):
        try:
            small = os.path.join(tmp, "mul_small.txt")
            open(small, "wb").write(open(src, "rb").read(65 << 21))
            src_full, src = src, small
            lines, status, dt = run_mul([cpu, "mul", "-f", blf, "-a", "cu", "-t", "64"])
            rep.append("%-78s: %3d lines  wall %5.1f s  status: %s" % ("the unmodified reference on the host cores, -t 64, the first 2^21 lines", len(lines), dt, status))
            src = src_full
        except Exception as e:  # the native build may not run on this host
            rep.append("the unmodified reference (ecloop_native) did not run here: %s" % str(e)[:80])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "%s_ref_binding.txt" % a.tag), "w").write("\n".join(rep) + "\n")
    print("\n".join(rep))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
