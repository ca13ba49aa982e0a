#!/usr/bin/env python3
"""Sustained run (GPU box): the C host program over 2^LOG2 keys (default 2^40, ~90 s) with the bench's 54 MB .blf.
Every bloom hit (expected: keys x 0.371^20 ~ 2.4e-9 per key) is re-derived on the device path that shares no kernel
with the walk (pk_verify_hash, ecl_hip_verify): one wrong hash160 among the hits ends the run with exit status 1.
Reports the sustained rate.  THREADS > 1 runs that many device threads on the one GPU (ECLOOP_HIP_SHARE_GPU): the
shared-counter hand-out of the multi-GPU path under load.
ADDR / ENDO / FILTER_N select the configs[2] shape: `python tools/soak.py 36 1 cu endo 1100000000` = `add -a cu -endo` with
the ~5.9 GB filter (random fill at the design density, written to disk as a .blf).
usage: python tools/soak.py [LOG2] [THREADS] [ADDR=c] [endo|noendo] [FILTER_N=10000000]"""
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

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 1
addr = sys.argv[3] if len(sys.argv) > 3 else "c"
endo = len(sys.argv) > 4 and sys.argv[4] == "endo"
filter_n = int(sys.argv[5]) if len(sys.argv) > 5 else bench.FILTER_N
if filter_n > 50_000_000:
    import torch
    torch.cuda.init()  # the big filter is generated with torch: its HIP runtime has to come up before the library's
cli = build_host_cli()
tmp = tempfile.mkdtemp(prefix="eclsoak")
import atexit, shutil
atexit.register(shutil.rmtree, tmp, ignore_errors=True)
blf, out = os.path.join(tmp, "bench.blf"), os.path.join(tmp, "found.txt")
d = Device(0)
size, offs, _ = bench.build_filter(d, bench.RANGE_A, 1 << 32, filter_n)
blf_save(blf, d.get_bloom(size))
d.close()
a = 1 << 44
t0 = time.time()
env = dict(os.environ, ECLOOP_HIP_SHARE_GPU=str(nthreads)) if nthreads > 1 else None
extra = ["-a", addr] + (["-endo"] if endo else [])
pr = subprocess.run([cli, "add", "-f", blf, "-r", "%x:%x" % (a, a + (1 << lg) - 1), "-t", str(nthreads), "-q", "-o", out] + extra, stdin=subprocess.DEVNULL,
                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
dt = time.time() - t0
status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
n = sum(1 for _ in open(out)) if os.path.exists(out) else 0
per_key = len(addr) * (6 if endo else 1)
exp = (1 << lg) * per_key * (0.375 if filter_n > 50_000_000 else 0.371) ** 20
rep = ["# tools/soak.py %s: ecloop-hip add -r %x:+2^%d -t %d %s, %.0f MB .blf, every hit re-derived by the independent device path" % (
    " ".join(sys.argv[1:]), a, lg, nthreads, " ".join(extra), size * 8 / 1e6),
       "exit status %d, wall %.1f s" % (pr.returncode, dt), "status: " + status,
       "hits %d (expected false positives at the filter's density, %d hashes per key: %.0f)" % (n, per_key, exp)]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "soak.txt" if len(sys.argv) <= 2 else "soak_%s.txt" % "_".join(sys.argv[1:])), "w").write("\n".join(rep) + "\n")
print("\n".join(rep))
sys.exit(pr.returncode)
