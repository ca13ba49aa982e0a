#!/usr/bin/env python3
"""Sustained run (GPU box): the C host program over 2^LOG2 keys (default 2^40, ~90 s) with the bench's 54 MB .blf.
Every bloom hit (expected: keys x 0.371^20 ~ 2.4e-9 per key) is re-derived on the device path that shares no kernel
with the walk (pk_verify_hash, ecl_hip_verify): one wrong hash160 among the hits ends the run with exit status 1.
Reports the sustained rate.  THREADS > 1 runs that many device threads on the one GPU (ECLOOP_HIP_SHARE_GPU): the
shared-counter hand-out of the multi-GPU path under load.
usage: python tools/soak.py [LOG2] [THREADS]"""
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
cli = build_host_cli()
tmp = tempfile.mkdtemp(prefix="eclsoak")
import atexit, shutil
atexit.register(shutil.rmtree, tmp, ignore_errors=True)
blf, out = os.path.join(tmp, "bench.blf"), os.path.join(tmp, "found.txt")
d = Device(0)
size, offs, _ = bench.build_filter(d, bench.RANGE_A, 1 << 32)
blf_save(blf, d.get_bloom(size))
d.close()
a = 1 << 44
t0 = time.time()
env = dict(os.environ, ECLOOP_HIP_SHARE_GPU=str(nthreads)) if nthreads > 1 else None
pr = subprocess.run([cli, "add", "-f", blf, "-r", "%x:%x" % (a, a + (1 << lg) - 1), "-t", str(nthreads), "-q", "-o", out], stdin=subprocess.DEVNULL,
                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
dt = time.time() - t0
status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
n = sum(1 for _ in open(out)) if os.path.exists(out) else 0
exp = (1 << lg) * 0.371 ** 20
rep = ["# tools/soak.py %d %d: ecloop-hip add -r %x:+2^%d -t %d, 54 MB .blf, every hit re-derived by the independent device path" % (lg, nthreads, a, lg, nthreads),
       "exit status %d, wall %.1f s" % (pr.returncode, dt), "status: " + status,
       "hits %d (expected false positives at density 0.371: %.0f)" % (n, exp)]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "soak.txt" if nthreads == 1 else "soak_t%d.txt" % nthreads), "w").write("\n".join(rep) + "\n")
print("\n".join(rep))
sys.exit(pr.returncode)
