#!/usr/bin/env python3
"""Stress of `mul -raw -t 4` on one GPU (ECLOOP_HIP_SHARE_GPU=4): 200 target pass phrases, twice, in 4.8 M filler lines; every run
must find each target exactly twice.  usage: repro_raw_fanout.py [runs=20]   (LD_LIBRARY_PATH selects another library build)"""
import hashlib, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tmp = "/tmp/raw_fanout"
os.makedirs(tmp, exist_ok=True)
targets = ["correct horse battery staple %d" % i for i in range(200)]
keys = [int.from_bytes(hashlib.sha256(t.encode()).digest(), "big") for t in targets]
hs = [orc.hash160(*orc.point_of(k % orc.N), True) for k in keys]
open(tmp + "/targets.txt", "w").write("".join("".join("%08x" % w for w in h) + "\n" for h in hs))
fill = ["filler phrase %07d" % i for i in range(2_400_000)]
with open(tmp + "/phrases.txt", "w") as f:
    for block in (fill, targets, [""], fill, targets):
        f.write("\n".join(block) + "\n")
key_of = {"%064x" % k: i for i, k in enumerate(keys)}
bad = 0
for r in range(runs):
    out = tmp + "/out.txt"
    if os.path.exists(out):
        os.unlink(out)
    pr = subprocess.run([os.path.join(ROOT, "ecloop_amd/host/ecloop-hip"), "mul", "-raw", "-f", tmp + "/targets.txt", "-t", "4", "-q", "-o", out],
                        stdin=open(tmp + "/phrases.txt", "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, ECLOOP_HIP_SHARE_GPU="4"))
    c = collections.Counter(l.split("\t")[2].strip() for l in open(out))
    miss = sorted((key_of[k], n) for k, n in c.items() if n != 2) + [(i, 0) for k, i in key_of.items() if k not in c]
    if miss or pr.returncode:
        bad += 1
        print("run", r, "rc", pr.returncode, "targets not found twice:", miss[:40], flush=True)
print("runs", runs, "bad", bad)
