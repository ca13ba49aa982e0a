set -u
O=gpurun_out/s4; mkdir -p $O
python -m pytest tests/test_gpu_blf.py tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
python - > $O/cli_bringup.txt 2>&1 <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from synth import synth_bloom_words, write_blf
from ecloop_amd.build import build_host_cli
cli = build_host_cli()
f = "/tmp/setup_54.blf"
write_blf(f, synth_bloom_words(54 * 131072 + 3, 9, "a"))
for n, rng in ((1, "100000000:1ffffffff"), (1, "100000000:1ffffffff"), (8, "100000000:1ffffffff"), (1, "100000000:10fffffff")):
    env = dict(os.environ, ECLOOP_HIP_SHARE_GPU=str(n), ECLOOP_HIP_STATS="1")
    t0 = time.time()
    pr = subprocess.run([cli, "add", "-f", f, "-r", rng, "-t", str(n), "-q", "-o", "/tmp/setup_out.txt"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    dt = time.time() - t0
    status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
    print("== -t %d -r %s: wall %.2fs | %s" % (n, rng, dt, status))
    print("\n".join(l for l in pr.stdout.decode().splitlines() if "bring-up" in l or l.startswith("setup")))
PY
bash tools/bench_rnd.sh 29 40 > $O/rnd_29.txt 2>&1
bash tools/bench_rnd.sh 32 10 > $O/rnd_32.txt 2>&1
bash tools/collect_profiles.sh r02 > $O/collect.log 2>&1
cat $O/tests.txt $O/cli_bringup.txt $O/rnd_29.txt $O/rnd_32.txt; tail -3 $O/collect.log; cat gpurun_out/prof_r02/r02_mul.json; grep TIME gpurun_out/prof_r02/calib.txt; grep -i "lsh\|ashr\|mul_u32_u24\|mul_lo_u32\|max_u32" gpurun_out/prof_r02/ubench8.txt
