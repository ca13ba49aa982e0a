set -u
O=gpurun_out/s2; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
python bench.py --cmd mul --steps 5 --warmup 2 > $O/bench_mul.json 2> $O/bench_mul.err
python bench.py --cmd mul --steps 5 --warmup 2 --addr c > $O/bench_mul_c.json 2>> $O/bench_mul.err
bash tools/bench_mul_cli.sh 16777216 > $O/mul_cli.txt 2>&1
# set-up of 1 vs 8 handles (one GPU shared by 8 device threads), 54 MB filter and 5.9 GB-class is covered by bench below
python - > $O/setup.txt 2>&1 <<'PY'
import os, subprocess, sys, time, re
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from synth import synth_bloom_words, write_blf
from ecloop_amd.build import build_host_cli
cli = build_host_cli()
for mb in (54, 1024):
    f = "/tmp/setup_%d.blf" % mb
    write_blf(f, synth_bloom_words(mb * 131072 + 3, 9, "a"))
    for n in (1, 8):
        env = dict(os.environ, ECLOOP_HIP_SHARE_GPU=str(n))
        t0 = time.time()
        pr = subprocess.run([cli, "add", "-f", f, "-r", "100000000:1ffffffff", "-t", str(n), "-q", "-o", "/tmp/setup_out.txt"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        banner = [l for l in pr.stdout.decode().splitlines() if l.startswith("setup:")]
        status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
        print("filter %4d MB, %d device thread(s): wall %.2fs | %s | %s" % (mb, n, dt, banner[0] if banner else "?", status))
PY
python tools/full_range_parity.py --filter-n 1100000000 --endo-log2 26 > $O/parity_big.txt 2>&1
python bench.py --filter-n 1100000000 --no-cpu > $O/bench_bigfilter.json 2> $O/bench_bigfilter.err
python bench.py --filter-n 1100000000 --no-cpu --addr cu --endo --keys-log2 30 > $O/bench_bigfilter_cu_endo.json 2>> $O/bench_bigfilter.err
python bench.py --no-cpu --addr cu --endo --keys-log2 30 > $O/bench_cu_endo.json 2>> $O/bench_bigfilter.err
cat $O/tests.txt; cat $O/bench_mul.json $O/bench_mul_c.json; cat $O/mul_cli.txt; cat $O/setup.txt; tail -30 $O/parity_big.txt; cat $O/bench_bigfilter.json $O/bench_bigfilter_cu_endo.json $O/bench_cu_endo.json
