set -u
O=gpurun_out/s3; mkdir -p $O
python -m pytest tests/test_cli.py tests/test_gpu_blf.py tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.txt
python tools/bringup_timing.py 54 > $O/bringup.txt 2>&1
ecloop_amd/csrc/tools/grpinv_bench > $O/grpinv.txt 2>&1
python bench.py --cmd mul --steps 5 --warmup 2 > $O/bench_mul.json 2> $O/bench_mul.err
python bench.py --cmd mul --steps 5 --warmup 2 --pageable > $O/bench_mul_pageable.json 2>> $O/bench_mul.err
python bench.py --cmd mul --steps 5 --warmup 2 --addr c > $O/bench_mul_c.json 2>> $O/bench_mul.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/mulstats -o m -- python $OLDPWD/bench.py --cmd mul --steps 3 --warmup 1 > $OLDPWD/$O/mulstats.log 2>&1
cd $OLDPWD
db=$(find $O/mulstats -name '*.db' | head -1); [ -n "$db" ] && python tools/rocprof_summary.py "$db" > $O/mul_kernel_stats.txt; rm -rf $O/mulstats
bash tools/bench_mul_cli.sh 16777216 > $O/mul_cli.txt 2>&1
python tools/full_range_parity.py --filter-n 1100000000 --endo-log2 26 > $O/parity_big.txt 2>&1
cat $O/tests.txt $O/bringup.txt $O/grpinv.txt $O/bench_mul.json $O/bench_mul_pageable.json $O/bench_mul_c.json; head -8 $O/mul_kernel_stats.txt; cat $O/mul_cli.txt; tail -40 $O/parity_big.txt
