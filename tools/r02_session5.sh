set -u
O=gpurun_out/s5; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
python tools/fuzz_gpu.py 420 201 > $O/fuzz.log 2>&1; cp gpurun_out/fuzz.txt $O/fuzz.txt
bash tools/bench_rnd.sh 29 40 > $O/rnd_29.txt 2>&1
bash tools/bench_rnd.sh 32 10 > $O/rnd_32.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
ECL_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_share2.json 2> $O/bench_share2.err
cat $O/tests.txt; tail -5 $O/fuzz.txt; cat $O/rnd_29.txt $O/rnd_32.txt; cat $O/bench.json; tail -2 $O/bench_share2.json
