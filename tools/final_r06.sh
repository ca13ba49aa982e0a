#!/bin/bash
# round 6, last GPU session on the final build: differential fuzz (add path, host program, mul), the 2^41-key soak, filter bring-up by size
# -> gpurun_out/r06_{fuzz,fuzz_cli,fuzz_mul,soak,bringup}.txt
cd "$(dirname "$0")/.."
python tools/fuzz_gpu.py 150 61 > /dev/null 2>&1; cp gpurun_out/fuzz.txt gpurun_out/r06_fuzz.txt
python tools/fuzz_gpu.py 100 62 > /dev/null 2>&1; cat gpurun_out/fuzz.txt >> gpurun_out/r06_fuzz.txt
python tools/fuzz_cli.py 150 63 > /dev/null 2>&1; cp gpurun_out/fuzz_cli.txt gpurun_out/r06_fuzz_cli.txt
python tools/fuzz_mul_gpu.py 240 64 > /dev/null 2>&1; cp gpurun_out/fuzz_mul.txt gpurun_out/r06_fuzz_mul.txt
python tools/fuzz_mul_gpu.py 120 65 > /dev/null 2>&1; cat gpurun_out/fuzz_mul.txt >> gpurun_out/r06_fuzz_mul.txt
python tools/soak.py 41 1 > gpurun_out/r06_soak.txt 2>&1
{ python tools/bringup_timing.py 54; python tools/bringup_timing.py 5900; } > gpurun_out/r06_bringup.txt 2>&1
tail -3 gpurun_out/r06_fuzz.txt gpurun_out/r06_fuzz_cli.txt gpurun_out/r06_fuzz_mul.txt gpurun_out/r06_soak.txt; grep -E "set_bloom|==" gpurun_out/r06_bringup.txt | head -20
