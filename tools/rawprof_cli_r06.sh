# round 6: HIP API + kernel + copy timeline of `ecloop-hip mul -raw` over 2^29 pass phrases -> gpurun_out/rawprof_cli/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rawprof_cli
gcc -O2 -pthread $R/tools/gen_phrases.c -o /tmp/gen_phrases && /tmp/gen_phrases $((1<<29)) 11 /dev/shm/mul_raw.txt 32
$R/ecloop_amd/host/ecloop-hip mul -raw -f $R/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt > /dev/null 2>&1
rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/rawprof_cli -o raw -- $R/ecloop_amd/host/ecloop-hip mul -raw -f $R/tests/golden/btc-bw-hash -a cu -q -o /tmp/o.txt < /dev/shm/mul_raw.txt > /dev/null 2>/tmp/e.txt
tr '\r' '\n' < /tmp/e.txt | grep -v "^[WE]2026" | tail -2
rm -f /dev/shm/mul_raw.txt
ls -la $R/gpurun_out/rawprof_cli
