# round 6: kernel + copy timeline of ecl_hip_mul_batch_raw calls (tools/raw_api_probe.py) -> gpurun_out/rawprof/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/rawprof
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/rawprof -o raw -- python $R/tools/raw_api_probe.py 24 2 ${1:-raw} ${2:-1} 2>&1 | grep -v "^[WE]2026"
