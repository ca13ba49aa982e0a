#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) run: per-kernel calls / total / average / percentage.
usage: tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_kernel_stats.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats, table top_kernels of", sys.argv[1])
print("%-110s %8s %16s %16s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for name, calls, total, avg, pct in c.execute("select * from top_kernels"):
    print("%-110s %8d %16.0f %16.0f %8.3f" % (name[:110], calls, total * 1e3, avg * 1e3, pct))
try:
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    want = [x for x in ("vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "grid_x", "workgroup_x") if x in cols]
    rows = list(c.execute("select name, count(*), %s from kernels group by name" % ", ".join("min(%s)" % w for w in want)))
    print("\n# per-kernel dispatch records:", ", ".join(want))
    for r in rows:
        print("%-110s calls=%d %s" % (r[0][:110], r[1], " ".join("%s=%s" % (w, v) for w, v in zip(want, r[2:]))))
except Exception as e:
    print("# (no register columns:", e, ")")
