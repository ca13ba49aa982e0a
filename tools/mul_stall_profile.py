#!/usr/bin/env python3
"""Where k_mul_check's issue slots go: wait / memory / translation counters PER DISPATCH (GPU box, run through gpurun).

  python tools/mul_stall_profile.py [--log2 24] [--windows 22,26] [--tag r05] [--min-ms 1.0]

One rocprofv3 pass per counter set (--kernel-trace --pmc only, never together with sys/hip/hsa tracing) over
`tools/bench_mul.py <log2> 3 <W> design` (three calls of 2^log2 scalars, -a cu, the design-density filter; the first call
builds the table).  Each k_mul_check dispatch is matched across passes by its position in the call sequence (the piece
schedule is deterministic), and only the full-size pieces (duration >= --min-ms in the trace of the same pass) are
reported: the ramp-up pieces at the head of a call cannot fill the chip and would only blur the ratios.
Ratios are sums over the selected dispatches (= time-weighted), never means of per-dispatch percentages.
-> gpurun_out/<tag>_mul_stall.txt (copy to profiles/)."""
import argparse
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecloop_amd.build import source_sha256  # noqa: E402

SETS = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE",
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS",
    "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32",
    "TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum",
    "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum",
    "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum",
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum",
    "FETCH_SIZE",
    "WRITE_SIZE",
    "VALUBusy",
]


def run_pass(out_dir, counters, cmd, env):
    shutil.rmtree(out_dir, ignore_errors=True)
    full = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out_dir, "-o", "p", "--"] + cmd
    pr = subprocess.run(full, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, cwd="/tmp")
    log = pr.stdout.decode(errors="replace")
    disp = collections.defaultdict(dict)  # dispatch id -> {counter: value}
    names = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            i = int(r["Dispatch_Id"])
            names[i] = r["Kernel_Name"]
            disp[i][r["Counter_Name"]] = disp[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    dur = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            names.setdefault(int(r["Dispatch_Id"]), r["Kernel_Name"])
    shutil.rmtree(out_dir, ignore_errors=True)
    rows = []  # k_mul_check dispatches in launch order: (ms, {counter: value})
    for i in sorted(names):
        if "k_mul_check" in names[i]:
            rows.append((dur.get(i, 0.0), disp.get(i, {})))
    return rows, pr.returncode, log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=24)
    ap.add_argument("--windows", default="22,26")
    ap.add_argument("--tag", default="r05")
    ap.add_argument("--min-ms", type=float, default=1.0)
    ap.add_argument("--label", default="")
    ap.add_argument("--passes", type=int, default=len(SETS), help="only the first N counter sets")
    ap.add_argument("--dump", action="store_true", help="also list every k_mul_check dispatch of the first pass (ms, VALU instructions, clock, wait share)")
    a = ap.parse_args()
    env = dict(os.environ, ECL_HIP_SKIP_SELFTEST="1", TMPDIR="/tmp")
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    rep = ["# tools/mul_stall_profile.py%s: k_mul_check<cu>, per dispatch, full-size pieces only (>= %.1f ms); calls of 2^%d scalars, design-density filter"
           % ((" [" + a.label + "]") if a.label else "", a.min_ms, a.log2),
           "# source_sha256 %s" % source_sha256(),
           "# one rocprofv3 --kernel-trace --pmc pass per counter set; sums over the selected dispatches (time-weighted), SQ cycle counters in quad-cycles"]
    for W in [int(w) for w in a.windows.split(",")]:
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_mul.py"), str(a.log2), "3", str(W), "design"]
        rep += ["", "== W = %d" % W]
        tot = collections.OrderedDict()
        shape = None
        for cs in SETS[: a.passes]:
            rows, rc, log = run_pass(os.path.join(out, "mulstall_tmp"), cs, cmd, env)
            full = [(ms, c) for ms, c in rows if ms >= a.min_ms]
            if rc != 0 or not full:
                rep.append("pass [%s]: FAILED rc %d, %d dispatches; %s" % (cs, rc, len(rows), log.strip().splitlines()[-1][:200] if log.strip() else ""))
                continue
            if shape is None:
                shape = [round(ms, 3) for ms, _ in rows]
                rep.append("dispatches per run: %d k_mul_check (ms each, first pass: %s)" % (len(rows), " ".join("%.2f" % m for m in shape)))
                if a.dump:
                    rep.append("  #   ms      VALU wave-instr   per ms (G)   clock GHz   WAIT_ANY/WAVE_CYCLES   WAIT_INST_ANY/WAVE_CYCLES   SIMD-clk per VALU")
                    for j, (ms, c) in enumerate(rows):
                        if not ms or "SQ_INSTS_VALU" not in c:
                            continue
                        ghz = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (ms * 1e-3) / 1e9
                        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
                        rep.append("  %2d %6.3f  %14.0f  %10.3f  %9.3f  %12.3f  %12.3f  %10.2f" % (
                            j, ms, c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] / ms / 1e6, ghz, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
                            c.get("GRBM_GUI_ACTIVE", 0) / 8 * 1024 / c["SQ_INSTS_VALU"]))
            ms_sum = sum(ms for ms, _ in full)
            line = "pass [%s]: %d full-size dispatches, %.3f ms" % (cs, len(full), ms_sum)
            rep.append(line)
            for name in cs.split():
                vals = [c.get(name) for _, c in full if name in c]
                if not vals:
                    rep.append("  %-36s (not reported)" % name)
                    continue
                s = sum(vals)
                tot[name] = (s, ms_sum, len(vals))
                rep.append("  %-36s sum %.6g  per-dispatch min %.6g max %.6g" % (name, s, min(vals), max(vals)))
        d = collections.OrderedDict()
        g = lambda n: tot[n][0] if n in tot else None
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if g(n) is not None:
                    d[n + " / SQ_WAVE_CYCLES"] = g(n) / wc
            if g("SQ_BUSY_CYCLES"):
                d["waves resident per SIMD (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / 4 SIMDs... as reported)"] = wc / g("SQ_BUSY_CYCLES")
        if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE sums the 8 XCDs' busy clocks; 1024 SIMDs
            d["SIMD-clocks per VALU instruction"] = g("GRBM_GUI_ACTIVE") / 8 * 1024 / g("SQ_INSTS_VALU")
            d["clock GHz (GRBM_GUI_ACTIVE / 8 / kernel time)"] = g("GRBM_GUI_ACTIVE") / 8 / (tot["GRBM_GUI_ACTIVE"][1] * 1e-3) / 1e9
        if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
            d["SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (other pass)"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
        if g("TCP_PENDING_STALL_CYCLES") and g("TCP_GATE_EN1"):
            d["TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1"] = g("TCP_PENDING_STALL_CYCLES") / g("TCP_GATE_EN1")
        if g("TCP_UTCL1_TRANSLATION_MISS_sum") is not None and g("TCP_UTCL1_REQUEST_sum"):
            d["UTCL1 translation miss / request"] = g("TCP_UTCL1_TRANSLATION_MISS_sum") / g("TCP_UTCL1_REQUEST_sum")
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum"):
            d["L2 hit rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
            d["L1->L2 read latency, cycles per request"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
        if g("VALUBusy") is not None:
            d["VALUBusy %, mean over full-size dispatches (derived metric; gfx94x formula)"] = g("VALUBusy") / tot["VALUBusy"][2]
        if g("FETCH_SIZE") is not None:
            d["FETCH_SIZE GB/s as reported"] = g("FETCH_SIZE") * 1024 / (tot["FETCH_SIZE"][1] * 1e-3) / 1e9
        rep.append("derived:")
        rep += ["  %-78s %.4f" % (k, v) for k, v in d.items()]
    path = os.path.join(out, "%s_mul_stall%s.txt" % (a.tag, ("_" + a.label) if a.label else ""))
    open(path, "w").write("\n".join(rep) + "\n")
    print("\n".join(rep))


if __name__ == "__main__":
    main()
