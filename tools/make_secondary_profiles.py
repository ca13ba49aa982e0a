#!/usr/bin/env python3
"""PMC passes of the non-headline kernels (tools/collect_profiles.sh: pmc_mul.txt, pmc_cu_endo.txt in gpurun_out/prof_<tag>/)
-> <tag>_roofline_mul.json, <tag>_roofline_cu_endo.json: VALU lane-operations and reported HBM bytes per unit (scalar / key)
of the build they were taken on.  bench.py multiplies them with the rate it measures (`secondary.*.roofline`).

usage: python tools/make_secondary_profiles.py gpurun_out/prof_r04 r04"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecloop_amd.build import source_sha256  # noqa: E402

SIMDS = 256 * 4


def read(path):
    """-> ({counter: (sum over the matching kernels' dispatches, dispatches)}, {kernel: [ns per dispatch]}, passes)"""
    pmc, trace, passes = {}, {}, 0
    for line in open(path):
        f = line.split()
        if line.startswith("# --pmc"):
            passes += 1
        if f and f[0] == "PMC":
            s, n = pmc.get(f[2], (0.0, 0))
            pmc[f[2]] = (s + float(f[3]), n + int(f[4]))
        if f and f[0] == "TRACE":
            trace.setdefault(f[1], []).append(int(f[2]))
    return pmc, trace, passes


def profile(path, tag, kernel, workload, units, unit_name):
    pmc, trace, passes = read(path)
    ns = sum(sum(v) for v in trace.values())
    out = {"tag": tag, "kernel": kernel, "workload": workload, "source_sha256": source_sha256(),
           "collected_by": "tools/collect_profiles.sh + tools/make_secondary_profiles.py (rocprofv3 --kernel-trace --pmc, one counter set per pass)",
           "units_per_pass": units, "unit": unit_name, "pmc": {k: v[0] for k, v in pmc.items()}, "dispatches": {k: v[1] for k, v in pmc.items()},
           "kernels": {k: {"dispatches_per_pass": len(v) // max(passes, 1), "ms_per_pass": sum(v) / max(passes, 1) / 1e6} for k, v in trace.items()},
           "derived": {}, "traffic": {}}
    d, t = out["derived"], out["traffic"]
    ms = ns / max(passes, 1) / 1e6
    if ns:
        d["kernel_ms_per_pass"] = ms
        d["kernel_m%ss_s" % unit_name] = units / ms / 1e3
    if "SQ_INSTS_VALU" in pmc:
        d["valu_lane_ops_per_" + unit_name] = pmc["SQ_INSTS_VALU"][0] * 64 / units  # wave instructions x 64 lanes
    if "GRBM_GUI_ACTIVE" in pmc and ms:
        # the counter sums the 8 XCDs, each counting while any of the pass's dispatches runs on it: busy clocks / kernel time
        d["clock_ghz"] = pmc["GRBM_GUI_ACTIVE"][0] / 8 / (ms * 1e-3) / 1e9
        if "SQ_INSTS_VALU" in pmc:
            d["simd_cycles_per_valu_instr"] = d["clock_ghz"] * 1e9 * (ms * 1e-3) * SIMDS / pmc["SQ_INSTS_VALU"][0]
    if "VALUBusy" in pmc:
        d["valu_busy_pct"] = pmc["VALUBusy"][0] / max(pmc["VALUBusy"][1], 1)
    if "TCP_PENDING_STALL_CYCLES" in pmc and pmc.get("TCP_GATE_EN1", (0, 0))[0]:
        d["tcp_pending_stall_pct"] = 100.0 * pmc["TCP_PENDING_STALL_CYCLES"][0] / pmc["TCP_GATE_EN1"][0]
    for c, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        if c in pmc:
            t[name + "_bytes_per_unit_reported"] = pmc[c][0] * 1024 / units
    if "fetch_bytes_per_unit_reported" in t and "write_bytes_per_unit_reported" in t:
        t["bytes_per_unit_reported"] = t["fetch_bytes_per_unit_reported"] + t["write_bytes_per_unit_reported"]
        t["note"] = ("counter x 1024 / units, as reported: FETCH_SIZE counts half of a wide coalesced read and one 64-byte request per random "
                     "8-byte or 64-byte access (profiles/*_fetch_calibration.txt)")
        if ms:
            t["reported_gbs"] = t["bytes_per_unit_reported"] * units / (ms * 1e-3) / 1e9
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    jobs = [("pmc_mul.txt", "roofline_mul", "mul kernels (k_mul*: window sums, hash160, probe)",
             "bench.py --cmd mul --steps 1 --warmup 1: 3 calls of 2^24 scalars from page-locked host memory, -a cu, 26-bit window table", 3 * (1 << 24), "scalar"),
            ("pmc_cu_endo.txt", "roofline_cu_endo", "k_add<addr33,addr65,endo>",
             "bench.py --addr cu --endo --filter-n 1100000000 --keys-log2 30: one 2^30-key launch, 12 hash160 per key, 5.9 GB filter", 1 << 30, "key")]
    for src, kind, kernel, workload, units, unit in jobs:
        p = os.path.join(d, src)
        if not os.path.exists(p):
            continue
        out = profile(p, tag, kernel, workload, units, unit)
        json.dump(out, open(os.path.join(d, f"{tag}_{kind}.json"), "w"), indent=1)
        print(kind, json.dumps(out["derived"]), file=sys.stderr)


if __name__ == "__main__":
    main()
