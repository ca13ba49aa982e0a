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
    """-> ({counter: (sum over the matching kernels' dispatches, dispatches)}, {kernel: [ns per dispatch]}, passes,
           [(pass, ns, counter, value)] per dispatch where the collection wrote DISPATCH rows)"""
    pmc, trace, passes, rows = {}, {}, 0, []
    for line in open(path):
        f = line.split()
        if line.startswith("# --pmc"):
            passes += 1
        if f and f[0] == "PMC":
            s, n = pmc.get(f[2], (0.0, 0))
            pmc[f[2]] = (s + float(f[3]), n + int(f[4]))
        if f and f[0] == "TRACE":
            trace.setdefault(f[1], []).append(int(f[2]))
        if f and f[0] == "DISPATCH":
            rows.append((passes, int(f[2]), f[3], float(f[4])))
    return pmc, trace, passes, rows


def full_size(rows, min_ms):
    """time-weighted figures over the dispatches that last at least min_ms (the full-size pieces of a `mul` call: the ramp-up pieces at the
    head of a call and the set-up launches cannot fill the chip; an unweighted mean over all of them says nothing - round-4 review)"""
    sel = [(ns, c, v) for _, ns, c, v in rows if ns >= min_ms * 1e6]
    out = {"min_ms": min_ms, "dispatches": sum(1 for ns, c, v in sel if c == "SQ_INSTS_VALU")}
    tot = lambda name: sum(v for ns, c, v in sel if c == name)
    ns_of = lambda name: sum(ns for ns, c, v in sel if c == name)
    if tot("SQ_INSTS_VALU") and tot("GRBM_GUI_ACTIVE"):
        out["clock_ghz"] = tot("GRBM_GUI_ACTIVE") / 8 / (ns_of("GRBM_GUI_ACTIVE") * 1e-9) / 1e9
        out["simd_cycles_per_valu_instr"] = tot("GRBM_GUI_ACTIVE") / 8 * SIMDS / tot("SQ_INSTS_VALU")
    if ns_of("VALUBusy"):
        out["valu_busy_pct_time_weighted"] = sum(v * ns for ns, c, v in sel if c == "VALUBusy") / ns_of("VALUBusy")
    return out


def profile(path, tag, kernel, workload, units, unit_name, min_ms=None):
    pmc, trace, passes, rows = read(path)
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
    if min_ms and rows:
        # a call of many unequal launches: the per-instruction and busy figures come from the full-size ones only, weighted by time
        fs = full_size(rows, min_ms)
        out["full_size_dispatches"] = fs
        for k in ("clock_ghz", "simd_cycles_per_valu_instr"):
            if k in fs:
                d[k] = fs[k]
        if "valu_busy_pct_time_weighted" in fs:
            d["valu_busy_pct"] = fs["valu_busy_pct_time_weighted"]
    elif "VALUBusy" in pmc:
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
             "bench.py --cmd mul --steps 1 --warmup 1: 3 calls of 2^24 scalars from page-locked host memory, -a cu, 26-bit window table", 3 * (1 << 24), "scalar", 1.0),
            ("pmc_cu_endo.txt", "roofline_cu_endo", "k_add<addr33,addr65,endo>",
             "bench.py --addr cu --endo --filter-n 1100000000 --keys-log2 30: one 2^30-key launch, 12 hash160 per key, 5.9 GB filter", 1 << 30, "key", None)]
    for src, kind, kernel, workload, units, unit, min_ms in jobs:
        p = os.path.join(d, src)
        if not os.path.exists(p):
            continue
        out = profile(p, tag, kernel, workload, units, unit, min_ms)
        if kind == "roofline_mul":  # the class mix of the kernel the counters were taken on (bench.py: `mix_ceiling` of the mul leg)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import isa_mix
                m = isa_mix.analyse_mul()
                out["static_mix"] = {"per_scalar_static": m["per_scalar_static"], "fingerprint": m["fingerprint"], "windows": m["windows"]}
            except Exception as e:
                print("static mix of k_mul_check unavailable:", e, file=sys.stderr)
        json.dump(out, open(os.path.join(d, f"{tag}_{kind}.json"), "w"), indent=1)
        print(kind, json.dumps(out["derived"]), file=sys.stderr)


if __name__ == "__main__":
    main()
