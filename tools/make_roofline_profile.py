#!/usr/bin/env python3
"""Turns the raw output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into profiles/<tag>_roofline.json:
the measured, build-specific inputs of bench.py's `roofline` object.  Nothing in that object is a constant typed
into bench.py: it is either measured in the timed process (kernel time, keys) or loaded from this file, whose
`source_sha256` / `fingerprint` say which build the counters were taken on.

usage: python tools/make_roofline_profile.py gpurun_out/prof_r02 r02 > profiles/r02_roofline.json"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_mix  # noqa: E402
from ecloop_amd.build import source_sha256  # noqa: E402

KEYS = 1 << 32  # one bench launch (bench.py default: 2^32 keys in one device call)
SIMDS = 256 * 4


def read_pmc(path):
    """PMC <kernel> <counter> <sum> <dispatches> / TRACE <kernel> <ns> lines of the k_add<addr33> passes"""
    pmc, trace = {}, []
    for line in open(path):
        f = line.split()
        if not f:
            continue
        if f[0] == "PMC" and "k_add" in f[1]:
            pmc[f[2]] = {"sum": float(f[3]), "dispatches": int(f[4])}
        elif f[0] == "TRACE" and "k_add" in f[1]:
            trace.append(int(f[2]))
    return pmc, trace


def read_calib(path):
    """-> {kernel: {"requests", "bytes", "FETCH_SIZE": [KB per launch...], "WRITE_SIZE": [...]}}"""
    out, order, counter = {}, [], None
    rows = {"FETCH_SIZE": [], "WRITE_SIZE": []}
    for line in open(path):
        f = line.split()
        if not f:
            continue
        if f[0] == "#":
            counter = f[2]
            order = []
        elif f[0] == "CALIB":
            out.setdefault(f[1], {"requests": int(f[2]), "bytes": int(f[3])})
            order.append(f[1])
        elif f[0] == "PMCROW" and f[2] in rows:
            rows[f[2]].append((f[1], float(f[3])))
    # dispatch order of the calibration program: stream16_rd, stream16_wr, random8 x2 (54 MB), random8 x2 (5.9 GB)
    names = ["stream16_rd", "stream16_wr", "random8_rd_54MB", "random8_rd_54MB", "random8_rd_5900MB", "random8_rd_5900MB"]
    for c, rr in rows.items():
        rr = [v for k, v in rr if "stream16" in k or "random8" in k]
        for name, v in zip(names, rr):
            out.setdefault(name, {}).setdefault(c, []).append(v)
    return out


def ubench_rows(path):
    rows = {}
    for line in open(path):
        m = re.match(r"(.{36})\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
        if m:
            rows[m.group(1).strip()] = float(m.group(3))
        m = re.match(r"(.{44})\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
        if m and "only" in m.group(1) or (m and "mix" in m.group(1)):
            rows["cal: " + m.group(1).strip()] = {"clock_ghz": float(m.group(2)), "cycles": float(m.group(3))}
    return rows


def main():
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r02")
    pmc, trace = read_pmc(os.path.join(d, "pmc.txt"))
    big = [t for t in trace if t > 50e6]  # the 2^32-key launches of the passes (ns)
    ms = sum(big) / len(big) / 1e6 if big else None
    v = lambda k: pmc[k]["sum"] if k in pmc else None
    der = {}
    if v("SQ_INSTS_VALU"):
        der["valu_wave_instr_per_64_keys"] = v("SQ_INSTS_VALU") / (KEYS / 64)
        der["valu_lane_ops_per_key"] = der["valu_wave_instr_per_64_keys"]
    if v("GRBM_GUI_ACTIVE") and ms:
        der["clock_ghz"] = v("GRBM_GUI_ACTIVE") / 8 / (ms * 1e-3) / 1e9  # the counter sums the 8 XCDs
        if v("SQ_INSTS_VALU"):
            der["simd_cycles_per_valu_instr"] = der["clock_ghz"] * 1e9 * (ms * 1e-3) * SIMDS / v("SQ_INSTS_VALU")
    for k, name in (("VALUBusy", "valu_busy_pct"), ("VALUUtilization", "valu_lane_utilization_pct")):
        if k in pmc:
            der[name] = pmc[k]["sum"] / max(pmc[k]["dispatches"], 1)
    for k, name in (("SQ_INSTS_VMEM_RD", "vmem_rd"), ("SQ_INSTS_VMEM_WR", "vmem_wr"), ("SQ_INSTS_SMEM", "smem"), ("SQ_INSTS_LDS", "lds"),
                    ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_VALU_INT32", "valu_int32"), ("SQ_INSTS_VALU_INT64", "valu_int64")):
        if v(k) is not None:
            der[name + "_wave_instr_per_64_keys"] = v(k) / (KEYS / 64)
    cal = read_calib(os.path.join(d, "calib.txt")) if os.path.exists(os.path.join(d, "calib.txt")) else {}
    calib = {}
    for name, c in cal.items():
        e = {"requests": c.get("requests"), "algorithmic_bytes": c.get("bytes")}
        for cn in ("FETCH_SIZE", "WRITE_SIZE"):
            if c.get(cn):
                kb = c[cn][-1]  # the last launch of the kind (caches warm for the small array)
                e[cn + "_KB"] = kb
                e[cn + "_bytes_per_request"] = kb * 1024 / c["requests"] if c.get("requests") else None
        calib[name] = e
    # corrections (guide: FETCH_SIZE counts half the bytes of a wide coalesced read; everything else: calibrate)
    corr = {}
    s = calib.get("stream16_rd", {})
    if s.get("FETCH_SIZE_KB"):
        corr["fetch_stream16_reported_over_actual"] = s["FETCH_SIZE_KB"] * 1024 / s["algorithmic_bytes"]
    s = calib.get("stream16_wr", {})
    if s.get("WRITE_SIZE_KB"):
        corr["write_stream16_reported_over_actual"] = s["WRITE_SIZE_KB"] * 1024 / s["algorithmic_bytes"]
    for n in ("random8_rd_54MB", "random8_rd_5900MB"):
        s = calib.get(n, {})
        if s.get("FETCH_SIZE_bytes_per_request") is not None:
            corr["fetch_reported_bytes_per_" + n] = s["FETCH_SIZE_bytes_per_request"]
    traffic = {}
    if v("FETCH_SIZE") is not None and v("WRITE_SIZE") is not None:
        fr, wr = v("FETCH_SIZE") * 1024 / KEYS, v("WRITE_SIZE") * 1024 / KEYS
        traffic = {"fetch_bytes_per_key_reported": fr, "write_bytes_per_key_reported": wr}
        # The chain is read and written as coalesced 16-byte-per-lane streams: 18 B/key each way (36 B per element,
        # one element per two keys).  Its share of the counters follows from the streaming calibration; what is left
        # of FETCH_SIZE are the bloom probes (+ the lane centres and spills, < 0.2 B/key).
        cf = corr.get("fetch_stream16_reported_over_actual", 0.5)
        cw = corr.get("write_stream16_reported_over_actual", 1.0)
        chain = 18.0
        probes_reported = max(fr - chain * cf, 0.0)
        traffic.update({"chain_bytes_per_key_each_way": chain, "probe_fetch_bytes_per_key_reported": probes_reported,
                        "fetch_bytes_per_key_corrected": chain + probes_reported, "write_bytes_per_key_corrected": wr / cw if cw else wr,
                        "bytes_per_key_corrected": chain + probes_reported + (wr / cw if cw else wr),
                        "note": "reported = counter x 1024 / keys; corrected = chain at its algorithmic 18 B/key each way "
                                "(streaming calibration factor applied to take its share out of FETCH_SIZE) + the probe share as reported"})
    ub = {}
    for w in ("8", "4"):
        p = os.path.join(d, "ubench%s.txt" % w)
        if os.path.exists(p):
            ub["waves_per_simd_" + w] = ubench_rows(p)
    a = isa_mix.analyse()
    out = {"tag": tag, "kernel": "k_add<addr33>", "workload": "bench.py default: one 2^32-key launch, 54 MB .blf",
           "collected_by": "tools/collect_profiles.sh + tools/make_roofline_profile.py (rocprofv3 --kernel-trace --pmc, one counter set per pass)",
           "source_sha256": source_sha256(), "fingerprint": a.get("fingerprint"), "per_key_static": a.get("per_key_static"),
           "vgpr": a.get("vgpr"), "scratch_bytes": a.get("scratch_bytes_own"),
           "profiled_launch_ms": ms, "keys_per_launch": KEYS, "pmc": {k: pmc[k]["sum"] for k in sorted(pmc)},
           "derived": der, "traffic": traffic, "calibration": calib, "corrections": corr, "ubench_cycles_per_wave_instr": ub}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
