#!/usr/bin/env python3
"""bench.py — `add` (addr33) key-search throughput on MI355X: BASELINE.json's metric on its configs[1].

    python bench.py [--gpus N --steps K --warmup W]              (N>1: launched by torch.distributed.run)

One step = one pass of the hot path over a contiguous range of 2^32 private keys (per GPU) against a `.blf`-format
bloom filter resident in HBM: batch affine additions, SHA-256 -> RIPEMD-160 of every compressed public key, bloom
probe, hits gathered on the host.  Inputs are synthetic and already in HBM when the timed region starts: the
filter holds 10^7 seeded pseudo-random hash160 values plus 16 planted keys of the scanned range (so the found list
is not empty and is checked).  The keyspace is range-partitioned: rank r scans [A + r*2^32, A + (r+1)*2^32); there
is no collective on the data path, only the timing barrier (weak scaling).

Prints ONE JSON line (rank 0).  `roofline` prices the fused kernel against the integer-VALU issue peak measured
by ecloop_amd/csrc/tools/ubench.hip; `cpu_baseline` is the unmodified reference binary
(oracle/_ref, built from /root/reference by oracle/Makefile) timed on this host's cores over a bounded sample of
the same range and filter, or the oracle port if the binary is not there.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RANGE_A = 0x100000000  # configs[1] / SURVEY §8d: add -r 100000000:1ffffffff
FILTER_N = 10_000_000
PLANTED = 16
# Work per addr33 key, priced as SURVEY.md §8d prescribes: the static VALU instruction count per key from the gfx950
# assembly (tools/isa_mix.py over the hot blocks, weighted by trip count: 3042, plus the amortised inversion and
# candidate-ring drains; total taken from the PMC count SQ_INSTS_VALU: 3127 per key) times the issue cost of each class measured by the dependency-free microbenchmark (profiles/ubench_r01.txt,
# SIMD-cycles per wave-instruction at the nominal clock):
#   422 v_mad_u64_u32 x 4.61 + 606 double-rate VOP2 (add/sub/and/or/xor/mov) x 2.55 + 2099 other VALU x 4.23
#   = 12 370 SIMD-cycles per 64 keys = 3092 lane-cycles per key (16 lanes per SIMD-cycle).
# (SURVEY's estimate before any code existed: 313 IMAD + 350 ALU + 2570 ALU = 3233 ops, ~4.2 k lane-cycles.)
VALU_PER_KEY = {"mad64": 422, "fast_vop2": 606, "other": 2099}
ISSUE_CYCLES = {"mad64": 4.61, "fast_vop2": 2.55, "other": 4.23}
OPS_PER_KEY = sum(VALU_PER_KEY.values())
LANE_CYCLES_PER_KEY = sum(VALU_PER_KEY[k] * ISSUE_CYCLES[k] for k in VALU_PER_KEY) / 4.0
# peak: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T lane-cycles/s (one 64-wide VALU instruction per SIMD per 4 clocks)
PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12
# HBM-side bytes per key from the PMC passes (profiles/r01_pmc_traffic.txt: FETCH_SIZE 117.6 B + WRITE_SIZE 18.0 B per key:
# 1.8 64-byte bloom sectors per key + the 36 B / 2 keys prefix-product chain each way + spills). Not the bound: 1.5 TB/s.
TRAFFIC_BYTES_PER_KEY = 135.6
HBM_PEAK_GBS = 8000.0
VALU_BUSY_PCT = 99.0  # rocprofv3 --pmc VALUBusy on the 2^32-key launch (profiles/r01_pmc_valu.txt)


def splitmix_hashes(n, seed):
    with np.errstate(over="ignore"):
        i = np.arange(1, n * 3 + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    w = z.view(np.uint32).reshape(n, 6)[:, :5]
    return np.ascontiguousarray(w)


def build_filter(dev, start, nkeys, filter_n=FILTER_N):
    """filter_n random entries + PLANTED keys of [start, start+nkeys) -> bloom words resident on `dev`.
    Above 5*10^7 entries (non-headline experiments, e.g. the ~6 GB filter of configs[2]) the bit array is filled with
    random words of the design density 0.375 instead of inserting that many hashes."""
    from ecloop_amd.engine import blf_size_words
    size = blf_size_words(filter_n)
    if filter_n > 50_000_000:
        import torch
        g = torch.Generator(device="cuda").manual_seed(2025)
        chunk, parts = 1 << 27, []
        for at in range(0, size, chunk):
            m = min(chunk, size - at)
            a, b, c = (torch.randint(-(1 << 63), (1 << 63) - 1, (m,), dtype=torch.int64, device="cuda", generator=g) for _ in range(3))
            parts.append((a & (b | c)).cpu().numpy().view(np.uint64))
        dev.set_bloom(np.concatenate(parts))
    else:
        dev.set_bloom(np.zeros(size, dtype=np.uint64))
        dev.bloom_insert(splitmix_hashes(filter_n, 2025))
    offs = [(nkeys // PLANTED) * i + 12345 * (i + 1) % 4096 for i in range(PLANTED)]
    xs, ys, ok = dev.diag_mulg([start + o for o in offs])
    h33, h65 = dev.diag_hash160(xs, ys)
    dev.bloom_insert(h33)
    dev.bloom_insert(h65)
    return size, offs, h33


def cpu_baseline(words, sample_keys_log2_max=33):
    """The reference binary on this host's cores, same filter, same range start; bounded to ~10-30 s."""
    cores = os.cpu_count() or 1
    refs = [os.path.join(ROOT, "oracle", "_ref", n) for n in ("ecloop_sane", "ecloop_avx2")]
    from ecloop_amd.engine import blf_save
    tmp = tempfile.mkdtemp(prefix="eclbench")
    blf = os.path.join(tmp, "bench.blf")
    blf_save(blf, words)

    def run(binary, log2n, threads):
        out = os.path.join(tmp, "found.txt")
        if os.path.exists(out):
            os.unlink(out)
        end = RANGE_A + (1 << log2n) - 1
        t0 = time.time()
        pr = subprocess.run([binary, "add", "-f", blf, "-r", f"{RANGE_A:x}:{end:x}", "-t", str(threads), "-q", "-o", out],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        dt = time.time() - t0
        if pr.returncode != 0:
            raise RuntimeError(f"reference exited with {pr.returncode}")
        status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
        m = re.search(r"([\d.]+)s ~ ([\d.]+) Mkeys/s", status)
        lines = sorted(l.strip() for l in open(out)) if os.path.exists(out) else []
        return float(m.group(2)), float(m.group(1)), dt, lines

    import atexit
    import shutil
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    for binary in refs:
        if not os.path.exists(binary):
            continue
        try:
            # the reference stops scaling at a few dozen threads (one mutex-guarded job counter + status line,
            # main.c:419-431): measured on the 2x EPYC 9575F box 64 Mkeys/s at -t 32/64, 58 at 128, 44 at 256
            threads = min(cores, 64)
            log2n = 30
            rate, secs, _, lines = run(binary, log2n, threads)
            rate1, _, _, _ = run(binary, 25, 1)
            return {"value": rate, "unit": "Mkeys/s", "cores": threads, "kind": "reference",
                    "sample": f"{os.path.basename(binary)} add -r {RANGE_A:x}:+2^{log2n} same .blf, -t {threads} ({secs:.1f}s); -t 1: {rate1:.2f} Mkeys/s",
                    "single_thread_mkeys": rate1}, (log2n, lines)
        except Exception as e:  # SIGILL on a host without SHA-NI, missing binary, ...
            sys.stderr.write(f"[bench] reference baseline {binary} failed: {e}\n")
    # fallback: the oracle port (same algorithm, plain C, pthreads)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    import orc
    flt = orc.OrcFilter(bloom_words=words)
    threads = min(cores, 64)
    log2n = 27
    t0 = time.time()
    rc, out, n, checked, hashed = orc.add_range(flt, RANGE_A, RANGE_A + (1 << log2n), verify=False, threads=threads)
    dt = time.time() - t0
    lines = sorted(orc.found_lines(out, n))
    return {"value": hashed / dt / 1e6, "unit": "Mkeys/s", "cores": threads, "kind": "port",
            "sample": f"oracle/orc.c add over 2^{log2n} keys, {threads} threads ({dt:.1f}s)"}, (log2n, lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--keys-log2", type=int, default=32, help="keys per GPU per step (default 2^32 = the named config)")
    ap.add_argument("--launch-log2", type=int, default=32, help="largest number of keys given to one device call")
    ap.add_argument("--half-group", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--addr", default="c", choices=["c", "u", "cu"], help="non-headline variants: -a u / -a cu")
    ap.add_argument("--endo", action="store_true", help="non-headline variant: -endo (6 images per key)")
    ap.add_argument("--filter-n", type=int, default=FILTER_N, help="bloom entries (default 10^7 = 54 MB; 1.1e9 = 5.9 GB)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (already set on the boxes)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # test hook (not used by the driver): ECL_BENCH_SHARE_GPU=1 lets several ranks run on one GPU with the gloo
    # backend, to exercise the N>1 code path on a single-GPU box
    share = os.environ.get("ECL_BENCH_SHARE_GPU") == "1"
    if share:
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)

    from ecloop_amd.build import build_library
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        build_library()  # no-op when the in-tree .so is current (it travels with the snapshot); builds it if it is missing
    if dist is not None:
        dist.barrier()
    from ecloop_amd.engine import Filter, KeySearch, calc_priv
    nkeys = 1 << args.keys_log2
    start = RANGE_A + rank * nkeys

    # --- inputs -> HBM (untimed)
    headline = args.addr == "c" and not args.endo and args.filter_n == FILTER_N
    ks = KeySearch(Filter(np.zeros(1, dtype=np.uint64)), device=local, a33="c" in args.addr, a65="u" in args.addr,
                   endo=args.endo, verify=True,
                   launch_keys=1 << args.launch_log2, half_group=args.half_group, max_lanes=args.lanes)
    size, planted_offs, planted_h = build_filter(ks.dev, start, nkeys, args.filter_n)
    words = ks.dev.get_bloom(size) if (rank == 0 and world == 1 and not args.no_cpu and headline) else None
    ks.dev.reserve(min(nkeys, 1 << args.launch_log2))  # walk buffers allocated with the inputs, outside the timed region

    def barrier():
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        ks.found.clear()
        ks.add_keys(start, nkeys)

    for _ in range(args.warmup):
        step()
    ks.dev.reset_timing()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # --- correctness of what was just timed: every planted key is in the found list with the right scalar
    found_pks = {r.pk for r in ks.found}
    missing = [o for o in planted_offs if calc_priv(start, 1, o, 0) not in found_pks]
    if missing:
        raise SystemExit(f"[bench] rank {rank}: planted keys not found: {missing}")
    kernel_ms, launches, kkeys = ks.dev.timing()

    ok_flag = torch.tensor([1], device="cpu" if share else "cuda") if dist is not None else None
    if dist is not None:
        dist.all_reduce(ok_flag)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_keys = nkeys * world * args.steps
    value = total_keys / dt / 1e6
    ms_launch = kernel_ms / max(launches, 1)
    keys_per_launch = kkeys / max(launches, 1)
    achieved = keys_per_launch * LANE_CYCLES_PER_KEY / (ms_launch * 1e-3) / 1e12 if ms_launch > 0 else 0.0
    res = {
        "metric": "Mkeys/sec (add, addr33)" if headline else f"Mkeys/sec (add -a {args.addr}{' -endo' if args.endo else ''})", "value": round(value, 2), "unit": "Mkeys/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"add addr33, 2^{args.keys_log2} contiguous keys per GPU from 0x{RANGE_A:x}, "
                               f".blf bloom ({args.filter_n} entries, {size * 8 / 1e6:.0f} MB) resident in HBM",
                   "keys_per_gpu_per_step": nkeys, "parallelism": f"range-sharded x{world}, no collective",
                   "found_per_step": len(ks.found), "planted_found": PLANTED - len(missing)},
        "roofline": {"bound": "valu-int32", "achieved": round(achieved, 3), "peak": round(PEAK_TOPS, 2), "unit": "T lane-cycles/s",
                     "frac": round(achieved / PEAK_TOPS, 4), "traffic": round(keys_per_launch * TRAFFIC_BYTES_PER_KEY),
                     "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, profiles/r01_pmc_traffic.txt)",
                     "kernel": "k_add<addr33>", "ms_per_launch": round(ms_launch, 3), "keys_per_launch": int(keys_per_launch),
                     "valu_instr_per_key": OPS_PER_KEY, "lane_cycles_per_key": round(LANE_CYCLES_PER_KEY, 1), "kernel_mkeys_s": round(keys_per_launch / (ms_launch * 1e3), 2) if ms_launch else 0,
                     "valu_busy_pct_profiled": VALU_BUSY_PCT,
                     "hbm_gbs": round(keys_per_launch * TRAFFIC_BYTES_PER_KEY / (ms_launch * 1e-3) / 1e9, 1) if ms_launch else 0,
                     "hbm_frac_of_peak": round(keys_per_launch * TRAFFIC_BYTES_PER_KEY / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_launch else 0},
    }
    if not headline:
        hashes_per_key = len(args.addr) * (6 if args.endo else 1)
        res["config"]["workload"] = res["config"]["workload"].replace("add addr33", f"add -a {args.addr}{' -endo' if args.endo else ''}")
        res["config"]["hashes_per_key"] = hashes_per_key
        res["roofline"] = {"bound": "valu-int32", "note": "non-headline variant: algorithmic op count not priced", "ms_per_launch": round(ms_launch, 3),
                           "keys_per_launch": int(keys_per_launch), "hash160_per_s_G": round(value * hashes_per_key / 1e3, 2)}
    if world == 1 and not args.no_cpu and headline:
        cb, (log2n, cpu_lines) = cpu_baseline(words)
        res["cpu_baseline"] = cb
        gpu_lines = sorted(r.line() for r in ks.found if r.pk < RANGE_A + (1 << log2n))
        res["config"]["found_list_matches_cpu_on_sample"] = (gpu_lines == cpu_lines) if log2n <= args.keys_log2 else None
        if log2n <= args.keys_log2 and gpu_lines != cpu_lines:
            sys.stderr.write(f"[bench] FOUND LIST MISMATCH on the CPU sample: gpu {len(gpu_lines)} cpu {len(cpu_lines)}\n")
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
