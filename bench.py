#!/usr/bin/env python3
"""bench.py — `add` (addr33) key-search throughput on MI355X: BASELINE.json's metric on its configs[1].

    python bench.py [--gpus N --steps K --warmup W]
        N>1 under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: one rank per GPU;
        N>1 WITHOUT a launcher: N device threads in this one process, one GPU each (same shards, same line).
        Fewer visible GPUs than N is an error in both shapes.

One step = one pass of the hot path over ONE contiguous range of 2^32 private keys from 0x1_0000_0000 against a
`.blf`-format bloom filter resident in HBM: batch affine additions, SHA-256 -> RIPEMD-160 of every compressed public
key, bloom probe, hits gathered on the host.  With N GPUs the range is cut into N contiguous shards, one per rank
(north_star: "a 2^32 contiguous range at 1, 2, 4 and 8 MI355X") - strong scaling, no collective on the data path, only
the timing rendezvous (barrier + MAX over workers; gloo by default, `--control nccl` for RCCL).  `weak_scaling` in the same line is a second, separately timed leg where every rank scans its own
2^32 keys (`--scaling weak` makes that leg the headline instead).  Inputs are synthetic and already in HBM when the
timed region starts: the filter holds 10^7 seeded pseudo-random hash160 values plus 16 planted keys per 2^32-key range
(so the found list is not empty and is checked: a missing planted key or a found list that differs from the reference
binary's on the CPU sample aborts the run).

Prints ONE JSON line (rank 0).  `roofline`: the kernel is integer-VALU issue bound; `achieved` = VALU lane-operations
per key (rocprofv3 SQ_INSTS_VALU of THIS build, loaded from the tracked profile named in the line) x keys per launch
/ the kernel's HIP-event time measured in this process.  Nothing in it is a constant typed into this file: every field
is measured here or loaded from profiles/<tag>_roofline.json (tools/collect_profiles.sh), and `profile.matches_build`
says whether that profile was taken on the sources being run.  `cpu_baseline` is the unmodified reference binary
(oracle/_ref, built from /root/reference by oracle/Makefile) timed on this host's cores over a bounded sample of the
same range and filter, or the oracle port if the binary is not there.

    python bench.py --cmd mul      non-headline: the `mul` path (2^24 seeded scalars per step through ecl_hip_mul_batch)
"""
import argparse
import glob
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
# hardware peaks (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz, HBM3E 8 TB/s).  A wave64 VALU instruction occupies
# its SIMD for 4 clocks (16 lanes per clock); the add/sub/and/or/xor/mov/bitop3 class issues in ~2 in long runs
# (32 lanes per clock: the guide's "v_fma_f32 2 cyc").  Measured per opcode by ecloop_amd/csrc/tools/ubench.hip.
PEAK_4CYCLE = 256 * 4 * 16 * 2.4e9 / 1e12
PEAK_2CYCLE = 2 * PEAK_4CYCLE
HBM_PEAK_GBS = 8000.0
DATA = "synthetic"  # what the line's `data` field says
SETUP = {}  # where the process spent its time before the first timed step (config.process_setup)
ALGO_BYTES_PER_KEY = 18 + 18 + 1.59 * 8  # chain element (36 B) written + read per two keys; 1.59 probes x 8 B


def splitmix_hashes(n, seed):
    with np.errstate(over="ignore"):
        i = np.arange(1, n * 3 + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    w = z.view(np.uint32).reshape(n, 6)[:, :5]
    return np.ascontiguousarray(w)


def planted_offsets(nkeys):
    return [(nkeys // PLANTED) * i + 12345 * (i + 1) % 4096 for i in range(PLANTED)]


def build_filter(dev, start, nkeys, filter_n=FILTER_N, ranges=1, cuda_index=0):
    """filter_n random entries + PLANTED keys in each of the `ranges` consecutive nkeys-key ranges from `start` -> bloom
    words resident on `dev` (identical on every rank: one .blf replicated per GPU).
    Above 5*10^7 entries (non-headline experiments, e.g. the ~6 GB filter of configs[2]) the bit array is filled with
    random words of the design density 0.375 instead of inserting that many hashes."""
    from ecloop_amd.engine import blf_size_words
    size = blf_size_words(filter_n)
    if filter_n > 50_000_000:
        import torch
        cuda = f"cuda:{cuda_index}"
        g = torch.Generator(device=cuda).manual_seed(2025)
        chunk, parts = 1 << 27, []
        for at in range(0, size, chunk):
            m = min(chunk, size - at)
            a, b, c = (torch.randint(-(1 << 63), (1 << 63) - 1, (m,), dtype=torch.int64, device=cuda, generator=g) for _ in range(3))
            parts.append((a & (b | c)).cpu().numpy().view(np.uint64))
        dev.set_bloom(np.concatenate(parts))
    else:
        dev.set_bloom(np.zeros(size, dtype=np.uint64))
        dev.bloom_insert(splitmix_hashes(filter_n, 2025))
    offs = planted_offsets(nkeys)
    xs, ys, ok = dev.diag_mulg([start + r * nkeys + o for r in range(ranges) for o in offs])
    h33, h65 = dev.diag_hash160(xs, ys)
    dev.bloom_insert(h33)
    dev.bloom_insert(h65)
    return size, offs, h33


def host_cores():
    """what the box has: logical CPUs online, physical cores (distinct (package, core) pairs of /proc/cpuinfo), sockets, model"""
    logical = os.cpu_count() or 1
    cores, sockets, model = set(), set(), None
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif k == "model name" and model is None:
                model = v
            elif not k and phys is not None:
                cores.add((phys, core)), sockets.add(phys)
                phys = core = None
        if phys is not None:
            cores.add((phys, core)), sockets.add(phys)
    except OSError:
        pass
    return {"logical": logical, "physical": len(cores) or logical, "sockets": len(sockets) or 1, "model": model, "usable_by_this_process": len(os.sched_getaffinity(0))}


def cpu_baseline(words):
    """The UNMODIFIED reference (oracle/_ref, built from /root/reference by oracle/Makefile) on this host's cores, same
    filter, same range start, under every compiler flag set that runs here (SURVEY §7 "CPU baseline fairness": the
    reference's own Makefile flags are pathological on some hosts, so one flag set is not a baseline):
      ecloop_sane   -march=x86-64-v2 -msha -mno-avx*   (SHA-NI + scalar RIPEMD)          headline sample: 2^30 keys
      ecloop_avx2   -march=x86-64-v3 -mno-sha          (portable SHA + AVX2 RIPEMD x8)   2^29 keys
      ecloop_native -march=native (the reference Makefile's flags, built on the BUILD host's CPU; may not run here)
    value = the best of them; bounded to ~10-40 s of CPU work in all.  Falls back to the oracle port if no binary runs."""
    cores = os.cpu_count() or 1
    from ecloop_amd.engine import blf_save
    tmp = tempfile.mkdtemp(prefix="eclbench")
    blf = os.path.join(tmp, "bench.blf")
    blf_save(blf, words)

    def run(binary, log2n, threads, limit):
        out = os.path.join(tmp, "found.txt")
        if os.path.exists(out):
            os.unlink(out)
        end = RANGE_A + (1 << log2n) - 1
        t0 = time.time()
        pr = subprocess.run([binary, "add", "-f", blf, "-r", f"{RANGE_A:x}:{end:x}", "-t", str(threads), "-q", "-o", out],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit)
        dt = time.time() - t0
        if pr.returncode != 0:
            raise RuntimeError(f"exit status {pr.returncode}")
        status = pr.stderr.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
        m = re.search(r"([\d.]+)s ~ ([\d.]+) Mkeys/s", status)
        lines = sorted(l.strip() for l in open(out)) if os.path.exists(out) else []
        return float(m.group(2)), float(m.group(1)), dt, lines

    import atexit
    import shutil
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    # the reference stops scaling at a few dozen threads (one mutex-guarded job counter + status line, main.c:419-431):
    # measured on the 2x EPYC 9575F box 64 Mkeys/s at -t 32/64, 58 at 128, 44 at 256
    threads = min(cores, 64)
    sets, sample = [], None
    for name, flags, log2n in (("ecloop_sane", "-O3 -ffast-math -march=x86-64-v2 -msha -mno-avx -mno-avx2 -mno-avx512f", 30),
                               ("ecloop_avx2", "-O3 -ffast-math -march=x86-64-v3 -mno-sha", 29),
                               ("ecloop_native", "-O3 -ffast-math -march=native (reference Makefile:3-8; native = the build host)", 29)):
        binary = os.path.join(ROOT, "oracle", "_ref", name)
        if not os.path.exists(binary):
            sets.append({"binary": name, "flags": flags, "error": "not built (no /root/reference at build time)"})
            continue
        try:
            rate, secs, _, lines = run(binary, log2n, threads, 180)
            rate1, _, _, _ = run(binary, 24, 1, 120)
            sets.append({"binary": name, "flags": flags, "mkeys": rate, "threads": threads, "keys_log2": log2n, "seconds": round(secs, 1),
                         "single_thread_mkeys": rate1})
            if sample is None:  # the found list of the first set that ran is what the GPU list is compared with
                sample = (log2n, lines)
        except Exception as e:  # SIGILL on a host without the instruction set, time limit, ...
            sets.append({"binary": name, "flags": flags, "error": str(e)[:120]})
            sys.stderr.write(f"[bench] reference baseline {name} failed: {e}\n")
    ran = [x for x in sets if "mkeys" in x]
    host = host_cores()
    if ran:
        best = max(ran, key=lambda x: x["mkeys"])
        # ... and that flag set against the thread count (the reference defaults -t to the online CPUs, main.c:833-834): the headline is
        # the best row of the sweep, `cores` the threads it used, `host_cores` what the box has
        sweep = [{"threads": best["threads"], "mkeys": best["mkeys"], "keys_log2": best["keys_log2"]}]
        for t in sorted({32, 128, min(cores, 256)} - {best["threads"]}):
            if t > cores:
                continue
            try:
                rate, secs, _, _ = run(os.path.join(ROOT, "oracle", "_ref", best["binary"]), 29, t, 120)
                sweep.append({"threads": t, "mkeys": rate, "keys_log2": 29})
            except Exception as e:
                sweep.append({"threads": t, "error": str(e)[:80]})
        sweep.sort(key=lambda x: x["threads"])
        top = max((x for x in sweep if "mkeys" in x), key=lambda x: x["mkeys"])
        return {"value": top["mkeys"], "unit": "Mkeys/s", "cores": top["threads"], "host_cores": host, "kind": "reference",
                "sample": f"{best['binary']} add -r {RANGE_A:x}:+2^{top['keys_log2']} same .blf, -t {top['threads']} of {host['logical']} logical / "
                          f"{host['physical']} physical cores (best row of thread_sweep); -t 1: {best['single_thread_mkeys']:.2f} Mkeys/s; the other flag sets are in flag_sets",
                "single_thread_mkeys": best["single_thread_mkeys"], "thread_sweep": sweep, "flag_sets": sets}, sample
    # fallback: the oracle port (same algorithm, plain C, pthreads)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    import orc
    flt = orc.OrcFilter(bloom_words=words)
    log2n = 27
    t0 = time.time()
    rc, out, n, checked, hashed = orc.add_range(flt, RANGE_A, RANGE_A + (1 << log2n), verify=False, threads=threads)
    dt = time.time() - t0
    lines = sorted(orc.found_lines(out, n))
    return {"value": hashed / dt / 1e6, "unit": "Mkeys/s", "cores": threads, "host_cores": host, "kind": "port",
            "sample": f"oracle/orc.c add over 2^{log2n} keys, {threads} threads ({dt:.1f}s)", "flag_sets": sets}, (log2n, lines)


# ----------------------------------------------------------------------------------------------- roofline inputs


def load_profile(kind="roofline"):
    """the newest tracked profiles/r<NN>_<kind>.json (written by tools/collect_profiles.sh on the GPU box)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s.json" % kind)))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception as e:
        sys.stderr.write(f"[bench] cannot read {files[-1]}: {e}\n")
        return None, None


def static_fingerprint(kernel=None):
    """instruction mix of the library being timed, from the assembly the build kept beside it (tools/isa_mix.py)"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_mix
        a = isa_mix.analyse(kernel=kernel) if kernel else isa_mix.analyse()
        return a.get("fingerprint"), a.get("per_key_static")
    except Exception as e:
        sys.stderr.write(f"[bench] static instruction mix unavailable: {e}\n")
        return None, None


def add_roofline(ms_launch, keys_per_launch):
    """`achieved` = VALU lane-operations per key (PMC, this build) x keys per launch / kernel time measured here.
    `peak` = the guide's VALU peak: a wave64 instruction over 2 clocks = 32 lanes per clock per SIMD (78.6 T lane-ops/s).
    Only the double-rate opcode class reaches it; `issue_cycles` prices the kernel's own instruction stream at the
    hardware minimum per class (2 clocks for the double-rate class, 4 for the rest) against the SIMDs' clock budget,
    and `mix_ceiling` at the per-class rates the microbenchmark measures."""
    from ecloop_amd.build import source_sha256
    prof, path = load_profile()
    fp, static = static_fingerprint()
    keys_s = keys_per_launch / (ms_launch * 1e-3) if ms_launch > 0 else 0.0
    r = {"bound": "valu-int32", "kernel": "k_add<addr33>", "unit": "T lane-ops/s", "peak": round(PEAK_2CYCLE, 2),
         "peak_definition": "MI355X_MICROARCH.md: wave64 VALU instruction over 2 clocks = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; measured "
                            "(profiles/ubench_r03.txt) only for add/sub/and/or/xor/mov/not/shift/bitop3 (2.3-2.6 clocks in long runs); rotates, "
                            "v_add3, v_perm, v_bfe, 32-bit multiplies, v_mad_u64_u32, carries take >= 4.1",
         "ms_per_launch": round(ms_launch, 3), "keys_per_launch": int(keys_per_launch), "kernel_mkeys_s": round(keys_s / 1e6, 2),
         "static": fp}
    if prof is None:
        r.update({"achieved": None, "frac": None, "traffic": None, "profile": None})
        return r
    d, t = prof.get("derived", {}), prof.get("traffic", {})
    ops = d.get("valu_lane_ops_per_key")
    matches = (fp == prof.get("fingerprint")) if fp else None
    r["profile"] = {"file": path, "matches_build": matches, "source_sha256_matches": prof.get("source_sha256") == source_sha256(),
                    "profiled_launch_ms": prof.get("profiled_launch_ms"), "clock_ghz": d.get("clock_ghz"),
                    "valu_busy_pct": d.get("valu_busy_pct"), "simd_cycles_per_valu_instr": d.get("simd_cycles_per_valu_instr")}
    if ops:
        ach = ops * keys_s / 1e12
        r.update({"achieved": round(ach, 3), "frac": round(ach / PEAK_2CYCLE, 4), "valu_lane_ops_per_key": round(ops, 1)})
        if static:
            tot = static["valu"]
            sh = {k: static[k] / tot for k in ("mad64", "fast", "other")}
            # the stream's own minimum: 2 clocks for a double-rate instruction, 4 for any other = lane-cycles at 16 lanes per clock
            work = ops * (sh["fast"] * 0.5 + (1.0 - sh["fast"]))
            r["issue_cycles"] = {"unit": "T lane-cycles/s (16 lanes per clock per SIMD)", "double_rate_share": round(sh["fast"], 3),
                                 "work_lane_cycles_per_key": round(work, 1), "achieved": round(work * keys_s / 1e12, 3),
                                 "peak": round(PEAK_4CYCLE, 2), "frac": round(work * keys_s / 1e12 / PEAK_4CYCLE, 4),
                                 "class_shares_from": "tools/isa_mix.py on the assembly of this library (static, scaled to the PMC count)"}
            ub = (prof.get("ubench_cycles_per_wave_instr") or {}).get("waves_per_simd_8") or {}
            c_fast = min([ub[k] for k in ("v_add_u32_e32   (distinct regs)", "v_bitop3_b32    (distinct regs)", "v_add_u32 (e32)") if k in ub] or [0])
            c_slow, c_mad = ub.get("v_alignbit_b32"), ub.get("v_mad_u64_u32")
            if c_fast and c_slow and c_mad:
                cyc = sh["mad64"] * c_mad + sh["fast"] * c_fast + sh["other"] * c_slow
                ceil_keys = PEAK_4CYCLE * 1e12 * 4.0 / (ops * cyc)
                r["mix_ceiling"] = {"cycles_per_class": {"mad64": c_mad, "double_rate": c_fast, "other": c_slow}, "mean_cycles_per_instr": round(cyc, 3),
                                    "ceiling_mkeys_s": round(ceil_keys / 1e6, 1), "frac": round(keys_s / ceil_keys, 4),
                                    "note": "per-class rates of the microbenchmark at the nominal 2.4 GHz, double-rate opcodes at their long-run rate (in the hash they occur singly)"}
    if t.get("bytes_per_key_corrected"):
        r["traffic"] = round(t["bytes_per_key_corrected"] * keys_per_launch)
        r["traffic_source"] = f"{path}: FETCH_SIZE/WRITE_SIZE passes, corrected with the known-byte-count calibration in the same file"
        r["hbm"] = {"achieved_gbs": round(ALGO_BYTES_PER_KEY * keys_s / 1e9, 1), "peak_gbs": HBM_PEAK_GBS,
                    "frac": round(ALGO_BYTES_PER_KEY * keys_s / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_key": round(ALGO_BYTES_PER_KEY, 1),
                    "measured_bytes_per_key": round(t["bytes_per_key_corrected"], 1)}
    else:
        r["traffic"] = None
    if matches is False:
        r["profile"]["note"] = "STALE: the kernel's instruction mix differs from the build the counters were taken on; rerun tools/collect_profiles.sh"
    return r


# ----------------------------------------------------------------------------------------------- the N workers of a run
# The workload has no exchange step (contiguous shards, filter replicated, hits gathered on the host), so
# the N>1 machinery is a rendezvous for the timed region and nothing else: barrier, MAX over workers, gather of small
# host objects.  Two shapes, same worker code:
#   * ranks   - one process per GPU under torch.distributed.run (how the driver launches N>1).  The rendezvous runs
#               over gloo on CPU tensors by default: nothing of the hot path crosses it, and it is the backend the
#               two-rank tests execute everywhere; `--control nccl` puts it on RCCL (device tensors) instead.
#   * threads - `python bench.py --gpus N` WITHOUT a launcher: N host threads in this process, one device context each,
#               the shape of the C host program's scan workers (ecloop_hip_cli.c: scan_worker; the reference's
#               cmd_add_worker threads, main.c:405-435).  The library calls release the GIL.
# Either way fewer visible GPUs than N is an error, never a silent n_gpus: 1.


class Solo:
    rank, world, kind = 0, 1, "single process"

    def barrier(self):
        pass

    def allmax(self, x):
        return float(x)

    def gather(self, obj):
        return [obj]

    def close(self):
        pass


class Ranks:
    kind = "torch.distributed.run: one process per GPU"

    def __init__(self, control, local):
        import datetime
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.rank, self.world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        patience = datetime.timedelta(seconds=300)  # a rank that aborts (failed check) must not leave the others waiting for long
        if control == "nccl":
            ndev = torch.cuda.device_count()
            idx = local if local < ndev else local % max(ndev, 1)  # a launcher that shows each rank only its own GPU
            torch.cuda.set_device(idx)
            dist.init_process_group("nccl", device_id=torch.device("cuda", idx), timeout=patience)
            self.tdev = "cuda"
        else:
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # the container hostname may not resolve
            dist.init_process_group("gloo", timeout=patience)
            self.tdev = "cpu"
        self.kind += f", rendezvous over {control}"

    def barrier(self):
        self.dist.barrier()

    def allmax(self, x):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        parts = [None] * self.world
        self.dist.all_gather_object(parts, obj)
        return parts

    def close(self):
        self.dist.destroy_process_group()


class Threads:
    """rendezvous of N device threads in one process; an exception in one thread breaks the barrier for all"""
    kind = "in-process device threads (no launcher)"

    def __init__(self, world):
        import threading
        self.world = world
        self._bar = threading.Barrier(world)
        self._slots = [None] * world
        self._tls = threading.local()

    def bind(self, rank):
        self._tls.rank = rank

    @property
    def rank(self):
        return self._tls.rank

    def barrier(self):
        self._bar.wait(timeout=600)

    def _exchange(self, obj):
        self._slots[self.rank] = obj
        self._bar.wait(timeout=600)
        out = list(self._slots)
        self._bar.wait(timeout=600)
        return out

    def allmax(self, x):
        return max(float(v) for v in self._exchange(x))

    def gather(self, obj):
        return self._exchange(obj)

    def abort(self):
        self._bar.abort()

    def close(self):
        pass


def device_fence(dev_index):
    """torch.cuda.synchronize() on the worker's GPU: the library's calls already return after their stream has drained,
    this is the contract's belt to those braces"""
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize(dev_index)


def visible_gpus():
    from ecloop_amd import capi
    return max(int(capi.load().ecl_hip_device_count()), 0)


def device_identity(dev_index):
    """what tells the GPUs of two ranks apart across processes: host + PCI address (+ uuid where the runtime reports one) of device
    `dev_index` as this process sees it - the same enumeration the library's hipSetDevice uses"""
    import socket
    import torch
    host = socket.gethostname()
    try:
        p = torch.cuda.get_device_properties(dev_index)
        return (host, getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), str(getattr(p, "uuid", "")))
    except Exception:  # no torch device behind the index (the CPU tests' stand-in device): the index is all there is
        return (host, "index", dev_index)


# ----------------------------------------------------------------------------------------------- mul (non-headline)


def page_locked_copy(lib, arr):
    """the array in page-locked host memory allocated by the runtime (hipHostMalloc through ecl_hip_alloc_host: pages on the NUMA node next
    to the GPU, what the C host program uses) - a numpy array registered in place (ecl_hip_pin_host) sits wherever its pages were first
    touched, and from the far socket the DMA runs at 30 instead of 57 GB/s: 0.93 G scalars/s for `mul`, seen in 2 of ~100 runs"""
    import ctypes as C
    ptr = lib.ecl_hip_alloc_host(arr.nbytes)
    if not ptr:
        raise SystemExit("[bench] no page-locked memory for the scalar array")
    buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=arr.shape)
    buf[:] = arr
    return ptr, buf


def bench_mul(args, sync, dev_index, emit):
    from ecloop_amd.engine import Filter, KeySearch
    rank, world = sync.rank, sync.world
    n = 1 << args.mul_log2
    addr = args.addr if "--addr" in sys.argv else "cu"  # configs[4] / `make mul`: -a cu
    ks = KeySearch(Filter(np.zeros(64, dtype=np.uint64)), device=dev_index, a33="c" in addr, a65="u" in addr, verify=False)
    rng = np.random.default_rng(1234 + rank)
    scal = rng.integers(0, 1 << 63, (n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    lib, h = ks.dev.lib, ks.dev.h
    import ctypes as C
    if not args.pageable:  # the C host program keeps its scalar arrays in page-locked memory too (ecl_hip_alloc_host)
        _, scal = page_locked_copy(lib, scal)
    out = np.zeros(64, dtype=np.dtype([("b", "u1", (32,))]))
    cnt = C.c_uint32()
    # steady state of a long run: the 26-bit window table at once (left alone a context starts on 22 bits and moves to 26
    # after 2^30 scalars - more than this bench multiplies); `--mul-window 0` measures the automatic choice instead
    ks.dev.set_mul_window(args.mul_window)
    t_tab = time.perf_counter()

    def step():
        rc = lib.ecl_hip_mul_batch(h, scal.ctypes.data, n, out.ctypes.data, 64, C.byref(cnt))
        if rc not in (0, -4):
            raise SystemExit(f"[bench] mul_batch failed: {rc}")

    step()  # builds the table
    t_first = time.perf_counter() - t_tab
    for _ in range(max(args.warmup, 1)):
        step()
    ks.dev.reset_timing()
    device_fence(dev_index)
    sync.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_fence(dev_index)
    sync.barrier()
    dt = sync.allmax(time.perf_counter() - t0)
    ms, calls, nsc = ks.dev.mul_timing()
    if rank != 0:
        return
    hashes = len(addr)
    prof, path = load_profile("roofline_mul")
    res = {"metric": f"M scalars/sec (mul -a {addr})", "value": round(n * world * args.steps / dt / 1e6, 2), "unit": "Mscalars/s", "n_gpus": world,
           "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": f"mul -a {addr}: 2^{args.mul_log2} seeded 256-bit scalars per GPU per step from HOST memory through "
                                  "ecl_hip_mul_batch (copies overlapped with the kernel), empty filter", "hashes_per_scalar": hashes,
                      "host_memory": "pageable (staged)" if args.pageable else "page-locked (direct DMA)", "launcher": sync.kind,
                      "window_bits": ks.dev.mul_window(), "first_call_ms_incl_table_build": round(t_first * 1e3, 1)},
           "roofline": {"bound": "valu-int32", "kernel": "k_mul_check", "ms_per_call_on_stream": round(ms / max(calls, 1), 3),
                        "device_mscalars_s": round(nsc / (ms * 1e-3) / 1e6, 2) if ms else None,
                        "pcie_gbs": round(nsc * 32 / (ms * 1e-3) / 1e9, 2) if ms else None}}
    if prof and prof.get("derived", {}).get("valu_lane_ops_per_scalar"):
        ops = prof["derived"]["valu_lane_ops_per_scalar"]
        ach = ops * nsc / (ms * 1e-3) / 1e12
        res["roofline"].update({"achieved": round(ach, 3), "peak": round(PEAK_2CYCLE, 2), "unit": "T lane-ops/s", "frac": round(ach / PEAK_2CYCLE, 4),
                                "frac_of_4_clock_issue": round(ach / PEAK_4CYCLE, 4), "valu_lane_ops_per_scalar": round(ops, 1), "profile": path,
                                "note": "same peak as the add kernel's roofline (2-clock VALU issue); the multiplication-heavy window sums are mostly 4-clock+ opcodes"})
    emit(res)


# ----------------------------------------------------------------------------------------------- add (the headline)


def bench_add(args, sync, dev_index, emit, t_process):
    """one worker (rank or device thread): its shard of every leg on GPU `dev_index`; worker 0 reports"""
    from ecloop_amd.engine import Filter, KeySearch, calc_priv, shard
    rank, world = sync.rank, sync.world
    nkeys = 1 << args.keys_log2
    # per-worker scan of each leg: strong = shard `rank` of the one range, weak = the worker's own range
    legs = {"strong": (RANGE_A + shard(nkeys, rank, world)[0], shard(nkeys, rank, world)[1]),
            "weak": (RANGE_A + rank * nkeys, nkeys)}
    order = [args.scaling] + ([m for m in ("strong", "weak") if m != args.scaling] if world > 1 and not args.no_second_leg else [])

    # --- inputs -> HBM (untimed)
    headline = args.addr == "c" and not args.endo and args.filter_n == FILTER_N
    t0 = time.perf_counter()
    ks = KeySearch(Filter(np.zeros(1, dtype=np.uint64)), device=dev_index, a33="c" in args.addr, a65="u" in args.addr,
                   endo=args.endo, verify=True,
                   launch_keys=1 << args.launch_log2, half_group=args.half_group, max_lanes=args.lanes)
    setup = dict(SETUP, context_open_and_selftest_s=round(time.perf_counter() - t0, 2))  # HIP init, streams, the device code's self-test
    t0 = time.perf_counter()
    size, planted_offs, _ = build_filter(ks.dev, RANGE_A, nkeys, args.filter_n, ranges=world, cuda_index=dev_index)
    words = ks.dev.get_bloom(size) if (rank == 0 and world == 1 and not args.no_cpu and headline) else None
    setup["filter_build_s"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    for m in order:  # walk buffers allocated with the inputs, outside the timed region
        ks.dev.reserve(min(legs[m][1], 1 << args.launch_log2))
    setup["reserve_walk_buffers_s"] = round(time.perf_counter() - t0, 2)
    t_setup = time.perf_counter() - t_process

    def run_leg(mode, steps, warmup):
        start, cnt = legs[mode]

        def step():
            ks.found.clear()
            ks.add_keys(start, cnt)

        for _ in range(warmup):
            step()
        ks.dev.reset_timing()
        device_fence(dev_index)
        sync.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        device_fence(dev_index)
        mine_dt = time.perf_counter() - t0
        sync.barrier()
        dt = sync.allmax(time.perf_counter() - t0)
        # correctness of what was just timed: every planted key inside this worker's scan is in the found list
        found_pks = {r.pk for r in ks.found}
        mine = [RANGE_A + r * nkeys + o for r in range(world) for o in planted_offs]
        mine = [k for k in mine if start <= k < start + cnt]
        missing = [hex(k) for k in mine if calc_priv(k, 1, 0, 0) not in found_pks]
        if missing:
            raise SystemExit(f"[bench] worker {rank} ({mode}): planted keys not found: {missing}")
        kernel_ms, launches, kkeys = ks.dev.timing()
        setup_ms, setups = ks.dev.setup_timing()
        total = (nkeys if mode == "strong" else nkeys * world) * steps
        shards = sync.gather({"gpu": dev_index, "worker": rank, "first_key": hex(start), "keys_per_step": cnt, "lanes": ks.dev.geometry()[1],
                              "ms_per_step": round(mine_dt / steps * 1e3, 3), "kernel_ms_per_step": round(kernel_ms / steps, 3),
                              "found_per_step": len(ks.found), "planted_checked": len(mine)})
        return {"dt": dt, "value": total / dt / 1e6, "ms_per_step": dt / steps * 1e3, "kernel_ms": kernel_ms, "launches": launches,
                "kkeys": kkeys, "setup_ms": setup_ms, "setups": setups, "planted_checked": len(mine), "found": len(ks.found),
                "found_lines": sorted(r.line() for r in ks.found), "shards": shards}

    out = {order[0]: run_leg(order[0], args.steps, args.warmup)}
    for m in order[1:]:
        out[m] = run_leg(m, min(args.steps, 5), 1)
    # N = 1 only: the per-GPU shard of the named range at N = 2, 4, 8 as timed steps on THIS GPU (the last shard of each cut; every step
    # re-positions the walk - set-up kernels, host work and hit handling included, as a rank of an N-GPU run pays them), so that the
    # projected strong-scaling efficiency is a measured number of this run, not prose.  A projection: N identical GPUs, no host contention.
    shard_steps = None
    if world == 1 and headline and args.keys_log2 == 32 and not args.no_secondary and order[0] == "strong":
        shard_steps = []
        for n in (2, 4, 8):
            first, cnt = shard(nkeys, n - 1, n)
            ks.dev.reserve(cnt)

            def sstep():
                ks.found.clear()
                ks.add_keys(RANGE_A + first, cnt)

            sstep()
            ks.dev.reset_timing()
            device_fence(dev_index)
            t0 = time.perf_counter()
            for _ in range(5):
                sstep()
            device_fence(dev_index)
            dt = (time.perf_counter() - t0) / 5
            kernel_ms, launches, _ = ks.dev.timing()
            setup_ms, setups = ks.dev.setup_timing()
            mine = [RANGE_A + o for o in planted_offs if first <= o < first + cnt]
            found_pks = {r.pk for r in ks.found}
            if any(calc_priv(k, 1, 0, 0) not in found_pks for k in mine):
                raise SystemExit(f"[bench] shard step N={n}: planted keys not found")
            shard_steps.append({"n_gpus": n, "keys": cnt, "first_key": hex(RANGE_A + first), "steps": 5, "ms_per_step": round(dt * 1e3, 3),
                                "kernel_ms_per_step": round(kernel_ms / 5, 3), "setup_ms_per_step_on_device": round(setup_ms / 5, 3), "setups": setups,
                                "geometry": ks.dev.plan_geometry(cnt), "mkeys": round(cnt / dt / 1e6, 1), "planted_checked": len(mine)})
    sync.barrier()  # a worker that aborted above never gets here: the others time out in the barrier, they do not report
    ks.close()
    if rank != 0:
        return

    main_leg = out[order[0]]
    ms_launch = main_leg["kernel_ms"] / max(main_leg["launches"], 1)
    keys_per_launch = main_leg["kkeys"] / max(main_leg["launches"], 1)
    per_gpu = legs[order[0]][1]
    what = (f"ONE range of 2^{args.keys_log2} contiguous keys from 0x{RANGE_A:x} cut into {world} contiguous shard(s)" if order[0] == "strong"
            else f"2^{args.keys_log2} contiguous keys per GPU from 0x{RANGE_A:x}")
    res = {
        "metric": "Mkeys/sec (add, addr33)" if headline else f"Mkeys/sec (add -a {args.addr}{' -endo' if args.endo else ''})",
        "value": round(main_leg["value"], 2), "unit": "Mkeys/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(main_leg["ms_per_step"], 3),
        "higher_is_better": True, "scaling": order[0], "vs_baseline": None, "dtype": "u32", "data": DATA,
        "config": {"workload": f"add addr33, {what}, .blf bloom ({args.filter_n} entries, {size * 8 / 1e6:.0f} MB) resident in HBM",
                   "keys_per_gpu_per_step": per_gpu, "parallelism": f"range-sharded x{world}, no collective", "launcher": sync.kind,
                   "found_per_step": sum(x["found_per_step"] for x in main_leg["shards"]),
                   "planted_checked": sum(x["planted_checked"] for x in main_leg["shards"]),
                   "setup_ms_per_step_on_device": round(main_leg["setup_ms"] / args.steps, 3), "process_setup_s": round(t_setup, 2),
                   "process_setup": setup, "shards": main_leg["shards"]},
        "roofline": add_roofline(ms_launch, keys_per_launch),
    }
    if shard_steps:
        full_ms = main_leg["ms_per_step"]
        res["shard_steps"] = {"what": "the last contiguous shard of the named 2^32-key range at N = 2 / 4 / 8, timed on this one GPU: 5 steps each, every step "
                                      "non-contiguous (walk re-positioned), set-up kernels + host work + hit verification inside the timed region",
                              "full_range_ms_per_step": round(full_ms, 3), "steps": shard_steps,
                              "projected_efficiency": {str(x["n_gpus"]): round(full_ms / x["n_gpus"] / x["ms_per_step"], 4) for x in shard_steps},
                              "projected_mkeys": {str(x["n_gpus"]): round(x["n_gpus"] * x["keys"] / x["ms_per_step"] / 1e3, 1) for x in shard_steps},
                              "label": "PROJECTION from one GPU (N identical GPUs, independent shards, no collective on the data path): not a measured N-GPU run"}
    for m in order[1:]:
        res[m + "_scaling"] = {"value": round(out[m]["value"], 2), "unit": "Mkeys/s", "ms_per_step": round(out[m]["ms_per_step"], 3),
                               "steps": min(args.steps, 5), "keys_per_gpu_per_step": legs[m][1]}
    if not headline:
        hashes_per_key = len(args.addr) * (6 if args.endo else 1)
        res["config"]["workload"] = res["config"]["workload"].replace("add addr33", f"add -a {args.addr}{' -endo' if args.endo else ''}")
        res["config"]["hashes_per_key"] = hashes_per_key
        res["roofline"] = {"bound": "valu-int32", "note": "non-headline variant: no PMC profile of this kernel is loaded", "ms_per_launch": round(ms_launch, 3),
                           "keys_per_launch": int(keys_per_launch), "hash160_per_s_G": round(main_leg["value"] * hashes_per_key / 1e3, 2)}
    if world == 1 and not args.no_cpu and headline:
        cb, (log2n, cpu_lines) = cpu_baseline(words)
        res["cpu_baseline"] = cb
        if log2n <= args.keys_log2:
            gpu_lines = [l for l in main_leg["found_lines"] if int(l.split("\t")[2], 16) < RANGE_A + (1 << log2n)]
            res["config"]["found_list_matches_cpu_on_sample"] = gpu_lines == cpu_lines
            if gpu_lines != cpu_lines:
                emit(res)
                raise SystemExit(f"[bench] FOUND LIST MISMATCH on the CPU sample: gpu {len(gpu_lines)} lines, cpu {len(cpu_lines)} lines")
    if world == 1 and headline and not args.no_secondary and args.keys_log2 == 32:
        res["secondary"] = secondary_legs(args, dev_index)  # configs[2], [3], [4] in the same run; a failed check aborts
    emit(res)



# ----------------------------------------------------------------------------------------------- secondary legs (N = 1)
# BASELINE.json's other configs, measured in the same run as the headline and reported under `secondary` (the headline
# fields are untouched): cfg2 = configs[2] on one GPU (add -a cu -endo, ~6 GB .blf), cfg3 = configs[3] (rnd -d 128:32
# windows through the C host program), cfg4 = configs[4] (mul: scalars through ecl_hip_mul_batch and hex lines through
# the host program's stdin).  Each leg checks what it found: planted keys must be there, and the found list over a
# bounded sample must equal the oracle's (tests/orc.py over oracle/liborc.so: the checker, never the thing timed).
# A failed check aborts the run like the headline's.


def _orc():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    import orc
    return orc


def _roofline_from(kind, ops_key, rate_per_s, what):
    """roofline object of a secondary kernel: VALU lane-ops per unit from profiles/rNN_roofline_<kind>.json x the rate
    measured here, against the same 2-clock VALU peak as the headline"""
    prof, path = load_profile("roofline_" + kind)
    r = {"bound": "valu-int32", "kernel": what, "unit": "T lane-ops/s", "peak": round(PEAK_2CYCLE, 2)}
    ops = (prof or {}).get("derived", {}).get(ops_key)
    if not ops:
        r.update({"achieved": None, "frac": None, "profile": None, "note": "no PMC profile of this kernel under profiles/"})
        return r
    from ecloop_amd.build import source_sha256
    ach = ops * rate_per_s / 1e12
    r.update({"achieved": round(ach, 3), "frac": round(ach / PEAK_2CYCLE, 4), ops_key: round(ops, 1),
              "profile": {"file": path, "source_sha256_matches": prof.get("source_sha256") == source_sha256(),
                          "valu_busy_pct": prof["derived"].get("valu_busy_pct"), "clock_ghz": prof["derived"].get("clock_ghz")}})
    t = prof.get("traffic") or {}
    if t.get("bytes_per_unit_reported") is not None:
        r["traffic_bytes_per_unit_reported"] = round(t["bytes_per_unit_reported"], 1)
    if kind == "mul":
        r["mix_ceiling"] = mul_mix_ceiling(ops, rate_per_s, prof)
    return r


def mul_mix_ceiling(ops, rate_per_s, prof):
    """what k_mul_check's own instruction stream allows: class shares of the library being timed (tools/isa_mix.py: analyse_mul - sum loop +
    8 trips of the window loop + walk-back loop, static) x the per-class issue rates of the microbenchmark (the headline profile's table)
    -> mean SIMD-clocks per instruction -> scalars/s at the nominal 2.4 GHz and at the clock this kernel really sustains (the multiply-add
    heavy stream is power-limited: 2.1-2.2 GHz by GRBM_GUI_ACTIVE against 2.33 for the add kernel)"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_mix
        m = isa_mix.analyse_mul()
        st = m["per_scalar_static"]
        head, _ = load_profile()
        ub = ((head or {}).get("ubench_cycles_per_wave_instr") or {}).get("waves_per_simd_8") or {}
        c_fast = min([ub[k] for k in ("v_add_u32_e32   (distinct regs)", "v_bitop3_b32    (distinct regs)", "v_add_u32 (e32)") if k in ub] or [0])
        c_slow, c_mad = ub.get("v_alignbit_b32"), ub.get("v_mad_u64_u32")
        if not (c_fast and c_slow and c_mad):
            return None
        sh = {k: st[k] / st["valu"] for k in ("mad64", "fast", "other")}
        cyc = sh["mad64"] * c_mad + sh["fast"] * c_fast + sh["other"] * c_slow
        ceil_nominal = PEAK_4CYCLE * 1e12 * 4.0 / (ops * cyc)
        out = {"class_shares": {k: round(v, 3) for k, v in sh.items()}, "cycles_per_class": {"mad64": c_mad, "double_rate": c_fast, "other": c_slow},
               "mean_cycles_per_instr": round(cyc, 3), "ceiling_mscalars_s": round(ceil_nominal / 1e6, 1), "frac": round(rate_per_s / ceil_nominal, 4),
               "fingerprint": m["fingerprint"], "fingerprint_matches_profile": (prof.get("static_mix") or {}).get("fingerprint") == m["fingerprint"],
               "note": "per-class rates of the microbenchmark at the nominal 2.4 GHz; shares: static, trip-weighted (tools/isa_mix.py), scaled to the PMC count"}
        # the double-rate opcodes only reach 2.5 clocks in long runs; between multiply-adds they issue like any other instruction
        out["mean_cycles_per_instr_double_rate_priced_singly"] = round(sh["mad64"] * c_mad + (1.0 - sh["mad64"]) * c_slow, 3)
        ghz = (prof.get("derived") or {}).get("clock_ghz")
        if ghz:
            out["at_sustained_clock"] = {"clock_ghz": round(ghz, 3), "ceiling_mscalars_s": round(ceil_nominal * ghz / 2.4 / 1e6, 1),
                                         "frac": round(rate_per_s / (ceil_nominal * ghz / 2.4), 4),
                                         "measured_simd_cycles_per_valu_instr": (prof.get("derived") or {}).get("simd_cycles_per_valu_instr")}
        return out
    except Exception as e:
        sys.stderr.write(f"[bench] mix ceiling of the mul kernel unavailable: {e}\n")
        return None


def leg_cu_endo(args, dev_index):
    """configs[2] on one GPU: add -a cu -endo (12 hash160 per key), filter of 1.1e9 entries (5.9 GB) resident in HBM"""
    from ecloop_amd.engine import Filter, KeySearch, calc_priv
    orc = _orc()
    nkeys, sample = 1 << args.cfg2_keys_log2, 1 << 21
    t0 = time.perf_counter()
    ks = KeySearch(Filter(np.zeros(1, dtype=np.uint64)), device=dev_index, a33=True, a65=True, endo=True, verify=True)
    size, planted_offs, _ = build_filter(ks.dev, RANGE_A, nkeys, args.cfg2_filter_n, cuda_index=dev_index)
    inside = [RANGE_A + 1000 + 4099 * i for i in range(16)]  # 16 more planted keys inside the oracle sample
    xs, ys, _ = ks.dev.diag_mulg(inside)
    h33, h65 = ks.dev.diag_hash160(xs, ys)
    ks.dev.bloom_insert(h33)
    ks.dev.bloom_insert(h65)
    ks.dev.reserve(nkeys)
    t_setup = time.perf_counter() - t0

    def step():
        ks.found.clear()
        ks.add_keys(RANGE_A, nkeys)

    step()
    ks.dev.reset_timing()
    device_fence(dev_index)
    t0 = time.perf_counter()
    for _ in range(args.cfg2_steps):
        step()
    device_fence(dev_index)
    dt = time.perf_counter() - t0
    kernel_ms, launches, kkeys = ks.dev.timing()
    found_pks = {r.pk for r in ks.found}
    want = [RANGE_A + o for o in planted_offs] + inside
    missing = [hex(k) for k in want if calc_priv(k, 1, 0, 0) not in found_pks]
    if missing:
        raise SystemExit(f"[bench] cfg2: planted keys not found: {missing}")
    nfound = len(ks.found)
    # the oracle on a sample of the same range and the same 5.9 GB of filter words (copied back from the device)
    t0 = time.perf_counter()
    words = ks.dev.get_bloom(size)
    ks.found.clear()
    ks.add_keys(RANGE_A, sample)
    gpu_lines = sorted(r.line() for r in ks.found)
    ks.close()
    threads = min(os.cpu_count() or 1, 64)
    rc, out, n, _, hashed = orc.add_range(orc.OrcFilter(bloom_words=words, borrow=True), RANGE_A, RANGE_A + sample, a65=True, endo=True,
                                          verify=False, threads=threads, cap=1 << 16)
    cpu_lines = sorted(orc.found_lines(out, n))
    t_check = time.perf_counter() - t0
    if rc != 0 or hashed != sample or gpu_lines != cpu_lines or len(cpu_lines) < 32:
        raise SystemExit(f"[bench] cfg2: FOUND LIST MISMATCH on the oracle sample: gpu {len(gpu_lines)} lines, oracle {len(cpu_lines)} (rc {rc}, hashed {hashed})")
    rate = nkeys * args.cfg2_steps / dt
    krate = kkeys / (kernel_ms * 1e-3) if kernel_ms else 0.0
    return {"metric": "Mkeys/sec (add -a cu -endo)", "value": round(rate / 1e6, 2), "unit": "Mkeys/s", "hash160_per_s_G": round(rate * 12 / 1e9, 2),
            "steps": args.cfg2_steps, "ms_per_step": round(dt / args.cfg2_steps * 1e3, 3),
            "config": {"workload": f"add -a cu -endo, 2^{args.cfg2_keys_log2} contiguous keys from 0x{RANGE_A:x} per step, .blf of {args.cfg2_filter_n} entries "
                                   f"({size * 8 / 1e6:.0f} MB, random words at the design density) resident in HBM, 1 GPU (configs[2]'s per-GPU shard shape)",
                       "hashes_per_key": 12, "found_per_step": nfound, "planted_checked": len(want),
                       "oracle_sample": f"first 2^21 keys (12 hash160 each) on {threads} host threads against the same filter words: {len(cpu_lines)} lines, equal",
                       "found_list_matches_oracle_on_sample": True, "setup_s": round(t_setup, 1), "check_s": round(t_check, 1)},
            "roofline": dict(_roofline_from("cu_endo", "valu_lane_ops_per_key", krate, "k_add<addr33,addr65,endo>"),
                             ms_per_launch=round(kernel_ms / max(launches, 1), 3), keys_per_launch=int(kkeys / max(launches, 1)),
                             kernel_mkeys_s=round(krate / 1e6, 2))}


def secondary_filter(dev_index, scalars_to_plant, mode="a&(b|c)", name="secondary"):
    """a 56 MB filter for the cfg3 / cfg4 legs: random words at the .blf design density 0.375 (mode "a&(b|c)"; the timed runs) or
    at density 0.5 (mode "a": ~4100 false positives per 2^32 hash160, so that an oracle sample has lines to compare; the check
    run of cfg3) + the hash160s of the given scalars' public keys; -> (words, path of the .blf, its directory)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import synth_bloom_words, write_blf
    from ecloop_amd.capi import Device
    words = synth_bloom_words(7000003, 41, mode)
    d = Device(dev_index, a33=True, a65=True)
    try:
        if scalars_to_plant:
            d.set_bloom(words)
            xs, ys, _ = d.diag_mulg(scalars_to_plant)
            h33, h65 = d.diag_hash160(xs, ys)
            d.bloom_insert(h33)
            d.bloom_insert(h65)
            words = d.get_bloom(len(words))
    finally:
        d.close()
    tmp = tempfile.mkdtemp(prefix="eclbench2")
    import atexit
    import shutil
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    path = os.path.join(tmp, name + ".blf")
    write_blf(path, words)
    return words, path, tmp


def _status_of(stderr_bytes):
    status = stderr_bytes.decode(errors="replace").replace("\x1b[2K", "\r").split("\r")[-1].strip()
    m = re.search(r"([\d.]+)s ~ ([\d.]+) Mkeys/s ~ ([\d,]+) / ([\d,]+)", status)
    if not m:
        raise SystemExit(f"[bench] cannot read the host program's status line: {status!r}")
    return float(m.group(1)), float(m.group(2)), int(m.group(3).replace(",", "")), int(m.group(4).replace(",", ""))


def leg_rnd(args, blf, dense_words, dense_blf, tmp):
    """configs[3] on one GPU: `ecloop-hip rnd -d 128:32` - random 2^32-key windows at stride 2^128, a new base point per window.
    Timed on the design-density filter (a window's handful of false positives says little); the same command line is then run for
    one window on a filter of density 0.5, whose found list over a slice of the printed window is compared with the oracle's."""
    from ecloop_amd.build import build_host_cli
    orc = _orc()
    cli = build_host_cli()
    lo, hi = (1 << 167) + 0x1234567, (1 << 168) - 0x7654321

    def rnd(filter_path, nwin, out):
        env = dict(os.environ, ECLOOP_HIP_RND_WINDOWS=str(nwin), ECLOOP_HIP_STATS="1")
        t0 = time.perf_counter()
        pr = subprocess.run([cli, "rnd", "-f", filter_path, "-r", f"{lo:x}:{hi:x}", "-d", "128:32", "-seed", "bench", "-t", "1", "-q", "-o", out],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        wall = time.perf_counter() - t0
        if pr.returncode != 0:
            raise SystemExit(f"[bench] cfg3: ecloop-hip rnd failed: {pr.stderr.decode(errors='replace')[-500:]}")
        text = pr.stdout.decode(errors="replace")
        secs, mkeys, found, checked = _status_of(pr.stderr)
        masks = [int(l.replace(" ", ""), 16) for l in text.splitlines() if re.fullmatch(r"[0-9a-f ]{67}", l)]
        m = re.search(r"gpu 0: (\d+) launches, ([\d.]+) ms in the search kernel, (\d+) set-ups, ([\d.]+) ms in set-up kernels", text)
        if len(masks) != 2 * nwin or checked != nwin << 32 or not m:
            raise SystemExit(f"[bench] cfg3: unexpected output of ecloop-hip rnd ({len(masks)} masks, {checked} checked)")
        lines = sorted(l.rstrip("\n") for l in open(out)) if os.path.exists(out) else []
        return secs, mkeys, found, checked, masks, float(m.group(2)), float(m.group(4)), wall, lines

    nwin = args.cfg3_windows
    secs, mkeys, found, checked, masks, kernel_ms, setup_ms, wall, _ = rnd(blf, nwin, os.path.join(tmp, "rnd.txt"))
    # the check run: one window, dense filter, the oracle over the first 2^24 keys of the printed window (stride 2^128, full-size jobs)
    t0 = time.perf_counter()
    _, _, _, _, cmasks, _, _, _, lines = rnd(dense_blf, 1, os.path.join(tmp, "rnd_check.txt"))
    threads, part, s0 = min(os.cpu_count() or 1, 64), 1 << 24, cmasks[0]
    rc, o, n, _, hashed = orc.add_range(orc.OrcFilter(bloom_words=dense_words), s0, s0 + ((part - 1) << 128), offs=128, rnd=True, verify=False,
                                        threads=threads, cap=1 << 16)
    want = sorted(orc.found_lines(o, n))
    mine = [l for l in lines if (int(l.split("\t")[2], 16) - s0) % (1 << 128) == 0 and 0 <= (int(l.split("\t")[2], 16) - s0) >> 128 < part]
    t_check = time.perf_counter() - t0
    if rc != 0 or hashed != part or mine != want or len(want) < 8:
        raise SystemExit(f"[bench] cfg3: FOUND LIST MISMATCH on the oracle sample: gpu {len(mine)} lines, oracle {len(want)} (rc {rc})")
    return {"metric": "Mkeys/sec (rnd -d 128:32)", "value": mkeys, "unit": "Mkeys/s", "windows": nwin, "seconds_by_status_line": secs,
            "config": {"workload": f"ecloop-hip rnd -d 128:32 -t 1: {nwin} random windows of 2^32 keys at stride 2^128 on a 168-bit range, 56 MB .blf at the design "
                                   "density; rate = the host program's own status line (set-up of every window included, process start-up not)",
                       "found": found, "checked": checked, "wall_s_incl_process_start": round(wall, 2),
                       "device_ms_search_kernel": kernel_ms, "device_ms_window_setup": setup_ms, "setup_share": round(setup_ms / (kernel_ms + setup_ms), 5),
                       "kernel_mkeys_s": round(checked / (kernel_ms * 1e-3) / 1e6, 2),
                       "oracle_sample": f"same command, one window, filter of density 0.5: first 2^24 keys of the printed window on {threads} host threads: "
                                        f"{len(want)} lines, equal",
                       "found_list_matches_oracle_on_sample": True, "check_s": round(t_check, 1)}}


def leg_small_jobs(args, blf, tmp):
    """The reference's scheduler against the library: `ecloop-hip add` over 2^34 keys handed out in the reference's own jobs of 2^21 keys
    from the shared counter (ECLOOP_HIP_JOB_KEYS=2097152 = MAX_JOB_SIZE, main.c:16,418-431) - one ecl_hip_add_range call per job, as the
    reference bound through the ABI makes them (that binary lives under oracle/ and is measured by tools/bench_ref_binding.py,
    not here).  With the look-ahead (default), without it (the library of round 5), and with eight worker threads on eight contexts of the
    one GPU; found lines of every run equal to the default large-call run's."""
    from ecloop_amd.build import build_host_cli
    cli = build_host_cli()
    log2 = args.small_jobs_log2
    rng = "%x:%x" % (RANGE_A, RANGE_A + (1 << log2) - 1)

    def run(env, threads=1):
        out = os.path.join(tmp, "small_jobs.txt")
        if os.path.exists(out):
            os.unlink(out)
        pr = subprocess.run([cli, "add", "-f", blf, "-r", rng, "-t", str(threads), "-q", "-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            stdin=subprocess.DEVNULL, timeout=900, env=dict(os.environ, **env))
        if pr.returncode != 0:
            raise SystemExit(f"[bench] small jobs: ecloop-hip add failed: {pr.stderr.decode(errors='replace')[-500:]}")
        secs, mk, found, checked = _status_of(pr.stderr)
        if checked != 1 << log2:
            raise SystemExit(f"[bench] small jobs: checked {checked} of 2^{log2} keys")
        return mk, secs, sorted(l.rstrip("\n") for l in open(out)) if os.path.exists(out) else []

    job = {"ECLOOP_HIP_JOB_KEYS": str(1 << 21)}
    base = run({})
    legs = {"lookahead": run(job), "lookahead_off": run(dict(job, ECL_HIP_LOOKAHEAD_LOG2="0")),
            "lookahead_8_threads_8_contexts": run(dict(job, ECLOOP_HIP_SHARE_GPU="8"), threads=8)}
    for k, v in legs.items():
        if v[2] != base[2]:
            raise SystemExit(f"[bench] small jobs ({k}): FOUND LIST differs from the large-call run's: {len(v[2])} against {len(base[2])} lines")
    return {"metric": "Mkeys/sec (add, addr33, handed out in the reference's 2^21-key jobs)", "value": legs["lookahead"][0], "unit": "Mkeys/s",
            "seconds_by_status_line": legs["lookahead"][1], "lookahead_off": legs["lookahead_off"][0],
            "eight_worker_threads_on_eight_contexts": legs["lookahead_8_threads_8_contexts"][0], "large_calls": base[0],
            "config": {"workload": f"ecloop-hip add -r {rng} (2^{log2} keys), 56 MB .blf, -t 1; every ecl_hip_add_range call is one 2^21-key job from the shared "
                                   "counter (ECLOOP_HIP_JOB_KEYS=2097152); rates by the host program's status line",
                       "found": len(base[2]), "found_lists_equal_to_the_large_call_run": True}}


def leg_mul(args, dev_index, words, blf, tmp, planted):
    """configs[4]: mul -a cu.  (a) 2^24 seeded scalars per call from page-locked host memory through ecl_hip_mul_batch;
    (b) 2^26 64-hex-digit lines through the host program's stdin (the reference's input format)."""
    import ctypes as C
    from ecloop_amd.build import build_host_cli
    from ecloop_amd.engine import Filter, KeySearch
    orc = _orc()
    n, nsample = 1 << args.cfg4_log2, 1 << 12
    rng = np.random.default_rng(4242)
    scal = rng.integers(0, 1 << 63, (n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    for i, k in enumerate(planted):  # the planted scalars sit inside the oracle sample
        scal[17 * i + 5] = [(k >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
    ks = KeySearch(Filter(words), device=dev_index, a33=True, a65=True, verify=False)
    lib, h = ks.dev.lib, ks.dev.h
    scal_ptr, scal = page_locked_copy(lib, scal)
    cap = 1 << 12
    out = np.zeros(cap, dtype=np.dtype([("key_offset", "<u8"), ("h160", "<u4", (5,)), ("endo", "u1"), ("compressed", "u1"), ("pad", "u1", (2,))]))
    cnt = C.c_uint32()
    ks.dev.set_mul_window(args.mul_window)

    def step():
        rc = lib.ecl_hip_mul_batch(h, scal.ctypes.data, n, out.ctypes.data, cap, C.byref(cnt))
        if rc != 0:
            raise SystemExit(f"[bench] cfg4: ecl_hip_mul_batch failed: {rc}")

    t0 = time.perf_counter()
    step()  # builds the table
    t_first = time.perf_counter() - t0
    step()
    ks.dev.reset_timing()
    device_fence(dev_index)
    t0 = time.perf_counter()
    each = []
    for _ in range(args.cfg4_steps):
        t1 = time.perf_counter()
        step()
        each.append(round((time.perf_counter() - t1) * 1e3, 3))  # (a call returns when its records are on the host: the steps do not overlap)
    device_fence(dev_index)
    dt = time.perf_counter() - t0
    ms, calls, nsc = ks.dev.mul_timing()
    wbits = ks.dev.mul_window()
    recs = out[: cnt.value].copy()
    # ... and on the widest table the library builds: 29 bits = 9 additions per scalar from 138 GB of HBM (on request only: it takes seconds to
    # build); same scalars, same filter, so the records must be the same ones
    wide = None
    if args.mul_window == 26 and args.cfg4_log2 >= 22:
        try:
            ks.dev.set_mul_window(29)

            def step29():
                rc = lib.ecl_hip_mul_batch(h, scal.ctypes.data, n, out.ctypes.data, cap, C.byref(cnt))
                if rc != 0:
                    raise RuntimeError(f"ecl_hip_mul_batch on the 29-bit table: {rc}")

            t0 = time.perf_counter()
            step29()
            t_first29 = time.perf_counter() - t0
            step29()
            device_fence(dev_index)
            t0 = time.perf_counter()
            for _ in range(args.cfg4_steps):
                step29()
            device_fence(dev_index)
            dt29 = time.perf_counter() - t0
            key = lambda a: sorted((int(r["key_offset"]), int(r["compressed"]), tuple(int(w) for w in r["h160"])) for r in a)
            if ks.dev.mul_window() != 29 or key(out[: cnt.value]) != key(recs):
                raise SystemExit("[bench] cfg4: the 29-bit table reports other records than the 26-bit one")
            wide = {"window_bits": 29, "value": round(n * args.cfg4_steps / dt29 / 1e6, 2), "unit": "Mscalars/s", "ms_per_step": round(dt29 / args.cfg4_steps * 1e3, 3),
                    "first_call_ms_incl_table_build": round(t_first29 * 1e3, 1), "table_gb": 138.4, "records_equal_to_the_26_bit_run": True}
        except Exception as e:  # no room for 138 GB beside whatever else holds HBM: reported, not fatal
            wide = {"window_bits": 29, "error": str(e)[:200]}

    def line(label, h160, k):
        return "%s\t%s\t%064x" % (label, "".join("%08x" % int(w) for w in h160), k)

    val = lambda row: sum(int(row[j]) << (64 * j) for j in range(4))
    gpu_lines = sorted(line("addr33" if r["compressed"] else "addr65", r["h160"], val(scal[int(r["key_offset"])])) for r in recs if r["key_offset"] < nsample)
    rc, o, no = orc.mul_batch(orc.OrcFilter(bloom_words=words), [val(scal[i]) for i in range(nsample)], a33=True, a65=True)
    cpu_lines = sorted(orc.found_lines(o, no))
    lib.ecl_hip_free_host(scal_ptr)
    ks.close()
    if rc != 0 or gpu_lines != cpu_lines or len(cpu_lines) < 2 * len(planted):
        raise SystemExit(f"[bench] cfg4: FOUND LIST MISMATCH on the oracle sample: gpu {len(gpu_lines)} lines, oracle {len(cpu_lines)}")
    drate = nsc / (ms * 1e-3) if ms else 0.0
    api = {"metric": "M scalars/sec (mul -a cu, ecl_hip_mul_batch)", "value": round(n * args.cfg4_steps / dt / 1e6, 2), "unit": "Mscalars/s",
           "steps": args.cfg4_steps, "ms_per_step": round(dt / args.cfg4_steps * 1e3, 3),
           "config": {"workload": f"2^{args.cfg4_log2} seeded 256-bit scalars per call from page-locked HOST memory (copies overlapped with the kernels inside the call), "
                                  "-a cu, 56 MB .blf at the design density", "window_bits": wbits, "first_call_ms_incl_table_build": round(t_first * 1e3, 1), "steps_ms": each,
                      "hits_per_call": int(cnt.value), "pcie_gbs": round(drate * 32 / 1e9, 2),
                      "oracle_sample": f"the first {nsample} scalars (with {len(planted)} planted) through the oracle's cmd_mul: {len(cpu_lines)} lines, equal",
                      "found_list_matches_oracle_on_sample": True},
           "widest_table": wide,
           "roofline": dict(_roofline_from("mul", "valu_lane_ops_per_scalar", drate, "mul kernels (window sums + hash160 + probe)"),
                            device_mscalars_s=round(drate / 1e6, 2), ms_per_call_on_stream=round(ms / max(calls, 1), 3))}
    # (b) the host program: lines of 64 hex digits on stdin - a regular file (2^30 lines = 70 GB on tmpfs where the box has the memory: a timed
    # window of about a second) and a pipe (`head -c ... | ecloop-hip mul`, the reference's usual form, main.c:542-576, over 2^28 lines)
    log2 = args.cfg4_cli_log2
    if log2 == 0:
        log2 = 26
        try:
            st = os.statvfs("/dev/shm")
            avail = int(next(l.split()[1] for l in open("/proc/meminfo") if l.startswith("MemAvailable"))) * 1024
            if st.f_bavail * st.f_frsize > 100e9 and avail > 200e9:
                log2 = 30
            elif st.f_bavail * st.f_frsize > 30e9 and avail > 60e9:
                log2 = 28
        except Exception:
            pass
    nl = 1 << log2
    gen = os.path.join(tmp, "gen_hex_lines")
    subprocess.run(["gcc", "-O2", "-pthread", os.path.join(ROOT, "tools", "gen_hex_lines.c"), "-o", gen], check=True)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and log2 >= 27 else tmp
    src = os.path.join(shm, "ecl_bench_mul_in_%d.txt" % os.getpid())
    import atexit
    atexit.register(lambda: os.path.exists(src) and os.unlink(src))
    t0 = time.perf_counter()
    subprocess.run([gen, str(nl), "7", src, str(min(64, os.cpu_count() or 1))], check=True)
    with open(src, "r+b") as f:  # planted keys among the first lines
        for i, k in enumerate(planted):
            f.seek(65 * (17 * i + 5))
            f.write(b"%064x" % k)
    t_gen = time.perf_counter() - t0
    head = [int(l, 16) for l in open(src, "rb").read(65 * nsample).split()]
    cli = build_host_cli()
    outp = os.path.join(tmp, "mul_out.txt")

    def run_cli(stdin, n_expected):
        if os.path.exists(outp):
            os.unlink(outp)
        t0 = time.perf_counter()
        pr = subprocess.run([cli, "mul", "-f", blf, "-a", "cu", "-q", "-o", outp], stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        wall = time.perf_counter() - t0
        if pr.returncode != 0:
            raise SystemExit(f"[bench] cfg4: ecloop-hip mul failed: {pr.stderr.decode(errors='replace')[-500:]}")
        secs, mk, found, checked = _status_of(pr.stderr)
        if checked != n_expected:
            raise SystemExit(f"[bench] cfg4: ecloop-hip mul checked {checked} of {n_expected} lines")
        return mk, secs, wall, found

    def check_found():
        lines = sorted(l.rstrip("\n") for l in open(outp))
        first = set(head)
        mine = [l for l in lines if int(l.split("\t")[2], 16) in first]
        rc, o, no = orc.mul_batch(orc.OrcFilter(bloom_words=words), head, a33=True, a65=True)
        want = sorted(orc.found_lines(o, no))
        if rc != 0 or mine != want or len(want) < 2 * len(planted):
            raise SystemExit(f"[bench] cfg4: FOUND LIST MISMATCH of the host program on the oracle sample: {len(mine)} lines, oracle {len(want)}")
        return len(want)

    run_cli(open(src, "rb"), nl)  # untimed: the first read of freshly written tmpfs pages runs at a fifth of the rate of the following ones
    runs = [run_cli(open(src, "rb"), nl) for _ in range(3)]
    nwant = check_found()
    rates = sorted(r[0] for r in runs)
    med = runs[[r[0] for r in runs].index(rates[1])]
    npipe = min(nl, 1 << 28)
    feeder = subprocess.Popen(["head", "-c", str(65 * npipe), src], stdout=subprocess.PIPE)
    pipe = run_cli(feeder.stdout, npipe)
    feeder.wait()
    check_found()
    os.unlink(src)
    clileg = {"metric": "M lines/sec (ecloop-hip mul -a cu, hex lines on stdin)", "value": med[0], "unit": "Mlines/s", "seconds_by_status_line": med[1],
              "runs_mlines_s": [r[0] for r in runs], "spread": round((rates[2] - rates[0]) / rates[1], 4),
              "pipe": {"value": pipe[0], "unit": "Mlines/s", "lines_log2": int(np.log2(npipe)), "seconds_by_status_line": pipe[1],
                       "what": "head -c <lines> file | ecloop-hip mul: bounded by the pipe (one reader, one writer, 1 MB in flight), not by the parser or the device"},
              "config": {"workload": f"2^{log2} lines of 64 hex digits (tools/gen_hex_lines.c, {t_gen:.0f} s to write) from a regular file on stdin, -a cu, same filter; "
                                     "rate = the host program's status line (clock from the end of bring-up to the last device call), median of 3 runs after one untimed pass",
                         "found": med[3], "wall_s_incl_process_start": round(med[2], 2),
                         "oracle_sample": f"the first {nsample} lines through the oracle's cmd_mul: {nwant} lines, equal (file and pipe runs)", "found_list_matches_oracle_on_sample": True}}
    # (c) the same program with -raw (main.c:503-527: the scalar of a line is its SHA-256 - pass phrases): hashed on the device, the text
    # and the line table travel with the pieces of a call (ecl_hip_mul_batch_raw)
    rawleg = leg_mul_raw(dev_index, cli, log2, tmp, shm, nsample)
    return {"api": api, "host_program": clileg, "host_program_raw": rawleg}


def leg_mul_raw(dev_index, cli, log2, tmp, shm, nsample):
    import hashlib
    orc = _orc()
    nl = 1 << log2
    gen = os.path.join(tmp, "gen_phrases")
    subprocess.run(["gcc", "-O2", "-pthread", os.path.join(ROOT, "tools", "gen_phrases.c"), "-o", gen], check=True)
    src = os.path.join(shm, "ecl_bench_mul_raw_%d.txt" % os.getpid())
    import atexit
    atexit.register(lambda: os.path.exists(src) and os.unlink(src))
    t0 = time.perf_counter()
    subprocess.run([gen, str(nl), "11", src, str(min(32, os.cpu_count() or 1))], check=True, stdout=subprocess.DEVNULL)
    t_gen = time.perf_counter() - t0
    with open(src, "rb") as f:
        head = f.read(26 * nsample).split(b"\n")[:nsample]
    scal = [int.from_bytes(hashlib.sha256(l).digest(), "big") for l in head]
    planted = [scal[17 * i + 5] for i in range(16)]  # the filter holds the public keys of 16 of the first lines
    words, blf, _ = secondary_filter(dev_index, planted, name="raw")
    outp = os.path.join(tmp, "mul_raw_out.txt")

    def run_cli():
        if os.path.exists(outp):
            os.unlink(outp)
        t0 = time.perf_counter()
        pr = subprocess.run([cli, "mul", "-raw", "-f", blf, "-a", "cu", "-q", "-o", outp], stdin=open(src, "rb"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        wall = time.perf_counter() - t0
        if pr.returncode != 0:
            raise SystemExit(f"[bench] cfg4: ecloop-hip mul -raw failed: {pr.stderr.decode(errors='replace')[-500:]}")
        secs, mk, found, checked = _status_of(pr.stderr)
        if checked != nl:
            raise SystemExit(f"[bench] cfg4: ecloop-hip mul -raw checked {checked} of {nl} lines")
        return mk, secs, wall, found

    run_cli()
    runs = [run_cli() for _ in range(3)]
    first = set(scal)
    mine = sorted(l.rstrip("\n") for l in open(outp) if int(l.split("\t")[2], 16) in first)
    rc, o, no = orc.mul_batch(orc.OrcFilter(bloom_words=words), scal, a33=True, a65=True)
    want = sorted(orc.found_lines(o, no))
    if rc != 0 or mine != want or len(want) < 2 * len(planted):
        raise SystemExit(f"[bench] cfg4: FOUND LIST MISMATCH of `mul -raw` on the oracle sample: {len(mine)} lines, oracle {len(want)}")
    os.unlink(src)
    rates = sorted(r[0] for r in runs)
    med = runs[[r[0] for r in runs].index(rates[1])]
    return {"metric": "M lines/sec (ecloop-hip mul -raw -a cu, pass phrases on stdin)", "value": med[0], "unit": "Mlines/s", "seconds_by_status_line": med[1],
            "runs_mlines_s": [r[0] for r in runs], "spread": round((rates[2] - rates[0]) / rates[1], 4),
            "config": {"workload": f"2^{log2} lines of 8..24 characters (tools/gen_phrases.c, {t_gen:.0f} s to write) from a regular file on stdin, SHA-256 of every line on the "
                                   "device, -a cu, 56 MB .blf at the design density; rate = the host program's status line, median of 3 runs after one untimed pass",
                       "found": med[3], "wall_s_incl_process_start": round(med[2], 2),
                       "oracle_sample": f"the SHA-256 of the first {nsample} lines (16 of them planted in the filter) through the oracle's cmd_mul: {len(want)} lines, equal",
                       "found_list_matches_oracle_on_sample": True}}


def secondary_legs(args, dev_index):
    out, t_all = {}, time.perf_counter()
    planted = [0x1000000000000000000000000000000000000000000000000000000000000000 + 0x9E3779B97F4A7C15 * (i + 1) for i in range(16)]
    t0 = time.perf_counter()
    out["cfg2"] = leg_cu_endo(args, dev_index)
    out["cfg2"]["leg_s"] = round(time.perf_counter() - t0, 1)
    words, blf, tmp = secondary_filter(dev_index, planted)
    dense_words, dense_blf, _ = secondary_filter(dev_index, [], mode="a", name="dense")
    t0 = time.perf_counter()
    out["cfg3"] = leg_rnd(args, blf, dense_words, dense_blf, tmp)
    out["cfg3"]["leg_s"] = round(time.perf_counter() - t0, 1)
    t0 = time.perf_counter()
    out["cfg4"] = leg_mul(args, dev_index, words, blf, tmp, planted)
    out["cfg4"]["leg_s"] = round(time.perf_counter() - t0, 1)
    t0 = time.perf_counter()
    out["small_jobs"] = leg_small_jobs(args, blf, tmp)
    out["small_jobs"]["leg_s"] = round(time.perf_counter() - t0, 1)
    out["seconds"] = round(time.perf_counter() - t_all, 1)
    return out


# ----------------------------------------------------------------------------------------------- main


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node; N>1 without a launcher runs N device threads in this process")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = ONE 2^keys-log2 range cut into N shards (default, the named config); weak = 2^keys-log2 keys per GPU")
    ap.add_argument("--control", default="gloo", choices=["gloo", "nccl"],
                    help="ranks under torch.distributed.run: backend of the timing rendezvous (no data-path collective exists)")
    ap.add_argument("--keys-log2", type=int, default=32, help="keys per step (strong) / per GPU per step (weak); default 2^32 = the named config")
    ap.add_argument("--launch-log2", type=int, default=32, help="largest number of keys given to one device call")
    ap.add_argument("--half-group", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-second-leg", action="store_true", help="N>1: skip the other scaling mode's leg")
    ap.add_argument("--addr", default="c", choices=["c", "u", "cu"], help="non-headline variants: -a u / -a cu")
    ap.add_argument("--endo", action="store_true", help="non-headline variant: -endo (6 images per key)")
    ap.add_argument("--filter-n", type=int, default=FILTER_N, help="bloom entries (default 10^7 = 54 MB; 1.1e9 = 5.9 GB)")
    ap.add_argument("--cmd", default="add", choices=["add", "mul"], help="mul: the non-headline `mul` path")
    ap.add_argument("--mul-log2", type=int, default=24)
    ap.add_argument("--mul-window", type=int, default=26, help="mul: window width of the table (0 = the library's automatic choice)")
    ap.add_argument("--pageable", action="store_true", help="mul: scalars in pageable host memory (staged through pinned buffers by the library)")
    ap.add_argument("--no-secondary", action="store_true", help="N=1 headline run: skip the `secondary` legs (configs[2], [3], [4])")
    ap.add_argument("--cfg2-keys-log2", type=int, default=30)
    ap.add_argument("--cfg2-steps", type=int, default=2)
    ap.add_argument("--cfg2-filter-n", type=int, default=1_100_000_000, help="entries of the cfg2 filter (1.1e9 = 5.9 GB, configs[2]'s '~6 GB bloom')")
    ap.add_argument("--cfg3-windows", type=int, default=3)
    ap.add_argument("--cfg4-log2", type=int, default=24)
    ap.add_argument("--cfg4-steps", type=int, default=3)
    ap.add_argument("--small-jobs-log2", type=int, default=34, help="keys of the `small_jobs` leg (the reference's 2^21-key hand-out through the host program)")
    ap.add_argument("--cfg4-cli-log2", type=int, default=0, help="lines fed to the host program; 0 = 2^30 where /dev/shm and RAM allow (70 GB), else 2^28 / 2^26")
    args = ap.parse_args()
    t_process = time.perf_counter()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver (already set on the boxes)
    launched = "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) >= 1 and "RANK" in os.environ
    world = int(os.environ["WORLD_SIZE"]) if launched else (args.gpus or 1)
    if world < 1:
        raise SystemExit("[bench] --gpus must be >= 1")
    if launched and args.gpus is not None and args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} contradicts the launcher's WORLD_SIZE {world}")
    local = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    # torch's HIP runtime (its own copy in the wheel) must come up BEFORE libecloop_hip's (/opt/rocm): the other way
    # round torch finds "no ROCm-capable device" - and torch is what fills multi-GB synthetic filters on the device
    t0 = time.perf_counter()
    import torch
    SETUP["import_torch_s"] = round(time.perf_counter() - t0, 2)  # 1-2 minutes on a fresh box while the image pages in
    t0 = time.perf_counter()
    if torch.cuda.is_available():
        torch.cuda.init()
        if local < torch.cuda.device_count():  # otherwise: refused below (or folded onto the GPUs that exist by the test hook)
            torch.cuda.set_device(local)
    SETUP["torch_cuda_init_s"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    if local == 0:
        from ecloop_amd.build import build_library, library_is_current
        SETUP["library_was_current"] = library_is_current()  # decided by the source hash stamped beside the .so, not by file times
        build_library()  # no-op when the in-tree .so is current (it travels with the snapshot); builds it if it is missing
    SETUP["library_build_check_s"] = round(time.perf_counter() - t0, 2)
    # test hook (not used by the driver): ECL_BENCH_SHARE_GPU=1 lets several workers run on the GPUs that exist
    # (worker g on device g mod count), to exercise the N>1 code path on a single-GPU box
    share = os.environ.get("ECL_BENCH_SHARE_GPU") == "1"

    def emit(res):
        print(json.dumps(res), flush=True)

    def work(sync, dev_index):
        if args.cmd == "mul":
            bench_mul(args, sync, dev_index, emit)
        else:
            bench_add(args, sync, dev_index, emit, t_process)

    if launched and world > 1:
        sync = Ranks(args.control, local)  # rendezvous first: rank 0 has built the library by the time the others load it
        sync.barrier()
        # one rank per PHYSICAL GPU.  A launcher may show every rank all GPUs (rank r takes device LOCAL_RANK) or only its own
        # (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank: device 0 of what the rank sees); what is refused is two ranks on the
        # same device - told by host + PCI address, gathered over the rendezvous - never a line that says n_gpus=N for fewer GPUs
        have = visible_gpus()
        if have < 1:
            raise SystemExit(f"[bench] rank {sync.rank}: 0 GPU(s) visible")
        dev = local if local < have else local % have
        ids = sync.gather(device_identity(dev))
        if len(set(ids)) < world and not share:
            raise SystemExit(f"[bench] {world} ranks on {len(set(ids))} distinct GPU(s) ({have} GPU(s) visible to rank {sync.rank}): refusing to report "
                             f"n_gpus={world} (one rank per GPU)")
        try:
            work(sync, dev)
        finally:
            sync.close()
        return
    have = visible_gpus()
    if have < world and not share:
        raise SystemExit(f"[bench] --gpus {world} but {have} GPU(s) visible: refusing to run (one device thread per GPU)")
    if world == 1:
        work(Solo(), 0)
        return
    import threading
    sync = Threads(world)
    errors = []

    def body(g):
        sync.bind(g)
        try:
            work(sync, g % have if share else g)
        except BaseException as e:  # SystemExit of a failed check included: release the others, report, fail the run
            errors.append((g, e))
            sync.abort()

    th = [threading.Thread(target=body, args=(g,), name=f"gpu{g}") for g in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    real = [(g, e) for g, e in errors if not isinstance(e, threading.BrokenBarrierError)]
    if errors:
        for g, e in real or errors:
            sys.stderr.write(f"[bench] device thread {g}: {e!r}\n")
        raise SystemExit(1)


if __name__ == "__main__":
    main()
