"""bench.py on the GPU box: the JSON contract at N=1, and the N>1 code path in both shapes - `--gpus 2` with no launcher
(two device threads) and two ranks under torch.distributed.run (rendezvous over gloo, and over RCCL when two GPUs are
visible).  On a one-GPU box the two workers share the GPU (ECL_BENCH_SHARE_GPU=1), so the sharded legs, the barrier /
MAX-over-workers timing and the planted-key checks of every worker run either way."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(out):
    lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
    assert lines, out.decode(errors="replace")[-2000:]
    return json.loads(lines[-1])


def test_bench_line_single_gpu_small():
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--keys-log2", "28", "--steps", "2", "--warmup", "1", "--no-cpu"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-2000:]
    r = last_json(pr.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["scaling"] == "strong" and r["config"]["planted_checked"] == 16 and r["value"] > 1000
    assert abs(r["value"] - (1 << 28) / (r["ms_per_step"] * 1e3)) / r["value"] < 1e-3
    rf = r["roofline"]
    assert rf["bound"] == "valu-int32" and rf["ms_per_launch"] > 0 and rf["keys_per_launch"] == 1 << 28
    assert rf["static"] and rf["static"]["kernel_valu"] > 3000  # instruction mix of the library that was timed
    # the PMC profile the roofline is priced with belongs to THIS build of k_add (tools/collect_profiles.sh after every
    # kernel change; on the CPU the same drift is a warning, tests/test_profiles_fresh.py)
    assert rf.get("profile") and rf["profile"]["matches_build"] is True, rf.get("profile")
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], abs=2e-3)


def test_bench_secondary_legs_at_reduced_size():
    """the `secondary` object of the default N = 1 run (BASELINE configs[2], [3], [4] beside the headline), at sizes that finish in a
    minute: every leg present, checked against the oracle on its sample (a mismatch makes bench.py exit non-zero), rates consistent"""
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu", "--cfg2-keys-log2", "27",
                         "--cfg2-steps", "1", "--cfg2-filter-n", "60000000", "--cfg3-windows", "1", "--cfg4-log2", "22", "--cfg4-steps", "2",
                         "--cfg4-cli-log2", "22", "--small-jobs-log2", "30"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, cwd=ROOT)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    r = last_json(pr.stdout)
    assert r["metric"] == "Mkeys/sec (add, addr33)" and r["config"]["keys_per_gpu_per_step"] == 1 << 32  # the headline is untouched
    s = r["secondary"]
    c2, c3, api, cli = s["cfg2"], s["cfg3"], s["cfg4"]["api"], s["cfg4"]["host_program"]
    assert c2["config"]["hashes_per_key"] == 12 and c2["config"]["found_list_matches_oracle_on_sample"] and c2["config"]["planted_checked"] == 32
    assert abs(c2["value"] - (1 << 27) / (c2["ms_per_step"] * 1e3)) / c2["value"] < 1e-3 and c2["value"] > 300
    assert c2["roofline"]["kernel"] == "k_add<addr33,addr65,endo>" and c2["roofline"]["keys_per_launch"] == 1 << 27
    assert c3["windows"] == 1 and c3["config"]["checked"] == 1 << 32 and c3["config"]["found_list_matches_oracle_on_sample"] and c3["value"] > 3000
    assert 0 < c3["config"]["setup_share"] < 0.05
    assert api["config"]["found_list_matches_oracle_on_sample"] and api["config"]["window_bits"] == 26 and api["value"] > 100
    wide = api["widest_table"]  # the same call on the 29-bit table (138 GB): same records, its own rate
    assert wide["window_bits"] == 29 and wide.get("records_equal_to_the_26_bit_run") and wide["value"] > 100, wide
    assert abs(api["value"] - (1 << 22) / (api["ms_per_step"] * 1e3)) / api["value"] < 1e-3
    assert cli["config"]["found_list_matches_oracle_on_sample"] and cli["value"] > 10
    assert len(cli["runs_mlines_s"]) == 3 and cli["value"] == sorted(cli["runs_mlines_s"])[1] and cli["pipe"]["value"] > 1 and cli["pipe"]["lines_log2"] == 22
    raw = s["cfg4"]["host_program_raw"]  # pass phrases, hashed on the device
    assert raw["config"]["found_list_matches_oracle_on_sample"] and raw["value"] > 10 and raw["value"] == sorted(raw["runs_mlines_s"])[1]
    # the reference's 2^21-key hand-out through the host program: found lists equal, the look-ahead ahead of plain launches
    sj = s["small_jobs"]
    assert sj["config"]["found_lists_equal_to_the_large_call_run"] and sj["value"] > sj["lookahead_off"] > 1000 and sj["eight_worker_threads_on_eight_contexts"] > 1000
    # the per-GPU shards of the named range at N = 2, 4, 8 as timed steps of this one GPU, and the efficiency projected from them
    sh = r["shard_steps"]
    assert [x["n_gpus"] for x in sh["steps"]] == [2, 4, 8] and [x["keys"] for x in sh["steps"]] == [1 << 31, 1 << 30, 1 << 29]
    assert all(x["setups"] == 5 and x["kernel_ms_per_step"] < x["ms_per_step"] for x in sh["steps"])  # every step re-positioned the walk
    for x in sh["steps"]:
        assert sh["projected_efficiency"][str(x["n_gpus"])] == pytest.approx(sh["full_range_ms_per_step"] / x["n_gpus"] / x["ms_per_step"], abs=1e-3)
    assert 0.85 < sh["projected_efficiency"]["8"] < 1.05 and "PROJECTION" in sh["label"]
    for leg in (c2, api):  # rooflines priced with the kernels' own PMC profiles when those are in the tree
        if leg["roofline"].get("profile"):
            assert leg["roofline"]["frac"] == pytest.approx(leg["roofline"]["achieved"] / leg["roofline"]["peak"], abs=2e-3)


def check_two(r, launcher):
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["keys_per_gpu_per_step"] == 1 << 26
    assert launcher in r["config"]["launcher"]
    assert abs(r["value"] - (1 << 27) / (r["ms_per_step"] * 1e3)) / r["value"] < 1e-3
    sh = sorted(r["config"]["shards"], key=lambda x: x["worker"])
    assert [x["keys_per_step"] for x in sh] == [1 << 26, 1 << 26] and sum(x["planted_checked"] for x in sh) == 16
    w = r["weak_scaling"]
    assert w["keys_per_gpu_per_step"] == 1 << 27 and abs(w["value"] - 2 * (1 << 27) / (w["ms_per_step"] * 1e3)) / w["value"] < 1e-3
    assert "cpu_baseline" not in r  # worker 0 at N=1 only


def two_gpu_env():
    import torch
    two = torch.cuda.device_count() >= 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not two:
        env["ECL_BENCH_SHARE_GPU"] = "1"  # both workers on the one GPU of this box
    return env, two


SMALL2 = ["--gpus", "2", "--keys-log2", "27", "--steps", "2", "--warmup", "1"]


def test_bench_gpus_2_without_a_launcher():
    """`python bench.py --gpus 2`, nothing else: two device threads in one process (two GPUs if the box has them)"""
    env, _ = two_gpu_env()
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL2, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                        timeout=1200, cwd=ROOT, env=env)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    check_two(last_json(pr.stdout), "in-process device threads")


@pytest.mark.parametrize("control", ["gloo", "nccl"])
def test_bench_two_ranks_strong_and_weak_legs(control):
    """the driver's N>1 form: torch.distributed.run, one rank per GPU; the rendezvous over gloo (default) and over RCCL
    (needs two real GPUs: RCCL refuses two ranks on one device)"""
    env, two = two_gpu_env()
    if control == "nccl" and not two:
        pytest.skip("one GPU on this box: RCCL cannot host two ranks on it")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517" if control == "gloo" else "29519", os.path.join(ROOT, "bench.py"), "--control", control] + SMALL2
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, cwd=ROOT, env=env)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    check_two(last_json(pr.stdout), "torch.distributed.run")


@pytest.mark.parametrize("launcher", ["threads", "ranks"])
def test_bench_gpus_8_shared(launcher):
    """the shape the driver's SCALE run ends on - `--gpus 8` - on whatever GPUs this box has (one: ECL_BENCH_SHARE_GPU=1 folds worker g
    onto device g mod count), in both launcher shapes: eight device threads in one process, and eight ranks under torch.distributed.run
    with the rendezvous over gloo.  Eight contexts, eight filters and eight sets of walk buffers live on the device(s) at once (the lane
    count is cut back to what the free HBM allows, ecloop_hip.hip: default_lanes); every shard finds the planted keys of ITS part of the
    range (bench.py aborts otherwise), and the line is timed over the slowest worker."""
    import torch
    ngpu = torch.cuda.device_count()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if ngpu < 8:
        env["ECL_BENCH_SHARE_GPU"] = "1"
    tail = ["--gpus", "8", "--keys-log2", "29", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-secondary"]
    if launcher == "threads":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--control", "gloo"] + tail
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500, cwd=ROOT, env=env)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    r = last_json(pr.stdout)
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["config"]["keys_per_gpu_per_step"] == 1 << 26
    assert ("torch.distributed.run" if launcher == "ranks" else "in-process device threads") in r["config"]["launcher"]
    sh = sorted(r["config"]["shards"], key=lambda x: x["worker"])
    assert [x["worker"] for x in sh] == list(range(8)) and [x["gpu"] for x in sh] == [g % max(ngpu, 1) if ngpu < 8 else g for g in range(8)]
    # contiguous shards that tile the one 2^29-key range
    assert sum(x["keys_per_step"] for x in sh) == 1 << 29
    assert [int(x["first_key"], 16) for x in sh] == [0x100000000 + g * (1 << 26) for g in range(8)]
    # the 16 planted keys (bench.py: build_filter) are spread over the range; each was found by the shard whose keys contain it
    assert sum(x["planted_checked"] for x in sh) == 16 and r["config"]["planted_checked"] == 16
    assert all(x["found_per_step"] >= x["planted_checked"] for x in sh) and sum(x["planted_checked"] > 0 for x in sh) >= 4
    # MAX over the workers: the step is as long as the slowest shard, and the value is the whole job over that time
    assert r["ms_per_step"] >= max(x["ms_per_step"] for x in sh) * 0.999
    assert abs(r["value"] - (1 << 29) / (r["ms_per_step"] * 1e3)) / r["value"] < 1e-3
    assert all(x["lanes"] >= 256 * 256 for x in sh)  # every context got a walk that fills the chip, none failed for memory
    w = r["weak_scaling"]  # the other leg: 2^29 keys per worker
    assert w["keys_per_gpu_per_step"] == 1 << 29 and abs(w["value"] - 8 * (1 << 29) / (w["ms_per_step"] * 1e3)) / w["value"] < 1e-3
    assert "cpu_baseline" not in r and "secondary" not in r


def test_bench_ranks_are_matched_to_physical_gpus():
    """under torch.distributed.run a rank takes device LOCAL_RANK, or - when the launcher shows every rank only its own GPU
    (HIP_VISIBLE_DEVICES per rank) - device 0 of what it sees; ranks that end up on the SAME physical GPU (told by PCI address over
    the rendezvous) are refused: one rank more than the box has GPUs, no sharing hook"""
    import torch
    ngpu = torch.cuda.device_count()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ECL_BENCH_SHARE_GPU", None)
    n = ngpu + 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", "29523", os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--keys-log2", "24", "--steps", "1", "--warmup", "0", "--no-cpu"]
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT, env=env)
    assert pr.returncode != 0 and b"distinct GPU(s)" in pr.stderr and b'"metric"' not in pr.stdout


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    n = torch.cuda.device_count() + 1
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--keys-log2", "24", "--no-cpu"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert pr.returncode != 0 and b"GPU(s) visible" in pr.stderr and not pr.stdout.strip()


def test_bench_cmd_mul_line():
    """`bench.py --cmd mul` (non-headline): one JSON line, the 22-bit window table in use, rates consistent"""
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cmd", "mul", "--mul-log2", "21", "--steps", "2", "--warmup", "1"],
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-2000:]
    r = last_json(pr.stdout)
    assert r["unit"] == "Mscalars/s" and r["config"]["window_bits"] == 26 and r["n_gpus"] == 1
    assert abs(r["value"] - (1 << 21) / (r["ms_per_step"] * 1e3)) / r["value"] < 1e-3 and r["value"] > 100
