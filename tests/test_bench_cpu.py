"""bench.py's N>1 plumbing on the CPU: `--gpus 2` with NO launcher (two device threads in one process) and under
torch.distributed.run (two ranks, gloo rendezvous), the GPU replaced by tests/fake_device.py through the launcher
tests/bench_on_fake_device.py (bench.py has no test hook of its own).  What is under test is
what the driver's scaling run depends on: the flag is honoured (n_gpus: 2, never a silent 1), the shards tile the
range, every worker checks its planted keys, MAX-over-workers timing and the gather of the per-GPU shard records.
The numbers in the line are meaningless here and the line says so (`data`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, os.environ.get("PYTHONPATH", "")]))
BENCH = os.path.join(ROOT, "tests", "bench_on_fake_device.py")  # swaps capi.Device for the oracle-backed stand-in, then bench.main()
SMALL = ["--keys-log2", "15", "--filter-n", "2000", "--steps", "2", "--warmup", "1", "--launch-log2", "14"]


def last_json(pr):
    lines = [l for l in pr.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    assert pr.returncode == 0 and len(lines) == 1, pr.stderr.decode(errors="replace")[-3000:]
    return json.loads(lines[0])


def check_two_gpu_line(r, launcher):
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and launcher in r["config"]["launcher"]
    assert r["config"]["keys_per_gpu_per_step"] == 1 << 14 and r["config"]["planted_checked"] == 16
    sh = sorted(r["config"]["shards"], key=lambda x: x["worker"])
    assert [x["gpu"] for x in sh] == [0, 1] and [x["keys_per_step"] for x in sh] == [1 << 14, 1 << 14]
    assert int(sh[0]["first_key"], 16) == 0x100000000 and int(sh[1]["first_key"], 16) == 0x100000000 + (1 << 14)
    assert sum(x["planted_checked"] for x in sh) == 16 and all(x["found_per_step"] >= x["planted_checked"] for x in sh)
    assert abs(r["value"] - (1 << 15) / (r["ms_per_step"] * 1e3)) / r["value"] < 3e-2  # two-decimal rounding of a sub-1 Mkeys/s stand-in
    w = r["weak_scaling"]
    assert w["keys_per_gpu_per_step"] == 1 << 15 and abs(w["value"] - 2 * (1 << 15) / (w["ms_per_step"] * 1e3)) / w["value"] < 3e-2
    assert "cpu_baseline" not in r and "TEST STAND-IN" in r["data"]


@pytest.mark.timeout(600)
def test_gpus_2_without_a_launcher_runs_two_device_threads():
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + SMALL, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=580, cwd=ROOT, env=ENV)
    check_two_gpu_line(last_json(pr), "in-process device threads")


@pytest.mark.timeout(600)
def test_gpus_2_under_torch_distributed_run():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2"] + SMALL
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=580, cwd=ROOT, env=ENV)
    check_two_gpu_line(last_json(pr), "torch.distributed.run")


def test_single_worker_line_and_contradicting_flags():
    pr = subprocess.run([sys.executable, BENCH] + SMALL + ["--no-cpu"], stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=580, cwd=ROOT, env=ENV)
    r = last_json(pr)
    assert r["n_gpus"] == 1 and r["config"]["shards"][0]["keys_per_step"] == 1 << 15 and "weak_scaling" not in r
    # a launcher's WORLD_SIZE that contradicts --gpus is an error, not a guess
    env = dict(ENV, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    pr = subprocess.run([sys.executable, BENCH, "--gpus", "4"] + SMALL, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=120, cwd=ROOT, env=env)
    assert pr.returncode != 0 and b"contradicts" in pr.stderr


def test_more_gpus_than_visible_is_refused():
    """with the real library and no GPU in this container (or fewer than asked for anywhere): loud failure"""
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--no-cpu"] + SMALL, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=300, cwd=ROOT, env=ENV)
    assert pr.returncode != 0 and b"GPU(s) visible" in pr.stderr and not pr.stdout.strip()
