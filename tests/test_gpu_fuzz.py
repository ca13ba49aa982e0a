"""A short slice of tools/fuzz_gpu.py inside the GPU suite: random ranges / geometries / selections / strides / filters,
found lists compared with the oracle (the long runs are recorded under profiles/)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_random_scans_equal_the_oracle(seed):
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "8", str(seed)], stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, timeout=300)
    out = pr.stdout.decode(errors="replace")
    assert pr.returncode == 0 and "ALL EQUAL" in out, out[-2000:]


@pytest.mark.gpu
def test_random_mul_batches_equal_the_oracle():
    """a short slice of tools/fuzz_mul_gpu.py: every scalar of every trial against orc.mul_hash160_many (never a device kernel)"""
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_mul_gpu.py"), "12", "31"], stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, timeout=600)
    out = pr.stdout.decode(errors="replace")
    assert pr.returncode == 0 and "ALL EQUAL" in out, out[-2000:]


@pytest.mark.gpu
def test_random_job_sequences_through_the_lookahead_equal_plain_launches():
    """a short slice of tools/fuzz_lookahead_gpu.py: scans handed out job by job (sizes, strides, selections, filters, sweep limits, worker
    threads on several contexts, jumps and odd jobs at random) - every call's records equal the same call's on a context without look-ahead"""
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_lookahead_gpu.py"), "15", "41"], stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, timeout=900)
    out = pr.stdout.decode(errors="replace")
    assert pr.returncode == 0 and "ALL EQUAL" in out, out[-2000:]
