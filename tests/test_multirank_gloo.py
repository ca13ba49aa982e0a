"""N>1 path on CPU: two processes (gloo) each scan their shard of the range and gather the found lists; the union
must equal the single-rank result and the reference's golden list.  The GPU call is replaced by the oracle here (no
GPU in this container); what is under test is the sharding / gather / timing-reduction logic bench.py and the CLI use."""
import json
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import orc
    from ecloop_amd.engine import gather_found, job_plan, max_over_ranks, shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs, re_ = 0x8000, 0xFFFFF
        job, njobs, hashed = job_plan(rs, re_)
        lo, cnt = shard(hashed, rank, world)
        flt = orc.OrcFilter(hashes=[h for h in orc.parse_hash_list(os.path.join(GOLD, "btc-puzzles-hash")) if h])
        rc, out, n, _, h = orc.add_range(flt, rs + lo, rs + lo + cnt, threads=2)
        assert rc == 0 and h == cnt
        lines = gather_found(orc.found_lines(out, n), dist)
        t = max_over_ranks(1.0 + rank, dist)
        dist.barrier()
        if rank == 0:
            q.put((sorted(lines), t, hashed))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_cover_the_range_and_gather():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    lines, t, hashed = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = json.load(open(os.path.join(GOLD, "golden.json")))["cases"]
    want = [l for l in g["make_add_8000_ffffff"]["lines"] if int(l.split("\t")[2], 16) <= 0xFFFFF + 2048]
    assert lines == sorted(want) and len(lines) == 5
    assert t == 2.0 and hashed == 1015808
