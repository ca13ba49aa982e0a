#!/usr/bin/env python3
"""Differential fuzz of the look-ahead (ecloop_amd/csrc/abi_lookahead.h) on the GPU box: random scans handed to the library job by job
the way the reference's scheduler does (main.c:405-435) - job sizes 2^11 .. 2^22, strides 2^0 .. 2^200, address / endo selections,
filters of several densities, sweep limits 2^22 .. 2^30, the scan's end told or not, 1 .. 4 worker threads on as many contexts pulling
from one counter, now and then a jump, a skipped job, a job of another size, a job that straddles what was swept - and every call's
records must equal those of the same call on a context with the look-ahead switched off (plain launches: the path the golden dumps and
tools/fuzz_gpu.py pin to the reference and the oracle).  A sample of the delivered hits is also put to the oracle directly.
usage: python tools/fuzz_lookahead_gpu.py [seconds=120] [seed=1]      -> gpurun_out/fuzz_lookahead.txt"""
import os
import random
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from ecloop_amd import Device  # noqa: E402
from synth import synth_bloom_words  # noqa: E402


def key(recs):
    return sorted((int(r["key_offset"]), int(r["endo"]), int(r["compressed"]), tuple(int(v) for v in r["h160"])) for r in recs)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rnd = random.Random(seed)
    t_end, trials, calls, served, swept, hits, checked = time.time() + budget, 0, 0, 0, 0, 0, 0
    while time.time() < t_end:
        a33, a65 = rnd.choice([(True, False), (True, False), (False, True), (True, True)])
        endo = rnd.random() < 0.3
        offs = rnd.choice([0, 0, 0, 1, 7, 64, 128, 200])
        nw = rnd.choice([64, 4099, 65539, (1 << 20) + 7])
        mode = rnd.choice(["a|(b&c)", "a|(b&c)", "a|b", "a", "a&(b|c)"])
        words = synth_bloom_words(nw, rnd.randrange(1 << 30), mode)
        job = 1 << rnd.choice([11, 12, 14, 16, 18, 20, 21, 22])
        if mode == "a|b" and job > (1 << 18):
            job = 1 << 18
        njobs = rnd.choice([3, 9, 40, 200, 700]) if job <= (1 << 16) else rnd.choice([3, 9, 40, 100])
        la_max = 1 << rnd.choice([22, 24, 26, 30])
        nctx = rnd.choice([1, 1, 2, 4])
        hint = rnd.random() < 0.6
        A = rnd.randrange(1 << 20, 1 << (250 - offs)) << offs | rnd.randrange(1 << offs) if offs else rnd.randrange(1 << 30, 1 << 250)
        kw = dict(a33=a33, a65=a65, endo=endo, ord_offs=offs)
        plain = Device(0, **kw)
        plain.set_lookahead(0)
        plain.set_bloom(words)
        ctxs = [Device(0, **kw) for _ in range(nctx)]
        per = (1 if a33 else 0) + (1 if a65 else 0)
        cap = 1 << 15
        # the sequence of calls: mostly the next job, sometimes something else
        seq, pos = [], 0
        for _ in range(njobs):
            r = rnd.random()
            if r < 0.03:
                pos += rnd.randrange(1, 6) * job                      # a jump ahead
            elif r < 0.05 and pos > 4 * job:
                seq.append((pos - rnd.randrange(1, 4) * job, job))    # back into what was covered
            elif r < 0.07:
                seq.append((pos + job // 2, job))                     # straddles two jobs
            elif r < 0.09:
                seq.append((pos, rnd.choice([1, 1000, job // 2])))    # another size
            seq.append((pos, job))
            pos += job
        end = A + ((pos - rnd.randrange(0, job)) << offs) if hint else None
        open(os.path.join(ROOT, "gpurun_out", "fuzz_lookahead_last_trial.txt"), "w").write("seed %d trial %d %r\n" % (
            seed, trials, dict(a33=a33, a65=a65, endo=endo, offs=offs, nw=nw, mode=mode, job=job, njobs=njobs, la_max=la_max, nctx=nctx, hint=hint, A=hex(A))))
        results, errors, lock, state = {}, [], threading.Lock(), {"next": 0}

        def worker(d):
            try:
                d.set_lookahead(la_max)
                d.set_bloom(words)
                d.set_scan_end(end)
                while True:
                    with lock:
                        i = state["next"]
                        state["next"] += 1
                    if i >= len(seq):
                        return
                    o, n = seq[i]
                    got, total = d.add_range((A + (o << offs)) % orc.N, n, cap=cap)
                    if total > cap:
                        got = np.concatenate([got, d.fetch_found(cap, total - cap)])
                    results[i] = (key(got), total)
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        try:
            ts = [threading.Thread(target=worker, args=(d,)) for d in ctxs]
            [t.start() for t in ts]
            [t.join() for t in ts]
            if errors:
                raise errors[0]
            flt = orc.OrcFilter(bloom_words=words)
            for i, (o, n) in enumerate(seq):
                want, total = plain.add_range((A + (o << offs)) % orc.N, n, cap=cap)
                if total > cap:
                    want = np.concatenate([want, plain.fetch_found(cap, total - cap)])
                if results[i] != (key(want), total):
                    print("MISMATCH", dict(seed=seed, trial=trials, call=i, offset=o, n=n, got=results[i][1], want=total))
                    sys.exit(1)
                hits += total
                if i % 17 == 0 and total and not endo and offs == 0:  # the oracle on a few of the delivered hits
                    for off, _, comp, h in results[i][0][:8]:
                        x, y = orc.point_of((A + o + off) % orc.N)
                        if tuple(orc.hash160(x, y, bool(comp))) != h or not flt.check(list(h)):
                            print("ORACLE MISMATCH", dict(seed=seed, trial=trials, call=i, key=hex(A + o + off)))
                            sys.exit(1)
                        checked += 1
            st = [d.lookahead_stats() for d in ctxs]
            served += sum(s[2] for s in st)
            swept += sum(s[1] for s in st)
            calls += len(seq)
        finally:
            plain.close()
            [d.close() for d in ctxs]
        trials += 1
    line = ("# tools/fuzz_lookahead_gpu.py %s %d: %d trials, %d calls (%d answered from sweeps, %d keys swept), %d compared hits, ALL EQUAL to plain launches; "
            "%d delivered hits re-derived by the oracle" % (budget, seed, trials, calls, served, swept, hits, checked))
    open(os.path.join(ROOT, "gpurun_out", "fuzz_lookahead.txt"), "w").write(line + "\n")
    print(line)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
