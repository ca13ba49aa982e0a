import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from ecloop_amd import Device
d = Device(0)
for n in (10**6, 10**7, 10**8):
    a = np.random.default_rng(n).integers(0, 1 << 32, (n, 5), dtype=np.uint64).astype(np.uint32)
    for rep in range(2):
        t = time.perf_counter(); got = d.sort_list(a); dt = time.perf_counter() - t
    print("sort_list %d entries: %.3f s (second call; copy in/out included), kept %d" % (n, dt, len(got)), flush=True)
d.close()
