"""Stand-in for capi.Device built on the oracle: same surface, no GPU.  Used ONLY by CPU tests of host logic
(engine.KeySearch's launch sizing / retry / confirm / verify, bench.py's N>1 plumbing) - test infrastructure like the
oracle it wraps; nothing in the product imports it."""
import numpy as np

import orc
from ecloop_amd import capi, engine


class FakeDevice:
    """Same surface as capi.Device; key ranges are answered by the oracle through an all-ones-free path: every key of
    the requested range is hashed by the oracle's own add_range with the filter that was set."""
    calls = []

    def __init__(self, device=0, a33=True, a65=False, endo=False, ord_offs=0):
        self.a33, self.a65, self.endo, self.offs = a33, a65, endo, ord_offs
        self.words = None
        self.list = None
        self.lanes, self.half = 512, 64

    def close(self):
        pass

    def set_bloom(self, words):
        self.words = np.array(words, dtype=np.uint64)

    def set_list(self, hashes):
        self.list = None if hashes is None else {tuple(int(w) for w in h) for h in hashes}

    def reserve(self, nkeys, cap=4096):
        pass

    def set_geometry(self, half_group=0, max_lanes=0):
        self.half = half_group or self.half
        self.lanes = max_lanes or self.lanes

    def geometry(self):
        return self.half, self.lanes

    def add_range(self, start, nkeys, cap=4096):
        FakeDevice.calls.append((start, nkeys, cap))
        stride = 1 << self.offs
        flt = orc.OrcFilter(bloom_words=self.words)
        recs = []
        lam = engine.LAMBDA
        if stride == 1:
            # the oracle hashes whole 2048-key groups: ask for the covering range and keep the keys inside
            batches = [orc.add_range(flt, start, start + nkeys, a33=self.a33, a65=self.a65, endo=self.endo, verify=False,
                                     threads=4, cap=1 << 18)]
        else:
            # with a stride, a one-key-wide range makes the oracle hash exactly one 2048-key group (main.c:442)
            batches = [orc.add_range(flt, (start + g * 2048 * stride) % orc.N, (start + g * 2048 * stride) % orc.N + 1, a33=self.a33,
                                     a65=self.a65, endo=self.endo, offs=self.offs, verify=False, cap=1 << 16)
                       for g in range((nkeys + 2047) // 2048)]
        for rc, out, n, _, hashed in batches:
          assert rc == 0
          for i in range(n):
            r = out[i]
            k = orc.val(r.pk)
            if r.endo in (1, 3, 5):
                k = (-k) % orc.N
            if r.endo in (2, 3):
                k = k * pow(lam, -1, orc.N) % orc.N
            if r.endo in (4, 5):
                k = k * pow(lam, -2, orc.N) % orc.N
            off = ((k - start) % orc.N) >> self.offs
            if off < nkeys and (self.list is None or tuple(int(w) for w in r.h160) in self.list):
                recs.append((off, [int(w) for w in r.h160], r.endo, r.compressed))
        arr = np.zeros(len(recs), dtype=capi.FOUND_DTYPE)
        for j, (off, h, e, c) in enumerate(recs):
            arr[j]["key_offset"], arr[j]["h160"], arr[j]["endo"], arr[j]["compressed"] = off, h, e, c
        self.kept = arr[: max(cap, self.keep_min)]  # what the device keeps of a call (the library: max(cap, 2^20) records)
        return arr[:cap].copy(), len(recs)

    keep_min = 1 << 20
    fetches = []

    def fetch_found(self, first, n):
        FakeDevice.fetches.append((first, n))
        return self.kept[first : first + n].copy()

    def diag_mulg(self, ks):
        pts = [orc.point_of(k) for k in ks]
        return [p[0] for p in pts], [p[1] for p in pts], np.ones(len(ks), dtype=np.uint8)

    def verify(self, ks):
        xs, ys, ok = self.diag_mulg(ks)
        h33, h65 = self.diag_hash160(xs, ys)
        return h33, h65, ok

    def diag_hash160(self, xs, ys):
        return (np.array([orc.hash160(x, y, True) for x, y in zip(xs, ys)], dtype=np.uint32).reshape(-1, 5),
                np.array([orc.hash160(x, y, False) for x, y in zip(xs, ys)], dtype=np.uint32).reshape(-1, 5))

    # ---- the rest of the surface bench.py touches (filter building, timers): trivial on the CPU
    def bloom_insert(self, hashes):
        engine.blf_add_host(self.words, np.ascontiguousarray(hashes, dtype=np.uint32).reshape(-1, 5))

    def get_bloom(self, nwords):
        return self.words[:nwords].copy()

    def reset_timing(self):
        pass

    def timing(self):
        return 1.0, 1, 1

    def setup_timing(self):
        return 0.0, 0
