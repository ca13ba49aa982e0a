"""Host-side mirror of the reference's command drivers (vladkens/ecloop main.c), driving the GPU through the C ABI.

What lives here is what the reference keeps on the host: filter loading, job arithmetic, the
sorted-list confirm, calc_priv, the pk_verify_hash self-check, the found sink, range sharding across GPUs.
All curve / hash work goes to the device (ecloop_amd.capi.Device); nothing here can compute a hash160 on the CPU.
Names follow the reference: load_filter, cmd_add, cmd_mul, calc_priv, pk_verify_hash, ctx_write_found.
"""
import math
import os
import struct

import numpy as np

from .capi import Device, EclError

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
P = 2**256 - 2**32 - 977
LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72  # A1, lib/ecc.c:36 (A2 = A1^2 mod n)
GROUP_INV_SIZE = 2048  # main.c:17
MAX_JOB_SIZE = 2 * 1024 * 1024  # main.c:16
BLF_MAGIC, BLF_VERSION = 0x45434246, 1  # lib/utils.c:274-275

# ----------------------------------------------------------------------------------------------- bloom filter (host side)


def blf_size_words(n):
    """lib/utils.c:421-427: m = n*ln(1e-9)/ln(1/2^ln2) bits, ceil to 64-bit words"""
    p = 1.0 / float(1000000000)
    m = int(n * math.log(p) / math.log(1.0 / math.pow(2.0, math.log(2.0))))
    return (m + 63) // 64


def blf_indices(h160):
    """lib/utils.c:290-306: the 20 probe positions of each hash (N x 5 uint32 words) -> N x 20 uint64"""
    h = np.ascontiguousarray(h160, dtype=np.uint32).reshape(-1, 5).astype(np.uint64)
    a = [h[:, 0] << np.uint64(32) | h[:, 1], h[:, 2] << np.uint64(32) | h[:, 3], h[:, 4] << np.uint64(32) | h[:, 0],
         h[:, 1] << np.uint64(32) | h[:, 2], h[:, 3] << np.uint64(32) | h[:, 4]]
    cols = []
    for s in (24, 28, 36, 40):
        for j in range(5):
            cols.append((a[j] << np.uint64(s)) | (a[(j + 1) % 5] >> np.uint64(s)))
    return np.stack(cols, axis=1)


def blf_add_host(words, h160):
    """blf_add for small lists on the host (bit-identical to the device bulk insert)"""
    idx = blf_indices(h160).reshape(-1)
    size = np.uint64(len(words))
    np.bitwise_or.at(words, ((idx >> np.uint64(6)) % size).astype(np.int64), np.uint64(1) << (idx & np.uint64(63)))


def blf_save(path, words):
    """lib/utils.c:328-360"""
    words = np.ascontiguousarray(words, dtype="<u8")
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQ", BLF_MAGIC, BLF_VERSION, len(words)))
        f.write(words.tobytes())


def blf_load(path):
    """lib/utils.c:362-396"""
    with open(path, "rb") as f:
        head = f.read(16)
        if len(head) != 16:
            raise ValueError("failed to read bloom filter header")
        magic, ver, size = struct.unpack("<IIQ", head)
        if magic != BLF_MAGIC or ver != BLF_VERSION:
            raise ValueError("invalid bloom filter version; create a new filter with blf-gen command")
        words = np.fromfile(f, dtype="<u8", count=size)
    if len(words) != size:
        raise ValueError("failed to read bloom filter bits")
    return words.astype(np.uint64)


class Filter:
    """ctx->blf + ctx->to_find_hashes (main.c:48-51). `hashes` is None in bloom-only mode."""

    def __init__(self, words, hashes=None):
        self.words = np.ascontiguousarray(words, dtype=np.uint64)
        self.hashes = hashes  # sorted unique (count x 5) uint32, or None
        # big-endian bytes of the 5 words compare like compare_160 (lib/addr.c:18-26): binary search on 20-byte keys
        self._keys = None if hashes is None else np.ascontiguousarray(np.asarray(hashes, dtype=">u4")).view("S20").reshape(-1)

    @property
    def count(self):
        return 0 if self.hashes is None else len(self.hashes)

    def confirm(self, h160):
        """second stage of ctx_check_hash (main.c:212-216): exact membership in list mode, always true in bloom mode"""
        if self._keys is None:
            return True
        key = np.asarray([int(v) for v in h160], dtype=">u4").tobytes()
        i = int(np.searchsorted(self._keys, np.bytes_(key)))
        return i < len(self._keys) and self._keys[i].ljust(20, b"\0") == key


def parse_hash_list(path):
    """main.c:96-110. fgets into a 41-byte buffer consumes a long line in 40-character chunks and every full chunk
    becomes an entry; short chunks are skipped.  Chunks that are not clean hex (the reference sscanf's garbage out of
    them: a quirk, it inflates the banner count by one for data/btc-bw-hash) are dropped here."""
    out = []
    with open(path, "rb") as f:
        for raw in f.read().decode("latin1").split("\n"):
            for i in range(0, max(len(raw), 1), 40):
                ch = raw[i : i + 40]
                if len(ch) != 40:
                    continue
                try:
                    out.append([int(ch[j : j + 8], 16) for j in range(0, 40, 8)])
                except ValueError:
                    pass
    return out


def load_filter(path):
    """main.c:71-131: `.blf` -> bloom-only mode; anything else -> hex list, sorted + deduplicated, plus an in-memory
    bloom of 2*count words."""
    if not path:
        raise ValueError("missing filter file")
    if not os.path.exists(path):
        raise ValueError(f"failed to open filter file: {path}")
    if os.path.splitext(path)[1] == ".blf":
        return Filter(blf_load(path))
    hs = np.array(parse_hash_list(path), dtype=np.uint32).reshape(-1, 5)
    if len(hs) == 0:
        raise ValueError("empty hash list")
    hs = np.unique(hs, axis=0)  # lexicographic on the 5 words == compare_160 (lib/addr.c:18-26)
    words = np.zeros(len(hs) * 2, dtype=np.uint64)
    blf_add_host(words, hs)
    return Filter(words, hs)


# ----------------------------------------------------------------------------------------------- scalar helpers (host)


def calc_priv(start, stride, off, endo):
    """main.c:267-276 with python integers (the reference's non-reducing fe_modn_add is a representation quirk:
    the value is the same residue mod n)"""
    k = (start + off * stride) % N
    if endo in (2, 3):
        k = k * LAMBDA % N
    elif endo in (4, 5):
        k = k * LAMBDA % N * LAMBDA % N
    if endo in (1, 3, 5):
        k = (-k) % N
    return k


def parse_range(raw):
    """arg_search_range (main.c:666-701): hex `A:B`, A > 0x800, B <= p, A < B; default 0x800:p"""
    if raw is None:
        return GROUP_INV_SIZE, P
    if ":" not in raw:
        raise ValueError("invalid search range, use format: -r 8000:ffff")
    a, b = raw.split(":", 1)
    rs, re_ = scalar_from_hex(a), scalar_from_hex(b)
    if rs <= GROUP_INV_SIZE:
        raise ValueError("invalid search range, start <= 0x800")
    if re_ > P:
        raise ValueError("invalid search range, end > FE_P")
    if rs >= re_:
        raise ValueError("invalid search range, start >= end")
    return rs, re_


def scalar_from_hex(s):
    """fe_modn_from_hex (lib/ecc.c:81-95,262-265): right-to-left, non-hex characters skipped, at most 64 digits"""
    digits = [c for c in s if c in "0123456789abcdefABCDEF"][-64:]
    v = int("".join(digits), 16) if digits else 0
    return v - N if v >= N else v


def job_plan(range_s, range_e, stride=1):
    """cmd_add + cmd_add_worker (main.c:405-454): job_size = min(B-A, 2^21) (scalar units, whatever the stride);
    a job advances range_s by job_size*stride and jobs are handed out until range_s >= range_e; each job hashes
    ceil(job_size/2048)*2048 keys.  Returns (job_size, njobs, keys_hashed): the keys actually hashed are the
    contiguous run range_s + i*stride, i < keys_hashed."""
    span = range_e - range_s
    job = span if span < MAX_JOB_SIZE else MAX_JOB_SIZE
    njobs = (span + job * stride - 1) // (job * stride)
    per_job = (job + GROUP_INV_SIZE - 1) // GROUP_INV_SIZE * GROUP_INV_SIZE
    hashed = (njobs - 1) * job + per_job
    return job, njobs, hashed


def shard(total, rank, world, align=GROUP_INV_SIZE):
    """contiguous range partition of `total` keys over `world` GPUs: (offset, count) for `rank`"""
    per = (total + world - 1) // world
    per = (per + align - 1) // align * align
    lo = min(total, rank * per)
    hi = min(total, lo + per)
    return lo, hi - lo


def gather_found(lines, dist=None):
    """found lists of all ranks -> every rank (the reference's single found sink, main.c:182-203); host objects only"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(lines)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(lines))
    return [l for part in parts for l in part]


def max_over_ranks(seconds, dist=None):
    """wall time of the slowest rank (the job is done when the last shard is)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------------------------- command drivers


class FoundRecord:
    __slots__ = ("label", "h160", "pk")

    def __init__(self, label, h160, pk):
        self.label, self.h160, self.pk = label, h160, pk

    def line(self):
        """outfile format of ctx_write_found (main.c:193-195)"""
        return "%s\t%s\t%064x" % (self.label, "".join("%08x" % int(w) for w in self.h160), self.pk)

    def stdout_line(self):
        """stdout format (main.c:187-189)"""
        return "%s: %s <- %064x" % (self.label, "".join("%08x" % int(w) for w in self.h160), self.pk)


class KeySearch:
    """ctx_t + cmd_add / cmd_mul for one GPU."""

    def __init__(self, flt, device=0, a33=True, a65=False, endo=False, ord_offs=0, verify=True, launch_keys=1 << 32,
                 half_group=0, max_lanes=0, device_cls=None):
        if not (a33 or a65):
            a33 = True  # main.c:825-827
        self.flt, self.a33, self.a65, self.endo, self.offs, self.verify = flt, a33, a65, endo, ord_offs, verify
        self.stride = 1 << ord_offs
        # device_cls: the GPU context (capi.Device); the CPU tests of the host logic pass a stand-in with the same surface
        self.dev = (device_cls or Device)(device, a33=a33, a65=a65, endo=endo, ord_offs=ord_offs)
        if half_group or max_lanes:
            self.dev.set_geometry(half_group, max_lanes)
        self.dev.set_bloom(flt.words)
        if flt.hashes is not None:
            self.dev.set_list(flt.hashes)  # exact confirm on the device too (main.c:212-216); the host check stays
        self.launch_keys = launch_keys
        self.k_checked = 0
        self.k_found = 0
        self.found = []

    def close(self):
        self.dev.close()

    # pk_verify_hash (main.c:248-263): re-derive the hit from its scalar by the window-table sum (not the walk kernel;
    # the self-test of the context pins that sum against the double-and-add kernel)
    def _verify(self, recs):
        if not recs:
            return
        h33, h65, ok = self.dev.verify([r.pk for r in recs])
        for i, r in enumerate(recs):
            h = h33[i] if r.label == "addr33" else h65[i]
            if not ok[i] or [int(v) for v in h] != [int(v) for v in r.h160]:
                raise EclError("[!] error: hash mismatch (%s) pk: %064x" % (r.label, r.pk))

    def _collect(self, raw, start):
        recs = []
        for r in raw:
            if not self.flt.confirm(r["h160"]):
                continue
            pk = calc_priv(start, self.stride, int(r["key_offset"]), int(r["endo"]))
            recs.append(FoundRecord("addr33" if r["compressed"] else "addr65", [int(v) for v in r["h160"]], pk))
        if self.verify:
            self._verify(recs)
        self.found.extend(recs)
        self.k_found += len(recs)
        return recs

    def add_keys(self, start, nkeys, cap=4096):
        """hash exactly nkeys keys from `start`, in launches of at most launch_keys keys"""
        b, lanes = self.dev.geometry()
        sweep = lanes * 2 * b  # launches that are whole sweeps keep every lane busy and continue without re-init
        per = max(sweep, self.launch_keys // sweep * sweep)
        done = 0
        while done < nkeys:
            n = nkeys - done
            if n > self.launch_keys:  # what fits one launch goes as one call: the library then sizes the lanes to it
                n = min(n, per)
            s = (start + done * self.stride) % N
            c = cap
            while True:
                raw, total = self.dev.add_range(s, n, cap=c)
                if total <= c:
                    break
                # overflow (dense filters only): the device kept up to max(cap, 2^20) records of the call - read the rest; only
                # if there were more than that is the launch repeated with a buffer that fits
                rest = self.dev.fetch_found(c, total - c)
                if len(rest) == total - c:
                    raw = np.concatenate([raw, rest])
                    break
                c = total
            self._collect(raw, s)
            done += n

    def cmd_add(self, range_s, range_e, rank=0, world=1):
        """reference semantics of `add -r A:B` (status counter included), the scan sharded over `world` GPUs"""
        job, njobs, hashed = job_plan(range_s, range_e, self.stride)
        lo, cnt = shard(hashed, rank, world)
        self.add_keys((range_s + lo * self.stride) % N, cnt)
        if rank == 0:
            self.k_checked += njobs * job * (6 if self.endo else 1)  # main.c:431
        return self.found

    def cmd_mul(self, scalars, cap=4096):
        """cmd_mul_worker body (main.c:530-535) for already parsed scalars"""
        chunk = 1 << 16
        for at in range(0, len(scalars), chunk):
            ks = scalars[at : at + chunk]
            c = max(cap, 2 * len(ks))
            raw, total = self.dev.mul_batch(ks, cap=c)
            for r in raw:
                if self.flt.confirm(r["h160"]):
                    self.found.append(FoundRecord("addr33" if r["compressed"] else "addr65", [int(v) for v in r["h160"]],
                                                  ks[int(r["key_offset"])]))
                    self.k_found += 1
            self.k_checked += len(ks)
        return self.found


def blf_gen(hashes, n, existing=None, device=0):
    """blf-gen (lib/utils.c:409-475): bloom of blf_size_words(n) words holding `hashes`; bulk insert on the GPU."""
    size = blf_size_words(n)
    words = np.zeros(size, dtype=np.uint64) if existing is None else np.ascontiguousarray(existing, dtype=np.uint64)
    if len(words) != size:
        raise ValueError("bloom filter size mismatch (%d != %d)" % (len(words), size))
    d = Device(device)
    try:
        d.set_bloom(words)
        d.bloom_insert(hashes)
        return d.get_bloom(size)
    finally:
        d.close()
