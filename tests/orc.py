"""ctypes view of oracle/liborc.so — the CPU oracle. Test infrastructure only (see oracle/orc.h)."""
import ctypes as C
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FE = C.c_uint64 * 4
H160 = C.c_uint32 * 5
MASK64 = (1 << 64) - 1
P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


class Pt(C.Structure):
    _fields_ = [("x", FE), ("y", FE), ("z", FE)]


class Filter(C.Structure):
    _fields_ = [("bits", C.POINTER(C.c_uint64)), ("size", C.c_uint64), ("list", C.POINTER(C.c_uint32)),
                ("list_count", C.c_uint64)]


class Found(C.Structure):
    _fields_ = [("h160", H160), ("compressed", C.c_uint8), ("endo", C.c_uint8), ("pad", C.c_uint8 * 2),
                ("pk", FE)]


class AddCfg(C.Structure):
    _fields_ = [("check33", C.c_int), ("check65", C.c_int), ("use_endo", C.c_int), ("ord_offs", C.c_uint32),
                ("verify", C.c_int), ("threads", C.c_int), ("rnd_jobs", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
        _lib.orc_gtable_init.restype = C.c_size_t
        _lib.orc_blf_gen_size.restype = C.c_uint64
        _lib.orc_blf_gen_size.argtypes = [C.c_uint64]
        _lib.orc_sn_add_stride.argtypes = [FE, FE, FE, C.c_uint64]
        _lib.orc_calc_priv.argtypes = [FE, FE, FE, C.c_uint64, C.c_uint8]
    return _lib


def fe(v):
    return FE(*[(v >> (64 * i)) & MASK64 for i in range(4)])


def val(f):
    return sum(int(f[i]) << (64 * i) for i in range(4))


def hex160(h):
    return "".join("%08x" % int(w) for w in h)


def point_of(k):
    x, y = FE(), FE()
    lib().orc_pt_mulg_affine(x, y, fe(k))
    return val(x), val(y)


def hash160(x, y, compressed=True):
    h = H160()
    (lib().orc_hash160_33 if compressed else lib().orc_hash160_65)(h, fe(x), fe(y))
    return [int(w) for w in h]


def parse_hash_list(path):
    """main.c:96-110: fgets into a 41-byte buffer reads 40-char CHUNKS; only full 40-char chunks count, each is
    parsed as 5 x %8x (a failed conversion leaves the previous/uninitialised word; we only accept clean hex)."""
    out = []
    data = open(path, "rb").read().decode("latin1")
    for line in data.split("\n"):
        # fgets semantics: successive 40-char chunks of a long line, then the remainder (+ newline)
        chunks = [line[i : i + 40] for i in range(0, len(line), 40)] or [""]
        for ch in chunks:
            if len(ch) != 40:
                continue
            try:
                out.append([int(ch[j : j + 8], 16) for j in range(0, 40, 8)])
            except ValueError:
                out.append(None)  # garbage entry in the reference (uninitialised words); not reproducible
    return out


class OrcFilter:
    def __init__(self, hashes=None, bloom_words=None, borrow=False):
        """borrow=True: use the caller's uint64 array in place (multi-GB filters: no second copy); it must stay alive"""
        self.f = Filter()
        self._borrowed = None
        if borrow:
            w = np.ascontiguousarray(bloom_words, dtype=np.uint64)
            self._borrowed = w
            self.f.bits = w.ctypes.data_as(C.POINTER(C.c_uint64))
            self.f.size = len(w)
        elif hashes is not None:
            arr = np.ascontiguousarray(np.array(hashes, dtype=np.uint32).reshape(-1, 5))
            lib().orc_filter_from_list(C.byref(self.f), arr.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(len(arr)))
        else:
            w = np.ascontiguousarray(bloom_words, dtype=np.uint64)
            lib().orc_filter_from_bloom(C.byref(self.f), w.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(len(w)))

    def bloom_words(self):
        return np.ctypeslib.as_array(self.f.bits, shape=(int(self.f.size),)).copy()

    def check(self, h):
        return bool(lib().orc_filter_check(C.byref(self.f), H160(*h)))

    def __del__(self):
        try:
            if self._borrowed is None:
                lib().orc_filter_free(C.byref(self.f))
        except Exception:
            pass


def found_lines(found, n):
    """Render found records exactly like ctx_write_found's outfile format (main.c:193-195)."""
    out = []
    for i in range(n):
        r = found[i]
        out.append("%s\t%s\t%064x" % ("addr33" if r.compressed else "addr65", hex160(r.h160), val(r.pk)))
    return out


def digest(lines):
    return hashlib.sha256(("\n".join(sorted(lines)) + "\n").encode()).hexdigest()


def add_range(flt, range_s, range_e, a33=True, a65=False, endo=False, offs=0, verify=True, threads=1, cap=1 << 16, rnd=False):
    """cmd_add over [range_s, range_e); rnd=True: one window of cmd_rnd (full-size jobs, main.c:624)"""
    cfg = AddCfg(int(a33), int(a65), int(endo), offs, int(verify), threads, int(rnd))
    out = (Found * cap)()
    nout, checked, hashed = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = lib().orc_add_range(C.byref(cfg), C.byref(flt.f), fe(range_s), fe(range_e), out, C.c_uint64(cap),
                             C.byref(nout), C.byref(checked), C.byref(hashed))
    return rc, out, nout.value, checked.value, hashed.value


def mul_batch(flt, scalars, a33=True, a65=False, cap=1 << 16):
    n = len(scalars)
    arr = (FE * n)(*[fe(k) for k in scalars])
    out = (Found * cap)()
    nout = C.c_uint64()
    rc = lib().orc_mul_batch(int(a33), int(a65), C.byref(flt.f), arr, C.c_uint64(n), out, C.c_uint64(cap), C.byref(nout))
    return rc, out, nout.value


def mul_hash160_many(K, a33=True, a65=False, threads=None):
    """hash160 of every scalar of K ((n, 4) uint64, little-endian limbs) the way cmd_mul's workers compute them (2048-scalar jobs:
    ec_gtable_mul, one grprdc, addr33 / addr65) -> (h33 or None, h65 or None, ok): (n, 5) uint32 each, ok[i] = 0 for k = 0 (mod n)"""
    K = np.ascontiguousarray(K, dtype=np.uint64)
    n = len(K)
    h33 = np.zeros((n, 5), np.uint32) if a33 else None
    h65 = np.zeros((n, 5), np.uint32) if a65 else None
    ok = np.zeros(n, np.uint8)
    if threads is None:
        threads = max(1, min(len(os.sched_getaffinity(0)), 128))
    lib().orc_mul_hash160_many(C.c_void_p(K.ctypes.data), C.c_uint64(n), C.c_void_p(h33.ctypes.data if a33 else None),
                               C.c_void_p(h65.ctypes.data if a65 else None), C.c_void_p(ok.ctypes.data), C.c_int(threads))
    return h33, h65, ok


def sn_from_hex(s):
    r = FE()
    lib().orc_sn_from_hex(r, s.encode())
    return val(r)
