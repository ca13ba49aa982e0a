"""ctypes binding of the C ABI in include/ecloop_hip.h (libecloop_hip.so, built in-tree by ecloop_amd.build).

There is no CPU fallback: if the HIP library is missing or a call fails, this raises."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ECLOOP_HIP_LIB") or os.path.join(PKG, "libecloop_hip.so")  # override: A/B builds

ADDR33, ADDR65, ENDO = 1, 2, 4
E_OVERFLOW = -4

U64x4 = C.c_uint64 * 4


class Found(C.Structure):
    _fields_ = [("key_offset", C.c_uint64), ("h160", C.c_uint32 * 5), ("endo", C.c_uint8),
                ("compressed", C.c_uint8), ("pad", C.c_uint8 * 2)]


FOUND_DTYPE = np.dtype([("key_offset", "<u8"), ("h160", "<u4", (5,)), ("endo", "u1"), ("compressed", "u1"),
                        ("pad", "u1", (2,))])
assert FOUND_DTYPE.itemsize == C.sizeof(Found) == 32

EXPORTS = [
    "ecl_hip_device_count", "ecl_hip_open", "ecl_hip_close", "ecl_hip_set_bloom", "ecl_hip_set_list", "ecl_hip_reserve", "ecl_hip_add_range",
    "ecl_hip_mul_batch", "ecl_hip_bloom_insert", "ecl_hip_get_bloom", "ecl_hip_set_geometry", "ecl_hip_get_geometry", "ecl_hip_get_timing", "ecl_hip_reset_timing", "ecl_hip_selftest", "ecl_hip_strerror",
    "ecl_hip_last_error", "ecl_hip_diag_fe", "ecl_hip_diag_mulg", "ecl_hip_diag_hash160", "ecl_hip_diag_bloom", "ecl_hip_diag_bloom_mod", "ecl_hip_set_lookahead", "ecl_hip_set_scan_end", "ecl_hip_get_lookahead_stats",
    "ecl_hip_get_setup_timing", "ecl_hip_get_mul_timing", "ecl_hip_bloom_insert_count", "ecl_hip_alloc_host", "ecl_hip_free_host", "ecl_hip_verify", "ecl_hip_sort_list", "ecl_hip_reserve_mul", "ecl_hip_mul_batch_raw", "ecl_hip_set_mul_window", "ecl_hip_get_mul_window", "ecl_hip_fetch_found", "ecl_hip_plan_geometry",
]

_lib = None


class EclError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EclError(f"{LIB_PATH} is missing: build it with `python -m ecloop_amd.build` "
                       "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    lib.ecl_hip_device_count.restype = C.c_int
    lib.ecl_hip_open.argtypes = [C.POINTER(P), C.c_int, C.c_uint32, C.c_uint32]
    lib.ecl_hip_close.argtypes = [P]
    lib.ecl_hip_close.restype = None
    lib.ecl_hip_set_bloom.argtypes = [P, C.c_void_p, C.c_uint64]
    lib.ecl_hip_set_list.argtypes = [P, C.c_void_p, C.c_uint64]
    lib.ecl_hip_reserve.argtypes = [P, C.c_uint64, C.c_uint32]
    lib.ecl_hip_add_range.argtypes = [P, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ecl_hip_mul_batch.argtypes = [P, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ecl_hip_fetch_found.argtypes = [P, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ecl_hip_bloom_insert.argtypes = [P, C.c_void_p, C.c_uint64]
    lib.ecl_hip_get_bloom.argtypes = [P, C.c_void_p, C.c_uint64]
    lib.ecl_hip_bloom_insert_count.argtypes = [P, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.ecl_hip_set_geometry.argtypes = [P, C.c_uint32, C.c_uint32]
    lib.ecl_hip_get_geometry.argtypes = [P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.ecl_hip_plan_geometry.argtypes = [P, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.ecl_hip_get_timing.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.ecl_hip_reset_timing.argtypes = [P]
    lib.ecl_hip_get_setup_timing.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.ecl_hip_get_mul_timing.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.ecl_hip_set_lookahead.argtypes = [P, C.c_uint64]
    lib.ecl_hip_set_scan_end.argtypes = [P, C.c_void_p]
    lib.ecl_hip_get_lookahead_stats.argtypes = [P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.ecl_hip_set_mul_window.argtypes = [P, C.c_uint32]
    lib.ecl_hip_reserve_mul.argtypes = [P, C.c_uint32, C.c_uint32]
    lib.ecl_hip_sort_list.argtypes = [P, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.ecl_hip_mul_batch_raw.argtypes = [P, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.ecl_hip_get_mul_window.argtypes = [P, C.POINTER(C.c_uint32)]
    lib.ecl_hip_verify.argtypes = [P, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ecl_hip_alloc_host.argtypes = [C.c_size_t]
    lib.ecl_hip_alloc_host.restype = C.c_void_p
    lib.ecl_hip_free_host.argtypes = [C.c_void_p]
    lib.ecl_hip_free_host.restype = None
    lib.ecl_hip_selftest.argtypes = [P]
    lib.ecl_hip_strerror.argtypes = [C.c_int]
    lib.ecl_hip_strerror.restype = C.c_char_p
    lib.ecl_hip_last_error.argtypes = [P]
    lib.ecl_hip_last_error.restype = C.c_char_p
    lib.ecl_hip_diag_fe.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.ecl_hip_diag_mulg.argtypes = [P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.ecl_hip_diag_hash160.argtypes = [P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.ecl_hip_diag_bloom.argtypes = [P, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.ecl_hip_diag_bloom_mod.argtypes = [P, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
    _lib = lib
    return lib


def limbs(v):
    """python int -> 4 little-endian u64 limbs (the reference's `fe`)"""
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limbs_array(vals):
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64).reshape(-1, 4)


def ints_of(arr):
    return [sum(int(r[i]) << (64 * i) for i in range(4)) for r in arr]


class Device:
    """One GPU context (ecl_hip handle)."""

    def __init__(self, device=0, a33=True, a65=False, endo=False, ord_offs=0):
        self.lib = load()
        self.h = C.c_void_p()
        self.a33, self.a65, self.endo = bool(a33), bool(a65), bool(endo)
        flags = (ADDR33 if a33 else 0) | (ADDR65 if a65 else 0) | (ENDO if endo else 0)
        rc = self.lib.ecl_hip_open(C.byref(self.h), device, flags, ord_offs)
        if rc != 0:
            msg = self.lib.ecl_hip_last_error(self.h).decode() if self.h else ""
            if self.h:
                self.lib.ecl_hip_close(self.h)
                self.h = None
            raise EclError(f"ecl_hip_open(device={device}): {self.lib.ecl_hip_strerror(rc).decode()} {msg}")

    def _chk(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise EclError(f"{self.lib.ecl_hip_strerror(rc).decode()}: {self.lib.ecl_hip_last_error(self.h).decode()}")
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.lib.ecl_hip_close(self.h)
            self.h = None

    __del__ = close

    def set_bloom(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        self._chk(self.lib.ecl_hip_set_bloom(self.h, w.ctypes.data, len(w)))

    def set_list(self, hashes):
        """sorted unique (n x 5) uint32 hash list for the on-device exact confirm; None / empty removes it"""
        H = np.zeros((0, 5), np.uint32) if hashes is None else np.ascontiguousarray(hashes, dtype=np.uint32).reshape(-1, 5)
        self._chk(self.lib.ecl_hip_set_list(self.h, H.ctypes.data if len(H) else None, len(H)))

    def reserve(self, nkeys, cap=4096):
        """allocate the device buffers of a later add_range(nkeys) now"""
        self._chk(self.lib.ecl_hip_reserve(self.h, nkeys, cap))

    def bloom_insert(self, hashes):
        H = np.ascontiguousarray(hashes, dtype=np.uint32).reshape(-1, 5)
        self._chk(self.lib.ecl_hip_bloom_insert(self.h, H.ctypes.data, len(H)))

    def bloom_insert_count(self, hashes):
        """blf-gen's insert loop (utils.c:455-470) on the device -> number of hashes that were new, in input order"""
        H = np.ascontiguousarray(hashes, dtype=np.uint32).reshape(-1, 5)
        added = C.c_uint64()
        self._chk(self.lib.ecl_hip_bloom_insert_count(self.h, H.ctypes.data, len(H), C.byref(added)))
        return added.value

    def get_bloom(self, nwords):
        w = np.zeros(nwords, dtype=np.uint64)
        self._chk(self.lib.ecl_hip_get_bloom(self.h, w.ctypes.data, nwords))
        return w

    def set_geometry(self, half_group=0, max_lanes=0):
        self._chk(self.lib.ecl_hip_set_geometry(self.h, half_group, max_lanes))

    def geometry(self):
        """-> (half_group, lanes); one sweep = lanes * 2 * half_group keys"""
        b, t = C.c_uint32(), C.c_uint32()
        self._chk(self.lib.ecl_hip_get_geometry(self.h, C.byref(b), C.byref(t)))
        return b.value, t.value

    def plan_geometry(self, nkeys):
        """-> (half_group, lanes, groups per lane) that add_range would use for a call of nkeys keys"""
        b, t, nb = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.lib.ecl_hip_plan_geometry(self.h, nkeys, C.byref(b), C.byref(t), C.byref(nb)))
        return b.value, t.value, nb.value

    def add_range(self, start, nkeys, cap=4096):
        """-> (records as numpy structured array, total hit count). Raises on overflow unless total <= cap."""
        out = np.zeros(cap, dtype=FOUND_DTYPE)
        n = C.c_uint32()
        s = limbs(start)
        rc = self.lib.ecl_hip_add_range(self.h, s.ctypes.data, nkeys, out.ctypes.data, cap, C.byref(n))
        self._chk(rc, allow=(E_OVERFLOW,))
        return out[: min(n.value, cap)], n.value

    def set_lookahead(self, max_keys):
        """look-ahead over small contiguous jobs: sweeps of up to max_keys keys (0 = off, else 2^22 .. 2^32; default 2^30)"""
        self._chk(self.lib.ecl_hip_set_lookahead(self.h, max_keys))

    def set_scan_end(self, end):
        """hint: the scalar at which the scan stops handing out jobs (None withdraws it)"""
        e = limbs(end) if end is not None else None
        self._chk(self.lib.ecl_hip_set_scan_end(self.h, e.ctypes.data if e is not None else None))

    def lookahead_stats(self):
        """-> (sweeps run by this context, their keys, calls answered from a sweep, their keys)"""
        v = [C.c_uint64() for _ in range(4)]
        self._chk(self.lib.ecl_hip_get_lookahead_stats(self.h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def fetch_found(self, first, n):
        """records [first, first + n) of the last add_range / mul_batch call that are still on the device (after an overflow:
        the call itself delivered [0, cap)); fewer come back only if the call produced more than the device kept"""
        out = np.zeros(n, dtype=FOUND_DTYPE)
        got = C.c_uint32()
        self._chk(self.lib.ecl_hip_fetch_found(self.h, first, out.ctypes.data, n, C.byref(got)))
        return out[: got.value]

    def mul_batch(self, scalars, cap=4096):
        k = limbs_array(scalars)
        out = np.zeros(cap, dtype=FOUND_DTYPE)
        n = C.c_uint32()
        rc = self.lib.ecl_hip_mul_batch(self.h, k.ctypes.data, len(k), out.ctypes.data, cap, C.byref(n))
        self._chk(rc, allow=(E_OVERFLOW,))
        return out[: min(n.value, cap)], n.value

    def mul_batch_raw(self, lines, cap=4096):
        """`mul -raw`: lines = list of bytes objects; their SHA-256 digests are the scalars (hashed on the device)"""
        text = b"".join(lines)
        table = np.zeros(len(lines), dtype=np.uint64)
        at = 0
        for i, l in enumerate(lines):
            table[i] = at | (len(l) << 32)
            at += len(l)
        buf = np.frombuffer(text, dtype=np.uint8) if text else np.zeros(1, dtype=np.uint8)
        out = np.zeros(cap, dtype=FOUND_DTYPE)
        n = C.c_uint32()
        rc = self.lib.ecl_hip_mul_batch_raw(self.h, buf.ctypes.data, len(text), table.ctypes.data, len(lines), out.ctypes.data, cap, C.byref(n))
        self._chk(rc, allow=(E_OVERFLOW,))
        return out[: min(n.value, cap)], n.value

    def set_mul_window(self, bits):
        """window width of this context's `mul` table (8..29 bits; 0 = automatic: 22, then 26 after 2^30 scalars)"""
        self._chk(self.lib.ecl_hip_set_mul_window(self.h, bits))

    def sort_list(self, h160):
        """(n, 5) uint32 hash160 words -> the sorted (compare_160 order), duplicate-free entries"""
        a = np.ascontiguousarray(h160, dtype=np.uint32).copy()
        kept = C.c_uint64()
        self._chk(self.lib.ecl_hip_sort_list(self.h, a.ctypes.data, len(a), C.byref(kept)))
        return a[: kept.value]

    def reserve_mul(self, n, cap=4096):
        self._chk(self.lib.ecl_hip_reserve_mul(self.h, n, cap))

    def mul_window(self):
        bits = C.c_uint32()
        self._chk(self.lib.ecl_hip_get_mul_window(self.h, C.byref(bits)))
        return bits.value

    def verify(self, ks):
        """pk_verify_hash for a batch: -> (h33, h65, ok) of the scalars' public keys (window-table path)"""
        K = limbs_array(ks)
        h33 = np.zeros((len(K), 5), dtype=np.uint32)
        h65 = np.zeros((len(K), 5), dtype=np.uint32)
        ok = np.zeros(len(K), dtype=np.uint8)
        self._chk(self.lib.ecl_hip_verify(self.h, K.ctypes.data, len(K), h33.ctypes.data, h65.ctypes.data, ok.ctypes.data))
        return h33, h65, ok

    def selftest(self):
        self._chk(self.lib.ecl_hip_selftest(self.h))

    def timing(self):
        ms, launches, keys = C.c_double(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.ecl_hip_get_timing(self.h, C.byref(ms), C.byref(launches), C.byref(keys)))
        return ms.value, launches.value, keys.value

    def reset_timing(self):
        self._chk(self.lib.ecl_hip_reset_timing(self.h))

    def setup_timing(self):
        """-> (ms spent in the set-up kernels of non-contiguous add_range calls, number of such calls)"""
        ms, n = C.c_double(), C.c_uint64()
        self._chk(self.lib.ecl_hip_get_setup_timing(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def mul_timing(self):
        """-> (ms of mul_batch on the device incl. the overlapped scalar copies, calls, scalars)"""
        ms, calls, n = C.c_double(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.ecl_hip_get_mul_timing(self.h, C.byref(ms), C.byref(calls), C.byref(n)))
        return ms.value, calls.value, n.value

    # ---- diagnostics
    def diag_fe(self, op, a, b=None):
        A = limbs_array(a)
        Bv = limbs_array(b) if b is not None else A
        R = np.zeros_like(A)
        self._chk(self.lib.ecl_hip_diag_fe(self.h, op, A.ctypes.data, Bv.ctypes.data, R.ctypes.data, len(A)))
        return ints_of(R)

    def diag_mulg(self, ks):
        K = limbs_array(ks)
        X, Y = np.zeros_like(K), np.zeros_like(K)
        ok = np.zeros(len(K), dtype=np.uint8)
        self._chk(self.lib.ecl_hip_diag_mulg(self.h, K.ctypes.data, X.ctypes.data, Y.ctypes.data, ok.ctypes.data, len(K)))
        return ints_of(X), ints_of(Y), ok

    def diag_hash160(self, xs, ys):
        X, Y = limbs_array(xs), limbs_array(ys)
        h33 = np.zeros((len(X), 5), dtype=np.uint32)
        h65 = np.zeros((len(X), 5), dtype=np.uint32)
        self._chk(self.lib.ecl_hip_diag_hash160(self.h, X.ctypes.data, Y.ctypes.data, h33.ctypes.data, h65.ctypes.data, len(X)))
        return h33, h65

    def diag_bloom(self, hashes):
        H = np.ascontiguousarray(hashes, dtype=np.uint32).reshape(-1, 5)
        hit = np.zeros(len(H), dtype=np.uint8)
        self._chk(self.lib.ecl_hip_diag_bloom(self.h, H.ctypes.data, hit.ctypes.data, len(H)))
        return hit

    def diag_bloom_mod(self, nwords, xs):
        X = np.ascontiguousarray(xs, dtype=np.uint64)
        R = np.zeros_like(X)
        self._chk(self.lib.ecl_hip_diag_bloom_mod(self.h, nwords, X.ctypes.data, R.ctypes.data, len(X)))
        return R
