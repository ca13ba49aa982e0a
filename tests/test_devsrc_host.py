"""The DEVICE source (ecloop_amd/csrc/*.h) compiled for the host with g++ and checked against the oracle on the CPU:
covers the kernels' arithmetic / hashing / probe logic without a GPU (the -m gpu tests cover the generated code)."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import orc
from synth import synth_bloom_words

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def D(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("devhost") / "libdevhost.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so,
                    os.path.join(ROOT, "ecloop_amd", "csrc", "tools", "devsrc_host.cpp")], check=True)
    return C.CDLL(so)


def test_field(D):
    rnd = random.Random(3)
    P = orc.P

    def op(o, a, b=0):
        r = orc.FE()
        D.dh_fe_op(o, r, orc.fe(a), orc.fe(b))
        return orc.val(r)

    edge = [0, 1, 2, P - 1, P - 2, 0x1000003D1, 1 << 255, P - 0x1000003D1, (1 << 32) - 1, (1 << 224) - 1,
            P - (1 << 29), (1 << 29) - 1, (1 << 232) - 1, P - 1 - (1 << 232), (1 << 256) - (1 << 232) - 1 - 0x1000003D1]
    vals = edge + [rnd.randrange(P) for _ in range(200)]
    for a in vals:
        for b in vals[:24]:
            assert op(0, a, b) == a * b % P
            assert op(3, a, b) == (a - b) % P
            assert op(4, a, b) == (a + b) % P
            # unnormalised (lazy) operands at the documented magnitude limits, parity of a magnitude-4 value
            assert op(6, a, b) == (3 * a * 2 * b) % P
            assert op(7, a, b) == ((a + b) ** 2) % P
            assert op(8, a, b) == (a * b) % P  # six negations: magnitude 7, value +b
            assert op(9, a, b) == (-2 * a * 2 * b) % P
            assert D.dh_parity(orc.fe(a), orc.fe(b)) == ((2 * a - b) % P) & 1
        assert op(1, a) == a * a % P
        assert op(5, a) == (-a) % P
        for b in vals[:6]:  # the interleaved pairs (fe_mul2 / fe_sqr2), results aliasing their operands
            assert op(13, a, b) == 2 * a * b % P
            assert op(14, a, b) == (a * a + b * b) % P
    for a in vals[1:40]:
        assert op(2, a) == pow(a, P - 2, P)


def test_inverse_by_division_steps_equals_the_addition_chain(D):
    """fe_inv_divsteps (600 Bernstein-Yang division steps in 20 rounds of 30) against pow(a, p - 2, p) and against the 255 S + 15 M
    chain of lib/ecc.c:463-520, on edge values (0, +-1, powers of two, values around the limb boundaries, inputs whose low words
    are all zeros / all ones so that whole rounds are pure halvings) and random ones; unnormalised input of magnitude 7."""
    rnd = random.Random(5)
    P = orc.P

    def op(o, a, b=0):
        r = orc.FE()
        D.dh_fe_op(o, r, orc.fe(a), orc.fe(b))
        return orc.val(r)

    edge = [0, 1, 2, 3, P - 1, P - 2, (P + 1) // 2, 0x1000003D1, P - 0x1000003D1, 1 << 255, (1 << 255) - 1, 1 << 30, 1 << 60, 1 << 240, (1 << 30) - 1,
            (1 << 240) + 1, ((1 << 256) - 1) % P, (1 << 232) - 1, P - (1 << 29), 0x3FFFFFFF << 30, 1 << 29, 1 << 58, 3 << 90, (1 << 200) - (1 << 100)]
    edge += [pow(2, -k, P) for k in (1, 30, 31, 60, 300, 590, 600)] + [pow(3, k, P) for k in (100, 200, 255)]
    vals = edge + [rnd.randrange(P) for _ in range(3000)]
    for a in vals:
        want = pow(a, P - 2, P)
        assert op(10, a) == want, hex(a)
    for a in vals[:200]:
        assert op(11, a) == pow(a, P - 2, P)
    for a in vals[:300]:
        b = rnd.randrange(P)
        assert op(12, a, b) == pow((4 * b - 2 * a) % P, P - 2, P)


def test_scalar_mul_and_hash160(D):
    rnd = random.Random(4)
    for k in [1, 2, 3, 5, 0xDC2A04, orc.N - 1, (orc.N + 1) // 2, 1 << 255] + [rnd.randrange(1, orc.N) for _ in range(12)]:
        x, y = orc.FE(), orc.FE()
        assert D.dh_mulg(x, y, orc.fe(k)) == 1
        assert (orc.val(x), orc.val(y)) == orc.point_of(k)
        h33, h65 = orc.H160(), orc.H160()
        D.dh_hash160(h33, h65, x, y)
        assert list(h33) == orc.hash160(orc.val(x), orc.val(y), True)
        assert list(h65) == orc.hash160(orc.val(x), orc.val(y), False)
    x, y = orc.FE(), orc.FE()
    assert D.dh_mulg(x, y, orc.fe(0)) == 0 and D.dh_mulg(x, y, orc.fe(orc.N)) == 0


def test_xyzz_lazy_sums_equal_the_scalar_sum(D):
    """`mul`'s window sums (ec.h: xyzz_mmadd_lazy / xyzz_madd_lazy, lazy magnitudes, no exceptional cases) against the
    oracle's k*G for the sum of the scalars; P = Q / P = -Q on the way must leave ZZ = 0 (the kernel's one test per scalar)."""
    rnd = random.Random(11)
    for n in [1, 2, 3, 5, 12, 13, 19]:
        for _ in range(4):
            ks = [rnd.randrange(1, orc.N) for _ in range(n)]
            pts = [orc.point_of(k) for k in ks]
            px = (C.c_uint64 * (4 * n))(*[w for p in pts for w in orc.fe(p[0])])
            py = (C.c_uint64 * (4 * n))(*[w for p in pts for w in orc.fe(p[1])])
            x, y = orc.FE(), orc.FE()
            assert D.dh_xyzz_sum(x, y, px, py, n) == 1
            assert (orc.val(x), orc.val(y)) == orc.point_of(sum(ks) % orc.N)
    for ks in [[5, 5], [5, orc.N - 5], [3, 4, 7, 9], [3, 4, orc.N - 7, 9], [2, 3, 4, 9, 11]]:
        n = len(ks)
        pts = [orc.point_of(k) for k in ks]
        px = (C.c_uint64 * (4 * n))(*[w for p in pts for w in orc.fe(p[0])])
        py = (C.c_uint64 * (4 * n))(*[w for p in pts for w in orc.fe(p[1])])
        x, y = orc.FE(), orc.FE()
        degenerate = any(sum(ks[:i]) % orc.N in (ks[i], orc.N - ks[i]) for i in range(1, n))
        assert D.dh_xyzz_sum(x, y, px, py, n) == (0 if degenerate else 1)


@pytest.mark.parametrize("nw,mode", [(12345, "a|(b&c)"), (1, "ones"), (2, "a|b"), (4099, "a|b"), (1 << 16, "a|b")])
def test_bloom_probe(D, nw, mode):
    w = np.full(1, 0xFFFFFFFFFFFFFFFF, np.uint64) if mode == "ones" else synth_bloom_words(nw, 5, mode)
    pw = w.ctypes.data_as(C.POINTER(C.c_uint64))
    rnd = random.Random(8)
    hits = 0
    for _ in range(5000):
        h = orc.H160(*[rnd.getrandbits(32) for _ in range(5)])
        a, b = D.dh_bloom_has(pw, C.c_uint64(nw), h), orc.lib().orc_blf_has(pw, C.c_uint64(nw), h)
        assert a == b
        hits += a
    if mode == "ones":
        assert hits == 5000


MOD_SIZES = [1, 2, 3, 65539, (1 << 24) - 1, 1 << 24, (1 << 24) + 3, 92_175_407, 737_403_255, (1 << 31) - 1, 1 << 31,
             (1 << 31) + 12345, (1 << 33) + 7, (1 << 40) - 87, (1 << 57) + 1, (1 << 58) - 1]


def mod_inputs(nw, n=4000, seed=17):
    """word indices (idx >> 6 < 2^58) that stress the reciprocal: random, multiples of nw +- 1, the extremes"""
    rnd = random.Random(seed ^ nw)
    top = (1 << 58) - 1
    xs = [0, 1, nw - 1, nw, nw + 1, top, top - 1, top // nw * nw, max(top // nw * nw - 1, 0)]
    xs += [rnd.getrandbits(58) for _ in range(n)]
    xs += [min(rnd.randrange(1, top // nw + 1) * nw + d, top) for _ in range(n // 4) for d in (-1, 0, 1)]
    return [x for x in xs if 0 <= x <= top]


@pytest.mark.parametrize("nw", MOD_SIZES)
def test_bloom_mod_all_width_classes(D, nw):
    """bloom_mod (the `% size` of utils.c:286-288 on the word index) against Python's %, including the 64-bit branch
    that serves filters of 2^31 words (16 GB) and more — no such filter needs to exist for this"""
    D.dh_bloom_mod.restype = C.c_uint64
    D.dh_bloom_mod.argtypes = [C.c_uint64, C.c_uint64]
    for x in mod_inputs(nw):
        assert D.dh_bloom_mod(nw, x) == x % nw, (nw, x)
