"""Builds the gfx950 shared library in-tree: ecloop_amd/libecloop_hip.so (hipcc cross-compiles without a GPU).

Staleness is decided by CONTENT, not by file times: the library and the host program each carry a sidecar stamp
(`<target>.stamp`) holding the sha256 of the sources + the compile command they were built from.  A snapshot copied to
another machine (gpurun) gets fresh mtimes in arbitrary order; with the stamp a shipped, current library is never
rebuilt inside a timed process, and a stale one always is."""
import hashlib
import os
import shutil
import subprocess
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libecloop_hip.so")
ASM = os.path.join(PKG, "libecloop_hip.gfx950.s")  # assembly of the library's code object (kept by the build)
SOURCES = ["ecloop_hip.hip", "exports.map", "setup_kernels.h", "mul_kernels.h", "aux_kernels.h", "abi_mul.h", "abi_diag.h", "abi_lookahead.h", "abi_lookahead_ctx.h", "add_kernel.h", "hash160.h", "fe256.h", "ec.h",
           "bloom.h", "scalar_host.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden"]
HOST_SOURCES = ["ecloop_hip_cli.c"]  # the translation unit; its parts (hashed for the stamp like it):
HOST_PARTS = ["cli_base.h", "cli_filter.h", "cli_report.h", "cli_add.h", "cli_mul.h", "cli_rnd_blf.h", "cli_keys.h"]
HOST_FLAGS = ["-O2", "-std=gnu11", "-Wall"]


def _library_sources():
    return [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(ROOT, "include", "ecloop_hip.h")]


def _host_sources():
    return [os.path.join(PKG, "host", s) for s in HOST_SOURCES + HOST_PARTS] + [os.path.join(ROOT, "include", "ecloop_hip.h")]


def _sha256_of(files, extra=()):
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    for e in extra:
        h.update(b"\1" + e.encode())
    return h.hexdigest()


def source_sha256():
    """sha256 over the device / ABI sources: identifies the build a profile under profiles/ was taken on"""
    return _sha256_of(_library_sources())


def _stamp_of(target):
    try:
        return open(target + ".stamp").read().strip()
    except OSError:
        return None


def _current(target, want):
    return os.path.exists(target) and _stamp_of(target) == want


def library_is_current():
    return _current(LIB, _sha256_of(_library_sources(), HIPCC_FLAGS)) and os.path.exists(ASM)


def build_library(force=False, verbose=False):
    want = _sha256_of(_library_sources(), HIPCC_FLAGS)
    if not force and _current(LIB, want) and os.path.exists(ASM):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -save-temps keeps the gfx950 assembly of the very code object that goes into the library: tools/isa_mix.py
    # reads the kernels' static instruction mix from it (bench.py's `roofline.static`, tests/test_profiles_fresh.py)
    tmp = tempfile.mkdtemp(prefix="eclbuild")
    try:
        cmd = [hipcc] + HIPCC_FLAGS + ["-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-save-temps", os.path.join(CSRC, "ecloop_hip.hip"), "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        if os.path.exists(LIB + ".stamp"):
            os.unlink(LIB + ".stamp")
        subprocess.run(cmd, check=True, cwd=tmp)
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f]
        if asm:
            shutil.copy(os.path.join(tmp, asm[0]), ASM)
        with open(LIB + ".stamp", "w") as f:
            f.write(want + "\n")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


def build_host_cli(force=False):
    """The C host program (ecloop-hip): plain C, links the C ABI only."""
    srcs = [os.path.join(PKG, "host", s) for s in HOST_SOURCES]
    out = os.path.join(PKG, "host", "ecloop-hip")
    if not all(os.path.exists(s) for s in srcs):
        return None
    want = _sha256_of(_host_sources(), HOST_FLAGS)
    if not force and _current(out, want):
        return out
    if os.path.exists(out + ".stamp"):
        os.unlink(out + ".stamp")
    subprocess.run(["gcc"] + HOST_FLAGS + ["-I", os.path.join(ROOT, "include")] + srcs +
                   ["-o", out, "-L", PKG, "-lecloop_hip", "-Wl,-rpath,$ORIGIN/..", "-lpthread", "-lm"], check=True)
    with open(out + ".stamp", "w") as f:
        f.write(want + "\n")
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_host_cli(force=True))
