"""Builds the gfx950 shared library in-tree: ecloop_amd/libecloop_hip.so (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libecloop_hip.so")
ASM = os.path.join(PKG, "libecloop_hip.gfx950.s")  # assembly of the library's code object (kept by the build)
SOURCES = ["ecloop_hip.hip", "add_kernel.h", "hash160.h", "fe256.h", "ec.h", "bloom.h", "scalar_host.h"]


def source_sha256():
    """sha256 over the device / ABI sources: identifies the build a profile under profiles/ was taken on"""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(ROOT, "include", "ecloop_hip.h")]:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(ROOT, "include", "ecloop_hip.h")]
    if not force and not _stale(LIB, deps) and os.path.exists(ASM):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # -save-temps keeps the gfx950 assembly of the very code object that goes into the library: tools/isa_mix.py
    # reads the kernels' static instruction mix from it (bench.py's `roofline.static`, tests/test_profiles_fresh.py)
    tmp = tempfile.mkdtemp(prefix="eclbuild")
    try:
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-save-temps",
               os.path.join(CSRC, "ecloop_hip.hip"), "-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=tmp)
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f]
        if asm:
            shutil.copy(os.path.join(tmp, asm[0]), ASM)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return LIB


def build_host_cli(force=False):
    """The C host program (ecloop-hip): plain C, links the C ABI only."""
    src = os.path.join(PKG, "host", "ecloop_hip_cli.c")
    out = os.path.join(PKG, "host", "ecloop-hip")
    if not os.path.exists(src):
        return None
    if not force and not _stale(out, [src, os.path.join(ROOT, "include", "ecloop_hip.h")]):
        return out
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", out,
                    "-L", PKG, "-lecloop_hip", "-Wl,-rpath,$ORIGIN/..", "-lpthread", "-lm"], check=True)
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
    print(build_host_cli(force=True))
