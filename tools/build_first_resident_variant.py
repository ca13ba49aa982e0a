#!/usr/bin/env python3
"""Measurement build for review lever 1(c) of round 5 (the ramp of ecl_hip_mul_batch): a copy of the library in which the FIRST piece of every
call needs no host-to-device copy - its scalars are kept in a device buffer from the first call on (valid because the benchmark sends the
same array every time).  That is the best case of any scheme that lets the first piece start without waiting for PCIe (a kernel reading
pinned host memory in place included): what it gains over the shipped library is the bound on lever (c).  The shipped sources are not
touched: the patch is applied to a temporary copy.   -> build_ab/first_resident.so;  measured by tools/ab_r05e.sh"""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp(prefix="eclvar")
shutil.copytree(os.path.join(ROOT, "ecloop_amd", "csrc"), os.path.join(tmp, "ecloop_amd", "csrc"), ignore=shutil.ignore_patterns("tools"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
p = os.path.join(tmp, "ecloop_amd", "csrc", "abi_mul.h")
s = open(p).read()
old = """    HIPCHK(h, hipMemcpyAsync(h->d_kbuf[b], src, (size_t)m * 32, hipMemcpyHostToDevice, h->copy_stream));
    HIPCHK(h, hipEventRecord(h->ev_copied[b], h->copy_stream));
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_copied[b], 0));
    mul_launch_piece(h, lane, h->d_kbuf[b], m, at, gtab, a);
"""
new = """    static u32* first_resident = nullptr;  // MEASUREMENT BUILD: the first piece's scalars stay on the device from the first call on
    static u32 first_resident_m = 0;
    const bool resident = c == 0 && first_resident && first_resident_m == m;
    if (c == 0 && !resident) {
      if (first_resident) (void)hipFree(first_resident);
      HIPCHK(h, hipMalloc(&first_resident, (size_t)m * 32));
      HIPCHK(h, hipMemcpy(first_resident, src, (size_t)m * 32, hipMemcpyHostToDevice));
      first_resident_m = m;
    }
    if (!resident || c != 0) {
      HIPCHK(h, hipMemcpyAsync(h->d_kbuf[b], src, (size_t)m * 32, hipMemcpyHostToDevice, h->copy_stream));
      HIPCHK(h, hipEventRecord(h->ev_copied[b], h->copy_stream));
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_copied[b], 0));
    }
    mul_launch_piece(h, lane, resident ? first_resident : h->d_kbuf[b], m, at, gtab, a);
"""
assert old in s
open(p, "w").write(s.replace(old, new))
out = os.path.join(ROOT, "build_ab", "first_resident.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", os.path.join(tmp, "ecloop_amd", "csrc", "ecloop_hip.hip"), "-o", out],
               check=True, cwd=tmp)
shutil.rmtree(tmp, ignore_errors=True)
print(out)
