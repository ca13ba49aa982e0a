#!/usr/bin/env python3
"""Builds oracle/_ref/ecloop_gpu: the REFERENCE's own host program (vladkens/ecloop main.c, unmodified but for six
one-line edits) with its hot path bound to libecloop_hip.so through include/ecloop_hip.h.

This is INTEGRATION.md's "binding a maintainer would add", compiled and run instead of described.  No reference text is
kept in this repository: the script reads /root/reference/main.c where it lies, locates its anchors by function names /
regular expressions, inserts OUR lines (calls into oracle/ref_binding/ecloop_gpu_binding.h, also ours), writes the
patched copy into a temporary directory, compiles it with -I/root/reference (main.c is a unity build: it #includes
lib/*.c) and deletes the copy.  Only the binary goes to oracle/_ref/ (git-ignored; travels to the GPU box like the
other reference binaries).  The reference's Makefile is not run.

The edits (each anchored on a unique pattern; the script fails loudly if an anchor is missing or ambiguous):
  1. ctx_t gains   struct ecl_hip *gpu[16]; int gpu_count; unsigned gpu_next;          before `} ctx_t;`   (main.c:69)
  2. #include "ecloop_gpu_binding.h"                                                 before `void batch_add(` (main.c:349)
  3. cmd_add_worker: `const int gpu_g = gpu_claim(ctx);` after its `ctx_t *ctx = ...` line             (main.c:406)
  4. cmd_add_worker: batch_add(ctx, pk, ctx->job_size)  ->  gpu_batch_add(ctx, gpu_g, pk, ctx->job_size) (main.c:430)
  5. cmd_add, cmd_rnd: `gpu_init(ctx);` after `ctx_precompute_gpoints(ctx);`                        (main.c:438,623)
  6. cmd_mul_worker: gpu_claim as in 3 (main.c:487); the four lines ec_gtable_mul loop .. check_found_mul
     -> gpu_mul_job(ctx, gpu_g, pk, job->count) (main.c:531-534); cmd_mul: ec_gtable_init() -> gpu_init(ctx) (main.c:543)
Optional: --job-log2 N rewrites MAX_JOB_SIZE (main.c:16; 2^21 keys are 0.2 ms of GPU work) - NOT used by the parity
tests, which keep the reference's job arithmetic bit for bit; used for the throughput note in INTEGRATION.md.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def sub_once(text, pattern, repl, what, count=1, flags=re.S):
    found = re.findall(pattern, text, flags)
    if len(found) != count:
        raise SystemExit(f"anchor for '{what}' matched {len(found)} times, expected {count}: the reference differs from v0.5.0")
    return re.sub(pattern, repl, text, flags=flags)


def in_function(text, name, edit):
    """apply `edit` to the body of `name` only (from its definition line to the next line that is just '}')"""
    m = re.search(r"^[\w \*]*\b%s\(.*?\) \{\n.*?^\}\n" % re.escape(name), text, re.S | re.M)
    if not m:
        raise SystemExit(f"function {name} not found in the reference")
    return text[: m.start()] + edit(m.group(0)) + text[m.end():]


def patch(src, job_log2=None):
    claim = r"\1\n  const int gpu_g = gpu_claim(ctx);"
    s = src
    s = sub_once(s, r"\n\} ctx_t;", "\n  struct ecl_hip *gpu[16]; int gpu_count; unsigned gpu_next; /* GPU binding */\n} ctx_t;", "ctx_t members")
    s = sub_once(s, r"\nvoid batch_add\(", '\n#include "ecloop_gpu_binding.h"\n\nvoid batch_add(', "binding include")
    s = in_function(s, "cmd_add_worker", lambda f: sub_once(
        sub_once(f, r"(ctx_t \*ctx = \(ctx_t \*\)arg;)", claim, "add worker: claim"),
        r"batch_add\(ctx, pk, ctx->job_size\);", "gpu_batch_add(ctx, gpu_g, pk, ctx->job_size);", "add worker: call"))
    for fn in ("cmd_add", "cmd_rnd"):
        s = in_function(s, fn, lambda f: sub_once(f, r"(ctx_precompute_gpoints\(ctx\);)", r"\1\n  gpu_init(ctx);", fn + ": init"))
    s = in_function(s, "cmd_mul_worker", lambda f: sub_once(
        sub_once(f, r"(ctx_t \*ctx = \(ctx_t \*\)arg;)", claim, "mul worker: claim"),
        r"for \(size_t i = 0; i < job->count; \+\+i\) ec_gtable_mul\(&cp\[i\], pk\[i\]\);.*?check_found_mul\(ctx, pk, cp, job->count\);",
        "(void)cp;\n    gpu_mul_job(ctx, gpu_g, pk, job->count);", "mul worker: body"))
    s = in_function(s, "cmd_mul", lambda f: sub_once(f, r"ec_gtable_init\(\);", "gpu_init(ctx);", "cmd_mul: init"))
    if job_log2 is not None:
        s = sub_once(s, r"#define MAX_JOB_SIZE [^\n]*", f"#define MAX_JOB_SIZE (1ull << {int(job_log2)})", "MAX_JOB_SIZE")
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "oracle", "_ref", "ecloop_gpu"))
    ap.add_argument("--job-log2", type=int, default=None)
    ap.add_argument("--show-diff", action="store_true", help="print the unified diff of the patch (for INTEGRATION.md) and exit")
    a = ap.parse_args()
    main_c = os.path.join(a.ref, "main.c")
    if not os.path.exists(main_c):
        print(f"reference not present at {a.ref}; keeping a prebuilt {os.path.relpath(a.out, ROOT)} if there is one")
        return 0
    src = open(main_c).read()
    out = patch(src, a.job_log2)
    if a.show_diff:
        import difflib
        sys.stdout.writelines(l for l in difflib.unified_diff(src.splitlines(True), out.splitlines(True), "main.c", "main.c (GPU binding)", n=0)
                              if l.startswith(("+", "-", "@")))
        return 0
    lib_dir = os.path.join(ROOT, "ecloop_amd")
    if not os.path.exists(os.path.join(lib_dir, "libecloop_hip.so")):
        raise SystemExit("build ecloop_amd/libecloop_hip.so first (python -m ecloop_amd.build)")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="eclrefgpu")
    try:
        patched = os.path.join(tmp, "main_gpu.c")
        open(patched, "w").write(out)
        # rpath relative to oracle/_ref/: the binary finds the library in the tree it travels with
        cmd = ["gcc", "-O3", "-ffast-math", "-w", "-march=x86-64-v2", "-msha", "-mno-avx", "-mno-avx2", "-mno-avx512f",
               "-I", a.ref, "-I", HERE, "-I", os.path.join(ROOT, "include"), patched, "-o", a.out,
               "-L", lib_dir, "-lecloop_hip", "-Wl,-rpath,$ORIGIN/../../ecloop_amd", "-lm", "-pthread"]
        subprocess.run(cmd, check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(a.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
