#!/usr/bin/env python3
"""Run lengths of double-rate VALU opcodes in a kernel's assembly: how many add / sub / logic / v_bitop3 / right-shift
instructions stand alone between 4-clock opcodes and how many sit in runs (profiles/ubench_r02.txt: a double-rate
opcode only pays off in a run).  usage: tools/isa_runs.py file.s [substring of the mangled kernel name ...]"""
import re
import sys
from collections import Counter

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32", "v_bitop3_b32",
        "v_lshrrev_b32", "v_ashrrev_i32"}


def kernels(path):
    s = open(path).read()
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end", s, re.M | re.S):
        yield m.group(1), m.group(2)


def analyse(body):
    runs, cur, valu, fast = Counter(), 0, 0, 0
    for line in body.split("\n"):
        m = re.match(r"\s+([a-z][a-z0-9_]+)", line)
        if not m:
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
        if not op.startswith("v_"):
            if not op.startswith("s_nop"):
                pass  # scalar instructions issue beside the VALU: they do not break a run
            continue
        valu += 1
        if op in FAST:
            cur += 1
            fast += 1
        else:
            if cur:
                runs[cur] += 1
            cur = 0
    if cur:
        runs[cur] += 1
    return valu, fast, runs


if __name__ == "__main__":
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if pats and not any(p in name for p in pats):
            continue
        valu, fast, runs = analyse(body)
        meta = re.search(r"\.vgpr_count:\s+(\d+)", open(path).read()[open(path).read().index(".name:           " + name) - 3000:][:6000]) if False else None
        tot = sum(k * v for k, v in runs.items())
        b = lambda lo, hi: sum(k * v for k, v in runs.items() if lo <= k <= hi)
        print(f"{name}: VALU {valu}, double-rate {fast} ({100.0 * fast / max(valu, 1):.1f} %); of those in runs of 1: {100.0 * b(1, 1) / max(tot, 1):.0f} %, "
              f"2-3: {100.0 * b(2, 3) / max(tot, 1):.0f} %, 4-7: {100.0 * b(4, 7) / max(tot, 1):.0f} %, 8-15: {100.0 * b(8, 15) / max(tot, 1):.0f} %, "
              f"16+: {100.0 * b(16, 10**9) / max(tot, 1):.0f} %")
