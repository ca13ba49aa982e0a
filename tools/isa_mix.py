#!/usr/bin/env python3
"""Instruction mix of k_add<addr33> from the gfx950 assembly (hipcc -save-temps): VALU instructions per basic block,
split into the issue classes the microbenchmark (profiles/ubench_r01.txt) prices differently:
  mad64 : v_mad_u64_u32                                   4.61 SIMD-cycles per wave-instruction (nominal clock)
  fast  : VOP2-encoded add / sub / and / or / xor / mov   2.55
  other : every other VALU instruction                    4.23
usage: tools/isa_mix.py path/to/ecloop_hip-hip-amdgcn-amd-amdhsa-gfx950.s [mangled kernel name]"""
import re
import sys

COST = {"mad64": 4.61, "fast": 2.55, "other": 4.23}
FAST = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32",
        "v_mov_b32_e32", "v_add_u32_e64"}


def classify(op):
    if op == "v_mad_u64_u32":
        return "mad64"
    return "fast" if op in FAST else "other"


def blocks(path, kernel):
    s = open(path).read()
    a = s.index(kernel + ":")
    b = s.index(".Lfunc_end", a)
    out, cur, name = [], {"mad64": 0, "fast": 0, "other": 0}, "entry"
    for line in s[a:b].split("\n")[1:]:
        line = line.strip()
        m = re.match(r"(\.LBB\d+_\d+):", line)
        br = re.match(r"s_cbranch_\w+|s_branch|s_endpgm", line)
        if m or br:
            if sum(cur.values()):
                out.append((name, cur))
            cur = {"mad64": 0, "fast": 0, "other": 0}
            if m:
                name = m.group(1)
            continue
        m = re.match(r"(v_[a-z0-9_]+)", line)
        if m:
            cur[classify(m.group(1))] += 1
    if sum(cur.values()):
        out.append((name, cur))
    return out


if __name__ == "__main__":
    kernel = sys.argv[2] if len(sys.argv) > 2 else "_Z5k_addILb1ELb0ELb0EEv8add_args"
    for name, c in blocks(sys.argv[1], kernel):
        n = sum(c.values())
        if n >= 8:
            cyc = sum(c[k] * COST[k] for k in c)
            print("%-12s VALU %5d  mad64 %4d  fast %4d  other %5d  -> %7.0f cycles (%.2f / instr)" % (name, n, c["mad64"], c["fast"], c["other"], cyc, cyc / n))
