#!/usr/bin/env python3
"""Static instruction mix of the add kernel from the gfx950 assembly the build keeps next to the library
(ecloop_amd/libecloop_hip.gfx950.s, written by ecloop_amd/build.py with -save-temps: the assembly OF the shipped
code object, not a second compilation).

The compiler annotates every basic block with the loop it belongs to, so the blocks can be attributed to the loop
nest of k_add (add_kernel.h): launch loop > {prefix-product loop, table loop > `which` loop > probe loop}.
VALU instructions are split into the issue classes the microbenchmark (ecloop_amd/csrc/tools/ubench.hip ->
profiles/ubench_r02.txt) prices differently:
  mad64 : v_mad_u64_u32
  fast  : add / sub / and / or / xor / mov / not and v_bitop3_b32 - the opcodes that reach ~2.3-2.5 SIMD-cycles per
          wave64 instruction in long runs (and ~4.2 when single between other instructions)
  other : every other VALU instruction (rotates, shifts, v_add3, v_perm, multiplies, carries, selects, ...): >= 4.1
What the numbers are used for:
  * `fingerprint`: bench.py puts it into its JSON line and compares it with the one stored in
    profiles/r02_roofline.json (the PMC counters in that file belong to a particular build); the CPU test
    tests/test_profiles_fresh.py fails when a freshly built library drifts more than 1 % from it;
  * `per_key_static`: one trip of the `which` loop (one key; this includes its rarely executed candidate-ring
    blocks, so it is an UPPER estimate) + half a trip of the table loop around it and of the prefix-product loop
    (each serves two keys).  The PMC count SQ_INSTS_VALU is the truth for the total; the class SHARES come from here.

usage: tools/isa_mix.py [file.s] [mangled kernel name] [--json]     |     tools/isa_mix.py --all   (every instantiation: profiles/rNN_static_mix.json)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "ecloop_amd", "libecloop_hip.gfx950.s")
K_ADD33 = "_Z5k_addILb1ELb0ELb0EEv8add_args"
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_not_b32", "v_bitop3_b32"}
KEYS = ("valu", "mad64", "fast", "other", "salu", "smem", "vmem", "lds", "scratch", "call", "scratch_at_calls")


def classify(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base == "v_mad_u64_u32":
        return "mad64"
    return "fast" if base in FAST else "other"


def zero():
    return {k: 0 for k in KEYS}


def add_to(c, op):
    if op.startswith("v_"):
        c["valu"] += 1
        c[classify(op)] += 1
    elif op.startswith("scratch_"):
        c["scratch"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")):
        c["vmem"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("s_load", "s_buffer_load")):
        c["smem"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
        if op.startswith("s_swappc"):
            c["call"] += 1


def blocks(path, kernel):
    """-> list of {label, header, depth, counts}: the basic blocks of `kernel` with the innermost loop they are in"""
    s = open(path).read()
    a = s.index("\n" + kernel + ":")
    b = s.index(".Lfunc_end", a)
    out = []
    cur = {"label": "entry", "header": None, "depth": 0, "c": zero()}
    lines = s[a:b].split("\n")[2:]
    i = 0
    while i < len(lines):
        line = lines[i]
        m = re.match(r"(\.LBB\d+_\d+):|; %bb\.(\d+):", line)
        if m:
            out.append(cur)
            label = m.group(1) or ("bb." + m.group(2))
            cur = {"label": label, "header": None, "depth": 0, "c": zero()}
            # the loop comments of this block: on this line and the comment-only lines that follow
            j, text = i, line
            while j + 1 < len(lines) and re.match(r"\s+;", lines[j + 1]):
                j += 1
                text += "\n" + lines[j]
            mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", text)
            if mm:
                cur["header"], cur["depth"] = label.lstrip(".L"), int(mm.group(1))
            else:
                mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", text)
                if mm:
                    cur["header"], cur["depth"] = mm.group(1), int(mm.group(2))
            i = j + 1
            continue
        mm = re.match(r"\s+([a-z][a-z0-9_]+)", line)
        if mm:
            add_to(cur["c"], mm.group(1))
        i += 1
    out.append(cur)
    for blk in out:  # a block that calls an out-of-line function passes its arguments through scratch: not spill traffic of the loop it sits in
        if blk["c"]["call"]:
            blk["c"]["scratch_at_calls"], blk["c"]["scratch"] = blk["c"]["scratch"], 0
    return out, s[b : b + 4000]


def loop_tree(path, kernel):
    """parent header of every loop header (from the `Parent Loop` comments)"""
    s = open(path).read()
    a = s.index("\n" + kernel + ":")
    b = s.index(".Lfunc_end", a)
    parent, depth = {}, {}
    for m in re.finditer(r"^\.L(BB\d+_\d+):((?:[^\n]*\n\s+;)*[^\n]*)", s[a:b], re.M):
        text = m.group(0)
        d = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", text)
        if not d:
            continue
        h = m.group(1)
        depth[h] = int(d.group(1))
        ps = re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", text)
        parent[h] = max(ps, key=lambda p: int(p[1]))[0] if ps else None
    return parent, depth


def analyse(path=ASM, kernel=K_ADD33):
    bl, tail = blocks(path, kernel)
    parent, depth = loop_tree(path, kernel)
    total = zero()
    excl = {h: zero() for h in parent}
    for b in bl:
        for k in KEYS:
            total[k] += b["c"][k]
            if b["header"] in excl:
                excl[b["header"]][k] += b["c"][k]
    res = {"kernel": kernel, "total": total,
           "loops": [{"header": h, "depth": depth[h], "parent": parent[h], **excl[h]} for h in sorted(parent, key=lambda h: (depth[h], h))]}
    m = re.search(r"\.private_seg_size, (\d+)", tail)
    res["scratch_bytes_own"] = int(m.group(1)) if m else None
    m = re.search(r"\.num_vgpr, (?:max\()?(\d+)", tail)
    res["vgpr"] = int(m.group(1)) if m else None
    # the loop nest of k_add: launch loop (depth 1, with children) > table loop (depth 2, with children) > `which` loop
    kids = {h: [c for c in parent if parent[c] == h] for h in parent}
    launch = [h for h in parent if depth[h] == 1 and kids[h]]
    if launch:
        L = max(launch, key=lambda h: sum(excl[c]["valu"] for c in kids[h]))
        table = [c for c in kids[L] if kids[c]]
        prefix = [c for c in kids[L] if not kids[c]]
        if table:
            T = table[0]
            W = max(kids[T], key=lambda h: excl[h]["valu"])
            est = {k: float(excl[W][k]) + excl[T][k] / 2.0 for k in ("valu", "mad64", "fast", "other")}
            if prefix:
                P = max(prefix, key=lambda h: excl[h]["valu"])
                for k in est:
                    est[k] += excl[P][k] / 2.0
                res["prefix_loop"] = excl[P]
            res["which_loop"], res["table_loop"], res["launch_loop"] = excl[W], excl[T], excl[L]
            res["per_key_static"] = est
            res["fingerprint"] = {"kernel_valu": total["valu"], "which_loop_valu": excl[W]["valu"], "which_loop_mad64": excl[W]["mad64"],
                                  "which_loop_fast": excl[W]["fast"], "table_loop_valu": excl[T]["valu"],
                                  "prefix_loop_valu": res.get("prefix_loop", zero())["valu"], "scratch_instr": total["scratch"] + total["scratch_at_calls"]}
            try:
                sp = spills(path).get(kernel)
                if sp:
                    res["fingerprint"]["vgpr_spill_count"] = sp["vgpr_spill_count"]
                    res["fingerprint"]["scratch_bytes"] = sp["scratch_bytes"]
            except Exception:
                pass
    return res


K_MUL_CU = "_Z11k_mul_checkILb1ELb1EEvPKjjj4wtab8add_argsPjjj"


def analyse_mul(path=ASM, kernel=K_MUL_CU, nwin=10):
    """k_mul_check (mul_kernels.h): loop nest = {sum loop (one trip per scalar) > window loop (nwin - 2 trips per scalar: windows 0 and 1
    are added before it), walk-back loop (one trip per scalar: two field multiplications pairs + the hash160s + stage-1 probes; its
    children are the rarely taken ring-drain loops), three ring-flush loops after it}.  Per-scalar static estimate of the class mix =
    sum loop + (nwin - 2) x window loop + walk-back loop, each exclusive of its children (the inversion - one per thread, shared by the
    thread's scalars - and the out-of-line complete sum are left out: the PMC count is the truth for the total, the SHARES come from here)."""
    bl, _ = blocks(path, kernel)
    parent, depth = loop_tree(path, kernel)
    excl = {h: zero() for h in parent}
    total = zero()
    for b in bl:
        for k in KEYS:
            total[k] += b["c"][k]
            if b["header"] in excl:
                excl[b["header"]][k] += b["c"][k]
    kids = {h: [c for c in parent if parent[c] == h] for h in parent}
    top = [h for h in parent if depth[h] == 1]
    # the window loop: the depth-2 loop with the most multiply-adds; the sum loop is its parent; the walk-back loop: the hash-heavy one
    win = max((h for h in parent if depth[h] == 2), key=lambda h: excl[h]["mad64"])
    summ = parent[win]
    back = max((h for h in top if h != summ), key=lambda h: excl[h]["other"])
    est = {k: float(excl[summ][k]) + (nwin - 2) * excl[win][k] + excl[back][k] for k in ("valu", "mad64", "fast", "other")}
    return {"kernel": kernel, "total": total, "sum_loop": excl[summ], "window_loop": excl[win], "walk_back_loop": excl[back], "windows": nwin,
            "per_scalar_static": est,
            "fingerprint": {"kernel_valu": total["valu"], "window_loop_valu": excl[win]["valu"], "window_loop_mad64": excl[win]["mad64"],
                            "walk_back_loop_valu": excl[back]["valu"], "scratch_instr": total["scratch"] + total["scratch_at_calls"],
                            "window_loop_scratch": excl[win]["scratch"]}}


def spills(path=ASM, prefix="_Z5k_add"):
    """vgpr_spill_count / private_segment_fixed_size / vgpr_count of every instantiation of the add kernel, from the code
    object's metadata (DESIGN.md states these numbers; tests/test_profiles_fresh.py compares)"""
    s = open(path).read()
    out = {}
    for m in re.finditer(r"- \.agpr_count:.*?\n(?=  - \.agpr_count:|amdhsa\.target|\.\.\.)", s, re.S):
        blk = m.group(0)
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or not name.group(1).startswith(prefix):
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        out[name.group(1)] = {"vgpr_count": g("vgpr_count"), "vgpr_spill_count": g("vgpr_spill_count"),
                              "scratch_bytes": g("private_segment_fixed_size")}
    return out


ADD_KERNELS = {"-a c": "_Z5k_addILb1ELb0ELb0EEv8add_args", "-a u": "_Z5k_addILb0ELb1ELb0EEv8add_args", "-a cu": "_Z5k_addILb1ELb1ELb0EEv8add_args",
               "-a c -endo": "_Z5k_addILb1ELb0ELb1EEv8add_args", "-a u -endo": "_Z5k_addILb0ELb1ELb1EEv8add_args", "-a cu -endo": "_Z5k_addILb1ELb1ELb1EEv8add_args"}
MUL_KERNELS = {"mul -a c": "_Z11k_mul_checkILb1ELb0EEvPKjjj4wtab8add_argsPjjj", "mul -a u": "_Z11k_mul_checkILb0ELb1EEvPKjjj4wtab8add_argsPjjj",
               "mul -a cu": K_MUL_CU}


def analyse_all(path=ASM):
    """every shipped instantiation of the two search kernels: fingerprint, registers / spills, and the scratch instructions inside the
    per-key loops (k_add: prefix-product, table and `which` loops; k_mul_check: window loop) - tests/test_profiles_fresh.py wants 0 there"""
    sp_add, sp_mul = spills(path), spills(path, "_Z11k_mul_check")
    out = {}
    for label, k in ADD_KERNELS.items():
        a = analyse(path, k)
        out[label] = {"kernel": k, "fingerprint": a["fingerprint"], "registers": sp_add.get(k),
                      "scratch_in_loops": {"which": a["which_loop"]["scratch"], "table": a["table_loop"]["scratch"], "prefix": a["prefix_loop"]["scratch"],
                                           "launch": a["launch_loop"]["scratch"]},
                      "per_key_static": {x: round(v, 1) for x, v in a["per_key_static"].items()}}
    for label, k in MUL_KERNELS.items():
        m = analyse_mul(path, k)
        out[label] = {"kernel": k, "fingerprint": m["fingerprint"], "registers": sp_mul.get(k),
                      "scratch_in_loops": {"window": m["window_loop"]["scratch"], "sum": m["sum_loop"]["scratch"], "walk_back": m["walk_back_loop"]["scratch"],
                                           "sum_at_the_call_of_the_complete_sum": m["sum_loop"]["scratch_at_calls"]},
                      "per_scalar_static": {x: round(v, 1) for x, v in m["per_scalar_static"].items()}}
    return out


def main():
    if "--all" in sys.argv:
        rest = [a for a in sys.argv[1:] if not a.startswith("--")]
        print(json.dumps(analyse_all(rest[0] if rest else ASM), indent=1))
        return
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else ASM
    kernel = args[1] if len(args) > 1 else K_ADD33
    if "k_mul_check" in kernel:
        a = analyse_mul(path, kernel)
        print(json.dumps(a, indent=1) if "--json" in sys.argv else "%s\nper scalar (static, %d windows): %s\nfingerprint: %s" % (
            kernel, a["windows"], {k: round(v, 1) for k, v in a["per_scalar_static"].items()}, a["fingerprint"]))
        return
    a = analyse(path, kernel)
    if "--json" in sys.argv:
        print(json.dumps(a, indent=1))
        return
    t = a["total"]
    print("%s: %d VALU (%d mad64, %d fast, %d other), %d SALU, %d VMEM, %d SMEM, %d LDS, %d scratch instr; %s VGPRs, %s B scratch (own)" %
          (kernel, t["valu"], t["mad64"], t["fast"], t["other"], t["salu"], t["vmem"], t["smem"], t["lds"], t["scratch"],
           a["vgpr"], a["scratch_bytes_own"]))
    for l in a["loops"]:
        print("  %sloop %-10s (in %s): VALU %5d  mad64 %4d  fast %4d  other %5d  vmem %d lds %d scratch %d" %
              ("  " * (l["depth"] - 1), l["header"], l["parent"], l["valu"], l["mad64"], l["fast"], l["other"], l["vmem"], l["lds"], l["scratch"]))
    if "per_key_static" in a:
        print("per key (static, upper estimate): %s" % {k: round(v, 1) for k, v in a["per_key_static"].items()})
        print("fingerprint: %s" % a["fingerprint"])
    for k, v in sorted(spills(path).items()):
        print("  %s: %s" % (k, v))


if __name__ == "__main__":
    main()
