#!/usr/bin/env python3
"""Exact-line overlap of a source file with the reference's C files (a development check, this container only:
/root/reference does not travel).  Lines are compared after stripping whitespace; blank lines, lone braces and
preprocessor includes are ignored.   usage: tools/overlap_with_reference.py [file ...]"""
import glob
import os
import re
import sys

REF = "/root/reference"
TRIVIAL = re.compile(r"^(|[{}();]+|else|else \{|\} else \{|break;|continue;|return;|return 0;|return NULL;|return false;|return true;|#include .*|#endif|#else|exit\(1\);|/\*|\*/|//.*)$")


def lines(path):
    out = []
    for i, l in enumerate(open(path, errors="replace"), 1):
        t = " ".join(l.strip().split())
        if not TRIVIAL.match(t):
            out.append((i, t))
    return out


def main():
    ref = set()
    for f in glob.glob(os.path.join(REF, "*.c")) + glob.glob(os.path.join(REF, "lib", "*.c")):
        ref.update(t for _, t in lines(f))
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ecloop_amd", "host")
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(host, "*.c")) + glob.glob(os.path.join(host, "*.h")))
    verbose = "-v" in files
    for f in [x for x in files if x != "-v"]:
        ls = lines(f)
        hit = [(i, t) for i, t in ls if t in ref]
        print(f"{f}: {len(hit)} of {len(ls)} non-trivial lines also occur in the reference ({100.0 * len(hit) / max(len(ls), 1):.1f} %)")
        if verbose:
            for i, t in hit:
                print(f"  {i}: {t}")


if __name__ == "__main__":
    main()
