#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's --save-temps assembly: counts per opcode, per class and per
basic block.  Usage: isa_mix.py file.s mangled_kernel_name [--blocks]"""
import collections
import re
import sys


def kernel_body(text, name):
    a = text.index("\n" + name + ":")
    b = text.index(".Lfunc_end", a)
    return text[a:b]


def classify(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_"): return "VALU"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    return "other"


def main():
    text = open(sys.argv[1]).read()
    body = kernel_body(text, sys.argv[2])
    ops = collections.Counter(); cls = collections.Counter(); blocks = []; cur = "entry"; n = collections.Counter()
    for line in body.split("\n")[1:]:
        l = line.strip()
        if not l or l.startswith((";", "//")): continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((cur, dict(n))); cur = l.split(":")[0]; n = collections.Counter(); continue
        if l.startswith(".") or l.endswith(":"): continue
        op = l.split()[0]
        ops[op] += 1; cls[classify(op)] += 1; n[classify(op)] += 1
    blocks.append((cur, dict(n)))
    print("class totals:", dict(cls))
    for k, v in ops.most_common(): print(f"  {k:30s} {v}")
    if "--blocks" in sys.argv:
        for b, d in blocks: print(b, d)


if __name__ == "__main__":
    main()
