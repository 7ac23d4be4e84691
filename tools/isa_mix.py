#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's assembly (-S --cuda-device-only): counts per opcode, per class and per
basic block.  Usage: isa_mix.py file.s mangled_kernel_name_or_prefix [--blocks] [--json]

--json prints {"valu": n, "half_rate_class_frac": .., "full_rate_class_frac": ..}: the kernel's vector opcodes split into the two
issue classes tools/ubench/valu_rates.hip measures on gfx950 (packed-16 / v_perm / min-max / shift / dot / three-operand forms at
~4.3 clk per wave-instruction and SIMD; add / sub / logic / mov / compare / f32 at ~2.5 clk).  Static: every opcode of the kernel
body counts once, whatever its trip count — tools/collect_profiles.sh stores it beside the dynamic SQ_INSTS_VALU count."""
import collections
import re
import sys


HALF_RATE = ("v_pk_", "v_perm_", "v_min", "v_max", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_bfi", "v_dot", "v_mad", "v_add3", "v_and_or",
             "v_or3", "v_xad", "v_mbcnt", "v_alignb", "v_sad", "v_msad", "v_cvt_pk", "v_mul_lo", "v_mul_hi", "v_med3", "v_fma", "v_add_lshl", "v_xor3")


def issue_class(op):
    return "half" if op.startswith(HALF_RATE) else "full"


def kernel_body(text, name):
    m = re.search(r"\n(" + re.escape(name) + r"[A-Za-z0-9_]*):", text)
    if not m:
        raise SystemExit(f"no kernel label starting with {name}")
    name = m.group(1)
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
    if "--json" in sys.argv:
        import json
        half = sum(v for k, v in ops.items() if classify(k) == "VALU" and issue_class(k) == "half")
        valu = cls["VALU"]
        print(json.dumps({"valu": valu, "salu": cls["SALU"], "lds": cls["LDS"], "half_rate_class_frac": round(half / max(valu, 1), 3),
                          "full_rate_class_frac": round(1 - half / max(valu, 1), 3)}))
        return
    print("class totals:", dict(cls))
    for k, v in ops.most_common(): print(f"  {k:30s} {v}")
    if "--blocks" in sys.argv:
        for b, d in blocks: print(b, d)


if __name__ == "__main__":
    main()
