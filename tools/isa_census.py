#!/usr/bin/env python3
"""Instruction census of a gfx950 kernel from the compiler's assembly (hipcc --save-temps):

    python3 tools/isa_census.py <file.s> <kernel name substring> [--regions]

Splits the kernel at its s_barrier instructions and at loop headers and counts instructions per class (packed / scalar
VALU, f64 VALU, LDS, VMEM load / store, SALU, SMEM, waits, branches) per region -- static counts, weighted by nothing:
the caller multiplies by the trip counts it knows."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_") and ("_f64" in op or op in ("v_rcp_f64_e32", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")):
        return "valu_f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_load", "global_load")):
        return "vmem_ld"
    if op.startswith(("buffer_store", "global_store")):
        return "vmem_st"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_body(text, name):
    m = re.search(r"^(_Z\S*" + re.escape(name) + r"\S*):.*?\n(.*?)\n\s*s_endpgm", text, re.S | re.M)
    if not m:
        raise SystemExit(f"kernel {name} not found")
    return m.group(1), m.group(2).split("\n")


def main():
    path, name = sys.argv[1], sys.argv[2]
    sym, lines = kernel_body(open(path).read(), name)
    regions, cur, label = [], collections.Counter(), "entry"
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
            continue
        if t.startswith(".LBB"):
            if "Loop Header" in ln:
                regions.append((label, cur))
                cur, label = collections.Counter(), "loop " + t.split(":")[0]
            continue
        op = t.split()[0]
        if op.startswith(";"):
            continue
        c = classify(op)
        cur[c] += 1
        if c == "barrier":
            regions.append((label, cur))
            cur, label = collections.Counter(), "after barrier"
    regions.append((label, cur))
    total = collections.Counter()
    for _, c in regions:
        total.update(c)
    cols = ["valu_pk", "valu", "valu_f64", "lds", "vmem_ld", "vmem_st", "salu", "smem", "wait", "branch"]
    print(f"kernel {sym}")
    print("| region | " + " | ".join(cols) + " |")
    print("|---|" + "---|" * len(cols))
    if "--regions" in sys.argv:
        for lab, c in regions:
            if sum(c.values()) >= 8:
                print(f"| {lab} | " + " | ".join(str(c.get(k, 0)) for k in cols) + " |")
    print("| **total (static)** | " + " | ".join(str(total.get(k, 0)) for k in cols) + " |")


if __name__ == "__main__":
    main()
