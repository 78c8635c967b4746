#!/usr/bin/env python3
"""Instruction mix of the loops of every kernel in a gfx950 assembly file (hipcc --save-temps):
    python tools/isa_mix.py file.s [min_mfma_per_loop]
For each backward-branch loop with at least that many MFMAs: counts per category and the commonest VALU opcodes."""
import re
import sys


def main(path, min_mfma=8):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si, s in enumerate(starts):
        try:
            e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        except StopIteration:
            continue
        body = lines[s:e]
        labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        shown = False
        for a, b in sorted(set(loops), key=lambda ab: ab[1] - ab[0]):
            seg = body[a:b + 1]
            n = sum("v_mfma" in x for x in seg)
            if n < min_mfma:
                continue
            cat = {}
            for x in seg:
                x = x.strip()
                if not x or x[0] in ";.":
                    continue
                op = x.split()[0]
                key = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "ds_read" if op.startswith("ds_read") else
                       "ds_write" if op.startswith("ds_write") else "vmem_ld" if op.startswith(("buffer_load", "global_load")) else
                       "vmem_st" if op.startswith(("buffer_store", "global_store", "global_atomic", "buffer_atomic")) else
                       "waitcnt" if op.startswith("s_waitcnt") else "barrier" if op.startswith("s_barrier") else
                       "branch" if op.startswith(("s_cbranch", "s_branch")) else "scratch" if op.startswith("scratch") else
                       "nop" if op.startswith("s_nop") else "salu" if op.startswith("s_") else "other")
                cat[key] = cat.get(key, 0) + 1
                if key == "valu":
                    cat["v:" + op] = cat.get("v:" + op, 0) + 1
            if not shown:
                print(lines[s][:150])
                shown = True
            print(f"  loop {a}-{b} ({b - a} lines):", {k: v for k, v in sorted(cat.items()) if not k.startswith("v:")})
            print("     ", sorted(((v, k[2:]) for k, v in cat.items() if k.startswith("v:")), reverse=True)[:8])


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
