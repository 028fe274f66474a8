#!/usr/bin/env python3
"""Per-kernel instruction mix from `hipcc --cuda-device-only -S` output: VALU / LDS / global / scratch counts,
overall and inside the largest loop bodies (backward-branch spans)."""
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:] or ["corr_cells_kernelILi8ELb0E", "track_block_kernelILi8ELb0E", "grid_cells_wave_kernelILi2"]
for m in re.finditer(r"^(_ZN3gyp\w+):[^\n]*\n(.*?)^\.Lfunc_end", txt, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not any(w in name for w in want):
        continue
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]

    def count(ls, pat):
        return sum(1 for l in ls if re.match(pat, l))

    def mix(ls):
        return {"valu": count(ls, r"v_"), "ds": count(ls, r"ds_"), "gload": count(ls, r"global_load"), "gstore": count(ls, r"global_store"),
                "sload": count(ls, r"scratch_load"), "sstore": count(ls, r"scratch_store"), "wait": count(ls, r"s_waitcnt"),
                "barrier": count(ls, r"s_barrier"), "salu": count(ls, r"s_(?!waitcnt|barrier|nop)")}

    print(name[:70], len(lines), mix(lines))
    label_at = {}
    for i, l in enumerate(lines):
        mm = re.match(r"(\.LBB\d+_\d+):", l)
        if mm:
            label_at[mm.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        mm = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
        if mm:
            tgt = mm.group(1) or mm.group(2)
            if tgt in label_at and label_at[tgt] < i:
                loops.append((i - label_at[tgt], label_at[tgt], i))
    for span, a, b in sorted(loops, reverse=True)[:4]:
        print("   loop", lines[a][:14], "span", span, mix(lines[a:b]))
