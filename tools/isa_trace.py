#!/usr/bin/env python3
"""Compressed instruction-category trace of a kernel's largest inner loop (from `hipcc --cuda-device-only -S`):
V = VALU run, D = LDS op, G = global load, Sl/Ss = scratch, W[..] = s_waitcnt, BAR = barrier."""
import re
import sys

txt = open(sys.argv[1]).read()
name = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
m = re.search(r"^(_ZN3gyp\w*" + name + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", txt, flags=re.S | re.M)
lines = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
label_at = {re.match(r"(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(lines) if re.match(r"(\.LBB\d+_\d+):", l)}
loops = []
for i, l in enumerate(lines):
    mm = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in label_at and label_at[t] < i:
            loops.append((i - label_at[t], label_at[t], i))
loops.sort(reverse=True)
print(m.group(1), "loops (span):", [s for s, _, _ in loops[:8]])
span, a, b = loops[which] if which >= 0 else min((l for l in loops if l[0] > 1500), default=loops[0])
print("tracing loop span", span)


def cat(l):
    if l.startswith("v_"): return "V"
    if l.startswith("ds_"): return "D"
    if l.startswith("global_load"): return "G"
    if l.startswith("global_store"): return "Gs"
    if l.startswith("scratch_load"): return "Sl"
    if l.startswith("scratch_store"): return "Ss"
    if l.startswith("s_waitcnt"): return "W[" + l.split(None, 1)[1].split(";")[0].strip() + "]"
    if l.startswith("s_barrier"): return "BAR"
    if l.startswith("s_"): return "s"
    if l.startswith(".LBB"): return "L"
    return "?"


out, prev, cnt = [], None, 0
for l in lines[a:b + 1]:
    c = cat(l)
    if c == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{prev}{cnt if cnt > 1 else ''}")
        prev, cnt = c, 1
out.append(f"{prev}{cnt}")
print(" ".join(out))
