#!/usr/bin/env python3
"""VALU / SALU / LDS instruction counts per source line inside a kernel (from `hipcc -S -gline-tables-only`), bucketed by
the enclosing function found in the source files.  usage: isa_lines.py file.s kernel_substring [min_count]"""
import collections
import re
import sys
from pathlib import Path

txt = open(sys.argv[1]).read()
name = sys.argv[2]
src_dir = Path(__file__).resolve().parents[1] / "gypsum_amd" / "csrc"
files = {int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+) "[^"]*" "([^"]+)"', txt)}
m = re.search(r"^(_ZN3gyp\w*" + name + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", txt, flags=re.S | re.M)
cur = (0, 0)
cnt = collections.Counter()
kinds = collections.Counter()
for l in m.group(2).split("\n"):
    l = l.strip()
    mm = re.match(r"\.loc\s+(\d+) (\d+)", l)
    if mm:
        cur = (int(mm.group(1)), int(mm.group(2)))
        continue
    if l.startswith("v_"):
        cnt[cur] += 1
    elif l.startswith("s_") and not l.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        kinds[(cur, "s")] += 1


def enclosing(fname, line):
    p = src_dir / fname
    if not p.exists():
        return fname
    ls = p.read_text().split("\n")
    for i in range(min(line, len(ls)) - 1, -1, -1):
        mm = re.match(r"^(?:template.*>\s*)?(?:__device__|__global__|static|inline|__forceinline__|\s)*[\w:<>,\s\*&]*?\b(\w+)\s*\([^;]*$", ls[i])
        if mm and not ls[i].startswith((" ", "\t", "//", "#")) and "(" in ls[i]:
            return mm.group(1)
    return fname


by_fn = collections.Counter()
for (f, line), c in cnt.items():
    by_fn[(files.get(f, "?"), enclosing(files.get(f, "?"), line))] += c
tot = sum(cnt.values())
print(m.group(1)[:60], "VALU total", tot)
for (f, fn), c in by_fn.most_common(40):
    print(f"{c:6d} {100*c/tot:5.1f}%  {f}:{fn}")
