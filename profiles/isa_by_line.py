#!/usr/bin/env python3
"""Static instruction count of one kernel per source line (from `hipcc -gline-tables-only --save-temps` assembly):
   python profiles/isa_by_line.py <file.s> <mangled kernel symbol> <source.hip> [top]
A wave executes every path any of its lanes needs, so for the divergent search kernel the static size of the paths a
trip visits is what a trip costs."""
import collections
import re
import sys

asm, sym, srcf = sys.argv[1], sys.argv[2], sys.argv[3]
top_n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
cur, cnt, kinds = None, collections.Counter(), collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)", l)
    if m:
        cur = int(m.group(1))
        continue
    t = l.strip()
    if not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    if re.match(r"(v_|s_|ds_|global_|buffer_|scratch_)", op):
        cnt[cur] += 1
        kinds[op.split("_")[0]] += 1
src = open(srcf).read().split("\n")
print("static instructions:", sum(cnt.values()), dict(kinds))
for ln, c in sorted(cnt.items(), key=lambda x: -x[1])[:top_n]:
    print("%5d  line %4s  %s" % (c, ln, src[ln - 1].strip()[:120] if ln and ln <= len(src) else ""))
