#!/usr/bin/env python3
"""Build container: static instruction counts of one kernel per source function (from `hipcc -gline-tables-only
--save-temps` assembly: .loc directives attribute every instruction to the innermost source line).
Usage: tools/asm_lines.py <file.s> <mangled kernel substring> <source.hip>"""
import collections, re, sys
asm, kname, src = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(src).read().splitlines()
# function start lines in the source (GDEV / __global__ definitions)
funcs = []
for i, l in enumerate(lines, 1):
    m = re.match(r'^(?:GDEV|__global__|template.*\n)?.*\b(g_\w+|gstep_kernel|grollout_kernel|gmax64|gsum|gmin|gmaxu|gballot|sortable\w*|lds_sync|gread\w*|bc\w*|pick_slot|lds_chain_sum_ring8|splitmix64)\s*\(.*\)\s*(?:const)?\s*\{', l)
    if m and (l.startswith('GDEV') or l.startswith('__global__') or l.startswith('__device__')):
        funcs.append((i, m.group(1)))
def func_of(line):
    name = '?'
    for s, n in funcs:
        if s <= line: name = n
    return name
s = open(asm).read()
a = s.index(kname + ':') if (kname + ':') in s else s.index(kname)
b = s.index('s_endpgm', a)
base = src.split('/')[-1]
by = collections.Counter(); byline = collections.Counter()
cur = None
for l in s[a:b].splitlines():
    if re.match(r'\s*\.loc\s', l):
        locs = re.findall(r'([\w./+-]+):(\d+):\d+', l.split(';', 1)[1]) if ';' in l else []
        cur = None
        for f, ln in locs:             # innermost first; take the innermost location inside the source file
            if f.endswith(base): cur = int(ln); break
        if cur is None and locs: cur = (locs[0][0].split('/')[-1], int(locs[0][1]))
        continue
    if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';')):
        if isinstance(cur, int):
            by[func_of(cur)] += 1; byline[cur] += 1
        else:
            by[str(cur)] += 1
tot = sum(by.values())
print("total", tot)
for k, v in by.most_common(): print("%5d  %s" % (v, k))
print("-- hottest lines")
for ln, v in byline.most_common(int(sys.argv[4]) if len(sys.argv) > 4 else 30): print("%5d  %d  %s" % (v, ln, lines[ln - 1].strip()[:120]))
