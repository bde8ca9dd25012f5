#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/ablate.py: per run of identical consecutive launches of the pricing
kernel, the kernel's average duration, the average gap to the next kernel's start and what sits in the gap.
usage: exp_agg_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0][-60:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps: a pricing kernel followed by anything up to the next pricing kernel
steps = []
i = 0
while i < len(ev):
    if "price_" in ev[i][0]:
        j = i + 1
        while j < len(ev) and "price_" not in ev[j][0]:
            j += 1
        if j < len(ev):
            others = ev[i + 1:j]
            steps.append((ev[i][0], ev[i][2] - ev[i][1], ev[j][1] - ev[i][1], [(o[0], o[2] - o[1]) for o in others]))
        i = j
    else:
        i += 1
# group consecutive steps with the same kernel name and the same followers
groups = []
for s in steps:
    key = (s[0], tuple(o[0] for o in s[3]))
    if groups and groups[-1][0] == key:
        groups[-1][1].append(s)
    else:
        groups.append((key, [s]))
for key, g in groups:
    if len(g) < 3:
        continue
    g = g[2:]          # the first launches of a run: warm-up
    dur = sum(s[1] for s in g) / len(g)
    period = sum(s[2] for s in g) / len(g)
    oth = sum(sum(o[1] for o in s[3]) for s in g) / len(g)
    print(f"{key[0][-40:]:40s} followers={len(key[1])} n={len(g):3d} kernel {dur/1e3:9.1f} us  start-to-start {period/1e3:9.1f} us  "
          f"other kernels {oth/1e3:6.1f} us  idle {(period - dur - oth)/1e3:7.1f} us")
