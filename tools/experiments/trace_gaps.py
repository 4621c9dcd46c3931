#!/usr/bin/env python3
"""from a rocprofv3 kernel trace of run_attempts_small.py: the kernels of the LAST attempt at each size in launch order, duration of
each (us) and the idle gap in front of it (us), then the means over the last 40 attempts"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"][r["Kernel_Name"].find("k_"):].split("(")[0] if "k_" in r["Kernel_Name"] else r["Kernel_Name"][:40],
       int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])) for r in rows]
# an attempt starts at its k_gen_cols / k_basis_front launch
starts = [i for i, e in enumerate(ev) if e[0].startswith("k_gen_cols") or e[0].startswith("k_basis_front")]
groups = defaultdict(list)
for a, b in zip(starts, starts[1:] + [len(ev)]):
    groups[ev[a][3]].append(ev[a:b])
for grid, atts in groups.items():
    from collections import Counter
    common = Counter(len(a) for a in atts).most_common(1)[0][0]
    atts = [a for a in atts if len(a) == common][-40:]
    if len(atts) < 2:
        continue
    print(f"== attempts whose first kernel has grid {grid}: {len(atts)} used, {len(atts[-1])} kernels each")
    tot_k = tot_g = 0.0
    for j in range(len(atts[-1])):
        dur = sum(a[j][2] - a[j][1] for a in atts) / len(atts) / 1e3
        gap = sum((a[j][1] - a[j - 1][2]) for a in atts) / len(atts) / 1e3 if j else 0.0
        tot_k += dur; tot_g += gap
        print(f"   {atts[-1][j][0]:44s} grid {atts[-1][j][3]:8d}  gap {gap:6.2f} us  runs {dur:6.2f} us")
    span = sum(a[-1][2] - a[0][1] for a in atts) / len(atts) / 1e3
    period = (atts[-1][0][1] - atts[0][0][1]) / (len(atts) - 1) / 1e3
    print(f"   kernels {tot_k:.1f} us + gaps {tot_g:.1f} us = {span:.1f} us from first start to last end; one attempt every {period:.1f} us")
