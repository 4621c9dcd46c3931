#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel trace / counter collection) into the small summaries kept under
profiles/.  usage: summarize_prof.py <dir with *_kernel_trace.csv / *_counter_collection.csv> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for k in ("k_h_pair", "k_h_wave", "k_phase", "k_camodc", "k_measure", "k_norm", "k_fill_random", "k_set_one"):
        if k in name:
            return k
    return name[-60:]


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {}
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if traces:
        dur = defaultdict(list)
        for f in traces:
            for r in csv.DictReader(open(f)):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        tot = sum(sum(v) for v in dur.values())
        out["kernel_stats"] = {k: dict(calls=len(v), total_ms=round(sum(v), 3), avg_ms=round(sum(v) / len(v), 4),
                                       min_ms=round(min(v), 4), max_ms=round(max(v), 4), pct=round(100 * sum(v) / tot, 2))
                               for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))}
    counters = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if counters:
        acc = defaultdict(lambda: defaultdict(list))
        for f in counters:
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out["counters"] = {k: {c: dict(launches=len(v), mean=sum(v) / len(v)) for c, v in cs.items()} for k, cs in acc.items()}
    print(json.dumps(out, indent=1))
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open(os.path.join("profiles", f"{tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
