#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel trace / counter collection) into the small summaries kept under
profiles/.
usage: summarize_prof.py <dir with *_kernel_trace.csv / *_counter_collection.csv> <tag> [--per-dispatch k_phase,k_camodc]
kernel_stats pools every launch of a kernel; kernel_stats_by_grid keeps launches of different grid sizes apart (a
register of another size is another grid), so that e.g. the n = 30 launches of k_h_pair -- the ones the bench line's
roofline stands on -- have a row of their own, with the one-line recomputation bytes / avg / 8 TB/s next to it.
--per-dispatch lists, for the named kernels, every dispatch in launch order (duration, counters), so that a script
that launches its cases in a fixed order (tools/experiments/probe_gates.py) can be matched case by case."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

NAMES = ("k_h_pair", "k_h_wave", "k_phase_lines", "k_phase", "k_camodc_table", "k_camodc_oop", "k_camodc", "k_measure", "k_meas_onepass", "k_meas_groups",
         "k_meas_walk", "k_meas_blocksum", "k_meas_composite", "k_meas_chain", "k_meas_prefix", "k_basis_front", "k_norm",
         "k_fill_random", "k_set_one", "k_fused_rounds", "k_fused_x8", "k_fused_q3", "k_gen_cols", "k_expand_compact", "k_fused", "k_swap_bits", "k_pack")


def short(name):
    for k in NAMES:
        if k in name:
            if k.startswith("k_fused") or k == "k_h_pair":               # keep the template arguments: they say which build ran
                i = name.find(k)
                return name[i:].split("(")[0][:60]
            return k
    return name.split("(")[0][-60:]


def main():
    d, tag = sys.argv[1], sys.argv[2]
    per = []
    if "--per-dispatch" in sys.argv:
        per = sys.argv[sys.argv.index("--per-dispatch") + 1].split(",")
    out = {}
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if traces:
        dur = defaultdict(list)
        rows = []
        for f in traces:
            for r in csv.DictReader(open(f)):
                ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                dur[short(r["Kernel_Name"])].append(ms)
                rows.append((int(r["Start_Timestamp"]), short(r["Kernel_Name"]), ms, r.get("VGPR_Count"), r.get("SGPR_Count"),
                             r.get("LDS_Block_Size"), r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X")))
        tot = sum(sum(v) for v in dur.values())
        out["kernel_stats"] = {k: dict(calls=len(v), total_ms=round(sum(v), 3), avg_ms=round(sum(v) / len(v), 4),
                                       min_ms=round(min(v), 4), max_ms=round(max(v), 4), pct=round(100 * sum(v) / tot, 2))
                               for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))}
        # the same, launches of different grid sizes kept apart; for the pair-form Hadamard kernel (one thread per pair) the
        # grid IS the register size: algorithmic bytes = 32 B x 2 x threads, against the 8 TB/s HBM peak
        byg = defaultdict(list)
        for _, k, ms, vg, sg, lds, grid, wg in rows:
            byg[(k, grid, wg)].append(ms)
        out["kernel_stats_by_grid"] = []
        for (k, grid, wg), v in sorted(byg.items(), key=lambda kv: -sum(kv[1])):
            row = dict(kernel=k, grid_threads=grid, workgroup=wg, calls=len(v), avg_ms=round(sum(v) / len(v), 4), min_ms=round(min(v), 4), max_ms=round(max(v), 4))
            if k.startswith("k_h_pair") and grid and "<1," in k:
                b = 64.0 * int(grid)
                row["algorithmic_bytes_per_launch"] = b
                row["hbm_frac_of_8TBs"] = round(b / (sum(v) / len(v) * 1e-3) / 8e12, 4)
            out["kernel_stats_by_grid"].append(row)
        res = {}
        for _, k, ms, vg, sg, lds, grid, wg in rows:
            res.setdefault(k, dict(vgpr=vg, sgpr=sg, lds_bytes=lds, grid=grid, workgroup=wg))
        out["kernel_resources_first_dispatch"] = res
        if per:
            rows.sort()
            out["dispatches"] = {k: [round(ms, 4) for _, kk, ms, *_ in rows if kk == k] for k in per}
    counters = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if counters:
        acc = defaultdict(lambda: defaultdict(list))
        rows = []
        for f in counters:
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                rows.append((int(r.get("Dispatch_Id", 0)), short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"])))
        out["counters"] = {k: {c: dict(launches=len(v), mean=sum(v) / len(v)) for c, v in cs.items()} for k, cs in acc.items()}
        if per:
            rows.sort()
            pd = {}
            for _, k, c, v in rows:
                if k in per:
                    pd.setdefault(k, {}).setdefault(c, []).append(v)
            out["counters_per_dispatch"] = pd
    print(json.dumps(out, indent=1)[:9000])
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open(os.path.join("profiles", f"{tag}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
