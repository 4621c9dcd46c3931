#!/usr/bin/env python3
"""print the fused-pass kernels of a rocprofv3 kernel trace in launch order with their durations (ms)"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"]
    if any(k in n for k in ("k_fused", "k_gen_cols", "k_expand_compact", "k_basis_front", "k_meas_onepass", "k_h_pair", "k_h_wave")) and int(r["Grid_Size_X"]) >= 4096:
        short = n[n.find("k_"):].split("(")[0].replace("k_fused_", "").replace(" ", "")
        out.append(f"{short}:{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.2f}")
print(" ".join(out))
