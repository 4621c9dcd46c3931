#!/usr/bin/env python3
"""Per-gate kernels at n = 30, one launch per gate: controlled phase (k_phase) and controlled modular multiply
(k_camodc) over the target / control positions that matter (mask bits below 3 = partial 128-B lines, inside / above the
LDS tile ...).  Each case is launched REPS times back to back in a fixed order, so that the per-dispatch rows of a
rocprofv3 --pmc pass of this script can be matched to the cases (tools/summarize_prof.py --per-dispatch).
Prints one JSON object: per case the HIP-event time and GB/s on the algorithmic bytes of SURVEY s8(d)
(phase: 32 * 2^(n-2), modular multiply: 32 * 2^(n-1)).
usage: probe_gates.py [-n 30] [--reps 3] [--out gpurun_out/probe_gates.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

PHASE_CASES = [(1, 0), (5, 0), (13, 1), (5, 2), (13, 2), (13, 5), (28, 5), (28, 13), (29, 28), (20, 3)]
CAM_CTLS = [5, 10, 11, 20, 29]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/probe_gates.json")
    a = ap.parse_args()
    n, M = a.n, 5
    out = {"n": n, "reps": a.reps, "phase": {}, "camodc": {}, "order": []}
    with qc.Register(n - M, M) as reg:
        reg.set_fusion(-1)
        reg.fill_random(1)
        qc.c_phase_shift_gate(n - 1, n - 2, 0.3, reg); reg.synchronize()          # warm-up (not in `order`: one extra k_phase dispatch first)
        for c, t in PHASE_CASES:
            c, t = min(c, n - 1), min(t, n - 2)
            ms = []
            for _ in range(a.reps):
                reg.timer_start()
                qc.c_phase_shift_gate(c, t, 0.3, reg)
                ms.append(reg.timer_stop())
            alg = 32.0 * 2.0 ** (n - 2)
            out["phase"][f"c{c}_t{t}"] = dict(ms=ms, best_ms=min(ms), gbs_on_quarter=alg / (min(ms) * 1e-3) / 1e9, algorithmic_bytes=alg)
            out["order"].append(["k_phase", f"c{c}_t{t}", a.reps])
            print(f"phase c={c:2d} t={t:2d}: {min(ms):7.3f} ms  {alg / (min(ms) * 1e-3) / 1e9:7.0f} GB/s on the touched quarter", flush=True)
        qc.c_amodc_gate(21, 2, n - 1, reg); reg.synchronize()
        for ctl in CAM_CTLS:
            ctl = min(ctl, n - 1)
            ms = []
            for _ in range(a.reps):
                reg.timer_start()
                qc.c_amodc_gate(21, 4, ctl, reg)
                ms.append(reg.timer_stop())
            alg = 32.0 * 2.0 ** (n - 1)
            out["camodc"][f"ctl{ctl}"] = dict(ms=ms, best_ms=min(ms), gbs_on_half=alg / (min(ms) * 1e-3) / 1e9, algorithmic_bytes=alg)
            out["order"].append(["k_camodc", f"ctl{ctl}", a.reps])
            print(f"camodc ctl={ctl:2d}: {min(ms):7.3f} ms  {alg / (min(ms) * 1e-3) / 1e9:7.0f} GB/s on the control half", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
