#!/usr/bin/env python3
"""time of the one-read measurement pass alone, with parts of its look-back switched off (meas_dbg): r = -1 makes the
walk stop at the first amplitude, so a wrong guess costs nothing here"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

n = 30
with qc.Register(n, 0) as reg:
    reg.set_fusion(-1)
    for dbg in [1, 0, 1, 0] + [int(x, 0) for x in sys.argv[1:]]:
        qc.tune(meas_dbg=dbg)
        best = 1e9
        for _ in range(3):
            reg.fill_random(5)
            reg.synchronize()
            reg.timer_start()
            qc.measure_state(reg, -1.0)
            best = min(best, reg.timer_stop())
        print(f"meas_dbg={dbg:#x}: scan + collapse {best:7.3f} ms", flush=True)
    qc.tune(meas_dbg=0)
