#!/usr/bin/env python3
"""measure_state at n = 30: the one-read scan (K4c: look-back + tree walk) against round 3's two-read scan (K4b), on the
Shor N = 21 final state (peaked) and on a dense random state (about thirty binade crossings of the running sum)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def stats():
    s, b = C.c_uint(0), C.c_uint(0)
    qc.lib().qcx_measure_last_stats(C.byref(s), C.byref(b))
    return s.value, b.value


n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L, M = n - 5, 5
with qc.Register(L, M) as reg:
    reg.set_fusion(-1)                          # eager collapse: the timed region holds the scan AND the collapse's memset
    for state in ("shor", "dense"):
        for onepass, dbg in ((1, 0), (0, 0)):        # (the column says "onepass": it is meas_fast since round 5)
            qc.tune(meas_dbg=dbg, meas_fast=onepass)
            for r in (0.3, 0.77, 0.999):
                if state == "shor":
                    reg.set_fusion(0); qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.set_fusion(-1)
                else:
                    reg.fill_random(5)
                reg.synchronize()
                reg.timer_start()
                p = reg.total_probability()          # the scan alone, run to the end (r = +inf)
                t_scan = reg.timer_stop()
                s_scan = stats()
                reg.timer_start()
                idx = qc.measure_state(reg, r)
                t_meas = reg.timer_stop()
                print(f"n={n} {state:5s} fast={onepass} dbg={dbg} r={r}: index {idx}  scan-to-end {t_scan:7.3f} ms (slow/blocks {s_scan}, P={p!r})  "
                      f"measure+collapse {t_meas:7.3f} ms (slow/blocks {stats()})", flush=True)
