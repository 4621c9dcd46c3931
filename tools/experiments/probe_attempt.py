#!/usr/bin/env python3
"""one period-finding attempt at n = 30 (Shor N = 21, a = 2, L = 25, M = 5): reset + quantum_computation + measure_state with no
flush in between -- the measurement reads the compact form a compact chain leaves behind -- against the same with the state
expanded first (fuse_compact_lazy = 0) and without compact chains; HIP events around the three calls, best of 5."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

for mode in (0, 2):
    for name, tune in (("compact chain, measured compact", dict(fuse_compact=1, fuse_compact_lazy=1)),
                       ("compact chain, expanded first", dict(fuse_compact=1, fuse_compact_lazy=0)),
                       ("no compact chain", dict(fuse_compact=0))):
        qc.tune(**tune)
        with qc.Register(25, 5) as reg:
            reg.set_fusion(mode)
            best, idx = 1e9, None
            for rep in range(6):
                reg.timer_start()
                qc.reset_register(reg); qc.quantum_computation(21, 2, reg); idx = qc.measure_state(reg, 0.37)
                t = reg.timer_stop()
                if rep:
                    best = min(best, t)
            print(f"mode {mode} {name:34s}: {best:7.3f} ms per attempt   index {idx} omega {qc.read_omega(idx, reg):.6f}", flush=True)
