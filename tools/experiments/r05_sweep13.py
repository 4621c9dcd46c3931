#!/usr/bin/env python3
"""round 5 experiment: the fused n = 30 Hadamard sweep on tiles of 2^13 amplitudes (one 128-KiB workgroup per CU, 256-byte store
runs) against the default 2^12 / c = 3 geometry (two 64-KiB workgroups per CU, 128-byte runs); bit-identical windows"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

n = 30
with qc.Register(n, 0) as reg:
    reg.set_fusion(1)
    ref = None
    for name, tune in (("T=12 c=3 (default)", dict(fuse_hsweep_T=12, fuse_hsweep_c=3)), ("T=13 c=4", dict(fuse_hsweep_T=13, fuse_hsweep_c=4)),
                       ("T=13 c=4, gates skipped", dict(fuse_hsweep_T=13, fuse_hsweep_c=4, fuse_dbg=1)), ("T=12 c=3, gates skipped", dict(fuse_hsweep_T=12, fuse_hsweep_c=3, fuse_dbg=1)),
                       ("T=12 c=4", dict(fuse_hsweep_T=12, fuse_hsweep_c=4))):
        qc.tune(fuse_dbg=0)
        qc.tune(**tune)
        best = 1e9
        for rep in range(4):
            reg.fill_random(7)
            reg.timer_start()
            for q in range(n):
                qc.hadamard_gate(q, reg)
            best = min(best, reg.timer_stop())
        win = np.concatenate([reg.read(s, 1 << 12) for s in (0, 12345 << 12, (1 << n) - (1 << 12))])
        same = ""
        if "skipped" not in name:
            if ref is None:
                ref = win
            same = "  windows identical to the default: " + str(bool(np.array_equal(win.view(np.uint64), ref.view(np.uint64))))
        print(f"{name:28s} {best:7.3f} ms per sweep{same}", flush=True)
