#!/usr/bin/env python3
"""controlled modular multiply at n = 30 (C = 21, M = 5): not reading the lines above row C (cam_skip), nontemporal stores of
the completely rewritten lines (cam_nt_lines), per control position"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

n, M = 30, 5
with qc.Register(n - M, M) as reg:
    reg.set_fusion(-1)
    reg.fill_random(1)
    for skip, ntl, logT in ((1, 0, 11), (1, 0, 10), (1, 0, 9), (1, 0, 8), (1, 0, 12)):
        qc.tune(cam_skip=skip, cam_nt_lines=ntl, cam_logT=logT)
        print("tile 2^%d" % logT)
        for Cn in (21, 31, 9):
            for ctl in (5, 11, 20, 29):
                qc.c_amodc_gate(Cn, 4, ctl, reg); reg.synchronize()
                best = 1e9
                for _ in range(4):
                    reg.timer_start(); qc.c_amodc_gate(Cn, 4, ctl, reg); best = min(best, reg.timer_stop())
                print(f"camodc skip={skip} nt_lines={ntl} C={Cn} ctl={ctl}: {best:.3f} ms", flush=True)
    qc.tune(cam_skip=1, cam_nt_lines=0, cam_logT=11)
