import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc
n, M = 30, 5
with qc.Register(n - M, M) as reg:
    reg.set_fusion(-1)
    reg.fill_random(1)
    for full in (0, 1):
        for cap in (4096, 8192, 16384, 0):
            qc.tune(cam_full=full, cam_grid_cap=cap)
            for ctl in (5, 11, 20, 29):
                qc.c_amodc_gate(21, 4, ctl, reg); reg.synchronize()
                best = 1e9
                for _ in range(3):
                    reg.timer_start(); qc.c_amodc_gate(21, 4, ctl, reg); best = min(best, reg.timer_stop())
                print(f"camodc full={full} cap={cap} ctl={ctl}: {best:.3f} ms", flush=True)
