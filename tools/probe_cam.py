import sys, time, os
sys.path.insert(0, os.getcwd())
import quantumcomputer_amd as qc
def timed(reg, fn, reps=3):
    best=1e9
    for _ in range(reps):
        reg.synchronize(); t0=time.perf_counter(); fn(); reg.synchronize(); best=min(best,time.perf_counter()-t0)
    return best
with qc.Register(25,5) as reg:
    reg.fill_random(1); reg.set_fusion(True)
    for name, f in (("2 cam ctl=29,28", lambda: [qc.c_amodc_gate(21, 2, 29, reg), qc.c_amodc_gate(21, 4, 28, reg)]),
                    ("10 cam ctl 20..29", lambda: [qc.c_amodc_gate(21, 2, 20 + k, reg) for k in range(10)]),
                    ("10 cam ctl=5 (local)", lambda: [qc.c_amodc_gate(21, 2, 5, reg) for k in range(10)]),
                    ("10 cam ctl=29 (ext)", lambda: [qc.c_amodc_gate(21, 2, 29, reg) for k in range(10)]),
                    ("10 H low", lambda: [qc.hadamard_gate(k % 4, reg) for k in range(10)])):
        p0=reg.fusion_stats()[0]; dt=timed(reg,f); passes=(reg.fusion_stats()[0]-p0)//3
        print(f"{name:24s}: {dt*1e3:7.2f} ms  passes={passes}", flush=True)
