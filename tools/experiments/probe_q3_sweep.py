import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc
n = 30
with qc.Register(n, 0) as reg:
    reg.set_fusion(1); reg.fill_random(1)
    def sweep():
        for q in range(n): qc.hadamard_gate(q, reg)
    for q3, cap, dbg in [(0, 3072, 0), (1, 1024, 0), (1, 2048, 0), (1, 3072, 0), (1, 4096, 0), (1, 8192, 0), (1, 24576, 0), (1, 3072, 1)]:
        qc.tune(fuse_q3=q3, fuse_q3_cap=cap, fuse_dbg=dbg)
        sweep(); reg.synchronize()
        best = 1e9
        for _ in range(3):
            reg.timer_start(); sweep(); best = min(best, reg.timer_stop())
        print(f"sweep30 q3={q3} cap={cap} dbg={dbg}: {best:.3f} ms", flush=True)
