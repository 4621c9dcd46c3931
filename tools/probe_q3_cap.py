import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc
with qc.Register(28, 0) as reg:
    reg.set_fusion(2); reg.fill_random(1)
    for cap in (1536, 2048, 2560, 3072, 4096, 6144):
        for dbg in (0, 1):
            qc.tune(fuse_q3_cap=cap, fuse_dbg=dbg)
            qc.inverse_QFT(reg); reg.synchronize()
            best = 1e9
            for _ in range(3):
                reg.timer_start(); qc.inverse_QFT(reg); best = min(best, reg.timer_stop())
            print(f"q3 cap={cap} dbg={dbg}: {best:.3f} ms", flush=True)
