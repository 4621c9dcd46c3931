import sys, time, os
sys.path.insert(0, os.getcwd())
import quantumcomputer_amd as qc
n=30
def timed(reg, fn, reps=3):
    best=1e9
    for _ in range(reps):
        reg.synchronize(); t0=time.perf_counter(); fn(); reg.synchronize(); best=min(best,time.perf_counter()-t0)
    return best
with qc.Register(n,0) as reg:
    reg.fill_random(1); reg.set_fusion(True)
    for T,c,dma,pg in ((11,4,0,0),(11,4,1,512),(11,4,1,768),(11,4,1,1024),(10,4,1,1024),(10,4,1,1536),(10,4,1,2048),(12,4,1,256),(12,4,1,512)):
        qc.tune(fuse_T=T, fuse_c=c, fuse_ldsdma=1, fuse_pipe=dma, fuse_pipe_grid=max(pg,1))
        for name, qs in (("low0-3", [0,1,2,3]), ("low0-10", list(range(0,11))), ("two low", [0,1]), ("hi11-17", list(range(11,18))), ("hi 2 gates", [11,12]),
                         ("hi22-28", list(range(22,29))), ("phase only x2", None), ("phase x16", "p16")):
            if qs is None:
                f=lambda: [qc.c_phase_shift_gate(29, 3, 0.3, reg), qc.c_phase_shift_gate(28, 2, 0.2, reg)]
            elif qs == "p16":
                f=lambda: [qc.c_phase_shift_gate(29, k, 0.3, reg) for k in range(16)]
            else:
                f=lambda qs=qs: [qc.hadamard_gate(q, reg) for q in qs]
            p0=reg.fusion_stats()[0]; dt=timed(reg,f); passes=(reg.fusion_stats()[0]-p0)//3
            print(f"T={T} c={c} pipe={dma} grid={pg} {name:14s}: {dt*1e3:7.2f} ms  passes={passes}", flush=True)
