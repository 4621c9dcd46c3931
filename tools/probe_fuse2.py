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
    for T,c,pipe,pg in ((11,4,1,512),(12,3,1,256),(12,3,1,512),(12,3,0,0),(11,4,0,0),(12,4,0,0),(11,3,1,512),(12,3,1,384)):
        qc.tune(fuse_T=T, fuse_c=c, fuse_pipe=pipe, fuse_pipe_grid=max(pg,1))
        p0=reg.fusion_stats()[0]; dt=timed(reg, lambda: [qc.hadamard_gate(q, reg) for q in range(n)]); passes=(reg.fusion_stats()[0]-p0)//3
        print(f"T={T} c={c} pipe={pipe} grid={pg}: sweep {dt*1e3:7.2f} ms passes={passes}", flush=True)
