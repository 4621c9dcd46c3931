"""fused Shor N=21 at n=30, part by part (each part flushed on its own): where the time of the circuit goes"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import quantumcomputer_amd as qc
L, M, C, a = 25, 5, 21, 2
n = L + M


def timed(reg, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        reg.synchronize(); t0 = time.perf_counter(); fn(); reg.flush(); reg.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


with qc.Register(L, M) as reg:
    reg.fill_random(1); reg.set_fusion(True)
    def hs():
        for l in range(M, n): qc.hadamard_gate(l, reg)
    def cams():
        x = a % C
        for l in range(M, n):
            qc.c_amodc_gate(C, x, l, reg); x = (x * x) % C
    def iqft(): qc.inverse_QFT(reg)
    def whole(): hs(); cams(); iqft()
    for name, f in (("H sweep over L", hs), ("25 controlled multiplies", cams), ("inverse QFT", iqft), ("whole circuit", whole)):
        p0 = reg.fusion_stats()[0]; dt = timed(reg, f); passes = (reg.fusion_stats()[0] - p0) // 3
        print(f"{name:28s}: {dt*1e3:7.2f} ms  passes={passes}", flush=True)
