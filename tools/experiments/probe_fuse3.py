"""Cost model probe for fused passes: time of one pass as a function of the number of phase gates in it.
   usage: python tools/probe_fuse3.py [n]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import quantumcomputer_amd as qc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
TH = 0.3


def timed(reg, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        reg.synchronize(); t0 = time.perf_counter(); fn(); reg.flush(); reg.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


with qc.Register(n, 0) as reg:
    reg.fill_random(1); reg.set_fusion(True)
    qc.tune(fuse_T=11, fuse_c=4)
    cases = {
        "4regs (ctl=1,tgt=0)": (1, 0),
        "2regs (ctl=4,tgt=0)": (4, 0),
        "1reg  (ctl=4,tgt=10)": (4, 10),
        "2regs ext (ctl=4,tgt=n-1)": (4, n - 1),
        "4regs wavebit (ctl=9,tgt=0)": (9, 0),
        "4regs ext2 (ctl=n-2,tgt=n-1)": (n - 2, n - 1),
    }
    for name, (c, t) in cases.items():
        res = []
        for k in (0, 32, 64, 128):
            def run():
                qc.hadamard_gate(4, reg); qc.hadamard_gate(4, reg)
                for _ in range(k): qc.c_phase_shift_gate(c, t, TH, reg)
            p0 = reg.fusion_stats()[0]
            dt = timed(reg, run)
            passes = (reg.fusion_stats()[0] - p0) // 3
            res.append((k, dt * 1e3, passes))
        slope = (res[-1][1] - res[1][1]) / (res[-1][0] - res[1][0])
        print(f"{name:32s} " + " ".join(f"k={k}:{ms:6.2f}ms/{p}p" for k, ms, p in res) + f"  slope {slope*1e3:6.1f} us/op", flush=True)

    # per-run overhead: R runs of one gate (H between them, same register bit) against one run of R gates
    for name, (c, t) in (("2regs (ctl=4,tgt=0)", (4, 0)), ("2regs ext (ctl=4,tgt=n-1)", (4, n - 1))):
        for R in (16, 64):
            def split():
                for _ in range(R): qc.hadamard_gate(4, reg); qc.c_phase_shift_gate(c, t, TH, reg)
            def joined():
                for _ in range(R): qc.hadamard_gate(4, reg)
                for _ in range(R): qc.c_phase_shift_gate(c, t, TH, reg)
            def honly():
                for _ in range(R): qc.hadamard_gate(4, reg)
            a, b, h = timed(reg, split), timed(reg, joined), timed(reg, honly)
            print(f"runs {name:28s} R={R}: split {a*1e3:6.2f} ms, joined {b*1e3:6.2f} ms, H only {h*1e3:6.2f} ms -> per-run overhead {(a-b)/R*1e6:6.1f} us, per H {(h)/R*1e6:6.1f} us", flush=True)
