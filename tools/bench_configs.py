#!/usr/bin/env python3
"""Times the BASELINE.json configurations that fit one GPU, through the C ABI, one launch per gate:
  config 2  n=26 Hadamard sweep                         (26 gates)
  config 3  n=28 IQFT-schedule QFT over all qubits      (28 H + 378 CPHASE)
  config 5' n=30 Shor N=21 a=2 (L=25, M=5) on ONE GPU   (50 H + 25 C_AMODC + 300 CPHASE + measure)
and per-kernel-class GB/s on the algorithmic bytes of SURVEY s8(d).  Writes one JSON object."""
import argparse

FUSION = -1
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc  # noqa: E402


def timed(reg, fn):
    reg.synchronize()
    t0 = time.perf_counter()
    fn()
    reg.synchronize()
    return time.perf_counter() - t0


def cfg2(n=26, reps=5):
    with qc.Register(n, 0) as reg:
        reg.fill_random(7)
        sweep = lambda: [qc.hadamard_gate(q, reg) for q in range(n)]
        sweep()
        dt = min(timed(reg, sweep) for _ in range(reps))
    return dict(config="n=%d H sweep" % n, gates=n, seconds=dt, amplitude_updates_per_s=n * 2.0 ** n / dt,
                hbm_gbs=n * 32 * 2.0 ** n / dt / 1e9)


def cfg3(n=28, reps=3):
    with qc.Register(n, 0) as reg:
        reg.set_fusion(FUSION)
        reg.fill_random(7)
        run = lambda: qc.inverse_QFT(reg)
        run()
        dt = min(timed(reg, run) for _ in range(reps))
        # per class
        th = timed(reg, lambda: [qc.hadamard_gate(q, reg) for q in range(n)])
        tp = timed(reg, lambda: [qc.c_phase_shift_gate(l, k, math.pi / (1 << (l - k)), reg)
                                 for l in range(n - 1, -1, -1) for k in range(l - 1, -1, -1)])
    nh, npz = n, n * (n - 1) // 2
    alg = nh * 32 * 2.0 ** n + npz * 32 * 2.0 ** (n - 2)
    return dict(config="n=%d IQFT schedule (%d H + %d CPHASE)" % (n, nh, npz), gates=nh + npz, seconds=dt,
                amplitude_updates_per_s=(nh + npz) * 2.0 ** n / dt, algorithmic_gbs=alg / dt / 1e9,
                h_only_seconds=th, cphase_only_seconds=tp, cphase_gbs_on_quarter=npz * 32 * 2.0 ** (n - 2) / tp / 1e9)


def cfg5(L=25, M=5, Cn=21, a=2, seed=12345):
    n = L + M
    rng = qc.Rng(seed)
    with qc.Register(L, M) as reg:
        reg.set_fusion(FUSION)
        qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); reg.synchronize()     # warm-up
        qc.reset_register(reg)
        dt = timed(reg, lambda: qc.quantum_computation(Cn, a, reg))
        norm = reg.norm2()
        t0 = time.perf_counter()
        idx = qc.measure_state(reg, rng)
        tm = time.perf_counter() - t0
        w = qc.read_omega(idx, reg)
        # C_AMODC alone
        reg.fill_random(3)
        tc = timed(reg, lambda: [qc.c_amodc_gate(Cn, 2 ** (1 << (k % 3)), M + k, reg) for k in range(L)])
    gates = 3 * L + L * (L - 1) // 2
    return dict(config="n=%d Shor N=%d a=%d L=%d M=%d on one GPU" % (n, Cn, a, L, M), gates=gates, circuit_seconds=dt,
                amplitude_updates_per_s=gates * 2.0 ** n / dt, measure_seconds=tm, measured_index=idx, omega=w,
                nearest_sixth=min((abs(w - k / 6.0), k) for k in range(7))[1], total_probability=norm,
                camodc_seconds_per_gate=tc / L, camodc_gbs_on_half=32 * 2.0 ** (n - 1) / (tc / L) / 1e9)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/bench_configs.json")
    ap.add_argument("--skip5", action="store_true")
    ap.add_argument("--fusion", type=int, default=-1, help="-1: one launch per gate (default here), 0: whole-circuit calls as fused passes (the library default), 1: everything queued")
    a = ap.parse_args()
    FUSION = a.fusion
    out = {"fusion_mode": FUSION, "config2": cfg2(), "config3": cfg3()}
    print(json.dumps(out["config2"])); print(json.dumps(out["config3"]), flush=True)
    if not a.skip5:
        out["config5_single_gpu"] = cfg5()
        print(json.dumps(out["config5_single_gpu"]))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
