#!/usr/bin/env python3
"""bench.py -- headline benchmark: Hadamard sweep over every qubit of an n-qubit register
(BASELINE.json: "amplitude-updates/s (gate*2^n/s) and HBM GB/s vs roofline, n=30 H-sweep").

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
              --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one sweep: hadamard_gate(q) for q = 0..n-1, one kernel launch per gate through the C ABI
(libqcx.so), state resident in HBM.  N = 1: n = 30 (16 GiB).  N > 1 is weak scaling: every rank keeps
a 2^30-amplitude shard (n = 30 + log2 N); the top log2 N qubits are global and cost all-to-all
exchanges over RCCL (quantumcomputer_amd/sharded.py).  value = steps * n * 2^n / seconds, where
seconds is the barrier-to-barrier wall time, max over ranks.

Extra objects on the JSON line:
  roofline      the dominant kernel (k_h_pair, the pair-form Hadamard used for q >= 3): algorithmic
                bytes per launch (32 B per amplitude = 32 * 2^n_local) / mean launch duration measured
                with HIP events recorded between the gates of the timed region, vs 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (oracle/, a port of the reference arithmetic, pairwise in place,
                OpenMP over the host cores) on a bounded sample of the same workload (smaller n).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(n_sample, budget_s=12.0):
    """oracle (kind "port") timed on this box's host cores: H-sweep at n_sample qubits"""
    from oracle import binding as ob
    import numpy as np
    # threads = the CPUs this process may use, capped at the GPU box's per-GPU CPU share
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("QCX_CPU_THREADS", "16"))))
    a = ob.fill_random(n_sample, 1)
    ob.hadamard(a, n_sample, 0, cores)                      # warm-up: page in, spin up the OpenMP team
    t0 = time.perf_counter()
    gates = 0
    while time.perf_counter() - t0 < budget_s:             # whole sweeps until ~budget_s of CPU work is done
        for q in range(n_sample):
            ob.hadamard(a, n_sample, q, cores)
            gates += 1
    dt = time.perf_counter() - t0
    # the literal reference algorithm (scan 4^n index pairs, build COO, mat-vec), 1 thread, for context
    lit_n = 11
    R = ob.LiteralRegister(lit_n, 0)
    R.set_state(ob.fill_random(lit_n, 1))
    t1 = time.perf_counter(); R.hadamard(lit_n - 1); lit_dt = time.perf_counter() - t1
    R.close()
    del a, np
    return {"value": gates * float(1 << n_sample) / dt, "unit": "amplitude-updates/s", "cores": cores, "kind": "port",
            "sample": f"{gates // n_sample} Hadamard sweeps q=0..{n_sample - 1} of an n={n_sample} register ({gates} gates, {dt:.1f} s), "
                      f"oracle pairwise in-place form, OpenMP {cores} threads",
            "literal_reference_algorithm": {"value": float(1 << lit_n) / lit_dt, "unit": "amplitude-updates/s", "cores": 1,
                                            "sample": f"one hadamard_gate at n={lit_n}: 4^n index-pair scan + COO mat-vec ({lit_dt * 1e3:.1f} ms)"}}


def load_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes, if any"""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):     # latest pass first
        try:
            return json.load(open(p)).get("k_h_pair_bytes_per_launch_n30")
        except Exception:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-local", type=int, default=30, help="qubits per GPU shard (30 = 16 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path (ShardedRegister) even at world size 1")
    ap.add_argument("--cpu-n", type=int, default=28)
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON); libraries that chat on fd 1 (RCCL prints a version banner
    # there when a communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import quantumcomputer_amd as qc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("QCX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))     # (override: test rigs only)
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    qc.lib()

    k = args.gpus.bit_length() - 1
    assert (1 << k) == args.gpus, "--gpus must be a power of two"
    n = args.n_local + k
    gates_per_step = n
    dim = float(1 << n)

    if args.gpus == 1 and not args.force_sharded:
        reg = qc.Register(n, 0)
        reg.fill_random(1)
        nev = args.steps * (gates_per_step + 1)
        reg.events_create(nev)

        def sweep(record_base=None):
            for q in range(n):
                if record_base is not None:
                    reg.event_record(record_base + q)
                qc.hadamard_gate(q, reg)
            if record_base is not None:
                reg.event_record(record_base + n)

        for _ in range(args.warmup):
            sweep()
        reg.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(args.steps):
            sweep(s * (gates_per_step + 1))
        reg.synchronize(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0

        per_q_ms = [[reg.event_elapsed(s * (n + 1) + q, s * (n + 1) + q + 1) for s in range(args.steps)] for q in range(n)]
        norm = reg.norm2()

        # reported separately: the same sweep with gate fusion on -- the 30 gate calls are queued
        # and run as a few fused LDS-tile passes, bit-identical results; bytes are counted once per pass
        fused = None
        if not args.no_fused:
            reg.set_fusion(True)
            sweep(); reg.synchronize()
            p0 = reg.fusion_stats()[0]
            tf0 = time.perf_counter()
            for s in range(args.steps):
                sweep()
            reg.synchronize()
            tf = time.perf_counter() - tf0
            passes = (reg.fusion_stats()[0] - p0) / args.steps
            fused = {"value": args.steps * gates_per_step * dim / tf, "unit": "amplitude-updates/s", "ms_per_step": tf / args.steps * 1e3,
                     "hbm_passes_per_sweep": passes, "hbm_gbs_per_pass": passes * 32.0 * dim / (tf / args.steps) / 1e9,
                     "note": "qcx_set_fusion(1): same 30 hadamard_gate calls, executed as fused passes over LDS tiles"}
            reg.set_fusion(False)
        reg.close()
        exchanges = 0
    else:
        import torch.distributed as dist
        from quantumcomputer_amd.sharded import ShardedRegister
        backend = os.environ.get("QCX_BENCH_BACKEND", "nccl")        # (gloo: rehearsal of the N > 1 path on one GPU, test rigs only)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        reg = ShardedRegister(n, 0, fusion=False)       # one launch per gate, like the N = 1 headline
        reg.fill_random(1)

        def sweep():
            for q in range(n):
                reg.hadamard_gate(q)          # queued; executed on flush with look-ahead eviction

        for _ in range(args.warmup):
            sweep()
        reg.synchronize(); dist.barrier(); torch.cuda.synchronize()
        ex0 = reg.exchanges
        reg.profile = []
        t0 = time.perf_counter()
        for s in range(args.steps):
            sweep()
        reg.synchronize(); dist.barrier(); torch.cuda.synchronize()
        dt_local = time.perf_counter() - t0
        t = torch.tensor([dt_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        prof, reg.profile = reg.profile, None
        # Hadamard launches by physical target position; a launch on a slice moves 32 * 2^bits bytes
        per_q_ms = [[] for _ in range(n)]
        h_bytes, h_ms = 0.0, 0.0
        for kind, pq, e0, e1, bits_ in prof:
            if kind != "h":
                continue
            ms = e0.elapsed_time(e1)
            h_bytes += 32.0 * float(1 << bits_); h_ms += ms
            per_q_ms[pq].append(ms * float(1 << (args.n_local - bits_)))      # scaled to a whole-shard launch
        exch_ms = []
        overlapped = reg.overlapped_gates
        norm = reg.norm2()
        exchanges = reg.exchanges - ex0
        fused = None

    if rank == 0:
        bytes_per_launch = 32.0 * float(1 << args.n_local)           # 16 B read + 16 B written per amplitude, per GPU
        # dominant kernel = pair-form Hadamard: every local target qubit >= 3 (wave-tile form below that);
        # on N > 1 the event interval of a global qubit also holds the all-to-all, so those are left out
        dom = [q for q in range(3, args.n_local)]
        dom_ms = [x for q in dom for x in per_q_ms[q]]
        avg_ms = sum(dom_ms) / len(dom_ms)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # (index = physical target bit; null where no launch hit that bit, e.g. the rank-id bits at N > 1)
        per_q_gbs = [round(bytes_per_launch / (min(per_q_ms[q]) * 1e-3) / 1e9, 1) if per_q_ms[q] else None for q in range(n)]
        out = {
            "metric": "amplitude-updates/s (gate*2^n/s), n=30 H-sweep" if args.gpus == 1 else
                      f"amplitude-updates/s (gate*2^n/s), n={n} H-sweep sharded over {args.gpus} GPUs",
            "value": args.steps * gates_per_step * dim / dt,
            "unit": "amplitude-updates/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"n={n} Hadamard sweep q=0..{n - 1} (config 2 of BASELINE.json at the headline size n=30), "
                                   f"one launch per gate through libqcx.so, complex128 state of {16 * dim / 2**30:.0f} GiB in HBM",
                       "qubits": n, "gates_per_step": gates_per_step, "shard_qubits": args.n_local,
                       "parallelism": "1 GPU" if args.gpus == 1 else f"state sharded by top {k} qubits over {args.gpus} ranks, "
                                      f"all-to-all qubit remap for global targets ({exchanges} exchanges in the {args.steps} timed sweeps"
                                      + (f", exchange overlapped with the neighbouring gates on {1 << reg.sigma} slices)" if args.gpus > 1 or args.force_sharded else ")")},
            "hbm_gbs_sweep_average": args.steps * gates_per_step * bytes_per_launch * args.gpus / dt / 1e9,
            "roofline": {"bound": "hbm", "kernel": "qcx::k_h_pair (Hadamard, pair form, target qubit >= 3)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                         "launches_timed": len(dom_ms),
                         "traffic": load_traffic() if args.n_local == 30 else None},     # PMC passes were taken at n = 30
            "per_qubit_gbs": per_q_gbs,
            "fused_sweep": fused if args.gpus == 1 else None,
            "total_probability_after": norm,
        }
        if not args.no_cpu_baseline and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if args.gpus > 1 or args.force_sharded:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
