#!/usr/bin/env python3
"""bench.py -- headline benchmark: Hadamard sweep over every qubit of an n-qubit register
(BASELINE.json: "amplitude-updates/s (gate*2^n/s) and HBM GB/s vs roofline, n=30 H-sweep").

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1, either form:
      python bench.py --gpus N ...                         (starts its N ranks itself, see self_launch)
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
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
                `traffic` is NOT measured in this run: it is the per-launch HBM byte count of the committed
                rocprofv3 --pmc passes of this same command (`traffic_source` names the file).
  cpu_baseline  the CPU oracle (oracle/, a port of the reference arithmetic) on bounded samples of the same
                workload: tier T3 = pairwise in place, OpenMP over the host cores (the headline `value`),
                T2 = the reference's COO mat-vec alone, T1 = the literal reference algorithm.
  fused_sweep   (N = 1) the same 30 gate calls with qcx_set_fusion(1): passes, GB/s per pass, roofline fraction.
  configs       (N = 1) the other BASELINE.json configurations that fit one GPU, driver-timed (plus kernels_n30: the per-gate
                kernels next to the headline one -- modular multiply, controlled phase, measurement scan, circuit front -- against
                their algorithmic bytes): config 2 (n=26 H sweep),
                config 3 (n=28 qcx_inverse_QFT: fused passes = the default, one launch per gate, and -- when the build
                has it -- the opt-in tolerance mode), config 5 on one GPU (n=30 qcx_quantum_computation(21, 2) +
                measure_state).  Each with ms, passes, GB/s per pass, FP64-op/s and the fraction of each roof.
  config4       (N > 1) BASELINE config 4: H on every global qubit vs a local one at n = 29 + log2 N
                (n = 32 on 8 GPUs), exchange GB/s per GPU.
  config5       (N > 1) BASELINE config 5 in its N-GPU form: n = 30 Shor N = 21 (L = 25, M = 5) sharded over the ranks, circuit
                time, exchanges, measured omega (also inside c_host for the one-process host).
  c_host        (N > 1) the same sweep and the config-4 shape through the ONE-process C-ABI sharded register
                (qcx_register_create_sharded: peer stores over xGMI, no RCCL), run by a fresh child process of rank 0
                before rank 0 touches a GPU; the register checks its exchange bit for bit at creation.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2**20
    except OSError:
        pass
    return 0.0


def cpu_baseline(n_sample, budget_s=12.0):
    """oracle (kind "port") timed on this box's host cores: H-sweep at n_sample qubits (tier T3), plus the
    reference's own two costs for context: its COO mat-vec alone (T2) and build + mat-vec (T1)"""
    from oracle import binding as ob
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
    del a
    # T1: the literal reference algorithm (scan 4^n index pairs, build COO, mat-vec), 1 thread
    lit_n = 11
    R = ob.LiteralRegister(lit_n, 0)
    R.set_state(ob.fill_random(lit_n, 1))
    t1 = time.perf_counter(); R.hadamard(lit_n - 1); lit_dt = time.perf_counter() - t1
    R.close()
    # T2: the reference's mat-vec (Q:396-413) on a COO matrix built in O(2^n) instead of by the 4^n scan, 1 thread
    spmv_n = 22
    R = ob.LiteralRegister(spmv_n, 0)
    R.set_state(ob.fill_random(spmv_n, 1))
    R.spmv_hadamard(0)                                      # first call grows the COO arrays
    t2 = time.perf_counter()
    for q in (0, spmv_n // 2, spmv_n - 1):
        R.spmv_hadamard(q)
    spmv_dt = (time.perf_counter() - t2) / 3
    R.close()
    return {"value": gates * float(1 << n_sample) / dt, "unit": "amplitude-updates/s", "cores": cores, "kind": "port", "tier": "T3",
            "sample": f"{gates // n_sample} Hadamard sweeps q=0..{n_sample - 1} of an n={n_sample} register ({gates} gates, {dt:.1f} s), "
                      f"oracle pairwise in-place form, OpenMP {cores} threads",
            "reference_matvec_only": {"tier": "T2", "value": float(1 << spmv_n) / spmv_dt, "unit": "amplitude-updates/s", "cores": 1,
                                      "sample": f"hadamard_gate at n={spmv_n}, q in (0, {spmv_n // 2}, {spmv_n - 1}): COO triplets laid down in "
                                                f"O(2^n), then the reference's mat-vec loop ({spmv_dt * 1e3:.0f} ms per gate)"},
            "literal_reference_algorithm": {"tier": "T1", "value": float(1 << lit_n) / lit_dt, "unit": "amplitude-updates/s", "cores": 1,
                                            "sample": f"one hadamard_gate at n={lit_n}: 4^n index-pair scan + COO mat-vec ({lit_dt * 1e3:.1f} ms)"}}


def load_traffic():
    """(bytes per launch, source file) of the dominant kernel from the latest committed rocprofv3 --pmc summary"""
    import glob
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):     # latest round first
        try:
            v = json.load(open(p)).get("k_h_pair_bytes_per_launch_n30")
        except Exception:
            continue
        if v:
            return v, os.path.relpath(p, ROOT)
    return None, None


def run_c_host(args):
    """rank 0, before it touches a GPU: the C-host leg in a fresh child process; returns its JSON object"""
    cmd = [sys.executable, os.path.abspath(__file__), "--c-host-child", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--n-local", str(args.n_local)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                           "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "QCX_FORCE_DEVICE")}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=float(os.environ.get("QCX_BENCH_CHOST_TIMEOUT_S", "200")))
    except subprocess.TimeoutExpired:
        return {"error": "the C-host child timed out"}
    for ln in reversed(r.stdout.decode(errors="replace").splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except ValueError:
                break
    return {"error": f"the C-host child failed (exit {r.returncode}): {r.stderr.decode(errors='replace')[-400:]}"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(ngpus, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (python -m torch.distributed.run, one rank per GPU) BEFORE anything in this process touches the GPU, relay
    rank 0's JSON line, return the children's exit status.  Nothing is exec'ed: this process only waits."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    line = None
    for ln in r.stdout.decode(errors="replace").splitlines():
        s = ln.strip()
        if s.startswith("{") and s.endswith("}"):
            line = s
        elif s:
            print(s, file=sys.stderr)
    if r.returncode != 0 or line is None:
        print(f"bench.py: the {ngpus}-rank run failed (exit {r.returncode}, JSON line {'present' if line else 'missing'})", file=sys.stderr)
        return r.returncode or 1
    print(line, flush=True)
    return 0


def exchange_selfcheck(make_reg, torch, dist, modes=("overlap", "sync", "pairwise")):
    """N > 1, before the timed region: H on a global qubit applied twice must give the state back (to rounding) --
    run once through the overlapped, sliced all-to-all exchange, if that fails once through the synchronous unsliced one,
    and if that fails too through the PAIRWISE form (one rank bit per exchange: half a shard to rank ^ 2^j with
    ncclSend/ncclRecv -- SURVEY s8(e)'s literal form, quantumcomputer_amd/sharded.py).  `modes` narrows the list.
    Returns the mode that works; raises if none does.  (A wrong exchange would still produce a plausible-looking
    throughput number; this keeps such a number from being printed.)"""
    for mode in modes:
        reg = make_reg(mode)
        reg.fill_random(5)
        reg.synchronize()
        before = reg.shard.clone()
        n = reg.num_qubits
        for q in (n - 1, 3, n - 1, 3):                       # global, local, and back
            reg.hadamard_gate(q)
        reg._identity()                                      # flush + restore the identity layout
        torch.cuda.synchronize()
        err = (reg.shard - before).abs().max()
        ref = before.abs().max()
        t = torch.stack([err, ref])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = bool(t[0] <= 1e-12 * t[1]) and reg.exchanges >= 1
        del reg, before
        if ok:
            return mode
    raise RuntimeError(f"sharded exchange self-check failed in every mode tried {tuple(modes)} (H.H != identity across ranks)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-local", type=int, default=30, help="qubits per GPU shard (30 = 16 GiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--no-config4", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path (ShardedRegister) even at world size 1")
    ap.add_argument("--cpu-n", type=int, default=0, help="register size of the CPU sample (0: 30 if the host has the memory, else 28)")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the configs object (BASELINE configs 2, 3, 5)")
    ap.add_argument("--no-c-host", action="store_true", help="N > 1: skip the one-process C-host leg")
    ap.add_argument("--c-host-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.c_host_child:
        return c_host_child(args)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line (the JSON); libraries that chat on fd 1 (RCCL prints a version banner
    # there when a communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # N > 1: the one-process C host over the same GPUs, in a fresh child process, before this rank initialises a GPU
    # (the other ranks wait for rank 0 in the rendezvous meanwhile)
    c_host = run_c_host(args) if (world > 1 and rank == 0 and not args.no_c_host) else None

    import torch
    import quantumcomputer_amd as qc

    local_rank = int(os.environ.get("QCX_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))     # (override: test rigs only)
    args.gpus = world
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        sys.exit(f"bench.py: rank {rank} wants device {local_rank} but {ndev} are visible "
                 "(one rank per GPU; QCX_FORCE_DEVICE + QCX_BENCH_BACKEND=gloo rehearse several ranks on one GPU)")
    torch.cuda.set_device(local_rank)
    qc.lib()

    k = args.gpus.bit_length() - 1
    assert (1 << k) == args.gpus, "--gpus must be a power of two"
    n = args.n_local + k
    gates_per_step = n
    dim = float(1 << n)
    sharded = args.gpus > 1 or args.force_sharded
    config4 = None
    config5 = None
    exchange_mode = None

    if not sharded:
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

        norm_before = reg.norm2()                        # (the synthetic fill is normalised only in expectation)
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
            gbs_pass = passes * 32.0 * dim / (tf / args.steps) / 1e9
            fused = {"value": args.steps * gates_per_step * dim / tf, "unit": "amplitude-updates/s", "ms_per_step": tf / args.steps * 1e3,
                     "hbm_passes_per_sweep": passes, "hbm_gbs_per_pass": gbs_pass, "roofline_frac": gbs_pass / HBM_PEAK_GBS,
                     "note": "qcx_set_fusion(1): same 30 hadamard_gate calls, executed as fused passes over LDS tiles; "
                             "32 B per amplitude are counted once per pass.  An all-Hadamard queue is planned on 2^12-amplitude "
                             "tiles (radix-8 rounds): 3 passes per 30-qubit sweep, CHAINED through the register's second buffer "
                             "(round 4): every pass reads whole 64-KiB tiles and stores 128-B runs under the layout the next pass "
                             "reads contiguously; the last one stores the identity layout.  In place (round 3) the same passes took "
                             "22.7 ms, chained 19.2-19.4; round 5: the workgroups of one XCD take the tiles whose 128-B runs are neighbours in "
                             "the output (fuse_stream_tile): 18.0-18.4 ms.  A pass WITHOUT gates in the same shell moves its tiles at "
                             "5.0-5.5 TB/s with 128-B runs (tools/experiments/tile_shell.hip, profiles/r05_tile_shell.txt): memory-bound"}
            reg.set_fusion(False)
        reg.close()
        exchanges = 0
        configs = None if (args.no_configs or args.n_local != 30) else single_gpu_configs(qc)
    else:
        import datetime
        import torch.distributed as dist
        from quantumcomputer_amd.sharded import ShardedRegister
        backend = os.environ.get("QCX_BENCH_BACKEND", "nccl")        # (gloo: rehearsal of the N > 1 path on one GPU, test rigs only)
        if world == 1:                                               # --force-sharded without a launcher: a one-rank job
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port()))
        tmo = datetime.timedelta(seconds=int(os.environ.get("QCX_BENCH_TIMEOUT_S", "300")))   # a hung collective ends the run, loudly
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

        def make(mode, nq, **kw):
            if mode in ("sync", "pairwise"):                         # unsliced, no overlap, async_op=False
                kw = dict(kw, slices_log2=0)
            if mode == "pairwise":                                   # one rank bit per exchange, half a shard to rank ^ 2^j
                kw = dict(kw, exchange="pairwise")
            r = ShardedRegister(nq, 0, fusion=False, **kw)           # one launch per gate, like the N = 1 headline
            if mode in ("sync", "pairwise"):
                r.overlap = r.async_exchange = False
            return r

        exchange_mode = "none (1 rank)"
        if args.gpus > 1:
            # QCX_SHARD_EXCHANGE=pairwise / QCX_SHARD_OVERLAP=0 pin the mode (still self-checked); default: the first that works
            if os.environ.get("QCX_SHARD_EXCHANGE", "").lower() == "pairwise":
                modes = ("pairwise",)
            elif os.environ.get("QCX_SHARD_OVERLAP", "1") == "0":
                modes = ("sync", "pairwise")
            else:
                modes = ("overlap", "sync", "pairwise")
            nchk = min(args.n_local, 22) + k
            exchange_mode = exchange_selfcheck(lambda m: make(m, nchk), torch, dist, modes)
        reg = make(exchange_mode, n)
        reg.fill_random(1)

        def sweep():
            for q in range(n):
                reg.hadamard_gate(q)          # queued; executed on flush with look-ahead eviction

        norm_before = reg.norm2()
        configs = None
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
        for kind, pq, e0, e1, bits_ in prof:
            if kind != "h":
                continue
            ms = e0.elapsed_time(e1)
            per_q_ms[pq].append(ms * float(1 << (args.n_local - bits_)))      # scaled to a whole-shard launch
        norm = reg.norm2()
        exchanges = reg.exchanges - ex0
        sigma = reg.sigma
        overlapped = reg.overlap and sigma > 0
        fused = None
        del reg
        torch.cuda.empty_cache()

        # BASELINE config 4: n = 32 over 8 GPUs (29 local qubits; the same shard size at other N), H on every global
        # qubit vs a local one.  A global H = pack pass + all-to-all of (W-1)/W of the shard + the local gate.
        if args.gpus > 1 and not args.no_config4:
            nl4 = min(29, args.n_local)
            r4 = make(exchange_mode, nl4 + k)

            def timed(fn):
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                t1 = time.perf_counter(); fn(); torch.cuda.synchronize(); dist.barrier()
                tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item())
            res = {}
            r4.fill_random(1); r4.synchronize()
            timed(lambda: (r4.hadamard_gate(nl4 - 8), r4.synchronize()))
            res["local_h_ms"] = timed(lambda: (r4.hadamard_gate(nl4 - 8), r4.synchronize())) * 1e3
            for q in range(nl4 + k - 1, nl4 - 1, -1):
                r4.fill_random(1); r4.synchronize()
                res[f"global_h_q{q}_ms"] = timed(lambda q=q: (r4.hadamard_gate(q), r4.synchronize())) * 1e3
            shard_bytes = 16.0 * (1 << nl4)
            g = [v for kk, v in res.items() if kk.startswith("global")]
            config4 = {"n": nl4 + k, "shard_GiB": shard_bytes / 2**30, "results_ms": res,
                       "amplitude_updates_per_s_global_h": float(1 << (nl4 + k)) / (sum(g) / len(g) * 1e-3),
                       "note": "BASELINE config 4 (n=32 on 8 GPUs): a global H = pack pass + all-to-all of (W-1)/W of the shard "
                               "+ the local gate; exchange rate = sent bytes / (global - local time)"}
            if min(g) > res["local_h_ms"]:
                config4["exchange_GBps_per_gpu"] = shard_bytes * (world - 1) / world / ((min(g) - res["local_h_ms"]) * 1e-3) / 1e9
            del r4
            torch.cuda.empty_cache()
            # BASELINE config 5 in its N-GPU form: n = 30 Shor N = 21 (L = 25, M = 5) sharded over the ranks, + measurement
            # (full-size runs only: small --n-local rehearsals skip it)
            try:
                if args.n_local < 27:
                    raise RuntimeError("skipped: --n-local below 27 (rehearsal run)")
                kw5 = {"slices_log2": 0} if exchange_mode in ("sync", "pairwise") else {}
                if exchange_mode == "pairwise":
                    kw5["exchange"] = "pairwise"
                r5 = ShardedRegister(25, 5, fusion=True, **kw5)
                if exchange_mode in ("sync", "pairwise"):
                    r5.overlap = r5.async_exchange = False

                def shor():
                    r5.reset_register(); r5.quantum_computation(21, 2); r5.synchronize()
                shor()
                ex5 = r5.exchanges
                t5 = min(timed(shor) for _ in range(2)) * 1e3
                ex5 = (r5.exchanges - ex5) / 2
                nrm5 = r5.norm2()
                idx5 = r5.measure_state(0.37)
                w5 = sum(((idx5 >> (29 - p)) & 1) << p for p in range(25)) / float(1 << 25)
                config5 = {"workload": "n=30 Shor N=21 a=2 L=25 M=5 (375 gates) over %d GPUs, then measure_state" % world, "circuit_ms": t5,
                           "exchanges_per_circuit": ex5, "total_probability": nrm5, "measured_index": idx5, "omega": w5,
                           "nearest_multiple_of_one_sixth": min((abs(w5 - k6 / 6.0), k6) for k6 in range(7))[1]}
                del r5
            except Exception as e:      # reported, never fatal for the line
                config5 = {"error": f"{type(e).__name__}: {e}"}

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
        traffic, traffic_src = load_traffic() if args.n_local == 30 else (None, None)     # the PMC passes were taken at n = 30
        if not sharded:
            par = "1 GPU"
        else:
            par = (f"state sharded by top {k} qubits over {args.gpus} ranks, "
                   + ("pairwise half-shard swap (ncclSend/ncclRecv with rank ^ 2^j) per global target "
                      if exchange_mode == "pairwise" else "all-to-all qubit remap for global targets ") +
                   f"({exchanges} exchanges in the {args.steps} timed sweeps, "
                   + (f"exchange overlapped with the neighbouring gates on {1 << sigma} slices)" if overlapped else "synchronous unsliced exchange)"))
        out = {
            "metric": "amplitude-updates/s (gate*2^n/s), n=30 H-sweep" if args.gpus == 1 else
                      f"amplitude-updates/s (gate*2^n/s), n={n} H-sweep sharded over {args.gpus} GPUs",
            "value": args.steps * gates_per_step * dim / dt,
            "unit": "amplitude-updates/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"n={n} Hadamard sweep q=0..{n - 1} (config 2 of BASELINE.json at the headline size n=30"
                                   + (f", weak-scaled: 2^{args.n_local} amplitudes per GPU" if args.gpus > 1 else "") + "), "
                                   f"one launch per gate through libqcx.so, complex128 state of {16 * dim / 2**30:.0f} GiB in HBM",
                       "qubits": n, "gates_per_step": gates_per_step, "shard_qubits": args.n_local, "parallelism": par},
            "hbm_gbs_sweep_average": args.steps * gates_per_step * bytes_per_launch * args.gpus / dt / 1e9,
            "roofline": {"bound": "hbm", "kernel": "qcx::k_h_pair (Hadamard, pair form, target qubit >= 3)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                         "launches_timed": len(dom_ms),
                         "traffic": traffic,
                         "traffic_source": (f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; "
                                            "not re-measured in this run") if traffic_src else None},
            "per_qubit_gbs": per_q_gbs,
            "fused_sweep": fused,
            "configs": configs,
            "total_probability_before": norm_before,
            "total_probability_after": norm,
        }
        if sharded:
            out["exchange_mode"] = exchange_mode
            out["exchange_form"] = "pairwise" if exchange_mode == "pairwise" else "alltoall"
            out["config4"] = config4
            out["config5"] = config5
            out["c_host"] = c_host
        if not args.no_cpu_baseline and args.gpus == 1:
            cpu_n = args.cpu_n or (30 if mem_available_gib() >= 48 else 28)
            out["cpu_baseline"] = cpu_baseline(cpu_n)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if sharded:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


# ---- legs that run with a GPU in hand (called from main after its imports, or in the C-host child process) -----------
FP64_VECTOR_PEAK = 39.3e12     # v_mul/v_add_f64: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (MI355X_MICROARCH.md); an FMA counts once


def single_gpu_configs(qc, reps=3):
    """BASELINE.json configs 2, 3 and 5 (one-GPU form) through the C ABI, timed with HIP events on the register's stream
    (qcx_timer_start/stop); min over `reps`.  Algorithmic figures: a pass or a Hadamard moves 32 B per amplitude; a
    controlled phase performs 6 FP64 operations on a quarter of the amplitudes, a Hadamard 4 per amplitude."""
    out = {}

    def timed(reg, fn, nrep=None):
        best, passes = 1e30, 0
        for _ in range(nrep or reps):
            reg.synchronize()
            p0 = reg.fusion_stats()[0]
            reg.timer_start(); fn(); ms = reg.timer_stop()
            if ms < best:
                best, passes = ms, reg.fusion_stats()[0] - p0
        return best, passes

    def roofs(ms, passes, n, flops, launches_bytes=None):
        d = {"ms": ms}
        if passes:
            gbs = passes * 32.0 * 2.0 ** n / (ms * 1e-3) / 1e9
            d.update(hbm_passes=passes, hbm_gbs_per_pass=gbs, hbm_roof_frac=gbs / HBM_PEAK_GBS)
        elif launches_bytes:
            gbs = launches_bytes / (ms * 1e-3) / 1e9
            d.update(algorithmic_gbs=gbs, hbm_roof_frac=gbs / HBM_PEAK_GBS)
        if flops:
            d.update(fp64_ops=flops, fp64_ops_per_s=flops / (ms * 1e-3), fp64_vector_roof_frac=flops / (ms * 1e-3) / FP64_VECTOR_PEAK)
        return d

    # config 1 on the GPU (BASELINE config 1 is the reference's own CPU case, n = 12 Shor N = 15): whole period-finding
    # attempts -- reset, circuit (16 H + 8 C_AMODC + 28 CPHASE), measurement -- in a warm process, wall clock
    # (and, beyond BASELINE's config 1, the same at n = 20 and n = 24 with N = 21: the sizes at which an attempt is mostly host time --
    #  the plan of a repeated circuit is kept)
    for label, (L, M, Cn, a) in (("n7_C15_L3_M4", (3, 4, 15, 7)), ("n12_C15_L8_M4", (8, 4, 15, 7)),
                                 ("n20_C21_L15_M5", (15, 5, 21, 2)), ("n24_C21_L19_M5", (19, 5, 21, 2))):
        rng = qc.Rng(1)
        with qc.Register(L, M) as reg:
            def attempt():
                qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); return qc.measure_state(reg, rng)
            for _ in range(20):
                attempt()
            reg.synchronize(); t0 = time.perf_counter()
            for _ in range(200):
                attempt()
            us = (time.perf_counter() - t0) / 200 * 1e6
        out.setdefault("config1_on_gpu", {"workload": "one period-finding attempt (reset_register + quantum_computation + measure_state), "
                                                      "warm process, 200 attempts, wall clock"})[label + "_us_per_attempt"] = us

    # config 2: n = 26 Hadamard sweep, one launch per gate
    n = 26
    with qc.Register(n, 0) as reg:
        reg.fill_random(7)
        sweep = lambda: [qc.hadamard_gate(q, reg) for q in range(n)]
        sweep()
        ms, _ = timed(reg, sweep)
        c2 = roofs(ms, 0, n, 0, n * 32.0 * 2.0 ** n)
        c2.update(workload="n=26 Hadamard sweep q=0..25, one launch per gate", gates=n, amplitude_updates_per_s=n * 2.0 ** n / (ms * 1e-3))
        out["config2"] = c2

    # config 3: n = 28 inverse_QFT schedule over all qubits (28 H + 378 controlled phases)
    n = 28
    nh, nph = n, n * (n - 1) // 2
    flops = nph * 6.0 * 2.0 ** (n - 2) + nh * 4.0 * 2.0 ** n
    alg_bytes = nh * 32.0 * 2.0 ** n + nph * 32.0 * 2.0 ** (n - 2)
    c3 = {"workload": f"n=28 qcx_inverse_QFT: {nh} H + {nph} controlled phases (Q:678-690 with M = 0)", "gates": nh + nph}
    modes = [("fused_default", 0), ("per_gate", -1)]
    if hasattr(qc, "FUSION_TOLERANCE"):
        modes.append(("tolerance_mode", qc.FUSION_TOLERANCE))
    with qc.Register(n, 0) as reg:
        for label, mode in modes:
            reg.set_fusion(mode)
            reg.fill_random(7)
            qc.inverse_QFT(reg)
            ms, passes = timed(reg, lambda: qc.inverse_QFT(reg))
            d = roofs(ms, passes, n, flops if mode != getattr(qc, "FUSION_TOLERANCE", None) else 0, alg_bytes)
            if mode == getattr(qc, "FUSION_TOLERANCE", None):
                d["note"] = "opt-in qcx_set_fusion(reg, 2): runs of phases sharing a qubit merged into one diagonal; NOT bit-exact (|delta amplitude| <= 1e-12 in the tests)"
            d["amplitude_updates_per_s"] = (nh + nph) * 2.0 ** n / (ms * 1e-3)
            c3[label] = d
        reg.set_fusion(0)
    out["config3"] = c3

    # config 5 on one GPU: n = 30 Shor N = 21, a = 2, L = 25, M = 5: 50 H + 25 modular multiplies + 300 phases, then measure
    L, M, Cn, a = 25, 5, 21, 2
    n = L + M
    gates = 3 * L + L * (L - 1) // 2
    c5 = {"workload": f"n=30 qcx_quantum_computation({Cn}, {a}) L={L} M={M}: {2 * L} H + {L} C_AMODC + {L * (L - 1) // 2} controlled phases, then measure_state",
          "gates": gates}
    rng = qc.Rng(12345)
    with qc.Register(L, M) as reg:
        def circuit():
            # (flush: the state is in HBM in index order when the timed region ends -- a compact chain expands its result here)
            qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); reg.flush()
        modes5 = [("circuit_fused_default", 0), ("circuit_per_gate", -1)]
        if hasattr(qc, "FUSION_TOLERANCE"):
            modes5.insert(1, ("circuit_tolerance_mode", qc.FUSION_TOLERANCE))
        for label, mode in modes5:
            reg.set_fusion(mode)
            circuit()
            ms, passes = timed(reg, circuit, 1 if mode < 0 else None)     # (reset included)
            exact = mode != getattr(qc, "FUSION_TOLERANCE", None)
            d = roofs(ms, 0, n, ((L * (L - 1) // 2) * 6.0 * 2.0 ** (n - 2) + 2 * L * 4.0 * 2.0 ** n) if exact and mode < 0 else 0)
            if mode >= 0:
                d["passes"] = passes
            d["amplitude_updates_per_s"] = gates * 2.0 ** n / (ms * 1e-3)
            c5[label] = d
            if mode >= 0:
                # one period-finding attempt as the host driver runs it: reset, circuit, measurement, nothing in between
                def attempt():
                    qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); return qc.measure_state(reg, 0.37)
                attempt()
                best = 1e30
                for _ in range(reps):
                    reg.synchronize(); t0 = time.perf_counter(); attempt(); best = min(best, (time.perf_counter() - t0) * 1e3)
                d["attempt_ms"] = best
        c5["note"] = ("fused modes: the circuit front (reset + Hadamard layer + modular-multiply ladder on the basis state) is generated "
                      "inside the first pass, which -- like the passes behind it -- works on a COMPACT copy of the state (the M register "
                      "stays on the 6 residues of the ladder's orbit: [L register][orbit column], 4 GiB instead of 16); `ms` includes "
                      "expanding it into the register at the end (qcx_flush); `attempt_ms` = reset + circuit + measure_state with the "
                      "measurement reading the compact form (wall clock of the three calls)")
        reg.set_fusion(0)
        circuit()
        c5["total_probability"] = reg.norm2()
        qc.measure_state(reg, rng)        # warm-up: the first scan of a register of this size allocates its record arrays (0.4-0.5 ms)
        circuit()                         # (flushed: the state is in the register, expanded -- the timed call is the scan of 16 * 2^n bytes alone)
        reg.synchronize()
        t0 = time.perf_counter()
        idx = qc.measure_state(reg, rng)
        c5["measure_ms"] = (time.perf_counter() - t0) * 1e3           # wall clock of the call: the scan + its read-back; the collapse is lazy
        c5["measure_hbm_roof_frac"] = 16.0 * 2.0 ** n / (c5["measure_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS   # algorithmic: one read of the state
        c5["measured_index"] = idx
        c5["omega"] = qc.read_omega(idx, reg)
        c5["nearest_multiple_of_one_sixth"] = min((abs(c5["omega"] - k / 6.0), k) for k in range(7))[1]
    out["config5_one_gpu"] = c5

    # the per-gate kernels next to the headline one, at n = 30 (HIP events on the register's stream, best of 4), against the
    # algorithmic bytes of SURVEY s8(d)
    kern = {}
    with qc.Register(L, M) as reg:
        reg.set_fusion(-1)
        reg.fill_random(3)

        def best(fn, reps=4):
            fn(); reg.synchronize()
            b = 1e30
            for _ in range(reps):
                reg.timer_start(); fn(); b = min(b, reg.timer_stop())
            return b

        def row(name, ms, nbytes, what):
            gbs = nbytes / (ms * 1e-3) / 1e9
            kern[name] = {"ms": ms, "algorithmic_bytes": nbytes, "algorithmic_gbs": gbs, "hbm_roof_frac": gbs / HBM_PEAK_GBS, "bytes": what}
        row("c_amodc_gate(C=21, ctl=17), M=5 [k_camodc]", best(lambda: qc.c_amodc_gate(21, 4, 17, reg)), 32.0 * 2.0 ** (n - 1),
            "32 B x the control-set half (rows >= C are neither read nor written: the kernel moves less)")
        row("c_phase_shift_gate(29, 12) [k_phase]", best(lambda: qc.c_phase_shift_gate(29, 12, 0.3, reg)), 32.0 * 2.0 ** (n - 2),
            "32 B x the quarter with both bits set")
        row("c_phase_shift_gate(20, 1) [k_phase_lines]", best(lambda: qc.c_phase_shift_gate(20, 1, 0.3, reg)), 32.0 * 2.0 ** (n - 1),
            "32 B x the touched 128-B lines (bit 1 lies inside a line: half the state's lines)")
        row("hadamard_gate(1) [k_h_wave]", best(lambda: qc.hadamard_gate(1, reg)), 32.0 * 2.0 ** n, "32 B per amplitude")
        row("measure_state scan to the end, dense random state [k_meas_onepass + k_meas_fast]", best(lambda: reg.total_probability()), 16.0 * 2.0 ** n,
            "one read of the state (qcx_total_probability: the exact scan without a collapse); the synthetic dense state is the scan's "
            "worst case: its running sum crosses 32 binades, each costs an exact rescan of one record on one wave")
        # the same scan on the state the workload measures: the final state of the n = 30 Shor circuit (expanded into the register)
        reg.set_fusion(0); qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.flush(); reg.set_fusion(-1)
        row("measure_state scan to the end, Shor N=21 final state [k_meas_onepass + k_meas_fast]", best(lambda: reg.total_probability()), 16.0 * 2.0 ** n,
            "one read of the state; 10 of 524288 records take the exact rescan")
        reg.fill_random(3)
        reg.set_fusion(0)

        def front():
            qc.reset_register(reg)
            for l in range(M, n):
                qc.hadamard_gate(l, reg)
            reg.flush()
        reg.set_fusion(1)
        row("reset_register + Hadamard layer [k_basis_front]", best(front), 16.0 * 2.0 ** n, "one write of the state, nothing read")
    out["kernels_n30"] = kern
    return out


def c_host_child(args):
    """(child process of rank 0, N > 1) the one-process C-ABI sharded register over the N GPUs: sweep + config-4 shape"""
    import quantumcomputer_amd as qc
    W = args.gpus
    k = W.bit_length() - 1
    n = args.n_local + k
    out = {"host": "qcx_register_create_sharded: one process, peer stores over xGMI (k_pack_push), no RCCL", "shards": W}
    try:
        import ctypes as C
        nd = C.c_int(0)
        qc.lib().qcx_device_count(C.byref(nd))
        out["visible_gpus"] = nd.value
        out["devices"] = qc.spread_devices(W)
        with qc.Register(n, 0, shards=W) as reg:            # devices=None: spread + pre-flight exchange check
            out["selfchecks_passed"] = reg.selfchecks
            out["exchange_checked_bit_for_bit"] = reg.selfchecks > 0
            reg.set_fusion(-1)                               # one launch per gate per shard, like the headline
            reg.fill_random(1)
            sweep = lambda: [qc.hadamard_gate(q, reg) for q in range(n)]
            for _ in range(args.warmup):
                sweep()
            reg.synchronize()
            e0 = reg.sharded_stats()[0]
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sweep()
            reg.synchronize()
            dt = time.perf_counter() - t0
            out.update(n=n, ms_per_step=dt / args.steps * 1e3, value=args.steps * n * 2.0 ** n / dt, unit="amplitude-updates/s",
                       exchanges_per_sweep=(reg.sharded_stats()[0] - e0) / args.steps,
                       slices_log2=reg.overlap_stats()[0], total_probability_after=reg.norm2())
            reg.set_fusion(1)
            sweep(); reg.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sweep()
            reg.synchronize()
            tf = time.perf_counter() - t0
            out["fused_sweep"] = {"ms_per_step": tf / args.steps * 1e3, "value": args.steps * n * 2.0 ** n / tf}
        nl4 = min(29, args.n_local)
        with qc.Register(nl4 + k, 0, shards=W) as reg:
            reg.set_fusion(-1)
            res = {}

            def timed(fn):
                reg.synchronize(); t1 = time.perf_counter(); fn(); reg.synchronize()
                return (time.perf_counter() - t1) * 1e3
            reg.fill_random(1)
            timed(lambda: qc.hadamard_gate(nl4 - 8, reg))
            res["local_h_ms"] = timed(lambda: qc.hadamard_gate(nl4 - 8, reg))
            for q in range(nl4 + k - 1, nl4 - 1, -1):
                reg.fill_random(1)
                res[f"global_h_q{q}_ms"] = timed(lambda q=q: qc.hadamard_gate(q, reg))
            g = [v for kk, v in res.items() if kk.startswith("global")]
            shard_bytes = 16.0 * 2.0 ** nl4
            c4 = {"n": nl4 + k, "results_ms": res, "amplitude_updates_per_s_global_h": 2.0 ** (nl4 + k) / (sum(g) / len(g) * 1e-3)}
            if min(g) > res["local_h_ms"]:
                c4["exchange_GBps_per_gpu"] = shard_bytes * (W - 1) / W / ((min(g) - res["local_h_ms"]) * 1e-3) / 1e9
            out["config4"] = c4
        # BASELINE config 5 in its N-GPU form through the C host: n = 30 Shor N = 21 (L = 25, M = 5), + measurement
        if n >= 30:
            with qc.Register(25, 5, shards=W) as reg:
                def shor():
                    qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.synchronize()
                shor()
                e0 = reg.sharded_stats()[0]
                best = 1e9
                for _ in range(2):
                    t1 = time.perf_counter(); shor(); best = min(best, time.perf_counter() - t1)
                ex = (reg.sharded_stats()[0] - e0) / 2
                nrm = reg.norm2()
                idx = qc.measure_state(reg, 0.37)
                w = qc.read_omega(idx, reg)
                out["config5"] = {"workload": "n=30 Shor N=21 a=2 L=25 M=5 (375 gates) over %d shards, then measure_state" % W,
                                  "circuit_ms": best * 1e3, "exchanges_per_circuit": ex, "total_probability": nrm, "measured_index": idx,
                                  "omega": w, "nearest_multiple_of_one_sixth": min((abs(w - k6 / 6.0), k6) for k6 in range(7))[1]}
    except Exception as e:      # reported, never fatal for the line
        out["error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
