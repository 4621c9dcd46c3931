"""CPU: the multi-rank host logic of quantumcomputer_amd/sharded.py (which rank skips a gate, how a
global target qubit is brought local by ONE all-to-all, how the measurement sum is handed from rank
to rank) run with world_size 2 and 4 over gloo.  The local shard arithmetic is injected as an
oracle-backed engine (tests may use the oracle as the checker; the product's only engine is HIP), and
the gathered result has to equal the unsharded oracle bit for bit."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """shard-level engine interface of sharded.HipEngine on CPU tensors, backed by oracle/"""

    def __init__(self):
        from oracle import binding as ob
        self.ob = ob

    def reset(self, t, n_local, holds_one):
        t.zero_()
        if holds_one:
            t[2] = 1.0

    def fill_random(self, t, n_local, first_global, seed, scale):
        import ctypes as C
        a = t.numpy()
        self.ob.lib().orc_fill_random(a.ctypes.data_as(C.POINTER(C.c_double)), first_global, 1 << n_local, seed, scale)

    def hadamard(self, t, n_local, q):
        self.ob.hadamard(t.numpy(), n_local, q)

    def phase(self, t, n_local, mask, c, s):
        v = t.numpy().reshape(-1, 2)
        idx = np.arange(v.shape[0], dtype=np.int64)
        sel = (idx & mask) == mask
        re, im = v[sel, 0].copy(), v[sel, 1].copy()
        v[sel, 0] = 0.0 + ((c * re) - (s * im))
        v[sel, 1] = 0.0 + ((c * im) + (s * re))

    def camodc(self, t, n_local, M, Cn, A, ctl_local):
        if ctl_local >= 0:
            self.ob.camodc(t.numpy(), n_local, M, Cn, A, ctl_local)
            return
        v = t.numpy().reshape(-1, 1 << M, 2)          # control is a rank bit that is 1: every block moves
        new = np.zeros_like(v)
        for f in range(1 << M):
            d = ((A * f) % Cn) & ((1 << M) - 1) if f < Cn else f
            new[:, d, :] += v[:, f, :]
        v[...] = new

    def run_ops(self, t, n_local, M, descs):
        """the fused gate-list entry point, emulated gate by gate (same semantics as qcx_shard_run_fused)"""
        for typ, q, mask, c, s, Cn, A in descs:
            if typ == 0:
                self.hadamard(t, n_local, q)
            elif typ == 1:
                self.phase(t, n_local, mask, c, s)
            else:
                self.camodc(t, n_local, M, Cn, A, -1 if q == 0xFFFFFFFF else q)

    def swap_bits(self, src, dst, n_local, pos_a, pos_b):
        j = np.arange(1 << n_local, dtype=np.int64)
        i = j.copy()
        for a, b in zip(pos_a, pos_b):                 # applied to the index in array order, like the kernel
            x = ((i >> a) ^ (i >> b)) & 1
            i ^= (x << a) | (x << b)
        dst.numpy().reshape(-1, 2)[...] = src.numpy().reshape(-1, 2)[i]

    def norm2(self, t, n_local):
        return self.ob.norm2(t.numpy(), n_local)

    def measure_scan(self, t, n_local, first_global, last_excluded, cum_in, r):
        return self.ob.measure_range(t.numpy(), first_global, 1 << n_local, last_excluded, cum_in, r)

    def collapse(self, t, n_local, local_index):
        t.zero_()
        if local_index >= 0:
            t[2 * local_index] = 1.0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, scenario, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        from quantumcomputer_amd.sharded import ShardedRegister
        out = scenario(rank, world, ob, lambda L, M, **kw: ShardedRegister(L, M, device="cpu", engine=OracleEngine(), **kw))
        if rank == 0:
            q.put(("ok", out))
    except Exception as e:      # pragma: no cover
        import traceback
        q.put(("err", f"rank {rank}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


def run(world, scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    status, out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", out
    return out


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# ---- scenarios (module level: spawn pickles them by name) ----------------------------------------
def sc_hadamard_sweep(rank, world, ob, make):
    n = 9
    reg = make(n, 0)
    reg.fill_random(5)
    want = ob.fill_random(n, 5)
    for rep in range(2):                       # second sweep starts in a permuted layout
        for qb in range(n):
            reg.hadamard_gate(qb)
            ob.hadamard(want, n, qb)
    reg.flush()
    exchanges = reg.exchanges              # before gather() restores the identity layout
    got = reg.gather()
    return bool(np.array_equal(bits(got), bits(want))), exchanges


def sc_shor(rank, world, ob, make):
    L, M, Cn, a = 6, 4, 15, 7
    n = L + M
    reg = make(L, M)
    reg.reset_register()
    reg.quantum_computation(Cn, a)
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, Cn, a)
    nrm = reg.norm2()
    got = reg.gather()
    same = bool(np.array_equal(bits(got), bits(want)))
    # seeded measurements: same index as the unsharded sequential scan, state collapses the same way
    rng = ob.Rng(12345)
    picks = []
    for _ in range(6):
        reg.reset_register(); reg.quantum_computation(Cn, a)
        r = rng.uniform()
        w = want.copy()
        picks.append((reg.measure_state(r), ob.measure(w, n, r)))
        same = same and bool(np.array_equal(bits(reg.gather()), bits(w)))
    return same, picks, nrm, reg.exchanges


def sc_shor_compact(rank, world, ob, make):
    """a register large enough for the companion register of compact circuits (ShardedRegister._try_compact; GPU engine only):
    Shor N = 21 with L = 12, M = 5 -- the orbit {1, 2, 4, 8, 11, 16} -> 8 columns"""
    L, M, Cn, a = 12, 5, 21, 2
    n = L + M
    reg = make(L, M)
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, Cn, a)
    same, picks = True, []
    rng = ob.Rng(777)
    for _ in range(3):
        reg.reset_register(); reg.quantum_computation(Cn, a)
        same = same and bool(np.array_equal(bits(reg.gather()), bits(want)))
        reg.reset_register(); reg.quantum_computation(Cn, a)
        r = rng.uniform()
        w = want.copy()
        picks.append((reg.measure_state(r), ob.measure(w, n, r)))
        same = same and bool(np.array_equal(bits(reg.gather()), bits(w)))
    return same, picks, reg.norm2(), (getattr(reg, "compact_circuits", 0), getattr(reg, "compact_measures", 0))


def sc_mixed_gates_in_swapped_layout(rank, world, ob, make):
    """controlled phases and modular multiplies whose control/target sit on global bits, issued
    while the qubit map is swapped (right after a global Hadamard)"""
    L, M, Cn = 6, 3, 7
    n = L + M
    reg = make(L, M)
    reg.fill_random(11)
    want = ob.fill_random(n, 11)
    top = n - 1
    prog = [("h", top), ("p", top, top - 1, 0.7), ("p", n - 4, top, math.pi / 8), ("c", 3, top), ("c", 5, top - 1),
            ("h", 4), ("p", top - 2, 0, 1.1), ("h", top - 1), ("c", 2, 4), ("p", top, 1, -2.0), ("h", top - 3), ("c", 6, n - 4)]
    for g in prog:
        if g[0] == "h":
            reg.hadamard_gate(g[1]); ob.hadamard(want, n, g[1])
        elif g[0] == "p":
            reg.c_phase_shift_gate(g[1], g[2], g[3]); ob.cphase(want, n, g[1], g[2], g[3])
        else:
            reg.c_amodc_gate(Cn, g[1], g[2]); ob.camodc(want, n, M, Cn, g[1], g[2])
    return bool(np.array_equal(bits(reg.gather()), bits(want))), reg.exchanges


def sc_measure_edges(rank, world, ob, make):
    n = 8
    reg = make(n, 0)
    res = []
    for r in (0.0, 0.3, 0.9999999, 1.0):
        reg.fill_random(21)
        want = ob.fill_random(n, 21)
        res.append((reg.measure_state(r), ob.measure(want, n, r)))
    reg.reset_register()                        # all-zero state: r beyond total probability -> last index
    reg.flush()                                 # (a reset is lazy on the HIP engine; the next line writes behind the register's back)
    reg.engine.collapse(reg.shard, reg.n_local, -1)
    res.append((reg.measure_state(0.5), (1 << n) - 1))
    return res


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hadamard_sweep(world):
    same, exchanges = run(world, sc_hadamard_sweep)
    assert same
    assert exchanges == 2          # look-ahead eviction: ONE all-to-all per sweep


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_shor_circuit_and_measurement(world):
    same, picks, nrm, exchanges = run(world, sc_shor)
    assert same
    assert all(g == w for g, w in picks), picks
    assert abs(nrm - 1.0) < 1e-13


def test_sharded_mixed_gates_world2():
    same, exchanges = run(2, sc_mixed_gates_in_swapped_layout)
    assert same and exchanges >= 2


def test_sharded_mixed_gates_world4():
    same, _ = run(4, sc_mixed_gates_in_swapped_layout)
    assert same


def test_sharded_measure_edges():
    res = run(2, sc_measure_edges)
    assert all(g == w for g, w in res), res


def sc_world8(rank, world, ob, make):
    """8 ranks (k = 3): two H sweeps, then a Shor circuit with seeded measurements"""
    n = 12
    reg = make(n, 0)
    reg.fill_random(31)
    want = ob.fill_random(n, 31)
    for rep in range(2):
        for qb in range(n):
            reg.hadamard_gate(qb); ob.hadamard(want, n, qb)
    reg.flush()
    ex = reg.exchanges
    ok = bool(np.array_equal(bits(reg.gather()), bits(want)))
    L, M, Cn, a = 9, 4, 15, 7
    reg2 = make(L, M)
    reg2.reset_register(); reg2.quantum_computation(Cn, a)
    w2 = np.zeros(2 << (L + M)); ob.reset(w2, L + M); ob.quantum_computation(w2, L + M, M, Cn, a)
    reg2.flush()
    ex2 = reg2.exchanges
    ok = ok and bool(np.array_equal(bits(reg2.gather()), bits(w2)))
    r = 0.6180339887
    ok = ok and reg2.measure_state(r) == ob.measure(w2, L + M, r)
    return ok, ex, ex2


def test_sharded_world8():
    ok, ex, ex2 = run(8, sc_world8)
    assert ok
    assert ex == 2             # one exchange per sweep
    assert ex2 <= 3            # whole Shor circuit: the top qubits are traded in once and out once


def sc_slice_counts(rank, world, ob, make):
    """the overlapped exchange with 1, 2, 4 and 8 slices, long and short queues: always the oracle's bits"""
    n = 11
    res = []
    for sl in (0, 1, 2, 3):
        for max_queue in (8192, 5):                 # 5: the queue is flushed mid-sweep (short look-ahead)
            reg = make(n, 0, slices_log2=sl, max_queue=max_queue)
            reg.fill_random(40 + sl)
            want = ob.fill_random(n, 40 + sl)
            for rep in range(2):
                for qb in list(range(n)) + [n - 1, 3, n - 2]:
                    reg.hadamard_gate(qb); ob.hadamard(want, n, qb)
                reg.c_phase_shift_gate(n - 1, 2, 0.3); ob.cphase(want, n, n - 1, 2, 0.3)
                reg.c_phase_shift_gate(n - 2, n - 3, 1.3); ob.cphase(want, n, n - 2, n - 3, 1.3)
            reg.flush()
            ov = reg.overlapped_gates
            res.append((sl, max_queue, reg.sigma, bool(np.array_equal(bits(reg.gather()), bits(want))), ov))
    return res


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_overlap_slice_counts(world):
    res = run(world, sc_slice_counts)
    assert all(r[3] for r in res), res
    assert any(r[2] >= 2 and r[4] > 10 for r in res)      # real overlap windows were exercised


def sc_random_programs(rank, world, ob, make):
    """seeded random programs (all gate kinds, random angles / controls / targets) on the sharded register"""
    res = []
    for seed in range(6):
        rs = np.random.RandomState(500 + seed)
        L, M = int(rs.randint(7, 10)), int(rs.randint(0, 4))
        n = L + M
        Cn = int(rs.randint(2, (1 << M) + 1)) if M else 2
        reg = make(L, M, slices_log2=int(rs.randint(0, 3)), max_queue=int(rs.choice([7, 8192])), fusion=bool(seed % 2))
        reg.fill_random(seed)
        want = ob.fill_random(n, seed)
        for _ in range(60):
            kind = rs.randint(0, 10)
            if kind < 4:
                qb = int(rs.randint(0, n)); reg.hadamard_gate(qb); ob.hadamard(want, n, qb)
            elif kind < 8 or M == 0:
                c, t = (int(x) for x in rs.choice(n, 2, replace=False))
                th = float(rs.uniform(-9, 9)) if rs.randint(0, 2) else math.pi / (1 << int(rs.randint(1, 20)))
                reg.c_phase_shift_gate(c, t, th); ob.cphase(want, n, c, t, th)
            else:
                atox, ctl = int(rs.randint(0, 1 << 16)), int(rs.randint(M, n))
                reg.c_amodc_gate(Cn, atox, ctl); ob.camodc(want, n, M, Cn, atox, ctl)
        same = bool(np.array_equal(bits(reg.gather()), bits(want)))
        r = float(rs.uniform(0, 1))
        w2 = want.copy()
        same = same and reg.measure_state(r) == ob.measure(w2, n, r)
        res.append((seed, same))
    return res


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_random_programs(world):
    res = run(world, sc_random_programs)
    assert all(ok for _, ok in res), res


# ---- the pairwise form of the exchange (north_star's "RCCL pairwise swaps"; QCX_SHARD_EXCHANGE=pairwise) --------------------
def sc_pairwise(rank, world, ob, make):
    """every scenario above once more with exchange="pairwise": ONE rank bit per exchange, half a shard to rank ^ 2^j
    (dist.batch_isend_irecv), relabel, nothing sent back -- H sweeps from permuted layouts, all gate kinds while the map is
    swapped, slices 1 ... 8 with and without short queues, a Shor circuit with seeded measurements; always the oracle's bits"""
    res = {}
    k = world.bit_length() - 1
    n = 9 + k
    reg = make(n, 0, exchange="pairwise")
    reg.fill_random(5)
    want = ob.fill_random(n, 5)
    for rep in range(2):
        for qb in range(n):
            reg.hadamard_gate(qb); ob.hadamard(want, n, qb)
    reg.flush()
    res["sweep_exchanges"], res["sweep_pairs"] = reg.exchanges, reg.pair_swaps
    res["sweep"] = bool(np.array_equal(bits(reg.gather()), bits(want)))
    res["identity_restored"] = reg.perm == list(range(n))
    # all gate kinds, controls / targets on rank bits, right after global Hadamards
    L, M, Cn = 6 + k, 3, 7
    n = L + M
    reg = make(L, M, exchange="pairwise")
    reg.fill_random(11)
    want = ob.fill_random(n, 11)
    top = n - 1
    prog = [("h", top), ("p", top, top - 1, 0.7), ("p", n - 4, top, math.pi / 8), ("c", 3, top), ("c", 5, top - 1),
            ("h", 4), ("p", top - 2, 0, 1.1), ("h", top - 1), ("c", 2, 4), ("p", top, 1, -2.0), ("h", top - 3), ("c", 6, n - 4),
            ("h", top), ("h", top - 1), ("h", top)]
    for g in prog:
        if g[0] == "h":
            reg.hadamard_gate(g[1]); ob.hadamard(want, n, g[1])
        elif g[0] == "p":
            reg.c_phase_shift_gate(g[1], g[2], g[3]); ob.cphase(want, n, g[1], g[2], g[3])
        else:
            reg.c_amodc_gate(Cn, g[1], g[2]); ob.camodc(want, n, M, Cn, g[1], g[2])
    res["mixed"] = bool(np.array_equal(bits(reg.gather()), bits(want)))
    # slices and queue lengths
    n = 11 + k
    ok, overlapped = True, 0
    for sl in (0, 1, 2, 3):
        for max_queue in (8192, 5):
            reg = make(n, 0, slices_log2=sl, max_queue=max_queue, exchange="pairwise")
            reg.fill_random(40 + sl)
            want = ob.fill_random(n, 40 + sl)
            for rep in range(2):
                for qb in list(range(n)) + [n - 1, 3, n - 2]:
                    reg.hadamard_gate(qb); ob.hadamard(want, n, qb)
                reg.c_phase_shift_gate(n - 1, 2, 0.3); ob.cphase(want, n, n - 1, 2, 0.3)
            reg.flush()
            overlapped += reg.overlapped_gates if reg.sigma >= 2 else 0
            ok = ok and bool(np.array_equal(bits(reg.gather()), bits(want)))
    res["slices"], res["overlapped"] = ok, overlapped
    # Shor circuit + seeded measurements
    L, M, Cn, a = 6 + k, 4, 15, 7
    n = L + M
    reg = make(L, M, exchange="pairwise")
    reg.reset_register(); reg.quantum_computation(Cn, a)
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, Cn, a)
    res["shor_norm"] = reg.norm2()
    res["shor"] = bool(np.array_equal(bits(reg.gather()), bits(want)))
    rng = ob.Rng(12345)
    picks = []
    for _ in range(4):
        reg.reset_register(); reg.quantum_computation(Cn, a)
        r = rng.uniform()
        w = want.copy()
        picks.append((reg.measure_state(r), ob.measure(w, n, r)))
        res["shor"] = res["shor"] and bool(np.array_equal(bits(reg.gather()), bits(w)))
    res["picks"] = picks
    return res


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_pairwise_exchange(world):
    res = run(world, sc_pairwise)
    k = world.bit_length() - 1
    assert res["sweep"] and res["identity_restored"] and res["mixed"] and res["slices"] and res["shor"], res
    assert all(g == w for g, w in res["picks"]), res["picks"]
    assert abs(res["shor_norm"] - 1.0) < 1e-13
    # one half-shard swap per global target per sweep -- k per sweep, against ONE all-to-all (of (W-1)/W of the shard)
    assert res["sweep_pairs"] == res["sweep_exchanges"] == 2 * k, res
    assert res["overlapped"] > 10                       # the sliced pipeline ran with the pairwise moves too


def sc_pairwise_random(rank, world, ob, make):
    res = []
    for seed in range(6):
        rs = np.random.RandomState(900 + seed)
        L, M = int(rs.randint(8, 11)) + (world.bit_length() - 1), int(rs.randint(0, 4))
        n = L + M
        Cn = int(rs.randint(2, (1 << M) + 1)) if M else 2
        reg = make(L, M, slices_log2=int(rs.randint(0, 3)), max_queue=int(rs.choice([7, 8192])), fusion=bool(seed % 2), exchange="pairwise")
        reg.fill_random(seed)
        want = ob.fill_random(n, seed)
        for _ in range(70):
            kind = rs.randint(0, 10)
            if kind < 5:
                qb = int(rs.randint(0, n)) if rs.randint(0, 2) else int(rs.randint(n - 3, n))
                reg.hadamard_gate(qb); ob.hadamard(want, n, qb)
            elif kind < 8 or M == 0:
                c, t = (int(x) for x in rs.choice(n, 2, replace=False))
                th = float(rs.uniform(-9, 9))
                reg.c_phase_shift_gate(c, t, th); ob.cphase(want, n, c, t, th)
            else:
                atox, ctl = int(rs.randint(0, 1 << 16)), int(rs.randint(M, n))
                reg.c_amodc_gate(Cn, atox, ctl); ob.camodc(want, n, M, Cn, atox, ctl)
        same = bool(np.array_equal(bits(reg.gather()), bits(want)))
        r = float(rs.uniform(0, 1))
        w2 = want.copy()
        same = same and reg.measure_state(r) == ob.measure(w2, n, r)
        res.append((seed, same, reg.pair_swaps))
    return res


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_pairwise_random_programs(world):
    res = run(world, sc_pairwise_random)
    assert all(ok for _, ok, _ in res), res
    assert sum(p for _, _, p in res) > 0
