"""GPU: the sharded register's sliced exchange (views, pack kernel, per-slice gate launches, measurement
hand-off) on real HIP kernels with 2 and 4 VIRTUAL ranks: threads of one process (rank r on GPU r mod visible GPUs:
all on the one GPU of a one-GPU box), with torch.distributed replaced by an in-process stand-in that performs the same
data movement (chunk c of rank r <-> chunk r of rank c).  RCCL itself is not involved here -- with one GPU per rank
tests/test_gpu_sharded_multiproc.py runs the same scenarios over nccl."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Work:
    def wait(self):
        return True


class FakeDist:
    """the subset of torch.distributed that quantumcomputer_amd.sharded uses, for W threads"""

    class ReduceOp:
        SUM = "sum"
        MIN = "min"
        MAX = "max"

    class P2POp:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    @staticmethod
    def isend(*a, **k): raise AssertionError("P2P ops go through batch_isend_irecv")
    @staticmethod
    def irecv(*a, **k): raise AssertionError("P2P ops go through batch_isend_irecv")

    def __init__(self, world):
        import torch
        self.torch = torch
        self.world = world
        self.tls = threading.local()
        self.barrier_obj = threading.Barrier(world)
        self.slots = [None] * world

    def get_world_size(self, group=None): return self.world
    def get_rank(self, group=None): return self.tls.rank
    def get_global_rank(self, group, r): return r

    def _sync_all(self):
        for d in range(max(1, self.torch.cuda.device_count())):          # ranks may sit on different GPUs (multi-GPU box)
            self.torch.cuda.synchronize(d)

    def _exchange(self, payload):
        self._sync_all()
        self.slots[self.tls.rank] = payload
        self.barrier_obj.wait()
        got = list(self.slots)
        self.barrier_obj.wait()
        return got

    def all_to_all_single(self, dst, src, group=None, async_op=False):
        W, r = self.world, self.tls.rank
        allsrc = self._exchange(src)
        n = src.numel() // W
        for c in range(W):
            dst[c * n:(c + 1) * n].copy_(allsrc[c][r * n:(r + 1) * n])
        self._sync_all()
        self.barrier_obj.wait()                      # nobody overwrites a source that is still being read
        return _Work()

    def all_reduce(self, t, op=None, group=None):
        vals = [v.to(t.device) for v in self._exchange(t.clone())]
        if op == "min":
            t.copy_(self.torch.stack(vals).amin(dim=0))
        elif op == "max":
            t.copy_(self.torch.stack(vals).amax(dim=0))
        else:
            t.copy_(sum(vals))

    def batch_isend_irecv(self, ops):
        """the pairwise form of the exchange: every rank posts one send and one receive with its partner"""
        sends = {o.peer: o.tensor for o in ops if o.op is FakeDist.isend}
        allsends = self._exchange(sends)
        r = self.tls.rank
        for o in ops:
            if o.op is FakeDist.irecv:
                o.tensor.copy_(allsends[o.peer][r])
        self._sync_all()
        self.barrier_obj.wait()                      # nobody overwrites a source that is still being read
        return [_Work()]

    def broadcast(self, t, src, group=None):
        vals = self._exchange(t.clone())
        t.copy_(vals[src])

    def all_gather(self, parts, t, group=None):
        vals = self._exchange(t.clone())
        for p, v in zip(parts, vals):
            p.copy_(v)


def run_virtual(world, body):
    import torch
    from quantumcomputer_amd import sharded
    fake = FakeDist(world)
    torch.cuda.init()
    ndev = max(1, torch.cuda.device_count())
    real = sharded.dist
    sharded.dist = fake
    out, err = [None] * world, []

    def worker(rank):
        try:
            torch.cuda.set_device(rank % ndev)      # spread over the visible GPUs (one GPU: all on 0)
            fake.tls.rank = rank
            out[rank] = body(rank, sharded.ShardedRegister)
        except Exception as e:      # pragma: no cover
            import traceback
            err.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
            fake.barrier_obj.abort()

    try:
        th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join(timeout=300) for t in th]
    finally:
        sharded.dist = real
    assert not err, err[0]
    return out


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("world,slices", [(2, 2), (4, 2), (2, 0), (4, 3)])
def test_virtual_ranks_sweeps_and_mixed_gates(ob, world, slices):
    n = 16

    def body(rank, SR):
        reg = SR(n, 0, slices_log2=slices)
        reg.fill_random(3)
        for rep in range(2):
            for q in range(n):
                reg.hadamard_gate(q)
            reg.c_phase_shift_gate(n - 1, 1, 0.7)
            reg.c_phase_shift_gate(n - 2, n - 5, -0.2)
        reg.flush()
        stats = (reg.exchanges, reg.overlapped_gates, reg.sigma)
        return reg.gather(), stats

    outs = run_virtual(world, body)
    want = ob.fill_random(n, 3)
    for rep in range(2):
        for q in range(n):
            ob.hadamard(want, n, q, 8)
        ob.cphase(want, n, n - 1, 1, 0.7, 8); ob.cphase(want, n, n - 2, n - 5, -0.2, 8)
    for got, stats in outs:
        assert np.array_equal(bits(got), bits(want))
    assert outs[0][1][0] == 2                       # one exchange per sweep
    if slices:
        assert outs[0][1][1] > 10


@pytest.mark.parametrize("world,fusion", [(2, False), (4, False), (2, True), (4, True), (2, 2), (4, 2)])
def test_virtual_ranks_shor_and_measurement(ob, world, fusion):
    L, M, Cn, a = 13, 5, 21, 2
    n = L + M
    r = 0.4142135623

    def body(rank, SR):
        reg = SR(L, M, fusion=fusion)
        reg.reset_register()
        reg.quantum_computation(Cn, a)
        state = reg.gather()
        nrm = reg.norm2()
        reg.reset_register(); reg.quantum_computation(Cn, a)
        idx = reg.measure_state(r)
        return state, nrm, idx, reg.gather(), reg.exchanges, reg.fronts

    outs = run_virtual(world, body)
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, Cn, a, threads=8)
    w2 = want.copy(); widx = ob.measure(w2, n, r)
    for state, nrm, idx, collapsed, ex, fronts in outs:
        assert fronts == (2 if fusion else 0)           # reset + Hadamard layer + multiply ladder: one write pass per rank, no exchange
        if fusion == 2:                                 # tolerance mode per shard: rounding-level differences only
            d = np.asarray(state) - want
            assert float(np.max(np.hypot(d[0::2], d[1::2]))) <= 1e-12 and abs(nrm - 1.0) < 1e-12 and idx == widx
            continue
        assert np.array_equal(bits(state), bits(want))
        assert abs(nrm - 1.0) < 1e-12
        assert idx == widx
        assert np.array_equal(bits(collapsed), bits(w2))


@pytest.mark.parametrize("world,slices,fusion", [(2, 0, False), (4, 2, True), (8, 1, True)])
def test_virtual_ranks_pairwise_exchange(ob, world, slices, fusion):
    """exchange="pairwise" (one rank bit per exchange, half a shard to rank ^ 2^j) on the HIP engine: sweeps from permuted
    layouts, phases on rank bits, and a Shor circuit with measurement -- the oracle's bits"""
    n = 17
    L, M, Cn, a = 13, 5, 21, 2
    r = 0.4142135623

    def body(rank, SR):
        reg = SR(n, 0, slices_log2=slices, fusion=fusion, exchange="pairwise")
        reg.fill_random(3)
        for rep in range(2):
            for q in range(n):
                reg.hadamard_gate(q)
            reg.c_phase_shift_gate(n - 1, 1, 0.7)
            reg.c_phase_shift_gate(n - 2, n - 5, -0.2)
        reg.flush()
        stats = (reg.exchanges, reg.pair_swaps)
        state = reg.gather()
        sh = SR(L, M, fusion=fusion, exchange="pairwise")
        sh.reset_register(); sh.quantum_computation(Cn, a)
        final = sh.gather()
        sh.reset_register(); sh.quantum_computation(Cn, a)
        return state, stats, final, sh.measure_state(r)

    outs = run_virtual(world, body)
    want = ob.fill_random(n, 3)
    for rep in range(2):
        for q in range(n):
            ob.hadamard(want, n, q, 8)
        ob.cphase(want, n, n - 1, 1, 0.7, 8); ob.cphase(want, n, n - 2, n - 5, -0.2, 8)
    w2 = np.zeros(2 << (L + M)); ob.reset(w2, L + M); ob.quantum_computation(w2, L + M, M, Cn, a, threads=8)
    k = world.bit_length() - 1
    for state, stats, final, idx in outs:
        assert np.array_equal(bits(state), bits(want))
        assert stats[0] == stats[1] == 2 * k             # one half-shard swap per global target per sweep
        assert np.array_equal(bits(final), bits(w2))
        assert idx == ob.measure(w2.copy(), L + M, r)
