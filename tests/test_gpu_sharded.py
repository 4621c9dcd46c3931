"""GPU: the shard-level C ABI driven through ShardedRegister + HipEngine on one rank (RCCL backend
initialised with world_size 1), against the oracle.  The multi-rank exchange logic is covered on CPU
by tests/test_sharded_gloo.py; the 8-GPU run itself belongs to the driver's scaling bench."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_single_rank_sharded_register_matches_oracle(pg, ob):
    from quantumcomputer_amd.sharded import ShardedRegister
    L, M, Cn, a = 7, 5, 21, 2
    n = L + M
    reg = ShardedRegister(L, M)
    reg.reset_register()
    reg.quantum_computation(Cn, a)
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, Cn, a)
    assert np.array_equal(bits(reg.gather()), bits(want))
    assert abs(reg.norm2() - 1.0) < 1e-13
    rng = ob.Rng(4357)
    for _ in range(4):
        reg.reset_register(); reg.quantum_computation(Cn, a)
        r = rng.uniform()
        w = want.copy()
        assert reg.measure_state(r) == ob.measure(w, n, r)
        assert np.array_equal(bits(reg.gather()), bits(w))


def test_shard_entry_points_with_global_bits(pg, ob, qc):
    """what rank 5 of 8 would execute for gates whose control sits in the rank id: the phase mask loses
    the global bit, the modular multiply gets ctl = -1, fill_random starts at the shard's global offset"""
    import ctypes as C
    import torch
    from quantumcomputer_amd.sharded import HipEngine
    n, k, rank = 13, 3, 5
    nl = n - k
    eng = HipEngine("cuda:0")
    t = torch.empty(2 << nl, dtype=torch.float64, device="cuda:0")
    scale = math.sqrt(6.0 / (1 << n))
    eng.fill_random(t, nl, rank << nl, 9, scale)
    full = ob.fill_random(n, 9)
    mine = full[(rank << nl) * 2:((rank + 1) << nl) * 2]
    assert np.array_equal(bits(t.cpu().numpy()), bits(mine))
    # CPHASE(control = qubit 12 (rank bit 2, set on rank 5), target = 4)
    th = math.pi / 16
    eng.phase(t, nl, 1 << 4, *qc.polar(th))
    ob.cphase(full, n, 12, 4, th)
    # C_AMODC with control = qubit 10 (rank bit 0, set on rank 5), M = 5
    eng.camodc(t, nl, 5, 21, 16, -1)
    ob.camodc(full, n, 5, 21, 16, 10)
    eng.hadamard(t, nl, 7)
    ob.hadamard(full, n, 7)
    assert np.array_equal(bits(t.cpu().numpy()), bits(full[(rank << nl) * 2:((rank + 1) << nl) * 2]))


@pytest.mark.parametrize("nl,M,Cn", [(15, 5, 21), (17, 13, 8191)], ids=["M=5", "M=13 (multiplies stand-alone, staged in place)"])
def test_shard_run_fused_gate_list(pg, ob, qc, nl, M, Cn):
    """qcx_shard_run_fused: a gate list on a shard (rank-bit controls folded in: always-on multiply, reduced phase
    masks) through the fusion scheduler equals the same gates applied one by one by the oracle"""
    import torch
    from quantumcomputer_amd.sharded import HipEngine
    rs = np.random.RandomState(77)
    eng = HipEngine("cuda:0")
    for trial in range(4):
        want = ob.random_state(nl, 90 + trial)
        t = torch.from_numpy(want.copy()).to("cuda:0")
        descs = []
        for _ in range(70):
            k = rs.randint(0, 10)
            if k < 4:
                q = int(rs.randint(0, nl)); descs.append((0, q, 0, 0.0, 0.0, 0, 0)); ob.hadamard(want, nl, q)
            elif k < 8:
                nb = int(rs.randint(0, 3))
                bits_ = [int(b) for b in rs.choice(nl, nb, replace=False)]
                mask = sum(1 << b for b in bits_)
                c, s = qc.polar(float(rs.uniform(-4, 4)))
                descs.append((1, 0, mask, c, s, 0, 0))
                v = want.reshape(-1, 2); idx = np.arange(v.shape[0]); sel = (idx & mask) == mask
                re, im = v[sel, 0].copy(), v[sel, 1].copy()
                v[sel, 0] = 0.0 + ((c * re) - (s * im)); v[sel, 1] = 0.0 + ((c * im) + (s * re))
            else:
                A = int(rs.randint(0, Cn))
                if rs.randint(0, 2):
                    ctl = int(rs.randint(M, nl)); descs.append((2, ctl, 0, 0.0, 0.0, Cn, A)); ob.camodc(want, nl, M, Cn, A, ctl)
                else:                                   # control = a rank bit that is 1: every block moves
                    descs.append((2, 0xFFFFFFFF, 0, 0.0, 0.0, Cn, A))
                    ext = np.concatenate([np.zeros_like(want), want]); ob.camodc(ext, nl + 1, M, Cn, A, nl)
                    want[:] = ext[want.size:]
        eng.run_ops(t, nl, M, descs)
        got = t.cpu().numpy()
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), trial


def test_rccl_all_to_all_single_float64_async(pg):
    """the one collective the exchange uses, on RCCL as far as one GPU allows (world size 1): float64 device buffers,
    async_op + Work.wait() ordering against kernels on the current stream, equal and explicit split sizes"""
    import torch
    src = torch.arange(1 << 21, dtype=torch.float64, device="cuda:0")
    dst = torch.zeros_like(src)
    src.mul_(0.5)                                            # a kernel on the current stream right before the collective
    w = pg.all_to_all_single(dst, src, async_op=True)
    w.wait()
    dst.add_(1.0)                                            # ... and one right after the wait
    assert torch.equal(dst, torch.arange(1 << 21, dtype=torch.float64, device="cuda:0") * 0.5 + 1.0)
    dst2 = torch.zeros_like(src)
    pg.all_to_all_single(dst2, src, output_split_sizes=[src.numel()], input_split_sizes=[src.numel()])
    assert torch.equal(dst2, src)
