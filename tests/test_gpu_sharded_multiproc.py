"""GPU: the multi-PROCESS sharded path -- 2 and 4 ranks, each its own process with its own HIP context and libqcx.
  * box with at least `world` GPUs: rank r on cuda:r, backend nccl (= RCCL over xGMI), device buffers straight into
    all_to_all_single -- the production path of bench.py --gpus N, overlapped sliced exchange included;
  * one-GPU box: all ranks on cuda:0, rendezvous over gloo, the exchange staged through the host (gloo has no device
    all-to-all) -- a rehearsal of the same host logic.
Which of the two ran is decided by torch.cuda.device_count() alone (QCX_TEST_BACKEND=gloo forces the rehearsal), so the
first multi-GPU box that runs `pytest -m gpu` exercises RCCL with more than one rank without anyone editing a test.
Same scenarios as the CPU gloo suite (tests/test_sharded_gloo.py), with the HIP kernels doing the work."""
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import test_sharded_gloo as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_worker(rank, world, port, scenario, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    use_nccl = pick_backend(world, torch.cuda.device_count()) == "nccl"
    dev = rank if use_nccl else 0
    torch.cuda.set_device(dev)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if use_nccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import binding as ob
        from quantumcomputer_amd.sharded import HipEngine, ShardedRegister

        def make(L, M, **kw):
            reg = ShardedRegister(L, M, **kw)
            assert isinstance(reg.engine, HipEngine) and reg.shard.is_cuda and reg.shard.device.index == dev
            assert reg._host_staged == (not use_nccl)
            return reg
        out = scenario(rank, world, ob, make)
        if rank == 0:
            q.put(("ok", out))
    except Exception as e:      # pragma: no cover
        import traceback
        q.put(("err", f"rank {rank}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


def pick_backend(world, visible_gpus):
    """nccl with one GPU per rank when the box has them, else the one-GPU gloo rehearsal"""
    if os.environ.get("QCX_TEST_BACKEND", "") == "gloo":
        return "gloo"
    return "nccl" if visible_gpus >= world else "gloo"


def run_gpu(world, scenario):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = G._free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        status, out = q.get(timeout=300)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
    assert status == "ok", out
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_hadamard_sweeps_across_processes(world):
    same, exchanges = run_gpu(world, G.sc_hadamard_sweep)
    assert same and exchanges >= 1


@pytest.mark.parametrize("world", [2, 4])
def test_shor_circuit_and_measurement_across_processes(world):
    same, picks, nrm, exchanges = run_gpu(world, G.sc_shor)
    assert same and abs(nrm - 1.0) < 1e-13
    assert all(a == b for a, b in picks), picks


@pytest.mark.parametrize("world", [2, 4])
def test_compact_circuits_across_processes(world):
    """the per-rank host runs the queue behind a circuit front on its companion register ([L register][orbit column]: fused
    passes and all-to-alls 2^(M - cb) times smaller) and every rank expands its part: whole states and measured indices
    against the oracle"""
    same, picks, nrm, compact = run_gpu(world, G.sc_shor_compact)
    assert same and all(a == b for a, b in picks), picks
    # six circuits ran on the companion; the three that were measured right behind the circuit were scanned THERE (round 5: the
    # register is not expanded for a measurement), the three that were gathered first were expanded
    assert compact == (6, 3), compact


@pytest.mark.parametrize("world", [2, 4])
def test_mixed_gates_in_swapped_layout_across_processes(world):
    same, exchanges = run_gpu(world, G.sc_mixed_gates_in_swapped_layout)
    assert same and exchanges >= 1


def test_measurement_edges_across_processes():
    assert all(a == b for a, b in run_gpu(2, G.sc_measure_edges))


@pytest.mark.parametrize("world", [2, 4])
def test_sliced_overlapped_exchange_across_processes(world):
    res = run_gpu(world, G.sc_slice_counts)
    assert all(r[3] for r in res), res
    assert any(r[2] >= 2 and r[4] > 10 for r in res)


@pytest.mark.parametrize("world", [2, 4])
def test_random_programs_across_processes(world):
    """all gate kinds, random slices / queue lengths, every second program through the fused-pass scheduler"""
    res = run_gpu(world, G.sc_random_programs)
    assert all(ok for _, ok in res), res
