"""CPU: where the sharded hosts and their GPU tests put shards, relays and ranks for a given number of visible GPUs.  The GPU
suites (tests/test_gpu_sharded_c.py, test_gpu_sharded_virtual.py, test_gpu_sharded_multiproc.py) carry no device
numbers of their own: they ask these helpers, so a one-GPU box gets today's virtual shards on device 0 and the first
multi-GPU box gets real peer stores / RCCL ranks from the very same tests."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shards", [1, 2, 4, 8, 16])
def test_one_visible_gpu_degrades_to_device_zero(qc, shards):
    assert qc.spread_devices(shards, visible=1) == [0] * shards


def test_spreading_over_several_gpus(qc):
    assert qc.spread_devices(8, visible=8) == list(range(8))                # config 4 / 5 of BASELINE.json
    assert qc.spread_devices(8, visible=16) == list(range(8))
    assert qc.spread_devices(8, visible=4) == [0, 0, 1, 1, 2, 2, 3, 3]      # neighbours share a GPU
    assert qc.spread_devices(8, visible=6) == [0, 0, 1, 1, 2, 2, 3, 3]      # a power-of-two number of GPUs is used
    assert qc.spread_devices(2, visible=8) == [0, 1]
    assert qc.spread_devices(16, visible=8) == [r // 2 for r in range(16)]
    assert qc.spread_devices(4, visible=2) == [0, 0, 1, 1]
    for bad in (0, 3, 32):
        with pytest.raises(qc.QcxError):
            qc.spread_devices(bad, visible=8)


def test_relays_are_idle_gpus_first(qc):
    assert qc.idle_devices([0, 1], 6, visible=8) == [2, 3, 4, 5, 6, 7]     # W = 2 with 6 relays: DESIGN s5's 3.5x case
    assert qc.idle_devices([0, 1, 2, 3], 4, visible=8) == [4, 5, 6, 7]
    assert qc.idle_devices([0, 0], 2, visible=1) == [0, 0]                  # one GPU: the shard device again
    assert qc.idle_devices(list(range(8)), 2, visible=8) == [0, 1]          # nothing idle: reuse shard devices
    assert qc.idle_devices([0, 1], 3, visible=3) == [2, 0, 1]


def test_multiprocess_suite_picks_rccl_when_every_rank_gets_a_gpu(monkeypatch):
    import test_gpu_sharded_multiproc as T
    monkeypatch.delenv("QCX_TEST_BACKEND", raising=False)
    assert T.pick_backend(2, 1) == "gloo" and T.pick_backend(4, 1) == "gloo"
    assert T.pick_backend(2, 2) == "nccl" and T.pick_backend(4, 8) == "nccl" and T.pick_backend(4, 2) == "gloo"
    monkeypatch.setenv("QCX_TEST_BACKEND", "gloo")
    assert T.pick_backend(2, 8) == "gloo"


def test_no_gpu_means_no_silent_placement(qc):
    """asking HIP for the device count without a GPU fails loudly (this container has none)"""
    import ctypes as C
    n = C.c_int(0)
    if qc.lib().qcx_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(qc.QcxError):
        qc.spread_devices(4)
