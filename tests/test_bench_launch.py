"""bench.py as the driver starts it: `python bench.py --gpus N` with no launcher around it must start its N ranks
itself (before anything touches the GPU), print ONE JSON line from rank 0 and pass the ranks' exit status on.

CPU: the launch path and the failure path (no GPU here: every rank must fail loudly, never fall back).
GPU: two ranks rehearsed on the one GPU of the box (gloo rendezvous, exchange staged through the host) -- the same
code path as the 2/4/8-GPU runs except for the collective's transport."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_fails_loudly_without_gpus():
    """no GPU in this container: the parent must have started the ranks (their message comes back), must not print a
    JSON line and must exit non-zero"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU half")
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--n-local", "12", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "2-rank run failed" in r.stderr
    assert "wants device" in r.stderr                       # the ranks really ran and refused: no CPU fallback
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_self_launch_happens_before_any_gpu_use():
    """the launcher must not import torch or load libqcx in the parent (an exec/fork after GPU init kills the box)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main()")]
    assert "import torch" not in head and "quantumcomputer_amd" not in head.split('"""', 2)[2].replace("quantumcomputer_amd/sharded.py", "")
    body = src[src.index("def main()"):]
    assert body.index("self_launch(") < body.index("import torch")


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_print_one_json_line():
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--n-local", "20", "--no-cpu-baseline"],
                  extra_env={"QCX_BENCH_BACKEND": "gloo", "QCX_FORCE_DEVICE": "0"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["qubits"] == 21 and out["config"]["shard_qubits"] == 20
    assert out["exchange_mode"] in ("overlap", "sync")
    assert abs(out["total_probability_after"] - 1.0) < 1e-2          # the synthetic state has norm 1 only in expectation
    c4 = out["config4"]
    assert c4["n"] == 21 and "global_h_q20_ms" in c4["results_ms"] and c4["results_ms"]["local_h_ms"] > 0
    assert out["roofline"]["frac"] > 0 and out["roofline"]["traffic"] is None      # PMC traffic exists for n_local = 30 only
    assert abs(out["total_probability_before"] - out["total_probability_after"]) < 1e-12     # conservation, shown on the line
    ch = out["c_host"]                                      # the one-process C host over the same GPUs, from a child of rank 0
    assert "error" not in ch, ch
    assert ch["shards"] == 2 and ch["n"] == 21 and ch["value"] > 0 and ch["exchanges_per_sweep"] >= 1
    assert ch["devices"] == [0, 0] or ch["exchange_checked_bit_for_bit"]         # several GPUs: the pre-flight check must have run
    assert "global_h_q20_ms" in ch["config4"]["results_ms"]


@pytest.mark.gpu
def test_sync_exchange_switch_is_reported():
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--n-local", "18", "--no-cpu-baseline", "--no-config4"],
                  extra_env={"QCX_BENCH_BACKEND": "gloo", "QCX_FORCE_DEVICE": "0", "QCX_SHARD_OVERLAP": "0"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert out["exchange_mode"] == "sync" and "synchronous" in out["config"]["parallelism"]


@pytest.mark.gpu
def test_pairwise_exchange_switch_is_reported():
    """QCX_SHARD_EXCHANGE=pairwise: the sweep runs through half-shard swaps with rank ^ 2^j (send/recv), self-checked first"""
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--n-local", "18", "--no-cpu-baseline", "--no-config4"],
                  extra_env={"QCX_BENCH_BACKEND": "gloo", "QCX_FORCE_DEVICE": "0", "QCX_SHARD_EXCHANGE": "pairwise"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert out["exchange_mode"] == "pairwise" and out["exchange_form"] == "pairwise" and "pairwise" in out["config"]["parallelism"]
    assert abs(out["total_probability_before"] - out["total_probability_after"]) < 1e-12
