"""GPU: a C program written in the reference's own style (gate calls with the matrix argument, Register by
value, INT_POW, gsl_rng) built against include/qcx_compat.h + libqcx.so reproduces the oracle bit for bit."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path_factory, name):
    out = str(tmp_path_factory.mktemp("c") / name)
    lib = os.path.join(ROOT, "quantumcomputer_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", name + ".c"), "-L", lib, "-lqcx", "-lm",
                    "-Wl,-rpath," + lib, "-o", out], check=True)
    return out


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return _build(tmp_path_factory, "refstyle_circuit")


@pytest.fixture(scope="module")
def exe_main(tmp_path_factory):
    return _build(tmp_path_factory, "refstyle_main")


@pytest.mark.parametrize("shards", [None, "4"])
@pytest.mark.parametrize("C,L,M,a", [(15, 3, 4, 7), (21, 9, 5, 2), (33, 5, 5, 7)])
def test_main_shaped_like_the_reference(exe_main, ob, C, L, M, a, shards):
    """tests/c/refstyle_main.c: the reference's main (qc_shor.c:1284-1347) call for call -- gsl_rng_alloc(gsl_rng_mt19937),
    gsl_vector_complex_alloc, gsl_spmatrix_complex_alloc_nzmax, operate_matrix, the GSL frees -- with no qcx_* call in
    it; unsharded and (QCX_SHARDS) sharded over the visible GPUs"""
    n = L + M
    if shards and n - 2 - max(M, 6) < 4:
        pytest.skip("register too small for 4 shards")
    env = dict(os.environ)
    env.pop("QCX_COMPAT_FUSION", None); env.pop("QCX_SHARD_DEVICES", None)
    if shards:
        env["QCX_SHARDS"] = shards
    r = subprocess.run([exe_main, str(C), str(L), str(M), str(a), "777"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split()
    got = np.array([int(x, 16) for x in lines[:2 << n]], dtype=np.uint64)
    want = np.zeros(2 << n); ob.reset(want, n)
    ob.quantum_computation(want, n, M, C, a, ref_intpow=True)
    assert np.array_equal(got, want.view(np.uint64))
    rng = ob.Rng(777)
    rng.uniform()                                                     # the program spends one draw before measuring
    assert int(lines[-1]) == ob.measure(want, n, rng.uniform())


@pytest.mark.parametrize("fusion", [None, "-1", "0"], ids=["queued (compat default)", "one launch per gate", "mode 0"])
@pytest.mark.parametrize("C,L,M,a", [(15, 3, 4, 7), (21, 5, 5, 2), (15, 8, 4, 7), (33, 5, 5, 7), (21, 9, 5, 2)])
def test_reference_style_c_program(exe, ob, C, L, M, a, fusion):
    n = L + M
    env = dict(os.environ)
    env.pop("QCX_COMPAT_FUSION", None)
    if fusion is not None:
        env["QCX_COMPAT_FUSION"] = fusion
    r = subprocess.run([exe, str(C), str(L), str(M), str(a), "12345"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split()
    got = np.array([int(x, 16) for x in lines[:2 << n]], dtype=np.uint64)
    want = np.zeros(2 << n); ob.reset(want, n)
    ob.quantum_computation(want, n, M, C, a, ref_intpow=True)        # the C program uses the reference's INT_POW
    assert np.array_equal(got, want.view(np.uint64))
    rng = ob.Rng(12345)
    assert int(lines[-1]) == ob.measure(want, n, rng.uniform())


def test_debug_helpers_of_the_reference(exe):
    """display_state / check_normalisation (testing_and_debug.c:7-37) through the compat header, same output format"""
    C, L, M, a = 21, 5, 5, 2
    n = L + M
    r = subprocess.run([exe, str(C), str(L), str(M), str(a), "12345", "debug"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0].startswith("Total Probability: ") and abs(float(out[0].split(": ")[1]) - 1.0) < 1e-13
    idx = int(out[1].split()[1])
    assert out[2] == "|" + format(idx, f"0{n}b") + "> 1.00"           # collapsed: exactly one basis state, amplitude 1
    assert out[3] == "Total Probability: 1.0000000000000000" and len(out) == 4
