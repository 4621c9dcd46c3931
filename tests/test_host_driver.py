"""host/: the C driver around the hot path.  CPU tests: classical helpers against the oracle's
restatement of the reference and CLI argument handling (exit codes of qc_shor.c:164-170).  GPU
tests: the reference's documented runs factor through the MI355X engine."""
import ctypes as C
import os
import subprocess

import numpy as np

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "host")


@pytest.fixture(scope="module")
def hostlib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "quantumcomputer_amd", "csrc"), "-s"], check=True)
    subprocess.run(["make", "-C", HOST, "-s"], check=True)
    L = C.CDLL(os.path.join(HOST, "libqcx_classical.so"))
    L.qcx_gcd.argtypes = [C.c_uint, C.c_uint]; L.qcx_gcd.restype = C.c_uint
    L.qcx_modpow.argtypes = [C.c_ulonglong] * 3; L.qcx_modpow.restype = C.c_ulonglong
    L.qcx_cf_denominators.argtypes = [C.c_double, C.c_uint, C.POINTER(C.c_uint)]
    L.qcx_read_omega.argtypes = [C.c_ulong, C.c_int, C.c_int]; L.qcx_read_omega.restype = C.c_double
    L.qcx_period_from_omega.argtypes = [C.c_double, C.c_uint, C.c_uint, C.c_int]; L.qcx_period_from_omega.restype = C.c_uint
    L.qcx_factors_from_period.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_uint)]
    return L


def test_classical_helpers_match_oracle(hostlib, ob):
    import random
    rnd = random.Random(3)
    for _ in range(300):
        a, b = rnd.randrange(0, 5000), rnd.randrange(0, 5000)
        assert hostlib.qcx_gcd(a, b) == ob.gcd(a, b)
        m = rnd.randrange(2, 10 ** 9)
        assert hostlib.qcx_modpow(a, b, m) == pow(a, b, m) == ob.modpow(a, b, m)
    for L, M in ((3, 4), (5, 5), (9, 5), (25, 5)):
        for _ in range(50):
            st = rnd.randrange(0, 1 << (L + M))
            assert hostlib.qcx_read_omega(st, L, M) == ob.read_omega(st, L, M)
    for x, L in [(1, 3), (2, 3), (3, 3), (5, 4), (21, 5), (27, 5), (341, 9), (85, 9), (171, 9), (11184811, 25)]:
        out = (C.c_uint * 15)()
        hostlib.qcx_cf_denominators(x / float(1 << L), 15, out)
        assert list(out) == ob.cf_denominators(x / float(1 << L))
    # a golden-ratio-like input: every coefficient 1 -> Fibonacci denominators
    out = (C.c_uint * 8)()
    hostlib.qcx_cf_denominators(0.6180339887498949, 8, out)
    assert list(out) == [1, 1, 2, 3, 5, 8, 13, 21]


def test_period_and_factors_for_the_documented_runs(hostlib):
    f = (C.c_uint * 2)()
    # C=15 a=7: x~ in {2, 4(->1/2), 6} of 8
    assert hostlib.qcx_period_from_omega(2 / 8.0, 7, 15, 0) == 4
    assert hostlib.qcx_period_from_omega(6 / 8.0, 7, 15, 0) == 4
    assert hostlib.qcx_period_from_omega(4 / 8.0, 7, 15, 0) == 4        # denominator 2, multiple m = 2
    assert hostlib.qcx_period_from_omega(0.0, 7, 15, 0) == 0
    assert hostlib.qcx_factors_from_period(7, 4, 15, 0, f) == 0 and list(f) == [5, 3]
    assert hostlib.qcx_factors_from_period(7, 10, 33, 0, f) == 0 and list(f) == [11, 3]
    assert hostlib.qcx_factors_from_period(2, 6, 21, 0, f) == 0 and list(f) == [3, 7]
    assert hostlib.qcx_factors_from_period(2, 3, 7, 0, f) == 1                                  # odd period
    assert hostlib.qcx_factors_from_period(2, 4, 5, 0, f) == 2                                  # 2^2 = -1 mod 5
    # reference quirk mode: 32-bit INT_POW wraps, so a large valid period is no longer recognised
    assert hostlib.qcx_period_from_omega(1 / 64.0, 3, 193, 0) == 64
    assert hostlib.qcx_period_from_omega(1 / 64.0, 3, 193, 1) != 64


def run_cli(*args):
    return subprocess.run([os.path.join(HOST, "qcx_shor"), *args], capture_output=True, text=True, timeout=300)


def test_cli_argument_errors(hostlib):
    r = run_cli("-L", "3", "-M", "4")
    assert r.returncode == 2 and "not given" in r.stderr and "Usage" in r.stdout
    r = run_cli("-C", "15", "-L", "3", "-M", "4", "-x")
    assert r.returncode == 2 and "Usage" in r.stdout
    r = run_cli("-C", "15", "-L", "0", "-M", "4")
    assert r.returncode == 2


def test_cli_without_gpu_fails_loudly(hostlib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = run_cli("-C", "15", "-L", "3", "-M", "4", "-a", "7")
    assert r.returncode != 0 and "could not create" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args,factors", [
    (["-C", "15", "-L", "3", "-M", "4", "-a", "7"], (5, 3)),
    (["-C", "33", "-L", "5", "-M", "5", "-f", "7"], (11, 3)),         # the reference's documented example, with -f
    (["-C", "21", "-L", "9", "-M", "5", "-a", "2"], (3, 7)),
    (["-C", "35", "-L", "5", "-M", "6"], None),                      # loop mode over trial integers
    (["-C", "21", "-L", "11", "-M", "5", "-a", "2", "-F"], (3, 7)),  # every gate call queued
    (["-C", "21", "-L", "11", "-M", "5", "-a", "2", "-G"], (3, 7)),  # one kernel launch per gate
])
def test_cli_factors_on_gpu(hostlib, args, factors):
    ok = 0
    for seed in range(1, 9):
        r = run_cli(*args, "-s", str(seed), "-v", "-j")
        assert r.returncode in (0, 3), r.stdout + r.stderr
        if r.returncode == 0:
            line = [l for l in r.stdout.splitlines() if "Factors of" in l][-1]
            a, b = [int(x) for x in line.split("(")[1].split(")")[0].split(",")]
            Cn = int(args[1])
            if factors is not None and "trivial" not in r.stdout:
                assert (a, b) == factors, r.stdout
            if a * b == Cn and 1 < a < Cn:
                ok += 1
    assert ok >= 3, "the period should be found for most seeds"


@pytest.mark.gpu
def test_cli_reference_histogram_seed(hostlib):
    """same MT19937 stream and measurement as the reference: with -s 12345 the first measured omega values
    follow the stream whose 500-shot histogram is pinned in tests/golden (spot check through the CLI)"""
    r = run_cli("-C", "15", "-L", "3", "-M", "4", "-a", "7", "-s", "12345", "-V")
    assert r.returncode in (0, 3)
    assert "omega = " in r.stdout


@pytest.mark.gpu
def test_cli_measures_the_same_state_in_every_fusion_mode(hostlib):
    """default (circuit as fused passes), -F (everything queued), -G (one launch per gate): same MT19937 seed ->
    the same measured basis state, because the amplitudes are the same bits"""
    outs = []
    for extra in ([], ["-F"], ["-G"]):
        r = run_cli("-C", "21", "-L", "12", "-M", "5", "-a", "2", "-s", "7", "-V", *extra)
        assert r.returncode in (0, 3), r.stdout + r.stderr
        outs.append([l.split("omega")[0] for l in r.stdout.splitlines() if "Measured state" in l])
    assert outs[0] and outs[0] == outs[1] == outs[2], outs


@pytest.mark.gpu
def test_cli_state_dumps(hostlib, tmp_path, qc):
    """-O: the state right after the circuit (normalised superposition); -o: after the measurement (one basis state)"""
    pre, post = tmp_path / "pre.qcx", tmp_path / "post.qcx"
    r = run_cli("-C", "15", "-L", "6", "-M", "4", "-a", "7", "-s", "3", "-V", "-O", str(pre), "-o", str(post))
    assert r.returncode in (0, 3), r.stdout + r.stderr
    L, M, a = qc.load_state_file(pre)
    assert (L, M) == (6, 4) and abs(float(np.sum(np.square(a))) - 1.0) < 1e-12 and np.count_nonzero(a) > 2
    L, M, b = qc.load_state_file(post)
    measured = [int(l.split("Measured state")[1].split(",")[0]) for l in r.stdout.splitlines() if "Measured state" in l][-1]
    assert np.count_nonzero(b) == 1 and b[2 * measured] == 1.0


def test_classical_helpers_under_asan_ubsan(tmp_path):
    """host/qcx_classical.c swept over ordinary and hostile inputs with -fsanitize=address,undefined (CPU only)"""
    exe = str(tmp_path / "classical_san")
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                    os.path.join(ROOT, "tests", "c", "classical_sanitize.c"), os.path.join(ROOT, "host", "qcx_classical.c"), "-lm"],
                   check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "sanitized calls ok" in r.stdout, r.stdout + r.stderr


def test_read_omega_beyond_32_register_bits(hostlib):
    """L > 32 (this engine holds n = 34 on one GPU): x~ no longer fits the reference's unsigned int"""
    L, M = 34, 0
    state = (1 << 33) | 1                       # bits 33 and 0 set -> reversed: bits 0 and 33 of x~
    hostlib.qcx_read_omega.restype = C.c_double
    hostlib.qcx_read_omega.argtypes = [C.c_ulong, C.c_int, C.c_int]
    want = (1 + (1 << 33)) / float(1 << 34)
    assert hostlib.qcx_read_omega(state, L, M) == want
