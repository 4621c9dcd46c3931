"""GPU: the C-ABI sharded register (qcx_register_create_sharded: ONE process, W shards, exchange by k_pack_push stores into
the peers' buffers).  The shards are SPREAD OVER THE VISIBLE GPUS (qc.spread_devices): on a one-GPU box they all land on
device 0 -- the code path of an 8-GPU node except that the peer stores stay on the device -- and on a multi-GPU box the
very same tests issue real peer stores over xGMI, cross-device event waits and relay forwarding, without an edit.
Everything is compared with the oracle bit for bit, through the ordinary entry points of include/qcx.h; the
reference-style C program and the C host driver are run sharded through QCX_SHARDS / -g."""
import math
import os
import random
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.asarray(a).view(np.uint64)


@pytest.mark.parametrize("shards,n", [(2, 10), (4, 12), (8, 16), (16, 20)])
@pytest.mark.parametrize("fusion", [1, -1])
def test_hadamard_sweeps(qc, ob, shards, n, fusion):
    with qc.Register(n, 0, shards=shards, devices=qc.spread_devices(shards)) as reg:
        assert reg.shards == shards
        reg.set_fusion(fusion)
        reg.fill_random(5)
        want = ob.fill_random(n, 5)
        assert np.array_equal(bits(reg.read()), bits(want))
        for _ in range(2):
            for q in range(n):
                qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
        assert np.array_equal(bits(reg.read()), bits(want))
        ex, _ = reg.sharded_stats()
        assert 2 <= ex <= 6                                  # one trade per sweep + restoring the identity layout


@pytest.mark.parametrize("shards,C,L,M,a", [(2, 15, 3, 4, 7), (2, 21, 6, 5, 2), (4, 21, 8, 5, 2), (8, 21, 12, 5, 2), (4, 33, 9, 6, 7),
                                            (2, 8191, 9, 13, 3), (4, 8191, 10, 13, 3)])       # M > 12: the staged in-place multiply on every shard
def test_shor_circuit_and_measurement(qc, ob, shards, C, L, M, a):
    n = L + M
    if n - (shards.bit_length() - 1) - max(M, 6) < 2 * (shards.bit_length() - 1):
        pytest.skip("register too small for that many shards")
    rng, orng = qc.Rng(12345), ob.Rng(12345)
    with qc.Register(L, M, shards=shards, devices=qc.spread_devices(shards)) as reg:
        picks = []
        for shot in range(4):
            qc.reset_register(reg)
            qc.quantum_computation(C, a, reg)
            want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a)
            if shot == 0:
                assert abs(reg.norm2() - 1.0) < 1e-13
                assert np.array_equal(bits(reg.read()), bits(want))
            picks.append((qc.measure_state(reg, rng), ob.measure(want, n, orng.uniform())))
            assert np.array_equal(bits(reg.read()), bits(want))          # collapsed the same way
        assert all(x == y for x, y in picks), picks
        k = shards.bit_length() - 1
        # compact circuits (round 4): with a small orbit and room for the companion register the whole queue ran on [L register][orbit
        # column] shards -- trades included -- and every shard expanded its part
        import ctypes as Ct
        cc = Ct.c_ulong(0)
        qc.lib().qcx_compact_stats(reg._h, Ct.byref(cc))
        if (C, L, M) in ((21, 12, 5), (33, 9, 6)):
            assert cc.value == 4, cc.value
            # round 5: shots 2-4 measure right behind the circuit -- on the companion's compact form, the register is never expanded
            cm = Ct.c_ulong(0)
            qc.lib().qcx_compact_measure_stats(reg._h, Ct.byref(cm))
            assert cm.value == 3, cm.value
        if n - k >= M + 6 and M <= 12:
            # the circuit front (reset + Hadamard layer, shard-id qubits included + multiply ladder) went out as one write
            # pass per shard, without an exchange: one front per shot
            assert reg.fusion_stats()[0] == 4


def test_measurement_edges(qc, ob):
    n = 12
    with qc.Register(n, 0, shards=4, devices=qc.spread_devices(4)) as reg:
        for r in (0.0, 1e-300, 0.25, 0.5, 0.999999999, 1.0 - 2.0 ** -53):
            reg.fill_random(9)
            for q in (n - 1, 3, n - 2):
                qc.hadamard_gate(q, reg)
            want = ob.fill_random(n, 9)
            for q in (n - 1, 3, n - 2):
                ob.hadamard(want, n, q)
            assert qc.measure_state(reg, r) == ob.measure(want, n, r)
            assert np.array_equal(bits(reg.read()), bits(want))


@pytest.mark.parametrize("shards,n,M", [(2, 9, 0), (4, 12, 3), (4, 13, 5), (8, 16, 4)])
def test_random_programs(qc, ob, shards, n, M):
    rnd = random.Random(7 * shards + n)
    with qc.Register(n - M, M, shards=shards, devices=qc.spread_devices(shards)) as reg:
        for trial in range(6):
            reg.set_fusion(1 if trial % 2 == 0 else -1)
            reg.fill_random(trial)
            want = ob.fill_random(n, trial)
            for _ in range(rnd.randrange(20, 90)):
                t = rnd.random()
                if t < 0.5:
                    q = rnd.randrange(n)
                    qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
                elif t < 0.85 or M == 0:
                    c, tq = rnd.sample(range(n), 2)
                    th = rnd.uniform(-3.0, 3.0)
                    qc.c_phase_shift_gate(c, tq, th, reg); ob.cphase(want, n, c, tq, th)
                else:
                    Cn, A, ctl = rnd.randrange(2, (1 << M) + 1), rnd.randrange(1, 200), rnd.randrange(M, n)
                    qc.c_amodc_gate(Cn, A, ctl, reg); ob.camodc(want, n, M, Cn, A, ctl)
                if rnd.random() < 0.03:
                    assert abs(reg.norm2() - ob.norm2(want, n)) < 1e-12       # observing in a swapped layout is fine
            assert np.array_equal(bits(reg.read()), bits(want)), (shards, n, M, trial)


def test_state_io_and_timers(qc, ob, tmp_path):
    n = 12
    with qc.Register(n, 0, shards=4, devices=qc.spread_devices(4)) as reg, qc.Register(n, 0) as plain:
        reg.fill_random(2)
        qc.hadamard_gate(n - 1, reg)                                            # leaves a swapped layout behind
        want = ob.fill_random(n, 2); ob.hadamard(want, n, n - 1)
        assert np.array_equal(bits(reg.read(100, 3000)), bits(want[200:6200]))     # a window across shard boundaries
        p = str(tmp_path / "s.qcx")
        reg.save(p)
        plain.load(p)
        assert np.array_equal(bits(plain.read()), bits(want))
        w2 = ob.fill_random(n, 77)
        reg.write(w2[2 * 1000:2 * 3500], first=1000)
        want[2 * 1000:2 * 3500] = w2[2 * 1000:2 * 3500]
        assert np.array_equal(bits(reg.read()), bits(want))
        reg.timer_start()
        for q in range(n):
            qc.hadamard_gate(q, reg)
        assert reg.timer_stop() > 0.0
        assert reg.device_pointer() == 0
        with pytest.raises(qc.QcxError):
            reg.events_create(4)


def test_inverse_qft_entry_point(qc, ob):
    L, M = 9, 4
    n = L + M
    with qc.Register(L, M, shards=4, devices=qc.spread_devices(4)) as reg:
        reg.fill_random(4)
        qc.inverse_QFT(reg)
        want = ob.fill_random(n, 4); ob.iqft(want, n, M)
        assert np.array_equal(bits(reg.read()), bits(want))


def test_reference_style_c_program_sharded_by_environment(ob, tmp_path):
    """the program of tests/c/refstyle_circuit.c (reference names, Register by value, gsl_rng ...) knows nothing about
    shards: QCX_SHARDS makes qcx_register_create shard the register"""
    out = str(tmp_path / "refstyle")
    lib = os.path.join(ROOT, "quantumcomputer_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "refstyle_circuit.c"), "-L", lib, "-lqcx", "-lm",
                    "-Wl,-rpath," + lib, "-o", out], check=True)
    C, L, M, a = 21, 9, 5, 2
    n = L + M
    env = dict(os.environ, QCX_SHARDS="4")           # (no device list: the library spreads the shards over the visible GPUs)
    env.pop("QCX_SHARD_DEVICES", None)
    r = subprocess.run([out, str(C), str(L), str(M), str(a), "12345"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split()
    got = np.array([int(x, 16) for x in lines[:2 << n]], dtype=np.uint64)
    want = np.zeros(2 << n); ob.reset(want, n)
    ob.quantum_computation(want, n, M, C, a, ref_intpow=True)
    assert np.array_equal(got, want.view(np.uint64))
    assert int(lines[-1]) == ob.measure(want, n, ob.Rng(12345).uniform())


def test_host_driver_sharded():
    exe = os.path.join(ROOT, "host", "qcx_shor")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "host"), "-s"], check=True)
    import json
    strip = lambda t: [ln for ln in t.splitlines() if not ln.startswith("{")]
    found = 0
    for seed in ("1", "2", "3", "4", "5", "6"):
        args = [exe, "-C", "21", "-L", "9", "-M", "5", "-a", "2", "-s", seed, "-j"]
        single = subprocess.run(args, capture_output=True, text=True, timeout=120)
        shard = subprocess.run(args + ["-g", "4"], capture_output=True, text=True, timeout=120)      # spread over the visible GPUs
        # same seed -> same measured states, same attempts, same verdict: only the JSON line's timing and shard fields differ
        assert single.returncode == shard.returncode and single.returncode in (0, 3), shard.stderr
        assert strip(single.stdout) == strip(shard.stdout)
        js = json.loads([ln for ln in shard.stdout.splitlines() if ln.startswith("{")][0])
        assert js["shards"] == 4 and js["exchanges"] >= 1
        if shard.returncode == 0:
            found += 1
            assert "Factors of 21 found: (3, 7)" in shard.stdout or "Factors of 21 found: (7, 3)" in shard.stdout
    assert found >= 1


@pytest.mark.parametrize("shards,n,relays", [(2, 14, 1), (2, 16, 6), (4, 16, 4), (8, 18, 2)])
def test_multi_path_striping_gives_the_same_bits(qc, ob, shards, n, relays):
    """relays (GPUs without a shard; on a one-GPU box the same GPU again) carry a share of every chunk of every trade through staging buffers and a
    forwarding copy; the amplitudes must come out exactly as without them"""
    devs = qc.spread_devices(shards)
    with qc.Register(n, 0, shards=shards, devices=devs) as reg:
        reg.set_relays(qc.idle_devices(devs, relays))                          # idle GPUs first; on one GPU: device 0 again
        assert reg.relay_stats()[0] == relays
        reg.fill_random(8)
        want = ob.fill_random(n, 8)
        for _ in range(2):
            for q in list(range(n - 1, n - 6, -1)) + [0, 5, n - 1]:
                qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
            qc.c_phase_shift_gate(n - 1, 2, 0.7, reg); ob.cphase(want, n, n - 1, 2, 0.7)
        assert np.array_equal(bits(reg.read()), bits(want))
        ex, _ = reg.sharded_stats()
        assert ex >= 2 and reg.relay_stats()[1] > 0                          # stripes really went through the relays
        reg.set_relays([])                                                     # and off again
        for q in (n - 1, n - 2):
            qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
        assert np.array_equal(bits(reg.read()), bits(want))


@pytest.mark.parametrize("slices_log2,overlap", [(0, 1), (1, 1), (2, 1), (3, 1), (3, 0)])
def test_sliced_exchange_windows(qc, ob, monkeypatch, slices_log2, overlap):
    """a trade cut into 2^sigma slices on a second stream per shard, with the gates around it run slice by slice in its
    pre- and post-window: same bits for every slice count, with and without windows"""
    monkeypatch.setenv("QCX_SHARD_SLICES_LOG2", str(slices_log2))
    monkeypatch.setenv("QCX_SHARD_OVERLAP", str(overlap))
    n, M, shards = 18, 4, 4
    rnd = random.Random(31 + slices_log2)
    with qc.Register(n - M, M, shards=shards, devices=qc.spread_devices(shards)) as reg:
        sg, _ = reg.overlap_stats()
        assert sg == (slices_log2 if overlap else 0)         # slices only serve the windows
        for trial in range(3):
            reg.set_fusion(1 if trial != 1 else -1)
            reg.fill_random(40 + trial)
            want = ob.fill_random(n, 40 + trial)
            for _ in range(120):
                t = rnd.random()
                if t < 0.45:
                    q = rnd.choice([n - 1, n - 2, rnd.randrange(n), rnd.randrange(M, n)])      # plenty of global targets
                    qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
                elif t < 0.85:
                    c, tq = rnd.sample(range(n), 2)
                    th = rnd.uniform(-3.0, 3.0)
                    qc.c_phase_shift_gate(c, tq, th, reg); ob.cphase(want, n, c, tq, th)
                else:
                    Cn, A, ctl = rnd.randrange(2, 17), rnd.randrange(1, 99), rnd.randrange(M, n)
                    qc.c_amodc_gate(Cn, A, ctl, reg); ob.camodc(want, n, M, Cn, A, ctl)
            assert np.array_equal(bits(reg.read()), bits(want)), (slices_log2, overlap, trial)
        _, in_windows = reg.overlap_stats()
        ex, _ = reg.sharded_stats()
        assert ex >= 3
        if slices_log2 and overlap:
            assert in_windows > ex                               # gates really ran inside the windows


def test_sharded_register_from_plain_c(tmp_path):
    """tests/c/sharded_api.c: include/qcx.h only, no Python in the loop"""
    out = str(tmp_path / "sharded_api")
    lib = os.path.join(ROOT, "quantumcomputer_amd")
    subprocess.run(["gcc", "-std=gnu11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "sharded_api.c"), "-L", lib, "-lqcx", "-lm",
                    "-Wl,-rpath," + lib, "-o", out], check=True)
    r = subprocess.run([out, "11", "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


@pytest.mark.parametrize("shards,n", [(2, 12), (2, 16), (4, 20), (8, 30)])
def test_exchange_selfcheck_passes_and_catches_a_wrong_trade(qc, monkeypatch, shards, n):
    """the pre-flight check of qcx_register_create_sharded (runs by itself when the shards sit on several GPUs; forced
    here so that a one-GPU box exercises it too): trade a small register there and back, compare every amplitude with
    the generator bit for bit.  With a corrupted amplitude injected after the trade, creation must FAIL."""
    n = min(n, 24)
    monkeypatch.setenv("QCX_SHARD_SELFCHECK", "1")
    with qc.Register(n, 0, shards=shards) as reg:                       # devices=None: the library's own spreading
        assert reg.shards == shards and reg.selfchecks == 1
        reg.selfcheck()
        assert reg.selfchecks == 2
        devs = qc.spread_devices(shards)
        reg.set_relays(qc.idle_devices(devs, 2))                        # ... and once more through the relays
        assert reg.selfchecks == (3 if reg.relay_stats()[0] == 2 else 2)    # (registers too small to stripe set up no relays)
    monkeypatch.setenv("QCX_SHARD_SELFCHECK_INJECT", "1")
    with pytest.raises(qc.QcxError, match="self-check FAILED"):
        qc.Register(n, 0, shards=shards)
    monkeypatch.setenv("QCX_SHARD_SELFCHECK", "0")                     # skipped: creation succeeds again
    with qc.Register(n, 0, shards=shards) as reg:
        assert reg.selfchecks == 0


@pytest.mark.parametrize("shards,L,M", [(2, 12, 0), (4, 13, 4), (8, 14, 5), (4, 16, 0)])
def test_tolerance_mode_on_a_sharded_register(qc, ob, shards, L, M):
    """qcx_set_fusion(reg, 2) on a sharded register: every shard's passes merge their phase runs; phases whose other qubit
    sits in the shard id join the diagonals as constant factors.  Within 1e-12 of the oracle, not bit-identical, and the
    same measured indices as the exact mode for the same draws."""
    n = L + M
    Cn, a = (21, 2) if M >= 5 else (15, 7) if M == 4 else (1, 1)
    with qc.Register(L, M, shards=shards, devices=qc.spread_devices(shards)) as reg:
        reg.set_fusion(2)
        reg.fill_random(6)
        qc.inverse_QFT(reg)
        got = reg.read()
        want = ob.fill_random(n, 6); ob.iqft(want, n, M, 8)
        d = got - want
        assert float(np.max(np.hypot(d[0::2], d[1::2]))) <= 1e-12
        assert not np.array_equal(bits(got), bits(want))
        if M:
            picks = []
            for mode in (1, 2):
                reg.set_fusion(mode)
                out = []
                for r in (0.11, 0.52, 0.93):
                    qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
                    out.append(qc.measure_state(reg, r))
                picks.append(out)
            assert picks[0] == picks[1]


@pytest.mark.parametrize("shards,n,M,slices,overlap", [(2, 12, 0, 0, 0), (4, 16, 4, 2, 1), (8, 18, 5, 3, 1), (4, 14, 0, 0, 0)])
def test_staged_exchange_without_peer_access(qc, ob, monkeypatch, shards, n, M, slices, overlap):
    """no peer access between two devices (forced here: QCX_SHARD_FORCE_STAGED=1): a trade packs every shard into its OWN
    spare buffer and the chunks travel as hipMemcpyPeerAsync copies instead of peer stores -- creation no longer fails, the
    bits are the same, with and without sliced exchange windows, and the pre-flight self-check runs through the same path"""
    monkeypatch.setenv("QCX_SHARD_FORCE_STAGED", "1")
    monkeypatch.setenv("QCX_SHARD_SLICES_LOG2", str(slices))
    monkeypatch.setenv("QCX_SHARD_OVERLAP", str(overlap))
    monkeypatch.setenv("QCX_SHARD_SELFCHECK", "1")
    rnd = random.Random(shards * 100 + n)
    Cn = 21 if M >= 5 else 15
    with qc.Register(n - M, M, shards=shards, devices=qc.spread_devices(shards)) as reg:
        assert reg.selfchecks == 1
        for trial in range(3):
            reg.set_fusion(1 if trial != 1 else -1)
            reg.fill_random(60 + trial)
            want = ob.fill_random(n, 60 + trial)
            for _ in range(100):
                t = rnd.random()
                if t < 0.45:
                    q = rnd.choice([n - 1, n - 2, rnd.randrange(n), rnd.randrange(M, n)])
                    qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
                elif t < 0.85 or M == 0:
                    c, tq = rnd.sample(range(n), 2)
                    th = rnd.uniform(-3.0, 3.0)
                    qc.c_phase_shift_gate(c, tq, th, reg); ob.cphase(want, n, c, tq, th)
                else:
                    A, ctl = rnd.randrange(1, 99), rnd.randrange(M, n)
                    qc.c_amodc_gate(Cn, A, ctl, reg); ob.camodc(want, n, M, Cn, A, ctl)
            assert np.array_equal(bits(reg.read()), bits(want)), (shards, n, trial)
        ex, _ = reg.sharded_stats()
        assert ex >= 3
        if M:
            qc.reset_register(reg); qc.quantum_computation(Cn, 2 if Cn == 21 else 7, reg)
            w2 = np.zeros(2 << n); ob.reset(w2, n); ob.quantum_computation(w2, n, M, Cn, 2 if Cn == 21 else 7)
            assert qc.measure_state(reg, 0.61) == ob.measure(w2, n, 0.61)
