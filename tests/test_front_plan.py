"""CPU: the circuit front on a basis state (K0b).  `qcx_front_plan` returns the host half -- which leading gates of a list
have the closed form and the parameters `k_basis_front` would get -- and this file restates the kernel's per-amplitude
rule in numpy (populated blocks, the residue chain, the sign parity, the magnitude v_k) and compares the resulting state
with the oracle applying the same gates one by one to the same basis state, bit for bit.  The GPU suite
(tests/test_gpu_basis_front.py) then only has to show that the kernel does what this emulator does."""
import numpy as np
import pytest


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def emulate_front(n, B):
    """k_basis_front (csrc/qcx_kernels.h) for every index at once"""
    M = B.M
    idx = np.arange(1 << n, dtype=np.uint64)
    low = np.uint64((1 << M) - 1)
    blk = idx & ~low
    f = np.full(idx.shape, B.basis & int(low), dtype=np.int64)
    for g in range(B.ncam):
        on = ((blk >> np.uint64(B.ctl[g])) & np.uint64(1)).astype(bool) & (f < B.C[g])
        f = np.where(on, (B.A[g] * f) % B.C[g], f)
    pop = (blk & np.uint64(B.fixed_mask)) == np.uint64(B.basis & B.fixed_mask)
    free_low = B.hmask & int(low)
    hit = ((((idx & low).astype(np.int64)) ^ f) & ~free_low) == 0
    par = np.zeros(idx.shape, dtype=np.int64)
    sm = B.sign_mask
    b = 0
    while sm >> b:
        if (sm >> b) & 1:
            par ^= ((idx >> np.uint64(b)) & np.uint64(1)).astype(np.int64)
        b += 1
    state = np.zeros(2 << n)
    val = np.where(par == 1, -B.v, B.v)
    sel = pop & hit
    state[0::2][sel] = val[sel]
    return state


def basis_state(n, b):
    s = np.zeros(2 << n)
    s[2 * b] = 1.0
    return s


@pytest.mark.parametrize("C,L,M,a", [(15, 8, 4, 7), (21, 9, 5, 2), (35, 7, 6, 2), (21, 6, 5, 2), (15, 4, 4, 7), (8191, 4, 13, 3), (16381, 3, 14, 2)])
def test_shor_front_from_the_reset_state(qc, ob, C, L, M, a):
    n = L + M
    descs = [(0, l, 0, 0.0, 0.0, 0, 0) for l in range(M, n)]
    x = a % C
    for l in range(M, n):
        descs.append((2, l, 0, 0.0, 0.0, C, x)); x = (x * x) % C
    descs.append((0, n - 1, 0, 0.0, 0.0, 0, 0))                       # the inverse QFT's first Hadamard: not part of the front
    used, B = qc.front_plan(n, M, 1, descs)
    assert used == 2 * L and B.ncam == L and B.hmask == ((1 << n) - 1) & ~((1 << M) - 1)
    want = basis_state(n, 1)
    for l in range(M, n):
        ob.hadamard(want, n, l)
    x = a % C
    for l in range(M, n):
        ob.camodc(want, n, M, C, x, l); x = (x * x) % C
    assert np.array_equal(bits(emulate_front(n, B)), bits(want))


@pytest.mark.parametrize("seed", range(12))
def test_random_fronts_from_random_basis_states(qc, ob, seed):
    rs = np.random.RandomState(900 + seed)
    n, M = int(rs.randint(8, 15)), int(rs.choice([0, 2, 4, 5]))
    basis = int(rs.randint(0, 1 << n))
    lo = M if rs.rand() < 0.7 else 0
    hs = [int(q) for q in rs.permutation(np.arange(lo, n))[: int(rs.randint(0, n - lo + 1))]]
    descs = [(0, q, 0, 0.0, 0.0, 0, 0) for q in hs]
    cams = []
    if M:
        for _ in range(int(rs.randint(0, 8))):
            Cn = int(rs.randint(2, (1 << M) + 1)); A = int(rs.randint(1, 3 * Cn)); ctl = int(rs.randint(M, n))
            cams.append((Cn, A, ctl)); descs.append((2, ctl, 0, 0.0, 0.0, Cn, A % Cn))
    tail = int(rs.randint(0, n))
    descs.append((0, hs[0] if hs else tail, 0, 0.0, 0.0, 0, 0))       # a repeated / later Hadamard ends the front
    used, B = qc.front_plan(n, M, basis, descs)
    in_low = any(q < M for q in hs)
    if hs:
        assert used == len(hs) + (0 if in_low else len(cams))          # multiplies join only while the M register is untouched
    else:
        assert used >= len(cams)                                        # (no Hadamard first: the trailing one may join behind zero multiplies)
    want = basis_state(n, basis)
    for d in descs[:used]:
        if d[0] == 0:
            ob.hadamard(want, n, d[1])
        else:
            ob.camodc(want, n, M, d[5], d[6], d[1])
    assert np.array_equal(bits(emulate_front(n, B)), bits(want)), (n, M, basis, hs, cams)


def test_front_respects_the_switch_and_the_table_form(qc):
    n, M = 12, 4
    descs = [(0, 5, 0, 0.0, 0.0, 0, 0), (2, 6, 0, 0.0, 0.0, 21, 2)]   # C = 21 > 2^4: table form, never in a front
    used, B = qc.front_plan(n, M, 1, descs)
    assert used == 1 and B.ncam == 0
    old = qc.lib().qcx_tune_get(b"fuse_front")
    try:
        qc.tune(fuse_front=0)
        assert qc.front_plan(n, M, 1, descs)[0] == 0
    finally:
        qc.tune(fuse_front=old)


def test_compact_plan_gives_the_orbit_of_the_ladder(qc):
    """qcx_compact_plan (pure host): behind the front of a Shor circuit the M register reads a residue of the multiply ladder's
    orbit; the compact form [L register][orbit column] exists when that orbit is small against 2^M"""
    import ctypes as C
    from quantumcomputer_amd._lib import GateDesc

    def plan(n, M, descs, basis=1):
        arr = (GateDesc * max(len(descs), 1))()
        for i, d in enumerate(descs):
            arr[i].type, arr[i].q, arr[i].mask, arr[i].c, arr[i].s, arr[i].C, arr[i].A = d
        used, cb, ncols = C.c_uint(0), C.c_uint(0), C.c_uint(0)
        orbit = (C.c_uint16 * 16)()
        st = qc.lib().qcx_compact_plan(n, M, basis, len(descs), C.cast(arr, C.c_void_p), C.byref(used), C.byref(cb), C.byref(ncols), orbit)
        assert st == 0
        return used.value, cb.value, [int(orbit[j]) for j in range(ncols.value)]

    def front(Cn, a, L, M):
        n = L + M
        d = [(0, l, 0, 0.0, 0.0, 0, 0) for l in range(M, n)]
        x = a % Cn
        for l in range(M, n):
            d.append((2, l, 0, 0.0, 0.0, Cn, x)); x = (x * x) % Cn
        return n, d + [(0, n - 1, 0, 0.0, 0.0, 0, 0)]

    n, d = front(21, 2, 12, 5)
    assert plan(n, 5, d) == (24, 3, [1, 2, 4, 8, 11, 16])
    n, d = front(15, 7, 10, 4)
    assert plan(n, 4, d) == (20, 2, [1, 4, 7, 13])
    n, d = front(255, 2, 9, 8)
    assert plan(n, 8, d) == (18, 3, [1, 2, 4, 8, 16, 32, 64, 128])
    n, d = front(15, 2, 10, 4)                    # orbit {1, 2, 4, 8}: four columns of 16 values
    assert plan(n, 4, d) == (20, 2, [1, 2, 4, 8])
    n, d = front(31, 3, 10, 5)                    # 3 generates all 30 units mod 31: no compact form
    assert plan(n, 5, d) == (20, 0, [])
    n, d = front(21, 2, 12, 5)                    # a Hadamard on an M-register qubit in the front: its low bits are free, no compact form
    assert plan(n, 5, [(0, 2, 0, 0.0, 0.0, 0, 0)] + d)[2] == []
    assert plan(12, 4, [(1, 0, (1 << 5) | (1 << 7), 1.0, 0.0, 0, 0)]) == (0, 0, [])       # no front at all
