"""CPU: a SECOND, independent derivation of the gates, to catch a misreading shared by the oracle and the kernels.

The oracle (oracle/qcx_oracle.c) restates the reference's loops; the HIP kernels are compared with it bit for bit.  If
both had misread the reference in the same way, those tests would still pass.  Here the gates are built the way the
reference DEFINES them, not the way it (or the oracle) computes them: dense 2^n x 2^n complex128 operators assembled
with Kronecker products from the 2x2 / 4x4 base matrices of Q:210-225 and the modular-multiply rule of Q:595-660 written
as a 0/1 matrix, applied with a dense matrix-vector product.  Different association order means different rounding, so
the agreement asked for is 1e-13 (north_star's amplitude tolerance is 1e-10), not bit equality.  n <= 9.

Also here: the measurement rule (Q:272-306) re-derived with numpy's running sum, and the report's Table I statistics
(R p.4) over 5 x 100 seeded shots."""
import math

import numpy as np
import pytest

S = 1.0 / math.sqrt(2.0)
H2 = np.array([[S, S], [S, -S]], dtype=np.complex128)                      # Q:210-213
I2 = np.eye(2, dtype=np.complex128)
P0 = np.array([[1, 0], [0, 0]], dtype=np.complex128)
P1 = np.array([[0, 0], [0, 1]], dtype=np.complex128)


def kron_all(factors_msb_first):
    out = np.array([[1.0 + 0j]])
    for f in factors_msb_first:
        out = np.kron(out, f)
    return out


def op_on(n, placed):
    """I (x) ... with the given 2x2 factors at the given qubits; qubit b = bit b of the index (Q:150-151), so the
    leftmost Kronecker factor is qubit n-1"""
    return kron_all([placed.get(q, I2) for q in range(n - 1, -1, -1)])


def dense_h(n, q):
    return op_on(n, {q: H2})


def dense_cphase(n, c, t, theta):
    # diag(1, 1, 1, e^{i theta}) on (c, t) (Q:220-225) = 1 + (e^{i theta} - 1) |1><1|_c |1><1|_t
    return np.eye(1 << n, dtype=np.complex128) + (np.exp(1j * theta) - 1.0) * op_on(n, {c: P1, t: P1})


def dense_camodc(n, M, Cn, atox, ctl):
    """Q:595-660 as a matrix: column k has its single 1 in row j, j = k unless the control bit of k is set and
    f = k mod 2^M < C, then the low M bits of j are (A f) mod C, A = atox mod C (Q:605)"""
    dim, A = 1 << n, atox % Cn
    U = np.zeros((dim, dim), dtype=np.complex128)
    for k in range(dim):
        f = k & ((1 << M) - 1)
        j = k
        if (k >> ctl) & 1 and f < Cn:
            j = (k - f) | ((A * f) % Cn)
        U[j, k] += 1.0
    return U


def as_complex(a):
    return a[0::2] + 1j * a[1::2]


def random_state(ob, n, seed):
    return ob.random_state(n, seed)


@pytest.mark.parametrize("n", [1, 2, 5, 8])
def test_hadamard_against_kronecker_definition(ob, n):
    for q in sorted({0, n // 2, n - 1}):
        a = random_state(ob, n, 10 + q)
        want = dense_h(n, q) @ as_complex(a)
        ob.hadamard(a, n, q)
        assert np.max(np.abs(as_complex(a) - want)) < 1e-13


@pytest.mark.parametrize("n,c,t,theta", [(2, 1, 0, math.pi / 2), (5, 4, 1, math.pi / 8), (5, 0, 3, -2.2), (8, 7, 6, math.pi / 2 ** 7),
                                         (8, 2, 5, 1.0), (7, 6, 0, math.pi)])
def test_cphase_against_kronecker_definition(ob, n, c, t, theta):
    a = random_state(ob, n, 77)
    want = dense_cphase(n, c, t, theta) @ as_complex(a)
    ob.cphase(a, n, c, t, theta)
    assert np.max(np.abs(as_complex(a) - want)) < 1e-13


@pytest.mark.parametrize("n,M,Cn,atox,ctl", [(7, 4, 15, 7, 4), (7, 4, 15, 4, 6), (8, 5, 21, 16, 7), (8, 5, 21, 2, 5), (6, 3, 7, 3, 5),
                                             (8, 4, 15, 6, 5)])            # the last one is not coprime: rows are SUMS (App. A#6)
def test_camodc_against_matrix_definition(ob, n, M, Cn, atox, ctl):
    a = random_state(ob, n, 5)
    want = dense_camodc(n, M, Cn, atox, ctl) @ as_complex(a)
    ob.camodc(a, n, M, Cn, atox, ctl)
    assert np.max(np.abs(as_complex(a) - want)) < 1e-13


@pytest.mark.parametrize("L,M", [(4, 0), (8, 0), (5, 3)])
def test_inverse_qft_schedule_against_dense_product(ob, L, M):
    """Q:678-690: l = n-1..M: H(l); k = l-1..M: CPHASE(l, k, pi / 2^(l-k)) -- as one product of dense operators"""
    n = L + M
    U = np.eye(1 << n, dtype=np.complex128)
    for l in range(n - 1, M - 1, -1):
        U = dense_h(n, l) @ U
        for k in range(l - 1, M - 1, -1):
            U = dense_cphase(n, l, k, math.pi / (1 << (l - k))) @ U
    a = random_state(ob, n, 9)
    want = U @ as_complex(a)
    ob.iqft(a, n, M)
    assert np.max(np.abs(as_complex(a) - want)) < 1e-13
    # the schedule has no final bit reversal and POSITIVE phases: on the M = 0 register it is the DFT matrix with
    # bit-reversed rows, F[rev(j), x] = exp(+2 pi i j x / 2^n) / sqrt(2^n)
    if M == 0:
        dim = 1 << n
        rev = np.array([int(format(j, f"0{n}b")[::-1], 2) for j in range(dim)])
        F = np.exp(2j * np.pi * np.outer(np.arange(dim), np.arange(dim)) / dim) / math.sqrt(dim)
        assert np.max(np.abs(U[rev, :] - F)) < 1e-12


@pytest.mark.parametrize("Cn,L,M,a", [(15, 3, 4, 7), (15, 4, 4, 7), (21, 4, 5, 2)])
def test_shor_circuit_against_dense_product(ob, Cn, L, M, a):
    """Q:712-737 with exact powers: H layer, C_AMODC ladder (control l carries a^(2^(l-M))), inverse QFT"""
    n = L + M
    v = np.zeros(1 << n, dtype=np.complex128)
    v[1] = 1.0                                                                # Q:318-324
    for l in range(M, n):
        v = dense_h(n, l) @ v
    for l in range(M, n):
        v = dense_camodc(n, M, Cn, pow(a, 1 << (l - M), Cn), l) @ v
    for l in range(n - 1, M - 1, -1):
        v = dense_h(n, l) @ v
        for k in range(l - 1, M - 1, -1):
            v = dense_cphase(n, l, k, math.pi / (1 << (l - k))) @ v
    s = np.zeros(2 << n); ob.reset(s, n); ob.quantum_computation(s, n, M, Cn, a)
    assert np.max(np.abs(as_complex(s) - v)) < 1e-13
    # and the physics: the x~ distribution (bits read reversed, Q:868-883) peaks at multiples of 2^L / period
    period = next(p for p in range(1, Cn) if pow(a, p, Cn) == 1)
    prob = np.abs(v) ** 2
    px = np.zeros(1 << L)
    for i in np.nonzero(prob > 1e-20)[0]:
        x = sum(((int(i) >> (n - 1 - p)) & 1) << p for p in range(L))
        px[x] += prob[i]
    top = np.argsort(px)[::-1][:period]
    assert all(min(abs(x - j * (1 << L) / period) for j in range(period + 1)) < 1.0 for x in top)


def test_measurement_rule_against_numpy_running_sum(ob):
    """Q:283-292: first index whose running sum of |amp|^2 reaches r, scanning 0 .. 2^n - 2; else the last index"""
    n = 9
    a = random_state(ob, n, 21)
    p = a[0::2] * a[0::2] + a[1::2] * a[1::2]
    cum = np.cumsum(p[:-1])                          # numpy accumulates sequentially in index order
    rs = np.random.RandomState(4)
    for r in list(rs.random_sample(200)) + [0.0, float(cum[0]), float(cum[100]), float(cum[-1]), 0.9999999999999999]:
        hit = np.nonzero(cum >= r)[0]
        want = int(hit[0]) if hit.size else (1 << n) - 1
        assert ob.measure(a.copy(), n, float(r)) == want          # (measure collapses its argument)


def test_report_table_one_statistics(ob):
    """R p.4 Table I: C=15, L=3, M=4, a=7, 5 x 100 shots: omega = 0, 1/4, 1/2, 3/4 occur 25.6 / 23.2 / 25.4 / 25.6 times per
    100 with sigma 3.93 / 1.47 / 3.00 / 2.87.  The reference seeds from time(NULL) (Q:1299), so only the statistics can be
    compared: 5 batches of 100 seeded shots of the oracle must be a sample the published means and sigmas are plausible
    for (each outcome Binomial(100, 1/4): mean 25, sigma 4.33)."""
    L, M, Cn, a = 3, 4, 15, 7
    n = L + M
    published_mean = {0.0: 25.6, 0.25: 23.2, 0.5: 25.4, 0.75: 25.6}
    published_sigma = {0.0: 3.93, 0.25: 1.47, 0.5: 3.00, 0.75: 2.87}
    final = np.zeros(2 << n); ob.reset(final, n); ob.quantum_computation(final, n, M, Cn, a)
    rng = ob.Rng(2024)
    batches = []
    for b in range(5):
        counts = {w: 0 for w in published_mean}
        for _ in range(100):
            idx = ob.measure(final.copy(), n, rng.uniform())
            counts[ob.read_omega(idx, L, M)] += 1                  # any other omega would be a KeyError: only four occur
        batches.append(counts)
    for w in published_mean:
        xs = np.array([b[w] for b in batches], dtype=float)
        assert abs(xs.mean() - 25.0) < 3 * 4.33 / math.sqrt(5)                  # our sample vs the exact law
        assert abs(published_mean[w] - 25.0) < 3 * 4.33 / math.sqrt(5)          # the report's sample vs the exact law
        assert abs(xs.mean() - published_mean[w]) < 3 * 4.33 * math.sqrt(2.0 / 5)   # the two samples vs each other
        assert 0.3 < xs.std(ddof=1) / 4.33 < 2.2 and 0.3 < published_sigma[w] / 4.33 < 2.2   # chi-square range for 4 dof


# ---- the reference's mat-vec, written out in Python floats (IEEE binary64, no contraction, no vectorisation) -----------------
def python_matvec(triplets, cur):
    """Q:393-413 literally: zero the new state; for every stored triplet, in insertion order,
    new[row] += M * cur[col] as four products and four sums in the order the reference writes them"""
    dim = len(cur) // 2
    new = [0.0] * (2 * dim)
    for row, col, mr, mi in triplets:
        cr, ci = cur[2 * col], cur[2 * col + 1]
        new[2 * row] = new[2 * row] + ((mr * cr) - (mi * ci))              # Q:409
        new[2 * row + 1] = new[2 * row + 1] + ((mr * ci) + (mi * cr))      # Q:412
    return np.array(new, dtype=np.float64)


def scan_triplets(n, free_bits, entry):
    """the reference's index-pair scan (Q:456-481, Q:529-562): rows ascending, columns ascending, keep (i, j) when they agree
    on every bit outside `free_bits`; entry(i, j) -> (re, im) of the base matrix element"""
    keep = ~sum(1 << b for b in free_bits)
    out = []
    for i in range(1 << n):
        for j in range(1 << n):
            if (i & keep) == (j & keep):
                mr, mi = entry(i, j)
                out.append((i, j, mr, mi))
    return out


@pytest.mark.parametrize("n,q", [(3, 0), (5, 2), (6, 5)])
def test_hadamard_rounding_order_against_python_floats(ob, n, q):
    s = 0.70710678118654752440                                             # M_SQRT1_2, Q:210-213
    trip = scan_triplets(n, [q], lambda i, j: ((-s if ((i >> q) & 1) and ((j >> q) & 1) else s), 0.0))
    a = ob.random_state(n, 3)
    want = python_matvec(trip, list(a))
    ob.hadamard(a, n, q)
    assert np.array_equal(a.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("n,c,t,theta", [(4, 3, 1, math.pi / 4), (6, 0, 5, 1.234), (5, 4, 2, math.pi / 2 ** 9)])
def test_cphase_rounding_order_against_python_floats(ob, n, c, t, theta):
    er, ei = ob.polar(theta)                                               # gsl_complex_polar(1, theta), Q:526

    def entry(i, j):                                                       # diag(1, 1, 1, e^{i theta}) on (c, t), Q:220-225
        bi, bj = 2 * ((i >> c) & 1) + ((i >> t) & 1), 2 * ((j >> c) & 1) + ((j >> t) & 1)
        if bi != bj:
            return (0.0, 0.0)                                              # explicit zeros are stored and multiplied too
        return (er, ei) if bi == 3 else (1.0, 0.0)
    trip = scan_triplets(n, [c, t], entry)
    a = ob.random_state(n, 4)
    want = python_matvec(trip, list(a))
    ob.cphase(a, n, c, t, theta)
    assert np.array_equal(a.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("n,M,Cn,atox,ctl", [(6, 3, 7, 3, 4), (7, 4, 15, 6, 5), (7, 4, 15, 7, 6)])
def test_camodc_rounding_order_against_python_floats(ob, n, M, Cn, atox, ctl):
    """Q:608-657: one triplet (j, k, 1 + 0i) per column k in ascending k; colliding rows (non-coprime multiplier) sum in that order"""
    A = atox % Cn
    trip = []
    for k in range(1 << n):
        f = k & ((1 << M) - 1)
        j = k
        if (k >> ctl) & 1 and f < Cn:
            j = (k - f) | ((A * f) % Cn)
        trip.append((j, k, 1.0, 0.0))
    a = ob.random_state(n, 6)
    want = python_matvec(trip, list(a))
    ob.camodc(a, n, M, Cn, atox, ctl)
    assert np.array_equal(a.view(np.uint64), want.view(np.uint64))
