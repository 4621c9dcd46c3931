"""ctypes binding of the CPU ORACLE (oracle/qcx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg -- never by quantumcomputer_amd/ or host/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libqcx_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("qcx_oracle.c", "qcx_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


class _Rng(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


class _Coo(C.Structure):
    _fields_ = [("row", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p),
                ("nz", C.c_size_t), ("cap", C.c_size_t), ("keep_zeros", C.c_int)]


class _Reg(C.Structure):
    _fields_ = [("L", C.c_int), ("M", C.c_int), ("n", C.c_uint), ("dim", C.c_uint64),
                ("buf", C.POINTER(C.c_double) * 2), ("cur", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        dp = C.POINTER(C.c_double)
        L.orc_rng_set.argtypes = [C.POINTER(_Rng), C.c_uint32]
        L.orc_rng_get.argtypes = [C.POINTER(_Rng)]; L.orc_rng_get.restype = C.c_uint32
        L.orc_rng_uniform.argtypes = [C.POINTER(_Rng)]; L.orc_rng_uniform.restype = C.c_double
        L.orc_coo_init.argtypes = [C.POINTER(_Coo), C.c_size_t, C.c_int]
        L.orc_coo_free.argtypes = [C.POINTER(_Coo)]
        L.orc_reg_init.argtypes = [C.POINTER(_Reg), C.c_int, C.c_int]
        L.orc_reg_free.argtypes = [C.POINTER(_Reg)]
        L.orc_reg_state.argtypes = [C.POINTER(_Reg)]; L.orc_reg_state.restype = dp
        L.orc_lit_reset.argtypes = [C.POINTER(_Reg)]
        L.orc_lit_hadamard.argtypes = [C.c_uint, C.POINTER(_Reg), C.POINTER(_Coo)]
        L.orc_spmv_hadamard.argtypes = [C.c_uint, C.POINTER(_Reg), C.POINTER(_Coo)]
        L.orc_lit_cphase.argtypes = [C.c_uint, C.c_uint, C.c_double, C.POINTER(_Reg), C.POINTER(_Coo)]
        L.orc_lit_camodc.argtypes = [C.c_uint, C.c_ulonglong, C.c_uint, C.POINTER(_Reg), C.POINTER(_Coo)]
        L.orc_lit_iqft.argtypes = [C.POINTER(_Reg), C.POINTER(_Coo)]
        L.orc_lit_quantum_computation.argtypes = [C.c_uint, C.c_uint, C.c_int, C.POINTER(_Reg), C.POINTER(_Coo)]
        L.orc_pair_reset.argtypes = [dp, C.c_uint]
        L.orc_pair_hadamard.argtypes = [dp, C.c_uint, C.c_uint, C.c_int]
        L.orc_pair_cphase.argtypes = [dp, C.c_uint, C.c_uint, C.c_uint, C.c_double, C.c_int]
        L.orc_pair_camodc.argtypes = [dp, C.c_uint, C.c_uint, C.c_uint, C.c_ulonglong, C.c_uint, C.c_int]
        L.orc_pair_iqft.argtypes = [dp, C.c_uint, C.c_uint, C.c_int]
        L.orc_pair_quantum_computation.argtypes = [dp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int]
        L.orc_measure.argtypes = [dp, C.c_uint, C.c_double]; L.orc_measure.restype = C.c_uint64
        L.orc_measure_range.argtypes = [dp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_double,
                                        C.POINTER(C.c_uint64), dp]
        L.orc_measure_range.restype = C.c_int
        L.orc_norm2.argtypes = [dp, C.c_uint]; L.orc_norm2.restype = C.c_double
        L.orc_fill_random.argtypes = [dp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double]
        L.orc_polar.argtypes = [C.c_double, dp, dp]
        L.orc_basis_iqft_window.argtypes = [C.c_uint64, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, dp]
        L.orc_basis_iqft_window.restype = None
        L.orc_shor_front_window.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_uint64, C.c_uint64, dp]
        L.orc_shor_front_window.restype = None
        L.orc_ref_intpow.argtypes = [C.c_double, C.c_double]; L.orc_ref_intpow.restype = C.c_uint
        L.orc_modpow.argtypes = [C.c_ulonglong] * 3; L.orc_modpow.restype = C.c_ulonglong
        L.orc_gcd.argtypes = [C.c_uint, C.c_uint]; L.orc_gcd.restype = C.c_uint
        L.orc_cf_denominators.argtypes = [C.c_double, C.c_uint, C.POINTER(C.c_uint)]
        L.orc_read_omega.argtypes = [C.c_uint64, C.c_int, C.c_int]; L.orc_read_omega.restype = C.c_double
    return _lib


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Rng:
    """GSL-flavoured MT19937 (gsl_rng_mt19937)."""

    def __init__(self, seed):
        self._s = _Rng()
        lib().orc_rng_set(C.byref(self._s), seed & 0xFFFFFFFF)

    def get(self):
        return int(lib().orc_rng_get(C.byref(self._s)))

    def uniform(self):
        return float(lib().orc_rng_uniform(C.byref(self._s)))


def random_state(n, seed=7):
    """The SURVEY s8(d) input: re, im ~ U(-0.5, 0.5) from MT19937(seed), L2-normalised.

    Uses numpy's MT19937 with the same legacy init_genrand seeding as GSL (seed != 0),
    one 32-bit draw per component, so it is cheap at large n and reproducible.
    """
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    raw = bg.random_raw(2 << n).astype(np.float64)
    a = raw / 4294967296.0 - 0.5
    a /= np.sqrt(np.sum(a * a))
    return np.ascontiguousarray(a)


def fill_random(n, seed, first=0, count=None):
    """window [first, first+count) of the synthetic state qcx_state_fill_random(seed) makes at n qubits"""
    if count is None:
        count = (1 << n) - first
    a = np.empty(2 * count, dtype=np.float64)
    lib().orc_fill_random(_dp(a), first, count, seed, float(np.sqrt(6.0 / float(1 << n))))
    return a


class LiteralRegister:
    """The reference's double-buffered register driven by the literal COO algorithm."""

    def __init__(self, L, M, keep_zeros=True):
        self._r = _Reg()
        self._m = _Coo()
        if lib().orc_reg_init(C.byref(self._r), L, M):
            raise MemoryError
        lib().orc_coo_init(C.byref(self._m), 2 << (L + M), int(keep_zeros))
        self.L, self.M, self.n = L, M, L + M

    def close(self):
        if self._r is not None:
            lib().orc_reg_free(C.byref(self._r)); lib().orc_coo_free(C.byref(self._m))
            self._r = None

    def __del__(self):
        self.close()

    def state(self):
        p = lib().orc_reg_state(C.byref(self._r))
        return np.ctypeslib.as_array(p, shape=(2 << self.n,))

    def set_state(self, a):
        self.state()[:] = a

    def reset(self): lib().orc_lit_reset(C.byref(self._r))
    def hadamard(self, q): lib().orc_lit_hadamard(q, C.byref(self._r), C.byref(self._m))
    def spmv_hadamard(self, q): lib().orc_spmv_hadamard(q, C.byref(self._r), C.byref(self._m))
    def cphase(self, c, t, th): lib().orc_lit_cphase(c, t, th, C.byref(self._r), C.byref(self._m))
    def camodc(self, Cn, atox, ctl): lib().orc_lit_camodc(Cn, atox, ctl, C.byref(self._r), C.byref(self._m))
    def iqft(self): lib().orc_lit_iqft(C.byref(self._r), C.byref(self._m))
    def quantum_computation(self, Cn, a, ref_intpow=False):
        lib().orc_lit_quantum_computation(Cn, a, int(ref_intpow), C.byref(self._r), C.byref(self._m))


# -- pairwise in-place layer on numpy arrays (float64, length 2*2^n) ----------
def reset(a, n): lib().orc_pair_reset(_dp(a), n)
def hadamard(a, n, q, threads=1): lib().orc_pair_hadamard(_dp(a), n, q, threads)
def cphase(a, n, c, t, theta, threads=1): lib().orc_pair_cphase(_dp(a), n, c, t, theta, threads)
def camodc(a, n, M, Cn, atox, ctl, threads=1): lib().orc_pair_camodc(_dp(a), n, M, Cn, atox, ctl, threads)
def iqft(a, n, M, threads=1): lib().orc_pair_iqft(_dp(a), n, M, threads)
def quantum_computation(a, n, M, Cn, aa, ref_intpow=False, threads=1):
    lib().orc_pair_quantum_computation(_dp(a), n, M, Cn, aa, int(ref_intpow), threads)
def measure(a, n, r): return int(lib().orc_measure(_dp(a), n, r))
def norm2(a, n): return float(lib().orc_norm2(_dp(a), n))


def measure_range(a, first, count, last_excluded, cum_in, r):
    idx = C.c_uint64(0); cum = C.c_double(0.0)
    hit = lib().orc_measure_range(_dp(a), first, count, last_excluded, cum_in, r, C.byref(idx), C.byref(cum))
    return bool(hit), int(idx.value), float(cum.value)


def basis_iqft_window(x, n, M, first, count):
    """amplitudes [first, first+count) of inverse_QFT applied to the basis state |x>, one scalar chain per index"""
    out = np.empty(2 * count, dtype=np.float64)
    lib().orc_basis_iqft_window(int(x), n, M, int(first), int(count), _dp(out))
    return out


def shor_front_window(n, M, Cn, a, first, count, ref_intpow=False):
    """amplitudes [first, first+count) after the Hadamard layer and the modular-multiply ladder of quantum_computation
    on the reset state (valid for gcd(a, C) = 1 and C <= 2^M)"""
    assert gcd(a, Cn) == 1 and Cn <= (1 << M)
    out = np.empty(2 * count, dtype=np.float64)
    lib().orc_shor_front_window(n, M, Cn, a, int(ref_intpow), int(first), int(count), _dp(out))
    return out


def polar(theta):
    re, im = C.c_double(0.0), C.c_double(0.0)
    lib().orc_polar(float(theta), C.byref(re), C.byref(im))
    return re.value, im.value


def ref_intpow(b, p): return int(lib().orc_ref_intpow(float(b), float(p)))
def modpow(a, e, m): return int(lib().orc_modpow(a, e, m))
def gcd(a, b): return int(lib().orc_gcd(a, b))
def read_omega(state, L, M): return float(lib().orc_read_omega(state, L, M))


def cf_denominators(omega, count=15):
    out = (C.c_uint * count)()
    lib().orc_cf_denominators(omega, count, out)
    return list(out)
