"""Host-side mirror of the reference's gate/state interface over libqcx.so.

Function names, argument order and qubit numbering follow qc_shor.c so that code
(and tests) written against the reference read the same here:

    reference (qc_shor.c)                       here
    ------------------------------------------  -------------------------------------
    Register + alloc/free (194-203, 1316-1333)  Register(L_size, M_size) / .close()
    reset_register(reg)              318-324    reset_register(reg)
    hadamard_gate(q, &reg, matrix)   442-484    hadamard_gate(q, reg, matrix=None)
    c_phase_shift_gate(c, q, th, ..) 513-565    c_phase_shift_gate(c, q, theta, reg)
    c_amodc_gate(C, atox, c, ..)     595-660    c_amodc_gate(C, atox, c, reg)
    inverse_QFT(&reg, matrix)        678-690    inverse_QFT(reg)
    quantum_computation(C, a, ..)    712-737    quantum_computation(C, a, reg)
    measure_state(reg, rng)          272-306    measure_state(reg, rng)
    swap_states(&reg)                242-249    swap_states(reg)   (no-op: in place)
    gsl_rng mt19937                  1296-1299  Rng(seed)

The `matrix` scratch argument of the reference is accepted and ignored: no gate
matrix is ever built.  Everything executes in the HIP library; a missing library
or GPU raises (see _lib.py).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class Rng:
    """gsl_rng_mt19937 equivalent (qc_shor.c:1296-1299)."""

    def __init__(self, seed=0):
        self._h = lib().qcx_rng_alloc()
        if not self._h:
            raise MemoryError("qcx_rng_alloc")
        lib().qcx_rng_set(self._h, seed & 0xFFFFFFFF)

    def set(self, seed):
        lib().qcx_rng_set(self._h, seed & 0xFFFFFFFF)

    def get(self):
        return int(lib().qcx_rng_get(self._h))

    def uniform(self):
        return float(lib().qcx_rng_uniform(self._h))

    def close(self):
        if self._h:
            lib().qcx_rng_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Register:
    """The qubit register (qc_shor.c:194-203): L_size, M_size, num_qubits, num_states and the
    state vector, which lives in HBM as 2^n interleaved (re, im) doubles, updated in place."""

    def __init__(self, L_size, M_size, shards=1, devices=None):
        """shards > 1: the register is sharded over that many GPUs by this process (qcx_register_create_sharded);
        devices = HIP device of each shard (default: spread over the visible GPUs, see spread_devices; entries may
        repeat; [-1] = dry run: the schedule only, see sharded_trace)"""
        h = C.c_void_p()
        if shards == 1 and devices is None:
            check(lib().qcx_register_create(int(L_size), int(M_size), C.byref(h)), "qcx_register_create")
        else:
            dv = None
            if devices is not None:
                devices = list(devices) + [devices[-1]] * max(0, int(shards) - len(devices))
                dv = (C.c_int * len(devices))(*devices)
            check(lib().qcx_register_create_sharded(int(L_size), int(M_size), int(shards), dv, C.byref(h)), "qcx_register_create_sharded")
        self._h = h
        self.L_size = int(L_size)
        self.M_size = int(M_size)
        self.num_qubits = int(lib().qcx_num_qubits(h))
        self.num_states = int(lib().qcx_num_states(h))

    # -- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().qcx_register_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state access (gsl_vector_complex_get/set uses; testing_and_debug.c:7-37) ------------
    def read(self, first=0, count=None):
        """Amplitudes [first, first+count) as a float64 array of 2*count (re, im) values."""
        if count is None:
            count = self.num_states - first
        out = np.empty(2 * count, dtype=np.float64)
        check(lib().qcx_state_read(self._h, first, count, out.ctypes.data_as(C.c_void_p)), "qcx_state_read")
        return out

    def write(self, amps, first=0):
        a = np.ascontiguousarray(amps, dtype=np.float64)
        if a.size % 2:
            raise ValueError("amplitudes are (re, im) pairs")
        check(lib().qcx_state_write(self._h, first, a.size // 2, a.ctypes.data_as(C.c_void_p)), "qcx_state_write")

    def save(self, path):
        """state file: 64-byte header + interleaved binary64 amplitudes (qcx_state_save); see load_state_file()"""
        check(lib().qcx_state_save(self._h, str(path).encode()), "qcx_state_save")

    def load(self, path):
        check(lib().qcx_state_load(self._h, str(path).encode()), "qcx_state_load")

    def fill_random(self, seed):
        """synthetic dense state generated on the device (include/qcx.h: qcx_state_fill_random)"""
        check(lib().qcx_state_fill_random(self._h, int(seed)), "qcx_state_fill_random")

    def norm2(self):
        """Total probability (testing_and_debug.c:28-37), tree-summed on the GPU."""
        out = C.c_double(0.0)
        check(lib().qcx_norm2(self._h, C.byref(out)), "qcx_norm2")
        return out.value

    @property
    def shards(self):
        return int(lib().qcx_register_shards(self._h))

    def sharded_stats(self):
        """(exchanges, pack passes) a sharded register has performed"""
        e, p = C.c_ulong(0), C.c_ulong(0)
        check(lib().qcx_sharded_stats(self._h, C.byref(e), C.byref(p)), "qcx_sharded_stats")
        return e.value, p.value

    def selfcheck(self):
        """the pre-flight exchange check on demand (runs by itself at creation when the shards sit on several GPUs)"""
        check(lib().qcx_sharded_selfcheck(self._h), "qcx_sharded_selfcheck")

    @property
    def selfchecks(self):
        return int(lib().qcx_sharded_selfchecks(self._h))

    def set_relays(self, devices):
        """multi-path striping: GPUs without a shard relay a share of every trade ([] = off)"""
        dv = (C.c_int * max(len(devices), 1))(*devices)
        check(lib().qcx_sharded_set_relays(self._h, len(devices), dv), "qcx_sharded_set_relays")

    def relay_stats(self):
        n, b = C.c_uint(0), C.c_ulong(0)
        check(lib().qcx_sharded_relay_stats(self._h, C.byref(n), C.byref(b)), "qcx_sharded_relay_stats")
        return n.value, b.value

    def overlap_stats(self):
        """(log2 of the slices a trade is cut into, gates that ran inside exchange windows so far)"""
        sg, g = C.c_uint(0), C.c_ulong(0)
        check(lib().qcx_sharded_overlap_stats(self._h, C.byref(sg), C.byref(g)), "qcx_sharded_overlap_stats")
        return sg.value, g.value

    def sharded_trace(self):
        """diagnostics: the steps a dry-run sharded register has scheduled since the last call, as text"""
        need = C.c_size_t(0)
        lib().qcx_sharded_trace(self._h, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        check(lib().qcx_sharded_trace(self._h, buf, need.value, C.byref(need)), "qcx_sharded_trace")
        return buf.value.decode()

    def sharded_restore_identity(self):
        check(lib().qcx_sharded_restore_identity(self._h), "qcx_sharded_restore_identity")

    def sharded_layout(self):
        """logical qubit -> physical index bit of a sharded register"""
        perm = (C.c_uint * self.num_qubits)()
        check(lib().qcx_sharded_layout(self._h, perm, self.num_qubits), "qcx_sharded_layout")
        return list(perm)

    def total_probability(self):
        """Total probability summed in the reference's order (testing_and_debug.c:28-37: index-ascending, one addition
        per amplitude) -- the exact scan of the measurement run to the end."""
        out = C.c_double(0.0)
        check(lib().qcx_total_probability(self._h, C.byref(out)), "qcx_total_probability")
        return out.value

    def set_fusion(self, enable=True):
        """Fused LDS-tile passes (bit-identical results).  True/1: every gate call is queued; False/0 (default): only
        the whole-circuit calls (inverse_QFT, quantum_computation) run as fused passes; -1: strictly one kernel launch
        per gate, inside the whole-circuit calls too; 2 (FUSION_TOLERANCE): opt-in tolerance mode -- runs of controlled
        phases sharing a qubit are merged into one diagonal, NOT bit-exact (rounding-level differences, include/qcx.h)."""
        if enable is True or enable is False:
            mode = int(enable)
        else:
            mode = -1 if int(enable) < 0 else min(int(enable), 2)
        check(lib().qcx_set_fusion(self._h, mode), "qcx_set_fusion")

    def flush(self):
        check(lib().qcx_flush(self._h), "qcx_flush")

    def fusion_stats(self):
        p, g = C.c_ulong(0), C.c_ulong(0)
        check(lib().qcx_fusion_stats(self._h, C.byref(p), C.byref(g)), "qcx_fusion_stats")
        return p.value, g.value

    def synchronize(self):
        check(lib().qcx_synchronize(self._h), "qcx_synchronize")

    def set_stream(self, hip_stream_ptr):
        check(lib().qcx_register_set_stream(self._h, C.c_void_p(hip_stream_ptr)), "qcx_register_set_stream")

    def device_pointer(self):
        return int(lib().qcx_device_pointer(self._h) or 0)

    def events_create(self, count):
        check(lib().qcx_events_create(self._h, count), "qcx_events_create")

    def event_record(self, slot):
        check(lib().qcx_event_record(self._h, slot), "qcx_event_record")

    def event_elapsed(self, a, b):
        ms = C.c_double(0.0)
        check(lib().qcx_event_elapsed(self._h, a, b, C.byref(ms)), "qcx_event_elapsed")
        return ms.value

    def timer_start(self):
        check(lib().qcx_timer_start(self._h), "qcx_timer_start")

    def timer_stop(self):
        ms = C.c_double(0.0)
        check(lib().qcx_timer_stop(self._h, C.byref(ms)), "qcx_timer_stop")
        return ms.value


# ---- the reference's free functions ----------------------------------------------------------
def reset_register(reg):
    check(lib().qcx_reset_register(reg._h), "reset_register")


def hadamard_gate(qubit_num, reg, matrix=None):
    check(lib().qcx_hadamard_gate(qubit_num, reg._h), "hadamard_gate")


def c_phase_shift_gate(c_qubit_num, qubit_num, theta, reg, matrix=None):
    check(lib().qcx_c_phase_shift_gate(c_qubit_num, qubit_num, float(theta), reg._h), "c_phase_shift_gate")


def c_amodc_gate(C_, atox, c_qubit_num, reg, matrix=None):
    check(lib().qcx_c_amodc_gate(C_, int(atox), c_qubit_num, reg._h), "c_amodc_gate")


def swap_states(reg):
    check(lib().qcx_swap_states(reg._h), "swap_states")


def inverse_QFT(reg, matrix=None):
    check(lib().qcx_inverse_QFT(reg._h), "inverse_QFT")


def quantum_computation(C_, a, reg, matrix=None, ref_intpow=False):
    """ref_intpow=True reproduces the reference's 32-bit INT_POW(a, x) (qc_shor.c:158-159, 729)."""
    check(lib().qcx_quantum_computation(C_, a, int(bool(ref_intpow)), reg._h), "quantum_computation")


def measure_state(reg, rng):
    """Collapse the register; returns the measured basis-state index (qc_shor.c:272-306).
    `rng` is an Rng, or a float r in [0,1) to inject the uniform draw directly."""
    out = C.c_ulong(0)
    if isinstance(rng, Rng):
        check(lib().qcx_measure_state(reg._h, rng._h, C.byref(out)), "measure_state")
    else:
        check(lib().qcx_measure_state_r(reg._h, float(rng), C.byref(out)), "measure_state")
    return int(out.value)


def read_omega(state_num, reg):
    """x~/2^L with the L register read in reversed bit order (qc_shor.c:868-883)."""
    x = 0
    for p in range(reg.L_size):
        x |= ((state_num >> (reg.L_size + reg.M_size - 1 - p)) & 1) << p
    return x / float(1 << reg.L_size)


def display_state(reg, file=None, limit=None):
    """testing_and_debug.c:7-26: one line per basis state with non-zero amplitude, qubits MSB first,
    followed by |amplitude| with two decimals.  Reads the whole state to the host (debug tool)."""
    import sys
    out = file or sys.stdout
    v = reg.read().reshape(-1, 2)
    mag = np.hypot(v[:, 0], v[:, 1])
    shown = 0
    for i in np.nonzero(mag)[0]:
        print("|" + format(int(i), "0%db" % reg.num_qubits) + "> %.2f" % mag[i], file=out)
        shown += 1
        if limit is not None and shown >= limit:
            break
    return shown


def check_normalisation(reg, file=None):
    """testing_and_debug.c:28-37: prints the total probability with 16 decimals and returns it"""
    import sys
    total = reg.total_probability()
    print("Total Probability: %.16f" % total, file=file or sys.stdout)
    return total


def load_state_file(path):
    """Read a state file written by Register.save / qcx_state_save on the host: returns (L, M, amplitudes as an
    interleaved float64 array, memory-mapped).  Verifies magic and version; the checksum is checked by Register.load."""
    import struct
    with open(path, "rb") as f:
        hdr = f.read(64)
    magic, version, bpa, L, M, dim, checksum = struct.unpack("<8sIIiiQQ", hdr[:40])
    if magic != b"QCXSTATE" or version != 1 or bpa != 16 or dim != 1 << (L + M):
        raise ValueError(f"{path}: not a qcx state file")
    return L, M, np.memmap(path, dtype="<f8", mode="r", offset=64, shape=(2 * dim,))
