"""ctypes loader for libqcx.so (include/qcx.h).

The HIP library is the product: if it is missing this module raises -- there is
no CPU fallback anywhere in quantumcomputer_amd.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QCX_LIB", os.path.join(_HERE, "libqcx.so"))      # (override: sanitizer builds of the host code, test rigs)

# status codes (include/qcx.h; 0..4 are the reference's ErrorCode, qc_shor.c:164-170)
NO_ERROR, INSUFFICIENT_MEMORY, BAD_ARGUMENTS, PERIOD_NOT_FOUND, UNKNOWN_ERROR = range(5)
HIP_ERROR, BAD_QUBIT, UNSUPPORTED = 5, 6, 7


class QcxError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib().qcx_status_string(status).decode()
        detail = lib().qcx_last_error().decode()
        super().__init__(f"{where}: {msg} ({status})" + (f" [{detail}]" if detail else ""))


# every symbol include/qcx.h declares, with its ctypes signature
_u, _ul, _ull, _i, _d, _p = C.c_uint, C.c_ulong, C.c_ulonglong, C.c_int, C.c_double, C.c_void_p
_u64, _i64 = C.c_uint64, C.c_int64
SIGNATURES = {
    "qcx_version": (C.c_char_p, []),
    "qcx_status_string": (C.c_char_p, [_i]),
    "qcx_last_error": (C.c_char_p, []),
    "qcx_device_count": (_i, [C.POINTER(_i)]),
    "qcx_set_device": (_i, [_i]),
    "qcx_register_create": (_i, [_i, _i, C.POINTER(_p)]),
    "qcx_register_destroy": (_i, [_p]),
    "qcx_register_create_sharded": (_i, [_i, _i, _u, C.POINTER(_i), C.POINTER(_p)]),
    "qcx_spread_devices": (_i, [_u, _i, C.POINTER(_i)]),
    "qcx_sharded_selfcheck": (_i, [_p]),
    "qcx_sharded_selfchecks": (_ul, [_p]),
    "qcx_register_shards": (_u, [_p]),
    "qcx_sharded_stats": (_i, [_p, C.POINTER(_ul), C.POINTER(_ul)]),
    "qcx_sharded_set_relays": (_i, [_p, _u, C.POINTER(_i)]),
    "qcx_num_qubits": (_u, [_p]),
    "qcx_num_states": (_ul, [_p]),
    "qcx_L_size": (_i, [_p]),
    "qcx_M_size": (_i, [_p]),
    "qcx_register_set_stream": (_i, [_p, _p]),
    "qcx_synchronize": (_i, [_p]),
    "qcx_reset_register": (_i, [_p]),
    "qcx_hadamard_gate": (_i, [_u, _p]),
    "qcx_c_phase_shift_gate": (_i, [_u, _u, _d, _p]),
    "qcx_c_amodc_gate": (_i, [_u, _ull, _u, _p]),
    "qcx_swap_states": (_i, [_p]),
    "qcx_inverse_QFT": (_i, [_p]),
    "qcx_quantum_computation": (_i, [_u, _u, _i, _p]),
    "qcx_ref_int_pow": (_u, [_d, _d]),
    "qcx_polar": (None, [_d, C.POINTER(_d), C.POINTER(_d)]),
    "qcx_set_fusion": (_i, [_p, _i]),
    "qcx_flush": (_i, [_p]),
    "qcx_fusion_stats": (_i, [_p, C.POINTER(_ul), C.POINTER(_ul)]),
    "qcx_measure_state": (_i, [_p, _p, C.POINTER(_ul)]),
    "qcx_measure_state_r": (_i, [_p, _d, C.POINTER(_ul)]),
    "qcx_state_read": (_i, [_p, _ul, _ul, _p]),
    "qcx_state_write": (_i, [_p, _ul, _ul, _p]),
    "qcx_norm2": (_i, [_p, C.POINTER(_d)]),
    "qcx_total_probability": (_i, [_p, C.POINTER(_d)]),
    "qcx_device_pointer": (_p, [_p]),
    "qcx_state_fill_random": (_i, [_p, _u64]),
    "qcx_shard_fill_random": (_i, [_p, _u, _u64, _u64, _d, _p]),
    "qcx_timer_start": (_i, [_p]),
    "qcx_timer_stop": (_i, [_p, C.POINTER(_d)]),
    "qcx_events_create": (_i, [_p, _u]),
    "qcx_event_record": (_i, [_p, _u]),
    "qcx_event_elapsed": (_i, [_p, _u, _u, C.POINTER(_d)]),
    "qcx_rng_alloc": (_p, []),
    "qcx_rng_set": (None, [_p, _ul]),
    "qcx_rng_get": (_ul, [_p]),
    "qcx_rng_uniform": (_d, [_p]),
    "qcx_rng_free": (None, [_p]),
    "qcx_shard_reset": (_i, [_p, _u, _i, _p]),
    "qcx_shard_hadamard": (_i, [_p, _u, _u, _p]),
    "qcx_shard_phase": (_i, [_p, _u, _u64, _d, _d, _p]),
    "qcx_shard_camodc": (_i, [_p, _u, _u, _u, _u, _i, _p]),
    "qcx_shard_swap_bits": (_i, [_p, _p, _u, _u, C.POINTER(_u), C.POINTER(_u), _p]),
    "qcx_shard_norm2": (_i, [_p, _u, C.POINTER(_d), _p]),
    "qcx_shard_run_fused": (_i, [_p, _u, _u, _u, _p, _p]),
    "qcx_shard_run_fused_mode": (_i, [_i, _p, _u, _u, _u, _p, _p]),
    "qcx_shard_release_stream": (_i, [_p]),
    "qcx_shard_basis_front": (_i, [_p, _u, _u64, _u, _u, _u64, _u, _p, C.POINTER(_u), _p]),
    "qcx_compact_plan": (_i, [_u, _u, _u64, _u, _p, C.POINTER(_u), C.POINTER(_u), C.POINTER(_u), C.POINTER(C.c_uint16)]),
    "qcx_shard_compact_front": (_i, [_p, _u, _u64, _u, _u, _u64, _u, _p, _u, _u, C.POINTER(C.c_uint16), _p]),
    "qcx_shard_expand_compact": (_i, [_p, _p, _u, _u, _u, _u, C.POINTER(C.c_uint16), _p]),
    "qcx_state_save": (_i, [_p, C.c_char_p]),
    "qcx_state_load": (_i, [_p, C.c_char_p]),
    "qcx_fusion_plan": (_i, [_u, _u, _u, _p, _p, _u, C.POINTER(_u), _p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "qcx_fusion_plan_mode": (_i, [_i, _u, _u, _u, _p, _p, _u, C.POINTER(_u), _p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "qcx_shard_measure_scan": (_i, [_p, _u, _u64, _u64, _d, _d, C.POINTER(_i), C.POINTER(_u64), C.POINTER(_d), _p]),
    "qcx_shard_collapse": (_i, [_p, _u, _i64, _p]),
    "qcx_shard_canon_zeros": (_i, [_p, _u, _p]),
}
# not in the public header: diagnostics / tuning hooks
_EXTRA = {
    "qcx_tune_set": (_i, [C.c_char_p, C.c_long]),
    "qcx_tune_get": (C.c_long, [C.c_char_p]),
    "qcx_measure_last_stats": (_i, [C.POINTER(_u), C.POINTER(_u)]),
    "qcx_sharded_trace": (_i, [_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "qcx_sharded_restore_identity": (_i, [_p]),
    "qcx_sharded_relay_stats": (_i, [_p, C.POINTER(_u), C.POINTER(_ul)]),
    "qcx_sharded_overlap_stats": (_i, [_p, C.POINTER(_u), C.POINTER(_ul)]),
    "qcx_sharded_layout": (_i, [_p, C.POINTER(_u), _u]),
    "qcx_front_plan": (_i, [_u, _u, _u64, _u, _p, C.POINTER(_u), _p, C.c_size_t]),
    "qcx_chain_stats": (_i, [_p, C.POINTER(_ul)]),
    "qcx_gen_stats": (_i, [_p, C.POINTER(_ul)]),
    "qcx_gen_cols_stats": (_i, [_p, C.POINTER(_ul)]),
    "qcx_compact_stats": (_i, [_p, C.POINTER(_ul)]),
    "qcx_compact_measure_stats": (_i, [_p, C.POINTER(_ul)]),
    "qcx_expanding_store_stats": (_i, [_p, C.POINTER(_ul)]),
    "qcx_plan_cache_stats": (_i, [_p, C.POINTER(_ul)]),
    "qcx_expand_store_plan": (_i, [_u, _p, _p, _u, _u, _p, _p, _p, C.POINTER(_i)]),
}


class BasisFront(C.Structure):
    """struct BasisFront (csrc/qcx_kernels.h): the parameters of k_basis_front"""
    _fields_ = [("first", C.c_uint64), ("basis", C.c_uint64), ("hmask", C.c_uint64), ("fixed_mask", C.c_uint64), ("sign_mask", C.c_uint64),
                ("v", C.c_double), ("M", C.c_uint), ("ncam", C.c_uint), ("C", C.c_uint32 * 64), ("A", C.c_uint32 * 64),
                ("ctl", C.c_uint8 * 64)]


def front_plan(n_local, M, basis, descs):
    """host half of the basis-state front (no GPU): (gates consumed, BasisFront)"""
    arr = (GateDesc * max(len(descs), 1))()
    for i, d in enumerate(descs):
        arr[i].type, arr[i].q, arr[i].mask, arr[i].c, arr[i].s, arr[i].C, arr[i].A = d
    used, B = C.c_uint(0), BasisFront()
    check(lib().qcx_front_plan(n_local, M, basis, len(descs), C.cast(arr, C.c_void_p), C.byref(used), C.byref(B), C.sizeof(B)), "qcx_front_plan")
    return used.value, B

class GateDesc(C.Structure):
    """qcx_gate_desc (include/qcx.h)"""
    _fields_ = [("type", C.c_uint32), ("q", C.c_uint32), ("mask", C.c_uint64), ("c", C.c_double), ("s", C.c_double),
                ("C", C.c_uint32), ("A", C.c_uint32)]


class FuseRecord(C.Structure):
    """qcx_fuse_record (include/qcx.h): one 32-byte record of a fused pass"""
    _fields_ = [("type", C.c_uint32), ("a", C.c_uint32), ("mask", C.c_uint64), ("c", C.c_double), ("s", C.c_double)]


class PlanAction(C.Structure):
    """qcx_plan_action (include/qcx.h)"""
    _fields_ = [("fused", C.c_int), ("first_gate", C.c_uint), ("ngates", C.c_uint), ("T", C.c_uint), ("c", C.c_uint),
                ("nh", C.c_uint), ("hbit", C.c_ubyte * 16), ("nopipe", C.c_uint), ("rounds_form", C.c_uint),
                ("rec_off", C.c_size_t), ("rec_cnt", C.c_size_t), ("nops", C.c_uint), ("table_bytes", C.c_uint),
                ("table_rec_off", C.c_uint), ("diag_cnt", C.c_uint), ("diag_rec_off", C.c_uint),
                ("chained", C.c_uint), ("tl", C.c_ubyte * 16), ("in_pos", C.c_ubyte * 16), ("st_loc", C.c_ubyte * 16),
                ("st_pos", C.c_ubyte * 16), ("nseg_in", C.c_ubyte), ("nseg_out", C.c_ubyte), ("nseg_lg", C.c_ubyte), ("pad_", C.c_ubyte),
                ("seg_in", C.c_ubyte * 64), ("seg_out", C.c_ubyte * 64), ("seg_lg", C.c_ubyte * 64)]      # segments: (src, dst, len, pad) x 16


def fusion_plan(n_local, M, descs, mode=1):
    """The pass planner alone (host code, no GPU needed).  descs: (type, q, mask, c, s, C, A) tuples as for
    qcx_shard_run_fused.  Returns (actions, records): a list of PlanAction and a ctypes array of FuseRecord.
    mode 1: the bit-exact plan; 2: the tolerance mode's plan (merged diagonals); | 4: with chained passes (PlanAction.chained)."""
    arr = (GateDesc * max(len(descs), 1))()
    for i, d in enumerate(descs):
        arr[i].type, arr[i].q, arr[i].mask, arr[i].c, arr[i].s, arr[i].C, arr[i].A = d
    na, nr = C.c_uint(0), C.c_size_t(0)
    cap_a, cap_r = len(descs) + 4, 4 * len(descs) + 64
    while True:
        acts = (PlanAction * cap_a)()
        recs = (FuseRecord * cap_r)()
        st = lib().qcx_fusion_plan_mode(mode, n_local, M, len(descs), C.cast(arr, C.c_void_p), C.cast(acts, C.c_void_p), cap_a, C.byref(na),
                                   C.cast(recs, C.c_void_p), cap_r, C.byref(nr))
        if st == 1 and (na.value > cap_a or nr.value > cap_r):        # QCX_INSUFFICIENT_MEMORY: grow and retry
            cap_a, cap_r = max(cap_a, na.value), max(cap_r, nr.value)
            continue
        check(st, "qcx_fusion_plan_mode")
        return [acts[k] for k in range(na.value)], recs, nr.value


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The torch wheel bundles its own libamdhip64.so (same SONAME as
    /opt/rocm's, different file); if libqcx.so pulled in the system copy first and torch its bundled
    copy later, the process would hold two HSA runtimes and the second to initialise sees no device.
    Loading torch's copy first (without importing torch) makes both resolve to that one file."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                      # torch already loaded its runtime; libqcx.so resolves to it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib():
    """Load libqcx.so once; raise loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C quantumcomputer_amd/csrc` "
                "(or __graft_entry__.build()); quantumcomputer_amd has no CPU fallback")
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in {**SIGNATURES, **_EXTRA}.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, where=""):
    if status != NO_ERROR:
        raise QcxError(status, where)


def spread_devices(shards, visible=None):
    """HIP device of each shard, as qcx_register_create_sharded(devices=NULL) places them: over the largest
    power-of-two number of devices <= min(shards, visible), neighbouring shards together; one visible GPU gives
    [0] * shards (the virtual shards of the one-GPU tests).  visible=None asks HIP (needs a GPU); an explicit count is
    pure arithmetic."""
    out = (C.c_int * int(shards))()
    check(lib().qcx_spread_devices(int(shards), 0 if visible is None else int(visible), out), "qcx_spread_devices")
    return list(out)


def idle_devices(shard_devices, count, visible=None):
    """`count` relay GPUs for multi-path striping: devices that hold no shard first, then (when there are not enough
    idle ones, e.g. on a one-GPU box) the shard devices again"""
    if visible is None:
        n = C.c_int(0)
        check(lib().qcx_device_count(C.byref(n)), "qcx_device_count")
        visible = n.value
    idle = [d for d in range(visible) if d not in set(shard_devices)]
    pool = idle + sorted(set(shard_devices))
    return [pool[i % len(pool)] for i in range(count)]


def polar(theta):
    """(cos, sin) of theta exactly as the gate path computes them (glibc sincos, include/qcx.h: qcx_polar)"""
    c, s = C.c_double(0.0), C.c_double(0.0)
    lib().qcx_polar(float(theta), C.byref(c), C.byref(s))
    return c.value, s.value


def tune(**kv):
    for k, v in kv.items():
        check(lib().qcx_tune_set(k.encode(), int(v)), f"tune {k}")
