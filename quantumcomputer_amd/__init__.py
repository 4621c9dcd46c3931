"""quantumcomputer_amd -- MI355X-native gate engine for the qc_shor.c hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI, built into libqcx.so),
the ctypes loader and the host-side mirror of the reference's gate interface.
"""
from ._lib import LIB_PATH, QcxError, front_plan, fusion_plan, idle_devices, lib, polar, spread_devices, tune  # noqa: F401
from .register import (Register, Rng, load_state_file, c_amodc_gate, c_phase_shift_gate, check_normalisation,  # noqa: F401
                       display_state, hadamard_gate,
                       inverse_QFT, measure_state, quantum_computation, read_omega,
                       reset_register, swap_states)

FUSION_TOLERANCE = 2      # qcx_set_fusion(reg, 2): the opt-in tolerance mode (include/qcx.h)

__all__ = ["FUSION_TOLERANCE", "Register", "Rng", "reset_register", "hadamard_gate", "c_phase_shift_gate", "c_amodc_gate",
           "swap_states", "inverse_QFT", "quantum_computation", "measure_state", "read_omega",
           "display_state", "check_normalisation", "lib", "tune", "polar", "QcxError", "LIB_PATH", "spread_devices", "idle_devices"]
