/*
 * qcx.h -- C ABI of libqcx.so: the MI355X (gfx950) state-vector gate engine that
 * replaces the gate-application path of adamalderton/QuantumComputer's
 * qc_shor.c (cited as Q:line).  Plain C, plain pointers and sizes; no GSL, no
 * torch types.  The library is HIP only: every entry point that computes
 * returns QCX_HIP_ERROR when no gfx950 device is usable -- there is no CPU
 * fallback.
 *
 * The reference has no library interface (every function is `static`, only
 * `main` is exported, Q:242-1284); the seam this ABI fills is the set of
 * gate/state functions the circuit builders call (Q:683, Q:687, Q:721, Q:729,
 * Q:922-923, Q:928) plus the allocation block in main (Q:1316-1333).
 * include/qcx_compat.h re-creates the reference's own names and signatures on
 * top of this header so that the bodies of inverse_QFT / quantum_computation
 * compile unchanged.
 *
 * Conventions kept from the reference
 *   - qubit b is bit b of the state index (LSB = qubit 0, GET_BIT Q:150-151);
 *     the M register is bits [0,M), the L register bits [M,M+L) (Q:620-652).
 *   - amplitudes are interleaved (re, im) binary64, index ascending (Q:405-406).
 *   - status values 0..4 are the reference's ErrorCode (Q:164-170).
 * Differences
 *   - gates return int status instead of void (a bad qubit index is an error
 *     here; the reference silently computes garbage);
 *   - the `gsl_spmatrix_complex *matrix` scratch argument is gone (no matrix is
 *     ever built); qcx_compat.h accepts and ignores it;
 *   - one state buffer, updated in place; swap_states is a no-op.
 * All gate calls are asynchronous on the register's HIP stream; read-back,
 * measurement and qcx_synchronize() wait.
 */
#ifndef QCX_H
#define QCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Q:164-170 ErrorCode, extended */
enum {
    QCX_NO_ERROR            = 0,
    QCX_INSUFFICIENT_MEMORY = 1,
    QCX_BAD_ARGUMENTS       = 2,
    QCX_PERIOD_NOT_FOUND    = 3,
    QCX_UNKNOWN_ERROR       = 4,
    QCX_HIP_ERROR           = 5,   /* HIP runtime / no device */
    QCX_BAD_QUBIT           = 6,   /* qubit index >= num_qubits, or control == target */
    QCX_UNSUPPORTED         = 7    /* valid request this build cannot serve */
};

typedef struct qcx_register qcx_register;   /* replaces Register, Q:194-203 */
typedef struct qcx_rng      qcx_rng;        /* replaces gsl_rng (mt19937), Q:1296-1299 */

const char *qcx_version(void);
const char *qcx_status_string(int status);
const char *qcx_last_error(void);      /* detail of the calling thread's last failure (HIP error string, file name ...) */
int  qcx_device_count(int *count);
int  qcx_set_device(int device);

/* ---- register lifecycle: replaces Q:1316-1324 / Q:1330-1332 -------------- */
int  qcx_register_create(int L_size, int M_size, qcx_register **out);
int  qcx_register_destroy(qcx_register *reg);
unsigned      qcx_num_qubits(const qcx_register *reg);   /* Register.num_qubits */
unsigned long qcx_num_states(const qcx_register *reg);   /* Register.num_states */
int  qcx_L_size(const qcx_register *reg);
int  qcx_M_size(const qcx_register *reg);
/* The same register sharded over `nshards` = 2, 4, 8 or 16 GPUs by THIS process (shard = top log2(nshards)
 * index bits).  The handle is an ordinary qcx_register: every function of this header works on it, so the reference's
 * circuit builders and main (Q:678-737, Q:1284-1347) run unchanged on the 8 GPUs of a node.  A Hadamard on a qubit held
 * in the shard id costs one exchange: each GPU's pack pass stores straight into its peers' buffers over xGMI.
 * devices[r] = HIP device of shard r (entries may repeat: several shards on one GPU).  NULL: the shards are spread over
 * the visible devices (qcx_spread_devices: 8 shards on 8 GPUs = shard r on device r, on 4 GPUs two neighbours per GPU,
 * on one GPU all on device 0).
 * Setting QCX_SHARDS=N in the environment makes qcx_register_create do this by itself (QCX_SHARD_DEVICES="0,1,..").
 * Pre-flight check: when the shards sit on more than one device, creation first trades a small register there and back
 * on the same devices and compares every amplitude, bit for bit, with what the layout says it must be; on a mismatch
 * creation fails with QCX_HIP_ERROR and qcx_last_error() names the shard (QCX_SHARD_SELFCHECK=0 skips, =1 forces the
 * check; qcx_sharded_set_relays runs it again through the relays).
 * M_size > 12 works like on one GPU (the modular multiply then runs in place through a per-device staging buffer instead of
 * LDS tiles; M_size <= 26).  Not available on a sharded register: qcx_register_set_stream, qcx_device_pointer (NULL),
 * the event pool. */
int  qcx_register_create_sharded(int L_size, int M_size, unsigned nshards, const int *devices, qcx_register **out);
int  qcx_spread_devices(unsigned nshards, int visible_devices /* <= 0: ask HIP */, int *devices_out /* [nshards] */);
int  qcx_sharded_selfcheck(qcx_register *reg);             /* the pre-flight exchange check on demand */
unsigned long qcx_sharded_selfchecks(const qcx_register *reg);   /* checks this register has passed */
unsigned qcx_register_shards(const qcx_register *reg);      /* 1 for an unsharded register */
int  qcx_sharded_stats(qcx_register *reg, unsigned long *exchanges, unsigned long *pack_passes);
/* Multi-path striping for fewer shards than GPUs on the node: the listed GPUs (which hold no shard)
 * relay a share of every chunk of every trade, so that a pair of shards exchanges over several xGMI links instead of its
 * one direct link.  nrelays = 0 turns it off (default; QCX_SHARD_RELAYS="4,5,6,7" sets it at creation).  Results are
 * the same bits either way. */
int  qcx_sharded_set_relays(qcx_register *reg, unsigned nrelays, const int *devices);
/* launch on a caller-owned hipStream_t (NULL = the register's own stream) */
int  qcx_register_set_stream(qcx_register *reg, void *hip_stream);
int  qcx_synchronize(qcx_register *reg);

/* ---- gate layer ----------------------------------------------------------- */
int  qcx_reset_register(qcx_register *reg);                                        /* Q:318-324 */
int  qcx_hadamard_gate(unsigned qubit_num, qcx_register *reg);                     /* Q:442-484 */
int  qcx_c_phase_shift_gate(unsigned c_qubit_num, unsigned qubit_num, double theta,
                            qcx_register *reg);                                    /* Q:513-565 */
int  qcx_c_amodc_gate(unsigned C, unsigned long long atox, unsigned c_qubit_num,
                      qcx_register *reg);                                          /* Q:595-660 */
int  qcx_swap_states(qcx_register *reg);                                           /* Q:242-249: no-op */
/* host-side gate schedules */
int  qcx_inverse_QFT(qcx_register *reg);                                           /* Q:678-690 */
/* atox per control qubit: intpow_mode 0 = exact a^(2^k) mod C, 1 = the
 * reference's 32-bit INT_POW(a, x) including its wrap (Q:158-159, Q:729) */
int  qcx_quantum_computation(unsigned C, unsigned a, int intpow_mode, qcx_register *reg); /* Q:712-737 */

/* e^{i theta} as the reference's gsl_complex_polar(1.0, theta) yields it under gcc -O2 + glibc: one sincos()
 * call (Q:526).  Hosts that compute the phase factor themselves (qcx_shard_phase) must use this. */
void qcx_polar(double theta, double *cos_out, double *sin_out);

/* the reference's INT_POW macro (Q:158-159) exactly as x86-64 gcc evaluates it, 32-bit wrap included */
unsigned qcx_ref_int_pow(double base, double power);

/* ---- gate fusion (no reference counterpart) ------------
 * Queued gates are executed as fused passes (one HBM round trip applies many gates to LDS-resident
 * tiles); in modes -1, 0 and 1 results are bit-identical to the per-gate kernels.  Every call that observes the state
 * flushes the queue; qcx_flush does so explicitly.  Behind reset_register + the front of quantum_computation (Hadamard layer,
 * multiply ladder) a flush may run on a COMPACT copy of the state -- only the M-register values of the ladder's orbit can hold
 * anything but +0 -- and a whole-circuit entry point may leave its result in that form: qcx_measure_state reads it there,
 * every other observer (qcx_flush included) first expands it into the register.  Same bits either way.
 *   enable =  0 (default): a gate call launches its own kernel; the whole-circuit entry points
 *                (qcx_inverse_QFT, qcx_quantum_computation) hand their complete gate list to the pass scheduler
 *   enable =  1: every gate call is queued
 *   enable = -1: strictly one kernel launch per gate, inside the whole-circuit entry points too
 *   enable =  2: TOLERANCE MODE, opt-in, NOT bit-exact: like 1, and every run of consecutive controlled phases that
 *                share a qubit (Q:682-689: all phases after H(l) share l) is merged into one diagonal -- one complex
 *                multiply per amplitude by the product of the factors of its set target bits, FMA allowed.  Amplitudes
 *                differ from the bit-exact modes by rounding only: |delta| <= 1e-14 * |amplitude| per merged diagonal
 *                (tests bound the whole n <= 16 circuits at 1e-12; north_star asks 1e-10); zero signs are not
 *                canonicalised.  On a sharded register (qcx_register_create_sharded) every shard's passes run in this mode. */
int  qcx_set_fusion(qcx_register *reg, int enable);
int  qcx_flush(qcx_register *reg);
/* passes_launched: fused passes plus circuit fronts executed (a front = the lazily pending basis state written together with
 * the closed-form prefix of the queue; also one that was generated inside its first pass); gates_fused: gates that went into
 * either.  Sharded register: passes_launched counts the fronts only (see qcx_sharded_stats), gates_fused is 0. */
int  qcx_fusion_stats(qcx_register *reg, unsigned long *passes_launched, unsigned long *gates_fused);

/* ---- measurement: Q:272-306 ----------------------------------------------- */
int  qcx_measure_state(qcx_register *reg, qcx_rng *rng, unsigned long *state_num);
int  qcx_measure_state_r(qcx_register *reg, double r, unsigned long *state_num);   /* r supplied */

/* ---- state access (replaces gsl_vector_complex_get/set uses, T:7-37) ------- */
int  qcx_state_read(qcx_register *reg, unsigned long first, unsigned long count, double *out_re_im);
int  qcx_state_write(qcx_register *reg, unsigned long first, unsigned long count, const double *in_re_im);
int  qcx_norm2(qcx_register *reg, double *total_probability);                      /* T:28-37, summed as a tree (fast) */
/* the same total with the reference's own summation order (index-ascending, one addition per amplitude): the bits
 * check_normalisation prints (T:28-37) */
int  qcx_total_probability(qcx_register *reg, double *total_probability);
/* State files (golden vectors, debugging, checkpoint): a 64-byte header ("QCXSTATE", version,
 * L, M, 2^n, FNV-1a 64 checksum) followed by the amplitudes as interleaved little-endian binary64 (re, im).  Streamed
 * in 64 MiB pieces.  Load requires a register of the same L and M and verifies the checksum. */
int  qcx_state_save(qcx_register *reg, const char *path);
int  qcx_state_load(qcx_register *reg, const char *path);
/* The amplitude buffer in HBM (for interop).  Flushes first.  Its contents are the register's state only after qcx_flush,
 * qcx_synchronize or a fresh call of this function: reset_register and the collapse of measure_state are lazy, and fused
 * passes may alternate between two buffers.  Calling it pins the state to the returned buffer from then on (the register
 * stops chaining passes through its second buffer), so the pointer stays valid until the register is destroyed.  NULL for a
 * sharded register. */
void *qcx_device_pointer(qcx_register *reg);
/* synthetic input for benches and full-size tests: component k (k = 2*index + {0 re, 1 im}) is
 * ((splitmix64(seed + k) >> 11) * 2^-53 - 0.5) * sqrt(6 / 2^n); generated on the device */
int  qcx_state_fill_random(qcx_register *reg, uint64_t seed);

/* ---- HIP-event timing on the register's stream (bench / roofline) ---------- */
int  qcx_timer_start(qcx_register *reg);
int  qcx_timer_stop(qcx_register *reg, double *milliseconds);    /* waits for the stop event */
/* a pool of events: record between gates inside a timed region, read the differences afterwards */
int  qcx_events_create(qcx_register *reg, unsigned count);
int  qcx_event_record(qcx_register *reg, unsigned slot);
int  qcx_event_elapsed(qcx_register *reg, unsigned from_slot, unsigned to_slot, double *milliseconds);

/* ---- MT19937 with gsl_rng_mt19937 semantics (Q:1296-1299, Q:281) ----------- */
qcx_rng      *qcx_rng_alloc(void);
void          qcx_rng_set(qcx_rng *rng, unsigned long seed);    /* seed 0 -> 4357 like GSL */
unsigned long qcx_rng_get(qcx_rng *rng);
double        qcx_rng_uniform(qcx_rng *rng);                    /* get / 2^32 */
void          qcx_rng_free(qcx_rng *rng);

/* ---- beyond the boundary ---------------------------------------------------
 * Two further interfaces of the same library live in headers of their own and are included here for convenience:
 *   qcx_shard.h   the per-rank (shard-level) entry points on caller-owned device memory: what a one-process-per-GPU host
 *                 calls between its exchanges (quantumcomputer_amd/sharded.py), plus the compact-circuit helpers;
 *   qcx_plan.h    the fused-pass planner alone, on the host: a test and tooling interface (tests/fuse_emulator.py).
 * A program that replaces the reference's gate path needs neither. */
#ifdef __cplusplus
}
#endif
#include "qcx_shard.h"
#include "qcx_plan.h"
#endif /* QCX_H */
