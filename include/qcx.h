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

/* ---- shard-level entry points on caller-owned device memory ----------------
 * One rank of a sharded register owns 2^n_local consecutive amplitudes; the
 * index bits above n_local are the rank id.  These are what a
 * multi-process host (one process per GPU) calls between its exchanges.
 * `stream` is a hipStream_t (NULL = default stream).                          */
int  qcx_shard_reset(void *amp, unsigned n_local, int holds_index_one, void *stream);
int  qcx_shard_fill_random(void *amp, unsigned n_local, uint64_t first_global, uint64_t seed,
                           double scale, void *stream);
int  qcx_shard_hadamard(void *amp, unsigned n_local, unsigned q_local, void *stream);
/* multiply by (cos_t + i sin_t) every amplitude whose local index has all bits of
 * `mask_local` set; mask_local has 0, 1 or 2 bits (global control bits that are 1
 * simply drop out of the mask; a global bit that is 0 means: do not call). */
int  qcx_shard_phase(void *amp, unsigned n_local, uint64_t mask_local, double cos_t, double sin_t, void *stream);
/* ctl_local < 0: the control is a global bit whose value on this rank is 1 */
int  qcx_shard_camodc(void *amp, unsigned n_local, unsigned M, unsigned C, unsigned A,
                      int ctl_local, void *stream);
/* dst[j] = src[j with index bits pos_a[m] <-> pos_b[m] exchanged, m < npairs <= 8]; out of place.
 * The pack pass of the sharded qubit remap (brings the bits to be traded with the rank id to the top). */
int  qcx_shard_swap_bits(const void *src, void *dst, unsigned n_local, unsigned npairs,
                         const unsigned *pos_a, const unsigned *pos_b, void *stream);
int  qcx_shard_norm2(const void *amp, unsigned n_local, double *out, void *stream);
/* a list of gates on a shard, executed through the fusion scheduler (fused LDS-tile passes, same bits as the
 * one-by-one entry points).  All bit positions are LOCAL index bits of the shard. */
typedef struct {
    uint32_t type;      /* 0: Hadamard, 1: phase, 2: controlled modular multiply */
    uint32_t q;         /* Hadamard: target bit;  modular multiply: control bit, or 0xffffffff = always on */
    uint64_t mask;      /* phase: local bits that must all be 1 (0 = every amplitude of the shard) */
    double   c, s;      /* phase: cos, sin (qcx_polar) */
    uint32_t C, A;      /* modular multiply: modulus and multiplier (A < C) */
} qcx_gate_desc;
int  qcx_shard_run_fused(void *amp, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates, void *stream);
/* the same in a fusion mode: 1 = bit-exact (qcx_shard_run_fused), 2 = tolerance mode (merged diagonals, see qcx_set_fusion) */
int  qcx_shard_run_fused_mode(int mode, void *amp, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates, void *stream);
/* This shard's part of the basis state |basis> of an (n, M) register -- amplitudes [first_global, first_global + 2^n_local)
 * -- written together with the longest prefix of `gates` that has a closed form on a basis state: Hadamards on distinct
 * qubits, then controlled modular multiplies (the front of Q:712-737).  Qubit numbers in `gates` are GLOBAL (identity
 * layout); a Hadamard or a control on a shard-id qubit costs nothing.  Every rank calls it with the same list and gets
 * the same *used (gates consumed; 0 = the plain basis state was written).  Replaces reset (Q:318-324) + those gates. */
int  qcx_shard_basis_front(void *amp, unsigned n_local, uint64_t first_global, unsigned n, unsigned M, uint64_t basis,
                           unsigned count, const qcx_gate_desc *gates, unsigned *used, void *stream);

/* qcx_shard_run_fused keeps record buffers per (device, stream); call this before destroying a stream it was used on */
int  qcx_shard_release_stream(void *stream);

/* The pass planner alone, on the host (no GPU needed; test and tooling interface).  Cuts a gate list into actions --
 * fused passes over LDS tiles, or single gates that run as their stand-alone kernel -- and returns the records the
 * pass kernels interpret, 32 bytes each, all passes back to back.  Record formats: csrc/qcx_kernels.h (FuseOp and
 * the ROUNDS form); tests/fuse_emulator.py interprets them on the CPU and compares with the oracle.
 * Returns QCX_INSUFFICIENT_MEMORY (with the needed counts in n_actions / n_records) when an array is too small. */
typedef struct {
    uint32_t type, a;
    uint64_t mask;
    double   c, s;
} qcx_fuse_record;
typedef struct {
    int      fused;                 /* 0: the single gate first_gate, stand-alone kernel; 1: one fused pass */
    unsigned first_gate, ngates;    /* the gates of the list this action covers (in order, no gaps between actions) */
    unsigned T, c, nh;              /* tile = 2^T amplitudes: the c lowest index bits + nh higher bits hbit[0..nh) */
    unsigned char hbit[16];
    unsigned nopipe;                /* 1: phase-dominated pass, planned on the smaller tile (fuse_T_phase) */
    unsigned rounds_form;           /* 1: records in ROUNDS form (rounds / items / runs), 0: plain gate list */
    size_t   rec_off, rec_cnt;      /* this pass's records (tables included) inside `records` */
    unsigned nops;                  /* records the kernel walks (the rest, if any, are tables) */
    unsigned table_bytes, table_rec_off;    /* folded modular-multiply tables: size, and offset in records from rec_off */
    unsigned diag_cnt, diag_rec_off;        /* tolerance mode: merged diagonals of the pass, record offset of their table area */
    /* tile addressing (round 4).  tl[j] = the qubit that is tile-local bit j in the records.  A CHAINED pass (chained = 1) reads
     * the register's current buffer under one logical -> physical layout and writes the other buffer under another one: tile-local
     * bit j is input index bit in_pos[j]; the j-th lowest output position of the tile's bits is output index bit st_pos[j] and
     * belongs to tile-local bit st_loc[j]; bits [src, src + len) of the tile number go to index bits [dst, dst + len) of the
     * input / output / logical index (seg_in / seg_out / seg_lg).  In place (chained = 0): out = in = logical, seg_in only. */
    unsigned chained;
    unsigned char tl[16], in_pos[16], st_loc[16], st_pos[16];
    unsigned char nseg_in, nseg_out, nseg_lg, pad_;
    struct { unsigned char src, dst, len, pad; } seg_in[16], seg_out[16], seg_lg[16];
} qcx_plan_action;
int  qcx_fusion_plan(unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates,
                     qcx_plan_action *actions, unsigned max_actions, unsigned *n_actions,
                     qcx_fuse_record *records, size_t max_records, size_t *n_records);
/* the same for a fusion mode: 1 = the bit-exact plan (what qcx_fusion_plan returns), 2 = the tolerance mode's plan;
 * | 4: as a register with a second buffer plans (runs of passes chained through it, see qcx_plan_action) */
int  qcx_fusion_plan_mode(int mode, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates,
                          qcx_plan_action *actions, unsigned max_actions, unsigned *n_actions,
                          qcx_fuse_record *records, size_t max_records, size_t *n_records);
/* sequential cumulative scan of |amp|^2 over this shard continuing from cum_in
 * (global index of local 0 = first_global; indices >= last_excluded are not
 * examined, Q:283).  Synchronous. */
int  qcx_shard_measure_scan(const void *amp, unsigned n_local, uint64_t first_global,
                            uint64_t last_excluded, double cum_in, double r,
                            int *found, uint64_t *index, double *cum_out, void *stream);
/* Compact circuits for a one-process-per-GPU host: behind the circuit front the M register reads one of the
 * residues of the multiply ladder's orbit; when nothing else in the queue touches it, the queue can run on a register of
 * L + cb qubits, [L register][orbit column], and every rank expands its part at the end.
 * qcx_compact_plan (pure host; same answer on every rank): *used = gates of the closed-form front, *ncols > 0: the compact form
 * exists, cb column bits, orbit16[0 .. *ncols) the populated M-register values ascending.
 * qcx_shard_compact_front writes a rank's part of the front in the compact form (n_local_compact = n_local - M + cb;
 * first_global = REAL global index of the rank's amplitude 0); qcx_shard_expand_compact turns a rank's compact part into its
 * part of the real register (n_local - M >= 6). */
int  qcx_compact_plan(unsigned n, unsigned M, uint64_t basis, unsigned count, const qcx_gate_desc *gates,
                      unsigned *used, unsigned *cb, unsigned *ncols, uint16_t *orbit16);
int  qcx_shard_compact_front(void *compact, unsigned n_local_compact, uint64_t first_global, unsigned n, unsigned M, uint64_t basis,
                             unsigned count, const qcx_gate_desc *gates, unsigned cb, unsigned ncols, const uint16_t *orbit16, void *stream);
int  qcx_shard_expand_compact(const void *compact, void *real, unsigned n_local, unsigned M, unsigned cb, unsigned ncols,
                              const uint16_t *orbit16, void *stream);
/* zero the shard; if 0 <= local_index < 2^n_local set that amplitude to (1,0) */
int  qcx_shard_collapse(void *amp, unsigned n_local, int64_t local_index, void *stream);
/* -0 components of the shard become +0: for amplitudes the caller wrote, before the first gate runs on them (the
 * reference's mat-vec canonicalises every amplitude at every gate, Q:393-413; the gate kernels only those they act on) */
int  qcx_shard_canon_zeros(void *amp, unsigned n_local, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QCX_H */
