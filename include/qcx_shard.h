/*
 * qcx_shard.h -- the PER-RANK interface of libqcx.so: the same gate kernels on caller-owned device
 * memory.  One rank of a sharded register owns 2^n_local consecutive amplitudes; the index bits above n_local are the
 * rank id.  A multi-process host (one process per GPU: quantumcomputer_amd/sharded.py over torch.distributed / RCCL) calls
 * these between its exchanges; the one-process sharded register of qcx.h (qcx_register_create_sharded) uses them inside.
 * Not part of the reference-facing boundary: a program that replaces qc_shor.c's gate path includes qcx.h /
 * qcx_compat.h only.
 */
#ifndef QCX_SHARD_H
#define QCX_SHARD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

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
#endif /* QCX_SHARD_H */
