/*
 * qcx_compat.h -- the reference's OWN names and signatures (qc_shor.c) on top of
 * libqcx.so, so that circuit-building code written against the reference --
 * the bodies of inverse_QFT (Q:678-690), quantum_computation (Q:712-737),
 * find_period (Q:912-964) -- compiles unchanged.  Header-only shims.
 *
 *   reference                                             here
 *   Register (Q:194-203)                                  the same eight fields
 *   gsl_vector_complex (state_a / state_b, Q:1316-1318)   a host object that owns the device register (created on first use)
 *   gsl_spmatrix_complex *matrix (scratch, Q:1320)        a token: accepted, ignored (may be NULL)
 *   gsl_rng * (mt19937, Q:1296-1299)                      alias of qcx_rng; gsl_rng_alloc/_set/_uniform/_free provided
 *   operate_matrix(matrix, reg) (Q:370)                   no-op (every gate leaves its result in the current state)
 * so main (Q:1284-1347) compiles as written, too.
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

/* ---- stand-ins for the GSL objects the reference's main and signatures name (Q:1286-1333) ------------------------
 * They carry no GSL semantics -- no matrix is built, the state lives in HBM -- they only let the reference's own lines
 * compile unchanged:
 *   gsl_spmatrix_complex   a token: gsl_spmatrix_complex_alloc_nzmax gives a non-NULL pointer (ALLOC_CHECK passes),
 *                          every gate accepts and ignores it, gsl_spmatrix_complex_free releases it;
 *   gsl_vector_complex     a small host object that OWNS the device register: gsl_vector_complex_alloc(num_states)
 *                          records the size, the device register is created at the first gate / reset through
 *                          *reg.current_state with the Register's L_size and M_size (the vector the reference calls
 *                          state_b is never touched: one in-place buffer, swap_states is a no-op, so it costs nothing);
 *   gsl_rng, gsl_rng_type, gsl_rng_mt19937, gsl_rng_alloc/_set/_uniform/_free     the MT19937 of libqcx. */
typedef struct gsl_spmatrix_complex { int unused; } gsl_spmatrix_complex;
#define GSL_SPMATRIX_COO 0
static inline gsl_spmatrix_complex *gsl_spmatrix_complex_alloc_nzmax(size_t n1, size_t n2, size_t nzmax, int sptype)
{
    (void)n1; (void)n2; (void)nzmax; (void)sptype;
    return (gsl_spmatrix_complex *)calloc(1, sizeof(gsl_spmatrix_complex));
}
static inline void gsl_spmatrix_complex_free(gsl_spmatrix_complex *m) { free(m); }

typedef struct gsl_vector_complex { size_t size; qcx_register *device; } gsl_vector_complex;
static inline gsl_vector_complex *gsl_vector_complex_alloc(size_t n)
{
    gsl_vector_complex *v = (gsl_vector_complex *)calloc(1, sizeof(gsl_vector_complex));
    if (v) v->size = n;
    return v;
}
static inline void gsl_vector_complex_free(gsl_vector_complex *v)
{
    if (v) { if (v->device) qcx_register_destroy(v->device); free(v); }
}

typedef qcx_rng gsl_rng;
typedef struct gsl_rng_type { const char *name; } gsl_rng_type;
static const gsl_rng_type qcx_compat_rng_mt19937 = { "mt19937" };
#define gsl_rng_mt19937 (&qcx_compat_rng_mt19937)
static inline gsl_rng *gsl_rng_alloc(const gsl_rng_type *type) { (void)type; return qcx_rng_alloc(); }       /* Q:1297 */
static inline void gsl_rng_set(gsl_rng *rng, unsigned long int seed) { qcx_rng_set(rng, seed); }              /* Q:1299 */
static inline double gsl_rng_uniform(gsl_rng *rng) { return qcx_rng_uniform(rng); }                           /* Q:281 */
static inline unsigned long int gsl_rng_get(gsl_rng *rng) { return qcx_rng_get(rng); }
static inline void gsl_rng_free(gsl_rng *rng) { qcx_rng_free(rng); }                                          /* Q:1333 */

typedef enum {                                              /* Q:164-170 */
    NO_ERROR = 0,
    INSUFFICIENT_MEMORY,
    BAD_ARGUMENTS,
    PERIOD_NOT_FOUND,
    UNKNOWN_ERROR,
} ErrorCode;

typedef struct {                                            /* Q:194-203, field for field */
    int L_size;
    int M_size;
    unsigned int num_qubits;
    unsigned long int num_states;
    gsl_vector_complex **current_state;     /* *current_state owns the device register (see above) */
    gsl_vector_complex **new_state;
    gsl_vector_complex *state_a;
    gsl_vector_complex *state_b;
} Register;

/* Q:150-151, Q:158-159 (INT_POW keeps the reference's 32-bit behaviour) */
#define GET_BIT(integer, n) ( ((integer) >> (n)) & 1 )
#define INT_POW(base, power) ( qcx_ref_int_pow((double)(base), (double)(power)) )

static inline void qcx_compat_die(int status, const char *what)
{
    if (status != QCX_NO_ERROR) {
        fprintf(stderr, "qcx: %s failed: %s (%s)\n", what, qcx_status_string(status), qcx_last_error());
        abort();
    }
}

/* The device register behind a Register: created on first use from L_size / M_size (the reference's main allocates
 * its vectors from num_states alone, Q:1316-1318, before any gate runs).  The reference's own circuit builders
 * (Q:678-690, Q:712-737) call the gate functions one by one; those calls are queued and run as fused passes: the
 * amplitudes are the same bits as with one kernel launch per gate and every call that looks at the state
 * (measure_state, state reads) flushes the queue first, so nothing else changes -- except the speed (n = 30 Shor circuit:
 * 0.74 s -> 0.07 s).  QCX_COMPAT_FUSION=-1|0|1|2 overrides (qcx_set_fusion). */
static inline qcx_register *qcx_compat_handle(const Register *reg)
{
    gsl_vector_complex *v = (reg->current_state && *reg->current_state) ? *reg->current_state : NULL;
    if (!v) { fprintf(stderr, "qcx: the Register has no state vector (gsl_vector_complex_alloc / register_alloc first)\n"); abort(); }
    if (!v->device) {
        const char *e = getenv("QCX_COMPAT_FUSION");
        qcx_compat_die(qcx_register_create(reg->L_size, reg->M_size, &v->device), "register allocation on the GPU");
        if (qcx_num_states(v->device) != (unsigned long)v->size) {
            fprintf(stderr, "qcx: state vector of %lu amplitudes for a register of L = %d, M = %d\n", (unsigned long)v->size, reg->L_size, reg->M_size);
            abort();
        }
        qcx_set_fusion(v->device, e ? atoi(e) : 1);
    } else if (qcx_L_size(v->device) != reg->L_size || qcx_M_size(v->device) != reg->M_size) {
        fprintf(stderr, "qcx: register sizes changed after the state vector was allocated\n");
        abort();
    }
    return v->device;
}

/* the allocation / free blocks of main (Q:1316-1324, Q:1330-1332) in one call each, for hosts that do not keep the
 * reference's own lines; fills num_qubits / num_states from L_size / M_size and creates the device register now */
static inline ErrorCode register_alloc(Register *reg)
{
    if (reg->L_size < 0 || reg->M_size < 0 || reg->L_size + reg->M_size < 1 || reg->L_size + reg->M_size > 40) return BAD_ARGUMENTS;
    reg->num_qubits = (unsigned)(reg->L_size + reg->M_size);
    reg->num_states = 1ul << reg->num_qubits;
    reg->state_a = gsl_vector_complex_alloc(reg->num_states);
    reg->state_b = NULL;
    if (!reg->state_a) return INSUFFICIENT_MEMORY;
    reg->current_state = &reg->state_a;
    reg->new_state = &reg->state_b;
    {
        const char *e = getenv("QCX_COMPAT_FUSION");
        int s = qcx_register_create(reg->L_size, reg->M_size, &reg->state_a->device);
        if (s != QCX_NO_ERROR) { gsl_vector_complex_free(reg->state_a); reg->state_a = NULL; }
        if (s == QCX_INSUFFICIENT_MEMORY) return INSUFFICIENT_MEMORY;
        if (s == QCX_BAD_ARGUMENTS) return BAD_ARGUMENTS;
        if (s != QCX_NO_ERROR) return UNKNOWN_ERROR;
        qcx_set_fusion(reg->state_a->device, e ? atoi(e) : 1);
    }
    return NO_ERROR;
}
static inline void register_free(Register *reg)
{
    gsl_vector_complex_free(reg->state_a); gsl_vector_complex_free(reg->state_b);
    reg->state_a = reg->state_b = NULL;
}

static inline void swap_states(Register *reg) { qcx_compat_die(qcx_swap_states(qcx_compat_handle(reg)), "swap_states"); }
static inline void reset_register(Register reg) { qcx_compat_die(qcx_reset_register(qcx_compat_handle(&reg)), "reset_register"); }
static inline unsigned long int measure_state(Register reg, gsl_rng *rng)
{
    unsigned long idx = 0;
    qcx_compat_die(qcx_measure_state(qcx_compat_handle(&reg), rng, &idx), "measure_state");
    return idx;
}
/* Q:370-420: the gates here leave their result in the current state themselves; nothing is left to apply */
static inline void operate_matrix(gsl_spmatrix_complex *matrix, Register *reg) { (void)matrix; (void)reg; }
static inline void hadamard_gate(unsigned int qubit_num, Register *reg, gsl_spmatrix_complex *matrix)
{
    (void)matrix;
    qcx_compat_die(qcx_hadamard_gate(qubit_num, qcx_compat_handle(reg)), "hadamard_gate");
}
static inline void c_phase_shift_gate(unsigned int c_qubit_num, unsigned int qubit_num, double theta,
                                      Register *reg, gsl_spmatrix_complex *matrix)
{
    (void)matrix;
    qcx_compat_die(qcx_c_phase_shift_gate(c_qubit_num, qubit_num, theta, qcx_compat_handle(reg)), "c_phase_shift_gate");
}
static inline void c_amodc_gate(unsigned int C, unsigned long long int atox, unsigned int c_qubit_num,
                                Register *reg, gsl_spmatrix_complex *matrix)
{
    (void)matrix;
    qcx_compat_die(qcx_c_amodc_gate(C, atox, c_qubit_num, qcx_compat_handle(reg)), "c_amodc_gate");
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
        qcx_compat_die(qcx_state_read(qcx_compat_handle(&reg), first, cnt, buf), "display_state");
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
    qcx_compat_die(qcx_total_probability(qcx_compat_handle(&reg), &total), "check_normalisation");     /* sequential sum, as T:28-37 */
    printf("Total Probability: %.16f\n", total);
}

#endif /* QCX_COMPAT_H */
