/*
 * A main() shaped like the reference's (qc_shor.c:1284-1347) -- same GSL-flavoured declarations and calls, in the same
 * order: an MT19937 from gsl_rng_alloc(gsl_rng_mt19937), a Register filled in from the command line (L_size, M_size,
 * num_qubits, num_states, as the reference's argument parser leaves it), two state vectors from
 * gsl_vector_complex_alloc(num_states), a scratch matrix from gsl_spmatrix_complex_alloc_nzmax, current_state /
 * new_state pointing at the two vectors, the circuit, the four frees -- compiled against include/qcx_compat.h with NOT ONE
 * qcx_* call in main.  This repository's own wording; only the call sequence follows the reference.
 * Prints the final state as hex doubles and the measured index; tests/test_gpu_compat_c.py compares with the oracle.
 */
#include <inttypes.h>
#include <stdio.h>
#include <string.h>

#include "qcx_compat.h"

#define ALLOC_CHECK(p) if ((p) == NULL) { fprintf(stderr, "Error: Insufficient memory.\n"); }

static void iqft_schedule(Register *reg, gsl_spmatrix_complex *matrix)
{
    for (int l = reg->L_size + reg->M_size - 1; l >= reg->M_size; l--) {
        hadamard_gate(l, reg, matrix);
        for (int k = l - 1; k >= reg->M_size; k--)
            c_phase_shift_gate(l, k, M_PI / INT_POW(2, l - k), reg, matrix);
    }
}

static void shor_circuit(unsigned int C, unsigned int a, Register *reg, gsl_spmatrix_complex *matrix)
{
    unsigned int x = 1;
    for (unsigned int l = reg->num_qubits - reg->L_size; l < reg->num_qubits; l++) hadamard_gate(l, reg, matrix);
    for (unsigned int l = reg->num_qubits - reg->L_size; l < reg->num_qubits; l++) {
        c_amodc_gate(C, INT_POW(a, x), l, reg, matrix);
        x *= 2;
    }
    iqft_schedule(reg, matrix);
}

int main(int argc, char *argv[])
{
    gsl_spmatrix_complex *matrix;
    Register reg;
    const gsl_rng_type *rng_type;
    gsl_rng *rng;
    unsigned int C, a;

    if (argc < 6) { fprintf(stderr, "usage: %s C L M a seed\n", argv[0]); return BAD_ARGUMENTS; }

    rng_type = gsl_rng_mt19937;
    rng = gsl_rng_alloc(rng_type);
    ALLOC_CHECK(rng);
    gsl_rng_set(rng, strtoul(argv[5], NULL, 10));           /* (the reference seeds with the time of day) */

    C = (unsigned int)atoi(argv[1]);
    a = (unsigned int)atoi(argv[4]);
    reg.L_size = atoi(argv[2]);
    reg.M_size = atoi(argv[3]);
    reg.num_qubits = reg.L_size + reg.M_size;
    reg.num_states = 1ul << reg.num_qubits;

    reg.state_a = gsl_vector_complex_alloc(reg.num_states);
    ALLOC_CHECK(reg.state_a);
    reg.state_b = gsl_vector_complex_alloc(reg.num_states);
    ALLOC_CHECK(reg.state_b);
    matrix = gsl_spmatrix_complex_alloc_nzmax(reg.num_states, reg.num_states, 2 * reg.num_states, GSL_SPMATRIX_COO);
    ALLOC_CHECK(matrix);

    reg.current_state = &reg.state_a;
    reg.new_state = &reg.state_b;

    reset_register(reg);
    shor_circuit(C, a, &reg, matrix);
    operate_matrix(matrix, &reg);                           /* nothing left to apply: accepted, does nothing */
    swap_states(&reg);

    {   /* test output: the state before the measurement, then the measured index */
        double *host = (double *)malloc(reg.num_states * 2 * sizeof(double));
        if (!host || qcx_state_read(qcx_compat_handle(&reg), 0, reg.num_states, host) != QCX_NO_ERROR) return UNKNOWN_ERROR;
        for (unsigned long i = 0; i < 2 * reg.num_states; i++) {
            uint64_t u; memcpy(&u, &host[i], 8);
            printf("%016" PRIx64 "\n", u);
        }
        free(host);
    }
    printf("draw %.17g\n", 0.0 * gsl_rng_uniform(rng));     /* consumes one draw, like a failed attempt would */
    printf("measured %lu\n", measure_state(reg, rng));

    gsl_vector_complex_free(reg.state_a);
    gsl_vector_complex_free(reg.state_b);
    gsl_spmatrix_complex_free(matrix);
    gsl_rng_free(rng);
    return NO_ERROR;
}
