/*
 * A circuit builder written the way qc_shor.c writes it (gate calls with the scratch-matrix argument,
 * Register passed by value / by pointer, INT_POW, M_PI) compiled against include/qcx_compat.h and run on
 * the GPU.  Prints the final state as hex doubles; tests/test_gpu_compat_c.py compares it with the oracle.
 * The two builder bodies below are this repository's own wording of the schedules of qc_shor.c:678-737.
 */
#include <inttypes.h>
#include <stdio.h>
#include <string.h>

#include "qcx_compat.h"

static void iqft_like_reference(Register *reg, gsl_spmatrix_complex *matrix)
{
    for (int l = reg->L_size + reg->M_size - 1; l >= reg->M_size; l--) {
        hadamard_gate(l, reg, matrix);
        for (int k = l - 1; k >= reg->M_size; k--)
            c_phase_shift_gate(l, k, M_PI / INT_POW(2, l - k), reg, matrix);
    }
}

static void circuit_like_reference(unsigned int C, unsigned int a, Register *reg, gsl_spmatrix_complex *matrix)
{
    unsigned int x = 1;
    for (unsigned int l = reg->num_qubits - reg->L_size; l < reg->num_qubits; l++) hadamard_gate(l, reg, matrix);
    for (unsigned int l = reg->num_qubits - reg->L_size; l < reg->num_qubits; l++) {
        c_amodc_gate(C, INT_POW(a, x), l, reg, matrix);
        x *= 2;
    }
    iqft_like_reference(reg, matrix);
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s C L M a seed\n", argv[0]); return 2; }
    Register reg;
    memset(&reg, 0, sizeof reg);
    unsigned C = (unsigned)atoi(argv[1]);
    reg.L_size = atoi(argv[2]);
    reg.M_size = atoi(argv[3]);
    unsigned a = (unsigned)atoi(argv[4]);
    gsl_rng *rng = qcx_rng_alloc();
    qcx_rng_set(rng, strtoul(argv[5], NULL, 10));
    if (register_alloc(&reg) != NO_ERROR) { fprintf(stderr, "register_alloc failed\n"); return 4; }
    gsl_spmatrix_complex *matrix = NULL;

    reset_register(reg);
    circuit_like_reference(C, a, &reg, matrix);
    swap_states(&reg);

    if (argc > 6 && strcmp(argv[6], "debug") == 0) {       /* the developer helpers of testing_and_debug.c */
        check_normalisation(reg);
        unsigned long m = measure_state(reg, rng);
        printf("measured %lu\n", m);
        display_state(reg);
        check_normalisation(reg);
        register_free(&reg);
        qcx_rng_free(rng);
        return 0;
    }
    double *host = (double *)malloc(reg.num_states * 2 * sizeof(double));
    if (qcx_state_read(qcx_compat_handle(&reg), 0, reg.num_states, host) != QCX_NO_ERROR) return 4;
    for (unsigned long i = 0; i < 2 * reg.num_states; i++) {
        uint64_t u; memcpy(&u, &host[i], 8);
        printf("%016" PRIx64 "\n", u);
    }
    unsigned long idx = measure_state(reg, rng);
    printf("measured %lu\n", idx);
    free(host);
    register_free(&reg);
    qcx_rng_free(rng);
    return 0;
}
