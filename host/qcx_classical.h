/*
 * qcx_classical.h -- the classical (host-only, O(15) scalar work per run) post-processing of Shor's
 * algorithm that surrounds the GPU hot path: SURVEY s8(f) rank 1.  Written from scratch in C; each
 * routine cites the reference lines whose behaviour it keeps (Q: = /root/reference/qc_shor.c).
 * Differences from the reference are deliberate and listed in host/README.md (correct modular
 * arithmetic instead of the 32-bit INT_POW, initialised flags, -f accepted next to -a).
 */
#ifndef QCX_CLASSICAL_H
#define QCX_CLASSICAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QCX_NUM_CONTINUED_FRACTIONS 15      /* Q:121 */
#define QCX_TRIALS_PER_DENOMINATOR  10      /* Q:122 */

unsigned qcx_gcd(unsigned a, unsigned b);                                         /* Q:756-779 */
unsigned long long qcx_modpow(unsigned long long base, unsigned long long e, unsigned long long m);
/* denominators of the first `count` convergents of omega, built the way Q:806-846 builds them */
void     qcx_cf_denominators(double omega, unsigned count, unsigned *denominators);
/* x~ read from the L register in reversed bit order (the IQFT has no swaps), Q:868-883 */
unsigned qcx_read_x_tilde(unsigned long state_num, int L_size, int M_size);
double   qcx_read_omega(unsigned long state_num, int L_size, int M_size);
/* smallest multiple m*d (d over the convergent denominators, m = 1..10) with a^(m*d) = 1 mod C,
 * searched in the reference's order (Q:941-955); 0 when none.  ref_intpow != 0 tests with the
 * reference's wrapping INT_POW instead of exact modular powers. */
unsigned qcx_period_from_omega(double omega, unsigned a, unsigned C, int ref_intpow);
/* Q:1030-1050: 0 = valid (factors written), 1 = odd period, 2 = a^(p/2) = -1 mod C */
int      qcx_factors_from_period(unsigned a, unsigned period, unsigned C, int ref_intpow, unsigned factors[2]);

#ifdef __cplusplus
}
#endif
#endif
