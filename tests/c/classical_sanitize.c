/* host/qcx_classical.c under AddressSanitizer + UBSan (CPU only): sweeps the classical helpers of the Shor driver over
 * ordinary and hostile inputs.  Any out-of-bounds access, signed overflow or invalid shift aborts the program. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../host/qcx_classical.h"

int main(void)
{
    unsigned long checks = 0;
    unsigned den[QCX_NUM_CONTINUED_FRACTIONS + 4];
    const double omegas[] = {0.0, 1e-300, 4.9e-324, 0.5, 1.0 / 3.0, 5.0 / 6.0, 0.8333333432674408, 0.9999999999999999, 1.0,
                             0.6180339887498949, 1.0 / 4294967296.0, 3.0 / 1024.0, 0.25, 0.75, 1.0 / 7.0};
    for (unsigned k = 0; k < sizeof omegas / sizeof omegas[0]; k++) {
        for (unsigned count = 0; count <= QCX_NUM_CONTINUED_FRACTIONS + 4; count += 3) {
            qcx_cf_denominators(omegas[k], count, den);
            checks++;
        }
        for (unsigned C = 2; C < 200; C += 7)
            for (unsigned a = 2; a + 1 < C; a += 3)
                for (int quirks = 0; quirks < 2; quirks++) {
                    unsigned p = qcx_period_from_omega(omegas[k], a, C, quirks);
                    unsigned f[2] = {0, 0};
                    if (p) (void)qcx_factors_from_period(a, p, C, quirks, f);
                    checks++;
                }
    }
    /* extremes of the integer helpers */
    const unsigned big[] = {0u, 1u, 2u, 3u, 65535u, 65536u, 2147483647u, 2147483648u, 4294967295u};
    for (unsigned i = 0; i < 9; i++)
        for (unsigned j = 0; j < 9; j++) {
            (void)qcx_gcd(big[i], big[j]);
            if (big[j]) (void)qcx_modpow(big[i], (unsigned long long)big[j] * 4294967311ull, big[j]);
            unsigned f[2];
            if (big[j] >= 2) (void)qcx_factors_from_period(big[i], big[i] | 1u, big[j], 0, f);
            if (big[j] >= 2) (void)qcx_factors_from_period(big[i], big[i] & ~1u, big[j], 1, f);
            checks++;
        }
    for (int L = 1; L <= 40; L += 3)
        for (int M = 0; M <= 12; M += 4) {
            if (L + M > 62) continue;
            const unsigned long top = (1ul << (L + M)) - 1;
            (void)qcx_read_x_tilde(top, L, M); (void)qcx_read_omega(top, L, M);
            (void)qcx_read_x_tilde(0, L, M);   (void)qcx_read_omega(top / 3, L, M);
            checks++;
        }
    printf("classical helpers: %lu sanitized calls ok\n", checks);
    return 0;
}
