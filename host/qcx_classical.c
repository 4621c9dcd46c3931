/* qcx_classical.c -- classical post-processing around the GPU hot path (see qcx_classical.h). */
#include "qcx_classical.h"

#include <math.h>

/* double -> unsigned the way the reference's casts behave on x86-64 (truncate through a signed
 * 64-bit integer, keep the low 32 bits; out of range or NaN gives 0) */
static unsigned trunc_u32(double d)
{
    if (!(d > -9223372036854775808.0 && d < 9223372036854775808.0)) return 0u;
    return (unsigned)(unsigned long long)(long long)d;
}

/* the reference's INT_POW (Q:158-159) with its 32-bit wrap */
static unsigned ref_int_pow(double base, double power) { return trunc_u32(pow(base, power) + 0.5); }

unsigned qcx_gcd(unsigned a, unsigned b)
{
    if (a == 0) return b;
    if (b == 0) return a;
    for (;;) {
        unsigned r = a % b;
        if (r == 0) return b;
        a = b;
        b = r;
    }
}

unsigned long long qcx_modpow(unsigned long long base, unsigned long long e, unsigned long long m)
{
    unsigned long long acc = 1 % m;
    base %= m;
    for (; e; e >>= 1) {
        if (e & 1) acc = (unsigned long long)(((__uint128_t)acc * base) % m);
        base = (unsigned long long)(((__uint128_t)base * base) % m);
    }
    return acc;
}

void qcx_cf_denominators(double omega, unsigned count, unsigned *den)
{
    unsigned coeff[64];
    if (count > 64) count = 64;
    for (unsigned i = 0; i < count; i++) {
        const double inv = 1.0 / omega;
        omega = inv - (double)trunc_u32(inv);           /* fractional part feeds the next level */
        coeff[i] = trunc_u32(inv - omega);
        /* fold the coefficients found so far (all but the newest) from the innermost outwards */
        unsigned d = 1, num = 0;
        for (unsigned c = i; c-- > 0;) {
            const unsigned keep = d;
            d = num + d * coeff[c];
            num = keep;
        }
        den[i] = d;
    }
}

/* the L register read in reversed bit order (Q:868-883), as a 64-bit value: L can exceed 32 on this engine */
static unsigned long long read_x_tilde64(unsigned long state_num, int L, int M)
{
    unsigned long long x = 0;
    for (int p = 0; p < L && p < 64; p++)
        x |= (unsigned long long)((state_num >> (L + M - 1 - p)) & 1UL) << p;
    return x;
}

unsigned qcx_read_x_tilde(unsigned long state_num, int L, int M)      /* the reference's unsigned int: low 32 bits */
{
    return (unsigned)read_x_tilde64(state_num, L, M);
}

double qcx_read_omega(unsigned long state_num, int L, int M)
{
    return (double)read_x_tilde64(state_num, L, M) / ldexp(1.0, L);
}

static int is_period(unsigned a, unsigned p, unsigned C, int ref_intpow)
{
    if (ref_intpow) return ref_int_pow((double)a, (double)p) % C == 1;
    return p != 0 && qcx_modpow(a, p, C) == 1 % C;
}

unsigned qcx_period_from_omega(double omega, unsigned a, unsigned C, int ref_intpow)
{
    unsigned den[QCX_NUM_CONTINUED_FRACTIONS];
    if (!ref_intpow && !(omega > 0.0)) return 0;        /* x~ = 0 carries no period information */
    qcx_cf_denominators(omega, QCX_NUM_CONTINUED_FRACTIONS, den);
    for (unsigned d = 0; d < QCX_NUM_CONTINUED_FRACTIONS; d++)
        for (unsigned m = 1; m <= QCX_TRIALS_PER_DENOMINATOR; m++)
            if (is_period(a, m * den[d], C, ref_intpow)) return m * den[d];
    return 0;
}

int qcx_factors_from_period(unsigned a, unsigned period, unsigned C, int ref_intpow, unsigned factors[2])
{
    if (period % 2 != 0) return 1;
    /* a^(p/2): the reference's 32-bit INT_POW, or the residue mod C (same gcds, no overflow) */
    const unsigned half = ref_intpow ? ref_int_pow((double)a, (double)(period / 2))
                                     : (unsigned)qcx_modpow(a, period / 2, C);
    if (half % C == C - 1) return 2;
    factors[0] = qcx_gcd(half + 1u, C);
    factors[1] = qcx_gcd(ref_intpow ? half - 1u : (half + C - 1u) % C, C);
    return 0;
}
