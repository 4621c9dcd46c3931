/*
 * qcx_oracle.c -- CPU ORACLE (test infrastructure, see qcx_oracle.h).
 *
 * Restates, in plain C, what /root/reference/qc_shor.c computes on its
 * gate-application path.  "Q:a-b" cites the reference lines each routine
 * follows.  Nothing here is used by the product path.
 *
 * Build:  gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC   (oracle/Makefile)
 * -ffp-contract=off matters: the reference's products and sums are separately
 * rounded binary64 operations (plain x86-64 gcc -O2 emits no FMA).
 */
#define _GNU_SOURCE            /* sincos */
#include "qcx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_SQRT1_2
#define M_SQRT1_2 0.70710678118654752440
#endif

/* gsl_complex_polar(1.0, theta) = (1*cos(theta), 1*sin(theta)) (Q:526, Q:323).  Under gcc -O2 the two calls
 * on the same argument become ONE glibc sincos() call -- in libgsl as much as in a program that inlines it --
 * and glibc's sincos can differ from its stand-alone sin()/cos() in the last bit.  The oracle pins that
 * behaviour by calling sincos explicitly instead of depending on what an optimiser does with cos()+sin(). */
void orc_polar(double theta, double *re, double *im)
{
    double sn, cs;
    sincos(theta, &sn, &cs);
    *re = 1.0 * cs;
    *im = 1.0 * sn;
}

/* bit b of x (qubit b == bit b of the state index, Q:150-151) */
static inline unsigned bit_of(uint64_t x, unsigned b) { return (unsigned)((x >> b) & 1u); }

/* ------------------------------------------------------------------------ */
/* MT19937 as shipped in GSL 2.6 (rng/mt.c): 2002 initialisation, seed 0 is */
/* replaced by 4357, uniform = u32 / 2^32.  Third-party algorithm restated   */
/* from the published MT19937 definition (Matsumoto & Nishimura).            */
/* ------------------------------------------------------------------------ */
void orc_rng_set(orc_rng *g, uint32_t seed)
{
    if (seed == 0) seed = 4357u;
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) {
        uint32_t prev = g->mt[i - 1];
        g->mt[i] = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
    }
    g->idx = 624;
}

static void mt_refill(orc_rng *g)
{
    uint32_t *mt = g->mt;
    for (int k = 0; k < 624; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[k] = v;
    }
    g->idx = 0;
}

uint32_t orc_rng_get(orc_rng *g)
{
    if (g->idx >= 624) mt_refill(g);
    uint32_t k = g->mt[g->idx++];
    k ^= (k >> 11);
    k ^= (k << 7) & 0x9d2c5680u;
    k ^= (k << 15) & 0xefc60000u;
    k ^= (k >> 18);
    return k;
}

double orc_rng_uniform(orc_rng *g) { return orc_rng_get(g) / 4294967296.0; }

/* ------------------------------------------------------------------------ */
/* COO container                                                             */
/* ------------------------------------------------------------------------ */
int orc_coo_init(orc_coo *m, size_t cap, int keep_zeros)
{
    if (cap == 0) cap = 1;
    m->row = (int32_t *)malloc(cap * sizeof(int32_t));
    m->col = (int32_t *)malloc(cap * sizeof(int32_t));
    m->val = (double *)malloc(cap * 2 * sizeof(double));
    m->nz = 0;
    m->cap = cap;
    m->keep_zeros = keep_zeros;
    return (m->row && m->col && m->val) ? 0 : 1;
}

void orc_coo_free(orc_coo *m)
{
    free(m->row); free(m->col); free(m->val);
    m->row = m->col = NULL; m->val = NULL; m->nz = m->cap = 0;
}

/* append a triplet; storage doubles when full (GSL's triplet growth rule) */
static void coo_put(orc_coo *m, uint64_t r, uint64_t c, double re, double im)
{
    if (!m->keep_zeros && re == 0.0 && im == 0.0) return;
    if (m->nz == m->cap) {
        size_t nc = m->cap * 2;
        m->row = (int32_t *)realloc(m->row, nc * sizeof(int32_t));
        m->col = (int32_t *)realloc(m->col, nc * sizeof(int32_t));
        m->val = (double *)realloc(m->val, nc * 2 * sizeof(double));
        m->cap = nc;
    }
    m->row[m->nz] = (int32_t)r;
    m->col[m->nz] = (int32_t)c;
    m->val[2 * m->nz] = re;
    m->val[2 * m->nz + 1] = im;
    m->nz++;
}

/* ------------------------------------------------------------------------ */
/* register                                                                  */
/* ------------------------------------------------------------------------ */
int orc_reg_init(orc_reg *r, int L, int M)
{
    r->L = L; r->M = M; r->n = (unsigned)(L + M);
    r->dim = (uint64_t)1 << r->n;
    r->buf[0] = (double *)calloc(2 * r->dim, sizeof(double));
    r->buf[1] = (double *)calloc(2 * r->dim, sizeof(double));
    r->cur = 0;
    return (r->buf[0] && r->buf[1]) ? 0 : 1;
}

void orc_reg_free(orc_reg *r) { free(r->buf[0]); free(r->buf[1]); r->buf[0] = r->buf[1] = NULL; }
double *orc_reg_state(orc_reg *r) { return r->buf[r->cur]; }

/* ------------------------------------------------------------------------ */
/* literal layer                                                             */
/* ------------------------------------------------------------------------ */
void orc_lit_reset(orc_reg *r)                                   /* Q:318-324 */
{
    double *s = orc_reg_state(r);
    memset(s, 0, 2 * r->dim * sizeof(double));
    orc_polar(0.0, &s[2], &s[3]);    /* polar(1, 0) */
}

/* Q:370-420: zero the other buffer, walk the triplets in insertion order
 * accumulating  new[row] += M * cur[col]  as four products and four sums,
 * empty the matrix, flip the buffers. */
static void lit_apply(orc_coo *m, orc_reg *r)
{
    const double *cur = r->buf[r->cur];
    double *nxt = r->buf[r->cur ^ 1];
    memset(nxt, 0, 2 * r->dim * sizeof(double));
    for (size_t k = 0; k < m->nz; k++) {
        uint32_t row = (uint32_t)m->row[k], col = (uint32_t)m->col[k];
        double mr = m->val[2 * k], mi = m->val[2 * k + 1];
        double cr = cur[2 * (size_t)col], ci = cur[2 * (size_t)col + 1];
        nxt[2 * (size_t)row]     += (mr * cr) - (mi * ci);
        nxt[2 * (size_t)row + 1] += (mr * ci) + (mi * cr);
    }
    m->nz = 0;
    r->cur ^= 1;
}

/* true when i and j agree on every qubit except the (up to two) free ones;
 * walks the bits one by one with early exit like Q:463-471 so that the cost
 * model of the literal tier stays that of the reference. */
static int same_elsewhere(uint64_t i, uint64_t j, unsigned n, unsigned f0, unsigned f1)
{
    uint64_t agree = ~(i ^ j);
    for (unsigned b = 0; b < n; b++) {
        if (b == f0 || b == f1) continue;
        if (((agree >> b) & 1u) == 0) return 0;
    }
    return 1;
}

void orc_lit_hadamard(unsigned q, orc_reg *r, orc_coo *m)        /* Q:442-484 */
{
    static const double h[2][2] = { { M_SQRT1_2, M_SQRT1_2 }, { M_SQRT1_2, -M_SQRT1_2 } };
    for (uint64_t i = 0; i < r->dim; i++)
        for (uint64_t j = 0; j < r->dim; j++)
            if (same_elsewhere(i, j, r->n, q, q))
                coo_put(m, i, j, h[bit_of(i, q)][bit_of(j, q)], 0.0);
    lit_apply(m, r);
}

void orc_lit_cphase(unsigned c, unsigned t, double theta, orc_reg *r, orc_coo *m) /* Q:513-565 */
{
    double er, ei;
    orc_polar(theta, &er, &ei);                                   /* gsl_complex_polar(1, theta), Q:526 */
    for (uint64_t i = 0; i < r->dim; i++)
        for (uint64_t j = 0; j < r->dim; j++) {
            if (!same_elsewhere(i, j, r->n, c, t)) continue;
            unsigned bi = 2 * bit_of(i, c) + bit_of(i, t);
            unsigned bj = 2 * bit_of(j, c) + bit_of(j, t);
            if (bi != bj)            coo_put(m, i, j, 0.0, 0.0);  /* off-diagonal of Q:220-225 */
            else if (bi == 3)        coo_put(m, i, j, er, ei);
            else                     coo_put(m, i, j, 1.0, 0.0);
        }
    lit_apply(m, r);
}

void orc_lit_camodc(unsigned C, unsigned long long atox, unsigned ctl, orc_reg *r, orc_coo *m) /* Q:595-660 */
{
    const unsigned A = (unsigned)(atox % C);                      /* Q:605 */
    const unsigned M = (unsigned)r->M;
    for (uint64_t k = 0; k < r->dim; k++) {
        if (!bit_of(k, ctl)) { coo_put(m, k, k, 1.0, 0.0); continue; }
        unsigned f = 0;
        for (unsigned b = 0; b < M; b++) f += bit_of(k, b) << b;
        if (f >= C) { coo_put(m, k, k, 1.0, 0.0); continue; }
        f = (A * f) % C;                                          /* 32-bit product, Q:639 */
        unsigned j = 0;
        for (unsigned b = 0; b < M; b++) j += bit_of(f, b) << b;
        for (unsigned b = M; b < r->n; b++) j += bit_of(k, b) << b;
        coo_put(m, j, k, 1.0, 0.0);
    }
    lit_apply(m, r);
}

/* theta of the ladder: pi / 2^d (Q:686).  2^d is exact for every d the
 * register sizes allow; the reference's 32-bit INT_POW would give 0 at d=32. */
static double ladder_theta(unsigned d) { return M_PI / (double)((uint64_t)1 << d); }

void orc_lit_iqft(orc_reg *r, orc_coo *m)                        /* Q:678-690 */
{
    for (int l = r->L + r->M - 1; l >= r->M; l--) {
        orc_lit_hadamard((unsigned)l, r, m);
        for (int k = l - 1; k >= r->M; k--)
            orc_lit_cphase((unsigned)l, (unsigned)k, ladder_theta((unsigned)(l - k)), r, m);
    }
}

/* a^(2^e) handed to the modular-multiply gate: either the reference's
 * INT_POW(a, x) with its 32-bit wrap (Q:729) or the mathematically right
 * residue (any representative works: the gate reduces mod C, Q:605). */
static unsigned long long ctrl_power(unsigned a, unsigned x_ref, unsigned e, unsigned C, int ref_intpow)
{
    if (ref_intpow) return orc_ref_intpow((double)a, (double)x_ref);
    unsigned long long v = a % C;
    for (unsigned s = 0; s < e; s++) v = (v * v) % C;
    return v;
}

void orc_lit_quantum_computation(unsigned C, unsigned a, int ref_intpow, orc_reg *r, orc_coo *m) /* Q:712-737 */
{
    const unsigned lo = r->n - (unsigned)r->L;
    for (unsigned l = lo; l < r->n; l++) orc_lit_hadamard(l, r, m);
    unsigned x = 1;
    for (unsigned l = lo; l < r->n; l++) {
        orc_lit_camodc(C, ctrl_power(a, x, l - lo, C, ref_intpow), l, r, m);
        x *= 2;
    }
    orc_lit_iqft(r, m);
}

void orc_spmv_hadamard(unsigned q, orc_reg *r, orc_coo *m)
{
    const uint64_t bitq = (uint64_t)1 << q;
    for (uint64_t i = 0; i < r->dim; i++) {     /* same row-major, j-ascending order as the scan */
        uint64_t j0 = i & ~bitq, j1 = i | bitq;
        coo_put(m, i, j0, M_SQRT1_2, 0.0);
        coo_put(m, i, j1, (i & bitq) ? -M_SQRT1_2 : M_SQRT1_2, 0.0);
    }
    lit_apply(m, r);
}

/* ------------------------------------------------------------------------ */
/* pairwise in-place layer.  Every output is written as the reference's      */
/* accumulation  0 + term0 + term1  with term = (mr*cr)-(mi*ci) etc., so the */
/* results carry the same bits (zero signs included) as the literal form.    */
/* ------------------------------------------------------------------------ */
void orc_pair_reset(double *amp, unsigned n)
{
    memset(amp, 0, ((size_t)2 << n) * sizeof(double));
    amp[2] = 1.0; amp[3] = 0.0;
}

void orc_pair_hadamard(double *amp, unsigned n, unsigned q, int threads)
{
    const double s = M_SQRT1_2, z = 0.0;
    const uint64_t half = (uint64_t)1 << (n - 1), bitq = (uint64_t)1 << q, low = bitq - 1;
    (void)threads;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (uint64_t p = 0; p < half; p++) {
        uint64_t i0 = ((p & ~low) << 1) | (p & low), i1 = i0 | bitq;
        double ar = amp[2 * i0], ai = amp[2 * i0 + 1], br = amp[2 * i1], bi = amp[2 * i1 + 1];
        double lo_r = 0.0, lo_i = 0.0, hi_r = 0.0, hi_i = 0.0;
        lo_r += (s * ar) - (z * ai);   lo_i += (s * ai) + (z * ar);     /* row i0, column i0 */
        lo_r += (s * br) - (z * bi);   lo_i += (s * bi) + (z * br);     /* row i0, column i1 */
        hi_r += (s * ar) - (z * ai);   hi_i += (s * ai) + (z * ar);     /* row i1, column i0 */
        hi_r += (-s * br) - (z * bi);  hi_i += (-s * bi) + (z * br);    /* row i1, column i1 */
        amp[2 * i0] = lo_r; amp[2 * i0 + 1] = lo_i;
        amp[2 * i1] = hi_r; amp[2 * i1 + 1] = hi_i;
    }
}

void orc_pair_cphase(double *amp, unsigned n, unsigned c, unsigned t, double theta, int threads)
{
    const double one = 1.0, z = 0.0;
    double er, ei;
    orc_polar(theta, &er, &ei);
    const uint64_t dim = (uint64_t)1 << n;
    const uint64_t both = ((uint64_t)1 << c) | ((uint64_t)1 << t);
    (void)threads;
#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
    for (uint64_t i = 0; i < dim; i++) {
        double re = amp[2 * i], im = amp[2 * i + 1], nr = 0.0, ni = 0.0;
        if ((i & both) == both) { nr += (er * re) - (ei * im);   ni += (er * im) + (ei * re); }
        else                    { nr += (one * re) - (z * im);   ni += (one * im) + (z * re); }
        amp[2 * i] = nr; amp[2 * i + 1] = ni;
    }
}

void orc_pair_camodc(double *amp, unsigned n, unsigned M, unsigned C,
                     unsigned long long atox, unsigned ctl, int threads)
{
    const unsigned A = (unsigned)(atox % C);
    const uint64_t blk = (uint64_t)1 << M, nblk = (uint64_t)1 << (n - M);
    const double one = 1.0, z = 0.0;
    (void)threads;
#pragma omp parallel num_threads(threads) if (threads > 1)
    {
        double *tmp = (double *)malloc(2 * blk * sizeof(double));
#pragma omp for schedule(static)
        for (uint64_t b = 0; b < nblk; b++) {
            double *s = amp + 2 * (b << M);
            const int on = (ctl >= M) ? (int)(((b << M) >> ctl) & 1u) : -1;   /* -1: control inside M */
            memset(tmp, 0, 2 * blk * sizeof(double));
            for (uint64_t f = 0; f < blk; f++) {
                int ctl_bit = (on >= 0) ? on : (int)((f >> ctl) & 1u);
                uint64_t dst = f;
                if (ctl_bit && f < C) dst = ((unsigned)((A * (unsigned)f) % C)) & (unsigned)(blk - 1);   /* only bits < M of f' land in j, Q:645-647 */
                tmp[2 * dst]     += (one * s[2 * f]) - (z * s[2 * f + 1]);
                tmp[2 * dst + 1] += (one * s[2 * f + 1]) + (z * s[2 * f]);
            }
            memcpy(s, tmp, 2 * blk * sizeof(double));
        }
        free(tmp);
    }
}

void orc_pair_iqft(double *amp, unsigned n, unsigned M, int threads)
{
    for (int l = (int)n - 1; l >= (int)M; l--) {
        orc_pair_hadamard(amp, n, (unsigned)l, threads);
        for (int k = l - 1; k >= (int)M; k--)
            orc_pair_cphase(amp, n, (unsigned)l, (unsigned)k, ladder_theta((unsigned)(l - k)), threads);
    }
}

void orc_pair_quantum_computation(double *amp, unsigned n, unsigned M, unsigned C,
                                  unsigned a, int ref_intpow, int threads)
{
    for (unsigned l = M; l < n; l++) orc_pair_hadamard(amp, n, l, threads);
    unsigned x = 1;
    for (unsigned l = M; l < n; l++) {
        orc_pair_camodc(amp, n, M, C, ctrl_power(a, x, l - M, C, ref_intpow), l, threads);
        x *= 2;
    }
    orc_pair_iqft(amp, n, M, threads);
}

/* ------------------------------------------------------------------------ */
/* per-index evaluation of whole circuits on a BASIS-STATE input             */
/*                                                                           */
/* Applied to a basis state, every Hadamard of inverse_QFT (Q:678-690) meets */
/* a pair of which exactly one member is non-zero (the bits it has not yet   */
/* reached still carry the input's values), so the amplitude of output index */
/* i is one scalar chain of the reference's per-gate sums -- the same        */
/* expressions as orc_pair_hadamard / orc_pair_cphase above, with the        */
/* partner's (+0, +0) written out.  This lets tests check windows of a 2^28  */
/* or 2^30 result bit for bit without any 2^n array on the host; the CPU     */
/* suite pins these chains to the full pairwise oracle at sizes it can hold. */
/* ------------------------------------------------------------------------ */
static void chain_hadamard(double *ar, double *ai, unsigned in_bit, unsigned out_bit)
{
    const double s = M_SQRT1_2, z = 0.0;
    /* the pair (lo = element with the target bit 0, hi = with 1); the input sits in lo or hi, the other is +0 */
    const double lr = in_bit ? 0.0 : *ar, li = in_bit ? 0.0 : *ai, hr = in_bit ? *ar : 0.0, hi = in_bit ? *ai : 0.0;
    double o_r = 0.0, o_i = 0.0;
    if (!out_bit) {
        o_r += (s * lr) - (z * li);   o_i += (s * li) + (z * lr);     /* row i0, column i0 */
        o_r += (s * hr) - (z * hi);   o_i += (s * hi) + (z * hr);     /* row i0, column i1 */
    } else {
        o_r += (s * lr) - (z * li);   o_i += (s * li) + (z * lr);     /* row i1, column i0 */
        o_r += (-s * hr) - (z * hi);  o_i += (-s * hi) + (z * hr);    /* row i1, column i1 */
    }
    *ar = o_r; *ai = o_i;
}

static void chain_cphase(double *ar, double *ai, int selected, double theta)
{
    const double one = 1.0, z = 0.0;
    double er, ei, nr = 0.0, ni = 0.0;
    orc_polar(theta, &er, &ei);
    if (selected) { nr += (er * *ar) - (ei * *ai);   ni += (er * *ai) + (ei * *ar); }
    else          { nr += (one * *ar) - (z * *ai);   ni += (one * *ai) + (z * *ar); }
    *ar = nr; *ai = ni;
}

/* amplitudes [first, first + count) of inverse_QFT (Q:678-690) applied to the basis state |x> of an n-qubit register
 * whose M register is bits [0, M) */
void orc_basis_iqft_window(uint64_t x, unsigned n, unsigned M, uint64_t first, uint64_t count, double *out)
{
    const uint64_t mlow = ((uint64_t)1 << M) - 1;
    for (uint64_t w = 0; w < count; w++) {
        const uint64_t i = first + w;
        double ar = 1.0, ai = 0.0;
        if ((i & mlow) != (x & mlow)) { out[2 * w] = 0.0; out[2 * w + 1] = 0.0; continue; }     /* the M register is not touched */
        for (int l = (int)n - 1; l >= (int)M; l--) {
            chain_hadamard(&ar, &ai, bit_of(x, (unsigned)l), bit_of(i, (unsigned)l));
            for (int k = l - 1; k >= (int)M; k--)       /* bit l is final (= i's), bit k still the input's (= x's) */
                chain_cphase(&ar, &ai, bit_of(i, (unsigned)l) && bit_of(x, (unsigned)k), ladder_theta((unsigned)(l - k)));
        }
        out[2 * w] = ar; out[2 * w + 1] = ai;
    }
}

/* amplitudes [first, first + count) after the first two stages of quantum_computation (Q:712-731) on the reset state
 * |0...01>: the Hadamard layer over the L register, then the controlled modular multiplies (exact powers or the
 * reference's INT_POW).  Index (l << M | f) is non-zero iff f = prod over the set bits j of l of atox_j, mod C,
 * applied to 1 in ascending j; its value is the chain of L Hadamards that all find their input in the lo element. */
void orc_shor_front_window(unsigned n, unsigned M, unsigned C, unsigned a, int ref_intpow,
                           uint64_t first, uint64_t count, double *out)
{
    const unsigned L = n - M;
    const uint64_t mlow = ((uint64_t)1 << M) - 1;
    double ar = 1.0, ai = 0.0;
    for (unsigned l = M; l < n; l++) chain_hadamard(&ar, &ai, 0, 0);
    /* (out_bit 0 and 1 give the same value when the input is in lo: s*lo + s*0 and s*lo + (-s)*0 differ only in the
     * sign of a zero term that the leading "0 +" absorbs; checked in the CPU suite against the pairwise oracle) */
    unsigned long long atox[64];
    unsigned xr = 1;
    for (unsigned j = 0; j < L && j < 64; j++) { atox[j] = ctrl_power(a, xr, j, C, ref_intpow) % C; xr *= 2; }
    for (uint64_t w = 0; w < count; w++) {
        const uint64_t i = first + w, l = i >> M;
        unsigned long long f = 1;
        int moved_ok = 1;
        for (unsigned j = 0; j < L; j++)
            if ((l >> j) & 1) {
                if (f < C) f = (atox[j] * f) % C;           /* Q:631-634: residues >= C stay */
            }
        (void)moved_ok;
        if ((i & mlow) == f && f <= mlow) { out[2 * w] = ar; out[2 * w + 1] = ai; }
        else { out[2 * w] = 0.0; out[2 * w + 1] = 0.0; }
    }
}

/* ------------------------------------------------------------------------ */
/* measurement, norm                                                         */
/* ------------------------------------------------------------------------ */
int orc_measure_range(const double *amp, uint64_t first, uint64_t count,
                      uint64_t last_excluded, double cum_in, double r,
                      uint64_t *idx, double *cum_out)
{
    double cum = cum_in;
    for (uint64_t k = 0; k < count; k++) {
        uint64_t g = first + k;
        if (g >= last_excluded) break;                /* loop bound dim-1, Q:283 */
        double x = amp[2 * k], y = amp[2 * k + 1];
        cum += x * x + y * y;                         /* gsl_complex_abs2, Q:286 */
        if (cum >= r) { *idx = g; *cum_out = cum; return 1; }
    }
    *cum_out = cum;
    return 0;
}

uint64_t orc_measure(double *amp, unsigned n, double r)          /* Q:272-306 */
{
    const uint64_t dim = (uint64_t)1 << n;
    uint64_t idx = dim - 1;
    double cum;
    orc_measure_range(amp, 0, dim, dim - 1, 0.0, r, &idx, &cum);
    memset(amp, 0, 2 * dim * sizeof(double));
    amp[2 * idx] = 1.0; amp[2 * idx + 1] = 0.0;
    return idx;
}

double orc_norm2(const double *amp, unsigned n)                  /* T:28-37 */
{
    const uint64_t dim = (uint64_t)1 << n;
    double s = 0.0;
    for (uint64_t i = 0; i < dim; i++) s += amp[2 * i] * amp[2 * i] + amp[2 * i + 1] * amp[2 * i + 1];
    return s;
}

/* ------------------------------------------------------------------------ */
/* host-side scalar helpers                                                  */
/* ------------------------------------------------------------------------ */
/* double -> unsigned int as x86-64 gcc does it for the reference's casts:
 * truncate to a signed 64-bit integer (out of range -> 0x8000000000000000)
 * and keep the low 32 bits. */
static unsigned u32_from_double(double d)
{
    if (!(d > -9223372036854775808.0 && d < 9223372036854775808.0)) return 0u;
    return (unsigned)(uint64_t)(int64_t)d;
}

unsigned orc_ref_intpow(double base, double power) { return u32_from_double(pow(base, power) + 0.5); }

unsigned long long orc_modpow(unsigned long long a, unsigned long long e, unsigned long long m)
{
    unsigned long long r = 1 % m;
    a %= m;
    while (e) {
        if (e & 1) r = (unsigned long long)(((__uint128_t)r * a) % m);
        a = (unsigned long long)(((__uint128_t)a * a) % m);
        e >>= 1;
    }
    return r;
}

unsigned orc_gcd(unsigned a, unsigned b)
{
    if (a == 0) return b;
    if (b == 0) return a;
    while (a % b) { unsigned t = a % b; a = b; b = t; }
    return b;
}

void orc_cf_denominators(double omega, unsigned count, unsigned *den)
{
    unsigned *co = (unsigned *)malloc((count ? count : 1) * sizeof(unsigned));
    for (unsigned i = 0; i < count; i++) {
        double inv = 1.0 / omega;
        omega = inv - (double)u32_from_double(inv);
        co[i] = u32_from_double(inv - omega);
        unsigned d = 1, nmr = 0;
        for (int c = (int)i - 1; c >= 0; c--) { unsigned t = d; d = nmr + d * co[c]; nmr = t; }
        den[i] = d;
    }
    free(co);
}

double orc_read_omega(uint64_t state, int L, int M)
{
    unsigned xt = 0;
    for (int p = 0; p < L; p++) xt += bit_of(state, (unsigned)(L + M - 1 - p)) << p;
    return (double)xt / (double)((uint64_t)1 << L);
}

/* ------------------------------------------------------------------------ */
/* CPU twin of the device-side synthetic-state generator (include/qcx.h,     */
/* qcx_state_fill_random): any window of the vector, same bits.             */
/* ------------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

void orc_fill_random(double *amp, uint64_t first, uint64_t count, uint64_t seed, double scale)
{
    for (uint64_t i = 0; i < count; i++) {
        uint64_t k = 2 * (first + i);
        amp[2 * i]     = ((double)(splitmix64(seed + k) >> 11) * 0x1p-53 - 0.5) * scale;
        amp[2 * i + 1] = ((double)(splitmix64(seed + k + 1) >> 11) * 0x1p-53 - 0.5) * scale;
    }
}
