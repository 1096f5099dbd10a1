/*
 * design_host.c -- host-side IIR design entry points of include/RTFilter.h:
 * butterworth, chebyshev1, chebyshev2, thiran.
 *
 * Filter design stays on the host (north star): it runs once per filter, is
 * O(order^2), and only produces the coefficient bank that the device kernels
 * consume.  These routines write b[0..nb) then a[0..na) into the filter's
 * existing bank and shrink nb/na, like the reference (src/RTFilter.c:1216-1238,
 * 1313-1335, 1415-1437, 1459-1495).
 *
 * Structure here is our own: a design is held as an explicit cascade of
 * low-order sections (gain, zero sections, pole sections) in scratch memory,
 * band-transformed section by section, then expanded into the two transfer-
 * function polynomials.  The ARITHMETIC (which libm call, which operand order,
 * which pairing of numerator and denominator factors into the gain) follows
 * the reference step for step so that the resulting coefficients have the same
 * bits; tests/test_design.py checks that against tests/golden/designs.json.
 * Built with -ffp-contract=off like the rest of the host code.
 *
 * Deliberate differences from the reference:
 *   - chebyshev1/chebyshev2 install RTIIRFilter_stable_init when `initialize`
 *     is NULL (the reference assigns only its local variable, :1321-1323,
 *     :1423-1425, leaving a NULL callback that the first update calls through);
 *   - the stray debug printf calls of chebyshev2 (:1384-1406) and the band-stop
 *     transform (:850) are not reproduced.
 */
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>

#include "RTFilter.h"
#include "RTFilterDesign.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* 1 + c[0] z^-1 + ... + c[deg-1] z^-deg */
typedef struct section {
    double c[4];
    int deg;
} section;

typedef struct cascade {
    double gain;
    section *zeros;
    size_t nzeros;
    section *poles;
    size_t npoles;
} cascade;

static int cascade_alloc(cascade *cs, size_t order)
{
    cs->zeros = (section *)DSP_MALLOC(sizeof(section) * (order + 1));
    cs->poles = (section *)DSP_MALLOC(sizeof(section) * (order + 1));
    cs->nzeros = cs->npoles = 0;
    cs->gain = 1.0;
    if (!cs->zeros || !cs->poles) {
        DSP_FREE(cs->zeros);
        DSP_FREE(cs->poles);
        return 1;
    }
    return 0;
}

static void cascade_free(cascade *cs)
{
    DSP_FREE(cs->zeros);
    DSP_FREE(cs->poles);
}

static void push1(section *list, size_t *n, double c1)
{
    list[*n].deg = 1;
    list[*n].c[0] = c1;
    (*n)++;
}

static void push2(section *list, size_t *n, double c1, double c2)
{
    list[*n].deg = 2;
    list[*n].c[0] = c1;
    list[*n].c[1] = c2;
    (*n)++;
}

/* ---- low-pass prototypes (bilinear-warped at w0) -------------------------- */

/* ref: src/RTFilter.c:1142-1205 */
static void prototype_butterworth(cascade *cs, size_t L, double w0)
{
    const size_t odd = L & 1;
    const size_t nsec = odd + (L - odd) / 2;
    const double W = tan(M_PI * w0 / 2);
    const double W2 = pow(W, 2.0);
    cs->gain = 1.0;
    for (size_t i = 0; i < nsec; i++) {
        double num, den;
        if (i < odd) {
            num = W;
            den = 1 + W;
            push1(cs->poles, &cs->npoles, -(1 - W) / den);
        } else {
            num = W2;
            const double zi = 2.0 * W * sin(M_PI * (2 * (i - odd) + 1) / (2 * L));
            const double z2 = W2;
            den = 1 + zi + z2;
            const double p1 = -2.0 * (1 - z2) / den;
            push2(cs->poles, &cs->npoles, p1, (1 - zi + z2) / den);
        }
        cs->gain *= num / den;
    }
    for (size_t i = 0; i < L; i++) push1(cs->zeros, &cs->nzeros, 1.0);   /* L zeros at z = -1 */
}

/* ref: src/RTFilter.c:1240-1311 */
static void prototype_chebyshev1(cascade *cs, size_t L, double ripple, double w0)
{
    const size_t odd = L & 1;
    const size_t nsec = odd + (L - odd) / 2;
    const double coshf = cosh(asinh(1.0 / ripple) / L);
    const double sinhf = sinh(asinh(1.0 / ripple) / L);
    const double W = tan(M_PI * w0 / 2);
    const double wsinhf = W * sinhf;
    const double W2 = pow(W, 2);
    cs->gain = odd ? 1.0 : 1.0 / pow(ripple * ripple + 1, .5);
    for (size_t i = 0; i < nsec; i++) {
        double num, den;
        if (i < odd) {
            num = wsinhf;
            den = 1 + wsinhf;
            push1(cs->poles, &cs->npoles, -(1 - wsinhf) / den);
        } else {
            const double phase = M_PI * (2 * (i - odd * 1.0) + 1) / (2 * L);
            const double sinphase = sin(phase);
            const double cosphase = cos(phase);
            const double rhok2 = pow(cosphase * coshf, 2) + pow(sinphase * sinhf, 2);
            const double rhokim = 2 * wsinhf * sinphase;
            num = W2 * rhok2;
            den = 1 + rhokim + num;
            const double p1 = -2.0 * (1 - num) / den;
            push2(cs->poles, &cs->npoles, p1, (1 - rhokim + num) / den);
        }
        cs->gain *= num / den;
    }
    for (size_t i = 0; i < L; i++) push1(cs->zeros, &cs->nzeros, 1.0);
}

/* ref: src/RTFilter.c:1338-1413.  Zeros sit on the unit circle, one conjugate
 * pair per second-order section (plus z = -1 for odd orders). */
static void prototype_chebyshev2(cascade *cs, size_t L, double ripple, double w0)
{
    const size_t odd = L & 1;
    const size_t nsec = odd + (L - odd) / 2;
    const double coshf2 = pow(cosh(asinh(1.0 / ripple) / L), 2);
    const double sinhf = sinh(asinh(1.0 / ripple) / L);
    const double W = tan(M_PI * w0 / 2);
    const double wsinhf = W * sinhf;
    const double W2 = pow(W, 2);
    cs->gain = 1.0;
    for (size_t i = 0; i < nsec; i++) {
        double num, den;
        if (i < odd) {
            num = W;
            push1(cs->zeros, &cs->nzeros, 1.0);
        } else {
            const double cos2 = pow(cos(M_PI * (2 * (i - odd * 1.0) + 1) / (2 * L)), 2);
            num = cos2 + W2;
            const double z1 = -2 * (cos2 - W2) / num;
            push2(cs->zeros, &cs->nzeros, z1, 1.0);
        }
        if (i < odd) {
            den = W + sinhf;
            push1(cs->poles, &cs->npoles, -(sinhf - W) / den);
        } else {
            const double phase = M_PI * (2 * (i - odd * 1.0) + 1) / (2 * L);
            const double sinphase = sin(phase);
            const double cosphase = cos(phase);
            const double rhok2 = pow(cosphase, 2) * coshf2 + pow(sinphase * sinhf, 2);
            const double rhokim = 2 * wsinhf * sinphase;
            den = rhok2 + rhokim + W2;
            const double p1 = -2.0 * (rhok2 - W2) / den;
            push2(cs->poles, &cs->npoles, p1, (rhok2 - rhokim + W2) / den);
        }
        cs->gain *= num / den;
    }
}

/* ---- band transforms, one section at a time ------------------------------- */

/* low-pass -> high-pass (ref: src/RTFilter.c:650-710).  Returns the section's
 * contribution to the gain ratio. */
static double hp_section(section *s, double alpha, double alpha2)
{
    double scale;
    if (s->deg == 1) {
        const double z = s->c[0];
        scale = 1 - alpha * z;
        s->c[0] = (alpha - z) / scale;
    } else {
        const double B1 = s->c[0], B2 = s->c[1];
        scale = 1 - alpha * B1 + alpha2 * B2;
        s->c[0] = (2 * alpha - B1 * (1 + alpha2) + 2 * alpha * B2) / scale;
        s->c[1] = (alpha2 - alpha * B1 + B2) / scale;
    }
    return scale;
}

typedef struct band_consts {
    double alpha, beta, beta2;
    double k[5];
    int negate;     /* band-stop feeds -c1 into the same formulas */
} band_consts;

/* low-pass -> band-pass / band-stop: every section doubles its degree
 * (ref: src/RTFilter.c:712-839 and :841-968). */
static double band_section(section *s, const band_consts *q)
{
    double scale;
    if (s->deg == 1) {
        const double B0 = q->negate ? -s->c[0] : s->c[0];
        scale = 1 - q->beta * B0;
        s->c[0] = q->alpha * (1 - B0) / scale;
        s->c[1] = (q->beta - B0) / scale;
        s->deg = 2;
    } else {
        const double B1 = q->negate ? -s->c[0] : s->c[0];
        const double B2 = s->c[1];
        scale = 1 - q->beta * B1 + q->beta2 * B2;
        s->c[0] = (q->k[0] - B1 * q->k[1] + q->k[2] * B2) / scale;
        s->c[1] = (q->k[3] - B1 * q->k[4] + B2 * q->k[3]) / scale;
        s->c[2] = (q->k[2] - B1 * q->k[1] + q->k[0] * B2) / scale;
        s->c[3] = (q->beta2 - B1 * q->beta + B2) / scale;
        s->deg = 4;
    }
    return scale;
}

static void band_constants(band_consts *q, double alpha, double beta, int negate)
{
    const double alpha2 = pow(alpha, 2);
    q->alpha = alpha;
    q->beta = beta;
    q->beta2 = pow(beta, 2);
    q->k[0] = 2 * alpha;
    q->k[1] = alpha * (1 + beta);
    q->k[2] = 2 * alpha * beta;
    q->k[3] = alpha2 + 2 * beta;
    q->k[4] = alpha2 + 1 + q->beta2;
    q->negate = negate;
}

/* Zero section i and pole section i are transformed together and their scale
 * factors enter the gain as one ratio, in index order -- the pairing fixes the
 * rounding of the gain, so it is kept (ref: the shared while loops at
 * :669-706, :777-832, :906-961). */
static void transform_cascade(cascade *cs, double wp, double wl, double wu)
{
    const size_t count = cs->nzeros > cs->npoles ? cs->nzeros : cs->npoles;
    if (wl <= 0.0) {
        /* low-pass: the prototype is already warped at wp == wu (ref :996-997, :588-590) */
        return;
    }
    if (wu <= 0.0) {                                   /* high-pass at wl */
        const double alpha = -cos(M_PI * (wp + wl) / 2.0) / cos(M_PI * (wp - wl) / 2.0);
        const double alpha2 = pow(alpha, 2);
        for (size_t i = 0; i < count; i++) {
            const double num = i < cs->nzeros ? hp_section(&cs->zeros[i], alpha, alpha2) : 1.0;
            const double den = i < cs->npoles ? hp_section(&cs->poles[i], alpha, alpha2) : 1.0;
            cs->gain *= num / den;
        }
        return;
    }
    band_consts q;
    if (wu > wl) {                                     /* band-pass [wl, wu] */
        const double chi = tan(M_PI * wp / 2.0) / tan(M_PI * (wu - wl) / 2.0);
        const double alpha = -2 * cos(M_PI * (wu + wl) / 2.0) / cos(M_PI * (wu - wl) / 2.0) * chi / (chi + 1.0);
        band_constants(&q, alpha, (chi - 1.0) / (chi + 1.0), 0);
    } else {                                           /* band-stop: edges swap roles (ref :1005-1006) */
        const double lo = wu, hi = wl;
        const double chi = tan(M_PI * wp / 2.0) * tan(M_PI * (hi - lo) / 2.0);
        const double alpha = -2 * cos(M_PI * (hi + lo) / 2.0) / cos(M_PI * (hi - lo) / 2.0) / (chi + 1.0);
        band_constants(&q, alpha, (1.0 - chi) / (chi + 1.0), 1);
    }
    for (size_t i = 0; i < count; i++) {
        const double num = i < cs->nzeros ? band_section(&cs->zeros[i], &q) : 1.0;
        const double den = i < cs->npoles ? band_section(&cs->poles[i], &q) : 1.0;
        cs->gain *= num / den;
    }
}

/* ---- cascade -> transfer-function coefficients ----------------------------- */

/* acc (degree *deg, ascending powers of z^-1) *= (1 + c[0] z^-1 + ...), in place,
 * with the accumulation order of the reference's polynomial product
 * (src/polynomials.c:370-402): the top coefficient is a single product; every
 * other one starts at 0.0 and adds acc[p]*f[n-p] for p descending.  A top
 * coefficient that comes out exactly 0.0 lowers the degree (:73-77, :400). */
static void expand_into(double *acc, size_t *deg, const section *s)
{
    double f[5];
    f[0] = 1.0;
    for (int j = 0; j < s->deg; j++) f[j + 1] = s->c[j];
    const size_t no = *deg, ni = (size_t)s->deg;
    size_t n = no + ni;
    acc[n] = acc[no] * f[ni];
    while (n) {
        n--;
        double val = 0.0;
        size_t p = n < no ? n : no;
        for (;;) {
            if (n - p > ni) break;
            val += acc[p] * f[n - p];
            if (p == 0) break;
            p--;
        }
        acc[n] = val;
    }
    size_t top = no + ni;
    while (top && acc[top] == 0.0) top--;
    *deg = top;
}

static void expand_list(double *dst, size_t ncoef, double lead, const section *list, size_t count)
{
    for (size_t i = 0; i < ncoef; i++) dst[i] = 0.0;
    dst[0] = lead;
    size_t deg = 0;
    for (size_t i = 0; i < count; i++) expand_into(dst, &deg, &list[i]);
}

/* Shared tail of the three classical designs. */
typedef void (*prototype_fn)(cascade *cs, size_t order, double ripple, double w0);

static void proto_bw(cascade *cs, size_t order, double ripple, double w0)
{
    (void)ripple;
    prototype_butterworth(cs, order, w0);
}

static prototype_fn family_prototype(int family)
{
    switch (family) {
        case DSP_FAMILY_BUTTERWORTH: return proto_bw;
        case DSP_FAMILY_CHEBYSHEV1: return prototype_chebyshev1;
        case DSP_FAMILY_CHEBYSHEV2: return prototype_chebyshev2;
    }
    return NULL;
}

/* prototype + band transform: the factored design every output form starts from */
static int build_cascade(cascade *cs, size_t order, double ripple, double wl, double wu, prototype_fn proto)
{
    const double w0 = wu < wl ? wl : wu;
    if (cascade_alloc(cs, order)) return 1;
    proto(cs, order, ripple, w0);
    transform_cascade(cs, w0, wl, wu);
    return 0;
}

static size_t design_ncoef(size_t order, double wl, double wu)
{
    return ((wu > 0.0 && wl > 0.0) ? 2 : 1) * order + 1;
}

/* transfer-function coefficients b[0..ncoef), a[0..ncoef) of one design (the arithmetic of the reference's
 * digital_prototype_to_IIRFilterBank, src/RTFilter.c:977-1140) */
static int design_tf(double *b, double *a, size_t ncoef, size_t order, double ripple, double wl, double wu,
                     prototype_fn proto)
{
    cascade cs;
    if (build_cascade(&cs, order, ripple, wl, wu, proto)) return 1;
    expand_list(b, ncoef, cs.gain, cs.zeros, cs.nzeros);
    expand_list(a, ncoef, 1.0, cs.poles, cs.npoles);
    cascade_free(&cs);
    return 0;
}

static int pack_sos(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu, prototype_fn proto);

static int classical_design(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu,
                            unsigned int flags, rtf_sample_fn initialize, prototype_fn proto)
{
    if (!rtif || wl == wu || !order) return 1;
    const size_t ncoef = design_ncoef(order, wl, wu);
    if (rtif->ifb.na < ncoef || rtif->ifb.fb.nb < ncoef) return 2;
    if (flags & FILTER_SOS) return pack_sos(rtif, order, ripple, wl, wu, proto);
    if (!initialize) rtif->rtf.initialize = RTIIRFilter_stable_init;

    double *bank = rtif->ifb.fb.b;
    if (design_tf(bank, bank + ncoef, ncoef, order, ripple, wl, wu, proto)) return 1;
    rtif->ifb.fb.nb = ncoef;
    rtif->ifb.na = ncoef;
    return 0;
}

int butterworth(RTIIRFilter *rtif, size_t order, double wl, double wu, unsigned int flags,
                rtf_sample_fn initialize)
{
    /* the reference stores `flags` and never reads them (ref :1216, :48-53); FILTER_SOS, reserved there "for
     * second order sections execution" (include/RTFilter.h:51,64-66), selects the cascade form here */
    return classical_design(rtif, order, 0.0, wl, wu, flags, initialize, proto_bw);
}

int chebyshev1(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu,
               unsigned int flags, rtf_sample_fn initialize)
{
    return classical_design(rtif, order, ripple, wl, wu, flags, initialize, prototype_chebyshev1);
}

int chebyshev2(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu,
               unsigned int flags, rtf_sample_fn initialize)
{
    return classical_design(rtif, order, ripple, wl, wu, flags, initialize, prototype_chebyshev2);
}

/* Thiran all-pole fractional-delay low-pass (ref: src/RTFilter.c:1459-1495):
 * b = [prod_{i=order+1}^{2 order} i / (2 tau + i)],
 * a[k] = (-1)^k C(order, k) prod_{i=0}^{order} (2 tau + i) / (2 tau + k + i). */
int thiran(RTIIRFilter *rtif, size_t order, double tau, rtf_sample_fn initialize)
{
    if (!rtif || tau == 0 || !order) return 1;
    if (rtif->ifb.fb.nb < 1 || rtif->ifb.na < order + 1) return 2;
    if (!initialize) rtif->rtf.initialize = RTIIRFilter_stable_init;

    double *b = rtif->ifb.fb.b;
    rtif->ifb.fb.nb = 1;
    b[0] = 1.0;
    for (size_t i = order + 1; i < 2 * order + 1; i++) b[0] *= i / (2 * tau + i);
    rtif->ifb.na = order + 1;
    double *a = b + 1;
    size_t binom = 1;
    for (size_t k = 0; k < order + 1; k++) {
        a[k] = (k & 1) ? -1.0 : 1.0;
        if (k) binom = binom * (order - k + 1) / k;
        for (size_t i = 0; i < order + 1; i++) a[k] *= (2 * tau + i) / (2 * tau + k + i);
        a[k] *= binom;
    }
    return 0;
}

/* ======================================================================== */
/* Second-order sections (no reference behaviour)      */

/* 1 + c[0] x + c[1] x^2 + c[2] x^3 + c[3] x^4  =  (1 + p[0] x + p[1] x^2)(1 + q[0] x + q[1] x^2), real factors.
 * Roots of the reversed (monic in z = 1/x) polynomial by Durand-Kerner in long double complex, grouped into
 * conjugate / real pairs, then the quadratic pair polished by Newton steps on the four coefficient equations. */
static void factor_quartic(const double c[4], double p[2], double q[2])
{
    long double complex r[4];
    const long double complex seed = 0.4L + 0.9L * I;
    r[0] = 1.0L;
    for (int k = 1; k < 4; k++) r[k] = r[k - 1] * seed;
    for (int it = 0; it < 200; it++) {
        long double move = 0.0L;
        for (int k = 0; k < 4; k++) {
            const long double complex z = r[k];
            const long double complex f = (((z + c[0]) * z + c[1]) * z + c[2]) * z + c[3];
            long double complex den = 1.0L;
            for (int j = 0; j < 4; j++)
                if (j != k) den *= z - r[j];
            const long double complex step = f / den;
            r[k] = z - step;
            move += cabsl(step);
        }
        if (move < 1e-30L) break;
    }
    /* pair: the root with the largest |imag| and the root closest to its conjugate; the other two remain */
    int i0 = 0;
    for (int k = 1; k < 4; k++)
        if (fabsl(cimagl(r[k])) > fabsl(cimagl(r[i0]))) i0 = k;
    int i1 = -1;
    for (int k = 0; k < 4; k++)
        if (k != i0 && (i1 < 0 || cabsl(r[k] - conjl(r[i0])) < cabsl(r[i1] - conjl(r[i0])))) i1 = k;
    int rest[2], nr = 0;
    for (int k = 0; k < 4; k++)
        if (k != i0 && k != i1) rest[nr++] = k;
    /* (z - u)(z - v) = z^2 - (u+v) z + u v   <=>   1 - (u+v) x + u v x^2 */
    long double P0 = -creall(r[i0] + r[i1]), P1 = creall(r[i0] * r[i1]);
    long double Q0 = -creall(r[rest[0]] + r[rest[1]]), Q1 = creall(r[rest[0]] * r[rest[1]]);
    /* polish: solve  P0+Q0 = c0, P1+Q1+P0 Q0 = c1, P0 Q1+P1 Q0 = c2, P1 Q1 = c3  by Newton (4x4, Gauss) */
    for (int it = 0; it < 4; it++) {
        long double F[4] = {P0 + Q0 - c[0], P1 + Q1 + P0 * Q0 - c[1], P0 * Q1 + P1 * Q0 - c[2], P1 * Q1 - c[3]};
        long double J[4][5] = {{1, 0, 1, 0, -F[0]}, {Q0, 1, P0, 1, -F[1]}, {Q1, Q0, P1, P0, -F[2]}, {0, Q1, 0, P1, -F[3]}};
        int ok = 1;
        for (int col = 0; col < 4 && ok; col++) {
            int piv = col;
            for (int row = col + 1; row < 4; row++)
                if (fabsl(J[row][col]) > fabsl(J[piv][col])) piv = row;
            if (fabsl(J[piv][col]) < 1e-300L) { ok = 0; break; }
            for (int k = 0; k < 5; k++) { long double t = J[col][k]; J[col][k] = J[piv][k]; J[piv][k] = t; }
            for (int row = 0; row < 4; row++) {
                if (row == col) continue;
                const long double m = J[row][col] / J[col][col];
                for (int k = col; k < 5; k++) J[row][k] -= m * J[col][k];
            }
        }
        if (!ok) break;
        P0 += J[0][4] / J[0][0];
        P1 += J[1][4] / J[1][1];
        Q0 += J[2][4] / J[2][2];
        Q1 += J[3][4] / J[3][3];
    }
    p[0] = (double)P0; p[1] = (double)P1;
    q[0] = (double)Q0; q[1] = (double)Q1;
}

/* A list of cascade sections (degrees 1, 2, 4) as quadratic factors 1 + f[0] x + f[1] x^2 plus at most one
 * leftover linear factor (f[1] == 0), linear factors multiplied in pairs.  Returns the factor count. */
static size_t quadratic_factors(const section *list, size_t count, double (*f)[2])
{
    size_t n = 0;
    int have_lin = 0;
    double lin = 0.0;
    for (size_t i = 0; i < count; i++) {
        const section *s = &list[i];
        if (s->deg == 1) {
            if (have_lin) {
                f[n][0] = lin + s->c[0];
                f[n][1] = lin * s->c[0];
                n++;
                have_lin = 0;
            } else {
                lin = s->c[0];
                have_lin = 1;
            }
        } else if (s->deg == 2) {
            f[n][0] = s->c[0];
            f[n][1] = s->c[1];
            n++;
        } else {
            factor_quartic(s->c, f[n], f[n + 1]);
            n += 2;
        }
    }
    if (have_lin) {
        f[n][0] = lin;
        f[n][1] = 0.0;
        n++;
    }
    return n;
}

static int design_sos(RTFilterSection *out, size_t max_sections, size_t *nsections, size_t order, double ripple,
                      double wl, double wu, prototype_fn proto)
{
    cascade cs;
    if (build_cascade(&cs, order, ripple, wl, wu, proto)) return 1;
    const size_t cap = 2 * order + 2;
    double (*num)[2] = (double (*)[2])DSP_MALLOC(sizeof(double[2]) * cap);
    double (*den)[2] = (double (*)[2])DSP_MALLOC(sizeof(double[2]) * cap);
    int rc = 0;
    if (!num || !den) rc = 1;
    if (!rc) {
        const size_t nn = quadratic_factors(cs.zeros, cs.nzeros, num);
        const size_t nd = quadratic_factors(cs.poles, cs.npoles, den);
        const size_t ns = nn > nd ? nn : nd;
        if (nsections) *nsections = ns;
        if (ns > max_sections) rc = 2;
        for (size_t k = 0; k < ns && !rc; k++) {
            const double g = k == 0 ? cs.gain : 1.0;                 /* the design's gain rides on section 0 */
            out[k].b[0] = g;
            out[k].b[1] = k < nn ? g * num[k][0] : 0.0;
            out[k].b[2] = k < nn ? g * num[k][1] : 0.0;
            out[k].a[0] = 1.0;
            out[k].a[1] = k < nd ? den[k][0] : 0.0;
            out[k].a[2] = k < nd ? den[k][1] : 0.0;
        }
    }
    DSP_FREE(num);
    DSP_FREE(den);
    cascade_free(&cs);
    return rc;
}

int dsp_design_sos(int family, size_t order, double ripple, double wl, double wu, RTFilterSection *sections,
                   size_t max_sections, size_t *nsections)
{
    prototype_fn proto = family_prototype(family);
    if (!proto || !order || wl == wu || (!sections && max_sections)) return 1;
    return design_sos(sections, max_sections, nsections, order, ripple, wl, wu, proto);
}

/* FILTER_SOS on a host filter: the sections are packed into the filter's own bank (which the reference sizes
 * for the transfer function: 2 ncoef doubles) as  g | n1' n2' d1 d2 | n1 n2 d1 d2 | ...  (section 0's numerator
 * pre-multiplied by the gain g; a first-order last section takes two doubles: n1 d1), nb = number of sections,
 * na = 1 if the last section is stored short, else 0; two state doubles per section.  The callbacks
 * become RTIIRFilter_sos_update / RTIIRFilter_sos_stable_init (rtfilter_host.c). */
static int pack_sos(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu, prototype_fn proto)
{
    const size_t room = rtif->ifb.fb.nb + rtif->ifb.na;
    RTFilterSection sec[DSP_MAX_SECTIONS];
    size_t ns = 0;
    int rc = design_sos(sec, DSP_MAX_SECTIONS, &ns, order, ripple, wl, wu, proto);
    if (rc) return rc;
    const int short_last = sec[ns - 1].b[2] == 0.0 && sec[ns - 1].a[2] == 0.0;   /* first-order: two doubles */
    if (1 + 4 * ns - (short_last ? 2 : 0) > room) return 2;
    double *bank = rtif->ifb.fb.b;
    bank[0] = sec[0].b[0];
    for (size_t k = 0; k < ns; k++) {
        double *p = bank + 1 + 4 * k;
        if (short_last && k + 1 == ns) {
            p[0] = sec[k].b[1];
            p[1] = sec[k].a[1];
        } else {
            p[0] = sec[k].b[1];
            p[1] = sec[k].b[2];
            p[2] = sec[k].a[1];
            p[3] = sec[k].a[2];
        }
    }
    rtif->ifb.fb.nb = ns;
    rtif->ifb.na = (size_t)short_last;
    rtif->rtf.flags |= FILTER_SOS;
    rtif->rtf.update = RTIIRFilter_sos_update;
    rtif->rtf.initialize = RTIIRFilter_sos_stable_init;
    for (size_t i = 0; i < 2 * ns; i++) rtif->state[i] = 0.0;
    RTFilter_reset(&rtif->rtf);
    return 0;
}

/* ======================================================================== */
/* Many designs at once: the same arithmetic as the    */
/* per-filter routines, one design per work item, over host threads           */

typedef struct batch_job {
    prototype_fn proto;
    size_t n, order, ld;
    const double *ripple, *wl, *wu;
    double *b, *a;
    size_t *ncoef;
    size_t next;                /* shared cursor, advanced in chunks under the mutex */
    int status;
    pthread_mutex_t mu;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *job = (batch_job *)arg;
    for (;;) {
        pthread_mutex_lock(&job->mu);
        const size_t lo = job->next;
        const size_t hi = lo + 64 < job->n ? lo + 64 : job->n;
        job->next = hi;
        pthread_mutex_unlock(&job->mu);
        if (lo >= hi) break;
        for (size_t i = lo; i < hi; i++) {
            const double wl = job->wl ? job->wl[i] : 0.0, wu = job->wu ? job->wu[i] : 0.0;
            const double rip = job->ripple ? job->ripple[i] : 0.0;
            const size_t nc = design_ncoef(job->order, wl, wu);
            int rc = (wl == wu || nc > job->ld) ? (wl == wu ? 1 : 2)
                                                : design_tf(job->b + i * job->ld, job->a + i * job->ld, nc, job->order,
                                                            rip, wl, wu, job->proto);
            if (job->ncoef) job->ncoef[i] = rc ? 0 : nc;
            if (rc) {
                pthread_mutex_lock(&job->mu);
                if (!job->status) job->status = rc;
                pthread_mutex_unlock(&job->mu);
            }
        }
    }
    return NULL;
}

int dsp_design_batch(int family, size_t n, size_t order, const double *ripple, const double *wl, const double *wu,
                     double *b, double *a, size_t ld, size_t *ncoef, int threads)
{
    prototype_fn proto = family_prototype(family);
    if (!proto || !order || !b || !a || (!wl && !wu)) return 1;
    if (family != DSP_FAMILY_BUTTERWORTH && !ripple) return 1;
    if (!n) return 0;
    batch_job job;
    job.proto = proto;
    job.n = n; job.order = order; job.ld = ld;
    job.ripple = ripple; job.wl = wl; job.wu = wu;
    job.b = b; job.a = a; job.ncoef = ncoef;
    job.next = 0;
    job.status = 0;
    pthread_mutex_init(&job.mu, NULL);
    if (threads <= 0) threads = 16;
    if ((size_t)threads > (n + 63) / 64) threads = (int)((n + 63) / 64);
    if (threads > 64) threads = 64;
    pthread_t tid[64];
    int started = 0;
    for (int t = 1; t < threads; t++)
        if (pthread_create(&tid[started], NULL, batch_worker, &job) == 0) started++;
    batch_worker(&job);
    for (int t = 0; t < started; t++) pthread_join(tid[t], NULL);
    pthread_mutex_destroy(&job.mu);
    return job.status;
}

/* ======================================================================== */
/* FIR design (the reference has only moving_average, src/RTFilter.c:569-585)  */

static double window_value(int window, size_t k, size_t taps)
{
    if (taps < 2) return 1.0;
    const double t = 2.0 * M_PI * (double)k / (double)(taps - 1);
    switch (window) {
        case DSP_WINDOW_HAMMING: return 0.54 - 0.46 * cos(t);
        case DSP_WINDOW_HANN: return 0.5 - 0.5 * cos(t);
        case DSP_WINDOW_BLACKMAN: return 0.42 - 0.5 * cos(t) + 0.08 * cos(2.0 * t);
    }
    return 1.0;                                           /* rectangular */
}

static double sinc_pi(double x)                            /* sin(pi x) / (pi x) */
{
    if (x == 0.0) return 1.0;
    const double px = M_PI * x;
    return sin(px) / px;
}

/* Windowed-sinc FIR: ideal response of the band [wl, wu] (units of Nyquist, the conventions of butterworth:
 * wl <= 0 low-pass at wu, wu <= 0 high-pass at wl, wl < wu band-pass, wl > wu band-stop), truncated to `taps`
 * and shaped by `window`; the low-pass / band-stop forms are normalised to unit gain at DC (sum b = 1, which
 * RTFIRFilter_stable_init assumes, src/RTFilter.c:250-265), band-pass to unit gain at the band centre,
 * high-pass at Nyquist.  Writes b[0..taps) into the filter's bank and shrinks nb, like moving_average. */
int fir_windowed_sinc(RTFIRFilter *rtff, size_t taps, double wl, double wu, int window, rtf_sample_fn initialize)
{
    if (!rtff || !taps || wl == wu) return 1;
    if (window < DSP_WINDOW_RECTANGULAR || window > DSP_WINDOW_BLACKMAN) return 1;
    if (rtff->fb.nb < taps) return 2;
    const int lowpass = wl <= 0.0, highpass = !lowpass && wu <= 0.0, stop = !lowpass && !highpass && wl > wu;
    if ((highpass || stop) && !(taps & 1)) return 1;      /* a type II filter has a zero at Nyquist */
    if (!initialize) rtff->rtf.initialize = RTFIRFilter_stable_init;
    double *b = rtff->fb.b;
    const double m = ((double)taps - 1.0) / 2.0;
    const double lo = stop ? wu : wl, hi = stop ? wl : wu;
    for (size_t k = 0; k < taps; k++) {
        const double t = (double)k - m;
        double h;
        if (lowpass) h = wu * sinc_pi(wu * t);
        else if (highpass) h = sinc_pi(t) - wl * sinc_pi(wl * t);
        else {
            h = hi * sinc_pi(hi * t) - lo * sinc_pi(lo * t);
            if (stop) h = sinc_pi(t) - h;
        }
        b[k] = h * window_value(window, k, taps);
    }
    /* normalise at the reference frequency of the form */
    const double wref = (lowpass || stop) ? 0.0 : (highpass ? 1.0 : 0.5 * (wl + wu));
    double re = 0.0, im = 0.0;
    for (size_t k = 0; k < taps; k++) {
        re += b[k] * cos(M_PI * wref * ((double)k - m));
        im += b[k] * sin(M_PI * wref * ((double)k - m));
    }
    const double mag = sqrt(re * re + im * im);
    if (!(mag > 0.0)) return 1;
    for (size_t k = 0; k < taps; k++) b[k] /= mag;
    rtff->fb.nb = taps;
    return 0;
}
