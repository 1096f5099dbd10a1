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
#include <math.h>
#include <stdlib.h>

#include "RTFilter.h"

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

static int classical_design(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu,
                            rtf_sample_fn initialize, prototype_fn proto)
{
    if (!rtif || wl == wu || !order) return 1;
    const int two_edges = wu > 0.0 && wl > 0.0;
    const size_t ncoef = (two_edges ? 2 : 1) * order + 1;
    if (rtif->ifb.na < ncoef || rtif->ifb.fb.nb < ncoef) return 2;
    if (!initialize) rtif->rtf.initialize = RTIIRFilter_stable_init;

    const double w0 = wu < wl ? wl : wu;
    cascade cs;
    if (cascade_alloc(&cs, order)) return 1;
    proto(&cs, order, ripple, w0);
    transform_cascade(&cs, w0, wl, wu);

    double *bank = rtif->ifb.fb.b;
    expand_list(bank, ncoef, cs.gain, cs.zeros, cs.nzeros);
    expand_list(bank + ncoef, ncoef, 1.0, cs.poles, cs.npoles);
    rtif->ifb.fb.nb = ncoef;
    rtif->ifb.na = ncoef;
    cascade_free(&cs);
    return 0;
}

int butterworth(RTIIRFilter *rtif, size_t order, double wl, double wu, unsigned int flags,
                rtf_sample_fn initialize)
{
    (void)flags;                                   /* accepted, never interpreted (ref :1216) */
    return classical_design(rtif, order, 0.0, wl, wu, initialize, proto_bw);
}

int chebyshev1(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu,
               unsigned int flags, rtf_sample_fn initialize)
{
    (void)flags;
    return classical_design(rtif, order, ripple, wl, wu, initialize, prototype_chebyshev1);
}

int chebyshev2(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu,
               unsigned int flags, rtf_sample_fn initialize)
{
    (void)flags;
    return classical_design(rtif, order, ripple, wl, wu, initialize, prototype_chebyshev2);
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
