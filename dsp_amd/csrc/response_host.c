/* response_host.c -- frequency response of a filter given as (b, a) or (zeros, poles, gain).
 * Host only; see include/filterutils.h for what each entry point replaces in the reference. */
#include "filterutils.h"

#include <complex.h>
#include <stdlib.h>

/* the grid the reference builds when the caller gives none: df accumulated step by step */
static double *default_grid(size_t ng)
{
    double *f = (double *)malloc(ng * sizeof(double));
    if (!f) return NULL;
    const double df = 1.0 / (double)(ng - 1);
    f[0] = 0.0;
    for (size_t i = 1; i < ng; ++i) f[i] = f[i - 1] + df;
    return f;
}

/* sum_k c[k] w^k by Horner */
static double complex horner(const double *c, size_t n, double complex w)
{
    double complex acc = 0.0;
    for (size_t k = n; k-- > 0;) acc = acc * w + c[k];
    return acc;
}

int filter_response_ab(double complex *gain, size_t ng, double *a, size_t na, double *b, size_t nb, double *freq)
{
    if (!gain) return 1;
    if (!ng) return 0;
    if (!a || !b || !na || !nb) return 1;
    double *grid = freq ? freq : default_grid(ng);
    if (!grid) return 1;
    for (size_t i = 0; i < ng; ++i) {
        const double complex zinv = cexp(-I * grid[i]);
        gain[i] = horner(b, nb, zinv) / horner(a, na, zinv);
    }
    if (!freq) free(grid);
    return 0;
}

int filter_response_pzg(double complex *gain, size_t ng, double complex *zeros, size_t nz, double complex *poles,
                        size_t np, double kgain, double *freq)
{
    if (!gain) return 1;
    if (!ng) return 0;
    double *grid = freq ? freq : default_grid(ng);
    if (!grid) return 1;
    const size_t n = nz > np ? nz : np;
    for (size_t i = 0; i < ng; ++i) {
        const double complex z = cexp(I * grid[i]);
        double complex h = kgain;
        for (size_t j = 0; j < n; ++j) {
            double complex factor = 1.0;
            if (j < nz) factor *= (z - zeros[j]);
            if (j < np) factor /= (z - poles[j]);
            h *= factor;
        }
        gain[i] = h;
    }
    if (!freq) free(grid);
    return 0;
}

static void unwrap(double *gain, double *phase, const double complex *h, size_t ng)
{
    for (size_t i = 0; i < ng; ++i) {
        gain[i] = cabs(h[i]);
        phase[i] = carg(h[i]);
    }
}

int filter_response_ab_noc(double *gain, double *phase, size_t ng, double *a, size_t na, double *b, size_t nb,
                           double *freq)
{
    if (!gain || !phase) return 1;
    if (!ng) return 0;
    double complex *h = (double complex *)malloc(ng * sizeof(double complex));
    if (!h) return 1;
    const int status = filter_response_ab(h, ng, a, na, b, nb, freq);
    if (!status) unwrap(gain, phase, h, ng);
    free(h);
    return status;
}

int filter_response_pzg_noc(double *gain, double *phase, size_t ng, double *zeros_r, double *zeros_i, size_t nz,
                            double *poles_r, double *poles_i, size_t np, double kgain, double *freq)
{
    if (!gain || !phase) return 1;
    if (!ng) return 0;
    double complex *h = (double complex *)malloc((ng + nz + np) * sizeof(double complex));
    if (!h) return 1;
    double complex *zeros = h + ng, *poles = zeros + nz;
    for (size_t i = 0; i < nz; ++i) zeros[i] = zeros_r[i] + I * zeros_i[i];
    for (size_t i = 0; i < np; ++i) poles[i] = poles_r[i] + I * poles_i[i];
    const int status = filter_response_pzg(h, ng, zeros, nz, poles, np, kgain, freq);
    if (!status) unwrap(gain, phase, h, ng);
    free(h);
    return status;
}

int pzg_to_RTIIRFilter(RTIIRFilter *rtif, double complex *poles, size_t N, double complex *zeros, size_t M,
                       double gain, double wl, double wu)
{
    (void)gain;
    if (!rtif || !(poles || zeros) || !(M || N) || wl == wu) return 1;
    return 1;       /* never succeeds in the reference either; see include/filterutils.h */
}
