/*
 * filterutils.h -- frequency-response helpers of the reference's public surface (host only, off the
 * per-sample path).  They are here because the reference's own ctypes binding resolves
 * filter_response_pzg_noc when it is imported (ref: python/cdsp.py:73-74) and python/dsp.py:126-144
 * calls it: a drop-in libdsp.so has to export the family.  Same prototypes as the reference header
 * (ref: include/filterutils.h:4-8).
 *
 * Status codes as in the reference: 0 ok (also for ng == 0: nothing written), 1 NULL output / no memory.
 * freq == NULL selects the reference's default grid freq[i] = i * 1/(ng-1) (accumulated, not multiplied:
 * freq[i] = freq[i-1] + df, ref: src/filterutils.c:57-61).
 */
#ifndef FILTERUTILS_H
#define FILTERUTILS_H

#include "RTFilter.h"

#ifdef __cplusplus
extern "C" {
#endif

#if !defined(__STDC_NO_COMPLEX__) && !defined(__cplusplus)
#include <complex.h>
/* gain[i] = B(z^-1) / A(z^-1) at z = exp(i freq[i]), b and a in ascending powers of z^-1.
 * DELIBERATE DIFFERENCE: the reference divides the numerator polynomial by itself (always 1+0i) and
 * dereferences a NULL grid when the caller passes `freq` (ref: src/filterutils.c:14-16, 35-36); this
 * build evaluates the transfer function the comment above those lines describes. */
EXPORT int filter_response_ab(double complex *gain, size_t ng, double *a, size_t na, double *b, size_t nb,
                              double *freq);
/* gain[i] = kgain * prod_j (z - zeros[j]) / (z - poles[j]), z = exp(i freq[i]); factor j carries zero j and
 * pole j together, in index order (ref: src/filterutils.c:66-81) -- same operations, same bits. */
EXPORT int filter_response_pzg(double complex *gain, size_t ng, double complex *zeros, size_t nz,
                               double complex *poles, size_t np, double kgain, double *freq);
/* pzg_to_RTIIRFilter is declared in RTFilter.h, where the reference has it (ref: include/RTFilter.h:180-181). */
#endif

/* The same without complex types in the signature: gain = |H|, phase = arg H in [-pi, pi]
 * (ref: src/filterutils.c:93-135, 137-191). */
EXPORT int filter_response_ab_noc(double *gain, double *phase, size_t ng, double *a, size_t na, double *b,
                                  size_t nb, double *freq);
EXPORT int filter_response_pzg_noc(double *gain, double *phase, size_t ng, double *zeros_r, double *zeros_i,
                                   size_t nz, double *poles_r, double *poles_i, size_t np, double kgain,
                                   double *freq);

#ifdef __cplusplus
}
#endif

#endif /* FILTERUTILS_H */
