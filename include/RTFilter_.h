/*
 * RTFilter_.h -- inline accessors over the public structs of RTFilter.h and the Chebyshev
 * ripple <-> epsilon conversions (ref: include/RTFilter_.h:8-71).  A C caller of the reference may
 * include this header; nothing here is exported from the shared object (the reference's are
 * `static inline` as well, ref :4 "need to fix so that I can export these").
 *
 * Coefficient bank layout the accessors encode (ref :23-25): one array, b[0..nb) followed by
 * a[0..na).
 */
#ifndef RTFILTER_ACCESSORS_H
#define RTFILTER_ACCESSORS_H

#include "RTFilter.h"

/* ref :8-16 */
static inline double *FilterBank_get_b(FilterBank *fb) { return fb->b; }
static inline size_t  FilterBank_get_b_size(FilterBank *fb) { return fb->nb; }
static inline size_t  FIR_order(FilterBank *fb) { return fb->nb; }

/* ref :17-28: a[] starts where b[] ends */
static inline double *IIRFilterBank_get_b(IIRFilterBank *ifb) { return ifb->fb.b; }
static inline size_t  IIRFilterBank_get_b_size(IIRFilterBank *ifb) { return ifb->fb.nb; }
static inline double *IIRFilterBank_get_a(IIRFilterBank *ifb) { return ifb->fb.b + ifb->fb.nb; }
static inline size_t  IIRFilterBank_get_a_size(IIRFilterBank *ifb) { return ifb->na; }

/* ref :29-46 */
static inline double *RTFIRFilter_get_b(RTFIRFilter *rtff) { return rtff->fb.b; }
static inline size_t  RTFIRFilter_get_b_size(RTFIRFilter *rtff) { return rtff->fb.nb; }
static inline double *RTIIRFilter_get_b(RTIIRFilter *rtif) { return rtif->ifb.fb.b; }
static inline size_t  RTIIRFilter_get_b_size(RTIIRFilter *rtif) { return rtif->ifb.fb.nb; }
static inline double *RTIIRFilter_get_a(RTIIRFilter *rtif) { return rtif->ifb.fb.b + rtif->ifb.fb.nb; }
static inline size_t  RTIIRFilter_get_a_size(RTIIRFilter *rtif) { return rtif->ifb.na; }

/* Chebyshev ripple conversions between the epsilon of the textbook definitions (what chebyshev1/2
 * take) and the decibel figures scipy / Matlab take (ref :48-69).
 *   type I  (pass band):  dB = 10 log10(1 + eps^2)        eps = sqrt(10^(dB/10) - 1)
 *   type II (stop band):  dB = 10 log10(1 + 1/eps^2)      eps = 1 / sqrt(10^(dB/10) - 1)            */
static inline double passband_ripple_epsilon_to_db(double epsilon) { return 10.0 * log10(1 + pow(epsilon, 2)); }
static inline double passband_ripple_db_to_epsilon(double db) { return sqrt(pow(10, db / 10.0) - 1); }
static inline double stopband_ripple_epsilon_to_db(double epsilon) { return 10.0 * log10(pow(1 / epsilon, 2) + 1); }
static inline double stopband_ripple_db_to_epsilon(double db) { return 1.0 / sqrt(pow(10, db / 10.0) - 1.0); }

#endif /* RTFILTER_ACCESSORS_H */
