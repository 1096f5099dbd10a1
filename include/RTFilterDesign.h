/*
 * RTFilterDesign.h -- additive host-side design entry points of libdsp (dsp_amd build): what SURVEY.md 8(f)
 * lists next to the hot path and the reference does not have.  Filter design stays on the host (north star);
 * everything here is plain C, no device needed.
 *
 *   - second-order sections of the classical designs (the factored form the reference builds and then
 *     multiplies out, src/RTFilter.c:977-1205; its FILTER_SOS flag is reserved but never read,
 *     include/RTFilter.h:51,64-66)
 *   - many designs at once over host threads, bit-identical to butterworth / chebyshev1 / chebyshev2
 *   - a windowed-sinc FIR design routine (the reference has only moving_average, src/RTFilter.c:569-585)
 */
#ifndef RTFILTERDESIGN_H
#define RTFILTERDESIGN_H

#include <stddef.h>
#include "RTFilter.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DSP_FAMILY_BUTTERWORTH 0
#define DSP_FAMILY_CHEBYSHEV1  1
#define DSP_FAMILY_CHEBYSHEV2  2

#define DSP_MAX_SECTIONS 64    /* sections one design may have here (order <= 64 band-pass) */

/* One section y = (b0 + b1 z^-1 + b2 z^-2) / (a0 + a1 z^-1 + a2 z^-2) x, a0 = 1; a first-order section has
 * b2 = a2 = 0.  A section is executed exactly like an RTIIRFilter with nb = na = 3 (src/RTFilter.c:396-475),
 * the cascade feeds section k's output to section k+1, sample by sample. */
typedef struct RTFilterSection {
    double b[3];
    double a[3];
} RTFilterSection;

/* Sections of butterworth / chebyshev1 / chebyshev2 (family = DSP_FAMILY_*; order, ripple, wl, wu as for those
 * routines).  The design's gain rides on section 0's numerator.  *nsections receives the count (also when
 * max_sections is too small: status 2).  0 ok, 1 bad argument, 2 too many sections for the caller's array. */
EXPORT int dsp_design_sos(int family, size_t order, double ripple, double wl, double wu,
                          RTFilterSection *sections, size_t max_sections, size_t *nsections);

/* Host execution of a cascade held in an RTIIRFilter: the callbacks butterworth / chebyshev1 / chebyshev2
 * install when called with FILTER_SOS in `flags`.  The filter's bank then holds the packed sections
 * (g | n1' n2' d1 d2 | n1 n2 d1 d2 | ... [| n1 d1], nb = number of sections, na = 1 if a first-order last
 * section is stored as two doubles, else 0 -- which always fits the bank the reference sizes for the transfer
 * function) and `state` two doubles per section; every section is seeded by RTIIRFilter_stable_init's rule on the first sample it sees. */
EXPORT int RTIIRFilter_sos_update(RTFilter *rtf, double sample);
EXPORT int RTIIRFilter_sos_stable_init(RTFilter *rtf, double sample);
/* Unpack the sections of such a filter (for RTFilterBank_from_filters and for callers that want to look). */
EXPORT int RTIIRFilter_sos_sections(const RTIIRFilter *rtif, RTFilterSection *sections, size_t max_sections,
                                    size_t *nsections);

/* n designs of one family and order: design i has band edges wl[i], wu[i] (a NULL array = all 0) and ripple[i]
 * (ignored for Butterworth), and its coefficients go to b + i*ld, a + i*ld (ld >= 2*order+1 doubles always
 * suffices); ncoef[i] (optional) receives the taps written: order+1 for low/high-pass, 2*order+1 for
 * band-pass/stop, 0 on failure of that design.  Same arithmetic, hence same bits, as the per-filter routines.
 * threads <= 0: up to 16 host threads.  Returns 0, or the first non-zero status of a design. */
EXPORT int dsp_design_batch(int family, size_t n, size_t order, const double *ripple, const double *wl,
                            const double *wu, double *b, double *a, size_t ld, size_t *ncoef, int threads);

#define DSP_WINDOW_RECTANGULAR 0
#define DSP_WINDOW_HAMMING     1
#define DSP_WINDOW_HANN        2
#define DSP_WINDOW_BLACKMAN    3

/* Windowed-sinc FIR design into the filter's existing bank (like moving_average): band conventions of
 * butterworth (wl <= 0: low-pass at wu; wu <= 0: high-pass at wl; wl < wu band-pass; wl > wu band-stop; units of
 * Nyquist), unit gain at DC / Nyquist / band centre.  High-pass and band-stop need an odd tap count.
 * 0 ok, 1 bad argument, 2 bank smaller than `taps`.  initialize == NULL installs RTFIRFilter_stable_init. */
EXPORT int fir_windowed_sinc(RTFIRFilter *rtff, size_t taps, double wl, double wu, int window,
                             rtf_sample_fn initialize);

#ifdef __cplusplus
}
#endif

#endif /* RTFILTERDESIGN_H */
