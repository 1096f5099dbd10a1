/*
 * RTFilter.h -- C API of libdsp (dsp_amd build), source-compatible with the
 * bondeje/DSP header of the same name.
 *
 * Every declaration below names the reference interface it replaces as
 * "ref: include/RTFilter.h:<line>" (declaration) and, where behaviour matters,
 * "src/RTFilter.c:<lines>" (definition).  Struct layouts are frozen ABI: the
 * reference's ctypes binding (python/cdsp.py:23-52) and C callers instantiate
 * them directly (LP64: RTFilter 40 B, FilterBank 16 B, IIRFilterBank 24 B,
 * RTFIRFilter 64 B, RTIIRFilter 72 B).
 *
 * The per-sample entry points in this header run on the host and produce the
 * reference's bits.  The MI355X path (many channels, device-resident blocks)
 * is the additive API in RTFilterBank.h.
 */
#ifndef RTFILTER_H
#define RTFILTER_H

#include <stddef.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ref: include/RTFilter.h:4 pulls <dsputils.h> (EXPORT = default visibility on Linux) */
#include "dsputils.h"

/* ref: include/RTFilter.h:9-20 -- allocator hooks, overridable at build time */
#ifndef DSP_MALLOC
#  define DSP_MALLOC malloc
#endif
#ifndef DSP_FREE
#  define DSP_FREE free
#endif
#ifndef DSP_CALLOC
#  define DSP_CALLOC calloc
#endif
#ifndef DSP_REALLOC
#  define DSP_REALLOC realloc
#endif

typedef struct RTFilter       RTFilter;
typedef struct RTIIRFilter    RTIIRFilter;
typedef struct RTFIRFilter    RTFIRFilter;
typedef struct IIRFilterState IIRFilterState;
typedef struct IIRFilterBank  IIRFilterBank;
typedef struct FilterBank     FilterBank;

/* Callback shapes of the RTFilter "interface" (ref: RTFilter.h:31-33).  The
 * typedefs are ours; the pointed-to types are identical, so reference callers
 * compile unchanged. */
typedef int  (*rtf_sample_fn)(RTFilter *rtf, double sample);
typedef void (*rtf_del_fn)(RTFilter *rtf);

/* ref: RTFilter.h:30-37.  offsets: 0, 8, 16, 24, 32, 36. */
struct RTFilter {
    rtf_sample_fn update;       /* steady-state step                                   */
    rtf_sample_fn initialize;   /* consumes samples while `initialized != 0`            */
    rtf_del_fn    del;          /* frees bank, state AND the struct (RTFilter.c:299-309) */
    double        filtered_value;
    unsigned int  flags;        /* stored, never interpreted (RTFilter.c:48-53)          */
    int           initialized;  /* FILTER_RESET, FILTER_INITIALIZED, or a >0 countdown   */
};

/* ref: RTFilter.h:39-42 */
struct FilterBank {
    double *b;
    size_t  nb;
};

/* ref: RTFilter.h:44-47.  One allocation holds b[0..nb) then a[0..na)
 * (ref: include/RTFilter_.h:23-25). */
struct IIRFilterBank {
    FilterBank fb;
    size_t     na;
};

/* ref: RTFilter.h:49-53; rtf first so (RTFilter *) casts work (RTFilter.c:396-397) */
struct RTIIRFilter {
    RTFilter      rtf;
    IIRFilterBank ifb;
    double       *state;        /* max(na, nb) doubles */
};

/* ref: RTFilter.h:55-59 */
struct RTFIRFilter {
    RTFilter   rtf;
    FilterBank fb;
    double    *state;           /* nb doubles */
};

/* ref: RTFilter.h:61-70 */
#define FILTER_INFINITE       0x0
#define FILTER_FINITE         0x1
#define FILTER_EXECUTION_MASK 0x2
#define FILTER_TF             0x0
#define FILTER_SOS            0x2
#define FILTER_RESET         -1
#define FILTER_INITIALIZED    0   /* must stay <= 0 */

/* ---- RTFilter: dispatch and batch loop (the hot path's entry) ------------ */
EXPORT void      RTFilter_reset(RTFilter *rtf);                              /* ref :72, c:55-58   */
EXPORT void      RTFilter_init(RTFilter *rtf, unsigned int flags,            /* ref :74-77, c:60-70 */
                               rtf_sample_fn initialize, rtf_sample_fn update, rtf_del_fn del);
EXPORT RTFilter *RTFilter_new(unsigned int flags,                            /* ref :79-82, c:72-83 */
                              rtf_sample_fn initialize, rtf_sample_fn update, rtf_del_fn del);
EXPORT void      RTFilter_del(RTFilter *rtf);                                /* ref :85, c:86-91   */
EXPORT void      RTFilter_initialize(RTFilter *rtf, double sample);          /* ref :87, c:94-96   */
EXPORT double    RTFilter_update(RTFilter *rtf, double sample);              /* ref :88, c:98-108  */
EXPORT int       RTFilter_updaten(double *out, RTFilter *rtf,                /* ref :89, c:111-119 */
                                  double *samples, size_t n);

/* ---- coefficient containers --------------------------------------------- */
EXPORT void        FilterBank_init(FilterBank *fb, double *b, size_t nb);    /* ref :91  */
EXPORT FilterBank *FilterBank_new(double *b, size_t nb);                     /* ref :92 (takes ownership of b) */
EXPORT void        FilterBank_del(FilterBank *fb);                           /* ref :93  */
EXPORT void        FilterBank_print(FilterBank *fb);                         /* ref :95  */

EXPORT void           IIRFilterBank_init(IIRFilterBank *ifb, double *bank, size_t na, size_t nb); /* ref :98  */
EXPORT IIRFilterBank *IIRFilterBank_new_empty(size_t na, size_t nb);                              /* ref :101 */
EXPORT IIRFilterBank *IIRFilterBank_new(double *a, size_t na, double *b, size_t nb);              /* ref :104 */
EXPORT IIRFilterBank *IIRFilterBank_new_coefs(size_t na, size_t nb, ...);   /* ref :107; na then nb doubles */
EXPORT void           IIRFilterBank_del(IIRFilterBank *ifb);                                      /* ref :109 */
EXPORT void           IIRFilterBank_print(IIRFilterBank *ifb);                                    /* ref :111 */

/* ---- FIR ----------------------------------------------------------------- */
EXPORT void         RTFIRFilter_del(RTFilter *rtf);                                               /* ref :113 */
EXPORT void         RTFIRFilter_init(RTFIRFilter *rtff, double *bank, double *state, size_t nb,   /* ref :115-116 */
                                     unsigned int flags, rtf_sample_fn initialize);
EXPORT RTFIRFilter *RTFIRFilter_new_empty(size_t nb, unsigned int flags, rtf_sample_fn initialize); /* ref :118-119 */
EXPORT RTFIRFilter *RTFIRFilter_new(double *b, size_t nb, unsigned int flags,                     /* ref :121-122 */
                                    rtf_sample_fn initialize);
EXPORT int          RTFIRFilter_update(RTFilter *rtf, double sample);        /* ref :124, c:235-247 */
EXPORT int          RTFIRFilter_stable_init(RTFilter *rtf, double sample);   /* ref :126, c:250-265 */
EXPORT int          RTFIRFilter_partial_init(RTFilter *rtf, double sample);  /* ref :129, c:273-297 */
EXPORT void         RTFIRFilter_print(RTFIRFilter *rtff);                    /* ref :131 */

/* ---- IIR ----------------------------------------------------------------- */
EXPORT void         RTIIRFilter_del(RTFilter *rtf);                                               /* ref :133 */
EXPORT void         RTIIRFilter_init(RTIIRFilter *rtif, double *bank, double *state,              /* ref :135-136 */
                                     size_t na, size_t nb, unsigned int flags, rtf_sample_fn initialize);
EXPORT RTIIRFilter *RTIIRFilter_new_empty(size_t na, size_t nb, unsigned int flags,               /* ref :138-139 */
                                          rtf_sample_fn initialize);
EXPORT RTIIRFilter *RTIIRFilter_new(double *a, size_t na, double *b, size_t nb,                   /* ref :141-142 */
                                    unsigned int flags, rtf_sample_fn initialize);
EXPORT int          RTIIRFilter_update(RTFilter *rtf, double sample);        /* ref :144, c:396-439 */
EXPORT int          RTIIRFilter_stable_init(RTFilter *rtf, double sample);   /* ref :146, c:441-475 */
EXPORT void         RTIIRFilter_print(RTIIRFilter *rtif);                    /* ref :148 */

/* ---- host-side design (writes into the filter's existing bank) ----------- */
EXPORT int moving_average(RTFIRFilter *rtff, size_t window, rtf_sample_fn initialize); /* ref :150-151, c:569-585 */

/* wl, wu in units of Nyquist.  wl<=0: low-pass at wu; wu<=0: high-pass at wl;
 * wl<wu: band-pass; wl>wu: band-stop; wl==wu: error 1.  (ref :153-163) */
EXPORT int butterworth(RTIIRFilter *rtif, size_t order, double wl, double wu,          /* ref :162-163, c:1216-1238 */
                       unsigned int flags, rtf_sample_fn initialize);
EXPORT int chebyshev1(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu, /* ref :165-166 */
                      unsigned int flags, rtf_sample_fn initialize);
EXPORT int chebyshev2(RTIIRFilter *rtif, size_t order, double ripple, double wl, double wu, /* ref :168-169 */
                      unsigned int flags, rtf_sample_fn initialize);
EXPORT int thiran(RTIIRFilter *rtif, size_t order, double tau, rtf_sample_fn initialize);   /* ref :171, c:1459-1495 */

#if !defined(__STDC_NO_COMPLEX__) && !defined(__cplusplus)
#include <complex.h>
#ifndef DEFAULT_COMPLEX_TOLERANCE
#define DEFAULT_COMPLEX_TOLERANCE 1e-7                                                      /* ref :176-178 */
#endif
/* ref :180-181, marked "internal only" there: an unfinished routine that prints debug text, overwrites the
 * filter's bank with intermediate values and returns non-zero on every path (src/RTFilter.c:1760-1825).
 * Exported for link compatibility: validates its arguments like the reference (1 on NULL / empty /
 * wl == wu) and otherwise reports failure (1) WITHOUT touching the filter. */
EXPORT int pzg_to_RTIIRFilter(RTIIRFilter *rtif, double complex *poles, size_t N, double complex *zeros,
                              size_t M, double gain, double wl, double wu);
#endif

#ifdef __cplusplus
}
#endif

#endif /* RTFILTER_H */
