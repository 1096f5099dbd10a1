/*
 * rtfilter_host.c -- host side of libdsp: the RTFilter object model and the
 * per-sample entry points of include/RTFilter.h.
 *
 * These run on the CPU by design: RTFilter_update is one FFI crossing per
 * sample (python/dsp.py:38 of the reference) and RTFilter_updaten on a single
 * host-resident filter is BASELINE.json config 1 ("via CPU DLL").  The
 * MI355X kernels for banks of channels live in bank.hip / kernels.hip and take
 * their coefficient and state buffers from the structs defined here.
 *
 * Arithmetic contract: IEEE binary64, one rounding per operator, no
 * contraction (built with -ffp-contract=off), evaluation order of
 * src/RTFilter.c:235-297 and :396-475 of the reference, so results are
 * bit-identical to it (tests/test_host_api.py checks against tests/golden).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "RTFilter.h"
#include "RTFilterDesign.h"

/* ------------------------------------------------------------------------ */
/* arithmetic leaves (shared by the callback entry points and the fused loop) */

static inline size_t max_sz(size_t p, size_t q) { return p > q ? p : q; }

/* ref: src/RTFilter.c:396-439 */
static inline double iir_advance(const double *b, size_t nb, const double *a, size_t na,
                                 double *s, double x)
{
    const size_t order = max_sz(nb, na) - 1;
    const double y = (x * b[0] + s[0]) / a[0];
    for (size_t i = 0; i < order; i++) {
        double t = 0.0;
        if (i + 1 < nb) t += b[i + 1] * x;
        if (i + 1 < na) t -= a[i + 1] * y;
        s[i] = s[i + 1] + t;
    }
    return y;
}

/* ref: src/RTFilter.c:441-475.  d uses only the first two taps of b and a. */
static inline double iir_seed(const double *b, size_t nb, const double *a, size_t na,
                              double *s, double x)
{
    double num = 0.0, den = 0.0;
    for (size_t j = 0; j < 2; j++) {
        if (j < nb) num += b[j];
        if (j < na) den += a[j];
    }
    const double d = num / den;
    size_t i = max_sz(nb, na) - 1;
    double tail = 0.0;
    s[i] = 0.0;
    for (; i > 0; i--) {
        if (i < nb) tail += b[i];
        if (i < na) tail -= a[i] * d;
        s[i - 1] = tail * x;
    }
    return d * x;
}

/* ref: src/RTFilter.c:235-247 */
static inline double fir_advance(const double *b, size_t nb, double *s, double x)
{
    const double y = x * b[0] + s[0];
    for (size_t i = 1; i < nb; i++)
        s[i - 1] = s[i] + b[i] * x;
    return y;
}

/* ref: src/RTFilter.c:250-265 */
static inline double fir_seed(const double *b, size_t nb, double *s, double x)
{
    double tail = 0.0;
    s[nb - 1] = 0.0;
    for (size_t i = nb - 1; i > 0; i--) {
        tail += b[i];
        s[i - 1] = tail * x;
    }
    return x;
}

/* ------------------------------------------------------------------------ */
/* RTFilter */

void RTFilter_reset(RTFilter *rtf)
{
    rtf->initialized = FILTER_RESET;
    rtf->filtered_value = 0.0;
}

void RTFilter_init(RTFilter *rtf, unsigned int flags, rtf_sample_fn initialize,
                   rtf_sample_fn update, rtf_del_fn del)
{
    rtf->update = update;
    rtf->initialize = initialize;
    rtf->del = del;
    rtf->flags = flags;
    RTFilter_reset(rtf);
}

RTFilter *RTFilter_new(unsigned int flags, rtf_sample_fn initialize, rtf_sample_fn update,
                       rtf_del_fn del)
{
    RTFilter *rtf = (RTFilter *)DSP_MALLOC(sizeof *rtf);
    if (rtf) RTFilter_init(rtf, flags, initialize, update, del);
    return rtf;
}

void RTFilter_del(RTFilter *rtf)
{
    if (rtf) rtf->del(rtf);
}

void RTFilter_initialize(RTFilter *rtf, double sample)
{
    rtf->initialize(rtf, sample);
}

double RTFilter_update(RTFilter *rtf, double sample)
{
    if (rtf->initialized == FILTER_INITIALIZED)
        rtf->update(rtf, sample);
    else
        rtf->initialize(rtf, sample);
    return rtf->filtered_value;
}

int RTFilter_updaten(double *out, RTFilter *rtf, double *samples, size_t n)
{
    if (!out || !rtf || !samples) return 1;
    size_t k = 0;
    /* Samples consumed by an initialiser (or by foreign callbacks) go through
     * the callback exactly as the reference does.  Once a library-owned filter
     * is running, the same arithmetic is applied in a fused loop without the
     * indirect call; in/out may alias since each sample is read before its
     * slot is written. */
    while (k < n) {
        if (rtf->initialized == FILTER_INITIALIZED && rtf->update == RTIIRFilter_update) {
            RTIIRFilter *f = (RTIIRFilter *)rtf;
            const double *b = f->ifb.fb.b;
            const size_t nb = f->ifb.fb.nb, na = f->ifb.na;
            const double *a = b + nb;
            double y = rtf->filtered_value;
            for (; k < n; k++) out[k] = y = iir_advance(b, nb, a, na, f->state, samples[k]);
            rtf->filtered_value = y;
        } else if (rtf->initialized == FILTER_INITIALIZED && rtf->update == RTFIRFilter_update) {
            RTFIRFilter *f = (RTFIRFilter *)rtf;
            double y = rtf->filtered_value;
            for (; k < n; k++) out[k] = y = fir_advance(f->fb.b, f->fb.nb, f->state, samples[k]);
            rtf->filtered_value = y;
        } else {
            out[k] = RTFilter_update(rtf, samples[k]);
            k++;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* coefficient containers */

void FilterBank_print(FilterBank *fb)
{
    for (size_t i = 0; i < fb->nb; i++) printf("%.10f ", fb->b[i]);
    printf("\n");
}

void FilterBank_init(FilterBank *fb, double *b, size_t nb)
{
    if (!fb) return;
    fb->b = b;
    fb->nb = nb;
}

FilterBank *FilterBank_new(double *b, size_t nb)
{
    FilterBank *fb = (FilterBank *)DSP_MALLOC(sizeof *fb);
    if (fb) FilterBank_init(fb, b, nb);
    return fb;
}

void FilterBank_del(FilterBank *fb)
{
    if (!fb) return;
    DSP_FREE(fb->b);
    fb->b = NULL;
    DSP_FREE(fb);
}

void IIRFilterBank_print(IIRFilterBank *ifb)
{
    const double *a = ifb->fb.b + ifb->fb.nb;
    printf("a coefficients (%zu):\n", ifb->na);
    for (size_t i = 0; i < ifb->na; i++) printf("%.10f ", a[i]);
    printf("\nb coefficients (%zu):\n", ifb->fb.nb);
    FilterBank_print(&ifb->fb);
}

void IIRFilterBank_init(IIRFilterBank *ifb, double *bank, size_t na, size_t nb)
{
    if (!ifb) return;
    FilterBank_init(&ifb->fb, bank, nb);
    ifb->na = na;
}

static double *zeroed_doubles(size_t count)
{
    double *p = (double *)DSP_MALLOC(sizeof(double) * count);
    if (p)
        for (size_t i = 0; i < count; i++) p[i] = 0.0;
    return p;
}

IIRFilterBank *IIRFilterBank_new_empty(size_t na, size_t nb)
{
    IIRFilterBank *ifb = (IIRFilterBank *)DSP_MALLOC(sizeof *ifb);
    if (!ifb) return NULL;
    double *bank = zeroed_doubles(na + nb);
    if (!bank) {
        DSP_FREE(ifb);
        return NULL;
    }
    IIRFilterBank_init(ifb, bank, na, nb);
    return ifb;
}

IIRFilterBank *IIRFilterBank_new(double *a, size_t na, double *b, size_t nb)
{
    IIRFilterBank *ifb = IIRFilterBank_new_empty(na, nb);
    if (!ifb) return NULL;
    double *dst = ifb->fb.b;
    for (size_t i = 0; i < nb; i++) dst[i] = b[i];
    for (size_t i = 0; i < na; i++) dst[nb + i] = a[i];
    return ifb;
}

IIRFilterBank *IIRFilterBank_new_coefs(size_t na, size_t nb, ...)
{
    if (!na || !nb) return NULL;
    IIRFilterBank *ifb = IIRFilterBank_new_empty(na, nb);
    if (!ifb) return NULL;
    double *dst = ifb->fb.b;
    va_list ap;
    va_start(ap, nb);
    for (size_t i = 0; i < na; i++) dst[nb + i] = va_arg(ap, double);  /* a first, then b */
    for (size_t i = 0; i < nb; i++) dst[i] = va_arg(ap, double);
    va_end(ap);
    return ifb;
}

void IIRFilterBank_del(IIRFilterBank *ifb)
{
    FilterBank_del(&ifb->fb);   /* fb is the first member: frees the bank and the struct */
}

/* ------------------------------------------------------------------------ */
/* FIR filter object */

int RTFIRFilter_update(RTFilter *rtf, double sample)
{
    RTFIRFilter *f = (RTFIRFilter *)rtf;
    rtf->filtered_value = fir_advance(f->fb.b, f->fb.nb, f->state, sample);
    return 0;
}

int RTFIRFilter_stable_init(RTFilter *rtf, double sample)
{
    RTFIRFilter *f = (RTFIRFilter *)rtf;
    rtf->filtered_value = fir_seed(f->fb.b, f->fb.nb, f->state, sample);
    rtf->initialized = FILTER_INITIALIZED;
    return 0;
}

/* Ramp-in for moving averages (ref: src/RTFilter.c:273-297): the first nb
 * samples are rescaled by nb/(samples seen); `initialized` counts down. */
int RTFIRFilter_partial_init(RTFilter *rtf, double sample)
{
    RTFIRFilter *f = (RTFIRFilter *)rtf;
    const double *b = f->fb.b;
    const size_t order = f->fb.nb;
    if (rtf->initialized == FILTER_RESET) {
        f->state[order - 1] = 0.0;
        rtf->filtered_value = sample;
        rtf->initialized = (int)(order - 1);
    } else {
        rtf->initialized--;
        rtf->filtered_value = order * (b[0] * sample + f->state[0]) / (order - rtf->initialized);
    }
    for (size_t i = 1; i < order; i++)
        f->state[i - 1] = f->state[i] + b[i] * sample;
    return 0;
}

void RTFIRFilter_print(RTFIRFilter *rtff)
{
    printf("FIR order %zu, coefficients:", rtff->fb.nb);
    FilterBank_print(&rtff->fb);
}

void RTFIRFilter_del(RTFilter *rtf)
{
    if (!rtf) return;
    RTFIRFilter *f = (RTFIRFilter *)rtf;
    DSP_FREE(f->fb.b);
    f->fb.b = NULL;
    DSP_FREE(f->state);
    f->state = NULL;
    DSP_FREE(rtf);
}

void RTFIRFilter_init(RTFIRFilter *rtff, double *bank, double *state, size_t nb,
                      unsigned int flags, rtf_sample_fn initialize)
{
    RTFilter_init(&rtff->rtf, flags, initialize, RTFIRFilter_update, RTFIRFilter_del);
    FilterBank_init(&rtff->fb, bank, nb);
    rtff->state = state;
    for (size_t i = 0; i < nb; i++) state[i] = 0.0;
}

RTFIRFilter *RTFIRFilter_new_empty(size_t nb, unsigned int flags, rtf_sample_fn initialize)
{
    if (!nb) return NULL;
    RTFIRFilter *f = (RTFIRFilter *)DSP_MALLOC(sizeof *f);
    if (!f) return NULL;
    double *state = (double *)DSP_MALLOC(sizeof(double) * nb);
    double *bank = state ? zeroed_doubles(nb) : NULL;
    if (!state || !bank) {
        DSP_FREE(state);
        DSP_FREE(f);
        return NULL;
    }
    RTFIRFilter_init(f, bank, state, nb, flags, initialize ? initialize : RTFIRFilter_stable_init);
    return f;
}

RTFIRFilter *RTFIRFilter_new(double *b, size_t nb, unsigned int flags, rtf_sample_fn initialize)
{
    RTFIRFilter *f = RTFIRFilter_new_empty(nb, flags, initialize);
    if (!f) return NULL;
    for (size_t i = 0; i < nb; i++) f->fb.b[i] = b[i];
    return f;
}

/* ------------------------------------------------------------------------ */
/* IIR filter object */

int RTIIRFilter_update(RTFilter *rtf, double sample)
{
    RTIIRFilter *f = (RTIIRFilter *)rtf;
    const double *b = f->ifb.fb.b;
    const size_t nb = f->ifb.fb.nb;
    rtf->filtered_value = iir_advance(b, nb, b + nb, f->ifb.na, f->state, sample);
    return 0;
}

int RTIIRFilter_stable_init(RTFilter *rtf, double sample)
{
    RTIIRFilter *f = (RTIIRFilter *)rtf;
    const double *b = f->ifb.fb.b;
    const size_t nb = f->ifb.fb.nb;
    rtf->filtered_value = iir_seed(b, nb, b + nb, f->ifb.na, f->state, sample);
    rtf->initialized = FILTER_INITIALIZED;
    return 0;
}

void RTIIRFilter_print(RTIIRFilter *rtif)
{
    if (rtif->rtf.update == RTIIRFilter_sos_update) {
        /* FILTER_SOS: the bank holds packed sections (design_host.c: pack_sos), nb = their number */
        RTFilterSection sec[DSP_MAX_SECTIONS];
        size_t ns = 0;
        if (RTIIRFilter_sos_sections(rtif, sec, DSP_MAX_SECTIONS, &ns)) return;
        printf("IIR as %zu second-order sections:\n", ns);
        for (size_t k = 0; k < ns; k++)
            printf("  b = [%.17g, %.17g, %.17g]  a = [%.17g, %.17g, %.17g]\n", sec[k].b[0], sec[k].b[1], sec[k].b[2],
                   sec[k].a[0], sec[k].a[1], sec[k].a[2]);
        return;
    }
    printf("IIR order %zu:\n", max_sz(rtif->ifb.na, rtif->ifb.fb.nb));
    IIRFilterBank_print(&rtif->ifb);
}

void RTIIRFilter_del(RTFilter *rtf)
{
    if (!rtf) return;
    RTIIRFilter *f = (RTIIRFilter *)rtf;
    DSP_FREE(f->ifb.fb.b);
    f->ifb.fb.b = NULL;
    DSP_FREE(f->state);
    f->state = NULL;
    DSP_FREE(f);
}

void RTIIRFilter_init(RTIIRFilter *rtif, double *bank, double *state, size_t na, size_t nb,
                      unsigned int flags, rtf_sample_fn initialize)
{
    RTFilter_init(&rtif->rtf, flags, initialize, RTIIRFilter_update, RTIIRFilter_del);
    IIRFilterBank_init(&rtif->ifb, bank, na, nb);
    rtif->state = state;
    const size_t slots = max_sz(na, nb);
    for (size_t i = 0; i < slots; i++) state[i] = 0.0;
}

RTIIRFilter *RTIIRFilter_new_empty(size_t na, size_t nb, unsigned int flags,
                                   rtf_sample_fn initialize)
{
    if (!na) return NULL;
    RTIIRFilter *f = (RTIIRFilter *)DSP_MALLOC(sizeof *f);
    if (!f) return NULL;
    double *bank = zeroed_doubles(na + nb);
    double *state = bank ? (double *)DSP_MALLOC(sizeof(double) * max_sz(na, nb)) : NULL;
    if (!bank || !state) {
        DSP_FREE(bank);
        DSP_FREE(f);
        return NULL;
    }
    RTIIRFilter_init(f, bank, state, na, nb, flags,
                     initialize ? initialize : RTIIRFilter_stable_init);
    return f;
}

RTIIRFilter *RTIIRFilter_new(double *a, size_t na, double *b, size_t nb, unsigned int flags,
                             rtf_sample_fn initialize)
{
    RTIIRFilter *f = RTIIRFilter_new_empty(na, nb, flags, initialize);
    if (!f) return NULL;
    double *dst = f->ifb.fb.b;
    for (size_t i = 0; i < nb; i++) dst[i] = b[i];
    for (size_t i = 0; i < na; i++) dst[nb + i] = a[i];
    return f;
}

/* ref: src/RTFilter.c:569-585.  Fills `window` equal taps into the filter's
 * existing bank and re-inits the filter around it (flags forced to 0). */
int moving_average(RTFIRFilter *rtff, size_t window, rtf_sample_fn initialize)
{
    if (!rtff || !window || rtff->fb.nb < window) return 1;
    double *w = rtff->fb.b;
    for (size_t i = 0; i < window; i++) w[i] = 1.0 / window;
    RTFIRFilter_init(rtff, w, rtff->state, window, 0,
                     initialize ? initialize : RTFIRFilter_partial_init);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* FILTER_SOS: a cascade of second-order sections held in an RTIIRFilter     */
/* (include/RTFilterDesign.h; the reference reserves the flag,               */
/* include/RTFilter.h:51,64-66, and never reads it)                          */

/* Packed bank: g | n1' n2' d1 d2 | n1 n2 d1 d2 | ... [| n1 d1]; section k as the arrays an RTIIRFilter would
 * hold.  `short_last` (the struct's na field): the last section is first-order and stored as two doubles. */
static inline void sos_unpack(const double *bank, size_t k, size_t ns, size_t short_last, double b[3], double a[3])
{
    const double *p = bank + 1 + 4 * k;
    b[0] = k == 0 ? bank[0] : 1.0;
    a[0] = 1.0;
    if (short_last && k + 1 == ns) {
        b[1] = p[0]; b[2] = 0.0;
        a[1] = p[1]; a[2] = 0.0;
    } else {
        b[1] = p[0]; b[2] = p[1];
        a[1] = p[2]; a[2] = p[3];
    }
}

/* Every section is RTIIRFilter_update with nb = na = 3 (ref: src/RTFilter.c:396-439); its never-written third
 * state slot is the constant 0.0, kept as an operand so that signed zeros come out as they would there. */
int RTIIRFilter_sos_update(RTFilter *rtf, double sample)
{
    RTIIRFilter *f = (RTIIRFilter *)rtf;
    const double *bank = f->ifb.fb.b;
    const size_t ns = f->ifb.fb.nb;
    double x = sample;
    for (size_t k = 0; k < ns; k++) {
        double b[3], a[3], s[3];
        sos_unpack(bank, k, ns, f->ifb.na, b, a);
        s[0] = f->state[2 * k];
        s[1] = f->state[2 * k + 1];
        s[2] = 0.0;
        x = iir_advance(b, 3, a, 3, s, x);
        f->state[2 * k] = s[0];
        f->state[2 * k + 1] = s[1];
    }
    rtf->filtered_value = x;
    return 0;
}

/* First sample: every section is seeded by RTIIRFilter_stable_init's rule (ref: src/RTFilter.c:441-475) on the
 * sample it sees, i.e. on the seeded output of the section before it. */
int RTIIRFilter_sos_stable_init(RTFilter *rtf, double sample)
{
    RTIIRFilter *f = (RTIIRFilter *)rtf;
    const double *bank = f->ifb.fb.b;
    const size_t ns = f->ifb.fb.nb;
    double x = sample;
    for (size_t k = 0; k < ns; k++) {
        double b[3], a[3], s[3];
        sos_unpack(bank, k, ns, f->ifb.na, b, a);
        x = iir_seed(b, 3, a, 3, s, x);
        f->state[2 * k] = s[0];
        f->state[2 * k + 1] = s[1];
    }
    rtf->filtered_value = x;
    rtf->initialized = FILTER_INITIALIZED;
    return 0;
}

int RTIIRFilter_sos_sections(const RTIIRFilter *rtif, RTFilterSection *sections, size_t max_sections,
                             size_t *nsections)
{
    if (!rtif || rtif->rtf.update != RTIIRFilter_sos_update) return 1;
    const size_t ns = rtif->ifb.fb.nb;
    if (nsections) *nsections = ns;
    if (ns > max_sections || (!sections && ns)) return 2;
    for (size_t k = 0; k < ns; k++) sos_unpack(rtif->ifb.fb.b, k, ns, rtif->ifb.na, sections[k].b, sections[k].a);
    return 0;
}
