/*
 * oracle/rtfilter_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the bondeje/DSP RTFilter hot path: the sample-by-sample
 * transposed-direct-form-II FIR/IIR update that RTFilter_updaten drives.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  Nothing under dsp_amd/ links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here bit-for-bit against (a) the known-answer vectors of SURVEY.md 8(a),
 * and (b) tests/golden/ fixtures captured from the reference itself compiled
 * from /root/reference by oracle/Makefile (target _ref/libdsp_ref.so).
 *
 * Arithmetic contract (reference build: src/make_dll.mak:16, -std=c99 -O3,
 * no -march): IEEE binary64, round-to-nearest, one rounding per operator,
 * no FMA contraction.  This file is compiled with -ffp-contract=off.
 *
 * Reference line citations are to /root/reference/src/RTFilter.c.
 */
#include <stddef.h>
#include <stdint.h>

/* init_mode values shared with tests (mirrors which `initialize` callback the
 * reference filter carries). */
#define ORC_INIT_IIR_STABLE   0   /* RTIIRFilter_stable_init   :441-475 */
#define ORC_INIT_FIR_STABLE   1   /* RTFIRFilter_stable_init   :250-265 */
#define ORC_INIT_FIR_PARTIAL  2   /* RTFIRFilter_partial_init  :273-297 */

#define ORC_KIND_IIR 0
#define ORC_KIND_FIR 1

/* One channel's mutable fields, as the reference keeps them in RTFilter /
 * RT{IIR,FIR}Filter (include/RTFilter.h:30-59). */
typedef struct orc_chan {
    double *state;          /* max(na,nb) doubles (IIR) or nb doubles (FIR) */
    double  filtered_value; /* RTFilter.filtered_value */
    int     initialized;    /* RTFilter.initialized: -1 reset, 0 running, >0 partial countdown */
} orc_chan;

static size_t orc_max(size_t p, size_t q) { return p >= q ? p : q; }

/* RTIIRFilter_update, :396-439.  bank layout b[0..nb) then a[0..na)
 * (include/RTFilter_.h:23-25) is resolved by the caller. */
double orc_iir_update(const double *b, size_t nb, const double *a, size_t na,
                      double *s, double x)
{
    size_t order = orc_max(na, nb) - 1;
    double y = (x * b[0] + s[0]) / a[0];                 /* :417 */
    for (size_t i = 0; i < order; ++i) {                 /* :422-433 */
        double t = 0.0;
        if (i + 1 < nb) t += b[i + 1] * x;
        if (i + 1 < na) t -= a[i + 1] * y;
        s[i] = s[i + 1] + t;
    }
    return y;
}

/* RTIIRFilter_stable_init, :441-475.  Returns filtered_value. */
double orc_iir_stable_init(const double *b, size_t nb, const double *a, size_t na,
                           double *s, double x)
{
    size_t i = orc_max(na, nb) - 1;
    double num = 0.0, den = 0.0;
    for (size_t j = 0; j <= 1; ++j) {                    /* :450-457, only j = 0,1 */
        if (j < nb) num += b[j];
        if (j < na) den += a[j];
    }
    double d = num / den;                                /* :458 */
    s[i] = 0.0;                                          /* :460 */
    double run = 0.0;
    while (i) {                                          /* :462-471 */
        if (i < nb) run += b[i];
        if (i < na) run -= a[i] * d;
        --i;
        s[i] = run * x;
    }
    return d * x;                                        /* :472 */
}

/* RTFIRFilter_update, :235-247. */
double orc_fir_update(const double *b, size_t nb, double *s, double x)
{
    double y = x * b[0] + s[0];                          /* :240 */
    for (size_t i = 0; i + 1 < nb; ++i)                  /* :242-245 */
        s[i] = s[i + 1] + b[i + 1] * x;
    return y;
}

/* RTFIRFilter_stable_init, :250-265. */
double orc_fir_stable_init(const double *b, size_t nb, double *s, double x)
{
    size_t i = nb - 1;
    s[i] = 0.0;
    double run = 0.0;
    while (i) {
        run += b[i];
        --i;
        s[i] = run * x;
    }
    return x;                                            /* :262 */
}

/* RTFIRFilter_partial_init, :273-297.  Mutates *initialized like the
 * reference mutates rtf->initialized. */
double orc_fir_partial_init(const double *b, size_t nb, double *s, double x,
                            int *initialized, double prev_value)
{
    double y = prev_value;
    size_t order = nb;
    if (*initialized == -1) {                            /* :279-282 */
        s[order - 1] = 0.0;
        y = x;
        *initialized = (int)(order - 1);
    } else {                                             /* :283-286 */
        *initialized -= 1;
        y = order * (b[0] * x + s[0]) / (order - *initialized);
    }
    for (size_t i = 0; i + 1 < order; ++i)               /* :287-291 */
        s[i] = s[i + 1] + b[i + 1] * x;
    return y;
}

/* RTFilter_update + RTFilter_updaten (:98-119) for one channel whose
 * callbacks are the library's own (kind/init_mode name them).
 * in/out may alias.  Returns 0, or 1 on NULL args like :112-114. */
int orc_updaten(int kind, int init_mode,
                const double *b, size_t nb, const double *a, size_t na,
                orc_chan *ch, double *out, const double *in, size_t n)
{
    if (!out || !ch || !in) return 1;
    for (size_t k = 0; k < n; ++k) {
        double x = in[k];
        if (ch->initialized == 0) {                      /* :101-103 */
            ch->filtered_value = (kind == ORC_KIND_IIR)
                ? orc_iir_update(b, nb, a, na, ch->state, x)
                : orc_fir_update(b, nb, ch->state, x);
        } else if (init_mode == ORC_INIT_IIR_STABLE) {   /* :104-106 */
            ch->filtered_value = orc_iir_stable_init(b, nb, a, na, ch->state, x);
            ch->initialized = 0;
        } else if (init_mode == ORC_INIT_FIR_STABLE) {
            ch->filtered_value = orc_fir_stable_init(b, nb, ch->state, x);
            ch->initialized = 0;
        } else {
            ch->filtered_value = orc_fir_partial_init(b, nb, ch->state, x,
                                                      &ch->initialized, ch->filtered_value);
        }
        out[k] = ch->filtered_value;
    }
    return 0;
}

/* Flat-array driver used by tests and the cpu_baseline leg: `nch` channels,
 * channel c reads in + c*ld_in, writes out + c*ld_out, and owns
 * state + c*state_ld, fval[c], init[c].  design[c] indexes per-design tables.
 * This is exactly nch independent orc_updaten calls (the reference has no
 * multichannel entry; a caller loops RTFilter_updaten per filter). */
int orc_bank_updaten(size_t nch, const int *design,
                     const int *kind, const int *init_mode,
                     const size_t *nb, const size_t *na,
                     const size_t *coef_off, const double *coefs,
                     double *state, size_t state_ld, double *fval, int *init,
                     double *out, size_t ld_out, const double *in, size_t ld_in,
                     size_t n, size_t c_begin, size_t c_end)
{
    if (c_end > nch) c_end = nch;
    for (size_t c = c_begin; c < c_end; ++c) {
        int d = design[c];
        const double *b = coefs + coef_off[d];
        const double *a = b + nb[d];
        orc_chan ch = { state + c * state_ld, fval[c], init[c] };
        int rc = orc_updaten(kind[d], init_mode[d], b, nb[d], a, na[d], &ch,
                             out + c * ld_out, in + c * ld_in, n);
        if (rc) return rc;
        fval[c] = ch.filtered_value;
        init[c] = ch.initialized;
    }
    return 0;
}

/* 80-bit "truth" run of the IIR recurrence (same order, long double), used by
 * tests only to REPORT the reference's own float64 rounding floor beside each
 * tolerance check.  State is long double throughout. */
void orc_iir_updaten_ld(const double *b, size_t nb, const double *a, size_t na,
                        long double *s, int *initialized,
                        double *out, const double *in, size_t n)
{
    size_t order = orc_max(na, nb) - 1;
    for (size_t k = 0; k < n; ++k) {
        long double x = in[k], y;
        if (*initialized != 0) {
            long double num = 0, den = 0;
            for (size_t j = 0; j <= 1; ++j) {
                if (j < nb) num += b[j];
                if (j < na) den += a[j];
            }
            long double d = num / den, run = 0;
            size_t i = order;
            s[i] = 0;
            while (i) {
                if (i < nb) run += b[i];
                if (i < na) run -= a[i] * d;
                --i;
                s[i] = run * x;
            }
            y = d * x;
            *initialized = 0;
        } else {
            y = (x * b[0] + s[0]) / a[0];
            for (size_t i = 0; i < order; ++i) {
                long double t = 0;
                if (i + 1 < nb) t += b[i + 1] * x;
                if (i + 1 < na) t -= a[i + 1] * y;
                s[i] = s[i + 1] + t;
            }
        }
        out[k] = (double)y;
    }
}

/* Deterministic synthetic input of SURVEY.md 8(d): uniform [-1,1) from
 * splitmix64(seed ^ (c*2^32 + n)).  Same formula as the device generator in
 * dsp_amd/csrc (kept separately there; this copy is the checker's). */
static uint64_t orc_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth(double *dst, uint64_t seed, uint64_t chan, uint64_t n0, size_t n)
{
    for (size_t k = 0; k < n; ++k) {
        uint64_t h = orc_splitmix64(seed ^ ((chan << 32) + (n0 + k)));
        dst[k] = (double)(int64_t)(h >> 11) * 0x1.0p-52 - 1.0;
    }
}
