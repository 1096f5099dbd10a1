/* iir_steps.h -- the per-sample arithmetic leaves of the IIR kernels. */
#ifndef DSP_IIR_STEPS_H
#define DSP_IIR_STEPS_H

#include "kernel_common.h"

namespace dsp {

/* One RTIIRFilter_update (ref: src/RTFilter.c:396-439) for na == nb == N+1, in
 * the reference's exact operation order:
 *   y    = ((x*b0) + s0) / a0
 *   t    = (0.0 + b[i+1]*x) - a[i+1]*y ;  s[i] = s[i+1] + t      i = 0..N-1
 * The literal `0.0 +` is kept: it turns a -0.0 product into +0.0 as the
 * reference's `state_value = 0.0; state_value += ...` does.  x / 1.0 == x
 * exactly, so kernels instantiated with kUnitA0 (chosen by the host when every
 * design of the launch group has a0 == 1, as all reference designs do:
 * src/RTFilter.c:1156,1255,1352) skip the IEEE divide; the other instantiation
 * always divides.  (A run-time `if (a0 != 1)` is if-converted by the compiler
 * into divide + select, i.e. pays the ~14-instruction divide on every sample.) */
template <int N, bool kUnitA0>
__device__ __forceinline__ double iir_step_ref(double x, double (&s)[N + 1],
                                               const double (&b)[N + 1],
                                               const double (&a)[N + 1])
{
    double t = x * b[0];
    t = t + s[0];
    double y = t;
    if (!kUnitA0) y = t / a[0];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double u = 0.0 + b[i + 1] * x;
        u = u - a[i + 1] * y;
        s[i] = s[i + 1] + u;
    }
    return y;
}

/* Same recurrence with explicit FMAs (2N+1 instructions instead of 4N+3+N):
 * used by the scan kernel only, which is re-associated anyway and is held to
 * the 1e-12 bar, not to bit-exactness. */
template <int N, bool kUnitA0, bool kEvenB = false>
__device__ __forceinline__ double iir_step_fused(double x, double (&s)[N + 1],
                                                 const double (&b)[N + 1],
                                                 const double (&a)[N + 1])
{
    double y = __builtin_fma(x, b[0], s[0]);
    if (!kUnitA0) y = y / a[0];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        /* kEvenB (the host has checked b[1] == b[3] == ... == 0 exactly): s + 0 x is s, no instruction for it */
        const double fed = (kEvenB && ((i + 1) & 1)) ? s[i + 1] : __builtin_fma(b[i + 1], x, s[i + 1]);
        s[i] = __builtin_fma(-a[i + 1], y, fed);
    }
    return y;
}

/* RTIIRFilter_stable_init (ref: src/RTFilter.c:441-475), na == nb == N+1 >= 2. */
template <int N>
__device__ __forceinline__ double iir_seed_ref(double x, double (&s)[N + 1],
                                               const double (&b)[N + 1],
                                               const double (&a)[N + 1])
{
    double num = 0.0 + b[0];
    num = num + b[1];
    double den = 0.0 + a[0];
    den = den + a[1];
    const double d = num / den;
    double run = 0.0;
    s[N] = 0.0;
#pragma unroll
    for (int i = N; i >= 1; --i) {
        run = run + b[i];
        run = run - a[i] * d;
        s[i - 1] = run * x;
    }
    return d * x;
}

/* ---- second-order sections (RTFB_KIND_SOS, include/RTFilterDesign.h) ------------------------------------------
 * S sections; section k is an RTIIRFilter with nb = na = 3 (taps sb[3k..3k+2], sa[3k..3k+2]) run on the output of
 * section k-1 -- per SAMPLE here, so that a block is read and written once for the whole cascade (round 2 ran
 * one pass over memory per section). */

/* reference operation order per section (iir_step_ref<2>), state 3 slots per section, s[3k+2] read, never written */
template <int S, bool kUnitA0>
__device__ __forceinline__ double sos_step_ref(double x, double (&s)[3 * S], const double (&sb)[3 * S],
                                               const double (&sa)[3 * S])
{
#pragma unroll
    for (int k = 0; k < S; ++k) {
        double t = x * sb[3 * k];
        t = t + s[3 * k];
        double y = t;
        if (!kUnitA0) y = t / sa[3 * k];
        double u = 0.0 + sb[3 * k + 1] * x;
        u = u - sa[3 * k + 1] * y;
        s[3 * k] = s[3 * k + 1] + u;
        u = 0.0 + sb[3 * k + 2] * x;
        u = u - sa[3 * k + 2] * y;
        s[3 * k + 1] = s[3 * k + 2] + u;
        x = y;
    }
    return x;
}

/* RTIIRFilter_stable_init of every section on the first sample of ITS input (the seeded output of the one before) */
template <int S>
__device__ __forceinline__ double sos_seed_ref(double x, double (&s)[3 * S], const double (&sb)[3 * S],
                                               const double (&sa)[3 * S])
{
#pragma unroll
    for (int k = 0; k < S; ++k) {
        double s3[3], b3[3] = {sb[3 * k], sb[3 * k + 1], sb[3 * k + 2]}, a3[3] = {sa[3 * k], sa[3 * k + 1], sa[3 * k + 2]};
        x = iir_seed_ref<2>(x, s3, b3, a3);
        s[3 * k] = s3[0];
        s[3 * k + 1] = s3[1];
        s[3 * k + 2] = s3[2];
    }
    return x;
}

/* The cascade as ONE state-space system of 2 S states z = (s^1_0, s^1_1, s^2_0, ...) for the time-parallel kernel
 * (fused multiply-adds; the third slot of a section is 0 in every state the library produces and is not an input
 * here); z has room for the kernel's tail slot z[2S], unused. */
template <int S, bool kUnitA0>
__device__ __forceinline__ double sos_step_fused(double x, double (&z)[2 * S + 1], const double (&sb)[3 * S],
                                                 const double (&sa)[3 * S])
{
#pragma unroll
    for (int k = 0; k < S; ++k) {
        double y = __builtin_fma(x, sb[3 * k], z[2 * k]);
        if (!kUnitA0) y = y / sa[3 * k];
        z[2 * k] = __builtin_fma(-sa[3 * k + 1], y, __builtin_fma(sb[3 * k + 1], x, z[2 * k + 1]));
        z[2 * k + 1] = __builtin_fma(-sa[3 * k + 2], y, sb[3 * k + 2] * x);
        x = y;
    }
    return x;
}

template <int S>
__device__ __forceinline__ double sos_seed_z(double x, double (&z)[2 * S + 1], const double (&sb)[3 * S],
                                             const double (&sa)[3 * S])
{
#pragma unroll
    for (int k = 0; k < S; ++k) {
        double s3[3], b3[3] = {sb[3 * k], sb[3 * k + 1], sb[3 * k + 2]}, a3[3] = {sa[3 * k], sa[3 * k + 1], sa[3 * k + 2]};
        x = iir_seed_ref<2>(x, s3, b3, a3);
        z[2 * k] = s3[0];
        z[2 * k + 1] = s3[1];
    }
    z[2 * S] = 0.0;
    return x;
}

}  // namespace dsp

#endif
