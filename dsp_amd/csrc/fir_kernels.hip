/*
 * fir_kernels.hip -- gfx950 kernels for FIR channels (RTFIRFilter_update with
 * RTFIRFilter_stable_init or RTFIRFilter_partial_init), bit-exact.
 *
 * The reference's transposed-form update (ref: src/RTFilter.c:235-247)
 *     y = x*b0 + s0 ;  s[i] = s[i+1] + b[i+1]*x        i = 0..K-2
 * unrolls, for any sample m of a block that starts with carried state s, to
 *     acc  = (m <= K-2) ? s[m] : s[K-1]
 *     acc  = acc + b[k]*x[m-k]     for k = min(m, K-1) down to 1   (oldest tap first)
 *     y[m] = x[m]*b0 + acc
 * with every product and sum rounded separately.  That
 * closed form has no dependence between outputs, so time is split freely:
 * a wavefront takes 64*T consecutive outputs of one channel, stages their
 * inputs plus a K-1 halo in LDS, and each lane accumulates R outputs at once
 * in exactly that order.  Built with -ffp-contract=off.
 *
 * Three launches per block, on one stream:
 *   prologue  per channel: RTFIRFilter_stable_init on sample 0 of a reset stream
 *             (ref :250-265), bookkeeping for the partial_init ramp (:273-297)
 *   main      the outputs
 *   epilogue  the state the reference would hold after the block's last sample,
 *             written to the other half of a ping-pong pair; filtered_value,
 *             initialized
 */
#include "dsp_internal.h"
#include "kernel_common.h"

#include "RTFilterBank.h"

namespace dsp {

namespace {

constexpr int T = kFirT;
constexpr int R = kFirR;
constexpr int S = kFirSeg;

/* LDS holds the segment's inputs at logical index e = (block-local sample index)
 * - (seg_base - (K-1)), skewed by one double per 32 so that lanes reading at a
 * stride of T = 32 doubles fall on distinct banks. */
__device__ __forceinline__ int skew(int e) { return e + (e >> 5); }

}  // namespace

__global__ void __launch_bounds__(256)
fir_prologue_kernel(GroupView g, BlockIO io)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= g.count) return;
    const int chan = group_channel(g, gi);
    const int dsn = g.design[gi];
    const int K = g.nb[dsn];
    const double *b = g.coefs + g.coef_off[dsn];
    double *s = g.state + (size_t)gi * g.state_ld;
    const int ini = g.init[gi];
    int t0 = 0, ramp0 = K;                       /* ramp0 >= K: no ramp in this block */
    if (g.kind[dsn] == RTFB_KIND_FIR) {
        if (ini != 0) {                          /* RTFIRFilter_stable_init, ref :250-265 */
            const double x0 = io.in[(size_t)chan * io.ld_in];
            double run = 0.0;
            s[K - 1] = 0.0;
            for (int i = K - 1; i >= 1; --i) {
                run = run + b[i];
                s[i - 1] = run * x0;
            }
            io.out[(size_t)chan * io.ld_out] = x0;
            t0 = 1;
        }
    } else {                                     /* RTFIRFilter_partial_init, ref :273-297 */
        if (ini == FILTER_RESET) {
            s[K - 1] = 0.0;
            ramp0 = 0;
        } else if (ini > 0) {
            ramp0 = K - ini;
        }
    }
    g.aux[2 * gi] = t0;
    g.aux[2 * gi + 1] = ramp0;
}

/* kHead: the segment may contain outputs that still draw on the carried state
 * (fewer than K-1 regular samples before them) or on the partial_init ramp. */
template <bool kHead>
__global__ void __launch_bounds__(kWave)
fir_main_kernel(GroupView g, BlockIO io, int seg0)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *lds = reinterpret_cast<double *>(smem_raw);

    const int lane = threadIdx.x;
    const int gi = blockIdx.y;
    const int seg = seg0 + blockIdx.x;
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    const int K = as_const(g.nb)[dsn];
    cdouble_p b = as_const(g.coefs) + as_const(g.coef_off)[dsn];
    const int t0 = g.aux[2 * gi];
    const int ramp0 = g.aux[2 * gi + 1];
    const bool partial = as_const(g.kind)[dsn] == RTFB_KIND_FIR_PARTIAL;

    const long long n = (long long)io.n;
    const long long seg_base = (long long)seg * S;        /* first output of this wavefront */
    const long long in_base = seg_base - (K - 1);           /* block-local index of LDS element 0 */
    const double *x = io.in + (size_t)chan * io.ld_in;
    double *out = io.out + (size_t)chan * io.ld_out;
    const double *st = g.state + (size_t)gi * g.state_ld;
    const double tail = st[K - 1];

    /* stage inputs (zero where the block has no sample) */
    const int total = S + K - 1;
    for (int e = lane; e < total; e += kWave) {
        const long long idx = in_base + e;
        lds[skew(e)] = (idx >= 0 && idx < n) ? x[idx] : 0.0;
    }
    wave_sync();

    const double b0 = b[0];
    double y[T];
#pragma unroll
    for (int blk = 0; blk < T / R; ++blk) {
        const int off = lane * T + blk * R;                 /* output offset inside the segment */
        const long long mb = seg_base + off;                /* block-local index of output 0    */
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = tail;
            if (kHead) {
                const long long mp = mb + r - t0;
                if (mp >= 0 && mp <= K - 2) acc[r] = st[mp];
            }
        }
        /* u = r - k: LDS element (off + K-1 + u) feeds output r through tap k = r - u */
        for (int u = -(K - 1); u <= R - 2; ++u) {
            const double xu = lds[skew(off + K - 1 + u)];
            const bool x_ok = !kHead || (mb + u >= t0);
            if (u >= R - K && u <= -1) {                    /* every r has a valid tap */
                cdouble_p bu = b - u;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double p = bu[r] * xu;
                    if (kHead) acc[r] = x_ok ? acc[r] + p : acc[r];
                    else acc[r] = acc[r] + p;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int k = r - u;
                    if (k >= 1 && k <= K - 1) {
                        const double p = b[k] * xu;
                        if (kHead) acc[r] = x_ok ? acc[r] + p : acc[r];
                        else acc[r] = acc[r] + p;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double xr = lds[skew(off + K - 1 + r)];
            double v = xr * b0;
            v = v + acc[r];
            if (kHead && partial) {                         /* ramp of RTFIRFilter_partial_init */
                const long long sidx = (long long)ramp0 + mb + r;   /* stream index */
                if (ramp0 < K && sidx <= K - 1) {
                    if (sidx == 0) v = xr;
                    else {
                        v = (double)(size_t)K * v;
                        v = v / (double)(size_t)(sidx + 1);
                    }
                }
            }
            y[blk * R + r] = v;
        }
    }
    wave_sync();

    /* park outputs in LDS (same skew), then store coalesced */
#pragma unroll
    for (int j = 0; j < T; ++j) lds[skew(lane * T + j)] = y[j];
    wave_sync();
    for (int e = lane; e < S; e += kWave) {
        const long long m = seg_base + e;
        if (m >= t0 && m < n) out[m] = lds[skew(e)];
    }
}

__global__ void __launch_bounds__(256)
fir_epilogue_kernel(GroupView g, BlockIO io)
{
    const int gi = blockIdx.x;
    const int chan = group_channel(g, gi);
    const int dsn = g.design[gi];
    const int K = g.nb[dsn];
    const double *b = g.coefs + g.coef_off[dsn];
    const double *s_in = g.state + (size_t)gi * g.state_ld;
    double *s_out = g.state_out + (size_t)gi * g.state_ld;
    const int t0 = g.aux[2 * gi];
    const int ramp0 = g.aux[2 * gi + 1];
    const long long n = (long long)io.n;
    const long long regular = n - t0;            /* samples that went through the plain update */
    const double *x = io.in + (size_t)chan * io.ld_in;
    const double tail = s_in[K - 1];

    /* s'[i] = ((base + b[i+J] x[n-J]) + ... ) + b[i+1] x[n-1],  J = min(regular, K-1-i),
     * base = s[K-1] when the chain reaches the end of the delay line, else the carried s[i+regular] */
    for (int i = threadIdx.x; i < K - 1; i += blockDim.x) {
        const long long room = K - 1 - i;
        const long long J = regular < room ? regular : room;
        double acc = (J == room) ? tail : s_in[i + regular];
        for (long long j = J; j >= 1; --j) acc = acc + b[i + j] * x[n - j];
        s_out[i] = acc;
    }
    if (threadIdx.x == 0) {
        s_out[K - 1] = tail;
        g.fval[gi] = io.out[(size_t)chan * io.ld_out + n - 1];
        int ini = 0;
        if (g.kind[dsn] == RTFB_KIND_FIR_PARTIAL && ramp0 < K) {
            const long long last = (long long)ramp0 + n - 1;    /* stream index of the last sample */
            ini = last >= K - 1 ? 0 : (int)(K - 1 - last);
        }
        g.init[gi] = ini;
    }
}

hipError_t launch_fir(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    const int nseg = (int)((io.n + S - 1) / S);
    const int pro_blocks = (g.count + 255) / 256;
    hipLaunchKernelGGL(fir_prologue_kernel, dim3(pro_blocks), dim3(256), 0, s, g, io);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    const int total = S + g.kmax - 1;
    const size_t lds_bytes = (size_t)(total + (total >> 5) + 1) * sizeof(double);
    /* segments whose outputs can touch the carried state or the ramp: m < K + 1 */
    int nhead = (g.kmax + 1 + S - 1) / S;
    if (nhead > nseg) nhead = nseg;
    if (nhead > 0) {
        hipLaunchKernelGGL(fir_main_kernel<true>, dim3(nhead, g.count), dim3(kWave), lds_bytes, s, g,
                           io, 0);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (nseg > nhead) {
        hipLaunchKernelGGL(fir_main_kernel<false>, dim3(nseg - nhead, g.count), dim3(kWave),
                           lds_bytes, s, g, io, nhead);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fir_epilogue_kernel, dim3(g.count), dim3(256), 0, s, g, io);
    return hipGetLastError();
}

}  // namespace dsp
