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
#include <type_traits>

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

/* g.aux: 2 ints per channel (prologue -> main / epilogue), then one 64-bit word per channel -- bit (item & 63) set
 * when the matrix-core kernels staged a non-finite sample for that 4096-output item -- then one int, nonzero when
 * any word is. */
__device__ __forceinline__ unsigned long long *nf_dirty(const GroupView &g)
{
    return reinterpret_cast<unsigned long long *>(g.aux + 2 * (size_t)g.count);
}
__device__ __forceinline__ int *nf_any(const GroupView &g) { return g.aux + 4 * (size_t)g.count; }
__device__ __forceinline__ void nf_mark(const GroupView &g, int gi, int item)
{
    atomicOr(nf_dirty(g) + gi, 1ull << (item & 63));
    atomicOr(nf_any(g), 1);
}
/* Is an Inf / NaN among the values a thread staged?  On the integer side (an FP64 compare per sample runs at the
 * matrix cores' expense): a double is non-finite iff its high word without the sign is >= 0x7FF00000; two running
 * maxima of the high words, one v_max3 each per pair of samples. */
struct NfKey {
    unsigned u = 0;     /* unsigned maximum: >= 0xFFF00000 iff a -Inf / negative NaN was among them */
    int s = 0;          /* signed maximum:   >= 0x7FF00000 iff a +Inf / positive NaN was */
    __device__ __forceinline__ void add(double a, double b)
    {
        const int ha = __double2hiint(a), hb = __double2hiint(b);
        u = max(max(u, (unsigned)ha), (unsigned)hb);
        s = max(max(s, ha), hb);
    }
    __device__ __forceinline__ void add(double a) { add(a, a); }
    __device__ __forceinline__ bool odd() const { return u >= 0xFFF00000u || s >= 0x7FF00000; }
};

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
    nf_dirty(g)[gi] = 0ull;                      /* items of this block that staged an Inf / NaN (matrix-core path) */
    if (gi == 0) *nf_any(g) = 0;
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
    /* with the MFMA kernel on, only channels inside a partial_init ramp keep this exact head */
    if (kHead && g.fir_mfma && !(partial && ramp0 < K)) return;

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

/* Steady-state segments, bit-exact.
 * A workgroup (4 waves) takes S = 2048 consecutive outputs of one channel; thread t owns the output PAIRS
 * (2t, 2t+1) + 512 j, j = 0..3.  Taps go oldest input first (the reference's order), two per step:
 *     tap k   (k even):  acc0 += b[k] x[m0-k],      acc1 += b[k] x[m0-k+1]
 *     tap k-1:           acc0 += b[k-1] x[m0-k+1],  acc1 += b[k-1] x[m0-k+2]
 * so a step needs the aligned input pair it already holds plus the next aligned pair: ONE 16-byte LDS read
 * (consecutive across lanes: conflict-free) per two taps and two outputs, against four 8-byte reads for
 * one output per lane.  Taps are wave-uniform scalars, every product and sum is rounded separately, eight
 * independent accumulation chains per lane, results stored straight from the registers (16 B per lane).
 * The staged inputs start on an even LDS index when K-1-k is even for even k, i.e. they are shifted by
 * one slot when K is even.  History: lane = 32 outputs tiling 2.0 / 1.2 / 0.5 TB/s at 2 / 32 / 128 taps;
 * one output per lane 4.9 / 3.1 / 1.0; this kernel: see tools/fir_sweep.py. */
constexpr int kDirThreads = 256;
constexpr int kDirPairs = S / (2 * kDirThreads);     /* 4 output pairs per thread */

__global__ void __launch_bounds__(kDirThreads)
fir_direct_kernel(GroupView g, BlockIO io, int seg0)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *lds = reinterpret_cast<double *>(smem_raw);

    const int tid = threadIdx.x;
    const int gi = blockIdx.y;
    const int seg = seg0 + blockIdx.x;
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    const int K = as_const(g.nb)[dsn];
    cdouble_p b = as_const(g.coefs) + as_const(g.coef_off)[dsn];
    const long long n = (long long)io.n;
    const long long seg_base = (long long)seg * S;            /* >= K+1: every output is in steady state */
    const double *x = io.in + (size_t)chan * io.ld_in + (seg_base - (K - 1));
    double *out = io.out + (size_t)chan * io.ld_out + seg_base;
    const double tail = as_const(g.state + (size_t)gi * g.state_ld)[K - 1];   /* written by earlier kernels only */

    /* stage x[seg_base-(K-1) .. seg_base+S) at LDS index pad + e; past the end of the block: the last sample
     * again (only outputs that are not stored ever see it) -- unconditional loads, all in flight together */
    const int pad = (K - 1) & 1;
    double *xs = lds + pad;
    const int total = S + K - 1;
    const long long last = n - 1 - (seg_base - (K - 1));
    const int emax = last < total - 1 ? (int)last : total - 1;
    for (int e = tid; e < total; e += kDirThreads) xs[e] = x[e < emax ? e : emax];
    lds_barrier();

    double acc0[kDirPairs], acc1[kDirPairs];
#pragma unroll
    for (int j = 0; j < kDirPairs; ++j) acc0[j] = acc1[j] = tail;
    const double *xp = xs + (K - 1) + 2 * tid;                /* xp[512 j - k] = x[m0_j - k] */
    int k = K - 1;
    if (k & 1) {                                              /* odd leading tap on its own */
        const double bk = b[k];
#pragma unroll
        for (int j = 0; j < kDirPairs; ++j) {
            const double p0 = bk * xp[2 * kDirThreads * j - k];
            const double p1 = bk * xp[2 * kDirThreads * j - k + 1];
            acc0[j] = acc0[j] + p0;
            acc1[j] = acc1[j] + p1;
        }
        --k;
    }
    /* k even: xp[.. - k] is 16-byte aligned */
    d2 cur[kDirPairs];
#pragma unroll
    for (int j = 0; j < kDirPairs; ++j) cur[j] = *reinterpret_cast<const d2 *>(xp + 2 * kDirThreads * j - k);
#pragma unroll 2
    for (; k >= 2; k -= 2) {
        const double bk = b[k], bk1 = b[k - 1];
#pragma unroll
        for (int j = 0; j < kDirPairs; ++j) {
            const d2 nxt = *reinterpret_cast<const d2 *>(xp + 2 * kDirThreads * j - k + 2);
            const double p0 = bk * cur[j].x, p1 = bk * cur[j].y;
            acc0[j] = acc0[j] + p0;
            acc1[j] = acc1[j] + p1;
            const double q0 = bk1 * cur[j].y, q1 = bk1 * nxt.x;
            acc0[j] = acc0[j] + q0;
            acc1[j] = acc1[j] + q1;
            cur[j] = nxt;
        }
    }
    /* k == 0: cur = (x[m0], x[m0+1]) */
    const double b0 = b[0];
    const bool vec = io.aligned16;
#pragma unroll
    for (int j = 0; j < kDirPairs; ++j) {
        double v0 = cur[j].x * b0, v1 = cur[j].y * b0;
        v0 = v0 + acc0[j];
        v1 = v1 + acc1[j];
        const int o = 2 * tid + 2 * kDirThreads * j;
        if (vec && seg_base + o + 1 < n) {
            *reinterpret_cast<d2 *>(out + o) = (d2){v0, v1};
        } else {
            if (seg_base + o < n) out[o] = v0;
            if (seg_base + o + 1 < n) out[o + 1] = v1;
        }
    }
}

/* End state.  s'[i] = ((base + b[i+J] x[n-J]) + ... ) + b[i+1] x[n-1],  J = min(regular, K-1-i),
 * base = s[K-1] when the chain reaches the end of the delay line, else the carried s[i+regular]
 * (oldest input first: the reference's order).  A lane owns kEpiR consecutive elements: at step j
 * they all take x[n-j] (one broadcast LDS read) times the sliding tap window b[i0+j .. i0+j+R-1],
 * which gains one new tap per step -- two LDS reads per kEpiR multiply-adds. */
constexpr int kEpiR = 8;
constexpr int kEpiThreads = 64;

__global__ void __launch_bounds__(kEpiThreads)
fir_epilogue_kernel(GroupView g, BlockIO io)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *lds = reinterpret_cast<double *>(smem_raw);

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

    /* the last min(regular, K-1) inputs, newest first (xl[j] = x[n-j]); the taps at pitch R+1 per R
     * (lanes read them R apart), zero beyond K-1 (never used: those steps are predicated off) */
    /* (xl stays dense although lanes walk their chains 8 steps apart -- xl[j] is a read at a stride of 8 doubles,
     * 8 lanes per bank, the 4.6e8 SQ_LDS_BANK_CONFLICT cycles per launch at 512 taps: skewing it like bl removes
     * the conflicts and costs more in per-step address arithmetic than they do, 1.41 -> 1.52 ms, round 2) */
    double *xl = lds;
    double *bl = lds + K;
    const int depth = regular < K - 1 ? (int)regular : K - 1;
    for (int j = 1 + threadIdx.x; j <= depth; j += kEpiThreads) xl[j] = x[n - j];
    for (int k = threadIdx.x; k < K + kEpiR; k += kEpiThreads) bl[k + k / kEpiR] = k < K ? b[k] : 0.0;
    wave_sync();

    for (int i0 = kEpiR * threadIdx.x; i0 < K - 1; i0 += kEpiR * kEpiThreads) {
        double acc[kEpiR], w[kEpiR];
        int J[kEpiR];
#pragma unroll
        for (int r = 0; r < kEpiR; ++r) {
            const long long room = K - 1 - (i0 + r);            /* <= 0 past the delay line: idle slot */
            J[r] = (int)(regular < room ? regular : (room > 0 ? room : 0));
            acc[r] = (room <= 0 || J[r] == room) ? tail : s_in[i0 + r + regular];
        }
        int j = J[0];                                           /* the longest chain of the R */
#pragma unroll
        for (int r = 0; r < kEpiR; ++r) { const int k = i0 + r + j; w[r] = bl[k + k / kEpiR]; }
        /* steps where some of the R chains have not started yet */
        for (; j >= 1 && j > J[kEpiR - 1]; --j) {
            const double xj = xl[j];
#pragma unroll
            for (int r = 0; r < kEpiR; ++r) {
                const double p = w[r] * xj;
                acc[r] = (j <= J[r]) ? acc[r] + p : acc[r];
            }
#pragma unroll
            for (int r = kEpiR - 1; r >= 1; --r) w[r] = w[r - 1];
            const int k = i0 + j - 1;
            w[0] = bl[k + k / kEpiR];
        }
#pragma unroll 8
        for (; j >= 1; --j) {
            const double xj = xl[j];
#pragma unroll
            for (int r = 0; r < kEpiR; ++r) {
                const double p = w[r] * xj;
                acc[r] = acc[r] + p;
            }
#pragma unroll
            for (int r = kEpiR - 1; r >= 1; --r) w[r] = w[r - 1];
            const int k = i0 + j - 1;
            w[0] = bl[k + k / kEpiR];
        }
#pragma unroll
        for (int r = 0; r < kEpiR; ++r)
            if (i0 + r < K - 1) s_out[i0 + r] = acc[r];
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

/* End state of blocks that hold at least K-1 regular samples (every chain runs its full length from s[K-1]):
 * the same chains, same order of operations, aligned on the TAP instead of on the sample.
 *     s'[i] = ((s[K-1] + b[K-1] x[n-(K-1-i)]) + b[K-2] x[n-(K-2-i)]) + ... + b[i+1] x[n-1]
 * At tap k chain i takes xl[k-i] (xl[j] = x[n-j]); every chain of the block starts at tap K-1 and chain i ends at
 * tap i+1.  A wavefront takes 128 consecutive chains of one channel (a PASS; lane = 2 consecutive chains), so its
 * lanes run nearly the same number of steps -- in fir_epilogue_kernel a lane owns 8 chains out of all K-1 and the
 * wavefront runs as long as the longest of them, half of the lane-steps idle.  The tap is wave-uniform: 16 at a
 * time through scalar loads; the 18 samples a lane needs for 16 steps arrive as nine 16-byte LDS reads at a lane
 * pitch of 16 bytes (conflict-free).  Steps outside the aligned 16-blocks (before the first, after a lane's last)
 * go one at a time with the tap from LDS.  Every product and sum is rounded separately, as in the reference. */
constexpr int kEpiTapR = 2;                               /* chains per lane */
constexpr int kEpiTapPass = kEpiTapR * kWave;            /* chains per wavefront */

__global__ void __launch_bounds__(kWave)
fir_epilogue_taps_kernel(GroupView g, BlockIO io)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *lds = reinterpret_cast<double *>(smem_raw);

    const int lane = threadIdx.x;
    const int gi = blockIdx.y;
    const int c0 = (int)blockIdx.x * kEpiTapPass;            /* first chain of this pass */
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    const int K = as_const(g.nb)[dsn];
    if (c0 > 0 && c0 >= K - 1) return;                        /* (pass 0 always runs: it keeps the books) */
    const double *b = g.coefs + as_const(g.coef_off)[dsn];
    cdouble_p cb = as_const(b);
    const double *s_in = g.state + (size_t)gi * g.state_ld;
    double *s_out = g.state_out + (size_t)gi * g.state_ld;
    const long long n = (long long)io.n;
    const double *x = io.in + (size_t)chan * io.ld_in;
    const double tail = as_const(s_in)[K - 1];

    /* xl[j] = x[n-j] for j = 1 .. K-1-c0 (what chain c0 reaches back to), xl[0] a readable dummy; then the taps */
    const int jmax = K - 1 - c0;
    double *xl = lds;
    double *bl = lds + ((jmax + 2) & ~1);
    for (int j = lane; j <= jmax; j += kWave) xl[j] = j >= 1 ? x[n - j] : 0.0;
    for (int k = lane; k < K; k += kWave) bl[k] = b[k];
    wave_sync();

    const int i0 = c0 + kEpiTapR * lane;                     /* this lane's chains: i0, i0 + 1 */
    double acc0 = tail, acc1 = tail;
    /* one tap for both chains; chain i is inside its run while i + 1 <= k (k <= K-1 always) */
    auto single = [&](int k) {
        const double t = bl[k];
        const int j = k - i0;
        const double x0 = xl[j > 0 ? j : 0], x1 = xl[j > 1 ? j - 1 : 0];
        const double p0 = t * x0, p1 = t * x1;
        acc0 = k >= i0 + 1 ? acc0 + p0 : acc0;
        acc1 = k >= i0 + 2 ? acc1 + p1 : acc1;
    };
    int k = K - 1;
    for (; (k & 15) != 15 && k > c0; --k) single(k);        /* wave-uniform: down to a block boundary */
    /* 16 taps per trip while both chains of the lane need all of them; lanes leave one after the other, the
     * ones still inside agree on k.  (Requesting the next block's operands before the arithmetic of the current
     * one, in a second register set, was measured and is slower: 1.6 vs 1.24 ms at 512 taps -- the other
     * wavefronts of the SIMD hide the wait better than the longer live ranges allow.) */
    while (k - 15 >= i0 + 2) {
        const int ku = __builtin_amdgcn_readfirstlane(k);
        double t[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) t[m] = cb[ku - 15 + m];
        const double *xp = xl + (k - i0 - 17);               /* even index: 16-byte aligned */
        d2 xv[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) xv[m] = *reinterpret_cast<const d2 *>(xp + 2 * m);
        auto xat = [&](int m) { return (m & 1) ? xv[m >> 1].y : xv[m >> 1].x; };
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {              /* tap k - sidx: x0 = xl[k-sidx-i0], x1 = the one before */
            const double p0 = t[15 - sidx] * xat(17 - sidx);
            const double p1 = t[15 - sidx] * xat(16 - sidx);
            acc0 = acc0 + p0;
            acc1 = acc1 + p1;
        }
        k -= 16;
    }
    for (; k >= i0 + 1; --k) single(k);                      /* the lane's last steps (its own k from here on) */

    if (i0 <= K - 2) s_out[i0] = acc0;
    if (i0 + 1 <= K - 2) s_out[i0 + 1] = acc1;
    if (blockIdx.x == 0 && lane == 0) {
        const int ramp0 = g.aux[2 * gi + 1];
        s_out[K - 1] = tail;
        g.fval[gi] = io.out[(size_t)chan * io.ld_out + n - 1];
        int ini = 0;
        if (as_const(g.kind)[dsn] == RTFB_KIND_FIR_PARTIAL && ramp0 < K) {
            const long long last = (long long)ramp0 + n - 1;    /* stream index of the last sample */
            ini = last >= K - 1 ? 0 : (int)(K - 1 - last);
        }
        g.init[gi] = ini;
    }
}

/* ------------------------------------------------------------------------ */
/* Long-tap FIR as a Toeplitz-tile contraction on the FP64 matrix cores         */
/* (v_mfma_f64_16x16x4_f64).  The other MFMA user is the order 9..16 IIR scan   */
/* (iir_scan16_kernel, kernels.hip).                                            */
/*                                                                              */
/* Steady-state outputs (at least K-1 regular samples behind them) are          */
/*     y[m] = s[K-1] + sum_k b[k] x[m-k].                                        */
/* A workgroup of 4 wavefronts takes 4096 consecutive outputs of one channel;   */
/* each wavefront owns 4 accumulator tiles of 256 outputs,                      */
/*     D[i][j] = y[m0 + 256 t + 16 j + i]   (i = row 0..15, j = column 0..15).   */
/* For every offset delta = -12, -8, ..., (a multiple of 16) >= K-1:            */
/*     D += A_delta * B_delta,   A_delta[i][kk] = b[delta + i - kk]   (a 16 x 4 */
/*     Toeplitz tile of the taps, identical for all columns and tiles),         */
/*     B_delta[kk][j] = x[m0 + 256 t + 16 j - delta + kk].                       */
/* Each (i, k) pair meets exactly one delta, so the sum is the convolution;     */
/* 4*(ceil((K-1)/16)+1) steps of 1024 FMAs per 256 outputs (132 for K = 512,    */
/* 97 % useful).  Operands come from LDS: taps zero-padded by 16 on both sides,  */
/* inputs with their halo skewed one double per 16 (row pitch 17) so that the B */
/* fetch is bank-conflict free.  Because the history length staged is a multiple */
/* of 16 and delta advances as -12 + 16 o + 4 u, the skew never carries: every   */
/* operand address is base(lane) + immediate, one pointer bump per 16 MFMAs.     */
/* FMA accumulation in a different order than the reference: NOT bit-exact,     */
/* bar 1e-12 relative to peak (measured ~1e-16); RTFB_MODE_SEQUENTIAL keeps     */
/* every FIR channel on the bit-exact VALU kernel above.  Head segments, the    */
/* prologue and the epilogue (state) are shared with that kernel.  Non-finite   */
/* inputs: the zero taps of the padding multiply up to 15 samples either side   */
/* of a tile (0 x Inf = NaN in outputs the sample never reaches in the           */
/* reference), so an item that staged an Inf / NaN marks itself in a per-channel */
/* word and fir_nf_fixup_kernel, launched behind the contraction kernels,        */
/* computes such items again with the reference's own sums (oldest tap first):   */
/* their outputs are then the reference's bits, non-finite ones included.        */

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kMfmaWaves = 4;
constexpr int kMfmaTiles = 4;                                    /* accumulator tiles per wavefront */
constexpr int kMfmaSeg = kMfmaWaves * kMfmaTiles * 256;          /* outputs per workgroup: 4096 */
constexpr int kPitch = 18;                                       /* LDS doubles per 16 inputs */
__host__ __device__ constexpr int mfma_steps16(int K) { return (K - 1 + 15) / 16 + 1; }   /* outer steps */
__device__ __forceinline__ int xpos(int e) { return e + 2 * (e >> 4); }
constexpr int kMfmaPre = 19;    /* input doubles per thread fetched one item ahead: covers K <= 753 */
constexpr int kMfmaPre2 = 10;   /* the same as 16-byte pairs when the rows allow it (x0 is then 16-byte aligned:
                                   seg_base and hist are multiples of 16) */
constexpr int kMfmaPreElems = 2 * kMfmaPre2 * kMfmaWaves * kWave;   /* 5120 >= 19 * 256: LDS is sized for it */

template <int NT>
struct MfmaOperandsT {
    double a[4];
    double b[4][NT];
};
using MfmaOperands = MfmaOperandsT<kMfmaTiles>;
/* Operand fetch with plain ds_read_b64 (two 32-lane groups, bank = dword mod 64: conflict-free at
 * pitch 18, 2 LDS cycles).  The four delta sub-steps use four base registers the compiler cannot
 * relate, otherwise it pairs neighbouring reads into ds_read2_b64, which is serviced in 16-lane
 * groups on 32 banks: 4x the LDS cycles plus conflicts (measured: LDS 49 % busy, 44 % of it conflicts). */
typedef const __attribute__((address_space(3))) double *lds_cdouble_p;
__device__ __forceinline__ lds_cdouble_p lds_opaque(const double *p)
{
    unsigned off = (unsigned)(uintptr_t)(lds_cdouble_p)p;
    asm volatile("" : "+v"(off));
    return (lds_cdouble_p)(uintptr_t)off;
}
/* Sub-step u of outer step o: its tap column and the input columns of the wavefront's NT tiles, which lie
 * `tile_stride` tiles apart (1: consecutive tiles, the persistent kernel; 4: the short heads, where the
 * four wavefronts take one tile each). */
template <int NT, int kTileStride = 1>
__device__ __forceinline__ void mfma_fetch(MfmaOperandsT<NT> &r, const lds_cdouble_p (&pa)[4],
                                           const lds_cdouble_p (&pb)[4], int u, int o)
{
    r.a[u] = pa[u][16 * o];
#pragma unroll
    for (int t = 0; t < NT; ++t) r.b[u][t] = pb[u][kPitch * 16 * kTileStride * t - kPitch * o];
}
template <int NT>
__device__ __forceinline__ void mfma_apply(d4 (&acc)[NT], const MfmaOperandsT<NT> &r, int u)
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(r.a[u], r.b[u][t], acc[t], 0, 0, 0);
}
}  // namespace

/* Edge items: the segments before `steady_lo` (outputs that still draw on the carried state, the
 * first regular sample, a ramp head handed over by the exact kernel) and from `steady_hi` on (the
 * ragged end of the block).  One workgroup per item, everything guarded.
 * `ramp_first`: first output this kernel writes for a channel that is inside a partial_init ramp
 * (the exact head kernel covers what is before it); every other channel is served from output t0. */
template <int NT>       /* NT = 4: wavefront w owns tiles 4w..4w+3 of the item; NT = 1 (heads of <= 1024 outputs): tile w */
__global__ void __launch_bounds__(kMfmaWaves * kWave)
fir_mfma_edge_kernel(GroupView g, BlockIO io, long long ramp_first, int steady_lo, int steady_hi, long long head_out)
{
    static_assert(NT == 1 || NT == kMfmaTiles, "one tile per wavefront, or the persistent kernel's four");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *lds = reinterpret_cast<double *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int gi = blockIdx.y;
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    const int K = as_const(g.nb)[dsn];
    const double *b = g.coefs + as_const(g.coef_off)[dsn];
    const int t0 = g.aux[2 * gi];
    const int ramp0 = g.aux[2 * gi + 1];
    const bool ramp = as_const(g.kind)[dsn] == RTFB_KIND_FIR_PARTIAL && ramp0 < K;
    const long long first = ramp ? ramp_first : (long long)t0;     /* first output written here */
    const long long n = (long long)io.n;
    const int seg = (int)blockIdx.x < steady_lo ? (int)blockIdx.x : (int)blockIdx.x - steady_lo + steady_hi;
    const long long seg_base = (long long)seg * kMfmaSeg;
    /* the item that holds head_out is shared with the persistent kernel, which writes its outputs from
     * max(head_out, first) on (a multiple of 256: whole accumulator tiles); this kernel stops there */
    const long long stop = (head_out > seg_base && head_out < seg_base + kMfmaSeg) ? head_out : seg_base + kMfmaSeg;
    if (stop <= first) return;                                     /* whole workgroup */
    const double *x = io.in + (size_t)chan * io.ld_in;
    double *out = io.out + (size_t)chan * io.ld_out;
    const double *st = g.state + (size_t)gi * g.state_ld;
    const double tail = st[K - 1];

    const int steps = mfma_steps16(K);                      /* delta = -12 + 16 o + 4 u, o < steps */
    const int hist = 16 * (steps - 1);                      /* inputs staged before seg_base (>= K-1) */
    const int nb_pad = hist + 32;                           /* taps at index k + 16 */
    double *bp = lds;
    double *xs = lds + nb_pad;
    const int span = (int)(stop - seg_base);                /* outputs of this item: 4096, or up to head_out */
    const int total = hist + span + 16;                     /* inputs seg_base-hist .. seg_base+span+15 */

    for (int e = tid; e < nb_pad; e += kMfmaWaves * kWave) {
        const int k = e - 16;
        bp[e] = (k >= 0 && k < K) ? b[k] : 0.0;
    }
    /* samples before t0 were consumed by stable_init (or do not exist): they feed nothing.  The loads are
     * unconditional on a clamped index and the guard is a select afterwards: guarded loads compile to a branch
     * each and wait one by one (this kernel ran at half the bandwidth of the steady one). */
    NfKey key;
    {
        constexpr int kThreads = kMfmaWaves * kWave;
        double xr[kMfmaPre];
        if (span == kMfmaSeg) {
#pragma unroll
            for (int k = 0; k < kMfmaPre; ++k) {
                long long idx = seg_base - hist + tid + kThreads * k;
                idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
                xr[k] = x[idx];
            }
#pragma unroll
            for (int k = 0; k < kMfmaPre; ++k) {            /* LDS holds kMfmaPreElems >= 19 * 256 inputs */
                const long long idx = seg_base - hist + tid + kThreads * k;
                const double v = (idx >= t0 && idx < n) ? xr[k] : 0.0;
                key.add(v);
                xs[xpos(tid + kThreads * k)] = v;
            }
        } else {                                            /* short head: only what its tiles read */
            for (int e = tid; e < total; e += kThreads) {
                long long idx = seg_base - hist + e;
                const bool ok = idx >= t0 && idx < n;
                idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
                const double v = ok ? x[idx] : 0.0;
                key.add(v);
                xs[xpos(e)] = v;
            }
        }
        for (int e = tid + kThreads * kMfmaPre; span == kMfmaSeg && e < total; e += kThreads) {     /* very long filters */
            const long long idx = seg_base - hist + e;
            const double v = (idx >= t0 && idx < n) ? x[idx] : 0.0;
            key.add(v);
            xs[xpos(e)] = v;
        }
    }
    if (__builtin_amdgcn_ballot_w64(key.odd()) && lane == 0) nf_mark(g, gi, seg);   /* fir_nf_fixup_kernel redoes the item */
    __syncthreads();
    if (256 * NT * wave >= span) return;                    /* this wavefront's tiles all lie past the stop */

    const int j = lane & 15, q = lane >> 4;
    d4 acc[NT];
    if (seg_base > K) {                                     /* steady state: s[K-1] under every output */
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (d4){tail, tail, tail, tail};
    } else {                                                /* outputs that still sit on the carried state */
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long mp = seg_base + 256 * (NT * wave + t) + 16 * j + 4 * q + r - t0;
                acc[t][r] = (mp >= 0 && mp <= K - 2) ? st[mp] : tail;
            }
    }

    /* Tile row i' stands for output offset pi(i') = 4 (i' & 3) + (i' >> 2) inside its 16, so that the
     * four results a lane holds (rows q, q+4, q+8, q+12 of column j) are the consecutive outputs
     * 16 j + 4 q + {0,1,2,3}: each lane stores 32 contiguous bytes, a wavefront a contiguous 2 KB tile.
     * A: row i' = j, kk = q: bp[16 + pi(j) - q + delta];  B: column j, kk = q, tile t:
     * xs element hist + 256 t + 16 j + q - delta = 16 (hist/16 + 16 t + j - o) + (12 + q - 4 u), no carry */
    const int pij = 4 * (j & 3) + (j >> 2);
    lds_cdouble_p pa[4], pb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        pa[u] = lds_opaque(bp + (4 + pij - q) + 4 * u);
        pb[u] = lds_opaque(xs + kPitch * (hist / 16 + 16 * NT * wave + j) + q + 4 * (3 - u));
    }
    MfmaOperandsT<NT> r;
#pragma unroll
    for (int u = 0; u < 4; ++u) mfma_fetch<NT>(r, pa, pb, u, 0);
    for (int o = 1; o < steps; ++o) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            mfma_apply<NT>(acc, r, u);
            mfma_fetch<NT>(r, pa, pb, u, o);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) mfma_apply<NT>(acc, r, u);

    /* D[row = q + 4 r][col = j]  ->  output 256 t + 16 j + 4 q + r, straight from the registers */
    const long long m0 = seg_base + 256 * NT * wave + 16 * j + 4 * q;
    const bool whole = first <= seg_base && seg_base + kMfmaSeg <= n && io.aligned16;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long long m = m0 + 256 * t;
        if (seg_base + 256 * (NT * wave + t) >= stop) break;              /* tiles of the persistent kernel */
        if (whole) {
            *reinterpret_cast<d2 *>(out + m) = (d2){acc[t][0], acc[t][1]};
            *reinterpret_cast<d2 *>(out + m + 2) = (d2){acc[t][2], acc[t][3]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (m + r >= first && m + r < n) out[m + r] = acc[t][r];
        }
    }
}

/* One steady item: kMfmaSeg outputs of one channel with the whole window inside the block. */
struct MfmaItem {
    const double *x0;       /* input of the first staged element: x + seg_base - hist (tiled block layout: the
                               channel's first tile, staged element e = sample p0 + e of the block) */
    int p0;                 /* tiled block layout only: seg_base - hist */
    int emax;               /* last staged element that exists (the 16 look-ahead samples of the block's
                               last segment do not: they repeat x[n-1] and only ever meet zero taps) */
    double *out;            /* out + seg_base */
    double tail;            /* s[K-1] */
    const double *b;
    int K, dsn, hist, steps;
    int emin;               /* first staged element that exists (> 0 only in the item that holds the block's
                               first samples: what lies before them only ever meets outputs that are skipped) */
    long long first;        /* block-local index of the first output this kernel writes for the channel */
    /* folded head (the block's first item, groups without partial_init channels): its first K-1 outputs sit on
     * the carried state instead of s[K-1], and what lies before the first regular sample t0 contributes nothing */
    bool head;
    int t0;                 /* 1 when stable_init consumed sample 0 of this block (the prologue wrote its output) */
    const double *st;       /* the channel's carried state s[0..K-1] */
};


template <bool kTiled = false>
__device__ __forceinline__ MfmaItem mfma_item(const GroupView &g, const BlockIO &io, int gi, int seg,
                                              long long head_out, long long ramp_first, bool fold)
{
    MfmaItem it;
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    it.dsn = as_const(g.design)[gi];
    it.K = as_const(g.nb)[it.dsn];
    it.b = g.coefs + as_const(g.coef_off)[it.dsn];
    it.steps = mfma_steps16(it.K);                  /* delta = -12 + 16 o + 4 u, o < steps */
    it.hist = 16 * (it.steps - 1);                  /* inputs staged before seg_base (>= K-1) */
    const long long seg_base = (long long)seg * kMfmaSeg;
    const long long last = (long long)io.n - 1 - (seg_base - it.hist);
    if constexpr (kTiled) {
        /* [n / 2048][nch][2048], io.ld_* = 2048: x0 / out are the channel's rows in the first tile; the 16
         * look-ahead samples (which only ever meet zero taps) are never fetched from the next tile */
        it.x0 = io.in + (size_t)chan * io.ld_in;
        it.p0 = (int)(seg_base - it.hist);
        it.emax = it.hist + kMfmaSeg - 1;
        it.out = io.out + (size_t)chan * io.ld_out;
    } else {
        it.x0 = io.in + (size_t)chan * io.ld_in + (seg_base - it.hist);
        it.p0 = 0;
        it.emax = last < it.hist + kMfmaSeg + 15 ? (int)last : it.hist + kMfmaSeg + 15;
        it.out = io.out + (size_t)chan * io.ld_out + seg_base;
    }
    it.st = g.state + (size_t)gi * g.state_ld;
    it.tail = as_const(it.st)[it.K - 1];            /* written by an earlier kernel */
    it.emin = seg_base < it.hist ? (int)(it.hist - seg_base) : 0;
    it.first = head_out;
    it.head = fold && seg == 0;
    it.t0 = it.head ? as_const(g.aux)[2 * gi] : 0;  /* written by the prologue kernel */
    if (ramp_first > head_out) {                    /* groups with partial_init channels: g.aux is this launch's */
        const bool ramp = as_const(g.kind)[it.dsn] == RTFB_KIND_FIR_PARTIAL && g.aux[2 * gi + 1] < it.K;
        if (ramp) it.first = ramp_first;            /* the exact head kernel wrote everything before it */
    }
    return it;
}

/* D[row = q + 4 r][col = j]  ->  output 256 t + 16 j + 4 q + r, straight from the registers */
template <bool kAligned16>
__device__ __forceinline__ void mfma_store(const d4 (&acc)[kMfmaTiles], double *o0, int skip)
{
#pragma unroll
    for (int t = 0; t < kMfmaTiles; ++t) {
        if (t < skip) continue;                     /* tiles before the channel's first steady output */
        if (kAligned16) {
            *reinterpret_cast<d2 *>(o0 + 256 * t) = (d2){acc[t][0], acc[t][1]};
            *reinterpret_cast<d2 *>(o0 + 256 * t + 2) = (d2){acc[t][2], acc[t][3]};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o0[256 * t + r] = acc[t][r];
        }
    }
}

/* Steady items (segments [steady_lo, steady_hi) of every channel: at least K regular samples behind
 * the first output, the 16 look-ahead samples inside the block), persistent: the grid is sized to
 * what the GPU holds at once and every workgroup walks a contiguous range of items (consecutive
 * segments of a channel, then the next channel).  The inputs of the next item are fetched into
 * registers before the contraction of the current one starts, so the global latency sits under
 * ~34k cycles of MFMA work instead of in front of it.  No guards anywhere: all loads of a thread
 * are in flight together and nothing makes the compiler drain vmcnt inside the item loop. */
/* kTiled: the tiled block layout [n / 2048][nch][2048] (RTFilterBank_updaten_tiled; aligned, n a multiple of the
 * item, every head folded).  seg_in / seg_out = doubles between consecutive tiles of a channel.  Workgroup w takes
 * channels w, w + grid, w + 2 grid, ... and walks each through its items in time order, so that the workgroups of
 * the grid, as long as they are in step, sweep ONE window of grid x 32 KiB per item instead of grid rows a pitch
 * apart (the reason the layout exists).  No schedule here: holding the prefetch of item k + 1 back until
 * launch start + (k + 1) x period, as the pipelined IIR kernel does, was measured on 65 536 x 64K at 32 taps with
 * periods from 5.3 to 6.2 TB/s and never engages -- a workgroup's own item loop (three per CU) sustains 5.1 TB/s
 * at most, every workgroup is late at once and runs free. */
template <bool kAligned16, bool kTiled = false>
__global__ void __launch_bounds__(kMfmaWaves * kWave, 3)      /* 3 waves per SIMD: <= 168 VGPRs */
fir_mfma_kernel(GroupView g, BlockIO io, int steady_lo, int steady_hi, long long head_out, long long ramp_first,
                int fold, size_t seg_in, size_t seg_out)
{
    static_assert(!kTiled || kAligned16, "the tiled block layout is 16-byte aligned by construction");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *lds = reinterpret_cast<double *>(smem_raw);
    constexpr int kThreads = kMfmaWaves * kWave;

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int nseg = steady_hi - steady_lo;
    const long long items = (long long)nseg * g.count;
    long long idx = kTiled ? 0 : items * blockIdx.x / gridDim.x;
    const long long idx_end = kTiled ? (long long)nseg * ((g.count - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x)
                                     : items * (blockIdx.x + 1) / gridDim.x;
    if (idx >= idx_end) return;
    int gi = kTiled ? (int)blockIdx.x : (int)(idx / nseg);
    int seg = kTiled ? 0 : (int)(idx - (long long)gi * nseg);
    /* staged element e of an item */
    auto xptr = [&](const MfmaItem &it, int e) -> const double * {
        if constexpr (kTiled) {
            const int p = it.p0 + e;
            return it.x0 + (size_t)(p >> 11) * seg_in + (p & 2047);
        } else
            return it.x0 + e;
    };

    MfmaItem cur = mfma_item<kTiled>(g, io, gi, steady_lo + seg, head_out, ramp_first, kAligned16 && fold != 0);
    /* emax is odd here (hist + 4096 - 1 when the block ends with the item, else hist + 4096 + 15), so a pair
     * clamped to emax >> 1 never reaches past it; clamped positions only ever meet zero taps */
    double xr[kMfmaPre];
    d2 xr2[kMfmaPre2];
    auto prefetch = [&](const MfmaItem &it) {
        if constexpr (kAligned16) {
#pragma unroll
            for (int k = 0; k < kMfmaPre2; ++k)
                xr2[k] = *reinterpret_cast<const d2 *>(xptr(it, 2 * max(min(tid + kThreads * k, it.emax >> 1), it.emin >> 1)));
        } else {
#pragma unroll
            for (int k = 0; k < kMfmaPre; ++k) xr[k] = *xptr(it, max(min(tid + kThreads * k, it.emax), it.emin));
        }
    };
    prefetch(cur);
    int staged_dsn = -1;
    d4 acc[kMfmaTiles];
    double *pending = nullptr;                                  /* where the previous item's results go */
    int pending_skip = 0;                                       /* ... and how many of this wavefront's tiles do not */
    long long seg_base = (long long)(steady_lo + seg) * kMfmaSeg;

    for (; idx < idx_end; ++idx) {
        double *bp = lds;                                       /* taps at index k + 16 */
        const int nb_pad = cur.hist + 32;
        double *xs = lds + nb_pad;
        const int total = cur.hist + kMfmaSeg + 16;             /* inputs seg_base-hist .. seg_base+4096+15 */

        /* head items: x[0] - s[K-1] and s[0..K-2] - s[K-1], behind the item's inputs (in a steady item the
         * staging below runs over that place: it writes whole rounds of the prefetch) */
        double *sa = xs + xpos(total) + 2;

        lds_barrier();                                        /* the previous item's operand reads are over */
        /* (first, so that the wait is for the prefetch only: the previous item's stores are younger) */
        NfKey key;                      /* Inf / NaN among the staged samples: the item is done again, see nf_mark */
        if (!cur.head) {
            if constexpr (kAligned16) {
#pragma unroll
                for (int k = 0; k < kMfmaPre2; ++k) {      /* elements 2p, 2p+1 share a 16-group: one 16-byte write */
                    *reinterpret_cast<d2 *>(xs + xpos(2 * (tid + kThreads * k))) = xr2[k];
                    key.add(xr2[k].x, xr2[k].y);
                }
            } else {
#pragma unroll
                for (int k = 0; k < kMfmaPre; ++k) {
                    xs[xpos(tid + kThreads * k)] = xr[k];
                    key.add(xr[k]);
                }
            }
            for (int e = tid + kThreads * (kAligned16 ? 2 * kMfmaPre2 : kMfmaPre); e < total; e += kThreads) {   /* very long filters */
                const double v = *xptr(cur, max(min(e, cur.emax), cur.emin));
                key.add(v);
                xs[xpos(e)] = v;
            }
        } else {
            /* The block's first item: zeros where there is no regular sample (before t0), nothing written past
             * the item's inputs, and the state loaded here, where the prefetch registers are free again -- the
             * other workgroups of the CU keep the matrix cores busy meanwhile. */
            const int zmin = cur.hist + cur.t0;
            if constexpr (kAligned16) {
#pragma unroll
                for (int k = 0; k < kMfmaPre2; ++k) {
                    const int e = 2 * (tid + kThreads * k);
                    d2 v = xr2[k];
                    v.x = e < zmin ? 0.0 : v.x;
                    v.y = e + 1 < zmin ? 0.0 : v.y;
                    key.add(v.x, v.y);
                    if (e < total) *reinterpret_cast<d2 *>(xs + xpos(e)) = v;      /* total is even */
                }
            } else {
#pragma unroll
                for (int k = 0; k < kMfmaPre; ++k) {
                    const int e = tid + kThreads * k;
                    const double v = e < zmin ? 0.0 : xr[k];
                    key.add(v);
                    if (e < total) xs[xpos(e)] = v;
                }
            }
            for (int e = tid + kThreads * (kAligned16 ? 2 * kMfmaPre2 : kMfmaPre); e < total; e += kThreads) {
                double v = *xptr(cur, max(min(e, cur.emax), cur.emin));
                v = e < zmin ? 0.0 : v;
                key.add(v);
                xs[xpos(e)] = v;
            }
            for (int e = tid; e < cur.K; e += kThreads) sa[e] = (e == 0 ? *xptr(cur, cur.hist) : cur.st[e - 1]) - cur.tail;
        }
        if (cur.dsn != staged_dsn) {
            for (int e = tid; e < nb_pad; e += kThreads) {
                const int k = e - 16;
                bp[e] = (k >= 0 && k < cur.K) ? cur.b[k] : 0.0;
            }
            staged_dsn = cur.dsn;
        }
        /* The previous item's results leave only now: issued before the prefetch they would sit
         * between it and the wait above, and the compiler, joining the loop entry (no stores) with
         * the back edge, would wait for them too. */
        if (__builtin_amdgcn_ballot_w64(key.odd()) && lane == 0) nf_mark(g, gi, steady_lo + seg);
        if (pending) mfma_store<kAligned16>(acc, pending, pending_skip);
        lds_barrier();

        /* next item: its inputs travel while this one is contracted */
        /* tiles of this item that lie before the channel's first steady output (whole tiles: head_out and
         * ramp_first are multiples of 256) */
        const long long lead = cur.first - seg_base - 256 * kMfmaTiles * wave;
        const int cur_skip = lead <= 0 ? 0 : (lead >= 256 * kMfmaTiles ? kMfmaTiles : (int)(lead >> 8));
        MfmaItem nxt = cur;
        const long long cur_base = seg_base;
        if (idx + 1 < idx_end) {
            if (++seg == nseg) {                                /* next channel: a chain of scalar loads */
                seg = 0;
                gi += kTiled ? (int)gridDim.x : 1;
                nxt = mfma_item<kTiled>(g, io, gi, steady_lo, head_out, ramp_first, kAligned16 && fold != 0);
            } else {                                            /* next segment of the same channel */
                nxt.head = false;
                const long long nb = (long long)(steady_lo + seg) * kMfmaSeg;
                if constexpr (kTiled) {
                    nxt.p0 += kMfmaSeg;
                    nxt.emin = 0;                               /* hist <= 2048 < one item */
                } else {
                    nxt.x0 += kMfmaSeg;
                    nxt.out += kMfmaSeg;
                    const long long last = (long long)io.n - 1 - (nb - cur.hist);
                    nxt.emax = last < cur.hist + kMfmaSeg + 15 ? (int)last : cur.hist + kMfmaSeg + 15;
                    nxt.emin = nb < cur.hist ? (int)(cur.hist - nb) : 0;
                }
            }
            seg_base = (long long)(steady_lo + seg) * kMfmaSeg;
            prefetch(nxt);
        }
        /* the next item's scalar loads end here: one pending into the contraction loop puts an lgkmcnt(0) at
         * its head (scalar loads return out of order), which every iteration would pay */
        asm volatile("" :: "s"(nxt.tail), "s"(nxt.t0));

#pragma unroll
        for (int t = 0; t < kMfmaTiles; ++t) acc[t] = (d4){cur.tail, cur.tail, cur.tail, cur.tail};

        /* Tile row i' stands for output offset pi(i') = 4 (i' & 3) + (i' >> 2) inside its 16, so that
         * the four results a lane holds (rows q, q+4, q+8, q+12 of column j) are the consecutive
         * outputs 16 j + 4 q + {0,1,2,3}: each lane stores 32 contiguous bytes, a wavefront a
         * contiguous 2 KB tile.  A: row i' = j, kk = q: bp[16 + pi(j) - q + delta];  B: column j,
         * kk = q, tile t: xs element hist + 256 t + 16 j + q - delta
         *                 = 16 (hist/16 + 16 t + j - o) + (12 + q - 4 u), no carry */
        const int pij = 4 * (j & 3) + (j >> 2);
        lds_cdouble_p pa[4], pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pa[u] = lds_opaque(bp + (4 + pij - q) + 4 * u);
            pb[u] = lds_opaque(xs + kPitch * (cur.hist / 16 + 64 * wave + j) + q + 4 * (3 - u));
        }
        /* one register set, software-pipelined: as soon as the four MFMAs of sub-step u have
         * issued (operands are read at issue), its registers take the operands of the next outer
         * step, which then have 12 MFMAs (768 cycles) to arrive */
        MfmaOperands r;
#pragma unroll
        for (int u = 0; u < 4; ++u) mfma_fetch<kMfmaTiles>(r, pa, pb, u, 0);
        for (int o = 1; o < cur.steps; ++o) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mfma_apply<kMfmaTiles>(acc, r, u);
                mfma_fetch<kMfmaTiles>(r, pa, pb, u, o);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mfma_apply<kMfmaTiles>(acc, r, u);

        /* Outputs t0 .. t0 + K - 2 of a head item sit on s[m - t0], not on s[K-1] (output 0 on x[0] when
         * stable_init consumed it: the contraction added exact zeros to a zero tail there).  Added here, in
         * place and in a wave-uniform branch, instead of at the initialisation: a conditional definition of
         * the accumulator tuples spills, and anything in front of the contraction lengthens the stretch in
         * which the workgroup issues no MFMA -- for every item. */
        if (cur.head) {
#pragma unroll
            for (int t = 0; t < kMfmaTiles; ++t) {
                const int mi0 = 256 * (kMfmaTiles * wave + t) + 1 - cur.t0;
                if (mi0 <= cur.K - 1) {                         /* wave-uniform: later tiles are steady */
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int mi = mi0 + 16 * j + 4 * q + r4;
                        const double c = sa[min(mi, cur.K - 1)];
                        acc[t][r4] += mi <= cur.K - 1 ? c : 0.0;
                    }
                }
            }
        }

        if constexpr (kTiled)       /* wavefronts 0, 1: the item's first tile of 2048 outputs; 2, 3: its second */
            pending = cur.out + (size_t)((cur_base >> 11) + (wave >> 1)) * seg_out + 256 * kMfmaTiles * (wave & 1) + 16 * j + 4 * q;
        else
            pending = cur.out + 256 * kMfmaTiles * wave + 16 * j + 4 * q;
        pending_skip = cur_skip;
        cur = nxt;
    }
    mfma_store<kAligned16>(acc, pending, pending_skip);
}

/* workgroups of fir_mfma_kernel the device holds at once for a given LDS size */
static int mfma_resident_groups(size_t lds_bytes)
{
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fir_mfma_kernel<true, false>, kMfmaWaves * kWave, lds_bytes) !=
            hipSuccess || per_cu < 1)
        per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    return per_cu * cus;
}

/* ------------------------------------------------------------------------ */
/* Short blocks on wide banks: one lane = one channel, the reference's loop itself.             */
/*                                                                                                */
/* The kernels above split TIME; a real-time producer hands over short blocks (tens to a few     */
/* thousand samples) of very many channels, where a segment per workgroup leaves most lanes      */
/* idle.  Here a wavefront owns 64 channels and walks the block in tiles of 32 samples exactly  */
/* as RTFilter_update does (ref: src/RTFilter.c:98-108, 235-297): delay line and taps of the     */
/* lane's channel in registers for the whole block (KP <= 32 slots, slots past the channel's own */
/* K never touched, so s[K-1] keeps the reference's "read, never written" role), stable_init /   */
/* partial_init inline on the samples they apply to.  Bit-exact by construction.  Tile staging   */
/* as in iir_seq_kernel: 64 rows x 32 doubles through padded LDS, unconditional 16-byte loads    */
/* with the next tile prefetched on full tiles.                                                  */

/* kUniform: every channel of the group has the same design -- taps and K are wave-uniform scalars (SGPRs,
 * scalar branches for the K bounds), which leaves the registers to the delay line: required for KP = 32. */
template <int KP, bool kUniform>
__global__ void __launch_bounds__(kWave)
fir_seq_kernel(GroupView g, BlockIO io)
{
    constexpr int TT = KP > 16 ? 16 : 32;               /* 32 slots: a shorter tile keeps the prefetch in registers */
    constexpr int kRowBytes = TT * 8 + 16;
    constexpr int kPiecesPerRow = TT / 2, kLoads = TT / 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char *smem = smem_raw;

    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * kWave;
    const int gi = g0 + lane;
    const bool live = gi < g.count;
    const int gc = live ? gi : g.count - 1;
    const int chan = group_channel(g, gc);
    const int dsn = kUniform ? g.single_design : g.design[gc];
    const int K = kUniform ? as_const(g.nb)[dsn] : g.nb[dsn];
    const bool partial = (kUniform ? as_const(g.kind)[dsn] : g.kind[dsn]) == RTFB_KIND_FIR_PARTIAL;

    double b[KP], s[KP];
    {
        const double *st = g.state + (size_t)gc * g.state_ld;
#pragma unroll
        for (int i = 0; i < KP; ++i) s[i] = i < K ? st[i] : 0.0;
        if (kUniform) {
            cdouble_p cf = as_const(g.coefs) + as_const(g.coef_off)[dsn];
#pragma unroll
            for (int i = 0; i < KP; ++i) b[i] = i < K ? cf[i] : 0.0;
        } else {
            const double *cf = g.coefs + g.coef_off[dsn];
#pragma unroll
            for (int i = 0; i < KP; ++i) b[i] = i < K ? cf[i] : 0.0;
        }
    }
    double y = g.fval[gc];
    int ini = g.init[gc];
    const double kd = (double)(size_t)K;

    auto one_sample = [&](double x) {
        if (ini == 0) {                                      /* RTFIRFilter_update, ref :235-247 */
            double v = x * b[0];
            y = v + s[0];
        } else if (!partial) {                               /* RTFIRFilter_stable_init, ref :250-265 */
            double run = 0.0;
#pragma unroll
            for (int i = KP - 1; i >= 1; --i) {
                s[i] = (i == K - 1) ? 0.0 : s[i];            /* a select: `if (i == K-1) s[i] = 0` folds into a
                                                                dynamically indexed store and sends s[] to scratch */
                if (i <= K - 1) {
                    run = run + b[i];
                    s[i - 1] = run * x;
                }
            }
            s[0] = (K == 1) ? 0.0 : s[0];
            y = x;
            ini = 0;
            return;                                          /* no delay-line update on this sample */
        } else if (ini == FILTER_RESET) {                    /* RTFIRFilter_partial_init, ref :279-282 */
#pragma unroll
            for (int i = 0; i < KP; ++i) s[i] = (i == K - 1) ? 0.0 : s[i];
            y = x;
            ini = K - 1;
        } else {                                             /* ... :283-286 */
            ini -= 1;
            double v = x * b[0];
            v = v + s[0];
            v = kd * v;
            y = v / (double)(size_t)(K - ini);
        }
#pragma unroll
        for (int i = 0; i < KP - 1; ++i) {                   /* :242-245 / :287-291 */
            if (i < K - 1) {
                const double p = b[i + 1] * x;
                s[i] = s[i + 1] + p;
            }
        }
    };

    const size_t n = io.n;
    const bool wide = g0 + kWave <= g.count && io.aligned16;
    const int st_r = lane / kPiecesPerRow, st_q = lane % kPiecesPerRow;
    constexpr int kRowsPerPass = kWave / kPiecesPerRow;
    d2 pre[kLoads];
    auto fetch = [&](size_t t0) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int cr = __shfl(chan, i * kRowsPerPass + st_r, kWave);
            pre[i] = *reinterpret_cast<const d2 *>(io.in + (size_t)cr * io.ld_in + t0 + 2 * st_q);
        }
    };
    if (wide && n >= (size_t)TT) fetch(0);

    for (size_t t0 = 0; t0 < n; t0 += TT) {
        const int cnt = (n - t0 < (size_t)TT) ? (int)(n - t0) : TT;
        char *row = smem + lane * kRowBytes;
        if (wide && cnt == TT) {
#pragma unroll
            for (int i = 0; i < kLoads; ++i)
                *reinterpret_cast<d2 *>(smem + (i * kRowsPerPass + st_r) * kRowBytes + st_q * 16) = pre[i];
            wave_sync();
            if (t0 + 2 * TT <= n) fetch(t0 + TT);
#pragma unroll 1
            for (int k = 0; k < TT; k += 2) {
                d2 xv = *reinterpret_cast<d2 *>(row + k * 8);
                one_sample(xv.x);
                xv.x = y;
                one_sample(xv.y);
                xv.y = y;
                *reinterpret_cast<d2 *>(row + k * 8) = xv;
            }
            wave_sync();
#pragma unroll
            for (int i = 0; i < kLoads; ++i) {
                const int cr = __shfl(chan, i * kRowsPerPass + st_r, kWave);
                *reinterpret_cast<d2 *>(io.out + (size_t)cr * io.ld_out + t0 + 2 * st_q) =
                    *reinterpret_cast<const d2 *>(smem + (i * kRowsPerPass + st_r) * kRowBytes + st_q * 16);
            }
            wave_sync();
            continue;
        }
        /* ragged tile: element-wise, guarded */
        for (int i = 0; i < kLoads * 2; ++i) {
            const int p = i * kWave + lane;                  /* element p: row p / TT, column p % TT */
            const int r = p / TT, c = p % TT;
            const int cr = __shfl(chan, r, kWave);
            double v = 0.0;
            if (g0 + r < g.count && c < cnt) v = io.in[(size_t)cr * io.ld_in + t0 + c];
            *reinterpret_cast<double *>(smem + r * kRowBytes + c * 8) = v;
        }
        wave_sync();
        for (int k = 0; k < cnt; ++k) {
            double *px = reinterpret_cast<double *>(row + k * 8);
            one_sample(*px);
            *px = y;
        }
        wave_sync();
        for (int i = 0; i < kLoads * 2; ++i) {
            const int p = i * kWave + lane;
            const int r = p / TT, c = p % TT;
            const int cr = __shfl(chan, r, kWave);
            if (g0 + r < g.count && c < cnt)
                io.out[(size_t)cr * io.ld_out + t0 + c] = *reinterpret_cast<const double *>(smem + r * kRowBytes + c * 8);
        }
        wave_sync();
    }

    if (live) {
        double *st = g.state + (size_t)gi * g.state_ld;
#pragma unroll
        for (int i = 0; i < KP; ++i)
            if (i < K) st[i] = s[i];
        g.fval[gi] = y;
        g.init[gi] = ini;
    }
}

/* Lane-per-channel kernel: when every filter of the group fits its registers, the bank is wide enough to
 * fill the GPU with lanes and the block is too short to give the time-split kernels whole segments. */
constexpr int kFirSeqMaxTaps = 32;
constexpr int kFirSeqMinChannels = 16384;
constexpr size_t kFirSeqMaxBlock = 4096;

static hipError_t launch_fir_seq(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    const int blocks = (g.count + kWave - 1) / kWave;
    const size_t lds = (size_t)kWave * (32 * 8 + 16);
    const bool uni = g.single_design >= 0;
    if (g.kmax <= 8) {
        if (uni) hipLaunchKernelGGL((fir_seq_kernel<8, true>), dim3(blocks), dim3(kWave), lds, s, g, io);
        else hipLaunchKernelGGL((fir_seq_kernel<8, false>), dim3(blocks), dim3(kWave), lds, s, g, io);
    } else if (g.kmax <= 16) {
        if (uni) hipLaunchKernelGGL((fir_seq_kernel<16, true>), dim3(blocks), dim3(kWave), lds, s, g, io);
        else hipLaunchKernelGGL((fir_seq_kernel<16, false>), dim3(blocks), dim3(kWave), lds, s, g, io);
    } else {
        hipLaunchKernelGGL((fir_seq_kernel<32, true>), dim3(blocks), dim3(kWave), lds, s, g, io);
    }
    return hipGetLastError();
}

/* The end-state kernels write the block's end state into g.state_out (their workgroups read g.state while others
 * already write); this puts it back into g.state, so that every call leaves the carried delay line in ONE fixed
 * buffer and a captured call can be replayed any number of times (round 2 swapped the two buffers on the host,
 * which a graph replay cannot see).  16 B per channel and tap: 0.1 ms of a 65 ms step at 65 536 x 512 taps. */
__global__ void __launch_bounds__(256) fir_commit_kernel(GroupView g)
{
    const size_t total = (size_t)g.count * g.state_ld;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256)
        g.state[i] = g.state_out[i];
}

static hipError_t launch_fir_commit(const GroupView &g, hipStream_t s)
{
    const size_t total = (size_t)g.count * g.state_ld;
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fir_commit_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    return hipGetLastError();
}

bool fir_uses_state_swap(const GroupView &g, const BlockIO &io)
{
    /* mixed designs keep their taps in vector registers: that fits up to 16 slots */
    const int max_taps = g.single_design >= 0 ? kFirSeqMaxTaps : 16;
    return !(g.kmax <= max_taps && g.count >= kFirSeqMinChannels && io.n <= kFirSeqMaxBlock);
}

static hipError_t launch_fir_nf_fixup(const GroupView &g, const BlockIO &io, size_t seg_in, size_t seg_out,
                                      long long ramp_first, hipStream_t s);
static hipError_t launch_fir_tile(const GroupView &g, const BlockIO &io, size_t seg_in, size_t seg_out, bool exact,
                                  double pace_ns, bool tiled, hipStream_t s);
/* the exact head kernel on the block's first 2048 samples, for the channels inside a partial_init ramp only */
static hipError_t launch_fir_ramp_heads(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    GroupView h = g;
    h.fir_mfma = 1;                                             /* "others leave at once" */
    const int total = S + g.kmax - 1;
    const size_t lds_bytes = (size_t)(total + (total >> 5) + 1) * sizeof(double);
    hipLaunchKernelGGL(fir_main_kernel<true>, dim3(1, g.count), dim3(kWave), lds_bytes, s, h, io, 0);
    return hipGetLastError();
}

hipError_t launch_fir(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    if (!fir_uses_state_swap(g, io)) return launch_fir_seq(g, io, s);      /* updates g.state in place */
    const int nseg = (int)((io.n + S - 1) / S);
    const int pro_blocks = (g.count + 255) / 256;
    hipLaunchKernelGGL(fir_prologue_kernel, dim3(pro_blocks), dim3(256), 0, s, g, io);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    /* Short filters on whole aligned tiles of a bank wide enough to fill the GPU with one workgroup per channel: the
     * streaming kernel (rows are tiles a pitch apart) -- in the bit-exact mode up to 128 taps (65 536 x 64K, 32 taps:
     * 3.68 -> 4.33 TB/s, 16 taps 4.46 -> 4.79, 7 taps 5.03 -> 5.24; 16 384 x 64K, 64 taps 2.46 -> 2.78, 128 taps 1.35
     * -> 1.54), otherwise up to 8 taps (5.17 -> 5.24, and the reference's bits); from 9 taps on the matrix-core
     * kernel leads on rows (4.99 against 4.69-4.80 at 32 taps, 2.97 against 2.36 at 128).  profiles/r3_fir_tile.txt */
    if (g.kmax >= 2 && g.kmax <= 128 && (!g.fir_mfma || g.kmax <= 8) && g.count >= 512 &&
        io.aligned16 && io.n % kMfmaSeg == 0 && io.n <= ((size_t)1 << 30)) {
        if (g.has_partial && (e = launch_fir_ramp_heads(g, io, s)) != hipSuccess) return e;
        if ((e = launch_fir_tile(g, io, 2048, 2048, !g.fir_mfma, 0.0, false, s)) != hipSuccess) return e;
        const int passes = (g.kmax - 1 + kEpiTapPass - 1) / kEpiTapPass;
        const size_t lds_taps = (size_t)(2 * g.kmax + 4) * sizeof(double);
        hipLaunchKernelGGL(fir_epilogue_taps_kernel, dim3(passes, g.count), dim3(kWave), lds_taps, s, g, io);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        return launch_fir_commit(g, s);
    }

    const int total = S + g.kmax - 1;
    const size_t lds_bytes = (size_t)(total + (total >> 5) + 1) * sizeof(double);
    /* segments whose outputs can touch the carried state or the ramp: m < K + 1 */
    int nhead = (g.kmax + 1 + S - 1) / S;
    if (nhead > nseg) nhead = nseg;
    if (g.fir_mfma) {
        /* exact head only for channels inside a partial_init ramp (the others leave at once) */
        if (nhead > 0 && g.has_partial) {
            hipLaunchKernelGGL(fir_main_kernel<true>, dim3(nhead, g.count), dim3(kWave), lds_bytes, s, g,
                               io, 0);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        const int hist = 16 * (mfma_steps16(g.kmax) - 1);
        int xtotal = hist + kMfmaSeg + 16;
        if (xtotal < kMfmaPreElems) xtotal = kMfmaPreElems;
        size_t mfma_lds = (size_t)(hist + 32 + xtotal + 2 * (xtotal >> 4) + 2) * sizeof(double);
        {   /* a folded head keeps K state words behind its inputs (mostly inside the slack of the line above) */
            const int ptotal = hist + kMfmaSeg + 16;
            const size_t with_state = (size_t)(hist + 32 + ptotal + 2 * (ptotal >> 4) + 2 + g.kmax) * sizeof(double);
            if (with_state > mfma_lds) mfma_lds = with_state;
        }
        const int nwg = (int)((io.n + kMfmaSeg - 1) / kMfmaSeg);       /* segments per channel */
        /* steady segments: first output has more than kmax samples behind it (and lies past any
         * ramp head: nhead*S <= steady_lo*kMfmaSeg), the whole segment is inside the block */
        /* The persistent kernel takes over at output head_out of every channel: the first multiple of 256
         * (whole accumulator tiles) with more than kmax samples behind it, so everything from there on is in
         * steady state; inside a partial_init ramp it takes over at ramp_first.  The item that holds head_out
         * belongs to both kernels: the edge kernel writes its outputs before head_out (a tile or two of
         * the 16), the persistent one the rest -- round 1 sent the whole first item of every channel, a
         * sixteenth of a 64K block, through the edge kernel at 3.4 TB/s. */
        long long head_out = 256LL * ((g.kmax + 1 + 255) / 256);
        int steady_lo = (int)(head_out / kMfmaSeg);                 /* the item that holds head_out */
        int steady_hi = (int)(io.n / kMfmaSeg);
        /* Groups without partial_init channels: the persistent kernel takes the head as well -- its first item
         * starts from the carried state, with zeros where there is no regular sample before the block. */
        /* (aligned rows only: the other instantiation is at its register limit) */
        const bool fold = !g.has_partial && steady_hi > 0 && io.aligned16;
        if (fold) {
            head_out = 0;
            steady_lo = 0;
        } else if (steady_lo >= steady_hi) {                        /* short block: no item is whole */
            steady_lo = steady_hi = steady_lo < nwg ? steady_lo : nwg;
            if (steady_hi > (int)(io.n / kMfmaSeg)) steady_hi = steady_lo = (int)(io.n / kMfmaSeg);
            head_out = 0;                                           /* nothing is shared */
        }
        const bool shared_head = head_out > (long long)steady_lo * kMfmaSeg;
        const int edge_lo = steady_lo + (shared_head ? 1 : 0);      /* leading items the edge kernel visits */
        const int nedge = edge_lo + (nwg - steady_hi);
        const long long ramp_first = g.has_partial ? (long long)nhead * S : 0;
        if (nedge > 0) {
            /* only short heads (no whole edge item in front of them, no ragged end behind): one tile per wavefront */
            if (shared_head && nedge == 1 && head_out <= 256 * kMfmaWaves) {
                /* LDS for what a head stages (taps, history, head_out + 16 inputs), not for a whole item: two to
                 * three times as many workgroups per CU overlap their staging with the others' contraction */
                const int htotal = hist + (int)head_out + 16;
                const size_t head_lds = (size_t)(hist + 32 + htotal + 2 * (htotal >> 4) + 2) * sizeof(double);
                hipLaunchKernelGGL(fir_mfma_edge_kernel<1>, dim3(nedge, g.count), dim3(kMfmaWaves * kWave), head_lds,
                                   s, g, io, (long long)nhead * S, edge_lo, steady_hi, head_out);
            }
            else
                hipLaunchKernelGGL(fir_mfma_edge_kernel<kMfmaTiles>, dim3(nedge, g.count), dim3(kMfmaWaves * kWave),
                                   mfma_lds, s, g, io, (long long)nhead * S, edge_lo, steady_hi, head_out);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (steady_hi > steady_lo) {
            const long long items = (long long)(steady_hi - steady_lo) * g.count;
            long long grid = mfma_resident_groups(mfma_lds);
            if (grid > items) grid = items;
            if (io.aligned16)
                hipLaunchKernelGGL((fir_mfma_kernel<true, false>), dim3((unsigned)grid), dim3(kMfmaWaves * kWave),
                                   mfma_lds, s, g, io, steady_lo, steady_hi, head_out, ramp_first, fold ? 1 : 0,
                                   (size_t)0, (size_t)0);
            else
                hipLaunchKernelGGL((fir_mfma_kernel<false, false>), dim3((unsigned)grid), dim3(kMfmaWaves * kWave),
                                   mfma_lds, s, g, io, steady_lo, steady_hi, head_out, ramp_first, fold ? 1 : 0,
                                   (size_t)0, (size_t)0);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = launch_fir_nf_fixup(g, io, 2048, 2048, (long long)nhead * S, s)) != hipSuccess) return e;
    } else {
        if (nhead > 0) {
            hipLaunchKernelGGL(fir_main_kernel<true>, dim3(nhead, g.count), dim3(kWave), lds_bytes, s, g,
                               io, 0);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (nseg > nhead) {
            hipLaunchKernelGGL(fir_direct_kernel, dim3(nseg - nhead, g.count), dim3(kDirThreads),
                               (size_t)(S + g.kmax + 2) * sizeof(double), s, g, io, nhead);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
    }
    if (io.n >= (size_t)g.kmax + 1 && g.kmax >= 2) {
        /* every channel has its K-1 regular samples inside the block (one may have gone to stable_init) */
        const int passes = (g.kmax - 1 + kEpiTapPass - 1) / kEpiTapPass;
        const size_t lds_taps = (size_t)(2 * g.kmax + 4) * sizeof(double);
        hipLaunchKernelGGL(fir_epilogue_taps_kernel, dim3(passes, g.count), dim3(kWave), lds_taps, s, g, io);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        return launch_fir_commit(g, s);
    }
    const size_t epi_lds = (size_t)(g.kmax + (g.kmax + kEpiR) + (g.kmax + kEpiR) / kEpiR + 2) * sizeof(double);
    hipLaunchKernelGGL(fir_epilogue_kernel, dim3(g.count), dim3(kEpiThreads), epi_lds, s, g, io);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return launch_fir_commit(g, s);
}

/* Items of the matrix-core kernels that staged an Inf / NaN (nf_mark), done again the reference's way: every output
 * from the carried state or s[K-1], taps added oldest first, every product and sum rounded on its own -- so a
 * non-finite sample shows in exactly the outputs it reaches through the taps, as in RTFIRFilter_update
 * (ref: src/RTFilter.c:235-247), and those items carry the reference's bits.  One launch per call on such a group;
 * without a mark every workgroup reads one word and leaves.  Addressing covers rows (seg_* = 2048, ld = the pitch)
 * and the tiled block layout (ld = 2048, seg_* = doubles between a channel's consecutive tiles) alike.
 * `ramp_first`: as for the edge kernel -- inside a partial_init ramp the matrix-core kernels own the outputs from
 * there on. */
__global__ void __launch_bounds__(256)
fir_nf_fixup_kernel(GroupView g, BlockIO io, size_t seg_in, size_t seg_out, long long ramp_first)
{
    if (*as_const(nf_any(g)) == 0) return;
    const int tid = threadIdx.x;
    const long long n = (long long)io.n;
    const int nitems = (int)((n + kMfmaSeg - 1) / kMfmaSeg);
    for (int gi = blockIdx.x; gi < g.count; gi += gridDim.x) {
        const unsigned long long dirty = nf_dirty(g)[gi];
        if (!dirty) continue;
        const int chan = g.chan ? g.chan[gi] : g.chan0 + gi;
        const int dsn = g.design[gi];
        const int K = g.nb[dsn];
        const double *b = g.coefs + g.coef_off[dsn];
        const double *st = g.state + (size_t)gi * g.state_ld;
        const double tail = st[K - 1];
        const int t0 = g.aux[2 * gi];
        const bool ramp = g.kind[dsn] == RTFB_KIND_FIR_PARTIAL && g.aux[2 * gi + 1] < K;
        const long long first = ramp ? ramp_first : (long long)t0;
        const double *x = io.in + (size_t)chan * io.ld_in;
        double *y = io.out + (size_t)chan * io.ld_out;
        auto at = [](long long m, size_t seg) { return (size_t)(m >> 11) * seg + (size_t)(m & 2047); };
        for (int item = 0; item < nitems; ++item) {
            if (!((dirty >> (item & 63)) & 1ull)) continue;
            const long long m_end = min((long long)(item + 1) * kMfmaSeg, n);
            for (long long m = (long long)item * kMfmaSeg + tid; m < m_end; m += blockDim.x) {
                if (m < first) {
                    /* sample 0 went to stable_init: its output is the sample (the prologue wrote it, the folded
                     * head of the persistent kernel wrote it again, with the zero taps' products added) */
                    if (!ramp) y[at(m, seg_out)] = x[at(m, seg_in)];
                    continue;
                }
                const long long mp = m - t0;
                double acc = mp <= K - 2 ? st[mp] : tail;
                for (int k = (int)min(mp, (long long)K - 1); k >= 1; --k) acc = acc + b[k] * x[at(m - k, seg_in)];
                y[at(m, seg_out)] = x[at(m, seg_in)] * b[0] + acc;
            }
        }
    }
}

static hipError_t launch_fir_nf_fixup(const GroupView &g, const BlockIO &io, size_t seg_in, size_t seg_out,
                                      long long ramp_first, hipStream_t s)
{
    const int grid = g.count < 2048 ? g.count : 2048;
    hipLaunchKernelGGL(fir_nf_fixup_kernel, dim3(grid), dim3(256), 0, s, g, io, seg_in, seg_out, ramp_first);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------ */
/* Short FIR (2 <= K <= 128 taps) on the tiled block layout and on rows of wide banks: */
/* the streaming kernel.  One workgroup of 256 threads per channel walks the      */
/* channel's tiles in time order -- the access pattern of the pipelined IIR       */
/* kernel and of its copy twin, which is what the tiled layout and the schedule   */
/* are made for: two tiles of loads in flight per workgroup, four     */
/* workgroups per CU, every launch one resident round of channels.  A thread      */
/* computes 8 consecutive outputs of a tile from LDS (tile + the 32 / 64 / 128     */
/* samples in front of it, 10 doubles per 8 samples: the 64-byte windows are      */
/* conflict-free), taps oldest first, two 8-sample blocks of the window in        */
/* registers at a time; the taps are wave-uniform scalars and only the K the      */
/* design has are ever multiplied: no padding products, so an Inf / NaN reaches   */
/* exactly the outputs it reaches in RTFIRFilter_update (ref: src/RTFilter.c:     */
/* 235-247).  kExact: every product and sum rounded on its own, the reference's   */
/* bits; else fused multiply-adds (the matrix-core path's accuracy class).        */
/* The outputs leave through LDS as well: stored straight from the registers,     */
/* 64 contiguous bytes per lane, every store instruction touches 64 lines a        */
/* quarter each and a 65 536-channel wave takes 16.1 ms; stored as 16 contiguous   */
/* bytes per lane (one more barrier per tile) 12.x ms.                             */
namespace {
constexpr int kTfThreads = 256;
constexpr int kTfMaxTaps = 128;                                /* the widest instantiation */
constexpr int tf_halo(int kp) { return kp < 32 ? 32 : kp; }    /* samples in front of a tile (>= K-1) */
constexpr int tf_lds_doubles(int kp) { return (2048 + tf_halo(kp)) / 8 * 10; }  /* the tile and the samples in front of it */
constexpr int kTfLdsOut = 2048;                                /* behind them: the tile's outputs on their way out */
}  // namespace

template <int KP, bool kExact>
__global__ void __launch_bounds__(kTfThreads, 4)
fir_tile_kernel(GroupView g, BlockIO io, int gi0, size_t seg_in, size_t seg_out, unsigned int pace_fx)
{
    static_assert(KP == 8 || KP == 16 || KP == 32 || KP == 64 || KP == 128, "window blocks of 8 samples");
    constexpr int kTfHalo = tf_halo(KP);
    constexpr int kTfLdsDoubles = tf_lds_doubles(KP);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double *xs = reinterpret_cast<double *>(smem_raw);
    const int tid = threadIdx.x;
    const int gi = gi0 + (int)blockIdx.x;
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    const int K = as_const(g.nb)[dsn];
    cdouble_p cb = as_const(g.coefs + as_const(g.coef_off)[dsn]);
    const double *st = g.state + (size_t)gi * g.state_ld;
    const double tail = as_const(st)[K - 1];
    const int t0 = as_const(g.aux)[2 * gi];                     /* 1: the prologue gave sample 0 to stable_init */
    /* a channel inside a partial_init ramp (ref: src/RTFilter.c:273-297): the exact head kernel has written its
     * first tile (the ramp is over after K - 1 <= 127 samples); from the second tile on it is in steady state */
    const bool ramp = as_const(g.kind)[dsn] == RTFB_KIND_FIR_PARTIAL && as_const(g.aux)[2 * gi + 1] < K;
    const int nseg = (int)(io.n / 2048);
    const double *src = io.in + (size_t)chan * io.ld_in;        /* tiled layout: ld = 2048, seg_* = nch x 2048; rows: */
    double *dst = io.out + (size_t)chan * io.ld_out;            /* ld = the pitch, seg_* = 2048                      */
    /* the taps: up to 32 stay in scalar registers for the whole launch; wider designs read them (scalar loads, the
     * constant cache holds them) where they are used -- tap by tap behind the branch on K: requested a window block
     * ahead and without the branch the compiler hoists all of them out of the tile loop and keeps them in lanes of
     * a vector register (two v_readlane per tap): 2.71 against 2.78 TB/s at 64 taps */
    constexpr bool kTapsResident = KP <= 32;
    double bk[kTapsResident ? KP : 1];
#pragma unroll
    for (int k = 0; k < (kTapsResident ? KP : 1); ++k) bk[k] = cb[k < K ? k : K - 1];
    auto tap = [&](int kk) -> double {                          /* kk < K */
        if constexpr (kTapsResident) return bk[kk];
        else return cb[kk];
    };

    /* Every thread issues the same five loads per tile and the same four stores, whatever the tile: past the end
     * of the block the loads go to the channel's first tile, and all threads fetch a piece of the 32 samples in
     * front of the tile (16 of them keep it).  With nothing conditional the compiler counts vmcnt exactly -- two
     * tiles of loads really stay in flight; with guarded loads it drained the queue at every tile (15.8 ms per
     * 65 536-channel wave instead of 12.x). */
    d2 buf0[4], buf1[4], h0, h1;
    auto load = [&](int s, d2(&buf)[4], d2 &h) {
        const int sc = s < nseg ? s : 0;
        const double *p = src + (size_t)sc * seg_in + 2 * tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[i] = stream_load(reinterpret_cast<const d2 *>(p + 512 * i));
        h = *reinterpret_cast<const d2 *>(src + (size_t)(sc > 0 ? sc - 1 : 0) * seg_in + (2048 - kTfHalo) +
                                          2 * (tid & (kTfHalo / 2 - 1)));
    };
    /* sample i of the tile (i >= -32) sits at LDS double (i + 32) + 2 ((i + 32) >> 3) */
    auto stage = [&](const d2(&buf)[4], const d2 &h, bool first) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 2 * (i * kTfThreads + tid) + kTfHalo;
            *reinterpret_cast<d2 *>(xs + e + 2 * (e >> 3)) = buf[i];
        }
        if (tid < kTfHalo / 2) {
            const int e = 2 * tid;
            d2 v = h;
            if (first) v = (d2){0.0, 0.0};
            *reinterpret_cast<d2 *>(xs + e + 2 * (e >> 3)) = v;
        }
    };
    const int o = 8 * tid;                                      /* this thread's outputs of a tile: o .. o + 7 */
    auto block = [&](int q, double(&w)[8]) {                    /* samples o + 8 q .. o + 8 q + 7 */
        const d2 *p = reinterpret_cast<const d2 *>(xs + 10 * (tid + q + kTfHalo / 8));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const d2 v = p[m];
            w[2 * m] = v.x;
            w[2 * m + 1] = v.y;
        }
    };
    /* kGuard: the block's first tile -- outputs that still sit on the carried state, samples that do not exist
     * (before the first regular sample t0) contribute NOTHING (not even a zero product) */
    auto compute = [&](int s, auto guard_tag) {
        constexpr bool kGuard = decltype(guard_tag)::value;
        constexpr int NB = KP / 8;
        double acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = tail;
            if constexpr (kGuard) {
                const int mp = o + j - t0;
                if (mp >= 0 && mp <= K - 2) acc[j] = st[mp];
            }
        }
        double lo[8], hi[8];
        block(-NB, lo);
#pragma unroll
        for (int q = -NB; q < 0; ++q) {
            block(q + 1, hi);
#pragma unroll
            for (int kk = -8 * q; kk >= -8 * q - 7; --kk) {
                if (kk > KP - 1) continue;
                if (kk < K) {                                   /* wave-uniform */
                    const double t = tap(kk);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int idx = j - kk - 8 * q;         /* 0 .. 14 within (lo, hi) */
                        const double x = idx < 8 ? lo[idx < 8 ? idx : 0] : hi[idx >= 8 ? idx - 8 : 0];
                        double r;
                        if constexpr (kExact) {
                            const double pr = t * x;
                            r = acc[j] + pr;
                        } else
                            r = __builtin_fma(t, x, acc[j]);
                        if constexpr (kGuard) acc[j] = (o + j - kk >= t0) ? r : acc[j];
                        else acc[j] = r;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] = hi[j];
        }
        double y[8];
        const double b0 = tap(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (kExact) {
                const double pr = lo[j] * b0;
                y[j] = pr + acc[j];
            } else
                y[j] = __builtin_fma(lo[j], b0, acc[j]);
        }
        if constexpr (kGuard) {                                 /* sample 0 went to stable_init: its output is the sample */
            if (tid == 0 && t0 > 0) y[0] = lo[0];
        }
        if (kGuard && ramp) return;                             /* (workgroup-uniform) */
        double *ys = xs + kTfLdsDoubles;
#pragma unroll
        for (int m = 0; m < 4; ++m) *reinterpret_cast<d2 *>(ys + o + 2 * m) = (d2){y[2 * m], y[2 * m + 1]};
        lds_barrier();
        double *out = dst + (size_t)s * seg_out + 2 * tid;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            stream_store(reinterpret_cast<d2 *>(out + 512 * i), *reinterpret_cast<const d2 *>(ys + 512 * i + 2 * tid));
    };

    load(0, buf0, h0);
    load(1, buf1, h1);
    unsigned long long due_fx = 0;
    if (pace_fx) due_fx = (__builtin_amdgcn_s_memrealtime() << 10) + pace_fx;
    auto pace = [&]() {                                         /* see pipe_kernel.hip: the prefetch keeps to a schedule */
        if (pace_fx) {
            while ((__builtin_amdgcn_s_memrealtime() << 10) < due_fx) __builtin_amdgcn_s_sleep(1);
            due_fx += pace_fx;
        }
    };
    /* one tile: park it, request the tile two ahead into the registers just freed, compute.  The block's first tile
     * (guarded: it loads state words under conditions) stays outside the loop, whose body then holds nothing
     * conditional. */
    auto tile = [&](int s, d2(&buf)[4], d2 &h, auto guard_tag) {
        lds_barrier();                                          /* the tile before has been read */
        stage(buf, h, decltype(guard_tag)::value);
        pace();
        load(s + 2, buf, h);
        lds_barrier();
        compute(s, guard_tag);
    };
    tile(0, buf0, h0, std::true_type{});
    tile(1, buf1, h1, std::false_type{});                       /* nseg is even: n is a multiple of 4096 */
    for (int s = 2; s < nseg; s += 2) {
        tile(s, buf0, h0, std::false_type{});
        tile(s + 1, buf1, h1, std::false_type{});
    }
}

/* rounds of at most `round` channels, each launch on the schedule when it fills the GPU */
/* Arithmetic: the reference's (every product and sum rounded on its own) when asked for (RTFB_MODE_SEQUENTIAL) and,
 * unasked, up to 16 taps, where it costs nothing (65 536 x 64K, tiled: 5.63 TB/s either way; 24 taps 5.28 against
 * 5.59 fused, 32 taps 4.5 against 5.54: profiles/r3_fir_tile.txt).  Schedule: from 9 taps on (below, the workgroups
 * stay in step by themselves and run 2-3 % faster without: 5.74-5.81 against 5.65 TB/s). */
static hipError_t launch_fir_tile(const GroupView &g, const BlockIO &io, size_t seg_in, size_t seg_out, bool exact,
                                  double pace_ns, bool tiled, hipStream_t s)
{
    if (g.kmax <= 16) exact = true;
    if (pace_ns < 0.0 && (g.kmax <= 8 || !tiled)) pace_ns = 0.0;
    /* one resident round (4 workgroups per CU) per launch, so that the workgroups of a launch stay in step; short
     * blocks (under 12 tiles: mostly pipeline fill, never paced) go as one launch */
    const int round = io.n / 2048 >= 12 ? 1024 : (g.count > 1024 ? g.count : 1024);
    const int kp = g.kmax <= 8 ? 8 : g.kmax <= 16 ? 16 : g.kmax <= 32 ? 32 : g.kmax <= 64 ? 64 : 128;
    const size_t lds = (size_t)(tf_lds_doubles(kp) + kTfLdsOut) * sizeof(double);
    for (int off = 0; off < g.count; off += round) {
        const int cnt = g.count - off < round ? g.count - off : round;
        unsigned int pace_fx = 0;
        if (pace_ns != 0.0 && 4 * cnt >= 3 * round && io.n / 2048 >= 12) {
            const double ns = pace_ns > 0.0 ? pace_ns : (double)cnt * 32768.0 / 5950.0;
            pace_fx = (unsigned int)(ns / 10.0 * 1024.0);
        }
#define DSP_TF(KP)                                                                                                  \
    if (exact)                                                                                                      \
        hipLaunchKernelGGL((fir_tile_kernel<KP, true>), dim3(cnt), dim3(kTfThreads), lds, s, g, io, off, seg_in,    \
                           seg_out, pace_fx);                                                                       \
    else                                                                                                            \
        hipLaunchKernelGGL((fir_tile_kernel<KP, false>), dim3(cnt), dim3(kTfThreads), lds, s, g, io, off, seg_in,   \
                           seg_out, pace_fx);
        if (kp == 8) { DSP_TF(8) }
        else if (kp == 16) { DSP_TF(16) }
        else if (kp == 32) { DSP_TF(32) }
        else if (kp == 64) { DSP_TF(64) }
        else { DSP_TF(128) }
#undef DSP_TF
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

/* The tiled block layout [n / 2048][nch][2048] for a group the matrix-core kernel serves with every head folded
 * (no partial_init channels, 2 <= K <= 2047) and n a multiple of the 4096-output item: prologue (sample 0 of a
 * reset stream sits at the same place in both layouts), ONE persistent launch, the end state from the block's last
 * tile (it holds the K-1 samples the state depends on) and the commit.  io.ld_* = 2048. */
/* Which kernel serves a group on the tiled layout: the streaming kernel up to 128 taps -- except where the matrix-core
 * kernel may run (AUTO, no partial_init channel) and is faster, from 33 taps on (16 384 x 64K, fused arithmetic against
 * matrix cores: 48 taps 4.66 / 4.78 TB/s, 64 taps 4.07 / 4.28, 128 taps 2.39 / 2.97; profiles/r3_fir_tile.txt). */
static bool tf_serves_tiled(const GroupView &g, bool exact)
{
    const bool mfma_can = !g.has_partial && !exact && g.fir_mfma;
    return g.kmax <= kTfMaxTaps && (!mfma_can || g.kmax <= 32);
}

bool fir_tiled_supported(const GroupView &g, size_t n, bool exact)
{
    if (g.kmax < 2 || n == 0 || n % kMfmaSeg || n > ((size_t)1 << 30)) return false;
    if (tf_serves_tiled(g, exact)) return true;                 /* the streaming kernel, either arithmetic, either initialiser */
    return !g.has_partial && !exact && g.fir_mfma && g.kmax <= 2047;    /* the matrix-core kernel, every head folded */
}

hipError_t launch_fir_tiled(const GroupView &g, const BlockIO &io, size_t seg_in, size_t seg_out, bool exact,
                            double pace_ns, hipStream_t s)
{
    const int pro_blocks = (g.count + 255) / 256;
    hipLaunchKernelGGL(fir_prologue_kernel, dim3(pro_blocks), dim3(256), 0, s, g, io);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (tf_serves_tiled(g, exact)) {
        if (g.has_partial && (e = launch_fir_ramp_heads(g, io, s)) != hipSuccess) return e;   /* tile 0: rows a tile wide */
        if ((e = launch_fir_tile(g, io, seg_in, seg_out, exact, pace_ns, true, s)) != hipSuccess) return e;
        BlockIO last = io;                                      /* the block's last tile as a block of its own */
        last.in = io.in + (io.n / 2048 - 1) * seg_in;
        last.out = io.out + (io.n / 2048 - 1) * seg_out;
        last.n = 2048;
        const int passes = (g.kmax - 1 + kEpiTapPass - 1) / kEpiTapPass;
        const size_t lds_taps = (size_t)(2 * g.kmax + 4) * sizeof(double);
        hipLaunchKernelGGL(fir_epilogue_taps_kernel, dim3(passes, g.count), dim3(kWave), lds_taps, s, g, last);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        return launch_fir_commit(g, s);
    }
    const int hist = 16 * (mfma_steps16(g.kmax) - 1);
    int xtotal = hist + kMfmaSeg + 16;
    if (xtotal < kMfmaPreElems) xtotal = kMfmaPreElems;
    size_t mfma_lds = (size_t)(hist + 32 + xtotal + 2 * (xtotal >> 4) + 2) * sizeof(double);
    const int ptotal = hist + kMfmaSeg + 16;
    const size_t with_state = (size_t)(hist + 32 + ptotal + 2 * (ptotal >> 4) + 2 + g.kmax) * sizeof(double);
    if (with_state > mfma_lds) mfma_lds = with_state;
    long long grid = mfma_resident_groups(mfma_lds);
    if (grid > g.count) grid = g.count;
    hipLaunchKernelGGL((fir_mfma_kernel<true, true>), dim3((unsigned)grid), dim3(kMfmaWaves * kWave), mfma_lds, s, g, io,
                       0, (int)(io.n / kMfmaSeg), 0LL, 0LL, 1, seg_in, seg_out);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = launch_fir_nf_fixup(g, io, seg_in, seg_out, 0, s)) != hipSuccess) return e;
    BlockIO last = io;                                          /* the block's last tile as a block of its own */
    last.in = io.in + (io.n / 2048 - 1) * seg_in;
    last.out = io.out + (io.n / 2048 - 1) * seg_out;
    last.n = 2048;
    const int passes = (g.kmax - 1 + kEpiTapPass - 1) / kEpiTapPass;
    const size_t lds_taps = (size_t)(2 * g.kmax + 4) * sizeof(double);
    hipLaunchKernelGGL(fir_epilogue_taps_kernel, dim3(passes, g.count), dim3(kWave), lds_taps, s, g, last);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return launch_fir_commit(g, s);
}

}  // namespace dsp
