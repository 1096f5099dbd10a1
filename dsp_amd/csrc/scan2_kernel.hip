/*
 * scan2_kernel.hip -- the two-wavefront time-parallel IIR kernel (gfx950).
 *
 * Same algorithm as iir_scan_kernel in kernels.hip (zero-state pass, prefix
 * combine of the affine chunk maps with powers of the companion matrix, true
 * pass), re-shaped around what profiling of the one-wave kernel showed on
 * MI355X (profiles/r1_scan_phases.txt):
 *
 *   - one wavefront per SIMD cannot issue FP64 VALU back to back (~5.6 cycles
 *     per instruction measured against 4) and exposes every LDS / scalar-load
 *     latency.  Here a channel is served by a 128-thread workgroup (2 waves),
 *     rows are 32 samples, the LDS footprint per workgroup stays ~40 KB, so a
 *     CU holds 4 workgroups = 2 waves per SIMD for a 1024-channel bank.
 *   - the matrix powers A^(32*2^k) live in LDS (broadcast ds_read_b128), not
 *     behind scalar loads whose latency a lone wave cannot hide.
 *   - combine levels whose matrix has decayed to nothing (every |entry| below
 *     kScan2Negligible, decided per design on the host) are skipped: a stable
 *     filter forgets its state, so for the config-2 design 4 of 7 levels matter.
 *
 * Segment = 128 rows x 32 samples = 4096 consecutive samples of the channel.
 * LDS tile rows are 256 B with the 16-byte column XOR-swizzled by (row & 15):
 * the per-lane row walk (ds_read_b128, lane = row) and the coalesced staging
 * (16 consecutive lanes = one row) are both bank-conflict free without padding.
 * HBM traffic is the algorithmic 16 B/sample.
 */
#include "dsp_internal.h"
#include "kernel_common.h"
#include "iir_steps.h"

namespace dsp {

namespace {

constexpr int T2 = kScan2T;
constexpr int kRows = kScan2Rows;
constexpr int kSeg2 = T2 * kRows;                 /* 4096 samples */
constexpr int kTileBytes = kRows * T2 * 8;        /* 32768 */
constexpr int kPiecesPerThread = kSeg2 / 2 / kRows;   /* 16 pieces of 16 B per thread */

__device__ __forceinline__ int tile_addr(int row, int gran)
{
    return row * (T2 * 8) + ((gran ^ (row & 15)) << 4);
}

__device__ __forceinline__ int tile_addr_elem(int row, int e)
{
    return tile_addr(row, e >> 1) + (e & 1) * 8;
}

}  // namespace

/* kStamp: diagnostic instantiation (never shipped): wave 0 sums s_memtime deltas per phase into
 * g.aux (8 uint64 per channel): park+barrier, prefetch issue, pass 1, combine, pass 2, store. */
template <int N, bool kUnitA0, bool kStamp = false>
__global__ void __launch_bounds__(kRows, 2)
iir_scan2_kernel(GroupView g, BlockIO io, int xbuf_rows)
{
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0;
    const unsigned long long rt_begin = kStamp ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int slot) {
        if (kStamp) {
            const unsigned long long now = __builtin_readcyclecounter();
            acc_t[slot] += now - t_prev;
            t_prev = now;
        }
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile = smem;
    double *mats = reinterpret_cast<double *>(smem + kTileBytes);
    double *xbuf0 = mats + g.levels2_max * N * N;         /* two exchange buffers, xbuf_rows x N each */
    double *xbuf1 = xbuf0 + xbuf_rows * N;
    double *sbuf = xbuf1 + xbuf_rows * N;                 /* N doubles: wave 0 lane 63 -> wave 1 lane 0 */
    double *cbuf = sbuf + N;                              /* N+2 doubles: carried state, last output */

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int gi = blockIdx.x;                            /* one workgroup per channel */
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    const int nlev = as_const(g.levels2)[dsn];
    cdouble_p gtab = as_const(g.pows) + as_const(g.gtab_off)[dsn];

    double b[N + 1], a[N + 1], carry[N + 1];
    {
        cdouble_p cf = as_const(g.coefs) + as_const(g.coef_off)[dsn];
#pragma unroll
        for (int i = 0; i <= N; ++i) {
            b[i] = cf[i];
            a[i] = cf[N + 1 + i];
        }
        const double *st = g.state + (size_t)gi * g.state_ld;
#pragma unroll
        for (int i = 0; i <= N; ++i) carry[i] = st[i];
    }
    double ylast = g.fval[gi];
    bool need_seed = g.init[gi] != 0;

    /* matrix powers of this design -> LDS, transposed to column-major */
    {
        const double *pw = g.pows + g.pow2_off[dsn];
        for (int e = tid; e < nlev * N * N; e += kRows) {
            const int lvl = e / (N * N), r = e % (N * N);
            mats[lvl * N * N + (r % N) * N + (r / N)] = pw[e];
        }
    }

    const size_t n = io.n;
    const double *src = io.in + (size_t)chan * io.ld_in;
    double *dst = io.out + (size_t)chan * io.ld_out;

    /* staging geometry: piece p = i*128 + tid -> row i*8 + (tid>>4), 16-byte column tid&15 */
    const int st_row = tid >> 4;
    const int st_even = st_row * (T2 * 8) + (((tid & 15) ^ st_row) << 4);          /* i even */
    const int st_odd = st_row * (T2 * 8) + (((tid & 15) ^ (st_row + 8)) << 4);     /* i odd  */

    d2 pre[kPiecesPerThread];
    auto fetch = [&](size_t base) {
        const size_t left = n - base;
        const int seg_len = left < (size_t)kSeg2 ? (int)left : kSeg2;
        if (seg_len == kSeg2 && io.aligned16) {
#pragma unroll
            for (int i = 0; i < kPiecesPerThread; ++i)
                pre[i] = stream_load(reinterpret_cast<const d2 *>(src + base + 2 * (i * kRows + tid)));
        } else {
#pragma unroll
            for (int i = 0; i < kPiecesPerThread; ++i) {
                const int e = 2 * (i * kRows + tid);
                d2 v = {0.0, 0.0};
                if (e < seg_len) v.x = src[base + e];
                if (e + 1 < seg_len) v.y = src[base + e + 1];
                pre[i] = v;
            }
        }
    };

    const int my_row = tid;
    const int rx = (my_row & 15) << 4;
    char *row = tile + my_row * (T2 * 8);

    fetch(0);
    unsigned long long rt0 = 0;
    if (kStamp) {
        t_prev = __builtin_readcyclecounter();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    for (size_t base = 0; base < n; base += kSeg2) {
        const size_t left = n - base;
        const int seg_len = left < (size_t)kSeg2 ? (int)left : kSeg2;
        const bool fast = seg_len == kSeg2 && !need_seed;

        /* ---- park the prefetched segment ---- */
#pragma unroll
        for (int i = 0; i < kPiecesPerThread; ++i)
            *reinterpret_cast<d2 *>(tile + i * (8 * T2 * 8) + ((i & 1) ? st_odd : st_even)) = pre[i];
        stamp(6);
        lds_barrier();   
        stamp(0);
        if (base + kSeg2 < n) fetch(base + kSeg2);
        stamp(1);

        /* samples this thread owns */
        int cnt = T2;
        if (!fast) {
            cnt = seg_len - my_row * T2;
            cnt = cnt < 0 ? 0 : (cnt > T2 ? T2 : cnt);
        }
        const bool seed_here = need_seed && tid == 0;
        const double tailv = need_seed ? 0.0 : carry[N];

        /* ---- pass 1: the state each row leaves behind when started from zero ----
         * full segments: v = sum_n G[n] x[n] with the per-design table G[n] = A^(31-n) B read
         * through scalar loads (N FMAs per sample); the carried state enters in the combine.
         * ragged / seeded segments: the recurrence itself, row 0 started from the carried state. */
        double v[N + 1];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (!fast && tid == 0) ? carry[i] : 0.0;
        v[N] = tailv;
        if (fast) {
#pragma unroll 2
            for (int q = 0; q < T2 / 2; ++q) {
                const d2 xv = *reinterpret_cast<const d2 *>(row + ((q << 4) ^ rx));
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = __builtin_fma(gtab[(2 * q) * N + i], xv.x, v[i]);
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = __builtin_fma(gtab[(2 * q + 1) * N + i], xv.y, v[i]);
            }
            if (tailv != 0.0) {
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = __builtin_fma(gtab[T2 * N + i], tailv, v[i]);
            }
        } else {
            int k = 0;
            if (seed_here) {
                (void)iir_seed_ref<N>(*reinterpret_cast<double *>(tile + tile_addr_elem(0, 0)), v, b, a);
                k = 1;
            }
            for (; k < cnt; ++k)
                (void)iir_step_fused<N, kUnitA0>(
                    *reinterpret_cast<double *>(tile + tile_addr_elem(my_row, k)), v, b, a);
        }

        stamp(2);
        /* ---- Kogge-Stone combine over the 128 rows: v <- end state after row j ---- */
        for (int lvl = 0; lvl < nlev; ++lvl) {
            const int d = 1 << lvl;
            /* this level's matrix: column-major in LDS, read by broadcast.  Columns are pulled
             * through a 3-deep register pipeline (a whole matrix in registers would not fit the
             * 256-VGPR budget of 2 waves per SIMD next to the prefetched segment); the first
             * columns are requested before the exchange so that their latency runs under the
             * barrier and the shuffles.  sched_barrier keeps hipcc from hoisting every load. */
            const double *M = mats + lvl * N * N;
            constexpr int kAhead = N < 3 ? N : 3;
            double mc[kAhead][N];
#pragma unroll
            for (int c = 0; c < kAhead; ++c) {
#pragma unroll
                for (int i = 0; i < N; ++i) mc[c][i] = M[c * N + i];
            }
            __builtin_amdgcn_sched_barrier(0);
            double *xb = (lvl & 1) ? xbuf1 : xbuf0;
            if (wave == 0 && lane >= 64 - d) {              /* rows 64-d..63 feed wave 1's rows 64..64+d-1 */
                double *slot = xb + (lane - (64 - d)) * N;
#pragma unroll
                for (int i = 0; i < N; ++i) slot[i] = v[i];
            }
            lds_barrier();   
            double u[N];
            if (d < 64) {
#pragma unroll
                for (int i = 0; i < N; ++i) u[i] = __shfl_up(v[i], d, kWave);
            }
            if (wave == 1 && lane < d) {
                const double *slot = xb + lane * N;
#pragma unroll
                for (int i = 0; i < N; ++i) u[i] = slot[i];
            }
            /* the carried state acts as a virtual row -1: at distance d it feeds row d-1 */
            const bool inject = fast && tid == d - 1;
            if (inject) {
#pragma unroll
                for (int i = 0; i < N; ++i) u[i] = carry[i];
            }
            /* v += M u, column by column: N independent accumulation chains per column */
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double col[N];
#pragma unroll
                for (int i = 0; i < N; ++i) col[i] = mc[j % kAhead][i];
                if (j + kAhead < N) {
#pragma unroll
                    for (int i = 0; i < N; ++i) mc[j % kAhead][i] = M[(j + kAhead) * N + i];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (tid >= d || inject) {
#pragma unroll
                    for (int i = 0; i < N; ++i) v[i] = __builtin_fma(col[i], u[j], v[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        stamp(3);
        /* ---- true start state of row j = end state of row j-1 ---- */
        if (tid == 63) {
#pragma unroll
            for (int i = 0; i < N; ++i) sbuf[i] = v[i];
        }
        lds_barrier();   
        double s[N + 1];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double prev = __shfl_up(v[i], 1, kWave);
            s[i] = prev;
        }
        if (tid == 64) {
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = sbuf[i];
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = carry[i];
        }
        s[N] = tailv;

        /* ---- pass 2: outputs, in place ---- */
        double y = ylast;
        if (fast) {
#pragma unroll 4
            for (int q = 0; q < T2 / 2; ++q) {
                d2 xv = *reinterpret_cast<d2 *>(row + ((q << 4) ^ rx));
                xv.x = iir_step_fused<N, kUnitA0>(xv.x, s, b, a);
                xv.y = y = iir_step_fused<N, kUnitA0>(xv.y, s, b, a);
                *reinterpret_cast<d2 *>(row + ((q << 4) ^ rx)) = xv;
            }
        } else {
            int k = 0;
            if (seed_here) {
                double *p0 = reinterpret_cast<double *>(tile + tile_addr_elem(0, 0));
                y = iir_seed_ref<N>(*p0, s, b, a);
                *p0 = y;
                k = 1;
            }
            for (; k < cnt; ++k) {
                double *pk = reinterpret_cast<double *>(tile + tile_addr_elem(my_row, k));
                y = iir_step_fused<N, kUnitA0>(*pk, s, b, a);
                *pk = y;
            }
        }

        stamp(4);
        /* ---- carry out: the row that consumed the segment's last sample ---- */
        const int jlast = (seg_len - 1) / T2;
        if (tid == jlast) {
#pragma unroll
            for (int i = 0; i <= N; ++i) cbuf[i] = s[i];
            cbuf[N + 1] = y;
        }
        need_seed = false;
        lds_barrier();   
#pragma unroll
        for (int i = 0; i <= N; ++i) carry[i] = cbuf[i];
        ylast = cbuf[N + 1];

        /* ---- store the segment, coalesced ---- */
        if (seg_len == kSeg2 && io.aligned16) {
#pragma unroll
            for (int i = 0; i < kPiecesPerThread; ++i)
                stream_store(reinterpret_cast<d2 *>(dst + base + 2 * (i * kRows + tid)),
                             *reinterpret_cast<const d2 *>(tile + i * (8 * T2 * 8) + ((i & 1) ? st_odd : st_even)));
        } else {
#pragma unroll 4
            for (int i = 0; i < kPiecesPerThread; ++i) {
                const int e = 2 * (i * kRows + tid);
                const d2 ov =
                    *reinterpret_cast<const d2 *>(tile + i * (8 * T2 * 8) + ((i & 1) ? st_odd : st_even));
                if (e < seg_len) dst[base + e] = ov.x;
                if (e + 1 < seg_len) dst[base + e + 1] = ov.y;
            }
        }
        stamp(5);
        lds_barrier();                                       /* tile is rewritten by the next park */
        stamp(7);
    }

    if (tid == 0) {
        double *st = g.state + (size_t)gi * g.state_ld;
#pragma unroll
        for (int i = 0; i <= N; ++i) st[i] = carry[i];
        g.fval[gi] = ylast;
        g.init[gi] = 0;
        if (kStamp) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(g.aux) + (size_t)gi * 8;
            for (int i = 0; i < 5; ++i) dbg[i] = acc_t[i];
            dbg[5] = rt_begin;                                    /* absolute 100 MHz ticks: WG start */
            dbg[6] = __builtin_amdgcn_s_memrealtime();            /* ... WG end */
            dbg[7] = dbg[6] - rt0;                                /* ticks over the stamped span */
        }
    }
}

template <int N>
static hipError_t launch_scan2_n(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    int xrows = 1;
    if (g.levels2_max > 0) xrows = 1 << (g.levels2_max - 1);
    if (xrows > 64) xrows = 64;
    const size_t lds = kTileBytes +
                       sizeof(double) * ((size_t)g.levels2_max * N * N + 2 * (size_t)xrows * N + N + (N + 2));
    if (N == 8 && g.unit_a0 && g.aux)             /* diagnostic build with phase stamps */
        hipLaunchKernelGGL((iir_scan2_kernel<8, true, true>), dim3(g.count), dim3(kRows), lds, s, g, io, xrows);
    else if (g.unit_a0)
        hipLaunchKernelGGL((iir_scan2_kernel<N, true>), dim3(g.count), dim3(kRows), lds, s, g, io, xrows);
    else
        hipLaunchKernelGGL((iir_scan2_kernel<N, false>), dim3(g.count), dim3(kRows), lds, s, g, io, xrows);
    return hipGetLastError();
}

hipError_t launch_iir_scan2(int order, const GroupView &g, const BlockIO &io, hipStream_t s)
{
    switch (order) {
        case 1: return launch_scan2_n<1>(g, io, s);
        case 2: return launch_scan2_n<2>(g, io, s);
        case 3: return launch_scan2_n<3>(g, io, s);
        case 4: return launch_scan2_n<4>(g, io, s);
        case 5: return launch_scan2_n<5>(g, io, s);
        case 6: return launch_scan2_n<6>(g, io, s);
        case 7: return launch_scan2_n<7>(g, io, s);
        case 8: return launch_scan2_n<8>(g, io, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dsp
