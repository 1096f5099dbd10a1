/*
 * pipe16_kernel.hip -- the pipelined time-parallel IIR kernel for orders 9..16 (gfx950).
 *
 * Same mathematics as iir_scan16_kernel (kernels.hip): a row of T samples is the affine map
 * s_end = A^T s_start + v, the zero-state vectors v come from the response table G, the rows are joined by a
 * prefix over matrix powers, and only then does the true recurrence run, one row per lane.  iir_scan16_kernel
 * runs these stages one after the other in ONE wave per channel (its 33 KiB tile lets a CU hold four waves), so
 * every stage's latency is exposed: measured, it moves 4.96 TB/s whatever the memory layout (tools/piece_probe.py).
 * Here a channel is served by a 2-wave workgroup with fixed roles, one 2048-sample segment apart (64 rows x 32
 * samples, two 16 KiB LDS tiles -- the arrangement of pipe_kernel.hip), and the table-driven stages run on the
 * FP64 matrix cores (v_mfma_f64_16x16x4f64; states above N are zero padding):
 *
 *   FRONT wave, segment i      park the prefetched x(i) in LDS, request x(i+1);
 *                              zero-state row vectors  V = G X                           32 MFMAs
 *                              up-sweep: pair and quad totals (rows 4q..4q+1, 4q..4q+3)   12 MFMAs
 *                              hand the four accumulator blocks to the back wave through LDS
 *   BACK wave, segment i-1     Kogge-Stone over the 16 quad totals with the carried state as
 *                              virtual quad -1 (levels that have decayed are skipped)   <= 16 MFMAs
 *                              down-sweep: true start state of every row                  12 MFMAs
 *                              lane <-> row transpose (permlane swaps), fused recurrence, store
 *
 * The prefix is work-efficient (72 MFMAs per segment against 112 for Kogge-Stone over all 64 rows) at no cost in
 * lane movement: accumulator block k holds the rows = k (mod 4), lane r of a block the quad r, so the up- and
 * down-sweeps are plain block-to-block products and only the 16-quad scan shifts lanes (DPP, inside a 16-lane
 * row).  Row j of a tile is therefore processed by lane 16 (j & 3) + (j >> 2).
 *
 * STREAM mode (Pipe16Ext.exch): one channel, segments dealt round-robin over the grid; see dsp_internal.h.  The
 * front wave also scans the quad totals from zero and publishes the segment's zero-state end vector, one step
 * before the workgroup that owns the next segment needs it.
 *
 * Whole, 16-byte aligned segments only (ragged launches stay on iir_scan16_kernel).  HBM traffic 16 B/sample (+ 512 B
 * per 32 KiB segment in stream mode).  Not bit-exact (re-associated); bar 1e-12 relative to peak, designs are
 * admitted by the host's conditioning probe of THIS algorithm (bank.hip: pipe16_probe).
 */
#include "dsp_internal.h"
#include "kernel_common.h"
#include "iir_steps.h"
#include "mfma_scan.h"

namespace dsp {

namespace {

constexpr int QT = kScan2T;                 /* 32 samples per row */
constexpr int kQSeg = QT * kWave;           /* 2048 samples */
constexpr int kQTile = kQSeg * 8;           /* 16384 bytes */
constexpr int kQPieces = kQSeg / 2 / kWave; /* 16 pieces of 16 B per lane */
constexpr int kQHand = kWave * 16 * 8;      /* hand-over buffer: 16 doubles per lane */
constexpr unsigned long long kQWaitTicks = 200000000ull;      /* 2 s of the 100 MHz realtime counter */

/* dst += M src for all 64 rows of a block pair: accumulator register t of lane group q holds state 4t + q, which
 * is the B layout of k-step t */
__device__ __forceinline__ void mat_apply(d4 &dst, const double (&M)[4], const d4 &src)
{
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dst = __builtin_amdgcn_mfma_f64_16x16x4f64(M[ks], src[ks], dst, 0, 0, 0);
}

/* one Kogge-Stone level over the 16 quads of a block: S_q += M S_(q-D); quad -1 is `cv` (lane 15 of a virtual
 * block in front: the carried state, or zero) */
template <int D>
__device__ __forceinline__ void quad_level(d4 &S, const double (&M)[4], const double (&cv)[4])
{
    double u[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) u[ks] = row_shift<D>(S[ks], cv[ks]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) S = __builtin_amdgcn_mfma_f64_16x16x4f64(M[ks], u[ks], S, 0, 0, 0);
}

/* levels of A^(32 2^k) that matter for the design: k = 0, 1 join rows into quads, k = 2 .. 5 join the quads */
__device__ __forceinline__ void quad_scan(d4 &S, const double (&Mq)[4][4], const double (&cv)[4], int nlev)
{
    if (nlev > 2) quad_level<1>(S, Mq[0], cv);
    if (nlev > 3) quad_level<2>(S, Mq[1], cv);
    if (nlev > 4) quad_level<4>(S, Mq[2], cv);
    if (nlev > 5) quad_level<8>(S, Mq[3], cv);
}

/* granules {value, 64-bit generation}: one 16-byte write-through store, one 16-byte load served past this CU's L1 */
/* four granules 64 bytes apart from four register tuples of their own, in ONE statement: no register of a store's
 * data is rewritten while an earlier store of the group may still be reading its own (with one tuple reused
 * for all four, the first granule of a group went out with a stale low word now and then -- only with two
 * waves per SIMD; tools/stream16_race.py) */
__device__ __forceinline__ void gran_store4(double *p, const d4 &val, double gen)
{
    d2 g0, g1, g2, g3;
    g0.x = val[0];
    g0.y = gen;
    g1.x = val[1];
    g1.y = gen;
    g2.x = val[2];
    g2.y = gen;
    g3.x = val[3];
    g3.y = gen;
    asm volatile(
        "s_nop 4\n\t"
        "global_store_dwordx4 %0, %1, off sc1\n\t"
        "global_store_dwordx4 %0, %2, off offset:64 sc1\n\t"
        "global_store_dwordx4 %0, %3, off offset:128 sc1\n\t"
        "global_store_dwordx4 %0, %4, off offset:192 sc1\n\t"
        "s_nop 4"
        ::"v"(p), "v"(g0), "v"(g1), "v"(g2), "v"(g3)
        : "memory");
}
__device__ __forceinline__ unsigned long long gran_gen(d2 v) { return (unsigned long long)__double_as_longlong(v.y); }

}  // namespace

template <int N, bool kUnitA0, bool kStream>
__global__ void __launch_bounds__(2 * kWave, 2)
iir_pipe16_kernel(GroupView g, BlockIO io, Pipe16Ext ext)
{
    static_assert(N > 8 && N <= 16, "orders 9..16");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile0 = smem;
    char *tile1 = smem + kQTile;
    char *hbuf = smem + 2 * kQTile;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    /* The recurrence of a back wave keeps its SIMD's vector ALU busy, the front wave's work is on the matrix
     * pipe: a SIMD should host one of each.  Workgroups arrive on a CU in the order of blockIdx / (number of CUs),
     * the first and the third on SIMDs 0-1, the second and the fourth on SIMDs 2-3, wave 0 first: the third and
     * fourth swap the roles of their waves (DSP_P16_NO_SWAP: diagnostic). */
#ifdef DSP_P16_SWAP
    const bool swap_roles = ((blockIdx.x >> 9) & 1) != 0;
#else
    const bool swap_roles = false;
#endif
    const bool front = (tid < kWave) != swap_roles;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int gi = kStream ? 0 : (int)blockIdx.x;
    const int w = kStream ? (int)blockIdx.x : 0;              /* first segment of this workgroup */
    const int G = kStream ? (int)gridDim.x : 1;               /* segment stride                  */
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    int nlev = as_const(g.levels2)[dsn];
    if (nlev > 6) nlev = 6;

    const int nseg_all = (int)(io.n / kQSeg);
    const int nseg = kStream ? (nseg_all > w ? (nseg_all - w + G - 1) / G : 0) : nseg_all;     /* this workgroup's */
    if (nseg <= 0) return;
    const double *src = io.in + (size_t)chan * io.ld_in;
    double *dst = io.out + (size_t)chan * io.ld_out;
    const bool seed_stream = as_const(g.init)[gi] != 0;
    const double tailv = seed_stream ? 0.0 : as_const(g.state)[(size_t)gi * g.state_ld + N];   /* s[N]: never written */
    cdouble_p cf = as_const(g.coefs) + as_const(g.coef_off)[dsn];
    const double *gt = g.pows + g.gtab_off[dsn];             /* G[n][i], n < 32, then Gc[i]        */
    const double *pm = g.pows + g.pow2_off[dsn];             /* A^(32 2^k), k < 7, [k][i][j]        */
    auto level_matrix = [&](double(&M)[4], int lvl) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + q4;
            const bool in = r16 < N && k < N;
            M[ks] = in ? pm[(size_t)lvl * N * N + (in ? r16 * N + k : 0)] : 0.0;
        }
    };
    /* tile geometry.  Row j sits at j * 256 bytes; its 16-byte column c at ((c ^ f(j)) << 4) with
     * f(j) = (j >> 2) ^ ((j & 3) << 2): conflict-free for the coalesced staging (16 lanes = 16 columns of a row),
     * for the matrix-core operand reads (16 lanes = the same column of 16 quads) and for the recurrence (16 lanes =
     * rows 4 r + k). */
#ifdef DSP_P16_STAMPS
    /* diagnostics: time (100 MHz ticks) spent between milestones, summed over the steps; slot s of wave role r at
     * stamps[(blockIdx * 2 + r) * 16 + s], behind the granules of the call (stream mode only) */
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memrealtime();
    auto stamp = [&](int slot) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        tacc[slot] += now - tprev;
        tprev = now;
    };
    auto stamps_out = [&](int role) {
        if (kStream && lane == 0) {
            unsigned long long *d = reinterpret_cast<unsigned long long *>(ext.exch + (size_t)nseg_all * 32) +
                                    ((size_t)blockIdx.x * 2 + role) * 16;
            for (int i = 0; i < 10; ++i) d[i] = tacc[i];
            d[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        /* HW_ID */
            d[11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       /* XCC_ID */
        }
    };
#else
    auto stamp = [](int) {};
    auto stamps_out = [](int) {};
#endif
    const int st_base = ((lane >> 4) << 8) | ((((lane & 15) ^ ((lane >> 4) << 2)) & 15) << 4);   /* piece i: ^ (i << 4), + i * 1024 */

    if (front) {
        /* =============================== FRONT =============================== */
        d2 pre[kQPieces];
        auto fetch = [&](int k) {
            if (k >= nseg) return;
            const double *p0 = src + ((size_t)w + (size_t)G * (size_t)k) * kQSeg + 2 * lane;
            const double *p1 = p0 + 4 * 2 * kWave, *p2 = p0 + 8 * 2 * kWave, *p3 = p0 + 12 * 2 * kWave;
            asm volatile(
                "global_load_dwordx4 %0, %16, off nt\n\t"
                "global_load_dwordx4 %1, %16, off offset:1024 nt\n\t"
                "global_load_dwordx4 %2, %16, off offset:2048 nt\n\t"
                "global_load_dwordx4 %3, %16, off offset:3072 nt\n\t"
                "global_load_dwordx4 %4, %17, off nt\n\t"
                "global_load_dwordx4 %5, %17, off offset:1024 nt\n\t"
                "global_load_dwordx4 %6, %17, off offset:2048 nt\n\t"
                "global_load_dwordx4 %7, %17, off offset:3072 nt\n\t"
                "global_load_dwordx4 %8, %18, off nt\n\t"
                "global_load_dwordx4 %9, %18, off offset:1024 nt\n\t"
                "global_load_dwordx4 %10, %18, off offset:2048 nt\n\t"
                "global_load_dwordx4 %11, %18, off offset:3072 nt\n\t"
                "global_load_dwordx4 %12, %19, off nt\n\t"
                "global_load_dwordx4 %13, %19, off offset:1024 nt\n\t"
                "global_load_dwordx4 %14, %19, off offset:2048 nt\n\t"
                "global_load_dwordx4 %15, %19, off offset:3072 nt"
                : "=&v"(pre[0]), "=&v"(pre[1]), "=&v"(pre[2]), "=&v"(pre[3]), "=&v"(pre[4]), "=&v"(pre[5]),
                  "=&v"(pre[6]), "=&v"(pre[7]), "=&v"(pre[8]), "=&v"(pre[9]), "=&v"(pre[10]), "=&v"(pre[11]),
                  "=&v"(pre[12]), "=&v"(pre[13]), "=&v"(pre[14]), "=&v"(pre[15])
                : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
        };
        /* piece i of the prefetched segment has arrived once at most 15 - i vector-memory operations are
         * outstanding: loads return in order, and this wave issues nothing behind a segment's loads that is still
         * in flight when the segment is parked except, in stream mode, the four granule stores of the step
         * before, which can only make the wait longer (see pipe_kernel.hip: park_wait for the operand tie) */
        auto park_wait = [&](int i, d2 &reg) {
            switch (kQPieces - 1 - i) {
#define DSP_PW(C) case C: asm volatile("s_waitcnt vmcnt(" #C ")" : "+v"(reg) : : "memory"); break;
                DSP_PW(0) DSP_PW(1) DSP_PW(2) DSP_PW(3) DSP_PW(4) DSP_PW(5) DSP_PW(6) DSP_PW(7)
                DSP_PW(8) DSP_PW(9) DSP_PW(10) DSP_PW(11) DSP_PW(12) DSP_PW(13) DSP_PW(14) DSP_PW(15)
#undef DSP_PW
            }
        };
        fetch(0);

        /* resident matrix-core operands (A layout: row = lane & 15, k = lane >> 4) */
        double Gr[QT / 4], Gt[4], M0[4], M1[4];
        {
            const bool live = r16 < N;
#pragma unroll
            for (int ks = 0; ks < QT / 4; ++ks) Gr[ks] = live ? gt[(4 * ks + q4) * N + (live ? r16 : 0)] : 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int st = 4 * t + q4;                                   /* accumulator layout */
                Gt[t] = st < N ? gt[QT * N + (st < N ? st : 0)] : 0.0;
            }
        }
        level_matrix(M0, 0);
        level_matrix(M1, 1);
        double Mq[kStream ? 4 : 1][4];
        if constexpr (kStream) {
#pragma unroll
            for (int l = 0; l < 4; ++l) level_matrix(Mq[l], 2 + l);
        }
        /* operand reads of pass 1: lane (r, q) of block k reads sample 4 ks + q of row 4 r + k */
        const int xlane = (r16 << 10) | ((q4 & 1) << 3) | ((((q4 >> 1) ^ r16) & 15) << 4);

        for (int it = 0; it <= nseg; ++it) {
            char *tile = (it & 1) ? tile1 : tile0;
            stamp(0);                                         /* tail of the step before: publish */
            lds_barrier();                                    /* barrier 0 */
            stamp(1);                                         /* waiting for the back wave */
            if (it < nseg) {
#pragma unroll
                for (int i = 0; i < kQPieces; ++i) {
                    park_wait(i, pre[i]);
                    *reinterpret_cast<d2 *>(tile + i * 1024 + (st_base ^ (i << 4))) = pre[i];
                }
            }
            stamp(2);                                         /* park: waiting for the segment's loads */
            lds_barrier();                                    /* barrier 1 */
            stamp(3);
            if (it >= nseg) break;
            fetch(it + 1);
            const long long m = (long long)w + (long long)G * it;            /* segment of the stream */
            const bool seeded = seed_stream && m == 0;

            /* ---- pass 1: zero-state end vectors of the 64 rows ---- */
            d4 acc[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[rb][t] = Gt[t] * tailv;
            }
#pragma unroll
            for (int ks = 0; ks < QT / 4; ++ks) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const double xv = *reinterpret_cast<const double *>(
                        tile + rb * 256 + (xlane ^ (((2 * ks) ^ (4 * rb)) << 4)));
                    acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gr[ks], xv, acc[rb], 0, 0, 0);
                }
            }
            if (seeded) {
                /* row 0 of a fresh stream: RTIIRFilter_stable_init on sample 0, then the recurrence; its end
                 * state replaces the row's zero-state vector (it does not depend on any carried state) */
                double e[N + 1];
#pragma unroll
                for (int i = 0; i <= N; ++i) e[i] = 0.0;
                if (lane == 0) {
                    double b[N + 1], a[N + 1];
#pragma unroll
                    for (int i = 0; i <= N; ++i) {
                        b[i] = cf[i];
                        a[i] = cf[N + 1 + i];
                    }
                    (void)iir_seed_ref<N>(*reinterpret_cast<const double *>(tile), e, b, a);
                    for (int k = 1; k < QT; ++k)
                        (void)iir_step_fused<N, kUnitA0>(
                            *reinterpret_cast<const double *>(tile + ((k >> 1) << 4) + (k & 1) * 8), e, b, a);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double sel = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (4 * t + q < N) {
                            const double ev = __shfl(e[4 * t + q < N ? 4 * t + q : 0], 0, kWave);
                            sel = q4 == q ? ev : sel;
                        }
                    }
                    if (r16 == 0) acc[0][t] = sel;
                }
            }

            stamp(4);                                         /* pass 1 */
            /* ---- up-sweep: block 1 <- rows 4q..4q+1, block 3 <- rows 4q..4q+3 ---- */
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(M0[ks], acc[0][ks], acc[1], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(M0[ks], acc[2][ks], acc[3], 0, 0, 0);
            }
            mat_apply(acc[3], M1, acc[1]);

            /* hand-over: 16 doubles per lane, unit u at (u ^ (lane & 7)) */
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d2 v;
                v.x = acc[u >> 1][2 * (u & 1)];
                v.y = acc[u >> 1][2 * (u & 1) + 1];
                *reinterpret_cast<d2 *>(hbuf + lane * 128 + ((u ^ (lane & 7)) << 4)) = v;
            }

            stamp(5);                                         /* up-sweep, hand-over */
            if constexpr (kStream) {
                /* the segment's zero-state end vector: the quad totals scanned from zero; lanes 15 hold it */
                d4 T = acc[3];
                const double z4[4] = {0.0, 0.0, 0.0, 0.0};
                quad_scan(T, Mq, z4, nlev);
                if (r16 == 15) {
                    double *p = ext.exch + ((size_t)m * 16 + q4) * 2;
                    const double gen = __longlong_as_double((long long)(ext.gen_base + (unsigned long long)m + 1ull));
                    gran_store4(p, T, gen);
                }
            }
        }
        stamps_out(0);
    } else {
        /* =============================== BACK ================================ */
        double b[N + 1], a[N + 1];
#pragma unroll
        for (int i = 0; i <= N; ++i) {
            b[i] = cf[i];
            a[i] = cf[N + 1 + i];
        }
        double M0[4], M1[4], Mq[4][4];
        level_matrix(M0, 0);
        level_matrix(M1, 1);
#pragma unroll
        for (int l = 0; l < 4; ++l) level_matrix(Mq[l], 2 + l);

        /* the state in front of a segment as virtual quad -1: lanes 15 hold state 4 ks + q */
        double cvirt[4] = {0.0, 0.0, 0.0, 0.0};
        auto cin_wait = [&](long long m_prev) {
            /* stream mode: the zero-state end vector of segment m_prev, published by the workgroup that owns it
             * while its front wave was one step ahead of where this wave is now */
            const double *p = ext.exch + ((size_t)m_prev * 16 + q4) * 2;
            const unsigned long long want = ext.gen_base + (unsigned long long)m_prev + 1ull;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                d2 v0, v1, v2, v3;
                asm volatile(
                    "global_load_dwordx4 %0, %4, off sc1\n\t"
                    "global_load_dwordx4 %1, %4, off offset:64 sc1\n\t"
                    "global_load_dwordx4 %2, %4, off offset:128 sc1\n\t"
                    "global_load_dwordx4 %3, %4, off offset:192 sc1\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                    : "v"(p)
                    : "memory");
                const bool ok = gran_gen(v0) == want && gran_gen(v1) == want && gran_gen(v2) == want && gran_gen(v3) == want;
                if (__all(ok)) {
                    cvirt[0] = r16 == 15 ? v0.x : 0.0;
                    cvirt[1] = r16 == 15 ? v1.x : 0.0;
                    cvirt[2] = r16 == 15 ? v2.x : 0.0;
                    cvirt[3] = r16 == 15 ? v3.x : 0.0;
                    return;
                }
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > kQWaitTicks) {
                    if (lane == 0) __hip_atomic_store(ext.error, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
            }
        };
        if (!seed_stream && (!kStream || w == 0)) {
            const double *st = g.state + (size_t)gi * g.state_ld;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 4 * ks + q4;
                const double c = k < N ? st[k < N ? k : 0] : 0.0;
                cvirt[ks] = r16 == 15 ? c : 0.0;
            }
        }
        double ylast = g.fval[gi];
        const int myrow = 4 * r16 + q4;                       /* the row this lane runs in pass 2 */
        const int rx = ((r16 ^ (q4 << 2)) & 15) << 4;         /* f(myrow) << 4 */

        lds_barrier();                                        /* step 0 belongs to the front wave alone */
        lds_barrier();
        double s[N + 1];
#pragma unroll
        for (int i = 0; i <= N; ++i) s[i] = 0.0;
        for (int it = 1; it <= nseg; ++it) {
            const int k = it - 1;
            const long long m = (long long)w + (long long)G * k;
            char *tile = (k & 1) ? tile1 : tile0;
            stamp(0);                                         /* stores issued */
            lds_barrier();                                    /* barrier 0 */
            stamp(1);                                         /* waiting for the front wave */
            d4 acc[4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const d2 v = *reinterpret_cast<const d2 *>(hbuf + lane * 128 + ((u ^ (lane & 7)) << 4));
                acc[u >> 1][2 * (u & 1)] = v.x;
                acc[u >> 1][2 * (u & 1) + 1] = v.y;
            }
            lds_barrier();                                    /* barrier 1 */
            stamp(2);
            const bool seeded = seed_stream && m == 0;
            if constexpr (kStream) {
                /* published a step ago by the front wave of the workgroup that owns segment m - 1.  (Requesting it
                 * a step early and carrying the registers over the loop's back edge would hide this round trip, but
                 * registers whose loads are in flight must not cross a point where the compiler may copy them:
                 * tools/check_prefetch_regs.py.) */
                if (m > 0) cin_wait(m - 1);
            }

            stamp(3);                                         /* start state from the neighbour */
            /* ---- the 16 quad totals scanned behind the carried state: true end state of every quad ---- */
            d4 S = acc[3];
            quad_scan(S, Mq, cvirt, nlev);
            d4 E;                                             /* true start state of every quad */
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) E[ks] = row_shift<1>(S[ks], cvirt[ks]);
            /* ---- down-sweep: end states of rows 4q, 4q+1, 4q+2 = start states of rows 4q+1, 4q+2, 4q+3 ---- */
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(M0[ks], E[ks], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(M1[ks], E[ks], acc[1], 0, 0, 0);
            }
            mat_apply(acc[2], M0, acc[1]);
            if constexpr (!kStream) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) cvirt[ks] = r16 == 15 ? S[ks] : 0.0;
            }

            /* lane (r, k) takes the 16 states of row 4 r + k: register a of group k <-> register k of group a */
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double w0 = E[t], w1 = acc[0][t], w2 = acc[1][t], w3 = acc[2][t];
                transpose4(w0, w1, w2, w3);
                if (4 * t + 0 < N) s[4 * t + 0 < N ? 4 * t + 0 : 0] = w0;
                if (4 * t + 1 < N) s[4 * t + 1 < N ? 4 * t + 1 : 0] = w1;
                if (4 * t + 2 < N) s[4 * t + 2 < N ? 4 * t + 2 : 0] = w2;
                if (4 * t + 3 < N) s[4 * t + 3 < N ? 4 * t + 3 : 0] = w3;
            }
            s[N] = tailv;

            stamp(4);                                         /* scan, down-sweep, transpose */
            /* ---- pass 2: the true recurrence, row per lane ---- */
            char *row = tile + myrow * 256;
            double y = ylast;
            if (!seeded) {
                constexpr int kBurst = QT / 4;
                auto burst = [&](int h) {
                    d2 xr[kBurst];
#pragma unroll
                    for (int q = 0; q < kBurst; ++q) xr[q] = *reinterpret_cast<const d2 *>(row + (((h + q) << 4) ^ rx));
#pragma unroll
                    for (int q = 0; q < kBurst; ++q) {
                        d2 xv = xr[q];
                        xv.x = iir_step_fused<N, kUnitA0>(xv.x, s, b, a);
                        xv.y = y = iir_step_fused<N, kUnitA0>(xv.y, s, b, a);
                        *reinterpret_cast<d2 *>(row + (((h + q) << 4) ^ rx)) = xv;
                    }
                };
                burst(0);
                burst(kBurst);
            } else {
                int kk = 0;
                if (lane == 0) {
                    double *p0 = reinterpret_cast<double *>(row + (0 ^ rx));
                    y = iir_seed_ref<N>(*p0, s, b, a);
                    *p0 = y;
                    kk = 1;
                }
                for (; kk < QT; ++kk) {
                    double *pk = reinterpret_cast<double *>(row + (((kk >> 1) << 4) ^ rx) + (kk & 1) * 8);
                    y = iir_step_fused<N, kUnitA0>(*pk, s, b, a);
                    *pk = y;
                }
            }
            ylast = y;

            stamp(5);                                         /* recurrence (+ the wait for the next start state) */
            /* store the segment, coalesced (this wave wrote the whole tile: in-order LDS) */
            wave_sync();
            double *drow = dst + (size_t)m * kQSeg;
#pragma unroll 8
            for (int i = 0; i < kQPieces; ++i)
                stream_store(reinterpret_cast<d2 *>(drow + 2 * (i * kWave + lane)),
                             *reinterpret_cast<const d2 *>(tile + i * 1024 + (st_base ^ (i << 4))));
        }

        stamps_out(1);
        /* the lane of row 63 of the stream's last segment holds the state the next call starts from */
        const long long m_last = (long long)w + (long long)G * (nseg - 1);
        if (lane == kWave - 1 && (!kStream || m_last == (long long)nseg_all - 1)) {
            double *st = g.state + (size_t)gi * g.state_ld;
#pragma unroll
            for (int i = 0; i <= N; ++i) st[i] = s[i];
            g.fval[gi] = ylast;
            g.init[gi] = 0;
        }
    }
}

template <int N>
static hipError_t launch_pipe16_n(const GroupView &g, const BlockIO &io, const Pipe16Ext &ext, int grid, hipStream_t s)
{
    const size_t lds = 2 * kQTile + kQHand;                  /* 40960 bytes: four workgroups fill a CU's 160 KiB */
    if (!io.aligned16 || io.n % kQSeg || io.n == 0) return hipErrorInvalidValue;
    if (ext.exch) {
        if (grid < 1 || grid > pipe_resident_round() || (size_t)grid > io.n / kQSeg) return hipErrorInvalidValue;
        if (g.unit_a0)
            hipLaunchKernelGGL((iir_pipe16_kernel<N, true, true>), dim3(grid), dim3(2 * kWave), lds, s, g, io, ext);
        else
            hipLaunchKernelGGL((iir_pipe16_kernel<N, false, true>), dim3(grid), dim3(2 * kWave), lds, s, g, io, ext);
    } else if (g.unit_a0)
        hipLaunchKernelGGL((iir_pipe16_kernel<N, true, false>), dim3(g.count), dim3(2 * kWave), lds, s, g, io, ext);
    else
        hipLaunchKernelGGL((iir_pipe16_kernel<N, false, false>), dim3(g.count), dim3(2 * kWave), lds, s, g, io, ext);
    return hipGetLastError();
}

hipError_t launch_iir_pipe16(int order, const GroupView &g, const BlockIO &io, const Pipe16Ext &ext, int grid,
                             hipStream_t s)
{
    switch (order) {
        case 9: return launch_pipe16_n<9>(g, io, ext, grid, s);
        case 10: return launch_pipe16_n<10>(g, io, ext, grid, s);
        case 11: return launch_pipe16_n<11>(g, io, ext, grid, s);
        case 12: return launch_pipe16_n<12>(g, io, ext, grid, s);
        case 13: return launch_pipe16_n<13>(g, io, ext, grid, s);
        case 14: return launch_pipe16_n<14>(g, io, ext, grid, s);
        case 15: return launch_pipe16_n<15>(g, io, ext, grid, s);
        case 16: return launch_pipe16_n<16>(g, io, ext, grid, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dsp
