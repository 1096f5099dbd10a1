/*
 * pipe_kernel.hip -- the pipelined time-parallel IIR kernel (gfx950), orders <= 8.
 *
 * Same mathematics as the other scan kernels (a chunk of T samples is the affine
 * map s_end = A^T s_start + v; see kernels.hip / scan2_kernel.hip), arranged so
 * that ONLY the part that truly depends on the previous segment sits on the
 * segment-to-segment critical path.  Measured on MI355X, the two-wave kernel is
 * neither HBM- nor VALU-bound: its phases run back to back per channel and a CU
 * only ever holds 4 channels of a 1024-channel bank (profiles/r1_scan_phases.txt).
 *
 * A channel is served by a 2-wave workgroup with FIXED ROLES, software-pipelined
 * one 2048-sample segment apart (64 rows x 32 samples, two LDS tiles):
 *
 *   FRONT wave, segment i      carry-independent work
 *       park prefetched x(i) in LDS, keep two further segments' loads in flight,
 *       zero-state row vectors  v_j = sum_n G[n] x[n]          (table G = A^(31-n) B)
 *       Kogge-Stone prefix over the 64 rows with ZERO carry-in (levels that have
 *       decayed below 2^-66 skipped), publish P_j = state row j would start from
 *       if the segment started from zero
 *   BACK wave, segment i-1     the critical path
 *       s_j = P_j + A^(32 j) carry    one N x N mat-vec per lane, the lane's own matrix
 *                                     A^(32 j) held in registers for the whole launch
 *       run the 32 samples from s_j (fused recurrence), write y in place,
 *       carry = last row's state, store the segment coalesced
 *
 * Two workgroup barriers per segment keep the roles one segment apart; all other
 * synchronisation is implicit in the in-order LDS queue of each wave.
 * HBM traffic: 16 B/sample (x once in, y once out).  Not bit-exact (re-associated);
 * bar 1e-12 relative to peak, designs are admitted by the host's conditioning probe.
 */
#include "dsp_internal.h"
#include "kernel_common.h"
#include "iir_steps.h"

namespace dsp {

namespace {

constexpr int PT = kScan2T;                 /* 32 samples per row */
constexpr int kPRows = kWave;               /* 64 rows per segment */
constexpr int kPSeg = PT * kPRows;          /* 2048 samples */
constexpr int kPTile = kPRows * PT * 8;     /* 16384 bytes */
constexpr int kPPieces = kPSeg / 2 / kWave; /* 16 pieces of 16 B per lane */
constexpr int kPLevels = 6;                 /* log2(64) */

}  // namespace

/* kWhole: the launch consists of full, 16-byte aligned segments only (n a multiple of 2048).  A separate
 * instantiation, not a run-time branch: with the guarded ragged loads in the same kernel, hipcc's waitcnt
 * pass has to assume they may be in flight wherever their registers are reused, and puts vmcnt waits behind
 * the prefetch that make the front wave wait for nearly all of it (13 us into a launch instead of ~6). */
/* Rotating issue priority.  The four workgroups that share a CU are otherwise arbitrated oldest-first, and the
 * one that arrived first finishes a 64K-sample block 10 % ahead of the other three, which then run the tail of
 * the launch on a half-empty memory system.  Each workgroup takes priority (slot + step) mod 4, slot = its
 * arrival order on the CU as far as the dispatch order tells (blockIdx / 256). */
__device__ __forceinline__ void rotate_priority(int k)
{
    switch (k & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

/* ---- chained launches (ChainView, dsp_internal.h) ------------------------------------------------------- */

constexpr unsigned long long kChainTimeoutTicks = 200000000ull;      /* 2 s of the 100 MHz realtime counter */

/* One 16-byte granule {value, generation}, written by ONE 16-byte sc1 (write-through) store and read by one
 * 16-byte sc1 load (served by L2 / memory, never by this CU's L1): a reader polls until the generation it
 * expects shows up (observed untorn on gfx950; MI355X_MICROARCH.md, "R2's granule needs no ordering at all"). */
__device__ __forceinline__ d2 granule_load(const double *p)
{
    d2 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void granule_store(double *p, double value, unsigned int gen)
{
    d2 v;
    v.x = value;
    v.y = __longlong_as_double((long long)(unsigned long long)gen);
    /* the trailing wait states keep whatever the compiler puts into v's registers next away from a store that
     * may still be reading them (a hazard it covers for stores it knows, not for inline ones) */
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned int granule_gen(d2 v) { return (unsigned int)__double_as_longlong(v.y); }

/* Lane i (i < cnt) of the calling wave fetches granule i of a record; returns once all cnt carry generation
 * `want` (bounded: after ~2 s the error word is raised and whatever was read is returned). */
__device__ __forceinline__ d2 record_load(const double *rec, int cnt, unsigned int want, unsigned int *error)
{
    const int lane = threadIdx.x & 63;
    const double *p = rec + 2 * (lane < cnt ? lane : cnt - 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const d2 v = granule_load(p);
        if (__all(granule_gen(v) == want)) return v;
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > kChainTimeoutTicks) {
            if (lane == 0) __hip_atomic_store(error, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v;
        }
    }
}

/* kChain (whole-segment launches only): the launch is one link of a chain, see ChainView.
 * kSos: every channel is a cascade of N / 2 second-order sections run as ONE state-space system of N states
 * (iir_steps.h: sos_step_fused): the tables G, A^(32 2^k), A^(32 j) are those of the cascade's block-triangular
 * system matrix (bank.hip), the true pass runs the sections one after the other per sample; coefficients and the
 * state row are 3 per section. */
template <int N, bool kUnitA0, bool kWhole, bool kChain, bool kSos = false>
__global__ void __launch_bounds__(2 * kWave, 2)
iir_pipe_kernel(GroupView g, BlockIO io, PipeExt ext)
{
    static_assert(kWhole || !kChain, "chained launches consist of whole segments");
    static_assert(!kSos || (N % 2 == 0 && !kChain), "two states per section; cascades are not chained");
    constexpr int kNC = kSos ? 3 * (N / 2) : N + 1;         /* taps per coefficient array */
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *tile0 = smem;
    char *tile1 = smem + kPTile;
    /* 64 x N start states, one slot per row.  The slot pitch in 16-byte units must be odd: lanes write and
     * read their slots with b128 accesses, 16 lanes per LDS cycle, and at a pitch of N = 8 doubles (4 units)
     * lanes l and l+4 meet in the same banks -- the 4.7e6 SQ_LDS_BANK_CONFLICT cycles per launch of round 1,
     * all of them from these two accesses (the swizzled sample tiles are conflict-free). */
    constexpr int kPPitch = (N % 2 == 0 && (N / 2) % 2 == 0) ? N + 2 : N;
    double *pbuf = reinterpret_cast<double *>(smem + 2 * kPTile);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool front = tid < kWave;
    int gi = blockIdx.x;
    bool chain_tail = false;                                /* state comes from a carry record, not from g.state */
    if constexpr (kChain) {
        if (tid == 0)
            __hip_atomic_fetch_add(ext.chain.started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ext.chain.prev_queue) {
            /* ticket = blockIdx.x: the channel that the (blockIdx.x + 1)-th finisher of the previous launch
             * published.  One poller; the decision (a channel, or give up) reaches both waves through LDS. */
            chain_tail = true;
            int *slot = reinterpret_cast<int *>(smem);
            if (tid == 0) {
                const unsigned long long *q = ext.chain.prev_queue + blockIdx.x;
                const unsigned int want = ext.chain.gen - 1;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                int got = -1;
                for (;;) {
                    const unsigned long long v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned int)(v >> 32) == want) {
                        got = (int)(unsigned int)v - 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kChainTimeoutTicks) {
                        __hip_atomic_store(ext.chain.error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                if (got >= g.count) got = -1;
                slot[0] = got;
            }
            lds_barrier();
            gi = __builtin_amdgcn_readfirstlane(slot[0]);
            if (gi < 0) return;                             /* both waves: the chain is broken, host reports it */
        }
    }
    const int prio_slot = (int)(blockIdx.x >> 8);
    const int chan = g.chan ? as_const(g.chan)[gi] : g.chan0 + gi;
    const int dsn = as_const(g.design)[gi];
    int nlev = as_const(g.levels2)[dsn];
    if (nlev > kPLevels) nlev = kPLevels;

    /* diagnostics (DSP_AMD_SCAN_STAMPS=1 allocates g.aux): 100 MHz timestamps of one workgroup's milestones */
#ifdef DSP_PIPE_STAMPS
    unsigned long long *dbg = g.aux ? reinterpret_cast<unsigned long long *>(g.aux) + (size_t)gi * 32 : nullptr;
#else
    constexpr unsigned long long *dbg = nullptr;       /* tools/variant.sh -DDSP_PIPE_STAMPS builds the stamped twin */
#endif
    if (dbg && lane == 0 && front) {
        dbg[0] = __builtin_amdgcn_s_memrealtime();
        dbg[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        /* HW_ID: simd, cu, sh, se */
        dbg[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       /* XCC_ID */
        dbg[16] = __builtin_amdgcn_s_memtime();                    /* shader clock, with [17] */
    }
    auto stamp = [&](int slot) {
        if (dbg && lane == 0) dbg[slot] = __builtin_amdgcn_s_memrealtime();
    };
    const size_t n = io.n;
    const int nseg = (int)((n + kPSeg - 1) / kPSeg);
    const double *src = io.in + (size_t)chan * io.ld_in;
    double *dst = io.out + (size_t)chan * io.ld_out;
    /* wave-uniform words of this launch's inputs come through scalar loads: a vector load issued next to the
     * segment prefetch below would make its first use wait for the whole prefetch (vmcnt counts in order) */
    bool seed_stream = as_const(g.init)[gi] != 0;
    double tail0 = kSos ? 0.0 : as_const(g.state)[(size_t)gi * g.state_ld + N];     /* s[N]: read, never written */
    double *rec = nullptr;
    if constexpr (kChain) {
        rec = ext.chain.records + (size_t)gi * (2 * (N + 2));
        /* behind at least one whole block the stream is initialised; s[N] rides in the record (the head launch
         * may have zeroed it) and reaches the front wave through LDS: the back wave, idle until segment 0 is
         * published, reads the record while the front's prefetch is in flight */
        if (chain_tail) seed_stream = false;
    }
    /* two words behind the start-state slots (chained launches only): [0] s[N] from the record, back -> front
     * before the first barrier; [1] this channel's position in the ready queue, front -> back */
    double *xch = reinterpret_cast<double *>(smem + 2 * kPTile) + kPRows * (N + 2);
    const size_t seg_in = kWhole ? ext.seg_in : (size_t)kPSeg, seg_out = kWhole ? ext.seg_out : (size_t)kPSeg;

    /* The front wave's first two segments are requested before anything else: the table loads below
     * then run under the cold-start HBM latency instead of in front of it. */
    d2 pre0[kPPieces], pre1[kPPieces];

    /* kWhole launches issue the prefetch as opaque instructions and count vmcnt by hand (park_wait below).
     * hipcc's waitcnt insertion is path-insensitive: because a fetch may issue nothing (past the end) it assumed
     * at every tile write that the youngest 16 loads in flight were the ones needed and drained BOTH prefetched
     * segments -- an effective prefetch distance of one -- and it made the front wait for loads of the back
     * wave's branch that follow it in the linearised flow graph.  By hand the distance is really two: while
     * segment i is parked, the loads of segment i+1 stay in flight.  The front wave issues no other vector-memory
     * instruction, so the counter holds these loads only (the stamped diagnostic build adds stores and therefore
     * waits for everything). */
#ifdef DSP_PIPE_STAMPS
    constexpr int kNewerInFlight = 0;
#else
    constexpr int kNewerInFlight = kPPieces;
#endif
    auto fetch = [&](int seg, d2(&pre)[kPPieces]) {
        const size_t base = (size_t)seg * kPSeg;
        if (!kWhole && base >= n) return;
        const size_t left = base < n ? n - base : 0;
        const int seg_len = left < (size_t)kPSeg ? (int)left : kPSeg;
        if constexpr (kWhole) {
            static_assert(kPPieces == 16, "four base pointers x four 1 KiB offsets");
            /* Past the end the same sixteen instructions are issued on one address for all lanes (the head of the
             * channel's block: sixteen cache lines, nothing to speak of): every park below then has exactly one
             * younger segment's worth of loads behind it, and ONE code path.  With a second path for the last
             * segment (no younger loads, smaller counts) the compiler resolved the join of the two by COPYING the
             * prefetch registers in front of the counted waits -- copies of registers whose loads were still in
             * flight; orders 1 and 2, whose steps are short, then ran their last segment on stale samples now and
             * then (tools/check_prefetch_regs.py looks for such copies in every instantiation at build time). */
            const double *p0 = base < n ? src + (size_t)seg * seg_in + 2 * lane : src;   /* piece i of the lane: + i * 1024 bytes */
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
        } else if (seg_len == kPSeg && io.aligned16) {
#pragma unroll
            for (int i = 0; i < kPPieces; ++i)
                pre[i] = stream_load(reinterpret_cast<const d2 *>(src + base + 2 * (i * kWave + lane)));
        } else {
#pragma unroll
            for (int i = 0; i < kPPieces; ++i) {
                const int e = 2 * (i * kWave + lane);
                d2 v = {0.0, 0.0};
                if (e < seg_len) v.x = src[base + e];
                if (e + 1 < seg_len) v.y = src[base + e + 1];
                pre[i] = v;
            }
        }
    };
    /* kWhole: piece i of a prefetched segment has arrived once at most (newer + 15 - i) loads are outstanding,
     * `newer` = loads issued after that segment's.  The wait clobbers "memory", so the tile write that follows
     * it (a memory operation) cannot be scheduled above it, and it names the loaded register as an in/out
     * operand: whatever reads that register afterwards reads the wait's output, so no read or copy of it can be
     * scheduled above the wait (a copy made BEFORE the wait would have to feed the wait's operand -- the compiler
     * has no reason to make one, and the listing shows none). */
    auto park_wait = [&](int i, bool newer_in_flight, d2 &reg) {
        if constexpr (kWhole) {
            const int cnt = (newer_in_flight ? kNewerInFlight : 0) + kPPieces - 1 - i;
            switch (cnt) {
#define DSP_PW(C) case C: asm volatile("s_waitcnt vmcnt(" #C ")" : "+v"(reg) : : "memory"); break;
                DSP_PW(0) DSP_PW(1) DSP_PW(2) DSP_PW(3) DSP_PW(4) DSP_PW(5) DSP_PW(6) DSP_PW(7)
                DSP_PW(8) DSP_PW(9) DSP_PW(10) DSP_PW(11) DSP_PW(12) DSP_PW(13) DSP_PW(14) DSP_PW(15)
                DSP_PW(16) DSP_PW(17) DSP_PW(18) DSP_PW(19) DSP_PW(20) DSP_PW(21) DSP_PW(22) DSP_PW(23)
                DSP_PW(24) DSP_PW(25) DSP_PW(26) DSP_PW(27) DSP_PW(28) DSP_PW(29) DSP_PW(30) DSP_PW(31)
#undef DSP_PW
            }
        }
    };

    /* Paced launch (PipeExt.pace_fx): the schedule starts now.  It lives in an LDS word behind the exchange words
     * (the next due time in 1/1024 ticks), not in scalar registers, which the front wave's table batches need. */
    const bool paced = kWhole && !kChain && ext.pace_fx != 0;
    unsigned long long *psch = reinterpret_cast<unsigned long long *>(xch + 2);
    if (paced && front && lane == 0)
        psch[0] = (__builtin_amdgcn_s_memrealtime() << 10) + ext.pace_fx;      /* step 1: one period after the start */
    if (front) {
        fetch(0, pre0);
        fetch(1, pre1);
    } else {
        /* The back wave, idle until segment 0 is published, warms the scalar cache of the CU for the front
         * wave: the front reads its two tables -- the zero-state response G and the prefix matrices
         * A^(32 2^k) -- through scalar loads, a few 64-byte lines per batch; cold, every batch is a trip to
         * L2 and, in a full launch, to HBM behind the prefetch of a thousand workgroups.  Touched here, the
         * lines are requested while that prefetch is still being issued.  All kPLevels matrices (the table
         * holds kScan2Levels >= kPLevels), whatever nlev is: a loop with a run-time bound would wait for
         * every line separately. */
        cint_p gw = (cint_p)(uintptr_t)(g.pows + as_const(g.gtab_off)[dsn]);
        int touched = 0;
#pragma unroll
        for (int l = 0; l < ((PT + 1) * N * 8 + 63) / 64; ++l) touched |= gw[l * 16];
        asm volatile("" ::"s"(touched));
        cint_p mw = (cint_p)(uintptr_t)(g.pows + as_const(g.pow2_off)[dsn]);
        touched = 0;
#pragma unroll
        for (int l = 0; l < (kPLevels * N * N * 8 + 63) / 64; ++l) touched |= mw[l * 16];
        asm volatile("" ::"s"(touched));
    }
    if (tid < N) pbuf[tid] = 0.0;                           /* row 0 never receives a prefix */

    /* staging geometry (both roles): piece p = i*64 + lane -> row i*4 + (lane>>4), column lane&15 */
    const int st_r = lane >> 4;
    int st_addr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
        st_addr[v] = st_r * (PT * 8) + (((lane & 15) ^ (v * 4 + st_r)) << 4);
    const int rx = (lane & 15) << 4;

    /* coefficients live in SGPRs; the front wave only needs them for the seeded first row and loads them
     * there, so that its scalar registers stay free for the table batches */
    auto load_coefs = [&](double(&b)[kNC], double(&a)[kNC]) {
        cdouble_p cf = as_const(g.coefs) + as_const(g.coef_off)[dsn];
#pragma unroll
        for (int i = 0; i < kNC; ++i) {
            b[i] = cf[i];
            a[i] = cf[kNC + i];
        }
    };
    auto seed = [](double x, double(&s)[N + 1], const double(&b)[kNC], const double(&a)[kNC]) {
        if constexpr (kSos) return sos_seed_z<N / 2>(x, s, b, a);
        else return iir_seed_ref<N>(x, s, b, a);
    };
    auto step = [](double x, double(&s)[N + 1], const double(&b)[kNC], const double(&a)[kNC]) {
        if constexpr (kSos) return sos_step_fused<N / 2, kUnitA0>(x, s, b, a);
        else return iir_step_fused<N, kUnitA0>(x, s, b, a);
    };

    if (front) {
        /* =============================== FRONT =============================== */
        cdouble_p gtab = as_const(g.pows) + as_const(g.gtab_off)[dsn];
        cdouble_p ptab = as_const(g.pows) + as_const(g.pow2_off)[dsn];      /* [level][i][j], row-major */
        auto front_step = [&](int it, d2(&pre)[kPPieces]) {
            char *tile = (it & 1) ? tile1 : tile0;
            rotate_priority(prio_slot + it);
            if (it == 0) stamp(1);                            /* set-up done */
            if (it == 8) stamp(11);
            if (it == 9) stamp(22);
            lds_barrier();                                    /* barrier 0 */
            if (it < nseg) {
                /* the loads issued after this segment's (segment it+1, or its stand-in past the end) are younger
                 * and may still be flying */
#pragma unroll
                for (int i = 0; i < kPPieces; ++i) {
                    park_wait(i, true, pre[i]);
                    *reinterpret_cast<d2 *>(tile + i * (4 * PT * 8) + st_addr[i & 3]) = pre[i];
                }
            }
            lds_barrier();                                    /* barrier 1 */
            if (it >= nseg) return;
            if constexpr (kChain) {
                if (chain_tail && it == 0) tail0 = xch[0];
            }
            if (it == 0) stamp(7);
            if (it == 8) stamp(12);
            if (it == 9) stamp(23);
            if (paced && it > 0) {
                /* the prefetch of step `it` is not issued before  launch start + it x period */
                const unsigned long long due_fx = psch[0];
                while (__builtin_amdgcn_s_memrealtime() < (due_fx >> 10)) __builtin_amdgcn_s_sleep(1);
                if (lane == 0) psch[0] = due_fx + ext.pace_fx;
            }
            fetch(it + 2, pre);

            const bool seeded = seed_stream && it == 0;
            const double tailv = seed_stream ? 0.0 : tail0;
            const char *row = tile + lane * (PT * 8);
            double v[N + 1];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = 0.0;
            v[N] = tailv;
            {
            /* the row comes out of LDS in one burst; the table arrives in scalar batches of kZB samples
             * (<= 32 SGPRs each), so a batch's wait is not also a wait for LDS */
            d2 xr[PT / 2];
#pragma unroll
            for (int q = 0; q < PT / 2; ++q) xr[q] = *reinterpret_cast<const d2 *>(row + ((q << 4) ^ rx));
            /* Scalar loads return out of order, so a wait is always a wait for ALL of them: a batch requested
             * after the wait for the one in use is the only kind that overlaps arithmetic.  Two register sets
             * take turns -- the first FMA of a batch carries the wait, then the next batch is requested, then
             * the remaining FMAs of the current one issue while it travels (~100 cycles from the scalar cache,
             * tools/sload_latency.hip, against 64 cycles of FMAs per batch). */
            constexpr int kZB = N <= 4 ? 4 : 2;
            static_assert(PT % (2 * kZB) == 0, "batches come in pairs");
            double ga[kZB][N], gb[kZB][N];
            auto zs_load = [&](double(&gk)[kZB][N], int k0) {
#pragma unroll
                for (int k = 0; k < kZB; ++k) {
#pragma unroll
                    for (int i = 0; i < N; ++i) gk[k][i] = gtab[(k0 + k) * N + i];
                }
            };
            auto zs_mac = [&](const double(&gk)[kZB][N], int k0, int from, int to) {
#pragma unroll
                for (int f = from; f < to; ++f) {
                    const int k = f / N, i = f % N;
                    const double xk = ((k0 + k) & 1) ? xr[(k0 + k) >> 1].y : xr[(k0 + k) >> 1].x;
                    v[i] = __builtin_fma(gk[k][i], xk, v[i]);
                }
            };
            zs_load(ga, 0);
#pragma unroll
            for (int k0 = 0; k0 < PT; k0 += 2 * kZB) {
                __builtin_amdgcn_sched_barrier(0);
                zs_mac(ga, k0, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                zs_load(gb, k0 + kZB);
                __builtin_amdgcn_sched_barrier(0);
                zs_mac(ga, k0, 1, kZB * N);
                __builtin_amdgcn_sched_barrier(0);
                zs_mac(gb, k0 + kZB, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (k0 + 2 * kZB < PT) zs_load(ga, k0 + 2 * kZB);
                __builtin_amdgcn_sched_barrier(0);
                zs_mac(gb, k0 + kZB, 1, kZB * N);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tailv != 0.0) {
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = __builtin_fma(gtab[PT * N + i], tailv, v[i]);
            }
            }
            if (it == 0) stamp(8);
            if (it == 8) stamp(13);
            if (seeded && lane == 0) {
                /* row 0 of a fresh stream: RTIIRFilter_stable_init on sample 0, then the recurrence;
                 * its end state does not depend on any carried state */
                double b[kNC], a[kNC];
                load_coefs(b, a);
#pragma unroll
                for (int i = 0; i <= N; ++i) v[i] = 0.0;
                const size_t left = n < (size_t)PT ? n : (size_t)PT;
                (void)seed(*reinterpret_cast<const double *>(row + (0 ^ rx)), v, b, a);
                for (int k = 1; k < (int)left; ++k)
                    (void)step(*reinterpret_cast<const double *>(row + (((k >> 1) << 4) ^ rx) + (k & 1) * 8), v, b, a);
            }

            /* Kogge-Stone over the 64 rows, zero carry-in.  The level's matrix A^(32 d) is wave-uniform: it
             * comes through the scalar cache in batches of kKR rows (<= 64 SGPRs), not through LDS, whose
             * pipe the two roles' tile traffic already keeps busy. */
            /* Row batches of the level matrices take turns in two register sets like the table above; the first
             * batch of the next level travels under the last FMAs of this one (the table holds kScan2Levels >
             * kPLevels matrices, so the request past the last level stays inside it). */
            constexpr int kKR = N <= 4 ? N : 2;
            constexpr int kKB = (N + kKR - 1) / kKR;
            static_assert(kScan2Levels > kPLevels, "the look-ahead load of level lvl + 1 needs a matrix there");
            double mbuf[2][kKR][N];
            auto px_load = [&](double(&m)[kKR][N], int lvl, int bi) {
#pragma unroll
                for (int r = 0; r < kKR; ++r) {
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        m[r][j] = (bi * kKR + r < N) ? ptab[(lvl * N + (bi * kKR + r < N ? bi * kKR + r : 0)) * N + j] : 0.0;
                }
            };
            px_load(mbuf[0], 0, 0);
            for (int lvl = 0; lvl < nlev; ++lvl) {
                const int d = 1 << lvl;
                double u[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double t = __shfl_up(v[i], d, kWave);
                    u[i] = lane >= d ? t : 0.0;
                }
#pragma unroll
                for (int bi = 0; bi < kKB; ++bi) {
                    double(&cur)[kKR][N] = mbuf[bi & 1];
                    double(&nxt)[kKR][N] = mbuf[(bi + 1) & 1];
                    auto px_mac = [&](int from, int to) {
#pragma unroll
                        for (int j = from; j < to; ++j) {
#pragma unroll
                            for (int r = 0; r < kKR; ++r)
                                if (bi * kKR + r < N)
                                    v[bi * kKR + r] = __builtin_fma(cur[r][j], u[j], v[bi * kKR + r]);
                        }
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    px_mac(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (bi + 1 < kKB) px_load(nxt, lvl, bi + 1);
                    else if (kKB % 2 == 0) px_load(nxt, lvl + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    px_mac(1, N);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kKB % 2 == 1) px_load(mbuf[0], lvl + 1, 0);
            }
            /* publish: row j+1 starts (with zero carry-in) from the state row j ends in */
            if (lane < kPRows - 1) {
                double *slot = pbuf + (lane + 1) * kPPitch;
#pragma unroll
                for (int i = 0; i < N; ++i) slot[i] = v[i];
            }
            if constexpr (kChain) {
                /* The front wave is done one segment before the back wave: it claims the channel's position in
                 * the ready queue now, so that the returning atomic is not waited for at the very end, where the
                 * workgroup would hold its slot idle for the round trip. */
                if (it == nseg - 1 && lane == 0) {
                    const unsigned long long pos =
                        __hip_atomic_fetch_add(ext.chain.finished, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                        ext.chain.finished_base;
                    xch[1] = (double)pos;
                }
            }
            if (it == 0) stamp(2);                            /* segment 0 published */
            if (it == 8) {
                stamp(14);
                if (dbg && lane == 0) dbg[17] = __builtin_amdgcn_s_memtime();
            }
        };

        for (int it = 0; it <= nseg; it += 2) {
            front_step(it, pre0);
            if (it + 1 <= nseg) front_step(it + 1, pre1);
        }

    } else {
        /* =============================== BACK ================================ */
        double b[kNC], a[kNC];
        load_coefs(b, a);
        double carry[N + 1];
        double ylast;
        if (kChain && chain_tail) {
            const d2 gv = record_load(rec, N + 2, ext.chain.gen - 1, ext.chain.error);
#pragma unroll
            for (int i = 0; i <= N; ++i) carry[i] = __shfl(gv.x, i, kWave);
            ylast = __shfl(gv.x, N + 1, kWave);
            if (lane == 0) xch[0] = carry[N];             /* for the front wave, read behind the first barriers */
        } else if constexpr (kSos) {
            const double *st = g.state + (size_t)gi * g.state_ld;       /* 3 slots per section, the third is 0 */
#pragma unroll
            for (int k = 0; k < N / 2; ++k) {
                carry[2 * k] = st[3 * k];
                carry[2 * k + 1] = st[3 * k + 1];
            }
            carry[N] = 0.0;
            ylast = g.fval[gi];
        } else {
            const double *st = g.state + (size_t)gi * g.state_ld;
#pragma unroll
            for (int i = 0; i <= N; ++i) carry[i] = st[i];
            ylast = g.fval[gi];
        }
        /* this lane's propagator A^(32*lane), row-major, resident for the whole launch */
        double tp[N * N];
        {
            const double *tt = g.pows + g.ttab_off[dsn] + lane;          /* table is [e][lane] */
#pragma unroll
            for (int e = 0; e < N * N; ++e) tp[e] = tt[e * kWave];
        }

        stamp(9);
        /* step 0 belongs to the front wave alone; its two barriers come before anything that needs the
         * tables loaded above, so that the front is not held at its first barrier until they have arrived */
        lds_barrier();
        lds_barrier();
        /* The propagator's loads must be retired HERE, on every path out of this branch: hipcc's waitcnt pass
         * works on the linearised flow graph, in which the front wave's code follows this branch, and a load it
         * believes pending on registers the front reuses (the loop below may run zero times, as far as it knows)
         * costs the front a vmcnt(0) -- a full trip to HBM for the prefetch it has just issued, every segment. */
#pragma unroll
        for (int e = 0; e < N * N; e += 8) {
            if constexpr (N * N - 0 >= 8) {
                if (e + 8 <= N * N)
                    asm volatile("" ::"v"(tp[e]), "v"(tp[e + 1]), "v"(tp[e + 2]), "v"(tp[e + 3]), "v"(tp[e + 4]),
                                 "v"(tp[e + 5]), "v"(tp[e + 6]), "v"(tp[e + 7]));
            }
        }
#pragma unroll
        for (int e = (N * N) / 8 * 8; e < N * N; ++e) asm volatile("" ::"v"(tp[e]));
        asm volatile("" ::"v"(ylast));
        double qpos = 0.0;
        for (int it = 1; it <= nseg; ++it) {
            const int seg = it - 1;
            char *tile = (seg & 1) ? tile1 : tile0;
            rotate_priority(prio_slot + it);
            if (seg == 9) stamp(19);
            lds_barrier();                                    /* barrier 0 */
            if constexpr (kChain) {
                if (it == nseg) qpos = xch[1];
            }
            double s[N + 1];
            {
                const double *slot = pbuf + lane * kPPitch;
#pragma unroll
                for (int i = 0; i < N; ++i) s[i] = slot[i];
            }
            lds_barrier();                                    /* barrier 1 */
            if (seg == 0) stamp(3);                           /* back starts segment 0 */
            if (seg == 8) stamp(15);
            if (seg == 9) stamp(21);

            const size_t base = (size_t)seg * kPSeg;
            const size_t left = n - base;
            const int seg_len = kWhole ? kPSeg : (left < (size_t)kPSeg ? (int)left : kPSeg);
            const bool seeded = seed_stream && seg == 0;
            const bool fast = (kWhole || seg_len == kPSeg) && !seeded;

            /* s_j = P_j + A^(32 j) carry  (row 0: P = 0, A^0 = I) */
            if (!seeded) {
#pragma unroll
                for (int q = 0; q < N; ++q) {
#pragma unroll
                    for (int i = 0; i < N; ++i) s[i] = __builtin_fma(tp[i * N + q], carry[q], s[i]);
                }
                s[N] = carry[N];
            } else {
                s[N] = 0.0;
            }

            char *row = tile + lane * (PT * 8);
            double y = ylast;
            if (fast) {
                /* the row in two bursts of LDS reads (a whole row at once would push the lane's propagator
                 * matrix out of the register file), then arithmetic only */
                constexpr int kBurst = PT / 4;
#pragma unroll
                for (int h = 0; h < PT / 2; h += kBurst) {
                    d2 xr[kBurst];
#pragma unroll
                    for (int q = 0; q < kBurst; ++q)
                        xr[q] = *reinterpret_cast<const d2 *>(row + (((h + q) << 4) ^ rx));
#pragma unroll
                    for (int q = 0; q < kBurst; ++q) {
                        d2 xv = xr[q];
                        xv.x = step(xv.x, s, b, a);
                        xv.y = y = step(xv.y, s, b, a);
                        *reinterpret_cast<d2 *>(row + (((h + q) << 4) ^ rx)) = xv;
                    }
                }
            } else {
                int cnt = seg_len - lane * PT;
                cnt = cnt < 0 ? 0 : (cnt > PT ? PT : cnt);
                int k = 0;
                if (seeded && lane == 0) {
                    double *p0 = reinterpret_cast<double *>(row + (0 ^ rx));
                    y = seed(*p0, s, b, a);
                    *p0 = y;
                    k = 1;
                }
                for (; k < cnt; ++k) {
                    double *pk = reinterpret_cast<double *>(row + (((k >> 1) << 4) ^ rx) + (k & 1) * 8);
                    y = step(*pk, s, b, a);
                    *pk = y;
                }
            }
            const int jlast = (seg_len - 1) / PT;
#pragma unroll
            for (int i = 0; i <= N; ++i) carry[i] = __shfl(s[i], jlast, kWave);
            ylast = __shfl(y, jlast, kWave);

            /* store the segment, coalesced (this wave wrote the whole tile: in-order LDS) */
            if (seg == 0) stamp(10);
            if (seg == 8) stamp(18);
            wave_sync();
            if (kWhole || (seg_len == kPSeg && io.aligned16)) {
                double *drow = dst + (size_t)seg * seg_out;
#pragma unroll 8
                for (int i = 0; i < kPPieces; ++i)
                    stream_store(reinterpret_cast<d2 *>(drow + 2 * (i * kWave + lane)),
                                 *reinterpret_cast<const d2 *>(tile + i * (4 * PT * 8) + st_addr[i & 3]));
            } else {
#pragma unroll 4
                for (int i = 0; i < kPPieces; ++i) {
                    const int e = 2 * (i * kWave + lane);
                    const d2 ov = *reinterpret_cast<const d2 *>(tile + i * (4 * PT * 8) + st_addr[i & 3]);
                    if (e < seg_len) dst[base + e] = ov.x;
                    if (e + 1 < seg_len) dst[base + e + 1] = ov.y;
                }
            }
        }

        if constexpr (kChain) {
            /* The state after this block goes into the channel's carry record (lane i: granule i) and the channel
             * joins the ready queue at the position the front wave claimed.  No wait and no order between the
             * two: the record's granules validate themselves, a reader that finds the queue entry first polls
             * them. */
            double val = ylast;
#pragma unroll
            for (int i = 0; i <= N; ++i) val = lane == i ? carry[i] : val;
            if (lane < N + 2) granule_store(rec + 2 * lane, val, ext.chain.gen);
            if (lane == 0) {
                const unsigned long long pos = (unsigned long long)qpos;
                if (pos < (unsigned long long)g.count)
                    __hip_atomic_store(ext.chain.my_queue + pos,
                                       ((unsigned long long)ext.chain.gen << 32) | (unsigned int)(gi + 1),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (lane == 0) {
            double *st = g.state + (size_t)gi * g.state_ld;
            if constexpr (kSos) {
#pragma unroll
                for (int k = 0; k < N / 2; ++k) {
                    st[3 * k] = carry[2 * k];
                    st[3 * k + 1] = carry[2 * k + 1];
                    st[3 * k + 2] = 0.0;                   /* what stable_init leaves there; never written afterwards */
                }
            } else {
#pragma unroll
                for (int i = 0; i <= N; ++i) st[i] = carry[i];
            }
            g.fval[gi] = ylast;
            g.init[gi] = 0;
            if (dbg) dbg[5] = __builtin_amdgcn_s_memrealtime();                            /* back done */
        }
    }
}

/* workgroups of the pipelined kernel the device holds at once (4 per CU: two 16 KiB tiles each, 2 waves per SIMD) */
int pipe_resident_round()
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    return 4 * cus;
}

template <int N>
static hipError_t launch_pipe_one(const GroupView &g, const BlockIO &io, hipStream_t s);

/* Groups a little wider than one resident round are launched round by round: a launch of 1.25 .. 4 rounds runs its
 * later workgroups against the drain of the first round and loses 1-5 % against the same channels as back-to-back
 * launches of one round each (2048 channels x 64K: 464 -> 438 us, profiles/r2_split_launch.txt); from ~6 rounds on
 * the single launch wins. */
template <int N>
static hipError_t launch_pipe_n(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    const int round = pipe_resident_round();
    if (g.count <= round || g.count > 4 * round) return launch_pipe_one<N>(g, io, s);
    for (int off = 0; off < g.count; off += round) {
        GroupView h = g;
        h.chan = g.chan ? g.chan + off : nullptr;
        h.chan0 = g.chan0 + off;
        h.count = g.count - off < round ? g.count - off : round;
        h.design = g.design + off;
        h.state = g.state + (size_t)off * g.state_ld;
        h.fval = g.fval + off;
        h.init = g.init + off;
        if (g.aux) h.aux = g.aux + (size_t)off * 64;         /* diagnostics: 32 stamps per workgroup */
        const hipError_t e = launch_pipe_one<N>(h, io, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int N>
static hipError_t launch_pipe_one(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    const size_t lds = 2 * kPTile + sizeof(double) * (size_t)kPRows * (N + 2);
    const bool whole = io.aligned16 && io.n % kPSeg == 0;
    PipeExt ext = {};
    ext.seg_in = ext.seg_out = kPSeg;
    if (g.unit_a0 && whole)
        hipLaunchKernelGGL((iir_pipe_kernel<N, true, true, false>), dim3(g.count), dim3(2 * kWave), lds, s, g, io, ext);
    else if (g.unit_a0)
        hipLaunchKernelGGL((iir_pipe_kernel<N, true, false, false>), dim3(g.count), dim3(2 * kWave), lds, s, g, io, ext);
    else if (whole)
        hipLaunchKernelGGL((iir_pipe_kernel<N, false, true, false>), dim3(g.count), dim3(2 * kWave), lds, s, g, io, ext);
    else
        hipLaunchKernelGGL((iir_pipe_kernel<N, false, false, false>), dim3(g.count), dim3(2 * kWave), lds, s, g, io, ext);
    return hipGetLastError();
}

/* whole segments with a segment stride and / or as a link of a chain */
template <int N>
static hipError_t launch_pipe_ext(const GroupView &g, const BlockIO &io, const PipeExt &ext, hipStream_t s)
{
    const size_t lds = 2 * kPTile + sizeof(double) * (size_t)kPRows * (N + 2);
    if (!io.aligned16 || io.n % kPSeg || ext.seg_in % 2 || ext.seg_out % 2) return hipErrorInvalidValue;
    if (ext.chain.records) {
        if (g.count > pipe_resident_round()) return hipErrorInvalidValue;
        const size_t ldc = lds + 16;                        /* + the two exchange words */
        if (g.unit_a0)
            hipLaunchKernelGGL((iir_pipe_kernel<N, true, true, true>), dim3(g.count), dim3(2 * kWave), ldc, s, g, io, ext);
        else
            hipLaunchKernelGGL((iir_pipe_kernel<N, false, true, true>), dim3(g.count), dim3(2 * kWave), ldc, s, g, io, ext);
    } else if (g.unit_a0)
        hipLaunchKernelGGL((iir_pipe_kernel<N, true, true, false>), dim3(g.count), dim3(2 * kWave), lds + 32, s, g, io, ext);
    else
        hipLaunchKernelGGL((iir_pipe_kernel<N, false, true, false>), dim3(g.count), dim3(2 * kWave), lds + 32, s, g, io, ext);
    return hipGetLastError();
}

hipError_t launch_iir_pipe_ext(int order, const GroupView &g, const BlockIO &io, const PipeExt &ext, hipStream_t s)
{
    switch (order) {
        case 1: return launch_pipe_ext<1>(g, io, ext, s);
        case 2: return launch_pipe_ext<2>(g, io, ext, s);
        case 3: return launch_pipe_ext<3>(g, io, ext, s);
        case 4: return launch_pipe_ext<4>(g, io, ext, s);
        case 5: return launch_pipe_ext<5>(g, io, ext, s);
        case 6: return launch_pipe_ext<6>(g, io, ext, s);
        case 7: return launch_pipe_ext<7>(g, io, ext, s);
        case 8: return launch_pipe_ext<8>(g, io, ext, s);
    }
    return hipErrorInvalidValue;
}

/* ---- chain plumbing: gate, unpack ------------------------------------------------------------------------- */

/* Runs on the stream of launch k+1 in front of it: returns once every workgroup of launch k (the other stream)
 * has started, so that no workgroup of k+1 can occupy a slot a workgroup of k still needs.  One lane, bounded. */
__global__ void __launch_bounds__(kWave) chain_gate_kernel(const unsigned long long *started,
                                                            unsigned long long target, unsigned int *error)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > kChainTimeoutTicks) {
            __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

hipError_t launch_chain_gate(const unsigned long long *started, unsigned long long target, unsigned int *error,
                             hipStream_t s)
{
    hipLaunchKernelGGL(chain_gate_kernel, dim3(1), dim3(kWave), 0, s, started, target, error);
    return hipGetLastError();
}

/* carry records -> the group's plain arrays (end of a chain; ordinary stream order, all links complete) */
__global__ void __launch_bounds__(256) chain_unpack_kernel(const double *records, double *state, double *fval,
                                                            int32_t *init, int count, int N, int state_ld)
{
    const int gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= count) return;
    const double *rec = records + (size_t)gi * (2 * (N + 2));
    for (int i = 0; i <= N; ++i) state[(size_t)gi * state_ld + i] = rec[2 * i];
    fval[gi] = rec[2 * (N + 1)];
    init[gi] = 0;
}

hipError_t launch_chain_unpack(const double *records, double *state, double *fval, int32_t *init, int count, int N,
                               int state_ld, hipStream_t s)
{
    hipLaunchKernelGGL(chain_unpack_kernel, dim3((count + 255) / 256), dim3(256), 0, s, records, state, fval, init,
                       count, N, state_ld);
    return hipGetLastError();
}

/* rows [nch][ld] <-> tiles [n / 2048][nch][2048] (harness / caller helper: dsp_tile_rows, dsp_untile_rows) */
__global__ void __launch_bounds__(256) tile_rows_kernel(double *dst, const double *src, size_t ld, size_t nch,
                                                         size_t nseg, int to_tiles)
{
    const size_t c = blockIdx.x, sg = blockIdx.y;
    const double *row = (to_tiles ? src : dst) + c * ld + sg * kPSeg;
    const double *tl = (to_tiles ? dst : src) + (sg * nch + c) * kPSeg;
    for (int k = threadIdx.x; k < kPSeg; k += 256) {
        if (to_tiles) const_cast<double *>(tl)[k] = row[k];
        else const_cast<double *>(row)[k] = tl[k];
    }
}

hipError_t launch_tile_rows(double *dst, const double *src, size_t ld, size_t nch, size_t n, int to_tiles,
                            hipStream_t s)
{
    if (n % kPSeg || nch == 0 || n == 0 || n / kPSeg > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)nch, (unsigned)(n / kPSeg)), dim3(256), 0, s, dst, src, ld,
                       nch, n / kPSeg, to_tiles);
    return hipGetLastError();
}

template <int S>
static hipError_t launch_pipe_sos_n(const GroupView &g, const BlockIO &io, hipStream_t s)
{
    constexpr int N = 2 * S;
    const size_t lds = 2 * kPTile + sizeof(double) * (size_t)kPRows * (N + 2);
    const bool whole = io.aligned16 && io.n % kPSeg == 0;
    const int round = pipe_resident_round();
    PipeExt ext = {};
    ext.seg_in = ext.seg_out = kPSeg;
    /* a little wider than one resident round: round by round, as launch_pipe_n */
    const int stepc = (g.count > round && g.count <= 4 * round) ? round : g.count;
    for (int off = 0; off < g.count; off += stepc) {
        GroupView h = g;
        h.chan = g.chan ? g.chan + off : nullptr;
        h.chan0 = g.chan0 + off;
        h.count = g.count - off < stepc ? g.count - off : stepc;
        h.design = g.design + off;
        h.state = g.state + (size_t)off * g.state_ld;
        h.fval = g.fval + off;
        h.init = g.init + off;
        if (whole)
            hipLaunchKernelGGL((iir_pipe_kernel<N, true, true, false, true>), dim3(h.count), dim3(2 * kWave), lds, s, h, io, ext);
        else
            hipLaunchKernelGGL((iir_pipe_kernel<N, true, false, false, true>), dim3(h.count), dim3(2 * kWave), lds, s, h, io, ext);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_iir_pipe_sos(int sections, const GroupView &g, const BlockIO &io, hipStream_t s)
{
    static_assert(kMaxFusedSections == 4, "one instantiation per section count");
    switch (sections) {
        case 1: return launch_pipe_sos_n<1>(g, io, s);
        case 2: return launch_pipe_sos_n<2>(g, io, s);
        case 3: return launch_pipe_sos_n<3>(g, io, s);
        case 4: return launch_pipe_sos_n<4>(g, io, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_iir_pipe(int order, const GroupView &g, const BlockIO &io, hipStream_t s)
{
    switch (order) {
        case 1: return launch_pipe_n<1>(g, io, s);
        case 2: return launch_pipe_n<2>(g, io, s);
        case 3: return launch_pipe_n<3>(g, io, s);
        case 4: return launch_pipe_n<4>(g, io, s);
        case 5: return launch_pipe_n<5>(g, io, s);
        case 6: return launch_pipe_n<6>(g, io, s);
        case 7: return launch_pipe_n<7>(g, io, s);
        case 8: return launch_pipe_n<8>(g, io, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dsp
