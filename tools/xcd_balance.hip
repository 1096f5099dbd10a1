/*
 * xcd_balance.hip -- does a row-per-workgroup stream run at the same rate on every XCD?
 *
 *   hipcc --offload-arch=gfx950 -O3 -o tools/_build/xcd_balance tools/xcd_balance.hip
 *
 * 1024 workgroups (one resident round, 4 per CU) each copy one row of 64K doubles, as the pipelined IIR kernel
 * does, and stamp their start and end (s_memrealtime, 100 MHz) and the XCC they ran on.  Variants: which row a
 * workgroup takes (blockIdx, blockIdx ^ 1, a shuffle), the row pitch, threads per workgroup.  Printed: mean life by
 * XCC, by row parity, and the spread -- to tell an address effect (follows the row) from a placement effect
 * (follows the XCD).
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

template <int U>
__global__ void __launch_bounds__(256) copy_rows_stamped(const d2 *__restrict__ in, d2 *__restrict__ out, size_t row16,
                                                          size_t pitch16, int map, unsigned long long *stamps)
{
    const size_t T = blockDim.x;
    unsigned b = blockIdx.x;
    unsigned r = b;
    if (map == 1) r = b ^ 1;
    else if (map == 2) r = ((b & 7) * 128) + (b >> 3);       /* rows of one XCD (b % 8) are neighbours */
    else if (map == 3) r = (b * 421u) & 1023u;              /* odd multiplier: a permutation of 0..1023 */
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    const d2 *src = in + r * pitch16;
    d2 *dst = out + r * pitch16;
    d2 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = __builtin_nontemporal_load(src + threadIdx.x + u * T);
    for (size_t off = 0; off < row16; off += T * U) {
        const bool more = off + T * U < row16;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = __builtin_nontemporal_load(src + off + T * U + threadIdx.x + u * T);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(cur[u], dst + off + threadIdx.x + u * T);
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[4 * b + 0] = t0;
        stamps[4 * b + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[4 * b + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;     /* XCC_ID */
        stamps[4 * b + 3] = r;
    }
}

/* Row copy with tail stealing: the balance experiment for the filter kernel's one-round launch.  One wave per
 * workgroup; workgroup b owns row b and claims its 16 KiB segments front to back through claims[b] (low 16 bits:
 * segments claimed by the owner, high 16: segments stolen from the END by other workgroups).  A workgroup whose row
 * is exhausted probes 64 other rows, picks the one with the most unclaimed segments (>= min_rem) and takes its last
 * unclaimed segment; with runup != 0 it also reads the segment in front of it (what the filter would need to
 * rebuild the carried state).  No waiting anywhere: every loop ends when a claim fails or no victim is found. */
template <int U>
__global__ void __launch_bounds__(64) copy_rows_steal(const d2 *__restrict__ in, d2 *__restrict__ out, unsigned rows,
                                                       unsigned nseg, size_t pitch16, unsigned *claims, int steal,
                                                       int runup, unsigned min_rem, unsigned long long *stamps,
                                                       double *sink)
{
    const unsigned b = blockIdx.x, lane = threadIdx.x;
    const size_t seg16 = (size_t)64 * U;
    unsigned long long t0 = 0;
    if (lane == 0) t0 = __builtin_amdgcn_s_memrealtime();
    auto claim_own = [&]() -> unsigned {                    /* returns the segment index or ~0u */
        unsigned old = 0;
        if (lane == 0) old = atomicAdd(&claims[b], 1u);
        old = __builtin_amdgcn_readfirstlane(old);
        const unsigned head = old & 0xffffu, stolen = old >> 16;
        return head + stolen < nseg ? head : ~0u;
    };
    auto load_seg = [&](unsigned row, unsigned seg, d2(&v)[U]) {
        const d2 *src = in + row * pitch16 + seg * seg16;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + lane + u * 64);
    };
    auto store_seg = [&](unsigned row, unsigned seg, const d2(&v)[U]) {
        d2 *dst = out + row * pitch16 + seg * seg16;
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], dst + lane + u * 64);
    };
    d2 cur[U], nxt[U];
    unsigned scur = claim_own();
    unsigned snxt = claim_own();
    if (scur != ~0u) load_seg(b, scur, cur);
    unsigned own_done = 0, stolen_done = 0;
    while (scur != ~0u) {
        if (snxt != ~0u) load_seg(b, snxt, nxt);
        const unsigned snn = snxt != ~0u ? claim_own() : ~0u;
        store_seg(b, scur, cur);
        ++own_done;
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        scur = snxt;
        snxt = snn;
    }
    double acc = 0.0;
    for (int tries = 0; steal && tries < 64; ++tries) {
        const unsigned cand = (b + 1 + 17 * lane) % rows;
        const unsigned w = __hip_atomic_load(&claims[cand], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned head = w & 0xffffu, st = w >> 16;
        unsigned rem = head + st < nseg ? nseg - head - st : 0;
        if (head == 0) rem = 0;                               /* owner not started */
        unsigned best = rem, who = lane;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const unsigned ob = __shfl_xor(best, d, 64), ow = __shfl_xor(who, d, 64);
            if (ob > best || (ob == best && ow < who)) { best = ob; who = ow; }
        }
        if (best < min_rem) break;
        const unsigned victim = (b + 1 + 17 * who) % rows;
        unsigned old = 0;
        if (lane == 0) old = atomicAdd(&claims[victim], 1u << 16);
        old = __builtin_amdgcn_readfirstlane(old);
        const unsigned h2 = old & 0xffffu, s2 = old >> 16;
        if (h2 + s2 >= nseg) continue;                        /* lost the race: probe again */
        const unsigned seg = nseg - 1 - s2;
        if (runup && seg > 0) {
            d2 r[U];
            load_seg(victim, seg - 1, r);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += r[u].x;
        }
        load_seg(victim, seg, cur);
        store_seg(victim, seg, cur);
        ++stolen_done;
    }
    if (acc == 1.2345e300) sink[0] = acc;
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) {
        stamps[4 * b + 0] = t0;
        stamps[4 * b + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[4 * b + 2] = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf) | (stolen_done << 8) | (own_done << 20);
        stamps[4 * b + 3] = b;
    }
}

int main()
{
    const size_t rows = 1024, L = 65536;
    const size_t maxpitch = ((size_t)1 << 20) + 4096;
    d2 *in, *out;
    CHECK(hipMalloc(&in, rows * maxpitch * 8));
    CHECK(hipMalloc(&out, rows * maxpitch * 8));
    CHECK(hipMemset(in, 0x11, rows * maxpitch * 8));
    CHECK(hipMemset(out, 0, rows * maxpitch * 8));
    unsigned long long *d_st;
    CHECK(hipMalloc(&d_st, rows * 4 * 8));
    std::vector<unsigned long long> st(rows * 4);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    struct Case { const char *name; size_t pitch; int map; int threads; };
    const Case cases[] = {
        {"pitch 2^20+128, row = blockIdx", ((size_t)1 << 20) + 128, 0, 128},
        {"pitch 2^20+128, row = blockIdx ^ 1", ((size_t)1 << 20) + 128, 1, 128},
        {"pitch 2^20+128, rows of an XCD adjacent", ((size_t)1 << 20) + 128, 2, 128},
        {"pitch 2^20+128, rows permuted", ((size_t)1 << 20) + 128, 3, 128},
        {"pitch 2^20+128, row = blockIdx, 256 threads", ((size_t)1 << 20) + 128, 0, 256},
        {"pitch 2^20+2176 (17 KiB past 8 MiB)", ((size_t)1 << 20) + 2176, 0, 128},
        {"pitch 2^20+640 (5 KiB past)", ((size_t)1 << 20) + 640, 0, 128},
        {"pitch 2^16 (dense block)", (size_t)1 << 16, 0, 128},
        {"pitch 2^16+128", ((size_t)1 << 16) + 128, 0, 128},
        {"pitch 2^16+640", ((size_t)1 << 16) + 640, 0, 128},
    };
    for (const Case &c : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((copy_rows_stamped<8>), dim3(rows), dim3(c.threads), 0, 0, in, out, L / 2, c.pitch / 2, c.map, d_st);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        CHECK(hipMemcpy(st.data(), d_st, rows * 4 * 8, hipMemcpyDeviceToHost));
        double by_xcc[8] = {0}, by_par[2] = {0}, by_xpar[2] = {0};
        int n_xcc[8] = {0}, n_par[2] = {0}, n_xpar[2] = {0};
        unsigned long long first = ~0ull, last = 0;
        double mean = 0, mx = 0, mn = 1e30;
        for (size_t b = 0; b < rows; ++b) {
            const double life = (double)(st[4 * b + 1] - st[4 * b]) / 100.0;       /* us */
            const int x = (int)st[4 * b + 2] & 7;
            const int r = (int)st[4 * b + 3];
            by_xcc[x] += life; n_xcc[x]++;
            by_par[r & 1] += life; n_par[r & 1]++;
            by_xpar[x & 1] += life; n_xpar[x & 1]++;
            first = std::min(first, st[4 * b]);
            last = std::max(last, st[4 * b + 1]);
            mean += life; mx = std::max(mx, life); mn = std::min(mn, life);
        }
        mean /= rows;
        printf("%-48s event %.1f us (%.0f GB/s)  span %.1f  life min %.1f mean %.1f max %.1f\n", c.name, best * 1e3,
               rows * L * 16 / (best * 1e-3) / 1e9, (double)(last - first) / 100.0, mn, mean, mx);
        printf("    by XCC:");
        for (int x = 0; x < 8; ++x) printf(" %d:%.1f(%d)", x, n_xcc[x] ? by_xcc[x] / n_xcc[x] : 0.0, n_xcc[x]);
        printf("\n    by row parity: even %.1f odd %.1f ; by XCC parity: even %.1f odd %.1f\n", by_par[0] / n_par[0],
               by_par[1] / n_par[1], by_xpar[0] / std::max(1, n_xpar[0]), by_xpar[1] / std::max(1, n_xpar[1]));
        fflush(stdout);
    }
    /* ---- tail stealing ---- */
    {
        const size_t pitch = ((size_t)1 << 20) + 128;
        const unsigned nseg = 32;                               /* 64K doubles = 32 segments of 16 KiB */
        unsigned *d_claims;
        double *d_sink;
        CHECK(hipMalloc(&d_claims, rows * 4));
        CHECK(hipMalloc(&d_sink, 64));
        struct SC { const char *name; int steal, runup; unsigned min_rem; };
        const SC scs[] = {{"one wave per row, claims only (no stealing)", 0, 0, 2},
                          {"tail stealing, min_rem 2", 1, 0, 2},
                          {"tail stealing, min_rem 3", 1, 0, 3},
                          {"tail stealing + run-up read, min_rem 2", 1, 1, 2},
                          {"tail stealing + run-up read, min_rem 3", 1, 1, 3}};
        for (const SC &c : scs) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipMemset(d_claims, 0, rows * 4));
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((copy_rows_steal<16>), dim3(rows), dim3(64), 0, 0, in, out, (unsigned)rows, nseg,
                                   pitch / 2, d_claims, c.steal, c.runup, c.min_rem, d_st, d_sink);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            CHECK(hipMemcpy(st.data(), d_st, rows * 4 * 8, hipMemcpyDeviceToHost));
            double mean = 0, mx = 0, mn = 1e30;
            unsigned long total_stolen = 0, total_own = 0, max_stolen = 0;
            for (size_t b = 0; b < rows; ++b) {
                const double life = (double)(st[4 * b + 1] - st[4 * b]) / 100.0;
                mean += life; mx = std::max(mx, life); mn = std::min(mn, life);
                const unsigned sd = (st[4 * b + 2] >> 8) & 0xfff, od = (st[4 * b + 2] >> 20) & 0xfff;
                total_stolen += sd; total_own += od; max_stolen = std::max<unsigned long>(max_stolen, sd);
            }
            printf("%-48s event %.1f us (%.0f GB/s)  life min %.1f mean %.1f max %.1f  segments own %lu stolen %lu (max %lu per WG)\n",
                   c.name, best * 1e3, rows * L * 16 / (best * 1e-3) / 1e9, mn, mean / rows, mx, total_own, total_stolen, max_stolen);
            fflush(stdout);
        }
    }
    return 0;
}
