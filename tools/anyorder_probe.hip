/*
 * anyorder_probe.hip -- can two consecutive launches of a stream overlap on gfx950, and in which order are
 * their workgroups dispatched?  (Round 3: the chained block entry point needs block k+1's workgroups to take
 * the slots block k's drain frees.)
 *
 *   hipcc --offload-arch=gfx950 -O3 -o tools/_build/anyorder_probe tools/anyorder_probe.hip
 *
 * Kernel `occupy`: 128 threads, 37 KiB of LDS (4 workgroups per CU like iir_pipe_kernel); every workgroup
 * stamps its start (s_memrealtime, 100 MHz), spins for dur(blockIdx) and stamps its end.  All spins are bounded.
 *
 *   test 1  A (1024 WGs) then B (1024 WGs) on ONE stream, B launched with hipExtAnyOrderLaunch
 *   test 2  the same with A = 4096 WGs: does any B workgroup start before the last A workgroup has started?
 *   test 3  A on stream 1, B on stream 2 (plain launches)
 *   test 4  chain: 16 launches, workgroup i of launch k+1 waits (bounded) until i+1 workgroups of launch k have
 *           finished (a counter in HBM), single stream AnyOrder vs plain stream order vs two streams: wall time
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

struct Stamp { unsigned long long start, end; unsigned int waited, xcc; };

__device__ __forceinline__ unsigned int dur_ticks(unsigned int i, unsigned int base, unsigned int spread)
{
    unsigned int h = i * 2654435761u;
    h ^= h >> 15;
    return base + (spread ? h % spread : 0);
}

/* prev_done: counter of finished workgroups of the previous launch (nullptr: no dependency);
 * my_done: this launch's counter.  Ticket = blockIdx.x. */
__global__ void __launch_bounds__(128) occupy(Stamp *st, unsigned int base, unsigned int spread,
                                              const unsigned int *prev_done, unsigned int *my_done,
                                              unsigned int *timeout_flag)
{
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned int waited = 0;
    if (prev_done && threadIdx.x == 0) {
        /* bounded: 20 ms at 100 MHz */
        while (__hip_atomic_load(prev_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < blockIdx.x + 1) {
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {
                *timeout_flag = 1;
                break;
            }
        }
        waited = (unsigned int)(__builtin_amdgcn_s_memrealtime() - t0);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const unsigned int d = dur_ticks(blockIdx.x, base, spread);
    if (threadIdx.x == 0) lds[0] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t1 < d) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0) {
        st[blockIdx.x].start = t0;
        st[blockIdx.x].end = __builtin_amdgcn_s_memrealtime();
        st[blockIdx.x].waited = waited;
        st[blockIdx.x].xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        if (my_done) __hip_atomic_fetch_add(my_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static const size_t kLds = 37 * 1024;

static void launch(hipStream_t s, bool any_order, int grid, Stamp *st, unsigned base, unsigned spread,
                   const unsigned *prev, unsigned *mine, unsigned *flag)
{
    if (any_order)
        hipExtLaunchKernelGGL(occupy, dim3(grid), dim3(128), kLds, s, nullptr, nullptr, hipExtAnyOrderLaunch, st, base,
                              spread, prev, mine, flag);
    else
        hipLaunchKernelGGL(occupy, dim3(grid), dim3(128), kLds, s, st, base, spread, prev, mine, flag);
    CHECK(hipGetLastError());
}

static void report(const char *name, const std::vector<Stamp> &a, const std::vector<Stamp> &b)
{
    unsigned long long a_s0 = ~0ull, a_s1 = 0, a_e0 = ~0ull, a_e1 = 0, b_s0 = ~0ull, b_s1 = 0, b_e1 = 0;
    for (const Stamp &s : a) {
        a_s0 = std::min(a_s0, s.start); a_s1 = std::max(a_s1, s.start);
        a_e0 = std::min(a_e0, s.end); a_e1 = std::max(a_e1, s.end);
    }
    size_t before_a_last_start = 0, before_a_end = 0;
    for (const Stamp &s : b) {
        b_s0 = std::min(b_s0, s.start); b_s1 = std::max(b_s1, s.start); b_e1 = std::max(b_e1, s.end);
        if (s.start < a_s1) ++before_a_last_start;
        if (s.start < a_e1) ++before_a_end;
    }
    auto us = [&](unsigned long long t) { return (double)(long long)(t - a_s0) / 100.0; };
    printf("%s\n  A: starts %.1f..%.1f us, ends %.1f..%.1f us\n  B: starts %.1f..%.1f us, last end %.1f us\n"
           "  B workgroups started before A's last END: %zu of %zu; before A's last START: %zu\n",
           name, us(a_s0), us(a_s1), us(a_e0), us(a_e1), us(b_s0), us(b_s1), us(b_e1), before_a_end, b.size(),
           before_a_last_start);
    fflush(stdout);
}

int main()
{
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CHECK(hipFuncSetAttribute((const void *)occupy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    const int maxg = 4096;
    Stamp *da, *db;
    CHECK(hipMalloc(&da, maxg * sizeof(Stamp)));
    CHECK(hipMalloc(&db, maxg * sizeof(Stamp)));
    unsigned *cnt;
    CHECK(hipMalloc(&cnt, 64 * sizeof(unsigned)));
    std::vector<Stamp> ha(maxg), hb(maxg);

    auto fetch = [&](int ga, int gb) {
        CHECK(hipDeviceSynchronize());
        ha.resize(ga); hb.resize(gb);
        CHECK(hipMemcpy(ha.data(), da, ga * sizeof(Stamp), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hb.data(), db, gb * sizeof(Stamp), hipMemcpyDeviceToHost));
    };
    /* warm-up */
    launch(s1, false, 1024, da, 1000, 0, nullptr, nullptr, cnt);
    CHECK(hipDeviceSynchronize());

    /* durations: 200 us +- 30 (20000 ticks + hash % 6000) */
    launch(s1, false, 1024, da, 17000, 6000, nullptr, nullptr, cnt);
    launch(s1, false, 1024, db, 17000, 6000, nullptr, nullptr, cnt);
    fetch(1024, 1024);
    report("test 0: A then B, one stream, plain launches (baseline: B must follow A)", ha, hb);

    launch(s1, false, 1024, da, 17000, 6000, nullptr, nullptr, cnt);
    launch(s1, true, 1024, db, 17000, 6000, nullptr, nullptr, cnt);
    fetch(1024, 1024);
    report("test 1: A then B, one stream, B with hipExtAnyOrderLaunch", ha, hb);

    launch(s1, true, 4096, da, 4000, 2000, nullptr, nullptr, cnt);
    launch(s1, true, 1024, db, 4000, 2000, nullptr, nullptr, cnt);
    fetch(4096, 1024);
    report("test 2: A = 4096 workgroups (4 rounds), B AnyOrder, one stream", ha, hb);

    launch(s1, false, 1024, da, 17000, 6000, nullptr, nullptr, cnt);
    launch(s2, false, 1024, db, 17000, 6000, nullptr, nullptr, cnt);
    fetch(1024, 1024);
    report("test 3: A on stream 1, B on stream 2, plain launches", ha, hb);

    launch(s1, false, 4096, da, 4000, 2000, nullptr, nullptr, cnt);
    launch(s2, false, 1024, db, 4000, 2000, nullptr, nullptr, cnt);
    fetch(4096, 1024);
    report("test 3b: A = 4096 workgroups on stream 1, B on stream 2", ha, hb);

    /* test 4: chains of 16 launches x 1024 workgroups x (200 +- 30 us) */
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipMemset(cnt, 0, 64 * sizeof(unsigned)));
            CHECK(hipDeviceSynchronize());
            const int K = 16;
            CHECK(hipEventRecord(e0, s1));
            if (variant == 2) CHECK(hipStreamWaitEvent(s2, e0, 0));
            for (int k = 0; k < K; ++k) {
                const unsigned *prev = k ? cnt + 1 + (k - 1) : nullptr;
                unsigned *mine = cnt + 1 + k;
                if (variant == 0) launch(s1, false, 1024, da, 17000, 6000, nullptr, mine, cnt);
                else if (variant == 1) launch(s1, k > 0, 1024, da, 17000, 6000, prev, mine, cnt);
                else launch((k & 1) ? s2 : s1, false, 1024, da, 17000, 6000, prev, mine, cnt);
            }
            if (variant == 2) {
                hipEvent_t j;
                CHECK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
                CHECK(hipEventRecord(j, s2));
                CHECK(hipStreamWaitEvent(s1, j, 0));
            }
            CHECK(hipEventRecord(e1, s1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned flag;
            CHECK(hipMemcpy(&flag, cnt, 4, hipMemcpyDeviceToHost));
            printf("test 4 %-44s: %d launches %.1f us = %.1f us per launch (timeout flag %u)\n",
                   variant == 0 ? "plain stream order (no dependency words)"
                   : variant == 1 ? "one stream, AnyOrder, ticket chain" : "two streams alternating, ticket chain",
                   K, ms * 1e3, ms * 1e3 / K, flag);
            fflush(stdout);
        }
    }
    return 0;
}
