/*
 * copy_ceiling.hip -- what a 1:1 read/write stream can reach on this MI355X, by access pattern.
 *
 *   hipcc --offload-arch=gfx950 -O3 -o tools/_build/copy_ceiling tools/copy_ceiling.hip
 *   gpurun -- ./tools/_build/copy_ceiling > gpurun_out/copy_ceiling.txt
 *
 * Patterns (all 16 B per lane per access, float64 payload, bytes counted = read + written):
 *   linear     persistent grid-stride copy, no row structure: workgroup w, step k touches the 16-byte
 *              pieces [(k G + w) T U, ...) -- the whole chip sweeps one contiguous window
 *   chunk      persistent, each workgroup takes contiguous chunks of CH bytes round-robin
 *   rows       one workgroup per row (the IIR kernels' pattern): R rows of L doubles at pitch P,
 *              each row walked front to back in 16 KiB segments, one segment of loads in flight
 *   read / write   one-sided sweeps (linear pattern)
 * The guide (MI355X_MICROARCH.md) records 6.29 TB/s for a float4 copy; round 1 measured 5.1 TB/s for the
 * row pattern.  This tool settles which of the two a kernel with the filter's row structure can be held to.
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

template <bool NT>
__device__ __forceinline__ d2 ld(const d2 *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st(d2 *p, d2 v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

/* n16 = number of 16-byte pieces; requires n16 % (blockDim * U) == 0 */
template <bool NT, int U>
__global__ void __launch_bounds__(1024) copy_linear(const d2 *__restrict__ in, d2 *__restrict__ out, size_t n16)
{
    const size_t T = blockDim.x;
    const size_t stride = (size_t)gridDim.x * T * U;
    for (size_t base = (size_t)blockIdx.x * T * U + threadIdx.x; base < n16; base += stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + base + u * T);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(out + base + u * T, v[u]);
    }
}

template <bool NT, int U>
__global__ void __launch_bounds__(1024) read_linear(const d2 *__restrict__ in, double *__restrict__ sink, size_t n16)
{
    const size_t T = blockDim.x;
    const size_t stride = (size_t)gridDim.x * T * U;
    double acc = 0.0;
    for (size_t base = (size_t)blockIdx.x * T * U + threadIdx.x; base < n16; base += stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + base + u * T);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

template <bool NT, int U>
__global__ void __launch_bounds__(1024) write_linear(d2 *__restrict__ out, size_t n16)
{
    const size_t T = blockDim.x;
    const size_t stride = (size_t)gridDim.x * T * U;
    const d2 v = {1.0, 2.0};
    for (size_t base = (size_t)blockIdx.x * T * U + threadIdx.x; base < n16; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(out + base + u * T, v);
    }
}

/* chunk16 = pieces per chunk; chunks dealt round-robin to workgroups; U pieces per thread in flight */
template <bool NT, int U>
__global__ void __launch_bounds__(1024) copy_chunk(const d2 *__restrict__ in, d2 *__restrict__ out, size_t n16,
                                                    size_t chunk16)
{
    const size_t T = blockDim.x;
    const size_t nchunk = n16 / chunk16;
    for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const size_t c0 = c * chunk16;
        for (size_t off = threadIdx.x; off < chunk16; off += T * U) {
            d2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + c0 + off + u * T);
#pragma unroll
            for (int u = 0; u < U; ++u) st<NT>(out + c0 + off + u * T, v[u]);
        }
    }
}

/* rows: workgroup r (grid-strided over rows) walks its row in segments of T*U pieces, one segment prefetched */
template <bool NT, int U>
__global__ void __launch_bounds__(1024) copy_rows(const d2 *__restrict__ in, d2 *__restrict__ out, size_t rows,
                                                   size_t row16, size_t pitch16)
{
    const size_t T = blockDim.x;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const d2 *src = in + r * pitch16;
        d2 *dst = out + r * pitch16;
        d2 cur[U], nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = ld<NT>(src + threadIdx.x + u * T);
        for (size_t off = 0; off < row16; off += T * U) {
            const bool more = off + T * U < row16;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = ld<NT>(src + off + T * U + threadIdx.x + u * T);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) st<NT>(dst + off + threadIdx.x + u * T, cur[u]);
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            }
        }
    }
}

/* tiled block layout [segment][row][T*U pieces]: workgroup r walks segment s of its row at (s * rows + r) * seg16 --
 * the workgroups that run together then sweep one contiguous window of rows * 16 KiB per step */
template <bool NT, int U>
__global__ void __launch_bounds__(1024) copy_rows_tiled(const d2 *__restrict__ in, d2 *__restrict__ out, size_t rows,
                                                         size_t row16)
{
    const size_t T = blockDim.x;
    const size_t seg16 = T * U;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        d2 cur[U], nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = ld<NT>(in + r * seg16 + threadIdx.x + u * T);
        const size_t nseg = row16 / seg16;
        for (size_t s = 0; s < nseg; ++s) {
            const bool more = s + 1 < nseg;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = ld<NT>(in + ((s + 1) * rows + r) * seg16 + threadIdx.x + u * T);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) st<NT>(out + (s * rows + r) * seg16 + threadIdx.x + u * T, cur[u]);
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            }
        }
    }
}

/* the tiled sweep with the workgroups out of step: workgroup r starts at segment (hash(r) % spread) and wraps
 * (what chained launches do to the lock-step of a launch: round 3) */
template <bool NT, int U>
__global__ void __launch_bounds__(1024) copy_rows_tiled_desync(const d2 *__restrict__ in, d2 *__restrict__ out, size_t rows,
                                                                size_t row16, int spread, int tiled)
{
    const size_t T = blockDim.x;
    const size_t seg16 = T * U;
    const size_t nseg = row16 / seg16;
    for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
        unsigned h = (unsigned)r * 2654435761u;
        h ^= h >> 15;
        const size_t s0 = spread > 1 ? h % (unsigned)spread : 0;
        auto at = [&](size_t s) { return tiled ? (s * rows + r) * seg16 : r * row16 + s * seg16; };
        d2 cur[U], nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = ld<NT>(in + at(s0 % nseg) + threadIdx.x + u * T);
        for (size_t k = 0; k < nseg; ++k) {
            const size_t s = (s0 + k) % nseg, sn = (s0 + k + 1) % nseg;
            const bool more = k + 1 < nseg;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) nxt[u] = ld<NT>(in + at(sn) + threadIdx.x + u * T);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) st<NT>(out + at(s) + threadIdx.x + u * T, cur[u]);
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            }
        }
    }
}

/* rows split in time: piece p of P per row is its own workgroup (no dependency: this is a copy);
 * workgroup id -> (row, piece) with piece-major order, so that concurrently running workgroups
 * touch P windows of the buffer instead of `rows` */
template <bool NT, int U>
__global__ void __launch_bounds__(1024) copy_rows_pieces(const d2 *__restrict__ in, d2 *__restrict__ out, size_t rows,
                                                          size_t row16, size_t pitch16, int pieces)
{
    const size_t T = blockDim.x;
    const size_t piece16 = row16 / pieces;
    for (size_t w = blockIdx.x; w < rows * (size_t)pieces; w += gridDim.x) {
        const size_t r = w % rows, p = w / rows;
        const d2 *src = in + r * pitch16 + p * piece16;
        d2 *dst = out + r * pitch16 + p * piece16;
        for (size_t off = threadIdx.x; off < piece16; off += T * U) {
            d2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld<NT>(src + off + u * T);
#pragma unroll
            for (int u = 0; u < U; ++u) st<NT>(dst + off + u * T, v[u]);
        }
    }
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
    template <typename F>
    double run(F f, int reps = 9)
    {
        std::vector<float> ms;
        for (int r = 0; r < reps + 2; ++r) {
            CHECK(hipEventRecord(a, 0));
            f();
            CHECK(hipEventRecord(b, 0));
            CHECK(hipEventSynchronize(b));
            float t;
            CHECK(hipEventElapsedTime(&t, a, b));
            if (r >= 2) ms.push_back(t);
        }
        CHECK(hipGetLastError());
        std::sort(ms.begin(), ms.end());
        return ms[ms.size() / 2];
    }
};

int main(int argc, char **argv)
{
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const bool tiled_only = argc > 1 && !strcmp(argv[1], "tiled");
    Timer tm;
    if (tiled_only) {
        /* round 3: dense rows vs the tiled block layout vs a linear sweep, the same 512 MiB each way, several
         * passes in one process (the pool's fast/slow state shows as a shift of all three) */
        const size_t rows = 1024, L = 65536, row16 = L / 2, bytes = rows * L * 8;
        d2 *in, *out;
        CHECK(hipMalloc(&in, bytes));
        CHECK(hipMalloc(&out, bytes));
        CHECK(hipMemset(in, 0x11, bytes));
        CHECK(hipMemset(out, 0, bytes));
        CHECK(hipDeviceSynchronize());
        size_t fr, tot;
        CHECK(hipMemGetInfo(&fr, &tot));
        printf("in %p out %p free %zu MiB of %zu MiB\n", (void *)in, (void *)out, fr >> 20, tot >> 20);
        auto gbs = [](size_t b, double ms) { return b / ms / 1e6; };
        for (int pass = 0; pass < 4; ++pass) {
            double t;
            t = tm.run([&] { hipLaunchKernelGGL((copy_rows<true, 4>), dim3(rows), dim3(256), 0, 0, in, out, rows, row16, row16); });
            const double a = gbs(2 * bytes, t);
            t = tm.run([&] { hipLaunchKernelGGL((copy_rows_tiled<true, 4>), dim3(rows), dim3(256), 0, 0, in, out, rows, row16); });
            const double b = gbs(2 * bytes, t);
            t = tm.run([&] { hipLaunchKernelGGL((copy_rows<true, 8>), dim3(rows), dim3(128), 0, 0, in, out, rows, row16, row16); });
            const double a2 = gbs(2 * bytes, t);
            t = tm.run([&] { hipLaunchKernelGGL((copy_rows_tiled<true, 8>), dim3(rows), dim3(128), 0, 0, in, out, rows, row16); });
            const double b2 = gbs(2 * bytes, t);
            t = tm.run([&] { hipLaunchKernelGGL((copy_linear<true, 4>), dim3(2048), dim3(256), 0, 0, in, out, rows * L / 2); });
            const double c = gbs(2 * bytes, t);
            t = tm.run([&] { hipLaunchKernelGGL((copy_linear<true, 4>), dim3(1024), dim3(256), 0, 0, in, out, rows * L / 2); });
            const double c2 = gbs(2 * bytes, t);
            printf("pass %d: dense rows 256xU4 %6.0f | tiled 256xU4 %6.0f | dense rows 128xU8 %6.0f | tiled 128xU8 %6.0f | "
                   "linear 2048x256 %6.0f | linear 1024x256 %6.0f GB/s\n", pass, a, b, a2, b2, c, c2);
            fflush(stdout);
        }
        for (int spread : {1, 2, 4, 8, 32}) {
            double t = tm.run([&] { hipLaunchKernelGGL((copy_rows_tiled_desync<true, 4>), dim3(rows), dim3(256), 0, 0, in, out, rows, row16, spread, 1); });
            const double a = gbs(2 * bytes, t);
            t = tm.run([&] { hipLaunchKernelGGL((copy_rows_tiled_desync<true, 4>), dim3(rows), dim3(256), 0, 0, in, out, rows, row16, spread, 0); });
            printf("workgroups out of step by up to %2d segments of 32: tiled %6.0f | dense rows %6.0f GB/s\n", spread, a, gbs(2 * bytes, t));
            fflush(stdout);
        }
        return 0;
    }
    const size_t maxbytes = (size_t)9 << 30;           /* 9 GiB each: holds 1024 rows at the bench's 8 MiB + 1 KiB pitch */
    d2 *in, *out;
    CHECK(hipMalloc(&in, maxbytes));
    CHECK(hipMalloc(&out, maxbytes));
    CHECK(hipMemset(in, 0x11, maxbytes));
    CHECK(hipMemset(out, 0, maxbytes));
    double *sink;
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipDeviceSynchronize());

    auto gbs = [](size_t bytes, double ms) { return bytes / ms / 1e6; };

    printf("== linear grid-stride copy (bytes = read + written) ==\n");
    const size_t sizes[] = {(size_t)512 << 20, (size_t)2 << 30, (size_t)8 << 30};
    for (size_t sz : sizes) {
        const size_t n16 = sz / 16;
        for (int wgs_per_cu : {2, 4, 8}) {
            for (int threads : {256, 512}) {
                const int grid = 256 * wgs_per_cu;
                double t;
                t = tm.run([&] { hipLaunchKernelGGL((copy_linear<false, 4>), dim3(grid), dim3(threads), 0, 0, in, out, n16); });
                const double p4 = gbs(2 * sz, t);
                t = tm.run([&] { hipLaunchKernelGGL((copy_linear<true, 4>), dim3(grid), dim3(threads), 0, 0, in, out, n16); });
                const double n4 = gbs(2 * sz, t);
                t = tm.run([&] { hipLaunchKernelGGL((copy_linear<true, 8>), dim3(grid), dim3(threads), 0, 0, in, out, n16); });
                const double n8 = gbs(2 * sz, t);
                t = tm.run([&] { hipLaunchKernelGGL((copy_linear<true, 1>), dim3(grid), dim3(threads), 0, 0, in, out, n16); });
                const double n1 = gbs(2 * sz, t);
                printf("size %5zu MiB each way  grid %4d x %3d : plain U4 %6.0f | nt U4 %6.0f | nt U8 %6.0f | nt U1 %6.0f GB/s\n",
                       sz >> 20, grid, threads, p4, n4, n8, n1);
                fflush(stdout);
                if (quick) break;
            }
        }
    }

    printf("== one-sided sweeps, 2 GiB, grid 2048 x 256 ==\n");
    {
        const size_t sz = (size_t)2 << 30, n16 = sz / 16;
        double t;
        t = tm.run([&] { hipLaunchKernelGGL((read_linear<false, 4>), dim3(2048), dim3(256), 0, 0, in, sink, n16); });
        printf("read  plain U4 %6.0f GB/s\n", gbs(sz, t));
        t = tm.run([&] { hipLaunchKernelGGL((read_linear<true, 4>), dim3(2048), dim3(256), 0, 0, in, sink, n16); });
        printf("read  nt    U4 %6.0f GB/s\n", gbs(sz, t));
        t = tm.run([&] { hipLaunchKernelGGL((read_linear<true, 8>), dim3(2048), dim3(256), 0, 0, in, sink, n16); });
        printf("read  nt    U8 %6.0f GB/s\n", gbs(sz, t));
        t = tm.run([&] { hipLaunchKernelGGL((write_linear<false, 4>), dim3(2048), dim3(256), 0, 0, out, n16); });
        printf("write plain U4 %6.0f GB/s\n", gbs(sz, t));
        t = tm.run([&] { hipLaunchKernelGGL((write_linear<true, 4>), dim3(2048), dim3(256), 0, 0, out, n16); });
        printf("write nt    U4 %6.0f GB/s\n", gbs(sz, t));
        fflush(stdout);
    }

    printf("== chunked persistent copy, 2 GiB each way, grid 1024 x 256, nt, U4 ==\n");
    {
        const size_t sz = (size_t)2 << 30, n16 = sz / 16;
        for (size_t chunk : {(size_t)16 << 10, (size_t)64 << 10, (size_t)512 << 10, (size_t)2 << 20}) {
            double t = tm.run([&] { hipLaunchKernelGGL((copy_chunk<true, 4>), dim3(1024), dim3(256), 0, 0, in, out, n16, chunk / 16); });
            printf("chunk %5zu KiB : %6.0f GB/s\n", chunk >> 10, gbs(2 * sz, t));
        }
        fflush(stdout);
    }

    printf("== row pattern: 1024 rows x 64K doubles (512 KiB per row, 1 GiB moved), one workgroup per row ==\n");
    {
        const size_t rows = 1024, L = 65536, row16 = L / 2;
        struct P { const char *name; size_t pitch; } pitches[] = {
            {"pitch 2^20+128 (bench: resident 1M stream)", ((size_t)1 << 20) + 128},
            {"pitch 2^16+128 (block buffer)", ((size_t)1 << 16) + 128},
            {"pitch 2^16 (dense block buffer)", (size_t)1 << 16},
        };
        for (auto &p : pitches) {
            const size_t pitch16 = p.pitch / 2;
            for (int threads : {128, 256}) {
                double t;
                t = tm.run([&] { hipLaunchKernelGGL((copy_rows<true, 4>), dim3(rows), dim3(threads), 0, 0, in, out, rows, row16, pitch16); });
                const double a = gbs(rows * L * 16, t);
                t = tm.run([&] { hipLaunchKernelGGL((copy_rows<true, 8>), dim3(rows), dim3(threads), 0, 0, in, out, rows, row16, pitch16); });
                const double b = gbs(rows * L * 16, t);
                t = tm.run([&] { hipLaunchKernelGGL((copy_rows<false, 8>), dim3(rows), dim3(threads), 0, 0, in, out, rows, row16, pitch16); });
                const double c = gbs(rows * L * 16, t);
                printf("%-46s threads %3d : nt U4 %6.0f | nt U8 %6.0f | plain U8 %6.0f GB/s\n", p.name, threads, a, b, c);
            }
            for (int pieces : {2, 4, 8, 16}) {
                double t = tm.run([&] { hipLaunchKernelGGL((copy_rows_pieces<true, 4>), dim3(2048), dim3(256), 0, 0, in, out, rows, row16, pitch16, pieces); });
                printf("%-46s %2d pieces per row, piece-major, grid 2048 x 256: %6.0f GB/s\n", p.name, pieces, gbs(rows * L * 16, t));
            }
            fflush(stdout);
        }
        /* same bytes as ONE linear sweep, for the launch-size effect: 512 MiB each way */
        double t = tm.run([&] { hipLaunchKernelGGL((copy_linear<true, 4>), dim3(2048), dim3(256), 0, 0, in, out, rows * L / 2); });
        printf("linear sweep of the same 512 MiB each way, grid 2048 x 256, nt U4: %6.0f GB/s (%.4f ms)\n", gbs(rows * L * 16, t), t);
    }
    printf("== row pattern, many rows: 65536 rows x 64K doubles at pitch 2^16+128 (64 GiB moved) ==\n");
    if (!quick) {
        CHECK(hipFree(in));
        CHECK(hipFree(out));
        const size_t rows = 65536, L = 65536, pitch = L + 128;
        CHECK(hipMalloc(&in, rows * pitch * 8));
        CHECK(hipMalloc(&out, rows * pitch * 8));
        CHECK(hipMemset(in, 0x11, rows * pitch * 8));
        CHECK(hipMemset(out, 0, rows * pitch * 8));
        double t = tm.run([&] { hipLaunchKernelGGL((copy_rows<true, 8>), dim3(rows), dim3(128), 0, 0, in, out, rows, L / 2, pitch / 2); }, 3);
        printf("one workgroup per row (65536 workgroups x 128): %6.0f GB/s\n", gbs(rows * L * 16, t));
        t = tm.run([&] { hipLaunchKernelGGL((copy_rows<true, 8>), dim3(1024), dim3(128), 0, 0, in, out, rows, L / 2, pitch / 2); }, 3);
        printf("persistent 1024 workgroups x 128 walking rows:  %6.0f GB/s\n", gbs(rows * L * 16, t));
        t = tm.run([&] { hipLaunchKernelGGL((copy_linear<true, 4>), dim3(2048), dim3(256), 0, 0, in, out, rows * L / 2); }, 3);
        printf("linear sweep 32 GiB each way:                   %6.0f GB/s\n", gbs(rows * L * 16, t));
    }
    return 0;
}
