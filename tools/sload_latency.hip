// Diagnostic: latency of a scalar-cache load batch (2 x s_load_dwordx16, the zero-state pass's table batch of
// the pipelined IIR kernel) on gfx950 -- idle, with every CU busy doing the same, and for broadcast LDS reads and
// a dependent v_fma_f64 chain for scale.  Prints shader-clock cycles per dependent step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef const double __attribute__((address_space(4))) *cdouble_p;

template <int kBytes>
__global__ void __launch_bounds__(64) sload_chain(const double *tab, unsigned long long *out, int iters, double *sink)
{
    cdouble_p t = (cdouble_p)(uintptr_t)tab;
    double acc = 0.0;
    // warm
    for (int o = 0; o < kBytes / 8; o += 8) acc += t[o];
    unsigned off = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        double g[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) g[k] = t[off + k];               // 128 B: two x16 loads
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += g[k];
        acc += s;
        // the next address depends on the loaded data (always 0 mod table, data are zeros): a true chain
        off = (off + 16 + (unsigned)__builtin_amdgcn_readfirstlane((int)s)) & (kBytes / 8 - 1);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345.0) *sink = acc;
}

__global__ void __launch_bounds__(64) fma_chain(unsigned long long *out, int iters, double *sink, double a, double b)
{
    double v = threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v = __builtin_fma(v, a, b);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (v == 12345.0) *sink = v;
}

__global__ void __launch_bounds__(64) lds_bcast_chain(unsigned long long *out, int iters, double *sink)
{
    __shared__ double tab[512];
    for (int i = threadIdx.x; i < 512; i += 64) tab[i] = 0.0;
    __syncthreads();
    double acc = 0.0;
    unsigned off = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += tab[off + k];               // all lanes the same address: broadcast
        acc += s;
        off = (off + 16 + (unsigned)(int)s) & 511;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 12345.0) *sink = acc;
}

static double median(std::vector<unsigned long long> v, int iters)
{
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2] / iters;
}

#include <algorithm>
int main()
{
    const int iters = 2000;
    double *tab, *sink;
    unsigned long long *out;
    hipMalloc(&tab, 1 << 20);
    hipMemset(tab, 0, 1 << 20);
    hipMalloc(&sink, 8);
    hipMalloc(&out, 8 * 4096);
    auto run = [&](const char *name, int grid, auto launch) {
        launch(grid);
        hipDeviceSynchronize();
        launch(grid);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), out, 8 * grid, hipMemcpyDeviceToHost);
        printf("%-58s grid %5d: %7.1f cycles per step (median over workgroups)\n", name, grid, median(h, iters));
    };
    for (int grid : {1, 256, 2048}) {
        run("s_load 128 B dependent chain, 2 KiB table", grid, [&](int g) {
            hipLaunchKernelGGL((sload_chain<2048>), dim3(g), dim3(64), 0, 0, tab, out, iters, sink); });
        run("s_load 128 B dependent chain, 8 KiB table", grid, [&](int g) {
            hipLaunchKernelGGL((sload_chain<8192>), dim3(g), dim3(64), 0, 0, tab, out, iters, sink); });
        run("s_load 128 B dependent chain, 64 KiB table (misses)", grid, [&](int g) {
            hipLaunchKernelGGL((sload_chain<65536>), dim3(g), dim3(64), 0, 0, tab, out, iters, sink); });
        run("16 broadcast ds_read_b64, dependent chain", grid, [&](int g) {
            hipLaunchKernelGGL(lds_bcast_chain, dim3(g), dim3(64), 0, 0, out, iters, sink); });
        run("16 dependent v_fma_f64", grid, [&](int g) {
            hipLaunchKernelGGL(fma_chain, dim3(g), dim3(64), 0, 0, out, iters, sink, 1.0, 0.0); });
    }
    return 0;
}
