// Diagnostic: attainable v_mfma_f64_16x16x4_f64 rate on this device (no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/mfma_f64_peak.hip -o tools/_build/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int kAcc>
__global__ void __launch_bounds__(256) burn(double *sink, int iters, long long *clk)
{
    d4 acc[kAcc];
    for (int t = 0; t < kAcc; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    long long c0 = __builtin_readcyclecounter();
    long long r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < kAcc; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    long long c1 = __builtin_readcyclecounter();
    long long r1 = wall_clock64();
    double s = 0;
    for (int t = 0; t < kAcc; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int kAcc>
void run(int wgs, int threads, int iters, double *sink, long long *clk)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    burn<kAcc><<<wgs, threads>>>(sink, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    burn<kAcc><<<wgs, threads>>>(sink, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double waves = (double)wgs * threads / 64;
    double flops = waves * iters * kAcc * 2048.0;
    printf("acc=%d wgs=%d threads=%d iters=%d: %.3f ms  %.2f TFLOP/s  shader cycles/MFMA/wave %.1f  (cycles %lld, 100MHz ticks %lld => %.2f GHz)\n",
           kAcc, wgs, threads, iters, ms, flops / ms / 1e9, (double)h[0] / ((double)iters * kAcc), h[0], h[1],
           h[1] ? (double)h[0] / (h[1] * 10.0) : 0.0);
}

int main()
{
    double *sink; long long *clk;
    hipMalloc(&sink, 8ull * 4096 * 1024);
    hipMalloc(&clk, 16);
    run<4>(256, 256, 20000, sink, clk);       // 1 wave / SIMD
    run<4>(512, 256, 20000, sink, clk);       // 2 waves / SIMD
    run<4>(1024, 256, 20000, sink, clk);      // 4 waves / SIMD
    run<1>(256, 256, 40000, sink, clk);       // dependent chain, 1 wave / SIMD
    run<2>(256, 256, 40000, sink, clk);
    run<4>(2048, 256, 40000, sink, clk);      // long run: sustained clock
    return 0;
}
