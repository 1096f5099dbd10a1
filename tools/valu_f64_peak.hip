// Diagnostic: attainable FP64 VALU rate on this GPU for independent streams of v_fma_f64, and of the unfused
// v_mul_f64 + v_add_f64 pairs the bit-exact FIR kernels are made of (8 independent chains per lane, 1..8 waves
// per SIMD).  Prints lane-operations per second.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(double *sink, int iters, double a, double b)
{
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) v[i] = __builtin_fma(v[i], a, b);
                else { double p = v[i] * a; asm volatile("" : "+v"(p)); v[i] = p + b; }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678) *sink = s;
}
int main()
{
    double *sink; hipMalloc(&sink, 8);
    const int iters = 4000;
    for (int wgs_per_cu : {1, 2, 4, 8}) {
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, sink, iters, 1.0000001, 1e-9);
                else hipLaunchKernelGGL(k<1>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, sink, iters, 1.0000001, 1e-9);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)256 * wgs_per_cu * 256 * iters * 64 * (mode == 0 ? 1 : 2);
            printf("%d waves/SIMD  %-18s: %.2f T lane-ops/s (%s)\n", wgs_per_cu, mode == 0 ? "v_fma_f64" : "v_mul_f64+v_add_f64",
                   ops / ms / 1e9, mode == 0 ? "x2 = TFLOP/s" : "= TFLOP/s");
        }
    }
    return 0;
}
