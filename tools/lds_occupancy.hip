// How many 128-thread workgroups with a given dynamic LDS size does a CU of this device hold?
// (the order 9..16 pipelined kernel wants 4 x 40960 bytes = all 160 KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(128, 2) probe(double *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *p = reinterpret_cast<double *>(smem);
    p[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[blockIdx.x * 128 + threadIdx.x] = p[127 - threadIdx.x];
}
int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n",
           (size_t)prop.sharedMemPerMultiprocessor, (size_t)prop.sharedMemPerBlock, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    const size_t sizes[] = {37920, 40448, 40960, 40992, 41024, 41472, 41984, 43008, 49152, 54272, 54613, 65536};
    for (size_t s : sizes) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe, 128, s);
        printf("dynamic LDS %6zu bytes: %d workgroups per CU (%s)\n", s, nb, hipGetErrorString(e));
    }
    return 0;
}
