// Diagnostic: prints what v_permlane16_swap / v_permlane32_swap / DPP row_shr, row_ror do on gfx950,
// lane by lane, to pin the semantics the order-16 scan kernel relies on.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned *out)
{
    const unsigned lane = threadIdx.x;
    unsigned a = lane, b = 100 + lane;
    uint2v r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    uint2v r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    unsigned shr3 = __builtin_amdgcn_update_dpp(999u, a, 0x110 + 3, 0xf, 0xf, false);   // row_shr:3, keep old
    unsigned ror3 = __builtin_amdgcn_update_dpp(999u, a, 0x120 + 3, 0xf, 0xf, false);   // row_ror:3
    unsigned wshr = __builtin_amdgcn_update_dpp(999u, a, 0x138, 0xf, 0xf, false);       // wave_shr:1
    out[lane * 8 + 0] = r16[0]; out[lane * 8 + 1] = r16[1];
    out[lane * 8 + 2] = r32[0]; out[lane * 8 + 3] = r32[1];
    out[lane * 8 + 4] = shr3; out[lane * 8 + 5] = ror3; out[lane * 8 + 6] = wshr;
}
int main()
{
    unsigned *d, h[64 * 8];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[7] = {"p16swap.vdst", "p16swap.src", "p32swap.vdst", "p32swap.src", "row_shr:3", "row_ror:3", "wave_shr:1"};
    for (int k = 0; k < 7; ++k) {
        printf("%-13s", names[k]);
        for (int l = 0; l < 64; ++l) printf(" %u", h[l * 8 + k]);
        printf("\n");
    }
    return 0;
}
