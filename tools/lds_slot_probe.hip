// Diagnostic: what HW_REG_LDS_ALLOC / HW_ID report for the workgroups that share a CU (2 waves + 37.9 KB of LDS each,
// like iir_pipe_kernel): is there a register from which a workgroup can tell its slot among its co-residents?
// build: hipcc --offload-arch=gfx950 -O2 tools/lds_slot_probe.hip -o tools/_build/lds_slot_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

__global__ void __launch_bounds__(128) probe(unsigned *out, int spin)
{
    extern __shared__ char smem[];
    volatile char *p = smem;
    p[threadIdx.x] = 1;
    if ((threadIdx.x & 63) == 0) {
        unsigned *o = out + (blockIdx.x * 2 + (threadIdx.x >> 6)) * 4;
        o[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID
        o[1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);     // LDS_ALLOC
        o[2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // XCC_ID
        o[3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 5);     // GPR_ALLOC
    }
    // stay resident for a while so that a full round of workgroups coexists
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(32);
}

int main()
{
    const int nwg = 2048, lds = 2 * 16384 + 8 * 64 * 10;
    unsigned *d;
    hipMalloc(&d, nwg * 2 * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(128), lds, 0, d, 20000);     // 200 us
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 2 * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<std::pair<int, unsigned>>> by_cu;        // (xcc, se, cu) -> (blockIdx, lds_alloc)
    for (int b = 0; b < nwg; ++b) {
        const unsigned hw = h[(b * 2) * 4], la = h[(b * 2) * 4 + 1], xcc = h[(b * 2) * 4 + 2] & 15;
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back({b, la});
    }
    printf("CUs seen: %zu\n", by_cu.size());
    int shown = 0;
    for (auto &kv : by_cu) {
        if (shown++ >= 6) break;
        printf("xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 255, (kv.first >> 4) & 1, kv.first & 15);
        for (auto &e : kv.second) printf("  wg %4d lds_alloc 0x%08x (base %u size %u)", e.first, e.second, e.second & 0xff, (e.second >> 12) & 0x1ff);
        printf("\n");
    }
    for (int w = 0; w < 2; ++w)
        printf("wg 0 wave %d: hw_id 0x%08x lds_alloc 0x%08x gpr_alloc 0x%08x\n", w, h[w * 4], h[w * 4 + 1], h[w * 4 + 3]);
    std::map<unsigned, int> bases;
    for (int b = 0; b < nwg; ++b) bases[h[(b * 2) * 4 + 1]]++;
    printf("distinct LDS_ALLOC values: %zu:", bases.size());
    for (auto &kv : bases) printf(" 0x%x x%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
