/* kernel_common.h -- device-side helpers shared by the kernel translation units. */
#ifndef DSP_KERNEL_COMMON_H
#define DSP_KERNEL_COMMON_H

#include "dsp_internal.h"

namespace dsp {

typedef double d2 __attribute__((ext_vector_type(2)));

/* Read-only tables (coefficients, matrix powers, design indices) are read
 * through the constant address space so that wave-uniform loads become scalar
 * loads (s_load_*) into SGPRs even though the kernel also stores to global
 * memory; the bank never aliases its tables with sample buffers. */
typedef const double __attribute__((address_space(4))) *cdouble_p;
typedef const int32_t __attribute__((address_space(4))) *cint_p;
__device__ __forceinline__ cdouble_p as_const(const double *p)
{
    return (cdouble_p)(uintptr_t)p;
}
__device__ __forceinline__ cint_p as_const(const int32_t *p)
{
    return (cint_p)(uintptr_t)p;
}

/* Ordering point between LDS phases of a ONE-WAVEFRONT workgroup.  A wave's LDS
 * instructions execute in issue order, so a ds_read issued after a ds_write
 * sees its data, other lanes' included; only the compiler has to be kept from
 * moving accesses across.  __syncthreads() must not be used here: it lowers to
 * s_waitcnt vmcnt(0) + s_barrier and would drain the global prefetch and the
 * output stores that these kernels keep in flight across phases. */
/* x is read once and y written once per launch: the nontemporal hint on both sides of a streaming kernel is
 * worth 4 % of HBM bandwidth on MI355X (tools/copy_nt.py: 4.89 -> 5.10 TB/s for a row copy; either side alone: 1 %). */
__device__ __forceinline__ d2 stream_load(const d2 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stream_store(d2 *p, d2 v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

/* Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence over
 * every address space, which on gfx950 drains vmcnt: global loads prefetched across the barrier
 * and stores still in flight would all be waited for. */
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ int group_channel(const GroupView &g, int idx)
{
    return g.chan ? g.chan[idx] : g.chan0 + idx;
}


}  // namespace dsp

#endif
