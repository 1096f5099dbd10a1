/* mfma_scan.h -- lane-movement helpers of the scan kernels that run their prefix on the FP64 matrix cores
 * (kernels.hip: iir_scan16_kernel; pipe16_kernel.hip). */
#ifndef DSP_MFMA_SCAN_H
#define DSP_MFMA_SCAN_H

#include "kernel_common.h"

namespace dsp {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lane_swap32(double &a, double &b)
{
    u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}

__device__ __forceinline__ void lane_swap16(double &a, double &b)
{
    u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi[0], (int)lo[0]);
    b = __hiloint2double((int)hi[1], (int)lo[1]);
}

/* element (register a, 16-lane group g) <-> (register g, group a) */
__device__ __forceinline__ void transpose4(double &r0, double &r1, double &r2, double &r3)
{
    lane_swap32(r0, r2);
    lane_swap32(r1, r3);
    lane_swap16(r0, r1);
    lane_swap16(r2, r3);
}

/* within every 16-lane group: lane r takes cur from lane r-D, lanes r < D take prev from lane r-D+16 */
template <int D>
__device__ __forceinline__ double row_shift(double cur, double prev)
{
    static_assert(D >= 1 && D <= 15, "row shift inside a 16-lane group");
    const int plo = __double2loint(prev), phi = __double2hiint(prev);
    int lo = __builtin_amdgcn_update_dpp(plo, plo, 0x120 + D, 0xf, 0xf, false);                  /* row_ror:D */
    int hi = __builtin_amdgcn_update_dpp(phi, phi, 0x120 + D, 0xf, 0xf, false);
    lo = __builtin_amdgcn_update_dpp(lo, __double2loint(cur), 0x110 + D, 0xf, 0xf, false);       /* row_shr:D */
    hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(cur), 0x110 + D, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

}  // namespace dsp

#endif
