// Precision selection for the factorisation / prediction kernels.  pmk_kmat.hip, pmk_chol.hip and
// pmk_predict.hip are compiled twice: as is (real = double, namespace pmk::f64) and with -DPMK_REAL_F32
// (real = float, namespace pmk::f32, v_mfma_f32_16x16x4_f32 at twice the fp64 MFMA rate).
#pragma once

#include "pmk_device.h"

#ifdef PMK_REAL_F32
#define PMK_NS f32
#else
#define PMK_NS f64
#endif

namespace pmk {
namespace PMK_NS {

#ifdef PMK_REAL_F32
typedef float real;
#else
typedef double real;
#endif
typedef real real2_t __attribute__((ext_vector_type(2)));
typedef real real4_t __attribute__((ext_vector_type(4)));

// D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] in both
// precisions; the accumulator layouts differ: register q of lane l holds D[i = frag_irow(l>>4, q)][j = l&15]
// with frag_irow = (l>>4) + 4q for fp64 and 4(l>>4) + q for fp32.
#ifdef PMK_REAL_F32
__device__ __forceinline__ real4_t mfma_real(real a_i, real b_j, real4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a_i, b_j, c, 0, 0, 0);
}
__device__ __forceinline__ constexpr int frag_irow(int lane_group, int q) { return 4 * lane_group + q; }
__device__ __forceinline__ real rsqrt_real(real d)
{
    real r = __builtin_amdgcn_rsqf(d);
    return __builtin_fmaf(r, __builtin_fmaf(-0.5f * d * r, r, 0.5f), r);
}
#else
__device__ __forceinline__ real4_t mfma_real(real a_i, real b_j, real4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a_i, b_j, c, 0, 0, 0);
}
__device__ __forceinline__ constexpr int frag_irow(int lane_group, int q) { return lane_group + 4 * q; }
// reciprocal square root from v_rsq_f64 + two Newton steps (~1 ulp; a short dependent chain instead of the
// IEEE sqrt + divide sequences on the serial pivot path of the tile factorisation)
__device__ __forceinline__ real rsqrt_real(real d)
{
    real r = __builtin_amdgcn_rsq(d);
    const real hd = 0.5 * d;
    r = __builtin_fma(r, __builtin_fma(-hd * r, r, 0.5), r);
    return __builtin_fma(r, __builtin_fma(-hd * r, r, 0.5), r);
}
#endif

// I index (within a 128-wide tile) of the element that lane `lane` holds in register q of fragment fi = 2 pi + ei
__device__ __forceinline__ int tile_i(int fi, int lane, int q)
{
    return 32 * (fi >> 1) + 2 * frag_irow(lane >> 4, q) + (fi & 1);
}

}  // namespace PMK_NS
}  // namespace pmk
