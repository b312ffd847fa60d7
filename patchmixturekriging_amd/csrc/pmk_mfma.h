// fp64 MFMA wave-tile routines shared by the Cholesky panel kernel and the prediction kernel.
//
// A wave owns a tile of NPI*32 "register-dimension" indices by NPJ*32 "lane-dimension" indices
// (layout: pmk_device.h).  Operands are never staged through LDS: an fp64 16x16x4 MFMA takes 64
// cycles on a gfx950 SIMD and consumes 1 KiB of operands, so a (4 x 2)-pair tile needs only six
// 16-byte loads per lane for every 32 MFMAs (2048 cycles) -- the loads are issued PF k-steps ahead
// straight into registers and the four waves of a workgroup share the I-operand through L1.
#pragma once

#include "pmk_device.h"

namespace pmk {

template <int NPI, int NPJ>
struct WaveTile {
    double4_t f[2 * NPI][2 * NPJ];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < 2 * NPI; ++i)
#pragma unroll
            for (int j = 0; j < 2 * NPJ; ++j) f[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
    }
};

template <int NPI, int NPJ>
struct OperandRegs {
    double2_t a[NPI];
    double2_t b[NPJ];
};

template <int NPI, int NPJ>
__device__ __forceinline__ void load_operands(OperandRegs<NPI, NPJ> &r, const double *opI, int64_t ldI,
                                              const double *opJ, int64_t ldJ, int k, int lane)
{
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi) r.a[pi] = load_pair(opI + 32 * pi + (int64_t)k * ldI, ldI, lane);
#pragma unroll
    for (int pj = 0; pj < NPJ; ++pj) r.b[pj] = load_pair(opJ + 32 * pj + (int64_t)k * ldJ, ldJ, lane);
}

template <int NPI, int NPJ>
__device__ __forceinline__ void mfma_step(WaveTile<NPI, NPJ> &t, const OperandRegs<NPI, NPJ> &r)
{
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi)
#pragma unroll
        for (int ei = 0; ei < 2; ++ei)
#pragma unroll
            for (int pj = 0; pj < NPJ; ++pj)
#pragma unroll
                for (int ej = 0; ej < 2; ++ej)
                    t.f[2 * pi + ei][2 * pj + ej] = mfma64(r.a[pi][ei], r.b[pj][ej], t.f[2 * pi + ei][2 * pj + ej]);
}

// t[I][J] += sum_{k < K} MI[I, k] * MJ[J, k]   (opI = &MI[I0, 0], opJ = &MJ[J0, 0]; both column-major
// with the tile index along the contiguous dimension).  K must be a positive multiple of 4*PF.
template <int NPI, int NPJ, int PF>
__device__ __forceinline__ void gemm_nt(WaveTile<NPI, NPJ> &t, const double *opI, int64_t ldI,
                                        const double *opJ, int64_t ldJ, int K, int lane)
{
    OperandRegs<NPI, NPJ> buf[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) load_operands(buf[s], opI, ldI, opJ, ldJ, 4 * s, lane);
    for (int k0 = 0; k0 < K; k0 += 4 * PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            mfma_step(t, buf[s]);
            int kn = k0 + 4 * (s + PF);
            kn = kn < K ? kn : K - 4;      // clamp: the tail re-reads the last step instead of branching
            load_operands(buf[s], opI, ldI, opJ, ldJ, kn, lane);
            // keep the refill of buffer s behind its own MFMAs: without this fence the scheduler hoists
            // every load of the group to the loop head and doubles the operand registers (spills)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// In-register triangular solve with an inverted 128 x 128 diagonal block:
//   t[I][J] <- sum_k Linv[I][k] * t[k][J],   I, k in [0,128)  (NPI = 4)
// Linv: column-major 128 x 128, lower triangular with explicit zeros above the diagonal.
// The accumulator registers of t are used directly as the MFMA B operand: register q of fragment
// (pi, ei) holds k = 32 pi + 8 q + 2 (lane>>4) + ei, which is a legal k-step when the A operand is
// gathered with the same k.  Output pair blocks are produced from the last to the first so the
// update is in place.
template <int NPJ>
__device__ __forceinline__ void tri_solve_inplace(WaveTile<4, NPJ> &t, const double *Linv, int lane)
{
#pragma unroll
    for (int pip = 3; pip >= 0; --pip) {
        double4_t o[2][2 * NPJ];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 2 * NPJ; ++j) o[e][j] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int pi = 0; pi <= pip; ++pi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int ei = 0; ei < 2; ++ei) {
                    const int kl = 32 * pi + 8 * q + 2 * (lane >> 4) + ei;
                    const double2_t a =
                        *reinterpret_cast<const double2_t *>(Linv + 32 * pip + 2 * (lane & 15) + kl * TILE);
#pragma unroll
                    for (int j = 0; j < 2 * NPJ; ++j) {
                        o[0][j] = mfma64(a[0], t.f[2 * pi + ei][j][q], o[0][j]);
                        o[1][j] = mfma64(a[1], t.f[2 * pi + ei][j][q], o[1][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * NPJ; ++j) {
            t.f[2 * pip + 0][j] = o[0][j];
            t.f[2 * pip + 1][j] = o[1][j];
        }
    }
}

}  // namespace pmk
