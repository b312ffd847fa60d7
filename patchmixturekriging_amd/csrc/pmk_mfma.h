// fp64 MFMA wave-tile routines shared by the Cholesky panel kernel and the prediction kernel.
//
// A wave owns a tile of NPI*32 "register-dimension" indices by NPJ*32 "lane-dimension" indices
// (layout: pmk_device.h).  Operands are never staged through LDS: an fp64 16x16x4 MFMA takes 64
// cycles on a gfx950 SIMD and consumes 1 KiB of operands, so a (4 x 2)-pair tile needs only six
// 16-byte loads per lane for every 32 MFMAs (2048 cycles) -- the loads are issued PF k-steps ahead
// straight into registers and the four waves of a workgroup share the I-operand through L1.
#pragma once

#include "pmk_real.h"

namespace pmk {
namespace PMK_NS {

// Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8) and every XCD has its own
// L2.  xcd_remap() turns the hardware block id into a logical id such that each XCD works on one
// contiguous range of logical ids, so workgroups that stream the same operand (the block rows of one
// patch in the panel kernel, the strips of one region in prediction) share it through ONE L2 instead of
// fetching it from HBM once per XCD.  Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// wave-uniform values that the compiler cannot prove uniform (loaded through a lane-indexed path): move them to
// scalar registers so that address arithmetic on them runs on the scalar unit
__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ const T *uniform_ptr(const T *p)
{
    return reinterpret_cast<const T *>(uniform_i64(reinterpret_cast<int64_t>(p)));
}

template <int NPI, int NPJ>
struct WaveTile {
    real4_t f[2 * NPI][2 * NPJ];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < 2 * NPI; ++i)
#pragma unroll
            for (int j = 0; j < 2 * NPJ; ++j) f[i][j] = real4_t{0, 0, 0, 0};
    }
};

// t[I][J] += sum_{k < K} MI[I, k] * MJ[J, k]   (opI = &MI[I0, 0], opJ = &MJ[J0, 0]; both column-major
// with the tile index along the contiguous dimension).  The two operands are prefetched into separate
// register rings: the I operand is shared by the workgroups of a patch / region and comes from L2
// (depth PFI k-steps), the J operand is this wave's own stream from HBM (depth PFJ, a multiple of PFI).
// K must be a positive multiple of 4*PFJ.
// NACT < NPI: only the first NACT index pairs of the I dimension are computed (the rest of the tile is known to be
// zero: identity padding of the last block row).
//
// OVER-READ: the rings are refilled without a bounds test, so the routine LOADS (and never uses) up to PFI k-steps of
// the I operand and PFJ k-steps of the J operand past K.  Every caller passes operands that lie inside a padded patch
// slab / strip with at least that many columns (rows) behind K.
//
// Waits.  The loop relies on counted waits (s_waitcnt vmcnt(N): the loads of the last PF-1 k-steps stay in flight).
// The wait at the loop head serves the entry from the prologue as well as the back edge, so the prologue must issue
// its loads in the order the loop consumes them and the scheduler must not reorder them: left alone it moved the loads
// of k-step 0 to the end of the prologue, the head wait became vmcnt(0), and every pass of the loop then waited for the
// loads it had issued a few instructions earlier (a wave alone on its SIMD ran at 73 % of the MFMA rate instead of
// 90 %; tools/gemm_probe.hip).  Each I slot is refilled right behind the MFMAs that consumed it, so the loads of a
// k-step are spread over its 16 MFMAs instead of queueing behind the last one.
// Scalar-base form (tools/gemm_probe.hip: 98.7 % of the fp64 MFMA peak against 94.4 % for the same loop with vector
// address arithmetic; a wave alone on its SIMD 97.6 % against 92.7 %).  On gfx950 a vector instruction is NOT hidden
// behind another wave's -- or the same wave's -- MFMAs: 16 extra fp64 FMAs per k-step cost the loop 8.5 %, and so do
// the handful of 64-bit address updates per k-step of the forms below.  Here every ring address is
//     (wave-uniform base in scalar registers, advanced on the scalar ALU)  +  (one 32-bit byte offset per lane),
// so the loop body holds nothing but MFMAs, loads and scalar adds.  The compiler does not emit that addressing mode
// for this pattern; the loads are inline asm, which also means that the waits are ours: s_waitcnt vmcnt(LOADS (PF-1))
// in front of every k-step (the ring registers are in/out operands of the wait, so no MFMA can be scheduled above it);
// past the end of K the refills read the last k-step again, and the ring is drained after the loop -- no over-read.
// Requirements: opI, opJ, ldI, ldJ wave-uniform; K a multiple of 4 PF, K >= 4 PF; the caller's own earlier loads are
// consumed (waited for) before the call or never touch the ring.  Do NOT use where the compiler spills around the call:
// it believes an asm load's destination is valid at once and could copy it to scratch before the wait.
template <int NPI, int NPJ, int PF, int NACT = NPI, bool PEEL = false>
__device__ __forceinline__ void gemm_nt_sbase(WaveTile<NPI, NPJ> &t, const real *opI, int64_t ldI, const real *opJ, int64_t ldJ,
                                              int K, int lane)
{
    static_assert(PF == 4, "the unrolled waits below are written for a ring of 4 k-steps");
    static_assert(NACT >= 1 && NACT <= 4 && NPJ >= 1 && NPJ <= 3, "ring shape");
    constexpr int ES = (int)sizeof(real);
    real2_t ra[PF][NACT], rb[PF][NPJ];
    const uint32_t offI = (uint32_t)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldI) * ES);
    const uint32_t offJ = (uint32_t)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldJ) * ES);
    const char *bI = uniform_ptr(reinterpret_cast<const char *>(opI));
    const char *bJ = uniform_ptr(reinterpret_cast<const char *>(opJ));
    const int64_t sI = uniform_i64(4 * ldI * ES), sJ = uniform_i64(4 * ldJ * ES);
#ifdef PMK_REAL_F32
#define PMK_LDI(dst, pi) asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(offI), "s"(bI), "n"(32 * ES * (pi)))
#define PMK_LDJ(dst, pj) asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(offJ), "s"(bJ), "n"(32 * ES * (pj)))
#else
#define PMK_LDI(dst, pi) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(offI), "s"(bI), "n"(32 * ES * (pi)))
#define PMK_LDJ(dst, pj) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(offJ), "s"(bJ), "n"(32 * ES * (pj)))
#endif
    // one k-step's refills / the wait in front of a k-step: every register of the slot is an in/out operand
    auto refill_I = [&](int s, int pi) {
        if (pi == 0) PMK_LDI(ra[s][0], 0);
        if (NACT > 1 && pi == 1) PMK_LDI(ra[s][NACT > 1 ? 1 : 0], 1);
        if (NACT > 2 && pi == 2) PMK_LDI(ra[s][NACT > 2 ? 2 : 0], 2);
        if (NACT > 3 && pi == 3) PMK_LDI(ra[s][NACT > 3 ? 3 : 0], 3);
    };
    auto refill_J = [&](int s) {
        PMK_LDJ(rb[s][0], 0);
        if (NPJ > 1) PMK_LDJ(rb[s][NPJ > 1 ? 1 : 0], 1);
        if (NPJ > 2) PMK_LDJ(rb[s][NPJ > 2 ? 2 : 0], 2);
    };
    constexpr int LOADS = NACT + NPJ;          // per k-step
    // the wait in front of a k-step; every register of the slot passes through an (empty) asm behind it, so that no MFMA
    // that reads the slot can be scheduled above the wait
#define PMK_WAITV(n, s)                                                                         \
    do {                                                                                        \
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ra[s][0]) : "n"(n));                          \
        if (NACT > 1) asm volatile("" : "+v"(ra[s][NACT > 1 ? 1 : 0]));                         \
        if (NACT > 2) asm volatile("" : "+v"(ra[s][NACT > 2 ? 2 : 0]));                         \
        if (NACT > 3) asm volatile("" : "+v"(ra[s][NACT > 3 ? 3 : 0]));                         \
        asm volatile("" : "+v"(rb[s][0]));                                                      \
        if (NPJ > 1) asm volatile("" : "+v"(rb[s][NPJ > 1 ? 1 : 0]));                           \
        if (NPJ > 2) asm volatile("" : "+v"(rb[s][NPJ > 2 ? 2 : 0]));                           \
    } while (0)
    auto mfmas = [&](int s, int pi) {
#pragma unroll
        for (int ei = 0; ei < 2; ++ei)
#pragma unroll
            for (int pj = 0; pj < NPJ; ++pj)
#pragma unroll
                for (int ej = 0; ej < 2; ++ej)
                    t.f[2 * pi + ei][2 * pj + ej] = mfma_real(ra[s][pi][ei], rb[s][pj][ej], t.f[2 * pi + ei][2 * pj + ej]);
    };
    __builtin_amdgcn_sched_barrier(0);
    const int nk = K / 4;                       // k-steps
#pragma unroll
    for (int s = 0; s < PF; ++s) {
#pragma unroll
        for (int pi = 0; pi < NACT; ++pi) refill_I(s, pi);
        refill_J(s);
        if (s + 1 < PF || nk > PF) {            // after the prologue the bases point at k-step min(PF, nk - 1)
            bI += sI;
            bJ += sJ;
        }
    }
    // refill r (counted from 0) fetches k-step min(PF + r, nk - 1): past the end the last k-step is simply read again
    // (into registers nobody uses any more), which keeps the number of loads in flight -- and with it every wait of the
    // loop -- the same in all passes, the last one included; the ring is drained after the loop
    int left = nk - PF - 1;                     // base advances still allowed
    auto pass = [&]() {                           // PF k-steps
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            if (s == 0) PMK_WAITV(LOADS * 3, 0);
            else if (s == 1) PMK_WAITV(LOADS * 3, 1);
            else if (s == 2) PMK_WAITV(LOADS * 3, 2);
            else PMK_WAITV(LOADS * 3, 3);
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi) {
                mfmas(s, pi);
                __builtin_amdgcn_sched_barrier(0);
                refill_I(s, pi);
                if (pi == NACT - 1) refill_J(s);
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool adv = left > 0;
            bI += adv ? sI : 0;
            bJ += adv ? sJ : 0;
            --left;
        }
    };
    if (PEEL) pass();                             // see gemm_nt: keeps the compiler's own waits out of the loop
    for (int k0 = PEEL ? 4 * PF : 0; k0 < K; k0 += 4 * PF) pass();
    // drain: the (redundant) refills of the last pass are still in flight, and the compiler must not reuse their
    // destination registers before they have landed
    PMK_WAITV(0, 0);
    PMK_WAITV(0, 1);
    PMK_WAITV(0, 2);
    PMK_WAITV(0, 3);
#undef PMK_LDI
#undef PMK_LDJ
#undef PMK_WAITV
}

#ifndef PMK_PEEL_DEFAULT
#define PMK_PEEL_DEFAULT false
#endif
template <int NPI, int NPJ, int PFI, int PFJ = PFI, int NACT = NPI, bool PEEL = PMK_PEEL_DEFAULT>
__device__ __forceinline__ void gemm_nt(WaveTile<NPI, NPJ> &t, const real *opI, int64_t ldI,
                                        const real *opJ, int64_t ldJ, int K, int lane)
{
    static_assert(PFJ % PFI == 0, "the J ring depth must be a multiple of the I ring depth");
    static_assert(NACT >= 1 && NACT <= NPI, "active pairs");
    real2_t ra[PFI][NACT], rb[PFJ][NPJ];
    // the lane's element of the next k-step to load: indices 2 rho, 2 rho + 1 at column (lane >> 4)
    const real *qI = opI + 2 * (lane & 15) + (int64_t)(lane >> 4) * ldI;
    const real *qJ = opJ + 2 * (lane & 15) + (int64_t)(lane >> 4) * ldJ;
    const int64_t sI = 4 * ldI, sJ = 4 * ldJ;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < PFJ; ++s) {
        if (s < PFI) {
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi) ra[s][pi] = *reinterpret_cast<const real2_t *>(qI + 32 * pi);
            qI += sI;
        }
        // the J operand is read by this wave only (its own rows / strip columns): a streaming hint keeps it from
        // displacing the operand that the workgroups of a patch or region share through L2
#pragma unroll
        for (int pj = 0; pj < NPJ; ++pj) rb[s][pj] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(qJ + 32 * pj));
        qJ += sJ;
        __builtin_amdgcn_sched_barrier(0);
    }
    auto pass = [&]() {                   // PFJ k-steps
#pragma unroll
        for (int s = 0; s < PFJ; ++s) {
            const int si = s % PFI;
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi) {
#pragma unroll
                for (int ei = 0; ei < 2; ++ei)
#pragma unroll
                    for (int pj = 0; pj < NPJ; ++pj)
#pragma unroll
                        for (int ej = 0; ej < 2; ++ej)
                            t.f[2 * pi + ei][2 * pj + ej] =
                                mfma_real(ra[si][pi][ei], rb[s][pj][ej], t.f[2 * pi + ei][2 * pj + ej]);
                ra[si][pi] = *reinterpret_cast<const real2_t *>(qI + 32 * pi);
                if (pi == NACT - 1) {
#pragma unroll
                    for (int pj = 0; pj < NPJ; ++pj)
                        rb[s][pj] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(qJ + 32 * pj));
                }
                // without this fence the scheduler hoists every load of the k-step group to the loop head and doubles
                // the operand registers (spills)
                __builtin_amdgcn_sched_barrier(0);
            }
            qI += sI;
            qJ += sJ;
        }
    };
    // PEEL: the first pass is peeled off the loop.  Whatever the code in front of the loop still has in flight when it
    // enters (a spilled accumulator being reloaded, the caller's last loads) is then waited for once, in the peeled
    // copy; in a rolled loop that wait sits at the loop head and is executed by every pass -- as vmcnt(0), i.e. for
    // the ring loads as well.  For callers under register pressure (prediction); it costs code size elsewhere.
    if (PEEL) pass();
    for (int k0 = PEEL ? 4 * PFJ : 0; k0 < K; k0 += 4 * PFJ) pass();
}

// The same product with every ring address recomputed from the k index (clamped at the end: no over-read).  Costs a
// few more address instructions per k-step but keeps no running pointers alive: the prediction kernel, which is out
// of registers around its GEMM, gets spill reloads in front of the loop with the pointer form (and with them a
// vmcnt(1) at the loop head); this form compiles there to clean counted waits.
template <int NPI, int NPJ, int PFI, int PFJ = PFI, int NACT = NPI, bool PEEL = true>
__device__ __forceinline__ void gemm_nt_indexed(WaveTile<NPI, NPJ> &t, const real *opI, int64_t ldI,
                                                const real *opJ, int64_t ldJ, int K, int lane)
{
    static_assert(PFJ % PFI == 0, "the J ring depth must be a multiple of the I ring depth");
    static_assert(NACT >= 1 && NACT <= NPI, "active pairs");
    real2_t ra[PFI][NACT], rb[PFJ][NPJ];
    const real *bI = opI + 2 * (lane & 15) + (int64_t)(lane >> 4) * ldI;
    const real *bJ = opJ + 2 * (lane & 15) + (int64_t)(lane >> 4) * ldJ;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < PFJ; ++s) {
        if (s < PFI) {
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi) ra[s][pi] = *reinterpret_cast<const real2_t *>(bI + 32 * pi + (int64_t)(4 * s) * ldI);
        }
#pragma unroll
        for (int pj = 0; pj < NPJ; ++pj)
            rb[s][pj] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(bJ + 32 * pj + (int64_t)(4 * s) * ldJ));
        __builtin_amdgcn_sched_barrier(0);
    }
    auto pass = [&](int k0) {             // PFJ k-steps
#pragma unroll
        for (int s = 0; s < PFJ; ++s) {
            const int si = s % PFI;
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi)
#pragma unroll
                for (int ei = 0; ei < 2; ++ei)
#pragma unroll
                    for (int pj = 0; pj < NPJ; ++pj)
#pragma unroll
                        for (int ej = 0; ej < 2; ++ej)
                            t.f[2 * pi + ei][2 * pj + ej] =
                                mfma_real(ra[si][pi][ei], rb[s][pj][ej], t.f[2 * pi + ei][2 * pj + ej]);
            // refill both slots behind their own MFMAs; the tail re-reads the last k-step instead of branching
            int ki = k0 + 4 * (s + PFI), kj = k0 + 4 * (s + PFJ);
            ki = ki < K ? ki : K - 4;
            kj = kj < K ? kj : K - 4;
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi) ra[si][pi] = *reinterpret_cast<const real2_t *>(bI + 32 * pi + (int64_t)ki * ldI);
#pragma unroll
            for (int pj = 0; pj < NPJ; ++pj)
                rb[s][pj] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(bJ + 32 * pj + (int64_t)kj * ldJ));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (PEEL) pass(0);                    // see gemm_nt
    for (int k0 = PEEL ? 4 * PFJ : 0; k0 < K; k0 += 4 * PFJ) pass(k0);
}

// Buffer-load form of the same product (tools/gemm_probe.hip: 96.2 % of the fp64 MFMA peak with two waves per SIMD,
// 94.0 % alone, against 94.0 / 91.7 % for the pointer form): every ring address is
//     (buffer resource in scalar registers, its base advanced on the scalar ALU once per k-step)  +  (one 32-bit byte
//     offset per lane, constant for the whole product)  +  (an immediate)
// so the loop holds no vector address arithmetic at all, and -- unlike gemm_nt_sbase -- the loads are builtins: the
// compiler counts their waits itself, which makes the form safe where it spills around the call.  The base is a full
// 64-bit address (no 2^31 limit on K x ld); past the end of K the refills read the last k-step again (no over-read).
// Requirements: opI, opJ, ldI, ldJ wave-uniform AND the call reached through wave-uniform control flow only (scalar
// branches): inside a divergent region the resources end up in vector registers and every load is waterfalled.
// K a positive multiple of 4 PF.
template <int NPI, int NPJ, int PF, int NACT = NPI, bool PEEL = true>
__device__ __forceinline__ void gemm_nt_buf(WaveTile<NPI, NPJ> &t, const real *opI, int64_t ldI, const real *opJ, int64_t ldJ,
                                            int K, int lane)
{
    static_assert(NACT >= 1 && NACT <= NPI, "active pairs");
    constexpr int ES = (int)sizeof(real);
    real2_t ra[PF][NACT], rb[PF][NPJ];
    const int vI = (int)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldI) * ES);
    const int vJ = (int)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldJ) * ES);
    const int64_t sI = uniform_i64(4 * ldI * ES), sJ = uniform_i64(4 * ldJ * ES);
    int64_t bI = uniform_i64(reinterpret_cast<int64_t>(opI)), bJ = uniform_i64(reinterpret_cast<int64_t>(opJ));
    const int64_t eI = bI + (int64_t)(K / 4 - 1) * sI, eJ = bJ + (int64_t)(K / 4 - 1) * sJ;      // the last k-step
    auto rsrc = [](int64_t base) {
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
    };
#ifdef PMK_REAL_F32
#define PMK_BUF_LD(r, voff, aux) __builtin_bit_cast(real2_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, aux))
#else
#define PMK_BUF_LD(r, voff, aux) __builtin_bit_cast(real2_t, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, aux))
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        const __amdgpu_buffer_rsrc_t rI = rsrc(bI), rJ = rsrc(bJ);
#pragma unroll
        for (int pi = 0; pi < NACT; ++pi) ra[s][pi] = PMK_BUF_LD(rI, vI + 32 * ES * pi, 0);
#pragma unroll
        for (int pj = 0; pj < NPJ; ++pj) rb[s][pj] = PMK_BUF_LD(rJ, vJ + 32 * ES * pj, 2);      // aux 2 = nt: the wave's own stream
        bI = bI < eI ? bI + sI : eI;
        bJ = bJ < eJ ? bJ + sJ : eJ;
        __builtin_amdgcn_sched_barrier(0);
    }
    auto pass = [&]() {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const __amdgpu_buffer_rsrc_t rI = rsrc(bI), rJ = rsrc(bJ);
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi) {
#pragma unroll
                for (int ei = 0; ei < 2; ++ei)
#pragma unroll
                    for (int pj = 0; pj < NPJ; ++pj)
#pragma unroll
                        for (int ej = 0; ej < 2; ++ej)
                            t.f[2 * pi + ei][2 * pj + ej] =
                                mfma_real(ra[s][pi][ei], rb[s][pj][ej], t.f[2 * pi + ei][2 * pj + ej]);
                ra[s][pi] = PMK_BUF_LD(rI, vI + 32 * ES * pi, 0);
                if (pi == NACT - 1) {
#pragma unroll
                    for (int pj = 0; pj < NPJ; ++pj) rb[s][pj] = PMK_BUF_LD(rJ, vJ + 32 * ES * pj, 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            bI = bI < eI ? bI + sI : eI;
            bJ = bJ < eJ ? bJ + sJ : eJ;
        }
    };
    if (PEEL) pass();                     // see gemm_nt
    for (int k0 = PEEL ? 4 * PF : 0; k0 < K; k0 += 4 * PF) pass();
#undef PMK_BUF_LD
}

// In-register triangular solve of a 128-row tile, by block forward substitution over its four
// 32-row blocks:   t  <-  -L^-1 t     (t: 128 x NJ*16, rows = the I dimension)
//   u_s = Ninv_s (t_s + sum_{j<s} L[s][j] u_j),   Ninv_s = -(L[s][s])^-1  (negated inverted 32 x 32 blocks)
// The operands are staged in LDS by stage_tri_operands(): the six 32 x 32 blocks of L strictly below
// the block diagonal (block (s, j) at index s(s-1)/2 + j) followed by the four Ninv blocks, each
// column-major with leading dimension 32.  Reading them from global memory inside the dependent
// stages exposed an L2 round trip per stage (a fixed ~30 us per block row of the panel launch).
// Every product is an MFMA whose B operand is an accumulator register of t itself: register q of
// fragment (pi, ei) holds row k = 32 pi + 8 q + 2 (lane>>4) + ei, a legal k-step once the A operand
// is gathered with the same k.  The fragment pairing permutes rows only inside a 32-block, so the
// block structure of L is preserved.
constexpr int TRI_LDS_DOUBLES = 10 * 32 * 32;

// cooperative copy by all `nthreads` threads of the workgroup; caller synchronises before and after.
// L: factored diagonal tile (column-major, ldl); ninv: its 4 negated inverted diagonal blocks (global).
// The copy is written as batches of independent loads followed by their LDS stores: left as a rolled
// load -> wait -> store loop (what hipcc makes of the obvious form) it is 20 serialised L2 round trips,
// ~10 us at the head of every workgroup.
template <int BATCH = 10>
__device__ __forceinline__ void stage_tri_operands(real *lds, const real *L, int64_t ldl, const real *ninv,
                                                   int tid, int nthreads)
{
    constexpr int PIECES = 10 * 512;            // 16-byte pieces: 512 per 32 x 32 block
    for (int e0 = tid; e0 < PIECES; e0 += BATCH * nthreads) {
        real2_t v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = e0 + u * nthreads;
            const int ec = e < PIECES ? e : tid;        // clamp (only when PIECES is not a multiple of the batch)
            const int b = ec >> 9, w = ec & 511;        // block, piece
            const int i = 2 * (w & 15), c = w >> 4;     // rows i, i+1 of column c
            const int sblk = (b >= 3) ? 3 : (b >= 1 ? 2 : 1);
            const int j = b - sblk * (sblk - 1) / 2;
            const real *src = (b < 6) ? L + 32 * sblk + i + (int64_t)(32 * j + c) * ldl : ninv + 1024 * (b - 6) + i + 32 * c;
            v[u] = *reinterpret_cast<const real2_t *>(src);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = e0 + u * nthreads;
            if (e < PIECES) {
                const int b = e >> 9, w = e & 511;
                const int i = 2 * (w & 15), c = w >> 4;
                *reinterpret_cast<real2_t *>(lds + 1024 * b + i + 32 * c) = v[u];
            }
        }
    }
}

template <int NPJ>
__device__ __forceinline__ void tri_solve_inplace(WaveTile<4, NPJ> &t, const real *lds, int lane)
{
    const real *base = lds + 2 * (lane & 15);     // + 32 * (k index): k = 2 frag_irow(lane >> 4, q) + e
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int pj = 0; pj < s; ++pj) {
            const real *blk = base + 1024 * (s * (s - 1) / 2 + pj);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const real2_t a = *reinterpret_cast<const real2_t *>(blk + 32 * (2 * frag_irow(lane >> 4, q) + e));
#pragma unroll
                    for (int j = 0; j < 2 * NPJ; ++j) {
                        t.f[2 * s + 0][j] = mfma_real(a[0], t.f[2 * pj + e][j][q], t.f[2 * s + 0][j]);
                        t.f[2 * s + 1][j] = mfma_real(a[1], t.f[2 * pj + e][j][q], t.f[2 * s + 1][j]);
                    }
                }
            }
        }
        real4_t o[2][2 * NPJ];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 2 * NPJ; ++j) o[e][j] = real4_t{0, 0, 0, 0};
        const real *blk = base + 1024 * (6 + s);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const real2_t a = *reinterpret_cast<const real2_t *>(blk + 32 * (2 * frag_irow(lane >> 4, q) + e));
#pragma unroll
                for (int j = 0; j < 2 * NPJ; ++j) {
                    o[0][j] = mfma_real(a[0], t.f[2 * s + e][j][q], o[0][j]);
                    o[1][j] = mfma_real(a[1], t.f[2 * s + e][j][q], o[1][j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * NPJ; ++j) {
            t.f[2 * s + 0][j] = o[0][j];
            t.f[2 * s + 1][j] = o[1][j];
        }
    }
}

// The same block substitution with its operands taken straight from global memory (the factor's diagonal tile in the
// slab, leading dimension ldl, and its four negated inverted 32 x 32 blocks at ninv, leading dimension 32) instead
// of an LDS copy: the ten 32 x 32 operand blocks are visited in the order of use and the 8 fragment loads of block
// b + 1 are issued before the 32 MFMAs of block b, so no load sits inside a dependent stage.  For callers that cannot
// afford the 80 KiB LDS copy or the barrier around staging it (prediction strips).
template <int NPJ>
__device__ __forceinline__ void tri_solve_global(WaveTile<4, NPJ> &t, const real *L, int64_t ldl, const real *ninv, int lane)
{
    // block list in order of use: s = 0: D0 | s = 1: (1,0) D1 | s = 2: (2,0) (2,1) D2 | s = 3: (3,0) (3,1) (3,2) D3
    // Addresses: a wave-uniform base (scalar registers; readfirstlane tells the compiler so) plus one 32-bit byte
    // offset per lane and operand -- written as 80 independent 64-bit lane addresses the compiler computed all of them
    // up front and spilled half the register file.
    // k index of fragment register (q, e) for this lane: 2 frag_irow(lane >> 4, q) + e = lane part + uniform part
    const uint32_t offL = (uint32_t)((2 * (lane & 15) + (int64_t)(2 * frag_irow(lane >> 4, 0)) * ldl) * (int64_t)sizeof(real));
    const uint32_t offN = (uint32_t)((2 * (lane & 15) + 32 * (2 * frag_irow(lane >> 4, 0))) * (int)sizeof(real));
    const char *Lb = uniform_ptr(reinterpret_cast<const char *>(L));
    const char *Nb = uniform_ptr(reinterpret_cast<const char *>(ninv));
    const int64_t ldu = uniform_i64(ldl);
    auto fetch = [&](real2_t (&dst)[8], int s, int pj) {        // pj == s: the inverted diagonal block
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ku = 2 * frag_irow(0, q) + e;          // uniform part of the k index
                dst[2 * q + e] =
                    (pj == s) ? *reinterpret_cast<const real2_t *>(Nb + (int64_t)(1024 * s + 32 * ku) * (int64_t)sizeof(real) + offN)
                              : *reinterpret_cast<const real2_t *>(Lb + (32 * s + (int64_t)(32 * pj + ku) * ldu) * (int64_t)sizeof(real) + offL);
            }
    };
    real2_t ring[2][8];
    fetch(ring[0], 0, 0);
    int b = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int pj = 0; pj <= s; ++pj, ++b) {
            // next block in the list
            const int ns = (pj == s) ? s + 1 : s, npj = (pj == s) ? 0 : pj + 1;
            if (ns < 4) fetch(ring[(b + 1) & 1], ns, npj);
            __builtin_amdgcn_sched_barrier(0);
            const real2_t(&a)[8] = ring[b & 1];
            if (pj < s) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int j = 0; j < 2 * NPJ; ++j) {
                            t.f[2 * s + 0][j] = mfma_real(a[2 * q + e][0], t.f[2 * pj + e][j][q], t.f[2 * s + 0][j]);
                            t.f[2 * s + 1][j] = mfma_real(a[2 * q + e][1], t.f[2 * pj + e][j][q], t.f[2 * s + 1][j]);
                        }
            } else {
                real4_t o[2][2 * NPJ];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int j = 0; j < 2 * NPJ; ++j) o[e][j] = real4_t{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int j = 0; j < 2 * NPJ; ++j) {
                            o[0][j] = mfma_real(a[2 * q + e][0], t.f[2 * s + e][j][q], o[0][j]);
                            o[1][j] = mfma_real(a[2 * q + e][1], t.f[2 * s + e][j][q], o[1][j]);
                        }
#pragma unroll
                for (int j = 0; j < 2 * NPJ; ++j) {
                    t.f[2 * s + 0][j] = o[0][j];
                    t.f[2 * s + 1][j] = o[1][j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

}  // namespace PMK_NS
}  // namespace pmk
