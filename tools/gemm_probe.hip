// Inner-loop probe for the fp64 MFMA wave-tile GEMM (pmk_mfma.h: gemm_nt) with the operand placement of the
// prediction strips: the I operand (a 128-row block row of a factor) is shared by the workgroups of an XCD and comes
// from L2, the J operand (the wave's own 32 strip columns) is private and streams from HBM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipatchmixturekriging_amd/csrc -Iinclude tools/gemm_probe.hip -o tools/gemm_probe
//   tools/gemm_probe [K=1024] [reps=16]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pmk_mfma.h"

using namespace pmk;
using namespace pmk::f64;

// pointer-stepping variant: no per-step index clamp (the ring over-reads up to PF k-steps past K: callers guarantee
// that memory exists), every I slot is refilled right behind the MFMAs that consumed it
template <int NPI, int NPJ, int PFI, int PFJ, int NACT, int TOUCH = 0, int NV = 0>
__device__ __forceinline__ void gemm_v2(WaveTile<NPI, NPJ> &t, const real *opI, int64_t ldI, const real *opJ, int64_t ldJ,
                                        int K, int lane)
{
    real2_t ra[PFI][NACT], rb[PFJ][NPJ];
    const real *qI = opI + 2 * (lane & 15) + (int64_t)(lane >> 4) * ldI;
    const real *qJ = opJ + 2 * (lane & 15) + (int64_t)(lane >> 4) * ldJ;
    const int64_t sI = 4 * ldI, sJ = 4 * ldJ;
    // TOUCH > 0: this wave's share of the I operand's cache lines TOUCH k-steps ahead is pulled into L2 by SCALAR
    // loads (lgkmcnt: they do not sit in the in-order vector-memory queue): wave w of the workgroup takes lines
    // 4w .. 4w+3 of the 32 that a k-step covers (8 waves) -- results are never used
    const real *tI = opI + (int64_t)TOUCH * sI;
    const int tw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int junk = 0;
    double dv[4] = {1.0, 2.0, 3.0, 4.0};      // NV > 0: independent fp64 FMAs between the MFMAs (does VALU work hide in the MFMA shadow?)
    const double da = 1.0000001, db = 1e-9;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < PFJ; ++s) {
        if (s < PFI) {
#pragma unroll
            for (int pi = 0; pi < NACT; ++pi) ra[s][pi] = *reinterpret_cast<const real2_t *>(qI + 32 * pi);
            qI += sI;
        }
#pragma unroll
        for (int pj = 0; pj < NPJ; ++pj) rb[s][pj] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(qJ + 32 * pj));
        qJ += sJ;
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int k0 = 0; k0 < K; k0 += 4 * PFJ) {
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
                            t.f[2 * pi + ei][2 * pj + ej] = mfma_real(ra[si][pi][ei], rb[s][pj][ej], t.f[2 * pi + ei][2 * pj + ej]);
                ra[si][pi] = *reinterpret_cast<const real2_t *>(qI + 32 * pi);
                if (pi == NACT - 1) {
#pragma unroll
                    for (int pj = 0; pj < NPJ; ++pj)
                        rb[s][pj] = __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(qJ + 32 * pj));
                }
                if (NV) {
#pragma unroll
                    for (int u = 0; u < NV / NACT; ++u)
                        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dv[u & 3]) : "v"(da), "v"(db));
                }
                if (TOUCH && pi == 0) {
                    // column (tw >> 1) of the k-step, rows 64 (tw & 1) .. +63: 4 lines of 128 B
                    const real *tp = tI + (int64_t)(tw >> 1) * ldI + 64 * (tw & 1);
#pragma unroll
                    for (int u = 0; u < 4; ++u)      // "+s": the register stays reserved while loads are in flight
                        asm volatile("s_load_dword %0, %1, %2" : "+s"(junk) : "s"(tp), "n"(128 * u));
                    tI += sI;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            qI += sI;
            qJ += sJ;
        }
    }
    if (TOUCH) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(junk));
    if (NV) t.f[0][0][0] += (dv[0] + dv[1] + dv[2] + dv[3]) * 1e-300;
}

// scalar-base form: every address of the rings is (wave-uniform base in SGPRs) + (one 32-bit byte offset per lane);
// the bases advance on the scalar ALU, so the loop body holds NO vector instruction besides the MFMAs and the loads.
// The loads are inline asm (the compiler does not emit the saddr form for this pattern), so the waits are explicit:
// vmcnt(5 (PF - 1)) in front of every k-step; the last PF k-steps run without loads and the ring is drained there.
template <int PF>
__device__ __forceinline__ void gemm_v3(WaveTile<4, 1> &t, const real *opI, int64_t ldI, const real *opJ, int64_t ldJ, int K, int lane)
{
    real2_t ra[PF][4], rb[PF];
    const uint32_t offI = (uint32_t)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldI) * 8);
    const uint32_t offJ = (uint32_t)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldJ) * 8);
    const char *bI = uniform_ptr(reinterpret_cast<const char *>(opI));
    const char *bJ = uniform_ptr(reinterpret_cast<const char *>(opJ));
    const int64_t sI = uniform_i64(4 * ldI * 8), sJ = uniform_i64(4 * ldJ * 8);
#define LDI(dst, pi) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(offI), "s"(bI), "n"(256 * (pi)))
#define LDJ(dst) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(offJ), "s"(bJ))
#define WAITV(n, s) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(ra[s][0]), "+v"(ra[s][1]), "+v"(ra[s][2]), "+v"(ra[s][3]), "+v"(rb[s]))
#pragma unroll
    for (int s = 0; s < PF; ++s) {
        LDI(ra[s][0], 0); LDI(ra[s][1], 1); LDI(ra[s][2], 2); LDI(ra[s][3], 3);
        LDJ(rb[s]);
        bI += sI; bJ += sJ;
    }
    auto mfmas = [&](int s, int pi) {
#pragma unroll
        for (int ei = 0; ei < 2; ++ei)
#pragma unroll
            for (int ej = 0; ej < 2; ++ej)
                t.f[2 * pi + ei][ej] = mfma_real(ra[s][pi][ei], rb[s][ej], t.f[2 * pi + ei][ej]);
    };
    static_assert(PF == 4, "wait counts below are written for PF = 4");
    for (int k0 = 0; k0 < K - 16; k0 += 16) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            if (s == 0) WAITV(15, 0); else if (s == 1) WAITV(15, 1); else if (s == 2) WAITV(15, 2); else WAITV(15, 3);
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) {
                mfmas(s, pi);
                __builtin_amdgcn_sched_barrier(0);
                if (pi == 0) LDI(ra[s][0], 0); else if (pi == 1) LDI(ra[s][1], 1); else if (pi == 2) LDI(ra[s][2], 2); else { LDI(ra[s][3], 3); LDJ(rb[s]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            bI += sI; bJ += sJ;
        }
    }
    // last PF k-steps: no refills; 15, 10, 5, 0 loads may still be in flight in front of each
    WAITV(15, 0);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) mfmas(0, pi);
    WAITV(10, 1);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) mfmas(1, pi);
    WAITV(5, 2);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) mfmas(2, pi);
    WAITV(0, 3);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) mfmas(3, pi);
#undef LDI
#undef LDJ
#undef WAITV
}

// buffer-load form: scalar resource descriptors + one 32-bit lane offset + a scalar k offset (soffset): no vector
// address arithmetic in the loop, and -- unlike the asm form -- the loads are builtins, so the compiler tracks their
// waits itself (safe where it spills).  Offsets are 32-bit: K * ld * 8 must stay below 2^31.
typedef int int4_v __attribute__((ext_vector_type(4)));
template <int PF>
__device__ __forceinline__ void gemm_buf(WaveTile<4, 1> &t, const real *opI, int64_t ldI, const real *opJ, int64_t ldJ, int K, int lane)
{
    real2_t ra[PF][4], rb[PF];
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void *)uniform_ptr(opI), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rJ = __builtin_amdgcn_make_buffer_rsrc((void *)uniform_ptr(opJ), 0, 0x7fffffff, 0x00020000);
    const int vI = (int)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldI) * 8);
    const int vJ = (int)((2 * (lane & 15) + (int64_t)(lane >> 4) * ldJ) * 8);
    const int sI = __builtin_amdgcn_readfirstlane((int)(4 * ldI * 8)), sJ = __builtin_amdgcn_readfirstlane((int)(4 * ldJ * 8));
    int oI = 0, oJ = 0;            // scalar byte offsets of the next k-step to load
    auto ldI4 = [&](int pi) { return __builtin_bit_cast(real2_t, __builtin_amdgcn_raw_buffer_load_b128(rI, vI + 256 * pi, oI, 0)); };
    auto ldJ1 = [&]() { return __builtin_bit_cast(real2_t, __builtin_amdgcn_raw_buffer_load_b128(rJ, vJ, oJ, 2)); };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < PF; ++s) {
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) ra[s][pi] = ldI4(pi);
        rb[s] = ldJ1();
        oI += sI; oJ += sJ;
        __builtin_amdgcn_sched_barrier(0);
    }
    const int endI = __builtin_amdgcn_readfirstlane((int)((K / 4 - 1) * (4 * ldI * 8))), endJ = __builtin_amdgcn_readfirstlane((int)((K / 4 - 1) * (4 * ldJ * 8)));
    for (int k0 = 0; k0 < K; k0 += 4 * PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) {
#pragma unroll
                for (int ei = 0; ei < 2; ++ei)
#pragma unroll
                    for (int ej = 0; ej < 2; ++ej)
                        t.f[2 * pi + ei][ej] = mfma_real(ra[s][pi][ei], rb[s][ej], t.f[2 * pi + ei][ej]);
                const int cI = oI < endI ? oI : endI;           // past the end: the last k-step again
                ra[s][pi] = __builtin_bit_cast(real2_t, __builtin_amdgcn_raw_buffer_load_b128(rI, vI + 256 * pi, cI, 0));
                if (pi == 3) {
                    const int cJ = oJ < endJ ? oJ : endJ;
                    rb[s] = __builtin_bit_cast(real2_t, __builtin_amdgcn_raw_buffer_load_b128(rJ, vJ, cJ, 2));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            oI += sI; oJ += sJ;
        }
    }
}

template <int VAR>
__global__ __launch_bounds__(512, 2) void probe(const double *__restrict__ L, int64_t ld, int64_t region_stride,
                                                const double *__restrict__ strips, int64_t strip_stride, int K, int reps,
                                                double *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *S = L + (int64_t)(blockIdx.x & 7) * region_stride;
    const double *V0 = strips + (int64_t)blockIdx.x * strip_stride + 32 * wave;
    WaveTile<4, 1> acc;
    acc.zero();
    for (int r = 0; r < reps; ++r) {
        const double *Li = S + (int64_t)(r & 15) * 128;
        const double *V = V0 + (int64_t)(r & 1) * 64 * 256;       // (a fixed J operand would be hoisted out of the loop)
        if (VAR == 0) gemm_nt<4, 1, 4, 4, 4>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 1) gemm_v2<4, 1, 4, 4, 4>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 2) gemm_v2<4, 1, 4, 8, 4>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 3) gemm_nt<4, 1, 4, 8, 4>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 4) gemm_v2<4, 1, 2, 4, 4>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 5) gemm_v2<4, 1, 4, 4, 4, 12>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 6) gemm_v2<4, 1, 4, 8, 4, 12>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 7) gemm_v2<4, 1, 4, 8, 4, 24>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 8) gemm_v2<4, 1, 4, 4, 4, 0, 16>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 9) gemm_v2<4, 1, 4, 4, 4, 0, 48>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 10) gemm_v3<4>(acc, Li, ld, V, 256, K, lane);
        if (VAR == 11) gemm_buf<4>(acc, Li, ld, V, 256, K, lane);
        __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc.f[i][j][0] + acc.f[i][j][1] + acc.f[i][j][2] + acc.f[i][j][3];
    if (s == 123.456) sink[0] = s;
}

// one wave per SIMD (512 registers a wave): 128 x 64 wave tile, deeper rings
template <int PFI, int PFJ>
__global__ __launch_bounds__(256, 1) void probe_wide(const double *__restrict__ L, int64_t ld, int64_t region_stride,
                                                     const double *__restrict__ strips, int64_t strip_stride, int K, int reps,
                                                     double *__restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *S = L + (int64_t)(blockIdx.x & 7) * region_stride;
    const double *V0 = strips + (int64_t)blockIdx.x * strip_stride + 64 * wave;
    WaveTile<4, 2> acc;
    acc.zero();
    for (int r = 0; r < reps; ++r) {
        const double *Li = S + (int64_t)(r & 15) * 128;
        const double *V = V0 + (int64_t)(r & 1) * 64 * 256;
        gemm_nt<4, 2, PFI, PFJ, 4>(acc, Li, ld, V, 256, K, lane);
        __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc.f[i][j][0] + acc.f[i][j][1] + acc.f[i][j][2] + acc.f[i][j][3];
    if (s == 123.456) sink[0] = s;
}

template <int PFI, int PFJ>
static void run_wide(const char *name, const double *L, int64_t ld, int64_t rs, const double *strips, int64_t ss, int K, int reps,
                     double *sink, int nwg)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((probe_wide<PFI, PFJ>), dim3(nwg), dim3(256), 0, 0, L, ld, rs, strips, ss, K, reps, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it > 0 && ms < best) best = ms;
    }
    const double flops = (double)nwg * 4 * reps * (K / 4) * 32.0 * 2048.0;
    printf("%-44s 4 waves/WG  %8.3f ms  %6.2f TFLOP/s  (%.1f %% of 78.6)\n", name, best, flops / best / 1e9, flops / best / 1e9 / 78.6 * 100);
}

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
    } while (0)

template <int VAR>
static void run(const char *name, int threads, const double *L, int64_t ld, int64_t rs, const double *strips, int64_t ss, int K,
                int reps, double *sink, int nwg)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(probe<VAR>, dim3(nwg), dim3(threads), 0, 0, L, ld, rs, strips, ss, K, reps, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    const double flops = (double)nwg * (threads / 64) * reps * (K / 4) * 16.0 * 2048.0;
    printf("%-44s %d waves/WG  %8.3f ms  %6.2f TFLOP/s  (%.1f %% of 78.6)\n", name, threads / 64, best, flops / best / 1e9,
           flops / best / 1e9 / 78.6 * 100);
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 1024;
    if (K < 64 || K > 1920 || K % 32) { printf("K must be a multiple of 32 in [64, 1920]\n"); return 1; }
    const int reps = argc > 2 ? atoi(argv[2]) : 16;
    const int64_t ld = 2048, rs = ld * 2048;
    const int nwg = 256;
    double *L, *strips, *sink;
    const int64_t ss = 2048 * 256;
    CK(hipMalloc(&L, sizeof(double) * rs * 8));
    const int64_t slack = 512 * 256;          // the rings over-read up to 8 k-steps past K; V is offset by up to 64 rows
    CK(hipMalloc(&strips, sizeof(double) * (ss * nwg + slack)));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(L, 0, sizeof(double) * rs * 8));
    CK(hipMemset(strips, 0, sizeof(double) * (ss * nwg + slack)));
    printf("K = %d, %d block rows per workgroup, 256 workgroups\n", K, reps);
    for (int threads : {512, 256}) {
        run<0>("shipped gemm_nt PFI 4 PFJ 4", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<3>("shipped gemm_nt PFI 4 PFJ 8", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<1>("pointer stepping, spread loads PFI 4 PFJ 4", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<2>("pointer stepping, spread loads PFI 4 PFJ 8", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<4>("pointer stepping, spread loads PFI 2 PFJ 4", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<5>("pointer stepping PFI 4 PFJ 4, L2 touch 12 ahead", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<6>("pointer stepping PFI 4 PFJ 8, L2 touch 12 ahead", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<7>("pointer stepping PFI 4 PFJ 8, L2 touch 24 ahead", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<8>("pointer stepping 4/4 + 16 fp64 FMAs per k-step", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<9>("pointer stepping 4/4 + 48 fp64 FMAs per k-step", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<10>("scalar bases, asm loads, explicit waits", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
        run<11>("buffer loads: scalar descriptor + soffset", threads, L, ld, rs, strips, ss, K, reps, sink, nwg);
    }
    run_wide<4, 4>("128 x 64 tile, 1 wave/SIMD, PFI 4 PFJ 4", L, ld, rs, strips, ss, K, reps, sink, nwg);
    run_wide<8, 8>("128 x 64 tile, 1 wave/SIMD, PFI 8 PFJ 8", L, ld, rs, strips, ss, K, reps, sink, nwg);
    run_wide<4, 8>("128 x 64 tile, 1 wave/SIMD, PFI 4 PFJ 8", L, ld, rs, strips, ss, K, reps, sink, nwg);
    return 0;
}
