// Self-test kernels (include/pmk_test.h): fragment-layout checks of the fp64 MFMA tile routines and
// an fp64 MFMA issue-rate microbenchmark (the fp64 matrix peak is not in the local hardware guide;
// the roofline in bench.py is priced against the datasheet value and this measurement is reported
// beside it).
#include "../../include/pmk_test.h"
#include "pmk_mfma.h"

namespace pmk {
using namespace f64;

__global__ __launch_bounds__(64) void selftest_gemm_kernel(int K, const double *MI, const double *MJ, double *C)
{
    const int lane = threadIdx.x;
    WaveTile<4, 1> t, u, v;
    t.zero();
    u.zero();
    v.zero();
    gemm_nt<4, 1, 4>(t, MI, 128, MJ, 32, K, lane);
    gemm_nt_indexed<4, 1, 4>(u, MI, 128, MJ, 32, K, lane);       // the form the prediction kernel uses
    gemm_nt_sbase<4, 1, 4>(v, MI, 128, MJ, 32, K, lane);         // scalar bases, asm loads, explicit waits
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                const int I = 32 * (fi >> 1) + 2 * ((lane >> 4) + 4 * q) + (fi & 1);
                const int J = 32 * (fj >> 1) + 2 * (lane & 15) + (fj & 1);
                // the three loops issue the same MFMAs in the same order: any difference is a bug -> NaN
                const bool same = t.f[fi][fj][q] == u.f[fi][fj][q] && t.f[fi][fj][q] == v.f[fi][fj][q];
                C[J + 32 * I] = same ? t.f[fi][fj][q] : __builtin_nan("");
            }
}

__global__ __launch_bounds__(64) void selftest_trisolve_kernel(const double *L, const double *ninv, const double *Tin, double *Tout)
{
    const int lane = threadIdx.x;
    WaveTile<4, 1> t;
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                const int I = 32 * (fi >> 1) + 2 * ((lane >> 4) + 4 * q) + (fi & 1);
                const int J = 32 * (fj >> 1) + 2 * (lane & 15) + (fj & 1);
                t.f[fi][fj][q] = Tin[J + 32 * I];
            }
    __shared__ double tri[TRI_LDS_DOUBLES];
    stage_tri_operands(tri, L, 128, ninv, lane, 64);
    __syncthreads();
    WaveTile<4, 1> g = t;                      // the same solve with operands straight from global memory
    tri_solve_inplace<1>(t, tri, lane);
    tri_solve_global<1>(g, L, 128, ninv, lane);
#pragma unroll
    for (int fi = 0; fi < 8; ++fi)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                const int I = 32 * (fi >> 1) + 2 * ((lane >> 4) + 4 * q) + (fi & 1);
                const int J = 32 * (fj >> 1) + 2 * (lane & 15) + (fj & 1);
                // the two routines issue the same MFMAs in the same order: any difference is a bug -> NaN
                Tout[J + 32 * I] = (t.f[fi][fj][q] == g.f[fi][fj][q]) ? t.f[fi][fj][q] : __builtin_nan("");
            }
}

// 16 independent accumulators per wave, operands in registers, no memory traffic in the loop
__global__ __launch_bounds__(256) void mfma_peak_kernel(int iters, double *sink)
{
    real4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = real4_t{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)   // inline asm: the builtin form makes hipcc shuttle accumulators VGPR<->AGPR
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;   // keep the loop alive
}

}  // namespace pmk

using namespace pmk;

extern "C" {

int pmk_selftest_gemm(pmk_ctx *ctx, int K, const double *MI, const double *MJ, double *C)
{
    if (!ctx || !MI || !MJ || !C || K < 16 || K % 16) { set_error("pmk_selftest_gemm: bad argument"); return -1; }
    PMK_HIP(hipSetDevice(ctx->device));
    double *dI, *dJ, *dC;
    // gemm_nt loads (and ignores) up to 4 k-steps = 16 columns past K: pad the operands
    PMK_HIP(hipMalloc((void **)&dI, sizeof(double) * 128 * (K + 16)));
    PMK_HIP(hipMalloc((void **)&dJ, sizeof(double) * 32 * (K + 16)));
    PMK_HIP(hipMemset(dI, 0, sizeof(double) * 128 * (K + 16)));
    PMK_HIP(hipMemset(dJ, 0, sizeof(double) * 32 * (K + 16)));
    PMK_HIP(hipMalloc((void **)&dC, sizeof(double) * 128 * 32));
    PMK_HIP(hipMemcpy(dI, MI, sizeof(double) * 128 * K, hipMemcpyHostToDevice));
    PMK_HIP(hipMemcpy(dJ, MJ, sizeof(double) * 32 * K, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_gemm_kernel, dim3(1), dim3(64), 0, ctx->stream, K, dI, dJ, dC);
    PMK_HIP(hipGetLastError());
    PMK_HIP(hipStreamSynchronize(ctx->stream));
    PMK_HIP(hipMemcpy(C, dC, sizeof(double) * 128 * 32, hipMemcpyDeviceToHost));
    (void)hipFree(dI); (void)hipFree(dJ); (void)hipFree(dC);
    return 0;
}

int pmk_selftest_trisolve(pmk_ctx *ctx, const double *L, const double *ninv, const double *T_in, double *T_out)
{
    if (!ctx || !L || !ninv || !T_in || !T_out) { set_error("pmk_selftest_trisolve: bad argument"); return -1; }
    PMK_HIP(hipSetDevice(ctx->device));
    double *dL, *dN, *dI, *dO;
    PMK_HIP(hipMalloc((void **)&dL, sizeof(double) * 128 * 128));
    PMK_HIP(hipMalloc((void **)&dN, sizeof(double) * 4096));
    PMK_HIP(hipMemcpy(dN, ninv, sizeof(double) * 4096, hipMemcpyHostToDevice));
    PMK_HIP(hipMalloc((void **)&dI, sizeof(double) * 128 * 32));
    PMK_HIP(hipMalloc((void **)&dO, sizeof(double) * 128 * 32));
    PMK_HIP(hipMemcpy(dL, L, sizeof(double) * 128 * 128, hipMemcpyHostToDevice));
    PMK_HIP(hipMemcpy(dI, T_in, sizeof(double) * 128 * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(selftest_trisolve_kernel, dim3(1), dim3(64), 0, ctx->stream, dL, dN, dI, dO);
    PMK_HIP(hipGetLastError());
    PMK_HIP(hipStreamSynchronize(ctx->stream));
    PMK_HIP(hipMemcpy(T_out, dO, sizeof(double) * 128 * 32, hipMemcpyDeviceToHost));
    (void)hipFree(dL); (void)hipFree(dN); (void)hipFree(dI); (void)hipFree(dO);
    return 0;
}

int pmk_selftest_mfma_peak(pmk_ctx *ctx, double *tflops)
{
    if (!ctx || !tflops) { set_error("pmk_selftest_mfma_peak: bad argument"); return -1; }
    PMK_HIP(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    PMK_HIP(hipGetDeviceProperties(&prop, ctx->device));
    double *sink;
    PMK_HIP(hipMalloc((void **)&sink, sizeof(double)));
    const int iters = 4000;
    const int blocks = prop.multiProcessorCount * 2;
    hipEvent_t a, b;
    PMK_HIP(hipEventCreate(&a));
    PMK_HIP(hipEventCreate(&b));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, ctx->stream, 100, sink);   // warm-up
    PMK_HIP(hipEventRecord(a, ctx->stream));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, ctx->stream, iters, sink);
    PMK_HIP(hipEventRecord(b, ctx->stream));
    PMK_HIP(hipEventSynchronize(b));
    float ms = 0;
    PMK_HIP(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * (2.0 * 16 * 16 * 4);
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(sink);
    return 0;
}

}  // extern "C"
