// fp64 issue-rate microbenchmarks for MI355X (gfx950): the local hardware guide has no fp64 MFMA
// numbers, so the roofline denominator is measured here.  Build: hipcc --offload-arch=gfx950 -O3
// tools/fp64_peak.hip -o tools/fp64_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma16(int iters, double *sink, unsigned long long *cyc, unsigned long long *rt)
{
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)   // inline asm: the builtin form makes hipcc shuttle accumulators VGPR<->AGPR
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

__global__ __launch_bounds__(256) void mfma4(int iters, double *sink, unsigned long long *cyc, unsigned long long *rt)
{
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456) sink[0] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

__global__ __launch_bounds__(256) void vfma(int iters, double *sink, unsigned long long *cyc, unsigned long long *rt)
{
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456) sink[0] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <typename F>
void run(const char *name, F launch, int blocks, int iters, double flops_per_wave_iter, int insts_per_iter)
{
    double *sink; unsigned long long *cyc, *rt;
    hipMalloc((void **)&sink, 8); hipMalloc((void **)&cyc, 8 * blocks); hipMalloc((void **)&rt, 8 * blocks);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(blocks, iters / 10, sink, cyc, rt);
    hipEventRecord(a);
    launch(blocks, iters, sink, cyc, rt);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> hc(blocks), hr(blocks);
    hipMemcpy(hc.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), rt, 8 * blocks, hipMemcpyDeviceToHost);
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    double c = (double)hc[blocks / 2], r = (double)hr[blocks / 2];
    double tf = (double)blocks * 4 * iters * flops_per_wave_iter / (ms * 1e-3) / 1e12;
    printf("%-28s blocks %4d  %8.3f ms  %7.2f TFLOP/s  cycles/inst/wave %.1f  clock %.2f GHz\n", name, blocks, ms, tf,
           c / ((double)iters * insts_per_iter), c / r * 0.1);
    hipFree(sink); hipFree(cyc); hipFree(rt);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    int cu = p.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        run("mfma_f64_16x16x4 1w/SIMD 16acc", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(mfma16<16>, dim3(g), dim3(256), 0, 0, it, s, c, r); }, cu, 40000, 16 * 2048.0, 16);
        run("mfma_f64_16x16x4 2w/SIMD 16acc", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(mfma16<16>, dim3(g), dim3(256), 0, 0, it, s, c, r); }, 2 * cu, 40000, 16 * 2048.0, 16);
        run("mfma_f64_16x16x4 1w/SIMD 4acc", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(mfma16<4>, dim3(g), dim3(256), 0, 0, it, s, c, r); }, cu, 160000, 4 * 2048.0, 4);
        run("mfma_f64_16x16x4 one CU", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(mfma16<16>, dim3(g), dim3(256), 0, 0, it, s, c, r); }, 1, 40000, 16 * 2048.0, 16);
        run("mfma_f64_4x4x4_4b 1w/SIMD", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(mfma4, dim3(g), dim3(256), 0, 0, it, s, c, r); }, cu, 80000, 16 * 512.0, 16);
        run("v_fma_f64 1w/SIMD", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(vfma, dim3(g), dim3(256), 0, 0, it, s, c, r); }, cu, 400000, 16 * 128.0, 16);
        run("v_fma_f64 2w/SIMD", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(vfma, dim3(g), dim3(256), 0, 0, it, s, c, r); }, 2 * cu, 400000, 16 * 128.0, 16);
        run("v_fma_f64 one CU", [](int g, int it, double *s, unsigned long long *c, unsigned long long *r) { hipLaunchKernelGGL(vfma, dim3(g), dim3(256), 0, 0, it, s, c, r); }, 1, 400000, 16 * 128.0, 16);
    }
    return 0;
}
