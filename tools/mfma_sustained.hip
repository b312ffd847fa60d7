// Sustained fp64 MFMA rate of the chip under its power management: a register-resident v_mfma_f64_16x16x4_f64 loop (no
// memory traffic at all) kept running for seconds, with operands that are (a) all zero, (b) random and changing every
// instruction.  tools/mfma_sustained.sh samples rocm-smi next to it.  The 17 ms bursts of tools/fp64_peak.hip reach
// 76.5-77.8 TFLOP/s at 2.34-2.40 GHz; this asks what the part holds once the power loop has settled.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_sustained tools/mfma_sustained.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int RANDOM>
__global__ __launch_bounds__(256, 2) void mfma_loop(int iters, double *sink, unsigned long long *clk)
{
    const int lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return RANDOM ? ((double)(h >> 8) * (1.0 / 16777216.0) - 0.5) : 0.0; };
    double a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = rnd(); b[i] = rnd(); }
    double4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = double4_t{0, 0, 0, 0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 7], b[(i + j) & 7], acc[i], 0, 0, 0);
        }
        if (RANDOM) {               // keep the sums bounded (and the operands toggling)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = -a[i];
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;
    if (blockIdx.x < 8 && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    (void)lane;
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2, iters = 20000;       // ~90 ms a launch
    double *sink; unsigned long long *clk, h[16];
    hipMalloc(&sink, 8); hipMalloc(&clk, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 3; ++w) {
            if (mode) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
            else hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
        }
        hipDeviceSynchronize();
        printf("MODE %s START\n", mode ? "random" : "zeros"); fflush(stdout);
        const auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        double el = 0;
        while (el < secs) {
            for (int w = 0; w < 4; ++w) {
                if (mode) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
                else hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
            }
            hipDeviceSynchronize();
            launches += 4;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double cyc = 0, tick = 0;
        for (int x = 0; x < 8; ++x) { cyc += (double)h[2 * x]; tick += (double)h[2 * x + 1]; }
        const double flops = (double)launches * blocks * 4 * iters * 16.0 * (2.0 * 16 * 16 * 4);
        printf("MODE %s END: %.2f s, %ld launches, %.2f TFLOP/s, shader clock %.3f GHz (last launch)\n", mode ? "random" : "zeros", el,
               launches, flops / el / 1e12, cyc / (tick * 10.0));
        fflush(stdout);
    }
    return 0;
}
