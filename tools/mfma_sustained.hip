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

// The same MFMA stream with the operand traffic of the fit's wave-tile GEMM: five 16-byte loads per lane for every 16 MFMAs
// (eight doubles of the shared operand, two of the wave's own), double-buffered one step ahead, and the MFMAs consume what
// was loaded.  SRC = 0: from global memory, an 8 KiB window per workgroup that stays in the CU's L1 (what the fit's
// shared operand does); SRC = 1: the same bytes from LDS (ds_read_b128).  What the two paths cost in clock and power at an
// equal MFMA rate is the question (tools/mfma_sustained.sh).
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int SRC>
__global__ __launch_bounds__(256, 2) void mfma_operand_loop(int iters, const double *__restrict__ src, double *sink,
                                                            unsigned long long *clk)
{
    __shared__ double lds[1024];                                      // 8 KiB
    const int tid = threadIdx.x;
    const double *win = src + (size_t)blockIdx.x * 1024;
    for (int e = tid; e < 1024; e += 256) lds[e] = win[e];
    __syncthreads();
    double4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = double4_t{0, 0, 0, 0};
    auto fetch = [&](double2_t *r, int it) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int at = (2 * (tid & 63) + 128 * ((it + u + (tid >> 6)) & 7)) & 1023;       // 16-byte pieces, 1 KiB per wave and load
            if (SRC) r[u] = *reinterpret_cast<const double2_t *>(lds + at);
            else r[u] = *reinterpret_cast<const double2_t *>(win + at);
        }
    };
    double2_t cur[5], nxt[5];
    fetch(cur, 0);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        fetch(nxt, it + 1);
        const double a[8] = {cur[0][0], cur[0][1], cur[1][0], cur[1][1], cur[2][0], cur[2][1], cur[3][0], cur[3][1]};
        const double b[2] = {cur[4][0], cur[4][1]};
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 7], b[i >> 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 5; ++u) cur[u] = nxt[u];
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;
    if (blockIdx.x < 8 && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

int main(int argc, char **argv)
{
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2, iters = 20000;       // ~90 ms a launch
    double *sink; unsigned long long *clk, h[16];
    hipMalloc(&sink, 8); hipMalloc(&clk, sizeof(h));
    double *src;
    {
        const size_t n = (size_t)blocks * 1024;
        double *h = (double *)malloc(n * sizeof(double));
        unsigned r = 12345u;
        for (size_t i = 0; i < n; ++i) { r = r * 1664525u + 1013904223u; h[i] = ((double)(r >> 8) * (1.0 / 16777216.0) - 0.5) * 1e-3; }
        hipMalloc(&src, n * sizeof(double));
        hipMemcpy(src, h, n * sizeof(double), hipMemcpyHostToDevice);
        free(h);
    }
    const char *names[4] = {"zeros", "random", "random + operands from L1", "random + operands from LDS"};
    auto launch = [&](int mode) {
        if (mode == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
        else if (mode == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
        else if (mode == 2) hipLaunchKernelGGL(mfma_operand_loop<0>, dim3(blocks), dim3(256), 0, 0, 8 * iters, (const double *)src, sink, clk);
        else hipLaunchKernelGGL(mfma_operand_loop<1>, dim3(blocks), dim3(256), 0, 0, 8 * iters, (const double *)src, sink, clk);
    };
    for (int mode = 0; mode < 4; ++mode) {
        for (int w = 0; w < 3; ++w) launch(mode);
        hipDeviceSynchronize();
        printf("MODE %s START\n", names[mode]); fflush(stdout);
        const auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        double el = 0;
        while (el < secs) {
            for (int w = 0; w < 4; ++w) launch(mode);
            hipDeviceSynchronize();
            launches += 4;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double cyc = 0, tick = 0;
        for (int x = 0; x < 8; ++x) { cyc += (double)h[2 * x]; tick += (double)h[2 * x + 1]; }
        // mfma_loop: iters counts MFMA groups of 16 x 8 per 8 iterations = 16 per iteration; the operand loop: 16 per iteration, 8 x the iterations
        const double per_launch = (mode < 2 ? (double)iters : 8.0 * iters) * 16.0;
        const double flops = (double)launches * blocks * 4 * per_launch * (2.0 * 16 * 16 * 4);
        printf("MODE %s END: %.2f s, %ld launches, %.2f TFLOP/s, shader clock %.3f GHz (last launch)\n", names[mode], el,
               launches, flops / el / 1e12, cyc / (tick * 10.0));
        fflush(stdout);
    }
    return 0;
}
