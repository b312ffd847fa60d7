// Probe of the v_mfma_f64_4x4x4_4b_f64 operand / result lane layout (one-hot operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double *out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 64 + lb) * 64 + lane] = d;
        }
}
int main()
{
    double *d; hipMalloc((void **)&d, 8 * 64 * 64 * 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
    // for every A lane: the set of B lanes it pairs with and the output lanes
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[(la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    return 0;
}
