// Stand-ins for every device-side launcher of libpmk_hip.so, for the HOST sanitizer build only (make asan): the host
// code of the C ABI (pmk_api.cpp, pmk_comm.cpp, pmk_bsp.cpp) is compiled with g++ -fsanitize=address,undefined and linked
// against these instead of the .hip objects, so that the CPU test-suite can run the argument checks, the exact host
// BSP, the sharding arithmetic and every error path under ASan / UBSan in a container without a GPU.  Never part of
// libpmk_hip.so: a product call that reaches one of these fails loudly.
#include "pmk_internal.h"

namespace pmk {

static int no_gpu(const char *what)
{
    set_error("%s: this is the host sanitizer build (no device code)", what);
    return -100;
}

#define PMK_STUB_REAL(NS)                                                                                          \
    namespace NS {                                                                                                  \
    int launch_kernel_matrix_slabs(const pmk_model *, const pmk_kernel_desc &, double, hipStream_t, int64_t, int64_t, int) { return no_gpu("launch_kernel_matrix_slabs"); } \
    int launch_cholesky(pmk_model *, hipStream_t, int64_t, int64_t, const hipEvent_t *, int) { return no_gpu("launch_cholesky"); } \
    int launch_backsolve(pmk_model *, hipStream_t, int64_t, int64_t) { return no_gpu("launch_backsolve"); }       \
    int launch_ninv_from_slabs(pmk_model *, hipStream_t) { return no_gpu("launch_ninv_from_slabs"); }             \
    int set_device_attributes() { return 0; }                                                                       \
    int build_strip_tasks(pmk_query *, hipStream_t) { return no_gpu("build_strip_tasks"); }                        \
    int launch_items(pmk_query *, const pmk_kernel_desc &, hipStream_t) { return no_gpu("launch_items"); }         \
    }
PMK_STUB_REAL(f64)
PMK_STUB_REAL(f32)

int launch_kernel_matrix_dense(const pmk_kernel_desc &, int, int64_t, const double *, int64_t, int64_t, const double *, int64_t,
                               double *, int64_t, bool, hipStream_t) { return no_gpu("launch_kernel_matrix_dense"); }
int set_plan_attributes() { return 0; }
int launch_iota(int32_t *, int64_t, hipStream_t) { return no_gpu("launch_iota"); }
int launch_plan_count(pmk_query *, double, double, hipStream_t) { return no_gpu("launch_plan_count"); }
int launch_plan_fill(pmk_query *, double, double, hipStream_t) { return no_gpu("launch_plan_fill"); }
int launch_sort_items(pmk_query *, hipStream_t) { return no_gpu("launch_sort_items"); }
int launch_mix(pmk_query *, const pmk_kernel_desc &, int64_t, int64_t, hipStream_t) { return no_gpu("launch_mix"); }
int launch_export_requests(pmk_query *, int64_t, int64_t, double *, int32_t *, hipStream_t) { return no_gpu("launch_export_requests"); }
int launch_export_results(pmk_query *, double *, double *, hipStream_t) { return no_gpu("launch_export_results"); }
int launch_explicit_items(pmk_query *, int *, hipStream_t) { return no_gpu("launch_explicit_items"); }
int launch_query_mean(const pmk_kernel_desc *, int, int, int64_t, const double *, int64_t, const double *, int64_t, const double *,
                      double *, hipStream_t) { return no_gpu("launch_query_mean"); }
int64_t exclusive_scan_i32_to_i64(const int32_t *, int64_t *, int64_t, void **, size_t *, hipStream_t) { return no_gpu("exclusive_scan"); }
int bsp_build_device(pmk_ctx *, int, int64_t, const double *, int, int, int, BspArrays &) { return no_gpu("bsp_build_device"); }
int bsp_assign_device(pmk_ctx *, const BspArrays &, int64_t, const double *, double, int64_t *, int64_t *, int64_t *, int64_t *)
{
    return no_gpu("bsp_assign_device");
}

}  // namespace pmk

extern "C" {
int pmk_selftest_gemm(pmk_ctx *, int, const double *, const double *, double *) { return pmk::no_gpu("pmk_selftest_gemm"); }
int pmk_selftest_trisolve(pmk_ctx *, const double *, const double *, const double *, double *) { return pmk::no_gpu("pmk_selftest_trisolve"); }
int pmk_selftest_mfma_peak(pmk_ctx *, double *) { return pmk::no_gpu("pmk_selftest_mfma_peak"); }
}
