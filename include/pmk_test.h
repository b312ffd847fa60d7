/*
 * pmk_test.h -- self-test entry points of libpmk_hip.so (used by tests/ only).
 * They exercise the fp64 MFMA tile routines on caller-supplied data so the fragment layout
 * (cdna_hip_programming.md section 3: v_mfma_f64_16x16x4_f64 does NOT use the f32 C/D map)
 * is checked with asymmetric integer data on the real device.
 */
#ifndef PMK_TEST_H
#define PMK_TEST_H

#include <stdint.h>

#include "pmk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* C[I][J] (128 x 32, J contiguous: C[J + 32*I]) = sum_k MI[I + 128 k] MJ[J + 32 k],
 * k < K (K a multiple of 16), through the three forms of the wave-tile GEMM loop (gemm_nt, gemm_nt_indexed,
 * gemm_nt_sbase) of one wave; elements on which they disagree come back as NaN.  Host buffers. */
int pmk_selftest_gemm(pmk_ctx *ctx, int K, const double *MI, const double *MJ, double *C);
/* T (128 x 32, T[J + 32*I]) <- -L^-1 T through tri_solve_inplace (operands staged in LDS) AND tri_solve_global (operands
 * prefetched from global memory); elements on which the two disagree come back as NaN.  L is 128 x 128 column-major
 * lower, ninv its four negated inverted 32 x 32 diagonal blocks (column-major, ld 32) */
int pmk_selftest_trisolve(pmk_ctx *ctx, const double *L, const double *ninv, const double *T_in, double *T_out);
/* sustained fp64 MFMA rate of the device in TFLOP/s (register-resident loop, all CUs) */
int pmk_selftest_mfma_peak(pmk_ctx *ctx, double *tflops);

/* make pmk_query_predict_sharded run its whole request / response exchange even when the communicator has one rank
 * (the rank then sends to itself through RCCL): the only way to drive that path on a one-GPU box */
int pmk_test_comm_force_exchange(pmk_comm *comm, int on);

/* choose the factorisation path of a model by hand: 0 = one workgroup per block row (batched), 1 = split path (deep
 * products cut along K; the triangular solves as one chained launch each where the blocks nearly fit on the chip, block
 * by block otherwise), 2 = split path with the block-by-block solves, 3 = split path with the chained solves whatever
 * the size; pmk_model_create picks by itself from P and the tile counts */
int pmk_test_model_set_split(pmk_model *model, int on);

#ifdef __cplusplus
}
#endif
#endif
