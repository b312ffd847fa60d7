// Device-side kernel functions (evalkernel) and the fp64 MFMA tile helpers.
#pragma once

#include <hip/hip_runtime.h>

#include "pmk_internal.h"

namespace pmk {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

// Everything in this header that feeds comparisons or is compared against the CPU oracle is
// evaluated without FMA contraction, in the operation order of the reference's Julia code.
#pragma clang fp contract(off)

// t^n for the spline profiles.  The reference evaluates `tmp^6` / `tmp^4` with Julia's pow (<= 1 ulp); a
// plain product chain t2*t2*t2 accumulates 3 roundings, which (1-r)^6 then carries into the result.
// Double-double products (error-free transformations on FMA) give the correctly rounded power.
struct dd_t { double hi, lo; };
__device__ __forceinline__ dd_t dd_sqr(dd_t a)
{
    dd_t r;
    r.hi = a.hi * a.hi;
    r.lo = __builtin_fma(a.hi, a.hi, -r.hi) + 2.0 * (a.hi * a.lo);
    const double s = r.hi + r.lo;
    r.lo = r.lo - (s - r.hi);
    r.hi = s;
    return r;
}
__device__ __forceinline__ dd_t dd_mul(dd_t a, dd_t b)
{
    dd_t r;
    r.hi = a.hi * b.hi;
    r.lo = __builtin_fma(a.hi, b.hi, -r.hi) + (a.hi * b.lo + a.lo * b.hi);
    const double s = r.hi + r.lo;
    r.lo = r.lo - (s - r.hi);
    r.hi = s;
    return r;
}
__device__ __forceinline__ double pow4_cr(double t)
{
    const dd_t t4 = dd_sqr(dd_sqr(dd_t{t, 0.0}));
    return t4.hi + t4.lo;
}
__device__ __forceinline__ double pow6_cr(double t)
{
    const dd_t t2 = dd_sqr(dd_t{t, 0.0});
    const dd_t t6 = dd_mul(dd_sqr(t2), t2);
    return t6.hi + t6.lo;
}

// x / 3.0, correctly rounded, without the ~30-instruction IEEE division sequence: q = RN(x * RN(1/3)),
// the residual r = x - 3 q is exact in an FMA, and RN(q + r * RN(1/3)) is the correctly rounded quotient
// (Markstein's division-by-constant correction; 3 has no all-ones significand).
__device__ __forceinline__ double div3_cr(double x)
{
    const double y = 1.0 / 3.0;
    const double q = x * y;
    const double r = __builtin_fma(-3.0, q, x);
    return __builtin_fma(r, y, q);
}

// evalkernel(tau, theta): src/RKHS/kernel.jl:299-381 of the reference.
// FAM != 0 fixes the family at compile time (the hot kernels are instantiated for Spline34 so the
// unrolled tile loops carry one straight-line formula instead of the whole switch).
template <int FAM = 0>
__device__ __forceinline__ double profile(const pmk_kernel_desc &th, double tau)
{
    switch (FAM ? FAM : th.family) {
    case PMK_SPLINE34: {
        double r = tau * th.p[0];
        double t = 1.0 - r;
        // branch-free form of `if sign(tmp) < 0 return 0`: outside the support evaluate at r = 1, t = 0,
        // which gives exactly +0.0 (selects, so the unrolled tile loops stay straight-line code)
        const bool outside = t < 0.0;
        r = outside ? 1.0 : r;
        t = outside ? 0.0 : t;
        const double t6 = pow6_cr(t);                        // tmp^6
        return div3_cr(((35.0 * (r * r) + 18.0 * r) + 3.0) * t6);
    }
    case PMK_SPLINE12: {
        double r = tau * th.p[0];
        double t = 1.0 - r;
        if (t < 0.0) return 0.0;
        return (3.0 * r + 1.0) * (t * t * t);
    }
    case PMK_SPLINE32: {
        double r = tau * th.p[0];
        double t = 1.0 - r;
        if (t < 0.0) return 0.0;
        return (4.0 * r + 1.0) * pow4_cr(t);               // tmp^4
    }
    case PMK_GAUSSIAN:
        return exp((-th.p[0]) * (tau * tau));
    case PMK_RQ: {
        double s = sqrt(th.p[0] + tau * tau);
        double sa = sqrt(th.p[0]);
        return (sa * sa * sa) / (s * s * s);
    }
    case PMK_TRQ: {
        double s = sqrt(th.p[0] + tau * tau);
        double sa = sqrt(th.p[0]);
        return (th.p[1] * (sa * sa * sa)) / (s * s * s);
    }
    case PMK_MODSQEXP:
        return exp((-th.p[0]) * (tau * tau)) * cos(th.p[1] * tau);
    default:
        return __builtin_nan("");
    }
}

// Brownian-bridge scalar kernels: kernel.jl:156-158, :218-225, :168-174, :176-193, :256-263
__device__ __forceinline__ double bb_scalar(const pmk_kernel_desc &th, double x, double z)
{
    if (th.flags & PMK_FLAG_SEMIINF) {
        x = x / (2.0 * (1.0 + x));
        z = z / (2.0 * (1.0 + z));
    }
    switch (th.family) {
    case PMK_BB10:
        return fmin(x, z) - x * z;
    case PMK_BB20: {
        const double m16 = -1.0 / 6.0;
        if (z < x) return ((m16 * z) * (1.0 - x)) * ((x * x + z * z) - 2.0 * x);
        return ((m16 * x) * (1.0 - z)) * ((x * x + z * z) - 2.0 * z);
    }
    case PMK_BB1EPS: {
        double e = th.p[0];
        double den = e * sinh(e);
        double num = sinh(e * fmin(x, z)) * sinh(e * (1.0 - fmax(x, z)));
        return num / den;
    }
    case PMK_BB2EPS: {
        double e = th.p[0];
        double s = x + z;
        double mn = fmin(x, z), mx = fmax(x, z), ad = fabs(x - z);
        double num = exp((-e) * s);
        double em1 = exp(2.0 * e) - 1.0;
        double den = (4.0 * (e * e * e)) * (em1 * em1);
        double mult = num / den;
        double t1 = exp(2.0 * e) * ((2.0 * e - e * s) - 1.0);
        double t2 = exp(4.0 * e) * (e * s + 1.0);
        double t3 = exp((2.0 * e) * ((1.0 + x) + z)) * ((2.0 * e - e * s) + 1.0);
        double t4 = exp((2.0 * e) * s) * (e * s - 1.0);
        double t5 = exp((2.0 * e) * (2.0 + mn)) * ((-e) * ad - 1.0);
        double t6 = exp((2.0 * e) * mx) * ((-e) * ad + 1.0);
        double t7 = exp((2.0 * e) * (1.0 + mn)) * ((1.0 - 2.0 * e) + e * ad);
        double t8 = exp((2.0 * e) * (1.0 + mx)) * ((1.0 + 2.0 * e) - e * ad);
        return mult * (((((((t1 + t2) + t3) + t4) + t5) + t6) + t7) + t8);
    }
    default:
        return __builtin_nan("");
    }
}

// evalkernel(p, q, theta) with p, q in registers.  Stationary: tau = norm(p-q) as a sequential
// sum of squares and one sqrt (kernel.jl:277-287); Brownian bridge: product over dimensions
// (kernel.jl:196-206).
template <int D, int FAM = 0>
__device__ __forceinline__ double kern_eval(const pmk_kernel_desc &th, const double *p, const double *q)
{
    const int fam = FAM ? FAM : th.family;
    if (fam >= PMK_BB10) {
        double out = bb_scalar(th, p[0], q[0]);
#pragma unroll
        for (int d = 1; d < D; ++d) out = out * bb_scalar(th, p[d], q[d]);
        return out;
    }
    if (fam == PMK_MODSQEXP && D > 1) return __builtin_nan("");
    double r0 = p[0] - q[0];
    double s = r0 * r0;
#pragma unroll
    for (int d = 1; d < D; ++d) {
        double r = p[d] - q[d];
        s = s + r * r;
    }
    return profile<FAM>(th, sqrt(s));
}

// dot(u, x) as the reference's short ddot: sequential multiply-add, no FMA
template <int D>
__device__ __forceinline__ double dot_seq(const double *u, const double *x)
{
    double s = u[0] * x[0];
#pragma unroll
    for (int d = 1; d < D; ++d) s = s + u[d] * x[d];
    return s;
}

#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------
// fp64 MFMA tile algebra.
//
// v_mfma_f64_16x16x4_f64: D[i][j] += sum_k A[i][k] B[k][j]; lane l supplies A[i = l&15][k = l>>4]
// and B[k = l>>4][j = l&15]; it holds D[i = (l>>4) + 4q][j = l&15] in register q (q = 0..3).
//
// A wave tile is indexed by a "lane dimension" J (the MFMA j index: contiguous in memory for
// every operand and result we touch) and a "register dimension" I (the MFMA i index).
// Fragments come in pairs that cover 32 consecutive indices: fragment (p, e) holds index
// 32 p + 2 rho + e at MFMA index rho, so one 16-byte load fetches the operands of both
// fragments of a pair and one 16-byte store writes two adjacent results.
//
// For lane l, fragment (pi, ei | pj, ej), register q:
//     J index = 32 pj + 2 (l & 15) + ej
//     I index = 32 pi + 2 ((l >> 4) + 4 q) + ei
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double4_t mfma64(double a_i, double b_j, double4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a_i, b_j, c, 0, 0, 0);
}

// operand pair load: base points at element (index0, k0) of a column-major matrix whose rows
// are the fragment index; lane reads indices index0 + 2 rho, +1 at column k0 + (l >> 4)
__device__ __forceinline__ double2_t load_pair(const double *base, int64_t ld, int lane)
{
    return *reinterpret_cast<const double2_t *>(base + 2 * (lane & 15) + (int64_t)(lane >> 4) * ld);
}

}  // namespace pmk
