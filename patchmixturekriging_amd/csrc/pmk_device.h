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
// t^6 = (t^3)^2 with t^3 carried as an unevaluated sum t3 + e3 (error-free products on FMA): 11 operations,
// result within the rounding of the true power in all but vanishingly rare cases.
__device__ __forceinline__ double pow6_cr(double t)
{
    const double t2 = t * t;
    const double e2 = __builtin_fma(t, t, -t2);
    const double t3 = t2 * t;
    const double e3 = __builtin_fma(e2, t, __builtin_fma(t2, t, -t3));
    const double t6 = t3 * t3;
    const double e6 = __builtin_fma(2.0 * t3, e3, __builtin_fma(t3, t3, -t6));
    return t6 + e6;
}

// sqrt for the distance of two points: x >= 0 and far from the subnormal / overflow ranges, so the range
// scaling of the library sqrt is dropped: v_rsq_f64 seed, one Goldschmidt step, two FMA corrections (the
// library's own tail) -> correctly rounded; 0 maps to 0 through the select.
__device__ __forceinline__ double sqrt_dist(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x == 0.0 || x > 1e300) ? x : g;
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
// unrolled tile loops carry one straight-line formula instead of the whole switch).  R = double is the
// parity path (operation order of the reference, correctly rounded tmp^6 and /3); R = float is the fp32
// configuration, which has no reference semantics (the reference is Float64-only) and uses plain float ops.
template <int FAM = 0, typename R = double>
__device__ __forceinline__ R profile(const pmk_kernel_desc &th, R tau)
{
    constexpr bool F64 = sizeof(R) == 8;
    const R p0 = (R)th.p[0], p1 = (R)th.p[1];
    switch (FAM ? FAM : th.family) {
    case PMK_SPLINE34: {
        R r = tau * p0;
        R t = (R)1 - r;
        // branch-free form of `if sign(tmp) < 0 return 0`: outside the support evaluate at r = 1, t = 0,
        // which gives exactly +0.0 (selects, so the unrolled tile loops stay straight-line code)
        const bool outside = t < (R)0;
        r = outside ? (R)1 : r;
        t = outside ? (R)0 : t;
        if constexpr (F64) {
            const double t6 = pow6_cr(t);                    // tmp^6
            return div3_cr(((35.0 * (r * r) + 18.0 * r) + 3.0) * t6);
        } else {
            const R t2 = t * t;
            return ((((R)35 * (r * r) + (R)18 * r) + (R)3) * (t2 * t2 * t2)) * (R)(1.0 / 3.0);
        }
    }
    case PMK_SPLINE12: {
        R r = tau * p0;
        R t = (R)1 - r;
        if (t < (R)0) return (R)0;
        return ((R)3 * r + (R)1) * (t * t * t);
    }
    case PMK_SPLINE32: {
        R r = tau * p0;
        R t = (R)1 - r;
        if (t < (R)0) return (R)0;
        if constexpr (F64) return (4.0 * r + 1.0) * pow4_cr(t);               // tmp^4
        else { const R t2 = t * t; return ((R)4 * r + (R)1) * (t2 * t2); }
    }
    case PMK_GAUSSIAN:
        return exp((-p0) * (tau * tau));
    case PMK_RQ: {
        R s = sqrt(p0 + tau * tau);
        R sa = sqrt(p0);
        return (sa * sa * sa) / (s * s * s);
    }
    case PMK_TRQ: {
        R s = sqrt(p0 + tau * tau);
        R sa = sqrt(p0);
        return (p1 * (sa * sa * sa)) / (s * s * s);
    }
    case PMK_MODSQEXP:
        return exp((-p0) * (tau * tau)) * cos(p1 * tau);
    default:
        return (R)__builtin_nan("");
    }
}

// Brownian-bridge scalar kernels: kernel.jl:156-158, :218-225, :168-174, :176-193, :256-263
template <typename R = double>
__device__ __forceinline__ R bb_scalar(const pmk_kernel_desc &th, R x, R z)
{
    if (th.flags & PMK_FLAG_SEMIINF) {
        x = x / ((R)2 * ((R)1 + x));
        z = z / ((R)2 * ((R)1 + z));
    }
    switch (th.family) {
    case PMK_BB10:
        return fmin(x, z) - x * z;
    case PMK_BB20: {
        const R m16 = (R)(-1.0 / 6.0);
        if (z < x) return ((m16 * z) * ((R)1 - x)) * ((x * x + z * z) - (R)2 * x);
        return ((m16 * x) * ((R)1 - z)) * ((x * x + z * z) - (R)2 * z);
    }
    case PMK_BB1EPS: {
        R e = (R)th.p[0];
        R den = e * sinh(e);
        R num = sinh(e * fmin(x, z)) * sinh(e * ((R)1 - fmax(x, z)));
        return num / den;
    }
    case PMK_BB2EPS: {
        R e = (R)th.p[0];
        R s = x + z;
        R mn = fmin(x, z), mx = fmax(x, z), ad = fabs(x - z);
        R num = exp((-e) * s);
        R em1 = exp((R)2 * e) - (R)1;
        R den = ((R)4 * (e * e * e)) * (em1 * em1);
        R mult = num / den;
        R t1 = exp((R)2 * e) * (((R)2 * e - e * s) - (R)1);
        R t2 = exp((R)4 * e) * (e * s + (R)1);
        R t3 = exp(((R)2 * e) * (((R)1 + x) + z)) * (((R)2 * e - e * s) + (R)1);
        R t4 = exp(((R)2 * e) * s) * (e * s - (R)1);
        R t5 = exp(((R)2 * e) * ((R)2 + mn)) * ((-e) * ad - (R)1);
        R t6 = exp(((R)2 * e) * mx) * ((-e) * ad + (R)1);
        R t7 = exp(((R)2 * e) * ((R)1 + mn)) * (((R)1 - (R)2 * e) + e * ad);
        R t8 = exp(((R)2 * e) * ((R)1 + mx)) * (((R)1 + (R)2 * e) - e * ad);
        return mult * (((((((t1 + t2) + t3) + t4) + t5) + t6) + t7) + t8);
    }
    default:
        return (R)__builtin_nan("");
    }
}

// evalkernel(p, q, theta) with p, q in registers.  Stationary: tau = norm(p-q) as a sequential
// sum of squares and one sqrt (kernel.jl:277-287); Brownian bridge: product over dimensions
// (kernel.jl:196-206).
template <int D, int FAM = 0, typename R = double>
__device__ __forceinline__ R kern_eval(const pmk_kernel_desc &th, const R *p, const R *q)
{
    const int fam = FAM ? FAM : th.family;
    if (fam >= PMK_BB10) {
        R out = bb_scalar<R>(th, p[0], q[0]);
#pragma unroll
        for (int d = 1; d < D; ++d) out = out * bb_scalar<R>(th, p[d], q[d]);
        return out;
    }
    if (fam == PMK_MODSQEXP && D > 1) return (R)__builtin_nan("");
    R r0 = p[0] - q[0];
    R s = r0 * r0;
#pragma unroll
    for (int d = 1; d < D; ++d) {
        R r = p[d] - q[d];
        s = s + r * r;
    }
    if constexpr (sizeof(R) == 8) return profile<FAM, R>(th, sqrt_dist(s));
    else return profile<FAM, R>(th, sqrt(s));
}

// dot(u, x) as the reference's short ddot.  mode 0: sequential multiply then add, no FMA; mode 1: a chain of
// fused multiply-adds (what a BLAS built with contraction runs) -- see dot_seq in pmk_bsp.cpp
template <int D>
__device__ __forceinline__ double dot_seq(const double *u, const double *x, int mode)
{
    double s = u[0] * x[0];
    if (mode) {
#pragma unroll
        for (int d = 1; d < D; ++d) s = __builtin_fma(u[d], x[d], s);
    } else {
#pragma unroll
        for (int d = 1; d < D; ++d) s = s + u[d] * x[d];
    }
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
}  // namespace pmk
