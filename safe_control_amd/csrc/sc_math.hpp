// Device math helpers shared by the CBF-QP and MPC-CBF kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

namespace sc {

template <typename T> struct num;
template <> struct num<float> {
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
    static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
    static __device__ __forceinline__ float eps_par() { return 1e-5f; }    // |sin| below which two rows count as parallel
    static __device__ __forceinline__ float tol_feas() { return 1e-5f; }   // relative infeasibility tolerance
};
template <> struct num<double> {
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
    static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
    static __device__ __forceinline__ double eps_par() { return 1e-12; }
    static __device__ __forceinline__ double tol_feas() { return 1e-9; }
};

__device__ __forceinline__ void sincos_(float x, float* s, float* c) { sincosf(x, s, c); }
// f64 sincos for arguments of ordinary size (headings): three-term Cody-Waite reduction by pi/2 with FMAs (exact products,
// good to an ulp for |x| < 1e5) and the fdlibm kernels on [-pi/4, pi/4] -- about 40 instructions in one basic block where
// the library routine carries the Payne-Hanek path and its branches; larger arguments go to the library.  Max error
// against the library over [-1e5, 1e5]: 1 ulp (tests/test_cbfqp_gpu.py pins the kernels that use it on the oracle).
__device__ __forceinline__ void sincos_(double x, double* s, double* c) {
    if (!(fabs(x) < 1.0e5)) { sincos(x, s, c); return; }
    const double k = rint(x * 0.63661977236758138);                     // 2 / pi
    double r = __builtin_fma(-k, 1.5707963267948966, x);
    r = __builtin_fma(-k, 6.123233995736766e-17, r);
    r = __builtin_fma(-k, -1.4973849048591698e-33, r);
    const double z = r * r;
    // __kernel_sin / __kernel_cos (fdlibm k_sin.c, k_cos.c)
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    const double ks = __builtin_fma(z * r, __builtin_fma(z, ps, -1.66666666666666324348e-01), r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double kc = w + (((1.0 - w) - hz) + z * (z * pc));
    const int n = (int)k & 3;
    const double sv = (n & 1) ? kc : ks, cv = (n & 1) ? ks : kc;
    *s = (n & 2) ? -sv : sv;
    *c = ((n + 1) & 2) ? -cv : cv;
}
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float fabs_(float x) { return fabsf(x); }
__device__ __forceinline__ double fabs_(double x) { return fabs(x); }
__device__ __forceinline__ float fmax_(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double fmax_(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float fmin_(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double fmin_(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float pow_(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double pow_(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float rint_(float a) { return rintf(a); }
__device__ __forceinline__ double rint_(double a) { return rint(a); }
__device__ __forceinline__ float atan2_(float a, float b) { return atan2f(a, b); }
__device__ __forceinline__ double atan2_(double a, double b) { return atan2(a, b); }
__device__ __forceinline__ float floor_(float a) { return floorf(a); }
__device__ __forceinline__ double floor_(double a) { return floor(a); }
__device__ __forceinline__ bool finite_(float a) { return __builtin_isfinite(a); }
__device__ __forceinline__ bool finite_(double a) { return __builtin_isfinite(a); }

// x**e with numpy/C pow semantics (signed base allowed for integer-valued e,
// robots/dynamic_unicycle2D.py:159-183).  Small integer exponents -- the only
// ones the reference's examples use (4, 6, 10) -- take a multiply chain.
template <typename T>
__device__ __forceinline__ T powi_chain(T x, int n) {     // n >= 0
    T r = T(1), b = x;
    while (n) {
        if (n & 1) r *= b;
        b *= b;
        n >>= 1;
    }
    return r;
}

__device__ __forceinline__ float fmod_(float a, float b) { return fmodf(a, b); }
__device__ __forceinline__ double fmod_(double a, double b) { return fmod(a, b); }

// Python float modulo (CPython float_rem): exact fmod, then the sign of the divisor (m > 0).
template <typename T>
__device__ __forceinline__ T pymod(T x, T m) {
    T r = fmod_(x, m);
    if (r < T(0)) r += m;
    return r;
}

template <typename T>
__device__ __forceinline__ T angle_normalize(T x) {       // robots/dynamic_unicycle2D.py:13-16
    const T pi = T(3.14159265358979323846);
    return pymod(x + pi, T(2) * pi) - pi;
}

}  // namespace sc
