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
__device__ __forceinline__ void sincos_(double x, double* s, double* c) { sincos(x, s, c); }
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
