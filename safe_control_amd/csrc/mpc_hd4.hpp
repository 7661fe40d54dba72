// Second-order forward-mode differentiation over four variables, in registers: value, gradient (4) and the upper triangle of the
// Hessian (10, by rows).  Used for the full-state discrete-time barriers of KinematicBicycle2D_C3BF / _DPCBF in mpc_gn.hip
// (dynamic_env/kinematic_bicycle2D_c3bf.py:83-109, kinematic_bicycle2D_dpcbf.py:91-136); oracle/mpc_kb_state.py: class HD carries the
// same arithmetic in numpy.  The barrier templates below run on `double` (line search: values only) and on `Hd4` (derivatives).
#pragma once
#include <hip/hip_runtime.h>

namespace sc {
namespace hd {

// No contraction in this header: the value part of an Hd4 expression and the same expression on doubles must round alike, because the
// interior point compares barrier values from the derivative pass (current iterate) with values from the line search (trial points).
#pragma clang fp contract(off)

struct Hd4 {
    double v, g[4], h[10];
};

__device__ __forceinline__ constexpr int tri(int a, int b) { return a * 4 - (a * (a - 1)) / 2 + (b - a); }   // a <= b

__device__ __forceinline__ Hd4 constant(double v) {
    Hd4 r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = 0.0;
#pragma unroll
    for (int t = 0; t < 10; ++t) r.h[t] = 0.0;
    return r;
}
__device__ __forceinline__ Hd4 variable(double v, int i) {
    Hd4 r = constant(v);
    r.g[i] = 1.0;
    return r;
}
// f(a) from f, f', f'' at a.v
__device__ __forceinline__ Hd4 chain(const Hd4& a, double f, double f1, double f2) {
    Hd4 r;
    r.v = f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = f1 * a.g[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) r.h[tri(i, j)] = f1 * a.h[tri(i, j)] + f2 * (a.g[i] * a.g[j]);
    return r;
}
__device__ __forceinline__ Hd4 operator+(const Hd4& a, const Hd4& b) {
    Hd4 r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] + b.g[i];
#pragma unroll
    for (int t = 0; t < 10; ++t) r.h[t] = a.h[t] + b.h[t];
    return r;
}
__device__ __forceinline__ Hd4 operator-(const Hd4& a, const Hd4& b) {
    Hd4 r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = a.g[i] - b.g[i];
#pragma unroll
    for (int t = 0; t < 10; ++t) r.h[t] = a.h[t] - b.h[t];
    return r;
}
__device__ __forceinline__ Hd4 operator-(double a, const Hd4& b) { return constant(a) - b; }
__device__ __forceinline__ Hd4 operator-(const Hd4& a, double b) { Hd4 r = a; r.v = a.v - b; return r; }
__device__ __forceinline__ Hd4 operator+(const Hd4& a, double b) { Hd4 r = a; r.v = a.v + b; return r; }
__device__ __forceinline__ Hd4 operator*(const Hd4& a, const Hd4& b) {
    Hd4 r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = a.v * b.g[i] + b.v * a.g[i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j)
            r.h[tri(i, j)] = a.v * b.h[tri(i, j)] + b.v * a.h[tri(i, j)] + (a.g[i] * b.g[j] + b.g[i] * a.g[j]);
    return r;
}
__device__ __forceinline__ Hd4 operator*(double a, const Hd4& b) {
    Hd4 r;
    r.v = a * b.v;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.g[i] = a * b.g[i];
#pragma unroll
    for (int t = 0; t < 10; ++t) r.h[t] = a * b.h[t];
    return r;
}
__device__ __forceinline__ Hd4 recip(const Hd4& a) {
    const double r = 1.0 / a.v;
    return chain(a, r, -r * r, 2.0 * r * r * r);
}
__device__ __forceinline__ Hd4 operator/(const Hd4& a, const Hd4& b) { return a * recip(b); }
__device__ __forceinline__ Hd4 div_(const Hd4& a, const Hd4& b) { return a * recip(b); }
__device__ __forceinline__ double div_(double a, double b) { return a * (1.0 / b); }          // the same two roundings as the Hd4 value
__device__ __forceinline__ Hd4 sqrt_(const Hd4& a) {
    const double r = sqrt(a.v);
    return chain(a, r, 0.5 / r, -0.25 / (r * a.v));
}
__device__ __forceinline__ double sqrt_(double a) { return sqrt(a); }
__device__ __forceinline__ void sincos_hd(const Hd4& a, Hd4& s, Hd4& c) {
    double sv, cv;
    sincos(a.v, &sv, &cv);
    s = chain(a, sv, cv, -sv);
    c = chain(a, cv, -sv, -cv);
}
__device__ __forceinline__ void sincos_hd(double a, double& s, double& c) { sincos(a, &s, &c); }
__device__ __forceinline__ double val(const Hd4& a) { return a.v; }
__device__ __forceinline__ double val(double a) { return a; }
__device__ __forceinline__ Hd4 lift(const Hd4&, double c) { return constant(c); }     // a constant of the other operand's type
__device__ __forceinline__ double lift(double, double c) { return c; }

// p_rel, v_rel of the state x = (px, py, theta, v) against an obstacle at (ox, oy).  The obstacle's velocity is zero: the reference's
// MPC hands the barrier a 1 x 7 row, whose `shape[0] > 3` test is False (oracle/mpc_kb_state.py: _rel).
template <class T>
__device__ __forceinline__ void rel(const T x[4], const double* o, T& px, T& py, T& vx, T& vy, T& pm2, T& vm) {
    px = o[0] - x[0]; py = o[1] - x[1];
    T s, c;
    sincos_hd(x[2], s, c);
    vx = 0.0 - x[3] * c; vy = 0.0 - x[3] * s;
    pm2 = px * px + py * py;
    vm = sqrt_(vx * vx + vy * vy);
}

// kinematic_bicycle2D_c3bf.py:83-109 (beta = 1.01): <p_rel, v_rel> + |p_rel| |v_rel| sqrt(max(|p_rel|^2 - ego^2, 0)) / |p_rel|
template <class T>
__device__ __forceinline__ T h_c3bf(const T x[4], const double* o, double radius) {
    T px, py, vx, vy, pm2, vm;
    rel(x, o, px, py, vx, vy, pm2, vm);
    const double ego = (o[2] + radius) * 1.01;
    const T pm = sqrt_(pm2);
    const T a = pm2 - ego * ego;
    const T root = val(a) > 0.0 ? sqrt_(a) : lift(a, 0.0);
    return px * vx + py * vy + div_(pm * vm * root, pm);
}

// kinematic_bicycle2D_dpcbf.py:91-136 (s = 1.05): line-of-sight frame, cos / sin of atan2(p_y, p_x) = p_x / |p|, p_y / |p|
template <class T>
__device__ __forceinline__ T h_dpcbf(const T x[4], const double* o, double radius) {
    T px, py, vx, vy, pm2, vm;
    rel(x, o, px, py, vx, vy, pm2, vm);
    const double s = 1.05, ego = (o[2] + radius) * s;
    const T pm = sqrt_(pm2);
    const T cr = div_(px, pm), sr = div_(py, pm);
    const T vn0 = cr * vx + sr * vy, vn1 = cr * vy - sr * vx;
    const T a = pm2 - ego * ego;
    const T dd = val(a) > 1e-6 ? a : lift(a, 1e-6);
    const double kl = 0.1 * sqrt(s * s - 1.0) / ego, km = 0.5 * sqrt(s * s - 1.0) / ego;
    const T rd = sqrt_(dd);
    return vn0 + div_(kl * rd, vm) * vn1 * vn1 + km * rd;
}

#pragma clang fp contract(fast)

}  // namespace hd
}  // namespace sc
