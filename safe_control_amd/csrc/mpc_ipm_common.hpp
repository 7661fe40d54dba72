// Pieces shared by the one-NLP-per-wavefront interior-point kernels mpc_lin.hip and mpc_gn.hip: wave reductions, the planar
// distance barrier, the LDS Cholesky for run-time orders (blocked, MFMA trailing update), the triangular solves and the
// out-of-line wrapper of the register Cholesky (mpc_chol.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "mpc_chol.hpp"

namespace sc {
namespace ipm {

// Wave reductions on DPP: a __shfl_xor butterfly is six dependent ds_bpermute round trips per 32-bit half (~100 cycles
// each for a lone wave) and an interior-point iteration does about twenty reductions.  Four DPP moves (xor 1, xor 2,
// half-row mirror, row mirror: VALU latency) leave every lane with its 16-lane row total; the four row totals are
// combined from v_readlane, so the result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_mv(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_value(double v, int src) {          // src: compile-time constant lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
template <typename F>
__device__ __forceinline__ double wred(double v, F f) {
    v = f(v, dpp_mv<0xB1>(v));           // quad_perm [1,0,3,2]
    v = f(v, dpp_mv<0x4E>(v));           // quad_perm [2,3,0,1]
    v = f(v, dpp_mv<0x141>(v));          // row_half_mirror
    v = f(v, dpp_mv<0x140>(v));          // row_mirror
    return f(f(row_value(v, 0), row_value(v, 16)), f(row_value(v, 32), row_value(v, 48)));
}
__device__ __forceinline__ double wsum(double v) { return wred(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wmin(double v) { return wred(v, [](double a, double b) { return fmin(a, b); }); }
__device__ __forceinline__ double wmax(double v) { return wred(v, [](double a, double b) { return fmax(a, b); }); }


// h, dh/dp, d2h/dp2 at a planar point: circle (every model) or superellipsoid (single_integrator2D.py:162-181: fabs,
// a, b >= 1e-3, e >= 2); oracle/mpc_cbf.py: barrier
__device__ inline void ipm_barrier(double px_, double py_, const double* o, double Rrob, double beta, bool circles_only, bool derivs,
                                   double& h, double& d0, double& d1, double& hxx, double& hxy, double& hyy) {
    if (circles_only || o[6] < 0.5) {
        const double d = Rrob + o[2];
        const double ex = px_ - o[0], ey = py_ - o[1];
        h = (ex * ex + ey * ey) - beta * d * d;
        d0 = 2.0 * ex; d1 = 2.0 * ey; hxx = 2.0; hxy = 0.0; hyy = 2.0;
        return;
    }
    const double a = fmax(fabs(o[2]), 1e-3) + Rrob, b = fmax(fabs(o[3]), 1e-3) + Rrob;
    const double e = fmax(fabs(o[4]), 2.0);
    double st, ct;
    sincos(o[5], &st, &ct);
    const double dx = px_ - o[0], dy = py_ - o[1];
    const double px = ct * dx + st * dy, py = -st * dx + ct * dy;
    const double ax = fabs(px) / a, ay = fabs(py) / b;
    h = pow(ax, e) + pow(ay, e) - 1.0;
    if (!derivs) { d0 = d1 = hxx = hxy = hyy = 0.0; return; }
    const double sx = px > 0 ? 1.0 : (px < 0 ? -1.0 : 0.0), sy = py > 0 ? 1.0 : (py < 0 ? -1.0 : 0.0);
    const double gpx = e * pow(ax, e - 1) / a * sx, gpy = e * pow(ay, e - 1) / b * sy;
    const double cxx = e * (e - 1) * pow(ax, e - 2) / (a * a), cyy = e * (e - 1) * pow(ay, e - 2) / (b * b);
    d0 = ct * gpx - st * gpy;
    d1 = st * gpx + ct * gpy;
    hxx = ct * ct * cxx + st * st * cyy;
    hxy = ct * st * cxx - st * ct * cyy;
    hyy = st * st * cxx + ct * ct * cyy;
}


// Cholesky in LDS for run-time orders (L = lower of A, row stride n), false on a pivot <= 0.  Right-looking with panels of
// 4 columns: the panel is factored column by column (updates confined to the panel), the trailing matrix then takes ONE
// rank-4 update per 16 x 16 tile as a v_mfma_f64_16x16x4_f64 (A = panel rows of the tile's rows, B' = panel rows of its
// columns): n^3 / 3 multiply-adds leave the LDS read-modify-write loop (n = 80: 533 k -> 60 k cycles).
typedef double ipm_c4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool cholesky_lds(double* A, int n, int lane) {
    bool ok = true;
    const int q = lane >> 4, l15 = lane & 15;
    for (int j0 = 0; j0 < n; j0 += 4) {
        const int pw = n - j0 < 4 ? n - j0 : 4;
        for (int jj = 0; jj < pw; ++jj) {
            const int j = j0 + jj;
            const double dd = A[j * n + j];
            if (!(dd > 0.0)) ok = false;
            const double inv = 1.0 / sqrt(dd);
            SC_SYNC();
            for (int i = j + lane; i < n; i += 64) A[i * n + j] = (i == j) ? dd * inv : A[i * n + j] * inv;
            SC_SYNC();
            // the remaining panel columns c = j+1 .. j0+pw-1, rows i >= c
            for (int e = lane; e < (n - j - 1) * (pw - jj - 1); e += 64) {
                const int c = j + 1 + e / (n - j - 1), i = j + 1 + e % (n - j - 1);
                if (i >= c) A[i * n + c] -= A[i * n + j] * A[c * n + j];
            }
            SC_SYNC();
            if (!ok) return false;                                         // uniform: every lane read the same pivot
        }
        const int t0 = j0 + pw;                                            // trailing matrix starts here
        if (t0 >= n) break;
        const int ntile = (n - t0 + 15) >> 4;
        for (int ti = 0; ti < ntile; ++ti) {
            for (int tj = 0; tj <= ti; ++tj) {
                const int ra = t0 + 16 * ti + l15, rb = t0 + 16 * tj + l15;
                const double a = (ra < n && q < pw) ? A[ra * n + j0 + q] : 0.0;
                const double b = (rb < n && q < pw) ? A[rb * n + j0 + q] : 0.0;
                ipm_c4 acc = {0.0, 0.0, 0.0, 0.0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t0 + 16 * ti + q + 4 * r, col = t0 + 16 * tj + l15;
                    if (row < n && col <= row) A[row * n + col] -= acc[r];
                }
            }
        }
        SC_SYNC();
    }
    return ok;
}
__device__ __forceinline__ void chol_solve_lds(const double* L, double* b, int n, int lane) {
    for (int j = 0; j < n; ++j) {
        if (lane == 0) b[j] = b[j] / L[j * n + j];
        SC_SYNC();
        const double yj = b[j];
        for (int i = j + 1 + lane; i < n; i += 64) b[i] -= L[i * n + j] * yj;
        SC_SYNC();
    }
    for (int j = n - 1; j >= 0; --j) {
        if (lane == 0) b[j] = b[j] / L[j * n + j];
        SC_SYNC();
        const double xj = b[j];
        for (int i = lane; i < j; i += 64) b[i] -= L[j * n + i] * xj;
        SC_SYNC();
    }
}

// register Cholesky for a compile-time order (mpc_chol.hpp); out of line like mpc_cbf.hip's (code size)
template <int nn>
__device__ __noinline__ bool chol_reg_solve(const double* M, const double* rhs, double* Lt, double* out, double delta, int lane) {
    double a[nn], diag;
    const int row = lane < nn ? lane : 0;
#pragma unroll
    for (int k = 0; k < nn; ++k) a[k] = M[row * nn + k] + (lane == k ? delta : 0.0);
    if (!chol_reg<nn>(a, lane, diag)) return false;
    const double x = chol_solve_reg<nn>(a, diag, rhs[row], Lt, lane);
    if (lane < nn) out[lane] = x;
    SC_SYNC();
    return true;
}


}  // namespace ipm
}  // namespace sc
