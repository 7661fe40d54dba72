// Batched MPC-CBF for gfx950: one receding-horizon NLP per wavefront, everything in LDS.
//
// Replaces, for B agents per launch, the per-robot path
//   MPCCBF.solve_control_problem   position_control/mpc_cbf.py:366-402
//   do-mpc -> casadi -> IPOPT      position_control/mpc_cbf.py:163,384
// for the problem MPCCBF.create_model / create_mpc / set_cbf_constraint define
// (mpc_cbf.py:108-160, 162-259, 295-325) with DynamicUnicycle2D.agent_barrier_dt
// (robots/dynamic_unicycle2D.py:188-238).  oracle/mpc_cbf.py is the float64 numpy statement of
// the same algorithm; this file mirrors it operation for operation.
//
// Formulation: single shooting on z = (u_0 .. u_{N-1}); the barrier depends on the position
// only, so every DT-CBF row is  w2 h(p_{k+2}) + w1 h(p_{k+1}) + w0 h(p_k) >= 0  over the predicted
// positions p_0..p_{N+1}.  Solver: primal-dual interior point with slacks, exact Hessian of the
// Lagrangian (closed-form second derivatives of the unicycle positions through suffix sums),
// inertia correction, fraction-to-the-boundary, l1-merit backtracking.
//
// Mapping: one problem per 64-lane wave (one wave per workgroup).  The condensed KKT matrix
// (2N x 2N), its Cholesky factor, the CBF Jacobian (N*K x 2N), position sensitivities, slacks
// and multipliers live in LDS (~37 KiB for N = 10, K = 8); lanes stride over matrix entries /
// constraint rows, reductions are wave shuffles.  All arithmetic is f64 (interior-point
// iterations drive slacks and mu to 1e-9); storage type of the I/O arrays is a parameter.
#include <hip/hip_runtime.h>

#include "sc_math.hpp"
#include "../../include/safe_control_amd.h"

namespace sc {

#define SC_SYNC() __syncthreads()

// -DSC_MPC_PROF: developer build that returns per-phase shader-clock totals in z_out instead of the solution
// (tools/exp_mpc_phases.py); never defined in the shipped library.
#ifdef SC_MPC_PROF
#define SC_PH(i) do { const long long t_ = clock64(); ph[i] += (double)(t_ - tph); tph = t_; } while (0)
#else
#define SC_PH(i) do { } while (0)
#endif

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wmin(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// ---- f64 MFMA tiles (v_mfma_f64_16x16x4_f64) for the two dense contractions of an iteration ------------
// Operand layout (MI355X guide, "f64 MFMA"): lane l feeds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// result register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15].
//
// tile (ti, tj) of  M += A' diag(sig) A  for a row-major A [rows][n] in LDS (the CBF Jacobian): the K-dimension
// of the contraction is the constraint index.
__device__ __forceinline__ d4_t mfma_tile_AtSA(const double* A, const double* sig, int rows, int n, int ti, int tj, int lane) {
    const int q = lane >> 4, cA = 16 * ti + (lane & 15), cB = 16 * tj + (lane & 15);
    const bool okA = cA < n, okB = cB < n;
    const int ca = okA ? cA : 0, cb = okB ? cB : 0;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < rows; k0 += 4) {
        const int row = k0 + q;
        const bool okr = row < rows;
        const int rr = okr ? row : 0;
        double a = A[(size_t)rr * n + ca] * sig[rr];
        double b = A[(size_t)rr * n + cb];
        a = (okA && okr) ? a : 0.0;
        b = (okB && okr) ? b : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// tile (ti, tj) of  sum_k dP_k' Om_k dP_k  (dP row-major [2 (N+2)][n], Om_k symmetric 2x2 as xx, xy, yy)
__device__ __forceinline__ d4_t mfma_tile_PtOP(const double* dP, const double* Om, int rows, int n, int ti, int tj, int lane) {
    const int q = lane >> 4, cA = 16 * ti + (lane & 15), cB = 16 * tj + (lane & 15);
    const bool okA = cA < n, okB = cB < n;
    const int ca = okA ? cA : 0, cb = okB ? cB : 0;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < rows; k0 += 4) {
        const int row = k0 + q;                          // row = 2 k + d
        const bool okr = row < rows;
        const int rr = okr ? row : 0, k = rr >> 1, d = rr & 1;
        double a = dP[(size_t)rr * n + ca];
        const double o0 = d ? Om[3 * k + 1] : Om[3 * k], o1 = d ? Om[3 * k + 2] : Om[3 * k + 1];
        double b = o0 * dP[(size_t)(2 * k) * n + cb] + o1 * dP[(size_t)(2 * k + 1) * n + cb];
        a = (okA && okr) ? a : 0.0;
        b = (okB && okr) ? b : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// add the lower-triangular tiles of a symmetric product into M (mirroring the off-diagonal tiles)
template <typename TileFn>
__device__ __forceinline__ void mfma_accumulate_sym(double* M, int n, int lane, TileFn tile) {
    const int nt = (n + 15) >> 4;
    for (int ti = 0; ti < nt; ++ti) {
        for (int tj = 0; tj <= ti; ++tj) {
            const d4_t acc = tile(ti, tj);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
                if (row < n && col < n) {
                    M[(size_t)row * n + col] += acc[r];
                    if (ti != tj) M[(size_t)col * n + row] += acc[r];
                }
            }
        }
    }
}

struct MpcMem {                 // LDS carve-up (doubles)
    double *z, *zt, *dz, *grad, *rhs, *zb;        // n
    double *TH, *V, *C, *S;                       // N+1
    double *pos, *PC, *PD, *q;                    // 2*(N+2)
    double *SA, *SB, *SV;                         // N+2 suffix sums
    double *obs;                                  // K*7
    double *hk, *mu;                              // (N+2)*K
    double *dh;                                   // (N+2)*K*2
    double *Hh;                                   // (N+2)*K*3
    double *Om;                                   // (N+2)*3
    double *g, *sl, *lam, *ds, *dlam, *w, *st;    // m
    double *J;                                    // mc*n   (CBF rows only)
    double *dP;                                   // (N+2)*2*n
    double *M, *L;                                // n*n, n*(n+1)
};

__host__ __device__ inline size_t mpc_lds_doubles(int N, int K) {
    const size_t n = 2 * (size_t)N, mc = (size_t)N * K, m = mc + 2 * N + 2 * n;
    return 6 * n + 4 * (N + 1) + 4 * 2 * (N + 2) + 3 * (N + 2) + (size_t)K * 7 + 2 * (size_t)(N + 2) * K +
           (size_t)(N + 2) * K * 2 + (size_t)(N + 2) * K * 3 + (size_t)(N + 2) * 3 + 7 * m + mc * n +
           (size_t)(N + 2) * 2 * n + 2 * n * n + n;
}

__device__ inline MpcMem carve(double* b, int N, int K) {
    const int n = 2 * N, mc = N * K, m = mc + 2 * N + 2 * n;
    MpcMem M;
    auto take = [&](size_t c) { double* r = b; b += c; return r; };
    M.z = take(n); M.zt = take(n); M.dz = take(n); M.grad = take(n); M.rhs = take(n); M.zb = take(n);
    M.TH = take(N + 1); M.V = take(N + 1); M.C = take(N + 1); M.S = take(N + 1);
    M.pos = take(2 * (N + 2)); M.PC = take(2 * (N + 2)); M.PD = take(2 * (N + 2)); M.q = take(2 * (N + 2));
    M.SA = take(N + 2); M.SB = take(N + 2); M.SV = take(N + 2);
    M.obs = take((size_t)K * 7);
    M.hk = take((size_t)(N + 2) * K); M.mu = take((size_t)(N + 2) * K);
    M.dh = take((size_t)(N + 2) * K * 2);
    M.Hh = take((size_t)(N + 2) * K * 3);
    M.Om = take((size_t)(N + 2) * 3);
    M.g = take(m); M.sl = take(m); M.lam = take(m); M.ds = take(m); M.dlam = take(m); M.w = take(m); M.st = take(m);
    M.J = take((size_t)mc * n);
    M.dP = take((size_t)(N + 2) * 2 * n);
    M.M = take((size_t)n * n); M.L = take((size_t)n * (n + 1));
    return M;
}

struct MpcConst {
    int N, K, n, mc, m;
    double dt, Qx, Qy, Qth, Qv, R0, R1, w0, w1, w2, vmax, amax, wmaxu, Rrob, beta;
    double x0, y0, th0, v0, up0, up1, gx, gy;
};

// ---- barrier h, dh/dp, d2h/dp2 at a position (oracle/mpc_cbf.py: barrier) ---------------------
// circle robots/dynamic_unicycle2D.py:194-202; superellipsoid :204-220 (fabs, clamps a,b>=1e-3, e>=2)
__device__ inline void barrier_at(double px_, double py_, const double* o, const MpcConst& c, bool derivs,
                                  double& h, double& d0, double& d1, double& hxx, double& hxy, double& hyy) {
    if (o[6] < 0.5) {
        const double d = c.Rrob + o[2];
        const double ex = px_ - o[0], ey = py_ - o[1];
        h = (ex * ex + ey * ey) - c.beta * d * d;
        d0 = 2.0 * ex; d1 = 2.0 * ey; hxx = 2.0; hxy = 0.0; hyy = 2.0;
        return;
    }
    const double a = fmax(fabs(o[2]), 1e-3) + c.Rrob, b = fmax(fabs(o[3]), 1e-3) + c.Rrob;
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

// ---- rollout + barrier values + g + f at a trial z (oracle: evaluate level 0) --------------------
__device__ inline double eval_values(const double* z, const MpcMem& W, const MpcConst& c, int lane, bool derivs) {
    const int N = c.N, K = c.K, n = c.n;
    if (lane <= N) {
        double th = c.th0, v = c.v0;
        for (int j = 0; j < lane; ++j) { th += c.dt * z[2 * j + 1]; v += c.dt * z[2 * j]; }
        W.TH[lane] = th; W.V[lane] = v;
        double sn, cs;
        sincos(th, &sn, &cs);
        W.C[lane] = cs; W.S[lane] = sn;
    }
    SC_SYNC();
    if (lane <= N + 1) {
        double pcx = 0, pcy = 0, pdx = 0, pdy = 0, px = c.x0, py = c.y0;
        for (int i = 0; i < lane; ++i) {
            const double ci = W.C[i], si = W.S[i], vi = W.V[i];
            pcx += ci; pcy += si; pdx += -vi * si; pdy += vi * ci;
            px += c.dt * vi * ci; py += c.dt * vi * si;
        }
        W.PC[2 * lane] = pcx; W.PC[2 * lane + 1] = pcy;
        W.PD[2 * lane] = pdx; W.PD[2 * lane + 1] = pdy;
        W.pos[2 * lane] = px; W.pos[2 * lane + 1] = py;
    }
    SC_SYNC();
    for (int e = lane; e < (N + 2) * K; e += 64) {
        const int k = e / K, j = e - k * K;
        double h, d0, d1, hxx, hxy, hyy;
        barrier_at(W.pos[2 * k], W.pos[2 * k + 1], W.obs + 7 * j, c, derivs, h, d0, d1, hxx, hxy, hyy);
        W.hk[e] = h;
        if (derivs) {
            W.dh[2 * e] = d0; W.dh[2 * e + 1] = d1;
            W.Hh[3 * e] = hxx; W.Hh[3 * e + 1] = hxy; W.Hh[3 * e + 2] = hyy;
        }
    }
    SC_SYNC();
    // g >= 0 : [CBF (k major) | v_max -/+ v_k (k = 1..N) | u_max - z | u_max + z]
    for (int i = lane; i < c.m; i += 64) {
        double gi;
        if (i < c.mc) {
            const int k = i / K, j = i - k * K;
            gi = c.w2 * W.hk[(k + 2) * K + j] + c.w1 * W.hk[(k + 1) * K + j] + c.w0 * W.hk[k * K + j];
        } else if (i < c.mc + 2 * N) {
            const int r = i - c.mc, k = (r >> 1) + 1;
            gi = (r & 1) ? (c.vmax + W.V[k]) : (c.vmax - W.V[k]);
        } else {
            const int r = i - c.mc - 2 * N;
            const int col = r < n ? r : r - n;
            const double ub = (col & 1) ? c.wmaxu : c.amax;
            gi = r < n ? (ub - z[col]) : (ub + z[col]);
        }
        W.g[i] = gi;
    }
    // f
    double part = 0.0;
    if (lane >= 1 && lane <= N) {
        const double ex = W.pos[2 * lane] - c.gx, ey = W.pos[2 * lane + 1] - c.gy;
        part = c.Qx * ex * ex + c.Qy * ey * ey + c.Qth * W.TH[lane] * W.TH[lane] + c.Qv * W.V[lane] * W.V[lane];
    }
    for (int i = lane; i < n; i += 64) {
        const double prev = i >= 2 ? z[i - 2] : ((i & 1) ? c.up1 : c.up0);
        const double du = z[i] - prev;
        part += ((i & 1) ? c.R1 : c.R0) * du * du;
    }
    SC_SYNC();
    return wsum(part);
}

// ---- first and second derivatives at W.z (oracle: evaluate level 1, 2) ---------------------------
// lam_scale = 1/sf converts the scaled problem's multipliers to those of the unscaled one.
__device__ inline void eval_derivs(const MpcMem& W, const MpcConst& c, int lane, double lam_scale) {
    const int N = c.N, K = c.K, n = c.n;
    const double dt = c.dt, dt2 = dt * dt;
    // dP[k][d][col] = d p_k / d z_col
    for (int e = lane; e < (N + 2) * n; e += 64) {
        const int k = e / n, col = e - k * n, j = col >> 1;
        double v0 = 0.0, v1 = 0.0;
        if (j + 1 <= k - 1) {
            const double* P = (col & 1) ? W.PD : W.PC;
            v0 = dt2 * (P[2 * k] - P[2 * (j + 1)]);
            v1 = dt2 * (P[2 * k + 1] - P[2 * (j + 1) + 1]);
        }
        W.dP[(size_t)(2 * k) * n + col] = v0;
        W.dP[(size_t)(2 * k + 1) * n + col] = v1;
    }
    // multipliers touching position k:  mu_kj = w2 lam_{k-2,j} + w1 lam_{k-1,j} + w0 lam_{k,j}
    for (int e = lane; e < (N + 2) * K; e += 64) {
        const int k = e / K, j = e - k * K;
        double mu = 0.0;
        if (k - 2 >= 0) mu += c.w2 * (W.lam[(k - 2) * K + j] * lam_scale);
        if (k >= 1 && k <= N) mu += c.w1 * (W.lam[(k - 1) * K + j] * lam_scale);
        if (k <= N - 1) mu += c.w0 * (W.lam[k * K + j] * lam_scale);
        W.mu[e] = mu;
    }
    SC_SYNC();
    // Om_k = d2 L / d p_k^2 (2x2 sym), q_k = d L / d p_k
    if (lane <= N + 1) {
        const int k = lane;
        double oxx = 0, oxy = 0, oyy = 0, q0 = 0, q1 = 0;
        for (int j = 0; j < K; ++j) {
            const int e = k * K + j;
            const double mu = W.mu[e];
            oxx -= mu * W.Hh[3 * e]; oxy -= mu * W.Hh[3 * e + 1]; oyy -= mu * W.Hh[3 * e + 2];
            q0 -= mu * W.dh[2 * e]; q1 -= mu * W.dh[2 * e + 1];
        }
        if (k >= 1 && k <= N) {
            oxx += 2.0 * c.Qx; oyy += 2.0 * c.Qy;
            q0 += 2.0 * c.Qx * (W.pos[2 * k] - c.gx);
            q1 += 2.0 * c.Qy * (W.pos[2 * k + 1] - c.gy);
        }
        W.Om[3 * k] = oxx; W.Om[3 * k + 1] = oxy; W.Om[3 * k + 2] = oyy;
        W.q[2 * k] = q0; W.q[2 * k + 1] = q1;
    }
    // CBF Jacobian rows
    for (int e = lane; e < c.mc * n; e += 64) {
        const int row = e / n, col = e - row * n, k = row / K, j = row - k * K;
        double acc = 0.0;
        const double wt[3] = {c.w0, c.w1, c.w2};
#pragma unroll
        for (int t = 2; t >= 0; --t) {
            const int kk = k + t, hh = kk * K + j;
            acc += wt[t] * (W.dh[2 * hh] * W.dP[(size_t)(2 * kk) * n + col] + W.dh[2 * hh + 1] * W.dP[(size_t)(2 * kk + 1) * n + col]);
        }
        W.J[e] = acc;
    }
    SC_SYNC();
    // suffix sums over stages i = 0..N:  A_i = qbar_i . (-s_i, c_i),  B_i = v_i qbar_i . (c_i, s_i),  qbar_i = sum_{k>i} q_k
    if (lane <= N) {
        const int i = lane;
        double qb0 = 0, qb1 = 0;
        for (int k = i + 1; k <= N + 1; ++k) { qb0 += W.q[2 * k]; qb1 += W.q[2 * k + 1]; }
        W.SA[i] = qb0 * (-W.S[i]) + qb1 * W.C[i];
        W.SB[i] = W.V[i] * (qb0 * W.C[i] + qb1 * W.S[i]);
    }
    SC_SYNC();
    if (lane == 0) {                                  // in-place suffix sums  SA[t] = sum_{i>=t} A_i
        double a = 0, b = 0;
        for (int i = N; i >= 0; --i) { a += W.SA[i]; b += W.SB[i]; W.SA[i] = a; W.SB[i] = b; }
        W.SA[N + 1] = 0; W.SB[N + 1] = 0;
    }
    // gradient of f
    for (int col = lane; col < n; col += 64) {
        const int j = col >> 1;
        double acc = 0.0;
        for (int k = 1; k <= N; ++k) {
            acc += W.dP[(size_t)(2 * k) * n + col] * (2.0 * c.Qx * (W.pos[2 * k] - c.gx)) +
                   W.dP[(size_t)(2 * k + 1) * n + col] * (2.0 * c.Qy * (W.pos[2 * k + 1] - c.gy));
            if (k > j) acc += (col & 1) ? 2.0 * c.Qth * W.TH[k] * dt : 2.0 * c.Qv * W.V[k] * dt;
        }
        const double Rc = (col & 1) ? c.R1 : c.R0;
        const double prev = col >= 2 ? W.z[col - 2] : ((col & 1) ? c.up1 : c.up0);
        acc += 2.0 * Rc * (W.z[col] - prev);
        if (col + 2 < n) acc -= 2.0 * Rc * (W.z[col + 2] - W.z[col]);
        W.grad[col] = acc;
    }
    SC_SYNC();
    // exact Hessian of the Lagrangian  W.M (full n x n)
    const double dt3 = dt2 * dt;
    for (int e = lane; e < n * n; e += 64) {
        const int r = e / n, cc = e - r * n;
        const int jr = r >> 1, jc = cc >> 1, jm = jr > jc ? jr : jc;
        double acc = 0.0;                                 // the sum_k dP_k' Om_k dP_k part is added by MFMA below
        const bool ra = !(r & 1), ca = !(cc & 1);
        if (ra && ca) acc += 2.0 * c.Qv * dt2 * (double)(N - jm);            // sum_k dV_k dV_k'
        else if (!ra && !ca) acc += 2.0 * c.Qth * dt2 * (double)(N - jm) - dt3 * W.SB[jm + 1];
        else acc += dt3 * W.SA[jm + 1];
        // input-rate penalty 2 D' R D
        const double Rc = (r & 1) ? c.R1 : c.R0;
        if (r == cc) acc += 2.0 * Rc * ((r + 2 < n) ? 2.0 : 1.0);
        else if (r == cc + 2 || cc == r + 2) acc -= 2.0 * Rc;
        W.M[e] = acc;
    }
    SC_SYNC();
    mfma_accumulate_sym(W.M, n, lane, [&](int ti, int tj) { return mfma_tile_PtOP(W.dP, W.Om, 2 * (N + 2), n, ti, tj, lane); });
    SC_SYNC();
}

// Cholesky of (W.L = lower of A) in place; returns false on a non-positive pivot.
__device__ inline bool cholesky(double* A, int n, int lane) {
    bool ok = true;
    for (int j = 0; j < n; ++j) {
        const double d = A[j * n + j];
        if (!(d > 0.0)) ok = false;
        const double piv = sqrt(d);
        SC_SYNC();
        for (int i = j + lane; i < n; i += 64) A[i * n + j] = (i == j) ? piv : A[i * n + j] / piv;
        SC_SYNC();
        const int rem = n - j - 1;
        for (int e = lane; e < rem * rem; e += 64) {
            const int i = j + 1 + e / rem, k = j + 1 + e % rem;
            if (k <= i) A[i * n + k] -= A[i * n + j] * A[k * n + j];
        }
        SC_SYNC();
        if (!ok) break;                       // uniform: every lane read the same pivot
    }
    return ok;
}

// solve L L' x = b in place (b in LDS), lanes cooperate column by column
__device__ inline void chol_solve(const double* L, double* b, int n, int lane) {
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

// ---- register-resident Cholesky for a compile-time order (n = 2 NT <= 64) -------------------------------------
// Lane i < n keeps row i of the matrix in VGPRs; a pivot row element is broadcast with v_readlane (the source lane
// is a compile-time constant in the fully unrolled loops), so a column step costs no LDS round trip and no barrier:
// ~1.3 k instructions for n = 20 against 60 barrier-separated LDS passes in cholesky()/chol_solve().
__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// In: a[k] = A[lane][k] (lower triangle used).  Out: a[k] = L[lane][k] for k <= lane, piv_out = L[lane][lane].
// Same operation order as cholesky(): right-looking, l_ij = a_ij / sqrt(d_j).  Returns false on a pivot <= 0.
template <int n>
__device__ __forceinline__ bool chol_reg(double (&a)[n], int lane, double& diag) {
    diag = 1.0;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const double d = bcast_lane(a[j], j);
        if (!(d > 0.0)) return false;                  // uniform
        const double piv = sqrt(d);
        a[j] = (lane == j) ? piv : a[j] / piv;
        diag = (lane == j) ? piv : diag;
#pragma unroll
        for (int k = j + 1; k < n; ++k) {
            const double lkj = bcast_lane(a[j], k);
            a[k] -= a[j] * lkj;                        // meaningful for lanes >= k; the upper triangle is never read
        }
    }
    return true;
}

// Solve L L' x = b with L row-held in a[] (from chol_reg), b = this lane's right-hand-side entry.  Lt is an LDS
// scratch of n (n + 1) doubles used once to transpose L (row stride n + 1 keeps the 64 banks conflict-free).
template <int n>
__device__ __forceinline__ double chol_solve_reg(double (&a)[n], double diag, double b, double* Lt, int lane) {
    const double dinv = 1.0 / diag;
    constexpr int ld = n + 1;
#pragma unroll
    for (int k = 0; k < n; ++k) {
        if (lane < n && k < lane) Lt[lane * ld + k] = a[k];
        a[k] = (lane > k && lane < n) ? a[k] : 0.0;    // strictly lower part only: the updates below need no select
    }
    // forward  L y = b
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const double yj = bcast_lane(b * dinv, j);
        b -= a[j] * yj;
    }
    b *= dinv;                                          // y_i
    SC_SYNC();
    double c[n];                                        // c[k] = L[k][lane] for k > lane (column of L)
#pragma unroll
    for (int k = 0; k < n; ++k) c[k] = (k > lane && lane < n) ? Lt[k * ld + lane] : 0.0;
    // backward  L' x = y
#pragma unroll
    for (int j = n - 1; j >= 0; --j) {
        const double xj = bcast_lane(b * dinv, j);
        b -= c[j] * xj;
    }
    return b * dinv;
}

// (J' w)[col] over all rows: CBF rows dense, speed rows -/+ dt for stages k > j, box rows -/+ identity
__device__ inline double jt_times(const MpcMem& W, const MpcConst& c, const double* w, int col) {
    const int N = c.N, n = c.n, j = col >> 1;
    double acc = 0.0;
    for (int r = 0; r < c.mc; ++r) acc += W.J[(size_t)r * n + col] * w[r];
    if (!(col & 1)) {
        for (int k = j + 1; k <= N; ++k) acc += c.dt * (w[c.mc + 2 * (k - 1) + 1] - w[c.mc + 2 * (k - 1)]);
    }
    acc += w[c.mc + 2 * N + n + col] - w[c.mc + 2 * N + col];
    return acc;
}

// (J v)[row]
__device__ inline double j_times(const MpcMem& W, const MpcConst& c, const double* v, int row) {
    const int N = c.N, n = c.n;
    if (row < c.mc) {
        double acc = 0.0;
        for (int col = 0; col < n; ++col) acc += W.J[(size_t)row * n + col] * v[col];
        return acc;
    }
    if (row < c.mc + 2 * N) {
        const int r = row - c.mc, k = (r >> 1) + 1;
        double acc = 0.0;
        for (int j = 0; j < k; ++j) acc += c.dt * v[2 * j];
        return (r & 1) ? acc : -acc;
    }
    const int r = row - c.mc - 2 * N;
    return r < n ? -v[r] : v[r - n];
}

// NT, KT > 0: horizon and obstacle count are compile-time constants (index arithmetic folds to shifts and
// multiplies); NT == 0: run-time sizes.
template <typename TIO, int NT, int KT>
__global__ __launch_bounds__(64) void mpccbf_kernel(const sc_mpccbf_params p, const long long B, const int K_rt,
                                                    const TIO* __restrict__ X, const TIO* __restrict__ u_prev,
                                                    const TIO* __restrict__ goal, const TIO* __restrict__ obs,
                                                    TIO* __restrict__ u_out, int* __restrict__ status_out,
                                                    int* __restrict__ iters_out, TIO* __restrict__ z_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x;
    const long long prob = blockIdx.x;
    if (prob >= B) return;
    MpcConst c;
    const int K = KT > 0 ? KT : K_rt;
    c.N = NT > 0 ? NT : p.horizon; c.K = K; c.n = 2 * c.N; c.mc = c.N * K; c.m = c.mc + 2 * c.N + 2 * c.n;
    c.dt = p.dt; c.Qx = p.Q[0]; c.Qy = p.Q[1]; c.Qth = p.Q[2]; c.Qv = p.Q[3]; c.R0 = p.R[0]; c.R1 = p.R[1];
    const double g1 = p.alpha1 + p.alpha2, g2 = p.alpha1 * p.alpha2;
    c.w0 = 1.0 - g1 + g2; c.w1 = g1 - 2.0; c.w2 = 1.0;
    c.vmax = p.v_max; c.amax = p.u_max[0]; c.wmaxu = p.u_max[1]; c.Rrob = p.robot_radius; c.beta = p.beta;
    c.x0 = (double)X[prob * 4 + 0]; c.y0 = (double)X[prob * 4 + 1]; c.th0 = (double)X[prob * 4 + 2]; c.v0 = (double)X[prob * 4 + 3];
    c.up0 = (double)u_prev[prob * 2 + 0]; c.up1 = (double)u_prev[prob * 2 + 1];
    c.gx = (double)goal[prob * 2 + 0]; c.gy = (double)goal[prob * 2 + 1];
    const int N = c.N, n = c.n, m = c.m;
    const MpcMem W = carve(sm, N, K);

    const TIO* osrc = obs + (p.obs_shared ? 0 : (size_t)prob * K * 7);
    for (int e = lane; e < K * 7; e += 64) W.obs[e] = (double)osrc[e];
    // set_initial_guess (mpc_cbf.py:369): u_prev at every stage, pulled strictly inside the box
    for (int i = lane; i < n; i += 64) {
        const double ub = (i & 1) ? c.wmaxu : c.amax;
        const double u = (i & 1) ? c.up1 : c.up0;
        W.z[i] = fmin(fmax(u, -0.99 * ub), 0.99 * ub);
    }
    SC_SYNC();

    double f = eval_values(W.z, W, c, lane, true);
    for (int i = lane; i < m; i += 64) W.lam[i] = 0.0;
    SC_SYNC();
    eval_derivs(W, c, lane, 1.0);
    double gmax = 0.0;
    for (int i = lane; i < n; i += 64) gmax = fmax(gmax, fabs(W.grad[i]));
    gmax = wmax(gmax);
    const double sf = fmin(1.0, 100.0 / fmax(1e-12, gmax));             // objective scaling
    double mu = p.mu_init;
    for (int i = lane; i < m; i += 64) {
        const double s = fmax(W.g[i], 1e-2);
        W.sl[i] = s;
        W.lam[i] = mu / s;
    }
    SC_SYNC();

    int status = SC_STATUS_INACCURATE, it = 0;
#ifdef SC_MPC_PROF
    double ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tph = clock64();
#endif
    const double tau = 0.995;
    double nu = 10.0, delta_last = 0.0, e_best = 1e300;
    for (int i = lane; i < n; i += 64) W.zb[i] = W.z[i];
    for (it = 1; it <= p.max_iter; ++it) {
        if (it > 1) f = eval_values(W.z, W, c, lane, true);
        SC_PH(0);
        eval_derivs(W, c, lane, 1.0 / sf);
        SC_PH(1);
        // residuals
        double e_d = 0.0, e_p = 0.0, e_c0 = 0.0, lmax = 0.0;
        for (int col = lane; col < n; col += 64) {
            const double rd = sf * W.grad[col] - jt_times(W, c, W.lam, col);
            e_d = fmax(e_d, fabs(rd));
        }
        for (int i = lane; i < m; i += 64) {
            e_p = fmax(e_p, fabs(W.g[i] - W.sl[i]));
            e_c0 = fmax(e_c0, fabs(W.sl[i] * W.lam[i]));
            lmax = fmax(lmax, W.lam[i]);
        }
        e_d = wmax(e_d); e_p = wmax(e_p); e_c0 = wmax(e_c0); lmax = wmax(lmax);
        const double e_opt = fmax(e_d, fmax(e_p, e_c0));
        if (e_opt < e_best) {                                            // remember the best iterate
            e_best = e_opt;
            for (int i = lane; i < n; i += 64) W.zb[i] = W.z[i];
        }
        if (e_opt <= p.tol) { status = SC_STATUS_OPTIMAL; break; }
        if (lmax > 1e10) { status = SC_STATUS_INFEASIBLE; break; }
        // barrier update
        for (;;) {
            double e_c = 0.0;
            for (int i = lane; i < m; i += 64) e_c = fmax(e_c, fabs(W.sl[i] * W.lam[i] - mu));
            e_c = wmax(e_c);
            const double e_mu = fmax(e_d, fmax(e_p, e_c));
            if (e_mu <= 10.0 * mu && mu > p.mu_min) mu = fmax(p.mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
            else break;
        }
        SC_PH(2);
        // condensed system  (sf W + J' Sigma J) dz = -sf grad + J' (mu/s - Sigma r_p)
        for (int i = lane; i < m; i += 64) {
            const double s = W.sl[i], sig = W.lam[i] / s;
            W.st[i] = sig;                                               // Sigma
            W.w[i] = mu / s - sig * (W.g[i] - s);
        }
        SC_SYNC();
        for (int col = lane; col < n; col += 64) W.rhs[col] = -sf * W.grad[col] + jt_times(W, c, W.w, col);
        for (int e = lane; e < n * n; e += 64) {
            const int r = e / n, cc = e - r * n;
            double acc = sf * W.M[e];                                   // + J_cbf' Sigma J_cbf by MFMA below
            if (!(r & 1) && !(cc & 1)) {                                  // speed rows: dt^2 sum_{k > max(jr,jc)} (sig+ + sig-)
                const int jm = (r > cc ? r : cc) >> 1;
                double sv = 0.0;
                for (int k = jm + 1; k <= N; ++k) sv += W.st[c.mc + 2 * (k - 1)] + W.st[c.mc + 2 * (k - 1) + 1];
                acc += c.dt * c.dt * sv;
            }
            if (r == cc) acc += W.st[c.mc + 2 * N + r] + W.st[c.mc + 2 * N + n + r];
            W.M[e] = acc;                                                 // M now holds the condensed matrix
        }
        SC_SYNC();
        mfma_accumulate_sym(W.M, n, lane, [&](int ti, int tj) { return mfma_tile_AtSA(W.J, W.st, c.mc, n, ti, tj, lane); });
        SC_SYNC();
        SC_PH(3);
        // inertia correction: M + delta I until the Cholesky succeeds
        double delta = 0.0;
        bool ok = false;
        if constexpr (NT > 0) {
            constexpr int nn = 2 * NT;
            double a[nn], diag;
            for (int t = 0; t < 40 && !ok; ++t) {
#pragma unroll
                for (int k = 0; k < nn; ++k) a[k] = (lane < nn) ? W.M[lane * nn + k] + (lane == k ? delta : 0.0) : 0.0;
                ok = chol_reg<nn>(a, lane, diag);
                if (!ok) delta = (delta == 0.0) ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
            }
            if (!ok) break;
            if (delta > 0.0) delta_last = delta;
            SC_PH(4);
            const double x = chol_solve_reg<nn>(a, diag, lane < nn ? W.rhs[lane] : 0.0, W.L, lane);
            if (lane < nn) W.dz[lane] = x;
            SC_SYNC();
        } else {
            for (int t = 0; t < 40 && !ok; ++t) {
                for (int e = lane; e < n * n; e += 64) W.L[e] = W.M[e] + ((e / n == e % n) ? delta : 0.0);
                SC_SYNC();
                ok = cholesky(W.L, n, lane);
                if (!ok) delta = (delta == 0.0) ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
            }
            if (!ok) break;
            if (delta > 0.0) delta_last = delta;
            SC_PH(4);
            for (int i = lane; i < n; i += 64) W.dz[i] = W.rhs[i];
            SC_SYNC();
            chol_solve(W.L, W.dz, n, lane);
        }
        SC_PH(5);
        // ds, dlam, step lengths
        double ap = 1.0, ad = 1.0, sum_ds_s = 0.0, sum_rp = 0.0, sum_log = 0.0;
        for (int i = lane; i < m; i += 64) {
            const double s = W.sl[i], lam = W.lam[i], rp = W.g[i] - s;
            const double ds = j_times(W, c, W.dz, i) + rp;
            const double dl = -W.st[i] * ds - (lam - mu / s);
            W.ds[i] = ds; W.dlam[i] = dl;
            if (ds < 0.0) ap = fmin(ap, -tau * s / ds);
            if (dl < 0.0) ad = fmin(ad, -tau * lam / dl);
            sum_ds_s += ds / s; sum_rp += fabs(rp); sum_log += log(s);
        }
        ap = wmin(ap); ad = wmin(ad); sum_ds_s = wsum(sum_ds_s); sum_rp = wsum(sum_rp); sum_log = wsum(sum_log);
        nu = fmax(nu, 1.1 * lmax);
        double gdz = 0.0;
        for (int i = lane; i < n; i += 64) gdz += sf * W.grad[i] * W.dz[i];
        gdz = wsum(gdz);
        const double phi0 = sf * f - mu * sum_log + nu * sum_rp;
        const double dphi = gdz - mu * sum_ds_s - nu * sum_rp;
        SC_PH(6);
        // l1-merit backtracking
        double alpha = ap;
        bool accepted = false;
        for (int ls = 0; ls < 12; ++ls) {                              // at most 12 halvings, then give up (best iterate)
            for (int i = lane; i < n; i += 64) W.zt[i] = W.z[i] + alpha * W.dz[i];
            SC_SYNC();
            const double ft = eval_values(W.zt, W, c, lane, false);
            double slog = 0.0, srp = 0.0;
            for (int i = lane; i < m; i += 64) {
                const double st = W.sl[i] + alpha * W.ds[i];
                slog += log(st); srp += fabs(W.g[i] - st);
            }
            slog = wsum(slog); srp = wsum(srp);
            const double phit = sf * ft - mu * slog + nu * srp;
            // Armijo, with an allowance for round-off in the merit function near convergence
            // (f is a sum of a few hundred terms of size |phi|: its noise is ~1e-13 |phi|)
            if (phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * fabs(phi0)) { accepted = true; break; }
            alpha *= 0.5;
        }
        SC_PH(7);
        if (!accepted) break;
        for (int i = lane; i < n; i += 64) W.z[i] = W.z[i] + alpha * W.dz[i];
        for (int i = lane; i < m; i += 64) {
            const double s = W.sl[i] + alpha * W.ds[i];
            double lam = W.lam[i] + ad * W.dlam[i];
            lam = fmin(fmax(lam, mu / (1e10 * s)), 1e10 * mu / s);       // IPOPT eq. (16) safeguard
            W.sl[i] = s; W.lam[i] = lam;
        }
        SC_SYNC();
        SC_PH(8);
    }
    if (it > p.max_iter) it = p.max_iter;
    if (status != SC_STATUS_OPTIMAL && e_best <= p.acceptable_tol) {
        // stalled at the precision limit (ill-conditioned condensed system at mu ~ 1e-9): the best iterate is
        // within the acceptable tolerance, like IPOPT's acceptable_tol exit
        SC_SYNC();
        for (int i = lane; i < n; i += 64) W.z[i] = W.zb[i];
        status = SC_STATUS_OPTIMAL;
    }
    SC_SYNC();
    eval_values(W.z, W, c, lane, false);
    if (status != SC_STATUS_OPTIMAL) {
        double gmin = 1e300;
        for (int i = lane; i < m; i += 64) gmin = fmin(gmin, W.g[i]);
        gmin = wmin(gmin);
        if (gmin < -1e-6) status = SC_STATUS_INFEASIBLE;
        else if (status != SC_STATUS_INFEASIBLE) status = SC_STATUS_INACCURATE;
    }
    if (lane == 0) {
        u_out[prob * 2 + 0] = (TIO)W.z[0];
        u_out[prob * 2 + 1] = (TIO)W.z[1];
        status_out[prob] = status;
        if (iters_out) iters_out[prob] = it;
    }
#ifdef SC_MPC_PROF
    if (z_out && lane < 12) z_out[prob * n + lane] = (TIO)ph[lane];
#else
    if (z_out) for (int i = lane; i < n; i += 64) z_out[prob * n + i] = (TIO)W.z[i];
#endif
}

size_t mpccbf_lds_bytes(int N, int K) { return mpc_lds_doubles(N, K) * sizeof(double); }

template <typename TIO, int NT, int KT>
static hipError_t mpc_launch_one(const sc_mpccbf_params& p, long long B, int K, const void* X, const void* u_prev,
                                 const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out,
                                 hipStream_t stream) {
    const size_t lds = mpc_lds_doubles(p.horizon, K) * sizeof(double);
    auto kern = mpccbf_kernel<TIO, NT, KT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64), lds, stream, p, B, K, (const TIO*)X, (const TIO*)u_prev,
                       (const TIO*)goal, (const TIO*)obs, (TIO*)u_out, status, iters, (TIO*)z_out);
    return hipGetLastError();
}

template <typename TIO>
static hipError_t mpc_launch_t(const sc_mpccbf_params& p, long long B, int K, const void* X, const void* u_prev,
                               const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out,
                               hipStream_t stream) {
    if (p.horizon == 10 && K == 8)            // BASELINE config 3
        return mpc_launch_one<TIO, 10, 8>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream);
    return mpc_launch_one<TIO, 0, 0>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream);
}

hipError_t mpccbf_launch(const sc_mpccbf_params& p, long long B, int K, const void* X, const void* u_prev,
                         const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out,
                         hipStream_t stream) {
    if (mpc_lds_doubles(p.horizon, K) * sizeof(double) > 160 * 1024) return hipErrorInvalidValue;
    if (p.io_dtype == SC_DTYPE_F32)
        return mpc_launch_t<float>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream);
    return mpc_launch_t<double>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream);
}

}  // namespace sc
