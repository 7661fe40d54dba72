// MPC-CBF for the reference's LINEAR robot models (SingleIntegrator2D, Quad3D), one NLP per wavefront.
//   MPCCBF (position_control/mpc_cbf.py:7-402) over robots/single_integrator2D.py and robots/quad3D.py:
//     prediction  x+ = x + (f(x) + g(x) u) dt = Ae x + Be u                                  mpc_cbf.py:135-141
//     cost        sum_k (x_k - xg)' Q (x_k - xg) + r-term R on delta u                        mpc_cbf.py:144,176-180
//     CBF         h(step(x_k, u_k)) - (1 - alpha) h(x_k) >= 0 per stage and obstacle          mpc_cbf.py:312-315
//                 (step = the robot's own one-step map: Euler for SI, RK4 of the linear system for Quad3D, quad3D.py:121-158)
//     bounds      box on the inputs                                                            mpc_cbf.py:183-187,219-223
//   Oracle: oracle/mpc_lin.py (problem functions) + oracle/mpc_cbf.py: solve (the interior-point method followed here).
//
// With linear dynamics every barrier point is affine in z = (u_0..u_{N-1}):  a_k = pos(x_k),  b_k = pos(As x_k + Bs u_k),
// points = const + G z with a CONSTANT G (4N x n) and the cost Hessian Hc (n x n) is constant too; both are built once
// per controller on the host (sc_mpclin_build_model) and shared by every problem of the batch (read through L1/L2).
// What is left per interior-point iteration is: a rollout (N sequential nx-wide matrix-vector steps), 2 N K barrier
// evaluations, the 4x4 stage blocks Phi_k of J' Sigma J - sum lam grad^2 g over the point pair (a_k, b_k), T = Phi G,
// M = sf Hc + G' T + box terms, one Cholesky of order n = N nu <= 64, and the line search (rollout + barriers only).
// Single shooting, slacks on all N K + 2 n inequalities, exact Hessian of the Lagrangian, inertia correction,
// fraction-to-boundary 0.995, l1-merit backtracking, monotone barrier decrease -- iterate for iterate the oracle's method.
// Arithmetic is f64; the caller's arrays are f32 or f64.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/safe_control_amd.h"
#include "sc_math.hpp"
#include "mpc_chol.hpp"
#include "mpc_ipm_common.hpp"
#include "mpc_cont.hpp"

namespace sc {

namespace {

using ipm::cholesky_lds;
using ipm::chol_solve_lds;
// Reductions over the TH threads that share a problem.  TH = 64: the wave reduction.  TH = 256 (the big layout, one problem
// per CU: its four waves sit on the four SIMDs): wave totals meet in LDS; two alternating slots, so ONE barrier per
// reduction (a slot is rewritten only after the barrier of the next reduction, which every reader of it has passed).
struct Red { double* buf; int par; };
template <int TH, typename F, typename WF>
__device__ __forceinline__ double block_red(double v, Red& R, F f, WF wf) {
    v = wf(v);
    if constexpr (TH > 64) {
        double* b = R.buf + 4 * R.par;
        R.par ^= 1;
        if ((threadIdx.x & 63) == 0) b[threadIdx.x >> 6] = v;
        SC_SYNC();
        v = f(f(b[0], b[1]), f(b[2], b[3]));
    }
    return v;
}
template <int TH> __device__ __forceinline__ double lsum(double v, Red& R) {
    return block_red<TH>(v, R, [](double a, double b) { return a + b; }, [](double a) { return ipm::wsum(a); });
}
template <int TH> __device__ __forceinline__ double lmin(double v, Red& R) {
    return block_red<TH>(v, R, [](double a, double b) { return fmin(a, b); }, [](double a) { return ipm::wmin(a); });
}
template <int TH> __device__ __forceinline__ double lmax_(double v, Red& R) {
    return block_red<TH>(v, R, [](double a, double b) { return fmax(a, b); }, [](double a) { return ipm::wmax(a); });
}

struct LinMem {
    double *Ae, *Be, *As2, *Bs2, *xg, *up, *lx;          // model, goal state, previous input, adjoint (2 nx)
    double *z, *zt, *zb, *dz, *gs, *rd, *rhs;            // n each
    double *xs, *pts, *y, *pdz, *obs, *hk, *dh, *hh;     // (N+1) nx | 4N | 4N | 4N | 7K | 2N K | 4N K | 6N K
    double *g, *s, *lam, *ds, *dlam, *vb;                // m each
    double *tel;                                         // N K: elastic variables of the feasibility restoration (mpc_ipm_common.hpp)
    double *cq;                                          // Q (12) | R (4) | u_lo (4) | u_hi (4)
    double *Phi, *T, *M, *L, *G;                         // 16 N | 4N n | n n | n (n + 1) | 4N n
    double *red;                                         // 8 reduction slots + the Cholesky status word (multi-wave layout)
    double *Hc, *clin;                                   // lean kernels: n n copy of the cost Hessian, n constant part of grad f
    // optimal decay (config-5 extension, oracle/od_mpc_rd1.py): one decay variable per stage
    double *rho, *rhot, *drho, *rhob, *w0k;              // N each: current, trial, step, best iterate; stage weight -(1 - alpha rho_k)
    double *Cv, *dinv, *rr, *rdr;                        // 4N | N | N | N: point-space coupling C_k, 1 / D_k, right-hand side, r_d of rho_k
};

struct LinDims { int N, K, nx, nu, n, m, mc; };

// lean = the compile-time instantiations: the Cholesky runs in registers, so its only LDS need is the n (n + 1) transpose
// scratch, which reuses the dead T | M region; the space goes to a copy of Hc (M assembly, line-search curvature and the
// gradient Hc z + c read it every iteration; 75 KB for Quad3D at N = 10, K = 8).  Reading G from global
// memory as well (49 KB, three problems per CU instead of two) was measured: 8 % faster at 65536 problems, 15 % slower
// at 4096 (the L1/L2 latency sits on every iteration's critical path) -- G stays in LDS.
// big (mode 2) = run-time sizes whose standard layout exceeds 80 KB, i.e. leaves one problem per CU or does not fit at all
// (Quad3D at N = 20: n = 80, 250 KB): the block is FOUR waves per problem (see mpclin_kernel), G is read from global memory
// (a model constant: L1 / L2 resident), T = Phi G is never stored (the M assembly forms its T operand on the fly from the 4 x 4
// stage blocks and four rows of G), M is assembled STRAIGHT INTO the Cholesky storage (and re-assembled in the rare
// inertia-correction retry), and that storage holds only the 16 x 16 tiles on and below the diagonal (ipm::LdTile).  Quad3D at
// N = 20, K = 8: 146 KB -> 77 KB, i.e. TWO problems per CU instead of one for a kernel that is latency bound.
enum { LIN_STD = 0, LIN_LEAN = 1, LIN_BIG = 2 };
__host__ __device__ inline size_t mpclin_lds_doubles(int N, int K, int nx, int nu, int mode = LIN_STD, bool od = false) {
    const bool lean = mode == LIN_LEAN;
    const size_t n = (size_t)N * nu, m = (size_t)N * K + 2 * n;
    return (od ? 12 * (size_t)N : (size_t)N * K) + (size_t)nx * nx + (size_t)nx * nu + 2 * nx + 2 * nu + nx + nu + 2 * nx + 24 + 10 + 7 * n + (size_t)(N + 1) * nx + 12 * N +
           7 * (size_t)K + 12 * (size_t)N * K + 6 * m + 16 * N +
           (mode == LIN_BIG ? ipm::LdTile::doubles((int)n)
                            : (lean ? 4 * (size_t)N * n + n * (n + 1) + n                             // G | M (n (n + 1): the factor's transposition scratch too) | clin
                                    : 4 * (size_t)N * n + n * n + 4 * (size_t)N * n + n * (n + 1)));   // T | M | G | L
}

__device__ inline LinMem carve_lin(double* b, const LinDims& d, int mode, bool od = false) {
    const bool lean = mode == LIN_LEAN;
    LinMem W;
    auto take = [&](size_t c) { double* r = b; b += c; return r; };
    const int N = d.N, K = d.K, nx = d.nx, nu = d.nu, n = d.n, m = d.m;
    W.Ae = take(nx * nx); W.Be = take(nx * nu); W.As2 = take(2 * nx); W.Bs2 = take(2 * nu);
    W.xg = take(nx); W.up = take(nu); W.lx = take(2 * nx); W.cq = take(24); W.red = take(10);
    W.z = take(n); W.zt = take(n); W.zb = take(n); W.dz = take(n); W.gs = take(n); W.rd = take(n); W.rhs = take(n);
    W.xs = take((N + 1) * nx); W.pts = take(4 * N); W.y = take(4 * N); W.pdz = take(4 * N);
    W.obs = take(7 * K); W.hk = take(2 * N * K); W.dh = take(4 * N * K); W.hh = take(6 * N * K);
    W.g = take(m); W.s = take(m); W.lam = take(m); W.ds = take(m); W.dlam = take(m); W.vb = take(m);
    W.tel = od ? nullptr : take((size_t)N * K);
    W.Phi = take(16 * N);
    W.T = W.M = nullptr;
    if (mode == LIN_STD) W.T = take((size_t)4 * N * n);
    if (mode != LIN_BIG) W.M = take(lean ? (size_t)n * (n + 1) : (size_t)n * n);       // lean: the scratch of chol_solve_reg (row stride n + 1) lies over M
    W.Hc = W.clin = nullptr;
    W.rho = W.rhot = W.drho = W.rhob = W.w0k = W.Cv = W.dinv = W.rr = W.rdr = nullptr;
    auto take_od = [&]() {
        if (!od) return;
        W.rho = take(N); W.rhot = take(N); W.drho = take(N); W.rhob = take(N); W.w0k = take(N);
        W.Cv = take(4 * N); W.dinv = take(N); W.rr = take(N); W.rdr = take(N);
    };
    if (mode == LIN_BIG) { W.G = nullptr; W.L = take(ipm::LdTile::doubles(n)); take_od(); return W; }
    W.G = take((size_t)4 * N * n);
    // lean (round 2): no T (the condensation forms that operand on the fly), no copy of Hc (read through L1 / L2: every problem of the
    // batch shares it), and the register factorisation writes its transposed factor over M, which it has read into registers by then
    // and which nothing reads again before the next condensation (chol_reg_solve leaves M alone when the factorisation fails):
    // Quad3D N = 10 75 -> 50 KB, three problems per CU
    if (lean) { W.L = W.M; W.clin = take(n); }
    else W.L = take((size_t)n * (n + 1));
    take_od();
    return W;
}

struct LinConst {
    double w0, Rrob, beta;
    int circles_only;
    int abs_r;                                           // input term R u^2 instead of do-mpc's delta-u penalty (optimal_decay = 2, OD)
    double alpha, ps, rf;                                // optimal decay: gain, penalty p_sb1, reference omega1
};

// rollout, barrier points, f, barrier values (and derivatives), g.  Points: a_k = index k, b_k = index N + k.
template <int TH, bool OD = false>
__device__ __forceinline__ double lin_eval(const double* zv, const double* rhov, const LinMem& W, const LinDims& d, const LinConst& c, int lane,
                                        bool derivs, Red& R) {
    const int N = d.N, K = d.K, nx = d.nx, nu = d.nu, n = d.n;
    for (int k = 0; k < N; ++k) {
        if (lane < nx) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nx; ++j) acc += W.Ae[lane * nx + j] * W.xs[k * nx + j];
#pragma unroll
            for (int j = 0; j < nu; ++j) acc += W.Be[lane * nu + j] * zv[k * nu + j];
            W.xs[(k + 1) * nx + lane] = acc;
        }
        SC_SYNC();
    }
    for (int e = lane; e < 2 * N; e += TH) {
        const int k = e >> 1, dd = e & 1;
        W.pts[e] = W.xs[k * nx + dd];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < nx; ++j) acc += W.As2[dd * nx + j] * W.xs[k * nx + j];
#pragma unroll
        for (int j = 0; j < nu; ++j) acc += W.Bs2[dd * nu + j] * zv[k * nu + j];
        W.pts[2 * N + e] = acc;
    }
    double part = 0.0;
    for (int e = lane; e < N * nx; e += TH) {
        const int k = e / nx + 1, i = e - (k - 1) * nx;
        const double dv = W.xs[k * nx + i] - W.xg[i];
        part += W.cq[i] * dv * dv;
    }
    for (int i = lane; i < n; i += TH) {
        const double prev = i >= nu ? zv[i - nu] : W.up[i];
        const double du = (OD || c.abs_r) ? zv[i] : zv[i] - prev;         // optimal decay: R u^2 (optimal_decay_mpc_cbf.py:178-179)
        part += W.cq[12 + i % nu] * du * du;
    }
    if constexpr (OD) {
        for (int k = lane; k < N; k += TH) {                              // row k:  h(b_k) - (1 - alpha rho_k) h(a_k)
            const double r = rhov[k];
            W.w0k[k] = -(1.0 - c.alpha * r);
            part += c.ps * (r - c.rf) * (r - c.rf);
        }
    }
    SC_SYNC();
    for (int e = lane; e < 2 * N * K; e += TH) {
        const int pt = e / K, j = e - pt * K;
        double h, d0, d1, hxx, hxy, hyy;
        ipm::ipm_barrier<true>(W.pts[2 * pt], W.pts[2 * pt + 1], W.obs + 7 * j, c.Rrob, c.beta, c.circles_only != 0, derivs, h, d0, d1, hxx, hxy, hyy);
        W.hk[e] = h;
        if (derivs) {
            W.dh[2 * e] = d0; W.dh[2 * e + 1] = d1;
            W.hh[3 * e] = hxx; W.hh[3 * e + 1] = hxy; W.hh[3 * e + 2] = hyy;
        }
    }
    SC_SYNC();
    for (int i = lane; i < d.m; i += TH) {
        double gi;
        if (i < d.mc) {
            const int k = i / K, j = i - k * K;
            gi = W.hk[(N + k) * K + j] + (OD ? W.w0k[k] : c.w0) * W.hk[k * K + j];
        } else if (i < d.mc + n) {
            const int col = i - d.mc;
            gi = W.cq[20 + col % nu] - zv[col];
        } else {
            const int col = i - d.mc - n;
            gi = zv[col] - W.cq[16 + col % nu];
        }
        W.g[i] = gi;
    }
    SC_SYNC();
    return lsum<TH>(part, R);
}

// gradient of the (unscaled) cost by the adjoint recursion:  l_k = 2 Q (x_k - xg) + Ae' l_{k+1},  df/du_{k-1} = Be' l_k
template <int TH, bool OD = false>
__device__ __forceinline__ void lin_grad(const LinMem& W, const LinDims& d, const LinConst& c, int lane, double sf) {
    const int N = d.N, nx = d.nx, nu = d.nu, n = d.n;
    for (int k = N; k >= 1; --k) {
        double* cur = W.lx + (k & 1) * nx;
        const double* nxt = W.lx + ((k + 1) & 1) * nx;
        if (lane < nx) {
            double acc = 2.0 * W.cq[lane] * (W.xs[k * nx + lane] - W.xg[lane]);
            if (k < N)
#pragma unroll
                for (int j = 0; j < nx; ++j) acc += W.Ae[j * nx + lane] * nxt[j];
            cur[lane] = acc;
        }
        SC_SYNC();
        if (lane < nu) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < nx; ++j) acc += W.Be[j * nu + lane] * cur[j];
            W.gs[(k - 1) * nu + lane] = acc;
        }
        SC_SYNC();
    }
    for (int i = lane; i < n; i += TH) {
        const double prev = i >= nu ? W.z[i - nu] : W.up[i];
        const bool absr = OD || c.abs_r;
        double gr = W.gs[i] + 2.0 * W.cq[12 + i % nu] * (absr ? W.z[i] : W.z[i] - prev);
        if (!absr && i + nu < n) gr -= 2.0 * W.cq[12 + i % nu] * (W.z[i + nu] - W.z[i]);
        W.gs[i] = sf * gr;
    }
    SC_SYNC();
}

// out = J' v for a row vector v (m):  G' (A' v) - v_hi + v_lo
// ycorr (optimal decay): the point-space vector subtracted from A' v before the G' product (Schur correction of the rhs)
template <int TH, bool OD = false>
__device__ __forceinline__ void lin_jt(const double* v, double* out, const LinMem& W, const LinDims& d, const LinConst& c,
                                    const double* G, int lane, bool ycorr = false) {
    const int N = d.N, K = d.K, n = d.n;
    for (int e = lane; e < 4 * N; e += TH) {
        const int pt = e >> 1, dd = e & 1, k = pt < N ? pt : pt - N;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) acc += v[k * K + j] * W.dh[2 * (pt * K + j) + dd];
        double yv = pt < N ? (OD ? W.w0k[k] : c.w0) * acc : acc;
        if constexpr (OD) {
            if (ycorr) yv -= W.Cv[4 * k + (pt < N ? dd : 2 + dd)] * (W.rr[k] * W.dinv[k]);
        }
        W.y[e] = yv;
    }
    SC_SYNC();
    for (int i = lane; i < n; i += TH) {
        double acc = 0.0;
#pragma unroll 8
        for (int r = 0; r < 4 * N; ++r) acc += G[(size_t)r * n + i] * W.y[r];
        out[i] = acc - v[d.mc + i] + v[d.mc + n + i];
    }
    SC_SYNC();
}

// M = sf Hc + G' T + diag(box) with v_mfma_f64_16x16x4_f64 (compile-time horizon).  Operand layout (MI355X guide, "f64
// MFMA"): lane l feeds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; result register r of lane l is
// D[row = (l >> 4) + 4 r][col = l & 15].  The contraction runs over the 4N point rows; k-step `st` is stage st (its a_k and
// b_k rows), so A[i][.] = G[row(st, q)][i] and B[.][j] = T[4 st + q][j].  Only the lower tiles are computed (T = Phi G with
// a symmetric Phi), and a row tile skips the stages whose points cannot depend on any of its columns (b_k depends on
// u_0..u_k): 28 MFMAs instead of 90 for n = 40.  `box` holds sig_hi + sig_lo per column.
typedef double lin_d4 __attribute__((ext_vector_type(4)));
// NT, NU > 0: compile-time horizon and input count (all operands of a tile are loaded first, then its MFMAs issue back to
// back); NT == 0: run-time sizes, operands in chunks of 4 stages.
// NW > 1: the tiles are dealt round-robin to the NW waves of the problem (tid = thread index among them).
// TILED (big layout): there is no T and no M in LDS -- the B operand T[4k + q][j] = sum_c Phi_k[q][c] G[row(k, c)][j] is formed
// from the stage block and four (cached) rows of G, and the lower tiles of M + delta I go straight into the tile-packed
// Cholesky storage W.L.
// FLY: the B operand formed on the fly with the output still in W.M (the lean layout: no T in LDS either).
template <int NT, int NU, int NW = 1, bool TILED = false, bool FLY = TILED>
__device__ __forceinline__ void lin_condense_mfma(const LinMem& W, const int N_rt, const int nu_rt, double sf, const double* Hc,
                                                  const double* G, const double* box, int tid, double delta = 0.0) {
    const int lane = tid & 63, wv = tid >> 6;
    int tile = 0;
    const int N = NT > 0 ? NT : N_rt, nu = NT > 0 ? NU : nu_rt, n = N * nu, nt = (n + 15) / 16;
    constexpr int S = NT > 0 ? ((FLY && !TILED && NT % 2 == 0) ? NT / 2 : NT) : 4;   // lean + on-the-fly operand: two chunks (all NT at once spills 100+ VGPRs)
    const int q = lane >> 4, l15 = lane & 15;
    for (int ti = 0; ti < nt; ++ti) {
        const int ia = 16 * ti + l15;
        const bool okA = ia < n;
        const int iac = okA ? ia : 0;
        const int k_lo = (16 * ti) / nu;
        for (int tj = 0; tj <= ti; ++tj) {
            if (NW > 1 && (tile++ & (NW - 1)) != wv) continue;
            const int jb = 16 * tj + l15;
            const bool okB = jb < n;
            const int jbc = okB ? jb : 0;
            lin_d4 acc = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = NT > 0 ? 0 : k_lo; k0 < N; k0 += S) {
                double a[S], b[S];
#pragma unroll
                for (int t = 0; t < S; ++t) {
                    const int k = k0 + t < N ? k0 + t : N - 1;
                    const int gr = q < 2 ? 2 * k + q : 2 * N + 2 * k + q - 2;
                    const double av = G[(size_t)gr * n + iac];
                    double bv;
                    if constexpr (FLY) {
                        const double* ph = W.Phi + 16 * k + 4 * q;
                        bv = ph[0] * G[(size_t)(2 * k) * n + jbc] + ph[1] * G[(size_t)(2 * k + 1) * n + jbc] +
                             ph[2] * G[(size_t)(2 * N + 2 * k) * n + jbc] + ph[3] * G[(size_t)(2 * N + 2 * k + 1) * n + jbc];
                    } else {
                        bv = W.T[(size_t)(4 * k + q) * n + jbc];
                    }
                    a[t] = okA ? av : 0.0;
                    b[t] = okB ? bv : 0.0;
                }
#pragma unroll
                for (int t = 0; t < S; ++t)
                    if (k0 + t >= k_lo && k0 + t < N) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + q + 4 * r, col = 16 * tj + l15;
                if (row < n && col < n) {
                    const double v = acc[r] + sf * Hc[(size_t)row * n + col] + (row == col ? box[row] : 0.0);
                    if constexpr (TILED) {
                        W.L[ipm::LdTile{}.at(row, col)] = v + (row == col ? delta : 0.0);   // (the strict upper part of a diagonal tile: don't-care)
                    } else {
                        W.M[(size_t)row * n + col] = v;
                        if (ti != tj) W.M[(size_t)col * n + row] = v;
                    }
                }
            }
        }
    }
}

#ifdef SC_LIN_PROF
#define LP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof[i] += (double)(t_ - tlast); tlast = t_; } while (0)
#else
#define LP(i) do { } while (0)
#endif

// NX, NU > 0: states and inputs are compile-time constants (the short matrix-vector loops unroll and their LDS loads are
// issued back to back; with run-time bounds every multiply-add waits a full LDS round trip); NT > 0 (needs NX > 0): the
// horizon too -- register Cholesky, lean LDS layout; KT > 0: obstacle rows.  0: run-time size.
// OD: optimal decay on a relative-degree-1 model (BASELINE config 5's extension, oracle/od_mpc_rd1.py): one decay variable
// rho_k per stage scales the gain of that stage's rows, h(b_k) - (1 - alpha rho_k) h(a_k) >= 0, at the price p_sb (rho_k - 1)^2;
// the input term is R u^2.  rho_k only meets the rows of stage k and enters them linearly, so its positive scalar block
// D_k = 2 sf p_sb + sum_j sig_kj (alpha h(a_kj))^2 is eliminated in POINT space: Phi_k -= C_k C_k' / D_k on the 4 x 4 stage block,
// the right-hand side loses G_k' C_k rr_k / D_k, and d rho_k = (rr_k - C_k' (G dz)_k) / D_k after the solve -- the condensed
// n x n system, its MFMA assembly and the Cholesky are untouched.
template <int NX, int NU, int NT, int KT, bool BIG = false, bool OD = false>
__global__ __launch_bounds__(BIG ? 256 : 64) __attribute__((amdgpu_waves_per_eu(BIG ? 2 : 1, BIG ? 2 : 8)))
void mpclin_kernel(const sc_mpclin_params p, const double* __restrict__ model, const long long B,
                                                    const int K_rt, const void* __restrict__ X, const void* __restrict__ u_prev,
                                                    const void* __restrict__ goal, const void* __restrict__ obs,
                                                    void* __restrict__ u_out, int* __restrict__ status_out,
                                                    int* __restrict__ iters_out, void* __restrict__ z_out,
                                                    void* __restrict__ rho_out, const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int NW = BIG ? 4 : 1, TH = 64 * NW;                 // threads per problem: the big layout leaves one problem per CU, so it takes all four SIMDs
    const int lane = threadIdx.x;                                 // index among the TH threads of the problem
    long long prob;
    if (!ipm::cont_problem(ct, B, prob)) return;                  // mpc_cont.hpp: block index, or an entry of the previous launch's queue
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };

    constexpr int NN = NT > 0 ? NT * NU : 0;                      // order of the condensed system when known at compile time
    LinDims d;
    d.N = NT > 0 ? NT : p.horizon; d.K = KT > 0 ? KT : K_rt; d.nx = NX > 0 ? NX : p.nx; d.nu = NX > 0 ? NU : p.nu;
    d.n = d.N * d.nu; d.mc = d.N * d.K; d.m = d.mc + 2 * d.n;
    const int K = d.K;
    const int N = d.N, nx = d.nx, nu = d.nu, n = d.n, m = d.m;
    constexpr bool LEAN = NT > 0;
    const LinMem W = carve_lin(sm, d, LEAN ? LIN_LEAN : (BIG ? LIN_BIG : LIN_STD), OD);
    Red R;
    R.buf = W.red; R.par = 0;
    LinConst c;
    c.w0 = -(1.0 - p.alpha); c.Rrob = p.robot_radius; c.beta = p.beta; c.circles_only = p.circles_only; c.abs_r = p.optimal_decay == 2;
    c.alpha = p.alpha; c.ps = p.od_p_sb; c.rf = p.od_omega_ref;
    if (lane < 12) W.cq[lane] = p.Q[lane];
    if (lane < 4) { W.cq[12 + lane] = p.R[lane]; W.cq[16 + lane] = p.u_lo[lane]; W.cq[20 + lane] = p.u_hi[lane]; }
    // model blob: Ae [nx nx] | Be [nx nu] | As2 [2 nx] | Bs2 [2 nu] | Hc [n n] | G [4N n]
    const int nmat = nx * nx + nx * nu + 2 * nx + 2 * nu;
    for (int e = lane; e < nmat; e += TH) W.Ae[e] = model[e];            // the four small matrices are contiguous in LDS too
    const double* __restrict__ Hcg = model + nmat;
    const double* __restrict__ Gg = Hcg + (size_t)n * n;
    const double* Hc = Hcg;
    if constexpr (!BIG) {
        for (int e = lane; e < 4 * N * n; e += TH) W.G[e] = Gg[e];
    }
    const double* G = BIG ? Gg : W.G;
    for (int i = lane; i < nx; i += TH) { W.xs[i] = ld(X, prob * nx + i); W.xg[i] = i < p.ng ? ld(goal, prob * p.ng + i) : 0.0; }
    for (int i = lane; i < nu; i += TH) W.up[i] = ld(u_prev, prob * nu + i);
    const size_t obase = p.obs_shared ? 0 : (size_t)prob * K * 7;
    for (int e = lane; e < K * 7; e += TH) W.obs[e] = ld(obs, obase + e);
    SC_SYNC();
    // the scalars of the interior-point loop (block-uniform); a continuation launch loads them with the arrays (mpc_cont.hpp)
    constexpr bool RESTO = !OD;
    const int nel = OD ? 0 : d.mc;                                      // elastic variables of the restoration
    double* const cst = ct.state ? ct.state + prob * ct.stride : nullptr;
    double f = 0.0, sf = 1.0, mu = p.mu_init;
    double nu_m = 10.0, delta_last = 0.0, e_best = 1e300;
    int n_acc = 0, it0 = 1;
    bool resto = false;
    int n_resto = 0, n_small = 0;                                       // n_small: consecutive tiny accepted steps at an infeasible z
    double theta_R = 0.0, mu_reg = mu;
    // stalled restorations (sc_resto_params.retry_max / stall_iter; oracle/mpc_cbf.py: solve)
    double delta_force = 0.0, theta_ref = 0.0;
    int n_retry = 0, n_stall = 0;
    if (ct.resume) {
        // the state a previous launch left: [scalars | z | zb | s | lam | obs | gs or clin | tel | rho | rhob]
        const double* a = cst + ipm::CONT_SCALARS;
        ipm::cont_copy(W.z, a, n, lane, TH); a += n;
        ipm::cont_copy(W.zb, a, n, lane, TH); a += n;
        ipm::cont_copy(W.s, a, m, lane, TH); a += m;
        ipm::cont_copy(W.lam, a, m, lane, TH); a += m;
        ipm::cont_copy(W.obs, a, K * 7, lane, TH); a += K * 7;
        ipm::cont_copy(LEAN ? W.clin : W.gs, a, n, lane, TH); a += n;
        if (nel) { ipm::cont_copy(W.tel, a, nel, lane, TH); a += nel; }
        if constexpr (OD) { ipm::cont_copy(W.rho, a, N, lane, TH); a += N; ipm::cont_copy(W.rhob, a, N, lane, TH); }
        it0 = (int)cst[0] + 1; mu = cst[1]; nu_m = cst[2]; delta_last = cst[3]; e_best = cst[4]; n_acc = (int)cst[5];
        resto = cst[6] != 0.0; n_resto = (int)cst[7]; n_small = (int)cst[8]; theta_R = cst[9]; mu_reg = cst[10]; sf = cst[11];
        delta_force = cst[12]; n_retry = (int)cst[13]; theta_ref = cst[14]; n_stall = (int)cst[15];
        SC_SYNC();
    } else {
    if (!c.circles_only) ipm::normalise_obstacle_flags(W.obs, K, lane, TH);
    SC_SYNC();
    // set_initial_guess (mpc_cbf.py:369): u_prev at every stage, pulled strictly inside the box
    for (int i = lane; i < n; i += TH) {
        const double lo = W.cq[16 + i % nu], hi = W.cq[20 + i % nu], pad = 0.005 * (hi - lo);
        W.z[i] = fmin(fmax(W.up[i % nu], lo + pad), hi - pad);
    }
    if constexpr (OD) for (int k = lane; k < N; k += TH) { W.rho[k] = c.rf; W.rhob[k] = c.rf; }   // decay variables start at their reference
    SC_SYNC();

    f = lin_eval<TH, OD>(W.z, W.rho, W, d, c, lane, true, R);
    if (ct.it_stop < 0) {
        // classify only (mpc_cont.hpp): is a CBF row violated at the initial guess?
        double th0 = 0.0;
        for (int i = lane; i < d.mc; i += TH) th0 += fmax(0.0, -W.g[i]);
        th0 = lsum<TH>(th0, R);
        if (lane == 0) ipm::cont_push(ct, prob, th0 > 0.0);
        return;
    }
    // steep (superellipsoid) barriers: IPOPT-style gradient-based row scaling from the initial guess, then a fresh evaluation
    if (!c.circles_only &&
        ipm::scale_steep_barriers(W.obs, K, W.dh, 2 * N, lane, TH, [&](double v) { return lmax_<TH>(v, R); }, [] { SC_SYNC(); }))
        f = lin_eval<TH, OD>(W.z, W.rho, W, d, c, lane, true, R);
    lin_grad<TH, OD>(W, d, c, lane, 1.0);
    if constexpr (LEAN) {
        // the cost is quadratic: grad f = Hc z + c with c fixed for the solve (one adjoint pass, here)
        for (int i = lane; i < n; i += TH) {
            double q = 0.0;
#pragma unroll
            for (int j = 0; j < n; ++j) q += Hc[(size_t)j * n + i] * W.z[j];
            W.clin[i] = W.gs[i] - q;
        }
        SC_SYNC();
    }
    double gmax = 0.0;
    for (int i = lane; i < n; i += TH) gmax = fmax(gmax, fabs(W.gs[i]));
    gmax = lmax_<TH>(gmax, R);
    sf = fmin(1.0, 100.0 / fmax(1e-12, gmax));                          // objective scaling
    for (int i = lane; i < m; i += TH) { const double s = fmax(W.g[i], 1e-2); W.s[i] = s; W.lam[i] = mu / s; }
    for (int i = lane; i < n; i += TH) W.zb[i] = W.z[i];
    SC_SYNC();
    }

    int status = SC_STATUS_INACCURATE, it = 0;
    const double tau = 0.995;
    bool fresh = false;
    const int acc_iter = p.acceptable_iter > 0 ? p.acceptable_iter : 15;
    // feasibility restoration (mpc_ipm_common.hpp; oracle/mpc_cbf.py: solve): block-uniform state
    bool regrad = false, pending = false;
    const double rho_R = p.resto.rho;
#ifdef SC_LIN_PROF
    double prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    for (it = it0; it <= p.max_iter; ++it) {
        LP(11);
        if (cst && it > ct.it_stop) { pending = true; break; }            // the cap of this launch: the solve goes on in the next one
        // fresh: the accepted trial point was evaluated with derivatives (a continuation launch evaluates: same point, same values)
        if ((it > 1 && !fresh) || (ct.resume && it == it0)) f = lin_eval<TH, OD>(W.z, W.rho, W, d, c, lane, true, R);
        LP(0);
        double theta = 0.0;                                               // l1 violation of the elastic (CBF) rows at z
        if constexpr (RESTO) {
            for (int i = lane; i < d.mc; i += TH) theta += fmax(0.0, -W.g[i]);
            theta = lsum<TH>(theta, R);
            if (resto && theta <= fmax(n_resto == 1 ? p.resto.kappa * theta_R : 0.0, p.resto.theta_tol)) {
                // enough of the violation is gone (first entry: a tenth of it; later entries run until nothing is left -- the regular
            // phase came back to the same stall): a fresh start of the regular phase at this z with the barrier parameter it left with
                resto = false; mu = mu_reg; regrad = true;
                for (int i = lane; i < m; i += TH) { const double s0 = fmax(W.g[i], 1e-2); W.s[i] = s0; W.lam[i] = mu / s0; }
                for (int i = lane; i < n; i += TH) W.zb[i] = W.z[i];
                nu_m = 10.0; n_acc = 0; e_best = 1e300;
                SC_SYNC();
            }
        }
        double zeta = (RESTO && resto) ? sqrt(mu) : 0.0;
        const double sfe = (RESTO && resto) ? 0.0 : sf;
        if (RESTO && resto) {                                             // objective of the restoration: zeta/2 |z - z_R|^2, z_R in W.zb
            for (int i = lane; i < n; i += TH) W.gs[i] = zeta * (W.z[i] - W.zb[i]);
            SC_SYNC();
        } else if constexpr (LEAN) {                                      // gs = sf grad f = sf (Hc z + c)
            for (int i = lane; i < n; i += TH) {
                double q = W.clin[i];
#pragma unroll
                for (int j = 0; j < n; ++j) q += Hc[(size_t)j * n + i] * W.z[j];
                W.gs[i] = sf * q;
            }
            SC_SYNC();
        } else if (it == 1 || regrad) {
            lin_grad<TH, OD>(W, d, c, lane, sf);                            // later iterations: gs += alpha sf Hc dz at the update (the cost is quadratic)
            regrad = false;
        }
        LP(1);
        lin_jt<TH, OD>(W.lam, W.rd, W, d, c, G, lane);
        LP(2);
        double e_d = 0.0, e_p = 0.0, e_c0 = 0.0, lmx = 0.0;
        for (int i = lane; i < n; i += TH) { const double r = W.gs[i] - W.rd[i]; W.rd[i] = r; e_d = fmax(e_d, fabs(r)); }
        if constexpr (OD) {
            for (int k = lane; k < N; k += TH) {                         // r_d of rho_k = sf 2 p_sb (rho_k - ref) - sum_j lam_kj alpha h(a_kj)
                double acc = sf * 2.0 * c.ps * (W.rho[k] - c.rf);
#pragma unroll
                for (int j = 0; j < K; ++j) acc -= W.lam[k * K + j] * c.alpha * W.hk[k * K + j];
                W.rdr[k] = acc;
                e_d = fmax(e_d, fabs(acc));
            }
        }
        for (int i = lane; i < m; i += TH) {
            const double s = W.s[i], l = W.lam[i];
            double rp = W.g[i] - s;
            if (RESTO && resto && i < d.mc) { const double t = W.tel[i]; rp += t; e_c0 = fmax(e_c0, fabs(t * (rho_R - l))); }
            e_p = fmax(e_p, fabs(rp)); e_c0 = fmax(e_c0, fabs(s * l)); lmx = fmax(lmx, l);
        }
        e_d = lmax_<TH>(e_d, R); e_p = lmax_<TH>(e_p, R); e_c0 = lmax_<TH>(e_c0, R); lmx = lmax_<TH>(lmx, R);
        const double e_opt = fmax(e_d, fmax(e_p, e_c0));
        if (!resto && e_opt < e_best) {
            e_best = e_opt;
            for (int i = lane; i < n; i += TH) W.zb[i] = W.z[i];
            if constexpr (OD) for (int k = lane; k < N; k += TH) W.rhob[k] = W.rho[k];
        }
        if (resto) {
            // A stationary point of the violation.  The restoration's KKT error is in units of its objective rho theta, so |grad theta|
            // <= e_opt / rho; over the input box (a few units across) theta cannot fall by more than ~10 e_opt / rho from here: the
            // certificate asks for more violation than that.
            if (e_opt <= p.resto.tol && theta > fmax(p.resto.theta_tol, 10.0 * e_opt / rho_R)) { status = SC_STATUS_INFEASIBLE; break; }
            if (e_opt <= p.tol) { break; }                             // solved, and (nearly) no violation left: nothing to certify
            if (p.resto.stall_iter > 0) {
                // no 1 % less violation within stall_iter iterations and violation left: a local minimiser of the violation at a kink
                const bool less = theta <= 0.99 * theta_ref;                  // (selects, not branches: the values are wave-uniform)
                theta_ref = less ? theta : theta_ref;
                n_stall = less ? 0 : n_stall + 1;
                if (n_stall >= p.resto.stall_iter) { if (theta > p.resto.stall_theta) status = SC_STATUS_INFEASIBLE; break; }   // (less violation: SC_STATUS_INACCURATE)
            }
        } else if (e_opt <= p.tol) {
            status = SC_STATUS_OPTIMAL;
            break;
        }
        n_acc = e_opt <= p.acceptable_tol ? n_acc + 1 : 0;
        if (n_acc >= acc_iter) {
            if (resto && theta > p.resto.theta_tol) status = SC_STATUS_INFEASIBLE;
            break;
        }
        if (!(e_opt < 1e300)) break;                                      // non-finite data: give up
        bool want_resto = RESTO && !resto && lmx > 1e10;                  // multipliers diverge: locally infeasible
        if (!RESTO && lmx > 1e10) { status = SC_STATUS_INFEASIBLE; break; }
        bool accepted = false;
        double alpha = 0.0, ad = 0.0;
        if (!want_resto) {
        const double mu_old = mu;
        for (;;) {                                                        // barrier update
            double e_c = 0.0;
            for (int i = lane; i < m; i += TH) {
                const double l = W.lam[i];
                e_c = fmax(e_c, fabs(W.s[i] * l - mu));
                if (RESTO && resto && i < d.mc) e_c = fmax(e_c, fabs(W.tel[i] * (rho_R - l) - mu));
            }
            e_c = lmax_<TH>(e_c, R);
            const double e_mu = fmax(e_d, fmax(e_p, e_c));
            if (e_mu <= 10.0 * mu && mu > p.mu_min) mu = fmax(p.mu_min, fmin(0.2 * mu, mu * sqrt(mu)));
            else break;
        }
        if (RESTO && resto && mu != mu_old) {                             // zeta = sqrt(mu): the proximity term follows the new mu
            zeta = sqrt(mu);
            for (int i = lane; i < n; i += TH) W.gs[i] = zeta * (W.z[i] - W.zb[i]);
        }
        LP(3);
        // rhs = -sf grad f + J' (mu / s - sig r_p);  elastic rows of the restoration: lam + dl0 and Sigma_eff (ipm::resto_row)
        for (int i = lane; i < m; i += TH) {
            const double s = W.s[i], l = W.lam[i];
            if (RESTO && resto && i < d.mc) {
                double rp, ise, vbe;
                ipm::resto_row(W.g[i], s, l, W.tel[i], mu, rho_R, rp, ise, vbe);
                W.vb[i] = l + (mu * ise - vbe);
                W.ds[i] = l * ise;
            } else {
                const double is = rcp_(s), sig = l * is;                  // v_rcp seed + two Newton steps (sc_qp2.hpp), as kernel 3
                W.vb[i] = mu * is - sig * (W.g[i] - s);
                W.ds[i] = sig;                                            // read by the Phi blocks below; ds proper is written after the solve
            }
        }
        SC_SYNC();
        if constexpr (OD) {
            // decay blocks of the stages: C_k (over the point pair: [w0 dh_a; dh_b] sig A - lam alpha [dh_a; 0], A = alpha h(a)),
            // 1 / D_k, and the right-hand side rr_k = -sf 2 p_sb (rho_k - ref) + sum_j vb_kj A_kj
            for (int e = lane; e < 5 * N; e += TH) {
                const int k = e / 5, r = e - 5 * k;
                double acc = 0.0, acc2 = 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int ea = k * K + j, eb = (N + k) * K + j;
                    const double A = c.alpha * W.hk[ea], sig = W.ds[ea], l = W.lam[ea];
                    if (r == 4) { acc += sig * A * A; acc2 += W.vb[ea] * A; }
                    else if (r < 2) acc += (sig * A * W.w0k[k] - l * c.alpha) * W.dh[2 * ea + r];
                    else acc += sig * A * W.dh[2 * eb + r - 2];
                }
                if (r == 4) { W.dinv[k] = 1.0 / (sf * 2.0 * c.ps + acc); W.rr[k] = -sf * 2.0 * c.ps * (W.rho[k] - c.rf) + acc2; }
                else W.Cv[4 * k + r] = acc;
            }
            SC_SYNC();
        }
        lin_jt<TH, OD>(W.vb, W.rhs, W, d, c, G, lane, true);
        for (int i = lane; i < n; i += TH) W.rhs[i] = -W.gs[i] + W.rhs[i];
        LP(4);
        // stage blocks Phi_k over (a_k, b_k):  sum_j sig_kj v v' (v = [w0 dh_a; dh_b])  -  sum_j lam_kj [w0 Hh_a, 0; 0, Hh_b]
        for (int e = lane; e < 16 * N; e += TH) {
            const int k = e >> 4, r = (e >> 2) & 3, cc = e & 3;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int ea = k * K + j, eb = (N + k) * K + j, row = k * K + j;
                const double l = W.lam[row], sig = W.ds[row];
                const double w0 = OD ? W.w0k[k] : c.w0;
                const double vr = r < 2 ? w0 * W.dh[2 * ea + r] : W.dh[2 * eb + r - 2];
                const double vc = cc < 2 ? w0 * W.dh[2 * ea + cc] : W.dh[2 * eb + cc - 2];
                acc += sig * vr * vc;
                if (r < 2 && cc < 2) acc -= l * w0 * W.hh[3 * ea + r + cc];
                if (r >= 2 && cc >= 2) acc -= l * W.hh[3 * eb + (r - 2) + (cc - 2)];
            }
            if constexpr (OD) acc -= W.Cv[4 * k + r] * W.Cv[4 * k + cc] * W.dinv[k];   // Schur complement of the stage's decay variable
            W.Phi[e] = acc;
        }
        SC_SYNC();
        LP(5);
        // T = Phi G (rows 4k..4k+3), then M = sf Hc + G' T + diag(sig_hi + sig_lo)
        if constexpr (!BIG && !LEAN)
        for (int e = lane; e < 4 * N * n; e += TH) {
            const int row = e / n, col = e - row * n, k = row >> 2, r = row & 3;
            double acc = 0.0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int gr = cc < 2 ? 2 * k + cc : 2 * N + 2 * k + cc - 2;
                acc += W.Phi[16 * k + 4 * r + cc] * G[(size_t)gr * n + col];
            }
            W.T[e] = acc;
        }
        SC_SYNC();
        for (int i = lane; i < n; i += TH)                                   // r_d is consumed: its space holds the box terms
            W.rd[i] = W.ds[d.mc + i] + W.ds[d.mc + n + i];                    // sigma of the two box rows (stored above)
        SC_SYNC();
        if constexpr (!BIG) {
            lin_condense_mfma<NT, NU, NW, false, LEAN>(W, N, nu, sfe, Hc, G, W.rd, lane);
            SC_SYNC();
        }
        LP(6);
        // inertia correction: M + delta I until the Cholesky succeeds (restoration: + zeta I, the proximity term)
        double delta = delta_force;                                      // 0 unless a failed restoration step is being retried
        bool ok = false;
        for (int t = 0; t < 40 && !ok; ++t) {
            if constexpr (NN > 0) {
                ok = ipm::chol_reg_solve<NN>(W.M, W.rhs, W.L, W.dz, delta + zeta, lane);
            } else if constexpr (BIG) {
                lin_condense_mfma<NT, NU, NW, true>(W, N, nu, sfe, Hc, G, W.rd, lane, delta + zeta);   // M + delta I, lower tiles, into the factor's storage
                SC_SYNC();
                ok = ipm::cholesky_ix<NW, ipm::LdTile>(W.L, n, ipm::LdTile{}, lane, W.red + 8);
            } else {
                for (int r = lane >> 6; r < n; r += NW)                      // lower triangle, row stride n | 1 (odd: no LDS bank conflicts)
                    for (int cc = lane & 63; cc <= r; cc += 64) W.L[r * (n | 1) + cc] = W.M[r * n + cc] + (cc == r ? delta + zeta : 0.0);
                SC_SYNC();
                ok = cholesky_lds<NW>(W.L, n, n | 1, lane, W.red + 8);
            }
            if (!ok) delta = (delta == 0.0) ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
        }
        if (!ok) break;
        if (delta > 0.0) delta_last = delta;
        if constexpr (NN == 0) {
            for (int i = lane; i < n; i += TH) W.dz[i] = W.rhs[i];
            SC_SYNC();
            if constexpr (BIG) ipm::chol_solve_ix<NW, ipm::LdTile>(W.L, W.dz, n, ipm::LdTile{}, lane);
            else chol_solve_lds<NW>(W.L, W.dz, n, n | 1, lane);
        }
        LP(7);
        // point displacements  G dz, then ds = J dz + r_p, dlam, step lengths
        for (int r = lane; r < 4 * N; r += TH) {
            double acc = 0.0;
#pragma unroll 8
            for (int i = 0; i < n; ++i) acc += G[(size_t)r * n + i] * W.dz[i];
            W.pdz[r] = acc;
        }
        // sf grad f . dz and the curvature sf dz' Hc dz: the cost is exactly quadratic in z, so the line search takes
        // f(z + a dz) - f(z) = a grad.dz + a^2/2 dz' Hc dz instead of the difference of two sums of size |f|
        // (restoration: the proximity term is quadratic too, curvature zeta |dz|^2)
        double gdz = 0.0, curv = 0.0;
        for (int i = lane; i < n; i += TH) {
            gdz += W.gs[i] * W.dz[i];
            if (RESTO && resto) { curv += W.dz[i] * W.dz[i]; continue; }
            double q = 0.0;
#pragma unroll 8
            for (int j = 0; j < n; ++j) q += Hc[(size_t)j * n + i] * W.dz[j];
            curv += q * W.dz[i];
            if constexpr (!LEAN) W.rd[i] = q;                               // (Hc dz)_i: r_d's space is free until the next iteration
        }
        if constexpr (OD) {
            // d rho_k = (rr_k - C_k' (G dz)_k) / D_k; the decay part of grad f . d and of the (exactly quadratic) curvature
            for (int k = lane; k < N; k += TH) {
                const double dr = (W.rr[k] - (W.Cv[4 * k] * W.pdz[2 * k] + W.Cv[4 * k + 1] * W.pdz[2 * k + 1] +
                                              W.Cv[4 * k + 2] * W.pdz[2 * N + 2 * k] + W.Cv[4 * k + 3] * W.pdz[2 * N + 2 * k + 1])) * W.dinv[k];
                W.drho[k] = dr;
                gdz += sf * 2.0 * c.ps * (W.rho[k] - c.rf) * dr;
                curv += 2.0 * c.ps * dr * dr;
            }
        }
        curv = ((RESTO && resto) ? zeta : sf) * lsum<TH>(curv, R);
        SC_SYNC();
        double rs_min = 0.0, rl_min = 0.0, sum_ds_s = 0.0, sum_rp = 0.0, sum_log = 0.0, sum_g = 0.0, sum_t = 0.0, sum_dt = 0.0;
        for (int i = lane; i < m; i += TH) {
            const double s = W.s[i], l = W.lam[i];
            double jd;
            if (i < d.mc) {
                const int k = i / K, j = i - k * K, ea = k * K + j, eb = (N + k) * K + j;
                jd = (OD ? W.w0k[k] : c.w0) * (W.dh[2 * ea] * W.pdz[2 * k] + W.dh[2 * ea + 1] * W.pdz[2 * k + 1]) +
                     (W.dh[2 * eb] * W.pdz[2 * N + 2 * k] + W.dh[2 * eb + 1] * W.pdz[2 * N + 2 * k + 1]);
                if constexpr (OD) jd += c.alpha * W.hk[ea] * W.drho[k];
            } else if (i < d.mc + n) {
                jd = -W.dz[i - d.mc];
            } else {
                jd = W.dz[i - d.mc - n];
            }
            const double is = rcp_(s);
            double dsi, dl, rp;
            if (RESTO && resto && i < d.mc) {
                // elastic row: dlam = -Sigma_eff J dz + dl0 (W.ds still holds Sigma_eff, W.vb = lam + dl0), dt from dlam
                const double t = W.tel[i];
                rp = W.g[i] + t - s;
                dl = -W.ds[i] * jd + (W.vb[i] - l);
                const double dt = ipm::resto_dt(l, t, dl, mu, rho_R);
                dsi = jd + dt + rp;
                const double rt = dt * rcp_(t);
                rs_min = fmin(rs_min, rt); rl_min = fmin(rl_min, -dl * rcp_(rho_R - l));
                sum_ds_s += rt; sum_t += t; sum_dt += dt; sum_log += log(t);
            } else {
                rp = W.g[i] - s;
                dsi = jd + rp;
                dl = -(l * is) * dsi - (l - mu * is);
            }
            const double rs = dsi * is, rl = dl * rcp_(l);
            rs_min = fmin(rs_min, rs); rl_min = fmin(rl_min, rl);
            sum_ds_s += rs; sum_rp += fabs(rp); sum_log += log(s); sum_g += fabs(W.g[i]);
            W.ds[i] = dsi; W.dlam[i] = dl;
        }
        rs_min = lmin<TH>(rs_min, R); rl_min = lmin<TH>(rl_min, R); sum_ds_s = lsum<TH>(sum_ds_s, R); sum_rp = lsum<TH>(sum_rp, R); sum_log = lsum<TH>(sum_log, R);
        gdz = lsum<TH>(gdz, R); sum_g = lsum<TH>(sum_g, R);
        const double ap = rs_min < 0.0 ? fmin(1.0, -tau / rs_min) : 1.0;
        ad = rl_min < 0.0 ? fmin(1.0, -tau / rl_min) : 1.0;
        nu_m = fmax(nu_m, 1.1 * lmx);
        double bar0 = sfe * f - mu * sum_log, dbar = gdz - mu * sum_ds_s, lin_t = 0.0;
        if (RESTO && resto) {
            double prox = 0.0;
            for (int i = lane; i < n; i += TH) { const double dzr = W.z[i] - W.zb[i]; prox += dzr * dzr; }
            prox = lsum<TH>(prox, R); sum_t = lsum<TH>(sum_t, R); sum_dt = lsum<TH>(sum_dt, R);
            bar0 = 0.5 * zeta * prox + rho_R * sum_t - mu * sum_log;
            lin_t = rho_R * sum_dt;                                       // rho sum t is linear along the step
            dbar += lin_t;
        }
        // not a descent direction of the merit function (the penalty is below the multipliers of the step): raise the penalty so that
        // the directional derivative is -0.1 nu |r_p|_1  (Nocedal & Wright (18.36))
        if (dbar - nu_m * sum_rp >= 0.0 && sum_rp > 0.0) nu_m = dbar / (0.9 * sum_rp);
        const double phi0 = bar0 + nu_m * sum_rp;
        const double dphi = dbar - nu_m * sum_rp;
        // round-off of the constraint part of the merit: a far-away dummy obstacle row has h ~ 2e6 (oracle: row_noise)
        const double noise_rows = 1e-15 * nu_m * sum_g;
        LP(8);
        alpha = ap;
        fresh = false;
        // slack reset of the line search: restoration sc_resto_params.slack_reset, regular phase sc_mpclin_params.slack_reset = 2
        const bool rreset = (RESTO && resto) ? p.resto.slack_reset != 0 : p.slack_reset == 2;
        const double thr_reset = mu * rcp_(nu_m);
        for (int ls = 0; ls < 12; ++ls) {
            for (int i = lane; i < n; i += TH) W.zt[i] = W.z[i] + alpha * W.dz[i];
            if constexpr (OD) for (int k = lane; k < N; k += TH) W.rhot[k] = W.rho[k] + alpha * W.drho[k];
            SC_SYNC();
            const bool full = ls == 0 && c.circles_only != 0;                  // the full step is taken most of the time: evaluate it once, with the
            const double f_t = lin_eval<TH, OD>(W.zt, W.rhot, W, d, c, lane, full, R);     // derivatives (circles: they cost nothing next to the rollout)
            double srp = 0.0, slog = 0.0;
            for (int i = lane; i < m; i += TH) {
                const double s_lin = W.s[i] + alpha * W.ds[i];
                double tot = W.g[i];
                if (RESTO && resto && i < d.mc) {
                    const double t = W.tel[i];
                    const double t_t = t + alpha * ipm::resto_dt(W.lam[i], t, W.dlam[i], mu, rho_R);
                    slog += log(t_t); tot += t_t;
                }
                // slack reset (both phases; t = 0 in the regular one): s = g + t where that is >= mu / nu
                const double s_t = (rreset && tot >= thr_reset) ? tot : s_lin;
                slog += log(s_t);
                srp += fabs(tot - s_t);
            }
            slog = lsum<TH>(slog, R); srp = lsum<TH>(srp, R);
            const double phit = phi0 + alpha * (gdz + lin_t) + 0.5 * alpha * alpha * curv - mu * (slog - sum_log) + nu_m * (srp - sum_rp);
            if (phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * fabs(phi0) + noise_rows) {
                accepted = true; fresh = full;
                if (fresh) f = f_t;
                break;
            }
            alpha *= 0.5;
        }
        LP(9);
        if (!accepted) {
            if (!RESTO) break;
            if (resto) {
                if (n_retry >= p.resto.retry_max) break;
                // the same z again, Levenberg-damped; the retry is an iteration of its own (LDS holds the last trial point's rows)
                ++n_retry; delta_force = fmax(1.0, 100.0 * fmax(delta_force, delta));
                SC_SYNC();
                f = lin_eval<TH, OD>(W.z, W.rho, W, d, c, lane, true, R);
                fresh = true;
                SC_SYNC();
                continue;
            }
            want_resto = true;
        } else if (RESTO && !resto) {
            // IPOPT hands over to the restoration when the step length falls below its alpha_min; here: small_iter consecutive
            // accepted steps shorter than small_alpha at an infeasible iterate (the accepted step is then not taken)
            n_small = (alpha < p.resto.small_alpha && theta > p.resto.theta_tol) ? n_small + 1 : 0;
            if (n_small >= p.resto.small_iter && n_resto < p.resto.max_entries && e_best > p.acceptable_tol) want_resto = true;
        }
        }
        if (want_resto) {
            // the regular phase cannot continue from z.  Nothing to restore at a feasible point or once the restoration has been
            // entered max_entries times
            if (e_best <= p.acceptable_tol || theta <= p.resto.theta_tol || n_resto >= p.resto.max_entries) break;
            SC_SYNC();
            f = lin_eval<TH, OD>(W.z, W.rho, W, d, c, lane, true, R);       // the rows and derivatives in LDS are the last trial point's
            fresh = true;
            resto = true; ++n_resto; n_small = 0; theta_R = theta; mu_reg = mu;
            delta_force = 0.0; n_retry = 0; theta_ref = theta; n_stall = 0;
            double vmax = 0.0;
            for (int i = lane; i < d.mc; i += TH) vmax = fmax(vmax, -W.g[i]);
            mu = fmax(mu, lmax_<TH>(vmax, R));                              // IPOPT: mu_R = max(mu, |c|_inf)
            for (int i = lane; i < m; i += TH) {
                // elastic rows start on their central path, the box rows like at the start of the solve
                const double gi = W.g[i];
                const double s0 = i < d.mc ? ipm::resto_central_slack(gi, mu, rho_R) : fmax(gi, 1e-2);
                if (i < d.mc) W.tel[i] = s0 - gi;
                W.s[i] = s0; W.lam[i] = mu / s0;
            }
            for (int i = lane; i < n; i += TH) W.zb[i] = W.z[i];            // z_R
            nu_m = 10.0; n_acc = 0;
            SC_SYNC();
            continue;
        }
        for (int i = lane; i < n; i += TH) {
            W.z[i] = W.z[i] + alpha * W.dz[i];
            if constexpr (!LEAN) { if (!(RESTO && resto)) W.gs[i] += alpha * sf * W.rd[i]; }
        }
        if constexpr (OD) for (int k = lane; k < N; k += TH) W.rho[k] = W.rho[k] + alpha * W.drho[k];
        delta_force = 0.0; n_retry = 0;
        {
        const bool rreset = (RESTO && resto) ? p.resto.slack_reset != 0 : p.slack_reset == 2;   // W.g holds the accepted trial point's rows
        const double thr_reset = mu * rcp_(nu_m);
        for (int i = lane; i < m; i += TH) {
            const double s_lin = W.s[i] + alpha * W.ds[i];
            const double l0 = W.lam[i], dl = W.dlam[i];
            double tn = 0.0;
            const bool el = RESTO && resto && i < d.mc;
            if (el) { const double t = W.tel[i]; tn = t + alpha * ipm::resto_dt(l0, t, dl, mu, rho_R); W.tel[i] = tn; }
            const double tot = W.g[i] + tn;
            const double s = (rreset && tot >= thr_reset) ? tot : s_lin;
            double l = l0 + ad * dl;
            const double mus = mu * rcp_(s);
            l = fmin(fmax(l, 1e-10 * mus), 1e10 * mus);                   // IPOPT eq. (16) safeguard
            if (el) l = ipm::resto_clamp_lam(l, tn, mu, rho_R);
            W.s[i] = s; W.lam[i] = l;
        }
        }
        SC_SYNC();
    }
    if (pending) {
        // hand-over (mpc_cont.hpp); W.g holds the rows of the current z on every path to the top of the loop
        SC_SYNC();
        double th = 0.0;
        for (int i = lane; i < d.mc; i += TH) th += fmax(0.0, -W.g[i]);
        th = lsum<TH>(th, R);
        double* a = cst + ipm::CONT_SCALARS;
        ipm::cont_copy(a, W.z, n, lane, TH); a += n;
        ipm::cont_copy(a, W.zb, n, lane, TH); a += n;
        ipm::cont_copy(a, W.s, m, lane, TH); a += m;
        ipm::cont_copy(a, W.lam, m, lane, TH); a += m;
        ipm::cont_copy(a, W.obs, K * 7, lane, TH); a += K * 7;
        ipm::cont_copy(a, LEAN ? W.clin : W.gs, n, lane, TH); a += n;
        if (nel) { ipm::cont_copy(a, W.tel, nel, lane, TH); a += nel; }
        if constexpr (OD) { ipm::cont_copy(a, W.rho, N, lane, TH); a += N; ipm::cont_copy(a, W.rhob, N, lane, TH); }
        if (lane == 0) {
            cst[0] = (double)(it - 1); cst[1] = mu; cst[2] = nu_m; cst[3] = delta_last; cst[4] = e_best; cst[5] = (double)n_acc;
            cst[6] = resto ? 1.0 : 0.0; cst[7] = (double)n_resto; cst[8] = (double)n_small; cst[9] = theta_R; cst[10] = mu_reg; cst[11] = sf;
            cst[12] = delta_force; cst[13] = (double)n_retry; cst[14] = theta_ref; cst[15] = (double)n_stall;
            status_out[prob] = SC_STATUS_PENDING_MPC;
            if (iters_out) iters_out[prob] = it - 1;
            ipm::cont_push(ct, prob, th > p.resto.theta_tol);
        }
        return;
    }
    if (it > p.max_iter) it = p.max_iter;
    if (status == SC_STATUS_INACCURATE && !resto && e_best <= p.acceptable_tol) {
        SC_SYNC();
        for (int i = lane; i < n; i += TH) W.z[i] = W.zb[i];
        if constexpr (OD) for (int k = lane; k < N; k += TH) W.rho[k] = W.rhob[k];
        status = SC_STATUS_OPTIMAL;
    }
    SC_SYNC();
    if constexpr (OD) {
        // optimal decay has no restoration phase: "infeasible" there still means "stopped at an infeasible iterate"
        lin_eval<TH, OD>(W.z, W.rho, W, d, c, lane, false, R);
        if (status != SC_STATUS_OPTIMAL) {
            double gmin = 1e300;
            for (int i = lane; i < m; i += TH) gmin = fmin(gmin, W.g[i]);
            gmin = lmin<TH>(gmin, R);
            if (gmin < -1e-6) status = SC_STATUS_INFEASIBLE;
        }
    }
    if (lane < nu) st(u_out, prob * nu + lane, W.z[lane]);
    if (lane == 0) {
        status_out[prob] = status;
        if (iters_out) iters_out[prob] = it;
    }
#ifdef SC_LIN_PROF
    if (z_out && lane == 0) for (int i = 0; i < 12; ++i) st(z_out, prob * n + i, prof[i]);
#else
    if (z_out) for (int i = lane; i < n; i += TH) st(z_out, prob * n + i, W.z[i]);
#endif
    if constexpr (OD) {
        if (rho_out) for (int k = lane; k < N; k += TH) st(rho_out, prob * N + k, W.rho[k]);
    }
}

}  // namespace

// the compile-time instantiations: both models at the reference's default horizon 10 (mpc_cbf.py:15), and the other
// horizons whose order n = N nu still fits the register Cholesky (n <= 64): SingleIntegrator2D N = 20, Quad3D N = 16
static bool mpclin_is_lean(int N, int nx, int nu) {
    if (nx == 12 && nu == 4) return N == 10 || N == 16;
    if (nx == 2 && nu == 2) return N == 10 || N == 20;
    return false;
}
static int mpclin_mode(int N, int K, int nx, int nu, bool od = false) {
    if (od && !(nx == 12 && nu == 4 && N == 10)) {                     // optimal decay: Quad3D, lean at the default horizon only
        return mpclin_lds_doubles(N, K, nx, nu, LIN_STD, true) * sizeof(double) > 80 * 1024 ? LIN_BIG : LIN_STD;
    }
    if (mpclin_is_lean(N, nx, nu) && mpclin_lds_doubles(N, K, nx, nu, LIN_LEAN, od) * sizeof(double) <= 160 * 1024) return LIN_LEAN;
    // the standard layout above 80 KB leaves one single-wave problem per CU (three SIMDs idle): the big layout with its four
    // waves per problem is faster from there on (measured, 4096 Quad3D problems: N = 12 15.9 -> 11.9 ms, N = 14 25.0 -> 17.4 ms;
    // SingleIntegrator2D N = 18 27.5 -> 20.4 ms; below, two or three single-wave problems per CU win: Quad3D N = 8 5.2 against 6.5 ms)
    return mpclin_lds_doubles(N, K, nx, nu, LIN_STD) * sizeof(double) > 80 * 1024 ? LIN_BIG : LIN_STD;
}
size_t mpclin_lds_bytes(int N, int K, int nx, int nu, bool od) { return mpclin_lds_doubles(N, K, nx, nu, mpclin_mode(N, K, nx, nu, od), od) * sizeof(double); }

// Host: constant matrices of the condensed problem from (Ae, Be, As, Bs, Q, R, N); layout of the blob as the kernel reads it
size_t mpclin_model_doubles(int nx, int nu, int N) {
    const size_t n = (size_t)N * nu;
    return (size_t)nx * nx + (size_t)nx * nu + 2 * nx + 2 * nu + n * n + 4 * (size_t)N * n;
}

bool mpclin_build_model(const sc_mpclin_params& p, const double* Ae, const double* Be, const double* As, const double* Bs,
                        double* out) {
    const int nx = p.nx, nu = p.nu, N = p.horizon, n = N * nu;
    double* o = out;
    for (int i = 0; i < nx * nx; ++i) *o++ = Ae[i];
    for (int i = 0; i < nx * nu; ++i) *o++ = Be[i];
    for (int i = 0; i < 2 * nx; ++i) *o++ = As[i];                        // rows 0..1 of As (the planar position)
    for (int i = 0; i < 2 * nu; ++i) *o++ = Bs[i];
    double* Hc = o;
    double* G = Hc + (size_t)n * n;
    // Phi_k = d x_k / d z (nx x n), Phi_0 = 0, Phi_{k+1} = Ae Phi_k + Be E_k
    double* Phi = static_cast<double*>(std::calloc((size_t)(N + 1) * nx * n, sizeof(double)));   // no exception may cross the C-ABI
    if (!Phi) return false;
    for (int k = 0; k < N; ++k) {
        const double* Pk = Phi + (size_t)k * nx * n;
        double* Pn = Phi + (size_t)(k + 1) * nx * n;
        for (int i = 0; i < nx; ++i)
            for (int col = 0; col < n; ++col) {
                double acc = 0.0;
                for (int j = 0; j < nx; ++j) acc += Ae[i * nx + j] * Pk[(size_t)j * n + col];
                Pn[(size_t)i * n + col] = acc;
            }
        for (int i = 0; i < nx; ++i)
            for (int cu = 0; cu < nu; ++cu) Pn[(size_t)i * n + k * nu + cu] += Be[i * nu + cu];
    }
    for (size_t e = 0; e < (size_t)n * n; ++e) Hc[e] = 0.0;
    for (int k = 1; k <= N; ++k) {
        const double* Pk = Phi + (size_t)k * nx * n;
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                double acc = 0.0;
                for (int i = 0; i < nx; ++i) acc += Pk[(size_t)i * n + a] * p.Q[i] * Pk[(size_t)i * n + b];
                Hc[(size_t)a * n + b] += 2.0 * acc;
            }
    }
    // r-term: sum_i R (z_i - z_{i - nu})^2  ->  2 D' R D   (optimal decay: sum_i R z_i^2 -> 2 R)
    for (int i = 0; i < n; ++i) {
        const double r = p.R[i % nu];
        Hc[(size_t)i * n + i] += 2.0 * r;
        if (i >= nu && !p.optimal_decay) {
            Hc[(size_t)(i - nu) * n + (i - nu)] += 2.0 * r;
            Hc[(size_t)i * n + (i - nu)] -= 2.0 * r;
            Hc[(size_t)(i - nu) * n + i] -= 2.0 * r;
        }
    }
    for (int k = 0; k < N; ++k) {
        const double* Pk = Phi + (size_t)k * nx * n;
        for (int dd = 0; dd < 2; ++dd)
            for (int col = 0; col < n; ++col) {
                G[(size_t)(2 * k + dd) * n + col] = Pk[(size_t)dd * n + col];
                double acc = 0.0;
                for (int j = 0; j < nx; ++j) acc += As[dd * nx + j] * Pk[(size_t)j * n + col];
                if (col >= k * nu && col < (k + 1) * nu) acc += Bs[dd * nu + (col - k * nu)];
                G[(size_t)(2 * N + 2 * k + dd) * n + col] = acc;
            }
    }
    std::free(Phi);
    return true;
}

// doubles of one problem's solver state in a continuation workspace (mpc_cont.hpp; the layout of mpclin_kernel's hand-over)
size_t mpclin_state_doubles(int N, int K, int nu) {
    const size_t n = (size_t)N * nu, mc = (size_t)N * K, m = mc + 2 * n;
    return ipm::CONT_SCALARS + 3 * n + 2 * m + 7 * (size_t)K + mc + 2 * (size_t)N;
}

hipError_t mpclin_launch(const sc_mpclin_params& p, const double* model, long long B, int K, const void* X, const void* u_prev,
                         const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out, void* rho_out,
                         hipStream_t stream, const ipm::Cont& ct) {
    const bool od = p.optimal_decay == 1;
    const size_t lds = mpclin_lds_bytes(p.horizon, K, p.nx, p.nu, od);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int mode = mpclin_mode(p.horizon, K, p.nx, p.nu, od);
    const unsigned threads = mode == LIN_BIG ? 256 : 64;                  // big layout: four waves per problem
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(threads), lds, stream, p, model, B, K, X, u_prev, goal, obs, u_out, status,
                           iters, z_out, rho_out, ct);
        return hipGetLastError();
    };
    if (od) {                                                           // the config-5 extension: Quad3D only
        if (!(p.nx == 12 && p.nu == 4)) return hipErrorInvalidValue;
        if (mode == LIN_LEAN) return launch(mpclin_kernel<12, 4, 10, 0, false, true>);
        return mode == LIN_BIG ? launch(mpclin_kernel<12, 4, 0, 0, true, true>) : launch(mpclin_kernel<12, 4, 0, 0, false, true>);
    }
    if (mode == LIN_LEAN) {
        if (p.horizon == 10 && p.nx == 12)                            // Quad3D at the reference's default horizon
            return K == 8 ? launch(mpclin_kernel<12, 4, 10, 8>) : launch(mpclin_kernel<12, 4, 10, 0>);
        if (p.horizon == 10 && p.nx == 2)                             // SingleIntegrator2D
            return K == 8 ? launch(mpclin_kernel<2, 2, 10, 8>) : launch(mpclin_kernel<2, 2, 10, 0>);
        if (p.horizon == 20 && p.nx == 2) return launch(mpclin_kernel<2, 2, 20, 0>);
        if (p.horizon == 16 && p.nx == 12) return launch(mpclin_kernel<12, 4, 16, 0>);
    }
    const bool big = mode == LIN_BIG;
    if (p.nx == 12 && p.nu == 4) return big ? launch(mpclin_kernel<12, 4, 0, 0, true>) : launch(mpclin_kernel<12, 4, 0, 0>);
    if (p.nx == 2 && p.nu == 2) return big ? launch(mpclin_kernel<2, 2, 0, 0, true>) : launch(mpclin_kernel<2, 2, 0, 0>);
    return big ? launch(mpclin_kernel<0, 0, 0, 0, true>) : launch(mpclin_kernel<0, 0, 0, 0>);
}

}  // namespace sc
