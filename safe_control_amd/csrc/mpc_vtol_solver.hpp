// VTOL2D MPC-CBF (robots/vtol2D.py:118-311,475, position_control/mpc_cbf.py:40-43,83-87,222): ONE NLP PER LANE.
//
// The problem is the one oracle/mpc_vtol.py states (N = 30 stages, 6 states, 4 inputs, K circles, rel-degree-2 DT-CBF rows through
// step o step, delta-u penalty, input box, |x_dot| <= v_max, z_dot >= -descent_max, |theta| <= pitch_max) and the algorithm is the
// interior point of oracle/mpc_cbf.py: solve() with the exact Hessian and the slack reset of the line search, restoration phase
// included.  What differs from kernels 3 / 7 / 8 is the linear algebra: n = 120 condensed variables do not fit a wave's LDS share,
// and nothing here needs the condensed matrix.  Every row of the problem is a function of ONE stage's (x_k, u_k), so
//   * the KKT residual  grad f - J' lam  is a backward costate sweep (p_k = l_x + A_k' p_{k+1},  r_k = l_u + B_k' p_{k+1}),
//   * the Newton system (W + J' Sigma J + delta I) dz = rhs  is an LQ problem over the linearised dynamics and is solved by a
//     Riccati recursion on the augmented state (dx_k, du_{k-1}) (the delta-u penalty couples neighbouring inputs): 30 stage blocks of
//     14 x 14 instead of one 120 x 120 factorisation; delta (inertia correction) and zeta (restoration) land on the 4 x 4 input block,
//     and "all 30 input blocks positive definite" is the same test as "condensed matrix positive definite",
//   * J dz, the merit function and its directional derivative are stage-local sums over the forward LQ rollout.
// Euler's position update uses the current velocity, so the three barrier points of a stage are p_k, p_k + dt v_k and
// p_k + 2 dt v_k + dt^2 a(x_k, u_k): one evaluation of the aero model per stage.  Its first and second derivatives with respect to
// (theta, x_dot, z_dot) -- everything the dynamics are nonlinear in -- come from second-order forward mode (D2 below), the same
// arithmetic as oracle/mpc_vtol.py: Dual2.
//
// This header is plain C++ (no HIP intrinsics): a lane runs solve() on its own problem with its work arrays behind `Mem`
// (lane-interleaved in HBM on the device; tools/vtol_host.cpp compiles the same code for the host as a debugging aid).
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef SC_HD
#define SC_HD __host__ __device__
#endif
// hooks for the device build's short reciprocal / sincos (sc_qp2.hpp, sc_math.hpp: an ulp or two from the IEEE results)
#ifndef SC_VTOL_RCP
#define SC_VTOL_RCP(a) (1.0 / (a))
#endif
#ifndef SC_VTOL_SINCOS
#define SC_VTOL_SINCOS(a, s, c) { s = sin(a); c = cos(a); }
#endif

namespace sc {
namespace vtol {

constexpr int NX = 6, NU = 4, NV = 10, NXB = 5;
constexpr int ST_OPTIMAL = 0, ST_INFEASIBLE = 1, ST_INACCURATE = 2;

struct Params {                      // filled from sc_mpcvtol_params by the launcher
    int N, K, max_iter, acceptable_iter, slack_reset, resto_reset, resto_gn;
    double dt, Q[6], R[4], alpha1, alpha2, beta, radius, u_lo[4], u_hi[4], v_max, descent_max, pitch_max;
    double tol, acceptable_tol, mu_init, mu_min, row_noise;
    double rho, kappa, theta_tol, resto_tol, small_alpha;
    int small_iter, max_entries;
    // airframe (vtol2D.py:56-111)
    double mass, inertia, S_wing, rho_air, C_L0, C_Lalpha, M, alpha_0, C_Ldelta_e, C_D0, C_Dalpha, C_Ddelta_e, C_m0, C_malpha, C_mdelta_e,
        chord, k_front, k_rear, k_pusher, ell_f, ell_r;
    double ps1 = 0.0, ps2 = 0.0, rf1 = 1.0, rf2 = 1.0;   // optimal decay (mpc_vtol_wave.hip, OD): penalties and references of the decay variables
    double eMa0sq;                   // exp(2 M alpha_0)
    double inv_m, inv_I, kf_m, kr_m, kp_m, lfkf_I, lrkr_I;
};

// the 21 airframe constants (sc_mpcvtol_params.airframe order) and what the aero model precomputes from them
SC_HD inline void set_airframe(Params& P, const double* a) {
    P.mass = a[0]; P.inertia = a[1]; P.S_wing = a[2]; P.rho_air = a[3]; P.C_L0 = a[4]; P.C_Lalpha = a[5]; P.M = a[6]; P.alpha_0 = a[7];
    P.C_Ldelta_e = a[8]; P.C_D0 = a[9]; P.C_Dalpha = a[10]; P.C_Ddelta_e = a[11]; P.C_m0 = a[12]; P.C_malpha = a[13]; P.C_mdelta_e = a[14];
    P.chord = a[15]; P.k_front = a[16]; P.k_rear = a[17]; P.k_pusher = a[18]; P.ell_f = a[19]; P.ell_r = a[20];
    P.eMa0sq = exp(2.0 * P.M * P.alpha_0);
    P.inv_m = 1.0 / P.mass; P.inv_I = 1.0 / P.inertia;
    P.kf_m = P.k_front / P.mass; P.kr_m = P.k_rear / P.mass; P.kp_m = P.k_pusher / P.mass;
    P.lfkf_I = P.ell_f * P.k_front / P.inertia; P.lrkr_I = P.ell_r * P.k_rear / P.inertia;
}

#ifdef SC_VTOL_WITH_C_PARAMS
inline Params from_c(const sc_mpcvtol_params& c, int K) {
    Params P;
    P.N = c.horizon; P.K = K; P.max_iter = c.max_iter; P.acceptable_iter = c.acceptable_iter; P.slack_reset = c.slack_reset;
    P.dt = c.dt;
    for (int i = 0; i < 6; ++i) P.Q[i] = c.Q[i];
    for (int i = 0; i < 4; ++i) { P.R[i] = c.R[i]; P.u_lo[i] = c.u_lo[i]; P.u_hi[i] = c.u_hi[i]; }
    P.alpha1 = c.alpha1; P.alpha2 = c.alpha2; P.beta = c.beta; P.radius = c.robot_radius;
    P.v_max = c.v_max; P.descent_max = c.descent_speed_max; P.pitch_max = c.pitch_max;
    P.tol = c.tol; P.acceptable_tol = c.acceptable_tol; P.mu_init = c.mu_init; P.mu_min = c.mu_min; P.row_noise = 1e-15;
    P.rho = c.resto.rho; P.kappa = c.resto.kappa; P.theta_tol = c.resto.theta_tol; P.resto_tol = c.resto.tol; P.small_alpha = c.resto.small_alpha;
    P.small_iter = c.resto.small_iter; P.max_entries = c.resto.max_entries; P.resto_reset = c.resto.slack_reset; P.resto_gn = c.resto.gauss_newton;
    set_airframe(P, c.airframe);
    return P;
}
#endif

// ---- second-order forward mode over q = (theta, x_dot, z_dot) ------------------------------------------------------------------
struct D2 {
    double v, d[3], h[6];            // h: (00, 01, 02, 11, 12, 22)
};
SC_HD inline D2 d2c(double c) { D2 r; r.v = c; for (int i = 0; i < 3; ++i) r.d[i] = 0; for (int i = 0; i < 6; ++i) r.h[i] = 0; return r; }
SC_HD inline D2 d2var(double v, int i) { D2 r = d2c(v); r.d[i] = 1.0; return r; }
SC_HD inline D2 chain(const D2& a, double f, double f1, double f2) {
    D2 r; r.v = f;
    for (int i = 0; i < 3; ++i) r.d[i] = f1 * a.d[i];
    int e = 0;
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j, ++e) r.h[e] = f1 * a.h[e] + f2 * a.d[i] * a.d[j];
    return r;
}
SC_HD inline D2 operator+(const D2& a, const D2& b) { D2 r; r.v = a.v + b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] + b.d[i]; for (int i = 0; i < 6; ++i) r.h[i] = a.h[i] + b.h[i]; return r; }
SC_HD inline D2 operator-(const D2& a, const D2& b) { D2 r; r.v = a.v - b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] - b.d[i]; for (int i = 0; i < 6; ++i) r.h[i] = a.h[i] - b.h[i]; return r; }
SC_HD inline D2 operator-(const D2& a) { D2 r; r.v = -a.v; for (int i = 0; i < 3; ++i) r.d[i] = -a.d[i]; for (int i = 0; i < 6; ++i) r.h[i] = -a.h[i]; return r; }
SC_HD inline D2 operator*(const D2& a, const D2& b) {
    D2 r; r.v = a.v * b.v;
    for (int i = 0; i < 3; ++i) r.d[i] = a.v * b.d[i] + b.v * a.d[i];
    int e = 0;
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j, ++e) r.h[e] = a.v * b.h[e] + b.v * a.h[e] + a.d[i] * b.d[j] + a.d[j] * b.d[i];
    return r;
}
SC_HD inline D2 operator+(const D2& a, double c) { D2 r = a; r.v += c; return r; }
SC_HD inline D2 operator+(double c, const D2& a) { D2 r = a; r.v += c; return r; }
SC_HD inline D2 operator-(const D2& a, double c) { D2 r = a; r.v -= c; return r; }
SC_HD inline D2 operator-(double c, const D2& a) { D2 r = -a; r.v += c; return r; }
SC_HD inline D2 operator*(const D2& a, double c) { D2 r; r.v = a.v * c; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] * c; for (int i = 0; i < 6; ++i) r.h[i] = a.h[i] * c; return r; }
SC_HD inline D2 operator*(double c, const D2& a) { return a * c; }
SC_HD inline D2 recip(const D2& a) { const double r = SC_VTOL_RCP(a.v); return chain(a, r, -r * r, 2.0 * r * r * r); }
SC_HD inline D2 operator/(const D2& a, const D2& b) { return a * recip(b); }
SC_HD inline D2 operator/(const D2& a, double c) { return a * (1.0 / c); }
SC_HD inline D2 sin_(const D2& a) { const double s = sin(a.v), c = cos(a.v); return chain(a, s, c, -s); }
SC_HD inline D2 cos_(const D2& a) { const double s = sin(a.v), c = cos(a.v); return chain(a, c, -s, -c); }
SC_HD inline D2 exp_(const D2& a) { const double e = exp(a.v); return chain(a, e, e, e); }
SC_HD inline D2 sqrt_(const D2& a) { const double r = sqrt(a.v); return chain(a, r, 0.5 / r, -0.25 / (r * a.v)); }
SC_HD inline D2 sq_(const D2& a) { return a * a; }
SC_HD inline D2 atan2_(const D2& y, const D2& x) {
    const double r2 = x.v * x.v + y.v * y.v, r4 = r2 * r2;
    const double ty = x.v / r2, tx = -y.v / r2;
    const double tyy = -2.0 * x.v * y.v / r4, txx = 2.0 * x.v * y.v / r4, txy = (y.v * y.v - x.v * x.v) / r4;
    D2 r; r.v = atan2(y.v, x.v);
    for (int i = 0; i < 3; ++i) r.d[i] = ty * y.d[i] + tx * x.d[i];
    int e = 0;
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j, ++e)
        r.h[e] = ty * y.h[e] + tx * x.h[e] + tyy * y.d[i] * y.d[j] + txx * x.d[i] * x.d[j] + txy * (y.d[i] * x.d[j] + x.d[i] * y.d[j]);
    return r;
}
SC_HD inline double sin_(double a) { return sin(a); }
SC_HD inline double cos_(double a) { return cos(a); }
SC_HD inline double exp_(double a) { return exp(a); }
SC_HD inline double sqrt_(double a) { return sqrt(a); }
SC_HD inline double sq_(double a) { return a * a; }
SC_HD inline double atan2_(double y, double x) { return atan2(y, x); }
SC_HD inline double recip_or(double a) { return SC_VTOL_RCP(a); }
SC_HD inline D2 recip_or(const D2& a) { return recip(a); }
SC_HD inline double val(double a) { return a; }
SC_HD inline double val(const D2& a) { return a.v; }

// ---- the airframe: vtol2D.py:333-452 as oracle/mpc_vtol.py: fg -------------------------------------------------------------------
// Same functions, fewer library calls than the literal form (ten transcendental calls per evaluation there, four here -- the rollout of
// an evaluation is thirty of these in a row): with alpha = atan2(-w_b, u_b) the sine and cosine of alpha are -w_b / V and u_b / V, those
// of theta + alpha follow from the addition theorems, exp(M (alpha + alpha_0)) is exp(M alpha_0)^2 / exp(-M (alpha - alpha_0)), and the
// elevator column (the model at delta_e = 1) shares everything but three constants with the drift term (delta_e = 0).
SC_HD inline void sincos_(double a, double& s, double& c) { SC_VTOL_SINCOS(a, s, c) }
SC_HD inline void sincos_(const D2& a, D2& s, D2& c) { double sv, cv; SC_VTOL_SINCOS(a.v, sv, cv) s = chain(a, sv, cv, -sv); c = chain(a, cv, -sv, -cv); }
SC_HD inline double inv_(double a) { return a > 0.0 ? SC_VTOL_RCP(a) : 0.0; }            // V = 0: every aerodynamic force carries the factor V^2
SC_HD inline D2 inv_(const D2& a) { return recip(a); }

// accelerations (x_ddot, z_ddot, theta_ddot) at (theta, x_dot, z_dot) with input u: acc[i] = f_i + sum_j g_ij u_j; gcol[j][i] = g_ij
template <typename T>
SC_HD inline void accel(const Params& P, const T& th, const T& xd, const T& zd, const double* u, T acc[3], T gcol[4][3]) {
    T c, sn;
    sincos_(th, sn, c);
    const T u_b = c * xd + sn * zd, w_b = c * zd - sn * xd;
    const T V2 = u_b * u_b + w_b * w_b;
    const T V = sqrt_(V2), iV = inv_(V);
    const T ca = u_b * iV, sa = -(w_b * iV);
    const T alpha = atan2_(-w_b, u_b);
    // lift blending (vtol2D.py:348-372)
    const T sig_a = exp_(-P.M * (alpha - P.alpha_0));
    const T sig_b = P.eMa0sq * recip_or(sig_a);
    const T sigma = (1.0 + sig_a + sig_b) * recip_or((1.0 + sig_a) * (1.0 + sig_b));
    const T CL_lin = P.C_L0 + P.C_Lalpha * alpha;
    const T CL_non = 2.0 * sa * ca;
    const T CL = (1.0 - sigma) * CL_lin + sigma * CL_non;
    const T CD = P.C_D0 + P.C_Dalpha * sq_(alpha);
    const T CM = P.C_m0 + P.C_malpha * alpha;
    const T qS = (0.5 * P.rho_air * P.S_wing) * V2;
    const T L0 = qS * CL, D0 = qS * CD, M0 = qS * CM * P.chord;
    const T Le = qS * (CL + P.C_Ldelta_e), De = qS * (CD + P.C_Ddelta_e), Me = qS * (CM + P.C_mdelta_e) * P.chord;
    const T ch = c * ca - sn * sa, sh = sn * ca + c * sa;              // cos, sin of theta + alpha
    const T fx = -(ch * D0) - sh * L0, fz = ch * L0 - sh * D0;         // wind -> inertial of (-D, L)
    const T ex = -(ch * De) - sh * Le, ez = ch * Le - sh * De;
    acc[0] = fx * P.inv_m; acc[1] = fz * P.inv_m - 9.81; acc[2] = M0 * P.inv_I;
    gcol[0][0] = -(sn * P.kf_m); gcol[0][1] = c * P.kf_m; gcol[0][2] = 0.0 * c + P.lfkf_I;
    gcol[1][0] = -(sn * P.kr_m); gcol[1][1] = c * P.kr_m; gcol[1][2] = 0.0 * c - P.lrkr_I;
    gcol[2][0] = c * P.kp_m; gcol[2][1] = sn * P.kp_m; gcol[2][2] = 0.0 * c;
    gcol[3][0] = ex * P.inv_m; gcol[3][1] = ez * P.inv_m; gcol[3][2] = Me * P.inv_I;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) acc[i] = acc[i] + gcol[j][i] * u[j];
}

// x+ = x + (f + g u) dt  (mpc_cbf.py:135-141); the pitch wrap of step() (vtol2D.py:299-307) changes no barrier point
SC_HD inline void step_plain(const Params& P, const double* x, const double* u, double* xn) {
    double acc[3], gc[4][3];
    accel<double>(P, x[2], x[3], x[4], u, acc, gc);
    xn[0] = x[0] + P.dt * x[3]; xn[1] = x[1] + P.dt * x[4]; xn[2] = x[2] + P.dt * x[5];
    xn[3] = x[3] + P.dt * acc[0]; xn[4] = x[4] + P.dt * acc[1]; xn[5] = x[5] + P.dt * acc[2];
}

// ---- work arrays of one problem ------------------------------------------------------------------------------------------------------
struct Layout {
    int N, K, n, m, m_el;
    int z, zt, dz, zR, zb, X, g, s, lam, t, sig, dl0, ds, dlam, st, gt, dtt, A, B, p, H, q, KK, kk, dx, total;
    SC_HD Layout(int N_, int K_) : N(N_), K(K_) {
        n = N * NU; m_el = N * K; m = m_el + NXB * N + 2 * n;
        int o = 0;
        auto take = [&](int c) { int r = o; o += c; return r; };
        z = take(n); zt = take(n); dz = take(n); zR = take(n); zb = take(n); X = take((N + 1) * NX);
        g = take(m); s = take(m); lam = take(m); sig = take(m); dl0 = take(m); ds = take(m); dlam = take(m); st = take(m); gt = take(m);
        t = take(m_el); dtt = take(m_el);
        A = take(N * 36); B = take(N * 24); p = take((N + 1) * NX); H = take(N * 55); q = take(N * NV); KK = take(N * 40); kk = take(N * NU);
        dx = take((N + 1) * NX);
        total = o;
    }
};
SC_HD inline int sym(int a, int b) { return a <= b ? a * NV - a * (a - 1) / 2 + (b - a) : b * NV - b * (b - 1) / 2 + (a - b); }   // upper-packed 10 x 10

struct Weights { double w0, w1, w2; };

template <typename Mem, typename Obs>
struct Solver {
    const Params& P;
    const Layout L;
    Mem W;
    Obs ob;                                   // ob(j, c): obstacle j, column c (0: x, 1: z, 2: radius)
    double x0[NX], uprev[NU], xg[2];
    Weights cw_;
    SC_HD Solver(const Params& P_, Mem W_, Obs ob_) : P(P_), L(P_.N, P_.K), W(W_), ob(ob_) {
        const double g1 = P.alpha1 + P.alpha2, g2 = P.alpha1 * P.alpha2;
        cw_.w0 = 1.0 - g1 + g2; cw_.w1 = g1 - 2.0; cw_.w2 = 1.0;
    }
    SC_HD int row_xb(int k, int r) const { return L.m_el + (k - 1) * NXB + r; }      // k = 1..N
    SC_HD int row_hi(int i) const { return L.m_el + NXB * L.N + i; }
    SC_HD int row_lo(int i) const { return L.m_el + NXB * L.N + L.n + i; }

    // barrier points of stage k from x_k and the acceleration there
    SC_HD void points(const double* x, const double* acc, double pt[3][2]) const {
        const double dt = P.dt;
        pt[0][0] = x[0]; pt[0][1] = x[1];
        pt[1][0] = x[0] + dt * x[3]; pt[1][1] = x[1] + dt * x[4];
        pt[2][0] = pt[1][0] + dt * (x[3] + dt * acc[0]); pt[2][1] = pt[1][1] + dt * (x[4] + dt * acc[1]);
    }
    SC_HD void bounds_rows(const double* x, double r[NXB]) const {
        r[0] = P.v_max - x[3]; r[1] = x[3] + P.v_max; r[2] = x[4] + P.descent_max; r[3] = P.pitch_max - x[2]; r[4] = x[2] + P.pitch_max;
    }

    // level 0: f (unscaled cost) and all rows g at the inputs stored at offset zo; rows to offset go.  Stores X when keepX.
    SC_HD double eval0(int zo, int go, bool keepX) {
        double x[NX], xn[NX], u[NU], up[NU], f = 0.0;
        for (int i = 0; i < NX; ++i) x[i] = x0[i];
        for (int j = 0; j < NU; ++j) up[j] = uprev[j];
        if (keepX) for (int i = 0; i < NX; ++i) W(L.X + i) = x[i];
        for (int k = 0; k < L.N; ++k) {
            for (int j = 0; j < NU; ++j) u[j] = W(zo + k * NU + j);
            double acc[3], gc[4][3], pt[3][2];
            accel<double>(P, x[2], x[3], x[4], u, acc, gc);
            points(x, acc, pt);
            for (int j = 0; j < L.K; ++j) {
                const double d = P.radius + ob(j, 2), cx = ob(j, 0), cz = ob(j, 1), off = P.beta * d * d;
                double hv[3];
                for (int p = 0; p < 3; ++p) { const double ex = pt[p][0] - cx, ez = pt[p][1] - cz; hv[p] = ex * ex + ez * ez - off; }
                W(go + k * L.K + j) = cw_.w0 * hv[0] + cw_.w1 * hv[1] + cw_.w2 * hv[2];
            }
            xn[0] = x[0] + P.dt * x[3]; xn[1] = x[1] + P.dt * x[4]; xn[2] = x[2] + P.dt * x[5];
            xn[3] = x[3] + P.dt * acc[0]; xn[4] = x[4] + P.dt * acc[1]; xn[5] = x[5] + P.dt * acc[2];
            for (int j = 0; j < NU; ++j) { const double du = u[j] - up[j]; f += P.R[j] * du * du; up[j] = u[j]; }
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
            { double e0 = x[0] - xg[0], e1 = x[1] - xg[1];
              f += P.Q[0] * e0 * e0 + P.Q[1] * e1 * e1 + P.Q[2] * x[2] * x[2] + P.Q[3] * x[3] * x[3] + P.Q[4] * x[4] * x[4] + P.Q[5] * x[5] * x[5]; }
            double rb[NXB]; bounds_rows(x, rb);
            for (int r = 0; r < NXB; ++r) W(go + row_xb(k + 1, r)) = rb[r];
            for (int j = 0; j < NU; ++j) { W(go + row_hi(k * NU + j)) = P.u_hi[j] - u[j]; W(go + row_lo(k * NU + j)) = u[j] - P.u_lo[j]; }
            if (keepX) for (int i = 0; i < NX; ++i) W(L.X + (k + 1) * NX + i) = x[i];
        }
        return f;
    }

    // Jacobians A_k, B_k of the prediction at the stored X, z (first derivatives only)
    SC_HD void linearise() {
        for (int k = 0; k < L.N; ++k) {
            double u[NU];
            for (int j = 0; j < NU; ++j) u[j] = W(L.z + k * NU + j);
            D2 acc[3], gc[4][3];
            accel<D2>(P, d2var(W(L.X + k * NX + 2), 0), d2var(W(L.X + k * NX + 3), 1), d2var(W(L.X + k * NX + 4), 2), u, acc, gc);
            const int a = L.A + k * 36, b = L.B + k * 24;
            for (int i = 0; i < 36; ++i) W(a + i) = 0.0;
            for (int i = 0; i < NX; ++i) W(a + i * 6 + i) = 1.0;
            W(a + 0 * 6 + 3) = P.dt; W(a + 1 * 6 + 4) = P.dt; W(a + 2 * 6 + 5) = P.dt;
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 3; ++c) W(a + (3 + i) * 6 + (2 + c)) += P.dt * acc[i].d[c];
            for (int i = 0; i < 24; ++i) W(b + i) = 0.0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < NU; ++j) W(b + (3 + i) * 4 + j) = P.dt * gc[j][i].v;
        }
    }

    // gradient of the three barrier points of stage k in v = (x_k, u_k): Gp[p][c][10]; acc from the stored A, B rows 3, 4
    SC_HD void point_jac(int k, double G2[2][NV]) const {
        // G0 = [e0; e1], G1 = [e0 + dt e3; e1 + dt e4] are constants; G2 = G1 + dt * rows (3, 4) of [A B]
        const int a = L.A + k * 36, b = L.B + k * 24;
        for (int c = 0; c < 2; ++c) {
            for (int i = 0; i < NX; ++i) G2[c][i] = P.dt * W(a + (3 + c) * 6 + i);
            for (int j = 0; j < NU; ++j) G2[c][6 + j] = P.dt * W(b + (3 + c) * 4 + j);
            G2[c][c] += 1.0; G2[c][3 + c] += P.dt;
        }
    }
    // gradient (10) of CBF row (k, j) given the points and G2
    SC_HD void cbf_row_grad(const double pt[3][2], const double G2[2][NV], int j, double r[NV]) const {
        const double cx = ob(j, 0), cz = ob(j, 1);
        const double e0x = pt[0][0] - cx, e0z = pt[0][1] - cz, e1x = pt[1][0] - cx, e1z = pt[1][1] - cz, e2x = pt[2][0] - cx, e2z = pt[2][1] - cz;
        for (int i = 0; i < NV; ++i) r[i] = 2.0 * cw_.w2 * (e2x * G2[0][i] + e2z * G2[1][i]);
        r[0] += 2.0 * (cw_.w0 * e0x + cw_.w1 * e1x); r[1] += 2.0 * (cw_.w0 * e0z + cw_.w1 * e1z);
        r[3] += 2.0 * cw_.w1 * e1x * P.dt; r[4] += 2.0 * cw_.w1 * e1z * P.dt;
    }
    SC_HD void stage_points(int k, double pt[3][2]) const {
        double x[NX], acc[2];
        for (int i = 0; i < NX; ++i) x[i] = W(L.X + k * NX + i);
        // acceleration from the stored next state: x_{k+1}[3:5] = x_k[3:5] + dt acc
        acc[0] = (W(L.X + (k + 1) * NX + 3) - x[3]) / P.dt; acc[1] = (W(L.X + (k + 1) * NX + 4) - x[4]) / P.dt;
        points(x, acc, pt);
    }

    // backward costate sweep with the multipliers at offset lo (NULL-like: lo < 0 means no rows): stores p_k, returns |r_d|_inf.
    //   cost weight cw on f; in the restoration the objective is zeta/2 |z - zR|^2 instead
    SC_HD double adjoint(int lo, double cw, bool resto, double zeta) {
        double p[NX], rd = 0.0;
        const int N = L.N;
        // terminal
        {
            const int xo = L.X + N * NX;
            p[0] = 2.0 * cw * P.Q[0] * (W(xo) - xg[0]); p[1] = 2.0 * cw * P.Q[1] * (W(xo + 1) - xg[1]);
            for (int i = 2; i < NX; ++i) p[i] = 2.0 * cw * P.Q[i] * W(xo + i);
            if (lo >= 0) {
                p[3] += W(lo + row_xb(N, 0)) - W(lo + row_xb(N, 1)); p[4] -= W(lo + row_xb(N, 2));
                p[2] += W(lo + row_xb(N, 3)) - W(lo + row_xb(N, 4));
            }
            for (int i = 0; i < NX; ++i) W(L.p + N * NX + i) = p[i];
        }
        for (int k = N - 1; k >= 0; --k) {
            double lv[NV];
            for (int i = 0; i < NV; ++i) lv[i] = 0.0;
            if (lo >= 0) {
                double pt[3][2], G2[2][NV];
                stage_points(k, pt); point_jac(k, G2);
                for (int j = 0; j < L.K; ++j) {
                    double r[NV]; cbf_row_grad(pt, G2, j, r);
                    const double l = W(lo + k * L.K + j);
                    for (int i = 0; i < NV; ++i) lv[i] -= l * r[i];
                }
                for (int j = 0; j < NU; ++j) lv[6 + j] += W(lo + row_hi(k * NU + j)) - W(lo + row_lo(k * NU + j));
                if (k >= 1) {
                    lv[3] += W(lo + row_xb(k, 0)) - W(lo + row_xb(k, 1)); lv[4] -= W(lo + row_xb(k, 2));
                    lv[2] += W(lo + row_xb(k, 3)) - W(lo + row_xb(k, 4));
                }
            }
            if (k >= 1) {
                const int xo = L.X + k * NX;
                lv[0] += 2.0 * cw * P.Q[0] * (W(xo) - xg[0]); lv[1] += 2.0 * cw * P.Q[1] * (W(xo + 1) - xg[1]);
                for (int i = 2; i < NX; ++i) lv[i] += 2.0 * cw * P.Q[i] * W(xo + i);
            }
            for (int j = 0; j < NU; ++j) {
                const double uk = W(L.z + k * NU + j), um = k ? W(L.z + (k - 1) * NU + j) : uprev[j];
                double gj = 2.0 * cw * P.R[j] * (uk - um);
                if (k + 1 < N) gj -= 2.0 * cw * P.R[j] * (W(L.z + (k + 1) * NU + j) - uk);
                if (resto) gj += zeta * (uk - W(L.zR + k * NU + j));
                lv[6 + j] += gj;
            }
            const int a = L.A + k * 36, b = L.B + k * 24;
            for (int j = 0; j < NU; ++j) {
                double r = lv[6 + j];
                for (int i = 0; i < NX; ++i) r += W(b + i * 4 + j) * p[i];
                rd = fmax(rd, fabs(r));
            }
            double pn[NX];
            for (int c = 0; c < NX; ++c) {
                double v = lv[c];
                for (int i = 0; i < NX; ++i) v += W(a + i * 6 + c) * p[i];
                pn[c] = v;
            }
            for (int i = 0; i < NX; ++i) { p[i] = pn[i]; W(L.p + k * NX + i) = pn[i]; }
        }
        return rd;
    }

    // stage blocks of the Newton system: H_k (10 x 10, packed) and q_k (negative gradient with lam + dl0) in v = (x_k, u_k);
    // x-terms of stage k >= 1 (cost, state bounds) are attached to stage k; the terminal stage's go to HT, qT.
    SC_HD void stage_blocks(double cw, bool resto, double zeta, double HT[21], double qT[NX]) {
        const int N = L.N;
        const double so = (resto && P.resto_gn) ? 0.0 : 1.0;
        for (int k = 0; k < N; ++k) {
            double H[55], q[NV], u[NU];
            for (int i = 0; i < 55; ++i) H[i] = 0.0;
            for (int i = 0; i < NV; ++i) q[i] = 0.0;
            for (int j = 0; j < NU; ++j) u[j] = W(L.z + k * NU + j);
            const int xo = L.X + k * NX;
            D2 acc[3], gc[4][3];
            accel<D2>(P, d2var(W(xo + 2), 0), d2var(W(xo + 3), 1), d2var(W(xo + 4), 2), u, acc, gc);
            double pt[3][2], G2[2][NV], x[NX];
            for (int i = 0; i < NX; ++i) x[i] = W(xo + i);
            { double a2[2] = {acc[0].v, acc[1].v}; points(x, a2, pt); }
            point_jac(k, G2);
            // rows of the stage
            double slam = 0.0, nu2[2] = {0.0, 0.0};
            for (int j = 0; j < L.K; ++j) {
                double r[NV]; cbf_row_grad(pt, G2, j, r);
                const int row = k * L.K + j;
                const double sg = W(L.sig + row), lq = W(L.lam + row) + W(L.dl0 + row), l = W(L.lam + row);
                for (int a = 0; a < NV; ++a) {
                    q[a] += lq * r[a];
                    for (int b = a; b < NV; ++b) H[sym(a, b)] += sg * r[a] * r[b];
                }
                slam += l;
                nu2[0] -= cw_.w2 * l * 2.0 * (pt[2][0] - ob(j, 0)); nu2[1] -= cw_.w2 * l * 2.0 * (pt[2][1] - ob(j, 1));
            }
            // curvature of h in the points: sum_p om_p G_p' G_p, om_p = -2 w_p sum_j lam_kj
            {
                // (so: 0 in a Gauss-Newton restoration, sc_resto_params.gauss_newton -- the second-order terms of rows and dynamics are dropped)
                const double o0 = -2.0 * cw_.w0 * slam * so, o1 = -2.0 * cw_.w1 * slam * so, o2 = -2.0 * cw_.w2 * slam * so;
                H[sym(0, 0)] += o0 + o1; H[sym(1, 1)] += o0 + o1;
                H[sym(0, 3)] += o1 * P.dt; H[sym(1, 4)] += o1 * P.dt; H[sym(3, 3)] += o1 * P.dt * P.dt; H[sym(4, 4)] += o1 * P.dt * P.dt;
                for (int a = 0; a < NV; ++a)
                    for (int b = a; b < NV; ++b) H[sym(a, b)] += o2 * (G2[0][a] * G2[0][b] + G2[1][a] * G2[1][b]);
            }
            // second derivatives of the dynamics, weighted by c = p_{k+1} + (d points_2 / d y1)' nu_2 (rows 3..5 matter)
            {
                const double c3 = W(L.p + (k + 1) * NX + 3) + P.dt * nu2[0], c4 = W(L.p + (k + 1) * NX + 4) + P.dt * nu2[1],
                             c5 = W(L.p + (k + 1) * NX + 5);
                const double cc[3] = {c3 * P.dt * so, c4 * P.dt * so, c5 * P.dt * so};
                int e = 0;
                for (int a = 0; a < 3; ++a)
                    for (int b = a; b < 3; ++b, ++e)
                        H[sym(2 + a, 2 + b)] += cc[0] * acc[0].h[e] + cc[1] * acc[1].h[e] + cc[2] * acc[2].h[e];
                for (int a = 0; a < 3; ++a)
                    for (int j = 0; j < NU; ++j)
                        H[sym(2 + a, 6 + j)] += cc[0] * gc[j][0].d[a] + cc[1] * gc[j][1].d[a] + cc[2] * gc[j][2].d[a];
            }
            // input box
            for (int j = 0; j < NU; ++j) {
                const int rh = row_hi(k * NU + j), rl = row_lo(k * NU + j);
                H[sym(6 + j, 6 + j)] += W(L.sig + rh) + W(L.sig + rl);
                q[6 + j] += -(W(L.lam + rh) + W(L.dl0 + rh)) + (W(L.lam + rl) + W(L.dl0 + rl));
                if (resto) q[6 + j] -= zeta * (u[j] - W(L.zR + k * NU + j));
            }
            if (k >= 1) x_terms(k, cw, H, q, true);
            for (int i = 0; i < 55; ++i) W(L.H + k * 55 + i) = H[i];
            for (int i = 0; i < NV; ++i) W(L.q + k * NV + i) = q[i];
        }
        double H[55], q[NV];
        for (int i = 0; i < 55; ++i) H[i] = 0.0;
        for (int i = 0; i < NV; ++i) q[i] = 0.0;
        x_terms(N, cw, H, q, true);
        int e = 0;
        for (int a = 0; a < NX; ++a) for (int b = a; b < NX; ++b, ++e) HT[e] = H[sym(a, b)];
        for (int i = 0; i < NX; ++i) qT[i] = q[i];
    }
    SC_HD void x_terms(int k, double cw, double* H, double* q, bool) {
        const int xo = L.X + k * NX;
        for (int i = 0; i < NX; ++i) H[sym(i, i)] += 2.0 * cw * P.Q[i];
        q[0] -= 2.0 * cw * P.Q[0] * (W(xo) - xg[0]); q[1] -= 2.0 * cw * P.Q[1] * (W(xo + 1) - xg[1]);
        for (int i = 2; i < NX; ++i) q[i] -= 2.0 * cw * P.Q[i] * W(xo + i);
        const int idx[NXB] = {3, 3, 4, 2, 2};
        const double sgn[NXB] = {-1.0, 1.0, 1.0, -1.0, 1.0};                // d row / d x[idx]
        for (int r = 0; r < NXB; ++r) {
            const int row = row_xb(k, r);
            H[sym(idx[r], idx[r])] += W(L.sig + row);
            q[idx[r]] += sgn[r] * (W(L.lam + row) + W(L.dl0 + row));
        }
    }

    // Riccati recursion over xi_k = (dx_k, du_{k-1}); returns false when an input block is not positive definite
    SC_HD bool riccati(double cw, double shift, const double HT[21], const double qT[NX]) {
        const int N = L.N;
        double Pm[NV][NV], pv[NV];
        for (int a = 0; a < NV; ++a) { pv[a] = 0.0; for (int b = 0; b < NV; ++b) Pm[a][b] = 0.0; }
        { int e = 0; for (int a = 0; a < NX; ++a) for (int b = a; b < NX; ++b, ++e) { Pm[a][b] = HT[e]; Pm[b][a] = HT[e]; } }
        for (int i = 0; i < NX; ++i) pv[i] = qT[i];
        for (int k = N - 1; k >= 0; --k) {
            const int a_ = L.A + k * 36, b_ = L.B + k * 24;
            double D[NU], rD[NU];
            for (int j = 0; j < NU; ++j) {
                D[j] = 2.0 * cw * P.R[j];
                const double uk = W(L.z + k * NU + j), um = k ? W(L.z + (k - 1) * NU + j) : uprev[j];
                rD[j] = -2.0 * cw * P.R[j] * (uk - um);
            }
            // PA = P * Abar (10 x 10; Abar = [A 0; 0 0]): columns 0..5 only;  PB = P * Bbar (10 x 4), Bbar = [B; I]
            double PA[NV][NX], PB[NV][NU];
            for (int r = 0; r < NV; ++r) {
                for (int c = 0; c < NX; ++c) { double v = 0.0; for (int i = 0; i < NX; ++i) v += Pm[r][i] * W(a_ + i * 6 + c); PA[r][c] = v; }
                for (int j = 0; j < NU; ++j) { double v = Pm[r][6 + j]; for (int i = 0; i < NX; ++i) v += Pm[r][i] * W(b_ + i * 4 + j); PB[r][j] = v; }
            }
            double Quu[NU][NU], Qux[NU][NV], qu[NU];
            for (int i = 0; i < NU; ++i) {
                for (int j = 0; j < NU; ++j) {
                    double v = PB[6 + i][j] + W(L.H + k * 55 + sym(6 + i, 6 + j));
                    for (int r = 0; r < NX; ++r) v += W(b_ + r * 4 + i) * PB[r][j];
                    Quu[i][j] = v;
                }
                Quu[i][i] += D[i] + shift;
                for (int c = 0; c < NX; ++c) {
                    double v = PA[6 + i][c] + W(L.H + k * 55 + sym(c, 6 + i));
                    for (int r = 0; r < NX; ++r) v += W(b_ + r * 4 + i) * PA[r][c];
                    Qux[i][c] = v;
                }
                for (int c = 0; c < NU; ++c) Qux[i][6 + c] = (c == i) ? -D[i] : 0.0;
                double v = W(L.q + k * NV + 6 + i) + rD[i] + pv[6 + i];
                for (int r = 0; r < NX; ++r) v += W(b_ + r * 4 + i) * pv[r];
                qu[i] = v;
            }
            // Cholesky of Quu
            double Lc[NU][NU];
            for (int i = 0; i < NU; ++i)
                for (int j = 0; j <= i; ++j) {
                    double v = Quu[i][j];
                    for (int c = 0; c < j; ++c) v -= Lc[i][c] * Lc[j][c];
                    if (i == j) { if (!(v > 0.0)) return false; Lc[i][i] = sqrt(v); }
                    else Lc[i][j] = v / Lc[j][j];
                }
            // KK = Quu^-1 Qux, kk = Quu^-1 qu
            double KKm[NU][NV], kkv[NU];
            for (int c = 0; c <= NV; ++c) {
                double y[NU];
                for (int i = 0; i < NU; ++i) {
                    double v = (c < NV) ? Qux[i][c] : qu[i];
                    for (int j = 0; j < i; ++j) v -= Lc[i][j] * y[j];
                    y[i] = v / Lc[i][i];
                }
                for (int i = NU - 1; i >= 0; --i) {
                    double v = y[i];
                    for (int j = i + 1; j < NU; ++j) v -= Lc[j][i] * y[j];
                    y[i] = v / Lc[i][i];
                }
                for (int i = 0; i < NU; ++i) { if (c < NV) KKm[i][c] = y[i]; else kkv[i] = y[i]; }
            }
            for (int i = 0; i < NU; ++i) {
                for (int c = 0; c < NV; ++c) W(L.KK + k * 40 + i * NV + c) = KKm[i][c];
                W(L.kk + k * NU + i) = kkv[i];
            }
            // P_k = Qxx - Qux' KK,  p_k = qx - Qux' kk;   Qxx = Hxx_aug + Abar' P Abar,  qx = q_aug + Abar' p
            double Pn[NV][NV], pn[NV];
            for (int r = 0; r < NV; ++r) {
                for (int c = r; c < NV; ++c) {
                    double v = 0.0;
                    if (r < NX && c < NX) { v = W(L.H + k * 55 + sym(r, c)); for (int i = 0; i < NX; ++i) v += W(a_ + i * 6 + r) * PA[i][c]; }
                    else if (r >= NX && c == r) v = D[r - NX];
                    for (int i = 0; i < NU; ++i) v -= Qux[i][r] * KKm[i][c];
                    Pn[r][c] = v; Pn[c][r] = v;
                }
                double v = 0.0;
                if (r < NX) { v = W(L.q + k * NV + r); for (int i = 0; i < NX; ++i) v += W(a_ + i * 6 + r) * pv[i]; }
                else v = -rD[r - NX];
                for (int i = 0; i < NU; ++i) v -= Qux[i][r] * kkv[i];
                pn[r] = v;
            }
            for (int r = 0; r < NV; ++r) { pv[r] = pn[r]; for (int c = 0; c < NV; ++c) Pm[r][c] = Pn[r][c]; }
        }
        return true;
    }

    // forward LQ rollout: dz and dx
    SC_HD void lq_forward() {
        double xi[NV];
        for (int i = 0; i < NV; ++i) xi[i] = 0.0;
        for (int i = 0; i < NX; ++i) W(L.dx + i) = 0.0;
        for (int k = 0; k < L.N; ++k) {
            double du[NU];
            for (int i = 0; i < NU; ++i) {
                double v = W(L.kk + k * NU + i);
                for (int c = 0; c < NV; ++c) v -= W(L.KK + k * 40 + i * NV + c) * xi[c];
                du[i] = v; W(L.dz + k * NU + i) = v;
            }
            double xn[NX];
            const int a_ = L.A + k * 36, b_ = L.B + k * 24;
            for (int r = 0; r < NX; ++r) {
                double v = 0.0;
                for (int c = 0; c < NX; ++c) v += W(a_ + r * 6 + c) * xi[c];
                for (int j = 0; j < NU; ++j) v += W(b_ + r * 4 + j) * du[j];
                xn[r] = v;
            }
            for (int i = 0; i < NX; ++i) { xi[i] = xn[i]; W(L.dx + (k + 1) * NX + i) = xn[i]; }
            for (int j = 0; j < NU; ++j) xi[6 + j] = du[j];
        }
    }
    // J dz of every row into ds (as jd)
    SC_HD void row_steps() {
        for (int k = 0; k < L.N; ++k) {
            double pt[3][2], G2[2][NV], v[NV];
            stage_points(k, pt); point_jac(k, G2);
            for (int i = 0; i < NX; ++i) v[i] = W(L.dx + k * NX + i);
            for (int j = 0; j < NU; ++j) v[6 + j] = W(L.dz + k * NU + j);
            for (int j = 0; j < L.K; ++j) {
                double r[NV]; cbf_row_grad(pt, G2, j, r);
                double s = 0.0;
                for (int i = 0; i < NV; ++i) s += r[i] * v[i];
                W(L.ds + k * L.K + j) = s;
            }
            for (int j = 0; j < NU; ++j) { W(L.ds + row_hi(k * NU + j)) = -v[6 + j]; W(L.ds + row_lo(k * NU + j)) = v[6 + j]; }
            const int xo = L.dx + (k + 1) * NX;
            W(L.ds + row_xb(k + 1, 0)) = -W(xo + 3); W(L.ds + row_xb(k + 1, 1)) = W(xo + 3); W(L.ds + row_xb(k + 1, 2)) = W(xo + 4);
            W(L.ds + row_xb(k + 1, 3)) = -W(xo + 2); W(L.ds + row_xb(k + 1, 4)) = W(xo + 2);
        }
    }
    SC_HD double grad_dot_dz(double cw) {                      // grad f(z)' dz of the (scaled) cost
        double v = 0.0;
        for (int k = 1; k <= L.N; ++k) {
            const int xo = L.X + k * NX, d = L.dx + k * NX;
            v += 2.0 * cw * (P.Q[0] * (W(xo) - xg[0]) * W(d) + P.Q[1] * (W(xo + 1) - xg[1]) * W(d + 1));
            for (int i = 2; i < NX; ++i) v += 2.0 * cw * P.Q[i] * W(xo + i) * W(d + i);
        }
        for (int k = 0; k < L.N; ++k)
            for (int j = 0; j < NU; ++j) {
                const double uk = W(L.z + k * NU + j), um = k ? W(L.z + (k - 1) * NU + j) : uprev[j];
                const double dk = W(L.dz + k * NU + j), dm = k ? W(L.dz + (k - 1) * NU + j) : 0.0;
                v += 2.0 * cw * P.R[j] * (uk - um) * (dk - dm);
            }
        return v;
    }
    SC_HD double violation(int go) const {
        double v = 0.0;
        for (int i = 0; i < L.m_el; ++i) v += fmax(0.0, -W(go + i));
        return v;
    }

    // ---- oracle/mpc_cbf.py: solve() ---------------------------------------------------------------------------------------------
    SC_HD void solve(int& status_out, int& iters_out) {
        const int n = L.n, m = L.m, m_el = L.m_el;
        for (int k = 0; k < L.N; ++k)
            for (int j = 0; j < NU; ++j) {
                const double lo = P.u_lo[j] + 0.005 * (P.u_hi[j] - P.u_lo[j]), hi = P.u_hi[j] - 0.005 * (P.u_hi[j] - P.u_lo[j]);
                W(L.z + k * NU + j) = fmin(fmax(uprev[j], lo), hi);
            }
        eval0(L.z, L.g, true);
        linearise();
        const double g0 = adjoint(-1, 1.0, false, 0.0);
        const double sf0 = fmin(1.0, 100.0 / fmax(1e-12, g0));
        double mu = P.mu_init;
        for (int i = 0; i < m; ++i) { const double s = fmax(W(L.g + i), 1e-2); W(L.s + i) = s; W(L.lam + i) = mu / s; }
        for (int i = 0; i < m_el; ++i) W(L.t + i) = 0.0;
        int status = ST_INACCURATE, it = 0;
        const double tau = 0.995;
        double nu = 10.0, delta_last = 0.0, e_best = INFINITY;
        int n_acc = 0;
        for (int i = 0; i < n; ++i) W(L.zb + i) = W(L.z + i);
        bool resto = false;
        const double rho = P.rho, theta_tol = P.theta_tol;
        int n_resto = 0, n_small = 0;
        double theta_R = 0.0, mu_reg = mu;
        const bool sreset = P.slack_reset != 0;
        for (it = 1; it <= P.max_iter; ++it) {
            double cw = resto ? 0.0 : sf0;
            double fraw = eval0(L.z, L.g, true);
            linearise();
            if (resto && violation(L.g) <= fmax(n_resto == 1 ? P.kappa * theta_R : 0.0, theta_tol)) {
                resto = false; mu = mu_reg; cw = sf0;
                for (int i = 0; i < m; ++i) { const double s = fmax(W(L.g + i), 1e-2); W(L.s + i) = s; W(L.lam + i) = mu / s; }
                nu = 10.0; n_acc = 0; e_best = INFINITY;
                for (int i = 0; i < n; ++i) W(L.zb + i) = W(L.z + i);
            }
            double zeta = resto ? sqrt(mu) : 0.0;
            const double rdn = adjoint(L.lam, cw, resto, zeta);
            // row residuals
            double rpn = 0.0, cs = 0.0, ct = 0.0, lmax = 0.0;
            for (int i = 0; i < m; ++i) {
                const bool el = resto && i < m_el;
                const double g = W(L.g + i), s = W(L.s + i), l = W(L.lam + i), t = el ? W(L.t + i) : 0.0;
                rpn = fmax(rpn, fabs(g + t - s)); cs = fmax(cs, fabs(s * l)); lmax = fmax(lmax, l);
                if (el) ct = fmax(ct, fabs(t * (rho - l)));
            }
            const double e_opt = fmax(fmax(rdn, rpn), fmax(cs, ct));
            if (!resto && e_opt < e_best) { e_best = e_opt; for (int i = 0; i < n; ++i) W(L.zb + i) = W(L.z + i); }
            if (resto) {
                const double theta = violation(L.g);
                if (e_opt <= P.resto_tol && theta > fmax(theta_tol, 10.0 * e_opt / rho)) { status = ST_INFEASIBLE; break; }
                if (e_opt <= P.tol) break;
            } else if (e_opt <= P.tol) { status = ST_OPTIMAL; break; }
            n_acc = e_opt <= P.acceptable_tol ? n_acc + 1 : 0;
            if (n_acc >= P.acceptable_iter) {
                if (resto && violation(L.g) > theta_tol) status = ST_INFEASIBLE;
                break;
            }
            bool want_resto = !resto && lmax > 1e10;
            double alpha = 0.0, ad = 0.0;
            if (!want_resto) {
                // barrier update
                for (;;) {
                    double cm = 0.0;
                    for (int i = 0; i < m; ++i) {
                        const double s = W(L.s + i), l = W(L.lam + i);
                        cm = fmax(cm, fabs(s * l - mu));
                        if (resto && i < m_el) cm = fmax(cm, fabs(W(L.t + i) * (rho - l) - mu));
                    }
                    const double e_mu = fmax(fmax(rdn, rpn), cm);
                    if (!(e_mu <= 10.0 * mu && mu > P.mu_min)) break;
                    mu = fmax(P.mu_min, fmin(0.2 * mu, mu * sqrt(mu)));
                }
                if (resto) zeta = sqrt(mu);
                // rows: Sigma and the multiplier step at dz = 0
                for (int i = 0; i < m; ++i) {
                    const double g = W(L.g + i), s = W(L.s + i), l = W(L.lam + i);
                    double sg = l / s, d0;
                    if (resto && i < m_el) {
                        const double t = W(L.t + i), nut = rho - l, sgt = nut / t, se = sg * sgt / (sg + sgt), rp = g + t - s;
                        d0 = -se * (rp + mu / nut - t) - (se / sg) * (l - mu / s);
                        sg = se;
                    } else d0 = -sg * (g - s) - l + mu / s;
                    W(L.sig + i) = sg; W(L.dl0 + i) = d0;
                }
                double HT[21], qT[NX];
                stage_blocks(cw, resto, zeta, HT, qT);
                double delta = 0.0;
                bool ok = false;
                for (int tr = 0; tr < 40; ++tr) {
                    if (riccati(cw, delta + zeta, HT, qT)) { ok = true; break; }
                    delta = delta == 0.0 ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
                }
                if (!ok) break;
                if (delta > 0.0) delta_last = delta;
                lq_forward();
                row_steps();
                // steps of the multipliers, slacks (and t); fraction to the boundary
                double ap = 1.0; ad = 1.0;
                double srp = 0.0, sum_dss = 0.0, sum_logs = 0.0, sum_t = 0.0, sum_dt = 0.0, sum_dtt = 0.0, sum_logt = 0.0, sum_absg = 0.0;
                for (int i = 0; i < m; ++i) {
                    const bool el = resto && i < m_el;
                    const double g = W(L.g + i), s = W(L.s + i), l = W(L.lam + i), jd = W(L.ds + i), t = el ? W(L.t + i) : 0.0;
                    const double rp = g + t - s;
                    const double dl = -W(L.sig + i) * jd + W(L.dl0 + i);
                    double dsi = jd + rp;
                    if (el) {
                        const double nut = rho - l, sgt = nut / t, dti = (mu / nut - t) + dl / sgt;
                        W(L.dtt + i) = dti; dsi += dti;
                        if (dti < 0.0) ap = fmin(ap, -tau * t / dti);
                        if (dl > 0.0) ad = fmin(ad, tau * nut / dl);
                        sum_t += t; sum_dt += dti; sum_dtt += dti / t; sum_logt += log(t);
                    }
                    W(L.ds + i) = dsi; W(L.dlam + i) = dl;
                    if (dsi < 0.0) ap = fmin(ap, -tau * s / dsi);
                    if (dl < 0.0) ad = fmin(ad, -tau * l / dl);
                    srp += fabs(rp); sum_dss += dsi / s; sum_logs += log(s); sum_absg += fabs(g);
                }
                nu = fmax(nu, 1.1 * lmax);
                double f, gdz;
                if (resto) {
                    double d2 = 0.0, gd = 0.0;
                    for (int i = 0; i < n; ++i) { const double d = W(L.z + i) - W(L.zR + i); d2 += d * d; gd += d * W(L.dz + i); }
                    f = 0.5 * zeta * d2; gdz = zeta * gd;
                } else { f = sf0 * fraw; gdz = grad_dot_dz(cw); }
                double bar0, dbar;
                if (resto) { bar0 = f + rho * sum_t - mu * (sum_logs + sum_logt); dbar = gdz + rho * sum_dt - mu * (sum_dss + sum_dtt); }
                else { bar0 = f - mu * sum_logs; dbar = gdz - mu * sum_dss; }
                if (dbar - nu * srp >= 0.0 && srp > 0.0) nu = dbar / (0.9 * srp);
                const double phi0 = bar0 + nu * srp, dphi = dbar - nu * srp;
                const double noise_rows = P.row_noise * nu * sum_absg;
                alpha = ap;
                bool accepted = false;
                for (int h = 0; h < 12; ++h) {
                    for (int i = 0; i < n; ++i) W(L.zt + i) = W(L.z + i) + alpha * W(L.dz + i);
                    const double ft = eval0(L.zt, L.gt, false);
                    double slog = 0.0, sabs = 0.0, st_t = 0.0, slogt = 0.0;
                    for (int i = 0; i < m; ++i) {
                        double st = W(L.s + i) + alpha * W(L.ds + i);
                        const double gt = W(L.gt + i);
                        if (resto) {
                            double tt = 0.0;
                            if (i < m_el) { tt = W(L.t + i) + alpha * W(L.dtt + i); st_t += tt; slogt += log(tt); }
                            if (P.resto_reset && gt + tt >= mu / nu) st = gt + tt;       // sc_resto_params.slack_reset
                            sabs += fabs(gt + tt - st);
                        } else {
                            if (sreset) st = (P.slack_reset == 1) ? fmax(st, gt) : (gt >= mu / nu ? gt : st);
                            sabs += fabs(gt - st);
                        }
                        W(L.st + i) = st; slog += log(st);
                    }
                    double phit;
                    if (resto) {
                        double d2 = 0.0;
                        for (int i = 0; i < n; ++i) { const double d = W(L.zt + i) - W(L.zR + i); d2 += d * d; }
                        phit = 0.5 * zeta * d2 + rho * st_t - mu * (slog + slogt) + nu * sabs;
                    } else phit = sf0 * ft - mu * slog + nu * sabs;
                    if (phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * fabs(phi0) + noise_rows) { accepted = true; break; }
                    alpha *= 0.5;
                }
                if (!accepted) {
                    if (resto) break;
                    want_resto = true;
                } else if (!resto) {
                    n_small = (alpha < P.small_alpha && violation(L.g) > theta_tol) ? n_small + 1 : 0;
                    if (n_small >= P.small_iter && n_resto < P.max_entries && e_best > P.acceptable_tol) want_resto = true;
                }
            }
            if (want_resto) {
                theta_R = violation(L.g);
                if (e_best <= P.acceptable_tol || theta_R <= theta_tol || n_resto >= P.max_entries) break;
                resto = true; n_resto += 1; n_small = 0;
                for (int i = 0; i < n; ++i) W(L.zR + i) = W(L.z + i);
                mu_reg = mu;
                double vmax = 0.0;
                for (int i = 0; i < m_el; ++i) vmax = fmax(vmax, -W(L.g + i));
                mu = fmax(mu, vmax);
                for (int i = 0; i < m; ++i) {
                    const double g = W(L.g + i);
                    double s;
                    if (i < m_el) { s = ((2.0 * mu + rho * g) + sqrt(rho * rho * g * g + 4.0 * mu * mu)) / (2.0 * rho); W(L.t + i) = s - g; }
                    else s = fmax(g, 1e-2);
                    W(L.s + i) = s; W(L.lam + i) = mu / s;
                }
                nu = 10.0; n_acc = 0;
                continue;
            }
            // take the step
            for (int i = 0; i < n; ++i) W(L.z + i) += alpha * W(L.dz + i);
            for (int i = 0; i < m; ++i) {
                const double s = ((sreset && !resto) || (P.resto_reset && resto)) ? W(L.st + i) : W(L.s + i) + alpha * W(L.ds + i);
                double l = W(L.lam + i) + ad * W(L.dlam + i);
                l = fmin(fmax(l, mu / (1e10 * s)), 1e10 * mu / s);
                if (resto && i < m_el) {
                    const double t = W(L.t + i) + alpha * W(L.dtt + i);
                    W(L.t + i) = t;
                    l = fmin(fmax(l, rho - 1e10 * mu / t), rho - mu / (1e10 * t));
                    l = fmin(fmax(l, 1e-300), rho * (1.0 - 1e-15));
                }
                W(L.s + i) = s; W(L.lam + i) = l;
            }
        }
        if (it > P.max_iter) it = P.max_iter;
        if (status != ST_OPTIMAL && status != ST_INFEASIBLE && e_best <= P.acceptable_tol && !resto) {
            for (int i = 0; i < n; ++i) W(L.z + i) = W(L.zb + i);
            status = ST_OPTIMAL;
        }
        status_out = status; iters_out = it;
    }
};

}  // namespace vtol
}  // namespace sc
