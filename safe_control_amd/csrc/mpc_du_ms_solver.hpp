// MPC-CBF AS DO-MPC POSES IT -- multiple shooting under IPOPT's filter line-search interior point: ONE NLP PER WAVEFRONT, FOUR LANES PER
// STAGE.  Written out below for DynamicUnicycle2D; the other robots it is instantiated for (Unicycle2D, DoubleIntegrator2D,
// KinematicBicycle2D) are described at `enum { M_DU, .. }`.  Kernel 13 in DESIGN.md; the algorithm is oracle/ms_ipopt.py (StageNLP with du_model()) in KERNEL_PROFILE
// (linear_solver = riccati, max_soc = 0, resto_elastic = "ineq", the stall rule), iterate for iterate.  The reference
// (position_control/mpc_cbf.py:162-174,366-369) hands IPOPT
//     variables    x_0 .. x_N (4 each), u_0 .. u_{N-1} (2 each)
//     objective    sum_k (x_k - goal)' Q (x_k - goal) + sum_k (u_k - u_{k-1})' R (u_k - u_{k-1})          mpc_cbf.py:144,176-180
//     equalities   x_0 = x0,  x_{k+1} = x_k + (f(x_k) + g(x_k) u_k) dt                                    mpc_cbf.py:135-141, dynamic_unicycle2D.py:42-73
//     inequalities -(dd_h + (a1 + a2) d_h + a1 a2 h)(x_k, u_k) <= 0 for every obstacle slot                mpc_cbf.py:295-325, dynamic_unicycle2D.py:188-238
//     bounds       |v_k| <= v_max (k = 0 .. N), |a_k| <= a_max, |w_k| <= w_max                              mpc_cbf.py:193-199
//     start        x_k = x0 for every k, u_k = u_prev                                                       mpc_cbf.py:366-369
// and whatever IPOPT holds at the end is applied (mpc_cbf.py:384, status hard-wired 'optimal', :10) -- on the ~10 % of config-3 NLPs that
// have no feasible point that is the restoration phase's iterate, which the condensed kernel (mpc_cbf.hip, kernel 3) does not reproduce
// (tools/exp_ms_vs_condensed.py on all 4096 draws: 29 % of the infeasible solves differ by more than 1e-3 in u_0).
//
//   lane k <= N    owns x_k and, for k < N, u_k, the four dynamics rows F(x_k, u_k) - x_{k+1} = 0 with their multipliers and the K rows
//                  d_kj <= 0 with slack, multiplier and slack-bound multiplier -- all in registers; the two barrier points beyond p_k come
//                  from the closed form of step o step (no rollout), first and second derivatives written out
//   Newton step    rows condensed into the stage's 6 x 6 block; Riccati recursion over the augmented state (dx_k, du_{k-1}) (the input-rate
//                  penalty couples neighbouring inputs): backward sweep with ONE ENTRY OF THE 8 x 8 STAGE MATRIX PER LANE (three LDS round
//                  trips per stage), forward sweep redundantly in every lane (no exchange at all); inertia = every 2 x 2 input block
//                  positive definite (Algorithm IC's delta_w ladder); dynamics multipliers from the value function
//   globalisation  IPOPT's filter, switching condition + Armijo, alpha_min -> the restoration phase as a mode of the same loop (elastic
//                  variables on the CBF rows, own filter, return test against the regular filter, infeasibility certificate), its row
//                  state in LDS; fraction to the boundary, monotone mu, gradient-based scaling, bound push / relaxation, least-square
//                  initial multipliers, kappa_sigma, safe slacks
//
// This header is plain C++ over a context `C` (LDS pointer, lane, barrier, wave reductions): csrc/mpc_du_ms.hip instantiates it for the
// device; tools/du_ms_host.cpp runs the SAME code on the host with one thread per lane and a barrier (a debugging aid, never shipped).
#pragma once
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/safe_control_amd.h"

#ifndef SC_HD
#define SC_HD __host__ __device__
#endif
#ifndef SC_DUMS_INLINE
#define SC_DUMS_INLINE inline
#endif
// -DSC_DUMS_PROF: cycles per phase of the iteration, summed over the solve, into row max_iter of the trace (tools/exp_du_ms_phases.py)
#ifdef SC_DUMS_PROF
#define DPROF_T0 long long _t0 = cx.clock();
#define DPROF_ADD(i) { const long long _t1 = cx.clock(); prof[i] += (double)(_t1 - _t0); _t0 = _t1; }
#else
#define DPROF_T0
#define DPROF_ADD(i)
#endif
#ifdef __HIP_DEVICE_COMPILE__
#define SC_UNROLL _Pragma("unroll")
#else
#define SC_UNROLL
#endif

namespace sc {
namespace dums {

constexpr int NX = 4, NU = 2, NV = 6, NA = 6, NQ = 8;
constexpr int KS_MAX = 16, NFILT = 128, TRACE_W = 8;      // NFILT: entries per filter (cleared with every decrease of mu; 28 seen on configs[2], 40+ at N = 20: tiny steps add one each until the stall rule ends the solve)
constexpr double EPS_ = 2.220446049250313e-16;
constexpr int ST_FILTER_FULL = 7;                 // (internal) the filter ran over: reported as SC_STATUS_INACCURATE

// Models (template parameter of Wave): both have x+ = x + (f(x) + g(x) u) dt with the inputs entering states 2 and 3 only, two-step barrier rows
// on the positions, and differ in the geometry, in its curvature and in the state bound:
//   M_DU  DynamicUnicycle2D  x = (px, py, theta, v), u = (a, omega); |v_k| <= v_max                        (the header comment above)
//   M_DI  DoubleIntegrator2D x = (px, py, vx, vy), u = (ax, ay), held here as (ay, ax) so that B = dt [[0, 0], [0, 0], [0, 1], [1, 0]] like
//         the unicycle's (the launcher swaps R, the bounds and the I/O); no state bound; the barrier's robot.step rescales the velocity to
//         norm v_max where it is above it (double_integrator2D.py:79-107,225-226): v_max is that norm            mpc_cbf.py:28-30,56-59,196-200
//   M_KB  KinematicBicycle2D x = (px, py, theta, v), u = (a, beta): x+ = x + dt (v (cos th - beta sin th), v (sin th + beta cos th), v beta / L_r, a);
//         |v_k| <= v_max; the barrier's robot.step clips the speed to [v_min, v_max] (kinematic_bicycle2D.py:112-123,175-199)     mpc_cbf.py:31-33,64-66,202-208
// M_KB runs the GENERAL stage layout: the inputs enter the positions directly, so a stage keeps the two non-trivial columns of A and all of B
// (16 values) where the other two keep four entries of A and know B; the recursion, the sweep, J'y and the rows read those.
//   M_UNI Unicycle2D x = (px, py, theta) -- held as four states, the last one idle (x_3+ = x_3 = 0, no cost) --, u = (v, omega): x+ = x + dt (v cos th,
//         v sin th, omega); no state bound; ONE-step rows h(p1) - (1 - alpha1) h(p0) >= 0 (alpha1 carries the reference's `alpha`)  mpc_cbf.py:22-24,52-53,188-192,312-315
//   M_SI  SingleIntegrator2D x = (px, py) -- two idle states --, u = (vx, vy): x+ = x + dt u; no state bound; one-step rows like M_UNI       mpc_cbf.py:19-21,49-51,183-187
enum { M_DU = 0, M_DI = 1, M_KB = 2, M_UNI = 3, M_SI = 4 };
constexpr bool general_layout(int model) { return model == M_KB || model == M_UNI || model == M_SI; }
struct Params {
    int N, K;
    double dt, Q[4], R[2], alpha1, alpha2, beta, radius, u_lo[2], u_hi[2], v_max;
    double v_min = 0.0, inv_Lr = 0.0;                   // M_KB: lower end of robot.step's speed clip; 1 / rear_ax_dist
};

SC_HD inline int sym6(int a, int b) { return a <= b ? a * 6 - a * (a - 1) / 2 + (b - a) : b * 6 - b * (b - 1) / 2 + (a - b); }

struct Lds {
    int OB, AB, H, G, C, KG, PX, LAM, XS, US, YS, Pa, Pb, T, QU, FP, FT, FP2, FT2, SC, Y0, RW, XR, total;
    int N, K, ABW, OBW;
    SC_HD Lds(int N_, int K_, bool gen = false, bool se = false) : N(N_), K(K_ < 1 ? 1 : K_), ABW(gen ? 16 : 4), OBW(se ? 8 : 3) {
        int o = 0;
        auto take = [&](int c) { int r = o; o += c; return r; };
        OB = take(OBW * K); AB = take(N * ABW); H = take((N + 1) * 21); G = take((N + 1) * 6); C = take((N + 1) * 4);
        KG = take(N * 14); PX = take(N * 28);
        // Slots whose lifetimes do not overlap share storage (19.7 KB per problem at N = 10, K = 8: eight problems per CU):
        //   YS (multipliers as the neighbours see them: written and read at the start of an evaluation) / LAM (costates = multiplier steps:
        //   written after the recursion, read at the update);
        //   the recursion's workspace T | QU lies over the exchange vectors XS | US, which an evaluation rewrites before it reads them.
        LAM = take((N + 2) * 4); YS = LAM;
        const int ex = (N + 2) * 6 < 72 ? 72 : (N + 2) * 6;
        XS = take(ex); US = XS + (N + 2) * 4; T = XS; QU = XS + 54;
        Pa = take(42); Pb = take(42);
        FP = take(NFILT); FT = take(NFILT); FP2 = take(NFILT); FT2 = take(NFILT); SC = take(4 + K); Y0 = take(4);
        RW = take(16 * K * N); XR = take(6 * (N + 1));
        total = o;
    }
};

// Obstacle rows of a launch that may hold superellipsoids (Wave<.., SE = true>; dynamic_unicycle2D.py:204-220, double_integrator2D.py:238-254,
// single_integrator2D.py:162-178: h = |q_x|^e / (a + R)^e + |q_y|^e / (b + R)^e - 1 in the obstacle's frame, with the reference's clamps) are
// packed once per problem: [ox, oy, off | 1 / (a + R)^e, 1 / (b + R)^e, e, cos th, sin th, flag]; a circle keeps off = beta (R + r)^2 in slot 2.
template <class PowF>
SC_HD inline void pack_obstacle(const double o[7], double radius, double beta, PowF powf, double out[8]) {
    out[0] = o[0]; out[1] = o[1];
    if (o[6] < 0.5) { const double d = radius + o[2]; out[2] = beta * d * d; out[3] = 0.0; out[4] = 2.0; out[5] = 1.0; out[6] = 0.0; out[7] = 0.0; return; }
    const double a = fmax(fabs(o[2]), 1e-3) + radius, b = fmax(fabs(o[3]), 1e-3) + radius, e = fmax(fabs(o[4]), 2.0);
    out[2] = 1.0 / powf(a, e); out[3] = 1.0 / powf(b, e); out[4] = e; out[5] = cos(o[5]); out[6] = sin(o[5]); out[7] = 1.0;
}
struct HP { double h, gx, gy, hxx, hxy, hyy; };      // a barrier at a point: value, gradient, Hessian over (p_x, p_y)

// ---- Riccati recursion with defects (oracle/ms_ipopt.py: _riccati_backward / _riccati_solve, hard dynamics) ----------------------------
// LDS in: AB[k] = (a02, a03, a12, a13) of A_k = I + [[0, 0, a02, a03], [0, 0, a12, a13], 0, 0] (B = dt [[0, 0], [0, 0], [0, 1], [1, 0]]),
// H[k] (upper-packed 6 x 6 over (x_k, u_k); H[N]: its x block), G[k] (gradient, 6), C[k + 1] (defect of stage k's dynamics), C[0] = dx_0;
// cpl[j] = 2 df R_j: the (u_{k-1}, u_k) cross term, -cpl on the (v, u) entries of stage k >= 1.
// Out: KG[k] = gains (K: 2 x 6 over (dx_k, du_{k-1}), then kff: 2), PX[k] = the x rows of P_k (4 x 6) and of p_k (4); false (wave-uniform)
// when an input block is not positive definite.
// Index t of the 8 x 8 stage matrix: 0..3 = x, 4..5 = v (= u_{k-1}), 6..7 = u; lane l owns entry (l >> 3, l & 7) -- computed as the entry
// (min, max), so the matrix is symmetric by construction (the oracle averages P with its transpose).  Per stage:
//     T = P G (6 x 8) and t = P c + p        G = [[A, 0, B], [0, 0, I]]: xi+ = G (xi, u) + (c, 0)
//     Q = S + G' T,  q = s + G' t            S = stage block (H over (x, u), -cpl on (v_i, u_i))
//     Quu = L L',  Y = L^-1 Q[u, :],  P' = Q - Y' Y,  p' = q - Y' L^-1 q_u,  K = -L^-T Y,  kff = -L^-T L^-1 q_u
// a lane's column of G~ = [G | (c, 0, 0)] (6 x 9, G = [[A, 0, B], [0, 0, I]]): what does not change from stage to stage is worked out once
//     column 0, 1: e_0, e_1;  2: (a02, a12, 1, 0, 0, 0);  3: (a03, a13, 0, 1, 0, 0);  4, 5: 0;  6: (0, 0, 0, dt, 1, 0);  7: (0, 0, dt, 0, 0, 1);  8: (c, 0, 0)
struct GCol {
    double b[NA], s2, s3, s8;
    SC_HD void set(int t, double dt) {
        b[0] = t == 0 ? 1.0 : 0.0; b[1] = t == 1 ? 1.0 : 0.0; b[2] = t == 2 ? 1.0 : (t == 7 ? dt : 0.0); b[3] = t == 3 ? 1.0 : (t == 6 ? dt : 0.0);
        b[4] = t == 6 ? 1.0 : 0.0; b[5] = t == 7 ? 1.0 : 0.0;
        s2 = t == 2 ? 1.0 : 0.0; s3 = t == 3 ? 1.0 : 0.0; s8 = t == 8 ? 1.0 : 0.0;
    }
    // (a02, a03, a12, a13) of the stage
    SC_HD void at(double a02, double a03, double a12, double a13, double g[NA]) const {
        g[0] = b[0] + s2 * a02 + s3 * a03; g[1] = b[1] + s2 * a12 + s3 * a13; g[2] = b[2]; g[3] = b[3]; g[4] = b[4]; g[5] = b[5];
    }
};

// general layout: AB[k] = the columns of [A_2 | A_3 | B_0 | B_1] (4 rows each, 16 values); column t of G~: 0, 1: e_t; 2, 3: A's column; 4, 5: 0;
// 6, 7: (B's column, e_{t-6});  8: (c, 0, 0)
struct GColG {
    double b[NA], sel, s8;
    int ci;
    SC_HD void set(int t, double) {
        SC_UNROLL for (int m = 0; m < NA; ++m) b[m] = 0.0;
        if (t < 2) b[t] = 1.0;
        if (t == 6) b[4] = 1.0;
        if (t == 7) b[5] = 1.0;
        sel = (t == 2 || t == 3 || t == 6 || t == 7) ? 1.0 : 0.0;
        ci = t == 3 ? 4 : (t == 6 ? 8 : (t == 7 ? 12 : 0));
        s8 = t == 8 ? 1.0 : 0.0;
    }
    template <class P>
    SC_HD void at(P ab, double g[NA]) const {
        SC_UNROLL for (int m = 0; m < 4; ++m) g[m] = b[m] + sel * ab[ci + m];
        g[4] = b[4]; g[5] = b[5];
    }
};

template <class Cx, bool GEN = false>
SC_HD SC_DUMS_INLINE bool riccati_backward(Cx& cx, const Lds& L, const int N, const double dt, const double cpl0, const double cpl1) {
    typename Cx::ptr lds = cx.lds;
    const int lane = cx.lane, r = lane >> 3, c = lane & 7;
    const int rr = r < c ? r : c, cc = r < c ? c : r;
    // T phase: lane l < 54 owns T~[l / 9][l % 9] (column 8 = the affine column P c + p)
    const int tr = lane < 54 ? lane / 9 : 0, tc = lane < 54 ? lane % 9 : 0;
    typename std::conditional<GEN, GColG, GCol>::type gT, gR, gC;
    gT.set(tc, dt); gR.set(rr, dt); gC.set(c, dt);
    // S[rr][cc]: an entry of H (index map 0..3 -> x, 6..7 -> u), or -cpl on (v_i, u_i) for k >= 1; s[c]: an entry of the gradient
    const int hr = rr < 4 ? rr : (rr >= 6 ? rr - 2 : -1), hc = cc < 4 ? cc : (cc >= 6 ? cc - 2 : -1), hq = c < 4 ? c : (c >= 6 ? c - 2 : -1);
    const bool inH = hr >= 0 && hc >= 0;
    const int oH = inH ? sym6(hr, hc) : 0, oG = hq >= 0 ? hq : 0;
    const double mH = inH ? 1.0 : 0.0, mG = hq >= 0 ? 1.0 : 0.0;
    const double cplv = (rr == 4 && cc == 6) ? -cpl0 : ((rr == 5 && cc == 7) ? -cpl1 : 0.0);
    int cur = L.Pa, nxt = L.Pb;
    // P_N = x block of H_N, p_N = g_N
    cx.sync();
    if (r < 6 && c < 6) lds[cur + r * 6 + c] = (r < 4 && c < 4) ? lds[L.H + N * 21 + sym6(r, c)] : 0.0;
    if (r == 6 && c < 6) lds[cur + 36 + c] = c < 4 ? lds[L.G + N * 6 + c] : 0.0;
    cx.sync();
    for (int k = N - 1; k >= 0; --k) {
        const typename Cx::ptr ab = lds + L.AB + k * L.ABW;
        const typename Cx::ptr cd = lds + L.C + (k + 1) * 4;
        double a02 = 0.0, a03 = 0.0, a12 = 0.0, a13 = 0.0;
        if constexpr (!GEN) { a02 = ab[0]; a03 = ab[1]; a12 = ab[2]; a13 = ab[3]; }
        double g[NA];
        // T~ = P G~ (+ p on the affine column)
        if constexpr (GEN) gT.at(ab, g); else gT.at(a02, a03, a12, a13, g);
        g[0] += gT.s8 * cd[0]; g[1] += gT.s8 * cd[1]; g[2] += gT.s8 * cd[2]; g[3] += gT.s8 * cd[3];
        {
            double v = gT.s8 * lds[cur + 36 + tr];
            SC_UNROLL for (int m = 0; m < NA; ++m) v += lds[cur + tr * 6 + m] * g[m];
            if (lane < 54) lds[L.T + (tc < 8 ? tr * 8 + tc : 48 + tr)] = v;
        }
        cx.sync();
        // Q[rr][cc] = S + G[:, rr]' T[:, cc];  q[c] = s[c] + G[:, c]' t  (every lane of column c)
        if constexpr (GEN) gR.at(ab, g); else gR.at(a02, a03, a12, a13, g);
        double qv = mH * lds[L.H + k * 21 + oH] + (k >= 1 ? cplv : 0.0);
        SC_UNROLL for (int m = 0; m < NA; ++m) qv += g[m] * lds[L.T + m * 8 + cc];
        if constexpr (GEN) gC.at(ab, g); else gC.at(a02, a03, a12, a13, g);
        double ql = mG * lds[L.G + k * 6 + oG];
        SC_UNROLL for (int m = 0; m < NA; ++m) ql += g[m] * lds[L.T + 48 + m];
        if (r >= 6) lds[L.QU + (r - 6) * 8 + c] = qv;
        if (r == 0 && c >= 6) lds[L.QU + 16 + (c - 6)] = ql;
        cx.sync();
        const double q66 = lds[L.QU + 6], q67 = lds[L.QU + 7], q77 = lds[L.QU + 8 + 7];
        if (!(q66 > 0.0) || !(q66 < 1e300)) return false;
        const double li0 = cx.rsqrt(q66), l10 = q67 * li0, d11 = q77 - l10 * l10;
        if (!(d11 > 0.0) || !(d11 < 1e300)) return false;
        const double li1 = cx.rsqrt(d11);
        const double ya0 = lds[L.QU + rr] * li0, ya1 = (lds[L.QU + 8 + rr] - l10 * ya0) * li1;
        const double yb0 = lds[L.QU + cc] * li0, yb1 = (lds[L.QU + 8 + cc] - l10 * yb0) * li1;
        const double yq0 = lds[L.QU + 16] * li0, yq1 = (lds[L.QU + 17] - l10 * yq0) * li1;
        if (r < 6 && c < 6) {
            const double pv = qv - (ya0 * yb0 + ya1 * yb1);
            lds[nxt + r * 6 + c] = pv;
            if (r < 4) lds[L.PX + k * 28 + r * 6 + c] = pv;
        }
        if (r == 0 && c < 6) {
            const double pl = ql - (yb0 * yq0 + yb1 * yq1);                // (row 0: rr = 0, cc = c)
            lds[nxt + 36 + c] = pl;
            if (c < 4) lds[L.PX + k * 28 + 24 + c] = pl;
        }
        if (r == 6) {                                                      // gains of column c (= rr for c < 6), kff on (6, 6): K = -L^-T y
            const double y0 = c < 6 ? ya0 : yq0, y1 = c < 6 ? ya1 : yq1;
            const double k1 = y1 * li1, k0 = (y0 - l10 * k1) * li0;
            if (c < 6) { lds[L.KG + k * 14 + c] = -k0; lds[L.KG + k * 14 + 6 + c] = -k1; }
            else if (c == 6) { lds[L.KG + k * 14 + 12] = -k0; lds[L.KG + k * 14 + 13] = -k1; }
        }
        cx.sync();
        const int t_ = cur; cur = nxt; nxt = t_;
    }
    return true;
}

// forward sweep, redundantly in every lane: lane k keeps xi_k = (dx_k, du_{k-1}) and du_k
template <class Cx, bool GEN = false>
SC_HD SC_DUMS_INLINE void riccati_forward(Cx& cx, const Lds& L, const int N, const double dt, const int k_me, double xi_me[NA], double du_me[NU]) {
    typename Cx::ptr lds = cx.lds;
    double xi[NA] = {lds[L.C + 0], lds[L.C + 1], lds[L.C + 2], lds[L.C + 3], 0.0, 0.0};
    SC_UNROLL for (int m = 0; m < NA; ++m) xi_me[m] = xi[m];
    du_me[0] = du_me[1] = 0.0;
    for (int k = 0; k < N; ++k) {
        const typename Cx::ptr kg = lds + L.KG + k * 14;
        double du0 = kg[12], du1 = kg[13];
        SC_UNROLL for (int m = 0; m < NA; ++m) { du0 += kg[m] * xi[m]; du1 += kg[6 + m] * xi[m]; }
        if (k == k_me) { du_me[0] = du0; du_me[1] = du1; }
        const typename Cx::ptr ab = lds + L.AB + k * L.ABW;
        const typename Cx::ptr cd = lds + L.C + (k + 1) * 4;
        double n0, n1, n2, n3;
        if constexpr (GEN) {
            double nn[4];
            SC_UNROLL for (int r = 0; r < 4; ++r) nn[r] = (r < 2 ? xi[r] : 0.0) + ab[r] * xi[2] + ab[4 + r] * xi[3] + ab[8 + r] * du0 + ab[12 + r] * du1 + cd[r];
            n0 = nn[0]; n1 = nn[1]; n2 = nn[2]; n3 = nn[3];
        } else {
            n0 = xi[0] + ab[0] * xi[2] + ab[1] * xi[3] + cd[0]; n1 = xi[1] + ab[2] * xi[2] + ab[3] * xi[3] + cd[1];
            n2 = xi[2] + dt * du1 + cd[2]; n3 = xi[3] + dt * du0 + cd[3];
        }
        xi[0] = n0; xi[1] = n1; xi[2] = n2; xi[3] = n3; xi[4] = du0; xi[5] = du1;
        if (k + 1 == k_me) SC_UNROLL for (int m = 0; m < NA; ++m) xi_me[m] = xi[m];
    }
}

SC_HD inline int group_lanes(int N) { return (N + 1) * 4 <= 64 ? 4 : ((N + 1) * 2 <= 64 ? 2 : 1); }
SC_HD inline bool cmp_le(double lhs, double rhs, double bas) { return lhs - rhs <= 10.0 * EPS_ * fabs(bas); }

template <class Cx, int MODEL = M_DU, bool SE = false>
struct Wave {
    static constexpr bool XB = MODEL == M_DU || MODEL == M_KB;          // the model has the state bound |x_3| <= v_max
    static constexpr bool GEN = general_layout(MODEL);
    static constexpr int NXT = MODEL == M_UNI ? 3 : (MODEL == M_SI ? 2 : NX);     // states of the reference's model (the count in the error scaling)
    Cx& cx;
    const Params& P;
    const sc_ipopt_params& O;
    typename Cx::ptr lds;
    const Lds L;
    const int lane, N, K;
    // G lanes per stage (4 while (N + 1) * 4 <= 64, else 2, else 1): they hold the stage's state in copies (same arithmetic, no exchange) and
    // SPLIT ITS ROWS -- lane q of the group walks rows q, q + G, .. --; row sums that the stage needs as a whole (the condensed block, J'y) are
    // added over the group (cx.gsum: DPP within a quad), everything else ends in a wave reduction anyway.  The group's first lane (`lead`)
    // owns the stage's share of the wave sums and its LDS slots.
    const int G, q;
    const bool act, stg;               // lane belongs to a state (k <= N) / a stage with inputs and rows (k < N)
    const bool acl, stl;               // ... and is the first lane of its group
    const int k;
    double x0[NX], uprev[NU], xg[2];
    double w0, w1, w2;
    double x[NX], u[NU], yc[NX];
    double xbL, xbU, ubL[NU], ubU[NU];                   // (relaxed, adjustable) bounds: v; inputs
    double zxL, zxU, zuL[NU], zuU[NU];
    double df;
    double rc[NX];                                       // scaled residuals of my dynamics rows (last evaluation)
    double dx[NX], dvv[NU], du[NU];                      // my part of the step: dx_k, du_{k-1}, du_k
    double dzxL, dzxU, dzuL[NU], dzuU[NU];
    int nfilt;
    bool filt_over = false;
    double dw_last, last_dw;
    bool rs = false;                   // inside the restoration phase (wave-uniform)
    double zeta = 0.0;
    int fpo, fto;                      // LDS offsets of the filter in use
#ifdef SC_DUMS_PROF
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // eval2 (errors), errors + mu, eval2 (build), riccati backward, finish_step, step lengths + barrier, line search, update
#endif
    // Row state in LDS, [slot][row][stage] (a lane walks its K rows in a loop: dynamic row index, no register arrays): the restoration's
    // n, p, their multipliers and the four steps; slack s, row multiplier yd, slack-bound multiplier vU, (relaxed, adjustable) slack bound sU,
    // scaled row value dv of the last evaluation, and the steps ds, dyd, dvU
    enum { R_N, R_P, R_ZN, R_ZP, R_DN, R_DP, R_DZN, R_DZP, R_S, R_YD, R_VU, R_SU, R_DV, R_DS, R_DYD, R_DVU };
    SC_HD int ri(int slot, int j) const { return L.RW + (slot * K + j) * N + k; }
    SC_HD int XRi(int i) const { return L.XR + i * (N + 1) + k; }
    SC_HD double dr2(int i) const { const double a = fabs(lds[XRi(i)]); return a > 1.0 ? 1.0 / (a * a) : 1.0; }      // D_R^2 = 1 / max(1, |w_R|)^2

    SC_HD Wave(Cx& cx_, const Params& P_, const sc_ipopt_params& O_)
        : cx(cx_), P(P_), O(O_), lds(cx_.lds), L(P_.N, P_.K, general_layout(MODEL), SE), lane(cx_.lane), N(P_.N), K(P_.K),
          G(group_lanes(P_.N)), q(cx_.lane % group_lanes(P_.N)), act(cx_.lane / group_lanes(P_.N) <= P_.N), stg(cx_.lane / group_lanes(P_.N) < P_.N),
          acl(cx_.lane / group_lanes(P_.N) <= P_.N && cx_.lane % group_lanes(P_.N) == 0), stl(cx_.lane / group_lanes(P_.N) < P_.N && cx_.lane % group_lanes(P_.N) == 0),
          k(cx_.lane / group_lanes(P_.N) <= P_.N ? cx_.lane / group_lanes(P_.N) : 0) {
        const double g1 = P.alpha1 + P.alpha2, g2 = P.alpha1 * P.alpha2;
        w0 = 1.0 - g1 + g2; w1 = g1 - 2.0; w2 = 1.0;
        if constexpr (MODEL == M_UNI || MODEL == M_SI) { w0 = P.alpha1 - 1.0; w1 = 1.0; w2 = 0.0; }              // d_h + alpha h_k
        nfilt = 0; dw_last = 0.0; last_dw = 0.0; fpo = L.FP; fto = L.FT;
    }
    SC_HD void sync() const { cx.sync(); }
    SC_HD double dgc(int i) const { return lds[L.SC + i]; }
    SC_HD double dgd(int j) const { return lds[L.SC + 4 + j]; }

    // ---- the stage's geometry: F(x, u), the three barrier points and (level 1) the derivatives of the points ---------------------------
    struct Geo {
        double c, s, c1, s1, v1;
        double F[NX], p1[2], p2[2];
        double a02, a03, a12, a13;       // d p1 / d (theta, v)  = the non-trivial entries of A
        double g02, g03, g12, g13;       // d (p2 - p1) / d (theta, v);  d / d a = dt (g03, g13),  d / d omega = dt (g02, g12)
        // general layout: ab = [A_2 | A_3 | B_0 | B_1] (d F / d (x_2, x_3, u_0, u_1), 4 rows each), j2 = d p2 / d (x_2, x_3, u_0, u_1) (2 x 4),
        // and what the curvature needs again (model-specific)
        double ab[16], j2[2][4], cw[8];
    };
    double tc_ = 1.0, ts_ = 0.0, tc1_ = 1.0, ts1_ = 0.0;        // cos / sin of theta_k and of theta_k + dt omega_k at the ITERATE (eval2 sets them, finish_step reuses them)
    SC_HD void geometry(const double* xs, const double* us, Geo& g, bool cached = false) const {
        const double dt = P.dt;
        if constexpr (MODEL == M_SI) {
            g.c = g.s = g.c1 = g.s1 = 0.0; g.v1 = 0.0;
            g.F[0] = xs[0] + dt * us[0]; g.F[1] = xs[1] + dt * us[1]; g.F[2] = xs[2]; g.F[3] = xs[3];
            g.p1[0] = g.F[0]; g.p1[1] = g.F[1]; g.p2[0] = g.F[0]; g.p2[1] = g.F[1];
            SC_UNROLL for (int i = 0; i < 16; ++i) g.ab[i] = 0.0;
            g.ab[2] = 1.0; g.ab[7] = 1.0; g.ab[8] = dt; g.ab[13] = dt;                          // A_2, A_3: the idle states; B_0 = dt e_0, B_1 = dt e_1
            SC_UNROLL for (int i = 0; i < 4; ++i) { g.j2[0][i] = g.ab[4 * i]; g.j2[1][i] = g.ab[4 * i + 1]; }
            g.a02 = g.a12 = g.a03 = g.a13 = 0.0;
            g.g02 = g.g03 = g.g12 = g.g13 = 0.0;
            return;
        }
        if constexpr (MODEL == M_UNI) {
            const double v = us[0];
            if (cached) { g.c = tc_; g.s = ts_; } else cx.sincos(xs[2], g.s, g.c);
            g.c1 = g.s1 = 0.0; g.v1 = v;
            g.F[0] = xs[0] + dt * v * g.c; g.F[1] = xs[1] + dt * v * g.s; g.F[2] = xs[2] + dt * us[1]; g.F[3] = xs[3];
            g.p1[0] = g.F[0]; g.p1[1] = g.F[1]; g.p2[0] = g.F[0]; g.p2[1] = g.F[1];              // (w2 = 0: the second point is not read)
            SC_UNROLL for (int i = 0; i < 16; ++i) g.ab[i] = 0.0;
            g.ab[0] = -dt * v * g.s; g.ab[1] = dt * v * g.c; g.ab[2] = 1.0;                     // A_2 (theta)
            g.ab[7] = 1.0;                                                                      // A_3 (the idle state)
            g.ab[8] = dt * g.c; g.ab[9] = dt * g.s;                                             // B_0 (v)
            g.ab[14] = dt;                                                                      // B_1 (omega)
            SC_UNROLL for (int i = 0; i < 4; ++i) { g.j2[0][i] = g.ab[4 * i]; g.j2[1][i] = g.ab[4 * i + 1]; }
            g.a02 = g.ab[0]; g.a12 = g.ab[1]; g.a03 = 0.0; g.a13 = 0.0;
            g.g02 = g.g03 = g.g12 = g.g13 = 0.0;
            return;
        }
        if constexpr (MODEL == M_KB) {
            const double v = xs[3], be = us[1], iL = P.inv_Lr;
            if (cached) { g.c = tc_; g.s = ts_; } else cx.sincos(xs[2], g.s, g.c);
            const double fv = g.c - be * g.s, pv = g.s + be * g.c, ph = v * fv, ps = v * pv;          // phi, psi and their d / dv
            g.F[0] = xs[0] + dt * ph; g.F[1] = xs[1] + dt * ps; g.F[2] = xs[2] + dt * (v * be * iL); g.F[3] = xs[3] + dt * us[0];
            if (cached) { g.c1 = tc1_; g.s1 = ts1_; } else cx.sincos(g.F[2], g.s1, g.c1);
            const bool in = g.F[3] >= P.v_min && g.F[3] <= P.v_max;
            const double v1 = in ? g.F[3] : (g.F[3] < P.v_min ? P.v_min : P.v_max), chi = in ? 1.0 : 0.0;      // robot.step's clip (casadi: slope 1 inside, 0 outside)
            const double fv1 = g.c1 - be * g.s1, pv1 = g.s1 + be * g.c1, ph1 = v1 * fv1, ps1 = v1 * pv1;
            g.v1 = v1;
            g.p1[0] = g.F[0]; g.p1[1] = g.F[1];
            g.p2[0] = g.F[0] + dt * ph1; g.p2[1] = g.F[1] + dt * ps1;
            // [A_2 | A_3 | B_0 | B_1]
            g.ab[0] = -dt * ps; g.ab[1] = dt * ph; g.ab[2] = 1.0; g.ab[3] = 0.0;
            g.ab[4] = dt * fv; g.ab[5] = dt * pv; g.ab[6] = dt * be * iL; g.ab[7] = 1.0;
            g.ab[8] = 0.0; g.ab[9] = 0.0; g.ab[10] = 0.0; g.ab[11] = dt;
            g.ab[12] = -dt * v * g.s; g.ab[13] = dt * v * g.c; g.ab[14] = dt * v * iL; g.ab[15] = 0.0;
            // q = (theta_1, v_1, beta) over (theta, v, a, beta)
            const double qt[4] = {1.0, dt * be * iL, 0.0, dt * v * iL}, qv[4] = {0.0, chi, chi * dt, 0.0};
            SC_UNROLL for (int i = 0; i < 4; ++i) {
                const double qb = i == 3 ? 1.0 : 0.0;
                g.j2[0][i] = g.ab[4 * i + 0] + dt * (-ps1 * qt[i] + fv1 * qv[i] - v1 * g.s1 * qb);
                g.j2[1][i] = g.ab[4 * i + 1] + dt * (ph1 * qt[i] + pv1 * qv[i] + v1 * g.c1 * qb);
            }
            g.cw[0] = chi; g.cw[1] = ph; g.cw[2] = ps; g.cw[3] = ph1; g.cw[4] = ps1;
            g.a02 = g.ab[0]; g.a12 = g.ab[1]; g.a03 = g.ab[4]; g.a13 = g.ab[5];
            g.g02 = g.g03 = g.g12 = g.g13 = 0.0;
            return;
        }
        if constexpr (MODEL == M_DI) {
            // p1 = p + dt v;  w = v + dt a,  p2 = p1 + dt w min(1, v_max / |w|)   (the rescaling of robot.step; the model's x_next has none)
            g.c = g.s = g.c1 = g.s1 = 0.0;
            g.F[0] = xs[0] + dt * xs[2]; g.F[1] = xs[1] + dt * xs[3]; g.F[2] = xs[2] + dt * us[1]; g.F[3] = xs[3] + dt * us[0];
            g.p1[0] = g.F[0]; g.p1[1] = g.F[1];
            g.a02 = dt; g.a03 = 0.0; g.a12 = 0.0; g.a13 = dt;
            const double w0_ = g.F[2], w1_ = g.F[3], vm = sqrt(w0_ * w0_ + w1_ * w1_);
            g.v1 = vm;
            if (vm > P.v_max) {
                const double sc = P.v_max / vm, i3 = P.v_max / (vm * vm * vm);
                g.p2[0] = g.F[0] + dt * (w0_ * sc); g.p2[1] = g.F[1] + dt * (w1_ * sc);
                g.g02 = dt * (sc - w0_ * w0_ * i3); g.g03 = dt * (-w0_ * w1_ * i3); g.g12 = g.g03; g.g13 = dt * (sc - w1_ * w1_ * i3);
            } else {
                g.p2[0] = g.F[0] + dt * w0_; g.p2[1] = g.F[1] + dt * w1_;
                g.g02 = dt; g.g03 = 0.0; g.g12 = 0.0; g.g13 = dt;
            }
            return;
        }
        g.F[2] = xs[2] + dt * us[1];
        if (cached) { g.c = tc_; g.s = ts_; g.c1 = tc1_; g.s1 = ts1_; }
        else { cx.sincos(xs[2], g.s, g.c); cx.sincos(g.F[2], g.s1, g.c1); }
        g.F[0] = xs[0] + dt * xs[3] * g.c; g.F[1] = xs[1] + dt * xs[3] * g.s; g.F[3] = xs[3] + dt * us[0];
        g.v1 = g.F[3];
        g.p1[0] = g.F[0]; g.p1[1] = g.F[1];
        g.p2[0] = g.F[0] + dt * g.v1 * g.c1; g.p2[1] = g.F[1] + dt * g.v1 * g.s1;
        g.a02 = -dt * xs[3] * g.s; g.a03 = dt * g.c; g.a12 = dt * xs[3] * g.c; g.a13 = dt * g.s;
        g.g02 = -dt * g.v1 * g.s1; g.g03 = dt * g.c1; g.g12 = dt * g.v1 * g.c1; g.g13 = dt * g.s1;
    }
    // cbf_j = w0 h(p0) + w1 h(p1) + w2 h(p2),  h(p) = |p - c_j|^2 - beta (R + r_j)^2;  a = grad cbf_j over (px, py, theta, v, a, omega) when asked
    // SE: h_j at a point with its first and second derivatives (circle or superellipsoid by the row's flag)
    SC_HD void hpoint(int j, double px, double py, HP& o) const {
        const typename Cx::ptr ob = lds + L.OB + 8 * j;
        const double ex = px - ob[0], ey = py - ob[1];
        if (ob[7] < 0.5) { o.h = ex * ex + ey * ey - ob[2]; o.gx = 2.0 * ex; o.gy = 2.0 * ey; o.hxx = 2.0; o.hxy = 0.0; o.hyy = 2.0; return; }
        const double e = ob[4], ct = ob[5], st = ob[6];
        const double qx = ct * ex + st * ey, qy = ct * ey - st * ex;
        const double px2 = cx.pow(fabs(qx), e - 2.0) * ob[2], py2 = cx.pow(fabs(qy), e - 2.0) * ob[3];      // |q|^(e - 2) / (a + R)^e
        o.h = qx * qx * px2 + qy * qy * py2 - 1.0;
        const double hx = e * qx * px2, hy = e * qy * py2, hxx = e * (e - 1.0) * px2, hyy = e * (e - 1.0) * py2;
        o.gx = ct * hx - st * hy; o.gy = st * hx + ct * hy;
        o.hxx = ct * ct * hxx + st * st * hyy; o.hxy = ct * st * (hxx - hyy); o.hyy = st * st * hxx + ct * ct * hyy;
    }
    SC_HD double row(const double* xs, const Geo& g, int j, double* a = nullptr, HP* hp = nullptr) const {
        if constexpr (SE) {
            HP h0, h1, h2;
            hpoint(j, xs[0], xs[1], h0); hpoint(j, g.p1[0], g.p1[1], h1);
            if (w2 != 0.0) hpoint(j, g.p2[0], g.p2[1], h2); else { h2.h = h2.gx = h2.gy = h2.hxx = h2.hxy = h2.hyy = 0.0; }
            if (a) {
                const double dt = P.dt;
                a[0] = w0 * h0.gx + w1 * h1.gx + w2 * h2.gx; a[1] = w0 * h0.gy + w1 * h1.gy + w2 * h2.gy;
                if constexpr (GEN) {
                    SC_UNROLL for (int i = 0; i < 4; ++i)
                        a[2 + i] = w1 * (h1.gx * g.ab[4 * i] + h1.gy * g.ab[4 * i + 1]) + w2 * (h2.gx * g.j2[0][i] + h2.gy * g.j2[1][i]);
                } else {
                    a[2] = w1 * (h1.gx * g.a02 + h1.gy * g.a12) + w2 * (h2.gx * (g.a02 + g.g02) + h2.gy * (g.a12 + g.g12));
                    a[3] = w1 * (h1.gx * g.a03 + h1.gy * g.a13) + w2 * (h2.gx * (g.a03 + g.g03) + h2.gy * (g.a13 + g.g13));
                    a[4] = w2 * dt * (h2.gx * g.g03 + h2.gy * g.g13);
                    a[5] = w2 * dt * (h2.gx * g.g02 + h2.gy * g.g12);
                }
            }
            if (hp) { hp[0] = h0; hp[1] = h1; hp[2] = h2; }
            return w0 * h0.h + w1 * h1.h + w2 * h2.h;
        }
        const double ox = lds[L.OB + 3 * j], oy = lds[L.OB + 3 * j + 1], d = P.radius + lds[L.OB + 3 * j + 2], off = P.beta * d * d;
        const double e0x = xs[0] - ox, e0y = xs[1] - oy, e1x = g.p1[0] - ox, e1y = g.p1[1] - oy, e2x = g.p2[0] - ox, e2y = g.p2[1] - oy;
        if (a) {
            const double dt = P.dt;
            a[0] = 2.0 * (w0 * e0x + w1 * e1x + w2 * e2x); a[1] = 2.0 * (w0 * e0y + w1 * e1y + w2 * e2y);
            if constexpr (GEN) {
                SC_UNROLL for (int i = 0; i < 4; ++i)
                    a[2 + i] = 2.0 * (w1 * (e1x * g.ab[4 * i] + e1y * g.ab[4 * i + 1]) + w2 * (e2x * g.j2[0][i] + e2y * g.j2[1][i]));
                return w0 * (e0x * e0x + e0y * e0y - off) + w1 * (e1x * e1x + e1y * e1y - off) + w2 * (e2x * e2x + e2y * e2y - off);
            }
            a[2] = 2.0 * (w1 * (e1x * g.a02 + e1y * g.a12) + w2 * (e2x * (g.a02 + g.g02) + e2y * (g.a12 + g.g12)));
            a[3] = 2.0 * (w1 * (e1x * g.a03 + e1y * g.a13) + w2 * (e2x * (g.a03 + g.g03) + e2y * (g.a13 + g.g13)));
            a[4] = 2.0 * w2 * dt * (e2x * g.g03 + e2y * g.g13);
            a[5] = 2.0 * w2 * dt * (e2x * g.g02 + e2y * g.g12);
        }
        return w0 * (e0x * e0x + e0y * e0y - off) + w1 * (e1x * e1x + e1y * e1y - off) + w2 * (e2x * e2x + e2y * e2y - off);
    }
    // objective share of lane k: l(x_k) (+ m(x_N)) + R (u_k - u_{k-1})^2; um = u_{k-1}
    SC_HD double cost_share(const double* xs, const double* us, const double* um) const {
        double f = 0.0;
        if (rs) {                                                          // restoration: sum D_R^2 (w - w_R)^2 over my variables (x zeta / 2 in barrier())
            if (acl) SC_UNROLL for (int i = 0; i < NX; ++i) { const double d = xs[i] - lds[XRi(i)]; f += dr2(i) * d * d; }
            if (stl) SC_UNROLL for (int j = 0; j < NU; ++j) { const double d = us[j] - lds[XRi(NX + j)]; f += dr2(NX + j) * d * d; }
            return f;
        }
        if (acl) {
            const double e0 = xs[0] - xg[0], e1 = xs[1] - xg[1];
            f = P.Q[0] * e0 * e0 + P.Q[1] * e1 * e1 + P.Q[2] * xs[2] * xs[2] + P.Q[3] * xs[3] * xs[3];
        }
        if (stl) SC_UNROLL for (int j = 0; j < NU; ++j) { const double d = us[j] - um[j]; f += P.R[j] * d * d; }
        return f;
    }
    // Log-barrier terms of a point, gathered while its rows are walked (the objective / mu-independent part): the log of the product of every
    // slack to a bound (one log per lane: the x / u bounds give six factors of order one, eight row slacks lie in [1e-11, 1e6]; the product is
    // flushed every eight rows, every two inside the restoration where n and p ride along), the sum of the one-sided slacks (damping), n + p
    struct BarAcc {
        double prod, lg, sa, sn;
        int nf;
        bool ok;
    };
    SC_HD void bar_begin(BarAcc& B, const double* xs, const double* us) const {
        B.prod = 1.0; B.lg = 0.0; B.sa = 0.0; B.sn = 0.0; B.ok = true; B.nf = 0;
        if (XB && acl) {
            const double a = xs[3] - xbL, b = xbU - xs[3];
            if (!(a > 0.0) || !(b > 0.0)) B.ok = false;
            B.prod = a * b;
        }
        if (stl) {
            SC_UNROLL for (int j = 0; j < NU; ++j) {
                const double a = us[j] - ubL[j], b = ubU[j] - us[j];
                if (!(a > 0.0) || !(b > 0.0)) B.ok = false;
                B.prod *= a * b;
            }
        }
    }
    // row j: slack to its bound sl = sU - s (and n, p of the restoration at step length a_np)
    SC_HD void bar_row(BarAcc& B, int j, double sl, double a_np) const {
        if (!(sl > 0.0)) B.ok = false;
        B.sa += sl;
        B.prod *= sl;
        if (rs) {
            const double nt = lds[ri(R_N, j)] + a_np * lds[ri(R_DN, j)], pt_ = lds[ri(R_P, j)] + a_np * lds[ri(R_DP, j)];
            if (!(nt > 0.0) || !(pt_ > 0.0)) B.ok = false;
            B.sn += nt + pt_;
            B.prod *= nt * pt_;
            if (++B.nf == 2) { B.lg += log(B.prod); B.prod = 1.0; B.nf = 0; }
        } else if (++B.nf == 8) { B.lg += log(B.prod); B.prod = 1.0; B.nf = 0; }
    }
    // barrier function (scaled objective + log barrier of every bound + damping of the one-sided ones) from the gathered terms
    SC_HD double barrier(double fsum, BarAcc& B, double mu) const {
        if (act) { B.lg += log(B.prod); B.prod = 1.0; }
        double v = -mu * B.lg;
        if (stg) {
            v += O.kappa_d * mu * B.sa;
            if (rs) v += (O.resto_penalty_parameter + O.kappa_d * mu) * B.sn;    // n, p >= 0: linear term rho_R, damping (one-sided)
        }
        const double bad = cx.wmax(B.ok ? 0.0 : 1.0);
        if (bad > 0.0) return INFINITY;
        return (rs ? 0.5 * zeta : df) * fsum + cx.wsum(v);
    }
    // trial evaluation at (xs, us) and slacks s + a_np ds (inside the restoration n, p at a_np as well): theta (l1 residual of the scaled rows),
    // the unscaled objective, the barrier terms; `safe`: IPOPT's safe-slack rule on the rows' slack bounds on the way; pmax (optional): largest residual
    SC_HD void eval0(const double* xs, const double* us, double& theta, double& fsum, BarAcc& B, double mu, double a_np, bool safe, double* pmax = nullptr) {
        sync();
        if (acl) SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.XS + k * 4 + i] = xs[i];
        if (stl) SC_UNROLL for (int j = 0; j < NU; ++j) lds[L.US + (k + 1) * 2 + j] = us[j];
        if (lane == 0) SC_UNROLL for (int j = 0; j < NU; ++j) lds[L.US + j] = uprev[j];
        sync();
        double th = 0.0, pm = 0.0;
        double um[NU];
        SC_UNROLL for (int j = 0; j < NU; ++j) um[j] = lds[L.US + k * 2 + j];
        bar_begin(B, xs, us);
        if (stg) {
            const double s_min = EPS_ * fmin(1.0, mu), move = 1.8189894035458565e-12;     // eps^(3/4)
            Geo g;
            geometry(xs, us, g);
            if (stl) SC_UNROLL for (int i = 0; i < NX; ++i) { const double r = fabs(dgc(i) * (g.F[i] - lds[L.XS + (k + 1) * 4 + i])); th += r; pm = fmax(pm, r); }
            for (int j = q; j < K; j += G) {
                const double sj = lds[ri(R_S, j)] + a_np * lds[ri(R_DS, j)];
                double bj = lds[ri(R_SU, j)];
                if (safe) { safe1(sj, bj, false, s_min, move); lds[ri(R_SU, j)] = bj; }
                double r = -dgd(j) * row(xs, g, j) - sj;
                if (rs) r += (lds[ri(R_N, j)] + a_np * lds[ri(R_DN, j)]) - (lds[ri(R_P, j)] + a_np * lds[ri(R_DP, j)]);
                th += fabs(r); pm = fmax(pm, fabs(r));
                bar_row(B, j, bj - sj, a_np);
            }
        }
        if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) { const double r = fabs(xs[i] - x0[i]); th += r; pm = fmax(pm, r); }
        theta = cx.wsum(th);
        if (pmax) *pmax = cx.wmax(pm);
        fsum = cx.wsum(cost_share(xs, us, um));
    }

    // fraction to the boundary: the largest step in (0, 1] that keeps sl + a dsl >= (1 - tau) sl
    SC_HD double ftb1(double tau, double sl, double dsl) const { return dsl < 0.0 ? fmin(1.0, -tau * sl / dsl) : 1.0; }
    // IPOPT's CalculateSafeSlack: a slack below eps min(1, mu) is raised to eps^(3/4) max(1, |bound|) by moving the bound
    SC_HD void safe1(double v, double& lo, bool lower, double s_min, double move) const {
        if (lower) { if (v - lo < s_min) lo = v - fmax(v - lo, move * fmax(1.0, fabs(lo))); }
        else { if (lo - v < s_min) lo = v + fmax(lo - v, move * fmax(1.0, fabs(lo))); }
    }
    SC_HD void safe_slacks_xu(const double* xs, const double* us, double mu) {      // (the rows' slack bounds: inside eval0's / the update's row loop)
        const double s_min = EPS_ * fmin(1.0, mu), move = 1.8189894035458565e-12;     // eps^(3/4)
        if (XB && act) { safe1(xs[3], xbL, true, s_min, move); safe1(xs[3], xbU, false, s_min, move); }
        if (stg) SC_UNROLL for (int j = 0; j < NU; ++j) { safe1(us[j], ubL[j], true, s_min, move); safe1(us[j], ubU[j], false, s_min, move); }
    }

    struct Eval2 {
        double Jty[NV];          // J' y at my (x_k, u_k)
        double gfx[NX], gfu[NU]; // scaled objective gradient
    };

    // condensed weight E_j and right-hand side b_j of row j (J dx - dy / E = b): the slack -- and in the restoration n and p -- are eliminated,
    //   1 / E = sum_v q_v,  q_v = 1 / (Sigma_v + dw);   b = -r + q_s rhs_s - q_n rhs_n + q_p rhs_p   (rhs_v = -(barrier gradient + sign_v y))
    struct RowW { double E, b, qs, qn, qp, rs_, rn, rp; };
    SC_HD RowW row_weights(int j, double mu, double dw) const {
        RowW w;
        const double stU = lds[ri(R_SU, j)] - lds[ri(R_S, j)];
        w.qs = 1.0 / (lds[ri(R_VU, j)] / stU + dw);
        w.rs_ = lds[ri(R_YD, j)] - (mu / stU - O.kappa_d * mu);
        double rd = lds[ri(R_DV, j)] - lds[ri(R_S, j)], e = w.qs, b = w.qs * w.rs_;
        w.qn = w.qp = w.rn = w.rp = 0.0;
        if (rs) {
            const double n = lds[ri(R_N, j)], p = lds[ri(R_P, j)], rho_R = O.resto_penalty_parameter;
            rd += n - p;
            w.qn = 1.0 / (lds[ri(R_ZN, j)] / n + dw); w.qp = 1.0 / (lds[ri(R_ZP, j)] / p + dw);
            w.rn = -(rho_R - mu / n + O.kappa_d * mu + lds[ri(R_YD, j)]); w.rp = -(rho_R - mu / p + O.kappa_d * mu - lds[ri(R_YD, j)]);
            e += w.qn + w.qp; b += -w.qn * w.rn + w.qp * w.rp;
        }
        w.E = 1.0 / e; w.b = -rd + b;
        return w;
    }

    // exchange through LDS: x_{k+1} (XS), u_{k-1} / u_{k+1} (US, slot k + 1 = u_k, slot 0 = u_prev), multipliers of the rows that DEFINE x_k (YS, slot k)
    SC_HD void publish() {
        sync();
        if (acl) SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.XS + k * 4 + i] = x[i];
        if (stl) {
            SC_UNROLL for (int j = 0; j < NU; ++j) lds[L.US + (k + 1) * 2 + j] = u[j];
            SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.YS + (k + 1) * 4 + i] = dgc(i) * yc[i];
        }
        if (lane == 0) {
            SC_UNROLL for (int j = 0; j < NU; ++j) lds[L.US + j] = uprev[j];
            SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.YS + i] = -lds[L.Y0 + i];
        }
        if (acl && k == N) SC_UNROLL for (int j = 0; j < NU; ++j) lds[L.US + (N + 1) * 2 + j] = 0.0;
        sync();
    }

    // Level 2 at the iterate: residuals, J'y, and -- `build` -- the stage block for the recursion: A, H (condensed rows, bounds, dw), gradient,
    // defects.  ls: the least-square multiplier system (W = 0, Sigma = 1) instead of the Newton system.
    BarAcc Bcur;                 // barrier terms of the iterate (eval2 gathers them)
    struct ErrAcc { double d, p, up, cmin, cmax, ysum, zsum; };
    ErrAcc Ecur;                 // the rows' share of the optimality error at the iterate (eval2 gathers it; errors())
    SC_HD void eval2(const bool build, const bool ls, Eval2& E, double mu, double dw, double& theta, double& fsum) {
        publish();
        bar_begin(Bcur, x, u);
        Ecur.d = 0.0; Ecur.p = 0.0; Ecur.up = 0.0; Ecur.cmin = INFINITY; Ecur.cmax = -INFINITY; Ecur.ysum = 0.0; Ecur.zsum = 0.0;
        double um[NU], un[NU];
        SC_UNROLL for (int j = 0; j < NU; ++j) { um[j] = lds[L.US + k * 2 + j]; un[j] = lds[L.US + (k + 2 <= N + 1 ? k + 2 : N + 1) * 2 + j]; }
        const bool last = k == N - 1;
        const double dt = P.dt;
        double th = 0.0;
        SC_UNROLL for (int i = 0; i < NX; ++i) E.gfx[i] = 0.0;
        if (act) {
            if (rs) SC_UNROLL for (int i = 0; i < NX; ++i) E.gfx[i] = zeta * dr2(i) * (x[i] - lds[XRi(i)]);
            else {
                E.gfx[0] = 2.0 * df * P.Q[0] * (x[0] - xg[0]); E.gfx[1] = 2.0 * df * P.Q[1] * (x[1] - xg[1]);
                E.gfx[2] = 2.0 * df * P.Q[2] * x[2]; E.gfx[3] = 2.0 * df * P.Q[3] * x[3];
            }
        }
        SC_UNROLL for (int j = 0; j < NU; ++j) {
            double gj = 0.0;
            if (stg) {
                if (rs) gj = zeta * dr2(NX + j) * (u[j] - lds[XRi(NX + j)]);
                else { gj = 2.0 * df * P.R[j] * (u[j] - um[j]); if (!last) gj -= 2.0 * df * P.R[j] * (un[j] - u[j]); }
            }
            E.gfu[j] = gj;
        }
        SC_UNROLL for (int i = 0; i < NV; ++i) E.Jty[i] = 0.0;
        // rows that define x_k: -I (scaled) on x_k from the dynamics of stage k - 1; +I from the initial-state rows on x_0
        if (act) SC_UNROLL for (int i = 0; i < NX; ++i) E.Jty[i] = -lds[L.YS + k * 4 + i];
        // diagonal of the block (objective, bound terms, dw) and the bound terms of the gradient
        double dg_[NV], gb[NV];
        SC_UNROLL for (int i = 0; i < NV; ++i) { dg_[i] = 0.0; gb[i] = 0.0; }
        if (build) {
            if (ls) {
                SC_UNROLL for (int i = 0; i < NV; ++i) dg_[i] = 1.0;
                gb[3] = -zxL + zxU;
                SC_UNROLL for (int j = 0; j < NU; ++j) gb[4 + j] = -zuL[j] + zuU[j];
            } else {
                SC_UNROLL for (int i = 0; i < NX; ++i) dg_[i] = ((rs && act) ? zeta * dr2(i) : 2.0 * df * P.Q[i]) + dw;
                if constexpr (XB) {
                    const double a = x[3] - xbL, b = xbU - x[3];
                    dg_[3] += zxL / a + zxU / b;
                    gb[3] = -mu / a + mu / b;
                }
                SC_UNROLL for (int j = 0; j < NU; ++j) {
                    const double a_ = u[j] - ubL[j], b_ = ubU[j] - u[j];
                    dg_[4 + j] = ((rs && stg) ? zeta * dr2(NX + j) : 2.0 * df * P.R[j] * (last ? 1.0 : 2.0)) + dw + zuL[j] / a_ + zuU[j] / b_;
                    gb[4 + j] = -mu / a_ + mu / b_;
                }
            }
        }
        // my share of the stage's rows (lane q of the group walks rows q, q + G, ..); the sums the stage needs as a whole -- J'y, and for the
        // recursion the condensed block and gradient, the multiplier sums of the curvature terms -- are added over the group afterwards
        // jr (6) | M (21) | gv (6) | sl, socx, socy  -- SE: per barrier point the om-weighted gradient (2) and Hessian (3) of the rows instead
        constexpr int NACC = SE ? 48 : 36;
        double acc[NACC];
        SC_UNROLL for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
        double* const jr = acc; double* const M = acc + 6; double* const gv = acc + 27;
        Geo g;
        double wy[NX] = {0.0, 0.0, 0.0, 0.0};
        if (stg) {
            geometry(x, u, g);
            tc_ = g.c; ts_ = g.s; tc1_ = g.c1; ts1_ = g.s1;
            SC_UNROLL for (int i = 0; i < NX; ++i) { rc[i] = dgc(i) * (g.F[i] - lds[L.XS + (k + 1) * 4 + i]); if (stl) th += fabs(rc[i]); }
            // J' y of my dynamics rows: [A | B]' (dgc yc)
            SC_UNROLL for (int i = 0; i < NX; ++i) wy[i] = dgc(i) * yc[i];
            E.Jty[0] += wy[0]; E.Jty[1] += wy[1];
            if constexpr (GEN) {
                if (build && stl) SC_UNROLL for (int i = 0; i < 16; ++i) lds[L.AB + k * 16 + i] = g.ab[i];
                SC_UNROLL for (int i = 0; i < 4; ++i) E.Jty[2 + i] += g.ab[4 * i] * wy[0] + g.ab[4 * i + 1] * wy[1] + g.ab[4 * i + 2] * wy[2] + g.ab[4 * i + 3] * wy[3];
            } else {
                if (build && stl) { lds[L.AB + k * 4 + 0] = g.a02; lds[L.AB + k * 4 + 1] = g.a03; lds[L.AB + k * 4 + 2] = g.a12; lds[L.AB + k * 4 + 3] = g.a13; }
                E.Jty[2] += g.a02 * wy[0] + g.a12 * wy[1] + wy[2];
                E.Jty[3] += g.a03 * wy[0] + g.a13 * wy[1] + wy[3];
                E.Jty[4] += dt * wy[3]; E.Jty[5] += dt * wy[2];
            }
            for (int j = q; j < K; j += G) {
                double a[NV];
                HP hp[3];
                const double cv = row(x, g, j, a, SE ? hp : nullptr);
                const double sc = dgd(j);
                const double dvj = -sc * cv, sj = lds[ri(R_S, j)], stU = lds[ri(R_SU, j)] - sj, ydj = lds[ri(R_YD, j)], vUj = lds[ri(R_VU, j)];
                lds[ri(R_DV, j)] = dvj;
                const double rd = dvj - sj + (rs ? lds[ri(R_N, j)] - lds[ri(R_P, j)] : 0.0);
                th += fabs(rd);
                bar_row(Bcur, j, stU, 0.0);
                {
                    const double cp = stU * vUj;
                    Ecur.d = fmax(Ecur.d, fabs(-ydj + vUj));
                    Ecur.p = fmax(Ecur.p, fabs(rd)); Ecur.up = fmax(Ecur.up, fabs(rd / sc));
                    Ecur.cmin = fmin(Ecur.cmin, cp); Ecur.cmax = fmax(Ecur.cmax, cp);
                    Ecur.ysum += fabs(ydj); Ecur.zsum += fabs(vUj);
                    if (rs) {
                        const double n = lds[ri(R_N, j)], pp = lds[ri(R_P, j)], zn = lds[ri(R_ZN, j)], zp = lds[ri(R_ZP, j)], rho_R = O.resto_penalty_parameter;
                        Ecur.d = fmax(Ecur.d, fmax(fabs(rho_R + ydj - zn), fabs(rho_R - ydj - zp)));
                        Ecur.cmin = fmin(Ecur.cmin, fmin(n * zn, pp * zp)); Ecur.cmax = fmax(Ecur.cmax, fmax(n * zn, pp * zp));
                        Ecur.zsum += fabs(zn) + fabs(zp);
                    }
                }
                const double om = sc * ydj;                                         // weight of grad^2 (-cbf_j) in the Hessian of the Lagrangian
                SC_UNROLL for (int i = 0; i < NV; ++i) jr[i] += om * a[i];             // yd_j * grad d_j = -om grad cbf_j
                if constexpr (SE) {
                    SC_UNROLL for (int i = 0; i < 3; ++i) {
                        acc[33 + 5 * i] += om * hp[i].gx; acc[34 + 5 * i] += om * hp[i].gy;
                        acc[35 + 5 * i] += om * hp[i].hxx; acc[36 + 5 * i] += om * hp[i].hxy; acc[37 + 5 * i] += om * hp[i].hyy;
                    }
                } else { acc[33] += om; acc[34] += om * lds[L.OB + 3 * j]; acc[35] += om * lds[L.OB + 3 * j + 1]; }
                if (build) {
                    double Ej, bd;
                    if (ls) { Ej = 1.0; bd = -vUj; }                                 // q = 1, rhs_t = -(0 + vU), rhs_g = 0: b = q rhs_t
                    else if (rs) { const RowW w = row_weights(j, mu, dw); Ej = w.E; bd = w.b; }
                    else {
                        Ej = vUj / stU + dw;
                        const double gt = mu / stU - O.kappa_d * mu;
                        bd = -rd + (ydj - gt) / Ej;
                    }
                    // row gradient = -sc a:  H += E (sc a)(sc a)',  g += E b sc a
                    const double ea = Ej * sc * sc, eb = Ej * bd * sc;
                    int e = 0;
                    SC_UNROLL for (int p = 0; p < NV; ++p) {
                        gv[p] += eb * a[p];
                        SC_UNROLL for (int r_ = p; r_ < NV; ++r_, ++e) M[e] += ea * a[p] * a[r_];
                    }
                }
            }
        } else {
            SC_UNROLL for (int i = 0; i < NX; ++i) rc[i] = 0.0;
        }
        if (build) cx.template gsum<NACC>(acc, G); else cx.template gsum<6>(acc, G);
        SC_UNROLL for (int i = 0; i < NV; ++i) E.Jty[i] -= jr[i];
        if (build && stl) {
            const double sl = SE ? 0.0 : acc[33], socx = SE ? 0.0 : acc[34], socy = SE ? 0.0 : acc[35];
            if (!ls) {
                // curvature of the Lagrangian: dynamics rows (weights dgc y on F_0, F_1, which are p1) and the rows' -cbf_j (weights om_j):
                //   -2 sl sum_p w_p Jp' Jp  -  2 sum_p w_p (S_p,x grad^2 p_p,x + S_p,y grad^2 p_p,y),   S_p = sum_j om_j (p_p - c_j) = sl p_p - soc
                // (SE: S_p = half the om-weighted gradient sum of the rows at point p; the constant 2 I of a circle becomes the rows' Hessian sum)
                const double s1x = SE ? 0.5 * acc[38] : sl * g.p1[0] - socx, s1y = SE ? 0.5 * acc[39] : sl * g.p1[1] - socy;
                const double s2x = SE ? 0.5 * acc[43] : sl * g.p2[0] - socx, s2y = SE ? 0.5 * acc[44] : sl * g.p2[1] - socy;
                const double nx_ = wy[0] - 2.0 * w1 * s1x - 2.0 * w2 * s2x, ny_ = wy[1] - 2.0 * w1 * s1y - 2.0 * w2 * s2y;    // on grad^2 p1
                const double kx = -2.0 * w2 * s2x, ky = -2.0 * w2 * s2y;                                                        // on grad^2 (p2 - p1)
                if constexpr (MODEL == M_SI) {
                    (void)nx_; (void)ny_; (void)kx; (void)ky;                                   // affine points: no curvature
                } else if constexpr (MODEL == M_UNI) {
                    // p1 = p + dt v (c, s) over (theta = entry 2, v = entry 4); no second point
                    (void)kx; (void)ky;
                    const double v = u[0];
                    M[sym6(2, 2)] += nx_ * (-dt * v * g.c) + ny_ * (-dt * v * g.s);
                    M[sym6(2, 4)] += nx_ * (-dt * g.s) + ny_ * (dt * g.c);
                } else if constexpr (MODEL == M_KB) {
                    // over (theta, v, a, beta) = entries 2 .. 5.  phi = v (c - b s), psi = v (s + b c):  phi_tt = -phi, phi_tv = -(s + b c), phi_tb = -v c,
                    // phi_vb = -s;  psi_tt = -psi, psi_tv = c - b s, psi_tb = -v s, psi_vb = c;  F_2 = theta + dt v b / L_r: (v, b) entry dt / L_r
                    const double v = x[3], be = u[1], iL = P.inv_Lr, c = g.c, s_ = g.s, c1 = g.c1, s1 = g.s1, v1 = g.v1, chi = g.cw[0];
                    const double ph = g.cw[1], ps = g.cw[2], ph1 = g.cw[3], ps1 = g.cw[4];
                    M[sym6(2, 2)] += dt * (nx_ * (-ph) + ny_ * (-ps));
                    M[sym6(2, 3)] += dt * (nx_ * (-(s_ + be * c)) + ny_ * (c - be * s_));
                    M[sym6(2, 5)] += dt * (nx_ * (-v * c) + ny_ * (-v * s_));
                    M[sym6(3, 5)] += dt * (nx_ * (-s_) + ny_ * c) + wy[2] * dt * iL;
                    // (p2 - p1) = dt (phi, psi)(q), q = (theta_1, v_1, beta):  J_q' S J_q + (kx phi_t + ky psi_t)(q) grad^2 theta_1
                    const double S00 = kx * (-ph1) + ky * (-ps1), S01 = kx * (-(s1 + be * c1)) + ky * (c1 - be * s1), S02 = kx * (-v1 * c1) + ky * (-v1 * s1), S12 = kx * (-s1) + ky * c1;
                    const double qt[4] = {1.0, dt * be * iL, 0.0, dt * v * iL}, qv[4] = {0.0, chi, chi * dt, 0.0}, qb[4] = {0.0, 0.0, 0.0, 1.0};
                    const double gt = kx * (-ps1) + ky * ph1;
                    SC_UNROLL for (int i = 0; i < 4; ++i) {
                        // (S J_q)[:, i]
                        const double t0 = S00 * qt[i] + S01 * qv[i] + S02 * qb[i], t1 = S01 * qt[i] + S12 * qb[i], t2 = S02 * qt[i] + S12 * qv[i];
                        SC_UNROLL for (int j = i; j < 4; ++j) {
                            double hv = qt[j] * t0 + qv[j] * t1 + qb[j] * t2;
                            if (i == 1 && j == 3) hv += gt * dt * iL;
                            M[sym6(2 + i, 2 + j)] += dt * hv;
                        }
                    }
                } else if constexpr (MODEL == M_DI) {
                    // p1 is affine; p2 - p1 = dt wt(w), wt = w v_max / |w| where |w| > v_max:  grad^2 wt_d = v_max (-(d_da w_b + d_db w_a + d_ab w_d) / |w|^3
                    // + 3 w_d w_a w_b / |w|^5) over w = (F_2, F_3), F_2 = x_2 + dt u_1, F_3 = x_3 + dt u_0
                    (void)nx_; (void)ny_;
                    const double wa = g.F[2], wb = g.F[3], vm = g.v1;
                    if (vm > P.v_max) {
                        const double i3 = P.v_max / (vm * vm * vm), i5 = 3.0 * i3 / (vm * vm);
                        const double hx00 = -3.0 * wa * i3 + wa * wa * wa * i5, hx01 = -wb * i3 + wa * wa * wb * i5, hx11 = -wa * i3 + wa * wb * wb * i5;
                        const double hy00 = -wb * i3 + wb * wa * wa * i5, hy01 = -wa * i3 + wb * wa * wb * i5, hy11 = -3.0 * wb * i3 + wb * wb * wb * i5;
                        const double s00 = dt * (kx * hx00 + ky * hy00), s01 = dt * (kx * hx01 + ky * hy01), s11 = dt * (kx * hx11 + ky * hy11), dt2 = dt * dt;
                        M[sym6(2, 2)] += s00; M[sym6(2, 5)] += dt * s00; M[sym6(5, 5)] += dt2 * s00;
                        M[sym6(3, 3)] += s11; M[sym6(3, 4)] += dt * s11; M[sym6(4, 4)] += dt2 * s11;
                        M[sym6(2, 3)] += s01; M[sym6(2, 4)] += dt * s01; M[sym6(3, 5)] += dt * s01; M[sym6(4, 5)] += dt2 * s01;
                    }
                } else {
                const double v = x[3], v1 = g.v1, c = g.c, s_ = g.s, c1 = g.c1, s1 = g.s1, dt2 = dt * dt, dt3 = dt2 * dt;
                M[sym6(2, 2)] += nx_ * (-dt * v * c) + ny_ * (-dt * v * s_) + kx * (-dt * v1 * c1) + ky * (-dt * v1 * s1);
                M[sym6(2, 3)] += nx_ * (-dt * s_) + ny_ * (dt * c) + kx * (-dt * s1) + ky * (dt * c1);
                M[sym6(2, 4)] += kx * (-dt2 * s1) + ky * (dt2 * c1);
                M[sym6(2, 5)] += kx * (-dt2 * v1 * c1) + ky * (-dt2 * v1 * s1);
                M[sym6(3, 5)] += kx * (-dt2 * s1) + ky * (dt2 * c1);
                M[sym6(4, 5)] += kx * (-dt3 * s1) + ky * (dt3 * c1);
                M[sym6(5, 5)] += kx * (-dt3 * v1 * c1) + ky * (-dt3 * v1 * s1);
                }
                // -2 sl (w0 J0'J0 + w1 J1'J1 + w2 J2'J2): J0 = [e_0; e_1], J1 = J0 + [a0.; a1.] on (theta, v), J2 = J1 + [g..] on (theta, v, a, omega)
                const double o0 = -2.0 * sl * w0, o1 = -2.0 * sl * w1, o2 = -2.0 * sl * w2;
                double j1x[NV] = {1.0, 0.0, g.a02, g.a03, 0.0, 0.0}, j1y[NV] = {0.0, 1.0, g.a12, g.a13, 0.0, 0.0};
                double j2x[NV] = {1.0, 0.0, g.a02 + g.g02, g.a03 + g.g03, dt * g.g03, dt * g.g02};
                double j2y[NV] = {0.0, 1.0, g.a12 + g.g12, g.a13 + g.g13, dt * g.g13, dt * g.g12};
                if constexpr (GEN) SC_UNROLL for (int i = 0; i < 4; ++i) { j1x[2 + i] = g.ab[4 * i]; j1y[2 + i] = g.ab[4 * i + 1]; j2x[2 + i] = g.j2[0][i]; j2y[2 + i] = g.j2[1][i]; }
                if constexpr (SE) {
                    // - sum_p w_p J_p' (sum_j om_j grad^2 h_j(p)) J_p
                    M[sym6(0, 0)] -= w0 * acc[35]; M[sym6(0, 1)] -= w0 * acc[36]; M[sym6(1, 1)] -= w0 * acc[37];
                    int e = 0;
                    SC_UNROLL for (int p = 0; p < NV; ++p)
                        SC_UNROLL for (int r_ = p; r_ < NV; ++r_, ++e)
                            M[e] -= w1 * (acc[40] * j1x[p] * j1x[r_] + acc[41] * (j1x[p] * j1y[r_] + j1y[p] * j1x[r_]) + acc[42] * j1y[p] * j1y[r_])
                                  + w2 * (acc[45] * j2x[p] * j2x[r_] + acc[46] * (j2x[p] * j2y[r_] + j2y[p] * j2x[r_]) + acc[47] * j2y[p] * j2y[r_]);
                } else {
                M[sym6(0, 0)] += o0; M[sym6(1, 1)] += o0;
                int e = 0;
                SC_UNROLL for (int p = 0; p < NV; ++p)
                    SC_UNROLL for (int r_ = p; r_ < NV; ++r_, ++e) M[e] += o1 * (j1x[p] * j1x[r_] + j1y[p] * j1y[r_]) + o2 * (j2x[p] * j2x[r_] + j2y[p] * j2y[r_]);
                }
            }
            int e = 0;
            SC_UNROLL for (int p = 0; p < NV; ++p) {
                SC_UNROLL for (int r_ = p; r_ < NV; ++r_, ++e) lds[L.H + k * 21 + e] = M[e] + (p == r_ ? dg_[p] : 0.0);
                lds[L.G + k * 6 + p] = gb[p] + gv[p] + (p < NX ? E.gfx[p] : E.gfu[p - NX]) + (ls ? 0.0 : E.Jty[p]);
            }
            // defects (unscaled) of my dynamics rows -> C[k + 1] = rc / dgc (ls: 0)
            SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.C + (k + 1) * 4 + i] = ls ? 0.0 : rc[i] / dgc(i);
        }
        if (build && acl && k == N) {                                       // terminal state: diagonal block, gradient
            SC_UNROLL for (int a = 0; a < NX; ++a) {
                SC_UNROLL for (int b = a; b < NX; ++b) lds[L.H + N * 21 + sym6(a, b)] = a == b ? dg_[a] : 0.0;
                lds[L.G + N * 6 + a] = gb[a] + E.gfx[a] + (ls ? 0.0 : E.Jty[a]);
            }
        }
        if (lane == 0) {
            SC_UNROLL for (int i = 0; i < NX; ++i) {
                const double r0 = x[i] - x0[i];
                th += fabs(r0);
                if (build) lds[L.C + i] = ls ? 0.0 : -r0;                  // C[0] = dx_0 = b_0 = -r0
            }
        }
        theta = cx.wsum(th);
        fsum = cx.wsum(cost_share(x, u, um));
        if (build) sync();
    }

    // after the recursion: the multiplier steps of the dynamics rows (from the value function), of my rows and bounds
    double sl_ap = 1.0, sl_az = 1.0, sl_v = 0.0;       // the rows' share of step_lengths (finish_step gathers it)
    SC_HD void finish_step(bool ls, double mu, double dw, double tau) {
        {
            double xi[NA];
            riccati_forward<Cx, GEN>(cx, L, N, P.dt, act ? k : -1, xi, du);
            SC_UNROLL for (int i = 0; i < NX; ++i) dx[i] = act ? xi[i] : 0.0;
            SC_UNROLL for (int j = 0; j < NU; ++j) { dvv[j] = act ? xi[4 + j] : 0.0; if (!stg) du[j] = 0.0; }
        }
        // lam_k = (P_k xi_k + p_k)_x with the x rows of P_k where riccati_backward left them; the terminal stage: lam_N = H_N dx_N + g_N
        sync();
        if (acl) {
            if (k < N) {
                const typename Cx::ptr px = lds + L.PX + k * 28;
                SC_UNROLL for (int a = 0; a < NX; ++a) {
                    double v = px[24 + a];
                    SC_UNROLL for (int b = 0; b < NX; ++b) v += px[a * 6 + b] * dx[b];
                    SC_UNROLL for (int j = 0; j < NU; ++j) v += px[a * 6 + 4 + j] * dvv[j];
                    lds[L.LAM + k * 4 + a] = v;
                }
            } else {
                SC_UNROLL for (int a = 0; a < NX; ++a) {
                    double v = lds[L.G + N * 6 + a];
                    SC_UNROLL for (int b = 0; b < NX; ++b) v += lds[L.H + N * 21 + sym6(a, b)] * dx[b];
                    lds[L.LAM + N * 4 + a] = v;
                }
            }
        }
        sync();
        // rows: dy_d = E (a . dw - b), ds = q (rhs_t + dy_d), dvU -- and, on the way, the rows' share of the fraction-to-the-boundary step lengths
        // and of the barrier function's directional derivative (step_lengths)
        sl_ap = 1.0; sl_az = 1.0; sl_v = 0.0;
        if (stg) {
            Geo g;
            geometry(x, u, g, true);
            const double dz[NV] = {dx[0], dx[1], dx[2], dx[3], du[0], du[1]};
            for (int j = q; j < K; j += G) {
                double a[NV];
                row(x, g, j, a);
                double adw = 0.0;
                SC_UNROLL for (int i = 0; i < NV; ++i) adw += a[i] * dz[i];
                adw *= -dgd(j);
                const double vUj = lds[ri(R_VU, j)];
                if (ls) { lds[ri(R_DYD, j)] = adw + vUj; lds[ri(R_DS, j)] = 0.0; lds[ri(R_DVU, j)] = 0.0; continue; }
                const double stU = lds[ri(R_SU, j)] - lds[ri(R_S, j)], istU = 1.0 / stU;
                double dyd_, ds_;
                if (rs) {                                                        // dy = E (a . dw - b);  dv = q_v (rhs_v - sign_v dy);  multipliers of n, p >= 0
                    const RowW w = row_weights(j, mu, dw);
                    const double n = lds[ri(R_N, j)], p = lds[ri(R_P, j)], zn = lds[ri(R_ZN, j)], zp = lds[ri(R_ZP, j)], rho_R = O.resto_penalty_parameter;
                    dyd_ = w.E * (adw - w.b);
                    ds_ = w.qs * (w.rs_ + dyd_);
                    const double dn = w.qn * (w.rn - dyd_), dp = w.qp * (w.rp + dyd_);
                    const double dzn = mu / n - zn - zn * dn / n, dzp = mu / p - zp - zp * dp / p;
                    lds[ri(R_DN, j)] = dn; lds[ri(R_DP, j)] = dp; lds[ri(R_DZN, j)] = dzn; lds[ri(R_DZP, j)] = dzp;
                    sl_ap = fmin(sl_ap, fmin(ftb1(tau, n, dn), ftb1(tau, p, dp)));
                    sl_az = fmin(sl_az, fmin(ftb1(tau, zn, dzn), ftb1(tau, zp, dzp)));
                    sl_v += (rho_R - mu / n + O.kappa_d * mu) * dn + (rho_R - mu / p + O.kappa_d * mu) * dp;
                } else {
                    const double Ej = vUj * istU + dw, gt = mu * istU - O.kappa_d * mu;
                    const double rd = lds[ri(R_DV, j)] - lds[ri(R_S, j)], rhs_t = lds[ri(R_YD, j)] - gt, iE = 1.0 / Ej, bd = -rd + rhs_t * iE;
                    dyd_ = Ej * (adw - bd);
                    ds_ = (rhs_t + dyd_) * iE;
                }
                const double dvU_ = mu * istU - vUj + vUj * ds_ * istU;
                lds[ri(R_DYD, j)] = dyd_; lds[ri(R_DS, j)] = ds_; lds[ri(R_DVU, j)] = dvU_;
                sl_ap = fmin(sl_ap, ftb1(tau, stU, -ds_)); sl_az = fmin(sl_az, ftb1(tau, vUj, dvU_));
                sl_v += (mu * istU - O.kappa_d * mu) * ds_;
            }
        }
        if (!ls) {
            if (XB && act) {
                const double a = x[3] - xbL, b = xbU - x[3];
                dzxL = mu / a - zxL - zxL * dx[3] / a; dzxU = mu / b - zxU + zxU * dx[3] / b;
            } else { dzxL = 0.0; dzxU = 0.0; }
            SC_UNROLL for (int j = 0; j < NU; ++j) {
                if (stg) {
                    const double a = u[j] - ubL[j], b = ubU[j] - u[j];
                    dzuL[j] = mu / a - zuL[j] - zuL[j] * du[j] / a; dzuU[j] = mu / b - zuU[j] + zuU[j] * du[j] / b;
                } else { dzuL[j] = 0.0; dzuU[j] = 0.0; }
            }
        }
    }

    // ---- optimality error (eq. (5)): E_mu and its parts ---------------------------------------------------------------------------------
    // The rows' share of the optimality error is gathered by eval2's row loop (ErrAcc): dual infeasibility of the slacks, residuals, multiplier
    // sums, and the smallest / largest complementarity product -- max |s z - mu| over a set is max(|max - mu|, |min - mu|), so the error for ANY
    // barrier parameter comes without another pass over the rows.  errors() adds the x / u share and the wave reductions.
    SC_HD void errors(const Eval2& E, double mu, double& E0, double& Emu, double& dinf, double& pinf, double& comp0, double& un_pinf, double* sc_out = nullptr) const {
        double d = Ecur.d, p = Ecur.p, up = Ecur.up, cmin = Ecur.cmin, cmax = Ecur.cmax, ysum = Ecur.ysum, zsum = Ecur.zsum;
        if (acl) {
            double gl[NX];
            SC_UNROLL for (int i = 0; i < NX; ++i) gl[i] = E.gfx[i] + E.Jty[i];
            if constexpr (XB) gl[3] += -zxL + zxU;
            SC_UNROLL for (int i = 0; i < NX; ++i) d = fmax(d, fabs(gl[i]));
            if constexpr (XB) {
                const double c1 = (x[3] - xbL) * zxL, c2 = (xbU - x[3]) * zxU;
                cmin = fmin(cmin, fmin(c1, c2)); cmax = fmax(cmax, fmax(c1, c2));
                zsum += fabs(zxL) + fabs(zxU);
            }
        }
        if (stl) {
            SC_UNROLL for (int j = 0; j < NU; ++j) {
                d = fmax(d, fabs(E.gfu[j] + E.Jty[4 + j] - zuL[j] + zuU[j]));
                const double c1 = (u[j] - ubL[j]) * zuL[j], c2 = (ubU[j] - u[j]) * zuU[j];
                cmin = fmin(cmin, fmin(c1, c2)); cmax = fmax(cmax, fmax(c1, c2));
                zsum += fabs(zuL[j]) + fabs(zuU[j]);
            }
            SC_UNROLL for (int i = 0; i < NX; ++i) { p = fmax(p, fabs(rc[i])); up = fmax(up, fabs(rc[i] / dgc(i))); ysum += fabs(yc[i]); }
        }
        if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) { const double r0 = fabs(x[i] - x0[i]); p = fmax(p, r0); up = fmax(up, r0); ysum += fabs(lds[L.Y0 + i]); }
        const bool any = act && cmax >= cmin;                               // (a lane without a product of its own: nothing to report)
        const double c0 = any ? fmax(fabs(cmax), fabs(cmin)) : 0.0, cm = any ? fmax(fabs(cmax - mu), fabs(cmin - mu)) : 0.0;
        dinf = cx.wmax(d); pinf = cx.wmax(p); comp0 = cx.wmax(c0); un_pinf = cx.wmax(up);
        const double compm = cx.wmax(cm);
        ysum = cx.wsum(ysum); zsum = cx.wsum(zsum);
        const double m = (double)(NXT * (N + 1) + N * K), nb = (double)((XB ? 2 * (N + 1) : 0) + 4 * N + N * K * (rs ? 3 : 1));
        const double sd = fmax(O.s_max, (ysum + zsum) / (m + nb)) / O.s_max, sc = fmax(O.s_max, zsum / nb) / O.s_max;
        E0 = fmax(fmax(dinf / sd, pinf), comp0 / sc);
        Emu = fmax(fmax(dinf / sd, pinf), compm / sc);
        if (sc_out) *sc_out = sc;
    }

    // fraction to the boundary over my primal / dual variables; directional derivative of the barrier function along the step
    SC_HD void step_lengths(const Eval2& E, double tau, double mu, double& a_max, double& a_z, double& gBD) const {
        double ap = 1.0, az = 1.0, v = 0.0;
        if (acl) {
            SC_UNROLL for (int i = 0; i < NX; ++i) v += E.gfx[i] * dx[i];
            if constexpr (XB) {
                const double a = x[3] - xbL, b = xbU - x[3];
                ap = fmin(ap, fmin(ftb1(tau, a, dx[3]), ftb1(tau, b, -dx[3])));
                az = fmin(az, fmin(ftb1(tau, zxL, dzxL), ftb1(tau, zxU, dzxU)));
                v += (-mu / a + mu / b) * dx[3];
            }
        }
        if (stg) { ap = fmin(ap, sl_ap); az = fmin(az, sl_az); v += sl_v; }      // (my rows: gathered by finish_step)
        if (stl) {
            SC_UNROLL for (int j = 0; j < NU; ++j) {
                ap = fmin(ap, ftb1(tau, u[j] - ubL[j], du[j])); ap = fmin(ap, ftb1(tau, ubU[j] - u[j], -du[j]));
                az = fmin(az, ftb1(tau, zuL[j], dzuL[j])); az = fmin(az, ftb1(tau, zuU[j], dzuU[j]));
                v += (E.gfu[j] - mu / (u[j] - ubL[j]) + mu / (ubU[j] - u[j])) * du[j];
            }
        }
        a_max = cx.wmin(ap); a_z = cx.wmin(az); gBD = cx.wsum(v);
    }

    // ---- filter ---------------------------------------------------------------------------------------------------------------------
    SC_HD bool filter_ok(double phi, double th) const {
        for (int i = 0; i < nfilt; ++i) {
            const double p = lds[fpo + i], t = lds[fto + i];
            if (!(cmp_le(phi, p, p) || cmp_le(th, t, t))) return false;
        }
        return true;
    }
    SC_HD void filter_add(double phi, double th) {
        sync();
        int n = 0;
        // (dominated entries are dropped; wave-uniform, every lane walks the list, lane 0 writes)
        for (int i = 0; i < nfilt; ++i) {
            const double p = lds[fpo + i], t = lds[fto + i];
            const bool keep = !(p >= phi && t >= th);
            sync();
            if (keep) { if (lane == 0) { lds[fpo + n] = p; lds[fto + n] = t; } ++n; }
            sync();
        }
        if (n >= NFILT) { n = NFILT - 1; filt_over = true; }              // (IPOPT's filter is unbounded: the solve ends 'inaccurate' instead of going on without an entry)
        if (lane == 0) { lds[fpo + n] = phi; lds[fto + n] = th; }
        nfilt = n + 1;
        sync();
    }

    SC_HD void push1(double& v, double lo, double hi, bool fl, bool fu) const {
        const double k1 = O.bound_push, k2 = O.bound_frac;
        const double rng = (fl && fu) ? hi - lo : INFINITY;
        if (fl) v = fmax(v, lo + fmin(k1 * fmax(1.0, fabs(lo)), k2 * rng));
        if (fu) v = fmin(v, hi - fmin(k1 * fmax(1.0, fabs(hi)), k2 * rng));
    }

    // ---- the solve ------------------------------------------------------------------------------------------------------------------
    SC_HD SC_DUMS_INLINE void solve(int& status_out, int& iters_out, double* trace) {
        const double rl = O.bound_relax_factor;
        // ---- start: x_k = x0, u_k = u_prev (set_initial_guess); scaling at that point; bounds relaxed; push ----
        SC_UNROLL for (int i = 0; i < NX; ++i) { x[i] = x0[i]; yc[i] = 0.0; dx[i] = 0.0; }
        SC_UNROLL for (int j = 0; j < NU; ++j) { u[j] = uprev[j]; du[j] = 0.0; dvv[j] = 0.0; }
        if (stg) for (int j = q; j < K; j += G) { lds[ri(R_S, j)] = 0.0; lds[ri(R_YD, j)] = 0.0; lds[ri(R_VU, j)] = 1.0; lds[ri(R_SU, j)] = rl; lds[ri(R_DS, j)] = 0.0; lds[ri(R_DYD, j)] = 0.0; lds[ri(R_DVU, j)] = 0.0; lds[ri(R_DV, j)] = 0.0; }
        xbL = -P.v_max - rl * fmax(1.0, fabs(P.v_max)); xbU = P.v_max + rl * fmax(1.0, fabs(P.v_max)); zxL = XB ? 1.0 : 0.0; zxU = zxL; dzxL = dzxU = 0.0;
        SC_UNROLL for (int j = 0; j < NU; ++j) {
            ubL[j] = P.u_lo[j] - rl * fmax(1.0, fabs(P.u_lo[j])); ubU[j] = P.u_hi[j] + rl * fmax(1.0, fabs(P.u_hi[j]));
            zuL[j] = 1.0; zuU[j] = 1.0; dzuL[j] = dzuU[j] = 0.0;
        }
        // gradient-based scaling at the user's starting point (every stage is the same point there): df, and the row scales into LDS
        {
            double gm = fmax(2.0 * P.Q[0] * fabs(x0[0] - xg[0]), 2.0 * P.Q[1] * fabs(x0[1] - xg[1]));
            gm = fmax(gm, fmax(2.0 * P.Q[2] * fabs(x0[2]), 2.0 * P.Q[3] * fabs(x0[3])));
            const double gmax = O.nlp_scaling_max_gradient, gmin = O.nlp_scaling_min_value;
            df = gm > gmax ? fmax(gmin, gmax / gm) : 1.0;
            Geo g;
            geometry(x0, uprev, g);
            double rm0 = fmax(1.0, fmax(fabs(g.a02), fabs(g.a03))), rm1 = fmax(1.0, fmax(fabs(g.a12), fabs(g.a13))), rm2 = fmax(1.0, P.dt), rm3 = rm2;
            if constexpr (GEN) {
                double rm[4];
                SC_UNROLL for (int r = 0; r < 4; ++r) rm[r] = fmax(fmax(1.0, fabs(g.ab[r])), fmax(fabs(g.ab[4 + r]), fmax(fabs(g.ab[8 + r]), fabs(g.ab[12 + r]))));
                rm0 = rm[0]; rm1 = rm[1]; rm2 = rm[2]; rm3 = rm[3];
            }
            sync();
            if (lane == 0) {
                lds[L.SC + 0] = rm0 > gmax ? fmax(gmin, gmax / rm0) : 1.0; lds[L.SC + 1] = rm1 > gmax ? fmax(gmin, gmax / rm1) : 1.0;
                lds[L.SC + 2] = rm2 > gmax ? fmax(gmin, gmax / rm2) : 1.0; lds[L.SC + 3] = rm3 > gmax ? fmax(gmin, gmax / rm3) : 1.0;
                SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = 0.0;
            }
            for (int j = 0; j < K; ++j) {
                double sc = 1.0;
                if (j < K) {
                    double a[NV];
                    row(x0, g, j, a);
                    double rm = 0.0;
                    SC_UNROLL for (int i = 0; i < NV; ++i) rm = fmax(rm, fabs(a[i]));
                    sc = rm > gmax ? fmax(gmin, gmax / rm) : 1.0;
                }
                if (lane == 0) lds[L.SC + 4 + j] = sc;
            }
            sync();
        }
        if constexpr (XB) push1(x[3], xbL, xbU, true, true);
        SC_UNROLL for (int j = 0; j < NU; ++j) push1(u[j], ubL[j], ubU[j], true, true);
        // One pass of the loop = one evaluation at the iterate + what the phase does with it:
        //   PH_INIT   first evaluation: slacks from the row values                          -> PH_LS
        //   PH_LS     least-square multiplier system: recursion, multipliers                  -> PH_START
        //   PH_START  evaluation with the multipliers: theta_0 for the filter's limits       -> (as PH_EVAL)
        //   PH_EVAL   errors, convergence tests, barrier parameter                            -> PH_BUILD
        //   PH_BUILD  Newton system with (mu, dw): recursion (Algorithm IC: PH_BUILD again with a larger dw), step, line search, update -> PH_EVAL
        enum { PH_INIT, PH_LS, PH_START, PH_EVAL, PH_BUILD };
        Eval2 E;
        double theta = 0.0, fsum = 0.0;
        double mu = O.mu_init, tau = fmax(O.tau_min, 1.0 - mu);
        double theta_max = INFINITY, theta_min = 0.0;
        const double mu_min = fmin(O.tol, O.compl_inf_tol) / (O.barrier_tol_factor + 1.0);
        int it = 0, n_acc = 0, status = SC_STATUS_INACCURATE, phase = PH_INIT;
        double last_alpha = 0.0, dw = 0.0;
        bool ic_first = true;
        // the regular algorithm's state while the restoration runs (its filter stays in FP / FT), and the iterate the restoration started from
        int o_nfilt = 0, o_nacc = 0;
        double o_mu = 0.0, o_theta = 0.0, o_phi = 0.0, o_pinf = 0.0, o_dw_last = 0.0, o_theta_max = 0.0, o_theta_min = 0.0;
        bool r_first = false, want_resto = false, presolved = false;
        int n_tiny = 0;                                                    // consecutive accepted steps below stall_alpha (of the phase the solve is in)
        int n_floor = 0;                                                   // consecutive regular iterates at the precision floor (sc_ipopt_params.floor_iter)
        for (;;) {
            if (filt_over) { status = SC_STATUS_INACCURATE; break; }
            if (want_resto) {
                // ---- enter the restoration phase (oracle/ms_ipopt.py: _Algo.restoration) at the current iterate; E, theta, fsum are of this iterate ----
                want_resto = false;
                if (rs) { status = SC_STATUS_INACCURATE; break; }           // (a failure INSIDE the restoration: resto_failed)
                double pm = 0.0;
                if (stg) {
                    SC_UNROLL for (int i = 0; i < NX; ++i) pm = fmax(pm, fabs(rc[i]));
                    for (int j = q; j < K; j += G) if (j < K) pm = fmax(pm, fabs(lds[ri(R_DV, j)] - lds[ri(R_S, j)]));
                }
                if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) pm = fmax(pm, fabs(x[i] - x0[i]));
                pm = cx.wmax(pm);
                if (pm <= O.resto_failure_feasibility_threshold) { status = SC_STATUS_INACCURATE; break; }      // "called at a point that is almost feasible"
                const double phi = barrier(fsum, Bcur, mu);
                filter_add(phi - O.gamma_phi * theta, (1.0 - O.gamma_theta) * theta);
                o_nfilt = nfilt; o_nacc = n_acc; o_mu = mu; o_theta = theta; o_phi = phi; o_pinf = pm; o_dw_last = dw_last; o_theta_max = theta_max; o_theta_min = theta_min;
                fpo = L.FP2; fto = L.FT2; nfilt = 0; n_acc = 0; dw_last = 0.0;
                mu = fmax(mu, pm); tau = fmax(O.tau_min, 1.0 - mu);
                zeta = O.resto_proximity_weight * sqrt(mu);
                const double rho_R = O.resto_penalty_parameter;
                if (act) {
                    if (acl) SC_UNROLL for (int i = 0; i < NX; ++i) lds[XRi(i)] = x[i];
                    zxL = fmin(rho_R, zxL); zxU = fmin(rho_R, zxU);
                }
                if (stg) {
                    SC_UNROLL for (int j = 0; j < NU; ++j) { if (stl) lds[XRi(NX + j)] = u[j]; zuL[j] = fmin(rho_R, zuL[j]); zuU[j] = fmin(rho_R, zuU[j]); }
                    for (int j = q; j < K; j += G) {
                        if (j < K) {                                        // eq. (33): the n, p >= 0 that minimise rho (n + p) - mu (log n + log p) on  r + n - p = 0
                            const double r = lds[ri(R_DV, j)] - lds[ri(R_S, j)], a = (mu - rho_R * r) / (2.0 * rho_R);
                            const double n = a + sqrt(a * a + mu * r / (2.0 * rho_R)), pp = r + n;
                            lds[ri(R_N, j)] = n; lds[ri(R_P, j)] = pp; lds[ri(R_ZN, j)] = mu / n; lds[ri(R_ZP, j)] = mu / pp;
                            lds[ri(R_VU, j)] = fmin(rho_R, lds[ri(R_VU, j)]); lds[ri(R_YD, j)] = 0.0;
                        }
                    }
                    SC_UNROLL for (int i = 0; i < NX; ++i) yc[i] = 0.0;
                }
                if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = 0.0;
                rs = true; r_first = true; n_tiny = 0;
                phase = PH_START;
                continue;
            }
            const bool ls = phase == PH_LS, build = phase == PH_LS || phase == PH_BUILD;
            // -DSC_DUMS_EARLY_BUILD: PH_START / PH_EVAL build the Newton system on the way (dw = 0) with the barrier parameter they arrive with, and
            // PH_BUILD skips its evaluation when the mu update left it alone (most iterations inside a restoration).  Measured on configs[2]: the
            // cycle counters of the longest solve say 116 k -> 109 k per iteration, the launch says 4.45 -> 4.53 ms (twice, alternating builds): not on.
            // (Neither is the variant that always skips the second pass by keeping the gradient as G0 + mu G1: 5.34 ms against 4.95.)
#ifdef SC_DUMS_EARLY_BUILD
            const bool early = phase == PH_START || phase == PH_EVAL;
#else
            const bool early = false;
#endif
            const double mu_eval = mu;
            DPROF_T0
            if (!(phase == PH_BUILD && presolved)) eval2(build || early, ls, E, mu, early ? 0.0 : dw, theta, fsum);
            presolved = false;
            if (phase == PH_INIT) {
                if (stg) for (int j = q; j < K; j += G) { double v = lds[ri(R_DV, j)]; push1(v, 0.0, lds[ri(R_SU, j)], false, true); lds[ri(R_S, j)] = v; }
                phase = PH_LS; dw = 0.0; ic_first = true;
                continue;
            }
            if (build) {
                const double cs = (ls || rs) ? 0.0 : 2.0 * df;               // (the restoration's objective has no input-rate term)
                DPROF_ADD(2)
                const bool okf = riccati_backward<Cx, GEN>(cx, L, N, P.dt, cs * P.R[0], cs * P.R[1]);
                DPROF_ADD(3)
                if (!okf) {                                                 // Algorithm IC: the same system with a larger delta_w
                    if (ic_first) {
                        ic_first = false;
                        dw = dw_last == 0.0 ? O.first_hessian_perturbation : fmax(O.min_hessian_perturbation, O.perturb_dec_fact * dw_last);
                    } else dw *= dw_last == 0.0 ? O.perturb_inc_fact_first : O.perturb_inc_fact;
                    if (dw > O.max_hessian_perturbation) {
                        if (ls) { phase = PH_START; continue; }             // (no least-square estimate: multipliers stay zero)
                        want_resto = true; continue;
                    }
                    continue;
                }
                if (dw > 0.0 && !ls) dw_last = dw;
                last_dw = dw;
                finish_step(ls, mu, dw, tau);
                DPROF_ADD(4)
                if (ls) {
                    double ym = 0.0;
                    if (stg) SC_UNROLL for (int i = 0; i < NX; ++i) ym = fmax(ym, fabs(lds[L.LAM + (k + 1) * 4 + i] / dgc(i)));
                    if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) ym = fmax(ym, fabs(lds[L.LAM + i]));
                    if (stg) for (int j = q; j < K; j += G) ym = fmax(ym, fabs(lds[ri(R_DYD, j)]));
                    ym = cx.wmax(ym);
                    if (ym <= O.constr_mult_init_max) {
                        if (stg) SC_UNROLL for (int i = 0; i < NX; ++i) yc[i] = lds[L.LAM + (k + 1) * 4 + i] / dgc(i);
                        if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = -lds[L.LAM + i];
                        if (stg) for (int j = q; j < K; j += G) lds[ri(R_YD, j)] = lds[ri(R_DYD, j)];
                    }
                    phase = PH_START;
                    continue;
                }
                // ---- filter line search ----
                double a_max, a_z, gBD;
                step_lengths(E, tau, mu, a_max, a_z, gBD);
                const double phi = barrier(fsum, Bcur, mu);
                double pw_t, pw_g;                                          // theta^s_theta, (-gBD)^s_phi: the switching condition (19) and alpha_min (23)
                cx.pow2(theta, O.s_theta, gBD < 0.0 ? -gBD : 1.0, O.s_phi, pw_t, pw_g);
                double a_min = O.gamma_theta;
                if (gBD < 0.0) {
                    a_min = fmin(a_min, O.gamma_phi * theta / (-gBD));
                    if (theta <= theta_min) a_min = fmin(a_min, O.delta * pw_t / pw_g);
                }
                a_min *= O.alpha_min_frac;
                const double sw_l = gBD < 0.0 ? pw_g : 0.0, sw_r = O.delta * pw_t;
                double alpha = a_max;
                bool first = true, accepted = false;
                DPROF_ADD(5)
                double xt[NX], ut[NU];
                double phi_t = 0.0, th_t = 0.0;
                while (alpha > a_min || first) {
                    SC_UNROLL for (int i = 0; i < NX; ++i) xt[i] = x[i] + alpha * dx[i];
                    SC_UNROLL for (int j = 0; j < NU; ++j) ut[j] = u[j] + alpha * du[j];
                    safe_slacks_xu(xt, ut, mu);
                    double f_t;
                    BarAcc Bt;
                    eval0(xt, ut, th_t, f_t, Bt, mu, alpha, true);
                    phi_t = barrier(f_t, Bt, mu);
                    if (phi_t < INFINITY && th_t == th_t && phi_t == phi_t) {
                        bool ok = th_t <= theta_max;
                        if (ok) {
                            const bool ftype = gBD < 0.0 && alpha * sw_l > sw_r;
                            if (alpha > 0.0 && ftype && theta <= theta_min) ok = cmp_le(phi_t - phi, O.eta_phi * alpha * gBD, phi);
                            else {
                                ok = true;
                                if (phi_t > phi) {
                                    const double bas = fabs(phi) > 10.0 ? fmax(1.0, log10(fabs(phi))) : 1.0;
                                    if (log10(phi_t - phi) > O.obj_max_inc + bas) ok = false;
                                }
                                if (ok) ok = cmp_le(th_t, (1.0 - O.gamma_theta) * theta, theta) || cmp_le(phi_t - phi, -O.gamma_phi * theta, phi);
                            }
                            if (ok) ok = filter_ok(phi_t, th_t);
                        }
                        if (ok) { accepted = true; break; }
                    }
                    first = false;
                    alpha *= O.alpha_red_factor;
                }
                if (!accepted) { want_resto = true; continue; }
                DPROF_ADD(6)
                {
                    const bool ftype = gBD < 0.0 && alpha * sw_l > sw_r;
                    const bool arm = cmp_le(phi_t - phi, O.eta_phi * alpha * gBD, phi);
                    if (!ftype || !arm) filter_add(phi - O.gamma_phi * theta, (1.0 - O.gamma_theta) * theta);
                }
                last_alpha = alpha;
                n_tiny = alpha < O.stall_alpha ? n_tiny + 1 : 0;
                SC_UNROLL for (int i = 0; i < NX; ++i) x[i] = xt[i];
                if (stg) SC_UNROLL for (int i = 0; i < NX; ++i) yc[i] += alpha * (lds[L.LAM + (k + 1) * 4 + i] / dgc(i));
                if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.Y0 + i] += alpha * (-lds[L.LAM + i]);
                SC_UNROLL for (int j = 0; j < NU; ++j) u[j] = ut[j];
                safe_slacks_xu(x, u, mu);
                // bound multipliers: z += a_z dz, then kappa_sigma; the rows in ONE pass: slack, multiplier, safe slack, slack-bound multiplier (n, p)
                {
                    const double ks = O.kappa_sigma, ksm = ks * mu, mks = mu / ks;
                    auto upd = [&](double& z, double dz, double sl) { const double is = 1.0 / sl; z += a_z * dz; z = fmax(fmin(z, ksm * is), mks * is); };
                    if (XB && act) { upd(zxL, dzxL, x[3] - xbL); upd(zxU, dzxU, xbU - x[3]); }
                    if (stg) {
                        SC_UNROLL for (int j = 0; j < NU; ++j) { upd(zuL[j], dzuL[j], u[j] - ubL[j]); upd(zuU[j], dzuU[j], ubU[j] - u[j]); }
                        const double s_min = EPS_ * fmin(1.0, mu), move = 1.8189894035458565e-12;     // eps^(3/4)
                        for (int j = q; j < K; j += G) {
                            const double sj = lds[ri(R_S, j)] + alpha * lds[ri(R_DS, j)];
                            lds[ri(R_S, j)] = sj;
                            lds[ri(R_YD, j)] += alpha * lds[ri(R_DYD, j)];
                            double bj = lds[ri(R_SU, j)];
                            safe1(sj, bj, false, s_min, move);
                            lds[ri(R_SU, j)] = bj;
                            double z = lds[ri(R_VU, j)];
                            upd(z, lds[ri(R_DVU, j)], bj - sj);
                            lds[ri(R_VU, j)] = z;
                            if (rs) {
                                const double n = lds[ri(R_N, j)] + alpha * lds[ri(R_DN, j)], pp = lds[ri(R_P, j)] + alpha * lds[ri(R_DP, j)];
                                double zn = lds[ri(R_ZN, j)], zp = lds[ri(R_ZP, j)];
                                upd(zn, lds[ri(R_DZN, j)], n); upd(zp, lds[ri(R_DZP, j)], pp);
                                lds[ri(R_N, j)] = n; lds[ri(R_P, j)] = pp; lds[ri(R_ZN, j)] = zn; lds[ri(R_ZP, j)] = zp;
                            }
                        }
                    }
                }
                ++it;
                DPROF_ADD(7)
                phase = PH_EVAL;
                continue;
            }
            // ---- PH_START / PH_EVAL: errors, convergence, barrier parameter ----
            DPROF_ADD(0)
            if (phase == PH_START) { theta_max = (rs ? O.resto_theta_max_fact : O.theta_max_fact) * fmax(1.0, theta); theta_min = O.theta_min_fact * fmax(1.0, theta); }
            double E0, Emu, dinf, pinf, comp, un_pinf;
            double sc_c = 1.0;
            errors(E, mu, E0, Emu, dinf, pinf, comp, un_pinf, &sc_c);
            if (trace && lane == 0) {
                double* t = trace + (size_t)(it < O.max_iter ? it : O.max_iter) * TRACE_W;
                t[0] = E0; t[1] = dinf; t[2] = pinf; t[3] = comp; t[4] = mu; t[5] = theta; t[6] = last_dw; t[7] = rs ? -last_alpha : last_alpha;      // (a negative step length marks an iterate of the restoration)
            }
            bool conv = false;
            if (rs) {
                // RestoFilterConvergenceCheck: back to the regular algorithm when (x, s) is acceptable to ITS filter and to the iterate the restoration
                // started from, with the infeasibility down to kappa_resto of what it was; and the restoration's own convergence tests (unscaled problem)
                double th_o = 0.0, f_o = 0.0, pm_o = 0.0;
                rs = false;
                BarAcc Bo;
                eval0(x, u, th_o, f_o, Bo, o_mu, 0.0, false, &pm_o);
                const double phi_o = barrier(f_o, Bo, o_mu);
                rs = true;
                bool leave = !r_first && pm_o <= O.required_infeasibility_reduction * o_pinf && phi_o < INFINITY && phi_o == phi_o;
                if (leave) {
                    for (int i = 0; i < o_nfilt; ++i) {
                        const double p = lds[L.FP + i], t = lds[L.FT + i];
                        if (!(cmp_le(phi_o, p, p) || cmp_le(th_o, t, t))) { leave = false; break; }
                    }
                }
                if (leave && phi_o > o_phi) {
                    const double bas = fabs(o_phi) > 10.0 ? fmax(1.0, log10(fabs(o_phi))) : 1.0;
                    if (log10(phi_o - o_phi) > O.obj_max_inc + bas) leave = false;
                }
                if (leave) leave = cmp_le(th_o, (1.0 - O.gamma_theta) * o_theta, o_theta) || cmp_le(phi_o - o_phi, -O.gamma_phi * o_theta, o_phi);
                if (leave) {
                    // bound multipliers of (x, s) come back (reset to 1 when one of them is beyond bound_mult_reset_threshold), the others start at zero
                    double zm = 0.0;
                    if (act) zm = fmax(zxL, zxU);
                    if (stg) {
                        SC_UNROLL for (int j = 0; j < NU; ++j) zm = fmax(zm, fmax(zuL[j], zuU[j]));
                        for (int j = q; j < K; j += G) if (j < K) zm = fmax(zm, lds[ri(R_VU, j)]);
                    }
                    zm = cx.wmax(zm);
                    if (zm > O.bound_mult_reset_threshold) {
                        if constexpr (XB) zxL = zxU = 1.0;
                        SC_UNROLL for (int j = 0; j < NU; ++j) { zuL[j] = 1.0; zuU[j] = 1.0; }
                        if (stg) for (int j = q; j < K; j += G) lds[ri(R_VU, j)] = 1.0;
                    }
                    SC_UNROLL for (int i = 0; i < NX; ++i) yc[i] = 0.0;
                    if (stg) for (int j = q; j < K; j += G) lds[ri(R_YD, j)] = 0.0;
                    if (lane == 0) SC_UNROLL for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = 0.0;
                    rs = false; n_tiny = 0; fpo = L.FP; fto = L.FT; nfilt = o_nfilt; n_acc = o_nacc; mu = o_mu; tau = fmax(O.tau_min, 1.0 - mu); dw_last = o_dw_last;
                    theta_max = o_theta_max; theta_min = o_theta_min;
                    phase = PH_EVAL;
                    continue;
                }
                if (E0 <= O.tol && dinf <= O.dual_inf_tol && pinf <= O.constr_viol_tol && comp <= O.compl_inf_tol) conv = true;
                else if (E0 <= O.acceptable_tol && dinf <= O.acceptable_dual_inf_tol && pinf <= O.acceptable_constr_viol_tol && comp <= O.acceptable_compl_inf_tol) {
                    if (++n_acc >= O.acceptable_iter) conv = true;
                } else n_acc = 0;
                if (conv) {                                                 // a stationary point of the violation: infeasible (certificate) unless it is feasible after all
                    status = pm_o <= 1e2 * O.tol ? SC_STATUS_INACCURATE : SC_STATUS_INFEASIBLE;
                    break;
                }
            } else {
                if (E0 <= O.tol && dinf / df <= O.dual_inf_tol && un_pinf <= O.constr_viol_tol && comp / df <= O.compl_inf_tol) { status = SC_STATUS_OPTIMAL; break; }
                if (E0 <= O.acceptable_tol && dinf / df <= O.acceptable_dual_inf_tol && un_pinf <= O.acceptable_constr_viol_tol && comp / df <= O.acceptable_compl_inf_tol) {
                    if (++n_acc >= O.acceptable_iter) { status = SC_STATUS_OPTIMAL; break; }
                } else n_acc = 0;
            }
            if (it >= O.max_iter) { status = SC_STATUS_INACCURATE; break; }
            if (!rs && O.floor_iter > 0) {                                  // (sc_ipopt_params.floor_iter)
                const bool at_floor = mu <= 10.0 * mu_min && n_acc == 0 && fmax(pinf, comp / sc_c) <= O.acceptable_tol && un_pinf <= O.acceptable_constr_viol_tol &&
                                      comp <= O.acceptable_compl_inf_tol * df;
                n_floor = at_floor ? n_floor + 1 : 0;
                if (n_floor >= O.floor_iter) { status = SC_STATUS_INACCURATE; break; }
            }
            if (O.stall_iter > 0 && n_tiny >= O.stall_iter) { status = SC_STATUS_INACCURATE; break; }       // (stall rule: see sc_ipopt_params)
            for (bool again = false;; again = true) {
                if (again) { double e0_, a, b, c, d; errors(E, mu, e0_, Emu, a, b, c, d); }     // (the barrier parameter went down: E_mu for the new one)
                if (Emu > O.barrier_tol_factor * mu || mu <= mu_min) break;
                const double mu_new = fmax(mu_min, fmin(O.mu_linear_decrease_factor * mu, pow(mu, O.mu_superlinear_decrease_power)));
                if (mu_new == mu) break;
                if (rs) {                                                   // the restoration's objective carries zeta = eta sqrt(mu): its gradient scales with it
                    const double sc_ = sqrt(mu_new / mu);
                    zeta *= sc_;
                    SC_UNROLL for (int i = 0; i < NX; ++i) E.gfx[i] *= sc_;
                    SC_UNROLL for (int j = 0; j < NU; ++j) E.gfu[j] *= sc_;
                }
                mu = mu_new; tau = fmax(O.tau_min, 1.0 - mu); nfilt = 0;
            }
            r_first = false;
            DPROF_ADD(1)
            phase = PH_BUILD; dw = 0.0; ic_first = true; presolved = early && mu == mu_eval;
        }
#ifdef SC_DUMS_PROF
        if (trace && lane == 0) { double* t = trace + (size_t)O.max_iter * TRACE_W; for (int i = 0; i < 8; ++i) t[i] = prof[i]; }
#endif
        status_out = status; iters_out = it;
    }
};

}  // namespace dums
}  // namespace sc
