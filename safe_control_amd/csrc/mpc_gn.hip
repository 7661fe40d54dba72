// MPC-CBF for the reference's planar models with a relative-degree-2 distance barrier and non-affine barrier points --
// DoubleIntegrator2D and Quad2D -- one NLP per wavefront.  (KinematicBicycle2D has the same structure -- its row of the
// table below is kept for reference -- but its fast heading dynamics need the second derivatives of the dynamics that
// this Gauss-Newton method drops: the oracle converges on under half of the test draws, so it is not served)
//   MPCCBF (position_control/mpc_cbf.py:7-402) over robots/double_integrator2D.py, robots/quad2D.py:
//     prediction  x+ = x + (f(x) + g(x) u) dt                                               mpc_cbf.py:135-141
//     cost        sum_k (x_k - xg)' Q (x_k - xg) + r-term R on delta u                       mpc_cbf.py:28-36,144,176-180
//     CBF         dd_h + (a1 + a2) d_h + a1 a2 h >= 0 with x1 = step(x_k, u_k), x2 = step(x1, u_k)  mpc_cbf.py:316-321;
//                 kinematic_bicycle2D.py:113-123,175-199 (speed clipped); double_integrator2D.py:79-107,222-272 (speed
//                 rescaled); quad2D.py:81-84,179-206
//     bounds      input boxes (KB adds |v_k| <= v_max: the state-bound rows are implemented, NB > 0)  mpc_cbf.py:193-216
//   Oracle: oracle/mpc_gn.py (problem functions, Gauss-Newton Hessian) + oracle/mpc_cbf.py: solve.
//
// The robot's own step() applies u_k twice and clips, so the barrier points b_k = pos(S(x_k, u_k)), c_k = pos(S(S(x_k, u_k), u_k))
// are functions of (x_k, u_k), not predicted positions.  Structure of one interior-point iteration:
//   * rollout and sensitivities: lane c < n owns COLUMN c of Phi_k = d x_k / d z and carries it through the horizon in
//     registers (Phi_{k+1} = A_k Phi_k + B_k E_k); the state, A_k, B_k and the Jacobians of S are wave-uniform and computed
//     redundantly by every lane, so the pass has no cross-lane step at all; it leaves Phi (cost, speed rows) and
//     G = d points / d z (6N x n) in LDS;
//   * Hessian of the Lagrangian, exact: sf (2 sum Phi_k' Q Phi_k + 2 D' R D) + G' Psi G + box terms (Psi = the 6 x 6 stage blocks
//     of J' Sigma J - sum lam grad^2 h over (a_k, b_k, c_k)) + sum_k V_k' H_k V_k, V_k = [Phi_k; E_k], where H_k collects the
//     second derivatives of the dynamics and of step o step weighted by the costates p_k of the Lagrangian (a backward
//     pass, again wave-uniform in registers; per model in closed form -- Quad2D: two scalars per stage);
//   * register Cholesky, fraction-to-boundary, l1-merit backtracking exactly as kernels 3 and 7.
// Arithmetic is f64; the caller's arrays are f32 or f64.
#include <hip/hip_runtime.h>

#include "../../include/safe_control_amd.h"
#include "sc_math.hpp"
#include "mpc_chol.hpp"
#include "mpc_ipm_common.hpp"
#include "mpc_cont.hpp"
// -DSC_GN_TRACE: developer build that writes 32 scalars per interior-point iteration (first 100 iterations) into z_out, which the
// tool then sizes as [B, 4096] doubles (tools/exp_gn_trace.py); never defined in the shipped library
#ifdef SC_GN_TRACE
#define GT(slot, val) do { if (lane == 0 && z_out && it <= 100) ((double*)z_out)[prob * 4096 + (it - 1) * 32 + (slot)] = (double)(val); } while (0)
#else
#define GT(slot, val)
#endif
#ifndef SC_CONT_LEVEL
#define SC_CONT_LEVEL 3      // developer switch (tools/build_variants.sh): 0 .. 2 compile parts of the continuation code out
#endif
#include "mpc_hd4.hpp"

// the superellipsoid powers: the multiply chain of ipm::pow3 for integer exponents, as mpc_lin.hip / mpc_cbf.hip.  Rounds 2 - 3 kept the
// library pow() here because the circles-only instantiations (Quad2D, KinematicBicycle2D at N = 10), which never execute that branch,
// returned garbage with the chain inlined: one of the faces of the compiler defect that round 4 root-caused (mpc_ipm_common.hpp: CHAIN;
// csrc/Makefile: SAFE_RA).  With the register-allocation flags of the Makefile the chain is right everywhere; -DSC_GN_CHAIN=false
// brings pow() back.
#ifndef SC_GN_CHAIN
#define SC_GN_CHAIN true
#endif

namespace sc {

namespace {

__device__ __forceinline__ double gsum(double v) { return ipm::wsum(v); }
__device__ __forceinline__ double gmin(double v) { return ipm::wmin(v); }
__device__ __forceinline__ double gmax_(double v) { return ipm::wmax(v); }

struct GnPar {                      // model constants (wave-uniform)
    double dt, Lr, v_min, v_max, mass, inertia, rad;
};

// ---- models: prediction F, the robot's own step S, their Jacobians (oracle/mpc_gn.py: kb_F, kb_S, di_F, di_S, q2_F) -------
template <int MODEL> struct GnModel;

template <> struct GnModel<SC_MODEL_DOUBLE_INTEGRATOR2D> {
    static constexpr int NX = 4, NB = 0, BIDX = 0;
    // Linear dynamics: the Gauss-Newton matrix IS the exact Hessian except where the speed rescaling of step() is active,
    // and including that curvature changes neither the convergence statistics nor the iteration counts on the test draws
    // while the backward pass costs 70 % more time per solve (measured) -- so it is left out (the oracle does the same).
    static constexpr bool EXACT = false;
    static constexpr int TIDX = 0, NH = 2, PD = 2, NP = 3;   // PD: dimension of a barrier point, NP: points per stage
    template <bool JAC, bool STEP>
    static __device__ __forceinline__ void map(const double* x, const double* u, const GnPar& q, double* xn, double (*A)[4], double (*B)[2]) {
        const double dt = q.dt;
        xn[0] = x[0] + dt * x[2]; xn[1] = x[1] + dt * x[3];
        double w0 = x[2] + dt * u[0], w1 = x[3] + dt * u[1];
        if constexpr (JAC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) A[i][j] = i == j ? 1.0 : 0.0;
                B[i][0] = 0.0; B[i][1] = 0.0;
            }
            A[0][2] = dt; A[1][3] = dt; B[2][0] = dt; B[3][1] = dt;
        }
        if constexpr (STEP) {                                               // speed rescaled to v_max
            const double vm = sqrt(w0 * w0 + w1 * w1);
            if (vm > q.v_max) {
                if constexpr (JAC) {
                    const double i1 = q.v_max / vm, i3 = q.v_max / (vm * vm * vm);
                    const double j00 = i1 - w0 * w0 * i3, j01 = -w0 * w1 * i3, j11 = i1 - w1 * w1 * i3;
                    // rows 2, 3 of [A | B] are [0 0 1 0 | dt 0], [0 0 0 1 | 0 dt] before the rescaling
                    A[2][2] = j00; A[2][3] = j01; A[3][2] = j01; A[3][3] = j11;
                    B[2][0] = j00 * dt; B[2][1] = j01 * dt; B[3][0] = j01 * dt; B[3][1] = j11 * dt;
                }
                const double sc_ = q.v_max / vm;
                w0 *= sc_; w1 *= sc_;
            }
        }
        xn[2] = w0; xn[3] = w1;
    }
};

template <> struct GnModel<SC_MODEL_QUAD2D> {
    static constexpr int NX = 6, NB = 0, BIDX = 0;
    // thrust direction: exact second derivatives take the solver from 97 % to 99 % optimal and from 21.5 to 13.8 iterations
    static constexpr bool EXACT = true;
    template <bool JAC, bool STEP>
    static __device__ __forceinline__ void map(const double* x, const double* u, const GnPar& q, double* xn, double (*A)[6], double (*B)[2]) {
        double s, c;
        sincos_(x[2], &s, &c);
        const double dt = q.dt, T = u[0] + u[1], im = 1.0 / q.mass, ri = q.rad / q.inertia;
        xn[0] = x[0] + dt * x[3]; xn[1] = x[1] + dt * x[4]; xn[2] = x[2] + dt * x[5];
        xn[3] = x[3] + dt * (-s * im) * T;
        xn[4] = x[4] + dt * (-9.81 + (c * im) * T);
        xn[5] = x[5] + dt * ri * (u[0] - u[1]);
        if constexpr (JAC) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j) A[i][j] = i == j ? 1.0 : 0.0;
                B[i][0] = 0.0; B[i][1] = 0.0;
            }
            A[0][3] = dt; A[1][4] = dt; A[2][5] = dt;
            A[3][2] = dt * (-c * im) * T; A[4][2] = dt * (-s * im) * T;
            B[3][0] = B[3][1] = dt * (-s * im); B[4][0] = B[4][1] = dt * (c * im); B[5][0] = dt * ri; B[5][1] = -dt * ri;
        }
    }
    // Second-order terms in closed form.  Only vx+ = vx - dt sin(theta) T / m and vz+ = vz + dt (cos(theta) T / m - g) are
    // nonlinear, in (theta, u): sum_i c_i grad^2 F_i = a e_t e_t' + b (e_t s' + s e_t'), a = c_3 dt sin T / m - c_4 dt cos T / m,
    // b = -(c_3 cos + c_4 sin) dt / m.  For stage k the weights are c = p_{k+1} + S2x' P' nu_2 (the point b_k is linear in
    // (x, u); c_k = pos(F(F(x, u), u)) reaches the nonlinear rows through x + dt v), the term D' H(y1; P' nu_2) D vanishes
    // (P' nu_2 has no velocity component).  Costates: p_N = mu_N, p_k = mu_k + (d points_k / d x_k)' nu + A_k' p_{k+1}, with
    // A_k = I + dt (e_0 e_3' + e_1 e_4' + e_2 e_5') + A32 e_3 e_2' + A42 e_4 e_2'.  Wave-uniform: every lane, in registers.
    static constexpr int TIDX = 2, NH = 2, PD = 2, NP = 3;                   // NH: second-order scalars per stage (a_k, b_k)
    static __device__ __forceinline__ void second_order(const double* xs, const double* z, const double* y, const double* cq,
                                                        const double* xg, double* ab, int N, double sf, const GnPar& q, int lane) {
        const double dt = q.dt, im = 1.0 / q.mass;
        double p[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) p[i] = sf * 2.0 * cq[i] * (xs[N * 6 + i] - xg[i]);
        for (int k = N - 1; k >= 0; --k) {
            double s, co;
            sincos_(xs[k * 6 + 2], &s, &co);
            const double T = z[2 * k] + z[2 * k + 1];
            const double n0x = -y[6 * k], n0y = -y[6 * k + 1], n1x = -y[6 * k + 2], n1y = -y[6 * k + 3], n2x = -y[6 * k + 4], n2y = -y[6 * k + 5];
            const double c3 = p[3] + dt * n2x, c4 = p[4] + dt * n2y;
            if (lane == 0) {
                ab[2 * k] = c3 * (dt * s * T * im) + c4 * (-dt * co * T * im);
                ab[2 * k + 1] = c3 * (-dt * co * im) + c4 * (-dt * s * im);
            }
            if (k >= 1) {
                const double A32 = dt * (-co * im) * T, A42 = dt * (-s * im) * T;
                double np_[6];
                np_[0] = p[0] + n0x + n1x + n2x;
                np_[1] = p[1] + n0y + n1y + n2y;
                np_[2] = p[2] + A32 * p[3] + A42 * p[4] + dt * (A32 * n2x + A42 * n2y);
                np_[3] = p[3] + dt * p[0] + dt * n1x + 2.0 * dt * n2x;
                np_[4] = p[4] + dt * p[1] + dt * n1y + 2.0 * dt * n2y;
                np_[5] = p[5] + dt * p[2];
#pragma unroll
                for (int i = 0; i < 6; ++i) p[i] = np_[i] + sf * 2.0 * cq[i] * (xs[k * 6 + i] - xg[i]);
            }
        }
    }
};

// KinematicBicycle2D (robots/kinematic_bicycle2D.py:75-123: x = (px, py, theta, v), u = (a, beta); f = [v cos, v sin, 0, 0],
// g = [[0, -v sin], [0, v cos], [0, v / Lr], [1, 0]]; step() = Euler + speed clipped to [v_min, v_max]); oracle/mpc_gn.py: kb_F,
// kb_S, kb_H.  The speed is a bounded state (mpc_cbf.py:205-207): NB = 1 pair of rows per stage on component BIDX = 3.
template <> struct GnModel<SC_MODEL_KINEMATIC_BICYCLE2D> {
    static constexpr int NX = 4, NB = 1, BIDX = 3;
    static constexpr bool EXACT = true;                     // heading dynamics: Gauss-Newton alone converges on a third of the draws
    static constexpr int TIDX = 2, NH = 10, PD = 2, NP = 3;  // NH: the 4 x 4 symmetric stage Hessian over (theta, v, a, beta)
    template <bool JAC, bool STEP>
    static __device__ __forceinline__ void map(const double* x, const double* u, const GnPar& q, double* xn, double (*A)[4], double (*B)[2]) {
        double s, c;
        sincos_(x[2], &s, &c);
        const double dt = q.dt, v = x[3], b = u[1], iL = 1.0 / q.Lr;
        xn[0] = x[0] + (v * c - v * s * b) * dt;
        xn[1] = x[1] + (v * s + v * c * b) * dt;
        xn[2] = x[2] + (v * iL * b) * dt;
        double vn = v + u[0] * dt;
        if constexpr (JAC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) A[i][j] = i == j ? 1.0 : 0.0;
                B[i][0] = 0.0; B[i][1] = 0.0;
            }
            A[0][2] = dt * (-v * s - v * c * b); A[0][3] = dt * (c - s * b);
            A[1][2] = dt * (v * c - v * s * b);  A[1][3] = dt * (s + c * b);
            A[2][3] = dt * b * iL;
            B[0][1] = -dt * v * s; B[1][1] = dt * v * c; B[2][1] = dt * v * iL; B[3][0] = dt;
        }
        if constexpr (STEP) {
            if (!(q.v_min <= vn && vn <= q.v_max)) {
                vn = fmin(fmax(vn, q.v_min), q.v_max);
                if constexpr (JAC) { A[3][3] = 0.0; B[3][0] = 0.0; }
            }
        }
        xn[3] = vn;
    }
    // Second-order terms (oracle/mpc_gn.py: evaluate, exact_hessian; kb_H).  sum_i c_i grad^2 F_i has four entries, on (theta, theta),
    // (theta, v), (theta, beta), (v, beta), the same for F and for step() (the clipped row is linear).  Per stage
    //   H_k = H(x_k, u_k; p_{k+1} + P'nu_1 + S2x'P'nu_2) + D' H(y1, u_k; P'nu_2) D,   y1 = step(x_k, u_k),  D = d(y1, u_k) / d(x_k, u_k),
    // a symmetric 4 x 4 over (theta, v, a, beta): ten scalars in Hk[10 k ..].  Costates: p_N = mu_N,
    // p_k = mu_k + P'nu_0 + S1x'(P'nu_1 + S2x'P'nu_2) + A_k' p_{k+1}; mu_k = sf 2 Q (x_k - xg) + (lam_hi - lam_lo) e_v.
    // Wave-uniform: every lane, in registers; lane 0 stores.
    static __device__ __forceinline__ void second_order(const double* xs, const double* z, const double* y, const double* cq,
                                                        const double* xg, const double* lam_v, double* Hk, int N, double sf,
                                                        const GnPar& q, int lane) {
        const double dt = q.dt, iL = 1.0 / q.Lr;
        double p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = sf * 2.0 * cq[i] * (xs[N * 4 + i] - xg[i]);
        p[3] += lam_v[2 * (N - 1)] - lam_v[2 * (N - 1) + 1];
        for (int k = N - 1; k >= 0; --k) {
            const double th = xs[k * 4 + 2], v = xs[k * 4 + 3], a = z[2 * k], b = z[2 * k + 1];
            double s, c, s1, c1;
            sincos_(th, &s, &c);
            const double th1 = th + (v * iL * b) * dt, v1u = v + a * dt;
            const bool cl = !(q.v_min <= v1u && v1u <= q.v_max);
            const double v1 = fmin(fmax(v1u, q.v_min), q.v_max);
            sincos_(th1, &s1, &c1);
            const double n0x = -y[6 * k], n0y = -y[6 * k + 1], n1x = -y[6 * k + 2], n1y = -y[6 * k + 3], n2x = -y[6 * k + 4], n2y = -y[6 * k + 5];
            // S2x rows 0, 1 (at y1): columns theta, v
            const double B02 = dt * (-v1 * s1 - v1 * c1 * b), B03 = dt * (c1 - s1 * b), B12 = dt * (v1 * c1 - v1 * s1 * b), B13 = dt * (s1 + c1 * b);
            const double w0 = n1x + n2x, w1 = n1y + n2y, w2 = B02 * n2x + B12 * n2y, w3 = B03 * n2x + B13 * n2y;   // P'nu_1 + S2x'P'nu_2
            const double ct0 = p[0] + w0, ct1 = p[1] + w1, ct2 = p[2] + w2;
            const double h22 = ct0 * (dt * (-v * c + v * s * b)) + ct1 * (dt * (-v * s - v * c * b));
            const double h23 = ct0 * (dt * (-s - c * b)) + ct1 * (dt * (c - s * b));
            const double h25 = ct0 * (-dt * v * c) + ct1 * (-dt * v * s);
            const double h35 = ct0 * (-dt * s) + ct1 * (dt * c) + ct2 * (dt * iL);
            const double g22 = n2x * (dt * (-v1 * c1 + v1 * s1 * b)) + n2y * (dt * (-v1 * s1 - v1 * c1 * b));
            const double g23 = n2x * (dt * (-s1 - c1 * b)) + n2y * (dt * (c1 - s1 * b));
            const double g25 = n2x * (-dt * v1 * c1) + n2y * (-dt * v1 * s1);
            const double g35 = n2x * (-dt * s1) + n2y * (dt * c1);
            if (lane == 0) {
                // rows of D for theta+, v+, beta over (theta, v, a, beta)
                const double d2[4] = {1.0, dt * b * iL, 0.0, dt * v * iL};
                const double d3[4] = {0.0, cl ? 0.0 : 1.0, cl ? 0.0 : dt, 0.0};
                const double d5[4] = {0.0, 0.0, 0.0, 1.0};
                double H[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        H[i][j] = g22 * d2[i] * d2[j] + g23 * (d2[i] * d3[j] + d3[i] * d2[j]) + g25 * (d2[i] * d5[j] + d5[i] * d2[j]) +
                                  g35 * (d3[i] * d5[j] + d5[i] * d3[j]);
                H[0][0] += h22; H[0][1] += h23; H[0][3] += h25; H[1][3] += h35;
                int o = 10 * k;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = i; j < 4; ++j) Hk[o++] = H[i][j];
            }
            if (k >= 1) {
                const double A02 = dt * (-v * s - v * c * b), A03 = dt * (c - s * b), A12 = dt * (v * c - v * s * b), A13 = dt * (s + c * b),
                             A23 = dt * b * iL;
                double np_[4];
                np_[0] = n0x + w0 + p[0];
                np_[1] = n0y + w1 + p[1];
                np_[2] = (w2 + A02 * w0 + A12 * w1) + (p[2] + A02 * p[0] + A12 * p[1]);
                np_[3] = ((cl ? 0.0 : w3) + A03 * w0 + A13 * w1 + A23 * w2) + (p[3] + A03 * p[0] + A13 * p[1] + A23 * p[2]);
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i] = np_[i] + sf * 2.0 * cq[i] * (xs[k * 4 + i] - xg[i]);
                p[3] += lam_v[2 * (k - 1)] - lam_v[2 * (k - 1) + 1];
            }
        }
    }
    // q_i' H_k q_j with q = (dtheta_k/dz_i, dv_k/dz_i, [i = a_k], [i = beta_k])
    static __device__ __forceinline__ double hess_term(const double* Ph, const double* Hk, int k, int i, int j, int n) {
        const double qi[4] = {Ph[(size_t)(k * 4 + 2) * n + i], Ph[(size_t)(k * 4 + 3) * n + i], i == 2 * k ? 1.0 : 0.0, i == 2 * k + 1 ? 1.0 : 0.0};
        const double qj[4] = {Ph[(size_t)(k * 4 + 2) * n + j], Ph[(size_t)(k * 4 + 3) * n + j], j == 2 * k ? 1.0 : 0.0, j == 2 * k + 1 ? 1.0 : 0.0};
        const double* H = Hk + 10 * k;
        const double r0 = H[0] * qj[0] + H[1] * qj[1] + H[2] * qj[2] + H[3] * qj[3];
        const double r1 = H[1] * qj[0] + H[4] * qj[1] + H[5] * qj[2] + H[6] * qj[3];
        const double r2 = H[2] * qj[0] + H[5] * qj[1] + H[7] * qj[2] + H[8] * qj[3];
        const double r3 = H[3] * qj[0] + H[6] * qj[1] + H[8] * qj[2] + H[9] * qj[3];
        return qi[0] * r0 + qi[1] * r1 + qi[2] * r2 + qi[3] * r3;
    }
};

// KinematicBicycle2D_C3BF / _DPCBF under the MPC (mpc_cbf.py:31-33,68-73,205-211,312-315): the bicycle's dynamics, cost and bounds with
// a rel-degree-1 discrete-time barrier of the FULL state, row = h(step(x_k, u_k)) - (1 - alpha) h(x_k); oracle/mpc_kb_state.py.
struct GnKbState : GnModel<SC_MODEL_KINEMATIC_BICYCLE2D> {
    static constexpr int PD = 4, NP = 2;
    // H_k = H(x_k, u_k; p_{k+1} + nu_1) (kb_H: entries (theta,theta), (theta,v), (theta,beta), (v,beta); the clipped row of step() is
    // linear); costates p_k = mu_k + nu_0 + S1x' nu_1 + A_k' p_{k+1}, nu_p = -w_p sum_j lam_kj dh_j(point p) = -y[8 k + 4 p ..].
    static __device__ __forceinline__ void second_order(const double* xs, const double* z, const double* y, const double* cq,
                                                        const double* xg, const double* lam_v, double* Hk, int N, double sf,
                                                        const GnPar& q, int lane) {
        const double dt = q.dt, iL = 1.0 / q.Lr;
        double p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = sf * 2.0 * cq[i] * (xs[N * 4 + i] - xg[i]);
        p[3] += lam_v[2 * (N - 1)] - lam_v[2 * (N - 1) + 1];
        for (int k = N - 1; k >= 0; --k) {
            const double th = xs[k * 4 + 2], v = xs[k * 4 + 3], a = z[2 * k], b = z[2 * k + 1];
            double s, c;
            sincos_(th, &s, &c);
            const double v1u = v + a * dt;
            const bool cl = !(q.v_min <= v1u && v1u <= q.v_max);
            double n0[4], n1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { n0[i] = -y[8 * k + i]; n1[i] = -y[8 * k + 4 + i]; }
            const double ct0 = p[0] + n1[0], ct1 = p[1] + n1[1], ct2 = p[2] + n1[2];
            if (lane == 0) {
                double* H = Hk + 10 * k;
#pragma unroll
                for (int t = 0; t < 10; ++t) H[t] = 0.0;
                H[0] = ct0 * (dt * (-v * c + v * s * b)) + ct1 * (dt * (-v * s - v * c * b));
                H[1] = ct0 * (dt * (-s - c * b)) + ct1 * (dt * (c - s * b));
                H[3] = ct0 * (-dt * v * c) + ct1 * (-dt * v * s);
                H[6] = ct0 * (-dt * s) + ct1 * (dt * c) + ct2 * (dt * iL);
            }
            if (k >= 1) {
                const double A02 = dt * (-v * s - v * c * b), A03 = dt * (c - s * b), A12 = dt * (v * c - v * s * b), A13 = dt * (s + c * b),
                             A23 = dt * b * iL;
                double np_[4];
                np_[0] = n0[0] + n1[0] + p[0];
                np_[1] = n0[1] + n1[1] + p[1];
                np_[2] = n0[2] + (n1[2] + A02 * n1[0] + A12 * n1[1]) + (p[2] + A02 * p[0] + A12 * p[1]);
                np_[3] = n0[3] + ((cl ? 0.0 : n1[3]) + A03 * n1[0] + A13 * n1[1] + A23 * n1[2]) + (p[3] + A03 * p[0] + A13 * p[1] + A23 * p[2]);
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i] = np_[i] + sf * 2.0 * cq[i] * (xs[k * 4 + i] - xg[i]);
                p[3] += lam_v[2 * (k - 1)] - lam_v[2 * (k - 1) + 1];
            }
        }
    }
    template <class F>
    static __device__ __forceinline__ void derivs_of(F fn, const double* xv, const double* o, double radius, double& h, double* g4, double* h10) {
        hd::Hd4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = hd::variable(xv[i], i);
        const hd::Hd4 r = fn(x, o, radius);
        h = r.v;
#pragma unroll
        for (int i = 0; i < 4; ++i) g4[i] = r.g[i];
#pragma unroll
        for (int t = 0; t < 10; ++t) h10[t] = r.h[t];
    }
};
template <> struct GnModel<SC_MODEL_KINEMATIC_BICYCLE2D_C3BF> : GnKbState {
    static __device__ __noinline__ void barrier_d(const double* xv, const double* o, double radius, double& h, double* g4, double* h10) {
        derivs_of([](const hd::Hd4* x, const double* ob, double r) { return hd::h_c3bf<hd::Hd4>(x, ob, r); }, xv, o, radius, h, g4, h10);
    }
    static __device__ __forceinline__ double barrier_v(const double* xv, const double* o, double radius) {
        const double x[4] = {xv[0], xv[1], xv[2], xv[3]};
        return hd::h_c3bf<double>(x, o, radius);
    }
};
template <> struct GnModel<SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF> : GnKbState {
    static __device__ __noinline__ void barrier_d(const double* xv, const double* o, double radius, double& h, double* g4, double* h10) {
        derivs_of([](const hd::Hd4* x, const double* ob, double r) { return hd::h_dpcbf<hd::Hd4>(x, ob, r); }, xv, o, radius, h, g4, h10);
    }
    static __device__ __forceinline__ double barrier_v(const double* xv, const double* o, double radius) {
        const double x[4] = {xv[0], xv[1], xv[2], xv[3]};
        return hd::h_dpcbf<double>(x, o, radius);
    }
};

struct GnMem {
    double *cq, *xg, *up;                                 // Q (6) | R (2) | u_lo (2) | u_hi (2) ; goal state ; previous input
    double *z, *zt, *zb, *dz, *gs, *rd, *rhs;             // n each
    double *xs, *Ph, *pts, *y, *pdz, *G, *T;              // (N+1) nx | (N+1) nx n | 6N | 6N | 6N | 6N n | 6N n
    double *obs, *hk, *dh, *hh;                           // 7K | 3N K | 6N K | 9N K
    double *g, *s, *lam, *ds, *dlam, *vb;                 // m each
    double *tel;                                          // N K: elastic variables of the feasibility restoration (mpc_ipm_common.hpp)
    // optimal decay (oracle/od_mpc_gn.py): two decay variables per stage
    double *rho, *rhot, *drho, *rhob, *rr, *rdr;          // 2 N each: current, trial, step, best iterate; right-hand side and r_d of the decay variables
    double *w0s, *w1s;                                    // N: stage weights of the rows (w2 = 1)
    double *A1, *A2;                                      // N K: d row / d rho_1, d rho_2
    double *Cod, *Dod;                                    // 12 N: point-space coupling C_k (6 x 2); 4 N: eigen form of D_k^-1 (vx, vy, 1/ls, 1/lw)
    double *Psi, *M, *L;                                  // 36 N | n n | (L: scratch in T)
    double *Hk;                                           // NH N: second-order terms of the dynamics per stage
};

struct GnDims { int N, K, n, m, mc, ms; bool circles; };

// circles: the barrier Hessian of a circle is 2 I, so its per-point table (9 N K) is not stored
__host__ __device__ inline size_t mpcgn_lds_doubles(int N, int K, int nx, int nb, bool circles, int nh = 2, int pd = 2, int np = 3, bool od = false) {
    const size_t n = 2 * (size_t)N, m = (size_t)N * K + 2 * (size_t)nb * N + 2 * n;
    const size_t rs = (size_t)pd * np, hs = (size_t)pd * (pd + 1) / 2;       // rows of a stage block (6 | 8), entries of a point Hessian (3 | 10)
    size_t tot = 12 + nx + 2 + 7 * n + (size_t)(N + 1) * nx + (size_t)(N + 1) * nx * n + 3 * rs * N + 2 * rs * N * n +
                 7 * (size_t)K + (size_t)np * (1 + pd + (circles ? 0 : hs)) * N * K + 5 * m + (size_t)N * K + rs * rs * N + n * n + (size_t)nh * N;
    const size_t need_l = n * (n + 1) + m, have = rs * N * n;                // Cholesky scratch L and the row vector vb live in T
    return tot + (need_l > have ? need_l - have : 0) + (od ? 30 * (size_t)N + 2 * (size_t)N * K : 0);
}

template <int NX, int NH, int PD, int NP>
__device__ inline GnMem carve_gn(double* b, const GnDims& d, bool od = false) {
    GnMem W;
    auto take = [&](size_t c) { double* r = b; b += c; return r; };
    const int N = d.N, K = d.K, n = d.n, m = d.m;
    W.cq = take(12); W.xg = take(NX); W.up = take(2);
    W.z = take(n); W.zt = take(n); W.zb = take(n); W.dz = take(n); W.gs = take(n); W.rd = take(n); W.rhs = take(n);
    W.xs = take((N + 1) * NX); W.Ph = take((size_t)(N + 1) * NX * n);
    constexpr int RS = PD * NP, HS = PD * (PD + 1) / 2;
    W.pts = take(RS * N); W.y = take(RS * N); W.pdz = take(RS * N);
    W.G = take((size_t)RS * N * n);
    W.obs = take(7 * K); W.hk = take(NP * N * K); W.dh = take(PD * NP * N * K); W.hh = take(d.circles ? 0 : HS * NP * N * K);
    W.g = take(m); W.s = take(m); W.lam = take(m); W.ds = take(m); W.dlam = take(m);
    W.tel = take((size_t)N * K);
    W.Psi = take(RS * RS * N); W.M = take((size_t)n * n);
    W.Hk = take((size_t)NH * N);
    W.T = take((size_t)RS * N * n); W.L = W.T;                     // T is dead once M is assembled
    W.vb = W.T + (size_t)n * (n + 1);                              // written before T is built and again after the solve
    W.rho = W.rhot = W.drho = W.rhob = W.rr = W.rdr = W.w0s = W.w1s = W.A1 = W.A2 = W.Cod = W.Dod = nullptr;
    if (od) {
        b = W.T + ((size_t)RS * N * n > (size_t)n * (n + 1) + m ? (size_t)RS * N * n : (size_t)n * (n + 1) + m);   // past T | L | vb
        W.rho = take(n); W.rhot = take(n); W.drho = take(n); W.rhob = take(n); W.rr = take(n); W.rdr = take(n);
        W.w0s = take(N); W.w1s = take(N); W.A1 = take((size_t)N * K); W.A2 = take((size_t)N * K);
        W.Cod = take((size_t)12 * N); W.Dod = take((size_t)4 * N);
    }
    return W;
}

struct GnConst {
    double w0, w1, w2, Rrob, beta, blo, bhi;
    int circles_only;
    double al1, al2, ps1, ps2, rf1, rf2;                  // optimal decay: DT gains, decay penalties and references
};
struct GnOd { double omega_ref[2], p_sb[2]; };

// rollout (+ sensitivities), f, barrier values (+ derivatives), g.  Point index: 3 k + p, p = 0 (a_k), 1 (b_k), 2 (c_k).
template <int MODEL, bool OD = false>
__device__ __forceinline__ double gn_eval(const double* zv, const GnMem& W, const GnDims& d, const GnConst& c, const GnPar& q,
                                          int lane, bool derivs, const double* rhov = nullptr) {
    using Mdl = GnModel<MODEL>;
    constexpr int NX = Mdl::NX;
    const int N = d.N, K = d.K, n = d.n;
    // every lane rolls the (wave-uniform) state out in registers; lane `lane` < n also carries column `lane` of Phi
    double x[NX], col[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { x[i] = W.xs[i]; col[i] = 0.0; }
    const int myk = lane >> 1, myc = lane & 1;                               // z index lane = input myc of stage myk
    if (!derivs) {
        // values only (line search): the prediction chain x_{k+1} = F(x_k, u_k) is sequential and wave-uniform; the barrier points
        // b_k = pos(S(x_k, u_k)), c_k = pos(S(S(x_k, u_k), u_k)) hang off x_k only, so lane k computes the two of stage k
        for (int k = 0; k < N; ++k) {
            const double u[2] = {zv[2 * k], zv[2 * k + 1]};
            double xn[NX];
            double (*nul4)[NX] = nullptr;
            double (*nul2)[2] = nullptr;
            Mdl::template map<false, false>(x, u, q, xn, nul4, nul2);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) W.xs[(k + 1) * NX + i] = xn[i];
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
        }
        SC_SYNC();
        for (int k = lane; k < N; k += 64) {
            const double u[2] = {zv[2 * k], zv[2 * k + 1]};
            double xk[NX], y1[NX], y2[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xk[i] = W.xs[k * NX + i];
            double (*nul4)[NX] = nullptr;
            double (*nul2)[2] = nullptr;
            Mdl::template map<false, true>(xk, u, q, y1, nul4, nul2);
            Mdl::template map<false, true>(y1, u, q, y2, nul4, nul2);
            W.pts[6 * k + 0] = xk[0]; W.pts[6 * k + 1] = xk[1];
            W.pts[6 * k + 2] = y1[0]; W.pts[6 * k + 3] = y1[1];
            W.pts[6 * k + 4] = y2[0]; W.pts[6 * k + 5] = y2[1];
        }
    } else
    for (int k = 0; k < N; ++k) {
        const double u[2] = {zv[2 * k], zv[2 * k + 1]};
        double xn[NX], y1[NX], y2[NX];
        if (derivs) {
            double A[NX][NX], B[NX][2], S1x[NX][NX], S1u[NX][2], S2x[NX][NX], S2u[NX][2];
            Mdl::template map<true, false>(x, u, q, xn, A, B);
            Mdl::template map<true, true>(x, u, q, y1, S1x, S1u);
            Mdl::template map<true, true>(y1, u, q, y2, S2x, S2u);
            const bool mine = (lane < n) && (myk == k);
            double nc[NX], Y1[NX], Y2[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = 0.0, b1 = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) { a += A[i][j] * col[j]; b1 += S1x[i][j] * col[j]; }
                nc[i] = a + (mine ? B[i][myc] : 0.0);
                Y1[i] = b1 + (mine ? S1u[i][myc] : 0.0);
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double b2 = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) b2 += S2x[i][j] * Y1[j];
                Y2[i] = b2 + (mine ? S2u[i][myc] : 0.0);
            }
            if (lane < n) {
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    W.G[(size_t)((3 * k + 0) * 2 + dd) * n + lane] = col[dd];
                    W.G[(size_t)((3 * k + 1) * 2 + dd) * n + lane] = Y1[dd];
                    W.G[(size_t)((3 * k + 2) * 2 + dd) * n + lane] = Y2[dd];
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) { col[i] = nc[i]; W.Ph[(size_t)((k + 1) * NX + i) * n + lane] = nc[i]; }
            }
        } else {
            double (*nul4)[NX] = nullptr;
            double (*nul2)[2] = nullptr;
            Mdl::template map<false, false>(x, u, q, xn, nul4, nul2);
            Mdl::template map<false, true>(x, u, q, y1, nul4, nul2);
            Mdl::template map<false, true>(y1, u, q, y2, nul4, nul2);
        }
        if (lane == 0) {
            W.pts[6 * k + 0] = x[0]; W.pts[6 * k + 1] = x[1];
            W.pts[6 * k + 2] = y1[0]; W.pts[6 * k + 3] = y1[1];
            W.pts[6 * k + 4] = y2[0]; W.pts[6 * k + 5] = y2[1];
#pragma unroll
            for (int i = 0; i < NX; ++i) W.xs[(k + 1) * NX + i] = xn[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
    }
    SC_SYNC();
    double part = 0.0;
    for (int e = lane; e < N * NX; e += 64) {
        const int k = e / NX + 1, i = e - (k - 1) * NX;
        const double dv = W.xs[k * NX + i] - W.xg[i];
        part += W.cq[i] * dv * dv;
    }
    for (int i = lane; i < n; i += 64) {
        const double prev = i >= 2 ? zv[i - 2] : W.up[i];
        const double du = OD ? zv[i] : zv[i] - prev;                          // OD: R u^2 (optimal_decay_mpc_cbf.py:178-179)
        part += W.cq[6 + (i & 1)] * du * du;
    }
    if constexpr (OD) {
        // stage weights of the rows  w1_k = s_k - 2,  w0_k = 1 - s_k + q_k  (w2 = 1)  and the decay penalty
        for (int k = lane; k < N; k += 64) {
            const double r1 = rhov[2 * k], r2 = rhov[2 * k + 1];
            const double sk = c.al1 * r1 + c.al2 * r2, qk = c.al1 * c.al2 * r1 * r2;
            W.w0s[k] = 1.0 - sk + qk; W.w1s[k] = sk - 2.0;
            part += c.ps1 * (r1 - c.rf1) * (r1 - c.rf1) + c.ps2 * (r2 - c.rf2) * (r2 - c.rf2);
        }
    }
    for (int e = lane; e < 3 * N * K; e += 64) {
        const int pt = e / K, j = e - pt * K;
        double h, d0, d1, hxx, hxy, hyy;
        ipm::ipm_barrier<SC_GN_CHAIN>(W.pts[2 * pt], W.pts[2 * pt + 1], W.obs + 7 * j, c.Rrob, c.beta, c.circles_only != 0, derivs, h, d0, d1, hxx, hxy, hyy);
        W.hk[e] = h;
        if (derivs) {
            W.dh[2 * e] = d0; W.dh[2 * e + 1] = d1;
            if (!d.circles) { W.hh[3 * e] = hxx; W.hh[3 * e + 1] = hxy; W.hh[3 * e + 2] = hyy; }
        }
    }
    SC_SYNC();
    for (int i = lane; i < d.m; i += 64) {
        double gi;
        if (i < d.mc) {
            const int k = i / K, j = i - k * K;
            const double h0 = W.hk[(3 * k) * K + j], h1 = W.hk[(3 * k + 1) * K + j], h2 = W.hk[(3 * k + 2) * K + j];
            if constexpr (OD) {
                gi = W.w0s[k] * h0 + W.w1s[k] * h1 + h2;
                if (derivs) {                                                 // d row / d rho_i = a_i (h1 - h0) + a1 a2 rho_other h0
                    const double aa = c.al1 * c.al2 * h0;
                    W.A1[i] = c.al1 * (h1 - h0) + aa * rhov[2 * k + 1];
                    W.A2[i] = c.al2 * (h1 - h0) + aa * rhov[2 * k];
                }
            } else {
                gi = c.w0 * h0 + c.w1 * h1 + c.w2 * h2;
            }
        } else if (i < d.mc + d.ms) {
            const int r = i - d.mc, k = (r >> 1) + 1;
            const double xv = W.xs[k * NX + Mdl::BIDX];
            gi = (r & 1) ? xv - c.blo : c.bhi - xv;
        } else if (i < d.mc + d.ms + n) {
            const int col_ = i - d.mc - d.ms;
            gi = W.cq[10 + (col_ & 1)] - zv[col_];
        } else {
            const int col_ = i - d.mc - d.ms - n;
            gi = zv[col_] - W.cq[8 + (col_ & 1)];
        }
        W.g[i] = gi;
    }
    SC_SYNC();
    return gsum(part);
}

// out = J' v:  G' (A' v) + Phi' (speed rows) - v_hi + v_lo
// OD, ycorr: the point-space vector C_k D_k^-1 rr_k is subtracted from A' v before the G' product (Schur correction of the right-hand side)
template <int MODEL, bool OD = false>
__device__ __forceinline__ void gn_jt(const double* v, double* out, const GnMem& W, const GnDims& d, const GnConst& c, int lane, bool ycorr = false) {
    using Mdl = GnModel<MODEL>;
    constexpr int NX = Mdl::NX;
    const int N = d.N, K = d.K, n = d.n;
    for (int e = lane; e < 6 * N; e += 64) {
        const int pt = e >> 1, dd = e & 1, k = pt / 3, p = pt - 3 * k;
        double acc = 0.0;
        for (int j = 0; j < K; ++j) acc += v[k * K + j] * W.dh[2 * (pt * K + j) + dd];
        double yv;
        if constexpr (OD) {
            yv = (p == 0 ? W.w0s[k] : (p == 1 ? W.w1s[k] : 1.0)) * acc;
            if (ycorr) {
                const double* Cm = W.Cod + 12 * k + 2 * (2 * p + dd);
                const double* Dk = W.Dod + 4 * k;
                const double vx = Dk[0], vy = Dk[1];
                const double ts = (vx * W.rr[2 * k] + vy * W.rr[2 * k + 1]) * Dk[2], tw = (-vy * W.rr[2 * k] + vx * W.rr[2 * k + 1]) * Dk[3];
                yv -= (Cm[0] * vx + Cm[1] * vy) * ts + (-Cm[0] * vy + Cm[1] * vx) * tw;
            }
        } else {
            yv = (p == 0 ? c.w0 : (p == 1 ? c.w1 : c.w2)) * acc;
        }
        W.y[e] = yv;
    }
    SC_SYNC();
    for (int i = lane; i < n; i += 64) {
        double acc = 0.0;
        for (int r = 0; r < 6 * N; ++r) acc += W.G[(size_t)r * n + i] * W.y[r];
        if constexpr (Mdl::NB > 0) {
            for (int k = 1; k <= N; ++k)                                       // rows hi - x: -Phi ; x - lo: +Phi
                acc += (v[d.mc + 2 * (k - 1) + 1] - v[d.mc + 2 * (k - 1)]) * W.Ph[(size_t)(k * NX + Mdl::BIDX) * n + i];
        }
        out[i] = acc - v[d.mc + d.ms + i] + v[d.mc + d.ms + n + i];
    }
    SC_SYNC();
}

// ---- full-state barrier points (KinematicBicycle2D_C3BF / _DPCBF; oracle/mpc_kb_state.py) -------------------------------------
// Stage k has two points of dimension four, x_k and y1 = step(x_k, u_k); row (k, j) is w0 h(x_k) + w1 h(y1) with w = (alpha - 1, 1).
// Point index 2 k + p; tables: hk[e], dh[4 e + d], hh[10 e + t] (upper triangle by rows) for e = point * K + j;
// G rows (2 k + p) * 4 + d; stage blocks Psi are 8 x 8.
template <int MODEL>
__device__ __forceinline__ double gn_eval_state(const double* zv, const GnMem& W, const GnDims& d, const GnConst& c, const GnPar& q,
                                                int lane, bool derivs) {
    using Mdl = GnModel<MODEL>;
    constexpr int NX = 4;
    const int N = d.N, K = d.K, n = d.n;
    double x[NX], col[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { x[i] = W.xs[i]; col[i] = 0.0; }
    const int myk = lane >> 1, myc = lane & 1;
    if (!derivs) {                                                          // values only: sequential chain, then lane k takes stage k
        for (int k = 0; k < N; ++k) {
            const double u[2] = {zv[2 * k], zv[2 * k + 1]};
            double xn[NX];
            double (*nul4)[NX] = nullptr;
            double (*nul2)[2] = nullptr;
            Mdl::template map<false, false>(x, u, q, xn, nul4, nul2);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) W.xs[(k + 1) * NX + i] = xn[i];
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
        }
        SC_SYNC();
        for (int k = lane; k < N; k += 64) {
            const double u[2] = {zv[2 * k], zv[2 * k + 1]};
            double xk[NX], y1[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xk[i] = W.xs[k * NX + i];
            double (*nul4)[NX] = nullptr;
            double (*nul2)[2] = nullptr;
            Mdl::template map<false, true>(xk, u, q, y1, nul4, nul2);
#pragma unroll
            for (int i = 0; i < NX; ++i) { W.pts[8 * k + i] = xk[i]; W.pts[8 * k + 4 + i] = y1[i]; }
        }
    } else
    for (int k = 0; k < N; ++k) {
        const double u[2] = {zv[2 * k], zv[2 * k + 1]};
        double xn[NX], y1[NX];
        if (derivs) {
            double A[NX][NX], B[NX][2], S1x[NX][NX], S1u[NX][2];
            Mdl::template map<true, false>(x, u, q, xn, A, B);
            Mdl::template map<true, true>(x, u, q, y1, S1x, S1u);
            const bool mine = (lane < n) && (myk == k);
            double nc[NX], Y1[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double a = 0.0, b1 = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j) { a += A[i][j] * col[j]; b1 += S1x[i][j] * col[j]; }
                nc[i] = a + (mine ? B[i][myc] : 0.0);
                Y1[i] = b1 + (mine ? S1u[i][myc] : 0.0);
            }
            if (lane < n) {
#pragma unroll
                for (int dd = 0; dd < NX; ++dd) {
                    W.G[(size_t)((2 * k + 0) * 4 + dd) * n + lane] = col[dd];
                    W.G[(size_t)((2 * k + 1) * 4 + dd) * n + lane] = Y1[dd];
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) { col[i] = nc[i]; W.Ph[(size_t)((k + 1) * NX + i) * n + lane] = nc[i]; }
            }
        } else {
            double (*nul4)[NX] = nullptr;
            double (*nul2)[2] = nullptr;
            Mdl::template map<false, false>(x, u, q, xn, nul4, nul2);
            Mdl::template map<false, true>(x, u, q, y1, nul4, nul2);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { W.pts[8 * k + i] = x[i]; W.pts[8 * k + 4 + i] = y1[i]; W.xs[(k + 1) * NX + i] = xn[i]; }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xn[i];
    }
    SC_SYNC();
    double part = 0.0;
    for (int e = lane; e < N * NX; e += 64) {
        const int k = e / NX + 1, i = e - (k - 1) * NX;
        const double dv = W.xs[k * NX + i] - W.xg[i];
        part += W.cq[i] * dv * dv;
    }
    for (int i = lane; i < n; i += 64) {
        const double prev = i >= 2 ? zv[i - 2] : W.up[i];
        const double du = zv[i] - prev;
        part += W.cq[6 + (i & 1)] * du * du;
    }
    for (int e = lane; e < 2 * N * K; e += 64) {
        const int pt = e / K, j = e - pt * K;
        if (derivs) {
            double h, g4[4], h10[10];
            Mdl::barrier_d(W.pts + 4 * pt, W.obs + 7 * j, c.Rrob, h, g4, h10);
            W.hk[e] = h;
#pragma unroll
            for (int t = 0; t < 4; ++t) W.dh[4 * e + t] = g4[t];
#pragma unroll
            for (int t = 0; t < 10; ++t) W.hh[10 * e + t] = h10[t];
        } else {
            W.hk[e] = Mdl::barrier_v(W.pts + 4 * pt, W.obs + 7 * j, c.Rrob);
        }
    }
    SC_SYNC();
    for (int i = lane; i < d.m; i += 64) {
        double gi;
        if (i < d.mc) {
            const int k = i / K, j = i - k * K;
            gi = c.w0 * W.hk[(2 * k) * K + j] + c.w1 * W.hk[(2 * k + 1) * K + j];
        } else if (i < d.mc + d.ms) {
            const int r = i - d.mc, k = (r >> 1) + 1;
            const double xv = W.xs[k * NX + Mdl::BIDX];
            gi = (r & 1) ? xv - c.blo : c.bhi - xv;
        } else if (i < d.mc + d.ms + n) {
            const int col_ = i - d.mc - d.ms;
            gi = W.cq[10 + (col_ & 1)] - zv[col_];
        } else {
            const int col_ = i - d.mc - d.ms - n;
            gi = zv[col_] - W.cq[8 + (col_ & 1)];
        }
        W.g[i] = gi;
    }
    SC_SYNC();
    return gsum(part);
}

template <int MODEL>
__device__ __forceinline__ void gn_jt_state(const double* v, double* out, const GnMem& W, const GnDims& d, const GnConst& c, int lane) {
    using Mdl = GnModel<MODEL>;
    constexpr int NX = 4;
    const int N = d.N, K = d.K, n = d.n;
    for (int e = lane; e < 8 * N; e += 64) {
        const int pt = e >> 2, dd = e & 3, k = pt >> 1, p = pt & 1;
        double acc = 0.0;
        for (int j = 0; j < K; ++j) acc += v[k * K + j] * W.dh[4 * (pt * K + j) + dd];
        W.y[e] = (p == 0 ? c.w0 : c.w1) * acc;
    }
    SC_SYNC();
    for (int i = lane; i < n; i += 64) {
        double acc = 0.0;
        for (int r = 0; r < 8 * N; ++r) acc += W.G[(size_t)r * n + i] * W.y[r];
        for (int k = 1; k <= N; ++k)
            acc += (v[d.mc + 2 * (k - 1) + 1] - v[d.mc + 2 * (k - 1)]) * W.Ph[(size_t)(k * NX + Mdl::BIDX) * n + i];
        out[i] = acc - v[d.mc + d.ms + i] + v[d.mc + d.ms + n + i];
    }
    SC_SYNC();
}

template <int MODEL, bool OD = false>
__device__ __forceinline__ double gn_eval_any(const double* zv, const GnMem& W, const GnDims& d, const GnConst& c, const GnPar& q,
                                              int lane, bool derivs, const double* rhov = nullptr) {
    if constexpr (GnModel<MODEL>::PD == 4) return gn_eval_state<MODEL>(zv, W, d, c, q, lane, derivs);
    else return gn_eval<MODEL, OD>(zv, W, d, c, q, lane, derivs, rhov);
}
template <int MODEL, bool OD = false>
__device__ __forceinline__ void gn_jt_any(const double* v, double* out, const GnMem& W, const GnDims& d, const GnConst& c, int lane, bool ycorr = false) {
    if constexpr (GnModel<MODEL>::PD == 4) gn_jt_state<MODEL>(v, out, W, d, c, lane);
    else gn_jt<MODEL, OD>(v, out, W, d, c, lane, ycorr);
}

#ifdef SC_GN_PROF
#define GP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof[i] += (double)(t_ - tlast); tlast = t_; } while (0)
#else
#define GP(i) do { } while (0)
#endif

// NT > 0: compile-time horizon (register Cholesky of order 2 NT); 0: run-time horizon, LDS Cholesky.
// OD: optimal-decay MPC-CBF (position_control/optimal_decay_mpc_cbf.py; oracle/od_mpc_gn.py) for KinematicBicycle2D and Quad2D: two decay
// variables per stage scale the DT-CBF gains of that stage's rows; their 2 x 2 blocks D_k are eliminated per stage in POINT space
// (Psi_k <- Psi_k - C_k D_k^-1 C_k', the right-hand side likewise), so the condensed system keeps its order 2 N; the input term is R u^2;
// no restoration phase (the optimal-decay oracle has none).
template <int MODEL, int NT, bool OD = false>
__global__ __launch_bounds__(64) void mpcgn_kernel(const sc_mpcgn_params p, const GnOd od, const long long B, const int K,
                                                   const void* __restrict__ X, const void* __restrict__ u_prev,
                                                   const void* __restrict__ goal, const void* __restrict__ obs,
                                                   void* __restrict__ u_out, int* __restrict__ status_out,
                                                   int* __restrict__ iters_out, void* __restrict__ z_out, void* __restrict__ rho_out,
                                                   const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    using Mdl = GnModel<MODEL>;
    constexpr int NX = Mdl::NX, NB = Mdl::NB;
    const int lane = threadIdx.x;
    long long prob;
#if SC_CONT_LEVEL >= 1
    if (!ipm::cont_problem(ct, B, prob)) return;                        // mpc_cont.hpp: block index, or an entry of the previous launch's queue
#else
    prob = blockIdx.x;
    if (prob >= B) return;
#endif
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };

    GnDims d;
    d.N = NT > 0 ? NT : p.horizon; d.K = K; d.n = 2 * d.N; d.mc = d.N * K; d.ms = 2 * NB * d.N; d.m = d.mc + d.ms + 2 * d.n;
    d.circles = Mdl::PD == 2 && p.circles_only != 0;         // full-state barriers keep their 4 x 4 Hessians per point
    const int N = d.N, n = d.n, m = d.m;
    const GnMem W = carve_gn<NX, Mdl::NH, Mdl::PD, Mdl::NP>(sm, d, OD);
    GnConst c;
    {
        const double g1 = p.alpha1 + p.alpha2, g2 = p.alpha1 * p.alpha2;
        c.w0 = 1.0 - g1 + g2; c.w1 = g1 - 2.0; c.w2 = 1.0;
        if constexpr (Mdl::PD == 4) { c.w0 = p.alpha1 - 1.0; c.w1 = 1.0; c.w2 = 0.0; }   // d_h + alpha h_k (mpc_cbf.py:312-315)
    }
    c.Rrob = p.robot_radius; c.beta = p.beta; c.circles_only = p.circles_only; c.blo = -p.v_max; c.bhi = p.v_max;
    c.al1 = p.alpha1; c.al2 = p.alpha2; c.ps1 = od.p_sb[0]; c.ps2 = od.p_sb[1]; c.rf1 = od.omega_ref[0]; c.rf2 = od.omega_ref[1];
    GnPar q;
    q.dt = p.dt; q.Lr = p.rear_ax_dist; q.v_min = p.v_min; q.v_max = p.v_max; q.mass = p.mass; q.inertia = p.inertia; q.rad = p.robot_radius;
    if (lane < 6) W.cq[lane] = p.Q[lane];
    if (lane < 2) { W.cq[6 + lane] = p.R[lane]; W.cq[8 + lane] = p.u_lo[lane]; W.cq[10 + lane] = p.u_hi[lane]; }
    for (int i = lane; i < NX; i += 64) { W.xs[i] = ld(X, prob * NX + i); W.xg[i] = i < 2 ? ld(goal, prob * 2 + i) : 0.0; }
    if (lane < 2) W.up[lane] = ld(u_prev, prob * 2 + lane);
    const size_t obase = p.obs_shared ? 0 : (size_t)prob * K * 7;
    for (int e = lane; e < K * 7; e += 64) W.obs[e] = ld(obs, obase + e);
    for (int e = lane; e < NX * n; e += 64) W.Ph[e] = 0.0;                  // Phi_0 = 0
    SC_SYNC();
    // the scalars of the interior-point loop (wave-uniform); a continuation launch loads them with the arrays (mpc_cont.hpp)
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
    if (SC_CONT_LEVEL >= 2 && ct.resume) {
        // the state a previous launch left: [scalars | z | zb | s | lam | obs | tel | rho | rhob]
        const double* a = cst + ipm::CONT_SCALARS;
        ipm::cont_copy(W.z, a, n, lane, 64); a += n;
        ipm::cont_copy(W.zb, a, n, lane, 64); a += n;
        ipm::cont_copy(W.s, a, m, lane, 64); a += m;
        ipm::cont_copy(W.lam, a, m, lane, 64); a += m;
        ipm::cont_copy(W.obs, a, K * 7, lane, 64); a += K * 7;
        if (nel) { ipm::cont_copy(W.tel, a, nel, lane, 64); a += nel; }
        if constexpr (OD) { ipm::cont_copy(W.rho, a, n, lane, 64); a += n; ipm::cont_copy(W.rhob, a, n, lane, 64); }
        it0 = (int)cst[0] + 1; mu = cst[1]; nu_m = cst[2]; delta_last = cst[3]; e_best = cst[4]; n_acc = (int)cst[5];
        resto = cst[6] != 0.0; n_resto = (int)cst[7]; n_small = (int)cst[8]; theta_R = cst[9]; mu_reg = cst[10]; sf = cst[11];
        delta_force = cst[12]; n_retry = (int)cst[13]; theta_ref = cst[14]; n_stall = (int)cst[15];
        SC_SYNC();
    } else {
    if (!c.circles_only) ipm::normalise_obstacle_flags(W.obs, K, lane, 64);
    SC_SYNC();
    for (int i = lane; i < n; i += 64) {                                   // set_initial_guess: u_prev, strictly inside the box
        const double lo = W.cq[8 + (i & 1)], hi = W.cq[10 + (i & 1)], pad = 0.005 * (hi - lo);
        W.z[i] = fmin(fmax(W.up[i & 1], lo + pad), hi - pad);
        if constexpr (OD) { W.rho[i] = (i & 1) ? c.rf2 : c.rf1; W.rhob[i] = W.rho[i]; }   // decay variables start at their references
    }
    SC_SYNC();
    }

    // grad f = sum_k Phi_k' 2 Q (x_k - xg) + r-term; gs = sf grad f
    auto grad_f = [&](double sf) {
        for (int i = lane; i < n; i += 64) {
            double acc = 0.0;
            for (int k = 1; k <= N; ++k) {
#pragma unroll
                for (int s_ = 0; s_ < NX; ++s_)
                    acc += W.Ph[(size_t)(k * NX + s_) * n + i] * (2.0 * W.cq[s_] * (W.xs[k * NX + s_] - W.xg[s_]));
            }
            const double prev = i >= 2 ? W.z[i - 2] : W.up[i];
            acc += 2.0 * W.cq[6 + (i & 1)] * (OD ? W.z[i] : W.z[i] - prev);
            if (!OD && i + 2 < n) acc -= 2.0 * W.cq[6 + (i & 1)] * (W.z[i + 2] - W.z[i]);
            W.gs[i] = sf * acc;
        }
        SC_SYNC();
    };

    if (!(SC_CONT_LEVEL >= 2 && ct.resume)) {
    f = gn_eval_any<MODEL, OD>(W.z, W, d, c, q, lane, true, W.rho);
    if (SC_CONT_LEVEL >= 2 && ct.it_stop < 0) {
        // classify only (mpc_cont.hpp): is a CBF row violated at the initial guess?
        double th0 = 0.0;
        for (int i = lane; i < d.mc; i += 64) th0 += fmax(0.0, -W.g[i]);
        th0 = gsum(th0);
        if (lane == 0) ipm::cont_push(ct, prob, th0 > 0.0);
        return;
    }
    // steep (superellipsoid) barriers: IPOPT-style gradient-based row scaling from the initial guess, then a fresh evaluation
    if (!c.circles_only &&
        ipm::scale_steep_barriers(W.obs, K, W.dh, 3 * N, lane, 64, [](double v) { return gmax_(v); }, [] { SC_SYNC(); }))
        f = gn_eval_any<MODEL, OD>(W.z, W, d, c, q, lane, true, W.rho);
    grad_f(1.0);
    double gmx = 0.0;
    for (int i = lane; i < n; i += 64) gmx = fmax(gmx, fabs(W.gs[i]));
    gmx = gmax_(gmx);
    sf = fmin(1.0, 100.0 / fmax(1e-12, gmx));
    for (int i = lane; i < m; i += 64) { const double s = fmax(W.g[i], 1e-2); W.s[i] = s; W.lam[i] = mu / s; }
    for (int i = lane; i < n; i += 64) W.zb[i] = W.z[i];
    SC_SYNC();
    }

    int status = SC_STATUS_INACCURATE, it = 0;
    const double tau = 0.995;
    const int acc_iter = p.acceptable_iter > 0 ? p.acceptable_iter : 15;
    // feasibility restoration (mpc_ipm_common.hpp; oracle/mpc_cbf.py: solve): wave-uniform state
    const double rho_R = p.resto.rho;
    bool pending = false;
#ifdef SC_GN_PROF
    double prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    for (it = it0; it <= p.max_iter; ++it) {
        GP(11);
        if (SC_CONT_LEVEL >= 3 && cst && it > ct.it_stop) { pending = true; break; }   // the cap of this launch: the solve goes on in the next one
        if (it > 1 || (SC_CONT_LEVEL >= 2 && ct.resume)) f = gn_eval_any<MODEL, OD>(W.z, W, d, c, q, lane, true, W.rho);
        GP(0);
        double theta = 0.0;                                               // l1 violation of the elastic (CBF) rows at z
        for (int i = lane; i < d.mc; i += 64) theta += fmax(0.0, -W.g[i]);
        theta = gsum(theta);
        if (resto && theta <= fmax(n_resto == 1 ? p.resto.kappa * theta_R : 0.0, p.resto.theta_tol)) {
            // enough of the violation is gone (first entry: a tenth of it; later entries run until nothing is left -- the regular
            // phase came back to the same stall): a fresh start of the regular phase at this z with the barrier parameter it left with
            resto = false; mu = mu_reg;
            for (int i = lane; i < m; i += 64) { const double s0 = fmax(W.g[i], 1e-2); W.s[i] = s0; W.lam[i] = mu / s0; }
            for (int i = lane; i < n; i += 64) W.zb[i] = W.z[i];
            nu_m = 10.0; n_acc = 0; e_best = 1e300;
            SC_SYNC();
        }
        // restoration: no objective but zeta/2 |z - z_R|^2 (z_R in W.zb), zeta = sqrt(mu)
        double zeta = resto ? sqrt(mu) : 0.0;
        const double sfe = resto ? 0.0 : sf;
        if (resto) {
            for (int i = lane; i < n; i += 64) W.gs[i] = zeta * (W.z[i] - W.zb[i]);
            SC_SYNC();
        } else {
            grad_f(sf);
        }
        gn_jt_any<MODEL, OD>(W.lam, W.rd, W, d, c, lane);
        GP(1);
        // ---- second derivatives of the dynamics and of step o step, weighted by the costates of the Lagrangian (oracle:
        // evaluate, exact_hessian).  W.y still holds A' lam per point (= -nu).  Leaves two scalars per stage in W.Hk.
        if constexpr (MODEL == SC_MODEL_KINEMATIC_BICYCLE2D || Mdl::PD == 4) Mdl::second_order(W.xs, W.z, W.y, W.cq, W.xg, W.lam + d.mc, W.Hk, N, sfe, q, lane);
        else if constexpr (Mdl::EXACT) Mdl::second_order(W.xs, W.z, W.y, W.cq, W.xg, W.Hk, N, sfe, q, lane);
        SC_SYNC();
        GP(2);
        double e_d = 0.0, e_p = 0.0, e_c0 = 0.0, lmx = 0.0;
        for (int i = lane; i < n; i += 64) { const double r = W.gs[i] - W.rd[i]; W.rd[i] = r; e_d = fmax(e_d, fabs(r)); }
        if constexpr (OD) {
            for (int i = lane; i < n; i += 64) {                         // r_d of rho_{k,i} = sf 2 p_sb (rho - ref) - sum_j lam_kj A_i[kj]
                const int k = i >> 1;
                const double* Ai = (i & 1) ? W.A2 : W.A1;
                double acc = sf * 2.0 * ((i & 1) ? c.ps2 : c.ps1) * (W.rho[i] - ((i & 1) ? c.rf2 : c.rf1));
                for (int j = 0; j < K; ++j) acc -= W.lam[k * K + j] * Ai[k * K + j];
                W.rdr[i] = acc;
                e_d = fmax(e_d, fabs(acc));
            }
        }
        for (int i = lane; i < m; i += 64) {
            const double s = W.s[i], l = W.lam[i];
            double rp = W.g[i] - s;
            if (resto && i < d.mc) { const double t = W.tel[i]; rp += t; e_c0 = fmax(e_c0, fabs(t * (rho_R - l))); }
            e_p = fmax(e_p, fabs(rp)); e_c0 = fmax(e_c0, fabs(s * l)); lmx = fmax(lmx, l);
        }
        e_d = gmax_(e_d); e_p = gmax_(e_p); e_c0 = gmax_(e_c0); lmx = gmax_(lmx);
        const double e_opt = fmax(e_d, fmax(e_p, e_c0));
        GT(0, it); GT(1, e_d); GT(2, e_p); GT(3, e_c0); GT(4, lmx); GT(5, mu); GT(6, theta); GT(7, f); GT(25, resto ? 1.0 : 0.0);
        if (!resto && e_opt < e_best) {
            e_best = e_opt;
            for (int i = lane; i < n; i += 64) { W.zb[i] = W.z[i]; if constexpr (OD) W.rhob[i] = W.rho[i]; }
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
        if (!(e_opt < 1e300)) break;
        bool want_resto = RESTO && !resto && lmx > 1e10;                  // multipliers diverge: locally infeasible
        if (!RESTO && lmx > 1e10) { status = SC_STATUS_INFEASIBLE; break; }
        bool accepted = false, sreset = false;
        double thr_reset = 0.0;
        double alpha = 0.0, ad = 0.0;
        if (!want_resto) {
        const double mu_old = mu;
        for (;;) {
            double e_c = 0.0;
            for (int i = lane; i < m; i += 64) {
                const double l = W.lam[i];
                e_c = fmax(e_c, fabs(W.s[i] * l - mu));
                if (resto && i < d.mc) e_c = fmax(e_c, fabs(W.tel[i] * (rho_R - l) - mu));
            }
            e_c = gmax_(e_c);
            const double e_mu = fmax(e_d, fmax(e_p, e_c));
            if (e_mu <= 10.0 * mu && mu > p.mu_min) mu = fmax(p.mu_min, fmin(0.2 * mu, mu * sqrt(mu)));
            else break;
        }
        if (resto && mu != mu_old) {                                      // zeta = sqrt(mu): the proximity term follows the new mu
            zeta = sqrt(mu);
            for (int i = lane; i < n; i += 64) W.gs[i] = zeta * (W.z[i] - W.zb[i]);
        }
        for (int i = lane; i < m; i += 64) {
            const double s = W.s[i], l = W.lam[i];
            if (resto && i < d.mc) {                                      // elastic row: lam + dl0 and Sigma_eff (ipm::resto_row)
                double rp, ise, vbe;
                ipm::resto_row(W.g[i], s, l, W.tel[i], mu, rho_R, rp, ise, vbe);
                W.vb[i] = l + (mu * ise - vbe);
                W.ds[i] = l * ise;
            } else {
                const double is = rcp_(s), sig = l * is;                  // v_rcp seed + two Newton steps (sc_qp2.hpp)
                W.vb[i] = mu * is - sig * (W.g[i] - s);
                W.ds[i] = sig;                                            // sigma, read below; ds proper is written after the solve
            }
        }
        SC_SYNC();
        if constexpr (OD) {
            // decay blocks of the stages.  Row (k, j) in point space: v = [w0 dh_a; w1 dh_b; dh_c]; d row / d rho_i = A_i; its mixed second
            // derivative  x_i = [(-a_i + a1 a2 rho_other) dh_a; a_i dh_b; 0].  C_k[r][i] = sum_j sig v_r A_i - lam x_i[r] (6 x 2),
            // D_k = sf 2 p_sb + sum_j sig A A' - lam a1 a2 h_a [[0, 1], [1, 0]],  rr_k = -sf 2 p_sb (rho - ref) + sum_j vb A.
            for (int e = lane; e < 17 * N; e += 64) {
                const int k = e / 17, r = e - 17 * k;
                const double r1 = W.rho[2 * k], r2 = W.rho[2 * k + 1];
                double acc = 0.0;
                for (int j = 0; j < K; ++j) {
                    const int row = k * K + j;
                    const double sig = W.ds[row], l = W.lam[row], a1v = W.A1[row], a2v = W.A2[row];
                    if (r < 12) {
                        const int pr = r >> 1, i = r & 1, pp = pr >> 1, dd = pr & 1;    // point-space component pr = 2 p + d, decay variable i
                        const double dhv = W.dh[2 * ((3 * k + pp) * K + j) + dd];
                        const double wv = pp == 0 ? W.w0s[k] : (pp == 1 ? W.w1s[k] : 1.0);
                        const double ai = i == 0 ? c.al1 : c.al2, ro = i == 0 ? r2 : r1;
                        const double xi = pp == 0 ? (-ai + c.al1 * c.al2 * ro) : (pp == 1 ? ai : 0.0);
                        acc += sig * (wv * dhv) * (i == 0 ? a1v : a2v) - l * xi * dhv;
                    } else if (r == 12) acc += sig * a1v * a1v;
                    else if (r == 13) acc += sig * a1v * a2v - l * c.al1 * c.al2 * W.hk[(3 * k) * K + j];
                    else if (r == 14) acc += sig * a2v * a2v;
                    else if (r == 15) acc += W.vb[row] * a1v;
                    else acc += W.vb[row] * a2v;
                }
                if (r < 12) W.Cod[12 * k + r] = acc;
                else if (r < 15) W.pdz[3 * k + (r - 12)] = acc;                  // D11, D12, D22 (pdz: free until the solve)
                else W.rr[2 * k + (r - 15)] = -sf * 2.0 * (r == 15 ? c.ps1 * (r1 - c.rf1) : c.ps2 * (r2 - c.rf2)) + acc;
            }
            SC_SYNC();
            for (int k = lane; k < N; k += 64) {
                // D_k^-1 in eigen form, shifted to positive definite (oracle/od_mpc_cbf.py: block_eig; csrc/mpc_cbf.hip: stage_pass)
                const double d11 = sf * 2.0 * c.ps1 + W.pdz[3 * k], d12 = W.pdz[3 * k + 1], d22 = sf * 2.0 * c.ps2 + W.pdz[3 * k + 2];
                const double tr = d11 + d22, df = d11 - d22, rad = sqrt(df * df + 4.0 * d12 * d12);
                double ls = 0.5 * (tr + rad), lw = (d11 * d22 - d12 * d12) / ls;
                double vx = df >= 0.0 ? df + rad : 2.0 * d12, vy = df >= 0.0 ? 2.0 * d12 : rad - df;
                const double vn = vx * vx + vy * vy;
                if (vn > 0.0) { const double rn = 1.0 / sqrt(vn); vx *= rn; vy *= rn; } else { vx = 1.0; vy = 0.0; }
                const double sh = fmax(0.0, 1e-8 * fmax(1.0, fabs(d11) + fabs(d22)) - lw);
                ls += sh; lw += sh;
                W.Dod[4 * k] = vx; W.Dod[4 * k + 1] = vy; W.Dod[4 * k + 2] = 1.0 / ls; W.Dod[4 * k + 3] = 1.0 / lw;
            }
            SC_SYNC();
        }
        gn_jt_any<MODEL, OD>(W.vb, W.rhs, W, d, c, lane, true);
        for (int i = lane; i < n; i += 64) W.rhs[i] = -W.gs[i] + W.rhs[i];
        GP(3);
        // stage blocks Psi_k (6 x 6 over a_k, b_k, c_k): sum_j sig v v' (v = [w0 dh_a; w1 dh_b; w2 dh_c]) - sum_j lam w_p Hh_p
        if constexpr (Mdl::PD == 4) {
            // 8 x 8 stage blocks over (x_k, y1): sum_j sig v v' (v = [w0 dh(x_k); w1 dh(y1)]) - sum_j lam w_p Hh_p on the diagonal blocks
            for (int e = lane; e < 64 * N; e += 64) {
                const int k = e >> 6, r = (e >> 3) & 7, cc = e & 7, pr = r >> 2, pc = cc >> 2, a = r & 3, b = cc & 3;
                const double wr = pr == 0 ? c.w0 : c.w1, wc = pc == 0 ? c.w0 : c.w1;
                const int lo = a < b ? a : b, hi = a < b ? b : a, t = lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo);
                double acc = 0.0;
                for (int j = 0; j < K; ++j) {
                    const int row = k * K + j, er = (2 * k + pr) * K + j, ec = (2 * k + pc) * K + j;
                    const double l = W.lam[row], sig = W.ds[row];
                    acc += sig * (wr * W.dh[4 * er + a]) * (wc * W.dh[4 * ec + b]);
                    if (pr == pc) acc -= l * wr * W.hh[10 * er + t];
                }
                W.Psi[e] = acc;
            }
        } else
        for (int e = lane; e < 36 * N; e += 64) {
            const int k = e / 36, r = (e - 36 * k) / 6, cc = e - 36 * k - 6 * r, pr = r >> 1, pc = cc >> 1;
            double wr = pr == 0 ? c.w0 : (pr == 1 ? c.w1 : c.w2), wc = pc == 0 ? c.w0 : (pc == 1 ? c.w1 : c.w2);
            if constexpr (OD) {
                wr = pr == 0 ? W.w0s[k] : (pr == 1 ? W.w1s[k] : 1.0);
                wc = pc == 0 ? W.w0s[k] : (pc == 1 ? W.w1s[k] : 1.0);
            }
            double acc = 0.0;
            for (int j = 0; j < K; ++j) {
                const int row = k * K + j, er = (3 * k + pr) * K + j, ec = (3 * k + pc) * K + j;
                const double l = W.lam[row], sig = W.ds[row];
                acc += sig * (wr * W.dh[2 * er + (r & 1)]) * (wc * W.dh[2 * ec + (cc & 1)]);
                if (pr == pc) acc -= l * wr * (d.circles ? ((r & 1) == (cc & 1) ? 2.0 : 0.0) : W.hh[3 * er + (r & 1) + (cc & 1)]);
            }
            if constexpr (OD) {                                             // Schur complement of the stage's decay block: - C D^-1 C'
                const double* Cm = W.Cod + 12 * k;
                const double* Dk = W.Dod + 4 * k;
                const double vx = Dk[0], vy = Dk[1];
                const double csr = Cm[2 * r] * vx + Cm[2 * r + 1] * vy, cwr = -Cm[2 * r] * vy + Cm[2 * r + 1] * vx;
                const double csc = Cm[2 * cc] * vx + Cm[2 * cc + 1] * vy, cwc = -Cm[2 * cc] * vy + Cm[2 * cc + 1] * vx;
                acc -= csr * csc * Dk[2] + cwr * cwc * Dk[3];
            }
            W.Psi[e] = acc;
        }
        GP(4);
        SC_SYNC();
        GP(5);
        if constexpr (Mdl::PD == 4) {
            for (int e = lane; e < 8 * N * n; e += 64) {
                const int row = e / n, col_ = e - row * n, k = row >> 3, r = row & 7;
                double acc = 0.0;
#pragma unroll
                for (int cc = 0; cc < 8; ++cc) acc += W.Psi[64 * k + 8 * r + cc] * W.G[(size_t)(8 * k + cc) * n + col_];
                W.T[e] = acc;
            }
        } else
        for (int e = lane; e < 6 * N * n; e += 64) {                          // T = Psi G
            const int row = e / n, col_ = e - row * n, k = row / 6, r = row - 6 * k;
            double acc = 0.0;
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) acc += W.Psi[36 * k + 6 * r + cc] * W.G[(size_t)(6 * k + cc) * n + col_];
            W.T[e] = acc;
        }
        SC_SYNC();
        for (int e = lane; e < n * (n + 1) / 2; e += 64) {                    // M = sf Hc + G' T + V' H V + speed rows + box
            int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            while ((i + 1) * (i + 2) / 2 <= e) ++i;
            while (i * (i + 1) / 2 > e) --i;
            const int j = e - i * (i + 1) / 2;
            double acc = 0.0;                                                 // Gauss-Newton cost Hessian 2 sum_k Phi_k' Q Phi_k + 2 D' R D
            for (int k = 1; k <= N; ++k) {
#pragma unroll
                for (int s_ = 0; s_ < NX; ++s_)
                    acc += W.cq[s_] * W.Ph[(size_t)(k * NX + s_) * n + i] * W.Ph[(size_t)(k * NX + s_) * n + j];
            }
            acc *= 2.0;
            const double ri = W.cq[6 + (i & 1)];
            if (i == j) acc += 2.0 * ri + ((!OD && i + 2 < n) ? 2.0 * ri : 0.0);   // OD: R u^2, no input-rate coupling
            if (!OD && i == j + 2) acc -= 2.0 * ri;
            acc *= sfe;
            for (int r = 0; r < Mdl::PD * Mdl::NP * N; ++r) acc += W.G[(size_t)r * n + i] * W.T[(size_t)r * n + j];
            if constexpr (NB > 0) {
                for (int k = 1; k <= N; ++k) {
                    const int r0 = d.mc + 2 * (k - 1);
                    acc += (W.ds[r0] + W.ds[r0 + 1]) * W.Ph[(size_t)(k * NX + Mdl::BIDX) * n + i] * W.Ph[(size_t)(k * NX + Mdl::BIDX) * n + j];
                }
            }
            if constexpr (MODEL == SC_MODEL_KINEMATIC_BICYCLE2D || Mdl::PD == 4) {
                for (int k = 0; k < N; ++k) acc += Mdl::hess_term(W.Ph, W.Hk, k, i, j, n);
            } else if constexpr (Mdl::EXACT) {
                // V_k' H_k V_k with H_k = a_k e_t e_t' + b_k (e_t s' + s e_t'): t = the model's angle state, s = the two inputs of stage k
                for (int k = 0; k < N; ++k) {
                    const double ti = W.Ph[(size_t)(k * NX + Mdl::TIDX) * n + i], tj = W.Ph[(size_t)(k * NX + Mdl::TIDX) * n + j];
                    acc += W.Hk[2 * k] * ti * tj + W.Hk[2 * k + 1] * (ti * ((j >> 1) == k ? 1.0 : 0.0) + ((i >> 1) == k ? 1.0 : 0.0) * tj);
                }
            }
            if (i == j) acc += W.ds[d.mc + d.ms + i] + W.ds[d.mc + d.ms + n + i];
            W.M[(size_t)i * n + j] = acc;
            W.M[(size_t)j * n + i] = acc;
        }
        SC_SYNC();
        GP(6);
        double delta = delta_force;                                      // 0 unless a failed restoration step is being retried
        bool ok = false;
        for (int t = 0; t < 40 && !ok; ++t) {
            if constexpr (NT > 0) {
                ok = ipm::chol_reg_solve<2 * NT>(W.M, W.rhs, W.L, W.dz, delta + zeta, lane);   // restoration: + zeta I, the proximity term
            } else {
                for (int r = 0; r < n; ++r)                                 // lower triangle, row stride n | 1 (odd: no LDS bank conflicts)
                    for (int cc = lane; cc <= r; cc += 64) W.L[r * (n | 1) + cc] = W.M[r * n + cc] + (cc == r ? delta + zeta : 0.0);
                SC_SYNC();
                ok = ipm::cholesky_lds(W.L, n, n | 1, lane);
            }
            if (!ok) delta = (delta == 0.0) ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
        }
        GT(8, delta); GT(9, ok ? 1.0 : 0.0);
#ifdef SC_GN_TRACE
        { double sm_ = 0.0, sr_ = 0.0, sd_ = 0.0;
          for (int i = 0; i < n; ++i) { sm_ += W.M[(size_t)i * n + i]; sr_ += W.rhs[i]; sd_ += W.dz[i]; }
          GT(22, sd_); GT(23, sr_); GT(24, sm_); }
#endif
        if (!ok) break;
        if (delta > 0.0) delta_last = delta;
        if constexpr (NT == 0) {
            for (int i = lane; i < n; i += 64) W.dz[i] = W.rhs[i];
            SC_SYNC();
            ipm::chol_solve_lds(W.L, W.dz, n, n | 1, lane);
        }
        GP(7);
        for (int r = lane; r < Mdl::PD * Mdl::NP * N; r += 64) {
            double acc = 0.0;
            for (int i = 0; i < n; ++i) acc += W.G[(size_t)r * n + i] * W.dz[i];
            W.pdz[r] = acc;
        }
        double gdz = 0.0;
        for (int i = lane; i < n; i += 64) gdz += W.gs[i] * W.dz[i];
        SC_SYNC();
        if constexpr (OD) {
            // back-substitution of the decay variables:  d rho_k = D_k^-1 (rr_k - C_k' (G dz)_k)
            for (int k = lane; k < N; k += 64) {
                const double* Cm = W.Cod + 12 * k;
                const double* Dk = W.Dod + 4 * k;
                double t0 = W.rr[2 * k], t1 = W.rr[2 * k + 1];
#pragma unroll
                for (int r = 0; r < 6; ++r) { const double pd = W.pdz[6 * k + r]; t0 -= Cm[2 * r] * pd; t1 -= Cm[2 * r + 1] * pd; }
                const double vx = Dk[0], vy = Dk[1];
                const double ps_ = (vx * t0 + vy * t1) * Dk[2], pw_ = (-vy * t0 + vx * t1) * Dk[3];
                const double dr1 = vx * ps_ - vy * pw_, dr2 = vy * ps_ + vx * pw_;
                W.drho[2 * k] = dr1; W.drho[2 * k + 1] = dr2;
                gdz += sf * 2.0 * (c.ps1 * (W.rho[2 * k] - c.rf1) * dr1 + c.ps2 * (W.rho[2 * k + 1] - c.rf2) * dr2);   // sf grad_rho f . d rho
            }
            SC_SYNC();
        }
        double rs_min = 0.0, rl_min = 0.0, sum_ds_s = 0.0, sum_rp = 0.0, sum_log = 0.0, sum_g = 0.0, sum_t = 0.0, sum_dt = 0.0;
        for (int i = lane; i < m; i += 64) {
            const double s = W.s[i], l = W.lam[i], sig = W.ds[i];
            sum_g += fabs(W.g[i]);
            double jd;
            if (i < d.mc) {
                const int k = i / K, j = i - k * K;
                jd = 0.0;
                if constexpr (Mdl::PD == 4) {
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const int ep = (2 * k + pp) * K + j;
                        double a4 = 0.0;
#pragma unroll
                        for (int t = 0; t < 4; ++t) a4 += W.dh[4 * ep + t] * W.pdz[8 * k + 4 * pp + t];
                        jd += (pp == 0 ? c.w0 : c.w1) * a4;
                    }
                } else
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {
                    const int ep = (3 * k + pp) * K + j;
                    double wv = pp == 0 ? c.w0 : (pp == 1 ? c.w1 : c.w2);
                    if constexpr (OD) wv = pp == 0 ? W.w0s[k] : (pp == 1 ? W.w1s[k] : 1.0);
                    jd += wv * (W.dh[2 * ep] * W.pdz[6 * k + 2 * pp] + W.dh[2 * ep + 1] * W.pdz[6 * k + 2 * pp + 1]);
                }
                if constexpr (OD) jd += W.A1[i] * W.drho[2 * k] + W.A2[i] * W.drho[2 * k + 1];
            } else if (i < d.mc + d.ms) {
                const int r = i - d.mc, k = (r >> 1) + 1;
                double acc = 0.0;
                for (int cc = 0; cc < n; ++cc) acc += W.Ph[(size_t)(k * NX + Mdl::BIDX) * n + cc] * W.dz[cc];
                jd = (r & 1) ? acc : -acc;
            } else if (i < d.mc + d.ms + n) {
                jd = -W.dz[i - d.mc - d.ms];
            } else {
                jd = W.dz[i - d.mc - d.ms - n];
            }
            const double is = rcp_(s);
            double dsi, dl, rp;
            if (resto && i < d.mc) {
                // elastic row: dlam = -Sigma_eff J dz + dl0 (the row's entries again: W.vb lies in the T region), dt from dlam
                const double t = W.tel[i];
                double ise, vbe;
                ipm::resto_row(W.g[i], s, l, t, mu, rho_R, rp, ise, vbe);
                dl = -sig * jd + (mu * ise - vbe);
                const double dt = ipm::resto_dt(l, t, dl, mu, rho_R);
                dsi = jd + dt + rp;
                const double rt = dt * rcp_(t);
                rs_min = fmin(rs_min, rt); rl_min = fmin(rl_min, -dl * rcp_(rho_R - l));
                sum_ds_s += rt; sum_t += t; sum_dt += dt; sum_log += log(t);
            } else {
                rp = W.g[i] - s;
                dsi = jd + rp;
                dl = -sig * dsi - (l - mu * is);
            }
            const double rs = dsi * is, rl = dl * rcp_(l);
            rs_min = fmin(rs_min, rs); rl_min = fmin(rl_min, rl);
            sum_ds_s += rs; sum_rp += fabs(rp); sum_log += log(s);
            W.dlam[i] = dl;
            W.vb[i] = dsi;                                                // ds (W.ds still holds sigma for other lanes' rows)
        }
        SC_SYNC();
        for (int i = lane; i < m; i += 64) W.ds[i] = W.vb[i];
        rs_min = gmin(rs_min); rl_min = gmin(rl_min); sum_ds_s = gsum(sum_ds_s); sum_rp = gsum(sum_rp); sum_log = gsum(sum_log);
        gdz = gsum(gdz); sum_g = gsum(sum_g);
        SC_SYNC();
        const double ap = rs_min < 0.0 ? fmin(1.0, -tau / rs_min) : 1.0;
        ad = rl_min < 0.0 ? fmin(1.0, -tau / rl_min) : 1.0;
        nu_m = fmax(nu_m, 1.1 * lmx);
        double bar0 = sfe * f - mu * sum_log, dbar = gdz - mu * sum_ds_s;
        if (resto) {
            double prox = 0.0;
            for (int i = lane; i < n; i += 64) { const double dzr = W.z[i] - W.zb[i]; prox += dzr * dzr; }
            prox = gsum(prox); sum_t = gsum(sum_t); sum_dt = gsum(sum_dt);
            bar0 = 0.5 * zeta * prox + rho_R * sum_t - mu * sum_log;
            dbar += rho_R * sum_dt;
        }
        // not a descent direction of the merit function (the penalty is below the multipliers of the step): raise the penalty so that
        // the directional derivative is -0.1 nu |r_p|_1  (Nocedal & Wright (18.36))
        if (dbar - nu_m * sum_rp >= 0.0 && sum_rp > 0.0) nu_m = dbar / (0.9 * sum_rp);
        const double phi0 = bar0 + nu_m * sum_rp;
        const double dphi = dbar - nu_m * sum_rp;
        const double noise_rows = 1e-15 * nu_m * sum_g;                  // round-off of far dummy-obstacle rows (oracle: row_noise)
        GT(10, rs_min); GT(11, rl_min); GT(12, sum_ds_s); GT(13, sum_rp); GT(14, sum_log); GT(15, gdz); GT(16, nu_m); GT(17, phi0); GT(18, dphi);
        GP(8);
        alpha = ap;
        sreset = !OD && (resto ? p.resto.slack_reset != 0 : p.slack_reset == 2);
        thr_reset = mu * rcp_(nu_m);
        for (int ls = 0; ls < 12; ++ls) {
            for (int i = lane; i < n; i += 64) {
                W.zt[i] = W.z[i] + alpha * W.dz[i];
                if constexpr (OD) W.rhot[i] = W.rho[i] + alpha * W.drho[i];
            }
            SC_SYNC();
            const double ft = gn_eval_any<MODEL, OD>(W.zt, W, d, c, q, lane, false, W.rhot);
            double srp = 0.0, slog = 0.0, st_ = 0.0, proxt = 0.0;
            for (int i = lane; i < m; i += 64) {
                const double s_lin = W.s[i] + alpha * W.ds[i];
                double tot = W.g[i];
                if (resto && i < d.mc) {
                    const double t = W.tel[i];
                    const double t_t = t + alpha * ipm::resto_dt(W.lam[i], t, W.dlam[i], mu, rho_R);
                    slog += log(t_t); st_ += t_t; tot += t_t;
                }
                // slack reset (oracle/mpc_cbf.py: solve): for fixed z (and t) the merit function is smallest at s = max(g + t, mu / nu);
                // regular phase: P["slack_reset"] = 2 of the bicycles, restoration: sc_resto_params.slack_reset
                const double s_t = (sreset && tot >= thr_reset) ? tot : s_lin;
                slog += log(s_t);
                srp += fabs(tot - s_t);
            }
            slog = gsum(slog); srp = gsum(srp);
            double phit = sfe * ft - mu * slog + nu_m * srp;
            if (resto) {
                for (int i = lane; i < n; i += 64) { const double dzr = W.zt[i] - W.zb[i]; proxt += dzr * dzr; }
                phit = 0.5 * zeta * gsum(proxt) + rho_R * gsum(st_) - mu * slog + nu_m * srp;
            }
            GT(19, alpha); GT(20, phit); GT(26, ft); GT(27, slog); GT(28, srp);
            if (phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * fabs(phi0) + noise_rows) { accepted = true; break; }
            alpha *= 0.5;
        }
        GT(21, accepted ? 1.0 : 0.0);
        GP(9);
        if (!accepted) {
            if (!RESTO) break;
            if (resto) {
                if (n_retry >= p.resto.retry_max) break;
                // the same z again, Levenberg-damped; the retry is an iteration of its own (W.g holds the last trial point's rows)
                ++n_retry; delta_force = fmax(1.0, 100.0 * fmax(delta_force, delta));
                SC_SYNC();
                gn_eval_any<MODEL, OD>(W.z, W, d, c, q, lane, false, W.rho);
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
            // the regular phase cannot continue from z.  Nothing to restore at a feasible point (kinks of step(), round-off at the
            // precision limit) or once the restoration has been entered max_entries times
            if (e_best <= p.acceptable_tol || theta <= p.resto.theta_tol || n_resto >= p.resto.max_entries) break;
            SC_SYNC();
            gn_eval_any<MODEL, OD>(W.z, W, d, c, q, lane, false, W.rho);    // W.g holds the last trial point's rows
            resto = true; ++n_resto; n_small = 0; theta_R = theta; mu_reg = mu;
            delta_force = 0.0; n_retry = 0; theta_ref = theta; n_stall = 0;
            double vmax = 0.0;
            for (int i = lane; i < d.mc; i += 64) vmax = fmax(vmax, -W.g[i]);
            mu = fmax(mu, gmax_(vmax));                                     // IPOPT: mu_R = max(mu, |c|_inf)
            for (int i = lane; i < m; i += 64) {
                // elastic rows start on their central path, the others like at the start of the solve
                const double gi = W.g[i];
                const double s0 = i < d.mc ? ipm::resto_central_slack(gi, mu, rho_R) : fmax(gi, 1e-2);
                if (i < d.mc) W.tel[i] = s0 - gi;
                W.s[i] = s0; W.lam[i] = mu / s0;
            }
            for (int i = lane; i < n; i += 64) W.zb[i] = W.z[i];            // z_R
            nu_m = 10.0; n_acc = 0;
            SC_SYNC();
            continue;
        }
        for (int i = lane; i < n; i += 64) {
            W.z[i] = W.z[i] + alpha * W.dz[i];
            if constexpr (OD) W.rho[i] = W.rho[i] + alpha * W.drho[i];
        }
        delta_force = 0.0; n_retry = 0;
        for (int i = lane; i < m; i += 64) {
            const double s_lin = W.s[i] + alpha * W.ds[i];                     // W.g holds the accepted trial point's rows
            const double l0 = W.lam[i], dl = W.dlam[i];
            double tn = 0.0;
            const bool el = resto && i < d.mc;
            if (el) { const double t = W.tel[i]; tn = t + alpha * ipm::resto_dt(l0, t, dl, mu, rho_R); W.tel[i] = tn; }
            const double tot = W.g[i] + tn;
            const double s = (sreset && tot >= thr_reset) ? tot : s_lin;
            double l = l0 + ad * dl;
            const double mus = mu * rcp_(s);
            l = fmin(fmax(l, 1e-10 * mus), 1e10 * mus);
            if (el) l = ipm::resto_clamp_lam(l, tn, mu, rho_R);
            W.s[i] = s; W.lam[i] = l;
        }
        SC_SYNC();
    }
    if (SC_CONT_LEVEL >= 3 && pending) {
        // hand-over (mpc_cont.hpp); W.g holds the rows of the current z on every path to the top of the loop
        SC_SYNC();
        double th = 0.0;
        for (int i = lane; i < d.mc; i += 64) th += fmax(0.0, -W.g[i]);
        th = gsum(th);
        double* a = cst + ipm::CONT_SCALARS;
        ipm::cont_copy(a, W.z, n, lane, 64); a += n;
        ipm::cont_copy(a, W.zb, n, lane, 64); a += n;
        ipm::cont_copy(a, W.s, m, lane, 64); a += m;
        ipm::cont_copy(a, W.lam, m, lane, 64); a += m;
        ipm::cont_copy(a, W.obs, K * 7, lane, 64); a += K * 7;
        if (nel) { ipm::cont_copy(a, W.tel, nel, lane, 64); a += nel; }
        if constexpr (OD) { ipm::cont_copy(a, W.rho, n, lane, 64); a += n; ipm::cont_copy(a, W.rhob, n, lane, 64); }
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
        for (int i = lane; i < n; i += 64) { W.z[i] = W.zb[i]; if constexpr (OD) W.rho[i] = W.rhob[i]; }
        status = SC_STATUS_OPTIMAL;
    }
    SC_SYNC();
    if constexpr (OD) {
        // optimal decay has no restoration phase: "infeasible" there still means "stopped at an infeasible iterate" (oracle/od_mpc_cbf.py)
        gn_eval_any<MODEL, OD>(W.z, W, d, c, q, lane, false, W.rho);
        if (status != SC_STATUS_OPTIMAL) {
            double g_min = 1e300;
            for (int i = lane; i < m; i += 64) g_min = fmin(g_min, W.g[i]);
            g_min = gmin(g_min);
            if (g_min < -1e-6) status = SC_STATUS_INFEASIBLE;
        }
        if (rho_out) for (int i = lane; i < n; i += 64) st(rho_out, prob * n + i, W.rho[i]);
    }
    if (lane < 2) st(u_out, prob * 2 + lane, W.z[lane]);
    if (lane == 0) {
        status_out[prob] = status;
        if (iters_out) iters_out[prob] = it;
    }
#ifdef SC_GN_PROF
    if (z_out && lane == 0) for (int i = 0; i < 12; ++i) st(z_out, prob * n + i, prof[i]);
#elif defined(SC_GN_TRACE)
#elif defined(SC_GN_DUMP)
    // developer build (tools/exp_gn_dump.py): the whole LDS image of the problem at the end of the solve, z_out sized [B, SC_GN_DUMP]
    if (z_out) { SC_SYNC(); for (int i = lane; i < SC_GN_DUMP; i += 64) ((double*)z_out)[prob * (long long)SC_GN_DUMP + i] = sm[i]; }
#else
    if (z_out) for (int i = lane; i < n; i += 64) st(z_out, prob * n + i, W.z[i]);
#endif
}

template <int MODEL, bool OD = false>
static hipError_t mpcgn_launch_m(const sc_mpcgn_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                                 const void* obs, void* u_out, int* status, int* iters, void* z_out, hipStream_t stream, const ipm::Cont& ct,
                                 const GnOd od = GnOd{{1.0, 1.0}, {0.0, 0.0}}, void* rho_out = nullptr) {
    const size_t lds = mpcgn_lds_doubles(p.horizon, K, GnModel<MODEL>::NX, GnModel<MODEL>::NB, GnModel<MODEL>::PD == 2 && p.circles_only != 0, GnModel<MODEL>::NH,
                                         GnModel<MODEL>::PD, GnModel<MODEL>::NP, OD) * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64), lds, stream, p, od, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, rho_out, ct);
        return hipGetLastError();
    };
    if (p.horizon == 10) return launch(mpcgn_kernel<MODEL, 10, OD>);
    return launch(mpcgn_kernel<MODEL, 0, OD>);
}

}  // namespace

size_t mpcgn_lds_bytes(int model_id, int N, int K, int circles_only) {
    const int nx = model_id == SC_MODEL_QUAD2D ? 6 : 4;
    const bool st = model_id == SC_MODEL_KINEMATIC_BICYCLE2D_C3BF || model_id == SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF;
    const bool kb = model_id == SC_MODEL_KINEMATIC_BICYCLE2D || st;
    return mpcgn_lds_doubles(N, K, nx, kb ? 1 : 0, !st && circles_only != 0, kb ? 10 : 2, st ? 4 : 2, st ? 2 : 3) * sizeof(double);
}

// optimal-decay MPC-CBF on the step()-barrier template: KinematicBicycle2D and Quad2D (optimal_decay_mpc_cbf.py:19)
size_t odmpcgn_lds_bytes(int model_id, int N, int K) {
    const bool kb = model_id == SC_MODEL_KINEMATIC_BICYCLE2D;
    return mpcgn_lds_doubles(N, K, kb ? 4 : 6, kb ? 1 : 0, true, kb ? 10 : 2, 2, 3, true) * sizeof(double);
}
hipError_t odmpcgn_launch(const sc_odmpcgn_params& q, long long B, int K, const void* X, const void* u_prev, const void* goal,
                          const void* obs, void* u_out, void* rho_out, int* status, int* iters, void* z_out, hipStream_t stream,
                          const ipm::Cont& ct0) {
    GnOd od;
    od.omega_ref[0] = q.omega_ref[0]; od.omega_ref[1] = q.omega_ref[1]; od.p_sb[0] = q.p_sb[0]; od.p_sb[1] = q.p_sb[1];
    if (q.mpc.model_id == SC_MODEL_KINEMATIC_BICYCLE2D)
        return mpcgn_launch_m<SC_MODEL_KINEMATIC_BICYCLE2D, true>(q.mpc, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct0, od, rho_out);
    if (q.mpc.model_id == SC_MODEL_QUAD2D)
        return mpcgn_launch_m<SC_MODEL_QUAD2D, true>(q.mpc, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct0, od, rho_out);
    return hipErrorInvalidValue;
}

// doubles of one problem's solver state in a continuation workspace (mpc_cont.hpp; the layout of mpcgn_kernel's hand-over)
size_t mpcgn_state_doubles(int N, int K) {
    const size_t n = 2 * (size_t)N, mc = (size_t)N * K, m = mc + 2 * (size_t)N + 2 * n;     // (at most one bounded state)
    return ipm::CONT_SCALARS + 2 * n + 2 * m + 7 * (size_t)K + mc + 2 * n;
}

hipError_t mpcgn_launch(const sc_mpcgn_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                        const void* obs, void* u_out, int* status, int* iters, void* z_out, hipStream_t stream, const ipm::Cont& ct) {
    switch (p.model_id) {
        case SC_MODEL_DOUBLE_INTEGRATOR2D:
            return mpcgn_launch_m<SC_MODEL_DOUBLE_INTEGRATOR2D>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
        case SC_MODEL_QUAD2D:
            return mpcgn_launch_m<SC_MODEL_QUAD2D>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
        case SC_MODEL_KINEMATIC_BICYCLE2D:
            return mpcgn_launch_m<SC_MODEL_KINEMATIC_BICYCLE2D>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
        case SC_MODEL_KINEMATIC_BICYCLE2D_C3BF:
            return mpcgn_launch_m<SC_MODEL_KINEMATIC_BICYCLE2D_C3BF>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
        case SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF:
            return mpcgn_launch_m<SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace sc
