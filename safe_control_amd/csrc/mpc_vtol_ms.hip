// MPC-CBF for VTOL2D as do-mpc poses it -- MULTIPLE SHOOTING -- solved by IPOPT's filter line-search interior point: ONE NLP PER
// WAVEFRONT, ONE STAGE PER LANE.  Kernel 12 in DESIGN.md; the algorithm is oracle/ms_ipopt.py with KERNEL_PROFILE (linear_solver = riccati,
// max_soc = 0, resto_elastic = "ineq", the stall rule), iterate for iterate.
//
// Why a second VTOL2D kernel.  mpc_vtol_wave.hip solves the CONDENSED problem in z = (u_0 .. u_{N-1}) from the rollout of u_prev.  The
// reference does not (position_control/mpc_cbf.py:162-174,366-369): do-mpc hands IPOPT the states x_0 .. x_N as variables with the
// dynamics as equality rows, and starts every stage's state at x0.  On the reference's own example scene (examples/test_vtol.py) one
// condensed solve fails on a FEASIBLE problem -- the rollout of an aggressive u_prev over 30 unstable stages diverges, the Newton steps
// of the single-shooting problem find no descent -- and the flight is lost; the multiple-shooting solve of the same NLP converges in
// 32 iterations and the flight lands (tools/exp_ms_vtol_flight.py, profiles/r05_ms_vtol_flight*.log).
//
//   variables   lane k <= N owns x_k (6) and, for k < N, u_k (4), the 6 dynamics rows  F(x_k, u_k) - x_{k+1} = 0  with their multipliers
//               and the K rows  d_kj = -cbf_j(x_k, u_k) <= 0  with slack, multiplier and slack-bound multiplier -- all in registers;
//               bounds on theta, x_dot, z_dot (>= only) and the inputs as IPOPT treats variable bounds (log barrier, z_L / z_U)
//   evaluation  NO rollout: every lane evaluates the aero model at its own (x_k, u_k) (second-order forward mode over (theta, x_dot,
//               z_dot): mpc_vtol_solver.hpp: accel<D2>), one evaluation per stage and function call instead of a chain of thirty
//   Newton step the inequality rows of a stage are condensed into its 10 x 10 block, the primal-dual system is an LQ problem with
//               defects and is solved by a Riccati recursion over the augmented state (dx_k, du_{k-1}) (the input-rate penalty couples
//               neighbouring inputs): the backward sweep on v_mfma_f64_16x16x4_f64 with the value function resident in the accumulator
//               registers (riccati_backward), the forward sweep with its recurrence in registers (v_readlane broadcasts); inertia
//               correction = "every 4 x 4 input block positive definite" (Algorithm IC's delta_w ladder); multipliers of the dynamics
//               rows from the value function, lam_k = (P_k xi_k + p_k)_x
//   globalisation  IPOPT's filter (theta = l1 norm of the row residuals, phi = barrier function), switching condition + Armijo,
//               alpha_min -> IPOPT's restoration phase, a mode of the same loop (elastic variables on the CBF rows, own filter, return test
//               against the regular filter, infeasibility certificate; its row state in a global workspace: sc_ipopt_params.resto_workspace
//               -- without one the solve ends SC_STATUS_NEEDS_RESTO and the host class hands it to the condensed kernel); fraction to the
//               boundary tau = max(0.99, 1 - mu), monotone mu, gradient-based scaling, bound push / relaxation, least-square initial
//               multipliers, kappa_sigma, safe slacks; OD = true: the optimal-decay NLP (decay rates = two more inputs of a stage)
//
// LDS per problem (N = 30): stage Jacobians [A | B] (compact), stage blocks, gains, gradient / defect / step vectors, exchange vectors, two
// filters: 40.3 KB, four problems per CU.  Compiled with LLVM's splitting register allocator and therefore guarded (csrc/Makefile: GUARDED;
// tools/check_exec_prologue.py on every link; tests/test_codegen_guard_gpu.py holds it bitwise equal to a build with the non-splitting one).
#include <hip/hip_runtime.h>

#include <utility>

#include "../../include/safe_control_amd.h"
#include "mpc_ipm_common.hpp"
#include "sc_qp2.hpp"
#define SC_VTOL_WITH_C_PARAMS
#define SC_VTOL_RCP(a) sc::rcp_(a)
#define SC_VTOL_SINCOS(a, s, c) sc::sincos_((a), &(s), &(c));
#include "mpc_vtol_solver.hpp"

namespace sc {

using namespace vtol;

namespace msk {

typedef __attribute__((address_space(3))) double ldsd;

#ifndef SC_MS_SOLVE_INLINE
#define SC_MS_SOLVE_INLINE __forceinline__
#endif
#ifdef SC_MS_PROF
#define MPROF_T0_AGAIN _t0 = __builtin_readcyclecounter();
#define MPROF_T0 long long _t0 = __builtin_readcyclecounter();
#define MPROF_ADD(i) { const long long _t1 = __builtin_readcyclecounter(); prof[i] += (double)(_t1 - _t0); _t0 = _t1; }
#else
#define MPROF_T0
#define MPROF_T0_AGAIN
#define MPROF_ADD(i)
#endif

constexpr int KS_MAX = 16;
constexpr int NFILT = 24;                       // filter entries kept (the filter is cleared with every decrease of mu)
constexpr int ABS = 21;                         // [A | B] of a stage, compact: rows 3..5 x columns (2, 3, 4 | 6..9); rows 0..2 are e_i + dt e_{i+3}, A[5][5] = 1
constexpr int TRACE_W = 8;

struct Lds {
    int OB, AB, H, KG, G, C, DX, DU, LAM, XS, US, YS, Pc, Pn, T, QU, Quu, pc, pn, FP, FT, FP2, FT2, SC, Y0, total;
    __host__ __device__ explicit Lds(int N) {
        int o = 0;
        auto take = [&](int c) { int r = o; o += c; return r; };
        OB = take(3 * KS_MAX); AB = take(N * ABS); H = take((N + 1) * 55); KG = take(N * 44); G = take((N + 1) * 10);
        // Slots whose lifetimes do not overlap share storage (39.9 KB per problem for N = 30: four problems per CU):
        //   the defects C are consumed by the forward sweep as it writes the step dx in their place;
        //   YS (multipliers as the neighbours see them: written and read at the start of an evaluation) / LAM (costates = multiplier steps:
        //   written after the recursion, read at the update);
        //   the recursion's workspace Pc | Pn | T lies over the exchange vectors XS | US, which an evaluation rewrites before it reads them.
        DX = take((N + 1) * 6); C = DX; DU = take(N * 4); LAM = take((N + 2) * 6); YS = LAM;
        const int ex = (N + 2) * 10 < 310 ? 310 : (N + 2) * 10;
        XS = take(ex); US = XS + (N + 2) * 6; Pc = XS; Pn = XS + 100; T = XS + 200;
        QU = take(44); Quu = take(16); pc = take(10); pn = take(10);
        FP = take(NFILT); FT = take(NFILT); FP2 = take(NFILT); FT2 = take(NFILT); SC = take(6 + KS_MAX); Y0 = take(6);      // (FP2 / FT2: the restoration's own filter)
        total = o;
    }
};

size_t lds_bytes(int horizon) { return (size_t)Lds(horizon).total * sizeof(double); }

// entry (i, c) of the 6 x 10 block [A | B] from its compact form (ABS); ci = compact column of c (ab_col), or -1
__device__ __forceinline__ int ab_col(int c) { return (c >= 2 && c <= 4) ? c - 2 : (c >= 6 ? c - 3 : -1); }
template <int I>
__device__ __forceinline__ double ab_at(const ldsd* A3, int c, int ci, double dt) {
    if constexpr (I < 3) return c == I ? 1.0 : (c == I + 3 ? dt : 0.0);
    else return ci >= 0 ? A3[(I - 3) * 7 + ci] : ((I == 5 && c == 5) ? 1.0 : 0.0);
}
template <typename F, int... Is>
__device__ __forceinline__ void for6(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
#define SC_FOR6(body) for6([&](auto I_) { constexpr int I = decltype(I_)::value; body }, std::make_integer_sequence<int, 6>{})

// ---- Riccati recursion with defects (oracle/ms_ipopt.py: _riccati_backward / _riccati_solve, hard dynamics) ----------------------------
// LDS in: AB[k] (compact [A | B]), H[k] (upper-packed 10 x 10 over (x_k, u_k); H[N]: its x block), G[k] (gradient, 10), C[k + 1] (defect of the
// dynamics of stage k), C[0] = dx_0;  cpl[j] = 2 df R_j: the (u_{k-1}, u_k) cross term, -cpl on the (v, u) entries of stage k >= 1.
// Out: KG[k] = gains (K: 4 x 10 over (dx_k, du_{k-1}), then kff: 4); false (wave-uniform) when an input block is not positive definite.
//
// One stage is three small matrix products, on v_mfma_f64_16x16x4_f64 with the value function living in the accumulator registers from one
// stage to the next.  Tile index t of the 16 x 16 tiles:  0..5 = x,  6..9 = v (= u_{k-1}),  10..13 = u,  14 = the affine column (defect /
// gradient / p),  15 unused.   Gt = [[A, 0, B, c], [0, 0, I, 0]] (10 x 15: xi+ = Gt (xi, u, 1)),  S = stage block (H over (x, u), -cpl on
// (v_i, u_i), gradient in column 14):
//     T = P Gt (+ p on column 14),   Q = S + Gt' T,   Quu = L L',  Y = L^-1 Q[u, :],   P' = Q - Y' Y   (column 14 of P' = the new p)
// Operand layout (MI355X guide, "f64 MFMA"): lane l feeds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; result register r of lane l is
// D[row = (l >> 4) + 4 r][col = l & 15] -- so register s of a result IS the B operand of k-step s of the next product (B[4 s + g][c]), and,
// for a symmetric result, the A operand as well: nothing moves between the three products.  Only the four u rows of Q cross lanes (through
// LDS: every lane factors Quu and solves for its own column), and the gains go to LDS for the forward sweep.
// (Before: entry-per-lane loops over LDS copies of P, T, Q: ~1000 instructions per stage for a lone wave, 40 % of an iteration.)
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ __attribute__((noinline)) bool riccati_backward(ldsd* lds, const Lds L, const int N, const int lane, const double dt, const double c0,
                                                           const double c1, const double c2, const double c3) {
    ldsd* QX = lds + L.QU;                                                 // 4 x 16 exchange: the u rows of Q  (QU | Quu | pc | pn: 80 doubles)
    const int c = lane & 15, g = lane >> 4;
    // tile index -> index in the (x, u) ordering of H / G (-1: the v block, the affine column, unused)
    auto hx = [](int t) { return t < 6 ? t : ((t >= 10 && t < 14) ? t - 4 : -1); };
    const int hc = hx(c);
    // S[g + 4 r][c], r = 0..3: an entry of H, of the gradient (column 14), or the cross-term coefficient
    // (every load is unconditional -- a lane without an entry reads the first one and drops it: one LDS round trip for the seven operands
    // instead of one per divergent branch): address of stage k = so + k * ss, sl = the lane does take the loaded value
    int so[4], ss[4];
    bool sl[4];
    double cp[4];
    const double cpl[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = g + 4 * r, hr = hx(row);
        sl[r] = true;
        if (hr >= 0 && hc >= 0) { so[r] = L.H + sym(hr, hc); ss[r] = 55; }
        else if (hr >= 0 && c == 14) { so[r] = L.G + hr; ss[r] = 10; }
        else { so[r] = L.H; ss[r] = 55; sl[r] = false; }
        cp[r] = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) if ((row == 6 + q && c == 10 + q) || (row == 10 + q && c == 6 + q)) cp[r] = -cpl[q];
    }
    // Gt[4 s + g][c], s = 0..2: a constant, or an entry of the stage's compact [A | B] (offset into AB[k]), or of its defect (offset into C[k + 1])
    double gk[3];
    int go[3], gs[3];
    bool gl[3];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
        const int kk = 4 * s_ + g;
        gk[s_] = 0.0; go[s_] = L.AB; gs[s_] = ABS; gl[s_] = false;
        if (kk < 3) gk[s_] = c == kk ? 1.0 : (c == kk + 3 ? dt : 0.0);
        else if (kk < 6) {
            const int ci = (c >= 2 && c <= 4) ? c - 2 : ((c >= 10 && c < 14) ? c - 7 : -1);      // compact column of tile column c
            if (ci >= 0) { go[s_] = L.AB + (kk - 3) * 7 + ci; gl[s_] = true; }
            else if (kk == 5 && c == 5) gk[s_] = 1.0;
        } else if (kk < 10) gk[s_] = c == kk + 4 ? 1.0 : 0.0;                                    // v+ = u
        if (kk < 6 && c == 14) { gk[s_] = 0.0; go[s_] = L.C + 6 + kk; gs[s_] = 6; gl[s_] = true; }     // defect of stage k: C[k + 1]
    }
    // value function of the last stage: the x block of H[N], gradient in column 14
    d4_t Pv;
    {
        const ldsd* HN = lds + L.H + N * 55; const ldsd* GN = lds + L.G + N * 10;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = g + 4 * r;
            double v = 0.0;
            if (row < NX && c < NX) v = HN[sym(row, c)];
            else if (row < NX && c == 14) v = GN[row];
            Pv[r] = v;
        }
    }
    for (int kk = N - 1; kk >= 0; --kk) {
        double G_[3], ld[7];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) ld[s_] = lds[go[s_] + kk * gs[s_]];
#pragma unroll
        for (int r = 0; r < 4; ++r) ld[3 + r] = lds[so[r] + kk * ss[r]];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) G_[s_] = gl[s_] ? ld[s_] : gk[s_];
        d4_t Sv;
#pragma unroll
        for (int r = 0; r < 4; ++r) Sv[r] = sl[r] ? ld[3 + r] : (kk >= 1 ? cp[r] : 0.0);
        // T = P Gt: A = P (symmetric: registers 0..2 of the previous result; its rows 10, 11 -- the u rows -- do not belong to P)
        d4_t Tv = {0.0, 0.0, 0.0, 0.0};
        Tv = __builtin_amdgcn_mfma_f64_16x16x4f64(Pv[0], G_[0], Tv, 0, 0, 0);
        Tv = __builtin_amdgcn_mfma_f64_16x16x4f64(Pv[1], G_[1], Tv, 0, 0, 0);
        Tv = __builtin_amdgcn_mfma_f64_16x16x4f64(g < 2 ? Pv[2] : 0.0, G_[2], Tv, 0, 0, 0);
        if (c == 14) {                                                      // + p (rows 0..9 of column 14)
            Tv[0] += Pv[0]; Tv[1] += Pv[1];
            if (g < 2) Tv[2] += Pv[2];
        }
        // Q = S + Gt' T
        d4_t Qv = Sv;
        Qv = __builtin_amdgcn_mfma_f64_16x16x4f64(G_[0], Tv[0], Qv, 0, 0, 0);
        Qv = __builtin_amdgcn_mfma_f64_16x16x4f64(G_[1], Tv[1], Qv, 0, 0, 0);
        Qv = __builtin_amdgcn_mfma_f64_16x16x4f64(G_[2], g < 2 ? Tv[2] : 0.0, Qv, 0, 0, 0);
        // rows 10..13 of Q to every lane: row 10 = (g 2, r 2), 11 = (g 3, r 2), 12 = (g 0, r 3), 13 = (g 1, r 3)
        __syncthreads();
        QX[((g + 2) & 3) * 16 + c] = g >= 2 ? Qv[2] : Qv[3];
        __syncthreads();
        {
            const double q00 = QX[10], q10 = QX[16 + 10], q11 = QX[16 + 11], q20 = QX[32 + 10], q21 = QX[32 + 11], q22 = QX[32 + 12], q30 = QX[48 + 10],
                         q31 = QX[48 + 11], q32 = QX[48 + 12], q33 = QX[48 + 13];
            const bool rhs = c < 10 || c == 14;                             // columns over (x, v) and the affine one; the others carry zeros
            double b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = rhs ? QX[i * 16 + c] : 0.0;
            const double r0 = rsqrt_(q00), l10 = q10 * r0, l20 = q20 * r0, l30 = q30 * r0;
            const double d1 = q11 - l10 * l10, r1 = rsqrt_(d1), l21 = (q21 - l20 * l10) * r1, l31 = (q31 - l30 * l10) * r1;
            const double d2 = q22 - l20 * l20 - l21 * l21, r2 = rsqrt_(d2), l32 = (q32 - l30 * l20 - l31 * l21) * r2;
            const double d3 = q33 - l30 * l30 - l31 * l31 - l32 * l32, r3 = rsqrt_(d3);
            if (!(q00 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0) || !(d3 > 0.0)) return false;     // wave-uniform: every lane factors the same block
            const double y0 = b[0] * r0, y1 = (b[1] - l10 * y0) * r1, y2 = (b[2] - l20 * y0 - l21 * y1) * r2, y3 = (b[3] - l30 * y0 - l31 * y1 - l32 * y2) * r3;
            if (rhs && g == 0) {                                            // gains K = -L^-T Y (columns over (x, v)), kff (column 14)
                const double x3 = y3 * r3, x2 = (y2 - l32 * x3) * r2, x1 = (y1 - l21 * x2 - l31 * x3) * r1, x0_ = (y0 - l10 * x1 - l20 * x2 - l30 * x3) * r0;
                ldsd* dst = lds + L.KG + kk * 44 + (c < NV ? c : 40);
                const int st = c < NV ? 10 : 1;
                dst[0] = -x0_; dst[st] = -x1; dst[2 * st] = -x2; dst[3 * st] = -x3;
            }
            // P' = Q - Y' Y  (A[i = c][k = g] = -Y[g][c], B[k = g][j = c] = Y[g][c])
            const double yg = g == 0 ? y0 : (g == 1 ? y1 : (g == 2 ? y2 : y3));
            Pv = __builtin_amdgcn_mfma_f64_16x16x4f64(-yg, yg, Qv, 0, 0, 0);
        }
        // the x rows of this stage's value function go where its H block was (consumed above): finish_step reads the multipliers of the dynamics
        // rows off them, lam_k = (P_k xi_k + p_k)_x, instead of running the costate recursion lam_k = h_k + A_k' lam_{k+1} (which multiplies the
        // rounding of every h_j, j > k, by the transition matrices of an unstable airframe; same iterates on the parity tests, one barrier-free sweep
        // less).  Layout (51 of the 55 doubles): x-x upper triangle (21), x-v block (24), linear term (6).
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = g + 4 * r;
            int idx = -1;
            if (row < NX) {
                if (c < NX) { if (c >= row) idx = row * NX - row * (row - 1) / 2 + (c - row); }
                else if (c < NV) idx = 21 + row * 4 + (c - NX);
                else if (c == 14) idx = 45 + row;
            }
            if (idx >= 0) lds[L.H + kk * 55 + idx] = Pv[r];
        }
    }
    __syncthreads();
    return true;
}

// forward LQ rollout: du_k = K (dx_k, du_{k-1}) + kff, dx_{k+1} = [A | B] (dx_k, du_k) + c_{k+1}.  Lanes 0..5 carry dx_k and lanes 0..3 du_{k-1} in
// REGISTERS; the ten numbers a stage needs from its predecessor are v_readlane broadcasts, so the recurrence has no LDS round trip and no barrier
// inside the loop (before: two barriers and ~25 dependent LDS loads per stage, 1.2 k cycles; the loads left -- gains, [A | B] rows, defects -- do
// not depend on the recurrence).  The steps go to LDS (DX over the defects C, which a lane reads before it writes the same slot; DU) on the way.
__device__ __attribute__((noinline)) void riccati_forward(ldsd* lds, const Lds L, const int N, const int lane, const double dt) {
    const int li = lane < NU ? lane : 0, lx = lane < NX ? lane : 0, lr = lx < 3 ? 0 : lx - 3;
    double xr = lds[L.C + lx], vr = 0.0;                                   // dx_0 = C[0]; du_{-1} = 0
    for (int kk = 0; kk < N; ++kk) {
        const ldsd* A3 = lds + L.AB + kk * ABS + lr * 7; const ldsd* KK = lds + L.KG + kk * 44 + li * 10;
        double kr[10], ar[7];
#pragma unroll
        for (int c = 0; c < 10; ++c) kr[c] = KK[c];
        const double kf = lds[L.KG + kk * 44 + 40 + li], cn = lds[L.C + (kk + 1) * 6 + lx];
#pragma unroll
        for (int t = 0; t < 7; ++t) ar[t] = A3[t];
        double d_[NX], v_[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) d_[i] = ipm::row_value(xr, i);
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) v_[jj] = ipm::row_value(vr, jj);
        const double du = (kf + kr[0] * d_[0] + kr[1] * d_[1]) + (kr[2] * d_[2] + kr[3] * d_[3] + kr[4] * d_[4]) + (kr[5] * d_[5] + kr[6] * v_[0] + kr[7] * v_[1]) +
                          (kr[8] * v_[2] + kr[9] * v_[3]);
        if (lane < NU) lds[L.DU + kk * NU + lane] = du;
        double u_[NU];
#pragma unroll
        for (int jj = 0; jj < NU; ++jj) u_[jj] = ipm::row_value(du, jj);
        // row lx of [A | B] (dx_k, du_k) + c: rows 0..2: dx_i + dt dx_{i+3}; rows 3..5: the stored entries (+ dx_5 for row 5)
        const double up3 = lx == 0 ? d_[3] : (lx == 1 ? d_[4] : d_[5]);
        const double lo = cn + xr + dt * up3;
        const double hi = (cn + ar[0] * d_[2] + ar[1] * d_[3]) + (ar[2] * d_[4] + ar[3] * u_[0] + ar[4] * u_[1]) + (ar[5] * u_[2] + ar[6] * u_[3]) + (lx == 5 ? d_[5] : 0.0);
        const double xn = lx < 3 ? lo : hi;
        if (lane < NX) lds[L.DX + (kk + 1) * 6 + lane] = xn;
        if (kk == 0 && lane < NX) lds[L.DX + lane] = xr;
        xr = xn; vr = du;
    }
    __syncthreads();
}

constexpr double EPSD = 2.220446049250313e-16;
constexpr double INF_ = 1e300;

__device__ __forceinline__ bool cmp_le(double lhs, double rhs, double bas) { return lhs - rhs <= 10.0 * EPSD * fabs(bas); }

// OD: OptimalDecayMPCCBF (optimal_decay_mpc_cbf.py:123-124,173-186,288-296; oracle/ms_ipopt.py: vtol_od_model) -- the two decay rates of a
// stage are two more (free) inputs of that stage: they scale the gains of its rows, cost p_sb (omega - ref)^2, input term R u^2 (no coupling of
// neighbouring inputs).  Their 2 x 2 block is eliminated from the stage block inside the 8-dimensional span (V | rho) before it is expanded.
template <int KS, bool OD = false>
struct Wave {
    const Params& P;
    const sc_ipopt_params& O;
    ldsd* lds;
    const Lds L;
    const int lane, N, K;
    const bool act, stg;               // lane owns a state (k <= N) / a stage with inputs and rows (k < N)
    const int k;
    double x0[NX], uprev[NU], xg[2];
    double w0, w1, w2;
    // iterate: registers hold what a lane alone touches; the vectors the neighbours / the recursion read live in LDS (XS, US, YS, DX, DU, LAM),
    // wave-uniform data (row scales, multipliers of the initial-state rows) in LDS as well
    double x[NX], u[NU], yc[NX];
    double xbL[3], xbU[2], ubL[NU], ubU[NU];              // (relaxed, adjustable) bounds: x idx 2, 3, 4 lower / 2, 3 upper; inputs
    double zxL[3], zxU[2], zuL[NU], zuU[NU];
    double s[KS], yd[KS], vU[KS], sU[KS];
    double rho[2], drho[2], odT[6][2], odt[2];           // OD: decay rates, their step; (D^-1 M_rho,v)' and D^-1 g_rho of the last Newton system
    double df;
    double rc[NX], dv[KS];                               // scaled residuals of my dynamics rows, scaled row values (last evaluation)
    double ds[KS], dyd[KS], dvU[KS], dzxL[3], dzxU[2], dzuL[NU], dzuU[NU];
    int nfilt;
    bool od_bad = false;
    double dw_last, last_dw;
    // Restoration phase (section 3.3 of the paper; oracle/ms_ipopt.py: _Resto with resto_elastic = "ineq"): the same algorithm on
    //     min  rho_R sum (n + p) + zeta / 2 |D_R (w - w_R)|^2    s.t.  dynamics rows (hard),  -dgd cbf_j - s_j + n_j - p_j = 0,  n, p >= 0
    // Its row state (n, p, their multipliers, the four steps) and the reference point live in a global workspace, rw (64 lanes x slot), so that
    // the regular iteration keeps its registers: rw == nullptr -> no restoration phase (SC_STATUS_NEEDS_RESTO).
    double* rw = nullptr;
    bool rs = false;                   // inside the restoration phase (wave-uniform)
    double zeta = 0.0;
    int fpo, fto;                      // LDS offsets of the filter in use
    static constexpr int R_N = 0, R_P = KS, R_ZN = 2 * KS, R_ZP = 3 * KS, R_DN = 4 * KS, R_DP = 5 * KS, R_DZN = 6 * KS, R_DZP = 7 * KS, R_XR = 8 * KS, R_SLOTS = 8 * KS + 12;
    __device__ __forceinline__ double& RW(int slot) const { return rw[slot * 64 + lane]; }
    __device__ __forceinline__ double dr2(int i) const { const double a = fabs(RW(R_XR + i)); return a > 1.0 ? 1.0 / (a * a) : 1.0; }      // D_R^2 = 1 / max(1, |w_R|)^2
#ifdef SC_MS_PROF
    double prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};            // eval2 (errors), errors + mu, eval2 (build), riccati backward, forward, finish_step, line search, update
#endif

    __device__ __forceinline__ Wave(const Params& P_, const sc_ipopt_params& O_, ldsd* lds_)
        : P(P_), O(O_), lds(lds_), L(P_.N), lane(threadIdx.x), N(P_.N), K(P_.K), act((int)threadIdx.x <= P_.N), stg((int)threadIdx.x < P_.N),
          k((int)threadIdx.x <= P_.N ? (int)threadIdx.x : 0) {
        const double g1 = P.alpha1 + P.alpha2, g2 = P.alpha1 * P.alpha2;
        w0 = 1.0 - g1 + g2; w1 = g1 - 2.0; w2 = 1.0;
        nfilt = 0; dw_last = 0.0; last_dw = 0.0; fpo = L.FP; fto = L.FT;
    }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ void od_weights(double r1, double r2) {       // stage weights of the rows for decay rates (r1, r2)
        const double sk = P.alpha1 * r1 + P.alpha2 * r2, qk = P.alpha1 * P.alpha2 * r1 * r2;
        w0 = 1.0 - sk + qk; w1 = sk - 2.0;
    }
    __device__ __forceinline__ double dgc(int i) const { return lds[L.SC + i]; }
    __device__ __forceinline__ double dgd(int j) const { return lds[L.SC + 6 + j]; }

    // ---- level 0: my stage's F(x, u), row values (unscaled cbf), cost share ----------------------------------------------------------
    __device__ __forceinline__ void points(const double* xs, const double a0, const double a1, double pt[3][2]) const {
        const double dt = P.dt;
        pt[0][0] = xs[0]; pt[0][1] = xs[1];
        pt[1][0] = xs[0] + dt * xs[3]; pt[1][1] = xs[1] + dt * xs[4];
        pt[2][0] = pt[1][0] + dt * (xs[3] + dt * a0); pt[2][1] = pt[1][1] + dt * (xs[4] + dt * a1);
    }
    __device__ __forceinline__ double cbf_value(const double pt[3][2], int j) const {
        const double cx = lds[L.OB + 3 * j], cz = lds[L.OB + 3 * j + 1], d = P.radius + lds[L.OB + 3 * j + 2], off = P.beta * d * d;
        double hv[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) { const double ex = pt[p][0] - cx, ez = pt[p][1] - cz; hv[p] = ex * ex + ez * ez - off; }
        return w0 * hv[0] + w1 * hv[1] + w2 * hv[2];
    }
    // objective share of lane k: l(x_k) (+ m(x_N)) + R (u_k - u_{k-1})^2; um = u_{k-1}
    __device__ __forceinline__ double cost_share(const double* xs, const double* us, const double* um, const double* rs_) const {
        double f = 0.0;
        if (rs) {                                                          // restoration: sum D_R^2 (w - w_R)^2 over my variables (x zeta / 2 in barrier())
            if (act) {
#pragma unroll
                for (int i = 0; i < NX; ++i) { const double d = xs[i] - RW(R_XR + i); f += dr2(i) * d * d; }
            }
            if (stg) {
#pragma unroll
                for (int j = 0; j < NU; ++j) { const double d = us[j] - RW(R_XR + NX + j); f += dr2(NX + j) * d * d; }
                if constexpr (OD) { const double d0 = rs_[0] - RW(R_XR + 10), d1 = rs_[1] - RW(R_XR + 11); f += dr2(10) * d0 * d0 + dr2(11) * d1 * d1; }
            }
            return f;
        }
        if (act) {
            const double e0 = xs[0] - xg[0], e1 = xs[1] - xg[1];
            f = P.Q[0] * e0 * e0 + P.Q[1] * e1 * e1 + P.Q[2] * xs[2] * xs[2] + P.Q[3] * xs[3] * xs[3] + P.Q[4] * xs[4] * xs[4] + P.Q[5] * xs[5] * xs[5];
        }
        if (stg) {
#pragma unroll
            for (int j = 0; j < NU; ++j) { const double d = OD ? us[j] : us[j] - um[j]; f += P.R[j] * d * d; }
            if constexpr (OD) f += P.ps1 * (rs_[0] - P.rf1) * (rs_[0] - P.rf1) + P.ps2 * (rs_[1] - P.rf2) * (rs_[1] - P.rf2);
        }
        return f;
    }
    // trial evaluation at (xs, us, ss): theta (l1 residual of the scaled rows) and the unscaled objective; xs of every lane goes through XS
    // a_np: step length applied to (n, p) inside the restoration; pmax (optional): largest residual
    __device__ __forceinline__ void eval0(const double* xs, const double* us, const double* ss, const double* rs_, double& theta, double& fsum, double a_np = 0.0,
                                          double* pmax = nullptr) {
        if constexpr (OD) od_weights(rs_[0], rs_[1]);
        sync();
        if (act) {
#pragma unroll
            for (int i = 0; i < NX; ++i) lds[L.XS + k * 6 + i] = xs[i];
        }
        if (stg) {
#pragma unroll
            for (int j = 0; j < NU; ++j) lds[L.US + (k + 1) * 4 + j] = us[j];
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j) lds[L.US + j] = uprev[j];
        }
        sync();
        double th = 0.0, pm = 0.0;
        double um[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) um[j] = lds[L.US + k * 4 + j];
        if (stg) {
            double acc[3], gc[4][3], xn[NX];
            accel<double>(P, xs[2], xs[3], xs[4], us, acc, gc);
            xn[0] = xs[0] + P.dt * xs[3]; xn[1] = xs[1] + P.dt * xs[4]; xn[2] = xs[2] + P.dt * xs[5];
            xn[3] = xs[3] + P.dt * acc[0]; xn[4] = xs[4] + P.dt * acc[1]; xn[5] = xs[5] + P.dt * acc[2];
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double r = fabs(dgc(i) * (xn[i] - lds[L.XS + (k + 1) * 6 + i])); th += r; pm = fmax(pm, r); }
            double pt[3][2];
            points(xs, acc[0], acc[1], pt);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j < K) {
                    double r = -dgd(j) * cbf_value(pt, j) - ss[j];
                    if (rs) r += (RW(R_N + j) + a_np * RW(R_DN + j)) - (RW(R_P + j) + a_np * RW(R_DP + j));
                    th += fabs(r); pm = fmax(pm, fabs(r));
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double r = fabs(xs[i] - x0[i]); th += r; pm = fmax(pm, r); }
        }
        theta = ipm::wsum(th);
        if (pmax) *pmax = ipm::wmax(pm);
        fsum = ipm::wsum(cost_share(xs, us, um, rs_));
    }

    // ---- barrier function (scaled objective + log barrier of every bound + damping of the one-sided ones) ------------------------------
    __device__ __forceinline__ double barrier(double fsum, const double* xs, const double* us, const double* ss, double mu, double a_np = 0.0) const {
        double v = 0.0;
        bool ok = true;
        if (act) {
            const double sl[5] = {xs[2] - xbL[0], xs[3] - xbL[1], xs[4] - xbL[2], xbU[0] - xs[2], xbU[1] - xs[3]};
            double pr = 1.0;
#pragma unroll
            for (int i = 0; i < 5; ++i) { if (!(sl[i] > 0.0)) ok = false; pr *= sl[i]; }
            v -= mu * log(pr);
            v += O.kappa_d * mu * sl[2];
        }
        if (stg) {
            double pr = 1.0;
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const double a = us[j] - ubL[j], b = ubU[j] - us[j];
                if (!(a > 0.0) || !(b > 0.0)) ok = false;
                pr *= a * b;
            }
            v -= mu * log(pr);
            double ps = 1.0, sa = 0.0;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j < K) {
                    const double a = sU[j] - ss[j];
                    if (!(a > 0.0)) ok = false;
                    sa += a;
                    // (slacks of far rows are ~1e3 .. 1e6: the product of eight stays far inside the double range; two logs for sixteen)
                    if (j == 8) { v -= mu * log(ps); ps = 1.0; }
                    ps *= a;
                }
            }
            v -= mu * log(ps);
            v += O.kappa_d * mu * sa;
            if (rs) {                                                       // n, p >= 0: linear term rho_R, log barrier, damping (one-sided)
                double sn = 0.0;
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    if (j < K) {
                        const double nt = RW(R_N + j) + a_np * RW(R_DN + j), pt_ = RW(R_P + j) + a_np * RW(R_DP + j);
                        if (!(nt > 0.0) || !(pt_ > 0.0)) ok = false;
                        sn += nt + pt_;
                        v -= mu * log(nt * pt_);
                    }
                }
                v += (O.resto_penalty_parameter + O.kappa_d * mu) * sn;
            }
        }
        const double bad = ipm::wmax(ok ? 0.0 : 1.0);
        if (bad > 0.0) return INF_;
        return (rs ? 0.5 * zeta : df) * fsum + ipm::wsum(v);
    }
    // IPOPT's CalculateSafeSlack: a slack below eps min(1, mu) is raised to eps^(3/4) max(1, |bound|) by moving the bound
    __device__ __forceinline__ void safe1(double v, double& lo, bool lower, double s_min, double move) const {
        if (lower) { if (v - lo < s_min) lo = v - fmax(v - lo, move * fmax(1.0, fabs(lo))); }
        else { if (lo - v < s_min) lo = v + fmax(lo - v, move * fmax(1.0, fabs(lo))); }
    }
    __device__ __forceinline__ void safe_slacks(const double* xs, const double* us, const double* ss, double mu) {
        const double s_min = EPSD * fmin(1.0, mu), move = 1.8189894035458565e-12;     // eps^(3/4)
        if (act) {
            safe1(xs[2], xbL[0], true, s_min, move); safe1(xs[3], xbL[1], true, s_min, move); safe1(xs[4], xbL[2], true, s_min, move);
            safe1(xs[2], xbU[0], false, s_min, move); safe1(xs[3], xbU[1], false, s_min, move);
        }
        if (stg) {
#pragma unroll
            for (int j = 0; j < NU; ++j) { safe1(us[j], ubL[j], true, s_min, move); safe1(us[j], ubU[j], false, s_min, move); }
#pragma unroll
            for (int j = 0; j < KS; ++j) if (j < K) safe1(ss[j], sU[j], false, s_min, move);
        }
    }

    struct Eval2 {
        double Jty[NV];          // J' y at my (x_k, u_k)
        double gfx[NX], gfu[NU]; // scaled objective gradient
        double Jr[2], gfr[2];    // OD: J' y and the objective gradient at my decay rates
    };
    // Every row gradient of a stage lies in the span of V = [G2[0], G2[1], e_0, e_1, e_3, e_4] (the gradient of the second barrier point and
    // four unit vectors):  grad cbf_j = V c_j.  The rows of a stage are therefore accumulated as 6-vectors / 6 x 6 matrices and expanded once.
    __device__ __forceinline__ double row_coeffs(const double pt[3][2], int j, double c[6], double* hh = nullptr, double* e01 = nullptr) const {
        const double cx = lds[L.OB + 3 * j], cz = lds[L.OB + 3 * j + 1], d = P.radius + lds[L.OB + 3 * j + 2], off = P.beta * d * d;
        const double e0x = pt[0][0] - cx, e0z = pt[0][1] - cz, e1x = pt[1][0] - cx, e1z = pt[1][1] - cz, e2x = pt[2][0] - cx, e2z = pt[2][1] - cz;
        c[0] = 2.0 * w2 * e2x; c[1] = 2.0 * w2 * e2z;
        c[2] = 2.0 * (w0 * e0x + w1 * e1x); c[3] = 2.0 * (w0 * e0z + w1 * e1z);
        c[4] = 2.0 * w1 * e1x * P.dt; c[5] = 2.0 * w1 * e1z * P.dt;
        if (hh) { hh[0] = e0x * e0x + e0z * e0z - off; hh[1] = e1x * e1x + e1z * e1z - off; e01[0] = e0x; e01[1] = e0z; e01[2] = e1x; e01[3] = e1z; }
        return w0 * (e0x * e0x + e0z * e0z - off) + w1 * (e1x * e1x + e1z * e1z - off) + w2 * (e2x * e2x + e2z * e2z - off);
    }

    // condensed weight E_j and right-hand side b_j of row j (J dx - dy / E = b): the slack -- and in the restoration n and p -- are eliminated,
    //   1 / E = sum_v q_v,  q_v = 1 / (Sigma_v + dw);   b = -r + q_s rhs_s - q_n rhs_n + q_p rhs_p   (rhs_v = -(barrier gradient + sign_v y))
    struct RowW { double E, b, qs, qn, qp, rs_, rn, rp; };
    __device__ __forceinline__ RowW row_weights(int j, double mu, double dw) const {
        RowW w;
        const double stU = sU[j] - s[j];
        w.qs = 1.0 / (vU[j] / stU + dw);
        w.rs_ = yd[j] - (mu / stU - O.kappa_d * mu);
        double rd = dv[j] - s[j], e = w.qs, b = w.qs * w.rs_;
        w.qn = w.qp = w.rn = w.rp = 0.0;
        if (rs) {
            const double n = RW(R_N + j), p = RW(R_P + j), rho_R = O.resto_penalty_parameter;
            rd += n - p;
            w.qn = 1.0 / (RW(R_ZN + j) / n + dw); w.qp = 1.0 / (RW(R_ZP + j) / p + dw);
            w.rn = -(rho_R - mu / n + O.kappa_d * mu + yd[j]); w.rp = -(rho_R - mu / p + O.kappa_d * mu - yd[j]);
            e += w.qn + w.qp; b += -w.qn * w.rn + w.qp * w.rp;
        }
        w.E = 1.0 / e; w.b = -rd + b;
        return w;
    }

    // exchange through LDS: x_{k+1} (XS), u_{k-1} / u_{k+1} (US, slot k + 1 = u_k, slot 0 = u_prev), multipliers of the rows that DEFINE x_k (YS, slot k)
    __device__ __forceinline__ void publish() {
        sync();
        if (act) {
#pragma unroll
            for (int i = 0; i < NX; ++i) lds[L.XS + k * 6 + i] = x[i];
        }
        if (stg) {
#pragma unroll
            for (int j = 0; j < NU; ++j) lds[L.US + (k + 1) * 4 + j] = u[j];
#pragma unroll
            for (int i = 0; i < NX; ++i) lds[L.YS + (k + 1) * 6 + i] = dgc(i) * yc[i];
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NU; ++j) lds[L.US + j] = uprev[j];
#pragma unroll
            for (int i = 0; i < NX; ++i) lds[L.YS + i] = -lds[L.Y0 + i];
        }
        if (lane == N) {
#pragma unroll
            for (int j = 0; j < NU; ++j) lds[L.US + (N + 1) * 4 + j] = 0.0;
        }
        sync();
    }

    // Level 2 at the iterate: residuals, J'y, and -- `build` -- the stage block for the recursion: [A | B], H (condensed rows, bounds, dw; every
    // entry computed once and stored), gradient, defects.  ls: the least-square multiplier system (W = 0, Sigma = 1) instead of the Newton system.
    template <bool build, bool ls>
    __device__ __forceinline__ void eval2(Eval2& E, double mu, double dw, double& theta, double& fsum) {
        publish();
        if constexpr (OD) od_weights(rho[0], rho[1]);
        double um[NU], un[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) { um[j] = lds[L.US + k * 4 + j]; un[j] = lds[L.US + (k + 2 <= N + 1 ? k + 2 : N + 1) * 4 + j]; }
        const bool last = OD || k == N - 1;                                 // (optimal decay: R u^2, nothing couples neighbouring inputs)
        const double dt = P.dt;
        double th = 0.0;
        // scaled objective gradient
#pragma unroll
        for (int i = 0; i < NX; ++i) E.gfx[i] = 0.0;
        if (act) {
            if (rs) {
#pragma unroll
                for (int i = 0; i < NX; ++i) E.gfx[i] = zeta * dr2(i) * (x[i] - RW(R_XR + i));
            } else {
                E.gfx[0] = 2.0 * df * P.Q[0] * (x[0] - xg[0]); E.gfx[1] = 2.0 * df * P.Q[1] * (x[1] - xg[1]);
#pragma unroll
                for (int i = 2; i < NX; ++i) E.gfx[i] = 2.0 * df * P.Q[i] * x[i];
            }
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            double gj = 0.0;
            if (stg) {
                if (rs) gj = zeta * dr2(NX + j) * (u[j] - RW(R_XR + NX + j));
                else { gj = 2.0 * df * P.R[j] * (OD ? u[j] : u[j] - um[j]); if (!last) gj -= 2.0 * df * P.R[j] * (un[j] - u[j]); }
            }
            E.gfu[j] = gj;
        }
        E.Jr[0] = E.Jr[1] = 0.0;
        E.gfr[0] = (OD && stg) ? 2.0 * df * P.ps1 * (rho[0] - P.rf1) : 0.0; E.gfr[1] = (OD && stg) ? 2.0 * df * P.ps2 * (rho[1] - P.rf2) : 0.0;
        if constexpr (OD) {
            if (rs && stg) { E.gfr[0] = zeta * dr2(10) * (rho[0] - RW(R_XR + 10)); E.gfr[1] = zeta * dr2(11) * (rho[1] - RW(R_XR + 11)); }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) E.Jty[i] = 0.0;
        // rows that define x_k: -I (scaled) on x_k from the dynamics of stage k - 1; +I from the initial-state rows on x_0
        if (act) {
#pragma unroll
            for (int i = 0; i < NX; ++i) E.Jty[i] = -lds[L.YS + k * 6 + i];
        }
        // diagonal of the block (objective, bound terms, dw) and the bound terms of the gradient
        double dg_[NV], gb[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { dg_[i] = 0.0; gb[i] = 0.0; }
        if (build) {
            if (ls) {
#pragma unroll
                for (int i = 0; i < NV; ++i) dg_[i] = 1.0;
                gb[2] = -zxL[0] + zxU[0]; gb[3] = -zxL[1] + zxU[1]; gb[4] = -zxL[2];
#pragma unroll
                for (int j = 0; j < NU; ++j) gb[6 + j] = -zuL[j] + zuU[j];
            } else {
#pragma unroll
                for (int i = 0; i < NX; ++i) dg_[i] = ((rs && act) ? zeta * dr2(i) : 2.0 * df * P.Q[i]) + dw;
                const double sL[3] = {x[2] - xbL[0], x[3] - xbL[1], x[4] - xbL[2]}, sUp[2] = {xbU[0] - x[2], xbU[1] - x[3]};
                dg_[2] += zxL[0] / sL[0] + zxU[0] / sUp[0]; dg_[3] += zxL[1] / sL[1] + zxU[1] / sUp[1]; dg_[4] += zxL[2] / sL[2];
                gb[2] = -mu / sL[0] + mu / sUp[0]; gb[3] = -mu / sL[1] + mu / sUp[1]; gb[4] = -mu / sL[2] + O.kappa_d * mu;
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    const double a = u[j] - ubL[j], b = ubU[j] - u[j];
                    dg_[6 + j] = ((rs && stg) ? zeta * dr2(NX + j) : 2.0 * df * P.R[j] * (last ? 1.0 : 2.0)) + dw + zuL[j] / a + zuU[j] / b;
                    gb[6 + j] = -mu / a + mu / b;
                }
            }
        }
        if (stg) {
            D2 acc[3], gc[4][3];
            accel<D2>(P, d2var(x[2], 0), d2var(x[3], 1), d2var(x[4], 2), u, acc, gc);
            {
                double xn[NX];
                xn[0] = x[0] + dt * x[3]; xn[1] = x[1] + dt * x[4]; xn[2] = x[2] + dt * x[5];
                xn[3] = x[3] + dt * acc[0].v; xn[4] = x[4] + dt * acc[1].v; xn[5] = x[5] + dt * acc[2].v;
#pragma unroll
                for (int i = 0; i < NX; ++i) { rc[i] = dgc(i) * (xn[i] - lds[L.XS + (k + 1) * 6 + i]); th += fabs(rc[i]); }
            }
            // rows 3..5 of [A | B] (rows 0..2 are e_i + dt e_{i+3}): columns 2, 3, 4 (theta, x_dot, z_dot) and 6..9 (inputs)
            double a3[3][7];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a3[i][c] = dt * acc[i].d[c] + ((i + 1 == c) ? 1.0 : 0.0);      // A[3+i][2+c]: the identity sits at column 3 + i
#pragma unroll
                for (int j = 0; j < NU; ++j) a3[i][3 + j] = dt * gc[j][i].v;
            }
            // (A[5][5] = 1: theta_dot's own column carries no derivative of the accelerations)
            if (build) {
                ldsd* ABo = lds + L.AB + k * ABS;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int c = 0; c < 7; ++c) ABo[i * 7 + c] = a3[i][c];
            }
            // J' y of my dynamics rows: [A | B]' (dgc yc)
            {
                double wy[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) wy[i] = dgc(i) * yc[i];
                E.Jty[0] += wy[0]; E.Jty[1] += wy[1]; E.Jty[2] += wy[2];
                E.Jty[3] += dt * wy[0]; E.Jty[4] += dt * wy[1]; E.Jty[5] += dt * wy[2] + wy[5];
#pragma unroll
                for (int c = 0; c < 3; ++c) E.Jty[2 + c] += a3[0][c] * wy[3] + a3[1][c] * wy[4] + a3[2][c] * wy[5];
#pragma unroll
                for (int j = 0; j < NU; ++j) E.Jty[6 + j] += a3[0][3 + j] * wy[3] + a3[1][3 + j] * wy[4] + a3[2][3 + j] * wy[5];
            }
            double pt[3][2], G2[2][NV];
            points(x, acc[0].v, acc[1].v, pt);
            // gradient of the second barrier point: e_c + 2 dt e_{3+c} + dt (row 3 + c of [A | B] minus its identity)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int i = 0; i < NV; ++i) G2[c][i] = 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) G2[c][2 + q] = dt * a3[c][q];
#pragma unroll
                for (int j = 0; j < NU; ++j) G2[c][6 + j] = dt * a3[c][3 + j];
                G2[c][c] += 1.0; G2[c][3 + c] += dt;
            }
            // rows: accumulated in the 6-dimensional span of V (row_coeffs)
            double M[21], gv[6], jv[6], sl = 0.0;
            // OD: the rho part of the 8 x 8 accumulation: Mr[r][0..5] = coupling with the span, Dr = (11, 12, 22) block, gr = gradient part
            double Mr[2][6], Dl12 = 0.0;
#pragma unroll
            for (int i = 0; i < 21; ++i) M[i] = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) { gv[i] = 0.0; jv[i] = 0.0; Mr[0][i] = 0.0; Mr[1][i] = 0.0; }
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j < K) {
                    double c[6], hh[2], e01[4], br[2] = {0.0, 0.0};
                    const double cv = row_coeffs(pt, j, c, hh, e01);
                    const double sc = dgd(j);
                    dv[j] = -sc * cv;
                    const double rd = dv[j] - s[j] + (rs ? RW(R_N + j) - RW(R_P + j) : 0.0);
                    th += fabs(rd);
                    const double om = sc * yd[j];                                   // weight of grad^2 (-cbf_j) in the Hessian of the Lagrangian
#pragma unroll
                    for (int i = 0; i < 6; ++i) jv[i] += om * c[i];                 // J' y: yd_j * grad d_j = -om V c
                    sl += om;
                    if constexpr (OD) {
                        // d cbf / d rho_i = a_i (h1 - h0) + a1 a2 rho_other h0;  d2 / d rho1 d rho2 = a1 a2 h0;
                        // d2 cbf / d (x, u) d rho_i = a_i (grad h1 - grad h0) + a1 a2 rho_other grad h0, in span coordinates (e_0, e_1 | e_3, e_4)
                        const double a12 = P.alpha1 * P.alpha2;
                        br[0] = P.alpha1 * (hh[1] - hh[0]) + a12 * rho[1] * hh[0]; br[1] = P.alpha2 * (hh[1] - hh[0]) + a12 * rho[0] * hh[0];
                        E.Jr[0] -= om * br[0]; E.Jr[1] -= om * br[1];
                        if (build && !ls) {
                            Dl12 -= om * a12 * hh[0];
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                const double ai = r == 0 ? P.alpha1 : P.alpha2, ro = r == 0 ? rho[1] : rho[0];
                                const double k0 = a12 * ro - ai;                     // factor of grad h0; grad h1 carries a_i
                                Mr[r][2] -= om * 2.0 * (ai * e01[2] + k0 * e01[0]); Mr[r][3] -= om * 2.0 * (ai * e01[3] + k0 * e01[1]);
                                Mr[r][4] -= om * 2.0 * ai * e01[2] * P.dt; Mr[r][5] -= om * 2.0 * ai * e01[3] * P.dt;
                            }
                        }
                    }
                    if (build && !OD) {
                        double Ej, bd;
                        if (ls) { Ej = 1.0; bd = -vU[j]; }                           // q = 1, rhs_t = -(0 + vU), rhs_g = 0: b = q rhs_t
                        else if (rs) { const RowW w = row_weights(j, mu, dw); Ej = w.E; bd = w.b; }
                        else {
                            const double stU = sU[j] - s[j];
                            Ej = vU[j] / stU + dw;
                            const double gt = mu / stU - O.kappa_d * mu;
                            bd = -rd + (yd[j] - gt) / Ej;
                        }
                        // a_j = -dgd V c; H += E a a' = V (ea c c') V', g -= a E b = V (eb c)
                        const double ea = Ej * sc * sc, eb = Ej * bd * sc;
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a) {
                            gv[a] += eb * c[a];
#pragma unroll
                            for (int b = a; b < 6; ++b, ++e) M[e] += ea * c[a] * c[b];
                        }
                    }
                } else dv[j] = 0.0;
            }
            if constexpr (OD) {
                if (build) {
                    // The decay rates leave the stage before the recursion: with a = -dgd (V c; br) the row's gradient in (span, rho), the 8 x 8 block is
                    // K = K0 + sum_j ea_j a_j a_j', K0 = [0, Mr'; Mr, S] from the Lagrangian alone (S = (objective + dw) I + cross term), and what the
                    // recursion needs is its Schur complement on the span.  Assembled first and eliminated afterwards, an active row at mu ~ 1e-9
                    // (ea br^2 ~ 1e13 against S ~ 1e-2) leaves nothing of S in D and nothing of K0 in ea c c' - (ea br c')' D^-1 (ea br c') -- measured:
                    // the dual infeasibility stalls at 1e-4 and the solve crawls for thousands of iterations where the oracle takes one step.  So S is
                    // eliminated first and the rows enter one at a time as rank-one updates of the Schur complement:
                    //   w = D^-1 br, r = c - T br, q = 1 / ea + br . w:   K/rho += r r' / q,  T += r w' / q,  D^-1 -= w w' / q       (T = Mr' D^-1)
                    //   sigma = (eb / ea - br . t) / q:                   g_span += sigma r,  t += sigma w                        (t = D^-1 g_rho)
                    // -- no difference of large numbers anywhere (tools/micro/seq_schur.py: 1e-13 where the assembled form is off by 1e+4).  Inertia:
                    // det (D + e b b') = det D (1 + e b' D^-1 b) and a positive semidefinite update lowers no eigenvalue, so every q < 0 turns one
                    // negative eigenvalue of D positive; S alone may be indefinite (its cross term against 2 df ps)
                    const double s11 = ls ? 1.0 : (rs ? zeta * dr2(10) : 2.0 * df * P.ps1) + dw, s22 = ls ? 1.0 : (rs ? zeta * dr2(11) : 2.0 * df * P.ps2) + dw, s12 = Dl12;
                    const double det = s11 * s22 - s12 * s12;
                    int nneg = det > 0.0 ? (s11 > 0.0 ? 0 : 2) : 1;
                    bool sing = !(fabs(det) > 0.0);
                    const double idet = 1.0 / det;
                    double Di[3] = {s22 * idet, -s12 * idet, s11 * idet};
                    const double g0 = E.gfr[0] + (ls ? 0.0 : E.Jr[0]), g1 = E.gfr[1] + (ls ? 0.0 : E.Jr[1]);
                    odt[0] = Di[0] * g0 + Di[1] * g1; odt[1] = Di[1] * g0 + Di[2] * g1;
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        odT[a][0] = Mr[0][a] * Di[0] + Mr[1][a] * Di[1]; odT[a][1] = Mr[0][a] * Di[1] + Mr[1][a] * Di[2];
                        gv[a] -= odT[a][0] * g0 + odT[a][1] * g1;
                    }
                    {
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int b = a; b < 6; ++b, ++e) M[e] -= odT[a][0] * Mr[0][b] + odT[a][1] * Mr[1][b];
                    }
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
                        if (j < K) {
                            double c[6], hh[2], e01[4];
                            row_coeffs(pt, j, c, hh, e01);
                            const double a12 = P.alpha1 * P.alpha2;
                            const double b0 = P.alpha1 * (hh[1] - hh[0]) + a12 * rho[1] * hh[0], b1 = P.alpha2 * (hh[1] - hh[0]) + a12 * rho[0] * hh[0];
                            const double sc = dgd(j);
                            double Ej, bd;
                            if (ls) { Ej = 1.0; bd = -vU[j]; }
                            else if (rs) { const RowW w = row_weights(j, mu, dw); Ej = w.E; bd = w.b; }
                            else {
                                const double stU = sU[j] - s[j];
                                Ej = vU[j] / stU + dw;
                                const double gt = mu / stU - O.kappa_d * mu;
                                bd = -(dv[j] - s[j]) + (yd[j] - gt) / Ej;
                            }
                            const double w0_ = Di[0] * b0 + Di[1] * b1, w1_ = Di[1] * b0 + Di[2] * b1;
                            const double q = 1.0 / (Ej * sc * sc) + b0 * w0_ + b1 * w1_;
                            if (q < 0.0) --nneg;
                            sing |= !(fabs(q) > 0.0);
                            const double iq = 1.0 / q, sig = (bd / sc - (b0 * odt[0] + b1 * odt[1])) * iq;
                            double r[6];
#pragma unroll
                            for (int a = 0; a < 6; ++a) r[a] = c[a] - (odT[a][0] * b0 + odT[a][1] * b1);
                            int e = 0;
#pragma unroll
                            for (int a = 0; a < 6; ++a) {
                                const double ra = r[a] * iq;
                                gv[a] += sig * r[a];
                                odT[a][0] += ra * w0_; odT[a][1] += ra * w1_;
#pragma unroll
                                for (int b = a; b < 6; ++b, ++e) M[e] += ra * r[b];
                            }
                            odt[0] += sig * w0_; odt[1] += sig * w1_;
                            Di[0] -= w0_ * w0_ * iq; Di[1] -= w0_ * w1_ * iq; Di[2] -= w1_ * w1_ * iq;
                        }
                    }
                    od_bad = sing || nneg != 0;                                     // not positive definite: the inertia correction takes it
                }
            }
            // Jty -= V jv
#pragma unroll
            for (int a = 0; a < NV; ++a) E.Jty[a] -= G2[0][a] * jv[0] + G2[1][a] * jv[1];
            E.Jty[0] -= jv[2]; E.Jty[1] -= jv[3]; E.Jty[3] -= jv[4]; E.Jty[4] -= jv[5];
            if (build) {
                // M index of (a, b), a <= b, in the 6 x 6 upper packing: 0:(0,0) 1:(0,1) .. 5:(0,5) 6:(1,1) .. 10:(1,5) 11:(2,2) .. 14:(2,5) 15:(3,3) 16:(3,4) 17:(3,5) 18:(4,4) 19:(4,5) 20:(5,5)
                double cc[3] = {0.0, 0.0, 0.0};
                if (!ls) {
                    // curvature of -cbf in the points: -2 sl (w0 G0'G0 + w1 G1'G1 + w2 G2'G2), in the same span
                    const double o0 = -2.0 * w0 * sl, o1 = -2.0 * w1 * sl, o2 = -2.0 * w2 * sl;
                    M[0] += o2; M[6] += o2;
                    M[11] += o0 + o1; M[15] += o0 + o1; M[13] += o1 * dt; M[17] += o1 * dt; M[18] += o1 * dt * dt; M[20] += o1 * dt * dt;
                    // second derivatives of the dynamics: weights dgc y (rows 3..5) + dt nu2 through the second barrier point, nu2 = -(jv[0], jv[1])
                    cc[0] = (dgc(3) * yc[3] - dt * jv[0]) * dt; cc[1] = (dgc(4) * yc[4] - dt * jv[1]) * dt; cc[2] = dgc(5) * yc[5] * dt;
                }
                auto Mi = [](int a, int b) { return a <= b ? a * 6 - a * (a - 1) / 2 + (b - a) : b * 6 - b * (b - 1) / 2 + (a - b); };
                // H[a][b] = sum_q t_a[q] V[b][q] with t_a[q] = sum_p V[a][p] M[p][q]  (+ dynamics curvature, + diagonal): computed once, stored
                const int U4[4] = {0, 1, 3, 4};
                ldsd* Ho = lds + L.H + k * 55; ldsd* Go = lds + L.G + k * 10;
#pragma unroll
                for (int a = 0; a < NV; ++a) {
                    double ta[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        double v = G2[0][a] * M[Mi(0, q)] + G2[1][a] * M[Mi(1, q)];
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) if (U4[pp] == a) v += M[Mi(2 + pp, q)];
                        ta[q] = v;
                    }
#pragma unroll
                    for (int b = a; b < NV; ++b) {
                        double v = ta[0] * G2[0][b] + ta[1] * G2[1][b];
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) if (U4[pp] == b) v += ta[2 + pp];
                        if (a == b) v += dg_[a];
                        if (a >= 2 && a <= 4 && b <= 4) {
                            const int e = (a - 2) * 3 - (a - 2) * (a - 3) / 2 + (b - a);         // (a-2, b-2) in the 3 x 3 upper packing 00 01 02 11 12 22
                            v += cc[0] * acc[0].h[e] + cc[1] * acc[1].h[e] + cc[2] * acc[2].h[e];
                        }
                        if (a >= 2 && a <= 4 && b >= 6) v += cc[0] * gc[b - 6][0].d[a - 2] + cc[1] * gc[b - 6][1].d[a - 2] + cc[2] * gc[b - 6][2].d[a - 2];
                        Ho[sym(a, b)] = v;
                    }
                    double gval = gb[a] + G2[0][a] * gv[0] + G2[1][a] * gv[1] + (a < NX ? E.gfx[a] : E.gfu[a - NX]) + (ls ? 0.0 : E.Jty[a]);
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) if (U4[pp] == a) gval += gv[2 + pp];
                    Go[a] = gval;
                }
                // defects (unscaled) of my dynamics rows -> C[k + 1] = rc / dgc (ls: 0)
#pragma unroll
                for (int i = 0; i < NX; ++i) lds[L.C + (k + 1) * 6 + i] = ls ? 0.0 : rc[i] / dgc(i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) rc[i] = 0.0;
#pragma unroll
            for (int j = 0; j < KS; ++j) dv[j] = 0.0;
            if (build && lane == N) {                                       // terminal state: diagonal block, gradient
                ldsd* Ho = lds + L.H + N * 55; ldsd* Go = lds + L.G + N * 10;
#pragma unroll
                for (int a = 0; a < NX; ++a) {
#pragma unroll
                    for (int b = a; b < NX; ++b) Ho[sym(a, b)] = a == b ? dg_[a] : 0.0;
                    Go[a] = gb[a] + E.gfx[a] + (ls ? 0.0 : E.Jty[a]);
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double r0 = x[i] - x0[i];
                th += fabs(r0);
                if (build) lds[L.C + i] = ls ? 0.0 : -r0;                  // C[0] = dx_0 = b_0 = -r0
            }
        }
        theta = ipm::wsum(th);
        fsum = ipm::wsum(cost_share(x, u, um, rho));
        if (build) sync();
    }

    // after riccati_backward / forward: the multiplier steps of the dynamics rows (from the value function), of my rows and bounds (dx, du stay in LDS: DX, DU; dy in LAM)
    __device__ __forceinline__ void finish_step(bool ls, double mu, double dw) {
        sync();
        double dx[NX], du[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) dx[i] = act ? lds[L.DX + k * 6 + i] : 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) du[j] = stg ? lds[L.DU + k * NU + j] : 0.0;
        // multipliers of the dynamics rows (and of the initial-state rows: k = 0): lam_k = (P_k xi_k + p_k)_x, xi_k = (dx_k, du_{k-1}), with the x rows
        // of P_k where riccati_backward left them (the H slot of stage k); the terminal stage has no recursion behind it: lam_N = H_N dx_N + g_N
        if (act) {
            const ldsd* Pk = lds + L.H + k * 55;
            if (k < N) {
                double dv_[NU];
#pragma unroll
                for (int j = 0; j < NU; ++j) dv_[j] = k > 0 ? lds[L.DU + (k - 1) * NU + j] : 0.0;
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    double v = Pk[45 + a];
#pragma unroll
                    for (int b = 0; b < NX; ++b) { const int lo_ = a < b ? a : b, hi_ = a < b ? b : a; v += Pk[lo_ * NX - lo_ * (lo_ - 1) / 2 + (hi_ - lo_)] * dx[b]; }
#pragma unroll
                    for (int j = 0; j < NU; ++j) v += Pk[21 + a * 4 + j] * dv_[j];
                    lds[L.LAM + k * 6 + a] = v;
                }
            } else {
                const ldsd* g = lds + L.G + k * 10;
#pragma unroll
                for (int a = 0; a < NX; ++a) {
                    double v = g[a];
#pragma unroll
                    for (int b = 0; b < NX; ++b) v += Pk[sym(a, b)] * dx[b];
                    lds[L.LAM + k * 6 + a] = v;
                }
            }
        }
        sync();
        // rows: dy_d = E (a . dw - b), ds = q (rhs_t + dy_d), dvU
        if (stg) {
            double acc[3], gc[4][3], pt[3][2];
            accel<double>(P, x[2], x[3], x[4], u, acc, gc);
            points(x, acc[0], acc[1], pt);
            const ldsd* A3 = lds + L.AB + k * ABS;
            // vd = V' (dx, du): grad p2_c = e_c + 2 dt e_{3+c} + dt (row 3 + c of [A | B] - e_{3+c})
            double vd[6] = {dx[0] + P.dt * dx[3], dx[1] + P.dt * dx[4], dx[0], dx[1], dx[3], dx[4]};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const ldsd* row = A3 + c * 7;
                const double v = row[0] * dx[2] + row[1] * dx[3] + row[2] * dx[4] + row[3] * du[0] + row[4] * du[1] + row[5] * du[2] + row[6] * du[3];
                vd[c] += P.dt * v;
            }
            if constexpr (OD) {                                              // d rho = -D^-1 (g_rho + M_rho,v vd) = -(t + T' vd)
                double t0 = odt[0], t1 = odt[1];
#pragma unroll
                for (int i = 0; i < 6; ++i) { t0 += odT[i][0] * vd[i]; t1 += odT[i][1] * vd[i]; }
                drho[0] = -t0; drho[1] = -t1;
            }
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j < K) {
                    double c[6], hh[2], e01[4];
                    row_coeffs(pt, j, c, hh, e01);
                    double adw = 0.0;
#pragma unroll
                    for (int i = 0; i < 6; ++i) adw += c[i] * vd[i];
                    if constexpr (OD) {
                        const double a12 = P.alpha1 * P.alpha2;
                        adw += (P.alpha1 * (hh[1] - hh[0]) + a12 * rho[1] * hh[0]) * drho[0] + (P.alpha2 * (hh[1] - hh[0]) + a12 * rho[0] * hh[0]) * drho[1];
                    }
                    adw *= -dgd(j);
                    if (ls) { dyd[j] = adw + vU[j]; ds[j] = 0.0; dvU[j] = 0.0; }
                    else if (rs) {                                                   // dy = E (a . dw - b);  dv = q_v (rhs_v - sign_v dy);  multipliers of n, p >= 0
                        const RowW w = row_weights(j, mu, dw);
                        const double stU = sU[j] - s[j], n = RW(R_N + j), p = RW(R_P + j), zn = RW(R_ZN + j), zp = RW(R_ZP + j);
                        dyd[j] = w.E * (adw - w.b);
                        ds[j] = w.qs * (w.rs_ + dyd[j]);
                        dvU[j] = mu / stU - vU[j] + vU[j] * ds[j] / stU;
                        const double dn = w.qn * (w.rn - dyd[j]), dp = w.qp * (w.rp + dyd[j]);
                        RW(R_DN + j) = dn; RW(R_DP + j) = dp;
                        RW(R_DZN + j) = mu / n - zn - zn * dn / n; RW(R_DZP + j) = mu / p - zp - zp * dp / p;
                    } else {
                        const double stU = sU[j] - s[j], sig = vU[j] / stU, Ej = sig + dw, gt = mu / stU - O.kappa_d * mu;
                        const double rd = dv[j] - s[j], rhs_t = yd[j] - gt, bd = -rd + rhs_t / Ej;
                        dyd[j] = Ej * (adw - bd);
                        ds[j] = (rhs_t + dyd[j]) / Ej;
                        dvU[j] = mu / stU - vU[j] + vU[j] * ds[j] / stU;
                    }
                } else { dyd[j] = 0.0; ds[j] = 0.0; dvU[j] = 0.0; }
            }
        } else {
#pragma unroll
            for (int j = 0; j < KS; ++j) { dyd[j] = 0.0; ds[j] = 0.0; dvU[j] = 0.0; }
            drho[0] = drho[1] = 0.0;
        }
        if (!ls) {
            if (act) {
                const double sL[3] = {x[2] - xbL[0], x[3] - xbL[1], x[4] - xbL[2]}, sUp[2] = {xbU[0] - x[2], xbU[1] - x[3]};
                const double dxl[3] = {dx[2], dx[3], dx[4]};
#pragma unroll
                for (int i = 0; i < 3; ++i) dzxL[i] = mu / sL[i] - zxL[i] - zxL[i] * dxl[i] / sL[i];
#pragma unroll
                for (int i = 0; i < 2; ++i) dzxU[i] = mu / sUp[i] - zxU[i] + zxU[i] * dxl[i] / sUp[i];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) dzxL[i] = 0.0;
                dzxU[0] = dzxU[1] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                if (stg) {
                    const double a = u[j] - ubL[j], b = ubU[j] - u[j];
                    dzuL[j] = mu / a - zuL[j] - zuL[j] * du[j] / a; dzuU[j] = mu / b - zuU[j] + zuU[j] * du[j] / b;
                } else { dzuL[j] = 0.0; dzuU[j] = 0.0; }
            }
        }
    }

    // ---- optimality error (eq. (5)): E_mu and its parts ---------------------------------------------------------------------------------
    __device__ __forceinline__ void errors(const Eval2& E, double mu, double& Emu, double& dinf, double& pinf, double& comp, double& un_pinf, double* sc_out = nullptr) const {
        double d = 0.0, p = 0.0, c = 0.0, ysum = 0.0, zsum = 0.0, up = 0.0;
        if (act) {
            double gl[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) gl[i] = E.gfx[i] + E.Jty[i];
            gl[2] += -zxL[0] + zxU[0]; gl[3] += -zxL[1] + zxU[1]; gl[4] += -zxL[2];
#pragma unroll
            for (int i = 0; i < NX; ++i) d = fmax(d, fabs(gl[i]));
            const double sL[3] = {x[2] - xbL[0], x[3] - xbL[1], x[4] - xbL[2]}, sUp[2] = {xbU[0] - x[2], xbU[1] - x[3]};
#pragma unroll
            for (int i = 0; i < 3; ++i) { c = fmax(c, fabs(sL[i] * zxL[i] - mu)); zsum += fabs(zxL[i]); }
#pragma unroll
            for (int i = 0; i < 2; ++i) { c = fmax(c, fabs(sUp[i] * zxU[i] - mu)); zsum += fabs(zxU[i]); }
        }
        if (stg) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                d = fmax(d, fabs(E.gfu[j] + E.Jty[6 + j] - zuL[j] + zuU[j]));
                c = fmax(c, fmax(fabs((u[j] - ubL[j]) * zuL[j] - mu), fabs((ubU[j] - u[j]) * zuU[j] - mu)));
                zsum += fabs(zuL[j]) + fabs(zuU[j]);
            }
            if constexpr (OD) d = fmax(d, fmax(fabs(E.gfr[0] + E.Jr[0]), fabs(E.gfr[1] + E.Jr[1])));
#pragma unroll
            for (int i = 0; i < NX; ++i) { p = fmax(p, fabs(rc[i])); up = fmax(up, fabs(rc[i] / dgc(i))); ysum += fabs(yc[i]); }
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j < K) {
                    d = fmax(d, fabs(-yd[j] + vU[j]));
                    double rd = dv[j] - s[j];
                    if (rs) {
                        const double n = RW(R_N + j), pp = RW(R_P + j), zn = RW(R_ZN + j), zp = RW(R_ZP + j), rho_R = O.resto_penalty_parameter;
                        rd += n - pp;
                        d = fmax(d, fmax(fabs(rho_R + yd[j] - zn), fabs(rho_R - yd[j] - zp)));
                        c = fmax(c, fmax(fabs(n * zn - mu), fabs(pp * zp - mu)));
                        zsum += fabs(zn) + fabs(zp);
                    }
                    p = fmax(p, fabs(rd)); up = fmax(up, fabs(rd / dgd(j)));
                    c = fmax(c, fabs((sU[j] - s[j]) * vU[j] - mu));
                    ysum += fabs(yd[j]); zsum += fabs(vU[j]);
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double r0 = fabs(x[i] - x0[i]); p = fmax(p, r0); up = fmax(up, r0); ysum += fabs(lds[L.Y0 + i]); }
        }
        dinf = ipm::wmax(d); pinf = ipm::wmax(p); comp = ipm::wmax(c); un_pinf = ipm::wmax(up);
        ysum = ipm::wsum(ysum); zsum = ipm::wsum(zsum);
        const double m = (double)(6 * (N + 1) + N * K), nb = (double)(5 * (N + 1) + 8 * N + N * K * (rs ? 3 : 1));
        const double sd = fmax(O.s_max, (ysum + zsum) / (m + nb)) / O.s_max, sc = fmax(O.s_max, zsum / nb) / O.s_max;
        Emu = fmax(fmax(dinf / sd, pinf), comp / sc);
        if (sc_out) *sc_out = sc;
    }

    // fraction to the boundary over my primal / dual variables; directional derivative of the barrier function along the step
    __device__ __forceinline__ double ftb1(double tau, double sl, double dsl) const { return dsl < 0.0 ? fmin(1.0, -tau * sl / dsl) : 1.0; }
    __device__ __forceinline__ void step_lengths(const Eval2& E, double tau, double mu, double& a_max, double& a_z, double& gBD) const {
        double ap = 1.0, az = 1.0, v = 0.0;
        if (act) {
            double dx[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) dx[i] = lds[L.DX + k * 6 + i];
            const double sL[3] = {x[2] - xbL[0], x[3] - xbL[1], x[4] - xbL[2]}, sUp[2] = {xbU[0] - x[2], xbU[1] - x[3]};
            ap = fmin(ap, ftb1(tau, sL[0], dx[2])); ap = fmin(ap, ftb1(tau, sL[1], dx[3])); ap = fmin(ap, ftb1(tau, sL[2], dx[4]));
            ap = fmin(ap, ftb1(tau, sUp[0], -dx[2])); ap = fmin(ap, ftb1(tau, sUp[1], -dx[3]));
#pragma unroll
            for (int i = 0; i < 3; ++i) az = fmin(az, ftb1(tau, zxL[i], dzxL[i]));
#pragma unroll
            for (int i = 0; i < 2; ++i) az = fmin(az, ftb1(tau, zxU[i], dzxU[i]));
#pragma unroll
            for (int i = 0; i < NX; ++i) v += E.gfx[i] * dx[i];
            v += (-mu / sL[0] + mu / sUp[0]) * dx[2] + (-mu / sL[1] + mu / sUp[1]) * dx[3] + (-mu / sL[2] + O.kappa_d * mu) * dx[4];
        }
        if (stg) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const double duj = lds[L.DU + k * NU + j];
                ap = fmin(ap, ftb1(tau, u[j] - ubL[j], duj)); ap = fmin(ap, ftb1(tau, ubU[j] - u[j], -duj));
                az = fmin(az, ftb1(tau, zuL[j], dzuL[j])); az = fmin(az, ftb1(tau, zuU[j], dzuU[j]));
                v += (E.gfu[j] - mu / (u[j] - ubL[j]) + mu / (ubU[j] - u[j])) * duj;
            }
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j < K) {
                    ap = fmin(ap, ftb1(tau, sU[j] - s[j], -ds[j])); az = fmin(az, ftb1(tau, vU[j], dvU[j]));
                    v += (mu / (sU[j] - s[j]) - O.kappa_d * mu) * ds[j];
                    if (rs) {
                        const double n = RW(R_N + j), p = RW(R_P + j), dn = RW(R_DN + j), dp = RW(R_DP + j), rho_R = O.resto_penalty_parameter;
                        ap = fmin(ap, fmin(ftb1(tau, n, dn), ftb1(tau, p, dp)));
                        az = fmin(az, fmin(ftb1(tau, RW(R_ZN + j), RW(R_DZN + j)), ftb1(tau, RW(R_ZP + j), RW(R_DZP + j))));
                        v += (rho_R - mu / n + O.kappa_d * mu) * dn + (rho_R - mu / p + O.kappa_d * mu) * dp;
                    }
                }
            }
            if constexpr (OD) v += E.gfr[0] * drho[0] + E.gfr[1] * drho[1];
        }
        a_max = ipm::wmin(ap); a_z = ipm::wmin(az); gBD = ipm::wsum(v);
    }

    // ---- filter ---------------------------------------------------------------------------------------------------------------------
    __device__ __forceinline__ bool filter_ok(double phi, double th) const {
        for (int i = 0; i < nfilt; ++i) {
            const double p = lds[fpo + i], t = lds[fto + i];
            if (!(cmp_le(phi, p, p) || cmp_le(th, t, t))) return false;
        }
        return true;
    }
    __device__ __forceinline__ void filter_add(double phi, double th) {
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
        if (n >= NFILT) n = NFILT - 1;
        if (lane == 0) { lds[fpo + n] = phi; lds[fto + n] = th; }
        nfilt = n + 1;
        sync();
    }

    // ---- the solve ------------------------------------------------------------------------------------------------------------------
    __device__ __forceinline__ void push1(double& v, double lo, double hi, bool fl, bool fu) const {
        const double k1 = O.bound_push, k2 = O.bound_frac;
        const double rng = (fl && fu) ? hi - lo : INF_;
        if (fl) v = fmax(v, lo + fmin(k1 * fmax(1.0, fabs(lo)), k2 * rng));
        if (fu) v = fmin(v, hi - fmin(k1 * fmax(1.0, fabs(hi)), k2 * rng));
    }

    __device__ SC_MS_SOLVE_INLINE void solve(int& status_out, int& iters_out, double* trace) {
        const double rl = O.bound_relax_factor;
        // ---- start: x_k = x0, u_k = u_prev (set_initial_guess); scaling at that point; bounds relaxed; push ----
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = x0[i]; yc[i] = 0.0; }
#pragma unroll
        for (int j = 0; j < NU; ++j) u[j] = uprev[j];
#pragma unroll
        for (int j = 0; j < KS; ++j) { s[j] = 0.0; yd[j] = 0.0; vU[j] = 1.0; sU[j] = rl; }
        rho[0] = P.rf1; rho[1] = P.rf2; drho[0] = drho[1] = 0.0;
        if constexpr (OD) od_weights(rho[0], rho[1]);
        {
            const double lo[3] = {-P.pitch_max, -P.v_max, -P.descent_max}, hi[2] = {P.pitch_max, P.v_max};
#pragma unroll
            for (int i = 0; i < 3; ++i) { xbL[i] = lo[i] - rl * fmax(1.0, fabs(lo[i])); zxL[i] = 1.0; }
#pragma unroll
            for (int i = 0; i < 2; ++i) { xbU[i] = hi[i] + rl * fmax(1.0, fabs(hi[i])); zxU[i] = 1.0; }
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                ubL[j] = P.u_lo[j] - rl * fmax(1.0, fabs(P.u_lo[j])); ubU[j] = P.u_hi[j] + rl * fmax(1.0, fabs(P.u_hi[j]));
                zuL[j] = 1.0; zuU[j] = 1.0;
            }
        }
        // gradient-based scaling at the user's starting point (every stage is the same point there): df, and the row scales into LDS
        {
            double gm = fmax(2.0 * P.Q[0] * fabs(x0[0] - xg[0]), 2.0 * P.Q[1] * fabs(x0[1] - xg[1]));
#pragma unroll
            for (int i = 2; i < NX; ++i) gm = fmax(gm, 2.0 * P.Q[i] * fabs(x0[i]));
            if constexpr (OD) {                                               // the input term is R u^2 there (not the rate): 2 R u at u_prev
#pragma unroll
                for (int j = 0; j < NU; ++j) gm = fmax(gm, 2.0 * P.R[j] * fabs(uprev[j]));
            }
            df = gm > O.nlp_scaling_max_gradient ? fmax(O.nlp_scaling_min_value, O.nlp_scaling_max_gradient / gm) : 1.0;
            D2 acc[3], gc[4][3];
            accel<D2>(P, d2var(x0[2], 0), d2var(x0[3], 1), d2var(x0[4], 2), uprev, acc, gc);
            const double dt = P.dt;
            double G2[2][NV], pt[3][2];
            double ABr[3][NV], scl[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int c = 0; c < NV; ++c) ABr[i][c] = (3 + i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) ABr[i][2 + c] += dt * acc[i].d[c];
#pragma unroll
                for (int j = 0; j < NU; ++j) ABr[i][6 + j] = dt * gc[j][i].v;
                double rm = 1.0;
#pragma unroll
                for (int c = 0; c < NV; ++c) rm = fmax(rm, fabs(ABr[i][c]));
                scl[i] = rm > O.nlp_scaling_max_gradient ? fmax(O.nlp_scaling_min_value, O.nlp_scaling_max_gradient / rm) : 1.0;
            }
            points(x0, acc[0].v, acc[1].v, pt);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int i = 0; i < NV; ++i) G2[c][i] = dt * ABr[c][i];
                G2[c][c] += 1.0; G2[c][3 + c] += dt;
            }
            sync();
            if (lane == 0) {
                lds[L.SC + 0] = 1.0; lds[L.SC + 1] = 1.0; lds[L.SC + 2] = 1.0; lds[L.SC + 3] = scl[0]; lds[L.SC + 4] = scl[1]; lds[L.SC + 5] = scl[2];
#pragma unroll
                for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                double sc = 1.0;
                if (j < K) {
                    double c[6], r[NV], hh[2], e01[4];
                    row_coeffs(pt, j, c, hh, e01);
#pragma unroll
                    for (int a = 0; a < NV; ++a) r[a] = G2[0][a] * c[0] + G2[1][a] * c[1];
                    r[0] += c[2]; r[1] += c[3]; r[3] += c[4]; r[4] += c[5];
                    double rm = 0.0;
#pragma unroll
                    for (int i = 0; i < NV; ++i) rm = fmax(rm, fabs(r[i]));
                    if constexpr (OD) {
                        const double a12 = P.alpha1 * P.alpha2;
                        rm = fmax(rm, fmax(fabs(P.alpha1 * (hh[1] - hh[0]) + a12 * rho[1] * hh[0]), fabs(P.alpha2 * (hh[1] - hh[0]) + a12 * rho[0] * hh[0])));
                    }
                    sc = rm > O.nlp_scaling_max_gradient ? fmax(O.nlp_scaling_min_value, O.nlp_scaling_max_gradient / rm) : 1.0;
                }
                if (lane == 0) lds[L.SC + 6 + j] = sc;
            }
            sync();
        }
        push1(x[2], xbL[0], xbU[0], true, true); push1(x[3], xbL[1], xbU[1], true, true); push1(x[4], xbL[2], 0.0, true, false);
#pragma unroll
        for (int j = 0; j < NU; ++j) push1(u[j], ubL[j], ubU[j], true, true);
        // One pass of the loop = one evaluation at the iterate + what the phase does with it (a single call site of every big routine keeps
        // the code, and the register allocator's problem, small):
        //   PH_INIT   first evaluation: slacks from the row values                          -> PH_LS
        //   PH_LS     least-square multiplier system: recursion, multipliers                  -> PH_START
        //   PH_START  evaluation with the multipliers: theta_0 for the filter's limits       -> (as PH_EVAL)
        //   PH_EVAL   errors, convergence tests, barrier parameter                            -> PH_BUILD
        //   PH_BUILD  Newton system with (mu, dw): recursion (Algorithm IC: PH_BUILD again with a larger dw), step, line search, update -> PH_EVAL
        enum { PH_INIT, PH_LS, PH_START, PH_EVAL, PH_BUILD };
        Eval2 E;
        double theta = 0.0, fsum = 0.0;
        double mu = O.mu_init, tau = fmax(O.tau_min, 1.0 - mu);
        double theta_max = INF_, theta_min = 0.0;
        const double mu_min = fmin(O.tol, O.compl_inf_tol) / (O.barrier_tol_factor + 1.0);
        int it = 0, n_acc = 0, status = SC_STATUS_INACCURATE, phase = PH_INIT;
        double last_alpha = 0.0, dw = 0.0;
        bool ic_first = true;
        // the regular algorithm's state while the restoration runs (its filter stays in FP / FT), and the iterate the restoration started from
        int o_nfilt = 0, o_nacc = 0;
        double o_mu = 0.0, o_theta = 0.0, o_phi = 0.0, o_pinf = 0.0, o_dw_last = 0.0, o_theta_max = 0.0, o_theta_min = 0.0;
        bool r_first = false, want_resto = false;
        int n_tiny = 0;                                                    // consecutive accepted steps below stall_alpha (of the phase the solve is in)
        int n_floor = 0;                                                   // consecutive regular iterates at the precision floor (sc_ipopt_params.floor_iter)
        for (;;) {
            MPROF_T0
            if (want_resto) {
                // ---- enter the restoration phase (oracle/ms_ipopt.py: _Algo.restoration) at the current iterate; E, theta, fsum are of this iterate ----
                want_resto = false;
                if (rw == nullptr || rs) { status = rs ? SC_STATUS_INACCURATE : SC_STATUS_NEEDS_RESTO; break; }      // (a failure INSIDE the restoration: resto_failed)
                double pm = 0.0;
                if (stg) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) pm = fmax(pm, fabs(rc[i]));
#pragma unroll
                    for (int j = 0; j < KS; ++j) if (j < K) pm = fmax(pm, fabs(dv[j] - s[j]));
                }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) pm = fmax(pm, fabs(x[i] - x0[i]));
                }
                pm = ipm::wmax(pm);
                if (pm <= O.resto_failure_feasibility_threshold) { status = SC_STATUS_INACCURATE; break; }      // "called at a point that is almost feasible"
                const double phi = barrier(fsum, x, u, s, mu);
                filter_add(phi - O.gamma_phi * theta, (1.0 - O.gamma_theta) * theta);
                o_nfilt = nfilt; o_nacc = n_acc; o_mu = mu; o_theta = theta; o_phi = phi; o_pinf = pm; o_dw_last = dw_last; o_theta_max = theta_max; o_theta_min = theta_min;
                fpo = L.FP2; fto = L.FT2; nfilt = 0; n_acc = 0; dw_last = 0.0;
                mu = fmax(mu, pm); tau = fmax(O.tau_min, 1.0 - mu);
                zeta = O.resto_proximity_weight * sqrt(mu);
                const double rho_R = O.resto_penalty_parameter;
                if (act) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) RW(R_XR + i) = x[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) zxL[i] = fmin(rho_R, zxL[i]);
                    zxU[0] = fmin(rho_R, zxU[0]); zxU[1] = fmin(rho_R, zxU[1]);
                }
                if (stg) {
#pragma unroll
                    for (int j = 0; j < NU; ++j) { RW(R_XR + NX + j) = u[j]; zuL[j] = fmin(rho_R, zuL[j]); zuU[j] = fmin(rho_R, zuU[j]); }
                    RW(R_XR + 10) = rho[0]; RW(R_XR + 11) = rho[1];
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
                        if (j < K) {                                        // eq. (33): the n, p >= 0 that minimise rho (n + p) - mu (log n + log p) on  r + n - p = 0
                            const double r = dv[j] - s[j], a = (mu - rho_R * r) / (2.0 * rho_R);
                            const double n = a + sqrt(a * a + mu * r / (2.0 * rho_R)), pp = r + n;
                            RW(R_N + j) = n; RW(R_P + j) = pp; RW(R_ZN + j) = mu / n; RW(R_ZP + j) = mu / pp;
                            vU[j] = fmin(rho_R, vU[j]); yd[j] = 0.0;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NX; ++i) yc[i] = 0.0;
                }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = 0.0;
                }
                rs = true; r_first = true; n_tiny = 0;
                phase = PH_START;
                continue;
            }
            const bool ls = phase == PH_LS, build = phase == PH_LS || phase == PH_BUILD;
            if (phase == PH_BUILD) eval2<true, false>(E, mu, dw, theta, fsum);
            else if (phase == PH_LS) eval2<true, true>(E, mu, dw, theta, fsum);
            else eval2<false, false>(E, mu, dw, theta, fsum);
            if (phase == PH_INIT) {
#pragma unroll
                for (int j = 0; j < KS; ++j) if (j < K) { double v = dv[j]; push1(v, 0.0, sU[j], false, true); s[j] = v; }
                phase = PH_LS; dw = 0.0; ic_first = true;
                continue;
            }
            if (build) {
                MPROF_ADD(2)
                const double cs = (ls || OD || rs) ? 0.0 : 2.0 * df;         // (the restoration's objective has no input-rate term)
                bool okf = true;
                if constexpr (OD) okf = ipm::wmax((stg && od_bad) ? 1.0 : 0.0) == 0.0;      // a decay block that is not positive definite
                if (okf) okf = riccati_backward(lds, L, N, lane, P.dt, cs * P.R[0], cs * P.R[1], cs * P.R[2], cs * P.R[3]);
                MPROF_ADD(3)
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
                riccati_forward(lds, L, N, lane, P.dt);
                MPROF_ADD(4)
                finish_step(ls, mu, dw);
                MPROF_ADD(5)
                if (ls) {
                    double ym = 0.0;
                    if (stg) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) ym = fmax(ym, fabs(lds[L.LAM + (k + 1) * 6 + i] / dgc(i)));
                    }
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) ym = fmax(ym, fabs(lds[L.LAM + i]));
                    }
#pragma unroll
                    for (int j = 0; j < KS; ++j) ym = fmax(ym, fabs(dyd[j]));
                    ym = ipm::wmax(ym);
#ifdef SC_MS_DBG
                    last_alpha = ym; last_dw = dw;
#endif
                    if (ym <= O.constr_mult_init_max) {
                        if (stg) {
#pragma unroll
                            for (int i = 0; i < NX; ++i) yc[i] = lds[L.LAM + (k + 1) * 6 + i] / dgc(i);
                        }
                        if (lane == 0) {
#pragma unroll
                            for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = -lds[L.LAM + i];
                        }
#pragma unroll
                        for (int j = 0; j < KS; ++j) yd[j] = dyd[j];
                    }
                    phase = PH_START;
                    continue;
                }
                // ---- filter line search ----
                double a_max, a_z, gBD;
                step_lengths(E, tau, mu, a_max, a_z, gBD);
                const double phi = barrier(fsum, x, u, s, mu);
                double a_min = O.gamma_theta;
                if (gBD < 0.0) {
                    a_min = fmin(a_min, O.gamma_phi * theta / (-gBD));
                    if (theta <= theta_min) a_min = fmin(a_min, O.delta * pow(theta, O.s_theta) / pow(-gBD, O.s_phi));
                }
                a_min *= O.alpha_min_frac;
                const double sw_l = gBD < 0.0 ? pow(-gBD, O.s_phi) : 0.0, sw_r = O.delta * pow(theta, O.s_theta);
                double alpha = a_max;
                bool first = true, accepted = false;
                double xt[NX], ut[NU], st[KS], rt[2] = {rho[0], rho[1]};
                double phi_t = 0.0, th_t = 0.0;
                while (alpha > a_min || first) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) xt[i] = x[i] + alpha * (act ? lds[L.DX + k * 6 + i] : 0.0);
#pragma unroll
                    for (int j = 0; j < NU; ++j) ut[j] = u[j] + alpha * (stg ? lds[L.DU + k * NU + j] : 0.0);
#pragma unroll
                    for (int j = 0; j < KS; ++j) st[j] = s[j] + alpha * ds[j];
                    if constexpr (OD) { rt[0] = rho[0] + alpha * drho[0]; rt[1] = rho[1] + alpha * drho[1]; }
                    safe_slacks(xt, ut, st, mu);
                    double f_t;
                    eval0(xt, ut, st, rt, th_t, f_t, alpha);
                    phi_t = barrier(f_t, xt, ut, st, mu, alpha);
                    if (phi_t < INF_ && th_t == th_t && phi_t == phi_t) {
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
                MPROF_ADD(6)
                {
                    const bool ftype = gBD < 0.0 && alpha * sw_l > sw_r;
                    const bool arm = cmp_le(phi_t - phi, O.eta_phi * alpha * gBD, phi);
                    if (!ftype || !arm) filter_add(phi - O.gamma_phi * theta, (1.0 - O.gamma_theta) * theta);
                }
                last_alpha = alpha;
                n_tiny = alpha < O.stall_alpha ? n_tiny + 1 : 0;
#pragma unroll
                for (int i = 0; i < NX; ++i) x[i] = xt[i];
                if (stg) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) yc[i] += alpha * (lds[L.LAM + (k + 1) * 6 + i] / dgc(i));
                }
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) lds[L.Y0 + i] += alpha * (-lds[L.LAM + i]);
                }
#pragma unroll
                for (int j = 0; j < NU; ++j) u[j] = ut[j];
#pragma unroll
                for (int j = 0; j < KS; ++j) { s[j] = st[j]; yd[j] += alpha * dyd[j]; }
                if constexpr (OD) { rho[0] = rt[0]; rho[1] = rt[1]; }
                safe_slacks(x, u, s, mu);
                // bound multipliers: z += a_z dz, then kappa_sigma
                {
                    const double ks = O.kappa_sigma;
                    auto upd = [&](double& z, double dz, double sl) { z += a_z * dz; z = fmax(fmin(z, ks * mu / sl), mu / (ks * sl)); };
                    if (act) {
                        upd(zxL[0], dzxL[0], x[2] - xbL[0]); upd(zxL[1], dzxL[1], x[3] - xbL[1]); upd(zxL[2], dzxL[2], x[4] - xbL[2]);
                        upd(zxU[0], dzxU[0], xbU[0] - x[2]); upd(zxU[1], dzxU[1], xbU[1] - x[3]);
                    }
                    if (stg) {
#pragma unroll
                        for (int j = 0; j < NU; ++j) { upd(zuL[j], dzuL[j], u[j] - ubL[j]); upd(zuU[j], dzuU[j], ubU[j] - u[j]); }
#pragma unroll
                        for (int j = 0; j < KS; ++j) if (j < K) upd(vU[j], dvU[j], sU[j] - s[j]);
                        if (rs) {
#pragma unroll
                            for (int j = 0; j < KS; ++j) {
                                if (j < K) {
                                    const double n = RW(R_N + j) + alpha * RW(R_DN + j), pp = RW(R_P + j) + alpha * RW(R_DP + j);
                                    double zn = RW(R_ZN + j), zp = RW(R_ZP + j);
                                    upd(zn, RW(R_DZN + j), n); upd(zp, RW(R_DZP + j), pp);
                                    RW(R_N + j) = n; RW(R_P + j) = pp; RW(R_ZN + j) = zn; RW(R_ZP + j) = zp;
                                }
                            }
                        }
                    }
                }
                ++it;
                MPROF_ADD(7)
                phase = PH_EVAL;
                continue;
            }
            // ---- PH_START / PH_EVAL: errors, convergence, barrier parameter ----
            MPROF_ADD(0)
            if (phase == PH_START) { theta_max = (rs ? O.resto_theta_max_fact : O.theta_max_fact) * fmax(1.0, theta); theta_min = O.theta_min_fact * fmax(1.0, theta); }
            double E0, dinf, pinf, comp, un_pinf;
            double sc_c = 1.0;
            errors(E, 0.0, E0, dinf, pinf, comp, un_pinf, &sc_c);
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
                eval0(x, u, s, rho, th_o, f_o, 0.0, &pm_o);
                const double phi_o = barrier(f_o, x, u, s, o_mu);
                rs = true;
                bool leave = !r_first && pm_o <= O.required_infeasibility_reduction * o_pinf && phi_o < INF_ && phi_o == phi_o;
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
                    if (act) { zm = fmax(fmax(zxL[0], zxL[1]), fmax(zxL[2], fmax(zxU[0], zxU[1]))); }
                    if (stg) {
#pragma unroll
                        for (int j = 0; j < NU; ++j) zm = fmax(zm, fmax(zuL[j], zuU[j]));
#pragma unroll
                        for (int j = 0; j < KS; ++j) if (j < K) zm = fmax(zm, vU[j]);
                    }
                    zm = ipm::wmax(zm);
                    if (zm > O.bound_mult_reset_threshold) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) zxL[i] = 1.0;
                        zxU[0] = zxU[1] = 1.0;
#pragma unroll
                        for (int j = 0; j < NU; ++j) { zuL[j] = 1.0; zuU[j] = 1.0; }
#pragma unroll
                        for (int j = 0; j < KS; ++j) vU[j] = 1.0;
                    }
#pragma unroll
                    for (int i = 0; i < NX; ++i) yc[i] = 0.0;
#pragma unroll
                    for (int j = 0; j < KS; ++j) yd[j] = 0.0;
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) lds[L.Y0 + i] = 0.0;
                    }
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
            if (!rs && O.floor_iter > 0) {                                  // (see sc_ipopt_params.floor_iter: the iterate is the optimum, the dual infeasibility sits on a precision floor)
                const bool at_floor = mu <= 10.0 * mu_min && n_acc == 0 && fmax(pinf, comp / sc_c) <= O.acceptable_tol && un_pinf <= O.acceptable_constr_viol_tol &&
                                      comp <= O.acceptable_compl_inf_tol * df;
                n_floor = at_floor ? n_floor + 1 : 0;
                if (n_floor >= O.floor_iter) { status = SC_STATUS_INACCURATE; break; }
            }
            if (O.stall_iter > 0 && n_tiny >= O.stall_iter) { status = SC_STATUS_INACCURATE; break; }       // (stall rule: see sc_ipopt_params)
            for (;;) {
                double Emu, a, b, c, d;
                errors(E, mu, Emu, a, b, c, d);
                if (Emu > O.barrier_tol_factor * mu || mu <= mu_min) break;
                const double mu_new = fmax(mu_min, fmin(O.mu_linear_decrease_factor * mu, pow(mu, O.mu_superlinear_decrease_power)));
                if (mu_new == mu) break;
                if (rs) {                                                   // the restoration's objective carries zeta = eta sqrt(mu): its gradient scales with it
                    const double sc_ = sqrt(mu_new / mu);
                    zeta *= sc_;
#pragma unroll
                    for (int i = 0; i < NX; ++i) E.gfx[i] *= sc_;
#pragma unroll
                    for (int j = 0; j < NU; ++j) E.gfu[j] *= sc_;
                    E.gfr[0] *= sc_; E.gfr[1] *= sc_;
                }
                mu = mu_new; tau = fmax(O.tau_min, 1.0 - mu); nfilt = 0;
            }
            r_first = false;
            MPROF_ADD(1)
            phase = PH_BUILD; dw = 0.0; ic_first = true;
        }
#ifdef SC_MS_PROF
        if (trace && lane == 0) { double* t = trace + (size_t)O.max_iter * TRACE_W; for (int i = 0; i < 8; ++i) t[i] = prof[i]; }
#endif
        status_out = status; iters_out = it;
    }
};

template <typename TIO, int KS, bool OD = false>
__global__ void __launch_bounds__(64) mpcvtol_ms_kernel(const Params P, const sc_ipopt_params O, long long B, int obs_shared, const TIO* __restrict__ X,
                                                        const TIO* __restrict__ u_prev, const TIO* __restrict__ goal, const TIO* __restrict__ obs,
                                                        TIO* __restrict__ u_out, int* __restrict__ status_out, int* __restrict__ iters_out,
                                                        TIO* __restrict__ plan_out, double* __restrict__ trace_out, TIO* __restrict__ rho_out) {
    extern __shared__ double ms_lds[];
    const long long b = blockIdx.x;
    if (b >= B) return;
    Wave<KS, OD> S(P, O, (ldsd*)ms_lds);
    if (O.resto_workspace) S.rw = (double*)O.resto_workspace + (size_t)b * (size_t)(Wave<KS, OD>::R_SLOTS * 64);
    const TIO* ob = obs + (obs_shared ? 0 : b * P.K * 7);
    if ((int)threadIdx.x < 3 * KS_MAX) {
        const int j = threadIdx.x / 3, c = threadIdx.x % 3;
        ms_lds[S.L.OB + threadIdx.x] = j < P.K ? (double)ob[7 * j + c] : 0.0;
    }
    for (int i = 0; i < NX; ++i) S.x0[i] = (double)X[b * NX + i];
    for (int j = 0; j < NU; ++j) S.uprev[j] = (double)u_prev[b * NU + j];
    S.xg[0] = (double)goal[b * 2]; S.xg[1] = (double)goal[b * 2 + 1];
    __syncthreads();
    int st, it;
    S.solve(st, it, trace_out ? trace_out + (size_t)b * (size_t)(O.max_iter + 1) * TRACE_W : nullptr);
    if (threadIdx.x == 0) {
        for (int j = 0; j < NU; ++j) u_out[b * NU + j] = (TIO)S.u[j];
        status_out[b] = st;
        if (iters_out) iters_out[b] = it;
    }
    if (plan_out && S.act) {
        // the plan: x_0 .. x_N (6 each), then u_0 .. u_{N-1} (4 each)
        TIO* po = plan_out + b * (long long)((P.N + 1) * NX + P.N * NU);
        for (int i = 0; i < NX; ++i) po[S.k * NX + i] = (TIO)S.x[i];
        if (S.stg) for (int j = 0; j < NU; ++j) po[(P.N + 1) * NX + S.k * NU + j] = (TIO)S.u[j];
    }
    if constexpr (OD) {
        if (rho_out && S.stg) { rho_out[b * (long long)(2 * P.N) + 2 * S.k] = (TIO)S.rho[0]; rho_out[b * (long long)(2 * P.N) + 2 * S.k + 1] = (TIO)S.rho[1]; }
    }
}

template <typename TIO, int KS, bool OD = false>
static hipError_t launch_t(const Params& P, const sc_mpcvtol_params& p, const sc_ipopt_params& O, long long B, size_t lds, const void* X, const void* u_prev,
                           const void* goal, const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out,
                           hipStream_t stream, void* rho_out = nullptr) {
    hipError_t e = hipFuncSetAttribute((const void*)mpcvtol_ms_kernel<TIO, KS, OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mpcvtol_ms_kernel<TIO, KS, OD>), dim3((unsigned)B), dim3(64), lds, stream, P, O, B, p.obs_shared, (const TIO*)X, (const TIO*)u_prev,
                       (const TIO*)goal, (const TIO*)obs, (TIO*)u_out, status_out, iters_out, (TIO*)plan_out, trace_out, (TIO*)rho_out);
    return hipGetLastError();
}

}  // namespace msk

size_t mpcvtol_ms_lds_bytes(int horizon) { return msk::lds_bytes(horizon); }

hipError_t mpcvtol_ms_launch(const sc_mpcvtol_params& p, const sc_ipopt_params& O, long long B, int K, const void* X, const void* u_prev, const void* goal,
                             const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out, hipStream_t stream) {
    const Params P = from_c(p, K);
    const size_t lds = msk::lds_bytes(p.horizon);
    if (p.io_dtype == SC_DTYPE_F64)
        return K <= 8 ? msk::launch_t<double, 8>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream)
                      : msk::launch_t<double, 16>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream);
    return K <= 8 ? msk::launch_t<float, 8>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream)
                  : msk::launch_t<float, 16>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream);
}

// optimal-decay MPC-CBF for VTOL2D in the multiple-shooting form (sc_odmpcvtol_params: the condensed entry's struct)
hipError_t odmpcvtol_ms_launch(const sc_odmpcvtol_params& q, const sc_ipopt_params& O, long long B, int K, const void* X, const void* u_prev, const void* goal,
                               const void* obs, void* u_out, void* rho_out, int* status_out, int* iters_out, void* plan_out, double* trace_out,
                               hipStream_t stream) {
    const sc_mpcvtol_params& p = q.mpc;
    Params P = from_c(p, K);
    P.ps1 = q.p_sb[0]; P.ps2 = q.p_sb[1]; P.rf1 = q.omega_ref[0]; P.rf2 = q.omega_ref[1];
    const size_t lds = msk::lds_bytes(p.horizon);
    if (p.io_dtype == SC_DTYPE_F64)
        return K <= 8 ? msk::launch_t<double, 8, true>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream, rho_out)
                      : msk::launch_t<double, 16, true>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream, rho_out);
    return K <= 8 ? msk::launch_t<float, 8, true>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream, rho_out)
                  : msk::launch_t<float, 16, true>(P, p, O, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream, rho_out);
}

}  // namespace sc
