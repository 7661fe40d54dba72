// MPC-CBF for VTOL2D, the kernel that serves it: ONE NLP PER WAVEFRONT, ONE STAGE PER LANE (K <= 16 obstacles, N <= 64).
//
// Same problem and same interior point as mpc_vtol_solver.hpp (the lane-per-problem code, which stays as the statement this kernel
// was developed and is still checked against: sc_mpcvtol_params.kernel = 1); what changes is who does what:
//   * lane k owns stage k: its input u_k, its K + 13 rows (CBF rows of (x_k, u_k), the five state bounds of x_{k+1}, the input box) with
//     their slacks, multipliers and steps IN REGISTERS (instantiated for 8 and 16 CBF row slots, slots >= K switched off), the aero
//     model's derivatives at (x_k, u_k), its stage block of the Newton system;
//   * the nonlinear rollout is a serial recursion: every lane runs it (wave-uniform, same cost as one lane) and keeps x_k, x_{k+1} and
//     the acceleration when the loop passes its stage, so a function evaluation is one rollout + one parallel row pass;
//   * the costate sweep is a 6-vector recursion over stage vectors the lanes left in LDS (six lanes, one barrier per stage); the Riccati
//     recursion walks the stages backwards with the 10 x 10 value function in LDS and the lanes spread over the ENTRIES of each product
//     (P [A | B], Qux | Quu, the new P) in four branch-free phases per stage; the 4 x 4 Cholesky and its eleven right-hand sides are
//     done redundantly, one column per lane; the forward LQ rollout takes four + six lanes and two barriers per stage;
//   * sums and maxima over rows are DPP wave reductions; every decision of the interior point is wave-uniform.
// LDS per problem: stage Jacobians and blocks, gains, costates, the Riccati workspace -- 39.3 KB for N = 30 (slots with disjoint lifetimes
// shared, WaveLds), four problems per CU, one per SIMD (512 VGPRs).  The serial routines are noinline functions on LDS-address-space
// pointers: compiled without the row state's register pressure.
// Kernel 11 in DESIGN.md.
#include <hip/hip_runtime.h>

#include "../../include/safe_control_amd.h"
#include "mpc_ipm_common.hpp"
#include "mpc_cont.hpp"
#define SC_VTOL_WITH_C_PARAMS
#define SC_VTOL_RCP(a) sc::rcp_(a)
#define SC_VTOL_SINCOS(a, s, c) sc::sincos_((a), &(s), &(c));
#include "mpc_vtol_solver.hpp"

namespace sc {

using namespace vtol;

#ifdef SC_VTOL_PROF
#define VPROF_T0 long long _t0 = __builtin_readcyclecounter();
#define VPROF_ADD(i) { const long long _t1 = __builtin_readcyclecounter(); prof[i] += _t1 - _t0; _t0 = _t1; }
#else
#define VPROF_T0
#define VPROF_ADD(i)
#endif

constexpr int ST_PENDING = -1;      // Wave::solve: stopped at the cap of this launch (or classified only)
constexpr int WKT_MAX = 16;            // CBF rows per stage held in registers: instantiated for 8 and 16

struct WaveLds {
    int U, UT, DU, AB, H, Q, XD, XQ, PS, OB, Pm, Pn, pv, pn, PAB, Quu, QX, qu, UP, Dl, DX, total;
    // Lifetimes that do not overlap share a slot (39.3 KB per problem for N = 30: four problems per CU, one per SIMD):
    //   gains KK_k | kk_k (44 numbers) of the Riccati recursion go into the slot of H_{k+1}, which stage k + 1 has finished with
    //     (H has N + 1 slots; a retry with a larger delta rebuilds the stage blocks first);
    //   dx of the forward LQ rollout overwrites the costates (read for the last time when the stage blocks are built);
    //   the trial inputs UT and the step DU overwrite the own terms XD, XQ (read for the last time by the Riccati recursion).
    __host__ __device__ explicit WaveLds(int N) {
        int o = 0;
        auto take = [&](int c) { int r = o; o += c; return r; };
        U = take(N * 4); AB = take(N * 60); H = take((N + 1) * 55); Q = take(N * 10);
        XD = take(N * 6); XQ = take(N * 6); PS = take((N + 1) * 6); OB = take(3 * WKT_MAX);
        Pm = take(100); Pn = take(100); pv = take(10); pn = take(10); PAB = take(100); Quu = take(16); QX = take(40); qu = take(4);
        UP = take(4); Dl = take(4);
        UT = XD; DX = PS; DU = XQ;
        total = o;
    }
};
#define VT_KK(kk) (lds + L.H + ((kk) + 1) * 55)
#define VT_kk(kk) (lds + L.H + ((kk) + 1) * 55 + 40)

size_t mpcvtol_wave_lds_bytes(int horizon) { return (size_t)WaveLds(horizon).total * sizeof(double); }

typedef __attribute__((address_space(3))) double ldsd;

// [A_k | B_k] of stage kk as one 6 x 10 row-major block: AB[i * 10 + c], c < 6: A, c >= 6: B
#define VT_AB(kk) (lds + L.AB + (kk) * 60)

// ---- Riccati recursion, lanes over matrix entries; false (wave-uniform) when an input block is not positive definite --------
// Value function of stage kk+1 in buffer Pc (10 x 10 full, symmetric by construction) with the own terms of x_{kk+1} already
// added; four barrier-separated, branch-free phases per stage:
//   A  P [A | B] (+ P's input columns under B): one 10 x 10 product, two rounds of 64 entries
//   B  [Qux | Quu] = rows 6.. of the product + H + B' (product), 40 lanes; qu on four more
//   C  the 4 x 4 factorisation in every lane (reciprocal square roots of the pivots) + one right-hand side per lane
//   D  the new P (upper triangle, mirrored on the write) and p, own terms of x_kk folded in for the next stage
__device__ __attribute__((noinline)) bool vtol_riccati(ldsd* lds, const WaveLds L, const int N, const int lane, const double shift) {
    ldsd* PAB = lds + L.PAB; ldsd* Quu = lds + L.Quu; ldsd* QX = lds + L.QX; ldsd* qu = lds + L.qu;
    const ldsd* Dl = lds + L.Dl;
    ldsd* Pc = lds + L.Pm; ldsd* Pn = lds + L.Pn; ldsd* pc = lds + L.pv; ldsd* pn = lds + L.pn;
    // lane-constant index maps
    const int ra = lane / 10, ca_ = lane % 10;                            // phase A, round 0: entry `lane` of 100
    const int e1 = lane < 36 ? lane + 64 : 99;
    const int rb = e1 / 10, cb = e1 % 10;                                 // round 1 (lanes < 36)
    const int bi = (lane < 40 ? lane : 0) / 10, bc = (lane < 40 ? lane : 0) % 10;   // phase B: [Qux | Quu] entry (bi, bc)
    int tr = 0, tc = 0;                                                   // phase D: upper-triangle entry `lane` of 55
    { int e = lane < 55 ? lane : 0; int r = 0; while (e >= NV - r) { e -= NV - r; ++r; } tr = r; tc = r + e; }
    const bool dxx = tc < NX;                                             // entry in the state block: has the A' (P A) term
    const int tcc = dxx ? tc : 0, trr = dxx ? tr : 0;
    for (int e = lane; e < 100; e += 64) { const int r = e / 10, c = e % 10; Pc[e] = (r == c && r < NX) ? lds[L.XD + (N - 1) * 6 + r] : 0.0; }
    if (lane < NV) pc[lane] = lane < NX ? lds[L.XQ + (N - 1) * 6 + lane] : 0.0;
    if (lane < 16) { const int i = lane / 4, c = lane % 4; QX[i * 10 + 6 + c] = (c == i) ? -Dl[i] : 0.0; }
    __syncthreads();
    for (int kk = N - 1; kk >= 0; --kk) {
        const ldsd* AB = VT_AB(kk); const ldsd* H = lds + L.H + kk * 55;
        const ldsd* q = lds + L.Q + kk * 10;
        const ldsd* Uk = lds + L.U + kk * NU; const ldsd* Um = kk ? lds + L.U + (kk - 1) * NU : lds + L.UP;   // rD_i = Dl_i (u_k - u_{k-1})_i
        // A
        {
            double a0[NX], b0[NX], a1[NX], b1[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { a0[i] = Pc[ra * 10 + i]; b0[i] = AB[i * 10 + ca_]; a1[i] = Pc[rb * 10 + i]; b1[i] = AB[i * 10 + cb]; }
            double v0 = ca_ >= NX ? Pc[ra * 10 + ca_] : 0.0, v1 = cb >= NX ? Pc[rb * 10 + cb] : 0.0;
#pragma unroll
            for (int i = 0; i < NX; ++i) { v0 += a0[i] * b0[i]; v1 += a1[i] * b1[i]; }
            PAB[lane] = v0;
            if (lane < 36) PAB[e1] = v1;
        }
        __syncthreads();
        // B
        {
            double a0[NX], b0[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) { a0[r] = AB[r * 10 + 6 + bi]; b0[r] = PAB[r * 10 + bc]; }
            double v = PAB[(6 + bi) * 10 + bc] + H[sym(bc, 6 + bi)];
#pragma unroll
            for (int r = 0; r < NX; ++r) v += a0[r] * b0[r];
            if (bc - NX == bi) v += Dl[bi] + shift;
            if (lane < 40) { if (bc < NX) QX[bi * 10 + bc] = v; else Quu[bi * 4 + bc - NX] = v; }
            if (lane >= 40 && lane < 44) {
                const int i = lane - 40;
                double w = q[6 + i] - Dl[i] * (Uk[i] - Um[i]) + pc[6 + i];
#pragma unroll
                for (int r = 0; r < NX; ++r) w += AB[r * 10 + 6 + i] * pc[r];
                qu[i] = w;
            }
        }
        __syncthreads();
        // C
        {
            const double q00 = Quu[0], q10 = Quu[4], q11 = Quu[5], q20 = Quu[8], q21 = Quu[9], q22 = Quu[10], q30 = Quu[12], q31 = Quu[13],
                         q32 = Quu[14], q33 = Quu[15];
            const int c = lane < 11 ? lane : 0;
            const ldsd* bsrc = c < NV ? QX + c : qu;
            const int bst = c < NV ? 10 : 1;
            const double b0 = bsrc[0], b1 = bsrc[bst], b2 = bsrc[2 * bst], b3 = bsrc[3 * bst];
            const double r0 = rsqrt_(q00), l10 = q10 * r0, l20 = q20 * r0, l30 = q30 * r0;
            const double d1 = q11 - l10 * l10, r1 = rsqrt_(d1), l21 = (q21 - l20 * l10) * r1, l31 = (q31 - l30 * l10) * r1;
            const double d2 = q22 - l20 * l20 - l21 * l21, r2 = rsqrt_(d2), l32 = (q32 - l30 * l20 - l31 * l21) * r2;
            const double d3 = q33 - l30 * l30 - l31 * l31 - l32 * l32, r3 = rsqrt_(d3);
            if (!(q00 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0) || !(d3 > 0.0)) return false;     // wave-uniform: every lane factors the same block
            const double y0 = b0 * r0, y1 = (b1 - l10 * y0) * r1, y2 = (b2 - l20 * y0 - l21 * y1) * r2, y3 = (b3 - l30 * y0 - l31 * y1 - l32 * y2) * r3;
            const double x3 = y3 * r3, x2 = (y2 - l32 * x3) * r2, x1 = (y1 - l21 * x2 - l31 * x3) * r1, x0_ = (y0 - l10 * x1 - l20 * x2 - l30 * x3) * r0;
            if (lane < 11) {
                ldsd* dst = c < NV ? VT_KK(kk) + c : VT_kk(kk);
                dst[0] = x0_; dst[bst] = x1; dst[2 * bst] = x2; dst[3 * bst] = x3;
            }
        }
        __syncthreads();
        // D
        {
            const ldsd* KKm = VT_KK(kk); const ldsd* kkv = VT_kk(kk);
            double a0[NX], b0[NX], c0[NU], d0[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) { a0[i] = AB[i * 10 + trr]; b0[i] = PAB[i * 10 + tcc]; }
#pragma unroll
            for (int i = 0; i < NU; ++i) { c0[i] = QX[i * 10 + tr]; d0[i] = KKm[i * NV + tc]; }
            double v = H[sym(trr, tcc)];
#pragma unroll
            for (int i = 0; i < NX; ++i) v += a0[i] * b0[i];
            v = dxx ? v : ((tr == tc) ? Dl[tr >= NX ? tr - NX : 0] : 0.0);
#pragma unroll
            for (int i = 0; i < NU; ++i) v -= c0[i] * d0[i];
            if (tr == tc && tr < NX && kk >= 1) v += lds[L.XD + (kk - 1) * 6 + tr];
            if (lane < 55) { Pn[tr * 10 + tc] = v; Pn[tc * 10 + tr] = v; }
            if (lane < NV) {
                const int r = lane, rx = r < NX ? r : 0;
                double w = q[rx];
#pragma unroll
                for (int i = 0; i < NX; ++i) w += AB[i * 10 + rx] * pc[i];
                if (kk >= 1) w += lds[L.XQ + (kk - 1) * 6 + rx];
                { const int ri = r >= NX ? r - NX : 0; w = r < NX ? w : Dl[ri] * (Uk[ri] - Um[ri]); }
#pragma unroll
                for (int i = 0; i < NU; ++i) w -= QX[i * 10 + r] * kkv[i];
                pn[r] = w;
            }
        }
        __syncthreads();
        { ldsd* tP = Pc; Pc = Pn; Pn = tP; ldsd* tp = pc; pc = pn; pn = tp; }
    }
    return true;
}

// ---- forward LQ rollout: four lanes for du, six for dx, two barriers per stage ---------------------------------------------------
__device__ __attribute__((noinline)) void vtol_lq_forward(ldsd* lds, const WaveLds L, const int N, const int lane) {
    if (lane < NX) lds[L.DX + lane] = 0.0;
    __syncthreads();
    const int li = lane < NU ? lane : 0, lx = lane < NX ? lane : 0;
    for (int kk = 0; kk < N; ++kk) {
        const ldsd* AB = VT_AB(kk);
        {
            double a0[NV], b0[NV];
#pragma unroll
            for (int c = 0; c < NX; ++c) { a0[c] = VT_KK(kk)[li * NV + c]; b0[c] = lds[L.DX + kk * 6 + c]; }
#pragma unroll
            for (int c = 0; c < NU; ++c) { a0[6 + c] = VT_KK(kk)[li * NV + 6 + c]; b0[6 + c] = kk > 0 ? lds[L.DU + (kk - 1) * NU + c] : 0.0; }
            double v = VT_kk(kk)[li];
#pragma unroll
            for (int c = 0; c < NV; ++c) v -= a0[c] * b0[c];
            if (lane < NU) lds[L.DU + kk * NU + lane] = v;
        }
        __syncthreads();
        {
            double a0[NV], b0[NV];
#pragma unroll
            for (int c = 0; c < NX; ++c) { a0[c] = AB[lx * 10 + c]; b0[c] = lds[L.DX + kk * 6 + c]; }
#pragma unroll
            for (int j = 0; j < NU; ++j) { a0[6 + j] = AB[lx * 10 + 6 + j]; b0[6 + j] = lds[L.DU + kk * NU + j]; }
            double v = 0.0;
#pragma unroll
            for (int c = 0; c < NV; ++c) v += a0[c] * b0[c];
            if (lane < NX) lds[L.DX + (kk + 1) * 6 + lane] = v;
        }
        __syncthreads();
    }
}

// ---- costate recursion  p_j = own_j + (j < N ? cx_j + A_j' p_{j+1} : 0),  j = N .. 1  (own_j sits in slot j - 1): six lanes ----------
__device__ __attribute__((noinline)) void vtol_costates(ldsd* lds, const WaveLds L, const int N, const int lane) {
    const int c = lane < NX ? lane : 0;
    if (lane < NX) lds[L.PS + N * 6 + lane] = lds[L.XQ + (N - 1) * 6 + lane];
    __syncthreads();
    for (int j = N - 1; j >= 1; --j) {
        const ldsd* AB = VT_AB(j);
        double a0[NX], b0[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { a0[i] = AB[i * 10 + c]; b0[i] = lds[L.PS + (j + 1) * 6 + i]; }
        double v = lds[L.XQ + (j - 1) * 6 + c] + lds[L.Q + j * 10 + c];
#pragma unroll
        for (int i = 0; i < NX; ++i) v += a0[i] * b0[i];
        if (lane < NX) lds[L.PS + j * 6 + lane] = v;
        __syncthreads();
    }
}

// OD: the optimal-decay NLP (optimal_decay_mpc_cbf.py:288-296; oracle/od_mpc_vtol.py): two decay variables per stage, which enter only
// that stage's CBF rows (through the stage weights w0 = 1 - s + q, w1 = s - 2, s = a1 rho1 + a2 rho2, q = a1 a2 rho1 rho2) and the cost
// p_sb (rho - ref)^2; their 2 x 2 block D_k is eliminated from the stage block before the Riccati recursion (H_k -= C_k D_k^-1 C_k',
// q_k -= C_k D_k^-1 rr_k, d rho_k = D_k^-1 (rr_k - C_k' (dx_k, du_k)) afterwards), the input term is R u^2 (no coupling between the stages'
// inputs: Dl = 0 in the recursion), no restoration phase (the optimal-decay oracle has none).
template <int WKT, bool OD = false>
struct Wave {
    static constexpr int WNR = WKT + 13;          // + 5 state bounds of x_{k+1} + 4 + 4 input box
    const Params& P;
    ldsd* lds;
    const WaveLds L;
    const int lane, N, K;
    const bool act;
    const int k;                       // my stage (0 for idle lanes, which contribute nothing)
    double x0[NX], uprev[NU], xg[2];
    double w0, w1, w2;
    // rows of my stage
    double g[WNR], s[WNR], lam[WNR], ds[WNR], dlam[WNR], t[WKT], dtt[WKT];
    // my stage's state, next state, acceleration (from the last evaluation), rows 3, 4 of [A B]
    double xk[NX], xk1[NX], ak[2], a34[2][NV];
    double zR[NU], zb[NU];
    // optimal decay: my stage's decay variables, their best iterate, and what the elimination of stage_blocks leaves for the step
    double dk[2], dkb[2], Cm[NV][2], rrv[2], dvx, dvy, dils, dilw;
#ifdef SC_VTOL_PROF
    long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // eval, linearise, adjoint, blocks, riccati, lq forward, rows + rest, trials
#endif

    __device__ __forceinline__ Wave(const Params& P_, ldsd* lds_) : P(P_), lds(lds_), L(P_.N), lane(threadIdx.x), N(P_.N), K(P_.K), act((int)threadIdx.x < P_.N),
                                                    k((int)threadIdx.x < P_.N ? (int)threadIdx.x : 0) {
        const double g1 = P.alpha1 + P.alpha2, g2 = P.alpha1 * P.alpha2;
        w0 = 1.0 - g1 + g2; w1 = g1 - 2.0; w2 = 1.0;
    }
    __device__ __forceinline__ bool valid(int r) const { return act && (r >= WKT || r < K); }
    // stage weights of the rows for decay variables (r1, r2)
    __device__ __forceinline__ void od_weights(double r1, double r2, double& a, double& b) const {
        const double sk = P.alpha1 * r1 + P.alpha2 * r2, qk = P.alpha1 * P.alpha2 * r1 * r2;
        a = 1.0 - sk + qk; b = sk - 2.0;
    }
    // h at the first two barrier points of obstacle j
    __device__ __forceinline__ void od_h01(const double pt[3][2], int j, double& h0, double& h1) const {
        const double cx = lds[L.OB + 3 * j], cz = lds[L.OB + 3 * j + 1], d = P.radius + lds[L.OB + 3 * j + 2], off = P.beta * d * d;
        const double e0x = pt[0][0] - cx, e0z = pt[0][1] - cz, e1x = pt[1][0] - cx, e1z = pt[1][1] - cz;
        h0 = e0x * e0x + e0z * e0z - off; h1 = e1x * e1x + e1z * e1z - off;
    }
    // d row / d rho_1, d rho_2
    __device__ __forceinline__ void od_row_drho(double h0, double h1, double& A1, double& A2) const {
        const double aa = P.alpha1 * P.alpha2 * h0;
        A1 = P.alpha1 * (h1 - h0) + aa * dk[1]; A2 = P.alpha2 * (h1 - h0) + aa * dk[0];
    }
    __device__ __forceinline__ void sync() const { __syncthreads(); }

    // ---- function evaluation: rollout (every lane) + my rows -----------------------------------------------------------------
    __device__ __forceinline__ void rows_from_state(const double* u, double* go, double w0, double w1) const {
        double pt[3][2];
        const double dt = P.dt;
        pt[0][0] = xk[0]; pt[0][1] = xk[1];
        pt[1][0] = xk[0] + dt * xk[3]; pt[1][1] = xk[1] + dt * xk[4];
        pt[2][0] = pt[1][0] + dt * (xk[3] + dt * ak[0]); pt[2][1] = pt[1][1] + dt * (xk[4] + dt * ak[1]);
#pragma unroll
        for (int j = 0; j < WKT; ++j) {
            double v = 0.0;
            if (j < K) {
                const double cx = lds[L.OB + 3 * j], cz = lds[L.OB + 3 * j + 1], d = P.radius + lds[L.OB + 3 * j + 2], off = P.beta * d * d;
                double hv[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) { const double ex = pt[p][0] - cx, ez = pt[p][1] - cz; hv[p] = ex * ex + ez * ez - off; }
                v = w0 * hv[0] + w1 * hv[1] + w2 * hv[2];
            }
            go[j] = v;
        }
        go[WKT + 0] = P.v_max - xk1[3]; go[WKT + 1] = xk1[3] + P.v_max; go[WKT + 2] = xk1[4] + P.descent_max;
        go[WKT + 3] = P.pitch_max - xk1[2]; go[WKT + 4] = xk1[2] + P.pitch_max;
#pragma unroll
        for (int j = 0; j < NU; ++j) { go[WKT + 5 + j] = P.u_hi[j] - u[j]; go[WKT + 9 + j] = u[j] - P.u_lo[j]; }
    }
    // rollout over the inputs at LDS offset zo; returns the unscaled cost, rows to go
    __device__ __forceinline__ double eval(int zo, double* go, double r1 = 1.0, double r2 = 1.0) {
        double x[NX], xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = x0[i];
        for (int kk = 0; kk < N; ++kk) {
            double u[NU], acc[3], gc[4][3];
#pragma unroll
            for (int j = 0; j < NU; ++j) u[j] = lds[zo + kk * NU + j];
            accel<double>(P, x[2], x[3], x[4], u, acc, gc);
            xn[0] = x[0] + P.dt * x[3]; xn[1] = x[1] + P.dt * x[4]; xn[2] = x[2] + P.dt * x[5];
            xn[3] = x[3] + P.dt * acc[0]; xn[4] = x[4] + P.dt * acc[1]; xn[5] = x[5] + P.dt * acc[2];
            if (kk == k) {
#pragma unroll
                for (int i = 0; i < NX; ++i) { xk[i] = x[i]; xk1[i] = xn[i]; }
                ak[0] = acc[0]; ak[1] = acc[1];
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = xn[i];
        }
        double u[NU], f = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            u[j] = lds[zo + k * NU + j];
            const double um = k ? lds[zo + (k - 1) * NU + j] : uprev[j], du = OD ? u[j] : u[j] - um;     // OD: R u^2 (optimal_decay_mpc_cbf.py:173-179)
            f += P.R[j] * du * du;
        }
        double wa = w0, wb = w1;
        if constexpr (OD) {
            od_weights(r1, r2, wa, wb);
            f += P.ps1 * (r1 - P.rf1) * (r1 - P.rf1) + P.ps2 * (r2 - P.rf2) * (r2 - P.rf2);
        }
        {
            const double e0 = xk1[0] - xg[0], e1 = xk1[1] - xg[1];
            f += P.Q[0] * e0 * e0 + P.Q[1] * e1 * e1 + P.Q[2] * xk1[2] * xk1[2] + P.Q[3] * xk1[3] * xk1[3] + P.Q[4] * xk1[4] * xk1[4] +
                 P.Q[5] * xk1[5] * xk1[5];
        }
        rows_from_state(u, go, wa, wb);
        return ipm::wsum(act ? f : 0.0);
    }

    // ---- derivatives of my stage -----------------------------------------------------------------------------------------------
    __device__ __forceinline__ void accel_d2(D2 acc[3], D2 gc[4][3]) const {
        double u[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) u[j] = lds[L.U + k * NU + j];
        accel<D2>(P, d2var(xk[2], 0), d2var(xk[3], 1), d2var(xk[4], 2), u, acc, gc);
    }
    __device__ __forceinline__ void linearise() {
        D2 acc[3], gc[4][3];
        accel_d2(acc, gc);
        double Ak[36], Bk[24];
#pragma unroll
        for (int i = 0; i < 36; ++i) Ak[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NX; ++i) Ak[i * 6 + i] = 1.0;
        Ak[0 * 6 + 3] = P.dt; Ak[1 * 6 + 4] = P.dt; Ak[2 * 6 + 5] = P.dt;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) Ak[(3 + i) * 6 + (2 + c)] += P.dt * acc[i].d[c];
#pragma unroll
        for (int i = 0; i < 24; ++i) Bk[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < NU; ++j) Bk[(3 + i) * 4 + j] = P.dt * gc[j][i].v;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < NX; ++i) a34[c][i] = Ak[(3 + c) * 6 + i];
#pragma unroll
            for (int j = 0; j < NU; ++j) a34[c][6 + j] = Bk[(3 + c) * 4 + j];
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
#pragma unroll
                for (int c = 0; c < NX; ++c) lds[L.AB + k * 60 + i * 10 + c] = Ak[i * 6 + c];
#pragma unroll
                for (int j = 0; j < NU; ++j) lds[L.AB + k * 60 + i * 10 + 6 + j] = Bk[i * 4 + j];
            }
        }
    }
    __device__ __forceinline__ void points(double pt[3][2]) const {
        const double dt = P.dt;
        pt[0][0] = xk[0]; pt[0][1] = xk[1];
        pt[1][0] = xk[0] + dt * xk[3]; pt[1][1] = xk[1] + dt * xk[4];
        pt[2][0] = pt[1][0] + dt * (xk[3] + dt * ak[0]); pt[2][1] = pt[1][1] + dt * (xk[4] + dt * ak[1]);
    }
    __device__ __forceinline__ void point_jac(double G2[2][NV]) const {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < NV; ++i) G2[c][i] = P.dt * a34[c][i];
            G2[c][c] += 1.0; G2[c][3 + c] += P.dt;
        }
    }
    __device__ __forceinline__ void cbf_row_grad(const double pt[3][2], const double G2[2][NV], int j, double r[NV]) const {
        const double cx = lds[L.OB + 3 * j], cz = lds[L.OB + 3 * j + 1];
        const double e0x = pt[0][0] - cx, e0z = pt[0][1] - cz, e1x = pt[1][0] - cx, e1z = pt[1][1] - cz, e2x = pt[2][0] - cx, e2z = pt[2][1] - cz;
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = 2.0 * w2 * (e2x * G2[0][i] + e2z * G2[1][i]);
        r[0] += 2.0 * (w0 * e0x + w1 * e1x); r[1] += 2.0 * (w0 * e0z + w1 * e1z);
        r[3] += 2.0 * w1 * e1x * P.dt; r[4] += 2.0 * w1 * e1z * P.dt;
    }

    // ---- costate sweep: |grad f - J' lam|_inf, costates to LDS ----------------------------------------------------------------
    __device__ __forceinline__ double adjoint(bool with_rows, double cw, bool resto, double zeta) {
        double cx[NX], cu[NU], own[NX];
        double rdr1 = 0.0, rdr2 = 0.0;                                     // OD: r_d of my decay variables
#pragma unroll
        for (int i = 0; i < NX; ++i) cx[i] = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) cu[j] = 0.0;
        if constexpr (OD) { rdr1 = 2.0 * cw * P.ps1 * (dk[0] - P.rf1); rdr2 = 2.0 * cw * P.ps2 * (dk[1] - P.rf2); }
        if (with_rows) {
            double pt[3][2], G2[2][NV];
            points(pt); point_jac(G2);
#pragma unroll
            for (int j = 0; j < WKT; ++j)
                if (j < K) {
                    double r[NV]; cbf_row_grad(pt, G2, j, r);
#pragma unroll
                    for (int i = 0; i < NX; ++i) cx[i] -= lam[j] * r[i];
#pragma unroll
                    for (int i = 0; i < NU; ++i) cu[i] -= lam[j] * r[6 + i];
                    if constexpr (OD) {
                        double h0, h1, A1, A2;
                        od_h01(pt, j, h0, h1); od_row_drho(h0, h1, A1, A2);
                        rdr1 -= lam[j] * A1; rdr2 -= lam[j] * A2;
                    }
                }
#pragma unroll
            for (int j = 0; j < NU; ++j) cu[j] += lam[WKT + 5 + j] - lam[WKT + 9 + j];
        }
        own[0] = 2.0 * cw * P.Q[0] * (xk1[0] - xg[0]); own[1] = 2.0 * cw * P.Q[1] * (xk1[1] - xg[1]);
#pragma unroll
        for (int i = 2; i < NX; ++i) own[i] = 2.0 * cw * P.Q[i] * xk1[i];
        if (with_rows) { own[3] += lam[WKT + 0] - lam[WKT + 1]; own[4] -= lam[WKT + 2]; own[2] += lam[WKT + 3] - lam[WKT + 4]; }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const double uk = lds[L.U + k * NU + j], um = k ? lds[L.U + (k - 1) * NU + j] : uprev[j];
            double gj = 2.0 * cw * P.R[j] * (OD ? uk : uk - um);
            if (!OD && k + 1 < N) gj -= 2.0 * cw * P.R[j] * (lds[L.U + (k + 1) * NU + j] - uk);
            if (resto) gj += zeta * (uk - zR[j]);
            cu[j] += gj;
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < NX; ++i) { lds[L.Q + k * 10 + i] = cx[i]; lds[L.XQ + k * 6 + i] = own[i]; }
#pragma unroll
            for (int j = 0; j < NU; ++j) lds[L.Q + k * 10 + 6 + j] = cu[j];
        }
        sync();
        vtol_costates(lds, L, N, lane);
        double rd = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            double r = cu[j];
#pragma unroll
            for (int i = 0; i < NX; ++i) r += lds[L.AB + k * 60 + i * 10 + 6 + j] * lds[L.PS + (k + 1) * 6 + i];
            rd = fmax(rd, fabs(r));
        }
        if constexpr (OD) rd = fmax(rd, fmax(fabs(rdr1), fabs(rdr2)));
        return ipm::wmax(act ? rd : 0.0);
    }

    // Sigma and the multiplier step at dz = 0 of row r
    __device__ __forceinline__ void row_sig(int r, bool resto, double mu, double& sg, double& d0) const {
        const double is = rcp_(s[r]);
        sg = lam[r] * is;
        d0 = -sg * (g[r] - s[r]) - lam[r] + mu * is;
    }
    __device__ __forceinline__ void row_sig_el(int j, double mu, double rho, double& sg, double& d0) const {
        const double l = lam[j], tt = t[j], nut = rho - l, is = rcp_(s[j]), sgs = l * is, sgt = nut * rcp_(tt), se = sgs * sgt * rcp_(sgs + sgt),
                     rp = g[j] + tt - s[j];
        d0 = -se * (rp + mu * rcp_(nut) - tt) - (se * rcp_(sgs)) * (l - mu * is);
        sg = se;
    }

    // ---- my stage block of the Newton system to LDS -----------------------------------------------------------------------------
    __device__ __forceinline__ void stage_blocks(double cw, bool resto, double zeta, double mu, double rho_R) {
        double H[55], q[NV];
        const double so = (resto && P.resto_gn) ? 0.0 : 1.0;
#pragma unroll
        for (int i = 0; i < 55; ++i) H[i] = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) q[i] = 0.0;
        D2 acc[3], gc[4][3];
        accel_d2(acc, gc);
        double pt[3][2], G2[2][NV];
        points(pt); point_jac(G2);
        double slam = 0.0, nu2[2] = {0.0, 0.0};
        double od11 = 0.0, od12 = 0.0, od22 = 0.0;
        if constexpr (OD) {
#pragma unroll
            for (int a = 0; a < NV; ++a) { Cm[a][0] = 0.0; Cm[a][1] = 0.0; }
            rrv[0] = -2.0 * cw * P.ps1 * (dk[0] - P.rf1); rrv[1] = -2.0 * cw * P.ps2 * (dk[1] - P.rf2);
        }
#pragma unroll
        for (int j = 0; j < WKT; ++j)
            if (j < K) {
                double r[NV]; cbf_row_grad(pt, G2, j, r);
                double sg, d0;
                if (resto) row_sig_el(j, mu, rho_R, sg, d0); else row_sig(j, false, mu, sg, d0);
                const double lq = lam[j] + d0, l = lam[j];
#pragma unroll
                for (int a = 0; a < NV; ++a) {
                    q[a] += lq * r[a];
#pragma unroll
                    for (int b = a; b < NV; ++b) H[sym(a, b)] += sg * r[a] * r[b];
                }
                if constexpr (OD) {
                    // decay block of my stage: C[a][i] = sum_j sig r_a A_i - lam x_i[a] (x_i: d2 row / d rho_i d(x, u) = a_i (grad h1 - grad h0) +
                    // a1 a2 rho_other grad h0), D = sum_j sig A A' - lam a1 a2 h0 [[0, 1], [1, 0]], rr = sum_j (lam + dl0) A
                    double h0, h1, A1, A2;
                    od_h01(pt, j, h0, h1); od_row_drho(h0, h1, A1, A2);
                    const double cx_ = lds[L.OB + 3 * j], cz_ = lds[L.OB + 3 * j + 1];
                    const double e0x = pt[0][0] - cx_, e0z = pt[0][1] - cz_, e1x = pt[1][0] - cx_, e1z = pt[1][1] - cz_;
                    const double m1 = -P.alpha1 + P.alpha1 * P.alpha2 * dk[1], m2 = -P.alpha2 + P.alpha1 * P.alpha2 * dk[0];   // weight of grad h0 in x_i
                    double x1[NV], x2[NV];
#pragma unroll
                    for (int a = 0; a < NV; ++a) { x1[a] = 0.0; x2[a] = 0.0; }
                    x1[0] = 2.0 * (m1 * e0x + P.alpha1 * e1x); x1[1] = 2.0 * (m1 * e0z + P.alpha1 * e1z);
                    x1[3] = 2.0 * P.alpha1 * e1x * P.dt; x1[4] = 2.0 * P.alpha1 * e1z * P.dt;
                    x2[0] = 2.0 * (m2 * e0x + P.alpha2 * e1x); x2[1] = 2.0 * (m2 * e0z + P.alpha2 * e1z);
                    x2[3] = 2.0 * P.alpha2 * e1x * P.dt; x2[4] = 2.0 * P.alpha2 * e1z * P.dt;
#pragma unroll
                    for (int a = 0; a < NV; ++a) { Cm[a][0] += sg * r[a] * A1 - l * x1[a]; Cm[a][1] += sg * r[a] * A2 - l * x2[a]; }
                    od11 += sg * A1 * A1; od12 += sg * A1 * A2 - l * P.alpha1 * P.alpha2 * h0; od22 += sg * A2 * A2;
                    rrv[0] += lq * A1; rrv[1] += lq * A2;
                }
                slam += l;
                nu2[0] -= w2 * l * 2.0 * (pt[2][0] - lds[L.OB + 3 * j]); nu2[1] -= w2 * l * 2.0 * (pt[2][1] - lds[L.OB + 3 * j + 1]);
            }
        {
            // (so: 0 in a Gauss-Newton restoration, sc_resto_params.gauss_newton -- the second-order terms of rows and dynamics are dropped)
            const double o0 = -2.0 * w0 * slam * so, o1 = -2.0 * w1 * slam * so, o2 = -2.0 * w2 * slam * so;
            H[sym(0, 0)] += o0 + o1; H[sym(1, 1)] += o0 + o1;
            H[sym(0, 3)] += o1 * P.dt; H[sym(1, 4)] += o1 * P.dt; H[sym(3, 3)] += o1 * P.dt * P.dt; H[sym(4, 4)] += o1 * P.dt * P.dt;
#pragma unroll
            for (int a = 0; a < NV; ++a)
#pragma unroll
                for (int b = a; b < NV; ++b) H[sym(a, b)] += o2 * (G2[0][a] * G2[0][b] + G2[1][a] * G2[1][b]);
        }
        {
            const double c3 = lds[L.PS + (k + 1) * 6 + 3] + P.dt * nu2[0], c4 = lds[L.PS + (k + 1) * 6 + 4] + P.dt * nu2[1],
                         c5 = lds[L.PS + (k + 1) * 6 + 5];
            const double cc[3] = {c3 * P.dt * so, c4 * P.dt * so, c5 * P.dt * so};
            int e = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = a; b < 3; ++b, ++e) H[sym(2 + a, 2 + b)] += cc[0] * acc[0].h[e] + cc[1] * acc[1].h[e] + cc[2] * acc[2].h[e];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int j = 0; j < NU; ++j) H[sym(2 + a, 6 + j)] += cc[0] * gc[j][0].d[a] + cc[1] * gc[j][1].d[a] + cc[2] * gc[j][2].d[a];
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            double sh, dh, sl, dl;
            row_sig(WKT + 5 + j, false, mu, sh, dh); row_sig(WKT + 9 + j, false, mu, sl, dl);
            H[sym(6 + j, 6 + j)] += sh + sl;
            q[6 + j] += -(lam[WKT + 5 + j] + dh) + (lam[WKT + 9 + j] + dl);
            if (resto) q[6 + j] -= zeta * (lds[L.U + k * NU + j] - zR[j]);
        }
        if constexpr (OD) {
            // input term R u^2: its Hessian and negative gradient sit in my stage block (the recursion's coupling term Dl is zero)
#pragma unroll
            for (int j = 0; j < NU; ++j) { H[sym(6 + j, 6 + j)] += 2.0 * cw * P.R[j]; q[6 + j] -= 2.0 * cw * P.R[j] * lds[L.U + k * NU + j]; }
            // D^-1 in eigen form, shifted to positive definite (oracle/od_mpc_cbf.py: block_eig; csrc/mpc_gn.hip), then the Schur complement
            const double d11 = 2.0 * cw * P.ps1 + od11, d12 = od12, d22 = 2.0 * cw * P.ps2 + od22;
            const double tr = d11 + d22, df = d11 - d22, rad = sqrt(df * df + 4.0 * d12 * d12);
            double ls = 0.5 * (tr + rad), lw = (d11 * d22 - d12 * d12) / ls;
            double vx = df >= 0.0 ? df + rad : 2.0 * d12, vy = df >= 0.0 ? 2.0 * d12 : rad - df;
            const double vn = vx * vx + vy * vy;
            if (vn > 0.0) { const double rn = 1.0 / sqrt(vn); vx *= rn; vy *= rn; } else { vx = 1.0; vy = 0.0; }
            const double sh = fmax(0.0, 1e-8 * fmax(1.0, fabs(d11) + fabs(d22)) - lw);
            ls += sh; lw += sh;
            dvx = vx; dvy = vy; dils = 1.0 / ls; dilw = 1.0 / lw;
            const double ts = (vx * rrv[0] + vy * rrv[1]) * dils, tw = (-vy * rrv[0] + vx * rrv[1]) * dilw;
            double cs[NV], cwv[NV];
#pragma unroll
            for (int a = 0; a < NV; ++a) { cs[a] = Cm[a][0] * vx + Cm[a][1] * vy; cwv[a] = -Cm[a][0] * vy + Cm[a][1] * vx; }
#pragma unroll
            for (int a = 0; a < NV; ++a) {
                q[a] -= cs[a] * ts + cwv[a] * tw;
#pragma unroll
                for (int b = a; b < NV; ++b) H[sym(a, b)] -= cs[a] * cs[b] * dils + cwv[a] * cwv[b] * dilw;
            }
        }
        // own terms of x_{k+1}: diagonal and negative gradient
        double xd[NX], xq[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xd[i] = 2.0 * cw * P.Q[i];
        xq[0] = -2.0 * cw * P.Q[0] * (xk1[0] - xg[0]); xq[1] = -2.0 * cw * P.Q[1] * (xk1[1] - xg[1]);
#pragma unroll
        for (int i = 2; i < NX; ++i) xq[i] = -2.0 * cw * P.Q[i] * xk1[i];
        {
            constexpr int idx[NXB] = {3, 3, 4, 2, 2};
            constexpr double sgn[NXB] = {-1.0, 1.0, 1.0, -1.0, 1.0};
#pragma unroll
            for (int r = 0; r < NXB; ++r) {
                double sg, d0;
                row_sig(WKT + r, false, mu, sg, d0);
                xd[idx[r]] += sg; xq[idx[r]] += sgn[r] * (lam[WKT + r] + d0);
            }
        }
        if (act) {
#pragma unroll
            for (int j = 0; j < NU; ++j)
                if (k == 0) { lds[L.Dl + j] = OD ? 0.0 : 2.0 * cw * P.R[j]; lds[L.UP + j] = uprev[j]; }
#pragma unroll
            for (int i = 0; i < 55; ++i) lds[L.H + k * 55 + i] = H[i];
#pragma unroll
            for (int i = 0; i < NV; ++i) lds[L.Q + k * 10 + i] = q[i];
#pragma unroll
            for (int i = 0; i < NX; ++i) { lds[L.XD + k * 6 + i] = xd[i]; lds[L.XQ + k * 6 + i] = xq[i]; }
        }
        sync();
    }

    __device__ __forceinline__ bool riccati(double shift) { return vtol_riccati(lds, L, N, lane, shift); }
    __device__ __forceinline__ void lq_forward(double dxk[NX], double dxk1[NX], double duk[NU]) {
        vtol_lq_forward(lds, L, N, lane);
#pragma unroll
        for (int i = 0; i < NX; ++i) { dxk[i] = lds[L.DX + k * 6 + i]; dxk1[i] = lds[L.DX + (k + 1) * 6 + i]; }
#pragma unroll
        for (int j = 0; j < NU; ++j) duk[j] = lds[L.DU + k * NU + j];
    }
    __device__ __forceinline__ double violation() const {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < WKT; ++j) if (valid(j)) v += fmax(0.0, -g[j]);
        return ipm::wsum(v);
    }

    // ---- oracle/mpc_cbf.py: solve(), wave-uniform control flow -----------------------------------------------------------------
    // doubles of one stage (lane) in a problem's hand-over record (mpc_cont.hpp): s | lam | t | zb | zR
    static constexpr int CONT_LANE = 2 * WNR + WKT + 2 * NU + (OD ? 4 : 0);     // (optimal decay: + the stage's decay variables and their best iterate)
    // ST_PENDING in status_out: the solve stopped at the cap of this launch and its state is in cst
    __device__ __forceinline__ void solve(int& status_out, int& iters_out, const ipm::Cont& ct, double* cst, bool& violated) {
        double fraw = 0.0, sf0 = 1.0, mu = P.mu_init;
        double nu = 10.0, delta_last = 0.0, e_best = INFINITY;
        int n_acc = 0, n_resto = 0, n_small = 0, it0 = 1;
        bool resto = false, have = true;
        double theta_R = 0.0, mu_reg = mu;
#pragma unroll
        for (int r = 0; r < WNR; ++r) { ds[r] = 0.0; dlam[r] = 0.0; }
#pragma unroll
        for (int j = 0; j < WKT; ++j) { t[j] = 0.0; dtt[j] = 0.0; }
        if (ct.resume) {
            // the state a previous launch left: [scalars | U (N NU) | per stage: s, lam, t, zb, zR]; the rows, the stage's states and the
            // cost come from an evaluation of U (the same function on the same inputs as the accepted trial's: the same values)
            const double* a = cst + ipm::CONT_SCALARS;
            for (int i = lane; i < N * NU; i += 64) lds[L.U + i] = a[i];
            a += N * NU + (size_t)k * CONT_LANE;
#pragma unroll
            for (int r = 0; r < WNR; ++r) { s[r] = a[r]; lam[r] = a[WNR + r]; }
#pragma unroll
            for (int j = 0; j < WKT; ++j) t[j] = a[2 * WNR + j];
#pragma unroll
            for (int j = 0; j < NU; ++j) { zb[j] = a[2 * WNR + WKT + j]; zR[j] = a[2 * WNR + WKT + NU + j]; }
            if constexpr (OD) {
                dk[0] = a[2 * WNR + WKT + 2 * NU]; dk[1] = a[2 * WNR + WKT + 2 * NU + 1]; dkb[0] = a[2 * WNR + WKT + 2 * NU + 2]; dkb[1] = a[2 * WNR + WKT + 2 * NU + 3];
                od_weights(dk[0], dk[1], w0, w1);
            }
            it0 = (int)cst[0] + 1; mu = cst[1]; nu = cst[2]; delta_last = cst[3]; e_best = cst[4]; n_acc = (int)cst[5];
            resto = cst[6] != 0.0; n_resto = (int)cst[7]; n_small = (int)cst[8]; theta_R = cst[9]; mu_reg = cst[10]; sf0 = cst[11];
            have = false;
            sync();
        } else {
        double uk[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const double lo = P.u_lo[j] + 0.005 * (P.u_hi[j] - P.u_lo[j]), hi = P.u_hi[j] - 0.005 * (P.u_hi[j] - P.u_lo[j]);
            uk[j] = fmin(fmax(uprev[j], lo), hi);
            if (act) lds[L.U + k * NU + j] = uk[j];
            zb[j] = uk[j]; zR[j] = uk[j];
        }
        sync();
        if constexpr (OD) { dk[0] = dkb[0] = P.rf1; dk[1] = dkb[1] = P.rf2; od_weights(dk[0], dk[1], w0, w1); }   // decay variables start at their references
        fraw = eval(L.U, g, dk[0], dk[1]);
        if (ct.it_stop < 0) { violated = violation() > 0.0; status_out = ST_PENDING; iters_out = 0; return; }   // classify only
        linearise();
        sync();
        const double g0 = adjoint(false, 1.0, false, 0.0);
        sf0 = fmin(1.0, 100.0 / fmax(1e-12, g0));
#pragma unroll
        for (int r = 0; r < WNR; ++r) { s[r] = fmax(g[r], 1e-2); lam[r] = mu / s[r]; }
        }
        int status = ST_INACCURATE, it = 0;
        const double tau = 0.995, rho = P.rho, theta_tol = P.theta_tol;
        const bool sreset = P.slack_reset != 0;
        bool pending = false;
        for (it = it0; it <= P.max_iter; ++it) {
            if (cst && it > ct.it_stop) { pending = true; break; }        // the cap of this launch: the solve goes on in the next one
            double cw = resto ? 0.0 : sf0;
            VPROF_T0
            if (!have) fraw = eval(L.U, g, dk[0], dk[1]);
            have = false;
            VPROF_ADD(0)
            linearise();
            sync();
            VPROF_ADD(1)
            if (resto && violation() <= fmax(n_resto == 1 ? P.kappa * theta_R : 0.0, theta_tol)) {
                resto = false; mu = mu_reg; cw = sf0;
#pragma unroll
                for (int r = 0; r < WNR; ++r) { s[r] = fmax(g[r], 1e-2); lam[r] = mu / s[r]; }
                nu = 10.0; n_acc = 0; e_best = INFINITY;
#pragma unroll
                for (int j = 0; j < NU; ++j) zb[j] = lds[L.U + k * NU + j];
            }
            double zeta = resto ? sqrt(mu) : 0.0;
            const double rdn = adjoint(true, cw, resto, zeta);
            VPROF_ADD(2)
            double rpn = 0.0, cs = 0.0, ct = 0.0, lmax = 0.0;
#pragma unroll
            for (int r = 0; r < WNR; ++r)
                if (valid(r)) {
                    const bool el = resto && r < WKT;
                    const double tt = el ? t[r < WKT ? r : 0] : 0.0;
                    rpn = fmax(rpn, fabs(g[r] + tt - s[r])); cs = fmax(cs, fabs(s[r] * lam[r])); lmax = fmax(lmax, lam[r]);
                    if (el) ct = fmax(ct, fabs(tt * (rho - lam[r])));
                }
            rpn = ipm::wmax(rpn); cs = ipm::wmax(cs); ct = ipm::wmax(ct); lmax = ipm::wmax(lmax);
            const double e_opt = fmax(fmax(rdn, rpn), fmax(cs, ct));
            if (!resto && e_opt < e_best) {
                e_best = e_opt;
#pragma unroll
                for (int j = 0; j < NU; ++j) zb[j] = lds[L.U + k * NU + j];
                if constexpr (OD) { dkb[0] = dk[0]; dkb[1] = dk[1]; }
            }
            if (resto) {
                const double theta = violation();
                if (e_opt <= P.resto_tol && theta > fmax(theta_tol, 10.0 * e_opt / rho)) { status = ST_INFEASIBLE; break; }
                if (e_opt <= P.tol) break;
            } else if (e_opt <= P.tol) { status = ST_OPTIMAL; break; }
            n_acc = e_opt <= P.acceptable_tol ? n_acc + 1 : 0;
            if (n_acc >= P.acceptable_iter) {
                if (resto && violation() > theta_tol) status = ST_INFEASIBLE;
                break;
            }
            bool want_resto = !resto && lmax > 1e10;
            if (OD && want_resto) { status = ST_INFEASIBLE; break; }       // (optimal decay: no restoration phase; oracle/od_mpc_cbf.py)
            double alpha = 0.0, ad = 0.0, ft_acc = fraw;
            double drho[2] = {0.0, 0.0}, rhot[2] = {1.0, 1.0};
            if (!want_resto) {
                for (;;) {
                    double cm = 0.0;
#pragma unroll
                    for (int r = 0; r < WNR; ++r)
                        if (valid(r)) {
                            cm = fmax(cm, fabs(s[r] * lam[r] - mu));
                            if (resto && r < WKT) cm = fmax(cm, fabs(t[r < WKT ? r : 0] * (rho - lam[r]) - mu));
                        }
                    cm = ipm::wmax(cm);
                    const double e_mu = fmax(fmax(rdn, rpn), cm);
                    if (!(e_mu <= 10.0 * mu && mu > P.mu_min)) break;
                    mu = fmax(P.mu_min, fmin(0.2 * mu, mu * sqrt(mu)));
                }
                if (resto) zeta = sqrt(mu);
                VPROF_ADD(6)
                double delta = 0.0;
                bool ok = false;
                for (int tr = 0; tr < 40; ++tr) {
                    stage_blocks(cw, resto, zeta, mu, rho);                 // rebuilt for a retry: the gains of the failed pass sit in the slots of H
                    VPROF_ADD(3)
                    if (riccati(delta + zeta)) { ok = true; break; }
                    sync();
                    delta = delta == 0.0 ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
                }
                if (!ok) break;
                if (delta > 0.0) delta_last = delta;
                VPROF_ADD(4)
                double dxk[NX], dxk1[NX], duk[NU];
                lq_forward(dxk, dxk1, duk);
                VPROF_ADD(5)
                // J dz of my rows, then the steps
                double jd[WNR];
                {
                    double pt[3][2], G2[2][NV], v[NV];
                    points(pt); point_jac(G2);
#pragma unroll
                    for (int i = 0; i < NX; ++i) v[i] = dxk[i];
#pragma unroll
                    for (int j = 0; j < NU; ++j) v[6 + j] = duk[j];
                    if constexpr (OD) {
                        // back-substitution of my decay variables: d rho = D^-1 (rr - C' (dx_k, du_k))
                        double t0 = rrv[0], t1 = rrv[1];
#pragma unroll
                        for (int a = 0; a < NV; ++a) { t0 -= Cm[a][0] * v[a]; t1 -= Cm[a][1] * v[a]; }
                        const double ps_ = (dvx * t0 + dvy * t1) * dils, pw_ = (-dvy * t0 + dvx * t1) * dilw;
                        drho[0] = dvx * ps_ - dvy * pw_; drho[1] = dvy * ps_ + dvx * pw_;
                    }
#pragma unroll
                    for (int j = 0; j < WKT; ++j) {
                        double sacc = 0.0;
                        if (j < K) {
                            double r[NV]; cbf_row_grad(pt, G2, j, r);
#pragma unroll
                            for (int i = 0; i < NV; ++i) sacc += r[i] * v[i];
                            if constexpr (OD) {
                                double h0, h1, A1, A2;
                                od_h01(pt, j, h0, h1); od_row_drho(h0, h1, A1, A2);
                                sacc += A1 * drho[0] + A2 * drho[1];
                            }
                        }
                        jd[j] = sacc;
                    }
                    jd[WKT + 0] = -dxk1[3]; jd[WKT + 1] = dxk1[3]; jd[WKT + 2] = dxk1[4]; jd[WKT + 3] = -dxk1[2]; jd[WKT + 4] = dxk1[2];
#pragma unroll
                    for (int j = 0; j < NU; ++j) { jd[WKT + 5 + j] = -duk[j]; jd[WKT + 9 + j] = duk[j]; }
                }
                double ap = 1.0;
                ad = 1.0;
                double srp = 0.0, sum_dss = 0.0, sum_logs = 0.0, sum_t = 0.0, sum_dt = 0.0, sum_dtt = 0.0, sum_logt = 0.0, sum_absg = 0.0;
                double prod_s = 1.0, prod_t = 1.0;                          // sum of logs = log of products, seven rows at a time
#pragma unroll
                for (int r = 0; r < WNR; ++r)
                    if (valid(r)) {
                        const bool el = resto && r < WKT;
                        const int rj = r < WKT ? r : 0;
                        double sg, d0;
                        if (el) row_sig_el(rj, mu, rho, sg, d0); else row_sig(r, false, mu, sg, d0);
                        const double tt = el ? t[rj] : 0.0, rp = g[r] + tt - s[r];
                        const double dl = -sg * jd[r] + d0;
                        double dsi = jd[r] + rp;
                        if (el) {
                            const double nut = rho - lam[r], inut = rcp_(nut), dti = (mu * inut - tt) + dl * tt * inut;
                            dtt[rj] = dti; dsi += dti;
                            if (dti < 0.0) ap = fmin(ap, -tau * tt * rcp_(dti));
                            if (dl > 0.0) ad = fmin(ad, tau * nut * rcp_(dl));
                            sum_t += tt; sum_dt += dti; sum_dtt += dti * rcp_(tt); prod_t *= tt;
                        }
                        ds[r] = dsi; dlam[r] = dl;
                        if (dsi < 0.0) ap = fmin(ap, -tau * s[r] * rcp_(dsi));
                        if (dl < 0.0) ad = fmin(ad, -tau * lam[r] * rcp_(dl));
                        srp += fabs(rp); sum_dss += dsi * rcp_(s[r]); prod_s *= s[r]; sum_absg += fabs(g[r]);
                        if ((r % 7) == 6) { sum_logs += log(prod_s); prod_s = 1.0; }
                    }
                sum_logs += log(prod_s);
                if (resto) sum_logt = log(prod_t);                          // at most eight factors
                ap = ipm::wmin(ap); ad = ipm::wmin(ad);
                srp = ipm::wsum(srp); sum_dss = ipm::wsum(sum_dss); sum_logs = ipm::wsum(sum_logs); sum_absg = ipm::wsum(sum_absg);
                if (resto) { sum_t = ipm::wsum(sum_t); sum_dt = ipm::wsum(sum_dt); sum_dtt = ipm::wsum(sum_dtt); sum_logt = ipm::wsum(sum_logt); }
                nu = fmax(nu, 1.1 * lmax);
                double f, gdz;
                if (resto) {
                    double d2 = 0.0, gd = 0.0;
#pragma unroll
                    for (int j = 0; j < NU; ++j) { const double d = lds[L.U + k * NU + j] - zR[j]; d2 += d * d; gd += d * duk[j]; }
                    d2 = ipm::wsum(act ? d2 : 0.0); gd = ipm::wsum(act ? gd : 0.0);
                    f = 0.5 * zeta * d2; gdz = zeta * gd;
                } else {
                    f = sf0 * fraw;
                    double v = 2.0 * cw * (P.Q[0] * (xk1[0] - xg[0]) * dxk1[0] + P.Q[1] * (xk1[1] - xg[1]) * dxk1[1]);
#pragma unroll
                    for (int i = 2; i < NX; ++i) v += 2.0 * cw * P.Q[i] * xk1[i] * dxk1[i];
#pragma unroll
                    for (int j = 0; j < NU; ++j) {
                        const double u_ = lds[L.U + k * NU + j], um = k ? lds[L.U + (k - 1) * NU + j] : uprev[j];
                        const double dm = k ? lds[L.DU + (k - 1) * NU + j] : 0.0;
                        v += OD ? 2.0 * cw * P.R[j] * u_ * duk[j] : 2.0 * cw * P.R[j] * (u_ - um) * (duk[j] - dm);
                    }
                    if constexpr (OD) v += 2.0 * cw * (P.ps1 * (dk[0] - P.rf1) * drho[0] + P.ps2 * (dk[1] - P.rf2) * drho[1]);
                    gdz = ipm::wsum(act ? v : 0.0);
                }
                double bar0, dbar;
                if (resto) { bar0 = f + rho * sum_t - mu * (sum_logs + sum_logt); dbar = gdz + rho * sum_dt - mu * (sum_dss + sum_dtt); }
                else { bar0 = f - mu * sum_logs; dbar = gdz - mu * sum_dss; }
                if (dbar - nu * srp >= 0.0 && srp > 0.0) nu = dbar / (0.9 * srp);
                const double phi0 = bar0 + nu * srp, dphi = dbar - nu * srp;
                const double noise_rows = P.row_noise * nu * sum_absg;
                alpha = ap;
                bool accepted = false;
                double gt[WNR];
                VPROF_ADD(6)
                for (int h = 0; h < 12; ++h) {
                    if (act) {
#pragma unroll
                        for (int j = 0; j < NU; ++j) lds[L.UT + k * NU + j] = lds[L.U + k * NU + j] + alpha * duk[j];
                    }
                    sync();
                    if constexpr (OD) { rhot[0] = dk[0] + alpha * drho[0]; rhot[1] = dk[1] + alpha * drho[1]; }
                    const double ft = eval(L.UT, gt, rhot[0], rhot[1]);
                    double slog = 0.0, sabs = 0.0, st_t = 0.0, slogt = 0.0, pr_s = 1.0, pr_t = 1.0;
                    const double thr = mu * rcp_(nu);
#pragma unroll
                    for (int r = 0; r < WNR; ++r)
                        if (valid(r)) {
                            double st = s[r] + alpha * ds[r];
                            if (resto) {
                                double tt = 0.0;
                                if (r < WKT) { tt = t[r < WKT ? r : 0] + alpha * dtt[r < WKT ? r : 0]; st_t += tt; pr_t *= tt; }
                                if (P.resto_reset && gt[r] + tt >= thr) st = gt[r] + tt;   // sc_resto_params.slack_reset
                                sabs += fabs(gt[r] + tt - st);
                            } else {
                                if (sreset) st = (P.slack_reset == 1) ? fmax(st, gt[r]) : (gt[r] >= thr ? gt[r] : st);
                                sabs += fabs(gt[r] - st);
                            }
                            pr_s *= st;
                            if ((r % 7) == 6) { slog += log(pr_s); pr_s = 1.0; }
                        }
                    slog += log(pr_s);
                    if (resto) slogt = log(pr_t);
                    slog = ipm::wsum(slog); sabs = ipm::wsum(sabs);
                    double phit;
                    if (resto) {
                        st_t = ipm::wsum(st_t); slogt = ipm::wsum(slogt);
                        double d2 = 0.0;
#pragma unroll
                        for (int j = 0; j < NU; ++j) { const double d = lds[L.UT + k * NU + j] - zR[j]; d2 += d * d; }
                        d2 = ipm::wsum(act ? d2 : 0.0);
                        phit = 0.5 * zeta * d2 + rho * st_t - mu * (slog + slogt) + nu * sabs;
                    } else phit = sf0 * ft - mu * slog + nu * sabs;
                    if (phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * fabs(phi0) + noise_rows) { accepted = true; ft_acc = ft; break; }
                    alpha *= 0.5;
                }
                VPROF_ADD(7)
                if (!accepted) {
                    if (resto) break;
                    want_resto = true;
                } else if (!resto && !OD) {
                    n_small = (alpha < P.small_alpha && violation() > theta_tol) ? n_small + 1 : 0;
                    if (n_small >= P.small_iter && n_resto < P.max_entries && e_best > P.acceptable_tol) want_resto = true;
                }
                if (!want_resto) {
                    // take the step: the accepted trial's state is the next iteration's evaluation
                    if (act) {
#pragma unroll
                        for (int j = 0; j < NU; ++j) lds[L.U + k * NU + j] = lds[L.UT + k * NU + j];
                    }
#pragma unroll
                    for (int r = 0; r < WNR; ++r) {
                        double sn = s[r] + alpha * ds[r];
                        if (sreset && !resto) sn = (P.slack_reset == 1) ? fmax(sn, gt[r]) : (gt[r] >= mu * rcp_(nu) ? gt[r] : sn);
                        if (resto && P.resto_reset) {
                            const double tot = gt[r] + (r < WKT ? t[r < WKT ? r : 0] + alpha * dtt[r < WKT ? r : 0] : 0.0);
                            if (tot >= mu * rcp_(nu)) sn = tot;
                        }
                        double l = lam[r] + ad * dlam[r];
                        const double mus = mu * rcp_(sn);
                        l = fmin(fmax(l, 1e-10 * mus), 1e10 * mus);
                        if (resto && r < WKT) {
                            const int rj = r < WKT ? r : 0;
                            const double tn = t[rj] + alpha * dtt[rj];
                            t[rj] = tn;
                            const double mut = mu * rcp_(tn);
                            l = fmin(fmax(l, rho - 1e10 * mut), rho - 1e-10 * mut);
                            l = fmin(fmax(l, 1e-300), rho * (1.0 - 1e-15));
                        }
                        s[r] = sn; lam[r] = l; g[r] = gt[r];
                    }
                    if constexpr (OD) { dk[0] = rhot[0]; dk[1] = rhot[1]; od_weights(dk[0], dk[1], w0, w1); }
                    fraw = ft_acc; have = true;
                    sync();
                    continue;
                }
                // the step is not taken: my registers hold the trial's state, bring the iterate's back
                if (OD) break;                                             // (optimal decay: no restoration to hand over to)
                fraw = eval(L.U, g);
            }
            // want_resto
            theta_R = violation();
            if (e_best <= P.acceptable_tol || theta_R <= theta_tol || n_resto >= P.max_entries) break;
            resto = true; n_resto += 1; n_small = 0;
#pragma unroll
            for (int j = 0; j < NU; ++j) zR[j] = lds[L.U + k * NU + j];
            mu_reg = mu;
            double vmax = 0.0;
#pragma unroll
            for (int j = 0; j < WKT; ++j) if (valid(j)) vmax = fmax(vmax, -g[j]);
            mu = fmax(mu, ipm::wmax(vmax));
#pragma unroll
            for (int r = 0; r < WNR; ++r) {
                double sn;
                if (r < WKT) { sn = ((2.0 * mu + rho * g[r]) + sqrt(rho * rho * g[r] * g[r] + 4.0 * mu * mu)) / (2.0 * rho); t[r < WKT ? r : 0] = sn - g[r]; }
                else sn = fmax(g[r], 1e-2);
                s[r] = sn; lam[r] = mu / sn;
            }
            nu = 10.0; n_acc = 0; have = true;
        }
        if (pending) {
            // hand-over (mpc_cont.hpp): g holds the rows of U on every path to the top of the loop (accepted trial, the evaluation
            // before a restoration entry, the initial one) -- except right after a resume, which never stops again at once
            sync();
            violated = violation() > theta_tol;
            double* a = cst + ipm::CONT_SCALARS;
            for (int i = lane; i < N * NU; i += 64) a[i] = lds[L.U + i];
            a += N * NU + (size_t)k * CONT_LANE;
            if (act) {
#pragma unroll
                for (int r = 0; r < WNR; ++r) { a[r] = s[r]; a[WNR + r] = lam[r]; }
#pragma unroll
                for (int j = 0; j < WKT; ++j) a[2 * WNR + j] = t[j];
#pragma unroll
                for (int j = 0; j < NU; ++j) { a[2 * WNR + WKT + j] = zb[j]; a[2 * WNR + WKT + NU + j] = zR[j]; }
                if constexpr (OD) { a[2 * WNR + WKT + 2 * NU] = dk[0]; a[2 * WNR + WKT + 2 * NU + 1] = dk[1]; a[2 * WNR + WKT + 2 * NU + 2] = dkb[0]; a[2 * WNR + WKT + 2 * NU + 3] = dkb[1]; }
            }
            if (lane == 0) {
                cst[0] = (double)(it - 1); cst[1] = mu; cst[2] = nu; cst[3] = delta_last; cst[4] = e_best; cst[5] = (double)n_acc;
                cst[6] = resto ? 1.0 : 0.0; cst[7] = (double)n_resto; cst[8] = (double)n_small; cst[9] = theta_R; cst[10] = mu_reg; cst[11] = sf0;
            }
            status_out = ST_PENDING; iters_out = it - 1;
            return;
        }
        if (it > P.max_iter) it = P.max_iter;
        if (status != ST_OPTIMAL && status != ST_INFEASIBLE && e_best <= P.acceptable_tol && !resto) {
            sync();
            if (act) {
#pragma unroll
                for (int j = 0; j < NU; ++j) lds[L.U + k * NU + j] = zb[j];
            }
            if constexpr (OD) { dk[0] = dkb[0]; dk[1] = dkb[1]; }
            status = ST_OPTIMAL;
        }
        sync();
        if constexpr (OD) {
            // optimal decay has no restoration phase: "infeasible" there means "stopped at an infeasible iterate" (oracle/od_mpc_cbf.py)
            if (status != ST_OPTIMAL) {
                eval(L.U, g, dk[0], dk[1]);
                double gm = 1e300;
#pragma unroll
                for (int r = 0; r < WNR; ++r) if (valid(r)) gm = fmin(gm, g[r]);
                gm = ipm::wmin(act ? gm : 1e300);
                status = (gm < -1e-6 || status == ST_INFEASIBLE) ? ST_INFEASIBLE : ST_INACCURATE;
            }
        }
        status_out = status; iters_out = it;
    }
};

template <typename TIO, int WKT, bool OD = false>
__global__ void __launch_bounds__(64) mpcvtol_wave_kernel(const Params P, long long B, int obs_shared, const TIO* __restrict__ X,
                                                          const TIO* __restrict__ u_prev, const TIO* __restrict__ goal,
                                                          const TIO* __restrict__ obs, TIO* __restrict__ u_out, int* __restrict__ status_out,
                                                          int* __restrict__ iters_out, TIO* __restrict__ z_out, const ipm::Cont ct,
                                                          TIO* __restrict__ rho_out) {
    extern __shared__ double vtol_lds[];
    long long b;
    if (!ipm::cont_problem(ct, B, b)) return;                           // mpc_cont.hpp: block index, or an entry of the previous launch's queue
    Wave<WKT, OD> S(P, (ldsd*)vtol_lds);
    const TIO* ob = obs + (obs_shared ? 0 : b * P.K * 7);
    if ((int)threadIdx.x < 3 * WKT_MAX) {
        const int j = threadIdx.x / 3, c = threadIdx.x % 3;
        vtol_lds[S.L.OB + threadIdx.x] = j < P.K ? (double)ob[7 * j + c] : 0.0;
    }
    for (int i = 0; i < NX; ++i) S.x0[i] = (double)X[b * NX + i];
    for (int j = 0; j < NU; ++j) S.uprev[j] = (double)u_prev[b * NU + j];
    S.xg[0] = (double)goal[b * 2]; S.xg[1] = (double)goal[b * 2 + 1];
    __syncthreads();
    int st, it;
    bool violated = false;
    S.solve(st, it, ct, ct.state ? ct.state + b * ct.stride : nullptr, violated);
    if (st == ST_PENDING) {
        if (threadIdx.x == 0) {
            if (ct.it_stop >= 0) { status_out[b] = SC_STATUS_PENDING_MPC; if (iters_out) iters_out[b] = it; }
            ipm::cont_push(ct, b, violated);
        }
        return;
    }
    if (threadIdx.x == 0) {
        for (int j = 0; j < NU; ++j) u_out[b * NU + j] = (TIO)vtol_lds[S.L.U + j];
        status_out[b] = st;
        if (iters_out) iters_out[b] = it;
    }
#ifdef SC_VTOL_PROF
    if (z_out && threadIdx.x == 0) { for (int i = 0; i < 8; ++i) z_out[b * (long long)(P.N * NU) + i] = (TIO)(double)S.prof[i]; }
    if (false)
#else
    if (z_out && S.act)
#endif
        for (int j = 0; j < NU; ++j) z_out[b * (long long)(P.N * NU) + S.k * NU + j] = (TIO)vtol_lds[S.L.U + S.k * NU + j];
    if constexpr (OD) {
        if (rho_out && S.act) { rho_out[b * (long long)(2 * P.N) + 2 * S.k] = (TIO)S.dk[0]; rho_out[b * (long long)(2 * P.N) + 2 * S.k + 1] = (TIO)S.dk[1]; }
    }
}

template <typename TIO, int WKT, bool OD = false>
static hipError_t wave_launch_t(const Params& P, const sc_mpcvtol_params& p, long long B, size_t lds, const void* X, const void* u_prev,
                                const void* goal, const void* obs, void* u_out, int* status_out, int* iters_out, void* z_out, hipStream_t stream,
                                const ipm::Cont& ct, void* rho_out = nullptr) {
    hipError_t e = hipFuncSetAttribute((const void*)mpcvtol_wave_kernel<TIO, WKT, OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mpcvtol_wave_kernel<TIO, WKT, OD>), dim3((unsigned)B), dim3(64), lds, stream, P, B, p.obs_shared, (const TIO*)X,
                       (const TIO*)u_prev, (const TIO*)goal, (const TIO*)obs, (TIO*)u_out, status_out, iters_out, (TIO*)z_out, ct, (TIO*)rho_out);
    return hipGetLastError();
}

// doubles of one problem's solver state in a continuation workspace (mpc_cont.hpp; the layout of Wave::solve's hand-over)
size_t mpcvtol_state_doubles(int N, int K, bool od) {
    if (od) return ipm::CONT_SCALARS + (size_t)N * NU + (size_t)N * (K <= 8 ? Wave<8, true>::CONT_LANE : Wave<16, true>::CONT_LANE);
    return ipm::CONT_SCALARS + (size_t)N * NU + (size_t)N * (K <= 8 ? Wave<8>::CONT_LANE : Wave<16>::CONT_LANE);
}

hipError_t mpcvtol_wave_launch(const sc_mpcvtol_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                               const void* obs, void* u_out, int* status_out, int* iters_out, void* z_out, hipStream_t stream,
                               const ipm::Cont& ct) {
    const Params P = from_c(p, K);
    const size_t lds = mpcvtol_wave_lds_bytes(p.horizon);
    if (p.io_dtype == SC_DTYPE_F64)
        return K <= 8 ? wave_launch_t<double, 8>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct)
                      : wave_launch_t<double, 16>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct);
    return K <= 8 ? wave_launch_t<float, 8>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct)
                  : wave_launch_t<float, 16>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct);
}

// optimal-decay MPC-CBF for VTOL2D (include/safe_control_amd.h: sc_odmpcvtol_params)
hipError_t odmpcvtol_wave_launch(const sc_odmpcvtol_params& q, long long B, int K, const void* X, const void* u_prev, const void* goal,
                                 const void* obs, void* u_out, void* rho_out, int* status_out, int* iters_out, void* z_out, hipStream_t stream,
                                 const ipm::Cont& ct) {
    const sc_mpcvtol_params& p = q.mpc;
    Params P = from_c(p, K);
    P.ps1 = q.p_sb[0]; P.ps2 = q.p_sb[1]; P.rf1 = q.omega_ref[0]; P.rf2 = q.omega_ref[1];
    const size_t lds = mpcvtol_wave_lds_bytes(p.horizon);
    if (p.io_dtype == SC_DTYPE_F64)
        return K <= 8 ? wave_launch_t<double, 8, true>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct, rho_out)
                      : wave_launch_t<double, 16, true>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct, rho_out);
    return K <= 8 ? wave_launch_t<float, 8, true>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct, rho_out)
                  : wave_launch_t<float, 16, true>(P, p, B, lds, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct, rho_out);
}

}  // namespace sc
