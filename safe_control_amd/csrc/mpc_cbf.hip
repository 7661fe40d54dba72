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
#include "sc_qp2.hpp"
#include "mpc_chol.hpp"
#include "mpc_ipm_common.hpp"
#include "mpc_cont.hpp"
#include "../../include/safe_control_amd.h"

namespace sc {

#define SC_SYNC() __syncthreads()

// -DSC_MPC_PROF: developer build that returns per-phase shader-clock totals in z_out instead of the solution
// (tools/exp_mpc_phases.py); never defined in the shipped library.
// An opaque copy of the lane id at the top of every phase: nothing derived from it can be hoisted out of the
// interior-point loop or shared between phases, which is what kept ~330 VGPRs live.  With it the kernel fits 256
// VGPRs without spills, i.e. two waves per SIMD.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

struct Prof {
#ifdef SC_MPC_PROF
    double ph[20];
    long long t;
    __device__ __forceinline__ void start() { for (int i = 0; i < 20; ++i) ph[i] = 0.0; t = clock64(); }
    __device__ __forceinline__ void mark(int i) { const long long t_ = clock64(); ph[i] += (double)(t_ - t); t = t_; }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
#endif
};
#define SC_PH(i) pf.mark(i)

// ---- wave reductions on DPP ---------------------------------------------------------------------------------------
// A __shfl_xor butterfly is six dependent ds_bpermute round trips (~100 cycles each for a lone wave).  Here four DPP
// moves (xor 1, xor 2, half-row mirror, row mirror: VALU latency) leave every lane with its 16-lane row total, and
// the four row totals are combined from v_readlane.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int src) {        // src: compile-time constant lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
template <typename Op>
__device__ __forceinline__ double wreduce(double v, Op op) {
    v = op(v, dpp_move<0xB1>(v));        // quad_perm [1,0,3,2]
    v = op(v, dpp_move<0x4E>(v));        // quad_perm [2,3,0,1]
    v = op(v, dpp_move<0x141>(v));       // row_half_mirror
    v = op(v, dpp_move<0x140>(v));       // row_mirror
    return op(op(lane_value(v, 0), lane_value(v, 16)), op(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ double wsum(double v) { return wreduce(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wmin(double v) { return wreduce(v, [](double a, double b) { return fmin(a, b); }); }
__device__ __forceinline__ double wmax(double v) { return wreduce(v, [](double a, double b) { return fmax(a, b); }); }
__device__ __forceinline__ double wprod(double v) { return wreduce(v, [](double a, double b) { return a * b; }); }
__device__ __forceinline__ int wsumi(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
// v[l] + v[l ^ 16] + v[l ^ 32] + v[l ^ 48]: the same lane of the four 16-lane rows, on the gfx950 row swaps
// (v_permlane16_swap exchanges odd rows of one register with even rows of the other, v_permlane32_swap the wave
// halves; called with two copies of v the two results are "mine" and "the partner row's" in every lane)
__device__ __forceinline__ double sum_rows4(double v) {
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v;
}

// sum of log(x_i) over the wave as log(prod mantissas) + ln2 * sum exponents: one log per wave instead of one per
// row.  A lane multiplies at most a few mantissas in [0.5, 1), the wave product stays above 2^-(64 * rows per lane).
struct LogSum {
    double pm = 1.0;
    int pe = 0;
    __device__ __forceinline__ void add(double x) { pm *= __builtin_amdgcn_frexp_mant(x); pe += __builtin_amdgcn_frexp_exp(x); }
    __device__ __forceinline__ double total() const { return log(wprod(pm)) + 0.6931471805599453 * (double)wsumi(pe); }
};

typedef double d4_t __attribute__((ext_vector_type(4)));

struct MpcMem {                 // LDS carve-up (doubles); NP = N + 2 predicted positions, n = 2 N, m rows
    double *z, *zt, *dz, *zb;                     // n        (the right-hand side is solved in place in dz)
    // cv: 3 n column-pass results (r_d | J'(1/s) | J'(sig r_p + lam)); the last two die with the right-hand side and
    // their space then holds dp (2 NP) and dV (N + 1)
    double *cv, *dp, *dV;
    double *TH, *V, *C, *S;                       // N+1
    double *pos;                                  // 2 NP
    double *SA, *SB, *SS;                         // NP suffix sums
    double *obs;                                  // K*7
    double *dh;                                   // NP*K*2
    double *g, *sl, *lam;                         // m
    double *tel;                                  // N K elastic variables of the feasibility restoration (see carve)
    double *ds, *dlam;                            // m   (aliases: is = ds, vb = dlam, dead before ds / dlam are written)
    double *dP;                                   // 2 NP * n : G = d p / d z
    // region A, max(n n, NP (26 + K)):  [Phi 2 NP * 10 | Y 3 * 2 NP | hk NP K]  then  M n*n
    //   Phi, Y die with the T and column passes, hk with the g rows; M lives from the condensation to the Cholesky
    double *Phi, *Y, *hk, *M;
    // region B, 2 NP * n:  PC, PD (2 NP each, inside eval_values only), T = Phi G (until the condensation), then
    //   L (transposition scratch / LDS Cholesky)
    double *PC, *PD, *T, *L;
    double *rhs, *is;                             // aliases of dz, ds
    // optimal-decay variant only
    double *rho, *rhot, *drho, *rhob;             // 2 N decay variables (current, trial, step, best iterate)
    double *w0s, *w1s;                            // N stage weights of the rows
    double *A1, *A2;                              // N K : d row / d rho_1, d rho_2
    double *ods;                                  // 22 N : per stage C (12), eigen form of D^-1 (4), t0 t1 t2 (6)
    double *XS;                                   // 39 N exchange scratch inside region B (E packed 21, c_v 18)
};

__host__ __device__ inline size_t mpc_region_b(int N, bool od) {
    const size_t n = 2 * (size_t)N, NP = (size_t)N + 2;
    return (od && 39 * (size_t)N > 2 * NP * n) ? 39 * (size_t)N : 2 * NP * n;
}
// The restoration's elastic variables t (one per CBF row, mpc_ipm_common.hpp).  Rows 0..3 of G = d p / d z belong to p_0 and p_1,
// which do not depend on z: 4 n = 8 N doubles that no product reads (every loop over the rows of G starts at row 4) -- for
// K <= 8 the N K elastic variables live there, and the compile-time kernel of BASELINE config 3 keeps its 20 KiB (eight
// problems per CU).  Unicycle2D (first live row 2) and larger K get their own N K doubles; optimal decay has no restoration.
__host__ __device__ inline bool mpc_tel_in_g(int K, bool od, bool uni) { return !od && !uni && K <= 8; }
__host__ __device__ inline size_t mpc_lds_doubles(int N, int K, bool od = false, bool uni = false) {
    const size_t n = 2 * (size_t)N, NP = (size_t)N + 2, m = (size_t)N * K + 2 * N + 2 * n;   // (rows: sized for the larger, DU, layout)
    const size_t regA = n * n > NP * (26 + K) ? n * n : NP * (26 + K);
    const size_t cvt = 2 * n > 3 * (size_t)N + 5 ? 2 * n : 3 * (size_t)N + 5;
    const size_t odx = (od ? 4 * n + 2 * (size_t)N + 2 * (size_t)N * K + 22 * (size_t)N : 0) +
                       ((od || mpc_tel_in_g(K, od, uni)) ? 0 : (size_t)N * K);
    return 4 * n + n + cvt + 4 * (N + 1) + 2 * NP + 3 * NP + (size_t)K * 7 + NP * K * 2 + 5 * m + 2 * NP * n + regA +
           mpc_region_b(N, od) + odx;
}

__device__ inline MpcMem carve(double* b, int N, int K, bool od = false, bool uni = false) {
    const int n = 2 * N, NP = N + 2, m = N * K + 2 * N + 2 * n;
    MpcMem M;
    auto take = [&](size_t c) { double* r = b; b += c; return r; };
    M.z = take(n); M.zt = take(n); M.dz = take(n); M.zb = take(n);
    const size_t cvt = 2 * n > 3 * N + 5 ? 2 * n : 3 * N + 5;
    M.cv = take(n + cvt); M.dp = M.cv + n; M.dV = M.dp + 2 * NP;
    M.TH = take(N + 1); M.V = take(N + 1); M.C = take(N + 1); M.S = take(N + 1);
    M.pos = take(2 * NP);
    M.SA = take(NP); M.SB = take(NP); M.SS = take(NP);
    M.obs = take((size_t)K * 7);
    M.dh = take((size_t)NP * K * 2);
    M.g = take(m); M.sl = take(m); M.lam = take(m); M.ds = take(m); M.dlam = take(m);
    M.dP = take((size_t)2 * NP * n);
    const size_t regA = (size_t)n * n > (size_t)NP * (26 + K) ? (size_t)n * n : (size_t)NP * (26 + K);
    double* A = take(regA);
    M.Phi = A; M.Y = A + (size_t)2 * NP * 10; M.hk = M.Y + 6 * NP; M.M = A;
    M.T = take(mpc_region_b(N, od)); M.L = M.T; M.PC = M.T; M.PD = M.T + 2 * NP; M.XS = M.T;
    M.rhs = M.dz; M.is = M.ds;
    M.rho = M.rhot = M.drho = M.rhob = M.w0s = M.w1s = M.A1 = M.A2 = M.ods = nullptr;
    M.tel = od ? nullptr : (mpc_tel_in_g(K, od, uni) ? M.dP : take((size_t)N * K));
    if (od) {
        M.rho = take(n); M.rhot = take(n); M.drho = take(n); M.rhob = take(n);
        M.w0s = take(N); M.w1s = take(N);
        M.A1 = take((size_t)N * K); M.A2 = take((size_t)N * K);
        M.ods = take((size_t)22 * N);
    }
    return M;
}

struct MpcConst {
    int N, K, n, mc, m, ns;                       // ns: speed-bound rows (2 N for DynamicUnicycle2D, 0 for Unicycle2D)
    double dt, Qx, Qy, Qth, Qv, R0, R1, w0, w1, w2, vmax, amax, wmaxu, Rrob, beta;
    double x0, y0, th0, v0, up0, up1, gx, gy;
    double al1, al2, ps1, ps2, rf1, rf2;          // optimal decay: CBF gains, decay penalties and references
};

// ---- barrier h, dh/dp, d2h/dp2 at a position (oracle/mpc_cbf.py: barrier) ---------------------
// circle robots/dynamic_unicycle2D.py:194-202; superellipsoid :204-220 (fabs, clamps a,b>=1e-3, e>=2)
// CHAIN: integer superellipsoid exponents by the multiply chain of ipm::pow3 -- the N = 20 and run-time-horizon kernels (config 5's
// superellipsoid scenes).  The compile-time N = 10 kernels are capped at 256 VGPRs for two waves per SIMD and the inlined chain costs them
// 31 - 134 spilled registers (config 3: 1.48 -> 1.65 ms, optimal decay 4.0 -> 4.5 ms per 4096 problems): they keep the library pow().
template <bool CHAIN>
__device__ inline void barrier_at(double px_, double py_, const double* o, const MpcConst& c, bool derivs,
                                  double& h, double& d0, double& d1, double& hxx, double& hxy, double& hyy) {
    if (o[6] == 0.0) {                      // LDS rows: 0 = circle, otherwise the row scale of a superellipsoid (mpc_ipm_common.hpp)
        const double d = c.Rrob + o[2];
        const double ex = px_ - o[0], ey = py_ - o[1];
        h = (ex * ex + ey * ey) - c.beta * d * d;
        d0 = 2.0 * ex; d1 = 2.0 * ey; hxx = 2.0; hxy = 0.0; hyy = 2.0;
        return;
    }
    const double a = fmax(fabs(o[2]), 1e-3) + c.Rrob, b = fmax(fabs(o[3]), 1e-3) + c.Rrob;
    const double e = fmax(fabs(o[4]), 2.0);
    double st, ct;
    if constexpr (CHAIN) sincos_(o[5], &st, &ct);
    else sincos(o[5], &st, &ct);                                           // as pow(): the register budget of the N = 10 kernels
    const double dx = px_ - o[0], dy = py_ - o[1];
    const double px = ct * dx + st * dy, py = -st * dx + ct * dy;
    const double ax = fabs(px) / a, ay = fabs(py) / b;
    const double sc = o[6];
    double xe, xe1, xe2, ye, ye1, ye2;
    if constexpr (CHAIN) {
        ipm::pow3(ax, e, derivs, xe, xe1, xe2);                            // integer exponents: one multiply chain instead of three pow()
        ipm::pow3(ay, e, derivs, ye, ye1, ye2);
    } else {
        xe = pow(ax, e); ye = pow(ay, e);
        xe1 = derivs ? pow(ax, e - 1) : 0.0; ye1 = derivs ? pow(ay, e - 1) : 0.0;
        xe2 = derivs ? pow(ax, e - 2) : 0.0; ye2 = derivs ? pow(ay, e - 2) : 0.0;
    }
    h = sc * (xe + ye - 1.0);
    if (!derivs) { d0 = d1 = hxx = hxy = hyy = 0.0; return; }
    const double sx = px > 0 ? 1.0 : (px < 0 ? -1.0 : 0.0), sy = py > 0 ? 1.0 : (py < 0 ? -1.0 : 0.0);
    const double gpx = e * xe1 / a * sx, gpy = e * ye1 / b * sy;
    const double cxx = e * (e - 1) * xe2 / (a * a), cyy = e * (e - 1) * ye2 / (b * b);
    d0 = sc * (ct * gpx - st * gpy);
    d1 = sc * (st * gpx + ct * gpy);
    hxx = sc * (ct * ct * cxx + st * st * cyy);
    hxy = sc * (ct * st * cxx - st * ct * cyy);
    hyy = sc * (st * st * cxx + ct * ct * cyy);
}

// ---- scans over the first lanes of the wave -----------------------------------------------------------------------
// ROW16 = true: all NP = N + 2 stages sit in lanes 0..15, one DPP row, and a scan step is a row_shr / row_shl move
// (VALU latency, zero fill at the row edge) instead of a ds_bpermute round trip.
// inclusive prefix sum  out[l] = sum_{l' <= l} v[l']
template <bool ROW16>
__device__ __forceinline__ double prefix_sum(double v, int lane) {
    if constexpr (ROW16) {
        v += dpp_move<0x111>(v); v += dpp_move<0x112>(v); v += dpp_move<0x114>(v); v += dpp_move<0x118>(v);   // row_shr:1,2,4,8
    } else {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
    }
    return v;
}
// inclusive suffix sum  out[l] = sum_{l' >= l} v[l']   (v must be 0 in lanes outside the range of interest)
template <bool ROW16>
__device__ __forceinline__ double suffix_sum(double v, int lane) {
    if constexpr (ROW16) {
        v += dpp_move<0x101>(v); v += dpp_move<0x102>(v); v += dpp_move<0x104>(v); v += dpp_move<0x108>(v);   // row_shl:1,2,4,8
    } else {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_down(v, o);
            if (lane + o < 64) v += t;
        }
    }
    return v;
}

// ---- rollout + barrier values + g + f at a trial z (oracle: evaluate level 0) --------------------
// Lane k holds stage k: heading / speed are exclusive prefix sums of the inputs, positions and the position
// sensitivities PC_k = sum_{i<k} (cos, sin), PD_k = sum_{i<k} v_i (-sin, cos) prefix sums of the stage terms.
// UNI = kinematic Unicycle2D (robots/unicycle2D.py): the first input IS the speed of its stage (no speed state), the
// position of stage k+1 therefore depends on v_k directly (G has dt (cos, sin)_j in the v_j column for k >= j + 1).
template <bool ROW16, bool OD, bool UNI = false>
__device__ inline double eval_values(const double* z, const double* rho, const MpcMem& W, const MpcConst& c, int lane, bool derivs,
                                     Prof& pf, int slot) {
    lane = opaque(lane);
    const int N = c.N, K = c.K, n = c.n, NP = N + 2;
    double ak = 0.0, wk = 0.0;
    if (lane < N) { ak = z[2 * lane]; wk = z[2 * lane + 1]; }
    const double th = c.th0 + c.dt * (prefix_sum<ROW16>(wk, lane) - wk);
    const double v = UNI ? ak : c.v0 + c.dt * (prefix_sum<ROW16>(ak, lane) - ak);
    double sn, cs;
    sincos(th, &sn, &cs);
    const bool stg = UNI ? lane < N : lane <= N;                      // stages whose (v, heading) move a later position
    const double tc = stg ? cs : 0.0, ts = stg ? sn : 0.0, tdx = stg ? -v * sn : 0.0, tdy = stg ? v * cs : 0.0;
    const double pcx = prefix_sum<ROW16>(tc, lane) - tc, pcy = prefix_sum<ROW16>(ts, lane) - ts;
    const double pdx = prefix_sum<ROW16>(tdx, lane) - tdx, pdy = prefix_sum<ROW16>(tdy, lane) - tdy;
    const double px = c.x0 + c.dt * pdy, py = c.y0 - c.dt * pdx;
    if (lane <= N) { W.TH[lane] = th; W.V[lane] = v; W.C[lane] = cs; W.S[lane] = sn; }
    if (lane < NP) {
        W.pos[2 * lane] = px; W.pos[2 * lane + 1] = py;
        if (derivs) { W.PC[2 * lane] = pcx; W.PC[2 * lane + 1] = pcy; W.PD[2 * lane] = pdx; W.PD[2 * lane + 1] = pdy; }
    }
    // f
    double part = 0.0;
    if (lane >= 1 && lane <= N) {
        const double ex = px - c.gx, ey = py - c.gy;
        part = c.Qx * ex * ex + c.Qy * ey * ey + c.Qth * th * th + c.Qv * v * v;
    }
    for (int i = lane; i < n; i += 64) {
        const double prev = i >= 2 ? z[i - 2] : ((i & 1) ? c.up1 : c.up0);
        const double du = OD ? z[i] : z[i] - prev;                    // OD: R u^2 (optimal_decay_mpc_cbf.py:178-179)
        part += ((i & 1) ? c.R1 : c.R0) * du * du;
    }
    if constexpr (OD) {
        // stage weights of the rows  w1_k = s_k - 2,  w0_k = 1 - s_k + q_k  and the decay penalty
        if (lane < N) {
            const double r1 = rho[2 * lane], r2 = rho[2 * lane + 1];
            if constexpr (UNI) {
                // relative degree 1 (config-5 extension, oracle/od_mpc_rd1.py):  h(p_k+1) - (1 - alpha rho1_k) h(p_k);  the
                // second decay variable of the layout is inert (it stays at its reference)
                W.w0s[lane] = -(1.0 - c.al1 * r1); W.w1s[lane] = 1.0;
            } else {
                const double sk = c.al1 * r1 + c.al2 * r2, qk = c.al1 * c.al2 * r1 * r2;
                W.w0s[lane] = 1.0 - sk + qk; W.w1s[lane] = sk - 2.0;
            }
            part += c.ps1 * (r1 - c.rf1) * (r1 - c.rf1) + c.ps2 * (r2 - c.rf2) * (r2 - c.rf2);
        }
    }
    SC_SYNC();
    SC_PH(slot);
    if (derivs) {
        // G = d p / d z:  dP[2k+d][col] = dt^2 (P[k][d] - P[j+1][d]) for stages j <= k - 2, P = PC (accel) | PD (omega);
        // the structural zeros (k < j + 2) are written once at kernel start
        const double dt2 = c.dt * c.dt;
        for (int e = lane; e < 3 * n; e += 64) {
            const int gq = e / n, col = e - gq * n, j = col >> 1;
            if (UNI && !(col & 1)) {                                  // d p_k / d v_j = dt (cos, sin)(theta_j), k >= j + 1
                const double g0 = c.dt * W.C[j], g1 = c.dt * W.S[j];
                for (int k = gq; k < NP; k += 3) {
                    if (k >= j + 1) { W.dP[(size_t)(2 * k) * n + col] = g0; W.dP[(size_t)(2 * k + 1) * n + col] = g1; }
                }
                continue;
            }
            const double* P = W.PC + ((col & 1) ? (W.PD - W.PC) : 0);
            const double b0 = P[2 * (j + 1)], b1 = P[2 * (j + 1) + 1];
            for (int k = gq; k < NP; k += 3) {
                if (k >= j + 2) {
                    W.dP[(size_t)(2 * k) * n + col] = dt2 * (P[2 * k] - b0);
                    W.dP[(size_t)(2 * k + 1) * n + col] = dt2 * (P[2 * k + 1] - b1);
                }
            }
        }
    }
    SC_PH(slot + 1);
    for (int e = lane; e < NP * K; e += 64) {
        const int k = e / K, j = e - k * K;
        double h, d0, d1, hxx, hxy, hyy;
        barrier_at<!ROW16>(W.pos[2 * k], W.pos[2 * k + 1], W.obs + 7 * j, c, derivs, h, d0, d1, hxx, hxy, hyy);
        W.hk[e] = h;
        if (derivs) {
            W.dh[2 * e] = d0; W.dh[2 * e + 1] = d1;
        }
    }
    SC_SYNC();
    SC_PH(slot + 2);
    // g >= 0 : [CBF (k major) | v_max -/+ v_k (k = 1..N) | u_max - z | u_max + z]
    for (int i = lane; i < c.m; i += 64) {
        double gi;
        if (i < c.mc) {
            const int k = i / K, j = i - k * K;
            const double h2 = W.hk[(k + 2) * K + j], h1 = W.hk[(k + 1) * K + j], h0 = W.hk[k * K + j];
            if constexpr (OD) {
                gi = c.w2 * h2 + W.w1s[k] * h1 + W.w0s[k] * h0;      // w2 = 1 (rel-degree 2) or 0 (rel-degree 1)
                if (derivs) {                                       // d row / d rho_i = a_i (h1 - h0) + a1 a2 rho_other h0
                    if constexpr (UNI) { W.A1[i] = c.al1 * h0; W.A2[i] = 0.0; }
                    else {
                        const double aa = c.al1 * c.al2 * h0;
                        W.A1[i] = c.al1 * (h1 - h0) + aa * rho[2 * k + 1];
                        W.A2[i] = c.al2 * (h1 - h0) + aa * rho[2 * k];
                    }
                }
            } else {
                gi = c.w2 * h2 + c.w1 * h1 + c.w0 * h0;
            }
        } else if (i < c.mc + c.ns) {
            const int r = i - c.mc, k = (r >> 1) + 1;
            gi = (r & 1) ? (c.vmax + W.V[k]) : (c.vmax - W.V[k]);
        } else {
            const int r = i - c.mc - c.ns;
            const int col = r < n ? r : r - n;
            const double ub = (col & 1) ? c.wmaxu : c.amax;
            gi = r < n ? (ub - z[col]) : (ub + z[col]);
        }
        W.g[i] = gi;
    }
    SC_SYNC();
    const double fsum = wsum(part);
    SC_PH(slot + 3);
    return fsum;
}

// ---- assembly in position space ------------------------------------------------------------------------------
// With G = d(p_0..p_{N+1})/dz (2 NP x n) and A = d g_cbf / d p (row (kappa, jo) has the three 1x2 blocks
// w_t dh[kappa + t, jo], t = 0..2), the CBF Jacobian is J = A G and is never formed:
//   J' v            = G' (A' v)                                  (A' v is a 2 NP vector, one 2-vector per position)
//   sf W + J' S J   = G' (sf Om + A' S A) G + structured terms   (Phi = sf Om + A' S A is block pentadiagonal)
//   J dz            = A (G dz)
// which replaces the N K x 2 N Jacobian build and every loop over its rows by work on 2 NP = 24 position rows.

// rows: sigma = lam / s, 1/s, sigma r_p + lam; returns the lane-partial residual norms
// resto: the CBF rows (i < mc) are the elastic rows of the feasibility restoration (ipm::resto_row)
__device__ __forceinline__ void row_pass(const MpcMem& W, const MpcConst& c, int lane, double& e_p, double& e_c0, double& lmax,
                                         bool resto = false, double mu = 0.0, double rho = 0.0) {
    lane = opaque(lane);
    double* is = W.is;
    double* vb = W.dlam;
    e_p = 0.0; e_c0 = 0.0; lmax = 0.0;
    for (int i = lane; i < c.m; i += 64) {
        const double s = W.sl[i], l = W.lam[i];
        double rp, inv, vbi;
        if (resto && i < c.mc) {
            const double t = W.tel[i];
            ipm::resto_row(W.g[i], s, l, t, mu, rho, rp, inv, vbi);
            e_c0 = fmax(e_c0, fabs(t * (rho - l)));
        } else {
            rp = W.g[i] - s;
            inv = rcp_(s);
            vbi = (l * inv) * rp + l;
        }
        is[i] = inv; vb[i] = vbi;
        e_p = fmax(e_p, fabs(rp));
        e_c0 = fmax(e_c0, fabs(s * l));
        lmax = fmax(lmax, l);
    }
    e_p = wmax(e_p); e_c0 = wmax(e_c0); lmax = wmax(lmax);
}

// positions: lane k < NP builds  q_k = d L / d p_k,  Om_k,  the position-space vectors A'(1/s), A'(sig r_p + lam),
// row block k of Phi, and the suffix sums the structured Hessian terms need.  Multipliers are those of the scaled
// problem (objective times sf), so everything here is already scaled.
template <int KT, bool ROW16, bool OD, bool UNI = false>
__device__ __forceinline__ double stage_pass(const MpcMem& W, const MpcConst& c, int lane, double sf) {
    lane = opaque(lane);
    // ROW16: the obstacle loop of stage k is split over lanes k, k + 16, k + 32, k + 48 (jo = part, part + 4, ..) and
    // the partial sums are added across the four rows; every row then holds the totals, row 0 stores them.
    const int N = c.N, K = c.K, NP = N + 2;
    const int part = ROW16 ? (lane >> 4) : 0, k = ROW16 ? (lane & 15) : lane;
    constexpr int PSTEP = ROW16 ? 4 : 1;
    const double* is = W.is;
    const double* vb = W.dlam;
    double q0 = 0.0, q1 = 0.0;
    double oxx = 0, oxy = 0, oyy = 0, ya0 = 0, ya1 = 0, yb0 = 0, yb1 = 0;
    double f0xx = 0, f0xy = 0, f0yy = 0, f1[4] = {0, 0, 0, 0}, f2[4] = {0, 0, 0, 0};
    // optimal decay, stage kappa = k: D = sum sig a a' (+ Hessian), lh = sum lam h0, t_v = sum a v, C (6 x 2)
    double S11 = 0, S12 = 0, S22 = 0, lh = 0, T0[2] = {0, 0}, T1[2] = {0, 0}, T2[2] = {0, 0};
    double Cm[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                 // Cm[(2 t + d) * 2 + i]
    if (k < NP) {
        const bool t0 = k <= N - 1, t1 = k >= 1 && k <= N, t2 = k >= 2;       // row kappa = k - t exists
        const int r0 = (t0 ? k : 0) * K, r1 = (t1 ? k - 1 : 0) * K, r2 = (t2 ? k - 2 : 0) * K;
        // weights of the three rows that touch position k (OD: they depend on the row's stage)
        const double w0 = t0 ? (OD ? W.w0s[t0 ? k : 0] : c.w0) : 0.0, w1 = t1 ? (OD ? W.w1s[t1 ? k - 1 : 0] : c.w1) : 0.0,
                     w2 = t2 ? c.w2 : 0.0;
        const double w1k = OD ? W.w1s[t0 ? k : 0] : c.w1;                 // t = 1 weight of row kappa = k
        double cr1 = 0.0, cr2 = 0.0;                                      // d2 row / d rho_i d p_kappa = cr_i dh0
        if constexpr (OD) {
            const int kk = t0 ? k : 0;
            cr1 = UNI ? c.al1 : -c.al1 + c.al1 * c.al2 * W.rho[2 * kk + 1];
            cr2 = UNI ? 0.0 : -c.al2 + c.al1 * c.al2 * W.rho[2 * kk];
        }
        const int k1 = k + 1 < NP ? k + 1 : k, k2 = k + 2 < NP ? k + 2 : k;
        const double pk0 = W.pos[2 * k], pk1 = W.pos[2 * k + 1];
        // not unrolled: both iterations' 24 loads in flight at once cost 16 VGPR spills under the 256-register cap
#pragma unroll 1
        for (int jo = part; jo < K; jo += PSTEP) {
            const double l0 = W.lam[r0 + jo], l1 = W.lam[r1 + jo], l2 = W.lam[r2 + jo];
            const double i0 = is[r0 + jo], i1 = is[r1 + jo], i2 = is[r2 + jo];
            const double b0 = vb[r0 + jo], b1 = vb[r1 + jo], b2 = vb[r2 + jo];
            const double s0 = l0 * i0, s1 = l1 * i1, s2 = l2 * i2;       // Sigma = lam / s
            const int e = k * K + jo, e1 = k1 * K + jo, e2 = k2 * K + jo;
            const double d0 = W.dh[2 * e], d1 = W.dh[2 * e + 1];
            const double a0 = W.dh[2 * e1], a1 = W.dh[2 * e1 + 1], g0 = W.dh[2 * e2], g1 = W.dh[2 * e2 + 1];
            double hxx = 2.0, hxy = 0.0, hyy = 2.0;                  // d2h/dp2 of a circle; superellipsoids recompute theirs
            if (W.obs[7 * jo + 6] != 0.0) {
                double h_, g0_, g1_;
                barrier_at<!ROW16>(pk0, pk1, W.obs + 7 * jo, c, true, h_, g0_, g1_, hxx, hxy, hyy);
            }
            const double ml = w0 * l0 + w1 * l1 + w2 * l2, ma = w0 * i0 + w1 * i1 + w2 * i2, mb = w0 * b0 + w1 * b1 + w2 * b2;
            const double c0 = w0 * w0 * s0 + w1 * w1 * s1 + w2 * w2 * s2;
            const double c1 = w1 * c.w2 * s1 + w0 * w1k * s0;       // rows kappa = k-1 (t = 1, 2) and kappa = k (t = 0, 1)
            const double c2 = w0 * c.w2 * s0;                        // row kappa = k (t = 0, 2)
            oxx -= ml * hxx; oxy -= ml * hxy; oyy -= ml * hyy;
            q0 -= ml * d0; q1 -= ml * d1;
            ya0 += ma * d0; ya1 += ma * d1; yb0 += mb * d0; yb1 += mb * d1;
            f0xx += c0 * d0 * d0; f0xy += c0 * d0 * d1; f0yy += c0 * d1 * d1;
            f1[0] += c1 * d0 * a0; f1[1] += c1 * d0 * a1; f1[2] += c1 * d1 * a0; f1[3] += c1 * d1 * a1;
            f2[0] += c2 * d0 * g0; f2[1] += c2 * d0 * g1; f2[2] += c2 * d1 * g0; f2[3] += c2 * d1 * g1;
            if constexpr (OD) {
                const double tz = t0 ? 1.0 : 0.0;
                const double aa1 = tz * W.A1[r0 + jo], aa2 = tz * W.A2[r0 + jo];
                const double l0z = tz * l0, s0z = tz * s0;
                S11 += s0z * aa1 * aa1; S12 += s0z * aa1 * aa2; S22 += s0z * aa2 * aa2;
                lh += l0z * W.hk[e];
                T0[0] -= l0z * aa1; T0[1] -= l0z * aa2;
                T1[0] += i0 * aa1; T1[1] += i0 * aa2;
                T2[0] += b0 * aa1; T2[1] += b0 * aa2;
                const double e01 = s0z * w0 * aa1 - l0z * cr1, e02 = s0z * w0 * aa2 - l0z * cr2;
                // d2 row / d rho_i d p_(k+1) = d w1 / d rho_i dh1: a_i for the rel-degree-2 row, nothing for the rel-degree-1 one
                const double e11 = s0z * w1k * aa1 - (UNI ? 0.0 : l0z * c.al1), e12 = s0z * w1k * aa2 - (UNI ? 0.0 : l0z * c.al2);
                const double e21 = s0z * c.w2 * aa1, e22 = s0z * c.w2 * aa2;
                Cm[0] += d0 * e01; Cm[1] += d0 * e02; Cm[2] += d1 * e01; Cm[3] += d1 * e02;
                Cm[4] += a0 * e11; Cm[5] += a0 * e12; Cm[6] += a1 * e11; Cm[7] += a1 * e12;
                Cm[8] += g0 * e21; Cm[9] += g0 * e22; Cm[10] += g1 * e21; Cm[11] += g1 * e22;
            }
        }
    }
    if constexpr (ROW16) {
        auto rows4 = [](double& v) { v = sum_rows4(v); };
        rows4(oxx); rows4(oxy); rows4(oyy); rows4(q0); rows4(q1); rows4(ya0); rows4(ya1); rows4(yb0); rows4(yb1);
        rows4(f0xx); rows4(f0xy); rows4(f0yy);
#pragma unroll
        for (int t = 0; t < 4; ++t) { rows4(f1[t]); rows4(f2[t]); }
        if constexpr (OD) {
            rows4(S11); rows4(S12); rows4(S22); rows4(lh);
#pragma unroll
            for (int t = 0; t < 2; ++t) { rows4(T0[t]); rows4(T1[t]); rows4(T2[t]); }
#pragma unroll
            for (int t = 0; t < 12; ++t) rows4(Cm[t]);
        }
    }
    double e_rho = 0.0;
    if constexpr (OD) {
        // eliminate the decay variables of stage kappa = k:  E = C D^-1 C',  c_v = C D^-1 t_v  go to the exchange scratch,
        // C, D^-1 and t_v stay in W.ods for the back-substitution after the solve
        if (k < N && part == 0) {
            const double r1 = W.rho[2 * k], r2 = W.rho[2 * k + 1];
            T0[0] += sf * 2.0 * c.ps1 * (r1 - c.rf1); T0[1] += sf * 2.0 * c.ps2 * (r2 - c.rf2);
            e_rho = fmax(fabs(T0[0]), fabs(T0[1]));
            double d11 = sf * 2.0 * c.ps1 + S11, d22 = sf * 2.0 * c.ps2 + S22;
            const double d12 = UNI ? 0.0 : S12 - c.al1 * c.al2 * lh;
            // D^-1 is applied through the eigen-decomposition  D = ls vs vs' + lw vw vw'.  With a1 ~ a2 the two
            // columns of C and the rows of D are nearly parallel: D has one huge (sum sig a a') and one small (the
            // penalty) eigenvalue, and an explicit inverse loses the huge direction against the small one
            // (relative error ~1e-3 in C D^-1 C' at sig ~ 1e6, which stalls the last iterations).
            const double tr = d11 + d22, df = d11 - d22, rad = sqrt(df * df + 4.0 * d12 * d12);
            double ls = 0.5 * (tr + rad), lw = (d11 * d22 - d12 * d12) / ls;
            double vx = df >= 0.0 ? df + rad : 2.0 * d12, vy = df >= 0.0 ? 2.0 * d12 : rad - df;
            const double vn = vx * vx + vy * vy;
            if (vn > 0.0) { const double rn = rsqrt_(vn); vx *= rn; vy *= rn; } else { vx = 1.0; vy = 0.0; }
            // shift to positive definite: + max(0, eps - lambda_min) I, eps = 1e-8 max(1, |d11| + |d22|)
            const double sh = fmax(0.0, 1e-8 * fmax(1.0, fabs(d11) + fabs(d22)) - lw);
            ls += sh; lw += sh;
            double ils = 1.0 / ls, ilw = 1.0 / lw;
            if constexpr (UNI) {                                          // one live decay variable: the block is the scalar d11 > 0
                vx = 1.0; vy = 0.0; ils = 1.0 / d11; ilw = 0.0;
            }
            double cs[6], cw[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                cs[r] = Cm[2 * r] * vx + Cm[2 * r + 1] * vy;
                cw[r] = -Cm[2 * r] * vy + Cm[2 * r + 1] * vx;
            }
            double* xs = W.XS + (size_t)39 * k;
            int o = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int q2 = r; q2 < 6; ++q2) xs[o++] = cs[r] * cs[q2] * ils + cw[r] * cw[q2] * ilw;
            const double p0s = (vx * T0[0] + vy * T0[1]) * ils, p0w = (-vy * T0[0] + vx * T0[1]) * ilw;
            const double p1s = (vx * T1[0] + vy * T1[1]) * ils, p1w = (-vy * T1[0] + vx * T1[1]) * ilw;
            const double p2s = (vx * T2[0] + vy * T2[1]) * ils, p2w = (-vy * T2[0] + vx * T2[1]) * ilw;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                xs[21 + r] = cs[r] * p0s + cw[r] * p0w;
                xs[27 + r] = cs[r] * p1s + cw[r] * p1w;
                xs[33 + r] = cs[r] * p2s + cw[r] * p2w;
            }
            double* od = W.ods + (size_t)22 * k;
#pragma unroll
            for (int t = 0; t < 12; ++t) od[t] = Cm[t];
            od[12] = vx; od[13] = vy; od[14] = ils; od[21] = ilw;
            od[15] = T0[0]; od[16] = T0[1]; od[17] = T1[0]; od[18] = T1[1]; od[19] = T2[0]; od[20] = T2[1];
        }
        SC_SYNC();
        if (k < NP) {
            // gather: Phi -= sum_kappa E_kappa,  Y1 -= c_1,  Y2 -= c_2 + c_0   (rows of stages kappa = k - t, t = 0..2)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int kap = k - t;
                if (kap < 0 || kap > N - 1) continue;
                const double* xs = W.XS + (size_t)39 * kap;
                auto E = [&](int r, int q2) { return xs[r * 6 - (r * (r - 1)) / 2 + (q2 - r)]; };
                f0xx -= E(2 * t, 2 * t); f0xy -= E(2 * t, 2 * t + 1); f0yy -= E(2 * t + 1, 2 * t + 1);
                if (t <= 1) {
                    f1[0] -= E(2 * t, 2 * t + 2); f1[1] -= E(2 * t, 2 * t + 3);
                    f1[2] -= E(2 * t + 1, 2 * t + 2); f1[3] -= E(2 * t + 1, 2 * t + 3);
                }
                if (t == 0) { f2[0] -= E(0, 4); f2[1] -= E(0, 5); f2[2] -= E(1, 4); f2[3] -= E(1, 5); }
                ya0 -= xs[27 + 2 * t]; ya1 -= xs[27 + 2 * t + 1];
                yb0 -= xs[33 + 2 * t] + xs[21 + 2 * t]; yb1 -= xs[33 + 2 * t + 1] + xs[21 + 2 * t + 1];
            }
        }
    }
    if (k < NP) {
        if (k >= 1 && k <= N) {
            oxx += sf * 2.0 * c.Qx; oyy += sf * 2.0 * c.Qy;
            q0 += sf * 2.0 * c.Qx * (W.pos[2 * k] - c.gx);
            q1 += sf * 2.0 * c.Qy * (W.pos[2 * k + 1] - c.gy);
        }
    }
    if (k < NP && part == 0) {
        W.Y[2 * k] = q0; W.Y[2 * k + 1] = q1;
        W.Y[2 * NP + 2 * k] = ya0; W.Y[2 * NP + 2 * k + 1] = ya1;
        W.Y[4 * NP + 2 * k] = yb0; W.Y[4 * NP + 2 * k + 1] = yb1;
        // Phi band rows: Phi[(2k+d) * 10 + (delta + 2) * 2 + d'] = Phi[(k, d), (k + delta, d')]
        double* R0 = W.Phi + (size_t)(2 * k) * 10;
        double* R1 = R0 + 10;
        R0[4] = f0xx + oxx; R0[5] = f0xy + oxy; R1[4] = f0xy + oxy; R1[5] = f0yy + oyy;
        if (k + 1 < NP) {
            R0[6] = f1[0]; R0[7] = f1[1]; R1[6] = f1[2]; R1[7] = f1[3];
            double* Q0 = W.Phi + (size_t)(2 * (k + 1)) * 10;      // transposed block at (k+1, k): delta = -1
            Q0[2] = f1[0]; Q0[3] = f1[2]; Q0[12] = f1[1]; Q0[13] = f1[3];
        }
        if (k + 2 < NP) {
            R0[8] = f2[0]; R0[9] = f2[1]; R1[8] = f2[2]; R1[9] = f2[3];
            double* Q0 = W.Phi + (size_t)(2 * (k + 2)) * 10;      // delta = -2
            Q0[0] = f2[0]; Q0[1] = f2[2]; Q0[10] = f2[1]; Q0[11] = f2[3];
        }
    }
    // suffix sums (wave scans): qbar_i = sum_{k > i} q_k;  A_i = qbar_i . (-s_i, c_i),  B_i = v_i qbar_i . (c_i, s_i);
    // SA[t] = sum_{i >= t} A_i, SB likewise (t = 0..N+1, zero at N+1);  SS[t] = sum_{k >= t} (sig+_k + sig-_k), k = 1..N
    if (k >= NP) { q0 = 0.0; q1 = 0.0; }
    const double qs0 = suffix_sum<ROW16>(q0, lane) - q0, qs1 = suffix_sum<ROW16>(q1, lane) - q1;
    double Ai = 0.0, Bi = 0.0, sk = 0.0;
    if (k <= N) {
        const double ci = W.C[k], si = W.S[k];
        Ai = qs0 * (-si) + qs1 * ci;
        Bi = W.V[k] * (qs0 * ci + qs1 * si);
        if (!UNI && k >= 1) {
            const int r = c.mc + 2 * (k - 1);
            sk = W.lam[r] * is[r] + W.lam[r + 1] * is[r + 1];
        }
    }
    // UNI: v_j only moves stage j, so the (v_j, omega_i) curvature is A_j itself (i < j), not a suffix sum
    const double sa = UNI ? Ai : suffix_sum<ROW16>(Ai, lane), sb = suffix_sum<ROW16>(Bi, lane), ss = suffix_sum<ROW16>(sk, lane);
    if (k < NP && part == 0) { W.SA[k] = sa; W.SB[k] = sb; W.SS[k] = ss; }
    return OD ? wmax(e_rho) : 0.0;                                    // |r_d| of the decay variables
}

// columns: cv[v][col] for the three row vectors  v = 0: r_d = sf grad f - J' lam;  1: J'(1/s);  2: J'(sig r_p + lam).
// Every inner loop has a trip count that does not depend on the lane (structural zeros of G, masks on the stage
// sums), so with compile-time N the loads of a lane are issued back to back instead of one round trip per term.
template <bool OD, bool UNI = false>
// zeta > 0 (restoration): the objective is zeta/2 |z - z_R|^2 with z_R kept in W.zb (sf = 0 switches the cost off)
__device__ __forceinline__ double col_pass(const MpcMem& W, const MpcConst& c, int lane, double sf, double zeta = 0.0) {
    lane = opaque(lane);
    const int N = c.N, n = c.n, NP = N + 2;
    double e_d = 0.0;
    for (int idx = lane; idx < 3 * n; idx += 64) {
        const int v = idx / n, col = idx - v * n, j = col >> 1;
        const double* y = W.Y + (size_t)v * 2 * NP;
        // (an offset select: a select between the pointers themselves makes the compiler spill MpcMem to scratch)
        const double* vec = W.lam + (v == 0 ? 0 : (v == 1 ? (W.is - W.lam) : (W.dlam - W.lam)));
        double acc = 0.0;
#pragma unroll
        for (int row = UNI ? 2 : 4; row < 2 * NP; ++row) acc += W.dP[(size_t)row * n + col] * y[row];   // leading rows of G are zero
        double sp = 0.0, sx = 0.0;
        const double* Xs = W.V + ((col & 1) ? (W.TH - W.V) : 0);
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            const double dv = UNI ? 0.0 : vec[c.mc + 2 * (k - 1) + 1] - vec[c.mc + 2 * (k - 1)], xk = Xs[k];
            sp += k > j ? dv : 0.0;                               // speed rows of stages k > j
            sx += k > j ? xk : 0.0;                               // sum_{k > j} theta_k | v_k
        }
        if (UNI && !(col & 1)) sx = 0.0;                          // the speed is an input, not a state: no state cost on it
        double sb = vec[c.mc + c.ns + n + col] - vec[c.mc + c.ns + col];
        if (!UNI && !(col & 1)) sb += c.dt * sp;
        if (v == 0) {
            const double Rc = (col & 1) ? c.R1 : c.R0, Qs = (col & 1) ? c.Qth : c.Qv;
            const double prev = col >= 2 ? W.z[col - 2] : ((col & 1) ? c.up1 : c.up0);
            double gr = 2.0 * Qs * c.dt * sx + 2.0 * Rc * (OD ? W.z[col] : W.z[col] - prev);
            if (!OD && col + 2 < n) gr -= 2.0 * Rc * (W.z[col + 2] - W.z[col]);
            acc += sf * gr - sb;
            if (zeta != 0.0) acc += zeta * (W.z[col] - W.zb[col]);
            e_d = fmax(e_d, fabs(acc));
        } else {
            acc += sb;
        }
        W.cv[idx] = acc;
    }
    return wmax(e_d);
}

// ---- f64 MFMA (v_mfma_f64_16x16x4_f64) -----------------------------------------------------------------------------
// Operand layout (MI355X guide, "f64 MFMA"): lane l feeds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// result register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15].
// p_0 and p_1 do not depend on z, so rows 0..3 of G vanish: both products run on the n = 2 N rows 4.. of G, T and
// Phi (G', T', Phi').  k-steps outside a tile's structural non-zeros are skipped:
//   G'[kk][col] = 0 for kk < 2 (col >> 1)   (block lower triangular),   Phi'[i][kk] = 0 for |kk/2 - i/2| > 2.

// k-steps lo, lo + 4, .. < hi of one tile: operands of S steps are loaded first (one LDS round trip), then S MFMAs
template <int S, typename LoadA, typename LoadB>
__device__ __forceinline__ d4_t mfma_ksteps(int lo, int hi, LoadA load_a, LoadB load_b) {
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int kb = lo; kb < hi; kb += 4 * S) {
        double a[S], b[S];
#pragma unroll
        for (int t = 0; t < S; ++t) { a[t] = load_a(kb + 4 * t); b[t] = load_b(kb + 4 * t); }
#pragma unroll
        for (int t = 0; t < S; ++t)
            if (kb + 4 * t < hi) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc, 0, 0, 0);
    }
    return acc;
}

// T' = Phi' G'   (9 MFMAs for N = 10)
// R0: first row of G that can be non-zero (4 for DynamicUnicycle2D: p_0, p_1 fixed; 2 for Unicycle2D: p_0 fixed)
template <int S, int R0 = 4>
__device__ __forceinline__ void phi_times_G(const MpcMem& W, const MpcConst& c, int lane) {
    lane = opaque(lane);
    const int n = c.n, nt = (n + 15) >> 4, q = lane >> 4, l15 = lane & 15;
    for (int ti = 0; ti < nt; ++ti) {
        const int i = 16 * ti + l15, R = i + R0, kR = R >> 1;
        const bool okA = i < n;
        const double* band = W.Phi + (size_t)(okA ? R : R0) * 10;
        for (int tj = 0; tj < nt; ++tj) {
            const int cB = 16 * tj + l15;
            const bool okB = cB < n;
            const int cb = okB ? cB : 0;
            int lo = 16 * ti - 4;
            lo = lo > 16 * tj ? lo : 16 * tj;
            lo = lo > 0 ? lo : 0;
            const int hi = (16 * ti + 20) < n ? (16 * ti + 20) : n;
            const d4_t acc = mfma_ksteps<S>(lo, hi,
                [&](int k0) {
                    const int kk = k0 + q, Cc = kk + R0, dl = (Cc >> 1) - kR;
                    const bool inband = dl >= -2 && dl <= 2, ok = okA && kk < hi && inband;
                    const double a = band[(inband ? dl + 2 : 2) * 2 + (Cc & 1)];
                    return ok ? a : 0.0;
                },
                [&](int k0) {
                    const int kk = k0 + q;
                    const bool ok = okB && kk < hi;
                    const double b = W.dP[(size_t)(ok ? kk + R0 : R0) * n + cb];
                    return ok ? b : 0.0;
                });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + q + 4 * r, col = 16 * tj + l15;
                if (row < n && col < n) W.T[(size_t)(row + R0) * n + col] = acc[r];
            }
        }
    }
}

// structured part of the condensed matrix at (r, cc)
template <bool OD, bool UNI = false>
__device__ __forceinline__ double condensed_base(const MpcMem& W, const MpcConst& c, double sf, int r, int cc) {
    const int N = c.N, n = c.n;
    const int jr = r >> 1, jc = cc >> 1, jm = jr > jc ? jr : jc;
    const double dt2 = c.dt * c.dt, dt3 = dt2 * c.dt;
    const bool ra = !(r & 1), ca = !(cc & 1);
    const double ss = W.SS[jm + 1], sb = W.SB[jm + 1], sa = W.SA[jm + 1];
    const double vaa = sf * 2.0 * c.Qv * dt2 * (double)(N - jm) + dt2 * ss;             // d v_k: objective + speed rows
    const double vww = sf * 2.0 * c.Qth * dt2 * (double)(N - jm) - dt3 * sb;
    double acc = (ra && ca) ? vaa : ((!ra && !ca) ? vww : dt3 * sa);
    if constexpr (UNI) {
        // (v_j, omega_i): dt^2 A_j for i < j (SA holds A itself);  (v, v): nothing;  (omega, omega): as above
        const int jv = ra ? jr : jc, jw = ra ? jc : jr;
        acc = (ra && ca) ? 0.0 : ((!ra && !ca) ? vww : ((jw < jv) ? dt2 * W.SA[jv] : 0.0));
    }
    const double Rc = (r & 1) ? c.R1 : c.R0;                                               // input-rate penalty 2 D' R D
    const int bx = c.mc + c.ns + r;
    const double dg = sf * 2.0 * Rc * ((!OD && r + 2 < n) ? 2.0 : 1.0) + W.lam[bx] * W.is[bx] + W.lam[bx + n] * W.is[bx + n];
    acc += (r == cc) ? dg : ((!OD && (r == cc + 2 || cc == r + 2)) ? -sf * 2.0 * Rc : 0.0);
    return acc;
}

// M = G'' T' + structured terms   (7 MFMAs for N = 10; lower tiles, mirrored)
template <int S, bool OD, bool UNI = false>
__device__ __forceinline__ void condense_mfma(const MpcMem& W, const MpcConst& c, int lane, double sf) {
    lane = opaque(lane);
    const int n = c.n, nt = (n + 15) >> 4, q = lane >> 4, l15 = lane & 15;
    for (int ti = 0; ti < nt; ++ti) {
        for (int tj = 0; tj <= ti; ++tj) {
            const int cA = 16 * ti + l15, cB = 16 * tj + l15;
            const bool okA = cA < n, okB = cB < n;
            const int ca = okA ? cA : 0, cb = okB ? cB : 0;
            double base[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + q + 4 * r, col = 16 * tj + l15;
                base[r] = (row < n && col < n) ? condensed_base<OD, UNI>(W, c, sf, row, col) : 0.0;
            }
            const d4_t acc = mfma_ksteps<S>(16 * ti, n,
                [&](int k0) {
                    const int kk = k0 + q;
                    const bool ok = okA && kk < n;
                    const double a = W.dP[(size_t)((ok ? kk : 0) + (UNI ? 2 : 4)) * n + ca];
                    return ok ? a : 0.0;
                },
                [&](int k0) {
                    const int kk = k0 + q;
                    const bool ok = okB && kk < n;
                    const double b = W.T[(size_t)((ok ? kk : 0) + (UNI ? 2 : 4)) * n + cb];
                    return ok ? b : 0.0;
                });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * ti + q + 4 * r, col = 16 * tj + l15;
                if (row < n && col < n) {
                    const double v = acc[r] + base[r];
                    W.M[(size_t)row * n + col] = v;
                    if (ti != tj) W.M[(size_t)col * n + row] = v;
                }
            }
        }
    }
}

// Factor M + delta I and solve for the right-hand side, all in registers (mpc_chol.hpp); false on a pivot <= 0.
// Arguments are offsets (doubles) into the workgroup's LDS block.  NOT inlined on purpose: unrolled for n = 40 the
// factorisation and the two triangular solves are ~45 KB of code, and inlined they push the interior-point loop of the
// N = 20 kernel past the +-128 KB reach of s_cbranch; the relaxed long branches need a free SGPR pair, which a kernel
// with a few hundred spilled SGPRs does not have -- the result was a kernel whose iterates changed with unrelated
// code motion.  Out of line the loop stays ~100 KB for every instantiation.
template <int nn>
__device__ __attribute__((noinline)) bool chol_factor_solve(int oM, int oRhs, int oLt, int oOut, double delta, int lane) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const double* M = lds + oM;
    double a[nn], diag;
    const int row = lane < nn ? lane : 0;                                  // lanes >= n: unused copies of row 0
#pragma unroll
    for (int k = 0; k < nn; ++k) a[k] = M[row * nn + k] + (lane == k ? delta : 0.0);
    if (!chol_reg<nn>(a, lane, diag)) return false;
    const double x = chol_solve_reg<nn>(a, diag, lds[oRhs + row], lds + oLt, lane);
    if (lane < nn) lds[oOut + lane] = x;
    SC_SYNC();
    return true;
}

// NT, KT > 0: horizon and obstacle count are compile-time constants (index arithmetic folds to shifts and
// multiplies); NT == 0: run-time sizes.
struct OdExtra { double omega_ref[2], p_sb[2]; };

// The element type of the caller's arrays (p.io_dtype) only matters for the handful of loads at the start and stores at
// the end, so it is a run-time switch: one body per (NT, KT, OD, UNI) instead of two.
template <int NT, int KT, bool OD = false, bool UNI = false>
__device__ __forceinline__ void mpccbf_body(double* sm, const sc_mpccbf_params& p, const long long B, const int K_rt,
                                            const void* __restrict__ X, const void* __restrict__ u_prev,
                                            const void* __restrict__ goal, const void* __restrict__ obs,
                                            void* __restrict__ u_out, int* __restrict__ status_out,
                                            int* __restrict__ iters_out, void* __restrict__ z_out, const ipm::Cont& ct,
                                            const OdExtra od = OdExtra{{1.0, 1.0}, {0.0, 0.0}}, void* __restrict__ rho_out = nullptr) {
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };
    const int lane = threadIdx.x;
    long long prob;
    if (!ipm::cont_problem(ct, B, prob)) return;                        // mpc_cont.hpp: block index, or an entry of the previous launch's queue
    MpcConst c;
    const int K = KT > 0 ? KT : K_rt;
    c.N = NT > 0 ? NT : p.horizon; c.K = K; c.n = 2 * c.N; c.mc = c.N * K; c.ns = UNI ? 0 : 2 * c.N; c.m = c.mc + c.ns + 2 * c.n;
    c.dt = p.dt; c.Qx = p.Q[0]; c.Qy = p.Q[1]; c.Qth = p.Q[2]; c.Qv = UNI ? 0.0 : p.Q[3]; c.R0 = p.R[0]; c.R1 = p.R[1];
    const double g1 = p.alpha1 + p.alpha2, g2 = p.alpha1 * p.alpha2;
    c.w0 = 1.0 - g1 + g2; c.w1 = g1 - 2.0; c.w2 = 1.0;
    if constexpr (UNI) { c.w0 = -(1.0 - p.alpha1); c.w1 = 1.0; c.w2 = 0.0; }   // relative degree 1: h(p_k+1) - (1 - alpha) h(p_k)
    c.vmax = p.v_max; c.amax = p.u_max[0]; c.wmaxu = p.u_max[1]; c.Rrob = p.robot_radius; c.beta = p.beta;
    c.x0 = ld(X, prob * 4 + 0); c.y0 = ld(X, prob * 4 + 1); c.th0 = ld(X, prob * 4 + 2); c.v0 = UNI ? 0.0 : ld(X, prob * 4 + 3);
    c.up0 = ld(u_prev, prob * 2 + 0); c.up1 = ld(u_prev, prob * 2 + 1);
    c.gx = ld(goal, prob * 2 + 0); c.gy = ld(goal, prob * 2 + 1);
    c.al1 = p.alpha1; c.al2 = p.alpha2; c.ps1 = od.p_sb[0]; c.ps2 = od.p_sb[1]; c.rf1 = od.omega_ref[0]; c.rf2 = od.omega_ref[1];
    const int N = c.N, n = c.n, m = c.m;
    const MpcMem W = carve(sm, N, K, OD, UNI);
    constexpr bool ROW16 = NT > 0 && NT + 2 <= 16;
    constexpr int MS = NT > 0 ? (2 * NT + 3) / 4 : 4;                  // MFMA k-steps batched per LDS round trip
    for (int e = lane; e < 2 * (N + 2) * n; e += 64) W.dP[e] = 0.0;      // structural zeros of G stay

    // the scalars of the interior-point loop (all wave-uniform); a continuation launch loads them with the arrays below
    constexpr bool RESTO = !OD;
    const int nel = OD ? 0 : c.mc;                                       // elastic variables of the restoration
    double* const cst = ct.state ? ct.state + prob * ct.stride : nullptr;
    Prof pf;
    pf.start();
    double f = 0.0, sf = 1.0, mu = p.mu_init;
    double nu = 10.0, delta_last = 0.0, e_best = 1e300;
    int n_acc = 0, it0 = 1;
    // feasibility restoration (mpc_ipm_common.hpp; oracle/mpc_cbf.py: solve)
    bool resto = false;
    int n_resto = 0, n_small = 0;                                       // n_small: consecutive tiny accepted steps at an infeasible z
    double theta_R = 0.0, mu_reg = mu;
    // stalled restorations (sc_resto_params.retry_max / stall_iter; oracle/mpc_cbf.py: solve): the damping a retried step starts its
    // inertia correction with, retries so far at this iterate, the violation the stall counter measures against, iterations since
    double delta_force = 0.0, theta_ref = 0.0;
    int n_retry = 0, n_stall = 0;
    if (ct.resume) {
        // the state a previous launch left (mpc_cont.hpp): [scalars | z | zb | sl | lam | obs | tel | rho | rhob]
        const double* a = cst + ipm::CONT_SCALARS;
        SC_SYNC();                                                       // (the zeros of G above: tel may live in its first rows)
        ipm::cont_copy(W.z, a, n, lane, 64); a += n;
        ipm::cont_copy(W.zb, a, n, lane, 64); a += n;
        ipm::cont_copy(W.sl, a, m, lane, 64); a += m;
        ipm::cont_copy(W.lam, a, m, lane, 64); a += m;
        ipm::cont_copy(W.obs, a, K * 7, lane, 64); a += K * 7;
        if (nel) { ipm::cont_copy(W.tel, a, nel, lane, 64); a += nel; }
        if constexpr (OD) { ipm::cont_copy(W.rho, a, n, lane, 64); a += n; ipm::cont_copy(W.rhob, a, n, lane, 64); }
        it0 = (int)cst[0] + 1; mu = cst[1]; nu = cst[2]; delta_last = cst[3]; e_best = cst[4]; n_acc = (int)cst[5];
        resto = cst[6] != 0.0; n_resto = (int)cst[7]; n_small = (int)cst[8]; theta_R = cst[9]; mu_reg = cst[10]; sf = cst[11];
        delta_force = cst[12]; n_retry = (int)cst[13]; theta_ref = cst[14]; n_stall = (int)cst[15];
        SC_SYNC();
    } else {
    const size_t obase = p.obs_shared ? 0 : (size_t)prob * K * 7;
    for (int e = lane; e < K * 7; e += 64) W.obs[e] = ld(obs, obase + e);
    SC_SYNC();
    ipm::normalise_obstacle_flags(W.obs, K, lane, 64);
    // set_initial_guess (mpc_cbf.py:369): u_prev at every stage, pulled strictly inside the box
    for (int i = lane; i < n; i += 64) {
        const double ub = (i & 1) ? c.wmaxu : c.amax;
        const double u = (i & 1) ? c.up1 : c.up0;
        W.z[i] = fmin(fmax(u, -0.99 * ub), 0.99 * ub);
        if constexpr (OD) W.rho[i] = (i & 1) ? c.rf2 : c.rf1;            // decay variables start at their references
    }
    SC_SYNC();

    f = eval_values<ROW16, OD, UNI>(W.z, W.rho, W, c, lane, true, pf, 12);
    if (ct.it_stop < 0) {
        // classify only (mpc_cont.hpp): is a CBF row violated at the initial guess?  Those are the solves that crawl or restore
        // feasibility; the next launch starts them first
        double th0 = 0.0;
        for (int i = lane; i < c.mc; i += 64) th0 += fmax(0.0, -W.g[i]);
        th0 = wsum(th0);
        if (lane == 0) ipm::cont_push(ct, prob, th0 > 0.0);
        return;
    }
    // steep (superellipsoid) barriers: IPOPT-style gradient-based row scaling from the initial guess, then a fresh evaluation
    if (ipm::scale_steep_barriers(W.obs, K, W.dh, N + 2, lane, 64, [](double v) { return wmax(v); }, [] { SC_SYNC(); }))
        f = eval_values<ROW16, OD, UNI>(W.z, W.rho, W, c, lane, true, pf, 12);
    // objective scaling from |grad f|_inf at the start: with lam = 0 the column pass returns r_d = grad f
    for (int i = lane; i < m; i += 64) { W.sl[i] = fmax(W.g[i], 1e-2); W.lam[i] = 0.0; }
    SC_SYNC();
    double e_p, e_c0, lmax;
    row_pass(W, c, lane, e_p, e_c0, lmax);
    SC_SYNC();
    stage_pass<KT, ROW16, OD, UNI>(W, c, lane, 1.0);
    SC_SYNC();
    const double gmax = col_pass<OD, UNI>(W, c, lane, 1.0);
    sf = fmin(1.0, 100.0 / fmax(1e-12, gmax));                          // objective scaling
    SC_SYNC();
    for (int i = lane; i < m; i += 64) W.lam[i] = mu * W.is[i];
    SC_SYNC();
    for (int i = lane; i < n; i += 64) { W.zb[i] = W.z[i]; if constexpr (OD) W.rhob[i] = W.rho[i]; }
    }

    int status = SC_STATUS_INACCURATE, it = 0;
    pf.start();
    const double tau = 0.995;
    const int acc_iter = p.acceptable_iter > 0 ? p.acceptable_iter : 15;
    const double rho_R = p.resto.rho;
    bool pending = false;
    for (it = it0; it <= p.max_iter; ++it) {
        if (cst && it > ct.it_stop) { pending = true; break; }            // the cap of this launch: the solve goes on in the next one
        if (it > 1 || ct.resume) f = eval_values<ROW16, OD, UNI>(W.z, W.rho, W, c, lane, true, pf, 12);
        SC_PH(0);
        double theta = 0.0;                                              // l1 violation of the elastic (CBF) rows at z
        if constexpr (RESTO) {
            for (int i = lane; i < c.mc; i += 64) theta += fmax(0.0, -W.g[i]);
            theta = wsum(theta);
            if (resto && theta <= fmax(n_resto == 1 ? p.resto.kappa * theta_R : 0.0, p.resto.theta_tol)) {
                // enough of the violation is gone (first entry: a tenth of it; later entries run until nothing is left -- the regular
                // phase came back to the same stall): a fresh start of the regular phase at this z with the barrier parameter it left with
                resto = false; mu = mu_reg;
                for (int i = lane; i < m; i += 64) { const double s0 = fmax(W.g[i], 1e-2); W.sl[i] = s0; W.lam[i] = mu * rcp_(s0); }
                for (int i = lane; i < n; i += 64) W.zb[i] = W.z[i];
                nu = 10.0; n_acc = 0; e_best = 1e300;
                SC_SYNC();
            }
        }
        double e_p, e_c0, lmax, e_d;
        bool stop = false, want_resto = false;
        for (int pass = 0;; ++pass) {
            // restoration: no objective but zeta/2 |z - z_R|^2, zeta = sqrt(mu); the elastic rows' entries depend on mu, so the
            // passes run again when the barrier update below has changed it
            const double sfe = resto ? 0.0 : sf, zeta = resto ? sqrt(mu) : 0.0;
            row_pass(W, c, lane, e_p, e_c0, lmax, resto, mu, rho_R);
            SC_SYNC();
            SC_PH(1);
            const double e_rho = stage_pass<KT, ROW16, OD, UNI>(W, c, lane, sfe);
            SC_SYNC();
            SC_PH(2);
            e_d = fmax(e_rho, col_pass<OD, UNI>(W, c, lane, sfe, zeta));
            SC_PH(3);
            if (pass == 1) break;
            const double e_opt = fmax(e_d, fmax(e_p, e_c0));
            if (!resto && e_opt < e_best) {                              // remember the best iterate
                e_best = e_opt;
                for (int i = lane; i < n; i += 64) { W.zb[i] = W.z[i]; if constexpr (OD) W.rhob[i] = W.rho[i]; }
            }
            if (resto) {
                // A stationary point of the violation.  The restoration's KKT error is in units of its objective rho theta, so |grad theta|
                // <= e_opt / rho; over the input box (a few units across) theta cannot fall by more than ~10 e_opt / rho from here: the
                // certificate asks for more violation than that.
                if (e_opt <= p.resto.tol && theta > fmax(p.resto.theta_tol, 10.0 * e_opt / rho_R)) { status = SC_STATUS_INFEASIBLE; stop = true; break; }
                if (e_opt <= p.tol) { stop = true; break; }                             // solved, and (nearly) no violation left: nothing to certify
                if (p.resto.stall_iter > 0) {
                    // no 1 % less violation within stall_iter iterations and violation left: a local minimiser of the violation at a kink
                    const bool less = theta <= 0.99 * theta_ref;                  // (selects, not branches: the values are wave-uniform)
                    theta_ref = less ? theta : theta_ref;
                    n_stall = less ? 0 : n_stall + 1;
                    if (n_stall >= p.resto.stall_iter) { if (theta > p.resto.stall_theta) status = SC_STATUS_INFEASIBLE; stop = true; break; }   // (less violation: SC_STATUS_INACCURATE)
                }
            } else if (e_opt <= p.tol) {
                status = SC_STATUS_OPTIMAL;
                stop = true; break;
            }
            // IPOPT's acceptable-point rule: acceptable_iter consecutive iterates within acceptable_tol end the solve (the
            // best iterate is returned below)
            n_acc = e_opt <= p.acceptable_tol ? n_acc + 1 : 0;
            if (n_acc >= acc_iter) {
                if (resto && theta > p.resto.theta_tol) status = SC_STATUS_INFEASIBLE;
                stop = true; break;
            }
            if (!resto && lmax > 1e10) {                                 // multipliers diverge: locally infeasible
                if constexpr (RESTO) want_resto = true;
                else { status = SC_STATUS_INFEASIBLE; stop = true; }
                break;
            }
            // barrier update
            const double mu_old = mu;
            for (;;) {
                double e_c = 0.0;
                for (int i = lane; i < m; i += 64) {
                    const double l = W.lam[i];
                    e_c = fmax(e_c, fabs(W.sl[i] * l - mu));
                    if (RESTO && resto && i < c.mc) e_c = fmax(e_c, fabs(W.tel[i] * (rho_R - l) - mu));
                }
                e_c = wmax(e_c);
                const double e_mu = fmax(e_d, fmax(e_p, e_c));
                if (e_mu <= 10.0 * mu && mu > p.mu_min) mu = fmax(p.mu_min, fmin(0.2 * mu, mu * sqrt(mu)));
                else break;
            }
            SC_PH(4);
            if (!(RESTO && resto && mu != mu_old)) break;
            SC_SYNC();
        }
        if (stop) break;
        bool accepted = false;
        double alpha = 0.0, ad = 0.0;
        if (!want_resto) {
        const double sfe = resto ? 0.0 : sf, zeta = resto ? sqrt(mu) : 0.0;
        phi_times_G<MS, UNI ? 2 : 4>(W, c, lane);
        SC_SYNC();
        SC_PH(5);
        // condensed system  (sf W + J' Sigma J) dz = -r_d + J' (mu/s - Sigma r_p - lam)
        for (int col = lane; col < n; col += 64) W.rhs[col] = -W.cv[col] + (mu * W.cv[n + col] - W.cv[2 * n + col]);
        condense_mfma<MS, OD, UNI>(W, c, lane, sfe);
        SC_SYNC();
        SC_PH(6);
        // inertia correction: M + delta I until the Cholesky succeeds (restoration: + zeta I, the proximity term)
        double delta = delta_force;                                      // 0 unless a failed restoration step is being retried
        bool ok = false;
        if constexpr (NT > 0) {
            constexpr int nn = NT > 0 ? 2 * NT : 2;
            // out of line (see chol_factor_solve): keeps the interior-point loop short enough for plain branches
            for (int t = 0; t < 40 && !ok; ++t) {
                ok = chol_factor_solve<nn>((int)(W.M - sm), (int)(W.rhs - sm), (int)(W.L - sm), (int)(W.dz - sm), delta + zeta, lane);
                if (!ok) delta = (delta == 0.0) ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
            }
            if (!ok) break;
            if (delta > 0.0) delta_last = delta;
            SC_PH(7);
        } else {
            for (int t = 0; t < 40 && !ok; ++t) {
                for (int r = 0; r < n; ++r)                  // lower triangle, odd row stride (mpc_ipm_common.hpp: no LDS bank conflicts)
                    for (int cc = lane; cc <= r; cc += 64) W.L[r * (n | 1) + cc] = W.M[r * n + cc] + (cc == r ? delta + zeta : 0.0);
                SC_SYNC();
                ok = ipm::cholesky_lds(W.L, n, n | 1, lane);
                if (!ok) delta = (delta == 0.0) ? fmax(1e-4, delta_last / 3.0) : delta * 8.0;
            }
            if (!ok) break;
            if (delta > 0.0) delta_last = delta;
            SC_PH(7);
            ipm::chol_solve_lds(W.L, W.dz, n, n | 1, lane);  // rhs is dz: solved in place
        }
        SC_PH(8);
        // position and speed displacements  dp = G dz,  dV_k = dt sum_{j < k} dz_{2j}
        for (int row = lane; row < 2 * (N + 2); row += 64) {
            double acc = 0.0;                                             // p_k depends on stages j <= k - 2 (zeros stored)
            if (row >= (UNI ? 2 : 4)) {                                   // rows of p_0 (and p_1): zero, their storage may hold W.tel
#pragma unroll
                for (int col = 0; col < n; ++col) acc += W.dP[(size_t)row * n + col] * W.dz[col];
            }
            W.dp[row] = acc;
        }
        for (int k = lane; k <= N; k += 64) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) { const double dzj = W.dz[2 * j]; acc += j < k ? dzj : 0.0; }
            W.dV[k] = c.dt * acc;
        }
        double gdz = 0.0, prox = 0.0;
        for (int i = lane; i < n; i += 64) {
            gdz += W.cv[i] * W.dz[i];                                     // r_d . dz
            if (RESTO && resto) { const double dzr = W.z[i] - W.zb[i]; prox += dzr * dzr; }
        }
        SC_SYNC();
        if constexpr (OD) {
            // back-substitution of the decay variables:  d rho_k = D_k^-1 (rhs_k - C_k' dp_{k..k+2}),  rhs = -t0 + mu t1 - t2
            for (int k = lane; k < N; k += 64) {
                const double* od_ = W.ods + (size_t)22 * k;
                double v0 = -od_[15] + mu * od_[17] - od_[19], v1 = -od_[16] + mu * od_[18] - od_[20];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const double dpr = W.dp[2 * k + r];
                    v0 -= od_[2 * r] * dpr; v1 -= od_[2 * r + 1] * dpr;
                }
                const double vx = od_[12], vy = od_[13];
                const double ps_ = (vx * v0 + vy * v1) * od_[14], pw_ = (-vy * v0 + vx * v1) * od_[21];
                const double dr1 = vx * ps_ - vy * pw_, dr2 = vy * ps_ + vx * pw_;
                W.drho[2 * k] = dr1; W.drho[2 * k + 1] = dr2;
                gdz += od_[15] * dr1 + od_[16] * dr2;                     // r_d of the decay variables . d rho
            }
            SC_SYNC();
        }
        SC_PH(9);
        // ds = J dz + r_p, dlam, step lengths.  Elastic rows of the restoration: dlam = -Sigma_eff J dz + dl0 with both taken from
        // the row pass's arrays (lam * is, mu * is - vb), dt from dlam, ds = J dz + dt + r_p; t and rho - lam join the ratios
        double rs_min = 0.0, rl_min = 0.0, sum_ds_s = 0.0, sum_rp = 0.0, sum_t = 0.0, sum_dt = 0.0;
        LogSum ls0;
        for (int i = lane; i < m; i += 64) {
            const double s = W.sl[i], lam = W.lam[i], isv = W.is[i];
            double jd;
            if (i < c.mc) {
                const int k = i / K, jo = i - k * K;
                const int e0 = k * K + jo, e1 = e0 + K, e2 = e1 + K;
                const double w0r = OD ? W.w0s[k] : c.w0, w1r = OD ? W.w1s[k] : c.w1;
                jd = w0r * (W.dh[2 * e0] * W.dp[2 * k] + W.dh[2 * e0 + 1] * W.dp[2 * k + 1]) +
                     w1r * (W.dh[2 * e1] * W.dp[2 * k + 2] + W.dh[2 * e1 + 1] * W.dp[2 * k + 3]) +
                     c.w2 * (W.dh[2 * e2] * W.dp[2 * k + 4] + W.dh[2 * e2 + 1] * W.dp[2 * k + 5]);
                if constexpr (OD) jd += W.A1[i] * W.drho[2 * k] + W.A2[i] * W.drho[2 * k + 1];
            } else if (i < c.mc + c.ns) {
                const int r = i - c.mc, k = (r >> 1) + 1;
                jd = (r & 1) ? W.dV[k] : -W.dV[k];
            } else {
                const int r = i - c.mc - c.ns;
                jd = r < n ? -W.dz[r] : W.dz[r - n];
            }
            double ds, dl, rp, rs;
            if (RESTO && resto && i < c.mc) {
                const double t = W.tel[i], nut = rho_R - lam;
                rp = W.g[i] + t - s;
                dl = -(lam * isv) * jd + (mu * isv - W.dlam[i]);            // W.dlam still holds vb of this row
                const double dt = ipm::resto_dt(lam, t, dl, mu, rho_R);
                ds = jd + dt + rp;
                rs = ds * rcp_(s);
                const double rt = dt * rcp_(t);
                rs_min = fmin(rs_min, rt); rl_min = fmin(rl_min, -dl * rcp_(nut));
                sum_ds_s += rt; sum_t += t; sum_dt += dt; ls0.add(t);
            } else {
                rp = W.g[i] - s;
                ds = jd + rp;
                dl = -(lam * isv) * ds - (lam - mu * isv);
                rs = ds * isv;
            }
            gdz += lam * jd;                                              // sf grad f . dz = r_d . dz + lam . (J dz)
            const double rl = dl * rcp_(lam);                             // fraction to the boundary: most negative ratios
            rs_min = fmin(rs_min, rs); rl_min = fmin(rl_min, rl);
            sum_ds_s += rs; sum_rp += fabs(rp); ls0.add(s);
            W.ds[i] = ds; W.dlam[i] = dl;
        }
        rs_min = wmin(rs_min); rl_min = wmin(rl_min); sum_ds_s = wsum(sum_ds_s); sum_rp = wsum(sum_rp);
        const double sum_log = ls0.total();
        const double ap = rs_min < 0.0 ? fmin(1.0, -tau / rs_min) : 1.0;
        ad = rl_min < 0.0 ? fmin(1.0, -tau / rl_min) : 1.0;
        gdz = wsum(gdz);
        nu = fmax(nu, 1.1 * lmax);
        double bar0 = sfe * f - mu * sum_log, dbar = gdz - mu * sum_ds_s;
        if (RESTO && resto) {
            prox = wsum(prox); sum_t = wsum(sum_t); sum_dt = wsum(sum_dt);
            bar0 = 0.5 * zeta * prox + rho_R * sum_t - mu * sum_log;
            dbar += rho_R * sum_dt;
        }
        // not a descent direction of the merit function (the penalty is below the multipliers of the step): raise the penalty so that
        // the directional derivative is -0.1 nu |r_p|_1  (Nocedal & Wright (18.36))
        if (dbar - nu * sum_rp >= 0.0 && sum_rp > 0.0) nu = dbar / (0.9 * sum_rp);
        const double phi0 = bar0 + nu * sum_rp;
        const double dphi = dbar - nu * sum_rp;
        SC_PH(10);
        // l1-merit backtracking
        alpha = ap;
        // slack reset of the line search: restoration sc_resto_params.slack_reset, otherwise sc_mpccbf_params.slack_reset = 2
        const bool rreset = (RESTO && resto) ? p.resto.slack_reset != 0 : p.slack_reset == 2;
        const double thr_reset = mu * rcp_(nu);
        for (int ls = 0; ls < 12; ++ls) {                              // at most 12 halvings, then give up (best iterate)
            for (int i = lane; i < n; i += 64) {
                W.zt[i] = W.z[i] + alpha * W.dz[i];
                if constexpr (OD) W.rhot[i] = W.rho[i] + alpha * W.drho[i];
            }
            SC_SYNC();
            const double ft = eval_values<ROW16, OD, UNI>(W.zt, W.rhot, W, c, lane, false, pf, 16);
            double srp = 0.0, st_ = 0.0, proxt = 0.0;
            LogSum lst;
            for (int i = lane; i < m; i += 64) {
                const double s_lin = W.sl[i] + alpha * W.ds[i];
                double tot = W.g[i];
                if (RESTO && resto && i < c.mc) {
                    const double t = W.tel[i];
                    const double tt = t + alpha * ipm::resto_dt(W.lam[i], t, W.dlam[i], mu, rho_R);
                    lst.add(tt); st_ += tt;
                    tot += tt;
                }
                // slack reset (t = 0 outside the restoration): s = g + t where that is >= mu / nu
                const double st = (rreset && tot >= thr_reset) ? tot : s_lin;
                lst.add(st);
                srp += fabs(tot - st);
            }
            if (RESTO && resto) {
                for (int i = lane; i < n; i += 64) { const double dzr = W.zt[i] - W.zb[i]; proxt += dzr * dzr; }
            }
            const double slog = lst.total();
            srp = wsum(srp);
            double phit = sfe * ft - mu * slog + nu * srp;
            if (RESTO && resto) phit = 0.5 * zeta * wsum(proxt) + rho_R * wsum(st_) - mu * slog + nu * srp;
            // Armijo, with an allowance for round-off in the merit function near convergence
            // (f is a sum of a few hundred terms of size |phi|: its noise is ~1e-13 |phi|)
            if (phit <= phi0 + 1e-4 * alpha * dphi + 1e-13 * fabs(phi0)) { accepted = true; break; }
            alpha *= 0.5;
        }
        if (!accepted) {
            if (!RESTO) break;
            if (resto) {
                if (n_retry >= p.resto.retry_max) break;
                // the same z again, Levenberg-damped; the retry is an iteration of its own (W.g holds the last trial point's rows)
                ++n_retry; delta_force = fmax(1.0, 100.0 * fmax(delta_force, delta));
                SC_SYNC();
                eval_values<ROW16, OD, UNI>(W.z, W.rho, W, c, lane, false, pf, 16);
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
            eval_values<ROW16, OD, UNI>(W.z, W.rho, W, c, lane, false, pf, 16);   // W.g holds the last trial point's rows
            resto = true; ++n_resto; n_small = 0; theta_R = theta; mu_reg = mu;
            delta_force = 0.0; n_retry = 0; theta_ref = theta; n_stall = 0;
            double vmax = 0.0;
            for (int i = lane; i < c.mc; i += 64) vmax = fmax(vmax, -W.g[i]);
            mu = fmax(mu, wmax(vmax));                                   // IPOPT: mu_R = max(mu, |c|_inf)
            for (int i = lane; i < m; i += 64) {
                // elastic rows start on their central path, the others like at the start of the solve
                const double gi = W.g[i];
                const double s0 = i < c.mc ? ipm::resto_central_slack(gi, mu, rho_R) : fmax(gi, 1e-2);
                if (i < c.mc) W.tel[i] = s0 - gi;
                W.sl[i] = s0; W.lam[i] = mu * rcp_(s0);
            }
            for (int i = lane; i < n; i += 64) W.zb[i] = W.z[i];          // z_R
            nu = 10.0; n_acc = 0;
            SC_SYNC();
            continue;
        }
        for (int i = lane; i < n; i += 64) {
            W.z[i] = W.z[i] + alpha * W.dz[i];
            if constexpr (OD) W.rho[i] = W.rho[i] + alpha * W.drho[i];
        }
        delta_force = 0.0; n_retry = 0;
        {
        const bool rreset = (RESTO && resto) ? p.resto.slack_reset != 0 : p.slack_reset == 2;   // W.g holds the accepted trial point's rows
        const double thr_reset = mu * rcp_(nu);
        for (int i = lane; i < m; i += 64) {
            const double s_lin = W.sl[i] + alpha * W.ds[i];
            const double l0 = W.lam[i], dl = W.dlam[i];
            double tn = 0.0;
            const bool el = RESTO && resto && i < c.mc;
            if (el) { const double t = W.tel[i]; tn = t + alpha * ipm::resto_dt(l0, t, dl, mu, rho_R); W.tel[i] = tn; }
            const double tot = W.g[i] + tn;
            const double s = (rreset && tot >= thr_reset) ? tot : s_lin;
            double lam = l0 + ad * dl;
            const double mus = mu * rcp_(s);
            lam = fmin(fmax(lam, 1e-10 * mus), 1e10 * mus);               // IPOPT eq. (16) safeguard
            if (el) lam = ipm::resto_clamp_lam(lam, tn, mu, rho_R);
            W.sl[i] = s; W.lam[i] = lam;
        }
        }
        SC_SYNC();
        SC_PH(11);
    }
    if (pending) {
        // hand-over (mpc_cont.hpp).  W.g holds the rows of the current z on every path to the top of the loop (the accepted trial
        // point, the evaluation before a restoration entry, the initial evaluation)
        SC_SYNC();
        double th = 0.0;
        for (int i = lane; i < c.mc; i += 64) th += fmax(0.0, -W.g[i]);
        th = wsum(th);
        double* a = cst + ipm::CONT_SCALARS;
        ipm::cont_copy(a, W.z, n, lane, 64); a += n;
        ipm::cont_copy(a, W.zb, n, lane, 64); a += n;
        ipm::cont_copy(a, W.sl, m, lane, 64); a += m;
        ipm::cont_copy(a, W.lam, m, lane, 64); a += m;
        ipm::cont_copy(a, W.obs, K * 7, lane, 64); a += K * 7;
        if (nel) { ipm::cont_copy(a, W.tel, nel, lane, 64); a += nel; }
        if constexpr (OD) { ipm::cont_copy(a, W.rho, n, lane, 64); a += n; ipm::cont_copy(a, W.rhob, n, lane, 64); }
        if (lane == 0) {
            cst[0] = (double)(it - 1); cst[1] = mu; cst[2] = nu; cst[3] = delta_last; cst[4] = e_best; cst[5] = (double)n_acc;
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
        // stalled at the precision limit (ill-conditioned condensed system at mu ~ 1e-9): the best iterate is
        // within the acceptable tolerance, like IPOPT's acceptable_tol exit
        SC_SYNC();
        for (int i = lane; i < n; i += 64) { W.z[i] = W.zb[i]; if constexpr (OD) W.rho[i] = W.rhob[i]; }
        status = SC_STATUS_OPTIMAL;
    }
    SC_SYNC();
    if constexpr (OD) {
        // optimal decay has no restoration phase: "infeasible" there still means "stopped at an infeasible iterate"
        eval_values<ROW16, OD, UNI>(W.z, W.rho, W, c, lane, false, pf, 16);
        if (status != SC_STATUS_OPTIMAL) {
            double gmin = 1e300;
            for (int i = lane; i < m; i += 64) gmin = fmin(gmin, W.g[i]);
            gmin = wmin(gmin);
            if (gmin < -1e-6) status = SC_STATUS_INFEASIBLE;
        }
    }
    if (lane == 0) {
        st(u_out, prob * 2 + 0, W.z[0]);
        st(u_out, prob * 2 + 1, W.z[1]);
        status_out[prob] = status;
        if (iters_out) iters_out[prob] = it;
    }
#ifdef SC_MPC_PROF
    if (z_out && lane < 20 && lane < n) st(z_out, prob * n + lane, pf.ph[lane]);
#elif defined(SC_EXP_TRACE)
#else
    if (z_out) for (int i = lane; i < n; i += 64) st(z_out, prob * n + i, W.z[i]);
#endif
    if constexpr (OD) {
        if (rho_out) for (int i = lane; i < n; i += 64) st(rho_out, prob * n + i, W.rho[i]);
    }
}

// Compile-time horizon N <= 10: capped at 256 VGPRs (two waves per SIMD; fits without spills).  N = 20 keeps ~59 KiB of
// LDS per problem, so at most two problems share a CU (half a wave per SIMD): there the cap bought nothing but 70-90
// spilled VGPRs on the dependency chain, and the kernels take the whole register file instead.  Run-time sizes: no cap --
// the run-time index arithmetic needs more registers and would spill heavily under it.
#define SC_MPC_WAVES(NT) __attribute__((amdgpu_waves_per_eu((NT) <= 10 ? 2 : 1, (NT) <= 10 ? 2 : 1)))
template <int NT, int KT>
__global__ __launch_bounds__(64) SC_MPC_WAVES(NT)
void mpccbf_kernel(const sc_mpccbf_params p, const long long B, const int K_rt, const void* __restrict__ X,
                   const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                   void* __restrict__ u_out, int* __restrict__ status_out, int* __restrict__ iters_out, void* __restrict__ z_out,
                   const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<NT, KT>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct);
}
__global__ __launch_bounds__(64)
void mpccbf_kernel_rt(const sc_mpccbf_params p, const long long B, const int K_rt, const void* __restrict__ X,
                      const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                      void* __restrict__ u_out, int* __restrict__ status_out, int* __restrict__ iters_out, void* __restrict__ z_out,
                   const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<0, 0>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct);
}

// kinematic Unicycle2D (robots/unicycle2D.py through position_control/mpc_cbf.py), K run-time
template <int NT>
__global__ __launch_bounds__(64) SC_MPC_WAVES(NT)
void mpccbf_uni_kernel(const sc_mpccbf_params p, const long long B, const int K_rt, const void* __restrict__ X,
                       const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                       void* __restrict__ u_out, int* __restrict__ status_out, int* __restrict__ iters_out, void* __restrict__ z_out,
                   const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<NT, 0, false, true>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct);
}
__global__ __launch_bounds__(64)
void mpccbf_uni_kernel_rt(const sc_mpccbf_params p, const long long B, const int K_rt, const void* __restrict__ X,
                          const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                          void* __restrict__ u_out, int* __restrict__ status_out, int* __restrict__ iters_out, void* __restrict__ z_out,
                   const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<0, 0, false, true>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct);
}

// optimal-decay variant (position_control/optimal_decay_mpc_cbf.py), K run-time.  Compile-time horizon: capped at 256
// VGPRs (about 60 spilled registers, still 23 % faster at large batches: 6 problems per CU instead of 4); run-time
// horizon: no cap.
template <int NT>
__global__ __launch_bounds__(64) SC_MPC_WAVES(NT)
void odmpccbf_kernel(const sc_mpccbf_params p, const OdExtra od, const long long B, const int K_rt, const void* __restrict__ X,
                     const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                     void* __restrict__ u_out, void* __restrict__ rho_out, int* __restrict__ status_out,
                     int* __restrict__ iters_out, void* __restrict__ z_out, const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<NT, 0, true>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct, od, rho_out);
}
__global__ __launch_bounds__(64)
void odmpccbf_kernel_rt(const sc_mpccbf_params p, const OdExtra od, const long long B, const int K_rt, const void* __restrict__ X,
                        const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                        void* __restrict__ u_out, void* __restrict__ rho_out, int* __restrict__ status_out,
                        int* __restrict__ iters_out, void* __restrict__ z_out, const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<0, 0, true>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct, od, rho_out);
}

// optimal decay on the kinematic Unicycle2D: BASELINE config 5's extension (oracle/od_mpc_rd1.py)
template <int NT>
__global__ __launch_bounds__(64) SC_MPC_WAVES(NT)
void odmpccbf_uni_kernel(const sc_mpccbf_params p, const OdExtra od, const long long B, const int K_rt, const void* __restrict__ X,
                         const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                         void* __restrict__ u_out, void* __restrict__ rho_out, int* __restrict__ status_out,
                         int* __restrict__ iters_out, void* __restrict__ z_out, const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<NT, 0, true, true>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct, od, rho_out);
}
__global__ __launch_bounds__(64)
void odmpccbf_uni_kernel_rt(const sc_mpccbf_params p, const OdExtra od, const long long B, const int K_rt, const void* __restrict__ X,
                            const void* __restrict__ u_prev, const void* __restrict__ goal, const void* __restrict__ obs,
                            void* __restrict__ u_out, void* __restrict__ rho_out, int* __restrict__ status_out,
                            int* __restrict__ iters_out, void* __restrict__ z_out, const ipm::Cont ct) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    mpccbf_body<0, 0, true, true>(sm, p, B, K_rt, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, ct, od, rho_out);
}

size_t mpccbf_lds_bytes(int N, int K, bool uni) { return mpc_lds_doubles(N, K, false, uni) * sizeof(double); }
size_t odmpccbf_lds_bytes(int N, int K) { return mpc_lds_doubles(N, K, true) * sizeof(double); }

static hipError_t odmpc_launch_t(const sc_odmpccbf_params& q, long long B, int K, const void* X, const void* u_prev,
                                 const void* goal, const void* obs, void* u_out, void* rho_out, int* status, int* iters,
                                 void* z_out, hipStream_t stream, const ipm::Cont& ct) {
    const sc_mpccbf_params& p = q.mpc;
    const size_t lds = mpc_lds_doubles(p.horizon, K, true) * sizeof(double);
    OdExtra od;
    od.omega_ref[0] = q.omega_ref[0]; od.omega_ref[1] = q.omega_ref[1]; od.p_sb[0] = q.p_sb[0]; od.p_sb[1] = q.p_sb[1];
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64), lds, stream, p, od, B, K, X, u_prev,
                           goal, obs, u_out, rho_out, status, iters, z_out, ct);
        return hipGetLastError();
    };
    if (p.model_id == SC_MODEL_UNICYCLE2D) {
        if (p.horizon == 10) return launch(odmpccbf_uni_kernel<10>);
        if (p.horizon == 20) return launch(odmpccbf_uni_kernel<20>);  // BASELINE config 5's horizon
        return launch(odmpccbf_uni_kernel_rt);
    }
    if (p.horizon == 10) return launch(odmpccbf_kernel<10>);
    if (p.horizon == 20) return launch(odmpccbf_kernel<20>);         // BASELINE config 5's horizon
    return launch(odmpccbf_kernel_rt);
}

hipError_t odmpccbf_launch(const sc_odmpccbf_params& q, long long B, int K, const void* X, const void* u_prev,
                           const void* goal, const void* obs, void* u_out, void* rho_out, int* status, int* iters,
                           void* z_out, hipStream_t stream, const ipm::Cont& ct) {
    if (mpc_lds_doubles(q.mpc.horizon, K, true) * sizeof(double) > 160 * 1024) return hipErrorInvalidValue;
    return odmpc_launch_t(q, B, K, X, u_prev, goal, obs, u_out, rho_out, status, iters, z_out, stream, ct);
}

template <int NT, int KT, bool UNI = false>
static hipError_t mpc_launch_one(const sc_mpccbf_params& p, long long B, int K, const void* X, const void* u_prev,
                                 const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out,
                                 hipStream_t stream, const ipm::Cont& ct) {
    const size_t lds = mpc_lds_doubles(p.horizon, K, false, UNI) * sizeof(double);
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(64), lds, stream, p, B, K, X, u_prev,
                           goal, obs, u_out, status, iters, z_out, ct);
        return hipGetLastError();
    };
    if constexpr (UNI && NT > 0) return launch(mpccbf_uni_kernel<NT>);
    else if constexpr (UNI) return launch(mpccbf_uni_kernel_rt);
    else if constexpr (NT > 0) return launch(mpccbf_kernel<NT, KT>);
    else return launch(mpccbf_kernel_rt);
}

static hipError_t mpc_launch_t(const sc_mpccbf_params& p, long long B, int K, const void* X, const void* u_prev,
                               const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out,
                               hipStream_t stream, const ipm::Cont& ct) {
    if (p.model_id == SC_MODEL_UNICYCLE2D) {
        if (p.horizon == 10) return mpc_launch_one<10, 0, true>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
        if (p.horizon == 20) return mpc_launch_one<20, 0, true>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
        return mpc_launch_one<0, 0, true>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
    }
    if (p.horizon == 10 && K == 8)            // BASELINE config 3
        return mpc_launch_one<10, 8>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
    if (p.horizon == 10)                      // the reference's default horizon (mpc_cbf.py:15) with any obstacle count
        return mpc_launch_one<10, 0>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
    if (p.horizon == 20)                      // BASELINE config 5's horizon
        return mpc_launch_one<20, 0>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
    return mpc_launch_one<0, 0>(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
}

// doubles of one problem's solver state in a continuation workspace (mpc_cont.hpp; the layout of mpccbf_body's hand-over)
size_t mpccbf_state_doubles(int N, int K, bool od) {
    const size_t n = 2 * (size_t)N, m = (size_t)N * K + 2 * N + 2 * n;
    return ipm::CONT_SCALARS + 2 * n + 2 * m + 7 * (size_t)K + (od ? 2 * n : (size_t)N * K);
}

hipError_t mpccbf_launch(const sc_mpccbf_params& p, long long B, int K, const void* X, const void* u_prev,
                         const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out,
                         hipStream_t stream, const ipm::Cont& ct) {
    if (mpc_lds_doubles(p.horizon, K, false, p.model_id == SC_MODEL_UNICYCLE2D) * sizeof(double) > 160 * 1024) return hipErrorInvalidValue;
    return mpc_launch_t(p, B, K, X, u_prev, goal, obs, u_out, status, iters, z_out, stream, ct);
}

}  // namespace sc
