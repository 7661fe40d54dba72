// Batched Backup-CBF QP for gfx950: rollout of the backup controller, finite-difference sensitivities,
// one CBF row per backup step plus the terminal row, and the exact 2-variable QP -- per agent, four lanes per agent.
//
// Replaces, for B agents per launch, the per-robot path
//   BackupCBF.solve_control_problem          position_control/backup_cbf_qp.py:563-794
//   BackupCBF._integrate_backup_trajectory   :236-320   (robot.step + forward differences, eps = 1e-5)
//   BackupCBF._h_safety / _h_terminal (+ FD gradients)  :343-561
//   cvxpy -> OSQP                            :717-726   (here: the exact minimiser)
// on the scenario the reference ships for it (examples/evade/test_evade.py --algo backupcbf): DoubleIntegrator2D
// (robots/double_integrator2D.py:46-107), EvadeBackupController (position_control/backup_controller.py:456-571), EvadeEnv
// (envs/evade_env.py) with its constant-speed bullet.  oracle/backup_cbf.py is the float64 statement of the same
// computation (pinned bit for bit on the reference's own run, tests/golden/backup_cbf.npz).
//
// Mapping: FOUR lanes per agent (a DPP quad), 16 agents per wave, one wave per workgroup.  The rollout is sequential in
// the backup step but every step needs the base successor and the four perturbed successors (forward differences): lane q
// of the quad evaluates the base and perturbation q, so the quad covers them in two evaluations instead of five.  Lane q
// also keeps COLUMN q of the sensitivity S_i; the four columns of A_i = d step / d x reach every lane with quad_perm DPP
// broadcasts (VALU latency, no LDS).  The same split serves the rows: lane q differences h along x_q, the gradient is
// broadcast, lane q forms (grad . S)_q.  Rows go to LDS (N x 3 doubles per agent) and the quad then walks the QP
// cooperatively: rows dealt round-robin over the four lanes, the most violated row not yet in the working set joins it
// (Seidel's incremental step in adaptive order: the optimum over the set plus that row lies on its line, clipped by the
// box and by the members of the set), interval ends combined with two DPP steps.
//
// Everything a forward difference amplifies by 1 / eps = 1e5 (step, backup controller, h) is written operation for
// operation like the reference's scalar numpy code and compiled WITHOUT floating-point contraction, so those values agree
// with the CPU to the last bit or two instead of to 1e-16 / 1e-5.
#include <hip/hip_runtime.h>

#include "sc_math.hpp"
#include "sc_qp2.hpp"
#include "../../include/safe_control_amd.h"

#pragma clang fp contract(off)

namespace sc {
namespace {

struct BkP {                      // constants of one launch, in registers / SGPRs
    double dt, T, eps, R, amax, vmax, sm, alpha, alphaT, kp, kd;
    double L, hw, pxmin, pxmax, pymin, pymax, gxmin, gxmax, bspeed, blen, bhalf_len, bhalf_wid, bshift, bstart;
    double cx, cy;
};

template <int J>
__device__ __forceinline__ double quad_bcast(double v) {               // value of lane J of the quad, in every lane
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), J * 0x55, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), J * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double quad_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int quad_move(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ double quad_max(double v) { v = fmax(v, quad_move<0xB1>(v)); return fmax(v, quad_move<0x4E>(v)); }
__device__ __forceinline__ double quad_min(double v) { v = fmin(v, quad_move<0xB1>(v)); return fmin(v, quad_move<0x4E>(v)); }

// _clamp_control (backup_controller.py:551-557) / the example's nominal clamp (test_evade.py:160-164)
__device__ __forceinline__ void bk_clamp(double& ax, double& ay, double amax) {
    const double am = sqrt(ax * ax + ay * ay);
    if (am > amax) { ax = ax * amax / am; ay = ay * amax / am; }
}

// EvadeNominalController.compute_control (examples/evade/test_evade.py:141-166)
__device__ __forceinline__ void bk_nominal(const double* x, const BkP& P, double& ax, double& ay) {
    ax = 2.0 * (P.vmax - x[2]);
    ay = 2.0 * (0.0 - x[1]) + 2.0 * (0.0 - x[3]);
    bk_clamp(ax, ay, P.amax);
}

// EvadeBackupController.compute_control (backup_controller.py:456-549)
__device__ __forceinline__ void bk_backup(const double* x, const BkP& P, double& ax, double& ay) {
    const double px = x[0], py = x[1], vx = x[2], vy = x[3];
    const double margin = P.R + 0.1;
    const double ddx = px - P.cx, ddy = py - P.cy;
    const double dist = sqrt(ddx * ddx + ddy * ddy);
    const bool in_goal = P.gxmin <= px && px <= P.gxmax && -P.hw <= py && py <= P.hw;
    const bool x_safe = P.pxmin + margin <= px && px <= P.pxmax - margin;
    const bool deep = x_safe && P.pymin + margin <= py && py <= P.pymax - margin && dist < 1.0;
    if (in_goal || deep) {
        ax = -P.kd * vx; ay = -P.kd * vy;
    } else if (P.pxmin - 2.0 <= px && px <= P.pxmax + 2.0) {
        const double ty = x_safe ? P.cy : (py > P.pymin ? fmax(py, 3.0) : 0.0);
        ax = P.kp * (P.cx - px) - P.kd * vx;
        ay = P.kp * (ty - py) - P.kd * vy;
    } else {
        const double ty = (py > P.pymin && px > P.pxmax) ? fmax(py, 3.0) : 0.0;
        const double ex = P.cx - px, ey = ty - py;
        const double sg = ex > 0.0 ? 1.0 : (ex < 0.0 ? -1.0 : 0.0);
        ax = P.kp * sg * fmin(fabs(ex), 3.0) - P.kd * vx;
        ay = P.kp * ey - P.kd * vy;
    }
    bk_clamp(ax, ay, P.amax);
}

// DoubleIntegrator2D.step (double_integrator2D.py:79-107): Euler, then the speed rescaled to v_max
__device__ __forceinline__ void bk_step(const double* x, double ax, double ay, const BkP& P, double* xn) {
    xn[0] = x[0] + x[2] * P.dt; xn[1] = x[1] + x[3] * P.dt;
    xn[2] = x[2] + ax * P.dt; xn[3] = x[3] + ay * P.dt;
    const double vm = sqrt(xn[2] * xn[2] + xn[3] * xn[3]);
    if (vm > P.vmax) { const double s = P.vmax / vm; xn[2] *= s; xn[3] *= s; }
}
__device__ __forceinline__ void bk_closed_step(const double* x, const BkP& P, double* xn) {      // step(x, backup(x))
    double ax, ay;
    bk_backup(x, P, ax, ay);
    bk_step(x, ax, ay, P, xn);
}

// BackupCBF._h_safety on EvadeEnv (backup_cbf_qp.py:359-392) with the rectangular bullet at time t (:403-430;
// envs/evade_env.py:386-406 and test_evade.py:373-384 for its predicted box)
__device__ __forceinline__ double bk_h_safety(double px, double py, double t, double bx, const BkP& P) {
    double h = py + P.hw - P.R;
    h = fmin(h, px - P.R);
    h = fmin(h, P.L - px - P.R);
    if (P.pxmin <= px && px <= P.pxmax) {
        h = fmin(h, P.pymax - py - P.R);
        if (py > P.hw) h = fmin(h, fmin(px - P.pxmin - P.R, P.pxmax - px - P.R));
    } else {
        h = fmin(h, P.hw - py - P.R);
    }
    const double ox = bx + P.bshift + P.bspeed * t;
    const double dx = fmax(fabs(px - ox) - P.bhalf_len, 0.0), dy = fmax(fabs(py - 0.0) - P.bhalf_wid, 0.0);
    return fmin(h, sqrt(dx * dx + dy * dy) - P.R - P.sm);
}

// BackupCBF._h_terminal (:481-494 pocket box, :524-535 speed, :537-541 safety at the end of the horizon)
__device__ __forceinline__ double bk_h_terminal(const double* x, double bx, const BkP& P) {
    const double m = P.R + 0.2;
    double h = fmin(fmin(x[0] - P.pxmin - m, P.pxmax - x[0] - m), fmin(x[1] - P.pymin - m, P.pymax - x[1] - m));
    h = fmin(h, P.vmax - sqrt(x[2] * x[2] + x[3] * x[3]));
    return fmin(h, bk_h_safety(x[0], x[1], P.T, bx, P));
}

__device__ __forceinline__ BkP make_bkp(const sc_backupcbf_params& p) {
    BkP P;
    P.dt = p.dt; P.T = p.backup_horizon; P.eps = p.fd_eps; P.R = p.robot_radius; P.amax = p.a_max; P.vmax = p.v_max;
    P.sm = p.safety_margin; P.alpha = p.alpha; P.alphaT = p.alpha_terminal; P.kp = p.backup_kp; P.kd = p.backup_kd;
    P.L = p.hallway_length; P.hw = p.half_width; P.pxmin = p.pocket_x_min; P.pxmax = p.pocket_x_max;
    P.pymin = p.pocket_y_min; P.pymax = p.pocket_y_max; P.gxmin = p.goal_x_min; P.gxmax = p.goal_x_max;
    P.bspeed = p.bullet_speed; P.blen = p.bullet_length; P.bstart = p.bullet_start_x;
    P.bshift = p.bullet_length / 6;                                       // evade_env.py:396
    P.bhalf_len = p.bullet_length * (1 + 1.0 / 3) / 2;                    // :395, backup_cbf_qp.py:424
    P.bhalf_wid = p.bullet_width / 2;
    P.cx = (p.pocket_x_min + p.pocket_x_max) / 2; P.cy = (p.pocket_y_min + p.pocket_y_max) / 2;   // evade_env.py:70-73
    return P;
}

// One agent per quad.  n_ctrl control steps per launch; advance != 0 steps the state, the bullet and the outcome code
// like the example's loop (test_evade.py:425-500), advance == 0 (with n_ctrl = 1) is the plain batched solve.
__global__ __launch_bounds__(64) void backupcbf_kernel(const sc_backupcbf_params p, const long long B, const int n_ctrl,
                                                       const int advance, void* __restrict__ X, const void* __restrict__ u_nom,
                                                       void* __restrict__ bullet_x, void* __restrict__ u_out,
                                                       int* __restrict__ status_out, int* __restrict__ using_backup_out,
                                                       void* __restrict__ h_min_out, int* __restrict__ n_rows_out,
                                                       double* __restrict__ rows_out, int* __restrict__ ret,
                                                       int* __restrict__ ret_step, const int step0) {
    extern __shared__ __attribute__((aligned(16))) double sm_rows[];      // [16 agents][N][3]: unit normal, offset
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };
    const int lane = threadIdx.x, q = lane & 3, slot = lane >> 2;
    const long long agent = (long long)blockIdx.x * 16 + slot;
    const bool active = agent < B;
    const long long ag = active ? agent : 0;
    const BkP P = make_bkp(p);
    const int N = p.n_steps;
    double* rows = sm_rows + (size_t)slot * N * 3;

    double xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xs[j] = ld(X, ag * 4 + j);
    double bx = ld(bullet_x, p.bullet_shared ? 0 : ag);
    int rcode = ret ? ret[ag] : 0, rstep = ret_step ? ret_step[ag] : -1;
    double uo0 = 0.0, uo1 = 0.0, h_min = 0.0;
    int qp_status = -1, using_backup = 0, n_rows = 0;

    for (int cs = 0; cs < n_ctrl; ++cs) {
        // ---- reference input: the caller's nominal control, or the example's nominal controller ----------------------
        double un0, un1;
        if (u_nom) { un0 = ld(u_nom, ag * 2); un1 = ld(u_nom, ag * 2 + 1); }
        else bk_nominal(xs, P, un0, un1);

        // ---- rollout, sensitivities and rows, fused (one pass over the backup horizon) ---------------------------------
        double x[4] = {xs[0], xs[1], xs[2], xs[3]};
        double Sc[4] = {q == 0 ? 1.0 : 0.0, q == 1 ? 1.0 : 0.0, q == 2 ? 1.0 : 0.0, q == 3 ? 1.0 : 0.0};   // column q of S_0 = I
        const double f0x = xs[2], f0y = xs[3];                            // f(x_0) = (vx, vy, 0, 0); g(x_0) selects the last two entries
        h_min = bk_h_safety(x[0], x[1], 0.0, bx, P);
        n_rows = 0;
        bool bad = false;
        // row i waits for phi_{i+1}: its pieces are kept one iteration
        double pg[4] = {0, 0, 0, 0}, pa = 0.0, pdh = 0.0, pah = 0.0, pl0 = 0.0, pl1 = 0.0;
        bool pending = false;
        auto emit = [&](double l0, double l1, double rhs) {
            // keep rule |lhs| > 1e-6 (:663-665); rows enter the QP in scaled variables u = a_max us (:684-716)
            const bool keep = sqrt(l0 * l0 + l1 * l1) > 1e-6;
            bad = bad || !(l0 == l0) || !(l1 == l1) || !(rhs == rhs);
            if (keep) {
                const double n0 = l0 * P.amax, n1 = l1 * P.amax;
                if (q == 0) {
                    if (rows_out && active) {
                        double* ro = rows_out + ((size_t)agent * N + n_rows) * 3;
                        ro[0] = n0; ro[1] = n1; ro[2] = rhs;
                    }
                    const double nn = n0 * n0 + n1 * n1;
                    const double inv = nn > 0.0 ? 1.0 / sqrt(nn) : 1.0;
                    rows[3 * n_rows] = n0 * inv; rows[3 * n_rows + 1] = n1 * inv; rows[3 * n_rows + 2] = -rhs * inv;
                }
                ++n_rows;
            }
        };
        for (int i = 1; i < N; ++i) {
            // successor of phi_{i-1} and its forward differences
            double xp[4], xn[4], xq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xp[j] = (j == q) ? x[j] + P.eps : x[j];
            bk_closed_step(x, P, xn);
            bk_closed_step(xp, P, xq);
            double col[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) col[r] = (xq[r] - xn[r]) / P.eps;  // column q of A
            if (pending) {                                                 // finish row i-1 with f_pi = (phi_i - phi_{i-1}) / dt
                double b = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) b += pg[j] * ((xn[j] - x[j]) / P.dt);
                emit(pl0, pl1, -pa + b - pdh - pah);
            }
            // S_i = A S_{i-1}, column q here
            double Sn[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double a0 = quad_bcast<0>(col[r]), a1 = quad_bcast<1>(col[r]), a2 = quad_bcast<2>(col[r]), a3 = quad_bcast<3>(col[r]);
                Sn[r] = a0 * Sc[0] + a1 * Sc[1] + a2 * Sc[2] + a3 * Sc[3];
            }
            const double xprev[4] = {x[0], x[1], x[2], x[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) { Sc[j] = Sn[j]; x[j] = xn[j]; }
            // row i at (phi_i, t_i)
            const double t = (double)i * P.dt;
            const double hb = bk_h_safety(x[0], x[1], t, bx, P);
            const double hq = bk_h_safety(q == 0 ? x[0] + P.eps : x[0], q == 1 ? x[1] + P.eps : x[1], t, bx, P);
            const double hdt = bk_h_safety(x[0], x[1], t + P.dt, bx, P);
            h_min = fmin(h_min, hb);
            const double gq = (hq - hb) / P.eps;
            const double g0 = quad_bcast<0>(gq), g1 = quad_bcast<1>(gq), g2 = quad_bcast<2>(gq), g3 = quad_bcast<3>(gq);
            const double gSq = g0 * Sc[0] + g1 * Sc[1] + g2 * Sc[2] + g3 * Sc[3];
            const double gS0 = quad_bcast<0>(gSq), gS1 = quad_bcast<1>(gSq), gS2 = quad_bcast<2>(gSq), gS3 = quad_bcast<3>(gSq);
            pg[0] = g0; pg[1] = g1; pg[2] = g2; pg[3] = g3;
            pa = gS0 * f0x + gS1 * f0y; pdh = (hdt - hb) / P.dt; pah = P.alpha * hb; pl0 = gS2; pl1 = gS3;
            pending = true;
            if (i == N - 1) {                                              // last safety row: f_pi = (phi_i - phi_{i-1}) / dt (:645-646)
                double b = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) b += pg[j] * ((x[j] - xprev[j]) / P.dt);
                emit(pl0, pl1, -pa + b - pdh - pah);
                pending = false;
            }
        }
        {   // terminal row at phi_{N-1} (:668-676)
            double xp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xp[j] = (j == q) ? x[j] + P.eps : x[j];
            const double hT = bk_h_terminal(x, bx, P), hq = bk_h_terminal(xp, bx, P);
            h_min = fmin(h_min, hT);
            const double gq = (hq - hT) / P.eps;
            const double g0 = quad_bcast<0>(gq), g1 = quad_bcast<1>(gq), g2 = quad_bcast<2>(gq), g3 = quad_bcast<3>(gq);
            const double gSq = g0 * Sc[0] + g1 * Sc[1] + g2 * Sc[2] + g3 * Sc[3];
            const double gS0 = quad_bcast<0>(gSq), gS1 = quad_bcast<1>(gSq), gS2 = quad_bcast<2>(gSq), gS3 = quad_bcast<3>(gSq);
            emit(gS2, gS3, -((gS0 * f0x + gS1 * f0y) + P.alphaT * hT));
        }
        __syncthreads();                                                   // rows of the quad are in LDS

        // ---- QP in scaled variables: min |us - us_ref|^2, rows, -1 <= us <= 1 (Q_u = [1, 1], :104) ----------------------
        double ur0 = un0, ur1 = un1;
        qp_status = -1; using_backup = 0;
        if (n_rows > 0) {                                                  // (quad-uniform)
            ur0 = fmin(fmax(un0, -P.amax), P.amax); ur1 = fmin(fmax(un1, -P.amax), P.amax);   // :699
        }
        const double sr0 = ur0 / P.amax, sr1 = ur1 / P.amax;
        CbfConsts<double> kb;
        kb.lo0 = -1.0; kb.hi0 = 1.0; kb.lo1 = -1.0; kb.hi1 = 1.0;
        QpState<double> S;
        qp_begin(S, sr0, sr1, kb);
        unsigned member = 0u;                                              // bit k: my row q + 4 k is in the working set
        bool infeasible = bad, done = n_rows == 0 || bad;
        for (int it = 0; it <= N; ++it) {
            // most violated row outside the working set
            double worst = 0.0;
            int wi = -1;
            if (!done) {
                for (int r = q, k = 0; r < n_rows; r += 4, ++k) {
                    const double s = rows[3 * r] * S.u0 + (rows[3 * r + 1] * S.u1 + rows[3 * r + 2]);
                    if (!((member >> k) & 1u) && s < worst) { worst = s; wi = r; }
                }
            }
            {
                double ow = quad_move<0xB1>(worst); int oi = quad_move<0xB1>(wi);
                if (ow < worst || (ow == worst && oi >= 0 && (wi < 0 || oi < wi))) { worst = ow; wi = oi; }
                ow = quad_move<0x4E>(worst); oi = quad_move<0x4E>(wi);
                if (ow < worst || (ow == worst && oi >= 0 && (wi < 0 || oi < wi))) { worst = ow; wi = oi; }
            }
            const bool has = wi >= 0;
            if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;           // wave-uniform: every quad is done
            if (has) {
                if ((wi & 3) == q) member |= 1u << (wi >> 2);
                LineQP<double> Ln;
                qp_row_violated(S, rows[3 * wi], rows[3 * wi + 1], rows[3 * wi + 2], Ln, kb);
                clip_box(Ln, kb);
                for (int r = q, k = 0; r < n_rows; r += 4, ++k)
                    if (((member >> k) & 1u) && r != wi) clip_row(Ln, rows[3 * r], rows[3 * r + 1], rows[3 * r + 2]);
                Ln.lo = quad_max(Ln.lo); Ln.hi = quad_min(Ln.hi);
                if (Ln.lo > Ln.hi + 1e-9) { infeasible = true; done = true; }   // the working set is empty on this line: so is the QP
                qp_row_commit(S, Ln, true);
            }
        }
        qp_finish_box(S, kb);
        if (n_rows > 0 && !bad) {
            double wm = num<double>::inf(), poison = 0.0;
            for (int r = q; r < n_rows; r += 4) wm = qp_row_margin(wm, rows[3 * r], rows[3 * r + 1], rows[3 * r + 2], S.u0, S.u1, poison);
            wm = quad_min(wm);
            const bool nanp = !(poison == poison) || !(S.u0 == S.u0) || !(S.u1 == S.u1);
            const unsigned long long nm = __builtin_amdgcn_ballot_w64(nanp);
            if (!(wm >= 0.0) || ((nm >> (lane & ~3)) & 0xFull)) infeasible = true;
        }
        __syncthreads();                                                   // rows are free for the next control step

        // ---- output selection (:737-774) ---------------------------------------------------------------------------------
        if (n_rows == 0 && !bad) {
            uo0 = un0; uo1 = un1;                                          // no rows: the reference input as it came
        } else if (!infeasible) {
            qp_status = 0;
            uo0 = P.amax * S.u0; uo1 = P.amax * S.u1;
            const double d0 = S.u0 - sr0, d1 = S.u1 - sr1;
            using_backup = sqrt(d0 * d0 + d1 * d1) > 0.1;
        } else {
            qp_status = 1;
            if (h_min > 0.01) { uo0 = ur0; uo1 = ur1; }
            else { bk_backup(xs, P, uo0, uo1); using_backup = 1; }
        }

        // ---- closed loop: the example's step (test_evade.py:456-500) -------------------------------------------------
        if (advance && rcode == 0) {
            const double pos0 = xs[0], pos1 = xs[1];
            double xn[4];
            bk_step(xs, uo0, uo1, P, xn);
            const double vm = sqrt(xn[2] * xn[2] + xn[3] * xn[3]);
            if (vm > P.vmax) { xn[2] = xn[2] * P.vmax / vm; xn[3] = xn[3] * P.vmax / vm; }
#pragma unroll
            for (int j = 0; j < 4; ++j) xs[j] = xn[j];
            bx += P.bspeed * P.dt;                                         // EvadeEnv.step_bullet (evade_env.py:360-384)
            if (bx > P.L + P.blen) bx = P.bstart;
            const double cxx = fmin(fmax(pos0, bx - P.blen / 2), bx + P.blen / 2 + P.blen / 3);
            const double cyy = fmin(fmax(pos1, -P.bhalf_wid), P.bhalf_wid);
            const double ddx = pos0 - cxx, ddy = pos1 - cyy;
            if (sqrt(ddx * ddx + ddy * ddy) < P.R) { rcode = -2; rstep = step0 + cs; }
            else if (P.gxmin <= pos0 && pos0 <= P.gxmax && -P.hw <= pos1 && pos1 <= P.hw) { rcode = 1; rstep = step0 + cs; }
        }
    }

    if (active && q == 0) {
        st(u_out, agent * 2, uo0); st(u_out, agent * 2 + 1, uo1);
        status_out[agent] = qp_status;
        if (using_backup_out) using_backup_out[agent] = using_backup;
        if (h_min_out) st(h_min_out, agent, h_min);
        if (n_rows_out) n_rows_out[agent] = n_rows;
        if (advance) {
#pragma unroll
            for (int j = 0; j < 4; ++j) st(X, agent * 4 + j, xs[j]);
            if (!p.bullet_shared) st(bullet_x, agent, bx);
            if (ret) ret[agent] = rcode;
            if (ret_step) ret_step[agent] = rstep;
        }
    }
}

}  // namespace

hipError_t backupcbf_launch(const sc_backupcbf_params& p, long long B, int n_ctrl, int advance, void* X, const void* u_nom,
                            void* bullet_x, void* u_out, int* status, int* using_backup, void* h_min, int* n_rows, double* rows_out,
                            int* ret, int* ret_step, int step0, hipStream_t stream) {
    const size_t lds = (size_t)16 * p.n_steps * 3 * sizeof(double);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(backupcbf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const unsigned blocks = (unsigned)((B + 15) / 16);
    hipLaunchKernelGGL(backupcbf_kernel, dim3(blocks), dim3(64), lds, stream, p, B, n_ctrl, advance, X, u_nom, bullet_x, u_out, status,
                       using_backup, h_min, n_rows, rows_out, ret, ret_step, step0);
    return hipGetLastError();
}

}  // namespace sc
