// Manipulator2D CBF-QP: three joint-velocity inputs, one CBF row per link circle per obstacle.
//   CBFQP.setup_control_problem / solve_control_problem, Manipulator2D branches   position_control/cbf_qp.py:94-104, :130-151
//   Manipulator2D.get_link_circles / get_points_jacobian / agent_barrier         robots/manipulator2D.py:129-224
//
// One agent per wavefront.  The reference discretises the three links into 25 circles and writes one row per circle and
// obstacle until `num_obs` rows are used (150 by default, tracking.py:134-138), so a QP has 3 variables and up to
// 250 + 6 rows: every lane builds and keeps `RPL` rows in registers (row r lives in lane r % 64, slot r / 64), and the
// wave solves
//        minimise ||u - u_ref||^2   s.t.  n_r . u + c_r >= 0  (CBF rows and the box |u_i| <= w_max as six more rows)
// with the dual active-set method of Goldfarb and Idnani specialised to an identity Hessian: start at u_ref, pick the
// most violated (normalised) row by a wave arg-min, move along the projection of its normal onto the null space of the
// active normals until the row is satisfied or a multiplier of an active row reaches zero (that row is dropped), repeat.
// The active set has at most three rows, so its QR (Gram-Schmidt) is recomputed from the stored normals each time --
// wave-uniform arithmetic every lane does redundantly; the only cross-lane traffic per round is the arg-min (a DPP
// reduction, a ballot and a v_readlane) and the broadcast of the chosen row (v_readlane).  The objective is strictly convex, so the minimiser is unique and the
// oracle's different algorithm (constraint generation around active-set enumeration, oracle/qp.py: solve_qpn) is a real
// check.  Arithmetic is f64; the caller's arrays are f32 or f64.
#include <hip/hip_runtime.h>

#include "../../include/safe_control_amd.h"
#include "sc_math.hpp"
#include "sc_qp2.hpp"

namespace sc {

namespace {

struct ArgMin { double v; int i; };

// Wave arg-min without a shuffle butterfly (18 dependent ds_bpermute round trips per call, and the solver calls it once
// per active-set round): the minimum VALUE comes from a DPP reduction, the lanes holding it answer a ballot, the first of
// them supplies the row index through v_readlane.  Ties (the joint circles of two links coincide, so identical rows are
// common) go to the lowest lane instead of the lowest row index -- the rows are identical, the QP does not care.
template <int CTRL>
__device__ __forceinline__ double am_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double am_lane(double v, int src) {            // src wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ ArgMin wave_argmin(ArgMin a, int& owner) {
    double v = a.v;
    v = fmin(v, am_dpp<0xB1>(v)); v = fmin(v, am_dpp<0x4E>(v)); v = fmin(v, am_dpp<0x141>(v)); v = fmin(v, am_dpp<0x140>(v));
    const double mn = fmin(fmin(am_lane(v, 0), am_lane(v, 16)), fmin(am_lane(v, 32), am_lane(v, 48)));
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(a.v == mn);
    owner = hit ? (int)__builtin_ctzll(hit) : 0;                          // hit == 0 only for NaN slacks: the caller stops
    ArgMin r;
    r.v = mn;
    r.i = __builtin_amdgcn_readlane(a.i, owner);
    return r;
}

// Orthonormal basis of the active normals and the triangular factor (N = Q R), q <= 3.
struct ActiveSet {
    int q;
    int idx[3];
    double n[3][3];      // active normals (unit length)
    double lam[3];       // their multipliers (>= 0)
};

struct Qr { double Q[3][3]; double R[3][3]; double Rinv[3]; };   // Rinv[j] = 1 / R[j][j]

__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ __forceinline__ void factor(const ActiveSet& A, Qr& F) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < A.q) {
            double v[3] = {A.n[j][0], A.n[j][1], A.n[j][2]};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < j) {
                    const double r = dot3(F.Q[i], A.n[j]);
                    F.R[i][j] = r;
                    v[0] -= r * F.Q[i][0]; v[1] -= r * F.Q[i][1]; v[2] -= r * F.Q[i][2];
                }
            }
            // the scalar chain of the active-set step is latency bound: v_rsq / v_rcp seeds + Newton (sc_qp2.hpp) instead of
            // the IEEE sqrt and divisions (an ulp or two apart; the active normals are independent by construction, |v| > 1e-6)
            const double vv = dot3(v, v);
            const double inv = rsqrt_(vv);
            F.R[j][j] = vv * inv;
            F.Rinv[j] = inv;
            F.Q[j][0] = v[0] * inv; F.Q[j][1] = v[1] * inv; F.Q[j][2] = v[2] * inv;
        }
    }
}

// z = (I - Q Q^T) np,  r = R^-1 Q^T np
__device__ __forceinline__ void directions(const ActiveSet& A, const Qr& F, const double* np, double* z, double* r) {
    double d[3] = {0, 0, 0};
    z[0] = np[0]; z[1] = np[1]; z[2] = np[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < A.q) {
            d[i] = dot3(F.Q[i], np);
            z[0] -= d[i] * F.Q[i][0]; z[1] -= d[i] * F.Q[i][1]; z[2] -= d[i] * F.Q[i][2];
        }
    }
    r[0] = r[1] = r[2] = 0.0;
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i < A.q) {
            double s = d[i];
#pragma unroll
            for (int j = 2; j > i; --j)
                if (j < A.q) s -= F.R[i][j] * r[j];
            r[i] = s * F.Rinv[i];
        }
    }
}

__device__ __forceinline__ void drop(ActiveSet& A, int l) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j >= l && j + 1 < A.q) {
            A.idx[j] = A.idx[j + 1]; A.lam[j] = A.lam[j + 1];
            A.n[j][0] = A.n[j + 1][0]; A.n[j][1] = A.n[j + 1][1]; A.n[j][2] = A.n[j + 1][2];
        }
    }
    A.q -= 1;
}

// One Manipulator2D CBF-QP for the wave: rows from the joint angles (q0, q1, q2) and `kv` obstacles read through
// getobs(o, f), minimiser of ||u - (ur0, ur1, ur2)||^2 in u[]; returns SC_STATUS_*.  puth(r, h) receives the barrier value
// of row r (0 for unused rows).  Everything is wave-uniform except the rows, which live RPL per lane.
template <int RPL, typename GetObs, typename PutH>
__device__ __forceinline__ int manip_qp(const sc_manip_cbfqp_params& p, const double q0, const double q1, const double q2,
                                        const double ur0, const double ur1, const double ur2, const int kv, const int lane,
                                        GetObs getobs, PutH puth, double (&u)[3]) {
    // ---- kinematic chain (wave-uniform): joints P0..P2, link vectors d0..d2 (manipulator2D.py:129-152) ----------
    double Px[3], Py[3], dx[3], dy[3];
    {
        double ang = 0.0, px = p.base_pos[0], py = p.base_pos[1];
        const double qs[3] = {q0, q1, q2};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ang += qs[i];
            double s, c;
            sincos_(ang, &s, &c);
            Px[i] = px; Py[i] = py;
            dx[i] = p.link_lengths[i] * c; dy[i] = p.link_lengths[i] * s;
            px += dx[i]; py += dy[i];
        }
    }
    const int c0 = p.link_steps[0] + 1, c1 = p.link_steps[1] + 1, c2 = p.link_steps[2] + 1;
    const int C = c0 + c1 + c2;
    const int Rr = min(p.num_rows, kv * C);                       // CBF rows in use (cbf_qp.py:126-128, :133)
    const int m = Rr + 6;                                         // + the box
    const double gain = p.cbf_mode == SC_CBF_MODE_HARD ? 1.0 / p.dt : p.alpha;    // cbf_qp.py:136-147

    double n0[RPL], n1[RPL], n2[RPL], cc[RPL];
    bool dead = false;                                            // a row 0.u + c >= 0 with c < 0, or non-finite data
#pragma unroll
    for (int s = 0; s < RPL; ++s) {
        const int r = lane + 64 * s;
        n0[s] = n1[s] = n2[s] = 0.0;
        cc[s] = num<double>::inf();                               // unused slot: never violated
        if (r < Rr) {
            const int o = r / C, ci = r - o * C;
            int li, j, ns;
            if (ci < c0) { li = 0; j = ci; ns = p.link_steps[0]; }
            else if (ci < c0 + c1) { li = 1; j = ci - c0; ns = p.link_steps[1]; }
            else { li = 2; j = ci - c0 - c1; ns = p.link_steps[2]; }
            const double t = (double)j / (double)ns;
            const double sx = li == 0 ? Px[0] : (li == 1 ? Px[1] : Px[2]), sy = li == 0 ? Py[0] : (li == 1 ? Py[1] : Py[2]);
            const double lx = li == 0 ? dx[0] : (li == 1 ? dx[1] : dx[2]), ly = li == 0 ? dy[0] : (li == 1 ? dy[1] : dy[2]);
            const double cx = sx + t * lx, cy = sy + t * ly;
            const double ox = getobs(o, 0), oy = getobs(o, 1);
            const double orad = getobs(o, 2);
            const double ex = cx - ox, ey = cy - oy;
            const double dmin = p.robot_radius + orad;
            const double h = ex * ex + ey * ey - p.beta * (dmin * dmin);
            // dh/dq_k = 2 (e . z x (c - P_k)) for k <= li (manipulator2D.py:154-182, :216)
            double a[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double jx = -(cy - Py[k]), jy = cx - Px[k];
                a[k] = k <= li ? 2.0 * (ex * jx + ey * jy) : 0.0;
            }
            const double b = gain * h;
            puth(r, h);
            const double nn = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
            if (!(finite_(nn) && finite_(b))) dead = true;
            if (nn > 0.0) {
                const double inv = rsqrt_(nn);
                n0[s] = a[0] * inv; n1[s] = a[1] * inv; n2[s] = a[2] * inv; cc[s] = b * inv;
            } else if (b < -num<double>::tol_feas() * fmax(1.0, fabs(b))) {
                dead = true;
            }
        } else if (r < m) {                                       // box rows: +-e_i . u + w_max >= 0
            const int bi = r - Rr, ax = bi >> 1;
            const double sg = (bi & 1) ? -1.0 : 1.0;
            n0[s] = ax == 0 ? sg : 0.0; n1[s] = ax == 1 ? sg : 0.0; n2[s] = ax == 2 ? sg : 0.0;
            cc[s] = p.w_max;
        }
        if (r >= Rr && r < p.num_rows) puth(r, 0.0);
    }

    u[0] = ur0; u[1] = ur1; u[2] = ur2;
    if (!(finite_(u[0]) && finite_(u[1]) && finite_(u[2]))) dead = true;
    int status = __any(dead) ? SC_STATUS_INFEASIBLE : -1;

    ActiveSet A;
    A.q = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) { A.idx[j] = -1; A.lam[j] = 0.0; A.n[j][0] = A.n[j][1] = A.n[j][2] = 0.0; }
    const double tol = num<double>::tol_feas();
    int budget = 8 * 64;                                          // far above what the method needs; guards the loop
    while (status < 0) {
        // most violated row (normalised slack); rows of the active set sit at ~0 and are never below -tol
        ArgMin best{num<double>::inf(), 0x7fffffff};
#pragma unroll
        for (int s = 0; s < RPL; ++s) {
            const double sl = n0[s] * u[0] + n1[s] * u[1] + n2[s] * u[2] + cc[s];
            const double mrg = sl + tol * fmax(1.0, fabs(cc[s]));
            if (mrg < best.v) { best.v = mrg; best.i = lane + 64 * s; }
        }
        int owner;
        best = wave_argmin(best, owner);
        if (!(best.v < 0.0)) { status = SC_STATUS_OPTIMAL; break; }
        const int pr = best.i, slot = pr >> 6;
        double np_[3], cp;
        {
            double a0 = n0[0], a1 = n1[0], a2 = n2[0], ac = cc[0];
#pragma unroll
            for (int s = 1; s < RPL; ++s)
                if (slot == s) { a0 = n0[s]; a1 = n1[s]; a2 = n2[s]; ac = cc[s]; }
            np_[0] = am_lane(a0, owner); np_[1] = am_lane(a1, owner); np_[2] = am_lane(a2, owner); cp = am_lane(ac, owner);
        }
        double sp = dot3(np_, u) + cp;                            // < 0
        double lam_p = 0.0;
        // bring row pr to equality, dropping active rows whose multiplier would turn negative
        while (true) {
            if (--budget < 0) { status = SC_STATUS_INACCURATE; break; }
            Qr F;
            factor(A, F);
            double z[3], r[3];
            directions(A, F, np_, z, r);
            const double zz = dot3(z, z);
            const bool dep = zz <= 1e-12;                         // np in the span of the active normals
            double t1 = num<double>::inf();
            int l = -1;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < A.q && r[j] > 1e-14) {
                    const double tj = A.lam[j] * rcp_(r[j]);
                    if (tj < t1) { t1 = tj; l = j; }
                }
            const double t2 = dep ? num<double>::inf() : -sp * rcp_(zz);
            if (l < 0 && dep) { status = SC_STATUS_INFEASIBLE; break; }
            const double t = fmin(t1, t2);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < A.q) A.lam[j] = fmax(A.lam[j] - t * r[j], 0.0);
            lam_p += t;
            if (!dep) {
                u[0] += t * z[0]; u[1] += t * z[1]; u[2] += t * z[2];
                sp += t * zz;
            }
            if (t2 <= t1) {                                       // full step: the row becomes active
                const int j = A.q;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (k == j) { A.idx[k] = pr; A.lam[k] = lam_p; A.n[k][0] = np_[0]; A.n[k][1] = np_[1]; A.n[k][2] = np_[2]; }
                A.q += 1;
                break;
            }
            drop(A, l);
        }
    }
    return status;
}

template <int RPL>
__global__ __launch_bounds__(256) void manip_cbfqp_kernel(const sc_manip_cbfqp_params p, const long long B, const int K,
                                                          const void* __restrict__ X, const void* __restrict__ u_ref,
                                                          const void* __restrict__ obs, const int* __restrict__ n_obs,
                                                          void* __restrict__ u_out, int* __restrict__ status_out,
                                                          void* __restrict__ h_out) {
    const int lane = threadIdx.x & 63;
    const long long agent = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (agent >= B) return;
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };
    const int kv = n_obs ? min(max(n_obs[agent], 0), K) : K;
    const size_t obase = p.obs_shared ? 0 : (size_t)agent * K * 7;
    double u[3];
    const int status = manip_qp<RPL>(p, ld(X, agent * 3 + 0), ld(X, agent * 3 + 1), ld(X, agent * 3 + 2), ld(u_ref, agent * 3 + 0),
                                     ld(u_ref, agent * 3 + 1), ld(u_ref, agent * 3 + 2), kv, lane,
                                     [&](int o, int f) { return ld(obs, obase + (size_t)o * 7 + f); },
                                     [&](int r, double h) { if (h_out) st(h_out, (size_t)agent * p.num_rows + r, h); }, u);
    if (lane == 0) {
        const bool ok = status == SC_STATUS_OPTIMAL;
        st(u_out, agent * 3 + 0, ok ? u[0] : num<double>::nan());
        st(u_out, agent * 3 + 1, ok ? u[1] : num<double>::nan());
        st(u_out, agent * 3 + 2, ok ? u[2] : num<double>::nan());
        status_out[agent] = status;
    }
}

// ---- closed loop for the arm: LocalTrackingController.control_step with a Manipulator2D robot, n_steps per launch ---------
//   tracking.py:497-535 (update_goal; goal_reached on the end effector :263-268), :559-668 (control_step),
//   manipulator2D.py:42-127 (end effector, Jacobian, nominal_input), :38-41 (step).  One arm per wavefront, joint angles in
//   registers across the steps; the obstacle table is shared, static and already ordered by distance to the base
//   (get_nearest_unpassed_obs ranks by the distance to robot.get_position(), which is the fixed base, robots/robot.py:354-356;
//   every obstacle passes the "unpassed" test because the model keeps the default angle of 2 pi, tracking.py:345-357).
//   Quirks kept: the collision test also uses the base position (tracking.py:445-495), so it is constant over the run; in
//   the step that leaves 'stop' the reference passes through update_goal's rotate branch, which compares the yaw (0) with
//   atan2(wp_y - X[1], wp_x - X[0]) -- joint angles where positions are meant -- and, when they differ by more than the
//   rotation threshold, returns the current waypoint without the reached test (tracking.py:505-516).
template <int RPL>
__global__ __launch_bounds__(64) void manip_rollout_kernel(const sc_manip_tracking_params t, const long long B, const int M,
                                                           void* __restrict__ X, const void* __restrict__ waypoints,
                                                           const int* __restrict__ n_wp, int* __restrict__ wp_index,
                                                           int* __restrict__ state_machine, void* __restrict__ goal,
                                                           const void* __restrict__ obs_table, void* __restrict__ u_last,
                                                           int* __restrict__ ret_out, int* __restrict__ ret_step,
                                                           void* __restrict__ traj_X, void* __restrict__ traj_U) {
    extern __shared__ __attribute__((aligned(16))) double table[];        // [M][7]
    const sc_manip_cbfqp_params& p = t.qp;
    const int lane = threadIdx.x;
    const long long agent = blockIdx.x;
    if (agent >= B) return;
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };
    for (int e = lane; e < M * 7; e += 64) table[e] = ld(obs_table, e);
    __syncthreads();
    double q[3] = {ld(X, agent * 3 + 0), ld(X, agent * 3 + 1), ld(X, agent * 3 + 2)};
    int wp = wp_index[agent], sm = state_machine[agent], ret = ret_out[agent], rstep = ret_step[agent];
    double gx = ld(goal, agent * 3 + 0), gy = ld(goal, agent * 3 + 1);
    bool gvalid = ld(goal, agent * 3 + 2) != 0.0;
    const int W = t.max_waypoints;
    const size_t wbase = t.waypoints_shared ? 0 : (size_t)agent * W * 2;
    const int nw = n_wp[t.waypoints_shared ? 0 : agent];
    const int kv = M < p.num_rows ? M : p.num_rows;                       // obstacles handed to the controller (tracking.py:584)
    bool hit = false;                                                     // the base never moves
    for (int mo = 0; mo < M; ++mo) {
        const double dx = p.base_pos[0] - table[7 * mo], dy = p.base_pos[1] - table[7 * mo + 1];
        hit |= sqrt(dx * dx + dy * dy) < table[7 * mo + 2] + p.robot_radius;
    }
    double ul[3] = {ld(u_last, agent * 3 + 0), ld(u_last, agent * 3 + 1), ld(u_last, agent * 3 + 2)};   // the last input applied so far

    auto end_effector = [&](double& ex, double& ey, double (&sn)[3], double (&cs)[3]) {
        double ang = 0.0;
        ex = p.base_pos[0]; ey = p.base_pos[1];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ang += q[i];
            sincos_(ang, &sn[i], &cs[i]);
            ex += p.link_lengths[i] * cs[i]; ey += p.link_lengths[i] * sn[i];
        }
    };
    auto update_goal = [&](bool from_stop) {
        if (from_stop && wp < nw) {                                       // rotate branch (see the header comment)
            const double ga = atan2(ld(waypoints, wbase + 2 * wp + 1) - q[1], ld(waypoints, wbase + 2 * wp) - q[0]);
            if (fabs(0.0 - ga) > t.rotation_threshold) { gx = ld(waypoints, wbase + 2 * wp); gy = ld(waypoints, wbase + 2 * wp + 1); gvalid = true; return; }
        }
        if (wp >= nw) { gvalid = false; return; }
        double ex, ey, sn[3], cs[3];
        end_effector(ex, ey, sn, cs);
        const double dx = ex - ld(waypoints, wbase + 2 * wp), dy = ey - ld(waypoints, wbase + 2 * wp + 1);
        if (sqrt(dx * dx + dy * dy) < t.reached_threshold) {
            wp += 1;
            if (wp >= nw) { sm = SC_SM_IDLE; gvalid = false; return; }
        }
        gx = ld(waypoints, wbase + 2 * wp); gy = ld(waypoints, wbase + 2 * wp + 1); gvalid = true;
    };

    for (int step = 0; step < t.n_steps; ++step) {
        const bool run = ret == 0;
        if (run) {
            if (sm == SC_SM_STOP) { sm = SC_SM_TRACK; update_goal(t.enable_rotation != 0); }   // has_stopped() is always true
            else update_goal(false);
        }
        // nominal input: Jacobian-transpose control, clipped (manipulator2D.py:110-127); no goal: stop() = zeros
        double ur[3] = {0.0, 0.0, 0.0};
        if (gvalid) {
            double ex, ey, sn[3], cs[3];
            end_effector(ex, ey, sn, cs);
            const double vx = t.Kp * (gx - ex), vy = t.Kp * (gy - ey);
            double jx = 0.0, jy = 0.0;
#pragma unroll
            for (int i = 2; i >= 0; --i) {
                jx -= p.link_lengths[i] * sn[i]; jy += p.link_lengths[i] * cs[i];
                ur[i] = fmin(fmax(jx * vx + jy * vy, -p.w_max), p.w_max);
            }
        }
        double u[3] = {ur[0], ur[1], ur[2]};
        int status = SC_STATUS_OPTIMAL;
        if (M > 0) status = manip_qp<RPL>(p, q[0], q[1], q[2], ur[0], ur[1], ur[2], kv, lane,
                                          [&](int o, int f) { return table[7 * o + f]; }, [](int, double) {}, u);
        const bool pre_fail = status != SC_STATUS_OPTIMAL || hit;
        const int code = pre_fail ? -2 : ((!gvalid && sm != SC_SM_STOP) ? -1 : 0);
        if (run) {
            if (!pre_fail) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { q[i] = q[i] + u[i] * p.dt; ul[i] = u[i]; }
            }
            if (code != 0) { ret = code; rstep = t.step_offset + step; }
        }
        if (lane < 3) {
            if (traj_X) st(traj_X, ((size_t)step * B + agent) * 3 + lane, lane == 0 ? q[0] : (lane == 1 ? q[1] : q[2]));
            if (traj_U) st(traj_U, ((size_t)step * B + agent) * 3 + lane, lane == 0 ? ul[0] : (lane == 1 ? ul[1] : ul[2]));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { st(X, agent * 3 + i, q[i]); st(u_last, agent * 3 + i, ul[i]); }
        wp_index[agent] = wp; state_machine[agent] = sm;
        st(goal, agent * 3 + 0, gx); st(goal, agent * 3 + 1, gy); st(goal, agent * 3 + 2, gvalid ? 1.0 : 0.0);
        ret_out[agent] = ret; ret_step[agent] = rstep;
    }
}

}  // namespace

hipError_t manip_cbfqp_launch(const sc_manip_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                              const void* obs, const int* n_obs, void* u_out, int* status, void* h_out, hipStream_t stream) {
    const int C = p.link_steps[0] + p.link_steps[1] + p.link_steps[2] + 3;
    long long rows = (long long)K * C;
    if (rows > p.num_rows) rows = p.num_rows;
    const int m = (int)rows + 6;
    const int rpl = (m + 63) / 64;
    const dim3 block(256), grid((unsigned)((B + 3) / 4));
    switch (rpl) {
        case 1: hipLaunchKernelGGL(manip_cbfqp_kernel<1>, grid, block, 0, stream, p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out); break;
        case 2: hipLaunchKernelGGL(manip_cbfqp_kernel<2>, grid, block, 0, stream, p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out); break;
        case 3: hipLaunchKernelGGL(manip_cbfqp_kernel<3>, grid, block, 0, stream, p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out); break;
        case 4: hipLaunchKernelGGL(manip_cbfqp_kernel<4>, grid, block, 0, stream, p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t manip_rollout_launch(const sc_manip_tracking_params& t, long long B, int M, void* X, const void* wps, const int* n_wp,
                                int* wp_index, int* sm, void* goal, const void* table, void* u_last, int* ret, int* ret_step,
                                void* tX, void* tU, hipStream_t stream) {
    const sc_manip_cbfqp_params& p = t.qp;
    const int C = p.link_steps[0] + p.link_steps[1] + p.link_steps[2] + 3;
    long long rows = (long long)(M < p.num_rows ? M : p.num_rows) * C;
    if (rows > p.num_rows) rows = p.num_rows;
    const int rpl = ((int)rows + 6 + 63) / 64;
    const size_t lds = (size_t)(M > 0 ? M : 1) * 7 * sizeof(double);
    const dim3 block(64), grid((unsigned)B);
    switch (rpl) {
        case 1: hipLaunchKernelGGL(manip_rollout_kernel<1>, grid, block, lds, stream, t, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU); break;
        case 2: hipLaunchKernelGGL(manip_rollout_kernel<2>, grid, block, lds, stream, t, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU); break;
        case 3: hipLaunchKernelGGL(manip_rollout_kernel<3>, grid, block, lds, stream, t, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU); break;
        case 4: hipLaunchKernelGGL(manip_rollout_kernel<4>, grid, block, lds, stream, t, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace sc
