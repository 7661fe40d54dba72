// Fused closed-loop control_step for B agents: one launch = n_steps iterations of
//   LocalTrackingController.control_step        tracking.py:559-668
//   LocalTrackingControllerDyn.control_step     dynamic_env/main.py:126-236 (moving obstacles)
// with the CBF-QP row builders / solver of the solve kernels behind the boundary.  One agent per
// lane; the agent's state (X, waypoint index, state machine, goal) lives in registers for the
// whole rollout; the shared obstacle table lives in LDS (and moves there when dyn_obs).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sc_qp2.hpp"
#include "sc_group.hpp"

namespace sc {

template <typename TIO> struct tvec2;
template <> struct tvec2<float> { using type = float2; };
template <> struct tvec2<double> { using type = double2; };

template <typename T>
struct TrackConsts {
    T reached, rot_thr, v_max, v_min, k_omega, k_a, k_v, delta_max, wheel_base, Lr, dt, a_max;
    int enable_rotation, dyn_obs, K;
};

// nominal_input: DU robots/dynamic_unicycle2D.py:80-104 ; KB robots/kinematic_bicycle2D.py:125-147
// (gains as BaseRobot forwards them, robots/robot.py:401-408)
template <typename T, int MODEL>
__device__ __forceinline__ void nominal_input(const T x, const T y, const T th, const T v, const T gx, const T gy,
                                              const TrackConsts<T>& t, T& u0, T& u1) {
    if constexpr (MODEL == SC_MODEL_SINGLE_INTEGRATOR2D || MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) {
        // single_integrator2D.py:72-90 / double_integrator2D.py:113-140 (th, v carry vx, vy for the double integrator):
        // dead-banded position error -> desired velocity, saturated in norm; DI: acceleration towards it, saturated in norm
        const T ex = gx - x, ey = gy - y;
        T vx = t.k_v * copysign(fmax_(fabs_(ex) - T(0.05), T(0)), ex), vy = t.k_v * copysign(fmax_(fabs_(ey) - T(0.05), T(0)), ey);
        const T vm = sqrt_(vx * vx + vy * vy);
        if (vm > t.v_max) { vx = vx * t.v_max / vm; vy = vy * t.v_max / vm; }
        if constexpr (MODEL == SC_MODEL_SINGLE_INTEGRATOR2D) { u0 = vx; u1 = vy; }
        else {
            T ax = t.k_a * (vx - th), ay = t.k_a * (vy - v);
            const T am = sqrt_(ax * ax + ay * ay);
            if (am > t.a_max) { ax = ax * t.a_max / am; ay = ay * t.a_max / am; }
            u0 = ax; u1 = ay;
        }
        return;
    }
    const T pi = T(3.14159265358979323846);
    const T dx = x - gx, dy = y - gy;
    const T dist = sqrt_(dx * dx + dy * dy);
    const T err = angle_normalize(atan2_(gy - y, gx - x) - th);
    T sn, cs;
    sincos_(err, &sn, &cs);
    if constexpr (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) {
        const T d = fmax_(dist - T(0.05), T(0));
        const T vd = (fabs_(err) > pi / T(2)) ? T(0) : fmin_(t.k_v * d * cs, t.v_max);
        u0 = t.k_a * (vd - v);
        u1 = t.k_omega * err;
    } else if constexpr (MODEL == SC_MODEL_UNICYCLE2D) {
        // robots/unicycle2D.py:69-85 with the gains BaseRobot forwards (robots/robot.py:404-405): d_min .05,
        // k_omega 2, k_v 1; the speed command is not clipped
        const T d = fmax_(dist - T(0.05), T(0.05));
        u0 = (fabs_(err) > pi / T(2)) ? T(0) : T(1) * d * cs;
        u1 = T(2) * err;
    } else {
        const T d = fmax_(dist - T(0.05), T(0.05));
        const T delta = fmin_(fmax_(t.k_omega * err, -t.delta_max), t.delta_max);
        u1 = atan(double((t.Lr / t.wheel_base) * tan(double(delta))));
        const T vcmd = t.k_v * d * fmax_(T(0), cs);
        const T vd = fmin_(fmax_(vcmd, t.v_min), t.v_max);
        u0 = t.k_a * (vd - v);
    }
}

// stop(): DU brakes with k_a (dynamic_unicycle2D.py:106-111); KB and Unicycle2D return zeros (kinematic_bicycle2D.py:149-150,
// unicycle2D.py:87-88)
// SingleIntegrator2D: zeros (single_integrator2D.py:100-103); DoubleIntegrator2D brakes both components
// (double_integrator2D.py:149-156; th, v carry vx, vy)
template <typename T, int MODEL>
__device__ __forceinline__ void stop_input(const T th, const T v, const T k_a, T& u0, T& u1) {
    if constexpr (MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) { u0 = k_a * (T(0) - th); u1 = k_a * (T(0) - v); }
    else { u0 = (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) ? k_a * (T(0) - v) : T(0); u1 = T(0); }
}
// has_stopped(): |v| < .05 (DU :113-114, KB :152-153); the kinematic unicycle always has (unicycle2D.py:90-92)
template <typename T, int MODEL>
__device__ __forceinline__ bool has_stopped(const T th, const T v) {
    if constexpr (MODEL == SC_MODEL_UNICYCLE2D || MODEL == SC_MODEL_SINGLE_INTEGRATOR2D) return true;
    else if constexpr (MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) return sqrt_(th * th + v * v) < T(0.05);   // |(vx, vy)|
    else return fabs_(v) < T(0.05);
}
// step(): X + (f + g u) dt, heading wrapped (DU :75-78 ; KB :113-123, speed clipped ; Unicycle2D unicycle2D.py:64-67,
// whose 4th state column is padding and stays as it is)
template <typename T, int MODEL>
__device__ __forceinline__ void robot_step(const Agent<T>& a, const T u0, const T u1, const T dt, const T Lr, const T v_min,
                                           const T v_max, T& nx, T& ny, T& nth, T& nv) {
    if constexpr (MODEL == SC_MODEL_SINGLE_INTEGRATOR2D) {               // single_integrator2D.py:64-66; columns 2, 3 are padding
        nx = a.x + (T(0) + u0) * dt; ny = a.y + (T(0) + u1) * dt; nth = a.th; nv = a.v;
        return;
    } else if constexpr (MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) {        // double_integrator2D.py:79-107: speed rescaled to v_max
        nx = a.x + (a.f0 + T(0)) * dt; ny = a.y + (a.f1 + T(0)) * dt;
        nth = a.f0 + (T(0) + u0) * dt; nv = a.f1 + (T(0) + u1) * dt;
        const T vm = sqrt_(nth * nth + nv * nv);
        if (vm > v_max) { const T sc = v_max / vm; nth *= sc; nv *= sc; }
        return;
    } else if constexpr (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) {
        nx = a.x + (a.f0) * dt; ny = a.y + (a.f1) * dt;
        nth = a.th + (T(0) + u1) * dt; nv = a.v + (T(0) + u0) * dt;
    } else if constexpr (MODEL == SC_MODEL_UNICYCLE2D) {
        nx = a.x + (T(0) + a.c * u0) * dt; ny = a.y + (T(0) + a.s * u0) * dt;
        nth = a.th + (T(0) + u1) * dt; nv = a.v;
    } else {
        nx = a.x + (a.f0 + (-a.f1) * u1) * dt;
        ny = a.y + (a.f1 + a.f0 * u1) * dt;
        nth = a.th + (T(0) + (a.v / Lr) * u1) * dt;
        nv = a.v + (T(0) + u0) * dt;
        nv = fmin_(fmax_(nv, v_min), v_max);                              // np.clip in KinematicBicycle2D.step
    }
    nth = angle_normalize(nth);
}

// tracking.py:445-495 known-obstacle collision test (circle / superellipsoid, geometry rule :428-443)
template <typename T>
__device__ __forceinline__ bool collides(const T x, const T y, const T* table, int M, T R) {
    bool hit = false;
    for (int m = 0; m < M; ++m) {
        const T* o = table + 7 * m;
        const bool superell = (fabs_(o[6] - T(1)) <= T(1e-8) + T(1e-5)) && (o[4] >= T(2));     // np.isclose(flag, 1)
        if (!superell) {
            const T dx = x - o[0], dy = y - o[1];
            hit |= sqrt_(dx * dx + dy * dy) < o[2] + R;
        } else {
            T st, ct;
            sincos_(o[5], &st, &ct);
            const T px = ct * (x - o[0]) + st * (y - o[1]);
            const T py = -st * (x - o[0]) + ct * (y - o[1]);
            const T h = pow_(px / (o[2] + R), o[4]) + pow_(py / (o[3] + R), o[4]) - T(1);
            hit |= h <= T(0);
        }
    }
    return hit;
}

template <typename TIO, typename TC, int KMAX, int MODEL>
__global__ __launch_bounds__(64) void tracking_rollout_kernel(
        const sc_tracking_params p, const long long B, const int M,
        TIO* __restrict__ X, const TIO* __restrict__ waypoints, const int* __restrict__ n_wp,
        int* __restrict__ wp_index, int* __restrict__ state_machine, TIO* __restrict__ goal,
        TIO* __restrict__ obs_table, TIO* __restrict__ u_last, int* __restrict__ ret_out, int* __restrict__ ret_step,
        TIO* __restrict__ traj_X, TIO* __restrict__ traj_U) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC* table = reinterpret_cast<TC*>(smem_raw);                     // [M][7]
    const int lane = threadIdx.x;
    const long long agent = (long long)blockIdx.x * 64 + lane;
    const bool active = agent < B;
    const long long ag = active ? agent : 0;

    for (int e = lane; e < M * 7; e += 64) table[e] = TC(obs_table[e]);
    __syncthreads();

    const CbfConsts<TC> k = make_consts<TC>(p.qp);
    TrackConsts<TC> t;
    t.reached = TC(p.reached_threshold); t.rot_thr = TC(p.rotation_threshold);
    t.v_max = TC(p.v_max); t.v_min = TC(p.v_min);
    t.k_omega = TC(p.k_omega); t.k_a = TC(p.k_a); t.k_v = TC(p.k_v);
    t.delta_max = TC(p.delta_max); t.wheel_base = TC(p.wheel_base); t.Lr = TC(p.qp.rear_ax_dist); t.dt = TC(p.qp.dt); t.a_max = TC(p.qp.u_max[0]);
    t.enable_rotation = p.enable_rotation; t.dyn_obs = p.dyn_obs; t.K = p.num_constraints;
    const TC pi = TC(3.14159265358979323846);
    const TC half_unpassed = (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) ? TC(1.2) * pi / TC(2) : pi;   // tracking.py:352-357

    // ---- agent state -> registers ------------------------------------------------------------
    TC x = TC(X[ag * 4 + 0]), y = TC(X[ag * 4 + 1]), th = TC(X[ag * 4 + 2]), v = TC(X[ag * 4 + 3]);
    int wp = wp_index[ag], sm = state_machine[ag];
    TC gx = TC(goal[ag * 3 + 0]), gy = TC(goal[ag * 3 + 1]);
    bool gvalid = goal[ag * 3 + 2] != TIO(0);
    int ret = active ? ret_out[ag] : -2;
    int rstep = active ? ret_step[ag] : -1;                          // kept for agents frozen in an earlier launch
    const int W = p.max_waypoints;
    const TIO* wps = waypoints + (p.waypoints_shared ? 0 : (size_t)ag * W * 2);
    const int nw = n_wp[p.waypoints_shared ? 0 : ag];
    TC ul0 = TC(u_last[ag * 2 + 0]), ul1 = TC(u_last[ag * 2 + 1]);     // the last input applied so far

    auto wp_x = [&](int i) { return TC(wps[2 * i]); };
    auto wp_y = [&](int i) { return TC(wps[2 * i + 1]); };

    // tracking.py:497-535
    auto update_goal = [&]() {
        if (sm == SC_SM_ROTATE) {
            const int i = wp < nw ? wp : nw - 1;
            const TC rx = wp_x(i), ry = wp_y(i);
            const TC goal_angle = atan2_(ry - y, rx - x);
            if (!t.enable_rotation) sm = SC_SM_TRACK;
            if (fabs_(th - goal_angle) > t.rot_thr) { gx = rx; gy = ry; gvalid = true; return; }
            sm = SC_SM_TRACK;
        }
        if (wp >= nw) { gvalid = false; return; }
        {
            const TC dx = x - wp_x(wp), dy = y - wp_y(wp);
            if (sqrt_(dx * dx + dy * dy) < t.reached) {
                wp += 1;
                if (wp >= nw) { sm = SC_SM_IDLE; gvalid = false; return; }
            }
        }
        gx = wp_x(wp); gy = wp_y(wp); gvalid = true;
    };

    for (int step = 0; step < p.n_steps; ++step) {
        const bool run = (ret == 0);
        if (run) {
            // ---- state machine / goal (tracking.py:569-577) ----------------------------------
            if (sm == SC_SM_STOP) {
                if (has_stopped<TC, MODEL>(th, v)) {
                    sm = t.enable_rotation ? SC_SM_ROTATE : SC_SM_TRACK;
                    update_goal();
                }
            } else {
                update_goal();
            }
        }
        // ---- nearest unpassed obstacles (tracking.py:345-403): K smallest centre distances ------
        TC sd[KMAX];
        int si[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) { sd[j] = num<TC>::inf(); si[j] = -1; }
        int n_unpassed = 0;
        for (int m = 0; m < M; ++m) {
            const TC ang = atan2_(table[7 * m + 1] - y, table[7 * m] - x);
            n_unpassed += (fabs_(angle_normalize(ang - th)) <= half_unpassed) ? 1 : 0;
        }
        const bool use_all = n_unpassed == 0;
        for (int m = 0; m < M; ++m) {
            const TC ox = table[7 * m], oy = table[7 * m + 1];
            const TC ang = atan2_(oy - y, ox - x);
            const bool pass = use_all || (fabs_(angle_normalize(ang - th)) <= half_unpassed);
            const TC dx = ox - x, dy = oy - y;
            TC cd = pass ? sqrt_(dx * dx + dy * dy) : num<TC>::inf();
            int ci = pass ? m : -1;
            bool moved = false;                    // stable: once the candidate is placed, everything after it shifts
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const bool sw = moved || (cd < sd[j]);
                moved = sw;
                const TC td = sd[j]; const int ti = si[j];
                sd[j] = sw ? cd : td; si[j] = sw ? ci : ti;
                cd = sw ? td : cd; ci = sw ? ti : ci;
            }
        }
        // ---- rows in registers (selection order = distance order, as the reference passes them) ---
        const Agent<TC> agn = make_agent_m<TC, MODEL>(x, y, th, v);
        TC n0[KMAX], n1[KMAX], c[KMAX];
        bool bad_obs = false;
        TC poison = TC(0);
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const bool used = (j < t.K) && (si[j] >= 0);
            const TC* orow = table + 7 * (si[j] >= 0 ? si[j] : 0);
            TC o[7];
#pragma unroll
            for (int f = 0; f < 7; ++f) o[f] = orow[f];
            TC h, a0, a1, cc;
            const bool ok = cbf_row<TC, MODEL, true>(agn, o, k, a0, a1, cc, h);
            bad_obs |= used && !ok;
            a0 = used ? a0 : TC(0); a1 = used ? a1 : TC(0); cc = used ? cc : TC(0);
            normalise_row(a0, a1, cc, poison);
            n0[j] = a0; n1[j] = a1; c[j] = cc;
        }
        // moving obstacles advance AFTER the selection (dynamic_env/main.py:147-150): the solve sees the old table
        __syncthreads();
        if (t.dyn_obs) {
            for (int m = lane; m < M; m += 64) {
                table[7 * m] += table[7 * m + 3] * t.dt;
                table[7 * m + 1] += table[7 * m + 4] * t.dt;
            }
        }
        __syncthreads();
        // ---- nominal input (tracking.py:589-604) ------------------------------------------------
        TC ur0, ur1;
        if (sm == SC_SM_ROTATE) {
            const TC ga = atan2_(gy - y, gx - x);
            ur0 = TC(0); ur1 = TC(2) * angle_normalize(ga - th);           // rotate_to, k = 2
        } else if (!gvalid) {
            stop_input<TC, MODEL>(th, v, t.k_a, ur0, ur1);                  // stop()
        } else {
            nominal_input<TC, MODEL>(x, y, th, v, gx, gy, t, ur0, ur1);
        }
        // ---- solve (cbf_qp.py:108-199) -----------------------------------------------------------
        TC u0, u1;
        int st;
        if (M == 0) { u0 = ur0; u1 = ur1; st = SC_STATUS_OPTIMAL; }         // obs_list None: u_ref unclipped
        else {
            st = qp2_solve<TC, KMAX>(n0, n1, c, t.K, ur0, ur1, poison, k, u0, u1);
            if (bad_obs) st = SC_STATUS_BAD_OBSTACLE;
        }
        // ---- collision / status / step (tracking.py:627-646) ---------------------------------------
        // pre-step: infeasible or already colliding -> -2, the robot does not move
        const bool pre_fail = (st != SC_STATUS_OPTIMAL) || collides<TC>(x, y, table, M, k.R);
        TC nx, ny, nth, nv;
        robot_step<TC, MODEL>(agn, u0, u1, t.dt, t.Lr, t.v_min, t.v_max, nx, ny, nth, nv);
        int code;
        if (pre_fail) code = -2;
        else if (collides<TC>(nx, ny, table, M, k.R)) code = -2;             // post-step: the robot HAS moved
        else code = (!gvalid && sm != SC_SM_STOP) ? -1 : 0;                  // tracking.py:666-667
        if (run) {
            if (!pre_fail) { x = nx; y = ny; th = nth; v = nv; ul0 = u0; ul1 = u1; }
            if (code != 0) { ret = code; rstep = p.step_offset + step; }
        }
        if (active && traj_X) {
            TIO* tx = traj_X + ((size_t)step * B + agent) * 4;
            tx[0] = TIO(x); tx[1] = TIO(y); tx[2] = TIO(th); tx[3] = TIO(v);
        }
        if (active && traj_U) {
            TIO* tu = traj_U + ((size_t)step * B + agent) * 2;
            tu[0] = TIO(ul0); tu[1] = TIO(ul1);
        }
    }

    if (active) {
        X[agent * 4 + 0] = TIO(x); X[agent * 4 + 1] = TIO(y); X[agent * 4 + 2] = TIO(th); X[agent * 4 + 3] = TIO(v);
        wp_index[agent] = wp; state_machine[agent] = sm;
        goal[agent * 3 + 0] = TIO(gx); goal[agent * 3 + 1] = TIO(gy); goal[agent * 3 + 2] = gvalid ? TIO(1) : TIO(0);
        u_last[agent * 2 + 0] = TIO(ul0); u_last[agent * 2 + 1] = TIO(ul1);
        ret_out[agent] = ret; ret_step[agent] = rstep;
    }
}

// ======================================================================================
// control_step split around the solve, for position controllers that are their own launch (MPC-CBF, optimal-decay
// MPC-CBF): `select` is everything before pos_controller.solve_control_problem (tracking.py:569-609), `apply`
// everything after it (:627-668).  One agent per lane (the solve between them dominates).
template <typename TIO, typename TC, int KMAX, int MODEL>
__global__ __launch_bounds__(64) void tracking_select_kernel(
        const sc_tracking_params p, const long long B, const int M,
        const TIO* __restrict__ X, const TIO* __restrict__ waypoints, const int* __restrict__ n_wp,
        int* __restrict__ wp_index, int* __restrict__ state_machine, TIO* __restrict__ goal,
        const TIO* __restrict__ obs_table, const int* __restrict__ ret_in,
        TIO* __restrict__ obs_out, TIO* __restrict__ goal_out, TIO* __restrict__ u_ref_out, int* __restrict__ track_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC* table = reinterpret_cast<TC*>(smem_raw);                     // [M][7]
    const int lane = threadIdx.x;
    const long long agent = (long long)blockIdx.x * 64 + lane;
    const bool active = agent < B;
    const long long ag = active ? agent : 0;
    for (int e = lane; e < M * 7; e += 64) table[e] = TC(obs_table[e]);
    __syncthreads();

    TrackConsts<TC> t;
    t.reached = TC(p.reached_threshold); t.rot_thr = TC(p.rotation_threshold);
    t.v_max = TC(p.v_max); t.v_min = TC(p.v_min);
    t.k_omega = TC(p.k_omega); t.k_a = TC(p.k_a); t.k_v = TC(p.k_v);
    t.delta_max = TC(p.delta_max); t.wheel_base = TC(p.wheel_base); t.Lr = TC(p.qp.rear_ax_dist); t.dt = TC(p.qp.dt); t.a_max = TC(p.qp.u_max[0]);
    t.enable_rotation = p.enable_rotation; t.dyn_obs = 0; t.K = p.num_constraints;
    const TC pi = TC(3.14159265358979323846);
    const TC half_unpassed = (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) ? TC(1.2) * pi / TC(2) : pi;

    const TC x = TC(X[ag * 4 + 0]), y = TC(X[ag * 4 + 1]), th = TC(X[ag * 4 + 2]), v = TC(X[ag * 4 + 3]);
    int wp = wp_index[ag], sm = state_machine[ag];
    TC gx = TC(goal[ag * 3 + 0]), gy = TC(goal[ag * 3 + 1]);
    bool gvalid = goal[ag * 3 + 2] != TIO(0);
    const bool run = active && ret_in[ag] == 0;
    const int W = p.max_waypoints;
    const TIO* wps = waypoints + (p.waypoints_shared ? 0 : (size_t)ag * W * 2);
    const int nw = n_wp[p.waypoints_shared ? 0 : ag];
    auto wp_x = [&](int i) { return TC(wps[2 * i]); };
    auto wp_y = [&](int i) { return TC(wps[2 * i + 1]); };
    auto update_goal = [&]() {                                       // tracking.py:497-535
        if (sm == SC_SM_ROTATE) {
            const int i = wp < nw ? wp : nw - 1;
            const TC rx = wp_x(i), ry = wp_y(i);
            const TC goal_angle = atan2_(ry - y, rx - x);
            if (!t.enable_rotation) sm = SC_SM_TRACK;
            if (fabs_(th - goal_angle) > t.rot_thr) { gx = rx; gy = ry; gvalid = true; return; }
            sm = SC_SM_TRACK;
        }
        if (wp >= nw) { gvalid = false; return; }
        {
            const TC dx = x - wp_x(wp), dy = y - wp_y(wp);
            if (sqrt_(dx * dx + dy * dy) < t.reached) {
                wp += 1;
                if (wp >= nw) { sm = SC_SM_IDLE; gvalid = false; return; }
            }
        }
        gx = wp_x(wp); gy = wp_y(wp); gvalid = true;
    };
    if (run) {
        if (sm == SC_SM_STOP) {
            if (has_stopped<TC, MODEL>(th, v)) {
                sm = t.enable_rotation ? SC_SM_ROTATE : SC_SM_TRACK;
                update_goal();
            }
        } else {
            update_goal();
        }
    }
    // nearest unpassed obstacles (tracking.py:345-403): K smallest centre distances
    TC sd[KMAX];
    int si[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { sd[j] = num<TC>::inf(); si[j] = -1; }
    int n_unpassed = 0;
    for (int m = 0; m < M; ++m) {
        const TC ang = atan2_(table[7 * m + 1] - y, table[7 * m] - x);
        n_unpassed += (fabs_(angle_normalize(ang - th)) <= half_unpassed) ? 1 : 0;
    }
    const bool use_all = n_unpassed == 0;
    for (int m = 0; m < M; ++m) {
        const TC ox = table[7 * m], oy = table[7 * m + 1];
        const TC ang = atan2_(oy - y, ox - x);
        const bool pass = use_all || (fabs_(angle_normalize(ang - th)) <= half_unpassed);
        const TC dx = ox - x, dy = oy - y;
        TC cd = pass ? sqrt_(dx * dx + dy * dy) : num<TC>::inf();
        int ci = pass ? m : -1;
        bool moved = false;                    // stable: once the candidate is placed, everything after it shifts
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const bool sw = moved || (cd < sd[j]);
            moved = sw;
            const TC td = sd[j]; const int ti = si[j];
            sd[j] = sw ? cd : td; si[j] = sw ? ci : ti;
            cd = sw ? td : cd; ci = sw ? ti : ci;
        }
    }
    TC ur0, ur1;
    if (sm == SC_SM_ROTATE) {
        const TC ga = atan2_(gy - y, gx - x);
        ur0 = TC(0); ur1 = TC(2) * angle_normalize(ga - th);
    } else if (!gvalid) {
        stop_input<TC, MODEL>(th, v, t.k_a, ur0, ur1);
    } else {
        nominal_input<TC, MODEL>(x, y, th, v, gx, gy, t, ur0, ur1);
    }
    if (active) {
        wp_index[agent] = wp; state_machine[agent] = sm;
        goal[agent * 3 + 0] = TIO(gx); goal[agent * 3 + 1] = TIO(gy); goal[agent * 3 + 2] = gvalid ? TIO(1) : TIO(0);
        // obstacle rows for the solve, padded like MPCCBF.update_tvp (mpc_cbf.py:338-364)
        TIO* oo = obs_out + (size_t)agent * t.K * 7;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (j >= t.K) break;
            const bool have = si[j] >= 0;
            const TC* orow = table + 7 * (have ? si[j] : 0);
#pragma unroll
            for (int f = 0; f < 7; ++f) oo[j * 7 + f] = have ? TIO(orow[f]) : (f < 2 ? TIO(1000) : TIO(0));
        }
        goal_out[agent * 2 + 0] = TIO(gvalid ? gx : x); goal_out[agent * 2 + 1] = TIO(gvalid ? gy : y);
        u_ref_out[agent * 2 + 0] = TIO(ur0); u_ref_out[agent * 2 + 1] = TIO(ur1);
        track_out[agent] = (run && sm == SC_SM_TRACK && gvalid) ? 1 : 0;
    }
}

template <typename TIO, typename TC, int MODEL>
__global__ __launch_bounds__(64) void tracking_apply_kernel(
        const sc_tracking_params p, const long long B, const int M, const int step_index,
        TIO* __restrict__ X, const int* __restrict__ state_machine, const TIO* __restrict__ goal,
        const TIO* __restrict__ obs_table, const TIO* __restrict__ u, const int* __restrict__ u_status,
        TIO* __restrict__ u_last, int* __restrict__ ret_out, int* __restrict__ ret_step) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC* table = reinterpret_cast<TC*>(smem_raw);
    const int lane = threadIdx.x;
    const long long agent = (long long)blockIdx.x * 64 + lane;
    const bool active = agent < B;
    const long long ag = active ? agent : 0;
    for (int e = lane; e < M * 7; e += 64) table[e] = TC(obs_table[e]);
    __syncthreads();
    const CbfConsts<TC> k = make_consts<TC>(p.qp);
    const TC dt = TC(p.qp.dt), Lr = TC(p.qp.rear_ax_dist);
    const TC x = TC(X[ag * 4 + 0]), y = TC(X[ag * 4 + 1]), th = TC(X[ag * 4 + 2]), v = TC(X[ag * 4 + 3]);
    const int sm = state_machine[ag];
    const bool gvalid = goal[ag * 3 + 2] != TIO(0);
    const TC u0 = TC(u[ag * 2 + 0]), u1 = TC(u[ag * 2 + 1]);
    const int st = u_status ? u_status[ag] : SC_STATUS_OPTIMAL;
    const bool run = active && ret_out[ag] == 0;
    const Agent<TC> agn = make_agent_m<TC, MODEL>(x, y, th, v);
    const bool pre_fail = (st != SC_STATUS_OPTIMAL) || collides<TC>(x, y, table, M, k.R);
    TC nx, ny, nth, nv;
    robot_step<TC, MODEL>(agn, u0, u1, dt, Lr, TC(p.v_min), TC(p.v_max), nx, ny, nth, nv);
    int code;
    if (pre_fail) code = -2;
    else if (collides<TC>(nx, ny, table, M, k.R)) code = -2;
    else code = (!gvalid && sm != SC_SM_STOP) ? -1 : 0;
    if (run) {
        if (!pre_fail) {
            X[agent * 4 + 0] = TIO(nx); X[agent * 4 + 1] = TIO(ny); X[agent * 4 + 2] = TIO(nth); X[agent * 4 + 3] = TIO(nv);
            u_last[agent * 2 + 0] = TIO(u0); u_last[agent * 2 + 1] = TIO(u1);
        }
        if (code != 0) { ret_out[agent] = code; ret_step[agent] = step_index; }
    }
}

// ======================================================================================
// Cooperative rollout: G lanes per agent (G = 8 or 16 >= num_constraints), one obstacle row per lane.
//
// With one agent per lane a 4096-agent rollout is 64 waves on 1024 SIMDs and a step is one wave's ~9 k
// instructions (K rows built one after the other, sorted insertion over M obstacles, sequential walk).  Here the
// G lanes of a group share one agent: the scalar parts (state machine, nominal input, Euler step) run redundantly
// in every lane, the obstacle scan / selection / row build / QP walk / collision tests are split over the lanes:
//   selection   lane sub scans obstacles sub, sub + G, ..; the rank of a candidate = number of candidates that
//               come before it in (distance, index) order, counted against group broadcasts; rank r < G goes to
//               lane r through a G-entry LDS slot -- the same order as the sorted insertion (ties: lower index)
//   rows / QP   the cooperative walk of the CBF-QP kernel (sc_group.hpp)
//   collisions  per-lane partial tests, group OR through the ballot
// Same arithmetic per row and per obstacle as the lane-per-agent kernel, which stays as the fallback for
// M > 4 G and as a cross-check (SC_TRACK_LANE_PER_AGENT=1).
template <typename T, int G>
__device__ __forceinline__ int group_sum_int(int v) {
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// one round of the rank count: candidate (slot Q2, lane I) of every group against this lane's CM candidates
template <typename TC, int G, int CM, int Q2, int I>
__device__ __forceinline__ void rank_step(const TC (&cd)[CM], int (&rank)[CM], int sub) {
    const TC od = group_bcast<TC, G, I>(cd[Q2], sub);
    const int m2 = I + Q2 * G;
#pragma unroll
    for (int q = 0; q < CM; ++q) {
        const int m = sub + q * G;
        rank[q] += ((od < cd[q]) || (od == cd[q] && m2 < m)) ? 1 : 0;
    }
}
template <typename TC, int G, int CM, int Q2, int... Is>
__device__ __forceinline__ void rank_round(const TC (&cd)[CM], int (&rank)[CM], int sub, std::integer_sequence<int, Is...>) {
    (rank_step<TC, G, CM, Q2, Is>(cd, rank, sub), ...);
}

template <typename T>
__device__ __forceinline__ bool collides_one(const T x, const T y, const T* o, T R) {
    const bool superell = (fabs_(o[6] - T(1)) <= T(1e-8) + T(1e-5)) && (o[4] >= T(2));     // np.isclose(flag, 1)
    if (!superell) {
        const T dx = x - o[0], dy = y - o[1];
        return sqrt_(dx * dx + dy * dy) < o[2] + R;
    }
    T st, ct;
    sincos_(o[5], &st, &ct);
    const T px = ct * (x - o[0]) + st * (y - o[1]);
    const T py = -st * (x - o[0]) + ct * (y - o[1]);
    return pow_(px / (o[2] + R), o[4]) + pow_(py / (o[3] + R), o[4]) - T(1) <= T(0);
}

template <typename TIO, typename TC, int G, int MODEL>
__global__ __launch_bounds__(64) void tracking_coop_kernel(
        const sc_tracking_params p, const long long B, const int M,
        TIO* __restrict__ X, const TIO* __restrict__ waypoints, const int* __restrict__ n_wp,
        int* __restrict__ wp_index, int* __restrict__ state_machine, TIO* __restrict__ goal,
        TIO* __restrict__ obs_table, TIO* __restrict__ u_last, int* __restrict__ ret_out, int* __restrict__ ret_step,
        TIO* __restrict__ traj_X, TIO* __restrict__ traj_U) {
    constexpr int APW = 64 / G;                                       // agents per wave
    constexpr int CM = 4;                                             // obstacles per lane (M <= CM * G)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TC* table = reinterpret_cast<TC*>(smem_raw);                     // [M][7]
    int* sel = reinterpret_cast<int*>(table + (size_t)(M > 0 ? M : 1) * 7);   // [64]: selected obstacle of every lane
    const int lane = threadIdx.x;
    const int sub = lane & (G - 1);
    const long long agent = (long long)blockIdx.x * APW + lane / G;
    const bool active = agent < B;
    const long long ag = active ? agent : 0;
    const unsigned long long grp = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane & ~(G - 1));

    for (int e = lane; e < M * 7; e += 64) table[e] = TC(obs_table[e]);
    __syncthreads();

    const CbfConsts<TC> k = make_consts<TC>(p.qp);
    TrackConsts<TC> t;
    t.reached = TC(p.reached_threshold); t.rot_thr = TC(p.rotation_threshold);
    t.v_max = TC(p.v_max); t.v_min = TC(p.v_min);
    t.k_omega = TC(p.k_omega); t.k_a = TC(p.k_a); t.k_v = TC(p.k_v);
    t.delta_max = TC(p.delta_max); t.wheel_base = TC(p.wheel_base); t.Lr = TC(p.qp.rear_ax_dist); t.dt = TC(p.qp.dt); t.a_max = TC(p.qp.u_max[0]);
    t.enable_rotation = p.enable_rotation; t.dyn_obs = p.dyn_obs; t.K = p.num_constraints;
    const TC pi = TC(3.14159265358979323846);
    const TC half_unpassed = (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) ? TC(1.2) * pi / TC(2) : pi;   // tracking.py:352-357

    TC x = TC(X[ag * 4 + 0]), y = TC(X[ag * 4 + 1]), th = TC(X[ag * 4 + 2]), v = TC(X[ag * 4 + 3]);
    int wp = wp_index[ag], sm = state_machine[ag];
    TC gx = TC(goal[ag * 3 + 0]), gy = TC(goal[ag * 3 + 1]);
    bool gvalid = goal[ag * 3 + 2] != TIO(0);
    int ret = active ? ret_out[ag] : -2;
    int rstep = active ? ret_step[ag] : -1;                          // kept for agents frozen in an earlier launch
    const int W = p.max_waypoints;
    const TIO* wps = waypoints + (p.waypoints_shared ? 0 : (size_t)ag * W * 2);
    const int nw = n_wp[p.waypoints_shared ? 0 : ag];
    TC ul0 = TC(u_last[ag * 2 + 0]), ul1 = TC(u_last[ag * 2 + 1]);     // the last input applied so far

    auto wp_x = [&](int i) { return TC(wps[2 * i]); };
    auto wp_y = [&](int i) { return TC(wps[2 * i + 1]); };
    auto update_goal = [&]() {                                       // tracking.py:497-535
        if (sm == SC_SM_ROTATE) {
            const int i = wp < nw ? wp : nw - 1;
            const TC rx = wp_x(i), ry = wp_y(i);
            const TC goal_angle = atan2_(ry - y, rx - x);
            if (!t.enable_rotation) sm = SC_SM_TRACK;
            if (fabs_(th - goal_angle) > t.rot_thr) { gx = rx; gy = ry; gvalid = true; return; }
            sm = SC_SM_TRACK;
        }
        if (wp >= nw) { gvalid = false; return; }
        {
            const TC dx = x - wp_x(wp), dy = y - wp_y(wp);
            if (sqrt_(dx * dx + dy * dy) < t.reached) {
                wp += 1;
                if (wp >= nw) { sm = SC_SM_IDLE; gvalid = false; return; }
            }
        }
        gx = wp_x(wp); gy = wp_y(wp); gvalid = true;
    };
    auto group_any = [&](bool b) { return (__builtin_amdgcn_ballot_w64(b) & grp) != 0ull; };
    auto collides_group = [&](TC px_, TC py_) {
        bool hit = false;
#pragma unroll
        for (int q = 0; q < CM; ++q) {
            if (q * G >= M) break;
            const int m = sub + q * G;
            if (m < M) hit |= collides_one<TC>(px_, py_, table + 7 * m, k.R);
        }
        return group_any(hit);
    };

    for (int step = 0; step < p.n_steps; ++step) {
        const bool run = (ret == 0);
        if (run) {
            if (sm == SC_SM_STOP) {                                   // tracking.py:569-577
                if (has_stopped<TC, MODEL>(th, v)) {
                    sm = t.enable_rotation ? SC_SM_ROTATE : SC_SM_TRACK;
                    update_goal();
                }
            } else {
                update_goal();
            }
        }
        // ---- nearest unpassed obstacles (tracking.py:345-403) ---------------------------------------
        TC cd[CM];
        int ci[CM];
        bool inview[CM];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < CM; ++q) { cd[q] = num<TC>::inf(); ci[q] = -1; inview[q] = false; }
#pragma unroll
        for (int q = 0; q < CM; ++q) {
            if (q * G >= M) break;                                    // uniform: only ceil(M / G) obstacles per lane exist
            const int m = sub + q * G;
            const bool valid = m < M;
            const TC* o = table + 7 * (valid ? m : 0);
            const TC ox = o[0], oy = o[1];
            const TC ang = atan2_(oy - y, ox - x);
            inview[q] = valid && (fabs_(angle_normalize(ang - th)) <= half_unpassed);
            cnt += inview[q] ? 1 : 0;
            const TC dx = ox - x, dy = oy - y;
            cd[q] = sqrt_(dx * dx + dy * dy);
            ci[q] = valid ? m : -1;
        }
        const bool use_all = group_sum_int<TC, G>(cnt) == 0;
#pragma unroll
        for (int q = 0; q < CM; ++q) {
            const bool pass = (ci[q] >= 0) && (use_all || inview[q]);
            cd[q] = pass ? cd[q] : num<TC>::inf();
            ci[q] = pass ? ci[q] : -1;
        }
        int rank[CM] = {0, 0, 0, 0};
        rank_round<TC, G, CM, 0>(cd, rank, sub, std::make_integer_sequence<int, G>{});
        if (M > G) rank_round<TC, G, CM, 1>(cd, rank, sub, std::make_integer_sequence<int, G>{});
        if (M > 2 * G) rank_round<TC, G, CM, 2>(cd, rank, sub, std::make_integer_sequence<int, G>{});
        if (M > 3 * G) rank_round<TC, G, CM, 3>(cd, rank, sub, std::make_integer_sequence<int, G>{});
        sel[lane] = -1;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CM; ++q)
            if (sub + q * G < M && rank[q] < G) sel[(lane & ~(G - 1)) + rank[q]] = ci[q];
        __syncthreads();
        const int si = sel[lane];
        // ---- this lane's row -------------------------------------------------------------------------
        const Agent<TC> agn = make_agent_m<TC, MODEL>(x, y, th, v);
        const bool used = (sub < t.K) && (si >= 0);
        TC a0, a1, cc;
        TC poison = TC(0);
        bool bad_mine;
        {
            const TC* orow = table + 7 * (si >= 0 ? si : 0);
            TC o[7];
#pragma unroll
            for (int f = 0; f < 7; ++f) o[f] = orow[f];
            TC h;
            const bool ok = cbf_row<TC, MODEL, true>(agn, o, k, a0, a1, cc, h);
            bad_mine = used && !ok;
            a0 = used ? a0 : TC(0); a1 = used ? a1 : TC(0); cc = used ? cc : TC(0);
            normalise_row(a0, a1, cc, poison);
        }
        // moving obstacles advance AFTER the selection (dynamic_env/main.py:147-150): the solve sees the old table
        __syncthreads();
        if (t.dyn_obs) {
            for (int m = lane; m < M; m += 64) {
                table[7 * m] += table[7 * m + 3] * t.dt;
                table[7 * m + 1] += table[7 * m + 4] * t.dt;
            }
        }
        __syncthreads();
        // ---- nominal input (tracking.py:589-604) ------------------------------------------------
        TC ur0, ur1;
        if (sm == SC_SM_ROTATE) {
            const TC ga = atan2_(gy - y, gx - x);
            ur0 = TC(0); ur1 = TC(2) * angle_normalize(ga - th);           // rotate_to, k = 2
        } else if (!gvalid) {
            stop_input<TC, MODEL>(th, v, t.k_a, ur0, ur1);                  // stop()
        } else {
            nominal_input<TC, MODEL>(x, y, th, v, gx, gy, t, ur0, ur1);
        }
        // ---- cooperative solve (cbf_qp.py:108-199) ---------------------------------------------------
        TC u0, u1;
        int st;
        if (M == 0) { u0 = ur0; u1 = ur1; st = SC_STATUS_OPTIMAL; }         // obs_list None: u_ref unclipped
        else {
            QpState<TC> S;
            qp_begin(S, ur0, ur1, k);
#ifndef SC_EXP_NOWALK                                     // developer builds: what the solve costs inside the rollout
            if constexpr (G == 8) coop_solve_all8<TC>(S, t.K, sub, lane, a0, a1, cc, k);       // all candidates at once (sc_group.hpp)
            else if constexpr (G == 16) coop_solve_all16<TC>(S, t.K, sub, lane, a0, a1, cc, k);
            else coop_walk_violated<TC, G>(S, t.K, sub, lane, a0, a1, cc, k);
#endif
            qp_finish_box(S, k);
            TC worst = qp_row_margin(num<TC>::inf(), a0, a1, cc, S.u0, S.u1, poison);
            worst = group_min<TC, G>(worst);
            if (group_any(!(poison == poison))) poison = num<TC>::nan();
            st = qp_status(S, worst, poison, k);
            if (group_any(bad_mine)) st = SC_STATUS_BAD_OBSTACLE;
            u0 = S.u0; u1 = S.u1;
        }
        // ---- collision / status / step (tracking.py:627-646) ---------------------------------------
        const bool pre_fail = (st != SC_STATUS_OPTIMAL) || collides_group(x, y);
        TC nx, ny, nth, nv;
        robot_step<TC, MODEL>(agn, u0, u1, t.dt, t.Lr, t.v_min, t.v_max, nx, ny, nth, nv);
        const bool post_hit = collides_group(nx, ny);
        int code;
        if (pre_fail) code = -2;
        else if (post_hit) code = -2;                                       // post-step: the robot HAS moved
        else code = (!gvalid && sm != SC_SM_STOP) ? -1 : 0;                  // tracking.py:666-667
        if (run) {
            if (!pre_fail) { x = nx; y = ny; th = nth; v = nv; ul0 = u0; ul1 = u1; }
            if (code != 0) { ret = code; rstep = p.step_offset + step; }
        }
        if (active && sub == 0 && traj_X) {
            TIO* tx = traj_X + ((size_t)step * B + agent) * 4;
            tx[0] = TIO(x); tx[1] = TIO(y); tx[2] = TIO(th); tx[3] = TIO(v);
        }
        if (active && sub == 0 && traj_U) {
            TIO* tu = traj_U + ((size_t)step * B + agent) * 2;
            tu[0] = TIO(ul0); tu[1] = TIO(ul1);
        }
    }

    if (active && sub == 0) {
        X[agent * 4 + 0] = TIO(x); X[agent * 4 + 1] = TIO(y); X[agent * 4 + 2] = TIO(th); X[agent * 4 + 3] = TIO(v);
        wp_index[agent] = wp; state_machine[agent] = sm;
        goal[agent * 3 + 0] = TIO(gx); goal[agent * 3 + 1] = TIO(gy); goal[agent * 3 + 2] = gvalid ? TIO(1) : TIO(0);
        u_last[agent * 2 + 0] = TIO(ul0); u_last[agent * 2 + 1] = TIO(ul1);
        ret_out[agent] = ret; ret_step[agent] = rstep;
    }
}

// Moving obstacles: every block of a rollout launch reads the table as it was at launch and advances its own LDS copy
// (a grid larger than one resident wave of blocks starts late blocks after early ones have finished, so nobody may
// write the table while the launch runs).  This stream-ordered follow-up leaves the table where n_steps of
// `obs[:, 0:2] += obs[:, 3:5] * dt` (dynamic_env/main.py:54-58) put it -- the same f64 additions in the same order.
template <typename TIO>
__global__ void advance_obstacle_table_kernel(TIO* __restrict__ obs_table, const int M, const int n_steps, const double dt) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    double x = double(obs_table[7 * m]), y = double(obs_table[7 * m + 1]);
    const double vx = double(obs_table[7 * m + 3]), vy = double(obs_table[7 * m + 4]);
    for (int s = 0; s < n_steps; ++s) { x += vx * dt; y += vy * dt; }
    obs_table[7 * m] = TIO(x); obs_table[7 * m + 1] = TIO(y);
}

template <typename TIO, typename TC, int G, int MODEL>
static hipError_t launch_track_coop(const sc_tracking_params& p, long long B, int M, void* X, const void* wps, const int* n_wp,
                                    int* wp_index, int* sm, void* goal, void* table, void* u_last, int* ret, int* ret_step,
                                    void* tX, void* tU, hipStream_t stream) {
    constexpr int APW = 64 / G;
    const unsigned blocks = (unsigned)((B + APW - 1) / APW);
    const size_t lds = (size_t)(M > 0 ? M : 1) * 7 * sizeof(TC) + 64 * sizeof(int);
    auto kern = tracking_coop_kernel<TIO, TC, G, MODEL>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, p, B, M, (TIO*)X, (const TIO*)wps, n_wp, wp_index, sm,
                       (TIO*)goal, (TIO*)table, (TIO*)u_last, ret, ret_step, (TIO*)tX, (TIO*)tU);
    return hipGetLastError();
}

template <typename TIO, typename TC, int KMAX, int MODEL>
static hipError_t launch_track(const sc_tracking_params& p, long long B, int M, void* X, const void* wps, const int* n_wp,
                               int* wp_index, int* sm, void* goal, void* table, void* u_last, int* ret, int* ret_step,
                               void* tX, void* tU, hipStream_t stream) {
    const unsigned blocks = (unsigned)((B + 63) / 64);
    const size_t lds = (size_t)(M > 0 ? M : 1) * 7 * sizeof(TC);
    auto kern = tracking_rollout_kernel<TIO, TC, KMAX, MODEL>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, p, B, M, (TIO*)X, (const TIO*)wps, n_wp, wp_index, sm,
                       (TIO*)goal, (TIO*)table, (TIO*)u_last, ret, ret_step, (TIO*)tX, (TIO*)tU);
    return hipGetLastError();
}

template <typename TIO, typename TC, int MODEL>
static hipError_t launch_track_k(const sc_tracking_params& p, long long B, int M, void* X, const void* wps, const int* n_wp,
                                 int* wp_index, int* sm, void* goal, void* table, void* u_last, int* ret, int* ret_step,
                                 void* tX, void* tU, hipStream_t stream) {
    const char* env_lpa = std::getenv("SC_TRACK_LANE_PER_AGENT");        // read per call: the tests flip it
    const bool lane_per_agent = env_lpa && env_lpa[0] == '1';
    if (!lane_per_agent) {
        if (p.num_constraints <= 8 && M <= 32)
            return launch_track_coop<TIO, TC, 8, MODEL>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
        if (M <= 64)
            return launch_track_coop<TIO, TC, 16, MODEL>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
    }
    if (p.num_constraints <= 8)
        return launch_track<TIO, TC, 8, MODEL>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
    return launch_track<TIO, TC, 16, MODEL>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
}

template <typename TIO, typename TC>
static hipError_t launch_track_m(const sc_tracking_params& p, long long B, int M, void* X, const void* wps, const int* n_wp,
                                 int* wp_index, int* sm, void* goal, void* table, void* u_last, int* ret, int* ret_step,
                                 void* tX, void* tU, hipStream_t stream) {
    switch (p.qp.model_id) {
        case SC_MODEL_DYNAMIC_UNICYCLE2D:
            return launch_track_k<TIO, TC, SC_MODEL_DYNAMIC_UNICYCLE2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
        case SC_MODEL_KINEMATIC_BICYCLE2D:
            return launch_track_k<TIO, TC, SC_MODEL_KINEMATIC_BICYCLE2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
        case SC_MODEL_UNICYCLE2D:
            return launch_track_k<TIO, TC, SC_MODEL_UNICYCLE2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
        case SC_MODEL_KINEMATIC_BICYCLE2D_C3BF:
            return launch_track_k<TIO, TC, SC_MODEL_KINEMATIC_BICYCLE2D_C3BF>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
        case SC_MODEL_SINGLE_INTEGRATOR2D:
            return launch_track_k<TIO, TC, SC_MODEL_SINGLE_INTEGRATOR2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
        case SC_MODEL_DOUBLE_INTEGRATOR2D:
            return launch_track_k<TIO, TC, SC_MODEL_DOUBLE_INTEGRATOR2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
        default:
            return launch_track_k<TIO, TC, SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
    }
}

template <typename TIO, int MODEL>
static hipError_t launch_select_m(const sc_tracking_params& p, long long B, int M, const void* X, const void* wps,
                                  const int* n_wp, int* wp_index, int* sm, void* goal, const void* table, const int* ret,
                                  void* obs_out, void* goal_out, void* u_ref_out, int* track_out, hipStream_t stream) {
    const unsigned blocks = (unsigned)((B + 63) / 64);
    const size_t lds = (size_t)(M > 0 ? M : 1) * 7 * sizeof(double);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, p, B, M, (const TIO*)X, (const TIO*)wps, n_wp, wp_index,
                           sm, (TIO*)goal, (const TIO*)table, ret, (TIO*)obs_out, (TIO*)goal_out, (TIO*)u_ref_out, track_out);
        return hipGetLastError();
    };
    if (p.num_constraints <= 8) return go(tracking_select_kernel<TIO, double, 8, MODEL>);
    return go(tracking_select_kernel<TIO, double, 16, MODEL>);
}

template <typename TIO>
static hipError_t launch_select_t(const sc_tracking_params& p, long long B, int M, const void* X, const void* wps,
                                  const int* n_wp, int* wp_index, int* sm, void* goal, const void* table, const int* ret,
                                  void* obs_out, void* goal_out, void* u_ref_out, int* track_out, hipStream_t stream) {
    if (p.qp.model_id == SC_MODEL_DYNAMIC_UNICYCLE2D)
        return launch_select_m<TIO, SC_MODEL_DYNAMIC_UNICYCLE2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, ret, obs_out, goal_out, u_ref_out, track_out, stream);
    if (p.qp.model_id == SC_MODEL_UNICYCLE2D)
        return launch_select_m<TIO, SC_MODEL_UNICYCLE2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, ret, obs_out, goal_out, u_ref_out, track_out, stream);
    if (p.qp.model_id == SC_MODEL_SINGLE_INTEGRATOR2D)
        return launch_select_m<TIO, SC_MODEL_SINGLE_INTEGRATOR2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, ret, obs_out, goal_out, u_ref_out, track_out, stream);
    if (p.qp.model_id == SC_MODEL_DOUBLE_INTEGRATOR2D)
        return launch_select_m<TIO, SC_MODEL_DOUBLE_INTEGRATOR2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, ret, obs_out, goal_out, u_ref_out, track_out, stream);
    return launch_select_m<TIO, SC_MODEL_KINEMATIC_BICYCLE2D>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, ret, obs_out, goal_out, u_ref_out, track_out, stream);
}

hipError_t tracking_select_launch(const sc_tracking_params& p, long long B, int M, const void* X, const void* wps,
                                  const int* n_wp, int* wp_index, int* sm, void* goal, const void* table, const int* ret,
                                  void* obs_out, void* goal_out, void* u_ref_out, int* track_out, hipStream_t stream) {
    if (p.qp.io_dtype == SC_DTYPE_F32)
        return launch_select_t<float>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, ret, obs_out, goal_out, u_ref_out, track_out, stream);
    return launch_select_t<double>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, ret, obs_out, goal_out, u_ref_out, track_out, stream);
}

template <typename TIO>
static hipError_t launch_apply_t(const sc_tracking_params& p, long long B, int M, int step_index, void* X, const int* sm,
                                 const void* goal, const void* table, const void* u, const int* u_status, void* u_last,
                                 int* ret, int* ret_step, hipStream_t stream) {
    const unsigned blocks = (unsigned)((B + 63) / 64);
    const size_t lds = (size_t)(M > 0 ? M : 1) * 7 * sizeof(double);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, p, B, M, step_index, (TIO*)X, sm, (const TIO*)goal,
                           (const TIO*)table, (const TIO*)u, u_status, (TIO*)u_last, ret, ret_step);
        return hipGetLastError();
    };
    if (p.qp.model_id == SC_MODEL_DYNAMIC_UNICYCLE2D) return go(tracking_apply_kernel<TIO, double, SC_MODEL_DYNAMIC_UNICYCLE2D>);
    if (p.qp.model_id == SC_MODEL_UNICYCLE2D) return go(tracking_apply_kernel<TIO, double, SC_MODEL_UNICYCLE2D>);
    if (p.qp.model_id == SC_MODEL_SINGLE_INTEGRATOR2D) return go(tracking_apply_kernel<TIO, double, SC_MODEL_SINGLE_INTEGRATOR2D>);
    if (p.qp.model_id == SC_MODEL_DOUBLE_INTEGRATOR2D) return go(tracking_apply_kernel<TIO, double, SC_MODEL_DOUBLE_INTEGRATOR2D>);
    return go(tracking_apply_kernel<TIO, double, SC_MODEL_KINEMATIC_BICYCLE2D>);
}

hipError_t tracking_apply_launch(const sc_tracking_params& p, long long B, int M, int step_index, void* X, const int* sm,
                                 const void* goal, const void* table, const void* u, const int* u_status, void* u_last,
                                 int* ret, int* ret_step, hipStream_t stream) {
    if (p.qp.io_dtype == SC_DTYPE_F32)
        return launch_apply_t<float>(p, B, M, step_index, X, sm, goal, table, u, u_status, u_last, ret, ret_step, stream);
    return launch_apply_t<double>(p, B, M, step_index, X, sm, goal, table, u, u_status, u_last, ret, ret_step, stream);
}

hipError_t tracking_launch(const sc_tracking_params& p, long long B, int M, void* X, const void* wps, const int* n_wp,
                           int* wp_index, int* sm, void* goal, void* table, void* u_last, int* ret, int* ret_step,
                           void* tX, void* tU, hipStream_t stream) {
    // closed loops amplify rounding: arithmetic is always f64 here; storage follows io_dtype
    const bool f32 = p.qp.io_dtype == SC_DTYPE_F32;
    hipError_t e = f32 ? launch_track_m<float, double>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream)
                       : launch_track_m<double, double>(p, B, M, X, wps, n_wp, wp_index, sm, goal, table, u_last, ret, ret_step, tX, tU, stream);
    if (e != hipSuccess || !p.dyn_obs || M == 0) return e;
    const unsigned blocks = (unsigned)((M + 255) / 256);
    if (f32) hipLaunchKernelGGL(advance_obstacle_table_kernel<float>, dim3(blocks), dim3(256), 0, stream, (float*)table, M, p.n_steps, p.qp.dt);
    else hipLaunchKernelGGL(advance_obstacle_table_kernel<double>, dim3(blocks), dim3(256), 0, stream, (double*)table, M, p.n_steps, p.qp.dt);
    return hipGetLastError();
}

}  // namespace sc
