// control_step around the solve for Quad2D and Quad3D (SURVEY 8f-1 over the 8f-3 models): the two kernels that bracket the
// MPC-CBF launch of a batched closed loop, one agent per lane.
//   quadtrack_select_kernel = tracking.py:559-609: state machine / update_goal (:497-535; Quad2D skips 'rotate', Quad3D goals
//     are 3-D), the K nearest obstacles (get_nearest_unpassed_obs :345-403 with angle_unpassed = 2 pi: every obstacle counts),
//     the reference input (nominal_input / stop / rotate_to of robots/quad2D.py:88-154, robots/quad3D.py:153-257)
//   quadtrack_apply_kernel  = tracking.py:627-668: collision tests (:445-495), robot.step (quad2D.py:83-86 Euler + wrap;
//     quad3D.py:113-151 RK4 of the linear model + three wraps), return codes.
// oracle/tracking_quad.py is the float64 statement (pinned on the reference's own run, tests/golden/closed_loop_quads.npz).
// Arithmetic is f64; the caller's arrays are f32 or f64 (run-time switch: a handful of loads / stores per agent).
#include <hip/hip_runtime.h>

#include "sc_math.hpp"
#include "../../include/safe_control_amd.h"
#include "mpc_vtol_solver.hpp"

namespace sc {
namespace {

struct QtP {
    int q3, vt, nx, nu, ng, K, enable_rotation;
    double dt, reached, rot_thr, R, mass, inertia, f_min, f_max, Ix, Iy, Iz, L, nu_c, u_min, u_max, pitch_limit;
};

__device__ __forceinline__ QtP make_qtp(const sc_quadtrack_params& p) {
    QtP P;
    P.q3 = p.model == SC_QUADTRACK_QUAD3D; P.vt = p.model == SC_QUADTRACK_VTOL2D;
    P.nx = P.q3 ? 12 : 6; P.nu = (P.q3 || P.vt) ? 4 : 2; P.ng = P.q3 ? 3 : 2; P.pitch_limit = p.pitch_limit;
    P.K = p.num_constraints; P.enable_rotation = p.enable_rotation;
    P.dt = p.dt; P.reached = p.reached_threshold; P.rot_thr = p.rotation_threshold; P.R = p.robot_radius;
    P.mass = p.mass; P.inertia = p.inertia; P.f_min = p.f_min; P.f_max = p.f_max;
    P.Ix = p.Ix; P.Iy = p.Iy; P.Iz = p.Iz; P.L = p.L; P.nu_c = p.nu; P.u_min = p.u_min; P.u_max = p.u_max;
    return P;
}

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// Quad2D.nominal_input (quad2D.py:88-143) with its default gains; stop() = nominal_input towards the own position (:145-154)
__device__ __forceinline__ void q2_nominal(const double* X, double gx, double gz, const QtP& P, double* u) {
    const double g = 9.81;
    const double ax = 3.0 * (gx - X[0]) + 0.5 * (-X[3]);
    const double az = 0.1 * (gz - X[1]) + 0.5 * (-X[4]) + g;
    const double T = P.mass * sqrt(ax * ax + az * az);
    const double th_d = -atan2(ax, az);
    double e = th_d - X[2];
    e = atan2(sin(e), cos(e));
    const double tau = clipd(0.05 * e + 0.05 * (-X[5]), -1.0, 1.0);
    u[0] = clipd((T + tau / P.R) / 2.0, P.f_min, P.f_max);
    u[1] = clipd((T - tau / P.R) / 2.0, P.f_min, P.f_max);
}

// motor forces from a wrench (F, tau_y, tau_x, tau_z): inverse of B2 (quad3D.py:84-89; the reference takes pinv(B2), B2 is
// square and regular), then the clip to [u_min, u_max]
__device__ __forceinline__ void q3_alloc(double F, double ty, double tx, double tz, const QtP& P, double* u) {
    const double a = 0.25 * (F + tz / P.nu_c), b = 0.25 * (F - tz / P.nu_c), cx = tx / (2.0 * P.L), cy = ty / (2.0 * P.L);
    u[0] = clipd(a + cx, P.u_min, P.u_max); u[1] = clipd(b + cy, P.u_min, P.u_max);
    u[2] = clipd(a - cx, P.u_min, P.u_max); u[3] = clipd(b - cy, P.u_min, P.u_max);
}
// Quad3D.nominal_input (quad3D.py:153-199: k_p 1, k_d 2, k_ang 5), stop (:201-228, k 1), rotate_to (:236-257, k 2)
__device__ __forceinline__ void q3_nominal(const double* X, double gx, double gy, double gz, const QtP& P, double* u) {
    const double G = 9.8;
    const double ax = 1.0 * (gx - X[0]) + 2.0 * (-X[6]), ay = 1.0 * (gy - X[1]) + 2.0 * (-X[7]), az = 1.0 * (gz - X[2]) + 2.0 * (-X[8]);
    const double th_d = ax / G, ph_d = -ay / G, F = P.mass * az;
    q3_alloc(F, P.Iy * (5.0 * (th_d - X[3]) + 2.0 * (-X[9])), P.Ix * (5.0 * (ph_d - X[4]) + 2.0 * (-X[10])),
             P.Iz * (5.0 * (0.0 - X[5]) + 2.0 * (-X[11])), P, u);
}
__device__ __forceinline__ void q3_stop(const double* X, const QtP& P, double* u) {
    const double G = 9.8, k = 1.0;
    const double th_d = (-k * X[6]) / G, ph_d = -(-k * X[7]) / G, F = P.mass * (-k * X[8]);
    q3_alloc(F, P.Iy * k * (th_d - X[3] - X[9] / k), P.Ix * k * (ph_d - X[4] - X[10] / k), P.Iz * k * (0.0 - X[5] - X[11] / k), P, u);
}
__device__ __forceinline__ void q3_rotate_to(const double* X, double ang, const QtP& P, double* u) {
    const double k = 2.0;
    q3_alloc(P.mass * 9.8, P.Iy * k * (0.0 - X[3] - X[9] / k), P.Ix * k * (0.0 - X[4] - X[10] / k), P.Iz * k * (ang - X[5] - X[11] / k), P, u);
}

// x' = A x + B u of quad3D.py:70-98
__device__ __forceinline__ void q3_rhs(const double* x, const double* w, const QtP& P, double* k) {
#pragma unroll
    for (int i = 0; i < 6; ++i) k[i] = x[6 + i];
    k[6] = 9.8 * x[3]; k[7] = -9.8 * x[4]; k[8] = w[0] / P.mass;
    k[9] = w[1] / P.Iy; k[10] = w[2] / P.Ix; k[11] = w[3] / P.Iz;
}

__device__ __forceinline__ bool qt_collides(double x, double y, const double* table, int M, double R) {
    bool hit = false;
    for (int m = 0; m < M; ++m) {                          // tracking.py:445-495 (circle; superellipsoid when flag 1 and e >= 2)
        const double* o = table + 7 * m;
        const bool superell = (fabs(o[6] - 1.0) <= 1e-8 + 1e-5) && (o[4] >= 2.0);
        if (!superell) {
            const double dx = x - o[0], dy = y - o[1];
            hit |= sqrt(dx * dx + dy * dy) < o[2] + R;
        } else {
            double st, ct;
            sincos(o[5], &st, &ct);
            const double px = ct * (x - o[0]) + st * (y - o[1]), py = -st * (x - o[0]) + ct * (y - o[1]);
            hit |= pow(px / (o[2] + R), o[4]) + pow(py / (o[3] + R), o[4]) - 1.0 <= 0.0;
        }
    }
    return hit;
}

template <int KMAX>
__global__ __launch_bounds__(64) void quadtrack_select_kernel(
        const sc_quadtrack_params p, const long long B, const int M, const void* __restrict__ X, const void* __restrict__ waypoints,
        const int* __restrict__ n_wp, int* __restrict__ wp_index, int* __restrict__ state_machine, void* __restrict__ goal,
        const void* __restrict__ obs_table, const int* __restrict__ ret_in, void* __restrict__ obs_out, void* __restrict__ goal_out,
        void* __restrict__ u_ref_out, int* __restrict__ track_out) {
    extern __shared__ __attribute__((aligned(16))) double qt_table[];                 // [M][7]
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };
    const int lane = threadIdx.x;
    const long long agent = (long long)blockIdx.x * 64 + lane;
    const bool active = agent < B;
    const long long ag = active ? agent : 0;
    for (int e = lane; e < M * 7; e += 64) qt_table[e] = ld(obs_table, e);
    __syncthreads();
    const QtP P = make_qtp(p);
    double Xs[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Xs[i] = i < P.nx ? ld(X, ag * P.nx + i) : 0.0;
    const double x = Xs[0], y = Xs[1], yaw = P.q3 ? Xs[5] : Xs[2];
    int wp = wp_index[ag], sm = state_machine[ag];
    double gx = ld(goal, ag * 4 + 0), gy = ld(goal, ag * 4 + 1), gz = ld(goal, ag * 4 + 2);
    bool gvalid = ld(goal, ag * 4 + 3) != 0.0;
    const bool run = active && ret_in[ag] == 0;
    const int W = p.max_waypoints;
    const size_t wbase = p.waypoints_shared ? 0 : (size_t)ag * W * 3;
    const int nw = n_wp[p.waypoints_shared ? 0 : ag];
    auto wpc = [&](int i, int c) { return ld(waypoints, wbase + 3 * (size_t)i + c); };
    auto update_goal = [&]() {                                                        // tracking.py:497-535
        if (sm == SC_SM_ROTATE) {
            const int i = wp < nw ? wp : nw - 1;
            const double goal_angle = atan2(wpc(i, 1) - y, wpc(i, 0) - x);
            if (!P.q3) sm = SC_SM_TRACK;                                              // Quad2D skips 'rotate' (:512-513)
            if (!P.enable_rotation) sm = SC_SM_TRACK;
            if (fabs(yaw - goal_angle) > P.rot_thr) { gx = wpc(i, 0); gy = wpc(i, 1); gz = wpc(i, 2); gvalid = true; return; }
            sm = SC_SM_TRACK;
        }
        if (wp >= nw) { gvalid = false; return; }
        {
            const double dx = x - wpc(wp, 0), dy = y - wpc(wp, 1);                    // goal_reached: planar distance (:264-269)
            if (sqrt(dx * dx + dy * dy) < P.reached) {
                wp += 1;
                if (wp >= nw) { sm = SC_SM_IDLE; gvalid = false; return; }
            }
        }
        gx = wpc(wp, 0); gy = wpc(wp, 1); gz = wpc(wp, 2); gvalid = true;
    };
    if (run) {
        if (sm == SC_SM_STOP) {
            const bool stopped = P.q3 ? (sqrt(Xs[6] * Xs[6] + Xs[7] * Xs[7] + Xs[8] * Xs[8]) < 0.05 &&
                                         sqrt(Xs[9] * Xs[9] + Xs[10] * Xs[10] + Xs[11] * Xs[11]) < 0.05)
                                      : (sqrt(Xs[3] * Xs[3] + Xs[4] * Xs[4]) < 0.05);
            if (stopped) {
                sm = P.enable_rotation ? SC_SM_ROTATE : SC_SM_TRACK;
                update_goal();
            }
        } else {
            update_goal();
        }
    }
    // the K nearest obstacle centres, ties by index (stable sorted insertion)
    double sd[KMAX];
    int si[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { sd[j] = __builtin_huge_val(); si[j] = -1; }
    // VTOL2D (like the unicycles, tracking.py:354-355): obstacles inside the 1.2 pi cone about the heading -- the pitch angle -- count as
    // unpassed and are preferred; with none in the cone the nearest of all are taken (:389-394)
    bool any_front = false;
    if (P.vt)
        for (int m = 0; m < M; ++m)
            any_front |= fabs(angle_normalize(atan2(qt_table[7 * m + 1] - y, qt_table[7 * m] - x) - yaw)) <= 0.6 * 3.141592653589793;
    for (int m = 0; m < M; ++m) {
        const double dx = qt_table[7 * m] - x, dy = qt_table[7 * m + 1] - y;
        if (P.vt && any_front && !(fabs(angle_normalize(atan2(dy, dx) - yaw)) <= 0.6 * 3.141592653589793)) continue;
        double cd = sqrt(dx * dx + dy * dy);
        int ci = m;
        bool moved = false;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const bool sw = moved || (cd < sd[j]);
            moved = sw;
            const double td = sd[j]; const int ti = si[j];
            sd[j] = sw ? cd : td; si[j] = sw ? ci : ti;
            cd = sw ? td : cd; ci = sw ? ti : ci;
        }
    }
    double ur[4] = {0, 0, 0, 0};
    if (P.vt) {
        // nominal_input, stop, rotate_to of vtol2D.py:459-465 are not implemented there: zeros
    } else if (sm == SC_SM_ROTATE) {
        const double ga = atan2(gy - y, gx - x);
        if (P.q3) q3_rotate_to(Xs, ga, P, ur);
        else { ur[0] = 0.0; ur[1] = 2.0 * angle_normalize(ga - Xs[2]); }             // quad2D.py:160-164 (never reached: Quad2D skips 'rotate')
    } else if (!gvalid) {
        if (P.q3) q3_stop(Xs, P, ur); else q2_nominal(Xs, Xs[0], Xs[1], P, ur);
    } else {
        if (P.q3) q3_nominal(Xs, gx, gy, gz, P, ur); else q2_nominal(Xs, gx, gy, P, ur);
    }
    if (active) {
        wp_index[agent] = wp; state_machine[agent] = sm;
        st(goal, agent * 4 + 0, gx); st(goal, agent * 4 + 1, gy); st(goal, agent * 4 + 2, gz); st(goal, agent * 4 + 3, gvalid ? 1.0 : 0.0);
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            if (j >= P.K) break;
            const bool have = si[j] >= 0;
            const double* orow = qt_table + 7 * (have ? si[j] : 0);
#pragma unroll
            for (int f = 0; f < 7; ++f) st(obs_out, ((size_t)agent * P.K + j) * 7 + f, have ? orow[f] : (f < 2 ? 1000.0 : 0.0));   // mpc_cbf.py:343,360
        }
        st(goal_out, agent * P.ng + 0, gvalid ? gx : x); st(goal_out, agent * P.ng + 1, gvalid ? gy : y);
        if (P.q3) st(goal_out, agent * 3 + 2, gvalid ? gz : Xs[2]);
        for (int i = 0; i < P.nu; ++i) st(u_ref_out, agent * P.nu + i, ur[i]);
        track_out[agent] = (run && sm == SC_SM_TRACK && gvalid) ? 1 : 0;
    }
}

__global__ __launch_bounds__(64) void quadtrack_apply_kernel(
        const sc_quadtrack_params p, const long long B, const int M, const int step_index, void* __restrict__ X,
        const int* __restrict__ state_machine, const void* __restrict__ goal, const void* __restrict__ obs_table,
        const void* __restrict__ u, void* __restrict__ u_last, int* __restrict__ ret_out, int* __restrict__ ret_step) {
    extern __shared__ __attribute__((aligned(16))) double qt_table[];
    const bool io32 = p.io_dtype == SC_DTYPE_F32;
    auto ld = [io32](const void* a, size_t i) { return io32 ? (double)((const float*)a)[i] : ((const double*)a)[i]; };
    auto st = [io32](void* a, size_t i, double v) { if (io32) ((float*)a)[i] = (float)v; else ((double*)a)[i] = v; };
    const int lane = threadIdx.x;
    const long long agent = (long long)blockIdx.x * 64 + lane;
    const bool active = agent < B;
    const long long ag = active ? agent : 0;
    for (int e = lane; e < M * 7; e += 64) qt_table[e] = ld(obs_table, e);
    __syncthreads();
    const QtP P = make_qtp(p);
    double Xs[12], Xn[12], U[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 12; ++i) Xs[i] = i < P.nx ? ld(X, ag * P.nx + i) : 0.0;
    for (int i = 0; i < P.nu; ++i) U[i] = ld(u, ag * P.nu + i);
    const int sm = state_machine[ag];
    const bool gvalid = ld(goal, ag * 4 + 3) != 0.0;
    const bool run = active && ret_out[ag] == 0;
    // VTOL2D: below the ground or past the pitch limit counts as a collision (tracking.py:490-495)
    auto vt_fail = [&](const double* Xc) { return P.vt && (Xc[1] < 0.0 || fabs(Xc[2]) > P.pitch_limit); };
    const bool pre_fail = qt_collides(Xs[0], Xs[1], qt_table, M, P.R) || vt_fail(Xs);
    if (P.q3) {                                                                        // quad3D.py:113-151
        const double w[4] = {U[0] + U[1] + U[2] + U[3], P.L * (U[1] - U[3]), P.L * (U[0] - U[2]), P.nu_c * (U[0] - U[1] + U[2] - U[3])};
        double k1[12], k2[12], k3[12], k4[12], t[12];
        q3_rhs(Xs, w, P, k1);
#pragma unroll
        for (int i = 0; i < 12; ++i) t[i] = Xs[i] + P.dt / 2 * k1[i];
        q3_rhs(t, w, P, k2);
#pragma unroll
        for (int i = 0; i < 12; ++i) t[i] = Xs[i] + P.dt / 2 * k2[i];
        q3_rhs(t, w, P, k3);
#pragma unroll
        for (int i = 0; i < 12; ++i) t[i] = Xs[i] + P.dt * k3[i];
        q3_rhs(t, w, P, k4);
#pragma unroll
        for (int i = 0; i < 12; ++i) Xn[i] = Xs[i] + P.dt / 6 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        Xn[3] = angle_normalize(Xn[3]); Xn[4] = angle_normalize(Xn[4]); Xn[5] = angle_normalize(Xn[5]);
    } else if (P.vt) {                                                                 // vtol2D.py:299-321: Euler + pitch wrap
        vtol::Params VP;
        vtol::set_airframe(VP, p.airframe);
        double acc[3], gc[4][3];
        vtol::accel<double>(VP, Xs[2], Xs[3], Xs[4], U, acc, gc);
        Xn[0] = Xs[0] + Xs[3] * P.dt; Xn[1] = Xs[1] + Xs[4] * P.dt; Xn[2] = angle_normalize(Xs[2] + Xs[5] * P.dt);
        Xn[3] = Xs[3] + acc[0] * P.dt; Xn[4] = Xs[4] + acc[1] * P.dt; Xn[5] = Xs[5] + acc[2] * P.dt;
#pragma unroll
        for (int i = 6; i < 12; ++i) Xn[i] = 0.0;
    } else {                                                                           // quad2D.py:46-86
        double s, c;
        sincos(Xs[2], &s, &c);
        const double T = U[0] + U[1];
        Xn[0] = Xs[0] + Xs[3] * P.dt; Xn[1] = Xs[1] + Xs[4] * P.dt; Xn[2] = angle_normalize(Xs[2] + Xs[5] * P.dt);
        Xn[3] = Xs[3] + (-s / P.mass * T) * P.dt;
        Xn[4] = Xs[4] + (-9.81 + c / P.mass * T) * P.dt;
        Xn[5] = Xs[5] + (P.R / P.inertia * (U[0] - U[1])) * P.dt;
#pragma unroll
        for (int i = 6; i < 12; ++i) Xn[i] = 0.0;
    }
    int code;
    if (pre_fail) code = -2;
    else if (qt_collides(Xn[0], Xn[1], qt_table, M, P.R) || vt_fail(Xn)) code = -2;
    else code = (!gvalid && sm != SC_SM_STOP) ? -1 : 0;
    if (run) {
        if (!pre_fail) {
            for (int i = 0; i < P.nx; ++i) st(X, agent * P.nx + i, Xn[i]);
            for (int i = 0; i < P.nu; ++i) st(u_last, agent * P.nu + i, U[i]);
        }
        if (code != 0) { ret_out[agent] = code; ret_step[agent] = step_index; }
    }
}

}  // namespace

hipError_t quadtrack_select_launch(const sc_quadtrack_params& p, long long B, int M, const void* X, const void* wps, const int* n_wp,
                                   int* wp_index, int* sm, void* goal, const void* obs_table, const int* ret, void* obs_out,
                                   void* goal_out, void* u_ref_out, int* track_out, hipStream_t stream) {
    const unsigned blocks = (unsigned)((B + 63) / 64);
    const size_t lds = (size_t)(M > 0 ? M : 1) * 7 * sizeof(double);
    if (p.num_constraints <= 8)
        hipLaunchKernelGGL(quadtrack_select_kernel<8>, dim3(blocks), dim3(64), lds, stream, p, B, M, X, wps, n_wp, wp_index, sm, goal, obs_table,
                           ret, obs_out, goal_out, u_ref_out, track_out);
    else
        hipLaunchKernelGGL(quadtrack_select_kernel<16>, dim3(blocks), dim3(64), lds, stream, p, B, M, X, wps, n_wp, wp_index, sm, goal, obs_table,
                           ret, obs_out, goal_out, u_ref_out, track_out);
    return hipGetLastError();
}

hipError_t quadtrack_apply_launch(const sc_quadtrack_params& p, long long B, int M, int step_index, void* X, const int* sm, const void* goal,
                                  const void* obs_table, const void* u, void* u_last, int* ret, int* ret_step, hipStream_t stream) {
    const unsigned blocks = (unsigned)((B + 63) / 64);
    const size_t lds = (size_t)(M > 0 ? M : 1) * 7 * sizeof(double);
    hipLaunchKernelGGL(quadtrack_apply_kernel, dim3(blocks), dim3(64), lds, stream, p, B, M, step_index, X, sm, goal, obs_table, u, u_last,
                       ret, ret_step);
    return hipGetLastError();
}

}  // namespace sc
