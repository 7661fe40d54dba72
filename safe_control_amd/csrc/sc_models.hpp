// Per-model CBF row builders: the device-side equivalent of
//   robot.f(), robot.g(), robot.agent_barrier(obs)      (robots/robot.py:389-436)
// followed by the row assembly of CBFQP.solve_control_problem
//   (position_control/cbf_qp.py:155-183).
// One call produces one row  n0*u0 + n1*u1 + c >= 0  and the barrier value h.
#pragma once
#include "sc_math.hpp"
#include "../../include/safe_control_amd.h"

namespace sc {

// Controller constants converted to the compute type once per kernel.
template <typename T>
struct CbfConsts {
    T R;           // robot radius
    T a1, a2;      // alpha1 (or alpha), alpha2
    T g1, g2;      // a1+a2, a1*a2  (cbf_qp.py:180-181)
    T inv_dt, inv_dt2;
    T lo0, hi0, lo1, hi1;
    T inv_Lr;      // 1 / rear_ax_dist (KB family)
    T inv_mass;    // 1 / mass (Quad2D)
    int hard;
};

template <typename T>
__device__ __forceinline__ CbfConsts<T> make_consts(const sc_cbfqp_params& p) {
    CbfConsts<T> k;
    k.R = T(p.robot_radius);
    k.a1 = T(p.alpha1);
    k.a2 = T(p.alpha2);
    k.g1 = T(p.alpha1 + p.alpha2);
    k.g2 = T(p.alpha1 * p.alpha2);
    k.inv_dt = T(1.0 / p.dt);
    k.inv_dt2 = T(1.0 / (p.dt * p.dt));
    k.lo0 = T(p.u_min[0]); k.hi0 = T(p.u_max[0]);
    k.lo1 = T(p.u_min[1]); k.hi1 = T(p.u_max[1]);
    k.inv_Lr = T(p.rear_ax_dist > 0 ? 1.0 / p.rear_ax_dist : 0.0);
    k.inv_mass = T(p.mass > 0 ? 1.0 / p.mass : 1.0);
    k.hard = p.cbf_mode == SC_CBF_MODE_HARD;
    return k;
}

template <typename T>
struct Agent {
    T x, y, th, v;
    T c, s;        // cos(theta), sin(theta)
    T f0, f1;      // v cos, v sin  (f(x)[0:2]; f(x)[2:4] = 0)
};

template <typename T>
__device__ __forceinline__ Agent<T> make_agent(T x, T y, T th, T v) {
    Agent<T> a;
    a.x = x; a.y = y; a.th = th; a.v = v;
    sincos_(th, &a.s, &a.c);
    a.f0 = v * a.c;
    a.f1 = v * a.s;
    return a;
}

// Integrator models keep (vx, vy) where the unicycle keeps (theta, v): f(x)[0:2] = (vx, vy) for the double
// integrator (robots/double_integrator2D.py:46-59), zero for the single integrator (single_integrator2D.py:45-55).
template <typename T, int MODEL>
__device__ __forceinline__ Agent<T> make_agent_m(T x0, T x1, T x2, T x3, T x4 = T(0)) {
    if constexpr (MODEL == SC_MODEL_QUAD2D) {
        // robots/quad2D.py:46-58: f(x)[0:2] = (vx, vz); the heading only enters through g
        Agent<T> a;
        a.x = x0; a.y = x1; a.th = x2; a.v = T(0);
        sincos_(x2, &a.s, &a.c);
        a.f0 = x3; a.f1 = x4;
        return a;
    } else if constexpr (MODEL == SC_MODEL_SINGLE_INTEGRATOR2D || MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) {
        Agent<T> a;
        a.x = x0; a.y = x1; a.th = T(0); a.v = T(0); a.c = T(1); a.s = T(0);
        a.f0 = (MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) ? x2 : T(0);
        a.f1 = (MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) ? x3 : T(0);
        return a;
    } else {
        return make_agent<T>(x0, x1, x2, x3);
    }
}

// ---- rel-deg-2 distance barrier -------------------------------------------
// DU robots/dynamic_unicycle2D.py:136-146 (beta 1.01), KB robots/kinematic_bicycle2D.py:160-173 (beta 1.1)
template <typename T>
__device__ __forceinline__ void hocbf_circle(const Agent<T>& a, const T* o, T R, T beta,
                                             T& h, T& hdot, T (&dhd)[4]) {
    const T ex = a.x - o[0], ey = a.y - o[1];
    const T dmin = o[2] + R;
    h = (ex * ex + ey * ey) - beta * dmin * dmin;
    hdot = T(2) * (ex * a.f0 + ey * a.f1);
    dhd[0] = T(2) * a.f0;
    dhd[1] = T(2) * a.f1;
    dhd[2] = T(2) * (ex * (-a.f1) + ey * a.f0);
    dhd[3] = T(2) * (ex * a.c + ey * a.s);
}

// DU superellipsoid branch, robots/dynamic_unicycle2D.py:148-183
template <typename T>
__device__ __forceinline__ void hocbf_superellipsoid(const Agent<T>& a, const T* o, T R,
                                                     T& h, T& hdot, T (&dhd)[4]) {
    const T ea = o[2] + R, eb = o[3] + R, e = o[4];
    T st, ct;
    sincos_(o[5], &st, &ct);
    const T dx = a.x - o[0], dy = a.y - o[1];
    const T px = ct * dx + st * dy;
    const T py = -st * dx + ct * dy;
    T px_e2, py_e2, ea_e, eb_e;          // px^(e-2), py^(e-2), ea^e, eb^e
    const T er = rint_(e);
    if (er == e && e >= T(2) && e <= T(64)) {
        const int n = (int)er;
        px_e2 = powi_chain(px, n - 2);
        py_e2 = powi_chain(py, n - 2);
        ea_e = powi_chain(ea, n);
        eb_e = powi_chain(eb, n);
    } else {
        px_e2 = pow_(px, e - T(2));
        py_e2 = pow_(py, e - T(2));
        ea_e = pow_(ea, e);
        eb_e = pow_(eb, e);
    }
    const T px_e1 = px_e2 * px, py_e1 = py_e2 * py;
    h = px_e1 * px / ea_e + py_e1 * py / eb_e - T(1);
    const T gx = e * px_e1 / ea_e;
    const T gy = e * py_e1 / eb_e;
    const T dhx = gx * ct - gy * st;
    const T dhy = gx * st + gy * ct;
    hdot = dhx * a.f0 + dhy * a.f1;
    const T ca = e * (e - T(1)) / ea_e * px_e2;
    const T cb = e * (e - T(1)) / eb_e * py_e2;
    const T hxx = ca * ct * ct + cb * st * st;
    const T hxy = (ca - cb) * ct * st;
    const T hyy = ca * st * st + cb * ct * ct;
    dhd[0] = hxx * a.f0 + hxy * a.f1;
    dhd[1] = hxy * a.f0 + hyy * a.f1;
    dhd[2] = dhx * (-a.f1) + dhy * a.f0;
    dhd[3] = dhx * a.c + dhy * a.s;
}

// Out-of-line copy for kernels that unroll the row loop: superellipsoids are the rare,
// code-heavy case (sincos + pow chains + divisions); one shared body instead of K inlined
// copies keeps the unrolled kernels inside the instruction cache.  Everything by value.
template <typename T>
struct SuperOut { T h, hdot, d0, d1, d2, d3; };

template <typename T>
__device__ __attribute__((noinline)) SuperOut<T> hocbf_superellipsoid_outlined(
        T ax, T ay, T ac, T as, T af0, T af1, T o0, T o1, T o2, T o3, T o4, T o5, T R) {
    Agent<T> a;
    a.x = ax; a.y = ay; a.th = T(0); a.v = T(0); a.c = ac; a.s = as; a.f0 = af0; a.f1 = af1;
    const T o[7] = {o0, o1, o2, o3, o4, o5, T(1)};
    SuperOut<T> r;
    T dhd[4];
    hocbf_superellipsoid(a, o, R, r.h, r.hdot, dhd);
    r.d0 = dhd[0]; r.d1 = dhd[1]; r.d2 = dhd[2]; r.d3 = dhd[3];
    return r;
}

// ---- rel-deg-1 barriers with moving obstacles ------------------------------
// dynamic_env/kinematic_bicycle2D_c3bf.py:15-75
template <typename T>
__device__ __forceinline__ void c3bf(const Agent<T>& a, const T* o, T R, T& h, T (&dh)[4]) {
    const T ovx = o[3], ovy = o[4];
    const T ego = (o[2] + R) * T(1.0);
    const T px = o[0] - a.x, py = o[1] - a.y;
    const T vx = ovx - a.f0, vy = ovy - a.f1;
    const T pm = sqrt_(px * px + py * py);
    const T vm = sqrt_(vx * vx + vy * vy);
    const T eps = T(1e-6);
    const T sq = sqrt_(fmax_(pm * pm - ego * ego, eps));
    const T cphi = sq / (pm + eps);
    h = (px * vx + py * vy) + pm * vm * cphi;
    const T k = (sq + eps) / vm;
    dh[0] = -vx - vm * px / (sq + eps);
    dh[1] = -vy - vm * py / (sq + eps);
    dh[2] = a.f1 * px - a.f0 * py + k * (a.v * (ovx * a.s - ovy * a.c));
    dh[3] = -a.c * px - a.s * py + k * (a.v - (ovx * a.c + ovy * a.s));
}

// dynamic_env/kinematic_bicycle2D_dpcbf.py:16-84 (k_lambda 0.1, k_mu 0.5, s 1.05).
// cos/sin of rot = atan2(py, px) are taken algebraically (px/|p|, py/|p|).
template <typename T>
__device__ __forceinline__ void dpcbf(const Agent<T>& a, const T* o, T R, T& h, T (&dh)[4]) {
    const T kl = T(0.1), km = T(0.5), sm = T(1.05);
    const T ovx = o[3], ovy = o[4];
    const T ego = (o[2] + R) * sm;
    const T px = o[0] - a.x, py = o[1] - a.y;
    const T vx = ovx - a.f0, vy = ovy - a.f1;
    const T pm2 = px * px + py * py;
    const T pm = sqrt_(pm2);
    const T vm = sqrt_(vx * vx + vy * vy);
    const T cr = px / pm, sr = py / pm;
    const T vnx = cr * vx + sr * vy;
    const T vny = -sr * vx + cr * vy;
    const T sd = sqrt_(fmax_(pm2 - ego * ego, T(1e-6)));
    const T sfac = sqrt_(sm * sm - T(1)) / ego;
    const T lam = kl * sd / vm * sfac;
    const T mu = km * sd * sfac;
    h = vnx + lam * vny * vny + mu;
    const T sin_rt = sr * a.c - cr * a.s;     // sin(rot - theta)
    const T cos_rt = cr * a.c + sr * a.s;     // cos(rot - theta)
    const T vny2 = vny * vny;
    const T vm3 = vm * vm * vm;
    dh[0] = py * vny / pm2 - kl * px * vny2 / vm / sd
            - T(2) * kl * sd / vm * vny * py / pm2 * vnx - km * px / sd;
    dh[1] = -px * vny / pm2 - kl * py * vny2 / vm / sd
            + T(2) * kl * sd / vm * vny * px / pm2 * vnx - km * py / sd;
    dh[2] = -a.v * sin_rt - kl * sd * a.v * (ovx * a.s - ovy * a.c) * vny2 / vm3
            - T(2) * kl * sd * vny * a.v * cos_rt / vm;
    dh[3] = -cos_rt - kl * sd / vm3 * (a.v - ovx * a.c - ovy * a.s) * vny2
            - T(2) * kl * sd * vny * sin_rt / vm;
}

// One CBF row for obstacle `o` (7 values, compute type).  Returns false when
// the obstacle flag is invalid for the model (DU needs flag 0 or 1).
// CIRCLES_ONLY (DynamicUnicycle2D): the caller has established that every obstacle of the wave is a circle (flag 0): the
// superellipsoid branch -- and with OUTLINE_RARE the function call that costs the kernel 30 VGPRs and its SGPR spills -- is not emitted
template <typename T, int MODEL, bool OUTLINE_RARE = false, bool CIRCLES_ONLY = false>
__device__ __forceinline__ bool cbf_row(const Agent<T>& a, const T* o, const CbfConsts<T>& k,
                                        T& n0, T& n1, T& c, T& h) {
    if constexpr (MODEL == SC_MODEL_UNICYCLE2D) {
        // robots/unicycle2D.py:100-128 (rel-deg 1): h = |p - o|^2 - beta d_min^2 - sigma(s), s = (p - o) . heading,
        // sigma(s) = k2 (e^(k1-s) - 1) / (e^(k1-s) + 1) = k2 tanh((k1 - s) / 2)  (k1 = .5, k2 = 1.8; the tanh form has
        // no overflow for obstacles far behind the robot, where the reference returns NaN); the flag is not looked at.
        // g = [[c, 0], [s, 0], [0, 1]], f = 0  =>  A = (dh/dp . heading, dh/dtheta),  b = alpha h  (cbf_qp.py:155-165).
        const T ex = a.x - o[0], ey = a.y - o[1];
        const T dmin = o[2] + k.R;
        const T s = ex * a.c + ey * a.s;
        const T th = T(tanh(double(T(0.5) * (T(0.5) - s))));
        const T dsig = T(-0.9) * (T(1) - th * th);
        h = (ex * ex + ey * ey) - T(1.01) * dmin * dmin - T(1.8) * th;
        n0 = T(2) * s - dsig;                                      // (2 e - dsig heading) . heading
        n1 = -dsig * (-a.s * ex + a.c * ey);
        c = k.hard ? (h * k.inv_dt) : (k.a1 * h);
    } else if constexpr (MODEL == SC_MODEL_QUAD2D) {
        // robots/quad2D.py:166-177 (circle, no flag test) with g of :68-81: both thrusts enter identically,
        // A = dh_dot_dx g = [a, a],  a = (2 ex (-sin th) + 2 ez cos th) / m;  L_f = 2 |v|^2 - 2 g ez
        const T ex = a.x - o[0], ez = a.y - o[1];
        const T dmin = o[2] + k.R;
        h = (ex * ex + ez * ez) - T(1.01) * dmin * dmin;
        const T hdot = T(2) * (ex * a.f0 + ez * a.f1);
        n0 = (T(2) * ex * (-a.s) + T(2) * ez * a.c) * k.inv_mass;
        n1 = n0;
        const T Lf = T(2) * a.f0 * a.f0 + T(2) * a.f1 * a.f1 + T(2) * ez * T(-9.81);
        c = k.hard ? (h * k.inv_dt2 + T(2) * hdot * k.inv_dt + Lf) : (Lf + k.g1 * hdot + k.g2 * h);
    } else if constexpr (MODEL == SC_MODEL_SINGLE_INTEGRATOR2D || MODEL == SC_MODEL_DOUBLE_INTEGRATOR2D) {
        // SI robots/single_integrator2D.py:119-149 (rel-deg 1), DI robots/double_integrator2D.py:167-220 (rel-deg 2);
        // g is the identity on the actuated pair, so A = dh/dp for both.
        const T flag = o[6];
        T dhx, dhy, hxx = T(2), hxy = T(0), hyy = T(2);
        if (flag == T(0)) {
            const T ex = a.x - o[0], ey = a.y - o[1];
            const T dmin = o[2] + k.R;
            h = (ex * ex + ey * ey) - T(1.01) * dmin * dmin;
            dhx = T(2) * ex; dhy = T(2) * ey;
        } else if (flag == T(1)) {
            // reuse the unicycle superellipsoid with heading 0 and speed 1: dhd[3] = dh/dx, and the Hessian
            // entries come back through dhd[0..1] evaluated for the two unit velocities
            Agent<T> ux = a; ux.c = T(1); ux.s = T(0); ux.f0 = T(1); ux.f1 = T(0);
            Agent<T> uy = a; uy.c = T(0); uy.s = T(1); uy.f0 = T(0); uy.f1 = T(1);
            T hd, d[4];
            hocbf_superellipsoid(ux, o, k.R, h, hd, d);
            dhx = d[3]; hxx = d[0]; hxy = d[1];
            hocbf_superellipsoid(uy, o, k.R, h, hd, d);
            dhy = d[3]; hyy = d[1];
        } else {
            n0 = n1 = c = h = T(0);
            return false;
        }
        n0 = dhx; n1 = dhy;
        if constexpr (MODEL == SC_MODEL_SINGLE_INTEGRATOR2D) {
            c = k.hard ? (h * k.inv_dt) : (k.a1 * h);                     // f = 0: b = alpha h  (cbf_qp.py:158-165)
        } else {
            const T hdot = dhx * a.f0 + dhy * a.f1;
            const T Lf = (hxx * a.f0 + hxy * a.f1) * a.f0 + (hxy * a.f0 + hyy * a.f1) * a.f1;
            c = k.hard ? (h * k.inv_dt2 + T(2) * hdot * k.inv_dt + Lf) : (Lf + k.g1 * hdot + k.g2 * h);
        }
    } else if constexpr (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D || MODEL == SC_MODEL_KINEMATIC_BICYCLE2D) {
        T hdot, dhd[4];
        if constexpr (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) {
            const T flag = o[6];
            if (CIRCLES_ONLY || flag == T(0)) {
                hocbf_circle(a, o, k.R, T(1.01), h, hdot, dhd);
            } else if (flag == T(1)) {
                if constexpr (OUTLINE_RARE) {
                    const SuperOut<T> r = hocbf_superellipsoid_outlined<T>(a.x, a.y, a.c, a.s, a.f0, a.f1, o[0], o[1],
                                                                           o[2], o[3], o[4], o[5], k.R);
                    h = r.h; hdot = r.hdot; dhd[0] = r.d0; dhd[1] = r.d1; dhd[2] = r.d2; dhd[3] = r.d3;
                } else {
                    hocbf_superellipsoid(a, o, k.R, h, hdot, dhd);
                }
            } else {
                n0 = n1 = c = h = T(0);
                return false;
            }
            n0 = dhd[3];                 // g = [[0,0],[0,0],[0,1],[1,0]]  (dynamic_unicycle2D.py:64-73)
            n1 = dhd[2];
        } else {
            hocbf_circle(a, o, k.R, T(1.1), h, hdot, dhd);
            n0 = dhd[3];                 // g(x) of kinematic_bicycle2D.py:93-111
            n1 = -a.f1 * dhd[0] + a.f0 * dhd[1] + a.v * k.inv_Lr * dhd[2];
        }
        const T Lf = dhd[0] * a.f0 + dhd[1] * a.f1;
        c = k.hard ? (h * k.inv_dt2 + T(2) * hdot * k.inv_dt + Lf)    // cbf_qp.py:170-177
                   : (Lf + k.g1 * hdot + k.g2 * h);                   // cbf_qp.py:178-183
    } else {
        T dh[4];
        if constexpr (MODEL == SC_MODEL_KINEMATIC_BICYCLE2D_C3BF) c3bf(a, o, k.R, h, dh);
        else dpcbf(a, o, k.R, h, dh);
        n0 = dh[3];
        n1 = -a.f1 * dh[0] + a.f0 * dh[1] + a.v * k.inv_Lr * dh[2];
        const T Lf = dh[0] * a.f0 + dh[1] * a.f1;
        c = k.hard ? (h * k.inv_dt + Lf)                               // cbf_qp.py:158-161
                   : (Lf + k.a1 * h);                                  // cbf_qp.py:162-165
    }
    return true;
}

}  // namespace sc
