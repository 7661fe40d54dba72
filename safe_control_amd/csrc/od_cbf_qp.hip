// Optimal-decay CBF-QP, one problem per lane.
//   OptimalDecayCBFQP.solve_control_problem   position_control/optimal_decay_cbf_qp.py:131-158
// The QP has 4 variables (u0, u1, omega1, omega2), a diagonal Hessian diag(1, 1, p1, p2), ONE general
// row  a0 u0 + a1 u1 + e1 w1 + e2 w2 + b >= 0  and a box on (u0, u1).  Strictly convex => unique
// minimiser; it is found exactly by checking the KKT conditions of the 1 + 9 possible active sets
// (row inactive; row active with each of u0, u1 free / at its lower / at its upper bound).
#include <hip/hip_runtime.h>

#include "sc_models.hpp"

namespace sc {

template <typename TIO> struct odv2;
template <> struct odv2<float> { using type = float2; };
template <> struct odv2<double> { using type = double2; };

template <typename T>
struct OdSol { T u0, u1, w1, w2, cost; bool ok; };

template <typename TIO, typename TC, int MODEL>
__global__ __launch_bounds__(256) void odcbfqp_kernel(const sc_odcbfqp_params p, const long long B,
                                                      const TIO* __restrict__ X, const TIO* __restrict__ u_ref,
                                                      const TIO* __restrict__ obs, const int* __restrict__ has_obs,
                                                      TIO* __restrict__ u_out, TIO* __restrict__ omega_out,
                                                      int* __restrict__ status_out, TIO* __restrict__ h_out) {
    const long long agent = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (agent >= B) return;
    using V2 = typename odv2<TIO>::type;
    const V2 ur = reinterpret_cast<const V2*>(u_ref)[agent];
    const CbfConsts<TC> k = make_consts<TC>(p.qp);
    Agent<TC> ag;
    if constexpr (MODEL == SC_MODEL_QUAD2D) {                   // six states per row: [x, z, theta, vx, vz, theta_dot] (robots/quad2D.py:41-44)
        const TIO* r = X + agent * 6;
        ag = make_agent_m<TC, MODEL>(TC(r[0]), TC(r[1]), TC(r[2]), TC(r[3]), TC(r[4]));
    } else {
        const V2* Xv = reinterpret_cast<const V2*>(X) + agent * 2;
        const V2 xa = Xv[0], xb = Xv[1];
        ag = make_agent<TC>(TC(xa.x), TC(xa.y), TC(xb.x), TC(xb.y));
    }
    const TC r0 = TC(ur.x), r1 = TC(ur.y);
    const TC wr1 = TC(p.omega_ref[0]), wr2 = TC(p.omega_ref[1]);
    const TC p1 = TC(p.p_sb[0]), p2 = TC(p.p_sb[1]);
    constexpr bool REL2 = (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D || MODEL == SC_MODEL_KINEMATIC_BICYCLE2D || MODEL == SC_MODEL_QUAD2D);

    // ---- the row: A = dh g, b = dh f (optimal_decay_cbf_qp.py:138-146), e1, e2 -------------------
    TC a0 = 0, a1 = 0, b = 0, e1 = 0, e2 = 0, h = 0;
    bool bad = false;
    const bool present = has_obs ? (has_obs[agent] != 0) : true;
    if (present) {
        TC o[7];
#pragma unroll
        for (int f = 0; f < 7; ++f) o[f] = TC(obs[agent * 7 + f]);
        if constexpr (MODEL == SC_MODEL_QUAD2D) {
            // optimal_decay_cbf_qp.py:38-45,105-115,141-146 over robots/quad2D.py:166-177 (circle, no flag test) and g of :68-81: both
            // thrusts enter alike, A = dh_dot_dx g = [a, a], a = 2 (-ex sin th + ez cos th) / m; b = dh_dot_dx f = 2 |v|^2 - 2 g ez
            const TC ex = ag.x - o[0], ez = ag.y - o[1];
            const TC dmin = o[2] + k.R;
            h = (ex * ex + ez * ez) - TC(1.01) * dmin * dmin;
            const TC hdot = TC(2) * (ex * ag.f0 + ez * ag.f1);
            a0 = (TC(2) * ex * (-ag.s) + TC(2) * ez * ag.c) * k.inv_mass;
            a1 = a0;
            b = TC(2) * ag.f0 * ag.f0 + TC(2) * ag.f1 * ag.f1 + TC(2) * ez * TC(-9.81);
            e1 = k.g1 * hdot;
            e2 = k.g2 * h;
        } else if constexpr (REL2) {
            TC hdot, d[4];
            if constexpr (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D) {
                if (o[6] == TC(0)) hocbf_circle(ag, o, k.R, TC(1.01), h, hdot, d);
                else if (o[6] == TC(1)) hocbf_superellipsoid(ag, o, k.R, h, hdot, d);
                else { bad = true; h = hdot = 0; d[0] = d[1] = d[2] = d[3] = 0; }
                a0 = d[3]; a1 = d[2];
            } else {
                hocbf_circle(ag, o, k.R, TC(1.1), h, hdot, d);
                a0 = d[3];
                a1 = -ag.f1 * d[0] + ag.f0 * d[1] + ag.v * k.inv_Lr * d[2];
            }
            b = d[0] * ag.f0 + d[1] * ag.f1;
            e1 = k.g1 * hdot;                    // (alpha1 + alpha2) h_dot
            e2 = k.g2 * h;                       // alpha1 alpha2 h
        } else {
            TC d[4];
            if constexpr (MODEL == SC_MODEL_KINEMATIC_BICYCLE2D_C3BF) c3bf(ag, o, k.R, h, d);
            else dpcbf(ag, o, k.R, h, d);
            a0 = d[3];
            a1 = -ag.f1 * d[0] + ag.f0 * d[1] + ag.v * k.inv_Lr * d[2];
            b = d[0] * ag.f0 + d[1] * ag.f1;
            e1 = k.a1 * h;                       // alpha h
            e2 = TC(0);
        }
    }

    // ---- exact solve by KKT enumeration --------------------------------------------------------------
    const TC tol = num<TC>::tol_feas();
    const TC c0 = fmin_(fmax_(r0, k.lo0), k.hi0), c1 = fmin_(fmax_(r1, k.lo1), k.hi1);
    OdSol<TC> best;
    best.ok = false; best.cost = num<TC>::inf(); best.u0 = c0; best.u1 = c1; best.w1 = wr1; best.w2 = wr2;
    const TC rowscale = fmax_(TC(1), fabs_(a0 * c0) + fabs_(a1 * c1) + fabs_(e1 * wr1) + fabs_(e2 * wr2) + fabs_(b));
    // (1) row inactive
    {
        const TC s = a0 * c0 + a1 * c1 + e1 * wr1 + e2 * wr2 + b;
        if (s >= -tol * rowscale) {
            best.ok = true;
            best.cost = (c0 - r0) * (c0 - r0) + (c1 - r1) * (c1 - r1);
        }
    }
    // (2) row active, u0 / u1 each free (0), at lo (1) or at hi (2)
    const TC iw1 = e1 * e1 / p1, iw2 = REL2 ? e2 * e2 / p2 : TC(0);
#pragma unroll
    for (int q0 = 0; q0 < 3; ++q0) {
#pragma unroll
        for (int q1 = 0; q1 < 3; ++q1) {
            const TC f0 = q0 == 1 ? k.lo0 : k.hi0, f1 = q1 == 1 ? k.lo1 : k.hi1;
            const TC x0 = q0 == 0 ? r0 : f0, x1 = q1 == 0 ? r1 : f1;            // fixed at bound, else reference
            const TC s = a0 * x0 + a1 * x1 + e1 * wr1 + e2 * wr2 + b;             // row value at that point
            const TC den = (q0 == 0 ? a0 * a0 : TC(0)) + (q1 == 0 ? a1 * a1 : TC(0)) + iw1 + iw2;
            // stationarity: 2 D (x - r) = lam a on the free variables, row = 0  =>  lam = -2 s / den
            const TC lam = TC(-2) * s / den;
            const TC u0 = q0 == 0 ? r0 + TC(0.5) * lam * a0 : f0;
            const TC u1 = q1 == 0 ? r1 + TC(0.5) * lam * a1 : f1;
            const TC w1 = wr1 + TC(0.5) * lam * e1 / p1;
            const TC w2 = REL2 ? wr2 + TC(0.5) * lam * e2 / p2 : wr2;
            bool ok = (den > TC(0)) && (lam >= -tol);
            const TC btol = tol * fmax_(TC(1), fmax_(fabs_(k.hi0), fabs_(k.hi1)));
            // free inputs inside the box, fixed inputs pushed against their bound (multiplier >= 0)
            if (q0 == 0) ok = ok && (u0 >= k.lo0 - btol) && (u0 <= k.hi0 + btol);
            if (q0 == 1) ok = ok && (TC(2) * (k.lo0 - r0) - lam * a0 >= -tol);
            if (q0 == 2) ok = ok && (lam * a0 - TC(2) * (k.hi0 - r0) >= -tol);
            if (q1 == 0) ok = ok && (u1 >= k.lo1 - btol) && (u1 <= k.hi1 + btol);
            if (q1 == 1) ok = ok && (TC(2) * (k.lo1 - r1) - lam * a1 >= -tol);
            if (q1 == 2) ok = ok && (lam * a1 - TC(2) * (k.hi1 - r1) >= -tol);
            const TC cost = (u0 - r0) * (u0 - r0) + (u1 - r1) * (u1 - r1) + p1 * (w1 - wr1) * (w1 - wr1) +
                            (REL2 ? p2 * (w2 - wr2) * (w2 - wr2) : TC(0));
            if (ok && cost < best.cost) {
                best.ok = true; best.cost = cost; best.u0 = u0; best.u1 = u1; best.w1 = w1; best.w2 = w2;
            }
        }
    }
    const bool finite = finite_(a0 + a1 + b + e1 + e2 + r0 + r1);
    int st = (best.ok && finite) ? SC_STATUS_OPTIMAL : SC_STATUS_INFEASIBLE;
    if (bad) st = SC_STATUS_BAD_OBSTACLE;
    TC u0 = fmin_(fmax_(best.u0, k.lo0), k.hi0), u1 = fmin_(fmax_(best.u1, k.lo1), k.hi1);
    TC w1 = best.w1, w2 = best.w2;
    if (st != SC_STATUS_OPTIMAL) { u0 = u1 = w1 = w2 = num<TC>::nan(); }
    V2 uo; uo.x = TIO(u0); uo.y = TIO(u1);
    reinterpret_cast<V2*>(u_out)[agent] = uo;
    V2 wo; wo.x = TIO(w1); wo.y = TIO(w2);
    reinterpret_cast<V2*>(omega_out)[agent] = wo;
    status_out[agent] = st;
    if (h_out) h_out[agent] = present ? TIO(h) : TIO(0);
}

template <typename TIO, typename TC>
static hipError_t od_launch_model(const sc_odcbfqp_params& p, long long B, const void* X, const void* u_ref, const void* obs,
                                  const int* has_obs, void* u_out, void* w_out, int* status, void* h_out, hipStream_t stream) {
    const unsigned threads = 256, blocks = (unsigned)((B + threads - 1) / threads);
#define SC_OD(M)                                                                                                   \
    hipLaunchKernelGGL((odcbfqp_kernel<TIO, TC, M>), dim3(blocks), dim3(threads), 0, stream, p, B, (const TIO*)X, \
                       (const TIO*)u_ref, (const TIO*)obs, has_obs, (TIO*)u_out, (TIO*)w_out, status, (TIO*)h_out)
    switch (p.qp.model_id) {
        case SC_MODEL_DYNAMIC_UNICYCLE2D: SC_OD(SC_MODEL_DYNAMIC_UNICYCLE2D); break;
        case SC_MODEL_KINEMATIC_BICYCLE2D: SC_OD(SC_MODEL_KINEMATIC_BICYCLE2D); break;
        case SC_MODEL_KINEMATIC_BICYCLE2D_C3BF: SC_OD(SC_MODEL_KINEMATIC_BICYCLE2D_C3BF); break;
        case SC_MODEL_QUAD2D: SC_OD(SC_MODEL_QUAD2D); break;
        default: SC_OD(SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF); break;
    }
#undef SC_OD
    return hipGetLastError();
}

hipError_t odcbfqp_launch(const sc_odcbfqp_params& p, long long B, const void* X, const void* u_ref, const void* obs,
                          const int* has_obs, void* u_out, void* w_out, int* status, void* h_out, hipStream_t stream) {
    if (p.qp.io_dtype == SC_DTYPE_F32) {
        if (p.qp.compute_dtype == SC_DTYPE_F32)
            return od_launch_model<float, float>(p, B, X, u_ref, obs, has_obs, u_out, w_out, status, h_out, stream);
        return od_launch_model<float, double>(p, B, X, u_ref, obs, has_obs, u_out, w_out, status, h_out, stream);
    }
    return od_launch_model<double, double>(p, B, X, u_ref, obs, has_obs, u_out, w_out, status, h_out, stream);
}

}  // namespace sc
