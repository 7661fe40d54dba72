// Kernel 13: MPC-CBF as do-mpc poses it -- multiple shooting under IPOPT's filter interior point, one NLP per wavefront, four
// lanes per stage -- for DynamicUnicycle2D, Unicycle2D, SingleIntegrator2D, DoubleIntegrator2D and KinematicBicycle2D (template parameter MODEL: dums::M_*).  The algorithm and what each lane holds: mpc_du_ms_solver.hpp (plain C++ over a context; the same code
// runs on the host, one thread per lane, in tools/du_ms_host.cpp).  This unit supplies the device context -- LDS through an address-space
// pointer, __syncthreads, the DPP wave reductions of mpc_ipm_common.hpp -- the kernel and its launcher.
//
// LDS per problem at N = 10, K = 8: stage blocks, gains, value-function rows, exchange vectors, two filters, and the per-row state (slacks,
// multipliers, their steps, the restoration's n / p: 16 slots x K rows x N stages): 19.9 KB, eight problems per CU (two waves per SIMD).  HBM per solve: X 4 + u_prev 2 + goal 2 + K x 7 obstacle values in, u 2 + status +
// iterations out (272 B at K = 8, f32 storage).
#include "mpc_du_ms_dev.hpp"

namespace sc {

size_t mpcdu_ms_order_bytes(long long B) { return (size_t)(B + 4) * sizeof(int); }
size_t mpcdu_ms_lds_bytes(int horizon, int K, int model_id, int se) {
    return (size_t)dums::Lds(horizon, K, model_id == SC_MODEL_KINEMATIC_BICYCLE2D || model_id == SC_MODEL_UNICYCLE2D || model_id == SC_MODEL_SINGLE_INTEGRATOR2D, se != 0).total * sizeof(double);
}

hipError_t mpcdu_ms_se_launch(const dums::Params& P, const sc_mpccbf_params& p, const sc_ipopt_params& O, long long B, const void* X, const void* u_prev, const void* goal,
                              const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out, hipStream_t stream);      // mpc_du_ms_se.hip

hipError_t mpcdu_ms_launch(const sc_mpccbf_params& p, const sc_ipopt_params& O, long long B, int K, const void* X, const void* u_prev, const void* goal,
                           const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out, void* order_ws, hipStream_t stream) {
    dums::Params P;
    P.N = p.horizon; P.K = K; P.dt = p.dt;
    for (int i = 0; i < 4; ++i) P.Q[i] = p.Q[i];
    for (int j = 0; j < 2; ++j) { P.R[j] = p.R[j]; P.u_lo[j] = -p.u_max[j]; P.u_hi[j] = p.u_max[j]; }
    P.alpha1 = p.alpha1; P.alpha2 = p.alpha2; P.beta = p.beta; P.radius = p.robot_radius; P.v_max = p.v_max;
    const bool f64 = p.io_dtype == SC_DTYPE_F64;
    if (p.superellipsoid_rows) {                                              // rows that may be superellipsoids: the SE instantiations (their own translation unit)
        if (p.model_id == SC_MODEL_DOUBLE_INTEGRATOR2D) for (int j = 0; j < 2; ++j) { P.R[j] = p.R[1 - j]; P.u_lo[j] = -p.u_max[1 - j]; P.u_hi[j] = p.u_max[1 - j]; }
        return mpcdu_ms_se_launch(P, p, O, B, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, stream);
    }
    if (p.model_id == SC_MODEL_DOUBLE_INTEGRATOR2D) {                          // inputs held as (ay, ax): mpc_du_ms_solver.hpp, M_DI
        for (int j = 0; j < 2; ++j) { P.R[j] = p.R[1 - j]; P.u_lo[j] = -p.u_max[1 - j]; P.u_hi[j] = p.u_max[1 - j]; }
        if (f64) return dums::launch_t<double, dums::M_DI>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
        return dums::launch_t<float, dums::M_DI>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
    }
    if (p.model_id == SC_MODEL_SINGLE_INTEGRATOR2D) {
        P.Q[2] = 0.0; P.Q[3] = 0.0;
        if (f64) return dums::launch_t<double, dums::M_SI>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
        return dums::launch_t<float, dums::M_SI>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
    }
    if (p.model_id == SC_MODEL_UNICYCLE2D) {
        P.Q[3] = 0.0;
        if (f64) return dums::launch_t<double, dums::M_UNI>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
        return dums::launch_t<float, dums::M_UNI>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
    }
    if (p.model_id == SC_MODEL_KINEMATIC_BICYCLE2D) {
        P.v_min = p.v_min; P.inv_Lr = 1.0 / p.rear_ax_dist;
        if (f64) return dums::launch_t<double, dums::M_KB>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
        return dums::launch_t<float, dums::M_KB>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
    }
    if (f64) return dums::launch_t<double, dums::M_DU>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
    return dums::launch_t<float, dums::M_DU>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, order_ws, stream);
}

}  // namespace sc
