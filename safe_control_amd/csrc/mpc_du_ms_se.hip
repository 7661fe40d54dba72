// Kernel 13 for launches whose obstacle rows may be SUPERELLIPSOIDS (the reference's DT barriers of DynamicUnicycle2D and DoubleIntegrator2D
// have that branch: dynamic_unicycle2D.py:204-220, double_integrator2D.py:238-254; so has SingleIntegrator2D's, single_integrator2D.py:162-178,
// but the compiler's code for that instantiation trips tools/check_exec_prologue.py -- AGPR reloads ahead of an EXEC restore -- and it is left
// out: such scenes of that robot run on csrc/mpc_lin.hip): the SE
// instantiations of mpc_du_ms_dev.hpp -- a row's barrier is evaluated per point with its gradient and Hessian (hpoint), the curvature block sums
// the rows' Hessians instead of using the circle's 2 I.  Selected by sc_mpccbf_params.superellipsoid_rows; circles-only launches stay on
// mpc_du_ms.hip.  Built with the basic register allocator (no guard build needed).
#include "mpc_du_ms_dev.hpp"

namespace sc {

hipError_t mpcdu_ms_se_launch(const dums::Params& P, const sc_mpccbf_params& p, const sc_ipopt_params& O, long long B, const void* X, const void* u_prev, const void* goal,
                              const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out, hipStream_t stream) {
    const bool f64 = p.io_dtype == SC_DTYPE_F64;
#define SC_SE(M) (f64 ? dums::launch_t<double, M, true>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, nullptr, stream) \
                      : dums::launch_t<float, M, true>(P, O, B, p.obs_shared, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out, nullptr, stream))
    if (p.model_id == SC_MODEL_DOUBLE_INTEGRATOR2D) return SC_SE(dums::M_DI);
    if (p.model_id == SC_MODEL_DYNAMIC_UNICYCLE2D) return SC_SE(dums::M_DU);
#undef SC_SE
    return hipErrorInvalidValue;                                              // (the other robots' barriers have no superellipsoid branch: api.hip refuses before this)
}

}  // namespace sc
