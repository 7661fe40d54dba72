// C-ABI entry points (include/safe_control_amd.h).  No exceptions cross this file.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
#include <cstring>

#include "../../include/safe_control_amd.h"
#include "mpc_slices_host.hpp"

namespace sc {
hipError_t cbfqp_launch(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                        const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                        hipStream_t stream);

hipError_t mpccbf_launch(const sc_mpccbf_params& p, long long B, int K, const void* X, const void* u_prev,
                         const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out,
                         hipStream_t stream, const ipm::Cont& ct);
size_t mpccbf_state_doubles(int N, int K, bool od);
size_t mpccbf_lds_bytes(int N, int K, bool uni);
size_t odmpccbf_lds_bytes(int N, int K);
hipError_t odmpccbf_launch(const sc_odmpccbf_params& q, long long B, int K, const void* X, const void* u_prev,
                           const void* goal, const void* obs, void* u_out, void* rho_out, int* status, int* iters,
                           void* z_out, hipStream_t stream, const ipm::Cont& ct);
hipError_t neighbors_launch(int io_dtype, long long B_all, long long first, long long B_local, int K, double r, const void* X,
                            void* out, hipStream_t stream);
hipError_t odcbfqp_launch(const sc_odcbfqp_params& p, long long B, const void* X, const void* u_ref, const void* obs,
                          const int* has_obs, void* u_out, void* w_out, int* status, void* h_out, hipStream_t stream);
size_t neighbors_workspace_bytes(int io_dtype, long long B_all, long long B_local, int K);
hipError_t neighbors_split_launch(int io_dtype, long long B_all, long long first, long long B_local, int K, double r,
                                  const void* X, void* out, void* ws, hipStream_t stream);
hipError_t tracking_select_launch(const sc_tracking_params& p, long long B, int M, const void* X, const void* wps,
                                  const int* n_wp, int* wp_index, int* sm, void* goal, const void* table, const int* ret,
                                  void* obs_out, void* goal_out, void* u_ref_out, int* track_out, hipStream_t stream);
hipError_t tracking_apply_launch(const sc_tracking_params& p, long long B, int M, int step_index, void* X, const int* sm,
                                 const void* goal, const void* table, const void* u, const int* u_status, void* u_last,
                                 int* ret, int* ret_step, hipStream_t stream);
hipError_t tracking_launch(const sc_tracking_params& p, long long B, int M, void* X, const void* wps, const int* n_wp,
                           int* wp_index, int* sm, void* goal, void* table, void* u_last, int* ret, int* ret_step,
                           void* tX, void* tU, hipStream_t stream);

hipError_t manip_cbfqp_launch(const sc_manip_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                              const void* obs, const int* n_obs, void* u_out, int* status, void* h_out, hipStream_t stream);

hipError_t quadtrack_select_launch(const sc_quadtrack_params& p, long long B, int M, const void* X, const void* wps, const int* n_wp,
                                   int* wp_index, int* sm, void* goal, const void* obs_table, const int* ret, void* obs_out,
                                   void* goal_out, void* u_ref_out, int* track_out, hipStream_t stream);
hipError_t quadtrack_apply_launch(const sc_quadtrack_params& p, long long B, int M, int step_index, void* X, const int* sm, const void* goal,
                                  const void* obs_table, const void* u, void* u_last, int* ret, int* ret_step, hipStream_t stream);
hipError_t backupcbf_launch(const sc_backupcbf_params& p, long long B, int n_ctrl, int advance, void* X, const void* u_nom,
                            void* bullet_x, void* u_out, int* status, int* using_backup, void* h_min, int* n_rows, double* rows_out,
                            int* ret, int* ret_step, int step0, hipStream_t stream);

size_t mpclin_lds_bytes(int N, int K, int nx, int nu, bool od = false);
size_t mpclin_model_doubles(int nx, int nu, int N);
bool mpclin_build_model(const sc_mpclin_params& p, const double* Ae, const double* Be, const double* As, const double* Bs,
                        double* out);
hipError_t mpclin_launch(const sc_mpclin_params& p, const double* model, long long B, int K, const void* X, const void* u_prev,
                         const void* goal, const void* obs, void* u_out, int* status, int* iters, void* z_out, void* rho_out,
                         hipStream_t stream, const ipm::Cont& ct);
size_t mpclin_state_doubles(int N, int K, int nu);
size_t mpcgn_state_doubles(int N, int K);
size_t mpcvtol_state_doubles(int N, int K, bool od = false);
hipError_t odmpcvtol_wave_launch(const sc_odmpcvtol_params& q, long long B, int K, const void* X, const void* u_prev, const void* goal,
                                 const void* obs, void* u_out, void* rho_out, int* status_out, int* iters_out, void* z_out, hipStream_t stream,
                                 const ipm::Cont& ct);

size_t mpcgn_lds_bytes(int model_id, int N, int K, int circles_only);
size_t odmpcgn_lds_bytes(int model_id, int N, int K);
size_t mpcvtol_workspace_bytes(const sc_mpcvtol_params& p, long long B, int K);
bool mpcvtol_uses_wave(const sc_mpcvtol_params& p, int K);
hipError_t mpcvtol_launch(const sc_mpcvtol_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                          const void* obs, void* u_out, int* status_out, int* iters_out, void* z_out, void* workspace, hipStream_t stream,
                          const ipm::Cont& ct);
hipError_t mpcvtol_ms_launch(const sc_mpcvtol_params& p, const sc_ipopt_params& O, long long B, int K, const void* X, const void* u_prev, const void* goal,
                             const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out, hipStream_t stream);
hipError_t mpcdu_ms_launch(const sc_mpccbf_params& p, const sc_ipopt_params& O, long long B, int K, const void* X, const void* u_prev, const void* goal,
                           const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out, void* order_ws, hipStream_t stream);
size_t mpcdu_ms_lds_bytes(int horizon, int K, int model_id, int se);
size_t mpcdu_ms_order_bytes(long long B);
hipError_t odmpcvtol_ms_launch(const sc_odmpcvtol_params& q, const sc_ipopt_params& O, long long B, int K, const void* X, const void* u_prev, const void* goal,
                               const void* obs, void* u_out, void* rho_out, int* status_out, int* iters_out, void* plan_out, double* trace_out,
                               hipStream_t stream);
hipError_t odmpcgn_launch(const sc_odmpcgn_params& q, long long B, int K, const void* X, const void* u_prev, const void* goal,
                          const void* obs, void* u_out, void* rho_out, int* status, int* iters, void* z_out, hipStream_t stream,
                          const ipm::Cont& ct);
hipError_t mpcgn_launch(const sc_mpcgn_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                        const void* obs, void* u_out, int* status, int* iters, void* z_out, hipStream_t stream, const ipm::Cont& ct);

hipError_t manip_rollout_launch(const sc_manip_tracking_params& t, long long B, int M, void* X, const void* wps, const int* n_wp,
                                int* wp_index, int* sm, void* goal, const void* table, void* u_last, int* ret, int* ret_step,
                                void* tX, void* tU, hipStream_t stream);

static thread_local char g_err[256] = "";

// The launch and hipFuncSetAttribute calls act on the CURRENT device of the calling thread.  A process that drives
// several GPUs may call with tensors of a device that is not current: derive the device from the stream (or, for the
// null stream, from the first data pointer), make it current for the call, restore it afterwards.  One-GPU processes
// (the deployment this library is built for: one process per GPU) skip all of it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    DeviceGuard(void* stream, const void* ptr) {
        static const int n_dev = [] { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 1; } return n; }();
        if (n_dev <= 1) return;
        int cur = 0, want = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
        if (stream) {
            hipDevice_t d;
            if (hipStreamGetDevice((hipStream_t)stream, &d) == hipSuccess) want = (int)d; else (void)hipGetLastError();
        } else if (ptr) {
            hipPointerAttribute_t a;
            if (hipPointerGetAttributes(&a, ptr) == hipSuccess) want = a.device; else (void)hipGetLastError();
        }
        if (want >= 0 && want != cur) { prev = cur; switched = hipSetDevice(want) == hipSuccess; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// the kernels read X / u_ref / u_out with 8- and 16-byte vector accesses
static bool misaligned(const void* p, size_t a) { return p && (reinterpret_cast<uintptr_t>(p) % a) != 0; }

static int fail(int code, const char* msg) {
    std::snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
static int fail_hip(hipError_t e, const char* where) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return e == hipErrorNoDevice ? SC_ERR_NO_DEVICE : SC_ERR_HIP;
}

static int check_cbfqp(const sc_cbfqp_params* p, int64_t B, int32_t K, const void* X, const void* u_ref,
                       const void* obs, const void* u_out, const void* status_out) {
    if (!p) return fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (B < 0) return fail(SC_ERR_INVALID_ARGUMENT, "B < 0");
    if (K < 1) return fail(SC_ERR_INVALID_ARGUMENT, "K < 1 (pass obs_list=None handling to the caller: u = u_ref)");
    if (K > SC_CBFQP_MAX_OBS) return fail(SC_ERR_UNSUPPORTED, "K exceeds SC_CBFQP_MAX_OBS");
    if (p->model_id < 0 || p->model_id >= SC_MODEL_COUNT)
        return fail(SC_ERR_INVALID_ARGUMENT, "unknown model_id");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64)
        return fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->compute_dtype != SC_DTYPE_F32 && p->compute_dtype != SC_DTYPE_F64)
        return fail(SC_ERR_INVALID_ARGUMENT, "compute_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->io_dtype == SC_DTYPE_F64 && p->compute_dtype == SC_DTYPE_F32)
        return fail(SC_ERR_UNSUPPORTED, "f64 storage with f32 arithmetic is not built");
    if (p->cbf_mode != SC_CBF_MODE_CBF && p->cbf_mode != SC_CBF_MODE_HARD)
        return fail(SC_ERR_INVALID_ARGUMENT, "cbf_mode must be 0 (cbf) or 1 (hard)");
    if (!(p->dt > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "dt must be > 0");
    if (p->model_id >= SC_MODEL_KINEMATIC_BICYCLE2D && p->model_id <= SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF &&
        !(p->rear_ax_dist > 0))
        return fail(SC_ERR_INVALID_ARGUMENT, "rear_ax_dist must be > 0 for the KinematicBicycle2D family");
    const int want_dim = p->model_id == SC_MODEL_QUAD2D ? 6 : 4;
    if (!((p->state_dim == 0 && want_dim == 4) || p->state_dim == want_dim))
        return fail(SC_ERR_INVALID_ARGUMENT, "state_dim must be 4 (or 0) for the 4-state models and 6 for Quad2D");
    if (p->model_id == SC_MODEL_QUAD2D && !(p->mass > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "mass must be > 0 for Quad2D");
    if (B > 0 && (!X || !u_ref || !obs || !u_out || !status_out))
        return fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    const size_t es = p->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    // 4-state rows are read with vector loads (a row is 16 / 32 bytes, so every row slice of an aligned array qualifies); Quad2D's
    // six-value rows are read element by element and only need element alignment (a [lo:] slice of an f32 array is 8-byte aligned)
    if (misaligned(X, want_dim == 4 ? 16 : es) || misaligned(u_ref, 2 * es) || misaligned(u_out, 2 * es))
        return fail(SC_ERR_INVALID_ARGUMENT, "X must be 16-byte aligned (Quad2D: element aligned), u_ref / u_out aligned to two elements (vector loads)");
    return SC_OK;
}
// sc_resto_params of every MPC entry point: a zero-initialised block (max_entries = 0) switches the restoration off and is valid --
// the statuses then have their pre-restoration meaning (a multiplier above 1e10 or a failed line search at an infeasible iterate
// ends the solve as SC_STATUS_INFEASIBLE / SC_STATUS_INACCURATE without a certificate); a partially filled one would divide by rho.
static int check_resto(const sc_resto_params& r, bool gauss_newton_ok = false) {
    if (r.gauss_newton != 0 && !(gauss_newton_ok && r.gauss_newton == 1))
        return fail(SC_ERR_UNSUPPORTED, "resto.gauss_newton: 0, or 1 on the VTOL2D entry points");
    if (r.max_entries < 0) return fail(SC_ERR_INVALID_ARGUMENT, "resto.max_entries must be >= 0");
    if (r.slack_reset != 0 && r.slack_reset != 1) return fail(SC_ERR_INVALID_ARGUMENT, "resto.slack_reset must be 0 or 1");
    if (r.retry_max < 0 || r.retry_max > 8) return fail(SC_ERR_INVALID_ARGUMENT, "resto.retry_max must be in [0, 8]");
    if (r.stall_iter < 0 || (r.stall_iter > 0 && !(r.stall_theta > 0))) return fail(SC_ERR_INVALID_ARGUMENT, "resto: stall_iter >= 0, stall_theta > 0 with stall_iter > 0");
    if (r.max_entries == 0) return SC_OK;
    if (!(r.rho > 0) || !(r.kappa > 0 && r.kappa < 1) || !(r.theta_tol > 0) || !(r.tol > 0) || !(r.small_alpha >= 0) || r.small_iter < 1)
        return fail(SC_ERR_INVALID_ARGUMENT, "resto: rho > 0, 0 < kappa < 1, theta_tol > 0, tol > 0, small_alpha >= 0, small_iter >= 1 "
                                             "are required when max_entries > 0");
    return SC_OK;
}
static ipm::Cont one_launch(int max_iter) { ipm::Cont ct{}; ct.it_stop = max_iter; return ct; }
static int check_slices(const sc_mpc_slices* sl, int max_iter, size_t need_bytes) {
    if (!slices_valid(sl)) return fail(SC_ERR_INVALID_ARGUMENT, "slices: 0 <= n_caps <= SC_MPC_MAX_SLICES, caps >= 1 and strictly increasing, order / classify_first in {0, 1}");
    if (slices_active(sl, max_iter)) {
        if (!sl->workspace) return fail(SC_ERR_INVALID_ARGUMENT, "slices.workspace is NULL");
        if (sl->workspace_bytes < need_bytes) return fail(SC_ERR_INVALID_ARGUMENT, "slices.workspace_bytes below sc_mpc*_slices_workspace_bytes()");
        if (misaligned(sl->workspace, 16)) return fail(SC_ERR_INVALID_ARGUMENT, "slices.workspace must be 16-byte aligned");
    }
    return SC_OK;
}

static int check_mpccbf(const sc_mpccbf_params* p, int64_t B, int32_t K, const void* X, const void* u_prev,
                        const void* goal, const void* obs, const void* u_out, const void* status_out) {
    if (!p) return fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (B < 0) return fail(SC_ERR_INVALID_ARGUMENT, "B < 0");
    if (K < 1) return fail(SC_ERR_INVALID_ARGUMENT, "K < 1 (pad with [1000,1000,0,...] rows like update_tvp)");
    if (p->model_id != SC_MODEL_DYNAMIC_UNICYCLE2D && p->model_id != SC_MODEL_UNICYCLE2D)
        return fail(SC_ERR_UNSUPPORTED, "MPC-CBF is built for DynamicUnicycle2D and Unicycle2D only");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64)
        return fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->horizon < 1 || p->horizon > SC_MPCCBF_MAX_HORIZON)
        return fail(SC_ERR_UNSUPPORTED, "horizon outside [1, SC_MPCCBF_MAX_HORIZON]");
    if (mpccbf_lds_bytes(p->horizon, K, p->model_id == SC_MODEL_UNICYCLE2D) > 160 * 1024)
        return fail(SC_ERR_UNSUPPORTED, "horizon x obstacles does not fit the 160 KiB LDS of one CU");
    if (!(p->dt > 0) || !(p->tol > 0) || !(p->acceptable_tol >= p->tol) || p->max_iter < 1 || !(p->mu_init > 0) || !(p->mu_min > 0))
        return fail(SC_ERR_INVALID_ARGUMENT, "dt, tol, mu_init, mu_min must be > 0 and max_iter >= 1");
    if (!(p->u_max[0] > 0) || !(p->u_max[1] > 0) || !(p->v_max > 0))
        return fail(SC_ERR_INVALID_ARGUMENT, "u_max and v_max must be > 0");
    if (B > 0 && (!X || !u_prev || !goal || !obs || !u_out || !status_out))
        return fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B > 0x7fffffffLL) return fail(SC_ERR_UNSUPPORTED, "B too large for one launch");
    if (p->slack_reset != 0 && p->slack_reset != 2) return fail(SC_ERR_INVALID_ARGUMENT, "slack_reset must be 0 or 2");
    return check_resto(p->resto);
}
static int check_mpclin_dims(const sc_mpclin_params* p) {
    if (!p) return fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (p->nx < 2 || p->nx > 12 || p->nu < 1 || p->nu > 4 || p->ng < 2 || p->ng > p->nx)
        return fail(SC_ERR_INVALID_ARGUMENT, "need 2 <= nx <= 12, 1 <= nu <= 4, 2 <= ng <= nx");
    if (p->horizon < 1 || p->horizon > 128 || p->horizon * p->nu > 128)           // horizon bounded first: the product cannot overflow
        return fail(SC_ERR_UNSUPPORTED, "need 1 <= horizon and nu * horizon <= 128");
    return SC_OK;
}
static int check_mpclin(const sc_mpclin_params* p, const double* model, int64_t B, int32_t K, const void* X, const void* u_prev,
                        const void* goal, const void* obs, const void* u_out, const void* status_out) {
    int rc = check_mpclin_dims(p);
    if (rc != SC_OK) return rc;
    if (B < 0) return fail(SC_ERR_INVALID_ARGUMENT, "B < 0");
    if (K < 1) return fail(SC_ERR_INVALID_ARGUMENT, "K < 1 (pad with [1000,1000,0,...] rows like update_tvp)");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64)
        return fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->optimal_decay < 0 || p->optimal_decay > 2) return fail(SC_ERR_INVALID_ARGUMENT, "optimal_decay must be 0, 1 or 2");
    if (p->slack_reset != 0 && p->slack_reset != 2) return fail(SC_ERR_INVALID_ARGUMENT, "slack_reset must be 0 or 2");
    if (p->optimal_decay == 1 && !(p->nx == 12 && p->nu == 4))
        return fail(SC_ERR_UNSUPPORTED, "the optimal-decay extension of this kernel is built for Quad3D (nx = 12, nu = 4)");
    if (p->optimal_decay == 1 && !(p->od_p_sb > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "od_p_sb must be > 0");
    if (mpclin_lds_bytes(p->horizon, K, p->nx, p->nu, p->optimal_decay == 1) > 160 * 1024)
        return fail(SC_ERR_UNSUPPORTED, "horizon x obstacles does not fit the 160 KiB LDS of one CU");
    if (!(p->tol > 0) || !(p->acceptable_tol >= p->tol) || p->max_iter < 1 || !(p->mu_init > 0) || !(p->mu_min > 0))
        return fail(SC_ERR_INVALID_ARGUMENT, "tol, mu_init, mu_min must be > 0 and max_iter >= 1");
    for (int i = 0; i < p->nu; ++i)
        if (!(p->u_hi[i] > p->u_lo[i])) return fail(SC_ERR_INVALID_ARGUMENT, "u_hi must be > u_lo");
    if (B > 0 && (!model || !X || !u_prev || !goal || !obs || !u_out || !status_out))
        return fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B > 0x7fffffffLL) return fail(SC_ERR_UNSUPPORTED, "B too large for one launch");
    return p->optimal_decay == 1 ? SC_OK : check_resto(p->resto);
}
static int check_mpcgn(const sc_mpcgn_params* p, int64_t B, int32_t K, const void* X, const void* u_prev, const void* goal,
                       const void* obs, const void* u_out, const void* status_out) {
    if (!p) return fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    const bool kb_family = p->model_id == SC_MODEL_KINEMATIC_BICYCLE2D || p->model_id == SC_MODEL_KINEMATIC_BICYCLE2D_C3BF ||
                           p->model_id == SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF;
    if (p->model_id != SC_MODEL_DOUBLE_INTEGRATOR2D && p->model_id != SC_MODEL_QUAD2D && !kb_family)
        return fail(SC_ERR_UNSUPPORTED, "this entry point serves DoubleIntegrator2D, Quad2D and the KinematicBicycle2D family");
    if (B < 0) return fail(SC_ERR_INVALID_ARGUMENT, "B < 0");
    if (K < 1) return fail(SC_ERR_INVALID_ARGUMENT, "K < 1 (pad with [1000,1000,0,...] rows like update_tvp)");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64)
        return fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->horizon < 1 || p->horizon > 32) return fail(SC_ERR_UNSUPPORTED, "horizon outside [1, 32]");
    if (p->slack_reset != 0 && p->slack_reset != 2) return fail(SC_ERR_INVALID_ARGUMENT, "slack_reset must be 0 or 2 on this entry point");
    if (mpcgn_lds_bytes(p->model_id, p->horizon, K, p->circles_only) > 160 * 1024)
        return fail(SC_ERR_UNSUPPORTED, "horizon x obstacles does not fit the 160 KiB LDS of one CU");
    if (!(p->dt > 0) || !(p->tol > 0) || !(p->acceptable_tol >= p->tol) || p->max_iter < 1 || !(p->mu_init > 0) || !(p->mu_min > 0))
        return fail(SC_ERR_INVALID_ARGUMENT, "dt, tol, mu_init, mu_min must be > 0 and max_iter >= 1");
    if (!(p->u_hi[0] > p->u_lo[0]) || !(p->u_hi[1] > p->u_lo[1])) return fail(SC_ERR_INVALID_ARGUMENT, "u_hi must be > u_lo");
    if (p->model_id == SC_MODEL_DOUBLE_INTEGRATOR2D && !(p->v_max > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "DoubleIntegrator2D needs v_max > 0");
    if (p->model_id == SC_MODEL_QUAD2D && (!(p->mass > 0) || !(p->inertia > 0))) return fail(SC_ERR_INVALID_ARGUMENT, "Quad2D needs mass, inertia > 0");
    if (kb_family && (!(p->rear_ax_dist > 0) || !(p->v_max > p->v_min)))
        return fail(SC_ERR_INVALID_ARGUMENT, "KinematicBicycle2D needs rear_ax_dist > 0 and v_max > v_min");
    if (B > 0 && (!X || !u_prev || !goal || !obs || !u_out || !status_out)) return fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B > 0x7fffffffLL) return fail(SC_ERR_UNSUPPORTED, "B too large for one launch");
    return check_resto(p->resto);
}
static int check_mpcvtol(const sc_mpcvtol_params* p, int64_t B, int32_t K, const void* X, const void* u_prev, const void* goal,
                         const void* obs, const void* u_out, const void* status_out) {
    if (!p) return fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (B < 0) return fail(SC_ERR_INVALID_ARGUMENT, "B < 0");
    if (K < 1) return fail(SC_ERR_INVALID_ARGUMENT, "K < 1 (pad with [1000,1000,0,...] rows like update_tvp)");
    if (K > 16) return fail(SC_ERR_UNSUPPORTED, "K > 16 obstacles per aircraft");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64)
        return fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->horizon < 1 || p->horizon > 64) return fail(SC_ERR_UNSUPPORTED, "horizon outside [1, 64]");
    if (!(p->dt > 0) || !(p->tol > 0) || !(p->acceptable_tol >= p->tol) || p->max_iter < 1 || !(p->mu_init > 0) || !(p->mu_min > 0))
        return fail(SC_ERR_INVALID_ARGUMENT, "dt, tol, mu_init, mu_min must be > 0 and max_iter >= 1");
    for (int i = 0; i < 4; ++i)
        if (!(p->u_hi[i] > p->u_lo[i])) return fail(SC_ERR_INVALID_ARGUMENT, "u_hi must be > u_lo");
    if (!(p->airframe[0] > 0) || !(p->airframe[1] > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "airframe: mass and inertia must be > 0");
    if (!(p->v_max > 0) || !(p->pitch_max > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "v_max and pitch_max must be > 0");
    if (p->slack_reset < 0 || p->slack_reset > 2) return fail(SC_ERR_INVALID_ARGUMENT, "slack_reset must be 0, 1 or 2");
    if (p->kernel < 0 || p->kernel > 2) return fail(SC_ERR_INVALID_ARGUMENT, "kernel must be 0 (auto) or 2 (wave per problem)");
    if (p->kernel == 1) return fail(SC_ERR_UNSUPPORTED, "kernel = 1 (one NLP per lane) was retired in round 6: the wave-per-problem kernel serves every case");
    if (p->kernel == 2 && !mpcvtol_uses_wave(*p, K)) return fail(SC_ERR_UNSUPPORTED, "the wave-per-problem kernel serves K <= 16, horizon <= 64");
    if (B > 0 && (!X || !u_prev || !goal || !obs || !u_out || !status_out)) return fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B > 0x7fffffffLL) return fail(SC_ERR_UNSUPPORTED, "B too large for one launch");
    if (p->resto.retry_max != 0 || p->resto.stall_iter != 0)
        return fail(SC_ERR_UNSUPPORTED, "the VTOL2D kernels run the restoration without damped retries and without the stall certificate (resto.retry_max = resto.stall_iter = 0)");
    return check_resto(p->resto, true);
}
static int check_manip(const sc_manip_cbfqp_params* p, int64_t B, int32_t K, const void* X, const void* u_ref,
                       const void* obs, const void* u_out, const void* status_out) {
    if (!p) return fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (B < 0) return fail(SC_ERR_INVALID_ARGUMENT, "B < 0");
    if (K < 1) return fail(SC_ERR_INVALID_ARGUMENT, "K < 1 (obs_list=None is the caller's branch: u = u_ref)");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64)
        return fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->cbf_mode != SC_CBF_MODE_CBF && p->cbf_mode != SC_CBF_MODE_HARD)
        return fail(SC_ERR_INVALID_ARGUMENT, "cbf_mode must be 0 (cbf) or 1 (hard)");
    if (p->num_rows < 1) return fail(SC_ERR_INVALID_ARGUMENT, "num_rows < 1");
    if (p->num_rows > SC_MANIP_MAX_ROWS) return fail(SC_ERR_UNSUPPORTED, "num_rows exceeds SC_MANIP_MAX_ROWS");
    for (int i = 0; i < 3; ++i) {
        if (p->link_steps[i] < 1 || p->link_steps[i] > 64) return fail(SC_ERR_INVALID_ARGUMENT, "link_steps must be in [1, 64]");
        if (!(p->link_lengths[i] > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "link_lengths must be > 0");
    }
    if (!(p->dt > 0) || !(p->w_max > 0)) return fail(SC_ERR_INVALID_ARGUMENT, "dt and w_max must be > 0");
    if (B > 0 && (!X || !u_ref || !obs || !u_out || !status_out))
        return fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B > 0x7fffffffLL) return fail(SC_ERR_UNSUPPORTED, "B too large for one launch");
    return SC_OK;
}
}  // namespace sc

extern "C" {

int sc_mpcgn_solve_batch(const sc_mpcgn_params* params, int64_t B, int32_t K, const void* X, const void* u_prev, const void* goal,
                         const void* obs, void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpcgn(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::mpcgn_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out,
                                    (hipStream_t)stream, sc::one_launch(params->max_iter));
    if (e != hipSuccess) return sc::fail_hip(e, "mpcgn kernel launch");
    return SC_OK;
}

size_t sc_mpcvtol_workspace_bytes(const sc_mpcvtol_params* params, int64_t B, int32_t K) {
    if (!params || B < 0 || K < 1 || params->horizon < 1) return 0;
    return sc::mpcvtol_workspace_bytes(*params, (long long)B, (int)K);
}

int sc_mpcvtol_solve_batch(const sc_mpcvtol_params* params, int64_t B, int32_t K, const void* X, const void* u_prev, const void* goal,
                           const void* obs, void* u_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* workspace,
                           size_t workspace_bytes, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpcvtol(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    const size_t need = sc::mpcvtol_workspace_bytes(*params, (long long)B, (int)K);
    if (need > 0 && !workspace) return sc::fail(SC_ERR_INVALID_ARGUMENT, "workspace is NULL");
    if (workspace_bytes < need) return sc::fail(SC_ERR_INVALID_ARGUMENT, "workspace smaller than sc_mpcvtol_workspace_bytes()");
    hipError_t e = sc::mpcvtol_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, workspace,
                                      (hipStream_t)stream, sc::one_launch(params->max_iter));
    if (e != hipSuccess) return sc::fail_hip(e, "mpcvtol kernel launch");
    return SC_OK;
}

int sc_mpcvtol_solve_batch_host(const sc_mpcvtol_params* params, int64_t B, int32_t K, const void* X, const void* u_prev,
                                const void* goal, const void* obs, void* u_out, int32_t* status_out, int32_t* iters_out,
                                void* z_out, int device) {
    int rc = sc::check_mpcvtol(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const size_t es = params->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t n = 4 * (size_t)params->horizon;
    const size_t nX = (size_t)B * 6 * es, nU = (size_t)B * 4 * es, nG = (size_t)B * 2 * es;
    const size_t nO = (params->obs_shared ? (size_t)K * 7 : (size_t)B * K * 7) * es;
    const size_t nS = (size_t)B * 4, nZ = (size_t)B * n * es, nW = sc::mpcvtol_workspace_bytes(*params, (long long)B, (int)K);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oX = 0, oU = oX + up(nX), oG = oU + up(nU), oO = oG + up(nG), oUo = oO + up(nO), oS = oUo + up(nU),
                 oI = oS + up(nS), oZ = oI + up(nS), oW = oZ + up(nZ), total = oW + up(nW);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    rc = SC_OK;
    do {
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_prev, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oG, goal, nG, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        e = sc::mpcvtol_launch(*params, (long long)B, (int)K, d + oX, d + oU, d + oG, d + oO, d + oUo, (int*)(d + oS),
                               iters_out ? (int*)(d + oI) : nullptr, z_out ? d + oZ : nullptr, d + oW, s, sc::one_launch(params->max_iter));
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (iters_out && (e = hipMemcpyAsync(iters_out, d + oI, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (z_out && (e = hipMemcpyAsync(z_out, d + oZ, nZ, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_mpcvtol_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

int sc_odmpcgn_solve_batch(const sc_odmpcgn_params* params, int64_t B, int32_t K, const void* X, const void* u_prev, const void* goal,
                           const void* obs, void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream) {
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpcgn(&params->mpc, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (params->mpc.model_id != SC_MODEL_KINEMATIC_BICYCLE2D && params->mpc.model_id != SC_MODEL_QUAD2D)
        return sc::fail(SC_ERR_UNSUPPORTED, "optimal-decay MPC-CBF on this entry point: KinematicBicycle2D and Quad2D (DynamicUnicycle2D: sc_odmpccbf_solve_batch)");
    if (!(params->p_sb[0] > 0) || !(params->p_sb[1] > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "p_sb must be > 0");
    if (sc::odmpcgn_lds_bytes(params->mpc.model_id, params->mpc.horizon, K) > 160 * 1024)
        return sc::fail(SC_ERR_UNSUPPORTED, "horizon x obstacles does not fit the 160 KiB LDS of one CU");
    if (B == 0) return SC_OK;
    hipError_t e = sc::odmpcgn_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, rho_out, status_out, iters_out, z_out,
                                      (hipStream_t)stream, sc::one_launch(params->mpc.max_iter));
    if (e != hipSuccess) return sc::fail_hip(e, "odmpcgn kernel launch");
    return SC_OK;
}

int sc_mpcgn_solve_batch_host(const sc_mpcgn_params* params, int64_t B, int32_t K, const void* X, const void* u_prev,
                              const void* goal, const void* obs, void* u_out, int32_t* status_out, int32_t* iters_out,
                              void* z_out, int device) {
    int rc = sc::check_mpcgn(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const size_t es = params->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t nx = params->model_id == SC_MODEL_QUAD2D ? 6 : 4, n = 2 * (size_t)params->horizon;
    const size_t nX = (size_t)B * nx * es, nU = (size_t)B * 2 * es, nG = nU;
    const size_t nO = (params->obs_shared ? (size_t)K * 7 : (size_t)B * K * 7) * es;
    const size_t nS = (size_t)B * 4, nZ = (size_t)B * n * es;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oX = 0, oU = oX + up(nX), oG = oU + up(nU), oO = oG + up(nG), oUo = oO + up(nO), oS = oUo + up(nU),
                 oI = oS + up(nS), oZ = oI + up(nS), total = oZ + up(nZ);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    rc = SC_OK;
    do {
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_prev, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oG, goal, nG, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        e = sc::mpcgn_launch(*params, (long long)B, (int)K, d + oX, d + oU, d + oG, d + oO, d + oUo, (int*)(d + oS),
                             iters_out ? (int*)(d + oI) : nullptr, z_out ? d + oZ : nullptr, s, sc::one_launch(params->max_iter));
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (iters_out && (e = hipMemcpyAsync(iters_out, d + oI, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (z_out && (e = hipMemcpyAsync(z_out, d + oZ, nZ, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_mpcgn_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

size_t sc_mpclin_model_doubles(int32_t nx, int32_t nu, int32_t horizon) {
    if (nx < 2 || nx > 12 || nu < 1 || nu > 4 || horizon < 1 || horizon > 128 || horizon * nu > 128) return 0;
    return sc::mpclin_model_doubles(nx, nu, horizon);
}

int sc_mpclin_build_model(const sc_mpclin_params* params, const double* Ae, const double* Be, const double* As,
                          const double* Bs, double* model_out) {
    int rc = sc::check_mpclin_dims(params);
    if (rc != SC_OK) return rc;
    if (!Ae || !Be || !As || !Bs || !model_out) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL matrix pointer");
    if (!sc::mpclin_build_model(*params, Ae, Be, As, Bs, model_out)) return sc::fail(SC_ERR_HIP, "out of host memory building the condensed model");
    return SC_OK;
}

int sc_mpclin_solve_batch(const sc_mpclin_params* params, const double* model, int64_t B, int32_t K, const void* X,
                          const void* u_prev, const void* goal, const void* obs, void* u_out, int32_t* status_out,
                          int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpclin(params, model, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    if (params->optimal_decay == 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "optimal_decay = 1: call sc_odmpclin_solve_batch");
    hipError_t e = sc::mpclin_launch(*params, model, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out,
                                     z_out, nullptr, (hipStream_t)stream, sc::one_launch(params->max_iter));
    if (e != hipSuccess) return sc::fail_hip(e, "mpclin kernel launch");
    return SC_OK;
}

int sc_odmpclin_solve_batch(const sc_mpclin_params* params, const double* model, int64_t B, int32_t K, const void* X,
                            const void* u_prev, const void* goal, const void* obs, void* u_out, void* rho_out,
                            int32_t* status_out, int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpclin(params, model, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (params->optimal_decay != 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "optimal_decay != 1: call sc_mpclin_solve_batch");
    if (B == 0) return SC_OK;
    hipError_t e = sc::mpclin_launch(*params, model, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out,
                                     z_out, rho_out, (hipStream_t)stream, sc::one_launch(params->max_iter));
    if (e != hipSuccess) return sc::fail_hip(e, "mpclin (optimal decay) kernel launch");
    return SC_OK;
}

int sc_mpclin_solve_batch_host(const sc_mpclin_params* params, const double* model, int64_t B, int32_t K, const void* X,
                               const void* u_prev, const void* goal, const void* obs, void* u_out, int32_t* status_out,
                               int32_t* iters_out, void* z_out, int device) {
    int rc = sc::check_mpclin(params, model, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (params->optimal_decay == 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "optimal_decay = 1: call sc_odmpclin_solve_batch (device pointers)");
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const size_t es = params->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t n = (size_t)params->nu * params->horizon;
    const size_t nM = sc::mpclin_model_doubles(params->nx, params->nu, params->horizon) * 8;
    const size_t nX = (size_t)B * params->nx * es, nU = (size_t)B * params->nu * es, nG = (size_t)B * params->ng * es;
    const size_t nO = (params->obs_shared ? (size_t)K * 7 : (size_t)B * K * 7) * es;
    const size_t nS = (size_t)B * 4, nZ = (size_t)B * n * es;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oM = 0, oX = oM + up(nM), oU = oX + up(nX), oG = oU + up(nU), oO = oG + up(nG), oUo = oO + up(nO),
                 oS = oUo + up(nU), oI = oS + up(nS), oZ = oI + up(nS), total = oZ + up(nZ);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    rc = SC_OK;
    do {
        if ((e = hipMemcpyAsync(d + oM, model, nM, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_prev, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oG, goal, nG, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        e = sc::mpclin_launch(*params, (const double*)(d + oM), (long long)B, (int)K, d + oX, d + oU, d + oG, d + oO, d + oUo,
                              (int*)(d + oS), iters_out ? (int*)(d + oI) : nullptr, z_out ? d + oZ : nullptr, nullptr, s, sc::one_launch(params->max_iter));
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (iters_out && (e = hipMemcpyAsync(iters_out, d + oI, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (z_out && (e = hipMemcpyAsync(z_out, d + oZ, nZ, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_mpclin_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

int sc_manip_tracking_rollout_batch(const sc_manip_tracking_params* params, int64_t B, int32_t M, void* X, const void* waypoints,
                                    const int32_t* n_wp, int32_t* wp_index, int32_t* state_machine, void* goal,
                                    const void* obs_table, void* u_last, int32_t* ret, int32_t* ret_step, void* traj_X,
                                    void* traj_U, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    static const double dummy = 0.0;
    int rc = sc::check_manip(&params->qp, B, 1, &dummy, &dummy, &dummy, &dummy, &dummy);
    if (rc != SC_OK) return rc;
    if (M < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "M < 0");
    if (params->n_steps < 0 || params->max_waypoints < 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "n_steps < 0 or max_waypoints < 1");
    if ((size_t)M * 7 * 8 > 64 * 1024) return sc::fail(SC_ERR_UNSUPPORTED, "obstacle table does not fit 64 KiB of LDS");
    if (B > 0 && (!X || !waypoints || !n_wp || !wp_index || !state_machine || !goal || !u_last || !ret || !ret_step || (M > 0 && !obs_table)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B == 0 || params->n_steps == 0) return SC_OK;
    hipError_t e = sc::manip_rollout_launch(*params, (long long)B, (int)M, X, waypoints, n_wp, wp_index, state_machine, goal,
                                            obs_table, u_last, ret, ret_step, traj_X, traj_U, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "manipulator rollout kernel launch");
    return SC_OK;
}

static int check_quadtrack(const sc_quadtrack_params* p, int64_t B, int32_t M) {
    if (!p) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (B < 0 || M < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "B < 0 or M < 0");
    if (p->model != SC_QUADTRACK_QUAD2D && p->model != SC_QUADTRACK_QUAD3D && p->model != SC_QUADTRACK_VTOL2D)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "model must be SC_QUADTRACK_QUAD2D, _QUAD3D or _VTOL2D");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64) return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->num_constraints < 1 || p->num_constraints > SC_TRACKING_MAX_CONSTRAINTS) return sc::fail(SC_ERR_UNSUPPORTED, "num_constraints outside 1..SC_TRACKING_MAX_CONSTRAINTS");
    if (p->max_waypoints < 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "max_waypoints < 1");
    if ((size_t)M * 7 * 8 > 64 * 1024) return sc::fail(SC_ERR_UNSUPPORTED, "obstacle table does not fit 64 KiB of LDS");
    if (!(p->dt > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "dt must be > 0");
    if (p->model != SC_QUADTRACK_VTOL2D && !(p->mass > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "mass must be > 0");
    if (p->model == SC_QUADTRACK_VTOL2D && (!(p->airframe[0] > 0) || !(p->airframe[1] > 0) || !(p->pitch_limit > 0)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "VTOL2D: airframe mass and inertia and pitch_limit must be > 0");
    if (p->model == SC_QUADTRACK_QUAD2D && (!(p->inertia > 0) || !(p->robot_radius > 0))) return sc::fail(SC_ERR_INVALID_ARGUMENT, "Quad2D: inertia and radius must be > 0");
    if (p->model == SC_QUADTRACK_QUAD3D && (!(p->Ix > 0) || !(p->Iy > 0) || !(p->Iz > 0) || !(p->L > 0) || !(p->nu > 0)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "Quad3D: Ix, Iy, Iz, L, nu must be > 0");
    return SC_OK;
}

int sc_quadtrack_select_batch(const sc_quadtrack_params* params, int64_t B, int32_t M, const void* X, const void* waypoints,
                              const int32_t* n_wp, int32_t* wp_index, int32_t* state_machine, void* goal, const void* obs_table,
                              const int32_t* ret, void* obs_out, void* goal_out, void* u_ref_out, int32_t* track_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_quadtrack(params, B, M);
    if (rc != SC_OK) return rc;
    if (B > 0 && (!X || !waypoints || !n_wp || !wp_index || !state_machine || !goal || !ret || !obs_out || !goal_out || !u_ref_out ||
                  !track_out || (M > 0 && !obs_table)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B == 0) return SC_OK;
    hipError_t e = sc::quadtrack_select_launch(*params, (long long)B, (int)M, X, waypoints, n_wp, wp_index, state_machine, goal, obs_table,
                                               ret, obs_out, goal_out, u_ref_out, track_out, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "quadrotor select kernel launch");
    return SC_OK;
}

int sc_quadtrack_apply_batch(const sc_quadtrack_params* params, int64_t B, int32_t M, int32_t step_index, void* X,
                             const int32_t* state_machine, const void* goal, const void* obs_table, const void* u, void* u_last,
                             int32_t* ret, int32_t* ret_step, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_quadtrack(params, B, M);
    if (rc != SC_OK) return rc;
    if (B > 0 && (!X || !state_machine || !goal || !u || !u_last || !ret || !ret_step || (M > 0 && !obs_table)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B == 0) return SC_OK;
    hipError_t e = sc::quadtrack_apply_launch(*params, (long long)B, (int)M, (int)step_index, X, state_machine, goal, obs_table, u, u_last,
                                              ret, ret_step, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "quadrotor apply kernel launch");
    return SC_OK;
}

static int check_backupcbf(const sc_backupcbf_params* p, int64_t B, const void* X, const void* bullet_x, const void* u_out,
                           const void* status_out) {
    if (!p) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (B < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "B < 0");
    if (p->io_dtype != SC_DTYPE_F32 && p->io_dtype != SC_DTYPE_F64)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (p->n_steps < 2) return sc::fail(SC_ERR_INVALID_ARGUMENT, "n_steps < 2 (int(backup_horizon / dt))");
    if (p->n_steps > 128) return sc::fail(SC_ERR_UNSUPPORTED, "n_steps > 128 backup states");
    if (!(p->dt > 0) || !(p->fd_eps > 0) || !(p->a_max > 0) || !(p->v_max > 0) || !(p->backup_horizon > 0))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "dt, fd_eps, a_max, v_max, backup_horizon must be > 0");
    if (!(p->pocket_x_min < p->pocket_x_max) || !(p->pocket_y_min < p->pocket_y_max) || !(p->half_width > 0) || !(p->hallway_length > 0))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "degenerate EvadeEnv geometry");
    if (B > 0 && (!X || !bullet_x || !u_out || !status_out)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    return SC_OK;
}

int sc_backupcbf_solve_batch(const sc_backupcbf_params* params, int64_t B, const void* X, const void* u_nom,
                             const void* bullet_x, void* u_out, int32_t* status_out, int32_t* using_backup_out,
                             void* h_min_out, int32_t* n_rows_out, double* rows_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_backupcbf(params, B, X, bullet_x, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::backupcbf_launch(*params, (long long)B, 1, 0, const_cast<void*>(X), u_nom, const_cast<void*>(bullet_x), u_out,
                                        status_out, using_backup_out, h_min_out, n_rows_out, rows_out, nullptr, nullptr, 0,
                                        (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "backup-CBF kernel launch");
    return SC_OK;
}

int sc_backupcbf_rollout_batch(const sc_backupcbf_params* params, int64_t B, int32_t n_ctrl, int32_t step_offset, void* X,
                               void* bullet_x, void* u_out, int32_t* status_out, int32_t* using_backup_out, void* h_min_out,
                               int32_t* ret, int32_t* ret_step, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_backupcbf(params, B, X, bullet_x, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (n_ctrl < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "n_ctrl < 0");
    if (B > 0 && (!ret || !ret_step)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B == 0 || n_ctrl == 0) return SC_OK;
    hipError_t e = sc::backupcbf_launch(*params, (long long)B, n_ctrl, 1, X, nullptr, bullet_x, u_out, status_out, using_backup_out,
                                        h_min_out, nullptr, nullptr, ret, ret_step, step_offset, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "backup-CBF rollout kernel launch");
    return SC_OK;
}

int sc_manip_cbfqp_solve_batch(const sc_manip_cbfqp_params* params, int64_t B, int32_t K, const void* X, const void* u_ref,
                               const void* obs, const int32_t* n_obs, void* u_out, int32_t* status_out, void* h_out,
                               void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_manip(params, B, K, X, u_ref, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::manip_cbfqp_launch(*params, (long long)B, (int)K, X, u_ref, obs, n_obs, u_out, status_out, h_out,
                                          (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "manipulator cbfqp kernel launch");
    return SC_OK;
}

int sc_manip_cbfqp_solve_batch_host(const sc_manip_cbfqp_params* params, int64_t B, int32_t K, const void* X,
                                    const void* u_ref, const void* obs, const int32_t* n_obs, void* u_out,
                                    int32_t* status_out, void* h_out, int device) {
    int rc = sc::check_manip(params, B, K, X, u_ref, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const size_t es = params->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t nX = (size_t)B * 3 * es, nU = nX;
    const size_t nO = (params->obs_shared ? (size_t)K * 7 : (size_t)B * K * 7) * es;
    const size_t nH = (size_t)B * params->num_rows * es, nS = (size_t)B * 4, nN = n_obs ? (size_t)B * 4 : 0;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oX = 0, oU = oX + up(nX), oO = oU + up(nU), oN = oO + up(nO), oUo = oN + up(nN),
                 oS = oUo + up(nU), oH = oS + up(nS), total = oH + up(nH);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    rc = SC_OK;
    do {
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_ref, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if (n_obs && (e = hipMemcpyAsync(d + oN, n_obs, nN, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        e = sc::manip_cbfqp_launch(*params, (long long)B, (int)K, d + oX, d + oU, d + oO,
                                   n_obs ? (const int*)(d + oN) : nullptr, d + oUo, (int*)(d + oS),
                                   h_out ? d + oH : nullptr, s);
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (h_out && (e = hipMemcpyAsync(h_out, d + oH, nH, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_manip_cbfqp_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

int sc_version(void) { return SC_VERSION_MAJOR * 1000 + SC_VERSION_MINOR; }

const char* sc_last_error(void) { return sc::g_err; }

int sc_device_count(int* count_out) {
    if (!count_out) return sc::fail(SC_ERR_INVALID_ARGUMENT, "count_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count_out = 0; return sc::fail_hip(e, "hipGetDeviceCount"); }
    *count_out = n;
    return SC_OK;
}

int sc_cbfqp_solve_batch(const sc_cbfqp_params* params, int64_t B, int32_t K, const void* X, const void* u_ref,
                         const void* obs, const int32_t* n_obs, void* u_out, int32_t* status_out, void* h_out,
                         void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_cbfqp(params, B, K, X, u_ref, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::cbfqp_launch(*params, (long long)B, (int)K, X, u_ref, obs, n_obs, u_out, status_out, h_out,
                                    (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "cbfqp kernel launch");
    return SC_OK;
}

int sc_cbfqp_solve_batch_host(const sc_cbfqp_params* params, int64_t B, int32_t K, const void* X,
                              const void* u_ref, const void* obs, const int32_t* n_obs, void* u_out,
                              int32_t* status_out, void* h_out, int device) {
    int rc = sc::check_cbfqp(params, B, K, X, u_ref, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const size_t es = params->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t nX = (size_t)B * (params->model_id == SC_MODEL_QUAD2D ? 6 : 4) * es, nU = (size_t)B * 2 * es;
    const size_t nO = (params->obs_shared ? (size_t)K * 7 : (size_t)B * K * 7) * es;
    const size_t nH = (size_t)B * K * es, nS = (size_t)B * 4, nN = n_obs ? (size_t)B * 4 : 0;
    // one allocation, 256-byte aligned sections
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oX = 0, oU = oX + up(nX), oO = oU + up(nU), oN = oO + up(nO), oUo = oN + up(nN),
                 oS = oUo + up(nU), oH = oS + up(nS), total = oH + up(nH);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    rc = SC_OK;
    do {
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_ref, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if (n_obs && (e = hipMemcpyAsync(d + oN, n_obs, nN, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        e = sc::cbfqp_launch(*params, (long long)B, (int)K, d + oX, d + oU, d + oO,
                             n_obs ? (const int*)(d + oN) : nullptr, d + oUo, (int*)(d + oS),
                             h_out ? d + oH : nullptr, s);
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (h_out && (e = hipMemcpyAsync(h_out, d + oH, nH, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_cbfqp_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

int sc_mpccbf_solve_batch(const sc_mpccbf_params* params, int64_t B, int32_t K, const void* X, const void* u_prev,
                          const void* goal, const void* obs, void* u_out, int32_t* status_out, int32_t* iters_out,
                          void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpccbf(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    sc::ipm::Cont ct{};
    ct.it_stop = params->max_iter;
    hipError_t e = sc::mpccbf_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out,
                                     z_out, (hipStream_t)stream, ct);
    if (e != hipSuccess) return sc::fail_hip(e, "mpccbf kernel launch");
    return SC_OK;
}

/* ---- continuation launches of the other MPC families (sc_mpc_slices) ------------------------------------------------------- */
size_t sc_mpcgn_slices_workspace_bytes(const sc_mpcgn_params* params, int64_t B, int32_t K) {
    if (!params || B < 0 || K < 1 || params->horizon < 1 || params->horizon > 32) return 0;
    return sc::slices_workspace_bytes((long long)B, sc::mpcgn_state_doubles(params->horizon, K));
}
int sc_mpcgn_solve_batch_sliced(const sc_mpcgn_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K, const void* X,
                                const void* u_prev, const void* goal, const void* obs, void* u_out, int32_t* status_out,
                                int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpcgn(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    rc = sc::check_slices(slices, params->max_iter, sc_mpcgn_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->max_iter, (long long)B, sc::mpcgn_state_doubles(params->horizon, K), (hipStream_t)stream,
                                  [&](const sc::ipm::Cont& ct) {
        return sc::mpcgn_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "mpcgn kernel launch (sliced)");
    return SC_OK;
}

size_t sc_mpclin_slices_workspace_bytes(const sc_mpclin_params* params, int64_t B, int32_t K) {
    if (!params || B < 0 || K < 1 || sc::check_mpclin_dims(params) != SC_OK) return 0;
    return sc::slices_workspace_bytes((long long)B, sc::mpclin_state_doubles(params->horizon, K, params->nu));
}
int sc_mpclin_solve_batch_sliced(const sc_mpclin_params* params, const sc_mpc_slices* slices, const double* model, int64_t B, int32_t K,
                                 const void* X, const void* u_prev, const void* goal, const void* obs, void* u_out, int32_t* status_out,
                                 int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpclin(params, model, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (params->optimal_decay == 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "optimal_decay = 1: call sc_odmpclin_solve_batch");
    rc = sc::check_slices(slices, params->max_iter, sc_mpclin_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->max_iter, (long long)B, sc::mpclin_state_doubles(params->horizon, K, params->nu),
                                  (hipStream_t)stream, [&](const sc::ipm::Cont& ct) {
        return sc::mpclin_launch(*params, model, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, nullptr,
                                 (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "mpclin kernel launch (sliced)");
    return SC_OK;
}

int sc_odmpcvtol_solve_batch(const sc_odmpcvtol_params* params, int64_t B, int32_t K, const void* X, const void* u_prev, const void* goal,
                             const void* obs, void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    int rc = sc::check_mpcvtol(&params->mpc, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (!sc::mpcvtol_uses_wave(params->mpc, K)) return sc::fail(SC_ERR_UNSUPPORTED, "optimal-decay MPC-CBF for VTOL2D runs on the wave-per-problem kernel (kernel = 0 / 2)");
    if (!(params->p_sb[0] > 0) || !(params->p_sb[1] > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "p_sb must be > 0");
    if (B == 0) return SC_OK;
    hipError_t e = sc::odmpcvtol_wave_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, rho_out, status_out, iters_out, z_out,
                                             (hipStream_t)stream, sc::one_launch(params->mpc.max_iter));
    if (e != hipSuccess) return sc::fail_hip(e, "odmpcvtol kernel launch");
    return SC_OK;
}
size_t sc_odmpcvtol_slices_workspace_bytes(const sc_odmpcvtol_params* params, int64_t B, int32_t K) {
    if (!params || B < 0 || K < 1 || K > 16 || params->mpc.horizon < 1 || params->mpc.horizon > 64) return 0;
    return sc::slices_workspace_bytes((long long)B, sc::mpcvtol_state_doubles(params->mpc.horizon, K, true));
}
int sc_odmpcvtol_solve_batch_sliced(const sc_odmpcvtol_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K, const void* X,
                                    const void* u_prev, const void* goal, const void* obs, void* u_out, void* rho_out, int32_t* status_out,
                                    int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    int rc = sc::check_mpcvtol(&params->mpc, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (!sc::mpcvtol_uses_wave(params->mpc, K)) return sc::fail(SC_ERR_UNSUPPORTED, "optimal-decay MPC-CBF for VTOL2D runs on the wave-per-problem kernel (kernel = 0 / 2)");
    if (!(params->p_sb[0] > 0) || !(params->p_sb[1] > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "p_sb must be > 0");
    rc = sc::check_slices(slices, params->mpc.max_iter, sc_odmpcvtol_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->mpc.max_iter, (long long)B, sc::mpcvtol_state_doubles(params->mpc.horizon, K, true),
                                  (hipStream_t)stream, [&](const sc::ipm::Cont& ct) {
        return sc::odmpcvtol_wave_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, rho_out, status_out, iters_out, z_out,
                                         (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "odmpcvtol kernel launch (sliced)");
    return SC_OK;
}

size_t sc_mpcvtol_ms_workspace_bytes(int64_t B, int32_t K) {
    if (B < 0 || K < 0 || K > 16) return 0;
    return (size_t)B * (size_t)((8 * (K <= 8 ? 8 : 16) + 12) * 64) * sizeof(double);       // Wave<KS>::R_SLOTS x 64 lanes (csrc/mpc_vtol_ms.hip)
}

static int check_ms_common(const sc_mpcvtol_params* params, const sc_ipopt_params* ipopt, int64_t B, int32_t K) {
    if (!params || !ipopt) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params / ipopt is NULL");
    if (B < 0 || K < 0 || K > 16) return sc::fail(SC_ERR_UNSUPPORTED, "the multiple-shooting kernel serves 0 <= K <= 16");
    if (params->horizon < 1 || params->horizon > 62) return sc::fail(SC_ERR_UNSUPPORTED, "the multiple-shooting kernel serves 1 <= horizon <= 62 (one stage per lane + the terminal state)");
    if (params->io_dtype != SC_DTYPE_F32 && params->io_dtype != SC_DTYPE_F64) return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype");
    if (!(params->dt > 0.0) || !(params->beta > 0.0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "dt and beta must be positive");
    if (ipopt->max_iter < 0 || ipopt->acceptable_iter < 1 || !(ipopt->tol > 0.0) || !(ipopt->mu_init > 0.0) || !(ipopt->tau_min > 0.0 && ipopt->tau_min < 1.0) ||
        !(ipopt->alpha_red_factor > 0.0 && ipopt->alpha_red_factor < 1.0) || !(ipopt->perturb_inc_fact > 1.0) || !(ipopt->perturb_inc_fact_first > 1.0) ||
        !(ipopt->first_hessian_perturbation > 0.0) || !(ipopt->s_max > 0.0) || !(ipopt->kappa_sigma > 1.0))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "sc_ipopt_params out of range");
    if (ipopt->stall_iter < 0 || ipopt->floor_iter < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "sc_ipopt_params: stall_iter and floor_iter must be >= 0");
    if (ipopt->resto_workspace) {
        if ((size_t)ipopt->resto_workspace_bytes < sc_mpcvtol_ms_workspace_bytes(B, K)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "resto_workspace smaller than sc_mpcvtol_ms_workspace_bytes(B, K)");
        if (!(ipopt->resto_penalty_parameter > 0.0) || !(ipopt->resto_proximity_weight >= 0.0) || !(ipopt->required_infeasibility_reduction > 0.0 && ipopt->required_infeasibility_reduction < 1.0))
            return sc::fail(SC_ERR_INVALID_ARGUMENT, "sc_ipopt_params: restoration options out of range");
    }
    return SC_OK;
}

int sc_mpcvtol_ms_solve_batch(const sc_mpcvtol_params* params, const sc_ipopt_params* ipopt, int64_t B, int32_t K, const void* X, const void* u_prev,
                              const void* goal, const void* obs, void* u_out, int32_t* status_out, int32_t* iters_out, void* plan_out, double* trace_out,
                              void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_ms_common(params, ipopt, B, K);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    if (!X || !u_prev || !goal || !u_out || !status_out || (K > 0 && !obs)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL buffer");
    hipError_t e = sc::mpcvtol_ms_launch(*params, *ipopt, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out,
                                         (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "mpcvtol multiple-shooting kernel launch");
    return SC_OK;
}

int sc_odmpcvtol_ms_solve_batch(const sc_odmpcvtol_params* params, const sc_ipopt_params* ipopt, int64_t B, int32_t K, const void* X, const void* u_prev,
                                const void* goal, const void* obs, void* u_out, void* rho_out, int32_t* status_out, int32_t* iters_out, void* plan_out,
                                double* trace_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    int rc = check_ms_common(&params->mpc, ipopt, B, K);
    if (rc != SC_OK) return rc;
    if (!(params->p_sb[0] > 0) || !(params->p_sb[1] > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "p_sb must be > 0");
    if (B == 0) return SC_OK;
    if (!X || !u_prev || !goal || !u_out || !status_out || (K > 0 && !obs)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL buffer");
    hipError_t e = sc::odmpcvtol_ms_launch(*params, *ipopt, (long long)B, (int)K, X, u_prev, goal, obs, u_out, rho_out, status_out, iters_out, plan_out,
                                           trace_out, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "odmpcvtol multiple-shooting kernel launch");
    return SC_OK;
}

static int check_ipopt_options(const sc_ipopt_params* ipopt) {
    if (!ipopt) return sc::fail(SC_ERR_INVALID_ARGUMENT, "ipopt is NULL");
    if (ipopt->max_iter < 0 || ipopt->acceptable_iter < 1 || !(ipopt->tol > 0.0) || !(ipopt->mu_init > 0.0) || !(ipopt->tau_min > 0.0 && ipopt->tau_min < 1.0) ||
        !(ipopt->alpha_red_factor > 0.0 && ipopt->alpha_red_factor < 1.0) || !(ipopt->perturb_inc_fact > 1.0) || !(ipopt->perturb_inc_fact_first > 1.0) ||
        !(ipopt->first_hessian_perturbation > 0.0) || !(ipopt->s_max > 0.0) || !(ipopt->kappa_sigma > 1.0))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "sc_ipopt_params out of range");
    if (ipopt->stall_iter < 0 || ipopt->floor_iter < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "sc_ipopt_params: stall_iter and floor_iter must be >= 0");
    if (!(ipopt->resto_penalty_parameter > 0.0) || !(ipopt->resto_proximity_weight >= 0.0) || !(ipopt->required_infeasibility_reduction > 0.0 && ipopt->required_infeasibility_reduction < 1.0))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "sc_ipopt_params: restoration options out of range");
    return SC_OK;
}

size_t sc_mpccbf_ms_workspace_bytes(int64_t B) { return B < 0 ? 0 : sc::mpcdu_ms_order_bytes((long long)B); }

size_t sc_mpccbf_ms_lds_bytes(int32_t horizon, int32_t K) {
    if (horizon < 1 || horizon > 62 || K < 1 || K > 16) return 0;
    return sc::mpcdu_ms_lds_bytes(horizon, K, SC_MODEL_KINEMATIC_BICYCLE2D, 1);      /* (the largest layout: 96 B per stage more than the unicycle's) */
}

int sc_mpccbf_ms_solve_batch(const sc_mpccbf_params* params, const sc_ipopt_params* ipopt, int64_t B, int32_t K, const void* X, const void* u_prev,
                             const void* goal, const void* obs, void* u_out, int32_t* status_out, int32_t* iters_out, void* plan_out, double* trace_out,
                             void* stream) {
    sc::DeviceGuard on_device(stream, X);
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    if (params->model_id != SC_MODEL_DYNAMIC_UNICYCLE2D && params->model_id != SC_MODEL_DOUBLE_INTEGRATOR2D && params->model_id != SC_MODEL_KINEMATIC_BICYCLE2D &&
        params->model_id != SC_MODEL_UNICYCLE2D && params->model_id != SC_MODEL_SINGLE_INTEGRATOR2D)
        return sc::fail(SC_ERR_UNSUPPORTED, "the multiple-shooting MPC-CBF kernel is built for DynamicUnicycle2D, Unicycle2D, SingleIntegrator2D, DoubleIntegrator2D and KinematicBicycle2D");
    if (params->model_id == SC_MODEL_KINEMATIC_BICYCLE2D && (!(params->rear_ax_dist > 0.0) || !(params->v_min <= params->v_max)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "KinematicBicycle2D: rear_ax_dist must be positive and v_min <= v_max");
    if (B < 0 || K < 1 || K > 16) return sc::fail(SC_ERR_UNSUPPORTED, "the multiple-shooting kernel serves 1 <= K <= 16 (pad with [1000,1000,0,...] rows like update_tvp)");
    if (B > 0x7fffffffLL) return sc::fail(SC_ERR_UNSUPPORTED, "B too large for one launch");
    if (params->horizon < 1 || params->horizon > 62) return sc::fail(SC_ERR_UNSUPPORTED, "the multiple-shooting kernel serves 1 <= horizon <= 62 (one stage per lane + the terminal state)");
    if (params->superellipsoid_rows && params->model_id != SC_MODEL_DYNAMIC_UNICYCLE2D && params->model_id != SC_MODEL_DOUBLE_INTEGRATOR2D)
        return sc::fail(SC_ERR_UNSUPPORTED, "superellipsoid rows: served for DynamicUnicycle2D and DoubleIntegrator2D (SingleIntegrator2D: sc_mpclin_solve_batch; the other robots' DT barriers have no such branch)");
    if (sc::mpcdu_ms_lds_bytes(params->horizon, K, params->model_id, params->superellipsoid_rows) > 160 * 1024) return sc::fail(SC_ERR_UNSUPPORTED, "horizon x obstacles does not fit the 160 KiB LDS of one CU");
    if (params->io_dtype != SC_DTYPE_F32 && params->io_dtype != SC_DTYPE_F64) return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype");
    if (!(params->dt > 0.0) || !(params->beta > 0.0) || !(params->u_max[0] > 0.0) || !(params->u_max[1] > 0.0) || !(params->v_max > 0.0))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "dt, beta, u_max and v_max must be positive");
    int rc = check_ipopt_options(ipopt);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    if (!X || !u_prev || !goal || !obs || !u_out || !status_out) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL buffer");
    // the optional launch-order workspace rides in the fields the VTOL2D entry uses for its restoration state (this kernel keeps that in LDS)
    void* order_ws = nullptr;
    if (ipopt->resto_workspace) {
        if ((size_t)ipopt->resto_workspace_bytes < sc::mpcdu_ms_order_bytes((long long)B))
            return sc::fail(SC_ERR_INVALID_ARGUMENT, "resto_workspace (launch-order workspace here) smaller than sc_mpccbf_ms_workspace_bytes(B)");
        order_ws = ipopt->resto_workspace;
    }
    hipError_t e = sc::mpcdu_ms_launch(*params, *ipopt, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, plan_out, trace_out,
                                       order_ws, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "mpccbf multiple-shooting kernel launch");
    return SC_OK;
}

size_t sc_mpcvtol_slices_workspace_bytes(const sc_mpcvtol_params* params, int64_t B, int32_t K) {
    if (!params || B < 0 || K < 1 || K > 16 || params->horizon < 1 || params->horizon > 64) return 0;
    return sc::slices_workspace_bytes((long long)B, sc::mpcvtol_state_doubles(params->horizon, K));
}
int sc_mpcvtol_solve_batch_sliced(const sc_mpcvtol_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K, const void* X,
                                  const void* u_prev, const void* goal, const void* obs, void* u_out, int32_t* status_out,
                                  int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpcvtol(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (!sc::mpcvtol_uses_wave(*params, K)) return sc::fail(SC_ERR_UNSUPPORTED, "continuation launches are served by the wave-per-problem kernel (kernel = 0 / 2)");
    rc = sc::check_slices(slices, params->max_iter, sc_mpcvtol_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->max_iter, (long long)B, sc::mpcvtol_state_doubles(params->horizon, K), (hipStream_t)stream,
                                  [&](const sc::ipm::Cont& ct) {
        return sc::mpcvtol_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, nullptr,
                                  (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "mpcvtol kernel launch (sliced)");
    return SC_OK;
}

size_t sc_mpccbf_slices_workspace_bytes(const sc_mpccbf_params* params, int64_t B, int32_t K) {
    if (!params || B < 0 || K < 1 || params->horizon < 1 || params->horizon > SC_MPCCBF_MAX_HORIZON) return 0;
    return sc::slices_workspace_bytes((long long)B, sc::mpccbf_state_doubles(params->horizon, K, false));
}

int sc_mpccbf_solve_batch_sliced(const sc_mpccbf_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K, const void* X,
                                 const void* u_prev, const void* goal, const void* obs, void* u_out, int32_t* status_out,
                                 int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpccbf(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    rc = sc::check_slices(slices, params->max_iter, sc_mpccbf_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->max_iter, (long long)B, sc::mpccbf_state_doubles(params->horizon, K, false),
                                  (hipStream_t)stream, [&](const sc::ipm::Cont& ct) {
        return sc::mpccbf_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out,
                                 (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "mpccbf kernel launch (sliced)");
    return SC_OK;
}

int sc_mpccbf_solve_batch_host(const sc_mpccbf_params* params, int64_t B, int32_t K, const void* X,
                               const void* u_prev, const void* goal, const void* obs, void* u_out,
                               int32_t* status_out, int32_t* iters_out, void* z_out, int device) {
    int rc = sc::check_mpccbf(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const size_t es = params->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t n = 2 * (size_t)params->horizon;
    const size_t nX = (size_t)B * 4 * es, nU = (size_t)B * 2 * es, nG = nU;
    const size_t nO = (params->obs_shared ? (size_t)K * 7 : (size_t)B * K * 7) * es;
    const size_t nS = (size_t)B * 4, nZ = (size_t)B * n * es;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oX = 0, oU = oX + up(nX), oG = oU + up(nU), oO = oG + up(nG), oUo = oO + up(nO), oS = oUo + up(nU),
                 oI = oS + up(nS), oZ = oI + up(nS), total = oZ + up(nZ);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    do {
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_prev, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oG, goal, nG, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        sc::ipm::Cont ct{};
        ct.it_stop = params->max_iter;
        e = sc::mpccbf_launch(*params, (long long)B, (int)K, d + oX, d + oU, d + oG, d + oO, d + oUo, (int*)(d + oS),
                              (int*)(d + oI), z_out ? d + oZ : nullptr, s, ct);
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (iters_out && (e = hipMemcpyAsync(iters_out, d + oI, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (z_out && (e = hipMemcpyAsync(z_out, d + oZ, nZ, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    rc = SC_OK;
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_mpccbf_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

static int check_odmpccbf(const sc_odmpccbf_params* q, int64_t B, int32_t K, const void* X, const void* u_prev,
                          const void* goal, const void* obs, const void* u_out, const void* status_out) {
    if (!q) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    int rc = sc::check_mpccbf(&q->mpc, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    // DynamicUnicycle2D: the reference's problem (optimal_decay_mpc_cbf.py).  Unicycle2D: BASELINE config 5's EXTENSION -- the
    // reference class rejects that model (:19-20); rows d_h + alpha rho_k h_k, one live decay variable per stage
    // (oracle/od_mpc_rd1.py).  The host classes only reach it with extension=True.
    if (q->mpc.model_id != SC_MODEL_DYNAMIC_UNICYCLE2D && q->mpc.model_id != SC_MODEL_UNICYCLE2D)
        return sc::fail(SC_ERR_UNSUPPORTED, "optimal-decay MPC-CBF is built for DynamicUnicycle2D and (extension) Unicycle2D");
    if (sc::odmpccbf_lds_bytes(q->mpc.horizon, K) > 160 * 1024)
        return sc::fail(SC_ERR_UNSUPPORTED, "horizon x obstacles does not fit the 160 KiB LDS of one CU");
    if (!(q->p_sb[0] > 0) || !(q->p_sb[1] > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "p_sb must be > 0");
    return SC_OK;
}

int sc_odmpccbf_solve_batch(const sc_odmpccbf_params* params, int64_t B, int32_t K, const void* X, const void* u_prev,
                            const void* goal, const void* obs, void* u_out, void* rho_out, int32_t* status_out,
                            int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_odmpccbf(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::odmpccbf_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, rho_out, status_out,
                                       iters_out, z_out, (hipStream_t)stream, sc::one_launch(params->mpc.max_iter));
    if (e != hipSuccess) return sc::fail_hip(e, "odmpccbf kernel launch");
    return SC_OK;
}

/* ---- continuation launches of the optimal-decay families (round 5): the same kernels hand their state over (decay variables included) ---- */
size_t sc_odmpccbf_slices_workspace_bytes(const sc_odmpccbf_params* params, int64_t B, int32_t K) {
    if (!params || B < 0 || K < 1 || params->mpc.horizon < 1) return 0;
    return sc::slices_workspace_bytes((long long)B, sc::mpccbf_state_doubles(params->mpc.horizon, K, true));
}
int sc_odmpccbf_solve_batch_sliced(const sc_odmpccbf_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K, const void* X,
                                   const void* u_prev, const void* goal, const void* obs, void* u_out, void* rho_out, int32_t* status_out,
                                   int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_odmpccbf(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    rc = sc::check_slices(slices, params->mpc.max_iter, sc_odmpccbf_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->mpc.max_iter, (long long)B, sc::mpccbf_state_doubles(params->mpc.horizon, K, true),
                                  (hipStream_t)stream, [&](const sc::ipm::Cont& ct) {
        return sc::odmpccbf_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, rho_out, status_out, iters_out, z_out,
                                   (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "odmpccbf kernel launch (sliced)");
    return SC_OK;
}
size_t sc_odmpcgn_slices_workspace_bytes(const sc_odmpcgn_params* params, int64_t B, int32_t K) {
    if (!params) return 0;
    return sc_mpcgn_slices_workspace_bytes(&params->mpc, B, K);
}
int sc_odmpcgn_solve_batch_sliced(const sc_odmpcgn_params* params, const sc_mpc_slices* slices, int64_t B, int32_t K, const void* X,
                                  const void* u_prev, const void* goal, const void* obs, void* u_out, void* rho_out, int32_t* status_out,
                                  int32_t* iters_out, void* z_out, void* stream) {
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpcgn(&params->mpc, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (params->mpc.model_id != SC_MODEL_KINEMATIC_BICYCLE2D && params->mpc.model_id != SC_MODEL_QUAD2D)
        return sc::fail(SC_ERR_UNSUPPORTED, "optimal-decay MPC-CBF on this entry point: KinematicBicycle2D and Quad2D");
    if (!(params->p_sb[0] > 0) || !(params->p_sb[1] > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "p_sb must be > 0");
    if (sc::odmpcgn_lds_bytes(params->mpc.model_id, params->mpc.horizon, K) > 160 * 1024)
        return sc::fail(SC_ERR_UNSUPPORTED, "horizon x obstacles does not fit the 160 KiB LDS of one CU");
    rc = sc::check_slices(slices, params->mpc.max_iter, sc_odmpcgn_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->mpc.max_iter, (long long)B, sc::mpcgn_state_doubles(params->mpc.horizon, K), (hipStream_t)stream,
                                  [&](const sc::ipm::Cont& ct) {
        return sc::odmpcgn_launch(*params, (long long)B, (int)K, X, u_prev, goal, obs, u_out, rho_out, status_out, iters_out, z_out,
                                  (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "odmpcgn kernel launch (sliced)");
    return SC_OK;
}
int sc_odmpclin_solve_batch_sliced(const sc_mpclin_params* params, const sc_mpc_slices* slices, const double* model, int64_t B, int32_t K,
                                   const void* X, const void* u_prev, const void* goal, const void* obs, void* u_out, void* rho_out,
                                   int32_t* status_out, int32_t* iters_out, void* z_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = sc::check_mpclin(params, model, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (params->optimal_decay != 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "optimal_decay != 1: call sc_mpclin_solve_batch_sliced");
    rc = sc::check_slices(slices, params->max_iter, sc_mpclin_slices_workspace_bytes(params, B, K));
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::run_slices(slices, params->max_iter, (long long)B, sc::mpclin_state_doubles(params->horizon, K, params->nu),
                                  (hipStream_t)stream, [&](const sc::ipm::Cont& ct) {
        return sc::mpclin_launch(*params, model, (long long)B, (int)K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, rho_out,
                                 (hipStream_t)stream, ct);
    });
    if (e != hipSuccess) return sc::fail_hip(e, "mpclin (optimal decay) kernel launch (sliced)");
    return SC_OK;
}

int sc_odmpccbf_solve_batch_host(const sc_odmpccbf_params* params, int64_t B, int32_t K, const void* X,
                                 const void* u_prev, const void* goal, const void* obs, void* u_out, void* rho_out,
                                 int32_t* status_out, int32_t* iters_out, void* z_out, int device) {
    int rc = check_odmpccbf(params, B, K, X, u_prev, goal, obs, u_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const sc_mpccbf_params* mp = &params->mpc;
    const size_t es = mp->io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t n = 2 * (size_t)mp->horizon;
    const size_t nX = (size_t)B * 4 * es, nU = (size_t)B * 2 * es, nG = nU;
    const size_t nO = (mp->obs_shared ? (size_t)K * 7 : (size_t)B * K * 7) * es;
    const size_t nS = (size_t)B * 4, nZ = (size_t)B * n * es;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oX = 0, oU = oX + up(nX), oG = oU + up(nU), oO = oG + up(nG), oUo = oO + up(nO), oS = oUo + up(nU),
                 oI = oS + up(nS), oZ = oI + up(nS), oR = oZ + up(nZ), total = oR + up(nZ);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    do {
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_prev, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oG, goal, nG, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        e = sc::odmpccbf_launch(*params, (long long)B, (int)K, d + oX, d + oU, d + oG, d + oO, d + oUo,
                                rho_out ? d + oR : nullptr, (int*)(d + oS), (int*)(d + oI), z_out ? d + oZ : nullptr, s,
                                sc::one_launch(params->mpc.max_iter));
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (iters_out && (e = hipMemcpyAsync(iters_out, d + oI, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (z_out && (e = hipMemcpyAsync(z_out, d + oZ, nZ, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (rho_out && (e = hipMemcpyAsync(rho_out, d + oR, nZ, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    rc = SC_OK;
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_odmpccbf_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

int sc_tracking_rollout_batch(const sc_tracking_params* params, int64_t B, int32_t M, void* X, const void* waypoints,
                              const int32_t* n_wp, int32_t* wp_index, int32_t* state_machine, void* goal,
                              void* obs_table, void* u_last, int32_t* ret, int32_t* ret_step, void* traj_X,
                              void* traj_U, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    const sc_cbfqp_params* q = &params->qp;
    if (B < 0 || M < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "B < 0 or M < 0");
    const bool integrator = q->model_id == SC_MODEL_SINGLE_INTEGRATOR2D || q->model_id == SC_MODEL_DOUBLE_INTEGRATOR2D;
    if (q->model_id < 0 || (q->model_id > SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF && q->model_id != SC_MODEL_UNICYCLE2D && !integrator))
        return sc::fail(SC_ERR_UNSUPPORTED, "the fused rollout is built for the unicycle / bicycle / integrator models");
    if (integrator && params->enable_rotation)
        return sc::fail(SC_ERR_UNSUPPORTED, "the integrators' rotate state needs an attitude controller: enable_rotation must be 0");
    if (q->io_dtype != SC_DTYPE_F32 && q->io_dtype != SC_DTYPE_F64)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (params->num_constraints < 1 || params->num_constraints > SC_TRACKING_MAX_CONSTRAINTS)
        return sc::fail(SC_ERR_UNSUPPORTED, "num_constraints outside [1, SC_TRACKING_MAX_CONSTRAINTS]");
    if (params->n_steps < 0 || params->max_waypoints < 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "n_steps < 0 or max_waypoints < 1");
    if (!(q->dt > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "dt must be > 0");
    if (q->model_id != SC_MODEL_DYNAMIC_UNICYCLE2D && q->model_id != SC_MODEL_UNICYCLE2D && !integrator &&
        (!(q->rear_ax_dist > 0) || !(params->wheel_base > 0)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "rear_ax_dist and wheel_base must be > 0 for the KinematicBicycle2D family");
    if ((size_t)M * 7 * 8 > 160 * 1024) return sc::fail(SC_ERR_UNSUPPORTED, "obstacle table does not fit the LDS");
    if (B > 0 && (!X || !waypoints || !n_wp || !wp_index || !state_machine || !goal || !u_last || !ret || !ret_step || (M > 0 && !obs_table)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B == 0 || params->n_steps == 0) return SC_OK;
    hipError_t e = sc::tracking_launch(*params, (long long)B, (int)M, X, waypoints, n_wp, wp_index, state_machine, goal,
                                       obs_table, u_last, ret, ret_step, traj_X, traj_U, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "tracking kernel launch");
    return SC_OK;
}

static int check_tracking_split(const sc_tracking_params* params, int64_t B, int32_t M) {
    if (!params) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    const sc_cbfqp_params* q = &params->qp;
    if (B < 0 || M < 0) return sc::fail(SC_ERR_INVALID_ARGUMENT, "B < 0 or M < 0");
    const bool integrator = q->model_id == SC_MODEL_SINGLE_INTEGRATOR2D || q->model_id == SC_MODEL_DOUBLE_INTEGRATOR2D;
    // the C3BF / DPCBF bicycles differ from KinematicBicycle2D in their barrier only: select / apply run the bicycle's kernels for them
    const bool bicycle = q->model_id == SC_MODEL_KINEMATIC_BICYCLE2D || q->model_id == SC_MODEL_KINEMATIC_BICYCLE2D_C3BF ||
                         q->model_id == SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF;
    if (q->model_id != SC_MODEL_DYNAMIC_UNICYCLE2D && !bicycle && q->model_id != SC_MODEL_UNICYCLE2D && !integrator)
        return sc::fail(SC_ERR_UNSUPPORTED, "select / apply are built for DynamicUnicycle2D, Unicycle2D, the KinematicBicycle2D family and the integrators");
    if (integrator && params->enable_rotation)
        return sc::fail(SC_ERR_UNSUPPORTED, "the integrators' rotate state needs an attitude controller: enable_rotation must be 0");
    if (q->io_dtype != SC_DTYPE_F32 && q->io_dtype != SC_DTYPE_F64)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (params->num_constraints < 1 || params->num_constraints > SC_TRACKING_MAX_CONSTRAINTS)
        return sc::fail(SC_ERR_UNSUPPORTED, "num_constraints outside [1, SC_TRACKING_MAX_CONSTRAINTS]");
    if (params->dyn_obs) return sc::fail(SC_ERR_UNSUPPORTED, "select / apply take a static obstacle table");
    if (!(q->dt > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "dt must be > 0");
    if (q->model_id != SC_MODEL_DYNAMIC_UNICYCLE2D && q->model_id != SC_MODEL_UNICYCLE2D && !integrator &&
        (!(q->rear_ax_dist > 0) || !(params->wheel_base > 0)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "rear_ax_dist and wheel_base must be > 0 for the KinematicBicycle2D family");
    if ((size_t)M * 7 * 8 > 160 * 1024) return sc::fail(SC_ERR_UNSUPPORTED, "obstacle table does not fit the LDS");
    return SC_OK;
}

int sc_tracking_select_batch(const sc_tracking_params* params, int64_t B, int32_t M, const void* X, const void* waypoints,
                             const int32_t* n_wp, int32_t* wp_index, int32_t* state_machine, void* goal,
                             const void* obs_table, const int32_t* ret, void* obs_out, void* goal_out, void* u_ref_out,
                             int32_t* track_out, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_tracking_split(params, B, M);
    if (rc != SC_OK) return rc;
    if (params->max_waypoints < 1) return sc::fail(SC_ERR_INVALID_ARGUMENT, "max_waypoints < 1");
    if (B > 0 && (!X || !waypoints || !n_wp || !wp_index || !state_machine || !goal || !ret || !obs_out || !goal_out ||
                  !u_ref_out || !track_out || (M > 0 && !obs_table)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B == 0) return SC_OK;
    hipError_t e = sc::tracking_select_launch(*params, (long long)B, (int)M, X, waypoints, n_wp, wp_index, state_machine, goal,
                                              obs_table, ret, obs_out, goal_out, u_ref_out, track_out, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "tracking select kernel launch");
    return SC_OK;
}

int sc_tracking_apply_batch(const sc_tracking_params* params, int64_t B, int32_t M, int32_t step_index, void* X,
                            const int32_t* state_machine, const void* goal, const void* obs_table, const void* u,
                            const int32_t* u_status, void* u_last, int32_t* ret, int32_t* ret_step, void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_tracking_split(params, B, M);
    if (rc != SC_OK) return rc;
    if (B > 0 && (!X || !state_machine || !goal || !u || !u_last || !ret || !ret_step || (M > 0 && !obs_table)))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (B == 0) return SC_OK;
    hipError_t e = sc::tracking_apply_launch(*params, (long long)B, (int)M, (int)step_index, X, state_machine, goal, obs_table,
                                             u, u_status, u_last, ret, ret_step, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "tracking apply kernel launch");
    return SC_OK;
}

static int check_od(const sc_odcbfqp_params* p, int64_t B, const void* X, const void* u_ref, const void* obs,
                    const void* u_out, const void* w_out, const void* st) {
    if (!p) return sc::fail(SC_ERR_INVALID_ARGUMENT, "params is NULL");
    int rc = sc::check_cbfqp(&p->qp, B, 1, X, u_ref, obs, u_out, st);
    if (rc != SC_OK) return rc;
    if (p->qp.model_id > SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF && p->qp.model_id != SC_MODEL_QUAD2D)
        return sc::fail(SC_ERR_UNSUPPORTED, "optimal-decay CBF-QP is built for the unicycle / bicycle models and Quad2D");
    if (!(p->p_sb[0] > 0) || !(p->p_sb[1] > 0)) return sc::fail(SC_ERR_INVALID_ARGUMENT, "p_sb must be > 0");
    if (B > 0 && !w_out) return sc::fail(SC_ERR_INVALID_ARGUMENT, "omega_out is NULL");
    return SC_OK;
}

int sc_odcbfqp_solve_batch(const sc_odcbfqp_params* params, int64_t B, const void* X, const void* u_ref, const void* obs,
                           const int32_t* has_obs, void* u_out, void* omega_out, int32_t* status_out, void* h_out,
                           void* stream) {
    sc::DeviceGuard on_device(stream, X);
    int rc = check_od(params, B, X, u_ref, obs, u_out, omega_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = sc::odcbfqp_launch(*params, (long long)B, X, u_ref, obs, has_obs, u_out, omega_out, status_out, h_out,
                                      (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "odcbfqp kernel launch");
    return SC_OK;
}

int sc_odcbfqp_solve_batch_host(const sc_odcbfqp_params* params, int64_t B, const void* X, const void* u_ref,
                                const void* obs, const int32_t* has_obs, void* u_out, void* omega_out,
                                int32_t* status_out, void* h_out, int device) {
    int rc = check_od(params, B, X, u_ref, obs, u_out, omega_out, status_out);
    if (rc != SC_OK) return rc;
    if (B == 0) return SC_OK;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return sc::fail_hip(e, "hipSetDevice");
    const size_t es = params->qp.io_dtype == SC_DTYPE_F64 ? 8 : 4;
    const size_t nX = (size_t)B * (params->qp.model_id == SC_MODEL_QUAD2D ? 6 : 4) * es, nU = (size_t)B * 2 * es, nO = (size_t)B * 7 * es,
                 nS = (size_t)B * 4, nH = (size_t)B * es;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t oX = 0, oU = oX + up(nX), oO = oU + up(nU), oN = oO + up(nO), oUo = oN + up(nS), oW = oUo + up(nU),
                 oS = oW + up(nU), oH = oS + up(nS), total = oH + up(nH);
    unsigned char* d = nullptr;
    e = hipMalloc((void**)&d, total);
    if (e != hipSuccess) return sc::fail_hip(e, "hipMalloc");
    hipStream_t s = nullptr;
    do {
        if ((e = hipMemcpyAsync(d + oX, X, nX, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oU, u_ref, nU, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d + oO, obs, nO, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if (has_obs && (e = hipMemcpyAsync(d + oN, has_obs, nS, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        e = sc::odcbfqp_launch(*params, (long long)B, d + oX, d + oU, d + oO, has_obs ? (const int*)(d + oN) : nullptr,
                               d + oUo, d + oW, (int*)(d + oS), h_out ? d + oH : nullptr, s);
        if (e != hipSuccess) break;
        if ((e = hipMemcpyAsync(u_out, d + oUo, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(omega_out, d + oW, nU, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(status_out, d + oS, nS, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (h_out && (e = hipMemcpyAsync(h_out, d + oH, nH, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    rc = SC_OK;
    if (e != hipSuccess) rc = sc::fail_hip(e, "sc_odcbfqp_solve_batch_host");
    (void)hipFree(d);
    return rc;
}

int sc_neighbor_obstacles_batch(int32_t io_dtype, int64_t B_all, int64_t first_local, int64_t B_local, int32_t K,
                                double neighbour_radius, const void* X_all, void* obs_out, void* stream) {
    sc::DeviceGuard on_device(stream, X_all);
    if (io_dtype != SC_DTYPE_F32 && io_dtype != SC_DTYPE_F64)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (B_all < 0 || B_local < 0 || first_local < 0 || first_local + B_local > B_all)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "local range outside [0, B_all)");
    if (K < 1 || K > SC_CBFQP_MAX_OBS) return sc::fail(SC_ERR_UNSUPPORTED, "K outside [1, SC_CBFQP_MAX_OBS]");
    if (B_all > 0x7fffffffLL) return sc::fail(SC_ERR_UNSUPPORTED, "more than 2^31 agents");
    if (B_local == 0) return SC_OK;
    if (!X_all || !obs_out) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    hipError_t e = sc::neighbors_launch(io_dtype, (long long)B_all, (long long)first_local, (long long)B_local, (int)K,
                                        neighbour_radius, X_all, obs_out, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "neighbour kernel launch");
    return SC_OK;
}

size_t sc_neighbor_workspace_bytes(int32_t io_dtype, int64_t B_all, int64_t B_local, int32_t K) {
    if ((io_dtype != SC_DTYPE_F32 && io_dtype != SC_DTYPE_F64) || B_all < 0 || B_local < 0 || K < 1 || K > SC_CBFQP_MAX_OBS) return 0;
    return sc::neighbors_workspace_bytes(io_dtype, (long long)B_all, (long long)B_local, (int)K);
}

int sc_neighbor_obstacles_batch_ws(int32_t io_dtype, int64_t B_all, int64_t first_local, int64_t B_local, int32_t K,
                                   double neighbour_radius, const void* X_all, void* obs_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    sc::DeviceGuard on_device(stream, X_all);
    if (io_dtype != SC_DTYPE_F32 && io_dtype != SC_DTYPE_F64)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "io_dtype must be SC_DTYPE_F32 or SC_DTYPE_F64");
    if (B_all < 0 || B_local < 0 || first_local < 0 || first_local + B_local > B_all)
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "local range outside [0, B_all)");
    if (K < 1 || K > SC_CBFQP_MAX_OBS) return sc::fail(SC_ERR_UNSUPPORTED, "K outside [1, SC_CBFQP_MAX_OBS]");
    if (B_all > 0x7fffffffLL) return sc::fail(SC_ERR_UNSUPPORTED, "more than 2^31 agents");
    if (B_local == 0) return SC_OK;
    if (!X_all || !obs_out || !workspace) return sc::fail(SC_ERR_INVALID_ARGUMENT, "NULL data pointer");
    if (workspace_bytes < sc::neighbors_workspace_bytes(io_dtype, (long long)B_all, (long long)B_local, (int)K))
        return sc::fail(SC_ERR_INVALID_ARGUMENT, "workspace smaller than sc_neighbor_workspace_bytes()");
    hipError_t e = sc::neighbors_split_launch(io_dtype, (long long)B_all, (long long)first_local, (long long)B_local, (int)K,
                                              neighbour_radius, X_all, obs_out, workspace, (hipStream_t)stream);
    if (e != hipSuccess) return sc::fail_hip(e, "neighbour kernels launch");
    return SC_OK;
}

}  // extern "C"
