// MPC-CBF for VTOL2D: the lane-per-problem kernel around mpc_vtol_solver.hpp.
//
// A problem's work arrays (about 12.4 k doubles for N = 30, K = 8: iterate, rows, stage Jacobians and blocks, Riccati gains) live in the
// caller's workspace, entry i of problem b at  ws[i * B + b]: neighbouring lanes touch neighbouring addresses in every pass, so a wave's
// access is one or a few full cache lines.  The card's 288 GB make the footprint (0.4 GB per 4096 problems) a non-issue; its latency is
// what a lane pays, and the answer to that is occupancy -- a launch of B problems runs `lanes` problems per 64-thread block
// (16 by default up to 16384 problems: 4096 problems are then 256 blocks, one per CU, instead of 64 full waves on a quarter of the chip).
// Each lane walks its own interior point; a wave ends with its slowest lane.  Kernel 11 in DESIGN.md.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/safe_control_amd.h"
#define SC_VTOL_WITH_C_PARAMS
#include "mpc_vtol_solver.hpp"
#include "mpc_cont.hpp"

namespace sc {

constexpr int VTOL_KMAX = 16;

struct LaneMem {
    double* base;               // &ws[b]
    long long stride;           // B
    __device__ double& operator()(int i) const { return base[(long long)i * stride]; }
};
struct LaneObs {
    const double* o;            // K x 3 in private memory
    __device__ double operator()(int j, int c) const { return o[3 * j + c]; }
};

template <typename TIO>
__global__ void __launch_bounds__(64) mpcvtol_kernel(const vtol::Params P, long long B, int lanes, int obs_shared, const TIO* __restrict__ X,
                                                     const TIO* __restrict__ u_prev, const TIO* __restrict__ goal, const TIO* __restrict__ obs,
                                                     TIO* __restrict__ u_out, int* __restrict__ status_out, int* __restrict__ iters_out,
                                                     TIO* __restrict__ z_out, double* __restrict__ ws) {
    if ((int)threadIdx.x >= lanes) return;
    const long long b = (long long)blockIdx.x * lanes + threadIdx.x;
    if (b >= B) return;
    double oc[3 * VTOL_KMAX];
    const TIO* ob = obs + (obs_shared ? 0 : b * P.K * 7);
    for (int j = 0; j < P.K; ++j) { oc[3 * j] = (double)ob[7 * j]; oc[3 * j + 1] = (double)ob[7 * j + 1]; oc[3 * j + 2] = (double)ob[7 * j + 2]; }
    vtol::Solver<LaneMem, LaneObs> S(P, LaneMem{ws + b, B}, LaneObs{oc});
    for (int i = 0; i < vtol::NX; ++i) S.x0[i] = (double)X[b * vtol::NX + i];
    for (int j = 0; j < vtol::NU; ++j) S.uprev[j] = (double)u_prev[b * vtol::NU + j];
    S.xg[0] = (double)goal[b * 2]; S.xg[1] = (double)goal[b * 2 + 1];
    int st, it;
    S.solve(st, it);
    for (int j = 0; j < vtol::NU; ++j) u_out[b * vtol::NU + j] = (TIO)S.W(S.L.z + j);
    status_out[b] = st;
    if (iters_out) iters_out[b] = it;
    if (z_out) for (int i = 0; i < S.L.n; ++i) z_out[b * S.L.n + i] = (TIO)S.W(S.L.z + i);
}

hipError_t mpcvtol_wave_launch(const sc_mpcvtol_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                               const void* obs, void* u_out, int* status_out, int* iters_out, void* z_out, hipStream_t stream,
                               const ipm::Cont& ct);

// which kernel serves (p, K): the wave-per-problem kernel (mpc_vtol_wave.hip: one stage per lane, the stage's rows in registers, instantiated
// for 8 and 16 row slots) unless p.kernel = 1 asks for one problem per lane out of the workspace
bool mpcvtol_uses_wave(const sc_mpcvtol_params& p, int K) {
    if (p.kernel == 1) return false;
    return K <= 16 && p.horizon <= 64;
}

size_t mpcvtol_workspace_bytes(const sc_mpcvtol_params& p, long long B, int K) {
    if (mpcvtol_uses_wave(p, K)) return 0;
    vtol::Layout L(p.horizon, K);
    return (size_t)L.total * (size_t)B * sizeof(double);
}

hipError_t mpcvtol_launch(const sc_mpcvtol_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                          const void* obs, void* u_out, int* status_out, int* iters_out, void* z_out, void* workspace, hipStream_t stream,
                          const ipm::Cont& ct) {
    if (mpcvtol_uses_wave(p, K)) return mpcvtol_wave_launch(p, B, K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct);
    if (ct.state || ct.queue_in || ct.it_stop < p.max_iter) return hipErrorInvalidValue;   // continuation launches: the wave kernel only
    const vtol::Params P = vtol::from_c(p, K);
    const int lanes = B <= 16384 ? 16 : 64;                               // problems per block of the one-NLP-per-lane kernel (kernel = 1)
    const unsigned blocks = (unsigned)((B + lanes - 1) / lanes);
    if (p.io_dtype == SC_DTYPE_F64)
        hipLaunchKernelGGL(mpcvtol_kernel<double>, dim3(blocks), dim3(64), 0, stream, P, B, lanes, p.obs_shared, (const double*)X,
                           (const double*)u_prev, (const double*)goal, (const double*)obs, (double*)u_out, status_out, iters_out, (double*)z_out,
                           (double*)workspace);
    else
        hipLaunchKernelGGL(mpcvtol_kernel<float>, dim3(blocks), dim3(64), 0, stream, P, B, lanes, p.obs_shared, (const float*)X,
                           (const float*)u_prev, (const float*)goal, (const float*)obs, (float*)u_out, status_out, iters_out, (float*)z_out,
                           (double*)workspace);
    return hipGetLastError();
}

}  // namespace sc
