// MPC-CBF for VTOL2D, condensed form: dispatch to the wave-per-problem kernel (mpc_vtol_wave.hip, kernel 11 in DESIGN.md).
// The lane-per-problem kernel that used to live here (sc_mpcvtol_params.kernel = 1: one NLP per lane around mpc_vtol_solver.hpp's Solver,
// work arrays in a caller workspace) was the reference the wave kernel was developed against; nothing but its own cross-check reached it
// and it was retired in round 6 -- the same Solver still builds for the host (tools/vtol_host.cpp, tests/test_vtol_solver_host.py).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/safe_control_amd.h"
#define SC_VTOL_WITH_C_PARAMS
#include "mpc_vtol_solver.hpp"
#include "mpc_cont.hpp"

namespace sc {

hipError_t mpcvtol_wave_launch(const sc_mpcvtol_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                               const void* obs, void* u_out, int* status_out, int* iters_out, void* z_out, hipStream_t stream,
                               const ipm::Cont& ct);

// the wave-per-problem kernel (mpc_vtol_wave.hip: one stage per lane, the stage's rows in registers, instantiated for 8 and 16 row slots)
// serves every (p, K) the C-ABI accepts; p.kernel = 1 (the retired lane-per-problem kernel) is refused there
bool mpcvtol_uses_wave(const sc_mpcvtol_params& p, int K) {
    if (p.kernel == 1) return false;
    return K <= 16 && p.horizon <= 64;
}

size_t mpcvtol_workspace_bytes(const sc_mpcvtol_params& p, long long B, int K) {
    (void)p; (void)B; (void)K;
    return 0;                                                               // (the wave kernel keeps everything in registers and LDS)
}

hipError_t mpcvtol_launch(const sc_mpcvtol_params& p, long long B, int K, const void* X, const void* u_prev, const void* goal,
                          const void* obs, void* u_out, int* status_out, int* iters_out, void* z_out, void* workspace, hipStream_t stream,
                          const ipm::Cont& ct) {
    if (mpcvtol_uses_wave(p, K)) return mpcvtol_wave_launch(p, B, K, X, u_prev, goal, obs, u_out, status_out, iters_out, z_out, stream, ct);
    if (ct.state || ct.queue_in || ct.it_stop < p.max_iter) return hipErrorInvalidValue;   // continuation launches: the wave kernel only
    (void)workspace;
    return hipErrorInvalidValue;                                           // (kernel = 1, the one-NLP-per-lane kernel: retired in round 6, refused by the C-ABI)
}

}  // namespace sc
