// CBF-QP kernels, storage float / arithmetic float (one translation unit per dtype pair so they build in parallel).
#include "cbf_qp_kernel.hpp"

namespace sc {
hipError_t cbfqp_launch_f32(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                        const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                        hipStream_t stream) {
    return launch_model<float, float>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
}
}  // namespace sc
