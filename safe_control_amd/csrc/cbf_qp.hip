// Dtype dispatch for the fused CBF-QP kernels (cbf_qp_kernel.hpp).
#include <hip/hip_runtime.h>

#include "../../include/safe_control_amd.h"

namespace sc {
#define SC_DECL(name)                                                                                         \
    hipError_t cbfqp_launch_##name(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref, \
                                   const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,  \
                                   hipStream_t stream);
SC_DECL(f32)
SC_DECL(f32c64)
SC_DECL(f64)
#undef SC_DECL

hipError_t cbfqp_launch(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                        const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                        hipStream_t stream) {
    if (p.io_dtype == SC_DTYPE_F32) {
        if (p.compute_dtype == SC_DTYPE_F32)
            return cbfqp_launch_f32(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        return cbfqp_launch_f32c64(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
    }
    return cbfqp_launch_f64(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
}
}  // namespace sc
