// K nearest OTHER agents of every local agent, written as moving circular obstacles
// (extension for BASELINE config 4; no reference counterpart -- examples/test_multi_robot.py:77-80
// steps its robots independently).  Brute force, N-body style: a workgroup of 256 local agents walks
// all agents in LDS tiles of 256; each lane keeps its K best (distance, index) pairs in registers
// (sorted insertion, fully unrolled); a tile is skipped by the whole wave when none of its candidates
// beats any lane's current K-th distance.
#include <hip/hip_runtime.h>

#include "sc_math.hpp"
#include "../../include/safe_control_amd.h"

namespace sc {

template <typename TIO, int KMAX>
__global__ __launch_bounds__(256) void neighbor_kernel(const long long B_all, const long long first_local,
                                                       const long long B_local, const int K, const float radius,
                                                       const TIO* __restrict__ X_all, TIO* __restrict__ obs_out) {
    using TD = TIO;                                   // distances in the storage precision (ties order as in numpy)
    __shared__ TD tx[256], ty[256];
    const int tid = threadIdx.x;
    const long long li = (long long)blockIdx.x * 256 + tid;       // local agent
    const bool active = li < B_local;
    const long long gi = first_local + (active ? li : 0);
    const TD x = (TD)X_all[gi * 4 + 0], y = (TD)X_all[gi * 4 + 1];
    const TD BIG = TD(1e30), INF = TD(__builtin_huge_valf());
    TD sd[KMAX];
    int si[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { sd[j] = INF; si[j] = -1; }
    for (long long base = 0; base < B_all; base += 256) {
        const long long src = base + tid;
        __syncthreads();
        tx[tid] = src < B_all ? (TD)X_all[src * 4 + 0] : BIG;
        ty[tid] = src < B_all ? (TD)X_all[src * 4 + 1] : BIG;
        __syncthreads();
        const int cnt = (int)((B_all - base) < 256 ? (B_all - base) : 256);
        for (int t = 0; t < cnt; ++t) {
            const TD dx = tx[t] - x, dy = ty[t] - y;
            TD cd = dx * dx + dy * dy;
            int ci = (int)(base + t);
            if (base + t == gi) cd = INF;                                   // not its own neighbour
            if (__builtin_amdgcn_ballot_w64(cd < sd[KMAX - 1]) == 0) continue;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const bool sw = cd < sd[j];
                const TD td = sd[j]; const int ti = si[j];
                sd[j] = sw ? cd : td; si[j] = sw ? ci : ti;
                cd = sw ? td : cd; ci = sw ? ti : ci;
            }
        }
    }
    if (!active) return;
    TIO* out = obs_out + (size_t)li * K * 7;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        if (j >= K) break;
        TIO row[7] = {TIO(1000), TIO(1000), TIO(0), TIO(0), TIO(0), TIO(0), TIO(0)};
        const int n = si[j];
        if (n >= 0 && sd[j] < TD(1e29)) {
            const double th = (double)X_all[(size_t)n * 4 + 2], v = (double)X_all[(size_t)n * 4 + 3];
            row[0] = X_all[(size_t)n * 4 + 0]; row[1] = X_all[(size_t)n * 4 + 1]; row[2] = TIO(radius);
            row[3] = TIO(v * cos(th)); row[4] = TIO(v * sin(th));
        }
#pragma unroll
        for (int f = 0; f < 7; ++f) out[j * 7 + f] = row[f];
    }
}

template <typename TIO>
static hipError_t nb_launch(long long B_all, long long first, long long B_local, int K, double r, const void* X, void* out,
                            hipStream_t stream) {
    const unsigned blocks = (unsigned)((B_local + 255) / 256);
#define SC_NB(KM) hipLaunchKernelGGL((neighbor_kernel<TIO, KM>), dim3(blocks), dim3(256), 0, stream, B_all, first, B_local, K, (float)r, (const TIO*)X, (TIO*)out)
    if (K <= 8) SC_NB(8);
    else if (K <= 16) SC_NB(16);
    else SC_NB(32);
#undef SC_NB
    return hipGetLastError();
}

hipError_t neighbors_launch(int io_dtype, long long B_all, long long first, long long B_local, int K, double r, const void* X,
                            void* out, hipStream_t stream) {
    if (io_dtype == SC_DTYPE_F32) return nb_launch<float>(B_all, first, B_local, K, r, X, out, stream);
    return nb_launch<double>(B_all, first, B_local, K, r, X, out, stream);
}

}  // namespace sc
