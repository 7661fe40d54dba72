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
            bool moved = false;                    // stable: once the candidate is placed, everything after it shifts
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const bool sw = moved || (cd < sd[j]);
                moved = sw;
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

// ---- candidate-split variant ------------------------------------------------------------------------------------------
// The kernel above gives every 64 local agents ONE wave that scans all B_all candidates: its time is that wave's scan
// whatever the shard size, so sharding the fleet over more GPUs does not shorten a step.  Here the candidate range is
// cut into W slices: wave (group, w) keeps the K best of its slice for its 64 agents and writes them to a workspace;
// a second kernel merges the W sorted lists of an agent (lower slices first, strict <, so ties order by index exactly
// like the single scan) and writes the obstacle rows.  W is chosen so that about 2048 waves are in flight.
template <typename TIO, int KMAX>
__global__ __launch_bounds__(64) void neighbor_partial_kernel(const long long B_all, const long long first_local,
                                                              const long long B_local, const int W, const long long slice,
                                                              const TIO* __restrict__ X_all, TIO* __restrict__ pd,
                                                              int* __restrict__ pi) {
    using TD = TIO;
    __shared__ TD tx[64], ty[64];
    const int tid = threadIdx.x, w = blockIdx.y;
    const long long li = (long long)blockIdx.x * 64 + tid;
    const bool active = li < B_local;
    const long long gi = first_local + (active ? li : 0);
    const TD x = (TD)X_all[gi * 4 + 0], y = (TD)X_all[gi * 4 + 1];
    const TD BIG = TD(1e30), INF = TD(__builtin_huge_valf());
    TD sd[KMAX];
    int si[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { sd[j] = INF; si[j] = -1; }
    const long long c0 = (long long)w * slice, c1 = (c0 + slice) < B_all ? (c0 + slice) : B_all;
    for (long long base = c0; base < c1; base += 64) {
        const long long src = base + tid;
        __syncthreads();
        tx[tid] = src < c1 ? (TD)X_all[src * 4 + 0] : BIG;
        ty[tid] = src < c1 ? (TD)X_all[src * 4 + 1] : BIG;
        __syncthreads();
        const int cnt = (int)((c1 - base) < 64 ? (c1 - base) : 64);
        for (int t = 0; t < cnt; ++t) {
            const TD dx = tx[t] - x, dy = ty[t] - y;
            TD cd = dx * dx + dy * dy;
            int ci = (int)(base + t);
            if (base + t == gi) cd = INF;
            if (__builtin_amdgcn_ballot_w64(cd < sd[KMAX - 1]) == 0) continue;
            bool moved = false;                    // stable: once the candidate is placed, everything after it shifts
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const bool sw = moved || (cd < sd[j]);
                moved = sw;
                const TD td = sd[j]; const int ti = si[j];
                sd[j] = sw ? cd : td; si[j] = sw ? ci : ti;
                cd = sw ? td : cd; ci = sw ? ti : ci;
            }
        }
    }
    if (!active) return;
    // [slice][entry][agent]: the merge reads lane-contiguous
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        pd[((size_t)w * KMAX + j) * B_local + li] = sd[j];
        pi[((size_t)w * KMAX + j) * B_local + li] = si[j];
    }
}

template <typename TIO, int KMAX>
__global__ __launch_bounds__(64) void neighbor_merge_kernel(const long long B_local, const int W, const int K, const float radius,
                                                            const TIO* __restrict__ X_all, const TIO* __restrict__ pd,
                                                            const int* __restrict__ pi, TIO* __restrict__ obs_out) {
    using TD = TIO;
    const long long li = (long long)blockIdx.x * 64 + threadIdx.x;
    const bool active = li < B_local;
    const long long l = active ? li : 0;
    const TD INF = TD(__builtin_huge_valf());
    TD sd[KMAX];
    int si[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { sd[j] = INF; si[j] = -1; }
    for (int w = 0; w < W; ++w) {
        for (int e = 0; e < KMAX; ++e) {
            TD cd = pd[((size_t)w * KMAX + e) * B_local + l];
            int ci = pi[((size_t)w * KMAX + e) * B_local + l];
            if (__builtin_amdgcn_ballot_w64(cd < sd[KMAX - 1]) == 0) break;     // the slice's list is sorted: the rest is no better
            bool moved = false;                    // stable: once the candidate is placed, everything after it shifts
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {
                const bool sw = moved || (cd < sd[j]);
                moved = sw;
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

static int nb_kmax(int K) { return K <= 8 ? 8 : (K <= 16 ? 16 : 32); }

// number of candidate slices for a shard of B_local agents: about 2048 waves in flight, slices of at least 256 candidates
static int nb_slices(long long B_all, long long B_local) {
    const long long groups = (B_local + 63) / 64;
    long long W = (2048 + groups - 1) / groups;
    const long long wmax = B_all / 256 > 1 ? B_all / 256 : 1;
    W = W < 1 ? 1 : (W > 64 ? 64 : W);
    return (int)(W > wmax ? wmax : W);
}

size_t neighbors_workspace_bytes(int io_dtype, long long B_all, long long B_local, int K) {
    const size_t es = io_dtype == SC_DTYPE_F32 ? 4 : 8;
    return (size_t)nb_slices(B_all, B_local) * nb_kmax(K) * (size_t)B_local * (es + 4);
}

template <typename TIO>
static hipError_t nb_split_launch(long long B_all, long long first, long long B_local, int K, double r, const void* X,
                                  void* out, void* ws, hipStream_t stream) {
    const int W = nb_slices(B_all, B_local), KM = nb_kmax(K);
    const long long slice = ((B_all + W - 1) / W + 63) / 64 * 64;
    const unsigned groups = (unsigned)((B_local + 63) / 64);
    TIO* pd = (TIO*)ws;
    int* pi = (int*)((unsigned char*)ws + (size_t)W * KM * B_local * sizeof(TIO));
#define SC_NBS(KMX)                                                                                                         \
    hipLaunchKernelGGL((neighbor_partial_kernel<TIO, KMX>), dim3(groups, (unsigned)W), dim3(64), 0, stream, B_all, first,   \
                       B_local, W, slice, (const TIO*)X, pd, pi);                                                           \
    hipLaunchKernelGGL((neighbor_merge_kernel<TIO, KMX>), dim3(groups), dim3(64), 0, stream, B_local, W, K, (float)r,       \
                       (const TIO*)X, pd, pi, (TIO*)out)
    if (KM == 8) { SC_NBS(8); }
    else if (KM == 16) { SC_NBS(16); }
    else { SC_NBS(32); }
#undef SC_NBS
    return hipGetLastError();
}

hipError_t neighbors_split_launch(int io_dtype, long long B_all, long long first, long long B_local, int K, double r,
                                  const void* X, void* out, void* ws, hipStream_t stream) {
    if (io_dtype == SC_DTYPE_F32) return nb_split_launch<float>(B_all, first, B_local, K, r, X, out, ws, stream);
    return nb_split_launch<double>(B_all, first, B_local, K, r, X, out, ws, stream);
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
