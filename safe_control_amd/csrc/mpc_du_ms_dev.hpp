// Device side of kernel 13 shared by its two translation units (mpc_du_ms.hip: circles only, the fast path; mpc_du_ms_se.hip: launches whose
// obstacle rows may be superellipsoids): the lane context over LDS, the launch-order pre-pass, the kernel and its launcher, templated on the
// storage type, the model (dums::M_*) and SE.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/safe_control_amd.h"
#include "mpc_ipm_common.hpp"
#define SC_HD __host__ __device__
#define SC_DUMS_INLINE __forceinline__
#include "mpc_du_ms_solver.hpp"

namespace sc {
namespace dums {

typedef __attribute__((address_space(3))) double ldsd;

struct DevCtx {
    typedef ldsd* ptr;
    ptr lds;
    int lane;
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ long long clock() const { return __builtin_readcyclecounter(); }
    __device__ __forceinline__ void sincos(double a, double& s, double& c) const { sc::sincos_(a, &s, &c); }
    // two powers of wave-uniform arguments at the price of one: even lanes take the first, odd lanes the second
    __device__ __forceinline__ void pow2(double x1, double e1, double x2, double e2, double& r1, double& r2) const {
        const bool odd = lane & 1;
        const double v = pow(odd ? x2 : x1, odd ? e2 : e1);
        r1 = ipm::row_value(v, 0); r2 = ipm::row_value(v, 1);
    }
    __device__ __forceinline__ double rsqrt(double v) const { return ::rsqrt(v); }
    __device__ __forceinline__ double pow(double x, double y) const { return ::pow(x, y); }
    // sum over the G (4 / 2 / 1) lanes of a group: DPP within the quad
    template <int n>
    __device__ __forceinline__ void gsum(double* v, int G) const {
        if (G >= 2) {
#pragma unroll
            for (int i = 0; i < n; ++i) v[i] += ipm::dpp_mv<0xB1>(v[i]);                   // quad_perm [1,0,3,2]
        }
        if (G == 4) {
#pragma unroll
            for (int i = 0; i < n; ++i) v[i] += ipm::dpp_mv<0x4E>(v[i]);                   // quad_perm [2,3,0,1]
        }
    }
    __device__ __forceinline__ double wsum(double v) const { return ipm::wsum(v); }
    __device__ __forceinline__ double wmax(double v) const { return ipm::wmax(v); }
    __device__ __forceinline__ double wmin(double v) const { return ipm::wmin(v); }
};

// Launch order: a launch ends with its longest solve, and the long solves of a batch are the NLPs without a feasible point (restoration
// phase: 94 iterations against a mean of 17.6 on configs[2]) -- every one of which starts with a violated CBF row at (x0, u_prev).  With a
// caller workspace (sc_mpccbf_ms_workspace_bytes) a pre-pass evaluates the K rows of stage 0 at the start point and sends the problems with a
// violated row to the front of the grid (two atomic counters; the order inside the two groups does not matter: a problem's result does not
// depend on where it ran): list scheduling on the 1024 resident slots then ends at the longest solve instead of 15 % later.
template <typename TIO, int MODEL>
__global__ void __launch_bounds__(256) mpcdu_ms_order_kernel(const Params P, long long B, int obs_shared, const TIO* __restrict__ X, const TIO* __restrict__ u_prev,
                                                             const TIO* __restrict__ obs, int* __restrict__ counters, int* __restrict__ perm) {
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const double x = (double)X[b * NX], y = (double)X[b * NX + 1], th = (double)X[b * NX + 2], v = (double)X[b * NX + 3];
    const double a = (double)u_prev[b * NU], w = (double)u_prev[b * NU + 1], dt = P.dt;
    double p1x, p1y, p2x, p2y;
    if constexpr (MODEL == M_SI) {                                            // (x, y), (vx, vy): a = vx, w = vy; one-step rows
        p1x = x + dt * a; p1y = y + dt * w; p2x = p1x; p2y = p1y;
    } else if constexpr (MODEL == M_UNI) {                                    // (x, y, theta), (v, omega): a = v; one-step rows
        p1x = x + dt * a * cos(th); p1y = y + dt * a * sin(th); p2x = p1x; p2y = p1y;
    } else if constexpr (MODEL == M_KB) {                                     // (x, y, theta, v), (a, beta): w = beta
        const double c = cos(th), s = sin(th), th1 = th + dt * v * w * P.inv_Lr, v1 = fmax(fmin(v + dt * a, P.v_max), P.v_min);
        p1x = x + dt * v * (c - w * s); p1y = y + dt * v * (s + w * c);
        p2x = p1x + dt * v1 * (cos(th1) - w * sin(th1)); p2y = p1y + dt * v1 * (sin(th1) + w * cos(th1));
    } else if constexpr (MODEL == M_DI) {                                     // (x, y, vx, vy), (ax, ay): th = vx, v = vy, a = ax, w = ay
        p1x = x + dt * th; p1y = y + dt * v;
        double wx = th + dt * a, wy = v + dt * w;
        const double vm = sqrt(wx * wx + wy * wy);
        if (vm > P.v_max) { wx *= P.v_max / vm; wy *= P.v_max / vm; }
        p2x = p1x + dt * wx; p2y = p1y + dt * wy;
    } else {
        const double th1 = th + dt * w, v1 = v + dt * a;
        p1x = x + dt * v * cos(th); p1y = y + dt * v * sin(th);
        p2x = p1x + dt * v1 * cos(th1); p2y = p1y + dt * v1 * sin(th1);
    }
    const double g1 = P.alpha1 + P.alpha2, g2 = P.alpha1 * P.alpha2;
    constexpr bool one_step = MODEL == M_UNI || MODEL == M_SI;
    const double w0 = one_step ? P.alpha1 - 1.0 : 1.0 - g1 + g2, w1 = one_step ? 1.0 : g1 - 2.0, w2 = one_step ? 0.0 : 1.0;
    const TIO* ob = obs + (obs_shared ? 0 : b * P.K * 7);
    bool viol = false;
    for (int j = 0; j < P.K; ++j) {
        const double ox = (double)ob[7 * j], oy = (double)ob[7 * j + 1], d = P.radius + (double)ob[7 * j + 2], off = P.beta * d * d;
        const double h0 = (x - ox) * (x - ox) + (y - oy) * (y - oy) - off, h1 = (p1x - ox) * (p1x - ox) + (p1y - oy) * (p1y - oy) - off;
        const double h2 = (p2x - ox) * (p2x - ox) + (p2y - oy) * (p2y - oy) - off;
        viol |= !(w0 * h0 + w1 * h1 + w2 * h2 >= 0.0);
    }
    const int pos = viol ? atomicAdd(&counters[0], 1) : (int)B - 1 - atomicAdd(&counters[1], 1);
    perm[pos] = (int)b;
}

#ifndef SC_DUMS_WAVES
#define SC_DUMS_WAVES 1
#endif
template <typename TIO, int MODEL, bool SE = false>
__global__ void __launch_bounds__(64, SC_DUMS_WAVES) mpcdu_ms_kernel(const Params P, const sc_ipopt_params O, long long B, int obs_shared, const TIO* __restrict__ X,
                                                      const TIO* __restrict__ u_prev, const TIO* __restrict__ goal, const TIO* __restrict__ obs,
                                                      TIO* __restrict__ u_out, int* __restrict__ status_out, int* __restrict__ iters_out,
                                                      TIO* __restrict__ plan_out, double* __restrict__ trace_out, const int* __restrict__ perm) {
    extern __shared__ double dums_lds[];
    if ((long long)blockIdx.x >= B) return;
    const long long b = perm ? (long long)perm[blockIdx.x] : (long long)blockIdx.x;
    DevCtx cx{(ldsd*)dums_lds, (int)threadIdx.x};
    Wave<DevCtx, MODEL, SE> S(cx, P, O);
    constexpr int U0 = MODEL == M_DI ? 1 : 0;                                 // M_DI keeps its inputs as (ay, ax): swapped on the way in and out
    const TIO* ob = obs + (obs_shared ? 0 : b * P.K * 7);
    if constexpr (SE) {
        if ((int)threadIdx.x < P.K) {                                      // one obstacle per lane: packed for hpoint()
            double o7[7], o8[8];
            for (int c = 0; c < 7; ++c) o7[c] = (double)ob[7 * threadIdx.x + c];
            pack_obstacle(o7, P.radius, P.beta, [](double a, double b) { return ::pow(a, b); }, o8);
            for (int c = 0; c < 8; ++c) dums_lds[S.L.OB + 8 * threadIdx.x + c] = o8[c];
        }
    } else if ((int)threadIdx.x < 3 * P.K) {
        const int j = threadIdx.x / 3, c = threadIdx.x % 3;
        dums_lds[S.L.OB + threadIdx.x] = j < P.K ? (double)ob[7 * j + c] : 0.0;
    }
    for (int i = 0; i < NX; ++i) S.x0[i] = ((MODEL == M_UNI && i == 3) || (MODEL == M_SI && i >= 2)) ? 0.0 : (double)X[b * NX + i];      // (Unicycle2D rows are [x, y, theta, unused], SingleIntegrator2D's [x, y, unused, unused])
    for (int j = 0; j < NU; ++j) S.uprev[j] = (double)u_prev[b * NU + (j ^ U0)];
    S.xg[0] = (double)goal[b * 2]; S.xg[1] = (double)goal[b * 2 + 1];
    __syncthreads();
    int st, it;
    S.solve(st, it, trace_out ? trace_out + (size_t)b * (size_t)(O.max_iter + 1) * TRACE_W : nullptr);
    if (threadIdx.x == 0) {
        for (int j = 0; j < NU; ++j) u_out[b * NU + (j ^ U0)] = (TIO)S.u[j];
        status_out[b] = st;
        if (iters_out) iters_out[b] = it;
    }
    if (plan_out && S.acl) {
        // the plan: x_0 .. x_N (4 each), then u_0 .. u_{N-1} (2 each)
        TIO* po = plan_out + b * (long long)((P.N + 1) * NX + P.N * NU);
        for (int i = 0; i < NX; ++i) po[S.k * NX + i] = (TIO)S.x[i];
        if (S.stg) for (int j = 0; j < NU; ++j) po[(P.N + 1) * NX + S.k * NU + (j ^ U0)] = (TIO)S.u[j];
    }
}

template <typename TIO, int MODEL, bool SE = false>
static hipError_t launch_t(const Params& P, const sc_ipopt_params& O, long long B, int obs_shared, const void* X, const void* u_prev, const void* goal,
                           const void* obs, void* u_out, int* status_out, int* iters_out, void* plan_out, double* trace_out, void* order_ws, hipStream_t stream) {
    const size_t lds = (size_t)Lds(P.N, P.K, general_layout(MODEL), SE).total * sizeof(double);
    int* perm = nullptr;
    if (order_ws && B > 1024 && !SE) {                                        // (the pre-pass evaluates circles)                                               // (up to 1024 problems are all resident at once: nothing to order)
        int* counters = (int*)order_ws;
        perm = counters + 4;
        hipError_t e0 = hipMemsetAsync(counters, 0, 4 * sizeof(int), stream);
        if (e0 != hipSuccess) return e0;
        hipLaunchKernelGGL((mpcdu_ms_order_kernel<TIO, MODEL>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, P, B, obs_shared, (const TIO*)X, (const TIO*)u_prev,
                           (const TIO*)obs, counters, perm);
    }
    hipError_t e = hipFuncSetAttribute((const void*)mpcdu_ms_kernel<TIO, MODEL, SE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mpcdu_ms_kernel<TIO, MODEL, SE>), dim3((unsigned)B), dim3(64), lds, stream, P, O, B, obs_shared, (const TIO*)X, (const TIO*)u_prev,
                       (const TIO*)goal, (const TIO*)obs, (TIO*)u_out, status_out, iters_out, (TIO*)plan_out, trace_out, (const int*)perm);
    return hipGetLastError();
}

}  // namespace dums
}  // namespace sc
