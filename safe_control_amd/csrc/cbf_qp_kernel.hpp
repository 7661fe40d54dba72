// Fused batched CBF-QP for gfx950: one launch = row assembly + exact QP solve
// + status + h(x) for B agents.  Replaces the per-robot Python path
//   CBFQP.solve_control_problem   position_control/cbf_qp.py:108-199
//   robot.agent_barrier / f / g    robots/robot.py:389-436
//   cvxpy -> GUROBI                position_control/cbf_qp.py:190
//
// Mapping: one QP per lane, one 64-agent wave per workgroup.  The wave's
// obstacle rows (64 x K x 7 contiguous values in the reference's [B,K,7]
// layout) are streamed HBM -> LDS with 16-byte LDS-DMA (global_load_lds, 1 KiB
// per wave instruction, no VGPR round trip) and read back per lane; X / u_ref
// / outputs are plain coalesced vector accesses.  All rows live in registers;
// the solve is sc_qp2.hpp.  HBM traffic is exactly the algorithmic bytes
// (every input read once, every output written once).
#pragma once
#include <hip/hip_runtime.h>

#include "sc_qp2.hpp"

namespace sc {

using as1_void = const __attribute__((address_space(1))) void;
using as3_void = __attribute__((address_space(3))) void;

template <typename TIO> struct vec2;
template <> struct vec2<float> { using type = float2; };
template <> struct vec2<double> { using type = double2; };

// Stage `bytes` contiguous bytes (multiple of 16, 16-byte aligned) from global
// to LDS with the wave's 64 lanes, 1 KiB per instruction.
__device__ __forceinline__ void wave_dma16(const unsigned char* __restrict__ src, unsigned char* lds,
                                           unsigned bytes, int lane) {
    const unsigned n16 = bytes >> 4;
    for (unsigned base = 0; base < n16; base += 64) {
        const unsigned idx = base + lane;
        if (idx < n16) {
            __builtin_amdgcn_global_load_lds((as1_void*)(src + (size_t)idx * 16),
                                             (as3_void*)(lds + (size_t)base * 16), 16, 0, 0);
        }
    }
}

// KMAX > 0: rows in registers, loops unrolled to KMAX (K <= KMAX <= 8).
// KMAX == 0: rows in LDS, run-time loops (any K up to SC_CBFQP_MAX_OBS).
template <typename TIO, typename TC, int KMAX, int MODEL>
__global__ __launch_bounds__(64) void cbfqp_kernel(const sc_cbfqp_params p, const long long B, const int K,
                                                   const TIO* __restrict__ X, const TIO* __restrict__ u_ref,
                                                   const TIO* __restrict__ obs, const int* __restrict__ n_obs,
                                                   TIO* __restrict__ u_out, int* __restrict__ status_out,
                                                   TIO* __restrict__ h_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TIO* lobs = reinterpret_cast<TIO*>(smem);

    const int lane = threadIdx.x;
    const long long first = (long long)blockIdx.x * 64;
    const long long agent = first + lane;
    const int nag = (int)((B - first) < 64 ? (B - first) : 64);
    const bool active = lane < nag;
    const int row_elems = K * 7;

    // ---- obstacles: HBM -> LDS ------------------------------------------------
    if (p.obs_shared) {
        for (int e = lane; e < row_elems; e += 64) lobs[e] = obs[e];
    } else {
        const TIO* src = obs + (size_t)first * row_elems;
        const unsigned bytes = (unsigned)nag * row_elems * sizeof(TIO);
        if ((bytes & 15u) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) {
            wave_dma16(reinterpret_cast<const unsigned char*>(src), smem, bytes, lane);
        } else {
            const int total = nag * row_elems;
            for (int e = lane; e < total; e += 64) lobs[e] = src[e];
        }
    }

    // ---- state / reference (coalesced, overlaps the DMA) -----------------------
    using V2 = typename vec2<TIO>::type;
    TC x = 0, y = 0, th = 0, v = 0, ur0 = 0, ur1 = 0;
    int nk = K;
    if (active) {
        const V2* Xv = reinterpret_cast<const V2*>(X) + agent * 2;
        const V2 xa = Xv[0], xb = Xv[1];
        const V2 ur = reinterpret_cast<const V2*>(u_ref)[agent];
        x = TC(xa.x); y = TC(xa.y); th = TC(xb.x); v = TC(xb.y);
        ur0 = TC(ur.x); ur1 = TC(ur.y);
        if (n_obs) {
            nk = n_obs[agent];
            nk = nk < 0 ? 0 : (nk > K ? K : nk);
        }
    }
    const CbfConsts<TC> k = make_consts<TC>(p);
    const Agent<TC> ag = make_agent<TC>(x, y, th, v);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // LDS-DMA landed (it is tracked by vmcnt)
    __syncthreads();

    const TIO* mine = lobs + (p.obs_shared ? 0 : lane * row_elems);
    bool bad_obs = false;
    TC u0, u1;
    int st;

    if constexpr (KMAX > 0) {
        // ---- row assembly (agent_barrier + cbf_qp.py:155-183), rows in registers ----
        TC n0[KMAX], n1[KMAX], c[KMAX];
        TIO hv[KMAX];
        if (K == KMAX) {
            // compile-time stride: the compiler turns these into wide ds_reads
            TIO flat[KMAX * 7];
#pragma unroll
            for (int e = 0; e < KMAX * 7; ++e) flat[e] = mine[e];
#pragma unroll
            for (int r = 0; r < KMAX; ++r) {
                TC o[7];
#pragma unroll
                for (int f = 0; f < 7; ++f) o[f] = TC(flat[r * 7 + f]);
                TC h;
                const bool ok = cbf_row<TC, MODEL>(ag, o, k, n0[r], n1[r], c[r], h);
                const bool used = r < nk;
                bad_obs |= used && !ok;
                n0[r] = used ? n0[r] : TC(0);
                n1[r] = used ? n1[r] : TC(0);
                c[r] = used ? c[r] : TC(0);
                hv[r] = used ? TIO(h) : TIO(0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < KMAX; ++r) {
                n0[r] = n1[r] = c[r] = TC(0);
                hv[r] = TIO(0);
                if (r < K) {
                    TC o[7];
#pragma unroll
                    for (int f = 0; f < 7; ++f) o[f] = TC(mine[r * 7 + f]);
                    TC h, a0, a1, cc;
                    const bool ok = cbf_row<TC, MODEL>(ag, o, k, a0, a1, cc, h);
                    const bool used = r < nk;
                    bad_obs |= used && !ok;
                    n0[r] = used ? a0 : TC(0);
                    n1[r] = used ? a1 : TC(0);
                    c[r] = used ? cc : TC(0);
                    hv[r] = used ? TIO(h) : TIO(0);
                }
            }
        }
        st = qp2_solve<TC, KMAX>(n0, n1, c, K, ur0, ur1, k, u0, u1);
        if (active && h_out) {
            TIO* hp = h_out + agent * K;
            if (K == KMAX) {
#pragma unroll
                for (int r = 0; r < KMAX; ++r) hp[r] = hv[r];
            } else {
#pragma unroll
                for (int r = 0; r < KMAX; ++r) if (r < K) hp[r] = hv[r];
            }
        }
    } else {
        // ---- rows in LDS: [K][3][64] of TC behind the obstacle stage -----------------
        const size_t stage_bytes = ((p.obs_shared ? (size_t)row_elems : (size_t)64 * row_elems) * sizeof(TIO) + 15) & ~(size_t)15;
        TC* rows = reinterpret_cast<TC*>(smem + stage_bytes);
        TIO* hp = h_out ? h_out + agent * K : nullptr;
#pragma nounroll
        for (int r = 0; r < K; ++r) {
            TC o[7];
#pragma unroll
            for (int f = 0; f < 7; ++f) o[f] = TC(mine[r * 7 + f]);
            TC h, a0, a1, cc;
            const bool ok = cbf_row<TC, MODEL>(ag, o, k, a0, a1, cc, h);
            const bool used = r < nk;
            bad_obs |= used && !ok;
            rows[(r * 3 + 0) * 64 + lane] = used ? a0 : TC(0);
            rows[(r * 3 + 1) * 64 + lane] = used ? a1 : TC(0);
            rows[(r * 3 + 2) * 64 + lane] = used ? cc : TC(0);
            if (active && hp) hp[r] = used ? TIO(h) : TIO(0);
        }
        st = qp2_solve_lds<TC>(rows, lane, K, ur0, ur1, k, u0, u1);
    }

    if (bad_obs) st = SC_STATUS_BAD_OBSTACLE;
    if (st != SC_STATUS_OPTIMAL) { u0 = num<TC>::nan(); u1 = num<TC>::nan(); }
    if (active) {
        V2 uo; uo.x = TIO(u0); uo.y = TIO(u1);
        reinterpret_cast<V2*>(u_out)[agent] = uo;
        status_out[agent] = st;
    }
}

// ---- host-side dispatch ----------------------------------------------------------
template <typename TIO, typename TC, int KMAX, int MODEL>
static hipError_t launch_one(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                             const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                             hipStream_t stream) {
    const unsigned blocks = (unsigned)((B + 63) / 64);
    size_t lds = (p.obs_shared ? (size_t)K * 7 : (size_t)64 * K * 7) * sizeof(TIO);
    if (KMAX == 0) lds = ((lds + 15) & ~(size_t)15) + (size_t)K * 3 * 64 * sizeof(TC);
    auto kern = cbfqp_kernel<TIO, TC, KMAX, MODEL>;
    if (lds > 64 * 1024) {
        static bool raised = false;        // dynamic LDS above 64 KiB needs the attribute once
        if (!raised) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            raised = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, stream, p, B, K,
                       (const TIO*)X, (const TIO*)u_ref, (const TIO*)obs, n_obs, (TIO*)u_out, status, (TIO*)h_out);
    return hipGetLastError();
}

template <typename TIO, typename TC, int MODEL>
static hipError_t launch_k(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                           const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                           hipStream_t stream) {
#define SC_K(KM) return launch_one<TIO, TC, KM, MODEL>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream)
    if (K <= 4) SC_K(4);
    if (K <= 8) SC_K(8);
    SC_K(0);
#undef SC_K
}

template <typename TIO, typename TC>
static hipError_t launch_model(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                               const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                               hipStream_t stream) {
    switch (p.model_id) {
        case SC_MODEL_DYNAMIC_UNICYCLE2D:
            return launch_k<TIO, TC, SC_MODEL_DYNAMIC_UNICYCLE2D>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        case SC_MODEL_KINEMATIC_BICYCLE2D:
            return launch_k<TIO, TC, SC_MODEL_KINEMATIC_BICYCLE2D>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        case SC_MODEL_KINEMATIC_BICYCLE2D_C3BF:
            return launch_k<TIO, TC, SC_MODEL_KINEMATIC_BICYCLE2D_C3BF>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        default:
            return launch_k<TIO, TC, SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
    }
}

}  // namespace sc
