// Cross-lane helpers for kernels that spread one agent over a group of G lanes (one obstacle row per lane):
// broadcast / min / max inside the group and the cooperative incremental QP walk.  Used by the cooperative CBF-QP
// kernel (cbf_qp_kernel.hpp) and the cooperative closed-loop rollout (tracking.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "sc_qp2.hpp"

namespace sc {

// Cross-lane moves inside a group.  For G = 8 (the headline launch) they are DPP moves -- quad permutes and the
// half-row mirror, VALU latency -- instead of ds_bpermute round trips (~100 cycles each for a wave that has
// nothing else to do): broadcast of lane I = quad broadcast + mirror into the other quad; min / max = xor 1, xor 2,
// mirror.  Wider groups keep the shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <typename T, int G, int I>
__device__ __forceinline__ T group_bcast(T v, int sub) {
    if constexpr (G == 8) {
        const T q = dpp_mov<(I & 3) * 0x55>(v);             // quad_perm [I&3, I&3, I&3, I&3]
        const T m = dpp_mov<0x141>(q);                      // row_half_mirror: the other quad's value
        return ((sub >> 2) == (I >> 2)) ? q : m;
    } else {
        return __shfl(v, I, G);
    }
}
template <typename T, int G>
__device__ __forceinline__ T group_max(T v) {
    if constexpr (G == 8) {
        v = fmax_(v, dpp_mov<0xB1>(v)); v = fmax_(v, dpp_mov<0x4E>(v)); v = fmax_(v, dpp_mov<0x141>(v));
    } else {
#pragma unroll
        for (int o = 1; o < G; o <<= 1) v = fmax_(v, __shfl_xor(v, o));
    }
    return v;
}
template <typename T, int G>
__device__ __forceinline__ T group_min(T v) {
    if constexpr (G == 8) {
        v = fmin_(v, dpp_mov<0xB1>(v)); v = fmin_(v, dpp_mov<0x4E>(v)); v = fmin_(v, dpp_mov<0x141>(v));
    } else {
#pragma unroll
        for (int o = 1; o < G; o <<= 1) v = fmin_(v, __shfl_xor(v, o));
    }
    return v;
}

// step I of the cooperative walk (I is a template parameter so that the broadcast lane is a DPP immediate)
template <typename TC, int G, int I>
__device__ __forceinline__ void coop_step(QpState<TC>& S, int K, int sub, TC a0, TC a1, TC cc, const CbfConsts<TC>& k) {
    if (I >= K) return;
    const TC bi0 = group_bcast<TC, G, I>(a0, sub), bi1 = group_bcast<TC, G, I>(a1, sub), bic = group_bcast<TC, G, I>(cc, sub);
    LineQP<TC> L;
    const bool viol = qp_row_violated(S, bi0, bi1, bic, L, k);
    if (__builtin_amdgcn_ballot_w64(viol) == 0) return;
    clip_box(L, k);
    if (sub < I) clip_row(L, a0, a1, cc);                 // rows j < i, one per lane, in parallel
    L.lo = group_max<TC, G>(L.lo);
    L.hi = group_min<TC, G>(L.hi);
    qp_row_commit(S, L, viol);
}
template <typename TC, int G, int... Is>
__device__ __forceinline__ void coop_walk(QpState<TC>& S, int K, int sub, TC a0, TC a1, TC cc, const CbfConsts<TC>& k,
                                          std::integer_sequence<int, Is...>) {
    (coop_step<TC, G, Is>(S, K, sub, a0, a1, cc, k), ...);
}

// The same walk, driven by the violated rows instead of by the row index.  Every lane tests its OWN row at the running
// optimum; the first violated row of a group (lowest index above the last committed one) is broadcast and committed;
// repeat until no group of the wave has a violated row left.  Rows that are satisfied when their turn comes change
// nothing in the index-driven walk either, so both visit the same rows in the same order with the same arithmetic --
// but a wave whose agents have no violated row (the common case) leaves after one test instead of K.
template <typename TC, int G>
__device__ __forceinline__ void coop_walk_violated(QpState<TC>& S, int K, int sub, int lane, TC a0, TC a1, TC cc,
                                                   const CbfConsts<TC>& k) {
    const int gbase = lane & ~(G - 1);
    const unsigned long long grp = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << gbase;
    const bool testable = (sub < K) && !((a0 == TC(0)) && (a1 == TC(0)));     // all-zero rows are never projected on
    int last = -1;
    for (int it = 0; it < G; ++it) {                          // at most K commits per agent
        const TC s = a0 * S.u0 + (a1 * S.u1 + cc);
        const bool v = testable && (sub > last) && (s < TC(0));
        const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
        if (m == 0ull) break;                                 // wave-uniform
        const unsigned long long mg = m & grp;
        const bool has = mg != 0ull;
        const int istar = has ? (__builtin_ctzll(mg) - gbase) : 0;
        const TC bi0 = __shfl(a0, istar, G), bi1 = __shfl(a1, istar, G), bic = __shfl(cc, istar, G);
        LineQP<TC> L;
        const bool viol = qp_row_violated(S, bi0, bi1, bic, L, k) && has;
        clip_box(L, k);
        if (sub < istar) clip_row(L, a0, a1, cc);             // rows j < i*, one per lane, in parallel
        L.lo = group_max<TC, G>(L.lo);
        L.hi = group_min<TC, G>(L.hi);
        qp_row_commit(S, L, viol);
        last = has ? istar : K;
    }
}

}  // namespace sc
