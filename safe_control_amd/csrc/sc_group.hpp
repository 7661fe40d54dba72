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
#ifndef SC_EXP_MAXIT
#define SC_EXP_MAXIT G                                     // developer builds cap it to time one pass of the loop
#endif
    for (int it = 0; it < SC_EXP_MAXIT; ++it) {               // at most K commits per agent
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

// ---- all candidates at once (G = 8) ---------------------------------------------------------------------------------
// The walk above commits one violated row per pass: its latency is (number of commits of the slowest agent of the WAVE + 1)
// passes of ~1000 cycles each for a lone wave (measured on the headline launch: 0.5 us for the first pass, 1.9 us for the five
// passes its slowest wave needs -- the launch ends with that wave).  This form has a fixed, shorter critical path: the
// minimiser of ||u - u_ref||^2 over {box, rows} is u_box = clamp(u_ref) when no row is violated there; otherwise it lies on the
// line of a row that IS violated at u_box (moving from the optimum towards u_box lowers the objective, so it must leave the
// feasible set through an active row, which u_box then violates), and on that line it is the point of the feasible interval
// closest to the foot of u_ref.  So every lane whose row is violated at u_box clips ITS line against the box and against the
// seven other rows of the group (seven xor-partners: DPP quad permutes and the half-row mirror, no LDS round trip), takes the
// closest point of the interval, and the group keeps the candidate with the smallest distance.  Same clip_box / clip_row /
// midpoint rule as the walk, so the chosen point is computed by the same arithmetic as the walk's last commit on that row;
// feasibility is again decided once, in slack space, by the caller.
template <typename T, int X>
__device__ __forceinline__ T group_xor8(T v, T mirrored) {          // value of lane (sub ^ X), X = 1..7; mirrored = lane (sub ^ 7)
    if constexpr (X == 1) return dpp_mov<0xB1>(v);
    else if constexpr (X == 2) return dpp_mov<0x4E>(v);
    else if constexpr (X == 3) return dpp_mov<0x1B>(v);
    else if constexpr (X == 4) return dpp_mov<0x1B>(mirrored);
    else if constexpr (X == 5) return dpp_mov<0x4E>(mirrored);
    else if constexpr (X == 6) return dpp_mov<0xB1>(mirrored);
    else return mirrored;
}
template <typename TC, int... Xs>
__device__ __forceinline__ void clip_partners8(LineQP<TC>& L, TC a0, TC a1, TC cc, std::integer_sequence<int, Xs...>) {
    const TC m0 = dpp_mov<0x141>(a0), m1 = dpp_mov<0x141>(a1), mc = dpp_mov<0x141>(cc);
    (clip_row(L, group_xor8<TC, Xs + 1>(a0, m0), group_xor8<TC, Xs + 1>(a1, m1), group_xor8<TC, Xs + 1>(cc, mc)), ...);
}
template <typename TC>
__device__ __forceinline__ void coop_solve_all8(QpState<TC>& S, int K, int sub, int lane, TC a0, TC a1, TC cc,
                                                const CbfConsts<TC>& k) {
    const TC inf = num<TC>::inf();
    const bool testable = (sub < K) && !((a0 == TC(0)) && (a1 == TC(0)));     // all-zero rows are never projected on
    LineQP<TC> L;
    const bool viol = qp_row_violated(S, a0, a1, cc, L, k) && testable;       // S holds u_box; L: the line of this lane's row
    if (__builtin_amdgcn_ballot_w64(viol) == 0ull) return;                   // wave-uniform: nothing violated anywhere
    clip_box(L, k);
    clip_partners8(L, a0, a1, cc, std::make_integer_sequence<int, 7>{});
    TC t = fmin_(fmax_(TC(0), L.lo), L.hi);
    const bool inverted = L.lo > L.hi;
    t = inverted ? TC(0.5) * (L.lo + L.hi) : t;                               // by a hair (rounding): split the difference, as the walk
    const bool empty = L.lo > L.hi + num<TC>::tol_feas() * fmax_(TC(1), fmax_(fabs_(L.lo), fabs_(L.hi)));
    const TC v0 = L.p0 + t * L.d0, v1 = L.p1 + t * L.d1;
    const TC e0 = v0 - S.ur0, e1 = v1 - S.ur1;
    TC cost = e0 * e0 + e1 * e1;
    cost = (viol && !empty && (cost == cost)) ? cost : inf;
    const TC best = group_min<TC, 8>(cost);
    const int gbase = lane & ~7;
    const unsigned long long mg = __builtin_amdgcn_ballot_w64((cost == best) && (best < inf)) & (0xffull << gbase);
    const bool has = mg != 0ull;
    const int win = has ? (__builtin_ctzll(mg) - gbase) : -1;
    const TC s0 = group_max<TC, 8>(sub == win ? v0 : -inf), s1 = group_max<TC, 8>(sub == win ? v1 : -inf);
    S.u0 = has ? s0 : S.u0;                                                   // no candidate: u_box stays and fails the slack check
    S.u1 = has ? s1 : S.u1;
}

// The same for 16 lanes per agent (K <= 16; one DPP row per agent): fifteen partners by row_ror, reductions by quad permutes and
// the two mirrors.
template <typename T>
__device__ __forceinline__ T row_min16(T v) {
    v = fmin_(v, dpp_mov<0xB1>(v)); v = fmin_(v, dpp_mov<0x4E>(v)); v = fmin_(v, dpp_mov<0x141>(v)); v = fmin_(v, dpp_mov<0x140>(v));
    return v;
}
template <typename T>
__device__ __forceinline__ T row_max16(T v) {
    v = fmax_(v, dpp_mov<0xB1>(v)); v = fmax_(v, dpp_mov<0x4E>(v)); v = fmax_(v, dpp_mov<0x141>(v)); v = fmax_(v, dpp_mov<0x140>(v));
    return v;
}
template <typename TC, int... Rs>
__device__ __forceinline__ void clip_partners16(LineQP<TC>& L, TC a0, TC a1, TC cc, std::integer_sequence<int, Rs...>) {
    (clip_row(L, dpp_mov<0x121 + Rs>(a0), dpp_mov<0x121 + Rs>(a1), dpp_mov<0x121 + Rs>(cc)), ...);     // row_ror:1 .. row_ror:15
}
template <typename TC>
__device__ __forceinline__ void coop_solve_all16(QpState<TC>& S, int K, int sub, int lane, TC a0, TC a1, TC cc,
                                                 const CbfConsts<TC>& k) {
    const TC inf = num<TC>::inf();
    const bool testable = (sub < K) && !((a0 == TC(0)) && (a1 == TC(0)));
    LineQP<TC> L;
    const bool viol = qp_row_violated(S, a0, a1, cc, L, k) && testable;
    if (__builtin_amdgcn_ballot_w64(viol) == 0ull) return;
    clip_box(L, k);
    clip_partners16(L, a0, a1, cc, std::make_integer_sequence<int, 15>{});
    TC t = fmin_(fmax_(TC(0), L.lo), L.hi);
    t = (L.lo > L.hi) ? TC(0.5) * (L.lo + L.hi) : t;
    const bool empty = L.lo > L.hi + num<TC>::tol_feas() * fmax_(TC(1), fmax_(fabs_(L.lo), fabs_(L.hi)));
    const TC v0 = L.p0 + t * L.d0, v1 = L.p1 + t * L.d1;
    const TC e0 = v0 - S.ur0, e1 = v1 - S.ur1;
    TC cost = e0 * e0 + e1 * e1;
    cost = (viol && !empty && (cost == cost)) ? cost : inf;
    const TC best = row_min16(cost);
    const int gbase = lane & ~15;
    const unsigned long long mg = __builtin_amdgcn_ballot_w64((cost == best) && (best < inf)) & (0xffffull << gbase);
    const bool has = mg != 0ull;
    const int win = has ? (__builtin_ctzll(mg) - gbase) : -1;
    const TC s0 = row_max16(sub == win ? v0 : -inf), s1 = row_max16(sub == win ? v1 : -inf);
    S.u0 = has ? s0 : S.u0;
    S.u1 = has ? s1 : S.u1;
}

}  // namespace sc
