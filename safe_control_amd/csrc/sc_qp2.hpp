// Exact 2-variable projection QP, one problem per lane.
//
//   minimise ||u - u_ref||^2   s.t.  n0[i] u0 + n1[i] u1 + c[i] >= 0  (i < K),  lo <= u <= hi
//
// This is what cvxpy -> GUROBI solves at position_control/cbf_qp.py:190.  The
// objective is strictly convex, so the minimiser is unique.  The kernel walks
// the constraints incrementally (Seidel-style): the running optimum u of
// {box, rows 0..i-1} is kept; if row i is violated at u, the optimum of
// {box, rows 0..i} lies on the line of row i, where the problem is the
// projection of u_ref onto that line clipped to the interval [lo, hi] the box
// and the earlier rows leave on it.
//
// Rows arrive NORMALISED (unit normal, see normalise_row): then for line i with
// unit direction d = (-n1, n0) and foot point p (projection of u_ref), an
// earlier row j restricts the line parameter t through
//       a t + r >= 0,   a = n_j . d  (sine of the angle between the rows),  r = n_j . p + c_j
// i.e. t >= -r/a for a >= 0 and t <= -r/a for a < 0.  IEEE arithmetic covers
// the degenerate cases without branches: a == 0 (parallel or all-zero row)
// gives t = -r * inf = -inf (no restriction) when the line satisfies row j and
// +inf (empty interval => infeasible) when it does not.
// If rounding leaves lo > hi by a hair the midpoint is taken and the walk goes
// on.  Feasibility is decided ONCE, at the end, in slack space: the returned
// point must satisfy every (normalised) row to tol_feas -- what a QP solver's
// feasibility tolerance means -- so a problem is reported infeasible exactly
// when even the best point found violates a row by more than the tolerance.
// Everything is predicated; the only branch is a wave-uniform skip when no
// lane violates row i.
#pragma once
#include "sc_models.hpp"

namespace sc {

__device__ __forceinline__ float rcp_(float a) { return __builtin_amdgcn_rcpf(a); }     // v_rcp_f32, 1 ulp
// f64: hardware seed (v_rcp_f64 / v_rsq_f64, ~2^-27) + two Newton steps = full double precision to an ulp
// or two, in ~6 instructions instead of the ~30 of an IEEE division.  A zero / infinite argument yields
// NaN through the Newton step; every caller discards that lane's value by a select (flat / parallel /
// zero-row predicates), exactly where the exact quotient would have been +-inf.
__device__ __forceinline__ double rcp_(double a) {
    double x = __builtin_amdgcn_rcp(a);
    x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
    x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
    return x;
}
__device__ __forceinline__ float rsqrt_(float a) { return __builtin_amdgcn_rsqf(a); }
__device__ __forceinline__ double rsqrt_(double a) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
}

// Scale a row to a unit normal.  All-zero rows (the reference's unused rows,
// cbf_qp.py:110-111) stay (0, 0, c).  `poison` accumulates 0 * (row entries): it turns
// NaN as soon as any entry is non-finite, which the solve reports as infeasible.
template <typename T>
__device__ __forceinline__ void normalise_row(T& n0, T& n1, T& c, T& poison) {
    const T nn = n0 * n0 + n1 * n1;
    poison += T(0) * (nn + c);
    const T inv = nn > T(0) ? rsqrt_(nn) : T(1);
    n0 *= inv; n1 *= inv; c *= inv;
}

template <typename T>
struct LineQP {        // state of one incremental step
    T p0, p1, d0, d1, lo, hi;
};

// interval the box leaves on the line p + t d
template <typename T>
__device__ __forceinline__ void clip_box(LineQP<T>& L, const CbfConsts<T>& k) {
    const T inf = num<T>::inf();
    const T tol = num<T>::tol_feas();
    // axis 0
    {
        const T r = rcp_(L.d0);
        const T t1 = (k.lo0 - L.p0) * r, t2 = (k.hi0 - L.p0) * r;
        const bool flat = L.d0 == T(0);
        const bool inside = (L.p0 >= k.lo0 - tol) && (L.p0 <= k.hi0 + tol);
        const T lo = flat ? (inside ? -inf : inf) : fmin_(t1, t2);
        const T hi = flat ? inf : fmax_(t1, t2);
        L.lo = lo; L.hi = hi;
    }
    {
        const T r = rcp_(L.d1);
        const T t1 = (k.lo1 - L.p1) * r, t2 = (k.hi1 - L.p1) * r;
        const bool flat = L.d1 == T(0);
        const bool inside = (L.p1 >= k.lo1 - tol) && (L.p1 <= k.hi1 + tol);
        const T lo = flat ? (inside ? -inf : inf) : fmin_(t1, t2);
        const T hi = flat ? inf : fmax_(t1, t2);
        L.lo = fmax_(L.lo, lo); L.hi = fmin_(L.hi, hi);
    }
}

// restriction an earlier (normalised) row (g0, g1, gc) puts on the line:  a t + r >= 0.
// q = r/|a|:  a >= 0 -> t >= -q ;  a < 0 -> t <= q.   Rows closer than eps_par to parallel
// (|a| = |sin angle|; duplicates of row i, all-zero rows) restrict nothing when the line
// satisfies them and empty the interval when it violates them beyond tolerance.
template <typename T>
__device__ __forceinline__ void clip_row(LineQP<T>& L, T g0, T g1, T gc) {
    const T inf = num<T>::inf();
    const T a = g0 * L.d0 + g1 * L.d1;
    const T r = g0 * L.p0 + (g1 * L.p1 + gc);
    const bool par = fabs_(a) <= num<T>::eps_par();
    const bool par_viol = r < -num<T>::tol_feas() * fmax_(T(1), fabs_(gc));
    T q = r * rcp_(fabs_(a));
    q = par ? (par_viol ? -inf : inf) : q;
    const bool up = par || (a >= T(0));
    L.lo = fmax_(L.lo, up ? -q : -inf);
    L.hi = fmin_(L.hi, up ? inf : q);
}

template <typename T>
struct QpState {
    T u0, u1, ur0, ur1;
};

template <typename T>
__device__ __forceinline__ void qp_begin(QpState<T>& S, T ur0, T ur1, const CbfConsts<T>& k) {
    S.ur0 = ur0; S.ur1 = ur1;
    S.u0 = fmin_(fmax_(ur0, k.lo0), k.hi0);          // optimum of the box alone
    S.u1 = fmin_(fmax_(ur1, k.lo1), k.hi1);
}

// first half of step i: is row i violated at the running optimum?  (sets up the line if so)
template <typename T>
__device__ __forceinline__ bool qp_row_violated(QpState<T>& S, T a0, T a1, T ci, LineQP<T>& L,
                                                const CbfConsts<T>& k) {
    const T s = a0 * S.u0 + (a1 * S.u1 + ci);
    // an all-zero row is "0 >= -c": nothing to project on (the final slack check reports c < 0)
    const bool zero_row = (a0 == T(0)) && (a1 == T(0));
    const bool viol = !zero_row && (s < T(0));
    const T sr = a0 * S.ur0 + (a1 * S.ur1 + ci);
    L.p0 = S.ur0 - sr * a0;                           // projection of u_ref on the line
    L.p1 = S.ur1 - sr * a1;
    L.d0 = -a1; L.d1 = a0;
    return viol;
}

// second half of step i, after the clips
template <typename T>
__device__ __forceinline__ void qp_row_commit(QpState<T>& S, const LineQP<T>& L, bool viol) {
    T t = fmin_(fmax_(T(0), L.lo), L.hi);            // closest point of [lo, hi] to the foot point (t = 0)
    t = (L.lo > L.hi) ? T(0.5) * (L.lo + L.hi) : t;  // empty (rounding, or truly infeasible): split the difference
    const T v0 = L.p0 + t * L.d0, v1 = L.p1 + t * L.d1;
    S.u0 = viol ? v0 : S.u0;
    S.u1 = viol ? v1 : S.u1;
}

// slack of one normalised row at the final point, folded into the running minimum
template <typename T>
__device__ __forceinline__ T qp_row_margin(T worst, T a0, T a1, T ci, T u0, T u1, T& poison) {
    const T s = a0 * u0 + (a1 * u1 + ci);
    const T m = s + num<T>::tol_feas() * fmax_(T(1), fabs_(ci));     // >= 0 when satisfied to tolerance
    poison += T(0) * s;                  // a NaN / inf slack (poisoned walk) must count as violated: min() drops NaN
    return fmin_(worst, m);
}

template <typename T>
__device__ __forceinline__ int qp_status(const QpState<T>& S, T worst, T poison, const CbfConsts<T>& k) {
    const bool finite = (poison == poison) && finite_(S.ur0 + S.ur1);
    const bool box_ok = (k.lo0 <= k.hi0) && (k.lo1 <= k.hi1);
    return (!(worst >= T(0)) || !finite || !box_ok) ? SC_STATUS_INFEASIBLE : SC_STATUS_OPTIMAL;
}

// the box is a hard actuator limit: clamp (removes the last-ulp overshoot of p + t d) before the check
template <typename T>
__device__ __forceinline__ void qp_finish_box(QpState<T>& S, const CbfConsts<T>& k) {
    S.u0 = fmin_(fmax_(S.u0, k.lo0), k.hi0);
    S.u1 = fmin_(fmax_(S.u1, k.lo1), k.hi1);
}

// ---- rows in registers, loops unrolled (K <= KMAX) -----------------------------------
template <typename T, int KMAX>
__device__ __forceinline__ int qp2_solve(const T (&n0)[KMAX], const T (&n1)[KMAX], const T (&c)[KMAX],
                                         int K, T ur0, T ur1, T poison, const CbfConsts<T>& k, T& u0, T& u1) {
    QpState<T> S;
    qp_begin(S, ur0, ur1, k);
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        if (i >= K) break;                         // K is wave-uniform
        LineQP<T> L;
        const bool viol = qp_row_violated(S, n0[i], n1[i], c[i], L, k);
        if (__builtin_amdgcn_ballot_w64(viol) == 0) continue;
        clip_box(L, k);
#pragma unroll
        for (int j = 0; j < i; ++j) clip_row(L, n0[j], n1[j], c[j]);
        qp_row_commit(S, L, viol);
    }
    qp_finish_box(S, k);
    T worst = num<T>::inf();
#pragma unroll
    for (int i = 0; i < KMAX; ++i)
        if (i < K) worst = qp_row_margin(worst, n0[i], n1[i], c[i], S.u0, S.u1, poison);
    u0 = S.u0; u1 = S.u1;
    return qp_status(S, worst, poison, k);
}

// ---- rows in LDS, run-time loops (any K) ------------------------------------------------
// Row (r, comp) of lane l lives at rows[(r*3 + comp)*64 + l]: lane-contiguous, conflict free.
template <typename T>
__device__ __forceinline__ int qp2_solve_lds(const T* rows, int lane, int K, T ur0, T ur1, T poison,
                                             const CbfConsts<T>& k, T& u0, T& u1) {
    QpState<T> S;
    qp_begin(S, ur0, ur1, k);
    const T* mine = rows + lane;
#pragma nounroll
    for (int i = 0; i < K; ++i) {
        LineQP<T> L;
        const bool viol = qp_row_violated(S, mine[(i * 3 + 0) * 64], mine[(i * 3 + 1) * 64],
                                          mine[(i * 3 + 2) * 64], L, k);
        if (__builtin_amdgcn_ballot_w64(viol) == 0) continue;
        clip_box(L, k);
#pragma nounroll
        for (int j = 0; j < i; ++j)
            clip_row(L, mine[(j * 3 + 0) * 64], mine[(j * 3 + 1) * 64], mine[(j * 3 + 2) * 64]);
        qp_row_commit(S, L, viol);
    }
    qp_finish_box(S, k);
    T worst = num<T>::inf();
#pragma nounroll
    for (int i = 0; i < K; ++i)
        worst = qp_row_margin(worst, mine[(i * 3 + 0) * 64], mine[(i * 3 + 1) * 64], mine[(i * 3 + 2) * 64], S.u0, S.u1, poison);
    u0 = S.u0; u1 = S.u1;
    return qp_status(S, worst, poison, k);
}

}  // namespace sc
