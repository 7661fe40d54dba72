// Exact 2-variable projection QP, one problem per lane.
//
//   minimise ||u - u_ref||^2   s.t.  n0[i] u0 + n1[i] u1 + c[i] >= 0  (i < K),  lo <= u <= hi
//
// This is what cvxpy -> GUROBI solves at position_control/cbf_qp.py:190.  The
// objective is strictly convex, so the minimiser is unique.  The kernel walks
// the constraints incrementally (Seidel-style): the running optimum u of
// {box, rows 0..i-1} is kept; if row i is violated at u, the optimum of
// {box, rows 0..i} lies on the line of row i, where the problem is the
// projection of u_ref onto that line clipped to the interval the earlier
// constraints leave.  Interval ends are kept as fractions (num, den > 0) and
// compared by cross-multiplication, so a solve costs at most one division per
// violated row.  Rows are held in registers; everything is predicated, the
// only branch is a wave-uniform skip when no lane violates row i.
#pragma once
#include "sc_models.hpp"

namespace sc {

template <typename T>
struct Interval {      // t in [lo_n/lo_d, hi_n/hi_d], dens >= 0 (0 = unbounded)
    T lo_n, lo_d, hi_n, hi_d;
    bool par_bad;      // a parallel earlier constraint excludes the whole line
};

// clip the line  u(t) = p + t d  against  a t + r >= 0   where  a = g.d, r = g.p + gc
template <typename T>
__device__ __forceinline__ void clip(Interval<T>& I, T a, T r, T par2, T tol_r) {
    const bool is_par = a * a <= par2;            // |sin angle| <= eps_par
    const bool up = a > T(0);
    // candidate bound t* = -r / a  as a fraction with positive denominator
    const T cn = up ? -r : r;
    const T cd = up ? a : -a;
    const bool take_lo = !is_par && up && (cn * I.lo_d > I.lo_n * cd);
    const bool take_hi = !is_par && !up && (cn * I.hi_d < I.hi_n * cd);
    I.lo_n = take_lo ? cn : I.lo_n;
    I.lo_d = take_lo ? cd : I.lo_d;
    I.hi_n = take_hi ? cn : I.hi_n;
    I.hi_d = take_hi ? cd : I.hi_d;
    I.par_bad |= is_par && (r < -tol_r);
}

// Returns SC_STATUS_OPTIMAL / SC_STATUS_INFEASIBLE; u0,u1 valid when optimal.
template <typename T, int KMAX>
__device__ __forceinline__ int qp2_solve(const T (&n0)[KMAX], const T (&n1)[KMAX], const T (&c)[KMAX],
                                         int K, T ur0, T ur1, const CbfConsts<T>& k, T& u0, T& u1) {
    const T tol = num<T>::tol_feas();
    const T epar = num<T>::eps_par();
    const T epar2 = epar * epar;
    bool infeas = (k.lo0 > k.hi0) || (k.lo1 > k.hi1);
    bool finite = finite_(ur0 + ur1);
    u0 = fmin_(fmax_(ur0, k.lo0), k.hi0);        // optimum of the box alone
    u1 = fmin_(fmax_(ur1, k.lo1), k.hi1);

#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
        if (i >= K) break;                         // K is wave-uniform
        const T a0 = n0[i], a1 = n1[i], ci = c[i];
        const T nn = a0 * a0 + a1 * a1;
        finite = finite && finite_(nn + ci);
        const bool zero_row = !(nn > T(0));
        infeas |= zero_row && (ci < -tol * fmax_(T(1), fabs_(ci)));
        const T s = a0 * u0 + a1 * u1 + ci;
        const bool viol = !zero_row && (s < T(0));
        if (__builtin_amdgcn_ballot_w64(viol) == 0) continue;

        const T inv = T(1) / nn;
        const T sr = (a0 * ur0 + a1 * ur1 + ci) * inv;
        const T p0 = ur0 - sr * a0, p1 = ur1 - sr * a1;   // projection of u_ref on the line
        const T d0 = -a1, d1 = a0;                        // line direction, |d|^2 = nn
        Interval<T> I{T(-1), T(0), T(1), T(0), false};
        const T par_box = epar2 * nn;
        const T tol_box = tol * fmax_(T(1), fmax_(fmax_(fabs_(k.lo0), fabs_(k.hi0)), fmax_(fabs_(k.lo1), fabs_(k.hi1))));
        clip(I, d0, p0 - k.lo0, par_box, tol_box);
        clip(I, -d0, k.hi0 - p0, par_box, tol_box);
        clip(I, d1, p1 - k.lo1, par_box, tol_box);
        clip(I, -d1, k.hi1 - p1, par_box, tol_box);
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const T g0 = n0[j], g1 = n1[j], gc = c[j];
            const T gg = g0 * g0 + g1 * g1;
            const T a = g0 * d0 + g1 * d1;
            const T r = g0 * p0 + g1 * p1 + gc;
            // zero rows (gg == 0) give a = 0, r = gc: handled as "parallel"; their
            // own feasibility was already accounted for above.
            clip(I, a, r, epar2 * gg * nn, tol * fmax_(T(1), fabs_(gc)));
        }
        // empty interval?  lo > hi (+ relative slack)
        const T gap = I.lo_n * I.hi_d - I.hi_n * I.lo_d;
        const T gsc = I.lo_d * I.hi_d + fabs_(I.lo_n) * I.hi_d + fabs_(I.hi_n) * I.lo_d;
        const bool empty = gap > tol * gsc;
        T t = T(0);
        if (I.lo_n > T(0)) t = I.lo_n / I.lo_d;           // lo_d > 0 whenever lo_n > 0
        else if (I.hi_n < T(0)) t = I.hi_n / I.hi_d;
        const T v0 = p0 + t * d0, v1 = p1 + t * d1;
        u0 = viol ? v0 : u0;
        u1 = viol ? v1 : u1;
        infeas |= viol && (empty || I.par_bad);
    }
    return (infeas || !finite) ? SC_STATUS_INFEASIBLE : SC_STATUS_OPTIMAL;
}

// Same walk with the rows held in LDS instead of registers (K > 8: 3*K values
// per lane no longer fit the register file in f64).  Row (r, comp) of lane l
// lives at rows[(r*3 + comp)*64 + l]: lane-contiguous, bank-conflict free.
template <typename T>
__device__ __forceinline__ int qp2_solve_lds(const T* rows, int lane, int K, T ur0, T ur1,
                                             const CbfConsts<T>& k, T& u0, T& u1) {
    const T tol = num<T>::tol_feas();
    const T epar = num<T>::eps_par();
    const T epar2 = epar * epar;
    bool infeas = (k.lo0 > k.hi0) || (k.lo1 > k.hi1);
    bool finite = finite_(ur0 + ur1);
    u0 = fmin_(fmax_(ur0, k.lo0), k.hi0);
    u1 = fmin_(fmax_(ur1, k.lo1), k.hi1);
    const T tol_box = tol * fmax_(T(1), fmax_(fmax_(fabs_(k.lo0), fabs_(k.hi0)), fmax_(fabs_(k.lo1), fabs_(k.hi1))));
    const T* mine = rows + lane;
#pragma nounroll
    for (int i = 0; i < K; ++i) {
        const T a0 = mine[(i * 3 + 0) * 64], a1 = mine[(i * 3 + 1) * 64], ci = mine[(i * 3 + 2) * 64];
        const T nn = a0 * a0 + a1 * a1;
        finite = finite && finite_(nn + ci);
        const bool zero_row = !(nn > T(0));
        infeas |= zero_row && (ci < -tol * fmax_(T(1), fabs_(ci)));
        const T s = a0 * u0 + a1 * u1 + ci;
        const bool viol = !zero_row && (s < T(0));
        if (__builtin_amdgcn_ballot_w64(viol) == 0) continue;

        const T inv = T(1) / nn;
        const T sr = (a0 * ur0 + a1 * ur1 + ci) * inv;
        const T p0 = ur0 - sr * a0, p1 = ur1 - sr * a1;
        const T d0 = -a1, d1 = a0;
        Interval<T> I{T(-1), T(0), T(1), T(0), false};
        const T par_box = epar2 * nn;
        clip(I, d0, p0 - k.lo0, par_box, tol_box);
        clip(I, -d0, k.hi0 - p0, par_box, tol_box);
        clip(I, d1, p1 - k.lo1, par_box, tol_box);
        clip(I, -d1, k.hi1 - p1, par_box, tol_box);
#pragma nounroll
        for (int j = 0; j < i; ++j) {
            const T g0 = mine[(j * 3 + 0) * 64], g1 = mine[(j * 3 + 1) * 64], gc = mine[(j * 3 + 2) * 64];
            const T gg = g0 * g0 + g1 * g1;
            const T a = g0 * d0 + g1 * d1;
            const T r = g0 * p0 + g1 * p1 + gc;
            clip(I, a, r, epar2 * gg * nn, tol * fmax_(T(1), fabs_(gc)));
        }
        const T gap = I.lo_n * I.hi_d - I.hi_n * I.lo_d;
        const T gsc = I.lo_d * I.hi_d + fabs_(I.lo_n) * I.hi_d + fabs_(I.hi_n) * I.lo_d;
        const bool empty = gap > tol * gsc;
        T t = T(0);
        if (I.lo_n > T(0)) t = I.lo_n / I.lo_d;
        else if (I.hi_n < T(0)) t = I.hi_n / I.hi_d;
        const T v0 = p0 + t * d0, v1 = p1 + t * d1;
        u0 = viol ? v0 : u0;
        u1 = viol ? v1 : u1;
        infeas |= viol && (empty || I.par_bad);
    }
    return (infeas || !finite) ? SC_STATUS_INFEASIBLE : SC_STATUS_OPTIMAL;
}

}  // namespace sc
