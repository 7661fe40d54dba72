// Register-resident Cholesky factorisation and triangular solves of one n x n system per wavefront (n <= 64,
// compile-time), used by the MPC-CBF kernels (mpc_cbf.hip) and testable on its own (tools/test_chol_reg.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "sc_qp2.hpp"

namespace sc {

#ifndef SC_SYNC
#define SC_SYNC() __syncthreads()
#endif

// ---- register-resident Cholesky for a compile-time order (n = 2 NT <= 64) -------------------------------------
// Lane i < n keeps row i of the matrix in VGPRs; a pivot row element is broadcast with v_readlane (the source lane
// is a compile-time constant in the fully unrolled loops), so a column step costs no LDS round trip and no barrier:
// ~1.3 k instructions for n = 20 against 60 barrier-separated LDS passes in cholesky()/chol_solve().
__device__ __forceinline__ double bcast_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// In: a[k] = A[lane][k] (lower triangle used).  Out: a[k] = L[lane][k] for k <= lane, dinv = 1 / L[lane][lane].
// Right-looking; the pivot's reciprocal square root (v_rsq_f64 + two Newton steps) replaces sqrt and the division.
// Returns false on a pivot <= 0.
template <int n>
__device__ __forceinline__ bool chol_reg(double (&a)[n], int lane, double& dinv) {
    dinv = 1.0;
    bool ok = true;                                    // no early exit: one basic block, so the scheduler overlaps the
#pragma unroll                                         // pivot chain of column j + 1 with the trailing update of column j
    for (int j = 0; j < n; ++j) {
        const double d = bcast_lane(a[j], j);
        ok = ok && (d > 0.0);                          // a pivot <= 0 poisons the rest with NaN; the caller retries
        const double r = rsqrt_(d);
        a[j] = (lane == j) ? d * r : a[j] * r;
        dinv = (lane == j) ? r : dinv;
#pragma unroll
        for (int k = j + 1; k < n; ++k) {
            const double lkj = bcast_lane(a[j], k);
            a[k] -= a[j] * lkj;                        // meaningful for lanes >= k; the upper triangle is never read
        }
    }
    return ok;
}

// Solve L L' x = b with L row-held in a[] (from chol_reg), b = this lane's right-hand-side entry.  Lt is an LDS
// scratch of n (n + 1) doubles used once to transpose L (row stride n + 1 keeps the 64 banks conflict-free).
template <int n>
__device__ __forceinline__ double chol_solve_reg(double (&a)[n], double dinv, double b, double* Lt, int lane) {
    constexpr int ld = n + 1;
    const bool act = lane < n;
    // whole rows go to the scratch, the upper-triangle garbage is never selected (one exec region, no per-entry branch)
    if (act) {
#pragma unroll
        for (int k = 0; k < n; ++k) Lt[lane * ld + k] = a[k];
    }
    // rows scaled to a unit diagonal: the forward step is  y_j = b_j;  b_i -= (L_ij / L_ii) y_j
#pragma unroll
    for (int k = 0; k < n; ++k) a[k] = (lane > k && act) ? a[k] * dinv : 0.0;   // strictly lower part: no select below
    b *= dinv;
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const double yj = bcast_lane(b, j);
        b -= a[j] * yj;
    }
    SC_SYNC();
    double c[n];                                        // c[k] = L[k][lane] / L[lane][lane] for k > lane (column of L)
    const int lc = act ? lane : 0;
#pragma unroll
    for (int k = 0; k < n; ++k) {
        const double v = Lt[k * ld + lc];
        c[k] = (k > lane && act) ? v * dinv : 0.0;
    }
    // backward  L' x = y
    b *= dinv;
#pragma unroll
    for (int j = n - 1; j >= 0; --j) {
        const double xj = bcast_lane(b, j);
        b -= c[j] * xj;
    }
    return b;
}

}  // namespace sc
