// Pieces shared by the one-NLP-per-wavefront interior-point kernels mpc_lin.hip and mpc_gn.hip: wave reductions, the planar
// distance barrier, the LDS Cholesky for run-time orders (blocked, MFMA trailing update), the triangular solves and the
// out-of-line wrapper of the register Cholesky (mpc_chol.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "mpc_chol.hpp"
#include "sc_math.hpp"

namespace sc {
namespace ipm {

// Wave reductions on DPP: a __shfl_xor butterfly is six dependent ds_bpermute round trips per 32-bit half (~100 cycles
// each for a lone wave) and an interior-point iteration does about twenty reductions.  Four DPP moves (xor 1, xor 2,
// half-row mirror, row mirror: VALU latency) leave every lane with its 16-lane row total; the four row totals are
// combined from v_readlane, so the result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_mv(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_value(double v, int src) {          // src: compile-time constant lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
template <typename F>
__device__ __forceinline__ double wred(double v, F f) {
    v = f(v, dpp_mv<0xB1>(v));           // quad_perm [1,0,3,2]
    v = f(v, dpp_mv<0x4E>(v));           // quad_perm [2,3,0,1]
    v = f(v, dpp_mv<0x141>(v));          // row_half_mirror
    v = f(v, dpp_mv<0x140>(v));          // row_mirror
    return f(f(row_value(v, 0), row_value(v, 16)), f(row_value(v, 32), row_value(v, 48)));
}
__device__ __forceinline__ double wsum(double v) { return wred(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wmin(double v) { return wred(v, [](double a, double b) { return fmin(a, b); }); }
__device__ __forceinline__ double wmax(double v) { return wred(v, [](double a, double b) { return fmax(a, b); }); }


// h, dh/dp, d2h/dp2 at a planar point: circle (every model) or superellipsoid (single_integrator2D.py:162-181: fabs,
// a, b >= 1e-3, e >= 2); oracle/mpc_cbf.py: barrier
// LDS obstacle rows of the kernels that have the superellipsoid branch are normalised on load (normalise_obstacle_flags):
// slot 6 is 0 for a circle and the ROW SCALE (> 0, 1 until scale_steep_barriers has run) for a superellipsoid.
// x^e, x^(e-1), x^(e-2) for x >= 0, e >= 2.  The exponents the reference's scenes use are small even integers (4, 6, 10:
// examples, dynamic_unicycle2D.py:159-183); for an integer e <= 64 ONE multiply chain gives x^(e-2) and two more products
// the other powers, instead of three library pow() calls of ~150 instructions each -- six per barrier evaluation, which is
// what an interior-point iteration on superellipsoid scenes (BASELINE config 5) mostly consisted of.
__device__ __forceinline__ void pow3(double x, double e, bool derivs, double& pe, double& pe1, double& pe2) {
    const double er = rint(e);
    if (er == e && e <= 64.0) {
        const int n = (int)er - 2;                                         // 0 .. 62: six square-and-multiply steps, straight-line
        double r = 1.0, b = x;                                             // (no lane-divergent loop next to the pow() call below: see CHAIN)
#pragma unroll
        for (int k = 0; k < 6; ++k) { r = ((n >> k) & 1) ? r * b : r; b *= b; }
        pe2 = r; pe1 = r * x; pe = pe1 * x;
    } else {
        pe = pow(x, e);
        pe1 = derivs ? pow(x, e - 1) : 0.0;
        pe2 = derivs ? pow(x, e - 2) : 0.0;
    }
}

// CHAIN: integer exponents by the multiply chain of pow3, non-integer ones by pow().
// History, now closed.  Rounds 1 - 3 met a "code-generation fragility" four times: the chain inlined made the Quad2D kernel -- which never
// executes it -- return garbage; an unrelated LDS-layout change broke mpclin_kernel<2, 2, 10, 0> for K = 6 only; a nested conditional
// made mpcgn_kernel<1, 10, false> stop every solve after five iterations with the feature switched off; the continuation code of round 4
// broke the same kernel again (and a sibling build broke Quad2D instead).  Round 4 root-caused it ("The code-generation
// fragility: root cause"; tools/check_exec_prologue.py): after a divergent loop the compiler re-enables the parked lanes with
// s_or_b64 exec at the top of the join block, and the ROCm 7.2 register allocator places the copies of a VGPR live-range split IN FRONT of
// that instruction when the block starts with reloads of spilled SGPRs -- the copies run under the narrow mask (or EXEC = 0), so the
// lanes that sat out the last loop pass lose loop-invariant per-lane values (LDS addresses, a counter) when they are copied back after
// the Cholesky call.  Which value is hit depends on where the allocator splits: any edit, inlining decision or scheduling change moves
// it.  The MPC translation units are now compiled with the basic VGPR allocator, which never splits a live range (csrc/Makefile:
// SAFE_RA), and the objects are scanned for the signature after every build (tests/test_codegen_guard.py).
template <bool CHAIN = false>
__device__ inline void ipm_barrier(double px_, double py_, const double* o, double Rrob, double beta, bool circles_only, bool derivs,
                                   double& h, double& d0, double& d1, double& hxx, double& hxy, double& hyy) {
    if (circles_only || o[6] == 0.0) {
        const double d = Rrob + o[2];
        const double ex = px_ - o[0], ey = py_ - o[1];
        h = (ex * ex + ey * ey) - beta * d * d;
        d0 = 2.0 * ex; d1 = 2.0 * ey; hxx = 2.0; hxy = 0.0; hyy = 2.0;
        return;
    }
    const double a = fmax(fabs(o[2]), 1e-3) + Rrob, b = fmax(fabs(o[3]), 1e-3) + Rrob;
    const double e = fmax(fabs(o[4]), 2.0);
    double st, ct;
    sincos(o[5], &st, &ct);
    const double dx = px_ - o[0], dy = py_ - o[1];
    const double px = ct * dx + st * dy, py = -st * dx + ct * dy;
    const double ax = fabs(px) / a, ay = fabs(py) / b;
    const double sc = o[6];
    double xe, xe1, xe2, ye, ye1, ye2;
    if constexpr (CHAIN) {
        pow3(ax, e, derivs, xe, xe1, xe2);
        pow3(ay, e, derivs, ye, ye1, ye2);
    } else {
        xe = pow(ax, e); ye = pow(ay, e);
        xe1 = derivs ? pow(ax, e - 1) : 0.0; ye1 = derivs ? pow(ay, e - 1) : 0.0;
        xe2 = derivs ? pow(ax, e - 2) : 0.0; ye2 = derivs ? pow(ay, e - 2) : 0.0;
    }
    h = sc * (xe + ye - 1.0);
    if (!derivs) { d0 = d1 = hxx = hxy = hyy = 0.0; return; }
    const double sx = px > 0 ? 1.0 : (px < 0 ? -1.0 : 0.0), sy = py > 0 ? 1.0 : (py < 0 ? -1.0 : 0.0);
    const double gpx = e * xe1 / a * sx, gpy = e * ye1 / b * sy;
    const double cxx = e * (e - 1) * xe2 / (a * a), cyy = e * (e - 1) * ye2 / (b * b);
    d0 = sc * (ct * gpx - st * gpy);
    d1 = sc * (st * gpx + ct * gpy);
    hxx = sc * (ct * ct * cxx + st * st * cyy);
    hxy = sc * (ct * st * cxx - st * ct * cyy);
    hyy = sc * (st * st * cxx + ct * ct * cyy);
}

// ---- feasibility restoration (include/safe_control_amd.h: sc_resto_params; oracle/mpc_cbf.py: solve) -----------------------
// When the regular phase cannot continue at an infeasible iterate z_R the kernels minimise
//     rho sum_i t_i + zeta/2 |z - z_R|^2   s.t.  g_i(z) + t_i - s_i = 0,  s_i, t_i >= 0   over the CBF rows i (the elastic rows),
// zeta = sqrt(mu), with every other row as in the regular phase, by the SAME primal-dual iteration.  An elastic row keeps its
// slack s and multiplier lam and gains ONE stored number, t; the multiplier of t >= 0 is rho - lam (stationarity in t, kept
// exactly by a common dual step) and dt is eliminated from the Newton system, so the row enters the condensed system as
//     Sigma_eff = Sigma_s Sigma_t / (Sigma_s + Sigma_t),  Sigma_s = lam / s,  Sigma_t = (rho - lam) / t,
// with its own multiplier step at dz = 0 (dl0).  The row pass hands both over in the two arrays the regular phase uses
// (is: lam * is = Sigma;  vb: mu * is - vb = dl0), so the assembly and the column passes do not know about the restoration.
// slack of an elastic row on the central path  mu / s + mu / t = rho,  t = s - g
__device__ __forceinline__ double resto_central_slack(double g, double mu, double rho) {
    return ((2.0 * mu + rho * g) + sqrt(rho * rho * g * g + 4.0 * mu * mu)) / (2.0 * rho);
}
__device__ __forceinline__ void resto_row(double g, double s, double lam, double t, double mu, double rho, double& rp, double& is_eff,
                                          double& vb_eff) {
    rp = g + t - s;
    const double nu = rho - lam, inv_s = rcp_(s);
    const double sig = lam * inv_s, sgt = nu * rcp_(t);
    const double se = sig * sgt * rcp_(sig + sgt);
    const double dl0 = -se * (rp + mu * rcp_(nu) - t) - (se * rcp_(sig)) * (lam - mu * inv_s);
    is_eff = se * rcp_(lam);
    vb_eff = mu * is_eff - dl0;
}
// dt of an elastic row from its multiplier step dl:  (rho - lam) dt - t dl = mu - t (rho - lam)
__device__ __forceinline__ double resto_dt(double lam, double t, double dl, double mu, double rho) {
    const double inu = rcp_(rho - lam);
    return (mu * inu - t) + dl * t * inu;
}
// multiplier safeguard of an elastic row after the step (IPOPT eq. (16) for both lam and rho - lam)
__device__ __forceinline__ double resto_clamp_lam(double lam, double t, double mu, double rho) {
    const double mut = mu * rcp_(t);
    lam = fmin(fmax(lam, rho - 1e10 * mut), rho - 1e-10 * mut);
    return fmin(fmax(lam, 1e-300), rho * (1.0 - 1e-15));
}

__device__ inline void normalise_obstacle_flags(double* obs, int K, int tid, int nthreads) {
    for (int j = tid; j < K; j += nthreads) obs[7 * j + 6] = obs[7 * j + 6] < 0.5 ? 0.0 : 1.0;
}

// Gradient-based scaling of the steep (superellipsoid) barriers -- oracle/mpc_cbf.py: barrier_scales, IPOPT's default NLP
// scaling (nlp_scaling_max_gradient = 100) per obstacle:  h_j <- sc_j h_j,  sc_j = min(1, 100 / max_pts |grad h_j|_inf) over
// the barrier points of the initial guess.  dh holds (d0, d1) of entry e = pt * K + j from the first evaluation (scale 1).
// Returns true when some scale is below 1: the caller re-evaluates.  block_max: block-wide maximum (the same value in
// every thread); sync: block barrier.
template <typename MaxFn, typename SyncFn>
__device__ inline bool scale_steep_barriers(double* obs, int K, const double* dh, int npts, int tid, int nthreads, MaxFn block_max,
                                            SyncFn sync) {
    bool any = false;
    for (int j = 0; j < K; ++j) {
        if (obs[7 * j + 6] == 0.0) continue;                              // block-uniform
        double gm = 0.0;
        for (int pt = tid; pt < npts; pt += nthreads) {
            const int e = pt * K + j;
            gm = fmax(gm, fmax(fabs(dh[2 * e]), fabs(dh[2 * e + 1])));
        }
        gm = block_max(gm);
        const double sc = fmax(fmin(1.0, 100.0 / fmax(gm, 1e-300)), 1e-30);
        any = any || sc < 1.0;
        sync();
        if (tid == 0) obs[7 * j + 6] = sc;
    }
    sync();
    return any;
}


// Cholesky in LDS for run-time orders (L = lower of A, row stride ld), false on a pivot <= 0.  ld is ODD (callers pass
// n + 1 for even n): a column walk A[i * ld + j] over the lanes then touches 32 distinct bank pairs, where the natural
// stride n = 80 doubles maps every row onto two (a 32-way conflict on each column scale, panel update and MFMA operand read).
// Right-looking with panels of 4 columns: the panel is factored column by column (updates confined to the panel), the
// trailing matrix then takes ONE rank-4 update per 16 x 16 tile as a v_mfma_f64_16x16x4_f64 (A = panel rows of the tile's
// rows, B' = panel rows of its columns).  The diagonal is left as 1 / L_jj: the solves multiply.
typedef double ipm_c4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double lane_value(double v, int src);
// n <= 128.  A lone wave hides no latency and issues one instruction every few cycles, so what this routine costs is its
// count of DEPENDENT LDS round trips and of instructions.  The panel is factored in registers (lane l holds the four panel
// entries of rows l and 64 + l; pivots and multipliers travel by v_readlane: one load burst and one store burst per panel
// instead of ~20 round trips and 12 barriers).  The trailing update walks the FIXED 16 x 16 tile grid: rows and columns
// that are not part of the trailing matrix (already final) are switched off by zeroing their MFMA operand, so a tile
// element needs no predicate (the strict upper triangle holds don't-care values); only the last tile row of an order that
// is not a multiple of 16 takes the guarded variant (rows and columns >= n do not exist in the n x ld scratch).
// Storage of the factor.  LdRect: n x ld rectangle (ld odd).  LdTile: only the 16 x 16 tiles on and below the diagonal, tile
// (ti, tj <= ti) at ((ti (ti + 1) / 2 + tj) * 272, rows of a tile 17 doubles apart (odd: conflict-free column walks): 4080
// instead of 6480 doubles at n = 80, which is what lets two order-80 problems share a CU's LDS.  Elements above the diagonal
// TILE ROW do not exist: `at` is only called with (r >> 4) >= (c >> 4); the call sites below that used to read don't-care
// values from there clamp their row / column first (`row_for`, `col_for`).
struct LdRect {
    int ld;
    __device__ __forceinline__ int at(int r, int c) const { return r * ld + c; }
    __device__ __forceinline__ int row_for(int r, int c) const { return r; }
    __device__ __forceinline__ int col_for(int r, int c) const { return c; }
};
struct LdTile {
    __device__ __forceinline__ int at(int r, int c) const {
        const int ti = r >> 4, tj = c >> 4;
        return (((ti * (ti + 1)) >> 1) + tj) * 272 + (r & 15) * 17 + (c & 15);
    }
    __device__ __forceinline__ int row_for(int r, int c) const { return (r >> 4) >= (c >> 4) ? r : c; }   // an existing row for column c
    __device__ __forceinline__ int col_for(int r, int c) const { return (r >> 4) >= (c >> 4) ? c : r; }   // an existing column for row r
    static __host__ __device__ inline size_t doubles(int n) { const size_t nt = ((size_t)n + 15) >> 4; return nt * (nt + 1) / 2 * 272; }
};

template <int U, bool GUARD, typename IX>
__device__ __forceinline__ void chol_tiles(double* A, const IX ix, int j0, int t0, int n, int pw, int ti, int tj0, double a, int q, int l15) {
    double b[U];
    ipm_c4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int rb = 16 * (tj0 + u) + l15;
        const double bv = A[ix.at(GUARD && rb >= n ? j0 : rb, j0 + q)];
        b[u] = (rb >= t0 && rb < n && q < pw) ? bv : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool in = !GUARD || (16 * ti + q + 4 * r < n && rb < n);
            acc[u][r] = in ? A[ix.at(16 * ti + q + 4 * r, 16 * (tj0 + u) + l15)] : 0.0;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[u], acc[u], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool in = !GUARD || (16 * ti + q + 4 * r < n && 16 * (tj0 + u) + l15 < n);
            if (in) A[ix.at(16 * ti + q + 4 * r, 16 * (tj0 + u) + l15)] = acc[u][r];
        }
    }
}
// One panel (columns j0 .. j0 + pw - 1) in the registers of ONE wave; false (wave-uniform) on a pivot <= 0.
template <typename IX>
__device__ __forceinline__ bool chol_panel(double* A, int n, const IX ix, int j0, int pw, int lane) {
    const int r0 = lane, r1 = 64 + lane;
    const int c1 = r1 < n ? r1 : r0;                                       // in-range stand-in row for the loads of lanes without a second row
    const int l0 = ix.row_for(r0 < n ? r0 : j0, j0), l1 = ix.row_for(c1 < n ? c1 : j0, j0);   // rows whose (., j0) entries exist
    double p0[4], p1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        p0[c] = A[ix.at(l0, j0 + (c < pw ? c : 0))];
        p1[c] = A[ix.at(l1, j0 + (c < pw ? c : 0))];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        if (jj < pw) {
            const int j = j0 + jj;
            const double dd = j < 64 ? lane_value(p0[jj], j) : lane_value(p1[jj], j - 64);
            if (!(dd > 0.0)) return false;
            const double inv = rsqrt_(dd);                                 // v_rsq_f64 + two Newton steps (sc_qp2.hpp), as mpc_chol.hpp
            p0[jj] = r0 == j ? inv : p0[jj] * inv;                         // (rows above the diagonal carry don't-care values)
            p1[jj] = r1 == j ? inv : p1[jj] * inv;
#pragma unroll
            for (int c = jj + 1; c < 4; ++c) {
                if (c < pw) {
                    const int rc = j0 + c;
                    const double lc = rc < 64 ? lane_value(p0[jj], rc) : lane_value(p1[jj], rc - 64);
                    p0[c] -= p0[jj] * lc;
                    p1[c] -= p1[jj] * lc;
                }
            }
        }
    }
    if (r0 >= j0 && r0 < n) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < pw) A[ix.at(r0, j0 + c)] = p0[c];
    }
    if (r1 >= j0 && r1 < n) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < pw) A[ix.at(r1, j0 + c)] = p1[c];
    }
    return true;
}
// NW waves share the factorisation (tid = thread index among them): wave 0 factors the panel and posts its verdict in
// `flag` (an LDS word), the tiles of the trailing update are dealt round-robin.  NW = 1: flag unused.
template <int NW = 1, typename IX = LdRect>
__device__ __forceinline__ bool cholesky_ix(double* A, int n, const IX ix, int tid, double* flag = nullptr) {
    const int lane = tid & 63, wv = tid >> 6;
    const int q = lane >> 4, l15 = lane & 15;
    for (int j0 = 0; j0 < n; j0 += 4) {
        const int pw = n - j0 < 4 ? n - j0 : 4;
        bool ok = true;
        if (NW == 1 || wv == 0) ok = chol_panel(A, n, ix, j0, pw, lane);
        if constexpr (NW > 1) {
            if (tid == 0) *flag = ok ? 1.0 : 0.0;
            SC_SYNC();
            ok = *flag != 0.0;
        } else {
            SC_SYNC();
        }
        if (!ok) return false;
        const int t0 = j0 + pw;                                            // trailing matrix starts here
        if (t0 >= n) break;
        const int tb = t0 >> 4, te = (n + 15) >> 4;
        int item = 0;
        for (int ti = tb; ti < te; ++ti) {
            const int ra = 16 * ti + l15;
            const double av = A[ix.at(ra < n ? ra : j0, j0 + q)];
            const double a = (ra >= t0 && ra < n && q < pw) ? -av : 0.0;
            int tj = tb;
            if (16 * ti + 16 <= n) {
                if constexpr (NW == 1) {
                    for (; tj + 3 <= ti; tj += 4) chol_tiles<4, false>(A, ix, j0, t0, n, pw, ti, tj, a, q, l15);
                    if (tj + 1 <= ti) { chol_tiles<2, false>(A, ix, j0, t0, n, pw, ti, tj, a, q, l15); tj += 2; }
                    if (tj <= ti) chol_tiles<1, false>(A, ix, j0, t0, n, pw, ti, tj, a, q, l15);
                } else {
                    for (; tj <= ti; ++tj)
                        if ((item++ & (NW - 1)) == wv) chol_tiles<1, false>(A, ix, j0, t0, n, pw, ti, tj, a, q, l15);
                }
            } else {
                for (; tj <= ti; ++tj)
                    if (NW == 1 || (item++ & (NW - 1)) == wv) chol_tiles<1, true>(A, ix, j0, t0, n, pw, ti, tj, a, q, l15);
            }
        }
        SC_SYNC();
    }
    return true;
}
template <int NW = 1>
__device__ __forceinline__ bool cholesky_lds(double* A, int n, int ld, int tid, double* flag = nullptr) {
    return cholesky_ix<NW, LdRect>(A, n, LdRect{ld}, tid, flag);
}
__device__ __forceinline__ double lane_value(double v, int src) {           // src: wave-uniform lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// L L' x = b in place, n <= 128, on ONE wave: lane l keeps entries l and 64 + l of the right-hand side AND the two
// reciprocal pivots of its rows in registers; the pivot entry and its reciprocal are broadcast with v_readlane, so the 2 n
// elimination steps run without a barrier and with no LDS access on their dependency chain: the multipliers of 16 steps
// (a block column of L going down, a block row going up) are loaded in one burst ahead of them.
// TWO: n > 64.  LO: the 16 pivots of the block are first-row entries (c0 < 64).
template <bool TWO, bool LO, bool FWD, typename IX>
__device__ __forceinline__ void chol_solve_block(const double* L, int n, const IX ix, int c0, int lane, int k0, int k1, double d0, double d1,
                                                 double& b0, double& b1) {
    const int i0 = lane, i1 = 64 + lane;
    double m0[16], m1[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int j = c0 + c < n ? c0 + c : n - 1;
        if (FWD) {                                                         // column j of L, rows of this lane (rows above it: unused, clamped)
            if (LO) m0[c] = L[ix.at(ix.row_for(k0, j), j)];
            if (TWO) m1[c] = L[ix.at(ix.row_for(k1, j), j)];
        } else {                                                           // row j of L, columns of this lane (columns right of it: unused, clamped)
            m0[c] = L[ix.at(j, ix.col_for(j, k0))];
            if (TWO && !LO) m1[c] = L[ix.at(j, ix.col_for(j, k1))];
        }
    }
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int c = FWD ? cc : 15 - cc, j = c0 + c;
        if (j < n) {                                                       // wave-uniform
            const int jl = j & 63;
            const double v = (LO ? lane_value(b0, jl) : lane_value(b1, jl)) * (LO ? lane_value(d0, jl) : lane_value(d1, jl));
            if (FWD) {
                if (LO) { b0 = i0 == j ? v : (i0 > j ? b0 - m0[c] * v : b0); if (TWO) b1 -= m1[c] * v; }
                else b1 = i1 == j ? v : (i1 > j ? b1 - m1[c] * v : b1);
            } else {
                if (LO) b0 = i0 == j ? v : (i0 < j ? b0 - m0[c] * v : b0);
                else { b1 = i1 == j ? v : (i1 < j ? b1 - m1[c] * v : b1); b0 -= m0[c] * v; }
            }
        }
    }
}
template <bool TWO, typename IX>
__device__ __forceinline__ void chol_solve_wave(const double* L, double* b, int n, const IX ix, int lane) {
    const int i0 = lane, i1 = 64 + lane;
    const int k0 = i0 < n ? i0 : 0, k1 = i1 < n ? i1 : k0;                 // in-range rows for the unpredicated loads
    double b0 = i0 < n ? b[i0] : 0.0, b1 = (TWO && i1 < n) ? b[i1] : 0.0;
    const double d0 = L[ix.at(k0, k0)], d1 = TWO ? L[ix.at(k1, k1)] : 0.0;
    const int nb = (n + 15) >> 4;
    for (int kb = 0; kb < nb; ++kb) {
        if (!TWO || kb < 4) chol_solve_block<TWO, true, true>(L, n, ix, 16 * kb, lane, k0, k1, d0, d1, b0, b1);
        else chol_solve_block<TWO, false, true>(L, n, ix, 16 * kb, lane, k0, k1, d0, d1, b0, b1);
    }
    for (int kb = nb - 1; kb >= 0; --kb) {
        if (!TWO || kb < 4) chol_solve_block<TWO, true, false>(L, n, ix, 16 * kb, lane, k0, k1, d0, d1, b0, b1);
        else chol_solve_block<TWO, false, false>(L, n, ix, 16 * kb, lane, k0, k1, d0, d1, b0, b1);
    }
    if (i0 < n) b[i0] = b0;
    if (TWO && i1 < n) b[i1] = b1;
}
template <int NW = 1, typename IX = LdRect>
__device__ __forceinline__ void chol_solve_ix(const double* L, double* b, int n, const IX ix, int tid) {
    if (NW == 1 || tid < 64) {                                             // the other waves of the problem wait for wave 0
        if (n > 64) chol_solve_wave<true>(L, b, n, ix, tid);
        else chol_solve_wave<false>(L, b, n, ix, tid);
    }
    SC_SYNC();
}
template <int NW = 1>
__device__ __forceinline__ void chol_solve_lds(const double* L, double* b, int n, int ld, int tid) {
    chol_solve_ix<NW, LdRect>(L, b, n, LdRect{ld}, tid);
}

// register Cholesky for a compile-time order (mpc_chol.hpp); out of line like mpc_cbf.hip's (code size)
template <int nn>
__device__ __noinline__ bool chol_reg_solve(const double* M, const double* rhs, double* Lt, double* out, double delta, int lane) {
    double a[nn], diag;
    const int row = lane < nn ? lane : 0;
#pragma unroll
    for (int k = 0; k < nn; ++k) a[k] = M[row * nn + k] + (lane == k ? delta : 0.0);
    if (!chol_reg<nn>(a, lane, diag)) return false;
    const double x = chol_solve_reg<nn>(a, diag, rhs[row], Lt, lane);
    if (lane < nn) out[lane] = x;
    SC_SYNC();
    return true;
}


}  // namespace ipm
}  // namespace sc
