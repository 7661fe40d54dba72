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

#include <utility>

#include <cstdlib>

#include "sc_qp2.hpp"
#include "sc_group.hpp"

namespace sc {

// batches up to this many agents use the cooperative (8 lanes per agent) kernel: below it the
// lane-per-QP kernel cannot fill the chip (65536 agents = 1024 waves = one per SIMD)
#ifndef SC_COOP_MAX_AGENTS
#define SC_COOP_MAX_AGENTS 32768
#endif

// test / profiling hook: SC_FORCE_LDS_KERNEL=1 routes K > 8 to the LDS-staged kernel instead of the
// row-per-lane kernel
static inline long long sc_coop_max_agents() {             // SC_COOP_MAX_AGENTS=<n> overrides the compiled default
    static const long long v = [] { const char* e = getenv("SC_COOP_MAX_AGENTS"); return e ? atoll(e) : (long long)SC_COOP_MAX_AGENTS; }();
    return v;
}
static inline bool sc_two_pass() {                         // SC_CBFQP_TWO_PASS=0: the single-launch register kernel for every model
    static const bool v = [] { const char* e = getenv("SC_CBFQP_TWO_PASS"); return !(e && e[0] == '0'); }();
    return v;
}
static inline bool sc_force_lds_kernel() {
    static const bool v = [] { const char* e = getenv("SC_FORCE_LDS_KERNEL"); return e && e[0] == '1'; }();
    return v;
}

using as1_void = const __attribute__((address_space(1))) void;
using as3_void = __attribute__((address_space(3))) void;

// State row -> (x0..x4) in the compute type.  4-state models: two 8/16-byte loads; Quad2D: six values per row.
template <typename TIO, typename TC, int MODEL>
__device__ __forceinline__ Agent<TC> load_agent(const TIO* __restrict__ X, long long agent) {
    if constexpr (MODEL == SC_MODEL_QUAD2D) {
        const TIO* r = X + agent * 6;
        return make_agent_m<TC, MODEL>(TC(r[0]), TC(r[1]), TC(r[2]), TC(r[3]), TC(r[4]));
    } else {
        const TIO* r = X + agent * 4;
        return make_agent_m<TC, MODEL>(TC(r[0]), TC(r[1]), TC(r[2]), TC(r[3]));
    }
}

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

// ======================================================================================
// K <= 8: register kernel.  No LDS at all: each lane pulls its own agent's K x 7 obstacle
// values straight into VGPRs with 16-byte loads (the block is contiguous and 16-byte aligned
// in the reference's [B,K,7] layout; every byte of every cache line is consumed by the lanes
// that touch it, the 8 partial touches of a line merge in the vector L1), builds the rows in
// registers with the row loop unrolled, and runs the unrolled walk.  With no LDS footprint the
// occupancy is set by VGPRs only, which is what hides the HBM latency of a pure streaming pass.
template <typename TIO> struct vec4io;
template <> struct vec4io<float> { using type = float4; static constexpr int N = 4; };
template <> struct vec4io<double> { using type = double2; static constexpr int N = 2; };

// PASS (DynamicUnicycle2D with f64 arithmetic only; 0 everywhere else = the plain one-launch kernel):
//   1  circles-only fast pass: a wave that finds any obstacle flag != 0 among its agents' rows writes SC_STATUS_PENDING for
//      them and leaves; every other wave runs a body WITHOUT the out-of-line superellipsoid call -- 166 instead of 195 VGPRs
//      (three waves per SIMD instead of two), no spilled SGPRs: 1.55 -> 1.35 ms at 2^24 agents;
//   2  the generic body for exactly the waves pass 1 deferred (it reads status_out first; the others leave at once).
#define SC_STATUS_PENDING (-1)
template <typename TIO, typename TC, int KMAX, int MODEL, int PASS = 0>
__global__ __launch_bounds__(256) void cbfqp_reg_kernel(const sc_cbfqp_params p, const long long B, const int K,
                                                        const TIO* __restrict__ X, const TIO* __restrict__ u_ref,
                                                        const TIO* __restrict__ obs, const int* __restrict__ n_obs,
                                                        TIO* __restrict__ u_out, int* __restrict__ status_out,
                                                        TIO* __restrict__ h_out) {
    const long long agent = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = agent < B;
    const long long ag_i = active ? agent : 0;           // inactive lanes compute on agent 0, store nothing
    using V2 = typename vec2<TIO>::type;
    using V4 = typename vec4io<TIO>::type;
    constexpr int VN = vec4io<TIO>::N;
    if constexpr (PASS == 2) {
        const bool pending = active && status_out[ag_i] == SC_STATUS_PENDING;
        if (__builtin_amdgcn_ballot_w64(pending) == 0ull) return;          // wave-uniform: pass 1 finished this wave
    }

    // ---- loads: obstacles first (longest latency), then state ---------------------------
    TIO flat[KMAX * 7];
    const TIO* src = obs + (p.obs_shared ? 0 : (size_t)ag_i * K * 7);
    if (K == KMAX && ((KMAX * 7) % VN == 0) && ((reinterpret_cast<uintptr_t>(obs) & 15u) == 0) &&
        ((KMAX * 7 * sizeof(TIO)) % 16 == 0)) {
        const V4* s4 = reinterpret_cast<const V4*>(src);
#pragma unroll
        for (int e = 0; e < KMAX * 7 / VN; ++e) {
            const V4 q = s4[e];
            if constexpr (VN == 4) {
                flat[e * 4 + 0] = q.x; flat[e * 4 + 1] = q.y; flat[e * 4 + 2] = q.z; flat[e * 4 + 3] = q.w;
            } else {
                flat[e * 2 + 0] = q.x; flat[e * 2 + 1] = q.y;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < KMAX; ++r) {
#pragma unroll
            for (int f = 0; f < 7; ++f) flat[r * 7 + f] = (r < K) ? src[r * 7 + f] : TIO(0);
        }
    }
    const V2 ur = reinterpret_cast<const V2*>(u_ref)[ag_i];
    int nk = K;
    if (n_obs) {
        nk = n_obs[ag_i];
        nk = nk < 0 ? 0 : (nk > K ? K : nk);
    }
    if constexpr (PASS == 1) {
        bool other = false;
#pragma unroll
        for (int r = 0; r < KMAX; ++r) other |= (r < nk) && (flat[r * 7 + 6] != TIO(0));
        if (__builtin_amdgcn_ballot_w64(other && active) != 0ull) {        // wave-uniform: leave this wave to pass 2
            if (active) status_out[agent] = SC_STATUS_PENDING;
            return;
        }
    }
    const TC ur0 = TC(ur.x), ur1 = TC(ur.y);
    const CbfConsts<TC> k = make_consts<TC>(p);
    const Agent<TC> ag = load_agent<TIO, TC, MODEL>(X, ag_i);

    // ---- rows: agent_barrier + cbf_qp.py:155-183, unrolled, in registers ------------------
    TC n0[KMAX], n1[KMAX], c[KMAX];
    TIO hv[KMAX];
    bool bad_obs = false;
    TC poison = TC(0);
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
        TC o[7];
#pragma unroll
        for (int f = 0; f < 7; ++f) o[f] = TC(flat[r * 7 + f]);
        TC h, a0, a1, cc;
        const bool ok = cbf_row<TC, MODEL, true, PASS == 1>(ag, o, k, a0, a1, cc, h);
        const bool used = r < nk;                   // nk <= K <= KMAX
        bad_obs |= used && !ok;
        a0 = used ? a0 : TC(0); a1 = used ? a1 : TC(0); cc = used ? cc : TC(0);
        normalise_row(a0, a1, cc, poison);
        n0[r] = a0; n1[r] = a1; c[r] = cc;
        hv[r] = used ? TIO(h) : TIO(0);
    }

    // ---- solve + outputs --------------------------------------------------------------------
    TC u0, u1;
    int st = qp2_solve<TC, KMAX>(n0, n1, c, K, ur0, ur1, poison, k, u0, u1);
    if (bad_obs) st = SC_STATUS_BAD_OBSTACLE;
    if (st != SC_STATUS_OPTIMAL) { u0 = num<TC>::nan(); u1 = num<TC>::nan(); }
    if (active) {
        V2 uo; uo.x = TIO(u0); uo.y = TIO(u1);
        reinterpret_cast<V2*>(u_out)[agent] = uo;
        status_out[agent] = st;
        if (h_out) {
            TIO* hp = h_out + agent * K;
            if (K == KMAX && (KMAX % VN == 0) && ((reinterpret_cast<uintptr_t>(h_out) & 15u) == 0)) {
                V4* h4 = reinterpret_cast<V4*>(hp);
#pragma unroll
                for (int e = 0; e < KMAX / VN; ++e) {
                    V4 q;
                    if constexpr (VN == 4) { q.x = hv[e * 4]; q.y = hv[e * 4 + 1]; q.z = hv[e * 4 + 2]; q.w = hv[e * 4 + 3]; }
                    else { q.x = hv[e * 2]; q.y = hv[e * 2 + 1]; }
                    h4[e] = q;
                }
            } else {
#pragma unroll
                for (int r = 0; r < KMAX; ++r) if (r < K) hp[r] = hv[r];
            }
        }
    }
}

// ======================================================================================
// K <= 8, small batches: cooperative kernel, 8 lanes per agent (one obstacle row per lane).
//
// At BASELINE's 4096-agent batch the lane-per-QP kernel runs 64 waves on a chip with 1024 SIMDs and
// its time is one wave's instruction stream (~1700 VALU).  Here an agent is spread over 8 lanes:
// each lane builds ONE row, the incremental walk broadcasts row i inside the 8-lane group, every
// lane clips the line against its own row in parallel, and the interval ends are combined with
// three xor-shuffle steps.  512 waves instead of 64, each ~4x shorter; with only 8 agents per wave
// the wave-uniform "nobody violates row i" skip fires most of the time.  Arithmetic per row is the
// same code as the lane-per-QP kernel (min/max reductions are exact), so both give the same answer.
// G lanes per agent (G = 8, 16 or 32 >= K), 64 / G agents per wave.
template <typename TIO, typename TC, int G, int MODEL>
#ifndef SC_COOP_WAVES
#define SC_COOP_WAVES 1                                   // waves per workgroup of the cooperative kernel (developer builds time 2 and 4)
#endif
__global__ __launch_bounds__(64 * SC_COOP_WAVES) void cbfqp_coop_kernel(const sc_cbfqp_params p, const long long B, const int K,
                                                        const TIO* __restrict__ X, const TIO* __restrict__ u_ref,
                                                        const TIO* __restrict__ obs, const int* __restrict__ n_obs,
                                                        TIO* __restrict__ u_out, int* __restrict__ status_out,
                                                        TIO* __restrict__ h_out) {
    constexpr int APW = 64 / G;                            // agents per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane & (G - 1);                        // obstacle row handled by this lane
    const long long agent = ((long long)blockIdx.x * SC_COOP_WAVES + (threadIdx.x >> 6)) * APW + lane / G;
    const bool active = agent < B;
    const long long ag_i = active ? agent : 0;
    using V2 = typename vec2<TIO>::type;

    TIO orow[7];
    const bool has_row = sub < K;
    const TIO* src = obs + (p.obs_shared ? 0 : (size_t)ag_i * K * 7) + (has_row ? sub * 7 : 0);
#pragma unroll
    for (int f = 0; f < 7; ++f) orow[f] = src[f];
    const V2 ur = reinterpret_cast<const V2*>(u_ref)[ag_i];
    int nk = K;
    if (n_obs) {
        nk = n_obs[ag_i];
        nk = nk < 0 ? 0 : (nk > K ? K : nk);
    }
    const TC ur0 = TC(ur.x), ur1 = TC(ur.y);
    const CbfConsts<TC> k = make_consts<TC>(p);
    const Agent<TC> ag = load_agent<TIO, TC, MODEL>(X, ag_i);

    // ---- this lane's row -------------------------------------------------------------------
    TC o[7];
#pragma unroll
    for (int f = 0; f < 7; ++f) o[f] = TC(orow[f]);
    TC h, a0, a1, cc;
#ifndef SC_EXP_NOROW
    const bool ok = cbf_row<TC, MODEL, false>(ag, o, k, a0, a1, cc, h);
#else
    const bool ok = true; a0 = o[0] - ag.x; a1 = o[1] - ag.y; cc = o[2]; h = o[3];
#endif
    const bool used = sub < nk;
    const bool bad_mine = used && !ok;
    a0 = used ? a0 : TC(0); a1 = used ? a1 : TC(0); cc = used ? cc : TC(0);
    TC poison = TC(0);
    normalise_row(a0, a1, cc, poison);

    // ---- cooperative walk ----------------------------------------------------------------------
    QpState<TC> S;
    qp_begin(S, ur0, ur1, k);
#ifndef SC_EXP_NOWALK                                     // developer builds only (tools/README.md): what the solve costs in the launch
#ifndef SC_COOP_OLD_WALK
    if constexpr (G == 8) coop_solve_all8<TC>(S, K, sub, lane, a0, a1, cc, k);          // fixed critical path (sc_group.hpp)
    else if constexpr (G == 16) coop_solve_all16<TC>(S, K, sub, lane, a0, a1, cc, k);
    else
#endif
    coop_walk_violated<TC, G>(S, K, sub, lane, a0, a1, cc, k);
#endif
    qp_finish_box(S, k);
    TC worst = qp_row_margin(num<TC>::inf(), a0, a1, cc, S.u0, S.u1, poison);
    worst = group_min<TC, G>(worst);
    // NaN anywhere in the group must poison the status: min() drops NaN, so combine the flags explicitly
    const bool nan_mine = !(poison == poison);
    const unsigned long long nan_mask = __builtin_amdgcn_ballot_w64(nan_mine);
    const unsigned long long bad_mask = __builtin_amdgcn_ballot_w64(bad_mine);
    const unsigned long long grp = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane & ~(G - 1));
    if (nan_mask & grp) poison = num<TC>::nan();
    int st = qp_status(S, worst, poison, k);
    if (bad_mask & grp) st = SC_STATUS_BAD_OBSTACLE;
    TC u0 = S.u0, u1 = S.u1;
    if (st != SC_STATUS_OPTIMAL) { u0 = num<TC>::nan(); u1 = num<TC>::nan(); }
    if (active) {
        if (sub == 0) {
            V2 uo; uo.x = TIO(u0); uo.y = TIO(u1);
            reinterpret_cast<V2*>(u_out)[agent] = uo;
            status_out[agent] = st;
        }
        if (h_out && has_row) h_out[agent * K + sub] = used ? TIO(h) : TIO(0);
    }
}

template <typename TIO, typename TC, int G, int MODEL>
static hipError_t launch_coop(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                              const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                              hipStream_t stream) {
    constexpr int APW = 64 / G;
    const unsigned nblk = (unsigned)((B + APW * SC_COOP_WAVES - 1) / (APW * SC_COOP_WAVES));
    hipLaunchKernelGGL((cbfqp_coop_kernel<TIO, TC, G, MODEL>), dim3(nblk), dim3(64 * SC_COOP_WAVES), 0, stream, p, B, K,
                       (const TIO*)X, (const TIO*)u_ref, (const TIO*)obs, n_obs, (TIO*)u_out, status, (TIO*)h_out);
    return hipGetLastError();
}

// ======================================================================================
// K > 8: LDS kernel.
// The kernel has three phases per 64-agent wave:
//   1. stage   obstacle rows HBM -> LDS (16-byte LDS-DMA), state / u_ref -> registers
//   2. assemble one CBF row per obstacle (rolled loop, one copy of the row builder) -> LDS
//              rows[(obstacle*3 + comp)*64 + lane]  (lane-contiguous: bank-conflict free)
//   3. solve   KMAX > 0: rows are pulled into registers and the walk is fully unrolled (K <= KMAX <= 8)
//              KMAX == 0: rows stay in LDS, run-time loops (any K up to SC_CBFQP_MAX_OBS)
//
// Phase 2 reads the staged obstacle rows with a per-lane rotation: lane l handles obstacle
// (r + l / P) mod K at step r, P = 32 / gcd(K, 32).  Lanes whose staged rows start on the same
// LDS bank (stride K*7 dwords) then read different rows, which removes the 8- (K = 8) to 16-way
// (K = 16) bank conflict of the straightforward order; the QP does not care about row order.
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

    // ---- 1. obstacles: HBM -> LDS ------------------------------------------------
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

    // state / reference (coalesced, overlaps the DMA)
    using V2 = typename vec2<TIO>::type;
    TC ur0 = 0, ur1 = 0;
    int nk = K;
    if (active) {
        const V2 ur = reinterpret_cast<const V2*>(u_ref)[agent];
        ur0 = TC(ur.x); ur1 = TC(ur.y);
        if (n_obs) {
            nk = n_obs[agent];
            nk = nk < 0 ? 0 : (nk > K ? K : nk);
        }
    }
    const CbfConsts<TC> k = make_consts<TC>(p);
    const Agent<TC> ag = load_agent<TIO, TC, MODEL>(X, active ? agent : 0);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // LDS-DMA landed (it is tracked by vmcnt)
    __syncthreads();

    // ---- 2. row assembly: agent_barrier + cbf_qp.py:155-183 ------------------------
    const size_t stage_bytes = ((p.obs_shared ? (size_t)row_elems : (size_t)64 * row_elems) * sizeof(TIO) + 15) & ~(size_t)15;
    TC* rows = reinterpret_cast<TC*>(smem + stage_bytes);          // [K][3][64]
    TIO* hl = reinterpret_cast<TIO*>(rows + (size_t)K * 3 * 64);   // [K][64], natural obstacle order
    const TIO* mine = lobs + (p.obs_shared ? 0 : lane * row_elems);
    int period = 32;                                               // P = 32 / gcd(K, 32)
    while (((K * (32 / period)) & 31) != 0 && period > 1) period >>= 1;
    // (period ends as the smallest power of two with K*32/period = 0 mod 32, i.e. 32/gcd(K,32))
    const int rot = p.obs_shared ? 0 : (lane / period) % K;
    bool bad_obs = false;
    TC poison = TC(0);                                             // NaN once any row entry is non-finite
#pragma nounroll
    for (int r = 0; r < K; ++r) {
        int ro = r + rot;
        ro = ro >= K ? ro - K : ro;
        TC o[7];
#pragma unroll
        for (int f = 0; f < 7; ++f) o[f] = TC(mine[ro * 7 + f]);
        TC h, a0, a1, cc;
        const bool ok = cbf_row<TC, MODEL>(ag, o, k, a0, a1, cc, h);
        const bool used = ro < nk;
        bad_obs |= used && !ok;
        a0 = used ? a0 : TC(0); a1 = used ? a1 : TC(0); cc = used ? cc : TC(0);
        normalise_row(a0, a1, cc, poison);
        // stored at the obstacle's own index: the walk order (hence the rounding) is the same for every lane
        rows[(ro * 3 + 0) * 64 + lane] = a0;
        rows[(ro * 3 + 1) * 64 + lane] = a1;
        rows[(ro * 3 + 2) * 64 + lane] = cc;
        if constexpr (KMAX > 0) {
            hl[ro * 64 + lane] = used ? TIO(h) : TIO(0);
        } else {
            // large K: no LDS left for an h stage (f64, K = 32 uses all 160 KiB) -- store directly
            if (active && h_out) h_out[agent * K + ro] = used ? TIO(h) : TIO(0);
        }
    }

    // ---- 3. solve ---------------------------------------------------------------------
    TC u0, u1;
    int st;
    if constexpr (KMAX > 0) {
        TC n0[KMAX], n1[KMAX], c[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const bool in = j < K;
            n0[j] = in ? rows[(j * 3 + 0) * 64 + lane] : TC(0);
            n1[j] = in ? rows[(j * 3 + 1) * 64 + lane] : TC(0);
            c[j] = in ? rows[(j * 3 + 2) * 64 + lane] : TC(0);
        }
        st = qp2_solve<TC, KMAX>(n0, n1, c, K, ur0, ur1, poison, k, u0, u1);
    } else {
        st = qp2_solve_lds<TC>(rows, lane, K, ur0, ur1, poison, k, u0, u1);
    }

    if (bad_obs) st = SC_STATUS_BAD_OBSTACLE;
    if (st != SC_STATUS_OPTIMAL) { u0 = num<TC>::nan(); u1 = num<TC>::nan(); }

    // ---- outputs ----------------------------------------------------------------------
    if (active) {
        V2 uo; uo.x = TIO(u0); uo.y = TIO(u1);
        reinterpret_cast<V2*>(u_out)[agent] = uo;
        status_out[agent] = st;
        if constexpr (KMAX > 0) {
            if (h_out) {
                TIO* hp = h_out + agent * K;
                if (K == KMAX) {                 // static indices: vector stores
                    TIO hv[KMAX];
#pragma unroll
                    for (int r = 0; r < KMAX; ++r) hv[r] = hl[r * 64 + lane];
#pragma unroll
                    for (int r = 0; r < KMAX; ++r) hp[r] = hv[r];
                } else {
                    for (int r = 0; r < K; ++r) hp[r] = hl[r * 64 + lane];
                }
            }
        }
    }
}

// ---- host-side dispatch ----------------------------------------------------------
template <typename TIO, typename TC, int KMAX, int MODEL>
static hipError_t launch_one(const sc_cbfqp_params& p, long long B, int K, const void* X, const void* u_ref,
                             const void* obs, const int* n_obs, void* u_out, int* status, void* h_out,
                             hipStream_t stream) {
    if constexpr (KMAX > 0) {
        if (B <= sc_coop_max_agents())                    // latency-bound regime: 8 lanes per agent
            return launch_coop<TIO, TC, 8, MODEL>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        const unsigned threads = 256;
        const unsigned nblk = (unsigned)((B + threads - 1) / threads);
        if constexpr (MODEL == SC_MODEL_DYNAMIC_UNICYCLE2D && sizeof(TC) == 8) {
            if (sc_two_pass() && B >= (1 << 18)) {            // below, the second launch costs more than the fast pass gains (2^16: 12.2 -> 13.4 us)
                hipLaunchKernelGGL((cbfqp_reg_kernel<TIO, TC, KMAX, MODEL, 1>), dim3(nblk), dim3(threads), 0, stream, p, B, K,
                                   (const TIO*)X, (const TIO*)u_ref, (const TIO*)obs, n_obs, (TIO*)u_out, status, (TIO*)h_out);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((cbfqp_reg_kernel<TIO, TC, KMAX, MODEL, 2>), dim3(nblk), dim3(threads), 0, stream, p, B, K,
                                   (const TIO*)X, (const TIO*)u_ref, (const TIO*)obs, n_obs, (TIO*)u_out, status, (TIO*)h_out);
                return hipGetLastError();
            }
        }
        hipLaunchKernelGGL((cbfqp_reg_kernel<TIO, TC, KMAX, MODEL>), dim3(nblk), dim3(threads), 0, stream, p, B, K,
                           (const TIO*)X, (const TIO*)u_ref, (const TIO*)obs, n_obs, (TIO*)u_out, status, (TIO*)h_out);
        return hipGetLastError();
    } else {
    if (!sc_force_lds_kernel()) {                          // K > 8: one row per lane, 16 or 32 lanes per agent
        if (K <= 16) return launch_coop<TIO, TC, 16, MODEL>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        return launch_coop<TIO, TC, 32, MODEL>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
    }
    const unsigned blocks = (unsigned)((B + 63) / 64);
    size_t lds = (p.obs_shared ? (size_t)K * 7 : (size_t)64 * K * 7) * sizeof(TIO);
    lds = ((lds + 15) & ~(size_t)15) + (size_t)K * 3 * 64 * sizeof(TC) + (KMAX > 0 ? (size_t)K * 64 * sizeof(TIO) : 0);
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
        case SC_MODEL_SINGLE_INTEGRATOR2D:
            return launch_k<TIO, TC, SC_MODEL_SINGLE_INTEGRATOR2D>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        case SC_MODEL_DOUBLE_INTEGRATOR2D:
            return launch_k<TIO, TC, SC_MODEL_DOUBLE_INTEGRATOR2D>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        case SC_MODEL_QUAD2D:
            return launch_k<TIO, TC, SC_MODEL_QUAD2D>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        case SC_MODEL_UNICYCLE2D:
            return launch_k<TIO, TC, SC_MODEL_UNICYCLE2D>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
        default:
            return launch_k<TIO, TC, SC_MODEL_KINEMATIC_BICYCLE2D_DPCBF>(p, B, K, X, u_ref, obs, n_obs, u_out, status, h_out, stream);
    }
}

}  // namespace sc
