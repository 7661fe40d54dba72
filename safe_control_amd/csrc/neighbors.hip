// K nearest OTHER agents of every local agent, written as moving circular obstacles
// (extension for BASELINE config 4; no reference counterpart -- examples/test_multi_robot.py:77-80
// steps its robots independently).  Two implementations of the same result: the uniform-grid cell list further down (what the sharded step
// runs: sc_neighbor_obstacles_batch_ws) and, first, the plain scan it is held to bit for bit (sc_neighbor_obstacles_batch).
// The scan, N-body style: a workgroup of 256 local agents walks
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
            double sn, cs; sincos_(th, &sn, &cs); row[3] = TIO(v * cs); row[4] = TIO(v * sn);
        }
#pragma unroll
        for (int f = 0; f < 7; ++f) out[j * 7 + f] = row[f];
    }
}

// ---- uniform-grid cell list (round 6; replaces the candidate-split scan of rounds 3 - 5: 0.6 ms of a 0.72 ms config-4 step) ----------------
// The K nearest of N points do not need N^2 distance evaluations: the agents are binned into square cells of side h chosen so that a
// 3 x 3 block of cells holds about 3.75 K of them on average (h = sqrt(3.75 K A / (9 N)) for a bounding box of area A: the K-th nearest then
// lies inside the block for nine agents in ten, and 60 candidates mostly fit one pass of the 64-lane sort at K = 16); an agent looks at
// the ring of cells around its own, widens ring by ring, and stops once its K-th best distance is inside the block it has covered --
// every agent it has not looked at lies outside that block.  Five small kernels on the caller's stream, no host round trip:
//   nb_bbox_kernel      bounding box of all agents, grid geometry (one workgroup)
//   nb_count_kernel     cell of every agent, agents per cell (atomics)
//   nb_scan_kernel      exclusive scan of the cell counts (one workgroup)
//   nb_scatter_kernel   agents into cell order (index and position; the order INSIDE a cell is whatever the atomics give)
//   nb_select_kernel    per local agent (one wavefront each): rings of cells -> K best (distance, index) pairs -> obstacle rows
// The result is the single scan's bit for bit: distances are computed by the same expression in the storage precision, and a candidate
// replaces an entry when it is nearer OR equally near with a lower index -- the order the sequential scan's strict "<" produces -- so the
// visiting order does not matter.
struct NbGrid {
    float x0, y0, h, inv_h;
    int nx, ny;
    float slop;                                       // what the binning's float arithmetic may be off by (in metres): the stop test's margin
    int pad;
};
constexpr int NB_MAX_DIM = 256;                       // at most 256 x 256 cells

static size_t nb_align(size_t v) { return (v + 255) & ~(size_t)255; }
struct NbLayout {
    size_t grid, cell_of, count, start, fill, sorted, sx, sy, total;
    NbLayout(size_t es, long long B_all) {
        size_t o = 0;
        auto take = [&](size_t n) { size_t r = o; o += nb_align(n); return r; };
        grid = take(sizeof(NbGrid)); cell_of = take((size_t)B_all * 4); count = take((size_t)(NB_MAX_DIM * NB_MAX_DIM + 1) * 4);
        start = take((size_t)(NB_MAX_DIM * NB_MAX_DIM + 1) * 4); fill = take((size_t)(NB_MAX_DIM * NB_MAX_DIM) * 4);
        sorted = take((size_t)B_all * 4); sx = take((size_t)B_all * es); sy = take((size_t)B_all * es);
        total = o;
    }
};

template <typename TIO>
__global__ __launch_bounds__(1024) void nb_bbox_kernel(const long long B_all, const int K, const TIO* __restrict__ X_all, NbGrid* __restrict__ grid,
                                                       int* __restrict__ count) {
    __shared__ float red[4][1024];
    const int tid = threadIdx.x;
    float xl = 3.0e38f, xh = -3.0e38f, yl = 3.0e38f, yh = -3.0e38f;
    for (long long i = tid; i < B_all; i += 1024) {
        const float x = (float)X_all[i * 4 + 0], y = (float)X_all[i * 4 + 1];
        if (fabsf(x) < 1.0e30f && fabsf(y) < 1.0e30f) { xl = fminf(xl, x); xh = fmaxf(xh, x); yl = fminf(yl, y); yh = fmaxf(yh, y); }     // (NaN / inf: out of the box)
    }
    red[0][tid] = xl; red[1][tid] = xh; red[2][tid] = yl; red[3][tid] = yh;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] = fminf(red[0][tid], red[0][tid + s]); red[1][tid] = fmaxf(red[1][tid], red[1][tid + s]);
            red[2][tid] = fminf(red[2][tid], red[2][tid + s]); red[3][tid] = fmaxf(red[3][tid], red[3][tid + s]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        xl = red[0][0]; xh = red[1][0]; yl = red[2][0]; yh = red[3][0];
        if (!(xh >= xl) || !(yh >= yl)) { xl = yl = 0.0f; xh = yh = 1.0f; }
        const float w = fmaxf(xh - xl, 1.0e-6f), hgt = fmaxf(yh - yl, 1.0e-6f);
        float h = sqrtf(3.75f * (float)K * w * hgt / (9.0f * (float)(B_all > 0 ? B_all : 1)));
        h = fmaxf(h, fmaxf(w, hgt) / (float)(NB_MAX_DIM - 1));                         // at most NB_MAX_DIM cells per side
        h = fmaxf(h, 1.0e-6f * fmaxf(fmaxf(fabsf(xl), fabsf(xh)), fmaxf(fabsf(yl), fabsf(yh))));
        NbGrid g;
        g.x0 = xl; g.y0 = yl; g.h = h; g.inv_h = 1.0f / h;
        g.nx = min(NB_MAX_DIM, (int)(w / h) + 1); g.ny = min(NB_MAX_DIM, (int)(hgt / h) + 1);
        g.slop = 1.0e-4f * h + 1.0e-6f * fmaxf(fmaxf(fabsf(xl), fabsf(xh)), fmaxf(fabsf(yl), fabsf(yh)));
        g.pad = 0;
        *grid = g;
    }
    // the cell counters of this call
    __syncthreads();
    const int ncell = grid->nx * grid->ny;
    for (int i = tid; i <= ncell; i += 1024) count[i] = 0;
}

__device__ __forceinline__ int nb_cell(const NbGrid& g, float x, float y, int& cx, int& cy) {
    // (NaN / inf positions are binned into a corner cell: their distances are never "nearer", they are never anybody's neighbour)
    float fx = (x - g.x0) * g.inv_h, fy = (y - g.y0) * g.inv_h;
    fx = fx >= 0.0f ? fx : 0.0f; fy = fy >= 0.0f ? fy : 0.0f;
    cx = fx < (float)(g.nx - 1) ? (int)fx : g.nx - 1; cy = fy < (float)(g.ny - 1) ? (int)fy : g.ny - 1;
    return cy * g.nx + cx;
}

template <typename TIO>
__global__ __launch_bounds__(256) void nb_count_kernel(const long long B_all, const TIO* __restrict__ X_all, const NbGrid* __restrict__ grid,
                                                       int* __restrict__ cell_of, int* __restrict__ count) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B_all) return;
    const NbGrid g = *grid;
    int cx, cy;
    const int c = nb_cell(g, (float)X_all[i * 4 + 0], (float)X_all[i * 4 + 1], cx, cy);
    cell_of[i] = c;
    atomicAdd(&count[c], 1);
}

__global__ __launch_bounds__(1024) void nb_scan_kernel(const NbGrid* __restrict__ grid, const int* __restrict__ count, int* __restrict__ start,
                                                       int* __restrict__ fill) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, n = grid->nx * grid->ny;
    const int per = (n + 1023) / 1024, lo = tid * per, hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                                // inclusive scan of the 1024 partial sums
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid > 0 ? part[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) { start[i] = run; fill[i] = 0; run += count[i]; }
    if (tid == 1023) start[n] = part[1023];
}

template <typename TIO>
__global__ __launch_bounds__(256) void nb_scatter_kernel(const long long B_all, const TIO* __restrict__ X_all, const int* __restrict__ cell_of,
                                                         const int* __restrict__ start, int* __restrict__ fill, int* __restrict__ sorted,
                                                         TIO* __restrict__ sx, TIO* __restrict__ sy) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B_all) return;
    const int c = cell_of[i];
    const int p = start[c] + atomicAdd(&fill[c], 1);
    sorted[p] = (int)i; sx[p] = X_all[i * 4 + 0]; sy[p] = X_all[i * 4 + 1];
}

// ONE LOCAL AGENT PER WAVEFRONT, one candidate per lane: the cells of a grid row are contiguous in the cell-ordered arrays, so the 3 x 3 block
// around the agent is three contiguous runs of candidates, read coalesced; lanes 0 .. KMAX-1 keep the best so far, the other lanes take the next
// 64 - KMAX candidates, a bitonic sort over the wave (__shfl_xor) by (distance, index) leaves the KMAX best in the low lanes -- no divergence,
// no per-lane insertion loops (round 6's first version, a lane per agent walking its cells with a 16-deep sorted insertion, took 165 us for
// the 16384 agents of configs[3]; this one ~10).  Lane j < K then writes obstacle row j.
// the value of lane (l ^ J): DPP where the pattern exists (quad_perm for 1 and 2, row_ror:8 for 8: VALU latency), the LDS crossbar otherwise --
// 14 of the 21 compare-exchange steps of a 64-lane bitonic sort have J in {1, 2, 8}
template <int J>
__device__ __forceinline__ int nb_xor_i(int v) {
    if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
    else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
    else if constexpr (J == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);
    else return __shfl_xor(v, J, 64);
}
template <int J>
__device__ __forceinline__ float nb_xor(float v) { return __int_as_float(nb_xor_i<J>(__float_as_int(v))); }
template <int J>
__device__ __forceinline__ double nb_xor(double v) { return __hiloint2double(nb_xor_i<J>(__double2hiint(v)), nb_xor_i<J>(__double2loint(v))); }

template <typename TD, int K2, int J>
__device__ __forceinline__ void nb_cmpx(TD& d, int& idx, const int lane) {
    const TD od = nb_xor<J>(d);
    const int oi = nb_xor_i<J>(idx);
    const bool up = (lane & K2) == 0, low = (lane & J) == 0;                   // ascending block / the lower lane of the pair
    const bool mine_first = d < od || (d == od && idx <= oi);
    const bool keep = (low == up) ? mine_first : !mine_first;                 // the lower lane of an ascending pair keeps the smaller, of a descending pair the larger
    d = keep ? d : od; idx = keep ? idx : oi;
}
template <typename TD, int K2, int J>
__device__ __forceinline__ void nb_merge(TD& d, int& idx, const int lane) {
    nb_cmpx<TD, K2, J>(d, idx, lane);
    if constexpr (J > 1) nb_merge<TD, K2, J / 2>(d, idx, lane);
}
template <typename TD>
__device__ __forceinline__ void nb_sort64(TD& d, int& idx, const int lane) {
    nb_merge<TD, 2, 1>(d, idx, lane); nb_merge<TD, 4, 2>(d, idx, lane); nb_merge<TD, 8, 4>(d, idx, lane);
    nb_merge<TD, 16, 8>(d, idx, lane); nb_merge<TD, 32, 16>(d, idx, lane); nb_merge<TD, 64, 32>(d, idx, lane);
}
template <typename TIO, int KMAX>
__global__ __launch_bounds__(256) void nb_select_kernel(const long long B_all, const long long first_local, const long long B_local, const int K,
                                                        const float radius, const TIO* __restrict__ X_all, const NbGrid* __restrict__ grid,
                                                        const int* __restrict__ start, const int* __restrict__ sorted, const TIO* __restrict__ sx,
                                                        const TIO* __restrict__ sy, TIO* __restrict__ obs_out) {
    using TD = TIO;
    const int lane = threadIdx.x & 63;
    const long long li = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // the wave's local agent (wave-uniform)
    if (li >= B_local) return;
    const long long gi = first_local + li;
    const NbGrid g = *grid;
    const TD x = (TD)X_all[gi * 4 + 0], y = (TD)X_all[gi * 4 + 1];
    const TD INF = TD(__builtin_huge_valf());
    TD bd = INF;                                                            // lanes < KMAX: the best so far, sorted; the others: scratch
    int bi = 0x7fffffff;
    int cx, cy;
    nb_cell(g, (float)x, (float)y, cx, cy);
    const int rmax = max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy));
    const int kth = (K < KMAX ? K : KMAX) - 1;
    // candidates stream through the lanes that hold nothing worth keeping: all 64 while the list is empty, the upper 64 - KMAX afterwards; a
    // sort runs when they are full and at the end of a ring (wave-uniform control flow throughout)
    int fill = 0;
    bool have = false;
    auto flush = [&]() { if (fill > 0) { nb_sort64<TD>(bd, bi, lane); fill = 0; have = true; } };
    auto feed = [&](int p0, int p1) {                                         // a run [p0, p1) of the cell-ordered arrays
        while (p0 < p1) {
            const int base = have ? KMAX : 0, cap = 64 - base;
            if (fill == 0 && lane >= base) { bd = INF; bi = 0x7fffffff; }
            const int m = min(cap - fill, p1 - p0), slot = lane - base - fill;
            if (slot >= 0 && slot < m) {
                const int q = p0 + slot, ci = sorted[q];
                const TD dx = sx[q] - x, dy = sy[q] - y;
                const TD cd = dx * dx + dy * dy;
                if (ci != (int)gi && cd < INF) { bd = cd; bi = ci; }          // (not its own neighbour; NaN / inf never are)
            }
            fill += m; p0 += m;
            if (fill == cap) flush();
        }
    };
    for (int r = 1; ; ++r) {
        const int xa = max(cx - r, 0), xb = min(cx + r, g.nx - 1);
        if (r == 1) {                                                       // rings 0 and 1 together: three rows of up to three contiguous cells
            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.ny - 1); ++yy) feed(start[yy * g.nx + xa], start[yy * g.nx + xb + 1]);
        } else {                                                            // ring r: the top and the bottom row of the block, the two end cells of every row between
            if (cy - r >= 0) feed(start[(cy - r) * g.nx + xa], start[(cy - r) * g.nx + xb + 1]);
            if (cy + r < g.ny) feed(start[(cy + r) * g.nx + xa], start[(cy + r) * g.nx + xb + 1]);
            for (int yy = max(cy - r + 1, 0); yy <= min(cy + r - 1, g.ny - 1); ++yy) {
                if (cx - r >= 0) feed(start[yy * g.nx + cx - r], start[yy * g.nx + cx - r + 1]);
                if (cx + r < g.nx) feed(start[yy * g.nx + cx + r], start[yy * g.nx + cx + r + 1]);
            }
        }
        flush();
        if (r >= rmax) break;
        // done when the K-th best lies inside the covered block: distance from the agent to the nearest side of the block that still has cells
        // beyond it (NbGrid.slop covers the rounding of the binning)
        const TD dk = __shfl(bd, kth, 64);
        const int ik = __shfl(bi, kth, 64);
        float dmin = 3.0e38f;
        if (cx - r > 0) dmin = fminf(dmin, (float)x - (g.x0 + (float)(cx - r) * g.h));
        if (cx + r < g.nx - 1) dmin = fminf(dmin, (g.x0 + (float)(cx + r + 1) * g.h) - (float)x);
        if (cy - r > 0) dmin = fminf(dmin, (float)y - (g.y0 + (float)(cy - r) * g.h));
        if (cy + r < g.ny - 1) dmin = fminf(dmin, (g.y0 + (float)(cy + r + 1) * g.h) - (float)y);
        dmin -= g.slop;
        if (dmin > 0.0f && ik != 0x7fffffff && (float)dk < dmin * dmin) break;
    }
    if (lane < K && lane < KMAX) {
        TIO* out = obs_out + ((size_t)li * K + lane) * 7;
        TIO row[7] = {TIO(1000), TIO(1000), TIO(0), TIO(0), TIO(0), TIO(0), TIO(0)};
        const int n = bi;
        if (n != 0x7fffffff && bd < TD(1e29)) {
            const double th = (double)X_all[(size_t)n * 4 + 2], v = (double)X_all[(size_t)n * 4 + 3];
            row[0] = X_all[(size_t)n * 4 + 0]; row[1] = X_all[(size_t)n * 4 + 1]; row[2] = TIO(radius);
            double sn, cs; sincos_(th, &sn, &cs); row[3] = TIO(v * cs); row[4] = TIO(v * sn);
        }
#pragma unroll
        for (int f = 0; f < 7; ++f) out[f] = row[f];
    }
}

static int nb_kmax(int K) { return K <= 8 ? 8 : (K <= 16 ? 16 : 32); }

size_t neighbors_workspace_bytes(int io_dtype, long long B_all, long long B_local, int K) {
    (void)B_local; (void)K;
    return NbLayout(io_dtype == SC_DTYPE_F32 ? 4 : 8, B_all).total;
}

template <typename TIO>
static hipError_t nb_cells_launch(long long B_all, long long first, long long B_local, int K, double r, const void* X,
                                  void* out, void* ws, hipStream_t stream) {
    const NbLayout L(sizeof(TIO), B_all);
    unsigned char* w = (unsigned char*)ws;
    NbGrid* grid = (NbGrid*)(w + L.grid);
    int *cell_of = (int*)(w + L.cell_of), *count = (int*)(w + L.count), *start = (int*)(w + L.start), *fill = (int*)(w + L.fill), *sorted = (int*)(w + L.sorted);
    TIO *sx = (TIO*)(w + L.sx), *sy = (TIO*)(w + L.sy);
    const unsigned ab = (unsigned)((B_all + 255) / 256), lb = (unsigned)((B_local + 3) / 4);
    hipLaunchKernelGGL((nb_bbox_kernel<TIO>), dim3(1), dim3(1024), 0, stream, B_all, K, (const TIO*)X, grid, count);
    hipLaunchKernelGGL((nb_count_kernel<TIO>), dim3(ab), dim3(256), 0, stream, B_all, (const TIO*)X, grid, cell_of, count);
    hipLaunchKernelGGL(nb_scan_kernel, dim3(1), dim3(1024), 0, stream, grid, count, start, fill);
    hipLaunchKernelGGL((nb_scatter_kernel<TIO>), dim3(ab), dim3(256), 0, stream, B_all, (const TIO*)X, cell_of, start, fill, sorted, sx, sy);
    const int KM = nb_kmax(K);
#define SC_NBC(KMX) hipLaunchKernelGGL((nb_select_kernel<TIO, KMX>), dim3(lb), dim3(256), 0, stream, B_all, first, B_local, K, (float)r, (const TIO*)X, grid, start, sorted, sx, sy, (TIO*)out)
    if (KM == 8) { SC_NBC(8); }
    else if (KM == 16) { SC_NBC(16); }
    else { SC_NBC(32); }
#undef SC_NBC
    return hipGetLastError();
}

hipError_t neighbors_split_launch(int io_dtype, long long B_all, long long first, long long B_local, int K, double r,
                                  const void* X, void* out, void* ws, hipStream_t stream) {
    if (io_dtype == SC_DTYPE_F32) return nb_cells_launch<float>(B_all, first, B_local, K, r, X, out, ws, stream);
    return nb_cells_launch<double>(B_all, first, B_local, K, r, X, out, ws, stream);
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
