// TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): the compiled, multi-core CPU baseline of BASELINE configs[2] that
// bench.py times beside kernel 13 (`cpu_baseline`, kind "port"), and a second check of oracle/ms_ipopt.py.
//
// What it is: the multiple-shooting MPC-CBF solve of position_control/mpc_cbf.py:162-174,366-402 (do-mpc -> IPOPT) in the form
// oracle/ms_ipopt.py states it (StageNLP with du_model(), KERNEL_PROFILE), compiled for the host from the SAME solver header the HIP kernel
// instantiates (safe_control_amd/csrc/mpc_du_ms_solver.hpp: plain C++ over a context).  The 64 lanes of a wavefront run as 64 cooperative
// fibers of ONE thread (a barrier is a yield around the ring, a wave reduction a pass over a small array); problems are spread over the
// host cores with OpenMP.  It is NOT an independent restatement -- parity claims rest on the numpy oracle, which tests/test_oracle_c.py
// holds this build to -- and nothing on the product path loads it.
//
//   g++ -O2 -fPIC -shared -fopenmp -o oracle/_build/libdu_ms_cpu.so oracle/c/mpc_du_ms_cpu.cpp      (oracle/Makefile)
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SC_HD
#include "../../safe_control_amd/csrc/mpc_du_ms_solver.hpp"

#if !defined(__x86_64__)
#error "the fiber switch below is written for x86-64 (System V): the image's host architecture"
#endif

// ---- a minimal cooperative context switch: callee-saved registers and the stack pointer ------------------------------------------------
extern "C" void sc_fiber_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl sc_fiber_switch
.type sc_fiber_switch,@function
sc_fiber_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size sc_fiber_switch,.-sc_fiber_switch
)");

namespace {

constexpr int WAVE = 64, STACK = 48 * 1024;

struct Problem {
    const sc::dums::Params* P;
    const sc_ipopt_params* O;
    const double *x0, *up, *goal, *obs;
    double* u_out;
    int *status, *iters;
    int model, se;
    double* trace;                                       // [max_iter + 1][8] or NULL (du_ms_cpu_solve_batch with trace_out: B = 1)
};

struct Wavefront {
    std::vector<double> lds;
    double red[WAVE];
    double gred[WAVE][48];
    void* sp[WAVE];
    void* main_sp = nullptr;
    bool done[WAVE];
    int cur = 0, arrived = 0;
    unsigned gen = 0;
    std::vector<unsigned char> stacks;
    Problem prob;
    Wavefront() : stacks((size_t)WAVE * STACK + 64) {}
};

thread_local Wavefront* tl_wave = nullptr;

void fiber_yield(Wavefront* w) {                       // to the next lane that is still running (or back to the caller when none is)
    const int me = w->cur;
    int nx = me;
    for (int s = 1; s <= WAVE; ++s) { const int c = (me + s) % WAVE; if (!w->done[c]) { nx = c; break; } }
    if (nx == me && !w->done[me]) return;
    if (w->done[me] && nx == me) { void* dummy; sc_fiber_switch(&dummy, w->main_sp); return; }
    w->cur = nx;
    sc_fiber_switch(&w->sp[me], w->sp[nx]);
}

struct FiberCtx {
    typedef double* ptr;
    double* lds;
    int lane;
    Wavefront* w;
    void sync() {
        const unsigned g = w->gen;
        if (++w->arrived == WAVE) { w->arrived = 0; ++w->gen; return; }
        while (w->gen == g) fiber_yield(w);
    }
    long long clock() const { return 0; }
    double rsqrt(double v) const { return 1.0 / std::sqrt(v); }
    void sincos(double a, double& s, double& c) const { s = std::sin(a); c = std::cos(a); }
    void pow2(double x1, double e1, double x2, double e2, double& r1, double& r2) const { r1 = std::pow(x1, e1); r2 = std::pow(x2, e2); }
    template <int n>
    void gsum(double* v, int G) {
        if (G == 1) return;
        sync();
        for (int i = 0; i < n; ++i) w->gred[lane][i] = v[i];
        sync();
        const int base = lane / G * G;
        for (int i = 0; i < n; ++i) { double a = 0.0; for (int l = 0; l < G; ++l) a += w->gred[base + l][i]; v[i] = a; }
        sync();
    }
    template <typename F>
    double reduce(double v, F f) {
        sync();
        w->red[lane] = v;
        sync();
        double a = w->red[0];
        for (int i = 1; i < WAVE; ++i) a = f(a, w->red[i]);
        sync();
        return a;
    }
    double wsum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
    double wmax(double v) { return reduce(v, [](double a, double b) { return std::fmax(a, b); }); }
    double wmin(double v) { return reduce(v, [](double a, double b) { return std::fmin(a, b); }); }
    double pow(double x, double y) const { return std::pow(x, y); }
};

template <int MODEL, bool SE = false>
void lane_body_m(Wavefront* w, int lane) {
    using namespace sc::dums;
    const Problem& q = w->prob;
    FiberCtx cx{w->lds.data(), lane, w};
    Wave<FiberCtx, MODEL, SE> S(cx, *q.P, *q.O);
    constexpr int U0 = MODEL == M_DI ? 1 : 0;            // (M_DI holds its inputs swapped: mpc_du_ms_solver.hpp)
    if constexpr (SE) {
        if (lane < q.P->K) pack_obstacle(q.obs + 7 * lane, q.P->radius, q.P->beta, [](double a, double b) { return std::pow(a, b); }, &w->lds[S.L.OB + 8 * lane]);
    } else if (lane < 3 * q.P->K) { const int j = lane / 3, c = lane % 3; w->lds[S.L.OB + lane] = q.obs[7 * j + c]; }
    for (int i = 0; i < NX; ++i) S.x0[i] = ((MODEL == M_UNI && i == 3) || (MODEL == M_SI && i >= 2)) ? 0.0 : q.x0[i];
    for (int j = 0; j < NU; ++j) S.uprev[j] = q.up[j ^ U0];
    S.xg[0] = q.goal[0]; S.xg[1] = q.goal[1];
    cx.sync();
    int st, it;
    S.solve(st, it, q.trace);
    if (lane == 0) { q.u_out[0 ^ U0] = S.u[0]; q.u_out[1 ^ U0] = S.u[1]; *q.status = st; *q.iters = it; }
}
void lane_body(Wavefront* w, int lane) {
    if (w->prob.se) {                                    // rows that may be superellipsoids (the three robots whose barrier has the branch)
        if (w->prob.model == sc::dums::M_DI) lane_body_m<sc::dums::M_DI, true>(w, lane);
        else if (w->prob.model == sc::dums::M_SI) lane_body_m<sc::dums::M_SI, true>(w, lane);
        else lane_body_m<sc::dums::M_DU, true>(w, lane);
        return;
    }
    if (w->prob.model == sc::dums::M_DI) lane_body_m<sc::dums::M_DI>(w, lane);
    else if (w->prob.model == sc::dums::M_KB) lane_body_m<sc::dums::M_KB>(w, lane);
    else if (w->prob.model == sc::dums::M_UNI) lane_body_m<sc::dums::M_UNI>(w, lane);
    else if (w->prob.model == sc::dums::M_SI) lane_body_m<sc::dums::M_SI>(w, lane);
    else lane_body_m<sc::dums::M_DU>(w, lane);
}

extern "C" void sc_fiber_entry() {
    Wavefront* w = tl_wave;
    const int lane = w->cur;
    lane_body(w, lane);
    w->done[lane] = true;
    for (;;) fiber_yield(w);                             // never returns: the last lane to finish switches back to the caller
}

void run_wave(Wavefront* w) {
    tl_wave = w;
    unsigned char* base = (unsigned char*)(((uintptr_t)w->stacks.data() + 63) & ~(uintptr_t)63);
    for (int l = 0; l < WAVE; ++l) {
        w->done[l] = false;
        void** top = (void**)(base + (size_t)(l + 1) * STACK);          // 64-byte aligned
        *--top = nullptr;                                               // (fake return address of the entry function: rsp = 8 mod 16 at its first instruction)
        *--top = (void*)&sc_fiber_entry;                                // popped by `ret`
        for (int r = 0; r < 6; ++r) *--top = nullptr;                   // rbp rbx r12 r13 r14 r15
        w->sp[l] = top;
    }
    w->cur = 0; w->arrived = 0; w->gen = 0;
    sc_fiber_switch(&w->main_sp, w->sp[0]);
}

}  // namespace

static double* g_trace = nullptr;
// the iteration trace of the next B = 1 solve (rows of 8: E_0, dual / primal infeasibility, complementarity, mu, theta, delta_w, step length; a negative
// step length marks an iterate of the restoration phase); NULL switches it off
extern "C" void du_ms_cpu_set_trace(double* trace) { g_trace = trace; }

// X [B,4], u_prev [B,2], goal [B,2], obs [B,K,7] (or [K,7] when obs_shared), float64; returns 0.  n_threads <= 0: every core.
extern "C" int du_ms_cpu_solve_batch(const sc_mpccbf_params* prm, const sc_ipopt_params* O, long B, int K, const double* X, const double* u_prev,
                                     const double* goal, const double* obs, double* u_out, int* status, int* iters, int n_threads) {
    using namespace sc::dums;
    if (!prm || !O || K < 1 || K > 16 || prm->horizon < 1 || prm->horizon > 62) return 1;
    Params P;
    P.N = prm->horizon; P.K = K; P.dt = prm->dt;
    for (int i = 0; i < 4; ++i) P.Q[i] = prm->Q[i];
    for (int j = 0; j < 2; ++j) { P.R[j] = prm->R[j]; P.u_lo[j] = -prm->u_max[j]; P.u_hi[j] = prm->u_max[j]; }
    P.alpha1 = prm->alpha1; P.alpha2 = prm->alpha2; P.beta = prm->beta; P.radius = prm->robot_radius; P.v_max = prm->v_max;
    const int model = prm->model_id == SC_MODEL_DOUBLE_INTEGRATOR2D ? M_DI : (prm->model_id == SC_MODEL_KINEMATIC_BICYCLE2D ? M_KB : (prm->model_id == SC_MODEL_UNICYCLE2D ? M_UNI : (prm->model_id == SC_MODEL_SINGLE_INTEGRATOR2D ? M_SI : M_DU)));
    if (model == M_UNI) P.Q[3] = 0.0;
    if (prm->model_id == SC_MODEL_SINGLE_INTEGRATOR2D) { P.Q[2] = 0.0; P.Q[3] = 0.0; }
    if (model == M_KB) { P.v_min = prm->v_min; P.inv_Lr = 1.0 / prm->rear_ax_dist; }
    if (model == M_DI) for (int j = 0; j < 2; ++j) { P.R[j] = prm->R[1 - j]; P.u_lo[j] = -prm->u_max[1 - j]; P.u_hi[j] = prm->u_max[1 - j]; }
    const size_t nl = (size_t)Lds(P.N, P.K, general_layout(model), prm->superellipsoid_rows != 0).total;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
#pragma omp parallel num_threads(n_threads)
    {
        Wavefront* w = new Wavefront();
        w->lds.assign(nl, 0.0);
#pragma omp for schedule(dynamic, 4)
        for (long b = 0; b < B; ++b) {
            std::fill(w->lds.begin(), w->lds.end(), 0.0);
            w->prob = Problem{&P, O, X + 4 * b, u_prev + 2 * b, goal + 2 * b, obs + (prm->obs_shared ? 0 : (size_t)b * K * 7), u_out + 2 * b, status + b, iters + b, model, prm->superellipsoid_rows != 0, B == 1 ? g_trace : nullptr};
            run_wave(w);
        }
        delete w;
    }
    return 0;
}

extern "C" int du_ms_cpu_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
