// Debugging aid, never shipped and never on the product path: compiles safe_control_amd/csrc/mpc_du_ms_solver.hpp -- the code the 64 lanes of a
// wavefront run in csrc/mpc_du_ms.hip -- for the host, ONE THREAD PER LANE with a barrier where the kernel has __syncthreads and reductions
// through a shared array, so that the kernel's algorithm can be stepped against oracle/ms_ipopt.py in a container without a GPU.
//   g++ -O2 -std=c++17 -shared -fPIC -pthread tools/du_ms_host.cpp -o /tmp/libdu_ms_host.so   (tools/dbg_du_ms_host.py)
#include <pthread.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define SC_HD
#include "../safe_control_amd/csrc/mpc_du_ms_solver.hpp"

namespace {
struct Shared {
    pthread_barrier_t bar;
    std::vector<double> lds;
    double red[64];
    double gred[64][48];
};
struct HostCtx {
    typedef double* ptr;
    double* lds;
    int lane;
    Shared* sh;
    void sync() { pthread_barrier_wait(&sh->bar); }
    long long clock() const { return 0; }
    void sincos(double a, double& s, double& c) const { s = std::sin(a); c = std::cos(a); }
    void pow2(double x1, double e1, double x2, double e2, double& r1, double& r2) const { r1 = std::pow(x1, e1); r2 = std::pow(x2, e2); }
    double rsqrt(double v) const { return 1.0 / std::sqrt(v); }
    template <typename F>
    double reduce(double v, F f) {
        pthread_barrier_wait(&sh->bar);
        sh->red[lane] = v;
        pthread_barrier_wait(&sh->bar);
        double a = sh->red[0];
        for (int i = 1; i < 64; ++i) a = f(a, sh->red[i]);
        pthread_barrier_wait(&sh->bar);
        return a;
    }
    // sum over the G lanes of a group (lanes l / G equal), every lane of the wave calls it
    template <int n>
    void gsum(double* v, int G) {
        if (G == 1) return;
        pthread_barrier_wait(&sh->bar);
        for (int i = 0; i < n; ++i) sh->gred[lane][i] = v[i];
        pthread_barrier_wait(&sh->bar);
        const int base = lane / G * G;
        for (int i = 0; i < n; ++i) { double a = 0.0; for (int l = 0; l < G; ++l) a += sh->gred[base + l][i]; v[i] = a; }
        pthread_barrier_wait(&sh->bar);
    }
    double wsum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
    double wmax(double v) { return reduce(v, [](double a, double b) { return std::fmax(a, b); }); }
    double wmin(double v) { return reduce(v, [](double a, double b) { return std::fmin(a, b); }); }
};

void lane_main(int lane, Shared* sh, const sc::dums::Params* P, const sc_ipopt_params* O, const double* x0, const double* up, const double* goal,
               const double* obs, double* u_out, double* plan, double* trace, int* status, int* iters) {
    using namespace sc::dums;
    HostCtx cx{sh->lds.data(), lane, sh};
    Wave<HostCtx> S(cx, *P, *O);
    if (lane < 3 * P->K) { const int j = lane / 3, c = lane % 3; sh->lds[S.L.OB + lane] = j < P->K ? obs[7 * j + c] : 0.0; }
    for (int i = 0; i < NX; ++i) S.x0[i] = x0[i];
    for (int j = 0; j < NU; ++j) S.uprev[j] = up[j];
    S.xg[0] = goal[0]; S.xg[1] = goal[1];
    cx.sync();
    int st, it;
    S.solve(st, it, trace);
    if (lane == 0) { u_out[0] = S.u[0]; u_out[1] = S.u[1]; *status = st; *iters = it; if (getenv("DUMS_DEBUG")) fprintf(stderr, "filt_over %d nfilt %d rs %d\n", (int)S.filt_over, S.nfilt, (int)S.rs); }
    if (plan && S.acl) {
        for (int i = 0; i < NX; ++i) plan[S.k * NX + i] = S.x[i];
        if (S.stg) for (int j = 0; j < NU; ++j) plan[(P->N + 1) * NX + S.k * NU + j] = S.u[j];
    }
}
}  // namespace

extern "C" int du_ms_host_lds_layout(int N, int K, int* out) {
    const sc::dums::Lds L(N, K);
    const int v[] = {L.OB, L.AB, L.H, L.G, L.C, L.KG, L.PX, L.LAM, L.XS, L.US, L.YS, L.Pa, L.Pb, L.T, L.QU, L.FP, L.FT, L.FP2, L.FT2, L.SC, L.Y0, L.RW, L.XR, L.total};
    for (int i = 0; i < 24; ++i) out[i] = v[i];
    return 24;
}

extern "C" int du_ms_host_solve(const sc_mpccbf_params* prm, const sc_ipopt_params* O, int K, const double* x0, const double* u_prev, const double* goal,
                                const double* obs, double* u_out, double* plan, double* trace, int* status, int* iters, double* lds_out) {
    using namespace sc::dums;
    Params P;
    P.N = prm->horizon; P.K = K; P.dt = prm->dt;
    for (int i = 0; i < 4; ++i) P.Q[i] = prm->Q[i];
    for (int j = 0; j < 2; ++j) { P.R[j] = prm->R[j]; P.u_lo[j] = -prm->u_max[j]; P.u_hi[j] = prm->u_max[j]; }
    P.alpha1 = prm->alpha1; P.alpha2 = prm->alpha2; P.beta = prm->beta; P.radius = prm->robot_radius; P.v_max = prm->v_max;
    Shared sh;
    pthread_barrier_init(&sh.bar, nullptr, 64);
    sh.lds.assign(Lds(P.N, P.K).total, 0.0);
    std::vector<std::thread> th;
    for (int l = 0; l < 64; ++l) {
        th.emplace_back(lane_main, l, &sh, &P, O, x0, u_prev, goal, obs, u_out, plan, trace, status, iters);
    }
    for (auto& t : th) t.join();
    if (lds_out) for (size_t i = 0; i < sh.lds.size(); ++i) lds_out[i] = sh.lds[i];
    pthread_barrier_destroy(&sh.bar);
    return 0;
}
