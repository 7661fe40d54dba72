// Debugging aid, never shipped and never on the product path: compiles safe_control_amd/csrc/mpc_vtol_solver.hpp -- the code each GPU
// lane runs -- for the host, so that the lane-per-problem VTOL2D solver can be stepped against oracle/mpc_vtol.py in a container
// without a GPU.   g++ -O2 -std=c++17 -shared -fPIC tools/vtol_host.cpp -o /tmp/libvtol_host.so   (tools/dbg_vtol_host.py)
#include <vector>
#include "../include/safe_control_amd.h"
#define SC_HD
#define SC_VTOL_WITH_C_PARAMS
#include "../safe_control_amd/csrc/mpc_vtol_solver.hpp"

namespace {
struct HostMem {
    double* p;
    double& operator()(int i) const { return p[i]; }
};
struct HostObs {
    const double* o;
    double operator()(int j, int c) const { return o[7 * j + c]; }
};
}  // namespace

extern "C" int vtol_host_solve(const sc_mpcvtol_params* prm, int K, const double* x0, const double* u_prev, const double* goal,
                               const double* obs, double* u_out, double* z_out, int* status, int* iters) {
    using namespace sc::vtol;
    Params P = from_c(*prm, K);
    Layout L(P.N, P.K);
    std::vector<double> ws(L.total, 0.0);
    Solver<HostMem, HostObs> S(P, HostMem{ws.data()}, HostObs{obs});
    for (int i = 0; i < NX; ++i) S.x0[i] = x0[i];
    for (int j = 0; j < NU; ++j) S.uprev[j] = u_prev[j];
    S.xg[0] = goal[0]; S.xg[1] = goal[1];
    S.solve(*status, *iters);
    for (int j = 0; j < NU; ++j) u_out[j] = ws[L.z + j];
    if (z_out) for (int i = 0; i < L.n; ++i) z_out[i] = ws[L.z + i];
    return 0;
}
