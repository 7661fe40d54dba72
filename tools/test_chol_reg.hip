// Standalone check of the register Cholesky (safe_control_amd/csrc/mpc_chol.hpp) against a host solve.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Isafe_control_amd/csrc tools/test_chol_reg.hip -o /tmp/test_chol && /tmp/test_chol
#include <cmath>
#include <cstdio>
#include <vector>

#include "mpc_chol.hpp"

template <int n>
__global__ void kern(const double* A, const double* rhs, double* x, int* okflag) {
    __shared__ double Lt[n * (n + 1)];
    const int lane = threadIdx.x;
    double a[n], dinv;
#pragma unroll
    for (int k = 0; k < n; ++k) a[k] = A[(lane < n ? lane : 0) * n + k];
    const bool ok = sc::chol_reg<n>(a, lane, dinv);
    const double v = sc::chol_solve_reg<n>(a, dinv, rhs[lane < n ? lane : 0], Lt, lane);
    if (lane < n) x[lane] = v;
    if (lane == 0) *okflag = ok;
}

template <int n>
static double run() {
    std::vector<double> A(n * n), b(n), x(n), B(n * n);
    unsigned s = 12345u + n;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (auto& v : B) v = rnd();
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = (i == j) ? 0.5 : 0.0;
            for (int k = 0; k < n; ++k) acc += B[i * n + k] * B[j * n + k];
            A[i * n + j] = acc;
        }
    for (auto& v : b) v = rnd();
    double *dA, *db, *dx; int* dok;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&db, sizeof(double) * n); hipMalloc(&dx, sizeof(double) * n); hipMalloc(&dok, 4);
    hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern<n>, dim3(1), dim3(64), 0, 0, dA, db, dx, dok);
    int ok = 0;
    hipMemcpy(x.data(), dx, sizeof(double) * n, hipMemcpyDeviceToHost);
    hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
    double rmax = 0;
    for (int i = 0; i < n; ++i) {
        double acc = -b[i];
        for (int j = 0; j < n; ++j) acc += A[i * n + j] * x[j];
        rmax = std::fmax(rmax, std::fabs(acc));
    }
    printf("n = %d ok = %d residual |A x - b|_inf = %.3e\n", n, ok, rmax);
    return rmax;
}

int main() {
    double r = 0;
    r = std::fmax(r, run<8>()); r = std::fmax(r, run<20>()); r = std::fmax(r, run<24>()); r = std::fmax(r, run<33>()); r = std::fmax(r, run<40>());
    return r < 1e-9 ? 0 : 1;
}
