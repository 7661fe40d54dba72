// Micro-benchmark: sustained VALU issue rate of gfx950 for f64 / f32 FMA streams (what "VALU-issue peak" means on this
// part under load).  hipcc --offload-arch=gfx950 -O3 tools/micro/valu_peak.hip -o exp_libs/valu_peak ; ./exp_libs/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int WPB>
__global__ __launch_bounds__(WPB * 64) void fma_stream(T* out, int iters, T seed) {
    T a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const T m = (T)1.0000001, c = (T)1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
            a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <typename T>
static void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;                 // 256-thread blocks: one wave per SIMD each
    T* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(T));
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fma_stream<T, 4>), dim3(blocks), dim3(256), 0, 0, out, iters, (T)1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * 4 * iters * 128;          // wave-instructions
        const double rate = insts / (ms * 1e-3);                         // wave-instructions / s on the whole chip
        printf("%s waves/SIMD %d: %.3f ms, %.1f G wave-instr/s = %.3f G per SIMD (4-cycle issue => %.2f GHz), %.1f TFLOP/s\n", name,
               waves_per_simd, ms, rate / 1e9, rate / 1e9 / 1024, rate / 1e9 / 1024 * 4, rate * 128 / 1e12);
    }
    hipFree(out);
}
int main() {
    run<double>("f64 fma", 1); run<double>("f64 fma", 2); run<double>("f64 fma", 4);
    run<float>("f32 fma", 1); run<float>("f32 fma", 2); run<float>("f32 fma", 4);
    return 0;
}
