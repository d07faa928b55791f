// Microbenchmark: issue cost (cycles per wave instruction at one wave per SIMD, independent operands) and dependent-chain
// latency of the transcendental / conversion instructions the kriging kernels' pivot chain is made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o tests/tools/micro/trans_rate tests/tools/micro/trans_rate.hip && tests/tools/micro/trans_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP1(name, ins, T, cons)                                                                      \
    __global__ __launch_bounds__(64) void name(T *out, int iters, T seed, int dep)                   \
    {                                                                                                \
        T g[8];                                                                                      \
        for (int i = 0; i < 8; ++i) g[i] = seed + (T)(threadIdx.x + i);                              \
        for (int it = 0; it < iters; ++it) {                                                         \
            if (dep) {                                                                               \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) asm volatile(ins " %0, %0" : "+" cons(g[0]));   \
            } else {                                                                                 \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ins " %0, %0" : "+" cons(g[i]));   \
            }                                                                                        \
        }                                                                                            \
        T s = 0;                                                                                     \
        for (int i = 0; i < 8; ++i) s += g[i];                                                       \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                      \
    }

OP1(k_rsq64, "v_rsq_f64", double, "v")
OP1(k_rcp64, "v_rcp_f64", double, "v")
OP1(k_sqrt64, "v_sqrt_f64", double, "v")
OP1(k_rsq32, "v_rsq_f32", float, "v")
OP1(k_exp32, "v_exp_f32", float, "v")

__global__ __launch_bounds__(64) void k_fma64(double *out, int iters, double seed, int dep)
{
    double g[8];
    for (int i = 0; i < 8; ++i) g[i] = seed + (double)(threadIdx.x + i);
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
        if (dep) {
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(g[0]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(g[i]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += g[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class T, class K> void run(const char *name, K kern)
{
    const int nwg = 1024, iters = 600000;           // one wave per SIMD
    T *out; (void)hipMalloc(&out, nwg * 64 * sizeof(T));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int dep = 0; dep < 2; ++dep) {
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64), 0, 0, out, 1000, (T)1.5, dep);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(64), 0, 0, out, iters, (T)1.5, dep);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);   // kHz
        const double cyc = ms * 1e-3 * clk * 1e3 / ((double)iters * 8);
        printf("%-12s %-11s %7.1f cycles per instruction (%.2f ms, %d MHz nominal)\n", name, dep ? "dependent" : "independent", cyc, ms, clk / 1000);
    }
    (void)hipFree(out);
}

int main()
{
    run<double>("v_fma_f64", k_fma64);
    run<double>("v_rsq_f64", k_rsq64);
    run<double>("v_rcp_f64", k_rcp64);
    run<double>("v_sqrt_f64", k_sqrt64);
    run<float>("v_rsq_f32", k_rsq32);
    run<float>("v_exp_f32", k_exp32);
    return 0;
}
