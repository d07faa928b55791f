// Microbenchmark: does a VALU instruction with only the first 16 lanes of a wave active issue faster than with all 64?
// (If the hardware skipped the passes of lane groups without an active lane, uniform work -- the kriging kernels' pivot
// chain -- could run on 16 lanes at a quarter of its issue cost.)
//   hipcc --offload-arch=gfx950 -O3 -o tests/tools/micro/exec_skip tests/tools/micro/exec_skip.hip && tests/tools/micro/exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void k(double *out, int iters, int lanes)
{
    double g[8];
    for (int i = 0; i < 8; ++i) g[i] = 1.5 + (double)(threadIdx.x + i);
    const double a = 1.0000001, b = 1e-9;
    if ((int)threadIdx.x < lanes) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(g[i]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += g[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main()
{
    const int nwg = 4096, iters = 200000;          // four waves per SIMD: issue-bound
    double *out; (void)hipMalloc(&out, nwg * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int lanes : {64, 32, 16, 1}) {
        hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, out, 1000, lanes);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, out, iters, lanes);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f64, %2d active lanes: %8.2f ms\n", lanes, ms);
    }
    return 0;
}
