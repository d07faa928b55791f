// Microbenchmark: issue cost of v_cvt_f64_f32 beside v_fma_f64 on gfx950 (64 per iteration, 16 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, float seed)
{
    float f[16];
    double g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { f[i] = seed + i + threadIdx.x * 1e-3f; g[i] = f[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // the bit flip keeps f[i] loop-variant without touching the fp pipes' rates
                if (MODE == 0) { g[i] = fma(g[i], 0.999, 1e-3); f[i] = __int_as_float(__float_as_int(f[i]) ^ 1); }
                else { g[i] = fma((double)f[i], 0.999, g[i]); f[i] = __int_as_float(__float_as_int(f[i]) ^ 1); }
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += g[i] + f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int w)
{
    const int nwg = 1024 * w, iters = 20000;
    double *out; (void)hipMalloc(&out, nwg * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64), 0, 0, out, 1000, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s w/SIMD=%d %8.2f ms  %.1f ns per 64-op iteration per wave\n", name, w, ms, ms * 1e6 / iters);
    (void)hipFree(out);
}

int main()
{
    for (int w = 1; w <= 4; w *= 4) {
        run<0>("(fma_f64 + xor) x64", w); run<1>("(cvt_f64_f32 + fma_f64 + xor) x64", w);
    }
    return 0;
}
