// Microbenchmark: v_mfma_f64_16x16x4_f64 vs v_fma_f64 issue rates on gfx950, alone and side by side.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: 16 MFMA / iter; 1: 16 MFMA + 64 fp32 VALU; 2: 64 fp32 VALU; 3: 64 fp64 FMA; 4: 16 MFMA + 64 fp64 FMA
template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, double seed)
{
    d4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = d4{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    float f[16];
    double g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { f[i] = (float)a + i; g[i] = a + i; }
    const float fm = (float)seed * 0.999f, fa = 1e-3f;
    const double gm = seed * 0.999, ga = 1e-3;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 1 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = fmaf(f[i], fm, fa);
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = fma(g[i], gm, ga);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w + f[i] + g[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int waves_per_simd)
{
    const int nwg = 1024 * waves_per_simd, iters = 40000;
    double *out; (void)hipMalloc(&out, nwg * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64), 0, 0, out, 2000, 1.0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64), 0, 0, out, iters, 1.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const bool mf = MODE == 0 || MODE == 1 || MODE == 4;
    const double tf_m = mf ? (double)nwg * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12 : 0;
    const double tf_v = (MODE == 3 || MODE == 4) ? (double)nwg * iters * 64 * 128.0 / (ms * 1e-3) / 1e12 : 0;
    const double gi = (MODE == 1 || MODE == 2) ? (double)nwg * iters * 64 / (ms * 1e-3) / 1e9 : 0;
    printf("%-26s w/SIMD=%d %8.2f ms  mfma %5.1f TF  fma64 %5.1f TF  fp32 %6.0f Gwinst/s  ns/iter %.0f\n", name, waves_per_simd, ms,
           tf_m, tf_v, gi, ms * 1e6 / iters);
    (void)hipFree(out);
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("mfma x16", w); run<3>("fma_f64 x64", w); run<2>("fma_f32 x64", w);
        run<1>("mfma x16 + fma_f32 x64", w); run<4>("mfma x16 + fma_f64 x64", w);
    }
    return 0;
}
