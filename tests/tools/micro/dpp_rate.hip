// Microbenchmark: issue rate of v_fmac_f64 with a DPP row_newbcast source against the plain form and against the form
// with a scalar (SGPR pair) multiplicand -- the three ways a wave-uniform weight can reach the daily kernels' fmacs --,
// and of the f32 <-> f64 conversions.
//   hipcc --offload-arch=gfx950 -O3 -o tests/tools/micro/dpp_rate tests/tools/micro/dpp_rate.hip && tests/tools/micro/dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, const double *in, int iters)
{
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = 1.0 + (double)(threadIdx.x + i);
    double z = in[threadIdx.x & 15], x = in[16 + (threadIdx.x & 63)];
    const double sz = in[blockIdx.x & 7];                    // wave-uniform: lives in an SGPR pair
    float xf[8];
    for (int i = 0; i < 8; ++i) xf[i] = (float)in[32 + i] + (float)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(z), "v"(x));
            if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(z), "v"(x));
            if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(sz), "v"(x));
            if (MODE == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(acc[i]) : "v"(xf[i]));
            if (MODE == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(xf[i]) : "v"(acc[i]));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i] + (double)xf[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, double *out, const double *in)
{
    const int nwg = 256 * 8, iters = 100000;                 // 8 work-groups of 4 waves per CU: 8 waves per SIMD, issue-bound
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, out, in, 1000);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 64 * 8.0 * iters * (nwg * 4.0);
    printf("%-34s %8.2f ms  %6.1f TFLOP/s if they were FMAs = %.2f cycles per wave instruction\n", name, ms, flops / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / (8.0 * iters * 8.0));   // 8 waves per SIMD x 8 instructions per iteration (2.4 GHz)
}

int main()
{
    double *out, *in; (void)hipMalloc(&out, 256 * 8 * 256 * 8); (void)hipMalloc(&in, 1024);
    (void)hipMemset(in, 0, 1024);
    run<0>("v_fmac_f64 (VGPR x VGPR)", out, in);
    run<1>("v_fmac_f64_dpp row_newbcast", out, in);
    run<2>("v_fma_f64 (SGPR x VGPR)", out, in);
    run<3>("v_cvt_f64_f32", out, in);
    run<4>("v_cvt_f32_f64", out, in);
    return 0;
}
