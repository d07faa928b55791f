// Microbenchmark: v_fmac_f64 with a DPP row_newbcast operand vs plain v_fma_f64 on gfx950; also checks the
// broadcast semantics (lane n of each 16-lane row feeds all lanes of that row).
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMAC_DPP(acc, bc, x, N) \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(x))

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, int iters, double seed)
{
    double g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = 0.0;
    double bc = seed * 1e-9 * (threadIdx.x + 1), x = 1.0 + threadIdx.x * 1e-6, bc2 = bc * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = fma(bc, x, g[i]);
            } else {
                FMAC_DPP(g[0], bc, x, 0); FMAC_DPP(g[1], bc, x, 1); FMAC_DPP(g[2], bc, x, 2); FMAC_DPP(g[3], bc, x, 3);
                FMAC_DPP(g[4], bc, x, 4); FMAC_DPP(g[5], bc, x, 5); FMAC_DPP(g[6], bc, x, 6); FMAC_DPP(g[7], bc, x, 7);
                FMAC_DPP(g[8], bc2, x, 8); FMAC_DPP(g[9], bc2, x, 9); FMAC_DPP(g[10], bc2, x, 10); FMAC_DPP(g[11], bc2, x, 11);
                FMAC_DPP(g[12], bc2, x, 12); FMAC_DPP(g[13], bc2, x, 13); FMAC_DPP(g[14], bc2, x, 14); FMAC_DPP(g[15], bc2, x, 15);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += g[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

__global__ void k_check(double *out)
{
    double acc = 0.0, bc = 100.0 + threadIdx.x, x = 1.0;
    FMAC_DPP(acc, bc, x, 5);
    out[threadIdx.x] = acc;          // expect 100 + 16*(lane/16) + 5
}

template <int MODE> void run(const char *name, int w)
{
    const int nwg = 1024 * w, iters = 40000;
    double *out; (void)hipMalloc(&out, nwg * 64 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64), 0, 0, out, 2000, 1.0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(64), 0, 0, out, iters, 1.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-24s w/SIMD=%d %8.2f ms  %5.1f TF\n", name, w, ms, (double)nwg * iters * 64 * 128.0 / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}

int main()
{
    double *o; (void)hipMalloc(&o, 64 * 8);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, o);
    double h[64]; (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) if (h[l] != 100.0 + 16 * (l / 16) + 5) ++bad;
    printf("row_newbcast:5 semantics: %s (lane0=%g lane17=%g lane63=%g)\n", bad ? "MISMATCH" : "ok", h[0], h[17], h[63]);
    for (int w = 1; w <= 4; w *= 2) { run<0>("v_fma_f64", w); run<1>("v_fmac_f64_dpp newbcast", w); }
    return 0;
}
