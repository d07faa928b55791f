// Accuracy of v_rcp_f64 / v_rsq_f64 seeds and of one / two Newton steps (decides the pivot arithmetic of the kriging kernels).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off rcp_f64.hip -o rcp_f64 && ./rcp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double y = __builtin_amdgcn_rcp(d);
    out[i] = y;
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    out[n + i] = y;
    e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    out[2 * n + i] = y;
    double r = __builtin_amdgcn_rsq(d);
    out[3 * n + i] = r;
    r = r * fma(-0.5 * d * r, r, 1.5);
    out[4 * n + i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(5 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 41) - 20); }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
    const char *nm[5] = {"rcp seed", "rcp + 1 NR", "rcp + 2 NR", "rsq seed", "rsq + 1 NR"};
    for (int v = 0; v < 5; ++v) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            long double ref = v < 3 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
            long double e = fabsl(((long double)o[v * n + i] - ref) / ref);
            if (e > worst) worst = e;
        }
        printf("%-12s max rel err %.3Le\n", nm[v], worst);
    }
    return 0;
}
