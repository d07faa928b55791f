// twx_agg.h -- monthly / annual aggregation of the daily product.
//
// _TairAggregate (twx/interp/tiling.py:1080-1166) behind write_ds_mthly (tiling.py:1169-1219,
// scripts/step27_create_monthly.py): per (year, month) group the mean over its days of every cell,
// per year the mean of its monthly means, the monthly means rounded to 2 decimals and packed into an
// 'i2' netCDF variable with scale_factor float32(0.01).
//
// HBM-bound streaming reduction: the daily cube [ndays][ncell] is read exactly once (2 B per cell-day
// for the raw int16 product), cells are the coalesced axis, every thread owns VEC adjacent cells and
// walks the days of one group after the other in chronological order (the summation order of
// np.ma.mean over axis 0, so the f8 means are bit-identical to numpy's).
#pragma once
#include "twx_device.h"

struct AggAxis {
    int ng, nyr, nmth;           // groups = nyr * nmth, year-major (tiling.py:1101-1107)
    const int32_t *gstart;       // [ng + 1] offsets into gday
    const int32_t *gday;         // day indices of each group, ascending
};

// one daily value as netCDF4 / numpy hand it to daily_to_mthly; false = masked
__device__ __forceinline__ bool agg_value(int16_t r, double &v)
{
    // auto mask-and-scale of the 'i2' variable: int16 * np.float32(0.01) is a float32 product
    v = (double)((float)r * 0.01f);
    return r != (int16_t)TWX_FILL_I2;
}
__device__ __forceinline__ bool agg_value(float r, double &v) { v = (double)r; return r == r; }
__device__ __forceinline__ bool agg_value(double r, double &v) { v = r; return r == r; }

// write_ds_mthly: np.ma.round(x, 2) then netCDF4 packing np.around(x / scale_factor)
__device__ __forceinline__ int16_t pack_mthly_i16(double mean)
{
    if (!(mean == mean)) return (int16_t)TWX_FILL_I2;
    const double r = rint(mean * 100.0) / 100.0;
    return (int16_t)(int)rint(r / (double)0.01f);
}

template <class T, int VEC> struct AggVec { T v[VEC]; };

template <class T, int VEC>
__global__ __launch_bounds__(256) void k_agg(const T *__restrict__ daily, int64_t ncell, AggAxis ax,
                                             double *__restrict__ mthly, int16_t *__restrict__ mthly_i16,
                                             double *__restrict__ ann)
{
    typedef AggVec<T, VEC> __attribute__((aligned(sizeof(T) * VEC))) vec_t;
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (c0 >= ncell) return;
    int g = 0;
    for (int y = 0; y < ax.nyr; ++y) {
        double ys[VEC];
        int yn[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { ys[e] = 0.0; yn[e] = 0; }
        for (int m = 0; m < ax.nmth; ++m, ++g) {
            double s[VEC];
            int n[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { s[e] = 0.0; n[e] = 0; }
            const int a = ax.gstart[g], b = ax.gstart[g + 1];
            int q = a;
            // 8 days in flight per thread: the loads are independent, the adds stay in day order
            for (; q + 8 <= b; q += 8) {
                vec_t r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    r[u] = *reinterpret_cast<const vec_t *>(daily + (int64_t)ax.gday[q + u] * ncell + c0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        double v;
                        if (agg_value(r[u].v[e], v)) { s[e] += v; n[e]++; }
                    }
            }
            for (; q < b; ++q) {
                const vec_t r = *reinterpret_cast<const vec_t *>(daily + (int64_t)ax.gday[q] * ncell + c0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    double v;
                    if (agg_value(r.v[e], v)) { s[e] += v; n[e]++; }
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                // np.ma.mean: dsum * 1. / count; an empty or fully masked group is masked (NaN here)
                const double mean = n[e] ? s[e] * 1.0 / (double)n[e] : __builtin_nan("");
                if (n[e]) { ys[e] += mean; yn[e]++; }
                const int64_t o = (int64_t)g * ncell + c0 + e;
                if (mthly) mthly[o] = mean;
                if (mthly_i16) mthly_i16[o] = pack_mthly_i16(mean);
            }
        }
        if (ann) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                ann[(int64_t)y * ncell + c0 + e] = yn[e] ? ys[e] * 1.0 / (double)yn[e] : __builtin_nan("");
        }
    }
}

template <class T, int VEC>
void launch_agg(const void *daily, int64_t ncell, const AggAxis &ax, double *mthly, int16_t *mi16, double *ann,
                hipStream_t stream)
{
    const int64_t nthr = (ncell + VEC - 1) / VEC;
    hipLaunchKernelGGL((k_agg<T, VEC>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                       static_cast<const T *>(daily), ncell, ax, mthly, mi16, ann);
}
