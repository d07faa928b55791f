// EXPERIMENT RECORD (round 6) -- NOT part of the library, not compiled by build.sh.
// k_bucket_order: descending neighbourhood size inside each XCD's eighth of every kriging bucket's item list (VERDICT r5 #4b), so
// that the long systems of a launch start first and its tail consists of the short ones.
// MEASURED (one gpurun call, tests/tools/ab_bench.sh, three alternating runs each, C2 step):
//     library        step 11.01 / 11.13 / 10.99 ms   kriging kernels 9.86 / 9.93 / 9.85 ms
//     with the sort  step 11.43 / 11.37 / 11.32 ms   kriging kernels 10.19 / 10.15 / 10.12 ms      (+0.27 ms = +2.8 %: KILLED)
// Why it loses: k_bucket_items leaves the (cell, month) items in roughly cell order, so the twelve systems of a cell -- which read
// the SAME 55 KB pair-distance slab -- sit next to each other in a list and meet in one XCD's L2 within microseconds of each other;
// sorted by k they are spread over the launch.  The tail it was meant to shorten is one round of ~62 (95 000 systems at 6 per CU).
// To try it again: put the kernel into twx_select.h, allocate a second list (Work::bucket_sorted, ncell * 12 * TWX_NBUCKET ints) and
// launch it between k_bucket_items' read-back and the kriging launches of run_uk_stage (twx_hip.hip):
//
//   #if TWX_BUCKET_ORDER   // (experiment: descending k inside each XCD's eighth of every bucket's list)
//       HIPCHK(w.bucket_sorted.ensure((size_t)ncell * 12 * TWX_NBUCKET * 4));
//       hipLaunchKernelGGL(k_bucket_order, dim3(TWX_NBUCKET * 8), dim3(256), 0, stream, w.ws, w.bucket_sorted.as<int32_t>());
//       w.ws.bucket_cells = w.bucket_sorted.as<int32_t>();
//   #endif
#pragma once
#ifndef TWX_BUCKET_ORDER
#define TWX_BUCKET_ORDER 0     // 1: k_bucket_order (an experiment of round 6: see there)
#endif
// ---------------------------------------------------------------------------------
// k_bucket_order (TWX_BUCKET_ORDER = 1 builds only): descending k inside each XCD's eighth of a bucket's item list.
// The kriging kernels deal a list to the 8 XCDs in contiguous eighths (xcd_contig) and an XCD takes its work-groups in
// index order: with the eighth sorted by descending neighbourhood size the long systems of a launch start first and its
// tail consists of the short ones.  One work-group per (bucket, eighth): counting sort by (largest k of the bucket - k)
// into a second list.  Measured (round 6, same-box A/B on the C2 step): see EXPERIMENTS.md -- below the 1.5 % that
// would have kept it.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bucket_order(SelWs ws, int32_t *sorted)
{
    __shared__ int s_cnt[16], s_off[16];
    const int b = blockIdx.x >> 3, x = blockIdx.x & 7, t = threadIdx.x;
    const int n = ws.bucket_cnt[b];
    const int per = (n + 7) >> 3, i0 = x * per, i1 = min(n, i0 + per);
    const int32_t *in = ws.bucket_cells + (int64_t)b * ws.ncell * 12;
    int32_t *out = sorted + (int64_t)b * ws.ncell * 12;
    if (t < 16) s_cnt[t] = 0;
    __syncthreads();
    int kmax = 0;
    for (int i = i0 + t; i < i1; i += 256) kmax = max(kmax, ws.kk[in[i]]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
    __shared__ int s_kmax[4];
    if ((t & 63) == 0) s_kmax[t >> 6] = kmax;
    __syncthreads();
    kmax = max(max(s_kmax[0], s_kmax[1]), max(s_kmax[2], s_kmax[3]));
    for (int i = i0 + t; i < i1; i += 256) atomicAdd(&s_cnt[min(kmax - ws.kk[in[i]], 15)], 1);
    __syncthreads();
    if (t == 0) { int a = 0; for (int q = 0; q < 16; ++q) { s_off[q] = a; a += s_cnt[q]; } }
    __syncthreads();
    for (int i = i0 + t; i < i1; i += 256) {
        const int item = in[i];
        out[i0 + atomicAdd(&s_off[min(kmax - ws.kk[item], 15)], 1)] = item;
    }
}

