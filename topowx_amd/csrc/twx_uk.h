// twx_uk.h -- universal-kriging kernel.
//
// One 256-thread workgroup per cell, twelve months in sequence.  For month m with
// k = nnghs_m neighbours (the k nearest, in rank order) the bordered matrix
//
//        M = [ C   B ]      C = k x k covariance, B = [1 x1 x2 x3 x4 | y | c0]
//            [ B'  0 ]
//
// is held ENTIRELY IN REGISTERS, distributed 2-D block-cyclically over the 16x16
// thread grid: thread (tr, tc) owns element (16a+tr, 16b+tc) of every 16x16 block
// (a >= b).  k right-looking Cholesky steps eliminate the C part; what is left in
// the trailing 7x7 block is -B'C^-1 B, i.e. every inner product the GLS predictor
// needs (X'C^-1X, X'C^-1y, X'C^-1c0, c0'C^-1c0, c0'C^-1y) -- no triangular solves.
// Each step broadcasts one scaled column through a double-buffered LDS vector
// (one workgroup barrier per step); all register indexing is compile-time.
//
// The pair distances h_ij (sp/gstat WGS84 great-circle, B.1) depend only on the
// cell's neighbour list, not on the month, so every thread computes the h of its
// own elements once per cell and keeps them in registers; a month only
// re-evaluates psill*exp(-h/range).  The seven RHS rows sit at the fixed rows
// NP-7..NP-1 of the padded matrix so the Schur complement lands in a fixed block.
#pragma once
#include "twx_select.h"

__device__ __forceinline__ constexpr int tri(int a, int b) { return a * (a + 1) / 2 + b; }

__device__ __forceinline__ double rsqrt_nr(double d)
{
    // v_rsq_f64 seed + two Newton steps: full fp64 accuracy
    double y = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}

template <int NB>
__global__ __launch_bounds__(256, (NB >= 8 ? 2 : (NB >= 6 ? 3 : 4)))
void k_uk(StnDev st, CellSrc src, SelWs ws, const int32_t *cell_list, int ncells)
{
    constexpr int NP = NB * 16, BR = NP - 7, NT = NB * (NB + 1) / 2;
    __shared__ double s_col[2][NP];
    __shared__ double s_B[7][NP];
    __shared__ double s_sph[NP], s_cph[NP], s_slh[NP], s_clh[NP];
    __shared__ double s_h0[NP];
    __shared__ double s_xs[3][NP];     // lon, lat, elev of the neighbours, minus the cell's
    __shared__ int s_idx[NP];
    __shared__ double s_red[4][4];
    __shared__ double s_S[49];
    __shared__ int s_err;

    const int t = threadIdx.x, tr = t & 15, tc = t >> 4, lane = t & 63, wv = t >> 6;
    if ((int)blockIdx.x >= ncells) return;
    const int64_t lc = cell_list[blockIdx.x];
    const int64_t c = ws.cell0 + lc;
    const int kmaxc = ws.kmaxc[lc];
    const CellVals cv = cell_load(src, c);
    const size_t n = (size_t)st.n;

    // ---- per-cell staging: neighbour trig, coordinates, cell->station distance ----
    for (int i = t; i < NP; i += 256) {
        int j = (i < kmaxc) ? ws.near_idx[lc * ws.ksel + i] : -1;
        s_idx[i] = j;
        if (j >= 0) {
            s_sph[i] = st.sph[j]; s_cph[i] = st.cph[j]; s_slh[i] = st.slh[j]; s_clh[i] = st.clh[j];
            double lo = st.lon[j], la = st.lat[j];
            s_xs[0][i] = lo - cv.lon; s_xs[1][i] = la - cv.lat; s_xs[2][i] = st.elev[j] - cv.elev;
            s_h0[i] = ellip_km(cv.lon, cv.lat, lo, la);
        } else {
            s_sph[i] = 0; s_cph[i] = 1; s_slh[i] = 0; s_clh[i] = 1;
            s_xs[0][i] = 0; s_xs[1][i] = 0; s_xs[2][i] = 0; s_h0[i] = 0;
        }
    }
    if (t == 0) s_err = 0;
    __syncthreads();

    // ---- pair distances of this thread's elements (once per cell) -------------------
    float H[NT];
#pragma unroll
    for (int a = 0; a < NB; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int i = 16 * a + tr, j = 16 * b + tc;
            float h = 0.f;
            if (i < kmaxc && j < kmaxc && i != j)
                h = (float)ellip_pair(s_sph[i], s_cph[i], s_slh[i], s_clh[i], s_sph[j], s_cph[j], s_slh[j], s_clh[j]);
            H[tri(a, b)] = h;
        }
    }

    for (int m0 = 0; m0 < 12; ++m0) {
        const int k = ws.kk[lc * 12 + m0];
        if (k <= 0) continue;                       // uniform
        const double nug = ws.vario[(lc * 12 + m0) * 3 + 0];
        const double psill = ws.vario[(lc * 12 + m0) * 3 + 1];
        const double rng = ws.vario[(lc * 12 + m0) * 3 + 2];
        const double c00 = nug + psill;
        const double irng = rng == 0.0 ? 0.0 : -1.0 / rng;
        const double plst = cell_lst(src, c, m0);
        const int nbk = (k + 15) >> 4;              // block rows holding C rows

        // ---- RHS columns: trend (shifted to the cell, scaled), y, c0 ----------------
        // NP <= 160 < 256: one neighbour per thread
        double xl = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
        if (t < k) {
            xl = st.lst[m0 * n + s_idx[t]] - plst;
            e0 = fabs(s_xs[0][t]); e1 = fabs(s_xs[1][t]); e2 = fabs(s_xs[2][t]); e3 = fabs(xl);
        }
        e0 = wave_max(e0); e1 = wave_max(e1); e2 = wave_max(e2); e3 = wave_max(e3);
        if (lane == 0) { s_red[wv][0] = e0; s_red[wv][1] = e1; s_red[wv][2] = e2; s_red[wv][3] = e3; }
        __syncthreads();
        double sc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double s = fmax(fmax(s_red[0][q], s_red[1][q]), fmax(s_red[2][q], s_red[3][q]));
            sc[q] = s > 0.0 ? 1.0 / s : 1.0;
        }
        if (t < NP) {
            bool in = t < k;
            s_B[0][t] = in ? 1.0 : 0.0;
            s_B[1][t] = in ? s_xs[0][t] * sc[0] : 0.0;
            s_B[2][t] = in ? s_xs[1][t] * sc[1] : 0.0;
            s_B[3][t] = in ? s_xs[2][t] * sc[2] : 0.0;
            s_B[4][t] = in ? xl * sc[3] : 0.0;
            s_B[5][t] = in ? st.norm[m0 * n + s_idx[t]] : 0.0;
            double h0 = s_h0[t];
            s_B[6][t] = in ? (h0 == 0.0 ? c00 : (rng == 0.0 ? 0.0 : psill * exp(h0 * irng))) : 0.0;
        }
        __syncthreads();

        // ---- build this thread's elements -------------------------------------------
        double A[NT];
#pragma unroll
        for (int a = 0; a < NB; ++a) {
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const int i = 16 * a + tr, j = 16 * b + tc;
                double v = 0.0;
                if (i < k) {
                    if (j < k) {
                        float h = H[tri(a, b)];
                        v = (i == j || h == 0.f) ? c00 : (rng == 0.0 ? 0.0 : psill * exp((double)h * irng));
                    }
                } else if (a == NB - 1 && tr >= 9) {
                    if (j < k) v = s_B[tr - 9][j];
                }
                A[tri(a, b)] = v;
            }
        }

        // ---- k elimination steps -------------------------------------------------------
#pragma unroll
        for (int bp = 0; bp < NB; ++bp) {
            const int qn = min(16, k - 16 * bp);
            if (qn > 0) {
                for (int q = 0; q < qn; ++q) {
                    const int p = 16 * bp + q;
                    double *col = s_col[p & 1];
                    if (wv == (q >> 2)) {           // the wave that owns column p
                        double d = __shfl(A[tri(bp, bp)], ((q & 3) << 4) | q, 64);
                        const bool bad = !(d > 0.0) || !finite_d(d);
                        const double rinv = bad ? 0.0 : rsqrt_nr(d);
                        if ((lane >> 4) == (q & 3)) {
#pragma unroll
                            for (int a = bp; a < NB; ++a) {
                                if (a < nbk || a == NB - 1) {
                                    double v = A[tri(a, bp)] * rinv;
                                    if (a == bp && tr <= q) v = 0.0;
                                    col[16 * a + tr] = v;
                                }
                            }
                            if (bad && tr == 0) s_err = 1;
                        }
                    }
                    __syncthreads();
                    double li[NB], lj[NB];
#pragma unroll
                    for (int a = bp; a < NB; ++a) {
                        if (a < nbk || a == NB - 1) { li[a] = col[16 * a + tr]; lj[a] = col[16 * a + tc]; }
                        else { li[a] = 0.0; lj[a] = 0.0; }
                    }
#pragma unroll
                    for (int a = bp; a < NB; ++a) {
                        if (a < nbk || a == NB - 1) {
#pragma unroll
                            for (int b = bp; b <= a; ++b)
                                A[tri(a, b)] = fma(-li[a], lj[b], A[tri(a, b)]);
                        }
                    }
                }
            }
        }

        // ---- Schur complement -> GLS predictor ----------------------------------------
        if (tr >= 9 && tc >= 9) s_S[(tr - 9) * 7 + (tc - 9)] = -A[tri(NB - 1, NB - 1)];
        __syncthreads();
        if (t == 0) {
            // N = X'C^-1X (5x5), r = X'C^-1y, q = X'C^-1c0, gg = c0'C^-1c0, gb = c0'C^-1y
            double L[5][5], beta[5], u[5];
            bool bad = s_err != 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    double s = s_S[i * 7 + j];
#pragma unroll
                    for (int p = 0; p < j; ++p) s -= L[i][p] * L[j][p];
                    if (i == j) { if (!(s > 0.0)) bad = true; L[i][i] = sqrt(s); }
                    else L[i][j] = s / L[j][j];
                }
            }
            double mean = 0.0, var = 0.0;
            {
#pragma unroll
                for (int i = 0; i < 5; ++i) {       // L z = r
                    double s = s_S[i * 7 + 5];
#pragma unroll
                    for (int p = 0; p < i; ++p) s -= L[i][p] * beta[p];
                    beta[i] = s / L[i][i];
                }
#pragma unroll
                for (int i = 4; i >= 0; --i) {      // L' beta = z
                    double s = beta[i];
#pragma unroll
                    for (int p = i + 1; p < 5; ++p) s -= L[p][i] * beta[p];
                    beta[i] = s / L[i][i];
                }
                // x0 = [1, 0, 0, 0, 0] (trend columns are shifted to the cell)
#pragma unroll
                for (int i = 0; i < 5; ++i) u[i] = (i == 0 ? 1.0 : 0.0) - s_S[i * 7 + 6];
                mean = s_S[6 * 7 + 5];
#pragma unroll
                for (int i = 0; i < 5; ++i) mean += u[i] * beta[i];
#pragma unroll
                for (int i = 0; i < 5; ++i) {       // L w = u
                    double s = u[i];
#pragma unroll
                    for (int p = 0; p < i; ++p) s -= L[i][p] * u[p];
                    u[i] = s / L[i][i];
                }
                var = c00 - s_S[48];
#pragma unroll
                for (int i = 0; i < 5; ++i) var += u[i] * u[i];
                if (!finite_d(mean) || !finite_d(var)) bad = true;
            }
            if (bad) ws.uk_stat[lc] = TWX_CELL_NUMERIC;
            ws.uk_mean[lc * 12 + m0] = mean;
            ws.uk_var[lc * 12 + m0] = var;
            s_err = 0;
        }
        __syncthreads();
    }
}
