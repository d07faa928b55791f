// twx_uk1.h -- universal-kriging kernel for SMALL systems: one wavefront per
// (cell, month) item.
//
// Same algorithm as k_uk (twx_uk.h): bordered matrix [[C, B], [B', 0]] held in
// registers, right-looking Cholesky in 4-column panels, Schur complement -B'C^-1B
// in the trailing 7x7 block.  Differences for k + 8 <= 80 rows:
//   * ONE 64-lane wave owns the whole matrix, 2-D cyclic over an 8x8 lane grid:
//     lane (tr, tc) holds element (8a+tr, 8b+tc) of every 8x8 block (a >= b).
//     No wave ever waits for another one (the 4-wave kernel idles three waves during
//     every panel factorisation), and the block granularity is 8 instead of 16, so
//     far fewer updates hit already finished rows / columns.
//   * a panel is half a block column: its four columns live in one half of the wave.
// The covariance build, the slab layout and the GLS epilogue (k_uk_solve) are shared.
#pragma once
#include "twx_uk.h"

// waves per SIMD the register budget is sized for (measured, see twx_uk.h)
#define TWX_UK1_WAVES(NB8) ((NB8) >= 9 ? 3 : ((NB8) >= 7 ? 4 : 5))

template <int NB8>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(TWX_UK1_WAVES(NB8), TWX_UK1_WAVES(NB8))))
void k_uk1(StnDev st, CellSrc src, SelWs ws, const int32_t *item_list, int nitems)
{
    constexpr int NP = NB8 * 8, NT = NB8 * (NB8 + 1) / 2;
    constexpr int PS = 6;       // slab row stride (doubles): 48-byte rows, 16-B aligned
    __shared__ __attribute__((aligned(16))) double s_pan[NP * PS];
    __shared__ double s_B[7][NP];
    __shared__ double s_trig[NP * 4];
    __shared__ double s_cphi[NP];
    __shared__ int s_err;

    const int lane = threadIdx.x, tr = lane & 7, tc = lane >> 3, tcl = tc & 3, half = tc >> 2;
    if ((int)blockIdx.x >= nitems) return;
    const int item = item_list[blockIdx.x];
    const int64_t lc = item / 12;
    const int m0 = item % 12;
    const int64_t c = ws.cell0 + lc;
    const int k = ws.kk[lc * 12 + m0];
    const size_t n = (size_t)st.n;
    const CellVals cv = cell_load(src, c);
    const double plst = cell_lst(src, c, m0);
    const double nug = ws.vario[(lc * 12 + m0) * 3 + 0];
    const double psill = ws.vario[(lc * 12 + m0) * 3 + 1];
    const double rng = ws.vario[(lc * 12 + m0) * 3 + 2];
    const double c00 = nug + psill;
    const double c2 = rng == 0.0 ? 0.0 : -1.4426950408889634 / rng;   // -log2(e) / range
    const float chi = (float)c2, clo = (float)(c2 - (double)chi);

    // ---- staging: neighbours t = lane, lane + 64 (NP <= 80) -----------------------------------
    double xs[2][4], yv[2], c0v[2];
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u;
        xs[u][0] = xs[u][1] = xs[u][2] = xs[u][3] = 0.0; yv[u] = 0.0; c0v[u] = 0.0;
        if (t < NP) {
            double sp = 0, cp = 1, sl = 0, cl = 1;
            if (t < k) {
                const int j = ws.near_idx[lc * ws.ksel + t];
                sp = st.sph[j]; cp = st.cph[j]; sl = st.slh[j]; cl = st.clh[j];
                const double lo = st.lon[j], la = st.lat[j];
                xs[u][0] = lo - cv.lon; xs[u][1] = la - cv.lat; xs[u][2] = st.elev[j] - cv.elev;
                xs[u][3] = st.lst[m0 * n + j] - plst;
                yv[u] = st.norm[m0 * n + j];
                const double *ct = ws.ctrig + lc * 4;
                const float h0 = ellip_pair_fast(ct[0], ct[1], ct[2], ct[3], fma(ct[1], ct[1], -(ct[0] * ct[0])),
                                                 sp, cp, sl, cl, fma(cp, cp, -(sp * sp)));
                const bool same = (lo == cv.lon && la == cv.lat) || h0 == 0.f;
                c0v[u] = same ? c00 : (rng == 0.0 ? 0.0 : psill * (double)exp2_neg_split(h0, chi, clo));
                e0 = fmax(e0, fabs(xs[u][0])); e1 = fmax(e1, fabs(xs[u][1]));
                e2 = fmax(e2, fabs(xs[u][2])); e3 = fmax(e3, fabs(xs[u][3]));
            }
            s_trig[t * 4 + 0] = sp; s_trig[t * 4 + 1] = cp; s_trig[t * 4 + 2] = sl; s_trig[t * 4 + 3] = cl;
            s_cphi[t] = fma(cp, cp, -(sp * sp));
        }
    }
    e0 = wave_max(e0); e1 = wave_max(e1); e2 = wave_max(e2); e3 = wave_max(e3);
    const double sc0 = e0 > 0.0 ? 1.0 / e0 : 1.0, sc1 = e1 > 0.0 ? 1.0 / e1 : 1.0;
    const double sc2 = e2 > 0.0 ? 1.0 / e2 : 1.0, sc3 = e3 > 0.0 ? 1.0 / e3 : 1.0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u;
        if (t < NP) {
            s_B[0][t] = t < k ? 1.0 : 0.0;
            s_B[1][t] = xs[u][0] * sc0; s_B[2][t] = xs[u][1] * sc1; s_B[3][t] = xs[u][2] * sc2; s_B[4][t] = xs[u][3] * sc3;
            s_B[5][t] = yv[u]; s_B[6][t] = c0v[u];
        }
    }
    if (lane == 0) s_err = 0;
    __syncthreads();

    // ---- build this lane's elements -----------------------------------------------------------------
    double A[NT];
    sfor<0, NB8>([&](auto a_) __attribute__((always_inline)) {
        constexpr int a = decltype(a_)::value;
        const int i = 8 * a + tr;
        const double spi = s_trig[i * 4], cpi = s_trig[i * 4 + 1], sli = s_trig[i * 4 + 2], cli = s_trig[i * 4 + 3];
        const double cphi = s_cphi[i];
        sfor<0, a + 1>([&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value;
            const int j = 8 * b + tc;
            double v = 0.0;
            if (i < k && j < k) {
                if (i == j) v = c00;
                else {
                    const float h = ellip_pair_fast(spi, cpi, sli, cli, cphi, s_trig[j * 4], s_trig[j * 4 + 1],
                                                    s_trig[j * 4 + 2], s_trig[j * 4 + 3], s_cphi[j]);
                    v = h == 0.f ? c00 : (rng == 0.0 ? 0.0 : psill * (double)exp2_neg_split(h, chi, clo));
                }
            }
            if (a == NB8 - 1 && tr >= 1 && j < k) v = s_B[tr - 1][j];   // RHS rows NP-7..NP-1
            A[tri(a, b)] = v;
        });
    });

    // ---- elimination: panels of four columns = half a block column -------------------------------------
    sfor<0, NB8>([&](auto bp_) __attribute__((always_inline)) {
        constexpr int bp = decltype(bp_)::value;
        const int ncb = k - 8 * bp;                          // C columns left in this block column
        if (ncb > 0) {
            const int npan = ncb > 4 ? 2 : 1;
            for (int h = 0; h < npan; ++h) {
                const int ncol = min(4, ncb - 4 * h);        // real columns in this panel
                if (half == h) {                             // the half-wave holding the panel's columns
                    sfor<0, 4>([&](auto cc_) __attribute__((always_inline)) {
                        constexpr int cc = decltype(cc_)::value;
                        if (cc < ncol) {                     // uniform
                            const int q = 4 * h + cc;        // column within the block; diagonal sits in lane 9q
                            double d = readlane_d(A[tri(bp, bp)], 9 * q);
                            const bool bad = !(d > 1e-9 * c00) || !finite_d(d);
                            const double rinv = bad ? 0.0 : rsqrt_nr(d);
                            if (bad && tr == 0 && tcl == 0) s_err = 1;
                            if (tcl == cc) {                 // lanes of column cc: scale and publish
                                sfor<bp, NB8>([&](auto a_) __attribute__((always_inline)) {
                                    constexpr int a = decltype(a_)::value;
                                    double v = A[tri(a, bp)] * rinv;
                                    if (a == bp && tr <= q) v = 0.0;      // rows at / above the diagonal
                                    A[tri(a, bp)] = v;
                                    s_pan[(8 * a + tr) * PS + cc] = v;
                                });
                            }
                            if constexpr (cc < 3) {          // remaining columns: a(i,p') -= l(i,p) l(p',p)
                                __builtin_amdgcn_wave_barrier();
                                const double lpp = s_pan[(8 * bp + 4 * h + tcl) * PS + cc];
                                sfor<bp, NB8>([&](auto a_) __attribute__((always_inline)) {
                                    constexpr int a = decltype(a_)::value;
                                    const double li = s_pan[(8 * a + tr) * PS + cc];
                                    if (tcl > cc) A[tri(a, bp)] = fma(-li, lpp, A[tri(a, bp)]);
                                });
                            }
                        } else if (tcl == cc) {
                            sfor<bp, NB8>([&](auto a_) __attribute__((always_inline)) {
                                constexpr int a = decltype(a_)::value;
                                A[tri(a, bp)] = 0.0;
                                s_pan[(8 * a + tr) * PS + cc] = 0.0;
                            });
                        }
                    });
                }
                __syncthreads();
                // rank-4 update, two block columns at a time
                sfor2<bp, NB8>([&](auto b_) __attribute__((always_inline)) {
                    constexpr int b = decltype(b_)::value;
                    constexpr bool two = (b + 1 < NB8);
                    const double2 p0 = *reinterpret_cast<const double2 *>(&s_pan[(8 * b + tc) * PS]);
                    const double2 p1 = *reinterpret_cast<const double2 *>(&s_pan[(8 * b + tc) * PS + 2]);
                    double2 q0 = p0, q1 = p1;
                    if constexpr (two) {
                        q0 = *reinterpret_cast<const double2 *>(&s_pan[(8 * (b + 1) + tc) * PS]);
                        q1 = *reinterpret_cast<const double2 *>(&s_pan[(8 * (b + 1) + tc) * PS + 2]);
                    }
                    sfor<b, NB8>([&](auto a_) __attribute__((always_inline)) {
                        constexpr int a = decltype(a_)::value;
                        const double2 u0 = *reinterpret_cast<const double2 *>(&s_pan[(8 * a + tr) * PS]);
                        const double2 u1 = *reinterpret_cast<const double2 *>(&s_pan[(8 * a + tr) * PS + 2]);
                        {
                            double acc = A[tri(a, b)];
                            acc = fma(-u0.x, p0.x, acc);
                            acc = fma(-u0.y, p0.y, acc);
                            acc = fma(-u1.x, p1.x, acc);
                            acc = fma(-u1.y, p1.y, acc);
                            A[tri(a, b)] = acc;
                        }
                        if constexpr (two && a >= b + 1) {
                            double acc = A[tri(a, b + 1)];
                            acc = fma(-u0.x, q0.x, acc);
                            acc = fma(-u0.y, q0.y, acc);
                            acc = fma(-u1.x, q1.x, acc);
                            acc = fma(-u1.y, q1.y, acc);
                            A[tri(a, b + 1)] = acc;
                        }
                    });
                });
                __syncthreads();        // the single slab is rewritten by the next panel
            }
        }
    });

    // ---- Schur complement out (k_uk_solve finishes) -------------------------------------------------------
    if (tr >= 1 && tc >= 1 && tr >= tc) {
        const int r = tr - 1, cq = tc - 1;
        ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + r * (r + 1) / 2 + cq] = -A[tri(NB8 - 1, NB8 - 1)];
    }
    if (lane == 0) ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + 28] = s_err ? 1.0 : 0.0;
}
