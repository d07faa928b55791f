// twx_vario.h -- variogram estimation + one-parameter range fit.
//
// R get_vario_params (twx/interp/rpy/interp.R:54-113) behind BuildKrigParams.get_krig_params
// (interp_tair.py:635-698, step22) and KrigTairAll.krigall (interp_tair.py:722-769, step21):
//   pass 0: OLS residuals of tair ~ lon + lat + elev + lst over the k neighbours, binned
//           semivariogram (5 km bins of the sp/gstat distance, cutoff 1.4 x the largest neighbour
//           distance), nugget fixed at min(gamma), total sill at var(residuals), range fitted by
//           weighted Gauss-Newton (weights np/h^2); pure-nugget fallback (interp.R:63-80)
//   GLS trend with that model: the kriging kernels (k_uk / k_ukw + k_uk_solve) deliver
//           beta = (X'C^-1X)^-1 X'C^-1 y for the same neighbourhood (interp.R:82-84)
//   pass 1: the same on the GLS residuals -> (nug, psill, range) (interp.R:85-112)
// gstat cannot run here: the estimator and the optimiser's stopping rule are restated (scheme in
// DESIGN.md section 8); parity with gstat itself is unpinned.
//
// One 256-thread workgroup per (point, month) item.
#pragma once
#include "twx_uk.h"

#define TWX_VBINS 512        // 5 km bins: cutoff up to 2 555 km

// ---- the semivariogram's pair distance.  ellip_pair_f64 (twx_uk.h) spends most of its ~120 instructions in two IEEE
// divisions and two IEEE square roots; here the same formula with the hardware reciprocal / reciprocal square root and
// Newton steps: the main term sqrt(S) to an ulp or two, the terms of the flattening correction (weight f = 1/298) to
// ~1e-14 -- the distance differs from ellip_pair_f64's by a few ulp, which moves a pair across a 5 km bin boundary
// with probability ~1e-15 and the bin's mean distance by less.  (Three quarters of k_vario's time was this loop.)
__device__ __forceinline__ double rcp_nr1(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    return fma(y, fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double sqrt_nr(double x)          // x > 0, normal
{
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    const double sq = x * y;
    return fma(fma(-sq, sq, x), 0.5 * y, sq);
}
// a, b: {sin, cos of the half latitude, sin, cos of the half longitude, cos(latitude)} of the two stations
__device__ __forceinline__ double ellip_pair_vario(const double (&a)[5], const double (&b)[5])
{
    if (a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3]) return 0.0;   // same location
    const double sG = fma(a[0], b[1], -(a[1] * b[0]));
    const double sL = fma(a[2], b[3], -(a[3] * b[2]));
    const double cc = a[4] * b[4];
    const double sG2 = sG * sG, sL2 = sL * sL;
    const double Sd = fma(cc, sL2, sG2);
    if (!(Sd > 0.0)) return 0.0;
    if (!(Sd < 4e-3)) return ellip_far_f64(Sd, cc, sG2);                           // > ~800 km: never in a neighbourhood
    const double cF2 = cc + sG2, sF2 = 1.0 - cF2, cG2 = 1.0 - sG2, Cd = 1.0 - Sd;
    double P = 0.01396484375;                                                       // asin(x) / x in x^2 = S (ellip_near_f64)
    P = fma(P, Sd, 0.017352764423076924); P = fma(P, Sd, 0.022372159090909092); P = fma(P, Sd, 0.030381944444444444);
    P = fma(P, Sd, 0.044642857142857144); P = fma(P, Sd, 0.075); P = fma(P, Sd, 0.16666666666666666); P = fma(P, Sd, 1.0);
    double yc = __builtin_amdgcn_rsq(Cd);                                           // sqrt(C), C ~ 1: one Newton step (2^-46)
    yc = yc * fma(-0.5 * Cd * yc, yc, 1.5);
    const double R3 = 3.0 * (Cd * yc) * rcp_nr1(P);
    const double H1 = (R3 - 1.0) * rcp_nr1(2.0 * Cd), H2 = (R3 + 1.0) * rcp_nr1(2.0 * Sd);
    return (2 * TWX_WGS84_A) * (sqrt_nr(Sd) * P) * (1 + TWX_WGS84_F * H1 * sF2 * cG2 - TWX_WGS84_F * H2 * cF2 * sG2);
}

__device__ __forceinline__ double block_sum(double v, double *s_tmp /*[4]*/)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_tmp[0] + s_tmp[1] + s_tmp[2] + s_tmp[3];
}

__device__ __forceinline__ double block_max(double v, double *s_tmp)
{
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(s_tmp[0], s_tmp[1]), fmax(s_tmp[2], s_tmp[3]));
}

// ---------------------------------------------------------------------------------
// k_group_dist64: the semivariogram's pair distances, once per point LIST.  The points of a list -- one station's 16
// bandwidths x 12 months in step21 -- share their location, excluded station and therefore their ranked neighbour list,
// and their neighbourhoods are nested: the k (k - 1) / 2 pairs of a point are the first entries of the list's
// ksel (ksel - 1) / 2.  Evaluated per (point, pass) the distance formula was ~70 of the ~110 instructions of k_vario's pair
// loop, 64 x redundantly in step21 (profiles/README.md, round 5).  One work-group per list; pair (i, j) at i (i - 1) / 2 + j.
// ---------------------------------------------------------------------------------
// INVARIANT the sharing rests on: every point of a list has the SAME ranked neighbour list.  The host forms the lists
// (upload_points, twx_hip.hip) from everything select_cell's ranking depends on per point -- location, the excluded station,
// the further exclusion list of twx_set_exclusions --; rm_zero_dist and the station table are per call.  A new per-point
// selection input must join that key, or k_vario would pair the first point's distances with another point's residuals.
__global__ __launch_bounds__(256) void k_group_dist64(StnDev st, CellSrc src, SelWs ws)
{
    __shared__ double s_trig[5][TWX_KSEL_MAX];
    const int64_t g = blockIdx.x;
    if (g >= ws.ntile) return;
    const int64_t lc = (src.ptfirst ? src.ptfirst[g] : g) - ws.cell0;      // the list's first point stands for all of them
    if (lc < 0 || lc >= ws.ncell) return;
    const int t = threadIdx.x;
    const int nn = min(ws.nnear[lc], ws.ksel);
    if (t < nn) {
        const int j = ws.near_idx[lc * ws.ksel + t];
        const double sp = st.sph[j], cp = st.cph[j];
        s_trig[0][t] = sp; s_trig[1][t] = cp; s_trig[2][t] = st.slh[j]; s_trig[3][t] = st.clh[j];
        s_trig[4][t] = fma(cp, cp, -(sp * sp));
    }
    __syncthreads();
    const int npair = nn * (nn - 1) / 2;
    double *out = ws.gd64 + g * (int64_t)(ws.ksel * (ws.ksel - 1) / 2);
    int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)t)) * 0.5);
    while (i * (i - 1) / 2 > t) --i;
    while ((i + 1) * i / 2 <= t) ++i;
    int j = t - i * (i - 1) / 2;
    for (int p = t; p < npair; p += 256, j += 256) {
        while (j >= i) { j -= i; ++i; }
        const double ta[5] = {s_trig[0][i], s_trig[1][i], s_trig[2][i], s_trig[3][i], s_trig[4][i]};
        const double tb[5] = {s_trig[0][j], s_trig[1][j], s_trig[2][j], s_trig[3][j], s_trig[4][j]};
        out[p] = ellip_pair_vario(ta, tb);
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void k_vario(StnDev st, CellSrc src, SelWs ws)
{
    __shared__ double s_e[TWX_KSEL_MAX];
    __shared__ double s_sh[TWX_VBINS], s_sg[TWX_VBINS], s_sn[TWX_VBINS];
    __shared__ double s_red[4], s_beta[5], s_nrm[20];
    __shared__ int s_bad;
    const int t = threadIdx.x, lane = t & 63;
    const int64_t item = blockIdx.x;
    if (item >= ws.ncell * 12) return;
    const int64_t lc = item / 12;
    const int m0 = (int)(item % 12);
    const int k = ws.cstat[lc] == 0 ? ws.kk[item] : 0;
    if (k <= 0) return;                                  // uniform
    const int64_t c = ws.cell0 + lc;
    const size_t n = (size_t)st.n;
    const CellVals cv = cell_load(src, c);
    const double plst = cell_lst(src, c, m0);
    if (t == 0) s_bad = 0;

    // ---- neighbours (rank order), trend columns shifted to the point exactly as in k_uk
    double x[5] = {0, 0, 0, 0, 0}, y = 0, dh = 0;
    if (t < k) {
        const int j = ws.near_idx[lc * ws.ksel + t];
        const double lo = st.lon[j], la = st.lat[j];
        x[0] = 1.0; x[1] = lo - cv.lon; x[2] = la - cv.lat; x[3] = st.elev[j] - cv.elev; x[4] = st.lst[m0 * n + j] - plst;
        y = st.norm[m0 * n + j];
        dh = ws.near_dist[lc * ws.ksel + t];
    }
    // (unscaled, as in k_uk: the GLS beta of pass 1 is expressed in this basis)
    const double cutoff = 1.4 * block_max(dh, s_red);    // interp.R:63 (ngh_dist = haversine km)

    // ---- residuals ---------------------------------------------------------------------------------
    if (PASS == 0) {
        // lm(FORMULA): normal equations (15 + 5 sums), 5x5 Cholesky by thread 0
        int q = 0;
#pragma unroll
        for (int a = 0; a < 5; ++a) {
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const double v = block_sum(x[a] * x[b], s_red);
                if (t == 0) s_nrm[q] = v;
                ++q;
            }
        }
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const double v = block_sum(x[a] * y, s_red);
            if (t == 0) s_nrm[15 + a] = v;
        }
        __syncthreads();
        if (t == 0) {
            double L[5][5], b[5];
            bool bad = false;
            for (int i = 0; i < 5; ++i)
                for (int j = 0; j <= i; ++j) {
                    double s = s_nrm[i * (i + 1) / 2 + j];
                    for (int p = 0; p < j; ++p) s -= L[i][p] * L[j][p];
                    if (i == j) { if (!(s > 0.0)) bad = true; L[i][i] = sqrt(s); }
                    else L[i][j] = s / L[j][j];
                }
            for (int i = 0; i < 5; ++i) {
                double s = s_nrm[15 + i];
                for (int p = 0; p < i; ++p) s -= L[i][p] * b[p];
                b[i] = s / L[i][i];
            }
            for (int i = 4; i >= 0; --i) {
                double s = b[i];
                for (int p = i + 1; p < 5; ++p) s -= L[p][i] * b[p];
                b[i] = s / L[i][i];
            }
            for (int i = 0; i < 5; ++i) { s_beta[i] = b[i]; if (!finite_d(b[i])) bad = true; }
            if (bad) s_bad = 1;
        }
    } else {
        if (t < 5) s_beta[t] = ws.uk_beta[item * 5 + t];       // GLS beta from k_uk_solve
        if (t == 0 && ws.uk_S[item * TWX_UK_SLEN + 28] != 0.0) s_bad = 1;
    }
    __syncthreads();
    double e = 0.0;
    if (t < k) {
        e = y - (s_beta[0] + s_beta[1] * x[1] + s_beta[2] * x[2] + s_beta[3] * x[3] + s_beta[4] * x[4]);
        s_e[t] = e;
    }
    const double emean = block_sum(e, s_red) / k;
    const double dev = t < k ? e - emean : 0.0;
    const double sill = block_sum(dev * dev, s_red) / (k - 1);  // var(residuals), interp.R:66,85

    // ---- binned semivariogram (gstat::variogram, width 5 km) -------------------------------------------
    const double width = 5.0;
    int nb = (int)ceil(cutoff / width) + 1;
    if (nb > TWX_VBINS) nb = TWX_VBINS;
    for (int b = t; b < TWX_VBINS; b += 256) { s_sh[b] = 0.0; s_sg[b] = 0.0; s_sn[b] = 0.0; }
    __syncthreads();
    // Bins per WAVE when a quarter of the space holds them (cutoff <= 635 km: every neighbourhood of the path): the atomics of
    // one wave are applied in program order and, within an instruction, in the LDS unit's fixed lane order, and the four
    // partial sums are added in a fixed order afterwards -- the same bits run after run (shared bins took the waves' additions in
    // whatever order they arrived: fits differed in the last bits between two runs), and a quarter of the collisions.
    const int bo_step = nb <= TWX_VBINS / 4 ? TWX_VBINS / 4 : 0;
    const int bo = bo_step * (t >> 6);
    const int npair = k * (k - 1) / 2;
    const double *gd = ws.gd64 + (int64_t)(src.ptile ? src.ptile[c] : c) * (int64_t)(ws.ksel * (ws.ksel - 1) / 2);
    // pair p = i (i - 1) / 2 + j, j < i: decoded once, then stepped (256 pairs ahead is at most a few rows down: k <= 152)
    int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)t)) * 0.5);
    while (i * (i - 1) / 2 > t) --i;
    while ((i + 1) * i / 2 <= t) ++i;
    int j = t - i * (i - 1) / 2;
    for (int p = t; p < npair; p += 256, j += 256) {
        while (j >= i) { j -= i; ++i; }
        const double h = gd[p];                             // the list's pair table (k_group_dist64): coalesced, L2-resident
        if (h <= cutoff) {
            int b = (int)floor(h / width);
            if (b > 0 && h == b * width) --b;
            if (b >= nb) b = nb - 1;
            const double d = s_e[i] - s_e[j];
            atomicAdd(&s_sh[bo + b], h); atomicAdd(&s_sg[bo + b], d * d); atomicAdd(&s_sn[bo + b], 1.0);
        }
    }
    __syncthreads();
    if (bo_step) {                                           // the four waves' bins, in a fixed order
        for (int b = t; b < nb; b += 256) {
            s_sh[b] = (s_sh[b] + s_sh[bo_step + b]) + (s_sh[2 * bo_step + b] + s_sh[3 * bo_step + b]);
            s_sg[b] = (s_sg[b] + s_sg[bo_step + b]) + (s_sg[2 * bo_step + b] + s_sg[3 * bo_step + b]);
            s_sn[b] = (s_sn[b] + s_sn[bo_step + b]) + (s_sn[2 * bo_step + b] + s_sn[3 * bo_step + b]);
        }
        __syncthreads();
    }

    // ---- constrained fit by wave 0 (bins over lanes); Gauss-Newton with step halving ----------------------
    if (t < 64) {
        double gmin = INFINITY, dmax = 0.0;
        for (int b = lane; b < nb; b += 64)
            if (s_sn[b] > 0.0) {
                const double hb = s_sh[b] / s_sn[b], gb = s_sg[b] / (2.0 * s_sn[b]);
                s_sh[b] = hb; s_sg[b] = gb;                     // now: mean distance, gamma
                gmin = fmin(gmin, gb); dmax = fmax(dmax, hb);
            }
        gmin = -wave_max(-gmin); dmax = wave_max(dmax);
        double m0v = sill, m1v = 0.0, m2v = 0.0;               // pure nugget unless the fit is usable
        const double nug = gmin, psill = sill - gmin;
        double r = 0.1 * dmax;
        bool usable = (dmax > 0.0) && (psill > 0.0) && (r > 0.0) && finite_d(psill);
        if (usable) {
            auto sse_of = [&](double rr) {
                double s = 0.0;
                for (int b = lane; b < nb; b += 64)
                    if (s_sn[b] > 0.0) {
                        const double hb = s_sh[b];
                        const double mm = nug + psill * (1.0 - exp(-hb / rr));
                        const double w = s_sn[b] / (hb * hb);
                        s += w * (s_sg[b] - mm) * (s_sg[b] - mm);
                    }
                return wave_sum(s);
            };
            double sse = sse_of(r);
            for (int it = 0; it < 200; ++it) {
                double num = 0.0, den = 0.0;
                for (int b = lane; b < nb; b += 64)
                    if (s_sn[b] > 0.0) {
                        const double hb = s_sh[b];
                        const double ex = exp(-hb / r);
                        const double mm = nug + psill * (1.0 - ex);
                        const double J = -psill * ex * hb / (r * r);
                        const double w = s_sn[b] / (hb * hb);
                        num += w * J * (s_sg[b] - mm); den += w * J * J;
                    }
                num = wave_sum(num); den = wave_sum(den);
                if (!(den > 0.0) || !finite_d(num)) { usable = false; break; }
                double step = num / den, rn = r, ssen = sse;
                bool ok = false;
                for (int hh = 0; hh < 30; ++hh) {
                    rn = r + step;
                    if (rn > 0.0 && finite_d(rn)) {
                        ssen = sse_of(rn);
                        if (ssen <= sse) { ok = true; break; }
                    }
                    step *= 0.5;
                }
                if (!ok) break;
                const double impr = sse - ssen;
                r = rn; sse = ssen;
                if (impr <= 1e-10 * sse) break;
            }
            if (usable && r > 0.0 && finite_d(r)) { m0v = nug; m1v = psill; m2v = r; }
        }
        if (lane == 0) {
            const bool bad = s_bad != 0 || !finite_d(m0v) || !finite_d(m1v) || !finite_d(m2v) || !(k >= 7);
            if (bad) ws.uk_stat[lc] = TWX_CELL_NUMERIC;
            double *out = (PASS == 0) ? (ws.vario + item * 3) : (ws.vfit + item * 3);
            out[0] = m0v; out[1] = m1v; out[2] = m2v;
        }
    }
}


// ---------------------------------------------------------------------------------
// twx_krigall_points, between its two kriging stages: the model fitted for a point's month (ws.vfit, k_vario<1>) becomes the
// variogram the second stage kriges with -- on the device, with the selection, the neighbour lists and the pair distances
// of the first stage.  A point whose fit failed is done: its status stands and it gets no second system.
// ---------------------------------------------------------------------------------
__global__ void k_vfit_to_vario(CellSrc src, SelWs ws, int32_t *st1, double *vout)
{
    const int64_t lc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= ws.ncell) return;
    const int64_t c = ws.cell0 + lc;
    const int m0 = src.mth[c] - 1;
    const int64_t item = lc * 12 + m0;
    const int s = ws.uk_stat[lc] ? ws.uk_stat[lc] : ws.cstat[lc];
    st1[lc] = s;
    if (s == 0 && ws.kk[item] > 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double v = ws.vfit[item * 3 + q];
            ws.vario[item * 3 + q] = v;
            vout[lc * 3 + q] = v;
        }
    } else {
        ws.kk[item] = 0;
        vout[lc * 3] = vout[lc * 3 + 1] = vout[lc * 3 + 2] = NAN;
    }
    ws.uk_stat[lc] = TWX_CELL_OK;
}
