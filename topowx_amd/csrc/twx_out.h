// twx_out.h -- small epilogue kernels: status merge, output formatting.
#pragma once
#include "twx_select.h"

// Grid epilogue (step25:154-172): a cell is written only if BOTH requested
// variables succeeded; normals / SE cast to f4, SE = sqrt(max(var, 0))
// (KrigTair.std_err_ci, interp_tair.py:816).
// gstat_n / gstat_x: GWR status of the batch (k_gwr_z; null when no daily output is asked for).  A cell
// whose hat row cannot be formed (np.linalg.inv raises in _gwr_series, interp_tair.py:1139) is abandoned as a
// whole by the worker: normals, SE and ninvalid stay at fill, exactly like a kriging failure.
// only: null, or [ncell] flags -- the cells to (re)write (the tie guard's second pass, run_tie_guard).
__global__ void k_finalize_grid(CellSrc src, SelWs wn, SelWs wx, int has_n, int has_x,
                                const int32_t *gstat_n, const int32_t *gstat_x, twx_grid_out out, int write_ninvalid,
                                const int32_t *only)
{
    const int64_t lc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const SelWs &w0 = has_n ? wn : wx;
    if (lc >= w0.ncell || (only && !only[lc])) return;
    const int64_t c = w0.cell0 + lc;
    const int64_t yx = (int64_t)src.Y * src.X;
    if (!cell_valid(src, c)) {
        if (out.status) out.status[c] = TWX_CELL_MASKED;
        return;
    }
    int s = 0;
    // a kriging failure precedes the selection's: systems are solved only for the months the reference's loop reaches
    // before the selection failure it meets (k_select), so a singular one among them is the FIRST failure of the point
    if (has_n) s = wn.uk_stat[lc] ? wn.uk_stat[lc] : wn.cstat[lc];
    if (!s && has_x) s = wx.uk_stat[lc] ? wx.uk_stat[lc] : wx.cstat[lc];
    if (!s && has_n && gstat_n) s = gstat_n[lc];
    if (!s && has_x && gstat_x) s = gstat_x[lc];
    if (out.status) out.status[c] = s;
    if (s) return;
    for (int m = 0; m < 12; ++m) {
        if (has_n) {
            double v = wn.uk_var[lc * 12 + m];
            if (out.norm_tmin) out.norm_tmin[m * yx + c] = (float)wn.uk_mean[lc * 12 + m];
            if (out.se_tmin) out.se_tmin[m * yx + c] = (float)(v >= 0.0 ? sqrt(v) : 0.0);
        }
        if (has_x) {
            double v = wx.uk_var[lc * 12 + m];
            if (out.norm_tmax) out.norm_tmax[m * yx + c] = (float)wx.uk_mean[lc * 12 + m];
            if (out.se_tmax) out.se_tmax[m * yx + c] = (float)(v >= 0.0 ? sqrt(v) : 0.0);
        }
    }
    if (write_ninvalid && out.ninvalid) out.ninvalid[c] = 0;
}

// Point epilogue of twx_krig_points (one month per point).
__global__ void k_finalize_krig_points(CellSrc src, SelWs ws, double *mean, double *var,
                                       int32_t *nnghs_used, int32_t *status)
{
    const int64_t lc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= ws.ncell) return;
    const int64_t c = ws.cell0 + lc;
    const int m0 = src.mth[c] - 1;
    int s = ws.uk_stat[lc] ? ws.uk_stat[lc] : ws.cstat[lc];      // (see k_finalize_grid)
    status[c] = s;
    if (nnghs_used) nnghs_used[c] = s ? 0 : ws.kk[lc * 12 + m0];
    if (s) return;
    mean[c] = ws.uk_mean[lc * 12 + m0];
    var[c] = ws.uk_var[lc * 12 + m0];
}

// Point epilogue of twx_interp_points (normals part).
__global__ void k_finalize_interp_points(SelWs ws, double *norms, double *se, int32_t *status)
{
    const int64_t lc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= ws.ncell) return;
    const int64_t c = ws.cell0 + lc;
    int s = ws.uk_stat[lc] ? ws.uk_stat[lc] : ws.cstat[lc];      // (see k_finalize_grid)
    status[c] = s;
    if (s) return;
    for (int m = 0; m < 12; ++m) {
        double v = ws.uk_var[lc * 12 + m];
        norms[c * 12 + m] = ws.uk_mean[lc * 12 + m];
        se[c * 12 + m] = v >= 0.0 ? sqrt(v) : 0.0;
    }
}

// StationSelect.set_ngh_stns outputs (station_select.py:164-192): the k nearest
// re-ordered by ascending station index, bisquare weights with the (k+1)-th
// distance as bandwidth.  One wavefront per point.  kfix > 0: same k for all
// points (twx_knn); kfix == 0: k = kk of the point's month (neighbours of krig).
__global__ __launch_bounds__(256) void k_sorted_neighbours(CellSrc src, SelWs ws, int kfix, int ld,
                                                           int32_t *idx, double *dist, double *wgt,
                                                           int32_t *status)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t lc = (int64_t)blockIdx.x * 4 + wv;
    if (lc >= ws.ncell) return;
    const int64_t c = ws.cell0 + lc;
    int k = kfix;
    if (kfix == 0) {
        int m0 = src.mth[c] - 1;
        k = (ws.cstat[lc] == 0) ? ws.kk[lc * 12 + m0] : 0;
    }
    const int nnear = ws.nnear[lc];
    int s = TWX_CELL_OK;
    if (kfix > 0) {
        if (k >= nnear) s = TWX_CELL_FEW_STATIONS;
        else if (!(ws.near_dist[lc * ws.ksel + k] > 0.0)) s = TWX_CELL_NUMERIC;
        if (lane == 0 && status) status[c] = s;
    }
    const int kfill = (s || k <= 0) ? 0 : k;      // slots [kfill, ld) are padding
    for (int r = kfill + lane; r < ld; r += 64) {
        idx[c * ld + r] = -1;
        if (dist) dist[c * ld + r] = 0.0;
        if (wgt) wgt[c * ld + r] = 0.0;
    }
    if (s || k <= 0) return;
    const double dbw = ws.near_dist[lc * ws.ksel + k];
    const int32_t *ni = ws.near_idx + lc * ws.ksel;
    for (int r = lane; r < k; r += 64) {
        int me = ni[r], pos = 0;
        for (int i = 0; i < k; ++i) pos += ni[i] < me;
        double d = ws.near_dist[lc * ws.ksel + r];
        idx[c * ld + pos] = me;
        if (dist) dist[c * ld + pos] = d;
        if (wgt) wgt[c * ld + pos] = bisq(d, dbw);
    }
}
