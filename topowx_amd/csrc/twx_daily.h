// twx_daily.h -- GWR hat rows, daily anomaly dot products, Tmin>=Tmax fixer and
// int16 packing.
#pragma once
#include "twx_select.h"

#define TWX_KZ 160   // z slots per (cell, month)

struct DayAxis {
    int ndays;
    int moff[13];             // month-major offsets: days of month m are [moff[m-1], moff[m])
    const int32_t *mm2chron;  // [ndays] month-major position -> chronological day
    const int32_t *chron2mm;  // [ndays]
    const int32_t *day_month; // [ndays] chronological, 1..12
    const int32_t *day_year;  // [ndays]
    int tail;
    int norm_ny;              // years of the normals period present in the day axis (0 = none)
    int norm_y0;              // first of them
    const int32_t *ym_start;  // [norm_ny * 12] first chronological day of (year, month)
    const int32_t *ym_cnt;    // [norm_ny * 12] days of (year, month)
};

#define TWX_UROWS 208 // observation rows of a (tile, month) staged in LDS by k_daily_tile (x 64 days x 4 B = 52 KB: three work-groups per CU)
#ifndef TWX_DT_WAVES
#define TWX_DT_WAVES 8                           // waves per work-group of k_daily_tile (see there)
#endif
#define TWX_DT_CPW (64 / TWX_DT_WAVES)             // cells per wave
// position of (tile-month tm, cell ci of the tile, table row u) in GwrWs.zd
__host__ __device__ __forceinline__ int64_t twx_zd_index(int64_t tm, int ci)
{
    return ((tm * TWX_DT_WAVES + ci / TWX_DT_CPW) * (TWX_UROWS / 16) * TWX_DT_CPW + ci % TWX_DT_CPW) * 16;
}
__host__ __device__ __forceinline__ int twx_zd_row(int u) { return (u >> 4) * (TWX_DT_CPW * 16) + (u & 15); }

struct GwrWs {
    double *z;        // [ncell][12][TWX_KZ]  hat row by neighbour rank (point mode, and the tile-months of grid mode that do
                      // not go through a table: see zd)
    double *zc;       // [ncell][12]          pt_norm - sum_j z_j norm_j
    double *zn;       // [ncell][12]          sum_j z_j norm_j itself (grid mode: k_tie_rezc re-forms zc from a re-kriged normal)
    int32_t *gstat;   // [ncell]
    uint32_t *noff;   // [ncell][ksel] byte offset of each ranked neighbour's observation row (k_row_offsets)
    // EVERY daily sum runs in ONE order: ascending station index of the cell's neighbours -- the order of the reference
    // (station_select.py:179-182 sorts the neighbours by station id; the database is id-sorted) and of a table walk
    // (table rows are numbered in candidate-list order = ascending station index; a row the cell does not use carries
    // weight 0 and fma(0, x, acc) = acc exactly).  Table path, gather paths, fixer and point entries therefore agree bit
    // for bit: which path a tile-month takes does not show in the packed int16 output.
    int32_t *perm;    // [ncell][ksel] the ranks 0 .. kp - 1 of a cell in ascending station-index order (k_perm)
    int32_t *kp;      // [ncell] largest GWR bandwidth of the cell's months = entries of perm
    // k_tile_uidx -> k_gwr_z_cell -> k_daily_tile: the stations the cells of an 8x8 tile use in a month, as rows of an LDS table
    int32_t *urow;    // [ntile][12][TWX_UROWS] station index of table row u
    int32_t *nurow;   // [ntile][12] rows in the table; -1 = more than TWX_UROWS (the tile-month gathers from global memory)
    uint16_t *uslot;  // [ntile][12][cmax] table row of the candidate at list position p (valid where a cell uses it)
    double *zd;       // [ntile][12][64 / CPW groups][14 chunks][CPW cells][16] the hat rows in table-row order (0 for
                      // rows a cell does not use), laid out as k_daily_tile's waves read them: one wave = one group of CPW
                      // cells, one chunk of 16 table rows at a time = CPW x 128 contiguous bytes (twx_zd_index).  Written
                      // ONCE, by k_gwr_z_cell (scattered in LDS, stored coalesced): a hat row of a table tile-month never
                      // exists in rank order in HBM
    const int32_t *nurow2;  // the OTHER variable's nurow (table mode): a (tile, month) uses the tables only when both unions fit
    int use_table;    // grid mode, both variables, 8x8 tiles, no gather flag: tile-months with both nurow >= 0 use zd, not z
};

// ---------------------------------------------------------------------------------
// k_gwr_z: one 16-lane DPP row per (cell, month) item, four items per wavefront.  _gwr_series
// (interp_tair.py:1099-1146): z = x0' (X'WX)^-1 X'W with X = [1, lon, lat, elev, tdi, lst] of the ka nearest
// stations and their own bisquare weights (:1128-1140).
//
// Round 2 gave an item a whole wave and paid 32 full-wave reductions for it (3 % of the fp64 rate: reduction
// latency, not arithmetic).  Here a lane walks the item's neighbours tr, tr + 16, ... and keeps the 21 sums of
// the normal matrix in registers; they are reduced ONCE over the 16 lanes of the row (row_shr steps, the total of
// lane 15 broadcast back with row_newbcast: no cross-row traffic, every lane of a row holds the same bits), the 6x6
// SPD system is Cholesky-solved redundantly by every lane (four items at a time), and a second walk over the
// neighbours forms the hat row.  Predictor columns are shifted to the cell (cancellation) and left unscaled: z is
// invariant to a column scaling and an unpivoted Cholesky only changes its roundings under one (see k_uk).
// ---------------------------------------------------------------------------------
#define TWX_GZ_SLOTS ((TWX_MAX_NNGHS + 15) / 16)   // neighbours per lane

// 1 / sqrt(d): hardware seed + two Newton steps (relative error ~1e-16; NaN / inf for d <= 0 as sqrt + division gave)
__device__ __forceinline__ double gz_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// sum over the 16 lanes of a DPP row, returned in all of them
__device__ __forceinline__ double row16_sum(double v)
{
    v = dpp_add_step<0x111, 0xf>(v);           // row_shr:1
    v = dpp_add_step<0x112, 0xf>(v);           // row_shr:2
    v = dpp_add_step<0x114, 0xf>(v);           // row_shr:4
    v = dpp_add_step<0x118, 0xf>(v);           // row_shr:8 -> lane 15 of the row holds the row's sum
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x15f, 0xf, 0xf, false);   // row_newbcast:15
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x15f, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void k_gwr_z(StnDev st, CellSrc src, SelWs ws, GwrWs gw,
                                               const double *pt_norm_in)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, tr = lane & 15, row = lane >> 4;
    const int64_t nitems = ws.ncell * 12;
    const int wgi = xcd_contig(blockIdx.x, (int)((nitems + 15) / 16));
    if (wgi < 0) return;
    const int64_t item = (int64_t)wgi * 16 + wv * 4 + row;   // the four items of a wave: mostly months of one cell
    const bool in_range = item < nitems;
    const int64_t lc = in_range ? item / 12 : 0;
    const int m0 = in_range ? (int)(item % 12) : 0;
    const int64_t c = ws.cell0 + lc;
    int ka = 0;
    if (in_range && ws.cstat[lc] == 0 && ws.uk_stat[lc] == 0) ka = ws.ka[lc * 12 + m0];
    if (!__any(ka > 0)) return;
    const size_t n = (size_t)st.n;
    const CellVals cv = cell_load(src, c);
    const double plst = cell_lst(src, c, m0);
    // (one reciprocal per item instead of one fp64 division per neighbour: z moves in its last bit, every consumer
    // reads the stored z)
    const double inv_dbw = 1.0 / (ka > 0 ? ws.near_dist[lc * ws.ksel + ka] : 1.0);
    const int32_t *ni = ws.near_idx + lc * ws.ksel;
    const double *nd = ws.near_dist + lc * ws.ksel;
    int kamax = ka;                                          // slots this wave walks (uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kamax = max(kamax, __shfl_xor(kamax, o, 64));
    const int nslot = (kamax + 15) >> 4;

    // ---- pass 1: M = X'WX (lower triangle, 21 sums) over this lane's neighbours ------------------------------
    double M[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) M[a][b] = 0.0;
    double w[TWX_GZ_SLOTS];
    bool nz[5] = {false, false, false, false, false};        // a predictor that is non-zero at some neighbour
#pragma unroll
    for (int s = 0; s < TWX_GZ_SLOTS; ++s) {
        w[s] = 0.0;
        if (s < nslot) {                                     // uniform
            const int r = tr + 16 * s;
            if (r < ka) {
                const int j = ni[r];
                const double wj = bisq_r(nd[r], inv_dbw);
                w[s] = wj;
                const double raw[5] = {st.lon[j], st.lat[j], st.elev[j], st.tdi[j], st.lst[m0 * n + j]};
                const double x[6] = {1.0, raw[0] - cv.lon, raw[1] - cv.lat, raw[2] - cv.elev, raw[3] - cv.tdi, raw[4] - plst};
#pragma unroll
                for (int q = 0; q < 5; ++q) nz[q] = nz[q] || raw[q] != 0.0;
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double wx = wj * x[a];
#pragma unroll
                    for (int b = 0; b <= a; ++b) M[a][b] = fma(wx, x[b], M[a][b]);
                }
            }
        }
    }
    bool bad = false;
    // a predictor that is exactly zero at every neighbour (e.g. TDI on flat terrain) gives X'WX an exactly zero row:
    // np.linalg.inv raises (interp_tair.py:1139) -- with the columns shifted to the cell that case would otherwise
    // only be collinear, which rounding can hide from the Cholesky pivots below.  (ballots, restricted to the row)
    const unsigned long long rowmask = 0xffffull << (16 * row);
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (!(__ballot(nz[q]) & rowmask)) bad = true;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) M[a][b] = row16_sum(M[a][b]);
    // Cholesky with one reciprocal per pivot (33 fp64 divisions -> 6: a division is a dozen instructions; the hat
    // row changes in the last bit only, all consumers use the same stored z)
    double inv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = M[i][j];
#pragma unroll
            for (int p = 0; p < j; ++p) s -= M[i][p] * M[j][p];
            // pivot: 1 / sqrt(s) from v_rsq_f64 + two Newton steps (full precision; a square root and a division are
            // ~45 instructions), sqrt(s) = s * rsqrt(s)
            if (i == j) { if (!(s > 0.0)) bad = true; inv[i] = gz_rsqrt(s); M[i][i] = s * inv[i]; }
            else M[i][j] = s * inv[j];
        }
    }
    // a = M^-1 e1
    double a[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = (i == 0) ? 1.0 : 0.0;
#pragma unroll
        for (int p = 0; p < i; ++p) s -= M[i][p] * a[p];
        a[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = a[i];
#pragma unroll
        for (int p = i + 1; p < 6; ++p) s -= M[p][i] * a[p];
        a[i] = s * inv[i];
    }
    // ---- pass 2: hat row z_j = w_j a'x_j (the neighbours' predictors come back from the caches) -----------------
    double zn = 0.0;
    double *zout = gw.z + (lc * 12 + m0) * TWX_KZ;
#pragma unroll
    for (int s = 0; s < TWX_GZ_SLOTS; ++s) {
        if (s < nslot) {
            const int r = tr + 16 * s;
            if (r < ka) {
                const int j = ni[r];
                double t = a[0];
                t = fma(a[1], st.lon[j] - cv.lon, t); t = fma(a[2], st.lat[j] - cv.lat, t);
                t = fma(a[3], st.elev[j] - cv.elev, t); t = fma(a[4], st.tdi[j] - cv.tdi, t);
                t = fma(a[5], st.lst[m0 * n + j] - plst, t);
                const double z = w[s] * t;
                zout[r] = z;
                zn = fma(z, st.norm[m0 * n + j], zn);
                if (!finite_d(z)) bad = true;
            }
        }
    }
    zn = row16_sum(zn);
    bad = (__ballot(bad) & rowmask) != 0;
    if (tr == 0 && ka > 0) {
        const double pn = pt_norm_in ? pt_norm_in[c] : ws.uk_mean[lc * 12 + m0];
        gw.zc[lc * 12 + m0] = pn - zn;
        if (bad) gw.gstat[lc] = TWX_CELL_NUMERIC;
    }
}

// ---------------------------------------------------------------------------------
// k_gwr_z_cell (grid mode): the same hat rows, one work-group of 12 DPP rows (3 waves) per CELL, row = month.
// k_gwr_z is bound by its gathers (14 scattered 8-byte loads per neighbour and month: the station columns of both
// passes); the twelve GWR neighbourhoods of a cell are nested, so here the month-independent columns (lon, lat, elev,
// tdi, shifted to the cell) and the distances are staged ONCE per cell in LDS and only lst / norm of the row's month
// are gathered (2 per neighbour and month; lst waits in registers for the second pass).  Sum_j z_j norm_j is formed
// as a'(X'W norm) from six more sums of pass 1, so pass 2 reads no station column at all.
// Same arithmetic per sum as k_gwr_z (lane = neighbour r mod 16, row_shr reduction) -- the two kernels give the same
// z bit for bit; zc differs in the last bits (a'v instead of sum z_j norm_j).
// With GwrWs.use_table the hat rows of a tile-month that has a table leave this kernel in table-row order (GwrWs.zd).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(192) void k_gwr_z_cell(StnDev st, CellSrc src, SelWs ws, GwrWs gw)
{
    __shared__ double s_x[TWX_KSEL_MAX][4];
    __shared__ double s_d[TWX_KSEL_MAX];
    __shared__ int s_j[TWX_KSEL_MAX];
    __shared__ uint16_t s_pos[TWX_KSEL_MAX];                 // position of the ranked neighbour in the tile's candidate list
    __shared__ double s_z[12][TWX_UROWS];                    // the twelve hat rows in table-row order (use_table)
    __shared__ int s_first[4];                               // lowest rank at which static column q is non-zero
    const int t = threadIdx.x, tr = t & 15, m0 = t >> 4, lane = t & 63, row = lane >> 4;
    const int lci = xcd_contig(blockIdx.x, (int)ws.ncell);
    if (lci < 0) return;
    const int64_t lc = lci;
    if (ws.cstat[lc] != 0 || ws.uk_stat[lc] != 0) return;   // (uniform)
    const int64_t c = ws.cell0 + lc;
    int kamax = 0;
#pragma unroll
    for (int m = 0; m < 12; ++m) kamax = max(kamax, ws.ka[lc * 12 + m]);
    if (kamax <= 0) return;
    const int ka = ws.ka[lc * 12 + m0];
    const size_t n = (size_t)st.n;
    const CellVals cv = cell_load(src, c);
    const double plst = cell_lst(src, c, m0);
    if (t < 4) s_first[t] = 0x7fffffff;
    __syncthreads();
    for (int r = t; r <= kamax && r < ws.ksel; r += 192) {   // (rank kamax carries the bandwidth distance of the largest month)
        const int j = ws.near_idx[lc * ws.ksel + r];
        s_j[r] = j;
        s_d[r] = ws.near_dist[lc * ws.ksel + r];
        if (gw.use_table) s_pos[r] = ws.near_pos[lc * ws.ksel + r];
        if (r < kamax) {
            const double4 sr = st.stat_s[j];
            const double raw[4] = {sr.x, sr.y, sr.z, sr.w};
            s_x[r][0] = raw[0] - cv.lon; s_x[r][1] = raw[1] - cv.lat; s_x[r][2] = raw[2] - cv.elev; s_x[r][3] = raw[3] - cv.tdi;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (raw[q] != 0.0) atomicMin(&s_first[q], r);
        }
    }
    __syncthreads();
    const double inv_dbw = 1.0 / (ka > 0 ? s_d[ka] : 1.0);
    const int nslot = (kamax + 15) >> 4;                     // (uniform: every row walks the largest month's slots)

    double M[6][6], v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        v[a] = 0.0;
#pragma unroll
        for (int b = 0; b <= a; ++b) M[a][b] = 0.0;
    }
    double w[TWX_GZ_SLOTS], xl[TWX_GZ_SLOTS];
    bool nzl = false;
#pragma unroll
    for (int s = 0; s < TWX_GZ_SLOTS; ++s) {
        w[s] = 0.0; xl[s] = 0.0;
        if (s < nslot) {
            const int r = tr + 16 * s;
            if (r < ka) {
                const int j = s_j[r];
                const double2 mr = st.mon_s[(size_t)j * 12 + m0];      // (lst, norm): one 16-byte load
                const double lraw = mr.x, nrm = mr.y;
                const double wj = bisq_r(s_d[r], inv_dbw);
                w[s] = wj;
                xl[s] = lraw - plst;
                nzl = nzl || lraw != 0.0;
                const double x[6] = {1.0, s_x[r][0], s_x[r][1], s_x[r][2], s_x[r][3], xl[s]};
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    const double wx = wj * x[a];
                    v[a] = fma(wx, nrm, v[a]);
#pragma unroll
                    for (int b = 0; b <= a; ++b) M[a][b] = fma(wx, x[b], M[a][b]);
                }
            }
        }
    }
    bool bad = false;
    const unsigned long long rowmask = 0xffffull << (16 * row);
    // an exactly-zero predictor column (see k_gwr_z): static columns by the first rank that is non-zero, lst per row
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (!(s_first[q] < ka)) bad = true;
    if (!(__ballot(nzl) & rowmask)) bad = true;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        v[a] = row16_sum(v[a]);
#pragma unroll
        for (int b = 0; b <= a; ++b) M[a][b] = row16_sum(M[a][b]);
    }
    double inv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = M[i][j];
#pragma unroll
            for (int p = 0; p < j; ++p) s -= M[i][p] * M[j][p];
            // pivot: 1 / sqrt(s) from v_rsq_f64 + two Newton steps (full precision; a square root and a division are
            // ~45 instructions), sqrt(s) = s * rsqrt(s)
            if (i == j) { if (!(s > 0.0)) bad = true; inv[i] = gz_rsqrt(s); M[i][i] = s * inv[i]; }
            else M[i][j] = s * inv[j];
        }
    }
    double a[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = (i == 0) ? 1.0 : 0.0;
#pragma unroll
        for (int p = 0; p < i; ++p) s -= M[i][p] * a[p];
        a[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = a[i];
#pragma unroll
        for (int p = i + 1; p < 6; ++p) s -= M[p][i] * a[p];
        a[i] = s * inv[i];
    }
    // Where the hat row goes: a tile-month with a table (k_tile_uidx: nurow >= 0) gets it in TABLE-ROW order --
    // scattered in LDS (zeros where the cell does not use a row, up to the next multiple of 16 rows), then stored once,
    // coalesced, in the layout k_daily_tile's waves read (GwrWs.zd); any other in rank order (gw.z).
    double *zout = gw.z + (lc * 12 + m0) * TWX_KZ;
    int nu = -1;
    int64_t tm = 0;
    int ci = 0;
    if (gw.use_table) {
        const int rr = (int)(c / src.X), qq = (int)(c % src.X);
        const int64_t tl = (int64_t)(rr / src.ts) * src.ntx + (qq / src.ts) - ws.tile0;
        tm = tl * 12 + m0;
        ci = (rr % src.ts) * src.ts + (qq % src.ts);
        nu = gw.nurow2[tm] >= 0 ? gw.nurow[tm] : -1;
    }
    const bool tab = nu >= 0;                                // (uniform within the 16-lane row of a month)
    const int nu16 = (nu + 15) & ~15;
    const uint16_t *uslot = gw.uslot + tm * (int64_t)ws.cmax;
    if (tab)
        for (int u = tr; u < nu16; u += 16) s_z[m0][u] = 0.0;
    __builtin_amdgcn_wave_barrier();                         // (LDS operations of one wave execute in order)
#pragma unroll
    for (int s = 0; s < TWX_GZ_SLOTS; ++s) {
        if (s < nslot) {
            const int r = tr + 16 * s;
            if (r < ka) {
                double tt = a[0];
                tt = fma(a[1], s_x[r][0], tt); tt = fma(a[2], s_x[r][1], tt);
                tt = fma(a[3], s_x[r][2], tt); tt = fma(a[4], s_x[r][3], tt);
                tt = fma(a[5], xl[s], tt);
                const double z = w[s] * tt;
                if (tab) s_z[m0][uslot[s_pos[r]]] = z;
                else zout[r] = z;
                if (!finite_d(z)) bad = true;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (tab && ka > 0) {
        double *zd = gw.zd + twx_zd_index(tm, ci);
        for (int u = tr; u < nu16; u += 16) zd[twx_zd_row(u)] = s_z[m0][u];
    }
    double zn = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) zn = fma(a[q], v[q], zn);
    if (!finite_d(zn)) bad = true;
    bad = (__ballot(bad) & rowmask) != 0;
    if (tr == 0 && ka > 0) {
        gw.zc[lc * 12 + m0] = ws.uk_mean[lc * 12 + m0] - zn;
        gw.zn[lc * 12 + m0] = zn;
        if (bad) gw.gstat[lc] = TWX_CELL_NUMERIC;
    }
}

// value of one (cell, month-major day): sum_j z_j * obs[j][dm] + zc over the cell's neighbours in ASCENDING STATION
// INDEX order (GwrWs.perm; see GwrWs), one fma chain: every kernel reproduces it bit for bit.  A rank outside the
// month's bandwidth takes weight 0 (its observation is finite, so the term adds exactly nothing).
__device__ __forceinline__ double daily_value(const StnDev &st, const SelWs &ws, const GwrWs &gw,
                                              int64_t lc, int m0, int ka, int ndays, int dm)
{
    const double *z = gw.z + (lc * 12 + m0) * TWX_KZ;
    const int32_t *ni = ws.near_idx + lc * ws.ksel;
    const int32_t *pm = gw.perm + lc * ws.ksel;
    const int kp = gw.kp[lc];
    double acc = 0.0;
#pragma unroll 4
    for (int q = 0; q < kp; ++q) {
        const int r = pm[q];
        const double zr = r < ka ? z[r] : 0.0;
        acc = fma(zr, (double)st.obs[(size_t)ni[r] * ndays + dm], acc);
    }
    return acc + gw.zc[lc * 12 + m0];
}

// Tmin and Tmax of one (cell, month-major day) together (the gather kernels: cell wave-uniform, so perm, hat rows and
// row offsets come through scalar loads; lanes = days).  Same order, same bits as daily_value.  When the observation
// matrix is smaller than 4 GiB (OFF32) the element address is base + a 32-bit byte offset: the neighbours' row
// offsets are precomputed per cell (k_row_offsets), so a step is one add, the load, one convert and the fma.
template <bool OFF32>
__device__ __forceinline__ void daily_value2(const StnDev &sn, const SelWs &wn, const GwrWs &gn, const StnDev &sx,
                                             const SelWs &wx, const GwrWs &gx, int64_t lc, int m0, int ndays, int dm,
                                             double &vn, double &vx)
{
    const int kn = wn.ka[lc * 12 + m0], kx = wx.ka[lc * 12 + m0];
    const double *zn = gn.z + (lc * 12 + m0) * TWX_KZ, *zx = gx.z + (lc * 12 + m0) * TWX_KZ;
    const int32_t *nn = wn.near_idx + lc * wn.ksel, *nx = wx.near_idx + lc * wx.ksel;
    const int32_t *pn = gn.perm + lc * wn.ksel, *px = gx.perm + lc * wx.ksel;
    const int kpn = gn.kp[lc], kpx = gx.kp[lc];
    const float *on = sn.obs + dm, *ox = sx.obs + dm;
    const char *bn = reinterpret_cast<const char *>(sn.obs), *bx = reinterpret_cast<const char *>(sx.obs);
    const uint32_t *fn = gn.noff + lc * wn.ksel, *fx = gx.noff + lc * wx.ksel;   // byte offsets of the neighbours' rows
    const unsigned dm4 = 4u * (unsigned)dm;
    auto ldn = [&](int r) __attribute__((always_inline)) {
        if (OFF32) return *reinterpret_cast<const float *>(bn + (fn[r] + dm4));
        return on[(size_t)nn[r] * ndays];
    };
    auto ldx = [&](int r) __attribute__((always_inline)) {
        if (OFF32) return *reinterpret_cast<const float *>(bx + (fx[r] + dm4));
        return ox[(size_t)nx[r] * ndays];
    };
    double an = 0.0, ax = 0.0;
    const int kc = min(kpn, kpx);
    int q = 0;
#pragma unroll 4
    for (; q < kc; ++q) {                                    // the two chains are independent: twice the loads in flight
        const int rn = pn[q], rx = px[q];
        an = fma(rn < kn ? zn[rn] : 0.0, (double)ldn(rn), an);
        ax = fma(rx < kx ? zx[rx] : 0.0, (double)ldx(rx), ax);
    }
#pragma unroll 4
    for (int t = q; t < kpn; ++t) { const int r = pn[t]; an = fma(r < kn ? zn[r] : 0.0, (double)ldn(r), an); }
#pragma unroll 4
    for (int t = q; t < kpx; ++t) { const int r = px[t]; ax = fma(r < kx ? zx[r] : 0.0, (double)ldx(r), ax); }
    vn = an + gn.zc[lc * 12 + m0];
    vx = ax + gx.zc[lc * 12 + m0];
}

// step25:163-164: np.round(x, 2) / np.float32(0.01) assigned into int16, i.e. with n = rint(x * 100) (np.round) and
// c = (double)float32(0.01) = 0.009999999776482582:  trunc( fl( fl(n / 100) / c ) ).
// For every integer |n| <= 70 000 (any temperature the int16 product can hold, and twice that) this IS n: c is smaller
// than 0.01 by 2.2e-8 relative, so the quotient exceeds |n| by |n| * 2.2e-8 -- far more than the two roundings
// (|n| * 2.2e-16) can take back and never as much as 1 -- and the truncation returns n (n = 0 trivially).  Checked
// exhaustively over that range against the two-division form (tests/test_oracle_golden.py::test_pack_identity, and the
// oracle keeps the literal form); the two fp64 divisions per value were ~15 % of k_daily_tile's instructions.
__device__ __forceinline__ int16_t pack_i16(double x)
{
    return (int16_t)(int)rint(x * 100.0);
}

// ---------------------------------------------------------------------------------
// k_daily_points: double degC output for point entries.  One workgroup per point,
// threads over month-major days.  single_month != 0: out[c][ld] holds only the days
// of the point's month (chronological within the month == month-major order).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_daily_points(StnDev st, CellSrc src, SelWs ws, GwrWs gw, DayAxis da,
                                                      double *out, int64_t ld, int single_month)
{
    const int64_t lc = blockIdx.x;
    if (lc >= ws.ncell) return;
    const int64_t c = ws.cell0 + lc;
    if (ws.cstat[lc] != 0 || ws.uk_stat[lc] != 0 || gw.gstat[lc] != 0) return;
    for (int m0 = 0; m0 < 12; ++m0) {
        const int ka = ws.ka[lc * 12 + m0];
        if (ka <= 0) continue;
        for (int dm = da.moff[m0] + threadIdx.x; dm < da.moff[m0 + 1]; dm += blockDim.x) {
            double v = daily_value(st, ws, gw, lc, m0, ka, da.ndays, dm);
            if (single_month) out[c * ld + (dm - da.moff[m0])] = v;
            else out[c * ld + da.mm2chron[dm]] = v;
        }
    }
}

// ---------------------------------------------------------------------------------
// k_xval_stats: XvalTairAnom.run_xval's statistics (optimize.py:521-541) for one (station, bandwidth, month)
// point per workgroup, without moving the series to the host: interp = GWR series of the month (the point is
// the left-out station itself), truth = that station's own observations.
//   xval_anom = obs - norm, interp_anom = interp - norm, difs = interp_anom - xval_anom
//   bias = mean(difs), mae = mean|difs|, r2 = corr(interp_anom, xval_anom)^2   (stats.linregress: centred sums)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double xv_block_sum(double v, double *s_red /*[4]*/)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(256) void k_xval_stats(StnDev st, CellSrc src, SelWs ws, GwrWs gw, DayAxis da,
                                                    const double *pt_norm, const int32_t *obs_idx, double *bias,
                                                    double *mae, double *r2)
{
    __shared__ double s_red[4];
    const int64_t lc = blockIdx.x;
    if (lc >= ws.ncell) return;
    const int64_t c = ws.cell0 + lc;
    if (ws.cstat[lc] != 0 || gw.gstat[lc] != 0) return;
    const int m0 = src.mth[c] - 1;
    const int ka = ws.ka[lc * 12 + m0];
    if (ka <= 0) return;
    const double nrm = pt_norm[c];
    const float *truth = st.obs + (size_t)obs_idx[c] * da.ndays;
    const int d0 = da.moff[m0], nd = da.moff[m0 + 1] - d0;
    // pass 1: means
    double sx = 0, sy = 0, sd = 0, sa = 0;
    for (int q = threadIdx.x; q < nd; q += 256) {
        const double ia = daily_value(st, ws, gw, lc, m0, ka, da.ndays, d0 + q) - nrm;
        const double xa = (double)truth[d0 + q] - nrm;
        const double dif = ia - xa;
        sx += ia; sy += xa; sd += dif; sa += fabs(dif);
    }
    const double n = (double)nd;
    const double mx = xv_block_sum(sx, s_red) / n, my = xv_block_sum(sy, s_red) / n;
    const double b = xv_block_sum(sd, s_red) / n, a = xv_block_sum(sa, s_red) / n;
    // pass 2: centred second moments
    double sxx = 0, syy = 0, sxy = 0;
    for (int q = threadIdx.x; q < nd; q += 256) {
        const double ia = daily_value(st, ws, gw, lc, m0, ka, da.ndays, d0 + q) - nrm - mx;
        const double xa = (double)truth[d0 + q] - nrm - my;
        sxx += ia * ia; syy += xa * xa; sxy += ia * xa;
    }
    sxx = xv_block_sum(sxx, s_red); syy = xv_block_sum(syy, s_red); sxy = xv_block_sum(sxy, s_red);
    if (threadIdx.x == 0) {
        bias[c] = b; mae[c] = a;
        const double r = sxy / sqrt(sxx * syy);
        r2[c] = r * r;
    }
}

// byte offset of every ranked neighbour's observation row (valid when the matrix is smaller than 4 GiB)
__global__ void k_row_offsets(SelWs ws, GwrWs gw, int ndays)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ws.ncell * ws.ksel) return;
    const int j = ws.near_idx[i];
    gw.noff[i] = j < 0 ? 0u : (uint32_t)j * (uint32_t)ndays * 4u;
}

// k_perm: one wave per cell.  perm = the ranks 0 .. kp - 1 (kp = the largest GWR bandwidth of the cell's months) in
// ascending station-index order -- the one order every daily sum runs in (GwrWs).  Rank by counting: position of rank
// r = number of ranks with a smaller station index (indices are distinct).
__global__ __launch_bounds__(256) void k_perm(SelWs ws, GwrWs gw)
{
    __shared__ int s_idx[4][TWX_KSEL_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t lc = (int64_t)blockIdx.x * 4 + wv;
    if (lc >= ws.ncell) return;
    int kp = 0;
    if (ws.cstat[lc] == 0) {
#pragma unroll
        for (int m = 0; m < 12; ++m) kp = max(kp, ws.ka[lc * 12 + m]);
    }
    kp = min(kp, ws.ksel);
    if (lane == 0) gw.kp[lc] = kp;
    int *idx = s_idx[wv];
    for (int r = lane; r < kp; r += 64) idx[r] = ws.near_idx[lc * ws.ksel + r];
    __builtin_amdgcn_wave_barrier();
    for (int r = lane; r < kp; r += 64) {
        const int mine = idx[r];
        int pos = 0;
        for (int q = 0; q < kp; ++q) pos += idx[q] < mine ? 1 : 0;
        gw.perm[lc * ws.ksel + pos] = r;
    }
}

// ---------------------------------------------------------------------------------
// k_tile_uidx: one work-group per (8x8-cell tile, month), BEFORE the hat rows are computed.  The GWR neighbourhoods of
// a tile's cells overlap almost completely: mark the candidates (positions in the tile's candidate list, k_select)
// that any cell of the tile uses this month and number them in list order (= ascending station index) -- the rows of
// the tile-month's table of 64-day f4 observation rows that k_daily_tile stages in LDS.  Outputs: urow (station of a
// row), nurow, and uslot (row of a candidate position), which k_gwr_z_cell uses to deliver every hat row directly
// in table order.  (Rounds 2-3: k_tile_union ran after the hat rows, re-read them in rank order (0.43 GB per variable
// and 10-year step) and wrote them again in table order.)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_uidx(CellSrc src, SelWs ws, GwrWs gw)
{
    __shared__ uint16_t s_slot[TWX_CAND_MAX];
    __shared__ int s_cnt[4], s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t tl = blockIdx.x / 12;                  // local tile
    const int m0 = blockIdx.x % 12;
    const int64_t tile = ws.tile0 + tl;
    const int ty = (int)(tile / src.ntx), tx = (int)(tile % src.ntx);
    const int r0 = ty * src.ts, q0 = tx * src.ts;
    const int ncand = min(ws.ncand[tl], ws.cmax);
    for (int p = t; p < ncand; p += 256) s_slot[p] = 0;
    if (t == 0) s_base = 0;
    __syncthreads();
    // mark: a wave per cell, lanes over ranks (a cell whose kriging / GWR fails later only adds rows nobody reads)
    const int ncl = src.ts * src.ts;                     // cells per tile (<= 64)
    for (int ci = wv; ci < ncl; ci += 4) {
        const int rr = r0 + ci / src.ts, qq = q0 + ci % src.ts;
        if (rr >= src.Y || qq >= src.X) continue;
        const int64_t lc = (int64_t)rr * src.X + qq - ws.cell0;
        if (lc < 0 || lc >= ws.ncell || ws.cstat[lc] != 0) continue;
        const int ka = ws.ka[lc * 12 + m0];
        for (int r = lane; r < ka; r += 64) s_slot[ws.near_pos[lc * ws.ksel + r]] = 1;
    }
    __syncthreads();
    // number the used candidates in list order (= ascending station index)
    uint16_t *uslot = gw.uslot + (tl * 12 + m0) * (int64_t)ws.cmax;
    for (int p0 = 0; p0 < ncand; p0 += 256) {
        const int p = p0 + t;
        const bool f = p < ncand && s_slot[p] != 0;
        const unsigned long long b = __ballot(f);
        const int pre = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[wv] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_cnt[w];
        if (f) {
            const int u = off + pre;
            uslot[p] = (uint16_t)u;
            if (u < TWX_UROWS) gw.urow[(tl * 12 + m0) * TWX_UROWS + u] = ws.cand[tl * ws.cmax + p];
        }
        __syncthreads();
        if (t == 0) s_base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    if (t == 0) gw.nurow[tl * 12 + m0] = s_base <= TWX_UROWS ? s_base : -1;
}

// ---------------------------------------------------------------------------------
// k_daily_tile: (8x8-cell tile) x (month) x (64 month-major days), both variables.  Per variable: the tile-month's
// station rows (k_tile_uidx) x 64 days are staged in LDS (f4, 256 B per row, coalesced loads), then every wave takes
// TWX_DT_CPW = 8 cells, lane = day, and walks the TABLE: each row is read from LDS (conflict-free ds_read_b32 with immediate
// offsets, lanes = consecutive days) and converted once and feeds one fmac per cell, weighted with the cell's hat row in
// table order (zero where the cell does not use the row; a DPP row broadcast) -- dt_value4.  Tmin values wait in
// registers while the table is re-staged for Tmax (whose rows are fetched while the Tmin sums run); cells with any
// tmin >= tmax are flagged for k_fix_cells; results are transposed through LDS and written as 16-byte runs (8 cells
// of a tile row) of the [ndays][Y][X] int16 output.
// A tile-month whose union exceeds TWX_UROWS rows gathers from global memory (daily_value2: rank-order sums).
// ---------------------------------------------------------------------------------
// TWX_DT_WAVES = 8 waves per work-group, THREE work-groups per CU (52 KB of LDS -- the output's transposition buffer shares the
// table's space --, 76 VGPRs: the Tmax rows are fetched after the Tmin sums, not beside them; round 4: 21.2 -> 19.9 ms per C4
// tile against two work-groups with a 224-row table).  8 cells per wave share each
// row's LDS read + convert (C4 tile: 16 waves x 4 cells 35.2 ms, 8 x 8 27.5 ms, 4 x 16 34.5 ms)

// Days with tmin >= tmax are recorded per cell as they are found (chronological day index, unordered; the count keeps
// running past the capacity): a cell with at most TWX_INV_CAP of them is fixed from those days' windows alone
// (k_fix_sparse), any other by recomputing its whole series (k_fix_cells).
//
// Near ties (the "tie guard"): the fixer's test tmin >= tmax (interp_tair.py:170) is a DISCONTINUITY of the product -- a day
// whose Tmax - Tmin is within the fast covariance build's ~1e-6 degC of 0 can fall on the other side of it than in an fp64
// evaluation of the reference's formulas, and then that day moves by the window's mean diurnal range, the cell's recomputed
// normals by ~0.05 degC and its ninvalid by 1: far outside the parity bars, and the product would not know.  So the daily
// kernels also record the cells that have ANY day with |Tmax - Tmin| < TWX_TIE_EPS (20 x the fast build's observed error of
// a normal); exactly those cells are kriged again on the fp64 covariance build (run_tie_guard, twx_hip.hip), their constants
// re-formed (k_tie_rezc) and their whole series recomputed by k_fix_cells -- every output of such a cell is then what a
// TWX_FLAG_UK_F64_ALL run gives, bit for bit.  On real data this is a cell in ~1e5 (each costs ~25 us); the guard itself is
// one subtraction and one compare per cell-day.
#define TWX_INV_CAP 256
#ifndef TWX_TIE_EPS
#define TWX_TIE_EPS 2e-5
#endif
// dtr = tmax - tmin of a day, called when dtr < TWX_TIE_EPS (rare): dtr <= 0 <=> tmin >= tmax exactly (IEEE subtraction of
// finite values has the exact sign)
__device__ __forceinline__ void note_day(int32_t *flag, int32_t *inv_cnt, int32_t *inv_day, int32_t *tie, int64_t lc, int d, double dtr)
{
    if (dtr <= 0.0) {
        flag[lc] = 1;
        const int slot = atomicAdd(&inv_cnt[lc], 1);
        if (slot < TWX_INV_CAP) inv_day[lc * TWX_INV_CAP + slot] = d;
    }
    if (tie && dtr > -TWX_TIE_EPS) tie[lc] = 1;
}

// lean argument block of k_daily_tile (the full workspaces would not fit the scalar registers: 97 spilled SGPRs)
struct DtVar {
    const float *obs;         // [n][ndays] month-major
    const int32_t *ka;        // [ncell][12]
    const double *z, *zc;     // hat rows / constants
    const double *zd;         // hat rows in table-row order, wave layout (GwrWs.zd)
    const int32_t *urow;      // [ntile][12][TWX_UROWS]
    const int32_t *nurow;     // [ntile][12]
};
struct DtArgs {
    DtVar n, x;
    const int32_t *okc;       // [ncell] 1 = both variables of the cell are fine (k_daily_ok)
    const int32_t *mm2chron;
    int16_t *out_n, *out_x;   // [ndays][Y][X]
    int32_t *flag;
    int32_t *inv_cnt, *inv_day;   // per cell: number of days with tmin >= tmax, and the first TWX_INV_CAP of them (k_fix_sparse)
    int32_t *tie;             // [ncell] set for cells with a day of |tmax - tmin| < TWX_TIE_EPS (null: guard off)
    int64_t cell0, ncell, tile0, ntile;
    int Y, X, ts, ntx, ndays, nblk_max, gather;
    int moff[13];
};

// cells whose two variables both came through selection, kriging and the GWR hat rows
__global__ void k_daily_ok(SelWs wn, SelWs wx, GwrWs gn, GwrWs gx, int32_t *okc)
{
    const int64_t lc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= wn.ncell) return;
    okc[lc] = wn.cstat[lc] == 0 && wn.uk_stat[lc] == 0 && gn.gstat[lc] == 0 && wx.cstat[lc] == 0 && wx.uk_stat[lc] == 0 &&
              gx.gstat[lc] == 0;
}

__device__ __forceinline__ int64_t dt_cell(const DtArgs &a, int r0, int q0, int cl)
{
    const int rr = r0 + cl / a.ts, qq = q0 + cl % a.ts;
    if (cl >= a.ts * a.ts || rr >= a.Y || qq >= a.X) return -1;
    const int64_t lc = (int64_t)rr * a.X + qq - a.cell0;
    if (lc < 0 || lc >= a.ncell || !a.okc[lc]) return -1;
    return lc;
}

// acc += bcast(z, lane N of this lane's 16-lane row) * x   (v_fmac_f64 with a DPP row_newbcast source, see twx_uk.h)
template <int N>
__device__ __forceinline__ void dt_fmac(double &acc, double z, double x)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(z), "v"(x), "n"(N));
}

// The same sums for the TWX_DT_CPW cells of a wave at once, walking the table's rows instead of each cell's own list:
// every row is read from LDS and converted once per wave (immediate offsets: no address arithmetic) and feeds one
// fmac per cell, whose weight -- the cell's hat row delivered in table order by k_gwr_z_cell, zero where the cell does
// not use the row -- is again a DPP row broadcast.  ~2 VALU instructions per useful term instead of 3.9 (1.1 per table
// term with 8 cells per wave; the table has ~1.8 x the rows a cell uses); the terms are
// added in table order, not in rank order (the sum differs from daily_value's in the last bits, far below the int16
// rounding step; observations are finite by construction of the infilled station matrix).
__device__ __forceinline__ void dt_value4(const DtVar &v, const int64_t (&lc)[TWX_DT_CPW], int m0, const char *tab, uint32_t lane4,
                                          int lane, int nu, const double *zbase, double (&acc)[TWX_DT_CPW])
{
    // The hat-row entries of the NEXT chunk travel while this chunk is summed.  For that the loads must be
    // unconditional (behind a branch the compiler cannot count the loads in flight and waits for all of them, the
    // prefetch included, before the first fmac of a chunk): a cell outside the grid reads whatever its slot holds, a chunk
    // past the table the last one of the 14; both results are discarded.  zbase = the wave's group in GwrWs.zd (+ lane & 15):
    // a chunk of the group's cells is 8 x 128 contiguous bytes, reached with immediate offsets.  And they are relaxed ATOMIC loads
    // (wavefront scope: plain global_load instructions), which the optimizer leaves where they are -- an ordinary load
    // it turns back into "load this chunk at the top of the next iteration", right in front of its use.
    // Two register sets take turns (chunk pairs), so that no copy is needed at the end of a chunk.
    const double *zp[TWX_DT_CPW];
    double za[TWX_DT_CPW], zb[TWX_DT_CPW];
#pragma unroll
    for (int i = 0; i < TWX_DT_CPW; ++i) {
        acc[i] = 0.0;
        zp[i] = zbase + 16 * i;                              // (a cell outside the grid reads whatever its slot holds: discarded)
        za[i] = zp[i][0];
    }
    auto prefetch = [&](double (&z)[TWX_DT_CPW], int u) __attribute__((always_inline)) {
        const int un = min(u, TWX_UROWS - 16) * TWX_DT_CPW;             // (chunk stride: CPW x 16 entries)
#pragma unroll
        for (int i = 0; i < TWX_DT_CPW; ++i) z[i] = __hip_atomic_load(zp[i] + un, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    auto chunk = [&](const double (&z)[TWX_DT_CPW], int u0) __attribute__((always_inline)) {
        const char *row = tab + (uint32_t)u0 * 256u + lane4;
        // rows in groups of four behind a uniform test: the table's last chunk is short by 7.5 rows on average (~7 % of
        // a ~105-row table), and the rows past its end carry weight 0 for every cell -- skipping them changes no bit
        sfor<0, 4>([&](auto g_) __attribute__((always_inline)) {
            constexpr int G = decltype(g_)::value;
            if (u0 + 4 * G < nu) {
                sfor<0, 4>([&](auto n_) __attribute__((always_inline)) {
                    constexpr int N = 4 * G + decltype(n_)::value;
                    const double x = (double)*reinterpret_cast<const float *>(row + 256 * N);
#pragma unroll
                    for (int i = 0; i < TWX_DT_CPW; ++i) dt_fmac<N>(acc[i], z[i], x);
                });
            }
        });
    };
    for (int u0 = 0; u0 < nu; u0 += 32) {
        prefetch(zb, u0 + 16);
        chunk(za, u0);
        if (u0 + 16 >= nu) break;                            // (uniform)
        prefetch(za, u0 + 32);
        chunk(zb, u0 + 16);
    }
#pragma unroll
    for (int i = 0; i < TWX_DT_CPW; ++i) acc[i] = lc[i] >= 0 ? acc[i] + v.zc[lc[i] * 12 + m0] : 0.0;
}

#define TWX_DT_RPW ((TWX_UROWS + TWX_DT_WAVES - 1) / TWX_DT_WAVES)   // table rows staged per wave

__global__ __launch_bounds__(64 * TWX_DT_WAVES) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_daily_tile(DtArgs a)
{
    __shared__ float s_tab[TWX_UROWS * 64];
    static_assert(sizeof(int16_t) * 2 * 64 * 66 <= sizeof(float) * TWX_UROWS * 64, "the transposition buffer shares the table's space");
    int16_t (*s_v)[64][66] = reinterpret_cast<int16_t (*)[64][66]>(s_tab);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Work-groups are dealt round-robin to the 8 XCDs (one L2 each).  A unit = 8 consecutive tiles (neighbours in x) x
    // one month goes to ONE XCD, walked block by block with the 8 tiles side by side: the tiles' 16-byte output runs
    // of a day meet in that L2 and leave as whole 128-byte lines (measured write amplification of the int16 output
    // 2.1 x when neighbouring tiles ran on different XCDs), and a tile-month's hat rows (133 KB for both variables)
    // come from HBM once and from the L2 for its other blocks.
    const int64_t seq = blockIdx.x >> 3;
    const int64_t unit = (seq / (8 * a.nblk_max)) * 8 + (blockIdx.x & 7);
    const int rem = (int)(seq % (8 * a.nblk_max));
    const int64_t tl = (unit / 12) * 8 + (rem & 7);     // local tile (same tiling for both variables)
    if (tl >= a.ntile) return;
    const int m0 = (int)(unit % 12);
    const int blk = rem >> 3;
    const int dm0 = a.moff[m0] + blk * 64, dm1 = a.moff[m0 + 1];
    if (dm0 >= dm1) return;
    const int nun = a.n.nurow[tl * 12 + m0], nux = a.x.nurow[tl * 12 + m0];
    if (nun < 0 || nux < 0 || a.gather) return;         // this tile-month gathers from global memory (k_daily_tile_gather)
    const int dm = dm0 + lane;
    const bool day_ok = dm < dm1;
    const int dmc = day_ok ? dm : dm1 - 1;              // clamped: staging loads stay inside the month
    const int64_t tile = a.tile0 + tl;
    const int r0 = (int)(tile / a.ntx) * a.ts, q0 = (int)(tile % a.ntx) * a.ts;
    const char *tab = reinterpret_cast<const char *>(s_tab);
    const uint32_t lane4 = 4u * (uint32_t)lane;

    // stage rows u = wv + TWX_DT_WAVES j of a variable's table: all of this wave's row loads in flight at once (the
    // row indices come through one vector load per row: lanes read the same word)
    float pre[TWX_DT_RPW];
    auto fetch = [&](const DtVar &v, int nu) __attribute__((always_inline)) {
        const int32_t *rows = v.urow + (tl * 12 + m0) * TWX_UROWS;
#pragma unroll
        for (int j = 0; j < TWX_DT_RPW; ++j) {
            const int u = wv + TWX_DT_WAVES * j;
            pre[j] = u < nu ? v.obs[(size_t)rows[u] * a.ndays + dmc] : 0.f;
        }
    };
    auto store = [&](int nu) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TWX_DT_RPW; ++j) {
            const int u = wv + TWX_DT_WAVES * j;
            if (u < ((nu + 15) & ~15)) s_tab[u * 64 + lane] = pre[j];   // (rows up to the next multiple of 16: zeros)
        }
    };

    // ---- Tmin: stage the tile-month's rows, then walk this wave's cells.  (Issuing the Tmax row loads before the Tmin
    // sums does not help: the hat-row loads of the sums queue behind them on the same in-order counter.)
    fetch(a.n, nun);
    store(nun);
    __syncthreads();
    int64_t lcs[TWX_DT_CPW];
#pragma unroll
    for (int i = 0; i < TWX_DT_CPW; ++i) lcs[i] = dt_cell(a, r0, q0, wv * TWX_DT_CPW + i);
    double vn[TWX_DT_CPW], vxs[TWX_DT_CPW];
    dt_value4(a.n, lcs, m0, tab, lane4, lane, nun, a.n.zd + twx_zd_index(tl * 12 + m0, wv * TWX_DT_CPW) + (lane & 15), vn);
    fetch(a.x, nux);     // (after the sums: the staging registers are not live beside the sums' -- 80 VGPRs, six waves per SIMD)
    __syncthreads();
    // ---- Tmax: re-stage the table, walk the cells, flag, pack
    store(nux);
    __syncthreads();
    dt_value4(a.x, lcs, m0, tab, lane4, lane, nux, a.x.zd + twx_zd_index(tl * 12 + m0, wv * TWX_DT_CPW) + (lane & 15), vxs);
    __syncthreads();     // (s_v shares the table's space: every wave is done with the table)
#pragma unroll
    for (int i = 0; i < TWX_DT_CPW; ++i) {
        const int cl = wv * TWX_DT_CPW + i;
        const int64_t lc = lcs[i];
        const double vx = vxs[i];
        const double dtr = vx - vn[i];
        if (lc >= 0 && day_ok && dtr < TWX_TIE_EPS) note_day(a.flag, a.inv_cnt, a.inv_day, a.tie, lc, a.mm2chron[dm], dtr);
        const bool okd = lc >= 0 && day_ok;
        s_v[0][lane][cl] = okd ? pack_i16(vn[i]) : TWX_FILL_I2;
        s_v[1][lane][cl] = okd ? pack_i16(vx) : TWX_FILL_I2;
    }
    __syncthreads();
    // write: each wave takes 64 / TWX_DT_WAVES days
    const int64_t yx = (int64_t)a.Y * a.X;
    if (a.ts == 8 && TWX_DT_CPW == 8 && q0 + 8 <= a.X && r0 + 8 <= a.Y && (a.X & 1) == 0) {
        // a tile wholly inside the grid: lane = (tile row, day of the wave's eight), one 16-byte store per variable -- the
        // run of a tile row's 8 cells (the cells that failed or are masked hold the fill value the output was
        // initialised with: writing it again changes nothing).  Element offsets are even (X even): 4-byte aligned.
        struct __attribute__((packed, aligned(4))) Run { int32_t w[4]; };
        const int rw = lane >> 3, dl = wv * TWX_DT_CPW + (lane & 7);
        if (dm0 + dl < dm1) {
            const int64_t e = (int64_t)a.mm2chron[dm0 + dl] * yx + (int64_t)(r0 + rw) * a.X + q0;
            *reinterpret_cast<Run *>(a.out_n + e) = *reinterpret_cast<const Run *>(&s_v[0][dl][rw * 8]);
            *reinterpret_cast<Run *>(a.out_x + e) = *reinterpret_cast<const Run *>(&s_v[1][dl][rw * 8]);
        }
        return;
    }
    // edge tiles: lane = cell of the tile, 2-byte stores
    const int64_t lcw = dt_cell(a, r0, q0, lane);
    if (lcw < 0) return;
    const int64_t c = a.cell0 + lcw;
    for (int i = 0; i < TWX_DT_CPW; ++i) {
        const int dl = wv * TWX_DT_CPW + i;
        if (dm0 + dl >= dm1) break;
        const int64_t d = a.mm2chron[dm0 + dl];
        a.out_n[d * yx + c] = s_v[0][dl][lane];
        a.out_x[d * yx + c] = s_v[1][dl][lane];
    }
}

// the same tile-month blocks gathered from global memory (daily_value2): tile-months whose union exceeds TWX_UROWS
// rows, or all of them with TWX_FLAG_DAILY_GATHER / TWX_FLAG_OBS_ADDR64
__global__ __launch_bounds__(256) void k_daily_tile_gather(StnDev stn, StnDev stx, CellSrc src, SelWs wn, SelWs wx,
                                                           GwrWs gn, GwrWs gx, DayAxis da, twx_grid_out out,
                                                           int32_t *flag, int32_t *inv_cnt, int32_t *inv_day, int32_t *tie, const int32_t *okc,
                                                           int nblk_max, int addr64, int gather)
{
    __shared__ int16_t s_v[2][64][66];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool off32 = !addr64 && (uint64_t)max(stn.n, stx.n) * (uint64_t)da.ndays < (1ull << 30) &&
                       max(stn.n, stx.n) < (1 << 24) && da.ndays < (1 << 22);
    const int64_t tl = blockIdx.x;
    const int m0 = blockIdx.y / nblk_max;
    const int blk = blockIdx.y % nblk_max;
    const int dm0 = da.moff[m0] + blk * 64;
    if (dm0 >= da.moff[m0 + 1]) return;
    if (!gather && gn.nurow[tl * 12 + m0] >= 0 && gx.nurow[tl * 12 + m0] >= 0) return;   // done by k_daily_tile
    const int dm = dm0 + lane;
    const bool day_ok = dm < da.moff[m0 + 1];
    const int64_t tile = wn.tile0 + tl;
    const int r0 = (int)(tile / src.ntx) * src.ts, q0 = (int)(tile % src.ntx) * src.ts;
    const int64_t yx = (int64_t)src.Y * src.X;
    auto cell_lc = [&](int cl) -> int64_t {
        const int rr = r0 + cl / src.ts, qq = q0 + cl % src.ts;
        if (cl >= src.ts * src.ts || rr >= src.Y || qq >= src.X) return -1;
        const int64_t lc = (int64_t)rr * src.X + qq - wn.cell0;
        if (lc < 0 || lc >= wn.ncell || !okc[lc]) return -1;
        return lc;
    };
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int cl = wv * 16 + i;
        const int64_t lc = cell_lc(cl);
        double va = 0.0, vb = 0.0;
        if (lc >= 0 && day_ok) {
            if (off32) daily_value2<true>(stn, wn, gn, stx, wx, gx, lc, m0, da.ndays, dm, va, vb);
            else daily_value2<false>(stn, wn, gn, stx, wx, gx, lc, m0, da.ndays, dm, va, vb);
            if (vb - va < TWX_TIE_EPS) note_day(flag, inv_cnt, inv_day, tie, lc, da.mm2chron[dm], vb - va);
        }
        const bool okd = lc >= 0 && day_ok;
        s_v[0][lane][cl] = okd ? pack_i16(va) : TWX_FILL_I2;
        s_v[1][lane][cl] = okd ? pack_i16(vb) : TWX_FILL_I2;
    }
    __syncthreads();
    const int64_t lcw = cell_lc(lane);
    if (lcw < 0) return;
    const int64_t c = wn.cell0 + lcw;
    for (int i = 0; i < 16; ++i) {
        const int dl = wv * 16 + i;
        if (dm0 + dl >= da.moff[m0 + 1]) break;
        const int64_t d = da.mm2chron[dm0 + dl];
        if (out.daily_tmin) out.daily_tmin[d * yx + c] = s_v[0][dl][lane];
        if (out.daily_tmax) out.daily_tmax[d * yx + c] = s_v[1][dl][lane];
    }
}

// ---------------------------------------------------------------------------------
// k_daily_grid: (64-cell row strip) x (month) x (64 month-major days).  Lane = day
// (coalesced obs reads), each wave walks 16 cells; results are staged in LDS and
// written as 128-byte rows of the [ndays][Y][X] int16 output.  With both variables
// present, cells having any day with tmin >= tmax are flagged for k_fix_cells.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_daily_grid(StnDev stn, StnDev stx, CellSrc src, SelWs wn, SelWs wx,
                                                    GwrWs gn, GwrWs gx, int has_n, int has_x, DayAxis da,
                                                    twx_grid_out out, int32_t *flag, int32_t *inv_cnt, int32_t *inv_day, int32_t *tie,
                                                    int nblk_max, int addr64)
{
    __shared__ int16_t s_v[2][64][66];
    const int lane = threadIdx.x & 63;
    // wave index as a scalar: the cell a wave works on is then provably wave-uniform, and its hat row and
    // neighbour indices come through scalar loads -- the only vector loads left are the observations
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const SelWs &w0 = has_n ? wn : wx;
    // both observation matrices below 4 GiB and every index / row pitch below 2^24: 32-bit element offsets
    const bool off32 = !addr64 && (uint64_t)max(stn.n, stx.n) * (uint64_t)da.ndays < (1ull << 30) &&
                       max(stn.n, stx.n) < (1 << 24) && da.ndays < (1 << 22);
    const int64_t strip = blockIdx.x;               // 64 consecutive local cells
    const int m0 = blockIdx.y / nblk_max;
    const int blk = blockIdx.y % nblk_max;
    const int dm0 = da.moff[m0] + blk * 64;
    if (dm0 >= da.moff[m0 + 1]) return;
    const int dm = dm0 + lane;
    const bool day_ok = dm < da.moff[m0 + 1];
    const int64_t yx = (int64_t)src.Y * src.X;
    for (int i = 0; i < 16; ++i) {
        const int cl = wv * 16 + i;
        const int64_t lc = strip * 64 + cl;
        bool ok = lc < w0.ncell;
        if (ok && has_n) ok = wn.cstat[lc] == 0 && wn.uk_stat[lc] == 0 && gn.gstat[lc] == 0;
        if (ok && has_x) ok = wx.cstat[lc] == 0 && wx.uk_stat[lc] == 0 && gx.gstat[lc] == 0;
        double vn = 0.0, vx = 0.0;
        if (ok && day_ok) {
            if (has_n && has_x) {
                if (off32) daily_value2<true>(stn, wn, gn, stx, wx, gx, lc, m0, da.ndays, dm, vn, vx);
                else daily_value2<false>(stn, wn, gn, stx, wx, gx, lc, m0, da.ndays, dm, vn, vx);
                if (vx - vn < TWX_TIE_EPS) note_day(flag, inv_cnt, inv_day, tie, lc, da.mm2chron[dm], vx - vn);
            } else if (has_n) vn = daily_value(stn, wn, gn, lc, m0, wn.ka[lc * 12 + m0], da.ndays, dm);
            else vx = daily_value(stx, wx, gx, lc, m0, wx.ka[lc * 12 + m0], da.ndays, dm);
        }
        s_v[0][lane][cl] = (ok && day_ok) ? pack_i16(vn) : TWX_FILL_I2;
        s_v[1][lane][cl] = (ok && day_ok) ? pack_i16(vx) : TWX_FILL_I2;
    }
    __syncthreads();
    // write: each wave takes 16 days; lane = cell of the strip
    const int64_t lc = strip * 64 + lane;
    bool ok = lc < w0.ncell;
    if (ok && has_n) ok = wn.cstat[lc] == 0 && wn.uk_stat[lc] == 0 && gn.gstat[lc] == 0;
    if (ok && has_x) ok = wx.cstat[lc] == 0 && wx.uk_stat[lc] == 0 && gx.gstat[lc] == 0;
    if (!ok) return;
    const int64_t c = w0.cell0 + lc;
    for (int i = 0; i < 16; ++i) {
        const int dl = wv * 16 + i;
        if (dm0 + dl >= da.moff[m0 + 1]) break;
        const int64_t d = da.mm2chron[dm0 + dl];
        if (has_n && out.daily_tmin) out.daily_tmin[d * yx + c] = s_v[0][dl][lane];
        if (has_x && out.daily_tmax) out.daily_tmax[d * yx + c] = s_v[1][dl][lane];
    }
}

// ---------------------------------------------------------------------------------
// k_fix_cells: tmin_tmax_fixer + normals recompute (interp_tair.py:143-197,579-590)
// for the cells flagged by k_daily_tile / k_daily_tile_gather.  One workgroup per flagged cell
// (grid-stride): recompute both fp64 series chronologically into scratch, list the
// invalid days (before any fix), fix them sequentially in day order (earlier fixes
// feed later windows), recompute the normals, repack the changed days.
// scratch: [gridDim][2][ndays] doubles; lists: [gridDim][ndays] int32.
// Also used by twx_fix_pair (series given, no recompute): src_series != null.
// ---------------------------------------------------------------------------------
struct FixArgs {
    const int32_t *cells;   // flagged local cells (grid mode) or null
    int ncells;             // twx_fix_pair: number of series
    const int32_t *ncells_dev; // grid mode: number of flagged cells (k_compact_flags), read on the device
    const int32_t *inv_cnt;  // grid mode: [ncell] days with tmin >= tmax found by the daily kernels
    const int32_t *inv_day;  // grid mode: [ncell][TWX_INV_CAP] the first of them (chronological day index, unordered)
    const int32_t *tie;      // grid mode: [ncell] cells of the tie guard (note_day; re-kriged on the fp64 build): k_fix_cells recomputes and
                             // REWRITES their whole series, k_fix_sparse leaves them alone (null: guard off)
    int sparse_ok;           // grid mode: k_fix_sparse can take the cells with <= TWX_INV_CAP invalid days (fix_sparse_usable, evaluated
                             // ONCE on the host: both kernels and the launch decision read this one flag)
    double *scratch;        // [gridDim][2][ndays]
    int32_t *lists;         // [gridDim][ndays]
    double *series_min;     // twx_fix_pair: [nseries][ndays] in/out (chronological), else null
    double *series_max;
    double *norm_min_out;   // twx_fix_pair: [nseries][12] or null
    double *norm_max_out;
    int32_t *ninv_out;      // twx_fix_pair / points
    int32_t *status_out;
};

__device__ void fix_series_block(double *tmin, double *tmax, int32_t *list, const DayAxis &da,
                                 int *s_n, int *s_err, double *s_norm /*[2][12]*/)
{
    // ordered list of invalid days (taken BEFORE any fix, interp_tair.py:173)
    __shared__ int s_cnt[16];                                // per wave (work-groups of 4 ... 16 waves)
    __shared__ int s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, nth = blockDim.x, nwv = nth >> 6;
    if (t == 0) { s_base = 0; *s_err = 0; }
    __syncthreads();
    for (int d0 = 0; d0 < da.ndays; d0 += nth) {
        int d = d0 + t;
        bool f = d < da.ndays && tmin[d] >= tmax[d];
        unsigned long long b = __ballot(f);
        int pre = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[wv] = __popcll(b);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wv; ++w) off += s_cnt[w];
        if (f) list[off + pre] = d;
        __syncthreads();
        if (t == 0) { int a = 0; for (int w = 0; w < nwv; ++w) a += s_cnt[w]; s_base += a; }
        __syncthreads();
    }
    const int ninv = s_base;
    if (t == 0) {
        *s_n = ninv;
        for (int q = 0; q < ninv; ++q) {
            int x = list[q];
            double tavg = (tmin[x] + tmax[x]) / 2.0;
            int s = x - da.tail, e = x + da.tail + 1;
            if (s < 0) s = 0;
            if (e > da.ndays) e = da.ndays;
            double sum = 0.0; int cnt = 0;
            for (int d = s; d < e; ++d)
                if (tmin[d] < tmax[d]) { sum += tmax[d] - tmin[d]; ++cnt; }
            if (cnt == 0) { *s_err = 1; break; }
            double half = (sum / (double)cnt) / 2.0;
            tmin[x] = tavg - half;
            tmax[x] = tavg + half;
        }
    }
    __syncthreads();
    if (ninv == 0 || *s_err) return;
    // normals: mean over years of the per-(year, month) means, normals period only
    // (interp_tair.py:468-481,583-590); the days of one (year, month) are contiguous
    if (da.norm_ny <= 0) { if (t < 24) s_norm[t] = NAN; __syncthreads(); return; }
    if (t < 24) {
        const int v = t / 12, m = t % 12;
        const double *ser = v ? tmax : tmin;
        double acc = 0.0;
        for (int y = 0; y < da.norm_ny; ++y) {
            const int s0 = da.ym_start[y * 12 + m], n = da.ym_cnt[y * 12 + m];
            double sum = 0.0;
            for (int d = 0; d < n; ++d) sum += ser[s0 + d];
            acc += sum / (double)n;
        }
        s_norm[t] = acc / (double)da.norm_ny;
    }
    __syncthreads();
}

// lane `l` (wave-uniform) of a double
__device__ __forceinline__ double readlane_dv(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// One variable's (weight, station) list of a (cell, month) in ascending station-index order -- from the tile-month's table
// (GwrWs.zd: the rows with a non-zero weight; a zero weight adds exactly nothing) or from the rank-ordered hat row through
// perm -- built by ONE wave (ordered compaction by ballot): what the fixers sum a day's value with, bit for bit the value
// k_daily_tile / the gather kernels produced.
__device__ __forceinline__ void fix_month_list(const SelWs &w, const GwrWs &g, int64_t lc, int64_t tl, int ci, int m0, int lane,
                                               double *zl, int *jl, int *nl)
{
    const int ka = w.ka[lc * 12 + m0];
    const int64_t tm = tl * 12 + m0;
    const int nu = (g.use_table && g.nurow2[tm] >= 0) ? g.nurow[tm] : -1;
    const int nrow = nu >= 0 ? nu : g.kp[lc];
    int n = 0;
    for (int u0 = 0; u0 < nrow; u0 += 64) {
        const int u = u0 + lane;
        double z = 0.0;
        int j = 0;
        if (u < nrow) {
            if (nu >= 0) { z = g.zd[twx_zd_index(tm, ci) + twx_zd_row(u)]; j = g.urow[tm * TWX_UROWS + u]; }
            else { const int r = g.perm[lc * w.ksel + u]; z = r < ka ? g.z[(lc * 12 + m0) * TWX_KZ + r] : 0.0; j = w.near_idx[lc * w.ksel + r]; }
        }
        const bool f = z != 0.0;
        const unsigned long long b = __ballot(f);
        if (f) { const int p = n + __popcll(b & ((1ull << lane) - 1ull)); zl[p] = z; jl[p] = j; }
        n += __popcll(b);
    }
    if (lane == 0) *nl = n;
}

// 8 waves per flagged cell (TWX_FIX_THREADS): the recompute is a chain of gathers per day (memory latency), and a tile has
// fewer flagged cells than the GPU has CUs (C4 tile: 229): 4 waves per cell 2.36 ms, 8 waves 1.54, 16 waves 1.64
__global__ __launch_bounds__(1024) void k_fix_cells(StnDev stn, StnDev stx, CellSrc src, SelWs wn, SelWs wx,
                                                   GwrWs gn, GwrWs gx, DayAxis da, twx_grid_out out, FixArgs fa)
{
    __shared__ int s_n, s_err;
    __shared__ double s_norm[24];
    __shared__ double s_zl[2][TWX_UROWS];                    // per variable: the month's weights / stations, ascending station index
    __shared__ int s_jl[2][TWX_UROWS];
    __shared__ int s_nl[2];
    double *tmin = fa.scratch + (size_t)blockIdx.x * 2 * da.ndays;
    double *tmax = tmin + da.ndays;
    int32_t *list = fa.lists + (size_t)blockIdx.x * da.ndays;
    const int64_t yx = (int64_t)src.Y * src.X;
    const int ncells = *fa.ncells_dev;
    const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
    for (int it = blockIdx.x; it < ncells; it += gridDim.x) {
        const int64_t lc = fa.cells[it];
        const bool tie = fa.tie && fa.tie[lc];                  // tie-guard cell: its normals have just been kriged again (fp64 build)
        if (!tie && fa.sparse_ok && fa.inv_cnt[lc] <= TWX_INV_CAP) continue;   // fixed from its windows by k_fix_sparse
        const int64_t c = wn.cell0 + lc;
        const int rr = (int)(c / src.X), qq = (int)(c % src.X);
        const int64_t tl = (int64_t)(rr / src.ts) * src.ntx + (qq / src.ts) - wn.tile0;
        const int ci = (rr % src.ts) * src.ts + (qq % src.ts);
        // Recompute both series with the bits k_daily_tile / the gather kernels produced: per month the cell's
        // (weight, station) lists (fix_month_list), then one fma chain per day.  Waves 0 / 1 build the Tmin / Tmax list,
        // every thread takes days.
        for (int m0 = 0; m0 < 12; ++m0) {
            if (wvi < 2) fix_month_list(wvi ? wx : wn, wvi ? gx : gn, lc, tl, ci, m0, lane, s_zl[wvi], s_jl[wvi], &s_nl[wvi]);
            __syncthreads();
            const int nn = s_nl[0], nx = s_nl[1];
            const double zcn = gn.zc[lc * 12 + m0], zcx = gx.zc[lc * 12 + m0];
            for (int dm = da.moff[m0] + threadIdx.x; dm < da.moff[m0 + 1]; dm += blockDim.x) {
                double an = 0.0, ax = 0.0;
#pragma unroll 4
                for (int i = 0; i < nn; ++i) an = fma(s_zl[0][i], (double)stn.obs[(size_t)s_jl[0][i] * da.ndays + dm], an);
#pragma unroll 4
                for (int i = 0; i < nx; ++i) ax = fma(s_zl[1][i], (double)stx.obs[(size_t)s_jl[1][i] * da.ndays + dm], ax);
                const int d = da.mm2chron[dm];
                tmin[d] = an + zcn;
                tmax[d] = ax + zcx;
            }
            __syncthreads();
        }
        __syncthreads();
        // (a tie-guard cell whose second kriging failed -- the fp64 build found a system singular that the fast build let
        // pass: k_finalize_grid has entered its status -- is abandoned as a whole, like any kriging failure)
        const int ukfail = tie ? (wn.uk_stat[lc] ? wn.uk_stat[lc] : wx.uk_stat[lc]) : 0;
        if (!ukfail) fix_series_block(tmin, tmax, list, da, &s_n, &s_err, s_norm);
        const int ninv = ukfail ? 0 : s_n;
        if (ukfail || s_err) {
            // the reference raises (interp_tair.py:192) -> the worker leaves fill values
            if (threadIdx.x == 0) {
                if (out.status) out.status[c] = ukfail ? ukfail : TWX_CELL_FIXER;
                if (out.ninvalid) out.ninvalid[c] = TWX_FILL_I4;
            }
            for (int m = threadIdx.x; m < 12; m += blockDim.x) {
                if (out.norm_tmin) out.norm_tmin[m * yx + c] = TWX_FILL_F4;
                if (out.se_tmin) out.se_tmin[m * yx + c] = TWX_FILL_F4;
                if (out.norm_tmax) out.norm_tmax[m * yx + c] = TWX_FILL_F4;
                if (out.se_tmax) out.se_tmax[m * yx + c] = TWX_FILL_F4;
            }
            for (int d = threadIdx.x; d < da.ndays; d += blockDim.x) {
                if (out.daily_tmin) out.daily_tmin[(int64_t)d * yx + c] = TWX_FILL_I2;
                if (out.daily_tmax) out.daily_tmax[(int64_t)d * yx + c] = TWX_FILL_I2;
            }
        } else {
            if (threadIdx.x == 0 && out.ninvalid) out.ninvalid[c] = ninv;
            if (ninv > 0) {
                for (int m = threadIdx.x; m < 12; m += blockDim.x) {
                    if (out.norm_tmin) out.norm_tmin[m * yx + c] = (float)s_norm[m];
                    if (out.norm_tmax) out.norm_tmax[m * yx + c] = (float)s_norm[12 + m];
                }
            }
            if (tie) {          // every day: the series now stands on the re-kriged normals (ninv == 0: k_finalize_grid wrote those)
                for (int d = threadIdx.x; d < da.ndays; d += blockDim.x) {
                    if (out.daily_tmin) out.daily_tmin[(int64_t)d * yx + c] = pack_i16(tmin[d]);
                    if (out.daily_tmax) out.daily_tmax[(int64_t)d * yx + c] = pack_i16(tmax[d]);
                }
            } else {
                for (int q = threadIdx.x; q < ninv; q += blockDim.x) {
                    int d = list[q];
                    if (out.daily_tmin) out.daily_tmin[(int64_t)d * yx + c] = pack_i16(tmin[d]);
                    if (out.daily_tmax) out.daily_tmax[(int64_t)d * yx + c] = pack_i16(tmax[d]);
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// k_fix_sparse: the same fixer for a cell with at most TWX_INV_CAP invalid days, from those days alone.
//
// k_fix_cells recomputes a flagged cell's WHOLE series -- 25 203 days x ~150 terms -- although the fixer only touches the
// invalid days and reads the +-tail days around each (interp_tair.py:173-195), and the normals recompute (:583-590)
// only needs per-(year, month) sums.  A tile whose cells are mostly flagged (one day in 69 years is enough) would spend
// several times its whole interpolation in that recompute.  Here:
//   * the invalid days come from the daily kernels themselves (note_day: the exact fp64 test vn >= vx at the
//     moment the value is packed), sorted ascending;
//   * only the days of their windows are recomputed (same lists, same fma chain, same bits as the daily kernels);
//   * the sequential fix runs on those (earlier fixes feed later windows: one thread, day order);
//   * the normals are the linear form  sum_{d in (y,m)} daily[d] = sum_j z_j S_j[y][m] + n zc  with the per-station
//     sums S of observations over every (year, month) of the normals period precomputed at twx_set_stations
//     (StnDev.ymsum), plus the changes of the fixed days.  The f8 normals differ from a day-by-day sum in the last
//     bits (f4 outputs: the same value unless it sits on a rounding boundary).
// One work-group of 256 threads per flagged cell (grid-stride), scratch [gridDim][2][ndays] as k_fix_cells.
// ---------------------------------------------------------------------------------
#define TWX_NORM_NY_MAX 40
// The ONE predicate of "the sparse path is usable" (host side: twx_hip.hip sets FixArgs.sparse_ok from it and skips the
// launch otherwise): both variables' (month, year) observation sums exist, the normals period fits the LDS table, a
// fixer window fits one wave.  A longer normals period (norm_yr0..norm_yr1 are public parameters) or fixer_tail > 31
// sends every flagged cell through k_fix_cells.
__host__ __device__ inline bool fix_sparse_usable(const void *ymsum_min, const void *ymsum_max, int norm_ny, int tail)
{
    return ymsum_min && ymsum_max && norm_ny > 0 && norm_ny <= TWX_NORM_NY_MAX && 2 * tail + 1 <= 64;
}
#define TWX_FIX_LCAP (TWX_MAX_NNGHS + 8)                     // entries of a month's (weight, station) list
__global__ __launch_bounds__(256) void k_fix_sparse(StnDev stn, StnDev stx, CellSrc src, SelWs wn, SelWs wx, GwrWs gn, GwrWs gx,
                                                    DayAxis da, twx_grid_out out, FixArgs fa)
{
    __shared__ int s_inv[TWX_INV_CAP], s_sorted[TWX_INV_CAP];
    __shared__ double s_fmin[TWX_INV_CAP], s_fmax[TWX_INV_CAP];  // the fixed values of the invalid days, by position in s_sorted
    __shared__ double s_zl[12][2][TWX_FIX_LCAP];              // all twelve months' lists of both variables
    __shared__ int s_jl[12][2][TWX_FIX_LCAP];
    __shared__ int s_nl[12][2], s_err;
    __shared__ double s_ym[2][12][TWX_NORM_NY_MAX];          // per variable: sum of the (fixed) daily values of (month, year)
    if (!fa.sparse_ok) return;                               // (never launched then; k_fix_cells takes every cell)
    double *tmin = fa.scratch + (size_t)blockIdx.x * 2 * da.ndays;
    double *tmax = tmin + da.ndays;
    const int64_t yx = (int64_t)src.Y * src.X;
    const int ncells = *fa.ncells_dev;
    const int t = threadIdx.x, lane = t & 63, wvi = t >> 6;
    const int ny = da.norm_ny, W = 2 * da.tail + 1;
    for (int it = blockIdx.x; it < ncells; it += gridDim.x) {
        const int64_t lc = fa.cells[it];
        const int ninv = fa.inv_cnt[lc];
        if (ninv > TWX_INV_CAP || ninv <= 0 || (fa.tie && fa.tie[lc])) continue;   // (uniform) the full recompute takes this cell
        const int64_t c = wn.cell0 + lc;
        const int rr = (int)(c / src.X), qq = (int)(c % src.X);
        const int64_t tl = (int64_t)(rr / src.ts) * src.ntx + (qq / src.ts) - wn.tile0;
        const int ci = (rr % src.ts) * src.ts + (qq % src.ts);
        // the invalid days in ascending order (rank by counting: they are distinct)
        if (t < ninv) s_inv[t] = fa.inv_day[lc * TWX_INV_CAP + t];
        // the (weight, station) lists of the twelve months, both variables: 24 lists, one wave each
        for (int q = wvi; q < 24; q += 4) {
            const int m0 = q >> 1, v = q & 1;
            fix_month_list(v ? wx : wn, v ? gx : gn, lc, tl, ci, m0, lane, s_zl[m0][v], s_jl[m0][v], &s_nl[m0][v]);
        }
        __syncthreads();
        if (t < ninv) {
            const int mine = s_inv[t];
            int pos = 0;
            for (int q = 0; q < ninv; ++q) pos += s_inv[q] < mine ? 1 : 0;
            s_sorted[pos] = mine;
        }
        __syncthreads();
        // the days of the invalid days' windows (overlapping windows write the same bits twice)
        for (int p = t; p < ninv * W; p += 256) {
            const int d = s_sorted[p / W] - da.tail + p % W;
            if (d < 0 || d >= da.ndays) continue;
            const int m0 = da.day_month[d] - 1, dm = da.chron2mm[d];
            const int nn = s_nl[m0][0], nx = s_nl[m0][1];
            const double *zn = s_zl[m0][0], *zx = s_zl[m0][1];
            const int *jn = s_jl[m0][0], *jx = s_jl[m0][1];
            double an = 0.0, ax = 0.0;
#pragma unroll 4
            for (int i = 0; i < nn; ++i) an = fma(zn[i], (double)stn.obs[(size_t)jn[i] * da.ndays + dm], an);
#pragma unroll 4
            for (int i = 0; i < nx; ++i) ax = fma(zx[i], (double)stx.obs[(size_t)jx[i] * da.ndays + dm], ax);
            tmin[d] = an + gn.zc[lc * 12 + m0];
            tmax[d] = ax + gx.zc[lc * 12 + m0];
        }
        // sum of every month's daily values per year of the normals period, from the stations' (month, year) sums
        for (int q = t; q < 24 * ny; q += 256) {
            const int y = q % ny, mv = q / ny, m0 = mv >> 1, v = mv & 1;
            const double *S = (v ? stx.ymsum : stn.ymsum) + (size_t)m0 * ny + y;
            const int n = s_nl[m0][v];
            const double *zl = s_zl[m0][v];
            const int *jl = s_jl[m0][v];
            double acc = 0.0;
#pragma unroll 4
            for (int i = 0; i < n; ++i) acc = fma(zl[i], S[(size_t)jl[i] * 12 * ny], acc);
            s_ym[v][m0][y] = acc + (double)da.ym_cnt[y * 12 + m0] * (v ? gx.zc[lc * 12 + m0] : gn.zc[lc * 12 + m0]);
        }
        if (t == 0) s_err = 0;
        __syncthreads();
        // the fix itself: day order, earlier fixes feed later windows (interp_tair.py:177-195).  One wave: lane = day of
        // the window (raw values from the scratch, which is not written here; the days fixed so far -- found by bisection
        // in the sorted list -- from LDS)
        if (wvi == 0) {
            for (int q = 0; q < ninv; ++q) {
                const int x = s_sorted[q];
                int s0 = x - da.tail, e = x + da.tail + 1;
                if (s0 < 0) s0 = 0;
                if (e > da.ndays) e = da.ndays;
                const int d = s0 + lane;
                double a = 0.0, b = 0.0;
                bool in = d < e;
                if (in) {
                    a = tmin[d]; b = tmax[d];
                    int lo = 0, hi = q;                       // is d one of the days already fixed (s_sorted[0 .. q))?
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_sorted[mid] < d) lo = mid + 1; else hi = mid; }
                    if (lo < q && s_sorted[lo] == d) { a = s_fmin[lo]; b = s_fmax[lo]; }
                }
                const bool ok = in && a < b;
                const int cnt = __popcll(__ballot(ok));
                // the window's sum of diurnal ranges in DAY order, as fix_series_block / twx_fix_pair / the oracle form it
                // (a butterfly sum differs in the last bits, which moved a packed value by 1 LSB now and then: the fixed
                // days of a cell must not depend on which of the two fixer kernels took it); a day outside the window or
                // invalid contributes an exact + 0.0
                const double dv = ok ? b - a : 0.0;
                double sum = 0.0;
                for (int i = 0; i < e - s0; ++i) sum += readlane_dv(dv, i);
                if (cnt == 0) { if (lane == 0) s_err = 1; break; }
                // this day's own (raw) values sit in lane x - s0
                const double xa = readlane_dv(a, x - s0), xb = readlane_dv(b, x - s0);
                const double tavg = (xa + xb) / 2.0, half = (sum / (double)cnt) / 2.0;
                if (lane == 0) {
                    const double nmin = tavg - half, nmax = tavg + half;
                    const int y = da.day_year[x] - da.norm_y0, m = da.day_month[x] - 1;
                    if (y >= 0 && y < ny) { s_ym[0][m][y] += nmin - xa; s_ym[1][m][y] += nmax - xb; }
                    s_fmin[q] = nmin; s_fmax[q] = nmax;
                }
                __builtin_amdgcn_wave_barrier();             // (LDS operations of one wave execute in order)
            }
        }
        __syncthreads();
        if (s_err) {
            // the reference raises (interp_tair.py:192) -> the worker leaves fill values
            if (t == 0) {
                if (out.status) out.status[c] = TWX_CELL_FIXER;
                if (out.ninvalid) out.ninvalid[c] = TWX_FILL_I4;
            }
            for (int m = t; m < 12; m += 256) {
                if (out.norm_tmin) out.norm_tmin[m * yx + c] = TWX_FILL_F4;
                if (out.se_tmin) out.se_tmin[m * yx + c] = TWX_FILL_F4;
                if (out.norm_tmax) out.norm_tmax[m * yx + c] = TWX_FILL_F4;
                if (out.se_tmax) out.se_tmax[m * yx + c] = TWX_FILL_F4;
            }
            for (int d = t; d < da.ndays; d += 256) {
                if (out.daily_tmin) out.daily_tmin[(int64_t)d * yx + c] = TWX_FILL_I2;
                if (out.daily_tmax) out.daily_tmax[(int64_t)d * yx + c] = TWX_FILL_I2;
            }
        } else {
            if (t == 0 && out.ninvalid) out.ninvalid[c] = ninv;
            if (t < 24) {       // normals: mean over the years of the (year, month) means (interp_tair.py:583-590)
                const int v = t / 12, m = t - v * 12;
                double acc = 0.0;
                for (int y = 0; y < ny; ++y) acc += s_ym[v][m][y] / (double)da.ym_cnt[y * 12 + m];
                const double nrm = acc / (double)ny;
                float *dst = v ? out.norm_tmax : out.norm_tmin;
                if (dst) dst[m * yx + c] = (float)nrm;
            }
            for (int q = t; q < ninv; q += 256) {
                const int d = s_sorted[q];
                if (out.daily_tmin) out.daily_tmin[(int64_t)d * yx + c] = pack_i16(s_fmin[q]);
                if (out.daily_tmax) out.daily_tmax[(int64_t)d * yx + c] = pack_i16(s_fmax[q]);
            }
        }
        __syncthreads();
    }
}

// twx_fix_pair: series supplied by the caller
__global__ __launch_bounds__(256) void k_fix_series(DayAxis da, FixArgs fa)
{
    __shared__ int s_n, s_err;
    __shared__ double s_norm[24];
    int32_t *list = fa.lists + (size_t)blockIdx.x * da.ndays;
    for (int it = blockIdx.x; it < fa.ncells; it += gridDim.x) {
        double *tmin = fa.series_min + (size_t)it * da.ndays;
        double *tmax = fa.series_max + (size_t)it * da.ndays;
        fix_series_block(tmin, tmax, list, da, &s_n, &s_err, s_norm);
        if (threadIdx.x == 0) {
            fa.ninv_out[it] = s_err ? 0 : s_n;
            fa.status_out[it] = s_err ? TWX_CELL_FIXER : TWX_CELL_OK;
        }
        if (!s_err && s_n > 0 && threadIdx.x < 12) {
            if (fa.norm_min_out) fa.norm_min_out[it * 12 + threadIdx.x] = s_norm[threadIdx.x];
            if (fa.norm_max_out) fa.norm_max_out[it * 12 + threadIdx.x] = s_norm[12 + threadIdx.x];
        }
        __syncthreads();
    }
}

// Tie guard, after the second kriging of the listed cells (run_tie_guard): the daily constants zc = normal - sum_j z_j norm_j
// from the NEW normals (the sum was kept by k_gwr_z_cell: the same subtraction, the same bits as a first pass on the fp64
// build), and the cells enter the fixer's list (k_fix_cells rewrites their series whether or not a day is invalid).
// One thread per (listed cell, variable, month).
__global__ void k_tie_rezc(const int32_t *list, const int32_t *count, SelWs wn, SelWs wx, GwrWs gn, GwrWs gx, int32_t *flag,
                           long long *stats)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) atomicAdd((unsigned long long *)&stats[4], (unsigned long long)*count);   // (twx_timing.tie_cells)
    const int64_t it = idx / 24;
    if (it >= *count) return;
    const int64_t lc = list[it];
    const int q = (int)(idx % 24), v = q / 12, m = q % 12;
    const SelWs &w = v ? wx : wn;
    const GwrWs &g = v ? gx : gn;
    if (w.ka[lc * 12 + m] > 0) g.zc[lc * 12 + m] = w.uk_mean[lc * 12 + m] - g.zn[lc * 12 + m];
    if (q == 0) flag[lc] = 1;
}

__global__ void k_pack(const double *x, int64_t n, int16_t *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pack_i16(x[i]);
}

// compact flagged cells into a list (order irrelevant)
__global__ void k_compact_flags(const int32_t *flag, int64_t n, int32_t *list, int32_t *count)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) list[atomicAdd(count, 1)] = (int32_t)i;
}
