// twx_select.h -- candidate tiles and per-cell station selection kernels.
//
// Reference behaviour (station_select.py:72-192): for every point, haversine to
// ALL N stations, argsort, take the k nearest, bandwidth = distance of the
// (k+1)-th, bisquare weights.  On the GPU the all-stations pass is done once per
// TILE of cells: the triangle inequality on the sphere bounds which stations can
// be among the KSEL nearest of any cell of the tile, so each cell only ranks the
// tile's candidate list (a few hundred stations) -- with results identical to the
// full sort.
#pragma once
#include "twx_device.h"

#define TWX_UK_SLEN 29
#define TWX_DIST_NB ((TWX_MAX_NNGHS + 15) / 16)                 // block rows of the distance cache
#define TWX_DIST_BLOCKS (TWX_DIST_NB * (TWX_DIST_NB + 1) / 2)

// Workspace of one (batch, variable)
#define TWX_TC_LDS_STNS 16384  // k_tile_cand keeps its distance row in LDS up to this many stations (64 KB)

struct SelWs {
    int ksel;            // nearest-list length kept per cell (<= TWX_KSEL_MAX)
    int cmax;            // candidate slots per tile of this batch (a tile with more candidates fails its cells with TWX_CELL_CAND_OVERFLOW;
                         // run_select_uk re-runs a grid batch with longer lists before it lets that happen)
    int init_nnghs;
    int reserved0;
    int64_t cell0;       // first global cell id of the batch
    int64_t ncell;       // cells in the batch
    int64_t tile0;       // first tile id of the batch
    int64_t ntile;
    int32_t *cand;       // [ntile][cmax] station indices, ascending
    int32_t *ncand;      // [ntile]
    int32_t *ncand_max;  // [1]
    float *dscratch;     // [gridDim][n] centre distances
    int32_t *near_idx;   // [ncell][ksel] by rank (nearest first), -1 padded
    double *near_dist;   // [ncell][ksel]
    uint16_t *near_pos;  // [ncell][ksel] position of the ranked neighbour in its tile's candidate list (grid mode, daily) or null
    int32_t *nnear;      // [ncell]
    int32_t *kk;         // [ncell][12] kriging bandwidth (0 = month not requested)
    int32_t *ka;         // [ncell][12] GWR bandwidth
    double *vario;       // [ncell][12][3]
    int32_t *cstat;      // [ncell] selection-stage status
    int32_t *cdup;       // [ncell] lowest rank i whose neighbour coincides with an earlier one (k_cell_dist): systems with k > i are singular
    int32_t *bucket_cnt; // [TWX_NBUCKET]: systems per bucket of twx_krig_bucket: 0..7 one-wave kernels (steps of 8 neighbours up to 96),
                         // 8..13 the multi-wave kernels; TWX_NFAST.. the same sizes for the ill-conditioned systems (uk_needs_f64):
                         // the kernels' fp64 covariance build (<.., 1> instances; without fp64 slabs -- TWX_FLAG_NO_HOST_SYNC --
                         // only the buckets of k_uk<7 / 10, 2, 2>)
    int32_t *bucket_cells; // [TWX_NBUCKET][ncell * 12] (cell, month) items per matrix-size bucket
    double *uk_mean;     // [ncell][12]
    double *uk_var;      // [ncell][12]
    int32_t *uk_stat;    // [ncell]
    double *uk_S;        // [ncell][12][TWX_UK_SLEN] lower triangle of B'C^-1B + error flag
    double *uk_beta;     // [ncell][12][5] GLS trend coefficients (basis of k_uk: columns shifted to the cell, unscaled)
    double *vfit;        // [ncell][12][3] fitted variogram (8f-1)
    double *gd64;        // [lists][ksel (ksel - 1) / 2] fit mode: the fp64 pair distances of a point LIST's ranked neighbours (k_group_dist64),
                         // pair (i, j), j < i, at i (i - 1) / 2 + j: shared by every (bandwidth, month) point of the list (k_vario)
    double *ctrig;       // [ncell][4] sin/cos of the cell's half latitude, half longitude
    float *dist;         // [ncell][TWX_DIST_BLOCKS][16 tc][16 tr] station-pair distances (km) of the cell's kriging
                         // neighbourhood in rank order, 16x16 blocks (a >= b) -- shared by the cell's 12 monthly systems
    float *h0;           // [ncell][ksel] cell -> neighbour distance (km, sp/gstat formula)
    float *hminp;        // [ncell][ksel] lower bound of the smallest pair distance among the neighbours of ranks <= r: the running
                         // minimum of their distances to their nearest other station (StnDev.nn_km; k_select) -- what decides
                         // whether a system needs the fp64 covariance build (uk_needs_f64)
    int fast_only;       // TWX_FLAG_UK_FAST_ONLY: never route a system to the fp64 build (diagnostic)
    int f64_all;         // TWX_FLAG_UK_F64_ALL: every system on the fp64 build
    int32_t *cellf64;    // [ncell] 0, or 1 + the cell's SLOT in the fp64 slabs: a month of the cell was routed to the fp64 build
                         // (k_bucket_items numbers the routed cells as it meets them; their count comes back with the bucket counts)
    int32_t *nf64;       // [1] number of cells with a routed month
    int32_t *f64_cells;  // [ncell] the routed cells by slot (k_cell_dist64's work list)
    int f64_sized;       // 1 = the host reads the counts back and sizes fp64 slabs (default): routed systems go to the bucket of
                         // their own matrix size; 0 (TWX_FLAG_NO_HOST_SYNC) = two worst-case buckets (k <= 104, k <= 152), no slabs
    // fp64 pair distances of the cells with cellf64 set, in the layout of dist (k_cell_dist64), and their cell -> neighbour
    // distances, BY SLOT: sized by the number of routed cells when a batch has routed systems and the host knows it (null with
    // TWX_FLAG_NO_HOST_SYNC: the fp64-build kernels then evaluate every element's distance themselves, per system)
    double *dist64;      // [nf64][TWX_DIST_BLOCKS][16 tc][16 tr]
    double *h064;        // [nf64][ksel]
    const int32_t *rerun; // tie guard (twx_daily.h: note_day; run_tie_guard): [ncell] flags -- when set, k_bucket_items lists ONLY the
                         // systems of the flagged cells, all of them on the fp64 build, and k_uk_solve finishes only those (null otherwise)
#ifdef TWX_UK_STAMP      // diagnostic build only (tests/tools/uk_stamps.sh): s_memtime stamps of the panel loop
    unsigned long long *dbg;
#endif
};

// cos(latitude) of every station, by the device function hav_km() itself would call (twx_set_stations)
__global__ void k_stn_coslat(const double *lat, double *out, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = cos_lat(lat[j]);
}

// ---------------------------------------------------------------------------------
// k_tile_cand: one workgroup per tile (grid-stride).  Distances from the tile
// centre to every station (a1), bisection for a radius T holding >= KSEL stations,
// candidates = stations within T + 2 * (centre -> farthest cell) (+ margin).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_tile_cand(StnDev st, CellSrc src, SelWs ws)
{
    __shared__ int s_cnt[16];                                // per wave (work-groups of 4 ... 16 waves)
    __shared__ int s_any;
    __shared__ int s_base;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, nth = blockDim.x, nwv = nth >> 6;
    // tile-centre -> station distances of the current tile: in LDS when the station table fits (the bisection reads
    // them ~20 times), else in a global scratch row of this work-group
    extern __shared__ float s_dsc[];
    float *dsc = st.n <= TWX_TC_LDS_STNS ? s_dsc : ws.dscratch + (int64_t)blockIdx.x * st.n;

    for (int64_t tl = blockIdx.x; tl < ws.ntile; tl += gridDim.x) {
        const int64_t tile = ws.tile0 + tl;
        double clon, clat, hd = 0.0;
        int excl = -1;
        const int32_t *more = nullptr;                       // further exclusions of the list's points (twx_set_exclusions)
        int any = 1;
        if (src.mode == 1) {
            const int64_t p = src.ptfirst ? src.ptfirst[tile] : tile;     // the list's first point stands for all of them
            clon = src.pts[p].lon; clat = src.pts[p].lat;
            excl = src.excl ? src.excl[p] : -1;
            if (src.excl_more) more = src.excl_more + p * src.nexcl;
        } else {
            int ty = (int)(tile / src.ntx), tx = (int)(tile % src.ntx);
            int r0 = ty * src.ts, r1 = min(r0 + src.ts, src.Y) - 1;
            int q0 = tx * src.ts, q1 = min(q0 + src.ts, src.X) - 1;
            clat = 0.5 * (src.lat[r0] + src.lat[r1]);
            clon = 0.5 * (src.lon[q0] + src.lon[q1]);
            hd = fmax(fmax(hav_km(clon, clat, src.lon[q0], src.lat[r0]), hav_km(clon, clat, src.lon[q1], src.lat[r0])),
                      fmax(hav_km(clon, clat, src.lon[q0], src.lat[r1]), hav_km(clon, clat, src.lon[q1], src.lat[r1])));
            // skip tiles without a single unmasked cell
            if (t == 0) s_any = 0;
            __syncthreads();
            int mine = 0;
            int nr = r1 - r0 + 1, nq = q1 - q0 + 1;
            for (int i = t; i < nr * nq; i += nth)
                mine |= src.mask[(int64_t)(r0 + i / nq) * src.X + (q0 + i % nq)] != 0;
            if (mine) s_any = 1;
            __syncthreads();
            any = s_any;
        }
        if (!any) {
            if (t == 0) ws.ncand[tl] = 0;
            __syncthreads();
            continue;
        }
        // distances (float is enough for a conservative bound; margin below)
        float dmax = 0.f;
        int nvalid = 0;
        const float ccos = (float)cos_lat(clat);
        for (int j = t; j < st.n; j += nth) {
            // a conservative bound is all that is needed here (margin below): fp32 trigonometry.  The haversine is
            // exactly 0 only for identical coordinates (sin(x) = 0 <=> x = 0 at these magnitudes): the zero-distance
            // test of rm_zero_dist_stns (station_select.py:111-119) is the coordinate comparison
            const double slon = st.lon[j], slat = st.lat[j];
            float f = hav_km_f32(clon, clat, ccos, slon, slat, (float)st.coslat[j]);
            bool drop = j == excl || (src.rm_zero && slon == clon && slat == clat);   // dropped (point mode)
            if (more)                                        // (uniform)
                for (int q = 0; q < src.nexcl; ++q) drop = drop || more[q] == j;
            if (drop) f = -1.f;
            else { dmax = fmaxf(dmax, f); ++nvalid; }
            dsc[j] = f;
        }
        dmax = (float)wave_max((double)dmax);
        nvalid = wave_sum_i(nvalid);
        __shared__ float s_dmax[16];
        if (lane == 0) { s_dmax[wv] = dmax; s_cnt[wv] = nvalid; }
        __syncthreads();
        dmax = 0.f; nvalid = 0;
        for (int w = 0; w < nwv; ++w) { dmax = fmaxf(dmax, s_dmax[w]); nvalid += s_cnt[w]; }
        __syncthreads();
        float T;
        if (nvalid <= ws.ksel) {
            T = 3.0e38f;
        } else {
            float lo = 0.f, hi = dmax;
            for (int it = 0; it < 24; ++it) {
                float mid = 0.5f * (lo + hi);
                int c = 0;                                   // per wave, scalar: ballot + population count
                for (int j0 = 0; j0 < st.n; j0 += nth) {
                    const int j = j0 + t;
                    const float f = j < st.n ? dsc[j] : -1.f;
                    c += __popcll(__ballot(f >= 0.f && f <= mid));
                }
                if (lane == 0) s_cnt[wv] = c;
                __syncthreads();
                c = 0;
                for (int w = 0; w < nwv; ++w) c += s_cnt[w];
                __syncthreads();
                if (c >= ws.ksel) hi = mid; else lo = mid;
                if (c >= ws.ksel && c <= ws.ksel + 8) break;     // (uniform) tight enough: any radius holding >= ksel stations is valid
            }
            T = hi;
        }
        const float R = (T > 1.0e38f) ? T : (T * 1.000001f + 2.02f * (float)hd + 0.05f);
        // ordered compaction
        if (t == 0) s_base = 0;
        __syncthreads();
        for (int j0 = 0; j0 < st.n; j0 += nth) {
            int j = j0 + t;
            bool f = false;
            if (j < st.n) { float d = dsc[j]; f = (d >= 0.f && d <= R); }
            unsigned long long b = __ballot(f);
            int pre = __popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) s_cnt[wv] = __popcll(b);
            __syncthreads();
            int off = s_base;
            for (int w = 0; w < wv; ++w) off += s_cnt[w];
            if (f && off + pre < ws.cmax) ws.cand[tl * ws.cmax + off + pre] = j;
            __syncthreads();
            if (t == 0) { int a = 0; for (int w = 0; w < nwv; ++w) a += s_cnt[w]; s_base += a; }
            __syncthreads();
        }
        if (t == 0) {
            ws.ncand[tl] = s_base;
            atomicMax(ws.ncand_max, s_base);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// k_select<WPB>: one wavefront per cell (WPB cells per workgroup).
//   a1/a2  distances to the tile's candidates, rank by counting (ties -> smaller
//          station index), nearest list by rank
//   a3     nnghs / nnghs_anom = int(rint(weighted mean of neighbours' optimum))
//   a4     variogram parameters = weighted means over Select(nnghs)
// Two launches cover a batch without the host ever looking at the candidate counts: <4> with room for
// TWX_CAND_SMALL candidates per cell in LDS takes the cells of ordinary tiles, <1> with room for a full list
// (ws.cmax) the cells of tiles that hold more (dense station clusters); each cell is handled by exactly one of
// them (clo < ncand <= chi; masked cells belong to the first).  Dynamic LDS: WPB * chi doubles.
// ---------------------------------------------------------------------------------
#define TWX_CAND_SMALL 512
// months whose gathers are in flight together in k_select's smoothing: bandwidths (GA) / variogram parameters (GB).
// Measured on the C2 bench (select_ms): 4/1 1.08, 6/2 1.11, 12/3 1.13, 12/4 1.25 (144 VGPRs), month by month 1.26:
// past a few months per round the registers cost more occupancy than the shorter dependency chain wins.
#ifndef TWX_SEL_GA
#define TWX_SEL_GA 4
#define TWX_SEL_GB 1
#endif
#ifndef TWX_CAND_MAX
#define TWX_CAND_MAX 4096    // candidate slots per tile in grid mode (k_select<1> ranks up to this many in LDS; round 3: 2 048)
#endif
// a grid batch whose longest list does not fit is run again with up to this many slots (run_select_uk): what k_select<1,1>
// can hold in a work-group's 160 KB of LDS at 10 bytes per candidate, beside its 3 KB of static arrays
#define TWX_CAND_LDS_MAX 15872
struct SmoothOut { int status; int k; };

// v with lane L (wave-uniform index) replaced by the wave-uniform value x; lane L of v as a uniform value.  An fp64
// division is ~30 instructions whether one lane needs it or 64: k_select parks the twelve months' numerators and
// denominators in twelve lanes and divides ONCE (same operands, same IEEE quotient as month-by-month).
__device__ __forceinline__ double put_lane(double v, double x, int L)
{
    return (int)(threadIdx.x & 63) == L ? x : v;
}
__device__ __forceinline__ double get_lane(double v, int L)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}

// bisquare weight with the bandwidth's reciprocal (the variogram smoothing: fp64 means, no integer behind them)
__device__ __forceinline__ double bisq_r(double d, double inv_dbw)
{
    const double r = d * inv_dbw;
    const double u = 1.0 - r * r;
    return u * u;
}

__device__ __forceinline__ double bisq(double d, double dbw)
{
    double r = d / dbw;
    double u = 1.0 - r * r;
    return u * u;
}

// weighted mean of field[idx] over the k nearest with finite field, weights of Select(k)
// (station_select.py:164-169).  All lanes return the same values.
__device__ __forceinline__ int smooth3(const double *snd, const int *sidx, int nnear, int k,
                                       const double *f0, const double *f1, const double *f2,
                                       int lane, double out[3])
{
    if (k >= nnear) return TWX_CELL_FEW_STATIONS;
    const double dbw = snd[k];
    if (!(dbw > 0.0)) return TWX_CELL_NUMERIC;
    double n0 = 0, n1 = 0, n2 = 0, den = 0;
    int cnt = 0;                                             // neighbours with a finite field value (scalar: ballot + popcount)
    for (int r0 = 0; r0 < k; r0 += 64) {
        const int r = r0 + lane;
        const int j = r < k ? sidx[r] : -1;
        const double v0 = j >= 0 ? f0[j] : NAN;
        const bool fin = finite_d(v0);
        cnt += __popcll(__ballot(fin));
        if (fin) {
            double w = bisq(snd[r], dbw);
            n0 += v0 * w; den += w;
            if (f1) { n1 += f1[j] * w; n2 += f2[j] * w; }
        }
    }
    n0 = wave_sum_dpp(n0); den = wave_sum_dpp(den);
    if (f1) { n1 = wave_sum_dpp(n1); n2 = wave_sum_dpp(n2); }
    if (cnt == 0) return -1; // caller maps to NNGHS / VARIO
    if (!(den != 0.0)) return TWX_CELL_NUMERIC;
    out[0] = n0 / den; out[1] = n1 / den; out[2] = n2 / den;
    return TWX_CELL_OK;
}

template <int WPB>
__device__ __forceinline__ void select_cell(const StnDev &st, const CellSrc &src, const SelWs &ws, int clo, int chi, int64_t lc)
{
    extern __shared__ double s_dyn[];
    __shared__ double s_nd[WPB][TWX_KSEL_MAX];
    __shared__ int s_ni[WPB][TWX_KSEL_MAX];
    __shared__ int s_np[WPB][TWX_KSEL_MAX];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    bool in_range = lc >= 0 && lc < ws.ncell;
    const int64_t c = ws.cell0 + (in_range ? lc : 0);       // global cell id
    bool valid = in_range && cell_valid(src, c);
    double *sd = s_dyn + (size_t)wv * chi;
    uint16_t *scj = reinterpret_cast<uint16_t *>(s_dyn + (size_t)WPB * chi) + (size_t)wv * chi;   // list position of a compacted candidate
    double *snd = s_nd[wv];
    int *sni = s_ni[wv];
    int *snp = s_np[wv];

    int ncand = 0;
    const int32_t *cand = nullptr;
    CellVals cv = {0, 0, 0, 0};
    int excl = -1;
    const int32_t *more = nullptr;
    bool too_many = false;
    if (valid) {
        int64_t tl = cell_tile(src, c) - ws.tile0;
        ncand = ws.ncand[tl];
        too_many = ncand > ws.cmax;                         // the tile's list was truncated: fail its cells, never guess
        if (too_many) ncand = 0;
        cand = ws.cand + tl * ws.cmax;
        cv = cell_load(src, c);
        if (src.mode == 1 && src.excl) excl = src.excl[c];
        if (src.mode == 1 && src.excl_more) more = src.excl_more + c * src.nexcl;
    }
    // non-finite predictors of the point, as scalar bits (wave-uniform): 0 elev, 1 tdi, 2 + m lst of month m
    unsigned nan_pred = 0;
    if (valid) {
        const unsigned nl = (unsigned)__ballot(lane < 12 && !finite_d(cell_lst(src, c, lane < 12 ? lane : 0)));
        nan_pred = (unsigned)__builtin_amdgcn_readfirstlane((int)((nl << 2) | (finite_d(cv.elev) ? 0u : 1u) | (finite_d(cv.tdi) ? 0u : 2u)));
    }
    // which of the two launches owns this cell (wave-uniform)
    const int nc_tile = valid ? (too_many ? 0 : ncand) : 0;
    if (!(nc_tile > clo && nc_tile <= chi) && !(clo == 0 && nc_tile == 0)) in_range = false;
    if (!in_range) { valid = false; ncand = 0; }
    // phase 1: distances
    int nv = 0;
    for (int j = lane; j < ncand; j += 64) {
        int s = cand[j];
        double d = hav_km(cv.lon, cv.lat, st.lon[s], st.lat[s]);
        bool drop = s == excl || (src.rm_zero && d == 0.0);
        if (more)                                            // (wave-uniform)
            for (int q = 0; q < src.nexcl; ++q) drop = drop || more[q] == s;
        if (drop) d = INFINITY; else ++nv;
        sd[j] = d;
    }
    for (int r = lane; r < TWX_KSEL_MAX; r += 64) snp[r] = -1;
    nv = wave_sum_i(nv);
    __builtin_amdgcn_wave_barrier();                         // (each wave works on its own cell and LDS region: LDS operations of a wave execute in order)
    // phase 2: the ksel nearest, ranked.  Ranking by counting costs (candidates)^2 / 64 compares per lane, and a tile's
    // list is 2-3 x ksel long: first a distance bound T with at least ksel candidates inside (bisection on the fp32
    // distance: a monotone key; every step is one compare per candidate and a population count), then the candidates
    // inside the bound are compacted in list order and only they are ranked.  Whatever lies outside is strictly
    // farther than everything inside, so the ranks < ksel are the ranks of the full list (ties: list order, as before).
    float T = __builtin_inff();
    if (nv > ws.ksel) {
        float lo = 0.f, hi = 0.f;
        for (int j = lane; j < ncand; j += 64) { const double d = sd[j]; if (d != INFINITY) hi = fmaxf(hi, (float)d); }
        hi = wave_max_f(hi);
        int chi_cnt = nv;
        for (int itr = 0; itr < 12 && chi_cnt > ws.ksel + 8; ++itr) {
            const float mid = 0.5f * (lo + hi);
            int cnt = 0;                                     // scalar: ballot + population count per chunk of 64
            for (int j0 = 0; j0 < ncand; j0 += 64) {
                const int j = j0 + lane;
                cnt += __popcll(__ballot(j < ncand && (float)sd[j] <= mid));
            }
            if (cnt >= ws.ksel) { hi = mid; chi_cnt = cnt; } else lo = mid;
        }
        T = hi;
    }
    int m = 0;                                               // candidates inside the bound (wave-uniform)
    for (int j0 = 0; j0 < ncand; j0 += 64) {
        const int j = j0 + lane;
        const double d = j < ncand ? sd[j] : INFINITY;
        const bool keep = d != INFINITY && (float)d <= T;
        const unsigned long long mask = __ballot(keep);
        if (keep) {                                          // in place: position <= j, and this chunk has been read
            const int pos = m + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            sd[pos] = d; scj[pos] = (uint16_t)j;
        }
        m += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();                         // (each wave works on its own cell and LDS region: LDS operations of a wave execute in order)
    // Equal distances are rare: the first pass counts only the strictly smaller ones (one compare + one add per pair).
    // Candidates that tie then share a rank and leave the next one empty -- a hole among the ranks 0 .. ksel (one past
    // the last rank kept, so that a tie across that boundary shows too) -- and only then the wave ranks again, ties in
    // list order.
    static_assert(TWX_MAX_NNGHS + 1 < TWX_KSEL_MAX, "the rank past the kept ones needs a slot (ksel <= TWX_MAX_NNGHS + 1, pick_ksel)");
    for (int p = lane; p < m; p += 64) {
        const double dj = sd[p];
        int rank = 0;
        for (int i = 0; i < m; ++i) rank += sd[i] < dj;
        if (rank <= ws.ksel && rank < TWX_KSEL_MAX) snp[rank] = p;
    }
    __builtin_amdgcn_wave_barrier();                         // (each wave works on its own cell and LDS region: LDS operations of a wave execute in order)
    {
        bool hole = false;
        for (int r = lane; r <= ws.ksel && r < m && r < TWX_KSEL_MAX; r += 64) hole = hole || snp[r] < 0;
        if (__ballot(hole)) {                                // (wave-uniform)
            for (int r = lane; r < TWX_KSEL_MAX; r += 64) snp[r] = -1;
            __builtin_amdgcn_wave_barrier();
            for (int p = lane; p < m; p += 64) {
                const double dj = sd[p];
                int rank = 0;
                for (int i = 0; i < m; ++i) {
                    const double di = sd[i];
                    rank += (di < dj) || (di == dj && i < p);
                }
                if (rank < ws.ksel) snp[rank] = p;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                         // (each wave works on its own cell and LDS region: LDS operations of a wave execute in order)
    const int nnear = min(nv, ws.ksel);
    for (int r = lane; r < TWX_KSEL_MAX; r += 64) {
        const int p = (r < nnear) ? snp[r] : -1;
        const int j = p >= 0 ? (int)scj[p] : 0;
        const int s = p >= 0 ? cand[j] : -1;
        const double d = p >= 0 ? sd[p] : INFINITY;
        snd[r] = d; sni[r] = s;
        if (valid && r < ws.ksel) {
            ws.near_idx[lc * ws.ksel + r] = s;
            ws.near_dist[lc * ws.ksel + r] = d;
            if (ws.near_pos) ws.near_pos[lc * ws.ksel + r] = (uint16_t)j;
        }
    }
    __builtin_amdgcn_wave_barrier();                         // (each wave works on its own cell and LDS region: LDS operations of a wave execute in order)
    if (!in_range) return;
    if (!valid) {
        if (lane == 0) { ws.cstat[lc] = TWX_CELL_MASKED; ws.nnear[lc] = 0; }
        if (lane < 12) { ws.kk[lc * 12 + lane] = 0; ws.ka[lc * 12 + lane] = 0; }   // (kk > 0 is what the kriging kernels go by)
        return;
    }
    // what routes a system to the fp64 covariance build (uk_needs_f64): a lower bound of the smallest pair distance among
    // the neighbours of ranks <= r -- the running minimum of the stations' distances to their nearest other station
    // (k_stn_nn; exact unless a neighbour's nearest partner lies outside the neighbourhood, and then on the safe side:
    // the system is routed).  Three gathers and three wave scans per cell, against a second pass over the tile's pair
    // table (rounds 3-5: k_tile_dist<1>, 0.26 ms per tile batch of a run on fitted variograms).
    if (src.do_krig) {
        float run = __builtin_inff();
        for (int r0 = 0; r0 < ws.ksel; r0 += 64) {
            const int r = r0 + lane;
            const int s = r < nnear ? sni[r] : -1;
            float v = s >= 0 ? st.nn_km[s] : __builtin_inff();
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) v = fminf(v, __shfl_up(v, o, 64));   // (lanes < o get their own value back)
            v = fminf(v, run);
            if (r < ws.ksel) ws.hminp[lc * ws.ksel + r] = v;
            run = __shfl(v, 63, 64);
        }
    }
    // phase 3: monthly smoothing (a3, a4) in the reference's order: krig then gwr per month
    int status = too_many ? TWX_CELL_CAND_OVERFLOW : TWX_CELL_OK;
    const int only = (src.mode == 1 && src.mth) ? src.mth[c] : 0;
    const int k_in = (src.mode == 1 && src.nnghs_in) ? src.nnghs_in[c] : 0;
    const size_t n = (size_t)st.n;
    const bool given_vario = src.mode == 1 && src.vario_in && finite_d(src.vario_in[c * 3]);
    if (only == 0 && k_in <= 0 && !given_vario) {
        // All twelve months at once (the grid path and whole-year point requests).  The month-by-month form below
        // runs 24-36 dependent rounds of gathers + wave reductions per cell; here the bandwidth smoothing of all
        // months shares its weights (Select(init_nnghs)) and its 12 (24) gathers are in flight together, and the
        // variogram smoothing runs several months per round.  Same sums per month; the statuses are then resolved
        // month by month in the reference's order (krig, then gwr).
        constexpr int NS = (TWX_MAX_NNGHS + 63) / 64;        // neighbour slots per lane
        constexpr int GA = TWX_SEL_GA, GB = TWX_SEL_GB;      // months per round (registers <-> rounds of gather latency)
        const int kinit = ws.init_nnghs;
        int rc_init = TWX_CELL_OK;                           // smooth3's checks of Select(init_nnghs): the same for every month
        if (kinit >= nnear) rc_init = TWX_CELL_FEW_STATIONS;
        else if (!(snd[kinit] > 0.0)) rc_init = TWX_CELL_NUMERIC;
        int jn[NS];
        double dn[NS], wi[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int r = lane + 64 * u;
            jn[u] = r < nnear ? sni[r] : -1;
            dn[u] = r < nnear ? snd[r] : INFINITY;
            wi[u] = (rc_init == TWX_CELL_OK && r < kinit) ? bisq(dn[u], snd[kinit]) : 0.0;
        }
        // weighted mean of one field per month over Select(init_nnghs) -> rounded bandwidth, smooth3's status
        auto bandwidths = [&](const double *field, int (&kq)[12], int (&rq)[12]) __attribute__((always_inline)) {
            double numv = 0.0, denv = 1.0;                   // lane m: month m's weighted sum / sum of weights
            // the sum of weights runs over the neighbours whose value is finite: the same neighbours month after
            // month in practice (a station outside the mask has no value in any month), so its wave reduction is
            // repeated only when the pattern of finite neighbours changes (same addends, same order: same bits)
            unsigned long long mprev[NS];
            double den_prev = 0.0;
            bool have_prev = false;
#pragma unroll
            for (int u = 0; u < NS; ++u) mprev[u] = 0;
#pragma unroll
            for (int g = 0; g < 12; g += GA) {               // GA months of gathers in flight
                double v[GA][NS];
#pragma unroll
                for (int q = 0; q < GA; ++q)
#pragma unroll
                    for (int u = 0; u < NS; ++u)      // station-major: the GA months of a neighbour are 8 GA contiguous bytes
                        v[q][u] = (jn[u] >= 0 && lane + 64 * u < kinit) ? field[(size_t)jn[u] * 12 + (g + q)] : NAN;
#pragma unroll
                for (int q = 0; q < GA; ++q) {
                    double n0 = 0, den = 0;
                    int cnt = 0;
                    bool same = have_prev;
#pragma unroll
                    for (int u = 0; u < NS; ++u) {
                        const bool fin = finite_d(v[q][u]);
                        const unsigned long long mk = __ballot(fin);
                        cnt += __popcll(mk);
                        same = same && mk == mprev[u];
                        mprev[u] = mk;
                        if (fin) { n0 += v[q][u] * wi[u]; den += wi[u]; }
                    }
                    n0 = wave_sum_dpp(n0);
                    den = same ? den_prev : wave_sum_dpp(den);   // (uniform branch)
                    den_prev = den; have_prev = true;
                    int rc = rc_init;
                    if (!rc && cnt == 0) rc = TWX_CELL_NNGHS;
                    if (!rc && !(den != 0.0)) rc = TWX_CELL_NUMERIC;
                    rq[g + q] = __builtin_amdgcn_readfirstlane(rc);   // scalars: SGPRs
                    if (!rc) { numv = put_lane(numv, n0, g + q); denv = put_lane(denv, den, g + q); }
                }
            }
            const double quot = rint(numv / denv);           // all twelve months at once (lanes 0..11)
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                int kk = 0, rc = rq[m];
                if (!rc) { kk = (int)get_lane(quot, m); if (kk < 1 || kk > TWX_MAX_NNGHS) rc = TWX_CELL_RANGE; }
                kq[m] = __builtin_amdgcn_readfirstlane(kk); rq[m] = __builtin_amdgcn_readfirstlane(rc);
            }
        };
        int kq[12], rq[12], kaq[12], raq[12], rvq[12];
#pragma unroll
        for (int m = 0; m < 12; ++m) { kq[m] = kaq[m] = 0; rq[m] = raq[m] = rvq[m] = TWX_CELL_OK; }
        if (!(src.do_krig && src.do_vario) && lane < 36) ws.vario[lc * 36 + lane] = 0.0;
        if (src.do_krig) bandwidths(st.optim_s, kq, rq);
        if (src.do_anom) {
            bandwidths(st.optim_anom_s, kaq, raq);
#pragma unroll
            for (int m = 0; m < 12; ++m) {                   // GwrTairAnom.__get_nnghs (interp_tair.py:245-259)
                if (!raq[m] && kaq[m] >= nnear) raq[m] = TWX_CELL_FEW_STATIONS;
                if (!raq[m] && !(snd[kaq[m]] > 0.0)) raq[m] = TWX_CELL_NUMERIC;
                raq[m] = __builtin_amdgcn_readfirstlane(raq[m]);
            }
        }
        if (src.do_krig) {
            // KrigTair.__get_vario_params (interp_tair.py:837-851): weights of Select(k_m), GB months per round;
            // lane 3 m + f collects month m's numerator f and denominator: one division and one coalesced store
            double vnum = 0.0, vden = 1.0;
#pragma unroll
            for (int g = 0; g < 12; g += GB) {
                double w[GB][NS], f[GB][3][NS];
                int rc[GB];
#pragma unroll
                for (int q = 0; q < GB; ++q) {
                    const int m = g + q, k = rq[m] ? 0 : kq[m];
                    rc[q] = rq[m];
                    if (!rc[q] && k >= nnear) rc[q] = TWX_CELL_FEW_STATIONS;
                    if (!rc[q] && !(snd[k] > 0.0)) rc[q] = TWX_CELL_NUMERIC;
                    const double inv_dbw = 1.0 / (rc[q] ? 1.0 : snd[k]);   // (one division per month instead of one per neighbour)
#pragma unroll
                    for (int u = 0; u < NS; ++u) {
                        const bool act = !rc[q] && src.do_vario && lane + 64 * u < k;
                        w[q][u] = act ? bisq_r(dn[u], inv_dbw) : 0.0;
                        const double *vs = st.vario_s + ((size_t)(act ? jn[u] : 0) * 12 + m) * 4;   // nug, psill, rng: 24 contiguous bytes
                        f[q][0][u] = act ? vs[0] : NAN;
                        f[q][1][u] = act ? vs[1] : 0.0;
                        f[q][2][u] = act ? vs[2] : 0.0;
                    }
                }
#pragma unroll
                for (int q = 0; q < GB; ++q) {
                    const int m = g + q;
                    if (src.do_vario) {
                        double n0 = 0, n1 = 0, n2 = 0, den = 0;
                        int cnt = 0;
#pragma unroll
                        for (int u = 0; u < NS; ++u) {
                            const bool fin = finite_d(f[q][0][u]);
                            cnt += __popcll(__ballot(fin));
                            if (fin) { n0 += f[q][0][u] * w[q][u]; n1 += f[q][1][u] * w[q][u]; n2 += f[q][2][u] * w[q][u]; den += w[q][u]; }
                        }
                        n0 = wave_sum_dpp(n0); n1 = wave_sum_dpp(n1); n2 = wave_sum_dpp(n2); den = wave_sum_dpp(den);
                        if (!rc[q] && cnt == 0) rc[q] = TWX_CELL_VARIO;
                        if (!rc[q] && !(den != 0.0)) rc[q] = TWX_CELL_NUMERIC;
                        if (rc[q]) n0 = n1 = n2 = 0.0, den = 1.0;
                        vnum = put_lane(vnum, n0, 3 * m); vnum = put_lane(vnum, n1, 3 * m + 1); vnum = put_lane(vnum, n2, 3 * m + 2);
                        vden = put_lane(vden, den, 3 * m); vden = put_lane(vden, den, 3 * m + 1); vden = put_lane(vden, den, 3 * m + 2);
                    }
                    rvq[m] = __builtin_amdgcn_readfirstlane(rc[q]);   // (without do_vario: only Select(k) must exist)
                }
            }
            if (src.do_vario && lane < 36) ws.vario[lc * 36 + lane] = vnum / vden;
        }
        // resolve month by month: the first failure sticks (the reference abandons the point).  A non-finite predictor of
        // the point fails a month's kriging (lon, lat, elev, lst) / GWR (+ tdi) with a floating-point error AFTER that month's
        // selection steps and BEFORE the next month's: it is entered here, in its place in the sequence (the kriging
        // kernels would find it too -- NaN results -- but only for cells that no LATER month's selection had failed)
        const unsigned nan_lst = nan_pred >> 2;
        const bool nan_k = nan_pred & 1u, nan_a = nan_pred & 3u;
#pragma unroll
        // kk[m] stays set for every month whose KRIGING the reference's loop reaches before it abandons the point
        // (interp_tair.py:429-437: krig(m), then gwr(m), month by month): the kriging kernels run those systems also for a
        // cell that fails LATER -- in this month's GWR selection or in a later month -- so that a singular system among
        // them reports TWX_CELL_NUMERIC, as the reference's loop would, and not the later failure's code (k_finalize_*).
        for (int m0 = 0; m0 < 12; ++m0) {
            if (status == TWX_CELL_OK && src.do_krig) status = rq[m0] ? rq[m0] : rvq[m0];
            if (status == TWX_CELL_OK && src.do_krig && (nan_k || ((nan_lst >> m0) & 1u))) status = TWX_CELL_NUMERIC;
            const bool krig_ok = status == TWX_CELL_OK;
            if (status == TWX_CELL_OK && src.do_anom) status = raq[m0];
            if (status == TWX_CELL_OK && src.do_anom && (nan_a || ((nan_lst >> m0) & 1u))) status = TWX_CELL_NUMERIC;
            if (lane == 0) {
                ws.kk[lc * 12 + m0] = krig_ok ? kq[m0] : 0;
                ws.ka[lc * 12 + m0] = status ? 0 : kaq[m0];
            }
        }
    } else
    for (int m0 = 0; m0 < 12; ++m0) {
        int k = 0, kan = 0;
        double vp[3] = {0, 0, 0};
        bool krig_ok = false;                                // (this month's kriging is reached and its selection holds: see above)
        if ((only == 0 || only == m0 + 1) && status == TWX_CELL_OK) {
            double tmp[3];
            if (src.do_krig) {
                // KrigTair.__get_nnghs (interp_tair.py:821-835)
                if (k_in > 0) k = k_in;
                else {
                    int rc = smooth3(snd, sni, nnear, ws.init_nnghs, st.optim + m0 * n, nullptr, nullptr, lane, tmp);
                    if (rc == -1) rc = TWX_CELL_NNGHS;
                    if (rc) status = rc; else k = (int)rint(tmp[0]);
                }
                if (!status && (k < 1 || k > TWX_MAX_NNGHS)) status = TWX_CELL_RANGE;
                // KrigTair.__get_vario_params (interp_tair.py:837-851), weights of Select(k)
                if (!status) {
                    bool given = src.mode == 1 && src.vario_in && finite_d(src.vario_in[c * 3]);
                    if (given) {
                        vp[0] = src.vario_in[c * 3]; vp[1] = src.vario_in[c * 3 + 1]; vp[2] = src.vario_in[c * 3 + 2];
                        if (k >= nnear) status = TWX_CELL_FEW_STATIONS;
                    } else if (src.do_vario) {
                        int rc = smooth3(snd, sni, nnear, k, st.nug + m0 * n, st.psill + m0 * n, st.rng + m0 * n, lane, vp);
                        if (rc == -1) rc = TWX_CELL_VARIO;
                        if (rc) status = rc;
                    } else {                       // variogram fitted afterwards: only Select(k) must exist
                        if (k >= nnear) status = TWX_CELL_FEW_STATIONS;
                        else if (!(snd[k] > 0.0)) status = TWX_CELL_NUMERIC;
                    }
                }
            }
            // (a non-finite predictor of the point: the month's kriging fails here, see the all-months form above)
            if (!status && src.do_krig && ((nan_pred & 1u) || ((nan_pred >> (2 + m0)) & 1u))) status = TWX_CELL_NUMERIC;
            krig_ok = !status;
            // GwrTairAnom.__get_nnghs (interp_tair.py:245-259)
            if (!status && src.do_anom) {
                if (k_in > 0) kan = k_in;
                else {
                    int rc = smooth3(snd, sni, nnear, ws.init_nnghs, st.optim_anom + m0 * n, nullptr, nullptr, lane, tmp);
                    if (rc == -1) rc = TWX_CELL_NNGHS;
                    if (rc) status = rc; else kan = (int)rint(tmp[0]);
                }
                if (!status && (kan < 1 || kan > TWX_MAX_NNGHS)) status = TWX_CELL_RANGE;
                if (!status && kan >= nnear) status = TWX_CELL_FEW_STATIONS;
                if (!status && !(snd[kan] > 0.0)) status = TWX_CELL_NUMERIC;
                if (!status && ((nan_pred & 3u) || ((nan_pred >> (2 + m0)) & 1u))) status = TWX_CELL_NUMERIC;
            }
        }
        if (!krig_ok) k = 0;
        if (status) kan = 0;
        if (lane == 0) {
            ws.kk[lc * 12 + m0] = k;
            ws.ka[lc * 12 + m0] = kan;
            ws.vario[(lc * 12 + m0) * 3 + 0] = vp[0];
            ws.vario[(lc * 12 + m0) * 3 + 1] = vp[1];
            ws.vario[(lc * 12 + m0) * 3 + 2] = vp[2];
        }
    }
    if (lane == 0) {
        ws.cstat[lc] = status;
        ws.nnear[lc] = nnear;
        ws.uk_stat[lc] = TWX_CELL_OK;
        {
            const double r = 3.14159265358979323846 / 180.0;
            ws.ctrig[lc * 4 + 0] = sin(cv.lat * r / 2.0); ws.ctrig[lc * 4 + 1] = cos(cv.lat * r / 2.0);
            ws.ctrig[lc * 4 + 2] = sin(cv.lon * r / 2.0); ws.ctrig[lc * 4 + 3] = cos(cv.lon * r / 2.0);
        }
    }
}

// Launch forms: BY_TILE = 0: work-group g takes the cells g WPB .. g WPB + WPB - 1 (one wave each; the launch of ordinary
// tiles).  BY_TILE = 1 (WPB = 1): work-group g takes the g-th tile of the batch and, when its candidate list is a long
// one (clo < ncand <= chi), walks its cells -- a handful of work-groups do real work and the launch costs nothing otherwise
// (one work-group per CELL with 40 KB of dynamic LDS each took 0.16 ms per C2 step just to find that out).
template <int WPB, int BY_TILE>
__global__ __launch_bounds__(64 * WPB) void k_select(StnDev st, CellSrc src, SelWs ws, int clo, int chi)
{
    if constexpr (BY_TILE) {
        const int64_t tl = blockIdx.x;
        if (tl >= ws.ntile) return;
        const int nc = ws.ncand[tl];
        if (!(nc > clo && nc <= chi)) return;               // (an overflowing tile's cells are failed by the first launch)
        const int64_t tile = ws.tile0 + tl;
        const int r0 = (int)(tile / src.ntx) * src.ts, q0 = (int)(tile % src.ntx) * src.ts;
        for (int ci = 0; ci < src.ts * src.ts; ++ci) {
            const int rr = r0 + ci / src.ts, qq = q0 + ci % src.ts;
            if (rr >= src.Y || qq >= src.X) continue;
            select_cell<WPB>(st, src, ws, clo, chi, (int64_t)rr * src.X + qq - ws.cell0);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        const int wgi = xcd_contig(blockIdx.x, (int)((ws.ncell + WPB - 1) / WPB));   // row-major cells: a tile's rows meet in one L2
        if (wgi < 0) return;
        select_cell<WPB>(st, src, ws, clo, chi, (int64_t)wgi * WPB + (threadIdx.x >> 6));
    }
}

// ---------------------------------------------------------------------------------
// k_bucket_items: counting sort of the (cell, month) kriging items by matrix size, in steps of 8 neighbours.
// With m = ceil(k / 16) block rows of 16:
//   k <= 16 m - 8   bordered form: k C rows + 7 border rows at the fixed rows NP-7..NP-1 of NP = 16 m rows (the last C
//                   column stays out of the 4-column panel that holds the first border column): k_ukw<m> / k_uk<m>
//   k >  16 m - 8   the border would open a block row of its own: k_ukwz<m> keeps it as columns beside NP = 16 m C rows
//                   (twx_ukw.h) up to m = 6; larger systems take the bordered form with m + 1 block rows
// (k = 16 m - 7 in the bordered form -- a last panel of ONE C column beside three border columns, their pivots masked --
// was built in round 5 and taken out again: no value of the bandwidth ladder 35, 39, 43, 47, 52, 57, 63, 69, 76, 84, 92,
// 101, 111, 122, 134, 147 (optimize.py:376-405) is of that form -- 122 needs 129 rows, not 128 -- and the masks on the
// pivot chain cost every multi-wave kernel ~5 %: DESIGN.md section 10.)
// LDS counters per workgroup, one global atomic per (workgroup, bucket).  Order inside a bucket is irrelevant.
// ---------------------------------------------------------------------------------
#define TWX_NFAST 14                                           // matrix-size buckets (one kriging launch each)
#define TWX_NBUCKET (2 * TWX_NFAST)
#define TWX_BUCKET_F64 TWX_NFAST                              // fp64-build buckets: TWX_BUCKET_F64 + twx_krig_bucket(k)
// 0..7: k <= 40, 48, .. 96 (one-wave kernels, twx_ukw.h); 8: <= 104 k_uk<7,2>; 10: <= 120 k_uk<8,4>; 12: <= 136 k_uk<9,2>;
// 13: <= 152 k_uk<10,2>.  Buckets 9 (105..112) and 11 (121..128) stay empty: their two-wave kernels with the border as
// columns were measured slower than the bordered kernel one size up (tests/tools/experiments/twx_ukz.h).
__host__ __device__ __forceinline__ int twx_krig_bucket(int k)
{
    const int e = (k + 7) / 8;                               // eighths: 5 (k <= 40) .. 19 (k <= 152)
    const int b = (e < 5 ? 5 : (e > 19 ? 19 : e)) - 5;      // 0 .. 14
    return b < 9 ? b : (b < 11 ? 10 : (b < 13 ? 12 : 13));
}

// fp64-build bucket of a system of the TIE GUARD's second pass (SelWs.rerun: a few dozen cells of a batch): one of FOUR sizes
// -- 64 rows (k_ukwz<4, 1>), 96 (k_ukwz<6, 1>), 128 (k_uk<8, 4, 1>), 160 (k_uk<10, 2, 1>) -- instead of its own of fourteen:
// 8 small launches for both variables instead of up to 28 (1.04 -> 0.54 ms for 32 guarded cells).  A system in a larger
// kernel eliminates identity rows (pivot 1, factors 0): same result.  Systems routed by their amplification keep the
// kernel of their own size whatever their number: a table may route MOST of its systems (then the padding of a larger
// kernel costs 13 %: 15.7 -> 17.8 ms on the all-routed C2 step), and a run that routes 0.7 % of them gained nothing from
// fewer launches (EXPERIMENTS.md, round 6).
__host__ __device__ __forceinline__ int twx_f64_coarse_bucket(int b) { return b <= 3 ? 3 : (b <= 7 ? 7 : (b <= 10 ? 10 : 13)); }

// Which systems need the fp64 covariance build.  The fast build forms every off-diagonal entry psill exp(-h / range)
// from an fp32 pair distance (relative error ~3e-7) with one fp32 fma + v_exp_f32 (~1.5e-7): an absolute perturbation
// of ~2e-7 psill per entry.  Two neighbours hmin apart have nearly equal rows; the difference of their kriging weights
// is governed by 2 (nug + psill (1 - exp(-hmin / range))) on the diagonal of the rotated system, so the perturbation
// reaches the prediction multiplied by
//     amp = psill / (2 (nug + psill (1 - exp(-hmin / range))))
// (measured on the GPU, tests/tools/gpu_closepair_scan.py with the fast build forced: error <= ~8e-7 amp degC on ten
// disagreeing pairs 50-300 m apart: 7.6e-6 at amp 10, 5e-5 at amp 100, 2.4e-3 at amp 9 000).  The reference's nugget is
// min(gamma) of an empirical variogram (interp.R:304-359): nothing bounds it away from 0, and step20 removes only exact
// duplicates (step20:51-57).  Systems with amp > TWX_F64_AMP are routed to the <.., 1> instance of their kernel: fp64 distances from the
// stations' half-angle trigonometry, fp64 exp.  A pure-nugget model has no off-diagonal entries to perturb.
#ifndef TWX_F64_AMP
#define TWX_F64_AMP 8.0
#endif
// necessary condition (1 - exp(.) >= 0): the nugget alone must be below psill / 16; k_bucket_items reads hminp only
// then (the synthetic benchmark never does).
__device__ __forceinline__ bool uk_may_need_f64(double nug, double psill, double rng)
{
    return rng > 0.0 && psill > 0.0 && (2.0 * TWX_F64_AMP) * nug < psill;
}
__device__ __forceinline__ bool uk_needs_f64(double nug, double psill, double rng, float hmin)
{
    const double t = -expm1(-(double)hmin / rng);            // hmin = +inf (a single neighbour): t = 1
    return (2.0 * TWX_F64_AMP) * (nug + psill * t) < psill;
}
// does any month of the cell meet the necessary condition?  Lanes 0..11 of the calling wave take a month each (one
// 24-byte load per lane, one ballot: wave-uniform result); vario = ws.vario + lc * 36
__device__ __forceinline__ bool cell_may_need_f64(const double *vario, int lane)
{
    bool mine = false;
    if (lane < 12) mine = uk_may_need_f64(vario[lane * 3], vario[lane * 3 + 1], vario[lane * 3 + 2]);
    return __ballot(mine) != 0ull;
}

__global__ __launch_bounds__(256) void k_bucket_items(SelWs ws)
{
    __shared__ int s_cnt[TWX_NBUCKET], s_base[TWX_NBUCKET];
    const int t = threadIdx.x;
    if (t < TWX_NBUCKET) s_cnt[t] = 0;
    __syncthreads();
    const int64_t item = (int64_t)blockIdx.x * 256 + t;
    int id = -1, rank = 0;
    if (item < ws.ncell * 12) {
        const int64_t lc = item / 12;
        const int k = ws.kk[item];                           // (0 for masked cells and for months the reference's loop does not reach)
        if (k > 0 && (!ws.rerun || ws.rerun[lc])) {
            id = twx_krig_bucket(k);
            const double nug = ws.vario[item * 3], psill = ws.vario[item * 3 + 1], rng = ws.vario[item * 3 + 2];
            if (ws.f64_all || ws.rerun || (!ws.fast_only && uk_may_need_f64(nug, psill, rng) &&
                               uk_needs_f64(nug, psill, rng, ws.hminp[lc * ws.ksel + min(k, ws.ksel) - 1]))) {
                // its own matrix size's fp64-build kernel (the tie guard's pass: one of four sizes, twx_f64_coarse_bucket);
                // without slabs the two-wave kernels of 112 / 160 rows take all of them
                id = TWX_BUCKET_F64 + (ws.f64_sized ? (ws.rerun ? twx_f64_coarse_bucket(id) : id) : (k > 104 ? 13 : 8));
                // the first routed month of a cell claims the cell a slot in the fp64 slabs (read by later kernels only)
                if (atomicCAS(&ws.cellf64[lc], 0, -1) == 0) {
                    const int slot = atomicAdd(ws.nf64, 1);
                    ws.f64_cells[slot] = (int32_t)lc;
                    ws.cellf64[lc] = slot + 1;
                }
            }
            rank = atomicAdd(&s_cnt[id], 1);
        }
    }
    __syncthreads();
    if (t < TWX_NBUCKET && s_cnt[t] > 0) s_base[t] = atomicAdd(&ws.bucket_cnt[t], s_cnt[t]);
    __syncthreads();
    if (id >= 0) ws.bucket_cells[(int64_t)id * ws.ncell * 12 + s_base[id] + rank] = (int32_t)item;
}

// launch statistics for twx_get_timing without a host read-back on the launch path: stats[0] += systems solved,
// stats[1] += kriging launches that had work (+ 1 for k_cell_dist), stats[2] += systems that took the fp64 build;
// a tie-guard pass (ws.rerun) counts apart: stats[3] += systems kriged a second time (stats[4], their cells: k_tie_rezc)
__global__ void k_bucket_stats(SelWs ws, long long *stats)
{
    const int b = threadIdx.x;
    const int c = b < TWX_NBUCKET ? ws.bucket_cnt[b] : 0;
    const int tot = wave_sum_i(c), nz = wave_sum_i(c > 0 ? 1 : 0), f64 = wave_sum_i(b >= TWX_BUCKET_F64 ? c : 0);
    if (ws.rerun) {
        if (b == 0) atomicAdd((unsigned long long *)&stats[3], (unsigned long long)tot);
        return;
    }
    if (b == 0) {
        atomicAdd((unsigned long long *)&stats[0], (unsigned long long)tot);
        atomicAdd((unsigned long long *)&stats[1], (unsigned long long)(nz + 1));
        atomicAdd((unsigned long long *)&stats[2], (unsigned long long)f64);
    }
}
