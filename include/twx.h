/*
 * twx.h -- C ABI of libtwxhip.so: the MI355X (gfx950) implementation of the
 * TopoWx moving-window regression-kriging / GWR interpolation hot path.
 *
 * The reference (jaredwo/topowx) has no FFI: its seam is a set of Python
 * classes called once per grid cell.  Each entry point
 * below names the reference interface it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every function returns an int status: 0 = ok, <0 = call-level failure
 *     (twx_last_error() has the text); it never throws and never exits.
 *   - per-cell / per-point failures are NOT call failures: they are reported in
 *     the caller's status[] array with the TWX_CELL_* codes below and the
 *     corresponding outputs are left untouched (the reference's worker catches
 *     the exception and leaves fill values, step25:154-160).
 *   - the library owns device memory behind the opaque context; the caller
 *     owns every buffer it passes.  "host" buffers are plain host memory;
 *     the *_dev entry takes device pointers and a hipStream_t and is
 *     asynchronous on that stream.
 *   - one context per GPU; calls on one context must be serialised by the
 *     caller, different contexts are independent.
 *   - months are 1..12; variables are TWX_TMIN / TWX_TMAX.
 */
#ifndef TWX_H
#define TWX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TWX_TMIN 0
#define TWX_TMAX 1
#define TWX_VAR_TMIN_BIT 1
#define TWX_VAR_TMAX_BIT 2

/* largest supported neighbourhood (the reference ladder tops out at 147,
 * step21:198); larger smoothed bandwidths give TWX_CELL_RANGE */
#define TWX_MAX_NNGHS 152

/* per-cell status codes */
#define TWX_CELL_OK 0
#define TWX_CELL_FEW_STATIONS 1 /* IndexError at station_select.py:164 */
#define TWX_CELL_NNGHS 2        /* "Cannot determine the optimal # of neighbors" interp_tair.py:252,829 */
#define TWX_CELL_VARIO 3        /* "Cannot determine variogram params!" interp_tair.py:843 */
#define TWX_CELL_NUMERIC 4      /* FloatingPointError (np.seterr, step25:319) / singular kriging system */
#define TWX_CELL_FIXER 5        /* 'No valid tmin/tmax in window' interp_tair.py:192 */
#define TWX_CELL_RANGE 6        /* bandwidth above TWX_MAX_NNGHS */
#define TWX_CELL_CAND_OVERFLOW 7 /* library limit, no reference counterpart: the candidate list of the cell's 8x8-cell tile
                                  * (stations that can be among the nearest TWX_MAX_NNGHS + 1 of any of its cells) holds
                                  * more stations than the selection kernel can rank: 15 872 per tile in the grid entries
                                  * (a batch with a list longer than 4 096 is run a second time with longer lists and
                                  * without the per-tile LDS tables: slower, same results), 4 096 under
                                  * TWX_FLAG_NO_HOST_SYNC (nothing is read back, so nothing can be re-run), 512 per point
                                  * in the point entries.  The cell is failed rather than ranked from a truncated list.
                                  * A station table of fewer than 15 872 stations can never reach it in the grid
                                  * entries (the reference's CONUS tables hold 11-12 000). */
#define TWX_CELL_MASKED (-1)    /* cell outside the interpolation mask: nothing computed */

/* netCDF4 default fill values the reference worker pre-fills with (step25:73-88) */
#define TWX_FILL_I2 ((int16_t)-32767)
#define TWX_FILL_F4 9.969209968386869e36f
#define TWX_FILL_I4 ((int32_t)-2147483647)

typedef struct twx_ctx twx_ctx;

/* Algorithm constants (module constants / hard-coded arguments in the reference) */
typedef struct {
    int32_t init_nnghs;      /* DFLT_INIT_NNGHS = 100, interp_tair.py:51 */
    int32_t fixer_tail;      /* tmin_tmax_fixer(tail=15), interp_tair.py:143 */
    int32_t norm_yr0;        /* 1981, interp_tair.py:468 */
    int32_t norm_yr1;        /* 2010 */
    int32_t tile_cells;      /* edge (cells) of the square candidate tile; 0 = default 8 */
    int32_t batch_cells;     /* cells per device batch; 0 = default */
    int32_t flags;           /* TWX_FLAG_* */
    int32_t reserved;
} twx_params;

/* address the observation matrix with 64-bit element offsets even when it is smaller than 4 GiB (the
 * path a > 4 GiB matrix takes -- a test / diagnostic switch).  It implies TWX_FLAG_DAILY_GATHER, see there for
 * how the results compare with the default path. */
#define TWX_FLAG_OBS_ADDR64 1
/* twx_interp_grid_dev / twx_stream_submit enqueue a whole batch without ANY host synchronisation: the kriging
 * launches then cover the worst case and surplus work-groups exit at once (about +9 % kriging time on the C2 tile).
 * Default (flag clear): one 64-byte read-back of the matrix-size counts per (batch, variable), i.e. the host
 * waits for the selection kernels; everything after them -- kriging, GWR, daily values, fixer -- is asynchronous. */
#define TWX_FLAG_NO_HOST_SYNC 2
/* daily values: gather every (cell, neighbour) observation row from global memory instead of staging the rows of a
 * tile-month in LDS (the path a tile-month with more than 208 distinct rows takes -- a test / diagnostic switch;
 * TWX_FLAG_OBS_ADDR64 implies it).  Every daily sum -- LDS-table walk, 32- / 64-bit gathers, single-variable requests,
 * the fixer's recompute, the point entries -- adds a cell-day's terms in ONE order, ascending station index (the
 * reference's neighbour order, station_select.py:179-182; a table row the cell does not use carries weight 0 and adds
 * exactly nothing): all paths give the same bits, so which path a tile-month takes (its station union <= 208 rows, i.e.
 * tiling and station density) does not show in any output (tests/test_gpu_configs.py).
 * Precondition of every daily path: observations are finite (the reference's database is serially complete,
 * station_data.py:547-616); twx_set_stations rejects a table whose obs hold NaN / Inf, because the table walk
 * multiplies every row of a tile-month by every cell's weight (0 for rows a cell does not use: 0 * NaN = NaN would
 * reach all 64 cells of the tile). */
#define TWX_FLAG_DAILY_GATHER 4
/* kriging: never route a system to the fp64 covariance build (diagnostic switch; tests/tools/gpu_closepair_scan.py
 * measures the fast build's error with it).  Default (flag clear): a system whose amplification
 * psill / (2 (nug + psill (1 - exp(-hmin / range)))) -- hmin = smallest distance between two of its neighbours --
 * exceeds 8 has its covariance matrix built from fp64 distances and fp64 exponentials (the <.., 1> instance of the kernel of its matrix size) instead of the
 * fp32 pair-distance cache + v_exp_f32: station pairs a few hundred metres apart with a nugget near 0 are legal
 * inputs of interp.R:223-231,256 (nugget = min gamma, interp.R:304-359) and amplify the fp32 rounding of an entry
 * beyond the 1e-4 degC parity bar (measured: tests/test_gpu_closepairs.py). */
#define TWX_FLAG_UK_FAST_ONLY 8
/* kriging: EVERY system on the fp64 covariance build (2.2-2.3 x the kriging time): the normals then agree with an fp64
 * evaluation of the reference's formulas to ~1e-11 degC instead of ~1e-6 (one ulp of the f4 outputs), and the 2e-5 of
 * the packed int16 daily values that sit within that 1e-6 of a rounding boundary stop flipping by one count
 * (tests/test_gpu_closepairs.py).  For comparisons against other fp64 implementations; not needed for the 1e-4 degC bar. */
#define TWX_FLAG_UK_F64_ALL 16
/* fixer: every flagged cell through the full recompute (k_fix_cells) -- a test / diagnostic switch.  Default (flag
 * clear): a cell with at most 256 invalid days is fixed from those days' +- fixer_tail windows alone (k_fix_sparse) when
 * that path is usable -- both variables have observations, the normals period present in the day axis is at most 40 years,
 * 2 fixer_tail + 1 <= 64 -- and by the full recompute otherwise; one predicate, evaluated on the host, decides for a whole
 * call.  Both paths give the same fixed days and the same packed int16 values (the window means are summed in day order
 * in both); their recomputed f8 normals can differ in the last bits (tests/test_gpu_parity.py). */
#define TWX_FLAG_FIX_FULL 32
/* fixer: switch the TIE GUARD off (a test / diagnostic switch: tests/tools/gpu_full_tile_parity.py measures what it catches).
 * Default (flag clear), grid entries with both variables and daily output: the fixer's test tmin >= tmax
 * (interp_tair.py:170) is a discontinuity -- a day whose Tmax - Tmin lies within the fast covariance build's ~1e-6 degC of 0
 * can fall on the other side of it than in an fp64 evaluation, and then the day moves by degrees, the cell's recomputed
 * normals by ~0.05 degC and its ninvalid by 1.  So every cell that has a day with |Tmax - Tmin| < 2e-5 degC is kriged a
 * second time on the fp64 covariance build and its whole series is recomputed from those normals: ninvalid and the fixed
 * days of every cell are then those of a TWX_FLAG_UK_F64_ALL run.  Costs one subtraction per cell-day plus ~25 us per such
 * cell (real data: about one cell in 1e5); twx_timing.tie_cells / tie_solves / tie_ms report it.  Not active under
 * TWX_FLAG_UK_F64_ALL (nothing to guard) and TWX_FLAG_UK_FAST_ONLY (no fp64 build wanted). */
#define TWX_FLAG_NO_TIE_GUARD 64

/* twx_set_precision modes */
#define TWX_PRECISION_FAST 0  /* default: fp32 pair distances + v_exp_f32, ill-conditioned systems and tie-guard cells on the fp64 build */
#define TWX_PRECISION_EXACT 1 /* every kriging system on the fp64 covariance build (= TWX_FLAG_UK_F64_ALL) */

/* Station table of ONE variable (replaces StationSerialDataDb.stns +
 * StationSelect's isnan(bad) mask: station_data.py:126-183,609,
 * interp_tair.py:483-487).  Good stations only, sorted by station id.  All
 * columns fp64 host arrays; NaN = missing (station_data.py:159-164). */
typedef struct {
    int64_t n;
    const double *lon, *lat, *elev, *tdi;              /* [n] */
    const double *lst, *norm;                          /* [12][n] */
    const double *optim_nnghs, *optim_nnghs_anom;      /* [12][n] */
    const double *vario_nug, *vario_psill, *vario_rng; /* [12][n] */
    const float *obs; /* [ndays][n] (time, station_id) as in the netCDF var
                         (station_data.py:578, post_infill.py:33); NULL = normals only */
} twx_station_table;

/* A point to interpolate to (replaces the structured scalar of
 * build_empty_pt, interp_tair.py:200-213); lst = LST of the variable asked for */
typedef struct {
    double lon, lat, elev, tdi;
    double lst[12];
} twx_pt;

/* Predictor planes of a grid of cells in their native dtypes (replaces the
 * f8[32,Y,X] wrk_chk the coordinator sends, tiling.py:190-213 / step25:136-144).
 * lat is north-up (descending), one value per row; lon one value per column. */
typedef struct {
    int32_t Y, X;
    const uint8_t *mask;    /* [Y][X] nonzero = interpolate */
    const double *lat;      /* [Y] */
    const double *lon;      /* [X] */
    const float *elev;      /* [Y][X] */
    const float *tdi;       /* [Y][X] */
    const int32_t *climdiv; /* [Y][X]; carried for format parity, never read (interp_tair.py:563 is dead) */
    const float *lst_night; /* [12][Y][X]  -> Tmin predictor (interp_tair.py:562) */
    const float *lst_day;   /* [12][Y][X]  -> Tmax predictor (interp_tair.py:571) */
} twx_grid;

/* Outputs of a grid call in the dtypes / layout of the reference's result
 * arrays (step25:68-88,163-172).  Any pointer may be NULL.  Every grid entry first
 * fills the buffers with the netCDF fill values below (TWX_FILL_*; status: every
 * cell is written), then writes the cells that succeed: failed and masked cells
 * hold the fill values, as in the reference's result arrays (step25:68-88,154-160). */
typedef struct {
    float *norm_tmin, *se_tmin; /* [12][Y][X] */
    float *norm_tmax, *se_tmax; /* [12][Y][X] */
    int16_t *daily_tmin;        /* [ndays][Y][X], degC * 100 (step25:163) */
    int16_t *daily_tmax;
    int32_t *ninvalid;          /* [Y][X] days fixed by tmin_tmax_fixer */
    int32_t *status;            /* [Y][X] TWX_CELL_* */
} twx_grid_out;

/* Device time of the kernels of the last grid call, measured with HIP events on
 * the call's stream (ms); filled when the call has completed.  The system
 * counters (uk_solves, uk_launches, uk_f64_solves) are those of the last entry
 * call, grid or points; the times are 0 after a point entry. */
typedef struct {
    float tile_cand_ms, select_ms, uk_ms, gwr_ms, daily_ms, fix_ms, total_ms;
    int64_t cells;        /* unmasked cells processed */
    int64_t uk_solves;    /* (cell, month, variable) kriging systems solved */
    int64_t uk_launches;  /* kernel launches of the kriging kernel */
    int64_t uk_f64_solves; /* of uk_solves: ill-conditioned systems that took the fp64 covariance build (TWX_FLAG_UK_FAST_ONLY) */
    int64_t tie_cells;    /* tie guard (TWX_FLAG_NO_TIE_GUARD): cells with a day of |Tmax - Tmin| < 2e-5 degC, kriged a second time */
    int64_t tie_solves;   /* ... and their kriging systems (both variables; not counted in uk_solves) */
    float tie_ms;         /* ... and the device time of that second pass (kriging + epilogues; their fixer time is in fix_ms) */
    float deflate_ms;     /* twx_stream_deflate: device time of the deflate kernels of the last tile (0 otherwise) */
} twx_timing;

/* ---- lifetime ---------------------------------------------------------- */
int twx_create(int device, const twx_params *params, twx_ctx **out);
void twx_destroy(twx_ctx *ctx);
const char *twx_last_error(const twx_ctx *ctx);
const char *twx_version(void);

/* Covariance build of the kriging systems of every later call on this context: TWX_PRECISION_FAST (the default routing, see
 * TWX_FLAG_UK_FAST_ONLY / TWX_FLAG_NO_TIE_GUARD) or TWX_PRECISION_EXACT (all systems on the fp64 build: what the flag
 * TWX_FLAG_UK_F64_ALL selects at twx_create).  A run that is bound by its copy-out (daily tiles streamed to the host: the GPU
 * idles most of the wall) can afford EXACT for nothing -- topowx_amd/driver.py decides per run (precision="auto").
 * No reference counterpart (the reference is fp64 throughout: interp.R:256). */
int twx_set_precision(twx_ctx *ctx, int mode);

/* day axis of the observation matrix: replaces StationSerialDataDb.days /
 * mth_idx (station_data.py:570-576) and PtInterpTair's normals masks
 * (interp_tair.py:468-481) */
int twx_set_days(twx_ctx *ctx, int64_t ndays, const int32_t *day_month, const int32_t *day_year);

/* replaces StationSerialDataDb(...) + StationSelect(stn_da, isnan(bad))
 * (interp_tair.py:483-487); uploads and re-lays the table and obs matrix */
int twx_set_stations(twx_ctx *ctx, int var, const twx_station_table *tbl);

/* ---- per-point entries (host buffers) ----------------------------------- */
/* stns_rm as an ARRAY of station ids (station_select.py:74-103 accepts any number and removes them with np.in1d): up to
 * TWX_MAX_EXCL station indices per point for the NEXT point-entry call on this context (twx_knn, twx_krig_points,
 * twx_krigall_points, twx_fit_vario_points, twx_gwr_points, twx_gwr_xval_points, twx_interp_points), whose npts must
 * equal this call's.  lists [npts][nmax], entries < 0 = unused, 1 <= nmax <= TWX_MAX_EXCL.  The entry's own `excl` argument
 * (one index per point -- what every caller on the reference's path passes) still applies: the union is removed.
 * Consumed by the next point-entry call that passes its argument checks, whether it then succeeds or not; npts = 0 clears a
 * pending list.  Library limit without reference
 * counterpart: more than TWX_MAX_EXCL ids per point are refused here (call-level failure), never truncated. */
#define TWX_MAX_EXCL 8
int twx_set_exclusions(twx_ctx *ctx, int64_t npts, int32_t nmax, const int32_t *lists);

/* StationSelect.set_ngh_stns (station_select.py:121-192).  excl: station index
 * to drop (stns_rm) or -1, per point, may be NULL.  Outputs [npts][k] in
 * ascending station-index (= id) order. */
int twx_knn(twx_ctx *ctx, int var, int64_t npts, const double *lon, const double *lat,
            int32_t k, const int32_t *excl, int rm_zero_dist, int32_t *idx, double *dist,
            double *wgt, int32_t *status);

/* KrigTair.krig(pt, mth, nnghs=None, vario_params=None, stns_rm=None)
 * (interp_tair.py:853-926) for npts independent (point, month) pairs.
 * nnghs: NULL or entries <= 0 -> smoothed from neighbours (:821-835);
 * vario: NULL or NaN nugget -> smoothed (:837-851).
 * ngh_idx (optional) [npts][TWX_MAX_NNGHS]: neighbours used, ascending, -1 padded. */
int twx_krig_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const int32_t *mth,
                    const int32_t *nnghs, const double *vario /*[npts][3]*/,
                    const int32_t *excl, int rm_zero_dist, double *mean, double *variance,
                    int32_t *nnghs_used, int32_t *ngh_idx, int32_t *status);

/* GwrTairAnom.gwr_mth(pt, mth, nnghs=None, stns_rm=None) (interp_tair.py:261-314).
 * pt_norm = pt[normMM].  out[npts][ld]: the days of mth in chronological order. */
int twx_gwr_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const double *pt_norm,
                   const int32_t *mth, const int32_t *nnghs, const int32_t *excl,
                   int rm_zero_dist, double *out, int64_t ld, int32_t *nnghs_used,
                   int32_t *status);

/* XvalTairAnom.run_xval's inner step (twx/interp/optimize.py:521-541) for npts (station, bandwidth, month)
 * points at once, statistics computed on the device: the GWR series of gwr_mth(pt, mth, nnghs, stns_rm) is
 * compared with the observations of station obs_idx[i] (the left-out station).  With norm = pt_norm[i]:
 * xval_anom = obs - norm, interp_anom = series - norm, difs = interp_anom - xval_anom;
 * bias = mean(difs), mae = mean(|difs|), r2 = squared correlation of interp_anom and xval_anom
 * (stats.linregress(...)[2] ** 2).  Failed points keep the caller's values. */
int twx_gwr_xval_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const double *pt_norm,
                        const int32_t *mth, const int32_t *nnghs, const int32_t *excl, int rm_zero_dist,
                        const int32_t *obs_idx, double *bias, double *mae, double *r2,
                        int32_t *nnghs_used, int32_t *status);

/* InterpTair.interp(pt, stns_rm) (interp_tair.py:396-439): 12 x (krig, gwr).
 * daily [npts][ndays] degC (NULL = normals only), norms/se [npts][12]. */
int twx_interp_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts,
                      const int32_t *excl, int rm_zero_dist, double *daily, double *norms,
                      double *se, int32_t *status);

/* KrigTairAll.krigall(pt, nnghs, stns_rm) (interp_tair.py:722-769) -> R krig_all (interp.R:148-159): the variogram
 * of each (point, month)'s neighbourhood is fitted (twx_fit_vario_points) and the SAME neighbourhood is kriged with the
 * fitted model (twx_krig_points with that variogram) -- in one call: one station selection and one set of pair
 * distances serve both stages, the fitted parameters never leave the device.  Same results, bit for bit, as the two
 * calls in sequence (tests/test_gpu_xval.py).  What step21's workers run per (station, bandwidth, month)
 * (optimize.py:236-266).  variance / vario / nnghs_used may be NULL; status = the first failure of either stage. */
int twx_krigall_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const int32_t *mth,
                       const int32_t *nnghs, const int32_t *excl, int rm_zero_dist, double *mean, double *variance,
                       double *vario, int32_t *nnghs_used, int32_t *status);

/* BuildKrigParams.get_krig_params(pt, mth) (interp_tair.py:635-698) and the first half of R
 * krig_all (interp.R:148-159): R get_vario_params (interp.R:54-113) -- OLS-residual variogram,
 * range fit, GLS-residual variogram, range fit -- on the nnghs nearest stations of each (point,
 * month).  nnghs NULL / <= 0 -> smoothed bandwidth (:665-678).  vario[npts][3] = (nug, psill, range),
 * (nug, 0, 0) for a pure-nugget result.  SURVEY.md 8f-1; parity with gstat unpinned. */
int twx_fit_vario_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const int32_t *mth,
                         const int32_t *nnghs, const int32_t *excl, int rm_zero_dist, double *vario,
                         int32_t *nnghs_used, int32_t *status);

/* tmin_tmax_fixer + normals recompute (interp_tair.py:143-197,579-590) on
 * nseries independent [ndays] degC series, in place.  norm_* [nseries][12] are
 * overwritten only for series with ninvalid > 0 (may be NULL). */
int twx_fix_pair(twx_ctx *ctx, int64_t nseries, double *tmin, double *tmax, int32_t *ninvalid,
                 double *norm_tmin, double *norm_tmax, int32_t *status);

/* int16 packing of daily degC (step25:44,163-164) */
int twx_pack_i16(twx_ctx *ctx, int64_t n, const double *x, int16_t *out);

/* ---- grid entries: the step25 worker loop (step25:126-172) ---------------- */
/* vars: TWX_VAR_*_BIT mask.  With both variables and daily outputs the
 * Tmin>=Tmax fixer runs as in PtInterpTair.interp_pt (interp_tair.py:526-592). */
int twx_interp_grid(twx_ctx *ctx, const twx_grid *grid, const twx_grid_out *out, int vars);

/* same, but every pointer in grid / out is a DEVICE pointer and the work is
 * enqueued on hip_stream (a hipStream_t; NULL = default stream).  Candidate lists and
 * the list of cells for the Tmin >= Tmax fixer are sized and counted on the device;
 * the matrix-size counts are read back once per (batch, variable) unless
 * TWX_FLAG_NO_HOST_SYNC is set (see there).  Inputs must stay valid until the stream
 * has drained.  (The first call of a shape may allocate workspace, which
 * synchronises the device once.) */
int twx_interp_grid_dev(twx_ctx *ctx, const twx_grid *grid_dev, const twx_grid_out *out_dev,
                        int vars, void *hip_stream);

/* ---- streamed tiles: the worker writes every chunk as soon as it is finished (step25:177-185,
 * tiling.py:488-537).  A twx_stream owns two device images and nslots pinned host blocks for tiles of one shape;
 * twx_stream_submit copies the predictors of a tile (host pointers) up, enqueues its kernels on the stream's own
 * compute stream and the copy-out of all outputs on a second (copy) stream: the copy-out of
 * tile t overlaps the kernels of tile t + 1.  The call is fully asynchronous only with TWX_FLAG_NO_HOST_SYNC; by
 * default it returns once the tile's selection kernels are done (one 64-byte read-back per variable, i.e. after the
 * previous tile's kernels have drained), with kriging, GWR, daily values, fixer and copy-out still in flight.  twx_stream_wait blocks until the slot's outputs are in host memory
 * and returns pointers into the slot's pinned block (valid until the slot is submitted again); device_ms
 * (optional) = device time of the tile's kernels.  One stream per context at a time; do not mix with other
 * calls on the context while tiles are in flight.  twx_destroy destroys a context's open streams (their handles are
 * dead afterwards: do not pass them to twx_stream_destroy). */
typedef struct twx_stream twx_stream;
int twx_stream_create(twx_ctx *ctx, int Y, int X, int vars, int daily, int nslots, twx_stream **out);
int twx_stream_submit(twx_stream *st, int slot, const twx_grid *grid);
int twx_stream_wait(twx_stream *st, int slot, twx_grid_out *views, float *device_ms);
/* after twx_stream_wait(slot): device time of the tile's kernels and of its copy-out to the pinned block (HIP events on the
 * stream's compute / copy streams; ms).  What precision="auto" of topowx_amd/driver.py compares. */
int twx_stream_times(twx_stream *st, int slot, float *device_ms, float *copy_ms);
void twx_stream_destroy(twx_stream *st);

/* ---- deflated daily outputs of a streamed tile.  The reference stores its products through netCDF4-python with zlib=True
 * (tiling.py:720,894,913,1035), i.e. as HDF5 chunks filtered by shuffle + deflate; on the host that is one core per ~60 MB/s.
 * twx_stream_deflate (once, before the first submit of a stream created with daily != 0; Y % chunk_y == 0, X % chunk_x == 0) makes
 * the stream form those chunk bytes ON THE DEVICE (csrc/twx_deflate.h): per variable and chunk of chunk_y x chunk_x cells x all
 * days one zlib stream (RFC 1950) of the shuffled chunk -- low bytes in stored blocks, high bytes run-length coded in
 * dynamic-Huffman blocks, one code per variable and tile -- that any inflate reads and H5Dwrite_chunk appends to a dataset
 * created with shuffle + deflate as it is (zlib level 1's size on the same chunks).
 * The daily arrays then stay on the device (views.daily_* are NULL) and the tile leaves it at ~0.5-0.6 of its size.
 * twx_stream_wait_deflated replaces twx_stream_wait for such a stream: the copy-out of a tile is enqueued by THIS call (the sizes
 * of its streams are known only when its kernels are done), so call it for tile t after submitting tile t + 1, as a pipelined
 * caller does anyway; a tile still not waited for when its slot or its device image is needed again is copied out by
 * twx_stream_submit itself.  Chunks are in row-major order of the tile's chunk grid. */
typedef struct twx_deflated {
    const uint8_t *data[2];   /* [TWX_TMIN], [TWX_TMAX]: the variable's chunk streams one after the other (pinned; valid until the
                                 slot is submitted again); NULL for a variable the stream does not have */
    const int64_t *offset[2]; /* [nchunks + 1] byte offsets of the chunks in data[v] */
    int32_t nchunks, chunk_y, chunk_x, reserved;
} twx_deflated;
int twx_stream_deflate(twx_stream *st, int chunk_y, int chunk_x);
int twx_stream_wait_deflated(twx_stream *st, int slot, twx_grid_out *views, twx_deflated *streams, float *device_ms);

/* free / total device memory of the context's GPU in bytes (hipMemGetInfo): what a caller sizes its batches by
 * (topowx_amd/xval.py: step21 pushes ~22 MB of workspace per cross-validated station through one call) */
int twx_device_memory(twx_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);

/* kernel times of the last grid call (synchronises on its events) */
int twx_get_timing(twx_ctx *ctx, twx_timing *t);

/* Diagnostic: the kriging bandwidths nnghs (KrigTair.__get_nnghs, interp_tair.py:821-835) of the
 * LAST device batch of the last grid / point call of variable var, [cells][12] (0 = month not
 * solved).  Copies min(capacity, cells*12) values; returns the number of cells in that batch or < 0. */
int64_t twx_last_bandwidths(twx_ctx *ctx, int var, int32_t *nnghs, int64_t capacity);

/* ---- monthly / annual aggregation of the daily product --------------------
 * Replaces _TairAggregate (twx/interp/tiling.py:1080-1166: daily_to_mthly, daily_to_ann,
 * mthly_to_ann) and the rounding + int16 packing of write_ds_mthly (tiling.py:1169-1219, driven by
 * scripts/step27_create_monthly.py).  The day axis is the one given to twx_set_days; groups are
 * (unique years) x (unique months) in year-major order, a (year, month) without days is an empty
 * group (tiling.py:1101-1107). */
#define TWX_DT_I16 0 /* raw 'i2' product: value = int16 * float32(0.01), -32767 = masked (tiling.py:36,448) */
#define TWX_DT_F32 1 /* degC, NaN = masked */
#define TWX_DT_F64 2

/* number of years / of distinct months on the day axis: mthly has nyr*nmth planes, ann has nyr */
int twx_aggregate_dims(twx_ctx *ctx, int32_t *nyr, int32_t *nmth);

/* daily [ndays][ncell] of dtype -> any of
 *   mthly     f8 [nyr*nmth][ncell]  group means, NaN = masked        (daily_to_mthly)
 *   mthly_i16 i2 [nyr*nmth][ncell]  np.ma.round(mean, 2) packed with scale_factor float32(0.01),
 *                                    -32767 = masked                   (write_ds_mthly)
 *   ann       f8 [nyr][ncell]       mean of the year's monthly means  (daily_to_ann / mthly_to_ann)
 * on_device != 0: all pointers are device pointers and the kernel is enqueued on hip_stream.
 * kernel_ms (optional): device time of the aggregation kernel (HIP events; synchronises). */
int twx_aggregate(twx_ctx *ctx, const void *daily, int dtype, int64_t ncell, int on_device,
                  double *mthly, int16_t *mthly_i16, double *ann, void *hip_stream, float *kernel_ms);

/* ---- point-mode predictor sampling -----------------------------------------
 * Replaces PredictorGrids.setPtValues (twx/interp/interp_tair.py:115-141) behind
 * PtInterpTair.interp_to_lonlat (:513-524).  order 0: the raster cell of GeoNc.get_row_col
 * (twx/utils/util_ncdf.py:292-301); status 1 where the reference raises IndexError (cell outside
 * the raster).  order 1: bilinear (mpl_toolkits.basemap.interp order=1, masked), else nearest, else
 * `missing`; row / col are -1. */
typedef struct {
    int32_t nrows, ncols;
    const double *lon; /* [ncols] cell-centre longitudes, ascending */
    const double *lat; /* [nrows] cell-centre latitudes, descending (north-up) */
    const float *data; /* [nrows][ncols], NaN = missing */
} twx_raster;

int twx_sample_points(twx_ctx *ctx, const twx_raster *raster, int64_t npts, const double *lon,
                      const double *lat, int order, double missing, double *val, int32_t *row,
                      int32_t *col, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* TWX_H */
