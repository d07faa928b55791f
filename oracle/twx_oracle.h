/*
 * twx_oracle.h -- CPU restatement (plain C, fp64) of the TopoWx interpolation
 * hot path.  TEST INFRASTRUCTURE ONLY: the checker the HIP path is compared
 * against and the CPU baseline that bench.py times.  Nothing in the product
 * path (topowx_amd/) may call into this library.
 *
 * PARITY STATUS
 *   - station selection, bisquare weights, nnghs / variogram smoothing, GWR,
 *     Tmin>=Tmax fixer, normals recompute, int16 packing: PINNED against
 *     golden vectors produced by executing the reference's own source slices
 *     (tests/golden/make_golden.py).
 *   - universal-kriging solve (orc_uk): PARITY UNPINNED.  The arithmetic lives
 *     in gstat 1.0-25 / sp 1.1-1 (R 3.2.0), which the reference calls
 *     (twx/interp/rpy/interp.R:256) but does not vendor and which cannot run
 *     here.  It is restated from interp.R:198-270 + the published gstat/sp
 *     algorithm and pinned by known-answer properties
 *     (tests/test_oracle_uk.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef TWX_ORACLE_H
#define TWX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-cell / per-point status (same numbering as include/twx.h) */
enum {
    ORC_OK = 0,
    ORC_ERR_FEW_STATIONS = 1, /* IndexError: stn_dists[nnghs], station_select.py:164 */
    ORC_ERR_NNGHS = 2,        /* "Cannot determine the optimal # of neighbors" interp_tair.py:252,829 */
    ORC_ERR_VARIO = 3,        /* "Cannot determine variogram params!" interp_tair.py:843 */
    ORC_ERR_NUMERIC = 4,      /* FloatingPointError (np.seterr all='raise', step25:319) / singular system */
    ORC_ERR_FIXER = 5,        /* 'No valid tmin/tmax in window' interp_tair.py:192 */
    ORC_ERR_RANGE = 6         /* nnghs outside the supported range */
};

/* Station database of ONE variable: good stations only (isnan(bad),
 * interp_tair.py:483-487), sorted by station id (SURVEY.md a2 note). */
typedef struct {
    int64_t n;
    const double *lon, *lat, *elev, *tdi;               /* [n] */
    const double *lst, *norm;                           /* [12][n] */
    const double *optim_nnghs, *optim_nnghs_anom;       /* [12][n], NaN = missing */
    const double *vario_nug, *vario_psill, *vario_rng;  /* [12][n], NaN = missing */
    const float *obs;          /* [ndays][n] (time, station_id) or NULL */
    int64_t ndays;
    const int32_t *day_month;  /* [ndays] 1..12 */
    const int32_t *day_year;   /* [ndays] */
} orc_db;

typedef struct {
    double lon, lat, elev, tdi;
    double lst[12];  /* LST of the variable being interpolated */
} orc_pt;

typedef struct {
    int32_t init_nnghs;       /* DFLT_INIT_NNGHS = 100, interp_tair.py:51 */
    int32_t fixer_tail;       /* 15, interp_tair.py:143 */
    int32_t norm_yr0, norm_yr1; /* 1981, 2010, interp_tair.py:468 */
} orc_params;

/* a1: util_geo.py:24-40 */
double orc_grt_circle_dist(double lon1, double lat1, double lon2, double lat2);
/* B.1: sp/gstat WGS84 great-circle distance (km) */
double orc_ellip_dist(double lon1, double lat1, double lon2, double lat2);

/* a2: station_select.py:72-119.  The ksel nearest remaining stations in
 * (distance, index) order.  Returns the number written (< ksel if fewer). */
/* stns_rm as an array of ids (station_select.py:74-103): up to 16 further indices dropped by orc_nearest for this thread's next calls */
void orc_set_exclusions(int n, const int32_t *idx);
int64_t orc_nearest(const orc_db *db, double lat, double lon, int32_t excl,
                    int rm_zero_dist, int64_t ksel, int32_t *idx, double *dist);

/* a2: station_select.py:121-192 given the nearest list; outputs in ascending
 * station-index (= id) order. */
int orc_select(const int32_t *near_idx, const double *near_dist, int64_t nnear,
               int k, int32_t *idx, double *dist, double *wgt);

/* a3: interp_tair.py:821-835 / :245-259 */
int orc_smooth_nnghs(const double *optim_m, const int32_t *idx, const double *wgt,
                     int k, int *nnghs);
/* a4: interp_tair.py:837-851 */
int orc_smooth_vario(const double *nug_m, const double *psill_m, const double *rng_m,
                     const int32_t *idx, const double *wgt, int k, double vario[3]);

/* a5: interp.R:198-270 -> gstat::krige (Appendix B.2).  X rows: lon,lat,elev,lst */
int orc_uk(int k, const double *lon, const double *lat, const double *elev,
           const double *lst, const double *y, double plon, double plat,
           double pelev, double plst, double nug, double psill, double rng,
           double *mean, double *var);

/* a7: interp_tair.py:1099-1146 hat row z = x0' (X'WX)^-1 X'W; X rows:
 * lon,lat,elev,tdi,lst (intercept added inside). */
int orc_gwr_hat(int k, const double *X5 /*[k][5]*/, const double *w,
                const double *x5, double *z);

/* a5 orchestration: interp_tair.py:853-926 */
int orc_krig(const orc_db *db, const orc_params *p, const orc_pt *pt, int mth,
             int nnghs /*<=0: smooth*/, const double *vario /*NULL: smooth*/,
             int32_t excl, int rm_zero_dist, double *mean, double *var,
             int *nnghs_used, int32_t *ngh_idx /*optional [nnghs]*/);

/* a7 orchestration: interp_tair.py:261-314.  out[D_m] for the days of mth in
 * chronological order. */
int orc_gwr_mth(const orc_db *db, const orc_params *p, const orc_pt *pt,
                double pt_norm, int mth, int nnghs /*<=0: smooth*/, int32_t excl,
                int rm_zero_dist, double *out, int *nnghs_used,
                double *z /*optional*/, int32_t *ngh_idx /*optional*/);

/* a8: InterpTair.interp interp_tair.py:396-439.  daily may be NULL (normals). */
int orc_interp(const orc_db *db, const orc_params *p, const orc_pt *pt,
               int32_t excl, int rm_zero_dist, double *daily, double *norms,
               double *se);

/* a9: tmin_tmax_fixer interp_tair.py:143-197 (in place) */
int orc_fixer(double *tmin, double *tmax, int64_t ndays, int tail, int32_t *ninvalid);
/* a9: normals recompute interp_tair.py:583-590 */
void orc_recompute_norms(const double *daily, int64_t ndays, const int32_t *day_month,
                         const int32_t *day_year, int yr0, int yr1, double *norms);
/* a12: step25:163-164 */
void orc_pack_i16(const double *x, int64_t n, int16_t *out);

/* step25:126-172 over a grid of cells.  Planes are the native-dtype predictor
 * planes (tiling.py:190-213).  Outputs pre-filled with fill values by the
 * caller; failed / masked cells are left untouched.  nthreads: OpenMP threads.
 * daily_* may be NULL (normals only; fixer skipped). */
int orc_interp_grid(const orc_db *tmin, const orc_db *tmax, const orc_params *p,
                    int Y, int X, const uint8_t *mask, const double *lat,
                    const double *lon, const float *elev, const float *tdi,
                    const float *lst_night, const float *lst_day,
                    float *norm_tmin, float *se_tmin, float *norm_tmax, float *se_tmax,
                    int16_t *daily_tmin, int16_t *daily_tmax, int32_t *ninvalid,
                    int32_t *status, int nthreads);

/* ---- second tier: variogram estimation + range fit ---------------------
 * R get_vario_params (interp.R:54-113) with my.autofit.gwvario (:290-428) -> gstat
 * variogram / fit.variogram / predict(BLUE=TRUE).  PARITY UNPINNED (gstat not runnable, its
 * optimiser's stopping rules are restated, not copied): see twx_oracle.c for the exact scheme. */

/* binned semivariogram of residuals e (interp.R:64,86): pairs within cutoff, 5 km bins of the
 * B.1 distance; returns the number of non-empty bins; arrays sized >= cutoff/width + 2 */
int orc_variogram(int k, const double *lon, const double *lat, const double *e, double cutoff,
                  double width, double *dist, double *gamma, double *np);

/* one-parameter weighted fit of the range of nug + psill (1 - exp(-h/range)), weights np/h^2
 * (fit.method = 7), Gauss-Newton from range0 with step halving; returns 0 and *range, or 1 if the
 * fit is unusable (the R code then falls back to a pure nugget, interp.R:73-80) */
int orc_fit_range(int nbin, const double *dist, const double *gamma, const double *np, double nug,
                  double psill, double range0, double *range);

/* get_vario_params (interp.R:54-113): vario = (nug, psill, range) or (nug, 0, 0) */
int orc_get_vario_params(int k, const double *lon, const double *lat, const double *elev,
                         const double *lst, const double *y, double max_ngh_dist, double vario[3]);

/* BuildKrigParams.get_krig_params (interp_tair.py:635-698): smoothed nnghs, no exclusion */
int orc_build_krig_params(const orc_db *db, const orc_params *p, const orc_pt *pt, int mth,
                          double vario[3], int *nnghs_used);

/* KrigTairAll.krigall (interp_tair.py:722-769) -> R krig_all (interp.R:148-159): per month fit
 * the variogram on the nnghs nearest (stns_rm / zero-distance removed) and krige with it */
int orc_krigall(const orc_db *db, const orc_params *p, const orc_pt *pt, int nnghs, int32_t excl,
                int rm_zero_dist, double norms[12], double vario_out[36] /*optional*/);

/* ---- second tier: monthly / annual aggregation ----------------------------
 * _TairAggregate (tiling.py:1080-1166) and the rounding + packing of write_ds_mthly
 * (tiling.py:1169-1219).  PINNED against the executed _TairAggregate slice
 * (tests/golden/make_golden_agg.py); the netCDF4 unpack / pack arithmetic around it is restated.
 * dtype: 0 = raw int16 (scale float32(0.01), fill -32767), 1 = f4, 2 = f8 (NaN = masked). */
int orc_agg_groups(int64_t ndays, const int32_t *day_year, const int32_t *day_month, int32_t *nyr,
                   int32_t *nmth, int32_t *day_group);
void orc_daily_to_mthly(const void *daily, int dtype, int64_t ndays, int64_t ncell,
                        const int32_t *day_group, int ng, double *mthly);
void orc_mthly_to_ann(const double *mthly, int nyr, int nmth, int64_t ncell, double *ann);
void orc_pack_mthly_i16(const double *x, int64_t n, int16_t *out);

/* ---- second tier: point-mode predictor sampling -----------------------------
 * GeoNc.get_row_col (util_ncdf.py:262-301; PINNED against the executed slice) and
 * PredictorGrids.setPtValues (interp_tair.py:115-141; order 1 goes through mpl_toolkits.basemap.interp,
 * not vendored: restated, PARITY UNPINNED). */
int orc_get_row_col(int nrows, int ncols, const double *lons, const double *lats, double lon, double lat,
                    int32_t *row, int32_t *col);
int orc_sample_point(int nrows, int ncols, const double *lons, const double *lats, const float *data,
                     double lon, double lat, int order, double missing, double *val, int32_t *row,
                     int32_t *col);

#ifdef __cplusplus
}
#endif
#endif
