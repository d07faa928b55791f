/*
 * twx_oracle.c -- CPU restatement of the TopoWx interpolation hot path.
 * TEST INFRASTRUCTURE ONLY (see twx_oracle.h for the parity status of each
 * part; the universal-kriging solve is PARITY UNPINNED).
 *
 * Written from the behaviour of the reference (file:line cited per function,
 * relative to /root/reference) and, for the kriging solve, from the published
 * gstat / sp algorithm.  Plain C99, fp64 throughout.
 */
#define _GNU_SOURCE
#include "twx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RADIAN_CONVERSION_FACTOR 0.017453292519943295 /* util_geo.py:21 */
#define AVG_EARTH_RADIUS_KM 6371.009                  /* util_geo.py:22 */
#define ORC_MAXK 512

/* ---------------------------------------------------------------- a1 ---- */
/* util_geo.py:24-40: haversine on the IUGG mean sphere, same operation order. */
double orc_grt_circle_dist(double lon1, double lat1, double lon2, double lat2)
{
    double lat1rad = lat1 * RADIAN_CONVERSION_FACTOR;
    double lat2rad = lat2 * RADIAN_CONVERSION_FACTOR;
    double lon1rad = lon1 * RADIAN_CONVERSION_FACTOR;
    double lon2rad = lon2 * RADIAN_CONVERSION_FACTOR;
    double dlat = lat1rad - lat2rad;
    double dlon = lon1rad - lon2rad;
    double s1 = sin(dlat / 2), s2 = sin(dlon / 2);
    double ca = 2 * asin(sqrt(s1 * s1 + cos(lat1rad) * cos(lat2rad) * (s2 * s2)));
    return AVG_EARTH_RADIUS_KM * ca;
}

/* Appendix B.1: sp / gstat great-circle distance on the WGS84 ellipsoid
 * (Andoyer-Lambert form, km).  Reached because the data carry a longlat CRS
 * (interp.R:218-221).  [upstream-recall: parity unpinned] */
double orc_ellip_dist(double lon1, double lat1, double lon2, double lat2)
{
    const double a = 6378.137, f = 1.0 / 298.257223563;
    const double eps = 2.220446049250313e-16;
    if (fabs(lat1 - lat2) < eps) {
        if (fabs(lon1 - lon2) < eps)
            return 0.0;
        if (fabs((fabs(lon1) + fabs(lon2)) - 360.0) < eps)
            return 0.0;
    }
    double de2ra = M_PI / 180.0;
    double lat1r = lat1 * de2ra, lat2r = lat2 * de2ra;
    double lon1r = lon1 * de2ra, lon2r = lon2 * de2ra;
    double F = (lat1r + lat2r) / 2.0, G = (lat1r - lat2r) / 2.0, L = (lon1r - lon2r) / 2.0;
    double sinG2 = sin(G) * sin(G), cosG2 = cos(G) * cos(G);
    double sinF2 = sin(F) * sin(F), cosF2 = cos(F) * cos(F);
    double sinL2 = sin(L) * sin(L), cosL2 = cos(L) * cos(L);
    double S = sinG2 * cosL2 + cosF2 * sinL2;
    double C = cosG2 * cosL2 + sinF2 * sinL2;
    double w = atan(sqrt(S / C));
    double R = sqrt(S * C) / w;
    double D = 2 * w * a;
    double H1 = (3 * R - 1) / (2 * C);
    double H2 = (3 * R + 1) / (2 * S);
    return D * (1 + f * H1 * sinF2 * cosG2 - f * H2 * cosF2 * sinG2);
}

/* ---------------------------------------------------------------- a2 ---- */
typedef struct { double d; int32_t i; } di_t;

static int di_less(const di_t *a, const di_t *b)
{
    return a->d < b->d || (a->d == b->d && a->i < b->i);
}

static void heap_sift_down(di_t *h, int64_t n, int64_t i)
{
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && di_less(&h[m], &h[l])) m = l;
        if (r < n && di_less(&h[m], &h[r])) m = r;
        if (m == i) return;
        di_t t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static int di_cmp(const void *a, const void *b)
{
    const di_t *x = a, *y = b;
    return di_less(x, y) ? -1 : (di_less(y, x) ? 1 : 0);
}

/* station_select.py:72-119: distances to ALL stations, ordered ascending,
 * excluded ids (and, if rm_zero_dist, zero-distance stations) dropped after the
 * sort.  Only the first ksel entries are ever read (station_select.py:164-166),
 * so a bounded max-heap replaces the full argsort; ties (undefined under the
 * reference's unstable argsort, :111) break on the smaller station index. */
/* stns_rm as an ARRAY of ids (station_select.py:74-103: np.in1d over any number of ids): further station indices that
 * orc_nearest drops, besides its own excl argument, for the calls of THIS thread that follow (n = 0 clears).  Thread-local:
 * the threaded grid entry never sets it. */
static __thread int32_t orc_more_excl[16];
static __thread int orc_n_more_excl = 0;
void orc_set_exclusions(int n, const int32_t *idx)
{
    orc_n_more_excl = n < 0 ? 0 : (n > 16 ? 16 : n);
    for (int i = 0; i < orc_n_more_excl; ++i) orc_more_excl[i] = idx[i];
}

int64_t orc_nearest(const orc_db *db, double lat, double lon, int32_t excl,
                    int rm_zero_dist, int64_t ksel, int32_t *idx, double *dist)
{
    if (ksel > db->n) ksel = db->n;
    if (ksel <= 0) return 0;
    di_t *h = malloc(sizeof(di_t) * (size_t)ksel);
    int64_t nh = 0;
    for (int64_t j = 0; j < db->n; ++j) {
        if (j == excl) continue;
        {
            int dropped = 0;
            for (int q = 0; q < orc_n_more_excl; ++q) dropped |= orc_more_excl[q] == j;
            if (dropped) continue;
        }
        double d = orc_grt_circle_dist(lon, lat, db->lon[j], db->lat[j]);
        if (rm_zero_dist && d == 0.0) continue;
        di_t e = { d, (int32_t)j };
        if (nh < ksel) {
            h[nh++] = e;
            if (nh == ksel)
                for (int64_t i = nh / 2 - 1; i >= 0; --i) heap_sift_down(h, nh, i);
        } else if (di_less(&e, &h[0])) {
            h[0] = e;
            heap_sift_down(h, nh, 0);
        }
    }
    qsort(h, (size_t)nh, sizeof(di_t), di_cmp);
    for (int64_t i = 0; i < nh; ++i) { idx[i] = h[i].i; dist[i] = h[i].d; }
    free(h);
    return nh;
}

static int i32_cmp_perm(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* station_select.py:121-192: k nearest, bandwidth = distance of the (k+1)-th
 * (:164), bisquare weights (:169), re-ordered by ascending station id (:179-182;
 * the table is id-sorted so index order == id order). */
int orc_select(const int32_t *near_idx, const double *near_dist, int64_t nnear,
               int k, int32_t *idx, double *dist, double *wgt)
{
    if (k < 0 || k >= nnear) return ORC_ERR_FEW_STATIONS; /* IndexError at :164 */
    double dbw = near_dist[k];
    if (!(dbw > 0.0)) return ORC_ERR_NUMERIC;             /* 0/0 or x/0 -> FloatingPointError */
    /* sort positions by station index */
    int32_t key[ORC_MAXK][2];
    if (k > ORC_MAXK) return ORC_ERR_RANGE;
    for (int i = 0; i < k; ++i) { key[i][0] = near_idx[i]; key[i][1] = i; }
    qsort(key, (size_t)k, sizeof(key[0]), i32_cmp_perm);
    for (int i = 0; i < k; ++i) {
        int p = key[i][1];
        double r = near_dist[p] / dbw;
        double t = 1.0 - r * r;
        idx[i] = near_idx[p];
        dist[i] = near_dist[p];
        wgt[i] = t * t;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------- a3, a4 ---- */
/* interp_tair.py:821-835 (and :245-259): weighted mean of the neighbours'
 * optimal bandwidth over finite entries, np.round (half-even) -> int. */
int orc_smooth_nnghs(const double *optim_m, const int32_t *idx, const double *wgt,
                     int k, int *nnghs)
{
    double num = 0.0, den = 0.0;
    int cnt = 0;
    for (int i = 0; i < k; ++i) {
        double v = optim_m[idx[i]];
        if (isfinite(v)) { num += v * wgt[i]; den += wgt[i]; ++cnt; }
    }
    if (cnt == 0) return ORC_ERR_NNGHS;
    if (!(den != 0.0)) return ORC_ERR_NUMERIC; /* np.average: weights sum to zero */
    *nnghs = (int)nearbyint(num / den);
    return ORC_OK;
}

/* interp_tair.py:837-851: weighted means over neighbours with finite nugget. */
int orc_smooth_vario(const double *nug_m, const double *psill_m, const double *rng_m,
                     const int32_t *idx, const double *wgt, int k, double vario[3])
{
    double sn = 0, sp = 0, sr = 0, den = 0;
    int cnt = 0;
    for (int i = 0; i < k; ++i) {
        int32_t j = idx[i];
        if (isfinite(nug_m[j])) {
            sn += nug_m[j] * wgt[i]; sp += psill_m[j] * wgt[i]; sr += rng_m[j] * wgt[i];
            den += wgt[i]; ++cnt;
        }
    }
    if (cnt == 0) return ORC_ERR_VARIO;
    if (!(den != 0.0)) return ORC_ERR_NUMERIC;
    vario[0] = sn / den; vario[1] = sp / den; vario[2] = sr / den;
    return ORC_OK;
}

/* ---------------------------------------------------------------- a5 ---- */
/* in-place lower Cholesky of a dense n x n (row-major, ld = n) */
static int chol_lower(double *A, int n)
{
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int p = 0; p < j; ++p) s -= A[j * n + p] * A[j * n + p];
        if (!(s > 0.0) || !isfinite(s)) return 1;
        double ljj = sqrt(s);
        A[j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int p = 0; p < j; ++p) t -= A[i * n + p] * A[j * n + p];
            A[i * n + j] = t / ljj;
        }
    }
    return 0;
}

static void fwd_solve(const double *L, int n, double *b)
{
    for (int i = 0; i < n; ++i) {
        double t = b[i];
        for (int p = 0; p < i; ++p) t -= L[i * n + p] * b[p];
        b[i] = t / L[i * n + i];
    }
}

static void bwd_solve_t(const double *L, int n, double *b)
{
    for (int i = n - 1; i >= 0; --i) {
        double t = b[i];
        for (int p = i + 1; p < n; ++p) t -= L[p * n + i] * b[p];
        b[i] = t / L[i * n + i];
    }
}

/* Covariance of the variogram model of interp.R:223-231 (Appendix B.2):
 * Exp(nug, psill, range): c(h>0) = psill exp(-h/range), c(0) = nug + psill;
 * range == 0 -> pure nugget of variance psill + nug. */
static double cov_model(double h, double nug, double psill, double rng)
{
    if (h == 0.0) return nug + psill;
    if (rng == 0.0) return 0.0;
    return psill * exp(-h / rng);
}

/* interp.R:198-270 -> gstat::krige(tair ~ lon+lat+elev+lst, model) with a
 * global neighbourhood of exactly the k stations passed (Appendix B.2):
 *   beta = (X'C^-1 X)^-1 X'C^-1 y
 *   mean = x0'beta + c0'C^-1 (y - X beta)
 *   var  = c(0) - c0'C^-1 c0 + (x0 - X'C^-1 c0)' (X'C^-1 X)^-1 (x0 - X'C^-1 c0)
 * The four trend columns are centred and scaled per neighbourhood; universal
 * kriging is invariant to that re-parameterisation (the basis keeps the
 * intercept) and it keeps the 5x5 normal matrix well conditioned.
 * [parity unpinned: gstat is not runnable here] */
int orc_uk(int k, const double *lon, const double *lat, const double *elev,
           const double *lst, const double *y, double plon, double plat,
           double pelev, double plst, double nug, double psill, double rng,
           double *mean, double *var)
{
    enum { P = 5 };
    if (k < 1) return ORC_ERR_RANGE;
    if (!isfinite(nug) || !isfinite(psill) || !isfinite(rng)) return ORC_ERR_NUMERIC;
    double *C = malloc(sizeof(double) * (size_t)k * k);
    double *A = malloc(sizeof(double) * (size_t)k * (P + 2)); /* cols: X(5), y, c0 */
    const double *cols[4] = { lon, lat, elev, lst };
    double p0[4] = { plon, plat, pelev, plst };
    double x0[P];
    int rc = ORC_OK;

    for (int i = 0; i < k; ++i) {
        C[i * k + i] = nug + psill;
        for (int j = 0; j < i; ++j) {
            double h = orc_ellip_dist(lon[i], lat[i], lon[j], lat[j]);
            C[i * k + j] = C[j * k + i] = cov_model(h, nug, psill, rng);
        }
    }
    x0[0] = 1.0;
    for (int i = 0; i < k; ++i) A[i * (P + 2)] = 1.0;
    for (int c = 0; c < 4; ++c) {
        double mu = 0, sc = 0;
        for (int i = 0; i < k; ++i) mu += cols[c][i];
        mu /= k;
        for (int i = 0; i < k; ++i) { double t = fabs(cols[c][i] - mu); if (t > sc) sc = t; }
        if (!(sc > 0.0)) sc = 1.0; /* constant column: left collinear -> singular below */
        for (int i = 0; i < k; ++i) A[i * (P + 2) + 1 + c] = (cols[c][i] - mu) / sc;
        x0[1 + c] = (p0[c] - mu) / sc;
    }
    for (int i = 0; i < k; ++i) {
        A[i * (P + 2) + P] = y[i];
        double h = orc_ellip_dist(plon, plat, lon[i], lat[i]);
        A[i * (P + 2) + P + 1] = cov_model(h, nug, psill, rng);
    }
    if (chol_lower(C, k)) { rc = ORC_ERR_NUMERIC; goto done; }
    /* A <- L^-1 A (all 7 columns) */
    for (int i = 0; i < k; ++i) {
        for (int c = 0; c < P + 2; ++c) {
            double t = A[i * (P + 2) + c];
            for (int p = 0; p < i; ++p) t -= C[i * k + p] * A[p * (P + 2) + c];
            A[i * (P + 2) + c] = t / C[i * k + i];
        }
    }
    {
        double N[P * P], r[P], q[P], gg = 0, gb = 0;
        memset(N, 0, sizeof N); memset(r, 0, sizeof r); memset(q, 0, sizeof q);
        for (int i = 0; i < k; ++i) {
            const double *a = &A[i * (P + 2)];
            for (int c = 0; c < P; ++c) {
                for (int d = 0; d <= c; ++d) N[c * P + d] += a[c] * a[d];
                r[c] += a[c] * a[P];
                q[c] += a[c] * a[P + 1];
            }
            gg += a[P + 1] * a[P + 1];
            gb += a[P + 1] * a[P];
        }
        for (int c = 0; c < P; ++c) for (int d = c + 1; d < P; ++d) N[c * P + d] = N[d * P + c];
        if (chol_lower(N, P)) { rc = ORC_ERR_NUMERIC; goto done; }
        double beta[P], u[P];
        memcpy(beta, r, sizeof r);
        fwd_solve(N, P, beta); bwd_solve_t(N, P, beta);
        for (int c = 0; c < P; ++c) u[c] = x0[c] - q[c];
        double m = gb, v;
        for (int c = 0; c < P; ++c) m += u[c] * beta[c]; /* x0'b + g'(b - A beta) */
        fwd_solve(N, P, u);
        v = (nug + psill) - gg;
        for (int c = 0; c < P; ++c) v += u[c] * u[c];
        if (!isfinite(m) || !isfinite(v)) { rc = ORC_ERR_NUMERIC; goto done; }
        *mean = m; *var = v;
    }
done:
    free(C); free(A);
    return rc;
}

/* ---------------------------------------------------------------- a7 ---- */
/* interp_tair.py:1099-1146 (_gwr_series): z = x' (X'WX)^-1 X'W with
 * X = [1 | model_x].  The reference inverts the raw 6x6 with np.linalg.inv
 * (:1136); here the five predictor columns are centred / scaled first (z is
 * invariant to it) and the SPD system is solved by Cholesky. */
int orc_gwr_hat(int k, const double *X5, const double *w, const double *x5, double *z)
{
    enum { P = 6 };
    double mu[5], sc[5], x0[P], M[P * P], a[P];
    if (k < 1) return ORC_ERR_RANGE;
    for (int c = 0; c < 5; ++c) {
        double m = 0, s = 0;
        for (int i = 0; i < k; ++i) m += X5[i * 5 + c];
        m /= k;
        for (int i = 0; i < k; ++i) { double t = fabs(X5[i * 5 + c] - m); if (t > s) s = t; }
        if (!(s > 0.0)) s = 1.0;
        mu[c] = m; sc[c] = s;
        x0[1 + c] = (x5[c] - m) / s;
    }
    x0[0] = 1.0;
    memset(M, 0, sizeof M);
    for (int i = 0; i < k; ++i) {
        double r[P];
        r[0] = 1.0;
        for (int c = 0; c < 5; ++c) r[1 + c] = (X5[i * 5 + c] - mu[c]) / sc[c];
        for (int c = 0; c < P; ++c)
            for (int d = 0; d <= c; ++d) M[c * P + d] += w[i] * r[c] * r[d];
    }
    for (int c = 0; c < P; ++c) for (int d = c + 1; d < P; ++d) M[c * P + d] = M[d * P + c];
    if (chol_lower(M, P)) return ORC_ERR_NUMERIC;
    memcpy(a, x0, sizeof a);
    fwd_solve(M, P, a); bwd_solve_t(M, P, a);
    for (int i = 0; i < k; ++i) {
        double t = a[0];
        for (int c = 0; c < 5; ++c) t += a[1 + c] * ((X5[i * 5 + c] - mu[c]) / sc[c]);
        z[i] = w[i] * t;
        if (!isfinite(z[i])) return ORC_ERR_NUMERIC;
    }
    return ORC_OK;
}

/* ------------------------------------------------------ orchestration ---- */
static int db_kmax(const orc_db *db)
{
    double m = 0;
    for (int64_t i = 0; i < 12 * db->n; ++i) {
        if (isfinite(db->optim_nnghs[i]) && db->optim_nnghs[i] > m) m = db->optim_nnghs[i];
        if (isfinite(db->optim_nnghs_anom[i]) && db->optim_nnghs_anom[i] > m) m = db->optim_nnghs_anom[i];
    }
    return (int)nearbyint(m);
}

/* nearest-station list of one (point, variable): computed once and reused by
 * all months, as the reference's point cache does (station_select.py:62-119) */
typedef struct {
    int32_t idx[ORC_MAXK + 1];
    double dist[ORC_MAXK + 1];
    int64_t n;
} near_t;

static int krig_with_near(const orc_db *db, const orc_params *p, const orc_pt *pt,
                          const near_t *nr, int mth, int nnghs, const double *vario,
                          double *mean, double *var, int *nnghs_used, int32_t *ngh_idx)
{
    int32_t idx[ORC_MAXK]; double dist[ORC_MAXK], wgt[ORC_MAXK];
    int m0 = mth - 1, rc;
    int64_t n = db->n;
    if (nnghs <= 0) { /* interp_tair.py:888-890 -> :821-835 */
        rc = orc_select(nr->idx, nr->dist, nr->n, p->init_nnghs, idx, dist, wgt);
        if (rc) return rc;
        rc = orc_smooth_nnghs(db->optim_nnghs + m0 * n, idx, wgt, p->init_nnghs, &nnghs);
        if (rc) return rc;
    }
    if (nnghs < 1 || nnghs > ORC_MAXK) return ORC_ERR_RANGE;
    rc = orc_select(nr->idx, nr->dist, nr->n, nnghs, idx, dist, wgt); /* :892 */
    if (rc) return rc;
    double vp[3];
    if (vario == NULL) { /* :894-895 */
        rc = orc_smooth_vario(db->vario_nug + m0 * n, db->vario_psill + m0 * n,
                              db->vario_rng + m0 * n, idx, wgt, nnghs, vp);
        if (rc) return rc;
    } else {
        vp[0] = vario[0]; vp[1] = vario[1]; vp[2] = vario[2];
    }
    double lo[ORC_MAXK], la[ORC_MAXK], el[ORC_MAXK], ls[ORC_MAXK], y[ORC_MAXK];
    for (int i = 0; i < nnghs; ++i) { /* :899-906 */
        int32_t j = idx[i];
        lo[i] = db->lon[j]; la[i] = db->lat[j]; el[i] = db->elev[j];
        ls[i] = db->lst[m0 * n + j]; y[i] = db->norm[m0 * n + j];
    }
    if (nnghs_used) *nnghs_used = nnghs;
    if (ngh_idx) memcpy(ngh_idx, idx, sizeof(int32_t) * (size_t)nnghs);
    return orc_uk(nnghs, lo, la, el, ls, y, pt->lon, pt->lat, pt->elev, pt->lst[m0],
                  vp[0], vp[1], vp[2], mean, var);
}

static int gwr_with_near(const orc_db *db, const orc_params *p, const orc_pt *pt,
                         double pt_norm, const near_t *nr, int mth, int nnghs,
                         double *out, int scatter, int *nnghs_used, double *z_out,
                         int32_t *ngh_idx)
{
    int32_t idx[ORC_MAXK]; double dist[ORC_MAXK], wgt[ORC_MAXK], z[ORC_MAXK];
    double X5[ORC_MAXK * 5], x5[5];
    int m0 = mth - 1, rc;
    int64_t n = db->n;
    if (nnghs <= 0) { /* interp_tair.py:290-293 -> :245-259 */
        rc = orc_select(nr->idx, nr->dist, nr->n, p->init_nnghs, idx, dist, wgt);
        if (rc) return rc;
        rc = orc_smooth_nnghs(db->optim_nnghs_anom + m0 * n, idx, wgt, p->init_nnghs, &nnghs);
        if (rc) return rc;
    }
    if (nnghs < 1 || nnghs > ORC_MAXK) return ORC_ERR_RANGE;
    rc = orc_select(nr->idx, nr->dist, nr->n, nnghs, idx, dist, wgt); /* :295 */
    if (rc) return rc;
    for (int i = 0; i < nnghs; ++i) { /* :303-304, GWR_TREND_VARS :47 */
        int32_t j = idx[i];
        X5[i * 5 + 0] = db->lon[j]; X5[i * 5 + 1] = db->lat[j]; X5[i * 5 + 2] = db->elev[j];
        X5[i * 5 + 3] = db->tdi[j]; X5[i * 5 + 4] = db->lst[m0 * n + j];
    }
    x5[0] = pt->lon; x5[1] = pt->lat; x5[2] = pt->elev; x5[3] = pt->tdi; x5[4] = pt->lst[m0];
    rc = orc_gwr_hat(nnghs, X5, wgt, x5, z);
    if (rc) return rc;
    if (nnghs_used) *nnghs_used = nnghs;
    if (z_out) memcpy(z_out, z, sizeof(double) * (size_t)nnghs);
    if (ngh_idx) memcpy(ngh_idx, idx, sizeof(int32_t) * (size_t)nnghs);
    if (out == NULL) return ORC_OK;
    /* :300, :1143, :312 -- obs is f4, promoted to f8 by the subtraction */
    int64_t o = 0;
    for (int64_t d = 0; d < db->ndays; ++d) {
        if (db->day_month[d] != mth) continue;
        const float *row = db->obs + d * n;
        double s = 0.0;
        for (int i = 0; i < nnghs; ++i)
            s += z[i] * ((double)row[idx[i]] - db->norm[m0 * n + idx[i]]);
        s += pt_norm;
        if (!isfinite(s)) return ORC_ERR_NUMERIC;
        if (scatter) out[d] = s; else out[o++] = s;
    }
    return ORC_OK;
}

static int64_t pick_ksel(const orc_db *db, const orc_params *p, int nnghs)
{
    int km = nnghs > 0 ? nnghs : db_kmax(db);
    if (km < p->init_nnghs) km = p->init_nnghs;
    if (km > ORC_MAXK) km = ORC_MAXK;
    return km + 1;
}

int orc_krig(const orc_db *db, const orc_params *p, const orc_pt *pt, int mth,
             int nnghs, const double *vario, int32_t excl, int rm_zero_dist,
             double *mean, double *var, int *nnghs_used, int32_t *ngh_idx)
{
    near_t nr;
    nr.n = orc_nearest(db, pt->lat, pt->lon, excl, rm_zero_dist, pick_ksel(db, p, nnghs),
                       nr.idx, nr.dist);
    return krig_with_near(db, p, pt, &nr, mth, nnghs, vario, mean, var, nnghs_used, ngh_idx);
}

int orc_gwr_mth(const orc_db *db, const orc_params *p, const orc_pt *pt,
                double pt_norm, int mth, int nnghs, int32_t excl, int rm_zero_dist,
                double *out, int *nnghs_used, double *z, int32_t *ngh_idx)
{
    near_t nr;
    nr.n = orc_nearest(db, pt->lat, pt->lon, excl, rm_zero_dist, pick_ksel(db, p, nnghs),
                       nr.idx, nr.dist);
    return gwr_with_near(db, p, pt, pt_norm, &nr, mth, nnghs, out, 0, nnghs_used, z, ngh_idx);
}

static int interp_with_ksel(const orc_db *db, const orc_params *p, const orc_pt *pt,
                            int32_t excl, int rm_zero_dist, int64_t ksel,
                            double *daily, double *norms, double *se)
{
    near_t nr;
    nr.n = orc_nearest(db, pt->lat, pt->lon, excl, rm_zero_dist, ksel, nr.idx, nr.dist);
    for (int mth = 1; mth <= 12; ++mth) { /* interp_tair.py:429-437 */
        double mean, var;
        int rc = krig_with_near(db, p, pt, &nr, mth, 0, NULL, &mean, &var, NULL, NULL);
        if (rc) return rc;
        norms[mth - 1] = mean;
        se[mth - 1] = var >= 0 ? sqrt(var) : 0.0; /* std_err_ci :816 */
        if (daily) {
            rc = gwr_with_near(db, p, pt, mean, &nr, mth, 0, daily, 1, NULL, NULL, NULL);
            if (rc) return rc;
        }
    }
    return ORC_OK;
}

int orc_interp(const orc_db *db, const orc_params *p, const orc_pt *pt, int32_t excl,
               int rm_zero_dist, double *daily, double *norms, double *se)
{
    return interp_with_ksel(db, p, pt, excl, rm_zero_dist, pick_ksel(db, p, 0),
                            daily, norms, se);
}

/* ---------------------------------------------------------------- a9 ---- */
/* interp_tair.py:143-197.  Sequential: earlier fixes feed later windows. */
int orc_fixer(double *tmin, double *tmax, int64_t ndays, int tail, int32_t *ninvalid)
{
    int32_t ninv = 0;
    /* the list of invalid days is taken BEFORE any fix (:173) */
    unsigned char *inv = malloc((size_t)ndays);
    for (int64_t d = 0; d < ndays; ++d) { inv[d] = tmin[d] >= tmax[d]; ninv += inv[d]; }
    for (int64_t x = 0; x < ndays; ++x) {
        if (!inv[x]) continue;
        double tavg = (tmin[x] + tmax[x]) / 2.0;
        int64_t s = x - tail, e = x + tail + 1;
        if (s < 0) s = 0;
        if (e > ndays) e = ndays;
        double sum = 0.0; int64_t cnt = 0;
        for (int64_t d = s; d < e; ++d)
            if (tmin[d] < tmax[d]) { sum += tmax[d] - tmin[d]; ++cnt; }
        if (cnt == 0) { free(inv); return ORC_ERR_FIXER; }
        double half = (sum / (double)cnt) / 2.0;
        tmin[x] = tavg - half;
        tmax[x] = tavg + half;
    }
    free(inv);
    *ninvalid = ninv;
    return ORC_OK;
}

/* interp_tair.py:583-590 with the masks of :468-481: per (year, month) means
 * over the normals period, then the mean over years for each month. */
void orc_recompute_norms(const double *daily, int64_t ndays, const int32_t *day_month,
                         const int32_t *day_year, int yr0, int yr1, double *norms)
{
    int ymin = 1 << 30, ymax = -(1 << 30);
    for (int64_t d = 0; d < ndays; ++d)
        if (day_year[d] >= yr0 && day_year[d] <= yr1) {
            if (day_year[d] < ymin) ymin = day_year[d];
            if (day_year[d] > ymax) ymax = day_year[d];
        }
    if (ymin > ymax) { for (int m = 0; m < 12; ++m) norms[m] = NAN; return; }
    int ny = ymax - ymin + 1;
    double *sum = calloc((size_t)ny * 12, sizeof(double));
    int64_t *cnt = calloc((size_t)ny * 12, sizeof(int64_t));
    for (int64_t d = 0; d < ndays; ++d) {
        int y = day_year[d];
        if (y < yr0 || y > yr1) continue;
        int c = (y - ymin) * 12 + day_month[d] - 1;
        sum[c] += daily[d]; cnt[c]++;
    }
    for (int m = 0; m < 12; ++m) {
        double s = 0.0;
        for (int y = 0; y < ny; ++y) s += sum[y * 12 + m] / (double)cnt[y * 12 + m];
        norms[m] = s / ny;
    }
    free(sum); free(cnt);
}

/* --------------------------------------------------------------- a12 ---- */
/* step25:163-164: np.round(x, 2) / np.float32(0.01) assigned into int16:
 * rint(x*100)/100 (half-even), divided by float32(0.01) widened to f8, C cast
 * (truncation toward zero). */
void orc_pack_i16(const double *x, int64_t n, int16_t *out)
{
    const double scale = (double)0.01f;
    for (int64_t i = 0; i < n; ++i) {
        double r = nearbyint(x[i] * 100.0) / 100.0;
        out[i] = (int16_t)(r / scale);
    }
}

/* ------------------------------------------------ second tier: 8f-1 ---- */
/* OLS residuals of y ~ 1 + lon + lat + elev + lst (lm(FORMULA), interp.R:66,64): columns
 * centred / scaled first (same fitted values), normal equations by Cholesky. */
static int ols_residuals(int k, const double *const cols[4], const double *y, double *e)
{
    enum { P = 5 };
    double mu[4], sc[4], N[P * P], r[P], b[P];
    for (int c = 0; c < 4; ++c) {
        double m = 0, s = 0;
        for (int i = 0; i < k; ++i) m += cols[c][i];
        m /= k;
        for (int i = 0; i < k; ++i) { double t = fabs(cols[c][i] - m); if (t > s) s = t; }
        mu[c] = m; sc[c] = s > 0.0 ? s : 1.0;
    }
    memset(N, 0, sizeof N); memset(r, 0, sizeof r);
    for (int i = 0; i < k; ++i) {
        double x[P] = { 1.0, (cols[0][i] - mu[0]) / sc[0], (cols[1][i] - mu[1]) / sc[1],
                        (cols[2][i] - mu[2]) / sc[2], (cols[3][i] - mu[3]) / sc[3] };
        for (int c = 0; c < P; ++c) {
            for (int d = 0; d <= c; ++d) N[c * P + d] += x[c] * x[d];
            r[c] += x[c] * y[i];
        }
    }
    for (int c = 0; c < P; ++c) for (int d = c + 1; d < P; ++d) N[c * P + d] = N[d * P + c];
    if (chol_lower(N, P)) return 1;
    memcpy(b, r, sizeof r);
    fwd_solve(N, P, b); bwd_solve_t(N, P, b);
    for (int i = 0; i < k; ++i) {
        double f = b[0];
        for (int c = 0; c < 4; ++c) f += b[1 + c] * ((cols[c][i] - mu[c]) / sc[c]);
        e[i] = y[i] - f;
    }
    return 0;
}

static double sample_var(const double *x, int n)
{
    double m = 0, s = 0;
    for (int i = 0; i < n; ++i) m += x[i];
    m /= n;
    for (int i = 0; i < n; ++i) s += (x[i] - m) * (x[i] - m);
    return s / (n - 1);
}

/* gstat::variogram(..., cutoff, width) on residuals [upstream-recall]: every pair i<j with
 * 0 < h <= cutoff goes to bin floor(h / width) (a pair exactly on a boundary to the lower bin),
 * gamma = sum (e_i - e_j)^2 / (2 np), dist = mean h; empty bins are dropped. */
int orc_variogram(int k, const double *lon, const double *lat, const double *e, double cutoff,
                  double width, double *dist, double *gamma, double *np)
{
    int nb = (int)ceil(cutoff / width) + 1;
    if (nb > 4096) nb = 4096;
    double *sh = calloc((size_t)nb, sizeof(double)), *sg = calloc((size_t)nb, sizeof(double));
    double *sn = calloc((size_t)nb, sizeof(double));
    for (int i = 1; i < k; ++i)
        for (int j = 0; j < i; ++j) {
            double h = orc_ellip_dist(lon[i], lat[i], lon[j], lat[j]);
            if (!(h <= cutoff)) continue;
            int b = (int)floor(h / width);
            if (b > 0 && h == b * width) --b;
            if (b >= nb) b = nb - 1;
            double d = e[i] - e[j];
            sh[b] += h; sg[b] += d * d; sn[b] += 1.0;
        }
    int n = 0;
    for (int b = 0; b < nb; ++b)
        if (sn[b] > 0) { dist[n] = sh[b] / sn[b]; gamma[n] = sg[b] / (2.0 * sn[b]); np[n] = sn[b]; ++n; }
    free(sh); free(sg); free(sn);
    return n;
}

static double fit_sse(int nbin, const double *dist, const double *gamma, const double *np,
                      double nug, double psill, double range)
{
    double s = 0;
    for (int b = 0; b < nbin; ++b) {
        double m = nug + psill * (1.0 - exp(-dist[b] / range));
        double w = np[b] / (dist[b] * dist[b]);
        s += w * (gamma[b] - m) * (gamma[b] - m);
    }
    return s;
}

/* fit.variogram(..., fit.sills = FALSE, fit.ranges = TRUE, fit.method = 7) restated: Gauss-Newton
 * on the range alone, weights np/h^2; a step is halved (up to 30 times) while it leaves range <= 0
 * or increases the weighted SSE; stop when the SSE improves by less than 1e-10 relative or after 200
 * iterations.  Unusable (-> pure-nugget fallback in the caller): psill <= 0, no bins, non-finite or
 * non-positive range, or a flat objective (no range information). */
int orc_fit_range(int nbin, const double *dist, const double *gamma, const double *np, double nug,
                  double psill, double range0, double *range)
{
    if (nbin < 1 || !(psill > 0.0) || !(range0 > 0.0)) return 1;
    double r = range0, sse = fit_sse(nbin, dist, gamma, np, nug, psill, r);
    for (int it = 0; it < 200; ++it) {
        double num = 0, den = 0;
        for (int b = 0; b < nbin; ++b) {
            double ex = exp(-dist[b] / r);
            double m = nug + psill * (1.0 - ex);
            double J = -psill * ex * dist[b] / (r * r);
            double w = np[b] / (dist[b] * dist[b]);
            num += w * J * (gamma[b] - m); den += w * J * J;
        }
        if (!(den > 0.0) || !isfinite(num)) return 1;
        double step = num / den, rn = r, ssen = sse;
        int ok = 0;
        for (int h = 0; h < 30; ++h) {
            rn = r + step;
            if (rn > 0.0 && isfinite(rn)) {
                ssen = fit_sse(nbin, dist, gamma, np, nug, psill, rn);
                if (ssen <= sse) { ok = 1; break; }
            }
            step *= 0.5;
        }
        if (!ok) break; /* no descent possible: converged at r */
        double impr = sse - ssen;
        r = rn; sse = ssen;
        if (impr <= 1e-10 * sse) break;
    }
    if (!(r > 0.0) || !isfinite(r)) return 1;
    *range = r;
    return 0;
}

/* GLS trend residuals y - X beta_gls for covariance model (nug, psill, range): what
 * predict(g, stns_ngh, BLUE = TRUE) returns is the trend X beta_gls (interp.R:82-84). */
static int gls_residuals(int k, const double *lon, const double *lat, const double *const cols[4],
                         const double *y, double nug, double psill, double rng, double *e)
{
    enum { P = 5 };
    double *C = malloc(sizeof(double) * (size_t)k * k), *A = malloc(sizeof(double) * (size_t)k * (P + 1));
    double mu[4], sc[4];
    int rc = 0;
    for (int i = 0; i < k; ++i) {
        C[i * k + i] = nug + psill;
        for (int j = 0; j < i; ++j)
            C[i * k + j] = C[j * k + i] = cov_model(orc_ellip_dist(lon[i], lat[i], lon[j], lat[j]), nug, psill, rng);
    }
    for (int c = 0; c < 4; ++c) {
        double m = 0, s = 0;
        for (int i = 0; i < k; ++i) m += cols[c][i];
        m /= k;
        for (int i = 0; i < k; ++i) { double t = fabs(cols[c][i] - m); if (t > s) s = t; }
        mu[c] = m; sc[c] = s > 0.0 ? s : 1.0;
    }
    for (int i = 0; i < k; ++i) {
        A[i * (P + 1)] = 1.0;
        for (int c = 0; c < 4; ++c) A[i * (P + 1) + 1 + c] = (cols[c][i] - mu[c]) / sc[c];
        A[i * (P + 1) + P] = y[i];
    }
    if (chol_lower(C, k)) { rc = 1; goto done; }
    {
        double *W = malloc(sizeof(double) * (size_t)k * (P + 1));
        memcpy(W, A, sizeof(double) * (size_t)k * (P + 1));
        for (int i = 0; i < k; ++i)
            for (int c = 0; c <= P; ++c) {
                double t = W[i * (P + 1) + c];
                for (int p = 0; p < i; ++p) t -= C[i * k + p] * W[p * (P + 1) + c];
                W[i * (P + 1) + c] = t / C[i * k + i];
            }
        double N[P * P], r[P];
        memset(N, 0, sizeof N); memset(r, 0, sizeof r);
        for (int i = 0; i < k; ++i)
            for (int c = 0; c < P; ++c) {
                for (int d = 0; d <= c; ++d) N[c * P + d] += W[i * (P + 1) + c] * W[i * (P + 1) + d];
                r[c] += W[i * (P + 1) + c] * W[i * (P + 1) + P];
            }
        for (int c = 0; c < P; ++c) for (int d = c + 1; d < P; ++d) N[c * P + d] = N[d * P + c];
        free(W);
        if (chol_lower(N, P)) { rc = 1; goto done; }
        fwd_solve(N, P, r); bwd_solve_t(N, P, r);
        for (int i = 0; i < k; ++i) {
            double f = 0;
            for (int c = 0; c < P; ++c) f += r[c] * A[i * (P + 1) + c];
            e[i] = y[i] - f;
        }
    }
done:
    free(C); free(A);
    return rc;
}

/* one variogram + constrained fit (interp.R:63-80 and :86-96): nugget fixed at min(gamma), total
 * sill fixed at `sill`, range fitted from 0.1 * max bin distance; fallback: pure nugget of `sill`. */
static void fit_or_nugget(int k, const double *lon, const double *lat, const double *e, double cutoff,
                          double sill, double m[3])
{
    int cap = (int)ceil(cutoff / 5.0) + 3;
    double *dist = malloc(sizeof(double) * 3 * (size_t)cap), *gam = dist + cap, *np = gam + cap;
    int nb = orc_variogram(k, lon, lat, e, cutoff, 5.0, dist, gam, np);
    m[0] = sill; m[1] = 0.0; m[2] = 0.0;
    if (nb > 0) {
        double gmin = gam[0], dmax = dist[0], rng;
        for (int b = 1; b < nb; ++b) { if (gam[b] < gmin) gmin = gam[b]; if (dist[b] > dmax) dmax = dist[b]; }
        if (orc_fit_range(nb, dist, gam, np, gmin, sill - gmin, 0.1 * dmax, &rng) == 0) {
            m[0] = gmin; m[1] = sill - gmin; m[2] = rng;
        }
    }
    free(dist);
}

/* interp.R:54-113 */
int orc_get_vario_params(int k, const double *lon, const double *lat, const double *elev,
                         const double *lst, const double *y, double max_ngh_dist, double vario[3])
{
    const double *cols[4] = { lon, lat, elev, lst };
    if (k < 7 || k > ORC_MAXK) return ORC_ERR_RANGE;
    double *e = malloc(sizeof(double) * (size_t)k);
    double m1[3];
    int rc = ORC_OK;
    const double cutoff = max_ngh_dist * 1.4;                         /* :63 */
    if (ols_residuals(k, cols, y, e)) { rc = ORC_ERR_NUMERIC; goto done; }
    fit_or_nugget(k, lon, lat, e, cutoff, sample_var(e, k), m1);      /* :64-80 */
    if (gls_residuals(k, lon, lat, cols, y, m1[0], m1[1], m1[2], e)) { rc = ORC_ERR_NUMERIC; goto done; } /* :82-84 */
    fit_or_nugget(k, lon, lat, e, cutoff, sample_var(e, k), vario);   /* :85-96, returned as (nug, psill, range) :102-112 */
    if (!isfinite(vario[0]) || !isfinite(vario[1]) || !isfinite(vario[2])) rc = ORC_ERR_NUMERIC;
done:
    free(e);
    return rc;
}

static int vario_with_near(const orc_db *db, const near_t *nr, int mth, int nnghs, double vario[3])
{
    int32_t idx[ORC_MAXK]; double dist[ORC_MAXK], wgt[ORC_MAXK];
    double lo[ORC_MAXK], la[ORC_MAXK], el[ORC_MAXK], ls[ORC_MAXK], y[ORC_MAXK];
    const int m0 = mth - 1;
    const int64_t n = db->n;
    int rc = orc_select(nr->idx, nr->dist, nr->n, nnghs, idx, dist, wgt);
    if (rc) return rc;
    double dmax = 0;
    for (int i = 0; i < nnghs; ++i) {
        int32_t j = idx[i];
        lo[i] = db->lon[j]; la[i] = db->lat[j]; el[i] = db->elev[j];
        ls[i] = db->lst[m0 * n + j]; y[i] = db->norm[m0 * n + j];
        if (dist[i] > dmax) dmax = dist[i];
    }
    return orc_get_vario_params(nnghs, lo, la, el, ls, y, dmax, vario);
}

/* interp_tair.py:635-698 */
int orc_build_krig_params(const orc_db *db, const orc_params *p, const orc_pt *pt, int mth,
                          double vario[3], int *nnghs_used)
{
    near_t nr;
    int32_t idx[ORC_MAXK]; double dist[ORC_MAXK], wgt[ORC_MAXK];
    int nnghs = 0;
    nr.n = orc_nearest(db, pt->lat, pt->lon, -1, 0, pick_ksel(db, p, 0), nr.idx, nr.dist);
    int rc = orc_select(nr.idx, nr.dist, nr.n, p->init_nnghs, idx, dist, wgt);         /* :667 */
    if (rc) return rc;
    rc = orc_smooth_nnghs(db->optim_nnghs + (mth - 1) * db->n, idx, wgt, p->init_nnghs, &nnghs); /* :669-678 */
    if (rc) return rc;
    if (nnghs < 1 || nnghs > ORC_MAXK) return ORC_ERR_RANGE;
    if (nnghs_used) *nnghs_used = nnghs;
    return vario_with_near(db, &nr, mth, nnghs, vario);                                 /* :681-698 */
}

/* interp_tair.py:722-769 + interp.R:148-159 */
int orc_krigall(const orc_db *db, const orc_params *p, const orc_pt *pt, int nnghs, int32_t excl,
                int rm_zero_dist, double norms[12], double vario_out[36])
{
    near_t nr;
    if (nnghs < 1 || nnghs > ORC_MAXK) return ORC_ERR_RANGE;
    nr.n = orc_nearest(db, pt->lat, pt->lon, excl, rm_zero_dist, pick_ksel(db, p, nnghs), nr.idx, nr.dist);
    for (int mth = 1; mth <= 12; ++mth) {
        double v[3], mean, var;
        int rc = vario_with_near(db, &nr, mth, nnghs, v);
        if (rc) return rc;
        if (vario_out) memcpy(vario_out + (mth - 1) * 3, v, sizeof v);
        rc = krig_with_near(db, p, pt, &nr, mth, nnghs, v, &mean, &var, NULL, NULL);
        if (rc) return rc;
        norms[mth - 1] = mean;
    }
    return ORC_OK;
}

/* ----------------------------------------------------- step25:126-172 ---- */
int orc_interp_grid(const orc_db *tmin, const orc_db *tmax, const orc_params *p,
                    int Y, int X, const uint8_t *mask, const double *lat,
                    const double *lon, const float *elev, const float *tdi,
                    const float *lst_night, const float *lst_day,
                    float *norm_tmin, float *se_tmin, float *norm_tmax, float *se_tmax,
                    int16_t *daily_tmin, int16_t *daily_tmax, int32_t *ninvalid,
                    int32_t *status, int nthreads)
{
    const int64_t ncell = (int64_t)Y * X;
    const int with_daily = daily_tmin != NULL || daily_tmax != NULL;
    const int64_t ndays = with_daily ? (tmin ? tmin->ndays : tmax->ndays) : 0;
    const int64_t ksel_n = tmin ? pick_ksel(tmin, p, 0) : 0;
    const int64_t ksel_x = tmax ? pick_ksel(tmax, p, 0) : 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        double *dn = with_daily ? malloc(sizeof(double) * (size_t)ndays) : NULL;
        double *dx = with_daily ? malloc(sizeof(double) * (size_t)ndays) : NULL;
        int16_t *pk = with_daily ? malloc(sizeof(int16_t) * (size_t)ndays) : NULL;
#pragma omp for schedule(dynamic, 8)
        for (int64_t c = 0; c < ncell; ++c) {
            if (!mask[c]) continue;
            int r = (int)(c / X), q = (int)(c % X);
            orc_pt pt;
            double nn[12], sn[12], nx[12], sx[12];
            int rc = ORC_OK;
            int32_t ninv = 0;
            pt.lat = lat[r]; pt.lon = lon[q];
            pt.elev = (double)elev[c]; pt.tdi = (double)tdi[c];
            if (tmin) { /* interp_tair.py:560-566 */
                for (int m = 0; m < 12; ++m) pt.lst[m] = (double)lst_night[m * ncell + c];
                rc = interp_with_ksel(tmin, p, &pt, -1, 0, ksel_n, daily_tmin ? dn : NULL, nn, sn);
            }
            if (!rc && tmax) { /* :569-575 */
                for (int m = 0; m < 12; ++m) pt.lst[m] = (double)lst_day[m * ncell + c];
                rc = interp_with_ksel(tmax, p, &pt, -1, 0, ksel_x, daily_tmax ? dx : NULL, nx, sx);
            }
            if (!rc && daily_tmin && daily_tmax) { /* :579-590 */
                rc = orc_fixer(dn, dx, ndays, p->fixer_tail, &ninv);
                if (!rc && ninv > 0) {
                    orc_recompute_norms(dn, ndays, tmin->day_month, tmin->day_year,
                                        p->norm_yr0, p->norm_yr1, nn);
                    orc_recompute_norms(dx, ndays, tmin->day_month, tmin->day_year,
                                        p->norm_yr0, p->norm_yr1, nx);
                }
            }
            if (status) status[c] = rc;
            if (rc) continue; /* step25:154-160: leave fill values */
            for (int m = 0; m < 12; ++m) { /* step25:166-172 */
                if (tmin) { norm_tmin[m * ncell + c] = (float)nn[m]; se_tmin[m * ncell + c] = (float)sn[m]; }
                if (tmax) { norm_tmax[m * ncell + c] = (float)nx[m]; se_tmax[m * ncell + c] = (float)sx[m]; }
            }
            if (ninvalid) ninvalid[c] = ninv;
            if (daily_tmin) {
                orc_pack_i16(dn, ndays, pk);
                for (int64_t d = 0; d < ndays; ++d) daily_tmin[d * ncell + c] = pk[d];
            }
            if (daily_tmax) {
                orc_pack_i16(dx, ndays, pk);
                for (int64_t d = 0; d < ndays; ++d) daily_tmax[d * ncell + c] = pk[d];
            }
        }
        free(dn); free(dx); free(pk);
    }
    return ORC_OK;
}

/* ------------------------------------------------ second tier: 8f-3 ---- */
/* _TairAggregate.__init__ (tiling.py:1085-1118): groups = u_yrs x u_mths in year-major order
 * (a (year, month) with no day is an EMPTY group -> masked mean); group id per day. */
int orc_agg_groups(int64_t ndays, const int32_t *day_year, const int32_t *day_month, int32_t *nyr,
                   int32_t *nmth, int32_t *day_group)
{
    if (ndays <= 0) return ORC_ERR_RANGE;
    int32_t y0 = day_year[0], y1 = day_year[0];
    int has[13] = {0};
    for (int64_t d = 0; d < ndays; ++d) {
        if (day_month[d] < 1 || day_month[d] > 12) return ORC_ERR_RANGE;
        if (day_year[d] < y0) y0 = day_year[d];
        if (day_year[d] > y1) y1 = day_year[d];
        has[day_month[d]] = 1;
    }
    /* np.unique(days[YEAR]): only years that occur */
    int32_t ny = 0;
    int32_t *ymap = (int32_t *)malloc(sizeof(int32_t) * (size_t)(y1 - y0 + 1));
    char *yhas = (char *)calloc((size_t)(y1 - y0 + 1), 1);
    for (int64_t d = 0; d < ndays; ++d) yhas[day_year[d] - y0] = 1;
    for (int32_t y = y0; y <= y1; ++y) ymap[y - y0] = yhas[y - y0] ? ny++ : -1;
    int mmap[13], nm = 0;
    for (int m = 1; m <= 12; ++m) mmap[m] = has[m] ? nm++ : -1;
    for (int64_t d = 0; d < ndays; ++d) day_group[d] = ymap[day_year[d] - y0] * nm + mmap[day_month[d]];
    *nyr = ny; *nmth = nm;
    free(ymap); free(yhas);
    return ORC_OK;
}

/* one daily value as netCDF4 / numpy hand it to daily_to_mthly: raw int16 is unpacked as
 * int16 * np.float32(0.01) (float32 product; tiling.py:36,448) with _FillValue masked */
static inline int agg_value(const void *daily, int dtype, int64_t i, double *v)
{
    if (dtype == 0) {
        int16_t r = ((const int16_t *)daily)[i];
        if (r == (int16_t)-32767) return 0;
        float f = (float)r * 0.01f;
        *v = (double)f;
        return 1;
    } else if (dtype == 1) {
        float f = ((const float *)daily)[i];
        if (isnan(f)) return 0;
        *v = (double)f;
        return 1;
    }
    double x = ((const double *)daily)[i];
    if (isnan(x)) return 0;
    *v = x;
    return 1;
}

/* daily_to_mthly (tiling.py:1120-1134): np.ma.mean(np.ma.take(tair, mask, axis=0), axis=0,
 * dtype=float) = (sum of the unmasked values, in day order, in f8) * 1. / count; NaN = masked */
void orc_daily_to_mthly(const void *daily, int dtype, int64_t ndays, int64_t ncell,
                        const int32_t *day_group, int ng, double *mthly)
{
    double *sum = (double *)calloc((size_t)ng, sizeof(double));
    int64_t *cnt = (int64_t *)calloc((size_t)ng, sizeof(int64_t));
    for (int64_t c = 0; c < ncell; ++c) {
        for (int g = 0; g < ng; ++g) { sum[g] = 0.0; cnt[g] = 0; }
        for (int64_t d = 0; d < ndays; ++d) {
            double v;
            if (agg_value(daily, dtype, d * ncell + c, &v)) { sum[day_group[d]] += v; cnt[day_group[d]]++; }
        }
        for (int g = 0; g < ng; ++g) mthly[(int64_t)g * ncell + c] = cnt[g] ? sum[g] * 1.0 / (double)cnt[g] : NAN;
    }
    free(sum); free(cnt);
}

/* mthly_to_ann (tiling.py:1151-1166): per year the mean of its nmth monthly means */
void orc_mthly_to_ann(const double *mthly, int nyr, int nmth, int64_t ncell, double *ann)
{
    for (int y = 0; y < nyr; ++y)
        for (int64_t c = 0; c < ncell; ++c) {
            double s = 0.0; int n = 0;
            for (int m = 0; m < nmth; ++m) {
                double v = mthly[((int64_t)y * nmth + m) * ncell + c];
                if (!isnan(v)) { s += v; n++; }
            }
            ann[(int64_t)y * ncell + c] = n ? s * 1.0 / (double)n : NAN;
        }
}

/* write_ds_mthly (tiling.py:1215-1216): np.ma.round(x, 2) assigned to an 'i2' netCDF variable with
 * scale_factor float32(0.01): netCDF4-python packs as np.around(x / scale_factor) (restated from its
 * published behaviour; netCDF4 is not installed here); masked -> _FillValue */
void orc_pack_mthly_i16(const double *x, int64_t n, int16_t *out)
{
    const double scale = (double)0.01f;
    for (int64_t i = 0; i < n; ++i) {
        if (isnan(x[i])) { out[i] = (int16_t)-32767; continue; }
        double r = nearbyint(x[i] * 100.0) / 100.0;
        out[i] = (int16_t)nearbyint(r / scale);
    }
}

/* ------------------------------------------------ second tier: 8f-4 ---- */
/* GeoNc.get_row_col (util_ncdf.py:262-301): GDAL-style geotransform from the first two cell
 * centres, int() truncation, abs().  Returns 1 when the cell is outside the raster (the reference
 * then raises IndexError at var[row, col], interp_tair.py:122). */
int orc_get_row_col(int nrows, int ncols, const double *lons, const double *lats, double lon, double lat,
                    int32_t *row, int32_t *col)
{
    const double ph = -fabs(lats[0] - lats[1]), pw = fabs(lons[0] - lons[1]);
    const double ox = lons[0] - pw / 2.0, oy = lats[0] + fabs(ph / 2.0);
    const double fc = (lon - ox) / pw, fr = (lat - oy) / ph;
    if (!(fabs(fc) < 2147483648.0) || !(fabs(fr) < 2147483648.0)) return 1;
    int64_t c = (int64_t)fc, r = (int64_t)fr;
    if (c < 0) c = -c;
    if (r < 0) r = -r;
    *row = (int32_t)r; *col = (int32_t)c;
    return (r >= nrows || c >= ncols) ? 1 : 0;
}

/* PredictorGrids.setPtValues (interp_tair.py:115-141).  order 0: the cell of get_row_col.  order 1:
 * mpl_toolkits.basemap.interp(order=1, masked=True) on the south-up copy of the raster, falling back to
 * its order=0 (nearest) and then to the missing value.  basemap is a third-party dependency that is not
 * vendored: its published algorithm is restated (PARITY UNPINNED for order 1).  data: [nrows][ncols]
 * north-up f4, NaN = missing; lats descending.  Returns 0 ok, 1 outside (order 0 only). */
static double bm_coord(double g0, double g1, int n, double v)
{
    /* regular axis: (len-1) * (out - in[0]) / (in[-1] - in[0]) */
    return (double)(n - 1) * (v - g0) / (g1 - g0);
}

int orc_sample_point(int nrows, int ncols, const double *lons, const double *lats, const float *data,
                     double lon, double lat, int order, double missing, double *val, int32_t *row, int32_t *col)
{
    if (order == 0) {
        int rc = orc_get_row_col(nrows, ncols, lons, lats, lon, lat, row, col);
        if (rc) return rc;
        *val = (double)data[(int64_t)(*row) * ncols + *col];
        return 0;
    }
    /* yGrid = np.sort(lat) (ascending), ncData = flipud(a): row i of the flipped copy = nrows-1-i */
    const double ylo = lats[nrows - 1], yhi = lats[0];
    const int outside = (lon < lons[0]) || (lon > lons[ncols - 1]) || (lat < ylo) || (lat > yhi);
    double xc = bm_coord(lons[0], lons[ncols - 1], ncols, lon);
    double yc = bm_coord(ylo, yhi, nrows, lat);
    if (xc < 0) xc = 0;
    if (xc > ncols - 1) xc = ncols - 1;
    if (yc < 0) yc = 0;
    if (yc > nrows - 1) yc = nrows - 1;
    *row = -1; *col = -1;
    if (!outside) {
        int xi = (int)xc, yi = (int)yc;
        int xip = xi + 1 > ncols - 1 ? ncols - 1 : xi + 1, yip = yi + 1 > nrows - 1 ? nrows - 1 : yi + 1;
        const double dx = xc - (double)(float)xi, dy = yc - (double)(float)yi;
#define FL(yy, xx) ((double)data[(int64_t)(nrows - 1 - (yy)) * ncols + (xx)])
        const double a = FL(yi, xi), b = FL(yip, xip), c = FL(yip, xi), d = FL(yi, xip);
        if (!isnan(a) && !isnan(b) && !isnan(c) && !isnan(d)) {
            *val = (1. - dx) * (1. - dy) * a + dx * dy * b + (1. - dx) * dy * c + dx * (1. - dy) * d;
            return 0;
        }
        /* order 0: np.around (half to even) */
        const int xn = (int)nearbyint(xc), yn = (int)nearbyint(yc);
        const double v = FL(yn, xn);
#undef FL
        if (!isnan(v)) { *val = v; return 0; }
    }
    *val = missing;
    return 0;
}
