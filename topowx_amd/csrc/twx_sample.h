// twx_sample.h -- point-mode predictor sampling.
//
// PredictorGrids.setPtValues (twx/interp/interp_tair.py:115-141): order 0 reads the raster cell of
// GeoNc.get_row_col (twx/utils/util_ncdf.py:262-301); order 1 is mpl_toolkits.basemap.interp(order=1,
// masked=True) on the south-up copy, falling back to its order 0 and then to the missing value (basemap
// is not vendored: published algorithm restated, parity unpinned).  One thread per point: a gather.
#pragma once
#include "twx_device.h"

struct RasterDev {
    int nrows, ncols;
    const double *lon;   // [ncols] cell centres, ascending
    const double *lat;   // [nrows] cell centres, descending (north-up)
    const float *data;   // [nrows][ncols], NaN = missing
};

__global__ void k_sample(RasterDev r, int64_t npts, const double *__restrict__ qlon, const double *__restrict__ qlat,
                         int order, double missing, double *__restrict__ val, int32_t *__restrict__ row,
                         int32_t *__restrict__ col, int32_t *__restrict__ status)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const double lon = qlon[i], lat = qlat[i];
    int ro = -1, co = -1, st = 0;
    double v = missing;
    if (order == 0) {
        const double ph = -fabs(r.lat[0] - r.lat[1]), pw = fabs(r.lon[0] - r.lon[1]);
        const double ox = r.lon[0] - pw / 2.0, oy = r.lat[0] + fabs(ph / 2.0);
        const double fc = (lon - ox) / pw, fr = (lat - oy) / ph;
        if (!(fabs(fc) < 2147483648.0) || !(fabs(fr) < 2147483648.0)) st = 1;
        else {
            int64_t c = (int64_t)fc, rr = (int64_t)fr;           // int(): truncation; abs()
            if (c < 0) c = -c;
            if (rr < 0) rr = -rr;
            ro = (int)rr; co = (int)c;
            if (rr >= r.nrows || c >= r.ncols) st = 1;           // IndexError in the reference
            else v = (double)r.data[rr * r.ncols + c];
        }
    } else {
        const double ylo = r.lat[r.nrows - 1], yhi = r.lat[0], x0 = r.lon[0], x1 = r.lon[r.ncols - 1];
        const bool outside = (lon < x0) || (lon > x1) || (lat < ylo) || (lat > yhi);
        double xc = (double)(r.ncols - 1) * (lon - x0) / (x1 - x0);
        double yc = (double)(r.nrows - 1) * (lat - ylo) / (yhi - ylo);
        xc = fmin(fmax(xc, 0.0), (double)(r.ncols - 1));
        yc = fmin(fmax(yc, 0.0), (double)(r.nrows - 1));
        if (!outside) {
            const int xi = (int)xc, yi = (int)yc;
            const int xip = min(xi + 1, r.ncols - 1), yip = min(yi + 1, r.nrows - 1);
            const double dx = xc - (double)(float)xi, dy = yc - (double)(float)yi;
            auto fl = [&](int yy, int xx) { return (double)r.data[(int64_t)(r.nrows - 1 - yy) * r.ncols + xx]; };
            const double a = fl(yi, xi), b = fl(yip, xip), c = fl(yip, xi), d = fl(yi, xip);
            if (a == a && b == b && c == c && d == d)
                v = (1. - dx) * (1. - dy) * a + dx * dy * b + (1. - dx) * dy * c + dx * (1. - dy) * d;
            else {
                const double n = fl((int)rint(yc), (int)rint(xc));   // order 0: np.around
                if (n == n) v = n;
            }
        }
    }
    val[i] = v;
    if (row) row[i] = ro;
    if (col) col[i] = co;
    if (status) status[i] = st;
}
