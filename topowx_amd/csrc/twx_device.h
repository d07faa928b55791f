// twx_device.h -- device-side data structures and math helpers shared by the
// kernels of libtwxhip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "twx.h"

#define TWX_KSEL_MAX 160            // nearest-list slots per cell (>= TWX_MAX_NNGHS + 1)
#define TWX_WAVE 64

// ---- device copy of one variable's station table (SoA, fp64) -----------------
struct StnDev {
    int n;
    int kmax;          // largest finite optim_nnghs / optim_nnghs_anom in the table
    const double *lon, *lat, *elev, *tdi;                       // [n]
    const double *lst, *norm, *optim, *optim_anom;              // [12][n]
    const double *nug, *psill, *rng;                            // [12][n]
    const double *sph, *cph, *slh, *clh;                        // sin/cos(lat/2), sin/cos(lon/2) [n]
    // station-major copies of the columns the monthly smoothing of k_select gathers for ALL twelve months of a
    // neighbour (month-major columns cost one scattered 8-byte load -- one cache line -- per (neighbour, month, field):
    // the selection was bound by the texture addresser, not by arithmetic)
    const double *optim_s, *optim_anom_s;                       // [n][12]
    const double *vario_s;                                      // [n][12][4] = nug, psill, rng, 0
    const double4 *stat_s;                                      // [n] station record (lon, lat, elev, tdi)
    const double2 *mon_s;                                       // [n][12] (lst, norm)
    const double *coslat;                                       // cos(lat * TWX_DEG2RAD) [n] (k_stn_coslat; k_tile_cand's fp32 bound)
    const float *nn_km;                                         // [n] distance to the nearest OTHER station of the table (k_stn_nn): a
                                                                // lower bound of the pair distances inside any neighbourhood (SelWs.hminp)
    const float *obs;                                           // [n][ndays_mm] month-major days, or null
    const double *ymsum;                                        // [n][12][norm_ny] sum of a station's observations over every (month, year) of
                                                                // the normals period (k_fix_sparse), or null
};

// ---- where the cells of a launch come from -------------------------------------
// mode 0: grid planes (twx_grid, native dtypes); cell id = row * X + col
// mode 1: point list (twx_pt AoS); cell id = point index, one tile per point
struct CellSrc {
    int mode;
    int Y, X;
    int ts;              // tile edge (cells), grid mode
    int ntx;             // tiles per row
    const uint8_t *mask;
    const double *lat, *lon;
    const float *elev, *tdi;
    const float *lst;    // [12][Y][X] plane of the variable being interpolated
    const twx_pt *pts;
    const int32_t *excl; // per point station index to drop or null
    const int32_t *excl_more;  // [npts][nexcl] further station indices to drop per point (< 0: unused), or null (twx_set_exclusions)
    int nexcl;
    const int32_t *mth;  // per point month 1..12 (0 / null = all twelve)
    const int32_t *nnghs_in;   // per point explicit bandwidth (<= 0 smooth) or null
    const double *vario_in;    // per point [3] explicit variogram (NaN nugget = smooth) or null
    // point mode: runs of consecutive points at one location with one excluded station (a cross-validated station x its
    // bandwidths x months) share ONE candidate list: ptile = list ("tile") of a point, ptfirst = first point of a list
    const int32_t *ptile, *ptfirst;
    int rm_zero;
    int do_krig;         // derive kriging bandwidth + variogram (a3, a4)
    int do_vario;        // smooth the variogram from the neighbours' (0: it is fitted afterwards, 8f-1)
    int do_anom;         // derive the GWR bandwidth (needs optim_nnghs_anom)
};

struct CellVals {
    double lon, lat, elev, tdi;
};

__device__ __forceinline__ bool cell_valid(const CellSrc &s, int64_t c)
{
    return s.mode == 1 ? true : (s.mask[c] != 0);
}

__device__ __forceinline__ CellVals cell_load(const CellSrc &s, int64_t c)
{
    CellVals v;
    if (s.mode == 1) {
        v.lon = s.pts[c].lon; v.lat = s.pts[c].lat; v.elev = s.pts[c].elev; v.tdi = s.pts[c].tdi;
    } else {
        int r = (int)(c / s.X), q = (int)(c % s.X);
        v.lon = s.lon[q]; v.lat = s.lat[r];
        v.elev = (double)s.elev[c]; v.tdi = (double)s.tdi[c];
    }
    return v;
}

__device__ __forceinline__ double cell_lst(const CellSrc &s, int64_t c, int m0)
{
    return s.mode == 1 ? s.pts[c].lst[m0] : (double)s.lst[(int64_t)m0 * s.Y * s.X + c];
}

__device__ __forceinline__ int64_t cell_tile(const CellSrc &s, int64_t c)
{
    if (s.mode == 1) return s.ptile ? s.ptile[c] : c;
    int r = (int)(c / s.X), q = (int)(c % s.X);
    return (int64_t)(r / s.ts) * s.ntx + (q / s.ts);
}

// ---- math ------------------------------------------------------------------------
#define TWX_DEG2RAD 0.017453292519943295   // util_geo.py:21
#define TWX_EARTH_KM 6371.009              // util_geo.py:22

// Haversine of the reference (util_geo.py:24-40), same operation order as the
// oracle so that neighbour ORDER (integer output) agrees.
__device__ __forceinline__ double hav_km(double lon1, double lat1, double lon2, double lat2)
{
    double lat1r = lat1 * TWX_DEG2RAD, lat2r = lat2 * TWX_DEG2RAD;
    double lon1r = lon1 * TWX_DEG2RAD, lon2r = lon2 * TWX_DEG2RAD;
    double dlat = lat1r - lat2r, dlon = lon1r - lon2r;
    double s1 = sin(dlat / 2), s2 = sin(dlon / 2);
    double h = __dadd_rn(__dmul_rn(s1, s1), __dmul_rn(__dmul_rn(cos(lat1r), cos(lat2r)), __dmul_rn(s2, s2)));
    return TWX_EARTH_KM * (2 * asin(sqrt(h)));
}

__device__ __forceinline__ double cos_lat(double lat) { return cos(lat * TWX_DEG2RAD); }

// fp32 haversine for CONSERVATIVE bounds only (k_tile_cand's candidate radius carries a 50 m margin; this is good to
// ~1 m at the few hundred km it is used for: the coordinate differences are formed in fp64)
__device__ __forceinline__ float hav_km_f32(double lon1, double lat1, float cos1, double lon2, double lat2, float cos2)
{
    const float dlat = (float)((lat1 - lat2) * (0.5 * TWX_DEG2RAD)), dlon = (float)((lon1 - lon2) * (0.5 * TWX_DEG2RAD));
    const float s1 = sinf(dlat), s2 = sinf(dlon);
    const float h = fmaf(cos1 * cos2, s2 * s2, s1 * s1);
    return (float)(2.0 * TWX_EARTH_KM) * asinf(fminf(sqrtf(h), 1.f));
}

#define TWX_WGS84_A 6378.137
#define TWX_WGS84_F (1.0 / 298.257223563)

// sp/gstat great-circle distance on the WGS84 ellipsoid from
// sin^2 / cos^2 of F, G, L.
__device__ __forceinline__ double ellip_core(double sG2, double cG2, double sF2, double cF2,
                                             double sL2, double cL2)
{
    double S = sG2 * cL2 + cF2 * sL2;
    double C = cG2 * cL2 + sF2 * sL2;
    double w = atan(sqrt(S / C));
    double R = sqrt(S * C) / w;
    double D = 2 * w * TWX_WGS84_A;
    double H1 = (3 * R - 1) / (2 * C);
    double H2 = (3 * R + 1) / (2 * S);
    return D * (1 + TWX_WGS84_F * H1 * sF2 * cG2 - TWX_WGS84_F * H2 * cF2 * sG2);
}

// from raw coordinates (cell -> station), with the identical-point test of sp
__device__ __forceinline__ double ellip_km(double lon1, double lat1, double lon2, double lat2)
{
    const double eps = 2.220446049250313e-16;
    if (fabs(lat1 - lat2) < eps) {
        if (fabs(lon1 - lon2) < eps) return 0.0;
        if (fabs((fabs(lon1) + fabs(lon2)) - 360.0) < eps) return 0.0;
    }
    const double r = 3.14159265358979323846 / 180.0;
    double F = (lat1 * r + lat2 * r) / 2.0, G = (lat1 * r - lat2 * r) / 2.0, L = (lon1 * r - lon2 * r) / 2.0;
    double sG = sin(G), cG = cos(G), sF = sin(F), cF = cos(F), sL = sin(L), cL = cos(L);
    return ellip_core(sG * sG, cG * cG, sF * sF, cF * cF, sL * sL, cL * cL);
}

// station pair: half-angle sines / cosines are precomputed per station, so F, G, L
// come from angle-addition products instead of six trig calls per pair
__device__ __forceinline__ double ellip_pair(double sp1, double cp1, double sl1, double cl1,
                                             double sp2, double cp2, double sl2, double cl2)
{
    if (sp1 == sp2 && cp1 == cp2 && sl1 == sl2 && cl1 == cl2) return 0.0; // same location
    double sG = sp1 * cp2 - cp1 * sp2, cG = cp1 * cp2 + sp1 * sp2;
    double sF = sp1 * cp2 + cp1 * sp2, cF = cp1 * cp2 - sp1 * sp2;
    double sL = sl1 * cl2 - cl1 * sl2, cL = cl1 * cl2 + sl1 * sl2;
    return ellip_core(sG * sG, cG * cG, sF * sF, cF * cF, sL * sL, cL * cL);
}

// Work-group -> position in a list of n units of work.  Work-groups are dealt round-robin to the 8 XCDs (one L2
// each); with this map every XCD takes a contiguous eighth of the list, in order, so that neighbouring units (the
// months of a cell, the cells of a tile row) meet in one L2.  The launch needs a multiple of 8 work-groups >= n;
// -1 = nothing to do.
__device__ __forceinline__ int xcd_contig(unsigned wg, int n)
{
#ifdef TWX_NO_XCD_ORDER
    return (int)wg < n ? (int)wg : -1;
#else
    const int per = (n + 7) >> 3;
    const int it = (int)(wg & 7u) * per + (int)(wg >> 3);
    return ((int)(wg >> 3) < per && it < n) ? it : -1;
#endif
}

// Wave-wide sum / maximum through DPP row shifts and row broadcasts (no LDS crossbar traffic: a __shfl_xor butterfly
// costs two ds_bpermute per step and double).  All lanes return the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_step(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);       // lanes without a source (or in a masked row) add +0
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v)
{
    const int l = __double2loint(v), h = __double2hiint(v);
    const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, ROW_MASK, 0xf, false);   // lanes without a source keep their own value
    const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, ROW_MASK, 0xf, false);
    return fmax(v, __hiloint2double(hi, lo));
}

__device__ __forceinline__ double readlane63(double v)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
    v = dpp_add_step<0x111, 0xf>(v);           // row_shr:1
    v = dpp_add_step<0x112, 0xf>(v);           // row_shr:2
    v = dpp_add_step<0x114, 0xf>(v);           // row_shr:4
    v = dpp_add_step<0x118, 0xf>(v);           // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add_step<0x142, 0xa>(v);           // row_bcast:15 into rows 1 and 3
    v = dpp_add_step<0x143, 0xc>(v);           // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return readlane63(v);
}

__device__ __forceinline__ double wave_max(double v)
{
    v = dpp_max_step<0x111, 0xf>(v);
    v = dpp_max_step<0x112, 0xf>(v);
    v = dpp_max_step<0x114, 0xf>(v);
    v = dpp_max_step<0x118, 0xf>(v);
    v = dpp_max_step<0x142, 0xa>(v);
    v = dpp_max_step<0x143, 0xc>(v);
    return readlane63(v);
}

__device__ __forceinline__ double wave_sum_dpp(double v) { return wave_sum(v); }
__device__ __forceinline__ double wave_max_dpp(double v) { return wave_max(v); }

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ bool finite_d(double v) { return fabs(v) <= 1.79769313486231570e308; }

// compile-time loop: the body gets the index as an integral_constant, so every
// register-array subscript is a constant by construction (a "#pragma unroll" on the
// outer block-column loop is refused by the optimizer for the larger NB)
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void sfor2(F &&f)       // step 2
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor2<I + 2, N>(f);
    }
}

