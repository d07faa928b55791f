// twx_hip.hip -- host side of libtwxhip.so: the C ABI of include/twx.h on top of
// the gfx950 kernels in twx_select.h / twx_uk.h / twx_daily.h / twx_out.h.
//
// Plain HIP runtime only (no torch, no CPU fallback): every entry point either
// runs the kernels on the context's GPU or fails with a negative status.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "twx.h"
#include "twx_daily.h"
#include "twx_out.h"
#include "twx_uk.h"
#include "twx_ukw.h"
#include "twx_vario.h"
#include "twx_agg.h"
#include "twx_sample.h"
#include "twx_deflate.h"

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct VarData {
    int n = 0, kmax = 0;
    bool has_obs = false;
    DevBuf cols, obs, ymsum;
    StnDev dev{};
};

struct Work {
    DevBuf cand, ncand, small, dscratch, near_idx, near_dist, nnear, kk, ka, vario, cstat, cdup,
        bucket_cells, uk_mean, uk_var, uk_stat, z, zc, gstat, ctrig, uk_S, uk_beta, vfit, dist, h0, hminp, noff, near_pos, urow,
        nurow, zd, perm, kp, uslot, cellf64, f64_cells, dist64, h064, gd64, zn;
    int cmax = TWX_CAND_SMALL;   // candidate slots per tile of the current batch
    SelWs ws{};
    GwrWs gw{};
    void release()
    {
        for (DevBuf *b : {&cand, &ncand, &small, &dscratch, &near_idx, &near_dist, &nnear, &kk, &ka, &vario, &cstat,
                          &cdup, &bucket_cells, &uk_mean, &uk_var, &uk_stat, &z, &zc, &gstat, &ctrig, &uk_S, &uk_beta, &vfit, &dist, &h0, &hminp, &noff,
                          &near_pos, &urow, &nurow, &zd, &perm, &kp, &uslot, &cellf64, &f64_cells, &dist64, &h064, &gd64, &zn})
            b->release();
    }
};

enum { EV_TILE = 0, EV_SELECT, EV_UK, EV_GWR, EV_DAILY, EV_FIX, EV_TIE, EV_DEFLATE, EV_NKIND };

struct EvPair { hipEvent_t a, b; int kind; };

}  // namespace

#ifndef TWX_UKW2
#define TWX_UKW2 1      // two systems per wave (k_ukw2, twx_ukw.h): 1 = the bucket of <= 40 neighbours (-6..8 % there), 2 = also
                        // 49..56 neighbours (same-box A/B: 518 vs 509 us -- the shared chain is paid back by 2 instead of 4 waves
                        // per SIMD), 0 = nowhere
#endif
#ifndef TWX_FIX_THREADS
#define TWX_FIX_THREADS 512
#endif
#ifndef TWX_DAILY_BATCH
#define TWX_DAILY_BATCH 65536   // cells per batch when daily output is requested (workspace ~110 KB per cell and variable: 14 GB; a C2 / C4 tile is one batch: 71.3 -> 68.6 ms per C4 tile against 32 768)
#endif

struct twx_ctx {
    int device = 0;
    bool select_lds_set = false;   // k_select<1,1> may use TWX_CAND_LDS_MAX candidates' worth of dynamic LDS (set on first need)
    twx_params p{};
    VarData var[2];
    Work work[2];
    // day axis
    int64_t ndays = 0;
    std::vector<int32_t> day_month, day_year, mm2chron, chron2mm;
    std::vector<int32_t> ym_start, ym_cnt;   // host copies of DayAxis.ym_start / ym_cnt ((year, month) runs of the normals period)
    int norm_y0 = 0, norm_ny = 0;
    DevBuf day_dev;
    DayAxis da{};
    // (year, month) groups of the aggregation entry (tiling.py:1085-1118)
    DevBuf agg_dev, agg_in, agg_out;
    AggAxis agg{};
    hipEvent_t ev_agg_a = nullptr, ev_agg_b = nullptr;
    // scratch for the point entries / fixer
    DevBuf pt_in, pt_aux, pt_out, fix_scratch, fix_lists, flags, flag_list, inv_cnt, inv_day, tie;
    DevBuf grid_in, grid_out;     // persistent device images of the host-buffer grid entry
    // every context-level device buffer (the per-variable ones live in var[] / work[]): twx_destroy releases these
    std::vector<DevBuf *> all_bufs()
    {
        return {&day_dev, &agg_dev, &agg_in, &agg_out, &pt_in, &pt_aux, &pt_out, &fix_scratch, &fix_lists, &flags, &flag_list, &inv_cnt, &inv_day, &tie, &stats,
                &grid_in, &grid_out};
    }
    std::string err;
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    twx_timing timing{};
    int64_t t_cells = 0;
    DevBuf stats;                 // [5] int64: kriging systems solved, kriging launches with work, systems on the fp64 build, systems / cells of
                                  // the tie guard's second pass (device-side counters: k_bucket_stats)
    int ncu = 256;                // compute units (sizes the fixed grids of the kriging launches)
    hipEvent_t ev_total_a = nullptr, ev_total_b = nullptr;
    bool have_total = false;
    std::vector<struct twx_stream *> streams;   // open twx_stream objects: twx_destroy closes them before the context goes away
    std::vector<int32_t> excl_lists;            // twx_set_exclusions: [excl_npts][excl_nmax] for the next point-entry call
    int64_t excl_npts = 0;
    int excl_nmax = 0;
};

namespace {

int fail(twx_ctx *ctx, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return -1;
}

#define HIPCHK(call)                                                         \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return fail(ctx, #call, e_);                   \
    } while (0)

struct EvScope {
    twx_ctx *ctx; hipStream_t s; size_t i;
    EvScope(twx_ctx *c, hipStream_t st, int kind) : ctx(c), s(st)
    {
        if (ctx->ev_used == ctx->ev_pool.size()) {
            EvPair e{};
            (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b);
            ctx->ev_pool.push_back(e);
        }
        i = ctx->ev_used++;
        ctx->ev_pool[i].kind = kind;
        (void)hipEventRecord(ctx->ev_pool[i].a, s);
    }
    ~EvScope() { (void)hipEventRecord(ctx->ev_pool[i].b, s); }
};

template <class T> T *carve(char *&cur, size_t count)
{
    T *r = reinterpret_cast<T *>(cur);
    cur += (count * sizeof(T) + 255) / 256 * 256;
    return r;
}

// ---- workspace of one (batch, variable) ---------------------------------------------
int prepare_work(twx_ctx *ctx, int v, int64_t cell0, int64_t ncell, int64_t tile0, int64_t ntile, int ksel,
                 int nblocks_tile, bool need_gwr, bool fit_vario = false, bool tile_tab = false, bool grid = false)
{
    Work &w = ctx->work[v];
    const int n = ctx->var[v].n;
    HIPCHK(w.cand.ensure((size_t)ntile * w.cmax * 4));
    HIPCHK(w.ncand.ensure((size_t)ntile * 4));
    HIPCHK(w.small.ensure(256));
    HIPCHK(w.dscratch.ensure((size_t)nblocks_tile * n * 4));
    HIPCHK(w.near_idx.ensure((size_t)ncell * ksel * 4));
    HIPCHK(w.near_dist.ensure((size_t)ncell * ksel * 8));
    HIPCHK(w.nnear.ensure((size_t)ncell * 4));
    HIPCHK(w.kk.ensure((size_t)ncell * 48));
    HIPCHK(w.ka.ensure((size_t)ncell * 48));
    HIPCHK(w.vario.ensure((size_t)ncell * 36 * 8));
    HIPCHK(w.cstat.ensure((size_t)ncell * 4));
    HIPCHK(w.cdup.ensure((size_t)ncell * 4));
    HIPCHK(w.bucket_cells.ensure((size_t)ncell * 12 * TWX_NBUCKET * 4));
    HIPCHK(w.uk_mean.ensure((size_t)ncell * 96));
    HIPCHK(w.uk_var.ensure((size_t)ncell * 96));
    HIPCHK(w.uk_stat.ensure((size_t)ncell * 4));
    HIPCHK(w.ctrig.ensure((size_t)ncell * 32));
    HIPCHK(w.uk_S.ensure((size_t)ncell * 12 * TWX_UK_SLEN * 8));
    HIPCHK(w.dist.ensure((size_t)ncell * TWX_DIST_BLOCKS * 256 * 4));   // pair distances shared by a cell's 12 systems
    HIPCHK(w.h0.ensure((size_t)ncell * ksel * 4));
    HIPCHK(w.hminp.ensure((size_t)ncell * ksel * 4));
    HIPCHK(w.cellf64.ensure((size_t)ncell * 4));
    HIPCHK(w.f64_cells.ensure((size_t)ncell * 4));
    if (need_gwr) {
        HIPCHK(w.z.ensure((size_t)ncell * 12 * TWX_KZ * 8));
        HIPCHK(w.noff.ensure((size_t)ncell * ksel * 4));
        HIPCHK(w.zc.ensure((size_t)ncell * 96));
        HIPCHK(w.zn.ensure((size_t)ncell * 96));
        HIPCHK(w.gstat.ensure((size_t)ncell * 4));
        HIPCHK(w.perm.ensure((size_t)ncell * ksel * 4));     // ranks in ascending station-index order (k_perm)
        HIPCHK(w.kp.ensure((size_t)ncell * 4));
    }
    if (grid) HIPCHK(w.near_pos.ensure((size_t)ncell * ksel * 2));   // position in the tile's candidate list (k_tile_dist, k_tile_uidx, k_gwr_z_cell)
    if (tile_tab) {      // grid mode with daily output: k_tile_uidx / k_gwr_z_cell / k_daily_tile
        HIPCHK(w.zd.ensure((size_t)ntile * 12 * 64 * TWX_UROWS * 8));   // per tile: 64 cell slots x TWX_UROWS rows, wave layout
        HIPCHK(w.urow.ensure((size_t)ntile * 12 * TWX_UROWS * 4));
        HIPCHK(w.nurow.ensure((size_t)ntile * 12 * 4));
        HIPCHK(w.uslot.ensure((size_t)ntile * 12 * w.cmax * 2));
    }
    if (fit_vario) {
        HIPCHK(w.uk_beta.ensure((size_t)ncell * 12 * 5 * 8));
        HIPCHK(w.vfit.ensure((size_t)ncell * 12 * 3 * 8));
        HIPCHK(w.gd64.ensure((size_t)ntile * ((size_t)ksel * (ksel - 1) / 2) * 8));   // the variogram's pair distances per point list
    }
    SelWs &s = w.ws;
    s.uk_beta = fit_vario ? w.uk_beta.as<double>() : nullptr;
    s.vfit = fit_vario ? w.vfit.as<double>() : nullptr;
    s.gd64 = fit_vario ? w.gd64.as<double>() : nullptr;
    s.ksel = ksel; s.cmax = w.cmax; s.init_nnghs = ctx->p.init_nnghs;
    s.cell0 = cell0; s.ncell = ncell; s.tile0 = tile0; s.ntile = ntile;
    s.cand = w.cand.as<int32_t>(); s.ncand = w.ncand.as<int32_t>();
    s.ncand_max = w.small.as<int32_t>();          // [0]
    s.nf64 = w.small.as<int32_t>() + 1;           // [1] cells with a month on the fp64 covariance build
    s.bucket_cnt = w.small.as<int32_t>() + 16;    // [16 .. 16 + TWX_NBUCKET)
    s.f64_sized = (ctx->p.flags & TWX_FLAG_NO_HOST_SYNC) ? 0 : 1;
    s.reserved0 = 0;
    s.dscratch = w.dscratch.as<float>();
    s.near_idx = w.near_idx.as<int32_t>(); s.near_dist = w.near_dist.as<double>();
    s.nnear = w.nnear.as<int32_t>(); s.kk = w.kk.as<int32_t>(); s.ka = w.ka.as<int32_t>();
    s.vario = w.vario.as<double>(); s.cstat = w.cstat.as<int32_t>(); s.cdup = w.cdup.as<int32_t>();
    s.bucket_cells = w.bucket_cells.as<int32_t>();
    s.uk_mean = w.uk_mean.as<double>(); s.uk_var = w.uk_var.as<double>(); s.uk_stat = w.uk_stat.as<int32_t>();
    s.ctrig = w.ctrig.as<double>(); s.uk_S = w.uk_S.as<double>();
    s.dist = w.dist.as<float>(); s.h0 = w.h0.as<float>(); s.hminp = w.hminp.as<float>();
    s.fast_only = (ctx->p.flags & TWX_FLAG_UK_FAST_ONLY) ? 1 : 0;
    s.f64_all = (ctx->p.flags & TWX_FLAG_UK_F64_ALL) ? 1 : 0;
    s.cellf64 = w.cellf64.as<int32_t>(); s.f64_cells = w.f64_cells.as<int32_t>(); s.dist64 = nullptr; s.h064 = nullptr;
    s.rerun = nullptr;
    w.gw.z = w.z.as<double>(); w.gw.zc = w.zc.as<double>(); w.gw.zn = w.zn.as<double>(); w.gw.gstat = w.gstat.as<int32_t>();
    w.gw.noff = w.noff.as<uint32_t>();
    w.gw.perm = w.perm.as<int32_t>(); w.gw.kp = w.kp.as<int32_t>();
    w.gw.uslot = tile_tab ? w.uslot.as<uint16_t>() : nullptr;
    w.gw.nurow2 = nullptr; w.gw.use_table = 0;                           // (twx_interp_grid_dev switches it on: both variables, 8x8 tiles, no gather flag)
    s.near_pos = grid ? w.near_pos.as<uint16_t>() : nullptr;
    w.gw.zd = tile_tab ? w.zd.as<double>() : nullptr;
    w.gw.urow = tile_tab ? w.urow.as<int32_t>() : nullptr;
    w.gw.nurow = tile_tab ? w.nurow.as<int32_t>() : nullptr;
    return 0;
}

// Grid of a kriging launch.  Default: the host reads the 16 bucket counts back once per (batch, variable) -- one
// 64-byte copy that waits for the selection kernels -- and launches exactly one work-group per system.  With
// TWX_FLAG_NO_HOST_SYNC the counts stay on the device and every launch covers the worst case (all systems of the
// batch in one bucket); surplus work-groups exit on their first instruction (measured on the C2 tile: 9 M surplus
// work-groups per step = +1.2 ms = +9 % kriging time, which is why it is not the default).
inline unsigned krig_grid(const int32_t *cnt, int bucket, int64_t max_items)
{
    // (a multiple of 8: the kernels deal the item list to the 8 XCDs in contiguous eighths, uk_item())
    return (unsigned)((std::max<int64_t>(1, cnt ? (int64_t)cnt[bucket] : max_items) + 7) / 8 * 8);
}

template <int NB, int PREC = 0>
void launch_uk(const int32_t *cnt, const StnDev &st, const CellSrc &src, const SelWs &ws, int bucket, int64_t max_items, hipStream_t s)
{
    const int32_t *cells = ws.bucket_cells + (int64_t)bucket * ws.ncell * 12;
    if (cnt && cnt[bucket] <= 0) return;
    const unsigned grid = krig_grid(cnt, bucket, max_items);
#ifdef TWX_UK_STAMP   // diagnostic build only: stamp the NB = 7 launch, dump the stamps next to the working directory
    static unsigned long long *dbg = nullptr;
    const size_t nb = (size_t)2048 * 40 * 4 * 4 * 8;
    SelWs w2 = ws;
    w2.dbg = nullptr;
    if (NB == 7) {
        if (!dbg) (void)hipMalloc(&dbg, nb);
        (void)hipMemsetAsync(dbg, 0, nb, s);
        w2.dbg = dbg;
    }
    hipLaunchKernelGGL((k_uk<NB, twx_uk_nw(NB), PREC>), dim3(grid), dim3(64 * twx_uk_nw(NB)), 0, s, st, src, w2, cells, ws.bucket_cnt + bucket);
    if (NB == 7) {
        std::vector<unsigned long long> h(nb / 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), dbg, nb, hipMemcpyDeviceToHost);
        if (FILE *f = fopen("gpurun_out/uk_stamps.bin", "wb")) { fwrite(h.data(), 1, nb, f); fclose(f); }
    }
#else
    hipLaunchKernelGGL((k_uk<NB, twx_uk_nw(NB), PREC>), dim3(grid), dim3(64 * twx_uk_nw(NB)), 0, s, st, src, ws, cells, ws.bucket_cnt + bucket);
#endif
}

template <int NBR, int PREC = 0>
void launch_ukwz(const int32_t *cnt, const StnDev &st, const CellSrc &src, const SelWs &ws, int bucket, int64_t max_items, hipStream_t s)
{
    const int32_t *cells = ws.bucket_cells + (int64_t)bucket * ws.ncell * 12;
    if (cnt && cnt[bucket] <= 0) return;
    const unsigned grid = krig_grid(cnt, bucket, max_items);
    hipLaunchKernelGGL((k_ukwz<NBR, PREC>), dim3(grid), dim3(64), 0, s, st, src, ws, cells, ws.bucket_cnt + bucket);
}

template <int NBR, int PREC = 0>
void launch_ukw2(const int32_t *cnt, const StnDev &st, const CellSrc &src, const SelWs &ws, int bucket, int64_t max_items, hipStream_t s)
{
    const int32_t *cells = ws.bucket_cells + (int64_t)bucket * ws.ncell * 12;
    if (cnt && cnt[bucket] <= 0) return;
    // two systems per wave: half as many work-groups (a multiple of 8, krig_grid)
    const unsigned grid = (unsigned)(((std::max<int64_t>(1, cnt ? (int64_t)cnt[bucket] : max_items) + 1) / 2 + 7) / 8 * 8);
    hipLaunchKernelGGL((k_ukw2<NBR, PREC>), dim3(grid), dim3(64), 0, s, st, src, ws, cells, ws.bucket_cnt + bucket);
}

template <int NBR, int PREC = 0>
void launch_ukw(const int32_t *cnt, const StnDev &st, const CellSrc &src, const SelWs &ws, int bucket, int64_t max_items, hipStream_t s)
{
    const int32_t *cells = ws.bucket_cells + (int64_t)bucket * ws.ncell * 12;
    if (cnt && cnt[bucket] <= 0) return;
    const unsigned grid = krig_grid(cnt, bucket, max_items);
    hipLaunchKernelGGL((k_ukw<NBR, PREC>), dim3(grid), dim3(64), 0, s, st, src, ws, cells, ws.bucket_cnt + bucket);
}

// The kriging stage of a (batch, variable) whose selection, variograms (ws.vario) and pair distances are in place: bucket the
// (cell, month) systems by matrix size, read the counts back (unless TWX_FLAG_NO_HOST_SYNC), fp64 slabs for routed cells,
// one launch per bucket, the 7x7 epilogue.  Returns 1 when a grid batch must be re-run with longer candidate lists
// (*cmax_wanted), 0 when done, -1 on error.  Also the second stage of twx_krigall_points (new variograms, same selection).
// rerun (the tie guard, run_tie_guard): [ncell] device flags -- only the flagged cells' systems are listed, all on the fp64 build; the
// bucket counts are reset first, the time goes to EV_TIE and the systems to the guard's own counters.
int run_uk_stage(twx_ctx *ctx, int v, const CellSrc &src, int64_t ncell, int ksel, hipStream_t stream, bool may_retry, int *cmax_wanted,
                 const int32_t *rerun = nullptr)
{
    Work &w = ctx->work[v];
    const StnDev &st = ctx->var[v].dev;
    const int evk = rerun ? EV_TIE : EV_UK;
    w.ws.rerun = rerun;
    if (rerun) {
        HIPCHK(hipMemsetAsync(w.small.as<int32_t>() + 1, 0, 4, stream));                              // routed cells
        HIPCHK(hipMemsetAsync(w.small.as<int32_t>() + 16, 0, (size_t)TWX_NBUCKET * 4, stream));       // bucket counts
    }
    HIPCHK(hipMemsetAsync(w.cellf64.p, 0, (size_t)ncell * 4, stream));
    hipLaunchKernelGGL(k_bucket_items, dim3((unsigned)((ncell * 12 + 255) / 256)), dim3(256), 0, stream, w.ws);
    int32_t small_host[16 + TWX_NBUCKET];                    // [0] longest candidate list, [1] cells on the fp64 build, [16..] bucket counts
    const int32_t *cnt = nullptr;
    if (!(ctx->p.flags & TWX_FLAG_NO_HOST_SYNC)) {
        HIPCHK(hipMemcpyAsync(small_host, w.small.p, sizeof small_host, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        cnt = small_host + 16;
        if (may_retry && src.mode == 0 && small_host[0] > w.cmax && w.cmax < TWX_CAND_LDS_MAX) {
            *cmax_wanted = std::min((small_host[0] + 255) / 256 * 256, TWX_CAND_LDS_MAX);
            return 1;
        }
    }
    // launch statistics: only for the pass whose kriging launches run (a batch re-run with longer candidate lists must not
    // count its buckets twice: bench.py quotes systems_on_fp64_covariance_build from these counters)
    HIPCHK(ctx->stats.ensure(64));
    hipLaunchKernelGGL(k_bucket_stats, dim3(1), dim3(64), 0, stream, w.ws, ctx->stats.as<long long>());
    const int nf64 = cnt ? small_host[1] : 0;
    if (nf64 > 0) {
        // this batch has systems on the fp64 covariance build: the fp64 pair distances of their cells, once per cell, in slabs
        // sized by the number of ROUTED cells (110 KB each; a dense allocation for the whole batch was 14.8 GB per variable)
        HIPCHK(w.dist64.ensure((size_t)nf64 * TWX_DIST_BLOCKS * 256 * 8));
        HIPCHK(w.h064.ensure((size_t)nf64 * ksel * 8));
        w.ws.dist64 = w.dist64.as<double>(); w.ws.h064 = w.h064.as<double>();
        EvScope ev(ctx, stream, evk);
        hipLaunchKernelGGL(k_cell_dist64, dim3((unsigned)nf64), dim3(256), 0, stream, st, w.ws);
    }
    {
        EvScope ev(ctx, stream, evk);
        const int64_t mi = ncell * 12;
        // buckets of 8 neighbours (twx_krig_bucket): bordered one-wave kernels, one-wave kernels with the border as
        // columns (k in the upper half of a block row), two- / four-wave kernels from 97 neighbours on
        if (!rerun) {
        launch_ukwz<6>(cnt, st, src, w.ws, 7, mi, stream);      // 88 < k <= 96
        launch_ukw<6>(cnt, st, src, w.ws, 6, mi, stream);    // 80 < k <= 88
#if TWX_UKW2
        launch_ukw2<3>(cnt, st, src, w.ws, 0, mi, stream);      //      k <= 40  (two systems per wave)
#else
        launch_ukw<3>(cnt, st, src, w.ws, 0, mi, stream);    //      k <= 40
#endif
        launch_ukwz<3>(cnt, st, src, w.ws, 1, mi, stream);      // 40 < k <= 48
#if TWX_UKW2 >= 2
        launch_ukw2<4>(cnt, st, src, w.ws, 2, mi, stream);      // 48 < k <= 56  (two systems per wave: measured, no gain)
#else
        launch_ukw<4>(cnt, st, src, w.ws, 2, mi, stream);    // 48 < k <= 56
#endif
        launch_ukwz<4>(cnt, st, src, w.ws, 3, mi, stream);      // 56 < k <= 64
        launch_ukw<5>(cnt, st, src, w.ws, 4, mi, stream);    // 64 < k <= 72
        launch_ukwz<5>(cnt, st, src, w.ws, 5, mi, stream);      // 72 < k <= 80
        launch_uk<7>(cnt, st, src, w.ws, 8, mi, stream);        // 96 < k <= 104
        launch_uk<8>(cnt, st, src, w.ws, 10, mi, stream);       // 104 < k <= 120
        launch_uk<9>(cnt, st, src, w.ws, 12, mi, stream);       // 120 < k <= 136
        launch_uk<10>(cnt, st, src, w.ws, 13, mi, stream);      // 136 < k <= 152
        }
        constexpr int F = TWX_BUCKET_F64;
        if (w.ws.dist64) {       // ill-conditioned systems (uk_needs_f64): the fp64 covariance build of their own matrix size,
                                 // distances from the cells' fp64 slabs
#if TWX_UKW2
            launch_ukw2<3, 1>(cnt, st, src, w.ws, F + 0, mi, stream);
#else
            launch_ukw<3, 1>(cnt, st, src, w.ws, F + 0, mi, stream);
#endif
            launch_ukwz<3, 1>(cnt, st, src, w.ws, F + 1, mi, stream);
            launch_ukw<4, 1>(cnt, st, src, w.ws, F + 2, mi, stream);
            launch_ukwz<4, 1>(cnt, st, src, w.ws, F + 3, mi, stream);
            launch_ukw<5, 1>(cnt, st, src, w.ws, F + 4, mi, stream);
            launch_ukwz<5, 1>(cnt, st, src, w.ws, F + 5, mi, stream);
            launch_ukw<6, 1>(cnt, st, src, w.ws, F + 6, mi, stream);
            launch_ukwz<6, 1>(cnt, st, src, w.ws, F + 7, mi, stream);
            launch_uk<7, 1>(cnt, st, src, w.ws, F + 8, mi, stream);
            launch_uk<8, 1>(cnt, st, src, w.ws, F + 10, mi, stream);
            launch_uk<9, 1>(cnt, st, src, w.ws, F + 12, mi, stream);
            launch_uk<10, 1>(cnt, st, src, w.ws, F + 13, mi, stream);
        } else if (!cnt) {       // TWX_FLAG_NO_HOST_SYNC: no host decision possible, per-element distances, two worst-case sizes
            launch_uk<7, 2>(cnt, st, src, w.ws, F + 8, mi, stream);
            launch_uk<10, 2>(cnt, st, src, w.ws, F + 13, mi, stream);
        }
        hipLaunchKernelGGL(k_uk_solve, dim3((unsigned)((ncell * 12 + 255) / 256)), dim3(256), 0, stream, w.ws);
    }
    w.ws.rerun = nullptr;
    HIPCHK(hipGetLastError());
    return 0;
}

// tile candidates -> per-cell selection -> kriging, for one (batch, variable)
int run_select_uk(twx_ctx *ctx, int v, const CellSrc &src, int64_t cell0, int64_t ncell, int64_t tile0,
                  int64_t ntile, int ksel, bool need_gwr, hipStream_t stream, bool fit_vario = false, int cmax_retry = 0)
{
    Work &w = ctx->work[v];
    const StnDev &st = ctx->var[v].dev;
    const int nblk = (int)std::min<int64_t>(ntile, 2048);
    // Candidate lists have a fixed stride, chosen before anything is known about the batch.  The longest list of the batch
    // comes back with the bucket counts (one readback per (batch, variable), unless TWX_FLAG_NO_HOST_SYNC): a grid batch
    // with a tile that holds more than TWX_CAND_MAX candidates (thousands of stations inside one tile's search radius) is
    // run again with the stride that fits, up to TWX_CAND_LDS_MAX (what k_select<1,1> can rank in LDS) -- cmax_retry.  Only
    // past that -- or without the readback -- do a tile's cells fail with TWX_CELL_CAND_OVERFLOW.
    w.cmax = cmax_retry ? cmax_retry : (src.mode == 1 ? TWX_CAND_SMALL : TWX_CAND_MAX);
    const bool long_lists = w.cmax > TWX_CAND_MAX;           // (retry batch: no per-tile LDS tables, k_cell_dist / gather paths)
    if (long_lists) {
        bool &lds_set = ctx->select_lds_set;                 // > 64 KB of dynamic LDS has to be asked for, once
        if (!lds_set) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_select<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       TWX_CAND_LDS_MAX * (int)(sizeof(double) + 2)));
            lds_set = true;
        }
    }
    if (prepare_work(ctx, v, cell0, ncell, tile0, ntile, ksel, nblk, need_gwr, fit_vario, need_gwr && src.mode == 0 && !long_lists, src.mode == 0)) return -1;
    HIPCHK(hipMemsetAsync(w.small.p, 0, 256, stream));
    {
        EvScope ev(ctx, stream, EV_TILE);
        // grid mode has few tiles per batch (512 ... 2 048: at most a few work-groups per CU): 16 waves each; point mode one
        // "tile" per point and many of them: 4 waves each
        hipLaunchKernelGGL(k_tile_cand, dim3(nblk), dim3(src.mode == 0 ? 1024 : 256), st.n <= TWX_TC_LDS_STNS ? (size_t)st.n * 4 : 0, stream,
                           st, src, w.ws);
    }
    {
        EvScope ev(ctx, stream, EV_SELECT);
        hipLaunchKernelGGL((k_select<4, 0>), dim3((unsigned)(((ncell + 3) / 4 + 7) / 8 * 8)), dim3(256), (size_t)4 * TWX_CAND_SMALL * (sizeof(double) + 2),
                           stream, st, src, w.ws, 0, TWX_CAND_SMALL);
        if (w.cmax > TWX_CAND_SMALL)      // cells of tiles with long candidate lists (dense station clusters): one work-group per TILE
            hipLaunchKernelGGL((k_select<1, 1>), dim3((unsigned)ntile), dim3(64), (size_t)w.cmax * (sizeof(double) + 2), stream, st, src,
                               w.ws, TWX_CAND_SMALL, w.cmax);
    }
    if (!src.do_krig) return 0;
    if (fit_vario) { // model 1: OLS-residual variogram -> ws.vario, then the kriging kernels give the GLS trend
        hipLaunchKernelGGL(k_group_dist64, dim3((unsigned)ntile), dim3(256), 0, stream, st, src, w.ws);   // pair distances, once per point list
        hipLaunchKernelGGL(k_vario<0>, dim3((unsigned)(ncell * 12)), dim3(256), 0, stream, st, src, w.ws);
    }
    {
        // pair distances of every cell's largest neighbourhood, shared by its 12 monthly systems
        EvScope ev(ctx, stream, EV_UK);
        if (w.ws.near_pos && !long_lists)       // grid mode: per tile, from a table of the tile's station pairs
            hipLaunchKernelGGL(k_tile_dist, dim3((unsigned)(ntile * TWX_TD_PARTS)), dim3(64 * TWX_TD_WAVES), 0, stream, st, src, w.ws);
        else
            hipLaunchKernelGGL(k_cell_dist, dim3((unsigned)ncell), dim3(256), 0, stream, st, src, w.ws);
    }
    {
        int cmax_wanted = 0;
        const int rc = run_uk_stage(ctx, v, src, ncell, ksel, stream, !cmax_retry, &cmax_wanted);
        if (rc < 0) return -1;
        if (rc == 1) return run_select_uk(ctx, v, src, cell0, ncell, tile0, ntile, ksel, need_gwr, stream, fit_vario, cmax_wanted);
    }
    if (fit_vario)   // model 2: GLS-residual variogram -> ws.vfit
        hipLaunchKernelGGL(k_vario<1>, dim3((unsigned)(ncell * 12)), dim3(256), 0, stream, st, src, w.ws);
    HIPCHK(hipGetLastError());
    return 0;
}

int pick_ksel(const twx_ctx *ctx, int v, int kextra)
{
    int k = std::max(ctx->p.init_nnghs, std::max(ctx->var[v].kmax, kextra));
    k = std::min(k, TWX_MAX_NNGHS) + 1;
    return std::min(k, TWX_KSEL_MAX);
}

// before the hat rows: the one order of every daily sum (k_perm) and, in table mode, which rows the tables of the
// (tile, month)s hold (k_tile_uidx) -- the hat rows are then delivered in that order
int run_gwr_prep(twx_ctx *ctx, int v, const CellSrc &src, hipStream_t stream)
{
    Work &w = ctx->work[v];
    EvScope ev(ctx, stream, EV_GWR);
    hipLaunchKernelGGL(k_perm, dim3((unsigned)((w.ws.ncell + 3) / 4)), dim3(256), 0, stream, w.ws, w.gw);
    if (w.gw.use_table)
        hipLaunchKernelGGL(k_tile_uidx, dim3((unsigned)(w.ws.ntile * 12)), dim3(256), 0, stream, src, w.ws, w.gw);
    return 0;
}

int run_gwr(twx_ctx *ctx, int v, const CellSrc &src, const double *pt_norm_dev, hipStream_t stream, bool prep = true)
{
    Work &w = ctx->work[v];
    HIPCHK(hipMemsetAsync(w.gstat.p, 0, (size_t)w.ws.ncell * 4, stream));
    if (prep && run_gwr_prep(ctx, v, src, stream)) return -1;
    EvScope ev(ctx, stream, EV_GWR);
    int64_t items = w.ws.ncell * 12;
    if (src.mode == 0 && !pt_norm_dev)       // grid: one work-group per cell, the month-independent columns staged once
        hipLaunchKernelGGL(k_gwr_z_cell, dim3((unsigned)((w.ws.ncell + 7) / 8 * 8)), dim3(192), 0, stream, ctx->var[v].dev, src, w.ws, w.gw);
    else                                     // points: most of a point's twelve months are not asked for
        hipLaunchKernelGGL(k_gwr_z, dim3((unsigned)(((items + 15) / 16 + 7) / 8 * 8)), dim3(256), 0, stream, ctx->var[v].dev, src, w.ws,
                           w.gw, pt_norm_dev);
    return 0;
}

// The tie guard (twx_daily.h, note_day): the cells of this batch that have a day with |Tmax - Tmin| < TWX_TIE_EPS are kriged a second
// time, both variables, every month on the fp64 covariance build; their normals / SE are written again, their daily constants
// re-formed, and they enter the fixer's list flagged for a rewrite of their whole series (k_fix_cells).  Default mode: the number of
// such cells is read back (one 4-byte copy; the host has been waiting on this stream's selection kernels anyway) and nothing is
// launched when it is 0 -- the normal case.  TWX_FLAG_NO_HOST_SYNC: every launch covers the worst case.
int run_tie_guard(twx_ctx *ctx, const CellSrc (&src)[2], int64_t ncell, const twx_grid_out &o, hipStream_t stream)
{
    int32_t *d_tie = ctx->tie.as<int32_t>(), *d_list = d_tie + ncell, *d_cnt = d_list + ncell;
    hipLaunchKernelGGL(k_compact_flags, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, d_tie, ncell, d_list, d_cnt);
    int64_t ntie = ncell;
    if (!(ctx->p.flags & TWX_FLAG_NO_HOST_SYNC)) {
        int32_t h = 0;
        HIPCHK(hipMemcpyAsync(&h, d_cnt, 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (h == 0) return 0;
        ntie = h;
    }
    for (int v = 0; v < 2; ++v) {
        int dummy = 0;
        if (run_uk_stage(ctx, v, src[v], ncell, ctx->work[v].ws.ksel, stream, false, &dummy, d_tie) < 0) return -1;
    }
    EvScope ev(ctx, stream, EV_TIE);
    hipLaunchKernelGGL(k_tie_rezc, dim3((unsigned)((ntie * 24 + 255) / 256)), dim3(256), 0, stream, d_list, d_cnt, ctx->work[0].ws,
                       ctx->work[1].ws, ctx->work[0].gw, ctx->work[1].gw, ctx->flags.as<int32_t>(), ctx->stats.as<long long>());
    hipLaunchKernelGGL(k_finalize_grid, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, src[0], ctx->work[0].ws,
                       ctx->work[1].ws, 1, 1, ctx->work[0].gw.gstat, ctx->work[1].gw.gstat, o, 1, d_tie);
    HIPCHK(hipGetLastError());
    return 0;
}

int check_var(twx_ctx *ctx, int v, bool need_obs)
{
    if (v < 0 || v > 1) return fail(ctx, "var must be TWX_TMIN or TWX_TMAX");
    if (ctx->var[v].n <= 0) return fail(ctx, "no station table set for this variable (twx_set_stations)");
    if (need_obs && !ctx->var[v].has_obs) return fail(ctx, "this entry needs observations (station table was set with obs == NULL)");
    return 0;
}

}  // namespace

// =====================================================================================
extern "C" {

const char *twx_version(void) { return "topowx_amd libtwxhip 0.1 (gfx950)"; }

const char *twx_last_error(const twx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int twx_create(int device, const twx_params *params, twx_ctx **out)
{
    if (!out) return -1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return -2; // no GPU: fail loudly, no CPU path
    if (device < 0 || device >= ndev) return -3;
    if (hipSetDevice(device) != hipSuccess) return -4;
    twx_ctx *ctx = new twx_ctx();
    ctx->device = device;
    if (params) ctx->p = *params;
    if (ctx->p.init_nnghs <= 0) ctx->p.init_nnghs = 100;
    if (ctx->p.fixer_tail <= 0) ctx->p.fixer_tail = 15;
    if (ctx->p.norm_yr0 == 0 && ctx->p.norm_yr1 == 0) { ctx->p.norm_yr0 = 1981; ctx->p.norm_yr1 = 2010; }
    if (ctx->p.tile_cells <= 0) ctx->p.tile_cells = 8;
    if (ctx->p.init_nnghs > TWX_MAX_NNGHS) { delete ctx; return -5; }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->ncu = prop.multiProcessorCount;
    }
    (void)hipEventCreate(&ctx->ev_total_a);
    (void)hipEventCreate(&ctx->ev_total_b);
    *out = ctx;
    return 0;
}

void twx_stream_destroy(struct twx_stream *st);

void twx_destroy(twx_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    while (!ctx->streams.empty()) twx_stream_destroy(ctx->streams.back());   // (removes itself from the list)
    for (int v = 0; v < 2; ++v) { ctx->var[v].cols.release(); ctx->var[v].obs.release(); ctx->var[v].ymsum.release(); ctx->work[v].release(); }
    for (DevBuf *b : ctx->all_bufs()) b->release();
    for (auto &e : ctx->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (ctx->ev_total_a) (void)hipEventDestroy(ctx->ev_total_a);
    if (ctx->ev_total_b) (void)hipEventDestroy(ctx->ev_total_b);
    if (ctx->ev_agg_a) (void)hipEventDestroy(ctx->ev_agg_a);
    if (ctx->ev_agg_b) (void)hipEventDestroy(ctx->ev_agg_b);
    delete ctx;
}

int twx_set_precision(twx_ctx *ctx, int mode)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (mode != TWX_PRECISION_FAST && mode != TWX_PRECISION_EXACT) return fail(ctx, "twx_set_precision: mode must be TWX_PRECISION_FAST or TWX_PRECISION_EXACT");
    if (mode == TWX_PRECISION_EXACT) ctx->p.flags |= TWX_FLAG_UK_F64_ALL;
    else ctx->p.flags &= ~TWX_FLAG_UK_F64_ALL;
    return 0;
}

int twx_set_days(twx_ctx *ctx, int64_t ndays, const int32_t *day_month, const int32_t *day_year)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (ndays <= 0 || !day_month || !day_year) return fail(ctx, "twx_set_days: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    for (int v = 0; v < 2; ++v)
        if (ctx->var[v].has_obs) return fail(ctx, "twx_set_days must be called before twx_set_stations with observations");
    ctx->ndays = ndays;
    ctx->day_month.assign(day_month, day_month + ndays);
    ctx->day_year.assign(day_year, day_year + ndays);
    ctx->mm2chron.resize(ndays); ctx->chron2mm.resize(ndays);
    DayAxis &da = ctx->da;
    int pos = 0;
    for (int m = 1; m <= 12; ++m) {
        da.moff[m - 1] = pos;
        for (int64_t d = 0; d < ndays; ++d) {
            if (day_month[d] < 1 || day_month[d] > 12) return fail(ctx, "twx_set_days: month outside 1..12");
            if (day_month[d] == m) { ctx->mm2chron[pos] = (int32_t)d; ctx->chron2mm[d] = pos; ++pos; }
        }
    }
    da.moff[12] = pos;
    // (year, month) runs of the normals period (interp_tair.py:468-481)
    int y0 = 1 << 30, y1 = -(1 << 30);
    for (int64_t d = 0; d < ndays; ++d)
        if (day_year[d] >= ctx->p.norm_yr0 && day_year[d] <= ctx->p.norm_yr1) { y0 = std::min(y0, day_year[d]); y1 = std::max(y1, day_year[d]); }
    int ny = y0 <= y1 ? y1 - y0 + 1 : 0;
    std::vector<int32_t> ys((size_t)ny * 12, 0), yc((size_t)ny * 12, 0);
    for (int64_t d = ndays - 1; d >= 0; --d) {
        int y = day_year[d];
        if (ny == 0 || y < y0 || y > y1) continue;
        size_t s = (size_t)(y - y0) * 12 + (day_month[d] - 1);
        ys[s] = (int32_t)d; yc[s]++;
    }
    size_t total = (size_t)ndays * 4 * 4 + (size_t)ny * 12 * 8 + 4096;
    HIPCHK(ctx->day_dev.ensure(total));
    char *cur = ctx->day_dev.as<char>();
    int32_t *d_mm2c = carve<int32_t>(cur, ndays), *d_c2mm = carve<int32_t>(cur, ndays);
    int32_t *d_dm = carve<int32_t>(cur, ndays), *d_dy = carve<int32_t>(cur, ndays);
    int32_t *d_ys = carve<int32_t>(cur, (size_t)ny * 12 + 1), *d_yc = carve<int32_t>(cur, (size_t)ny * 12 + 1);
    HIPCHK(hipMemcpy(d_mm2c, ctx->mm2chron.data(), ndays * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_c2mm, ctx->chron2mm.data(), ndays * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_dm, day_month, ndays * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_dy, day_year, ndays * 4, hipMemcpyHostToDevice));
    if (ny) {
        HIPCHK(hipMemcpy(d_ys, ys.data(), ys.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_yc, yc.data(), yc.size() * 4, hipMemcpyHostToDevice));
    }
    da.ndays = (int)ndays; da.mm2chron = d_mm2c; da.chron2mm = d_c2mm; da.day_month = d_dm; da.day_year = d_dy;
    da.tail = ctx->p.fixer_tail; da.norm_ny = ny; da.norm_y0 = ny ? y0 : 0; da.ym_start = d_ys; da.ym_cnt = d_yc;
    ctx->ym_start = ys; ctx->ym_cnt = yc; ctx->norm_y0 = ny ? y0 : 0; ctx->norm_ny = ny;

    // _TairAggregate.__init__: groups = unique years x unique months, year-major; days ascending
    {
        int ya = day_year[0], yb = day_year[0];
        bool mhas[13] = {false};
        for (int64_t d = 0; d < ndays; ++d) { ya = std::min(ya, day_year[d]); yb = std::max(yb, day_year[d]); mhas[day_month[d]] = true; }
        std::vector<int> ymap((size_t)(yb - ya + 1), -1);
        for (int64_t d = 0; d < ndays; ++d) ymap[day_year[d] - ya] = 0;
        int nyr = 0, nmth = 0, mmap[13];
        for (auto &v : ymap) if (v == 0) v = nyr++;
        for (int m = 1; m <= 12; ++m) mmap[m] = mhas[m] ? nmth++ : -1;
        const int ng = nyr * nmth;
        std::vector<int32_t> gs((size_t)ng + 1, 0), gd((size_t)ndays);
        for (int64_t d = 0; d < ndays; ++d) gs[(size_t)ymap[day_year[d] - ya] * nmth + mmap[day_month[d]] + 1]++;
        for (int g = 0; g < ng; ++g) gs[g + 1] += gs[g];
        std::vector<int32_t> fillp(gs.begin(), gs.end() - 1);
        for (int64_t d = 0; d < ndays; ++d) gd[fillp[(size_t)ymap[day_year[d] - ya] * nmth + mmap[day_month[d]]]++] = (int32_t)d;
        HIPCHK(ctx->agg_dev.ensure(((size_t)ng + 1 + ndays) * 4 + 1024));
        char *ac = ctx->agg_dev.as<char>();
        int32_t *d_gs = carve<int32_t>(ac, (size_t)ng + 1), *d_gd = carve<int32_t>(ac, ndays);
        HIPCHK(hipMemcpy(d_gs, gs.data(), gs.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_gd, gd.data(), gd.size() * 4, hipMemcpyHostToDevice));
        ctx->agg.ng = ng; ctx->agg.nyr = nyr; ctx->agg.nmth = nmth; ctx->agg.gstart = d_gs; ctx->agg.gday = d_gd;
    }
    return 0;
}

int twx_set_stations(twx_ctx *ctx, int var, const twx_station_table *t)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (var < 0 || var > 1 || !t || t->n <= 0) return fail(ctx, "twx_set_stations: bad arguments");
    if (t->n > (1 << 24)) return fail(ctx, "twx_set_stations: too many stations");
    HIPCHK(hipSetDevice(ctx->device));
    VarData &vd = ctx->var[var];
    const size_t n = (size_t)t->n;
    // one allocation: 4 static + 7 monthly + 4 trig columns + cos(lat) (filled on the device) + the station-major
    // copies of optim_nnghs / optim_nnghs_anom ([n][12]) and of the variogram parameters ([n][12][4])
    const size_t ncol = 4 + 7 * 12 + 4 + 1 + 12 + 12 + 48 + 1;   // (+ nn_km: n floats in a column of their own, filled on the device)
    // ... + station records: (lon, lat, elev, tdi)[n] and (lst, norm)[n][12], 32-byte aligned behind the columns: what
    // the kriging / GWR staging reads of a neighbour comes with two or three 16-byte loads instead of five gathers
    const size_t rec0 = (ncol * n + 3) / 4 * 4;              // (in doubles)
    std::vector<double> host(rec0 + 4 * n + 24 * n);
    double *h = host.data();
    auto put = [&](const double *srcp, size_t cnt) { std::memcpy(h, srcp, cnt * 8); h += cnt; };
    put(t->lon, n); put(t->lat, n); put(t->elev, n); put(t->tdi, n);
    put(t->lst, 12 * n); put(t->norm, 12 * n); put(t->optim_nnghs, 12 * n); put(t->optim_nnghs_anom, 12 * n);
    put(t->vario_nug, 12 * n); put(t->vario_psill, 12 * n); put(t->vario_rng, 12 * n);
    const double r = 3.14159265358979323846 / 180.0;
    for (size_t i = 0; i < n; ++i) h[i] = std::sin(t->lat[i] * r / 2.0);
    for (size_t i = 0; i < n; ++i) h[n + i] = std::cos(t->lat[i] * r / 2.0);
    for (size_t i = 0; i < n; ++i) h[2 * n + i] = std::sin(t->lon[i] * r / 2.0);
    for (size_t i = 0; i < n; ++i) h[3 * n + i] = std::cos(t->lon[i] * r / 2.0);
    {
        double *os = h + 5 * n, *oa = os + 12 * n, *vs = oa + 12 * n;      // (h[4n .. 5n) is cos(lat), filled on the device)
        for (size_t i = 0; i < n; ++i)
            for (size_t m = 0; m < 12; ++m) {
                os[i * 12 + m] = t->optim_nnghs[m * n + i];
                oa[i * 12 + m] = t->optim_nnghs_anom[m * n + i];
                double *v = vs + (i * 12 + m) * 4;
                v[0] = t->vario_nug[m * n + i]; v[1] = t->vario_psill[m * n + i]; v[2] = t->vario_rng[m * n + i]; v[3] = 0.0;
            }
    }
    double km = 0;
    for (size_t i = 0; i < 12 * n; ++i) {
        if (std::isfinite(t->optim_nnghs[i])) km = std::max(km, t->optim_nnghs[i]);
        if (std::isfinite(t->optim_nnghs_anom[i])) km = std::max(km, t->optim_nnghs_anom[i]);
    }
    {
        double *ss = host.data() + rec0, *ms = ss + 4 * n;
        for (size_t i = 0; i < n; ++i) {
            ss[4 * i + 0] = t->lon[i]; ss[4 * i + 1] = t->lat[i]; ss[4 * i + 2] = t->elev[i]; ss[4 * i + 3] = t->tdi[i];
            for (size_t m = 0; m < 12; ++m) { ms[(i * 12 + m) * 2] = t->lst[m * n + i]; ms[(i * 12 + m) * 2 + 1] = t->norm[m * n + i]; }
        }
    }
    HIPCHK(vd.cols.ensure(host.size() * 8));
    HIPCHK(hipMemcpy(vd.cols.p, host.data(), host.size() * 8, hipMemcpyHostToDevice));
    const double *d = vd.cols.as<double>();
    StnDev &s = vd.dev;
    s.n = (int)n; s.kmax = (int)std::lrint(std::min(km, 1.0e6));
    s.lon = d; s.lat = d + n; s.elev = d + 2 * n; s.tdi = d + 3 * n;
    const double *mcol = d + 4 * n;
    s.lst = mcol; s.norm = mcol + 12 * n; s.optim = mcol + 24 * n; s.optim_anom = mcol + 36 * n;
    s.nug = mcol + 48 * n; s.psill = mcol + 60 * n; s.rng = mcol + 72 * n;
    const double *tcol = mcol + 84 * n;
    s.sph = tcol; s.cph = tcol + n; s.slh = tcol + 2 * n; s.clh = tcol + 3 * n;
    s.coslat = tcol + 4 * n;
    s.optim_s = tcol + 5 * n; s.optim_anom_s = s.optim_s + 12 * n; s.vario_s = s.optim_anom_s + 12 * n;
    s.stat_s = reinterpret_cast<const double4 *>(d + rec0); s.mon_s = reinterpret_cast<const double2 *>(d + rec0 + 4 * n);
    s.nn_km = reinterpret_cast<const float *>(s.vario_s + 48 * n);
    hipLaunchKernelGGL(k_stn_coslat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, s.lat, const_cast<double *>(s.coslat), (int)n);
    hipLaunchKernelGGL(k_stn_nn, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, s, const_cast<float *>(s.nn_km));
    HIPCHK(hipGetLastError());
    // a setup call: wait here, so that no later launch on a non-blocking stream (twx_stream_*, a caller's stream in
    // twx_interp_grid_dev) can read cos(lat) -- k_tile_cand's conservative radius -- before it is written, and so
    // that an asynchronous fault of these kernels is reported by this call
    HIPCHK(hipStreamSynchronize(nullptr));
    s.obs = nullptr; s.ymsum = nullptr;
    vd.n = (int)n; vd.kmax = s.kmax; vd.has_obs = false;
    if (t->obs) {
        if (ctx->ndays <= 0) return fail(ctx, "twx_set_stations: call twx_set_days before passing observations");
        const size_t nd = (size_t)ctx->ndays;
        // (time, station) -> [station][month-major day]: days become the contiguous axis
        // The database is serially complete (station_data.py:547-616): a NaN / Inf observation is rejected here, because
        // the daily table walk multiplies every row of a tile-month by every cell's weight (0 * NaN would reach all 64
        // cells of a tile, where the reference would only lose the cells that use the station)
        std::vector<float> tr(n * nd);
        const size_t B = 64;
        uint32_t nonfinite = 0;                            // set when an exponent field is all ones
        for (size_t j0 = 0; j0 < n; j0 += B)
            for (size_t p0 = 0; p0 < nd; p0 += B)
                for (size_t p = p0; p < std::min(nd, p0 + B); ++p) {
                    const float *row = t->obs + (size_t)ctx->mm2chron[p] * n;
                    for (size_t j = j0; j < std::min(n, j0 + B); ++j) {
                        const float x = row[j];
                        uint32_t b; std::memcpy(&b, &x, 4);
                        nonfinite |= ((b & 0x7f800000u) == 0x7f800000u) ? 1u : 0u;
                        tr[j * nd + p] = x;
                    }
                }
        if (nonfinite) return fail(ctx, "twx_set_stations: the observation matrix holds NaN / Inf (it must be serially complete)");
        HIPCHK(vd.obs.ensure(tr.size() * 4));
        HIPCHK(hipMemcpy(vd.obs.p, tr.data(), tr.size() * 4, hipMemcpyHostToDevice));
        s.obs = vd.obs.as<float>();
        vd.has_obs = true;
        // per station: the sum of its observations over every (month, year) of the normals period -- what the sparse fixer
        // forms a cell's recomputed normals from (k_fix_sparse) instead of re-evaluating ~11 000 daily values per cell
        s.ymsum = nullptr;
        const int ny = ctx->norm_ny;
        bool full_years = ny > 0;
        for (int q = 0; q < ny * 12; ++q) full_years = full_years && ctx->ym_cnt[q] > 0;
        if (full_years) {
            std::vector<double> ym(n * 12 * (size_t)ny, 0.0);
            for (int y = 0; y < ny; ++y)
                for (int m = 0; m < 12; ++m) {
                    const int d0 = ctx->ym_start[y * 12 + m], cntd = ctx->ym_cnt[y * 12 + m];
                    for (int d = d0; d < d0 + cntd; ++d) {
                        const float *row = t->obs + (size_t)d * n;
                        for (size_t j = 0; j < n; ++j) ym[(j * 12 + m) * ny + y] += (double)row[j];
                    }
                }
            HIPCHK(vd.ymsum.ensure(ym.size() * 8));
            HIPCHK(hipMemcpy(vd.ymsum.p, ym.data(), ym.size() * 8, hipMemcpyHostToDevice));
            s.ymsum = vd.ymsum.as<double>();
        }
    }
    return 0;
}

// ---- point entries -------------------------------------------------------------------
namespace {
struct PtDev {
    twx_pt *pts = nullptr;
    int32_t *mth = nullptr, *nnghs = nullptr, *excl = nullptr, *ptile = nullptr, *ptfirst = nullptr, *excl_more = nullptr;
    int nexcl = 0;
    double *vario = nullptr, *pt_norm = nullptr;
    int64_t nlists = 0;       // candidate lists: runs of consecutive points with the same location and excluded station
};

int upload_points(twx_ctx *ctx, int64_t npts, const twx_pt *pts, const double *lon, const double *lat,
                  const int32_t *mth, const int32_t *nnghs, const double *vario, const int32_t *excl,
                  const double *pt_norm, PtDev &pd)
{
    ctx->ev_used = 0;   // point entries do not report kernel timing: recycle the event pool
    ctx->have_total = false;
    HIPCHK(ctx->stats.ensure(64));                           // ... but they do count their systems (twx_get_timing: uk_solves, uk_f64_solves
    HIPCHK(hipMemsetAsync(ctx->stats.p, 0, 64, nullptr));    // of the LAST entry call, grid or points)
    // a pending exclusion list (twx_set_exclusions) belongs to THIS call: taken over and cleared, whatever happens next
    std::vector<int32_t> more;
    more.swap(ctx->excl_lists);
    const int64_t more_npts = ctx->excl_npts;
    const int nmore = ctx->excl_nmax;
    ctx->excl_npts = 0; ctx->excl_nmax = 0;
    if (!more.empty() && more_npts != npts) return fail(ctx, "the pending twx_set_exclusions list was given for another number of points");
    size_t bytes = (size_t)npts * (sizeof(twx_pt) + 5 * 4 + 4 * 8 + (size_t)nmore * 4) + 8192;
    HIPCHK(ctx->pt_in.ensure(bytes));
    char *cur = ctx->pt_in.as<char>();
    pd.pts = carve<twx_pt>(cur, npts);
    std::vector<twx_pt> tmp;
    if (!pts) {
        tmp.resize(npts);
        std::memset(tmp.data(), 0, npts * sizeof(twx_pt));
        for (int64_t i = 0; i < npts; ++i) { tmp[i].lon = lon[i]; tmp[i].lat = lat[i]; }
        pts = tmp.data();
    }
    HIPCHK(hipMemcpy(pd.pts, pts, npts * sizeof(twx_pt), hipMemcpyHostToDevice));
    if (mth) { pd.mth = carve<int32_t>(cur, npts); HIPCHK(hipMemcpy(pd.mth, mth, npts * 4, hipMemcpyHostToDevice)); }
    if (nnghs) { pd.nnghs = carve<int32_t>(cur, npts); HIPCHK(hipMemcpy(pd.nnghs, nnghs, npts * 4, hipMemcpyHostToDevice)); }
    if (excl) { pd.excl = carve<int32_t>(cur, npts); HIPCHK(hipMemcpy(pd.excl, excl, npts * 4, hipMemcpyHostToDevice)); }
    if (!more.empty()) {
        pd.excl_more = carve<int32_t>(cur, npts * nmore); pd.nexcl = nmore;
        HIPCHK(hipMemcpy(pd.excl_more, more.data(), more.size() * 4, hipMemcpyHostToDevice));
    }
    if (vario) { pd.vario = carve<double>(cur, npts * 3); HIPCHK(hipMemcpy(pd.vario, vario, npts * 24, hipMemcpyHostToDevice)); }
    if (pt_norm) { pd.pt_norm = carve<double>(cur, npts); HIPCHK(hipMemcpy(pd.pt_norm, pt_norm, npts * 8, hipMemcpyHostToDevice)); }
    // one candidate list per run of points that share location and excluded station (cross-validation asks for a
    // station x 16 bandwidths x 12 months: 192 points, one search through the station table instead of 192)
    std::vector<int32_t> ptile(npts), ptfirst;
    for (int64_t i = 0; i < npts; ++i) {
        const bool same = i > 0 && pts[i].lon == pts[i - 1].lon && pts[i].lat == pts[i - 1].lat &&
                          (excl ? excl[i] == excl[i - 1] : true) &&
                          (more.empty() || std::equal(more.begin() + i * nmore, more.begin() + (i + 1) * nmore, more.begin() + (i - 1) * nmore));
        if (!same) ptfirst.push_back((int32_t)i);
        ptile[i] = (int32_t)ptfirst.size() - 1;
    }
    pd.nlists = (int64_t)ptfirst.size();
    pd.ptile = carve<int32_t>(cur, npts); HIPCHK(hipMemcpy(pd.ptile, ptile.data(), npts * 4, hipMemcpyHostToDevice));
    pd.ptfirst = carve<int32_t>(cur, npts); HIPCHK(hipMemcpy(pd.ptfirst, ptfirst.data(), ptfirst.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

CellSrc point_src(const PtDev &pd, int rm_zero, int do_krig, int do_anom)
{
    CellSrc s{};
    s.mode = 1; s.pts = pd.pts; s.excl = pd.excl; s.excl_more = pd.excl_more; s.nexcl = pd.nexcl; s.mth = pd.mth; s.nnghs_in = pd.nnghs; s.vario_in = pd.vario;
    s.ptile = pd.ptile; s.ptfirst = pd.ptfirst;
    s.rm_zero = rm_zero; s.do_krig = do_krig; s.do_anom = do_anom; s.do_vario = 1;
    return s;
}

int max_k(const int32_t *nnghs, int64_t n)
{
    int m = 0;
    if (nnghs) for (int64_t i = 0; i < n; ++i) m = std::max(m, nnghs[i]);
    return m;
}
}  // namespace

int twx_set_exclusions(twx_ctx *ctx, int64_t npts, int32_t nmax, const int32_t *lists)
{
    if (!ctx) return -1;
    ctx->err.clear();
    ctx->excl_lists.clear(); ctx->excl_npts = 0; ctx->excl_nmax = 0;
    if (npts == 0) return 0;
    if (npts < 0 || nmax < 1 || nmax > TWX_MAX_EXCL || !lists)
        return fail(ctx, "twx_set_exclusions: 1 <= nmax <= TWX_MAX_EXCL station indices per point (more are refused, never truncated)");
    ctx->excl_lists.assign(lists, lists + npts * nmax);
    ctx->excl_npts = npts; ctx->excl_nmax = nmax;
    return 0;
}

int twx_knn(twx_ctx *ctx, int var, int64_t npts, const double *lon, const double *lat, int32_t k,
            const int32_t *excl, int rm_zero_dist, int32_t *idx, double *dist, double *wgt, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (check_var(ctx, var, false)) return -1;
    if (npts <= 0 || !lon || !lat || !idx || k < 1 || k >= TWX_KSEL_MAX) return fail(ctx, "twx_knn: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    PtDev pd;
    if (upload_points(ctx, npts, nullptr, lon, lat, nullptr, nullptr, nullptr, excl, nullptr, pd)) return -1;
    CellSrc src = point_src(pd, rm_zero_dist, 0, 0);
    if (run_select_uk(ctx, var, src, 0, npts, 0, pd.nlists, k + 1, false, nullptr)) return -1;
    size_t ob = (size_t)npts * k * (4 + 8 + 8) + (size_t)npts * 4 + 4096;
    HIPCHK(ctx->pt_out.ensure(ob));
    char *cur = ctx->pt_out.as<char>();
    int32_t *d_idx = carve<int32_t>(cur, npts * k);
    double *d_dist = carve<double>(cur, npts * k), *d_wgt = carve<double>(cur, npts * k);
    int32_t *d_st = carve<int32_t>(cur, npts);
    hipLaunchKernelGGL(k_sorted_neighbours, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, nullptr, src,
                       ctx->work[var].ws, (int)k, (int)k, d_idx, d_dist, d_wgt, d_st);
    HIPCHK(hipMemcpy(idx, d_idx, npts * k * 4, hipMemcpyDeviceToHost));
    if (dist) HIPCHK(hipMemcpy(dist, d_dist, npts * k * 8, hipMemcpyDeviceToHost));
    if (wgt) HIPCHK(hipMemcpy(wgt, d_wgt, npts * k * 8, hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, d_st, npts * 4, hipMemcpyDeviceToHost));
    return 0;
}

int twx_krig_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const int32_t *mth,
                    const int32_t *nnghs, const double *vario, const int32_t *excl, int rm_zero_dist,
                    double *mean, double *variance, int32_t *nnghs_used, int32_t *ngh_idx, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (check_var(ctx, var, false)) return -1;
    if (npts <= 0 || !pts || !mth || !mean || !variance || !status) return fail(ctx, "twx_krig_points: bad arguments");
    for (int64_t i = 0; i < npts; ++i)
        if (mth[i] < 1 || mth[i] > 12) return fail(ctx, "twx_krig_points: month outside 1..12");
    HIPCHK(hipSetDevice(ctx->device));
    PtDev pd;
    if (upload_points(ctx, npts, pts, nullptr, nullptr, mth, nnghs, vario, excl, nullptr, pd)) return -1;
    CellSrc src = point_src(pd, rm_zero_dist, 1, 0);
    const int ksel = pick_ksel(ctx, var, max_k(nnghs, npts));
    if (run_select_uk(ctx, var, src, 0, npts, 0, pd.nlists, ksel, false, nullptr)) return -1;
    const int ld = TWX_MAX_NNGHS;
    size_t ob = (size_t)npts * (8 + 8 + 4 + 4) + (size_t)npts * ld * 4 + 4096;
    HIPCHK(ctx->pt_out.ensure(ob));
    char *cur = ctx->pt_out.as<char>();
    double *d_mean = carve<double>(cur, npts), *d_var = carve<double>(cur, npts);
    int32_t *d_used = carve<int32_t>(cur, npts), *d_st = carve<int32_t>(cur, npts);
    int32_t *d_ngh = carve<int32_t>(cur, npts * ld);
    hipLaunchKernelGGL(k_finalize_krig_points, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, nullptr, src,
                       ctx->work[var].ws, d_mean, d_var, d_used, d_st);
    std::vector<double> hm(npts), hv(npts);
    HIPCHK(hipMemcpy(status, d_st, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hm.data(), d_mean, npts * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hv.data(), d_var, npts * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < npts; ++i)
        if (status[i] == 0) { mean[i] = hm[i]; variance[i] = hv[i]; }
    if (nnghs_used) HIPCHK(hipMemcpy(nnghs_used, d_used, npts * 4, hipMemcpyDeviceToHost));
    if (ngh_idx) {
        hipLaunchKernelGGL(k_sorted_neighbours, dim3((unsigned)((npts + 3) / 4)), dim3(256), 0, nullptr, src,
                           ctx->work[var].ws, 0, ld, d_ngh, (double *)nullptr, (double *)nullptr, (int32_t *)nullptr);
        HIPCHK(hipMemcpy(ngh_idx, d_ngh, npts * ld * 4, hipMemcpyDeviceToHost));
    }
    return 0;
}

int twx_krigall_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const int32_t *mth,
                       const int32_t *nnghs, const int32_t *excl, int rm_zero_dist, double *mean, double *variance,
                       double *vario, int32_t *nnghs_used, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (check_var(ctx, var, false)) return -1;
    if (npts <= 0 || !pts || !mth || !mean || !status) return fail(ctx, "twx_krigall_points: bad arguments");
    for (int64_t i = 0; i < npts; ++i)
        if (mth[i] < 1 || mth[i] > 12) return fail(ctx, "twx_krigall_points: month outside 1..12");
    HIPCHK(hipSetDevice(ctx->device));
    PtDev pd;
    if (upload_points(ctx, npts, pts, nullptr, nullptr, mth, nnghs, nullptr, excl, nullptr, pd)) return -1;
    CellSrc src = point_src(pd, rm_zero_dist, 1, 0);
    src.do_vario = 0;
    const int ksel = pick_ksel(ctx, var, max_k(nnghs, npts));
    // stage 1: selection, pair distances, the two variogram fits around the GLS kriging (twx_fit_vario_points)
    if (run_select_uk(ctx, var, src, 0, npts, 0, pd.nlists, ksel, false, nullptr, true)) return -1;
    Work &w = ctx->work[var];
    size_t ob = (size_t)npts * (8 + 8 + 4 + 4 + 4 + 24) + 4096;
    HIPCHK(ctx->pt_out.ensure(ob));
    char *cur = ctx->pt_out.as<char>();
    double *d_mean = carve<double>(cur, npts), *d_var = carve<double>(cur, npts), *d_vario = carve<double>(cur, npts * 3);
    int32_t *d_used = carve<int32_t>(cur, npts), *d_st = carve<int32_t>(cur, npts), *d_st1 = carve<int32_t>(cur, npts);
    // stage 2: the fitted models become the variograms, the same neighbourhoods are kriged again (interp.R:148-159)
    hipLaunchKernelGGL(k_vfit_to_vario, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, nullptr, src, w.ws, d_st1, d_vario);
    HIPCHK(hipMemsetAsync(w.small.as<int32_t>() + 1, 0, 4, nullptr));                              // routed cells
    HIPCHK(hipMemsetAsync(w.small.as<int32_t>() + 16, 0, (size_t)TWX_NBUCKET * 4, nullptr));       // bucket counts
    int dummy = 0;
    if (run_uk_stage(ctx, var, src, npts, ksel, nullptr, false, &dummy) < 0) return -1;
    hipLaunchKernelGGL(k_finalize_krig_points, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, nullptr, src, w.ws, d_mean,
                       d_var, d_used, d_st);
    std::vector<double> hm(npts), hv(npts), hvo((size_t)npts * 3);
    std::vector<int32_t> s1(npts), s2(npts);
    HIPCHK(hipMemcpy(s1.data(), d_st1, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(s2.data(), d_st, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hm.data(), d_mean, npts * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hv.data(), d_var, npts * 8, hipMemcpyDeviceToHost));
    if (vario) HIPCHK(hipMemcpy(hvo.data(), d_vario, (size_t)npts * 24, hipMemcpyDeviceToHost));
    if (nnghs_used) HIPCHK(hipMemcpy(nnghs_used, d_used, npts * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < npts; ++i) {
        status[i] = s1[i] ? s1[i] : s2[i];
        if (vario && s1[i] == 0) std::memcpy(vario + i * 3, hvo.data() + i * 3, 24);
        if (status[i] == 0) { mean[i] = hm[i]; if (variance) variance[i] = hv[i]; }
        else if (nnghs_used) nnghs_used[i] = 0;
    }
    return 0;
}

int twx_fit_vario_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const int32_t *mth,
                         const int32_t *nnghs, const int32_t *excl, int rm_zero_dist, double *vario,
                         int32_t *nnghs_used, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (check_var(ctx, var, false)) return -1;
    if (npts <= 0 || !pts || !mth || !vario || !status) return fail(ctx, "twx_fit_vario_points: bad arguments");
    for (int64_t i = 0; i < npts; ++i)
        if (mth[i] < 1 || mth[i] > 12) return fail(ctx, "twx_fit_vario_points: month outside 1..12");
    HIPCHK(hipSetDevice(ctx->device));
    PtDev pd;
    if (upload_points(ctx, npts, pts, nullptr, nullptr, mth, nnghs, nullptr, excl, nullptr, pd)) return -1;
    CellSrc src = point_src(pd, rm_zero_dist, 1, 0);
    src.do_vario = 0;
    const int ksel = pick_ksel(ctx, var, max_k(nnghs, npts));
    if (run_select_uk(ctx, var, src, 0, npts, 0, pd.nlists, ksel, false, nullptr, true)) return -1;
    HIPCHK(hipDeviceSynchronize());
    Work &w = ctx->work[var];
    std::vector<int32_t> cs(npts), us(npts), kk((size_t)npts * 12);
    std::vector<double> vf((size_t)npts * 36);
    HIPCHK(hipMemcpy(cs.data(), w.ws.cstat, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(us.data(), w.ws.uk_stat, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(kk.data(), w.ws.kk, npts * 48, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vf.data(), w.ws.vfit, vf.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < npts; ++i) {
        status[i] = cs[i] ? cs[i] : us[i];
        const int m0 = mth[i] - 1;
        if (nnghs_used) nnghs_used[i] = status[i] ? 0 : kk[i * 12 + m0];
        if (status[i] == 0) std::memcpy(vario + i * 3, vf.data() + (i * 12 + m0) * 3, 24);
    }
    return 0;
}

int twx_gwr_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const double *pt_norm,
                   const int32_t *mth, const int32_t *nnghs, const int32_t *excl, int rm_zero_dist,
                   double *out, int64_t ld, int32_t *nnghs_used, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (check_var(ctx, var, true)) return -1;
    if (npts <= 0 || !pts || !mth || !pt_norm || !out || !status) return fail(ctx, "twx_gwr_points: bad arguments");
    int maxd = 0;
    for (int m = 0; m < 12; ++m) maxd = std::max(maxd, ctx->da.moff[m + 1] - ctx->da.moff[m]);
    for (int64_t i = 0; i < npts; ++i) {
        if (mth[i] < 1 || mth[i] > 12) return fail(ctx, "twx_gwr_points: month outside 1..12");
        if (ctx->da.moff[mth[i]] - ctx->da.moff[mth[i] - 1] > ld) return fail(ctx, "twx_gwr_points: ld smaller than the days of the month");
    }
    HIPCHK(hipSetDevice(ctx->device));
    PtDev pd;
    if (upload_points(ctx, npts, pts, nullptr, nullptr, mth, nnghs, nullptr, excl, pt_norm, pd)) return -1;
    CellSrc src = point_src(pd, rm_zero_dist, 0, 1);
    const int ksel = pick_ksel(ctx, var, max_k(nnghs, npts));
    if (run_select_uk(ctx, var, src, 0, npts, 0, pd.nlists, ksel, true, nullptr)) return -1;
    if (run_gwr(ctx, var, src, pd.pt_norm, nullptr)) return -1;
    Work &w = ctx->work[var];
    HIPCHK(ctx->pt_out.ensure((size_t)npts * ld * 8 + 4096));
    double *d_out = ctx->pt_out.as<double>();
    HIPCHK(hipMemset(d_out, 0, (size_t)npts * ld * 8));
    hipLaunchKernelGGL(k_daily_points, dim3((unsigned)npts), dim3(256), 0, nullptr, ctx->var[var].dev, src, w.ws, w.gw,
                       ctx->da, d_out, ld, 1);
    std::vector<int32_t> cs(npts), gs(npts), ka((size_t)npts * 12);
    HIPCHK(hipMemcpy(cs.data(), w.ws.cstat, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gs.data(), w.gw.gstat, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ka.data(), w.ws.ka, npts * 48, hipMemcpyDeviceToHost));
    std::vector<double> ho((size_t)npts * ld);
    HIPCHK(hipMemcpy(ho.data(), d_out, ho.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < npts; ++i) {
        status[i] = cs[i] ? cs[i] : gs[i];
        if (nnghs_used) nnghs_used[i] = status[i] ? 0 : ka[i * 12 + mth[i] - 1];
        if (status[i] == 0) std::memcpy(out + i * ld, ho.data() + i * ld, (size_t)ld * 8);
    }
    return 0;
}

int twx_gwr_xval_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const double *pt_norm,
                        const int32_t *mth, const int32_t *nnghs, const int32_t *excl, int rm_zero_dist,
                        const int32_t *obs_idx, double *bias, double *mae, double *r2, int32_t *nnghs_used,
                        int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (check_var(ctx, var, true)) return -1;
    if (npts <= 0 || !pts || !mth || !pt_norm || !obs_idx || !bias || !mae || !r2 || !status)
        return fail(ctx, "twx_gwr_xval_points: bad arguments");
    for (int64_t i = 0; i < npts; ++i) {
        if (mth[i] < 1 || mth[i] > 12) return fail(ctx, "twx_gwr_xval_points: month outside 1..12");
        if (obs_idx[i] < 0 || obs_idx[i] >= ctx->var[var].n) return fail(ctx, "twx_gwr_xval_points: obs_idx outside the station table");
    }
    HIPCHK(hipSetDevice(ctx->device));
    PtDev pd;
    if (upload_points(ctx, npts, pts, nullptr, nullptr, mth, nnghs, nullptr, excl, pt_norm, pd)) return -1;
    CellSrc src = point_src(pd, rm_zero_dist, 0, 1);
    const int ksel = pick_ksel(ctx, var, max_k(nnghs, npts));
    if (run_select_uk(ctx, var, src, 0, npts, 0, pd.nlists, ksel, true, nullptr)) return -1;
    if (run_gwr(ctx, var, src, pd.pt_norm, nullptr)) return -1;
    Work &w = ctx->work[var];
    HIPCHK(ctx->pt_out.ensure((size_t)npts * (3 * 8 + 4) + 4096));
    HIPCHK(ctx->pt_aux.ensure((size_t)npts * 4 + 256));
    char *cur = ctx->pt_out.as<char>();
    double *d_bias = carve<double>(cur, npts), *d_mae = carve<double>(cur, npts), *d_r2 = carve<double>(cur, npts);
    int32_t *d_oi = ctx->pt_aux.as<int32_t>();
    HIPCHK(hipMemcpy(d_oi, obs_idx, npts * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ctx->pt_out.p, 0, (size_t)npts * 24));
    hipLaunchKernelGGL(k_xval_stats, dim3((unsigned)npts), dim3(256), 0, nullptr, ctx->var[var].dev, src, w.ws, w.gw, ctx->da,
                       pd.pt_norm, d_oi, d_bias, d_mae, d_r2);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> cs(npts), gs(npts), ka((size_t)npts * 12);
    std::vector<double> hb(npts), hm(npts), hr(npts);
    HIPCHK(hipMemcpy(cs.data(), w.ws.cstat, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gs.data(), w.gw.gstat, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ka.data(), w.ws.ka, npts * 48, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hb.data(), d_bias, npts * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hm.data(), d_mae, npts * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hr.data(), d_r2, npts * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < npts; ++i) {
        status[i] = cs[i] ? cs[i] : gs[i];
        if (nnghs_used) nnghs_used[i] = status[i] ? 0 : ka[i * 12 + mth[i] - 1];
        if (status[i] == 0) { bias[i] = hb[i]; mae[i] = hm[i]; r2[i] = hr[i]; }
    }
    return 0;
}

int twx_interp_points(twx_ctx *ctx, int var, int64_t npts, const twx_pt *pts, const int32_t *excl,
                      int rm_zero_dist, double *daily, double *norms, double *se, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (check_var(ctx, var, daily != nullptr)) return -1;
    if (npts <= 0 || !pts || !norms || !se || !status) return fail(ctx, "twx_interp_points: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    PtDev pd;
    if (upload_points(ctx, npts, pts, nullptr, nullptr, nullptr, nullptr, nullptr, excl, nullptr, pd)) return -1;
    CellSrc src = point_src(pd, rm_zero_dist, 1, daily ? 1 : 0);
    if (run_select_uk(ctx, var, src, 0, npts, 0, pd.nlists, pick_ksel(ctx, var, 0), daily != nullptr, nullptr)) return -1;
    Work &w = ctx->work[var];
    const int64_t nd = ctx->ndays;
    size_t ob = (size_t)npts * (96 * 2 + 4) + (daily ? (size_t)npts * nd * 8 : 0) + 4096;
    HIPCHK(ctx->pt_out.ensure(ob));
    char *cur = ctx->pt_out.as<char>();
    double *d_norm = carve<double>(cur, npts * 12), *d_se = carve<double>(cur, npts * 12);
    int32_t *d_st = carve<int32_t>(cur, npts);
    double *d_daily = daily ? carve<double>(cur, npts * nd) : nullptr;
    hipLaunchKernelGGL(k_finalize_interp_points, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, nullptr, w.ws,
                       d_norm, d_se, d_st);
    std::vector<int32_t> gs(npts, 0);
    if (daily) {
        if (run_gwr(ctx, var, src, nullptr, nullptr)) return -1;
        hipLaunchKernelGGL(k_daily_points, dim3((unsigned)npts), dim3(256), 0, nullptr, ctx->var[var].dev, src, w.ws,
                           w.gw, ctx->da, d_daily, nd, 0);
        HIPCHK(hipMemcpy(gs.data(), w.gw.gstat, npts * 4, hipMemcpyDeviceToHost));
    }
    std::vector<double> hn((size_t)npts * 12), hs((size_t)npts * 12);
    HIPCHK(hipMemcpy(status, d_st, npts * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hn.data(), d_norm, hn.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hs.data(), d_se, hs.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < npts; ++i) {
        if (status[i] == 0 && gs[i]) status[i] = gs[i];
        if (status[i]) continue;
        std::memcpy(norms + i * 12, hn.data() + i * 12, 96);
        std::memcpy(se + i * 12, hs.data() + i * 12, 96);
    }
    if (daily) {
        // one copy per run of consecutive successful points (failed points keep the caller's values)
        for (int64_t i = 0; i < npts;) {
            if (status[i]) { ++i; continue; }
            int64_t j = i;
            while (j < npts && status[j] == 0) ++j;
            HIPCHK(hipMemcpy(daily + i * nd, d_daily + i * nd, (size_t)(j - i) * nd * 8, hipMemcpyDeviceToHost));
            i = j;
        }
    }
    return 0;
}

int twx_fix_pair(twx_ctx *ctx, int64_t nseries, double *tmin, double *tmax, int32_t *ninvalid, double *norm_tmin,
                 double *norm_tmax, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (nseries <= 0 || !tmin || !tmax || !ninvalid || !status) return fail(ctx, "twx_fix_pair: bad arguments");
    if (ctx->ndays <= 0) return fail(ctx, "twx_fix_pair: call twx_set_days first");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nd = (size_t)ctx->ndays;
    const int nblk = (int)std::min<int64_t>(nseries, 1024);
    HIPCHK(ctx->fix_scratch.ensure((size_t)nseries * nd * 16 + (size_t)nseries * (8 + 192) + 4096));
    HIPCHK(ctx->fix_lists.ensure((size_t)nblk * nd * 4));
    char *cur = ctx->fix_scratch.as<char>();
    FixArgs fa{};
    fa.ncells = (int)nseries;
    fa.series_min = carve<double>(cur, nseries * nd); fa.series_max = carve<double>(cur, nseries * nd);
    fa.norm_min_out = carve<double>(cur, nseries * 12); fa.norm_max_out = carve<double>(cur, nseries * 12);
    fa.ninv_out = carve<int32_t>(cur, nseries); fa.status_out = carve<int32_t>(cur, nseries);
    fa.lists = ctx->fix_lists.as<int32_t>();
    HIPCHK(hipMemcpy(fa.series_min, tmin, nseries * nd * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(fa.series_max, tmax, nseries * nd * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fix_series, dim3(nblk), dim3(256), 0, nullptr, ctx->da, fa);
    HIPCHK(hipMemcpy(ninvalid, fa.ninv_out, nseries * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, fa.status_out, nseries * 4, hipMemcpyDeviceToHost));
    std::vector<double> a((size_t)nseries * nd), b((size_t)nseries * nd), nn((size_t)nseries * 12), nx((size_t)nseries * 12);
    HIPCHK(hipMemcpy(a.data(), fa.series_min, a.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data(), fa.series_max, b.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nn.data(), fa.norm_min_out, nn.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nx.data(), fa.norm_max_out, nx.size() * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nseries; ++i) {
        if (status[i]) continue; // the reference raises: the caller's series stay as they were
        std::memcpy(tmin + i * nd, a.data() + i * nd, nd * 8);
        std::memcpy(tmax + i * nd, b.data() + i * nd, nd * 8);
        if (ninvalid[i] > 0) {
            if (norm_tmin) std::memcpy(norm_tmin + i * 12, nn.data() + i * 12, 96);
            if (norm_tmax) std::memcpy(norm_tmax + i * 12, nx.data() + i * 12, 96);
        }
    }
    return 0;
}

int twx_pack_i16(twx_ctx *ctx, int64_t n, const double *x, int16_t *out)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (n <= 0 || !x || !out) return fail(ctx, "twx_pack_i16: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(ctx->pt_out.ensure((size_t)n * 10 + 512));
    char *cur = ctx->pt_out.as<char>();
    double *dx = carve<double>(cur, n);
    int16_t *dout = carve<int16_t>(cur, n);
    HIPCHK(hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dx, n, dout);
    HIPCHK(hipMemcpy(out, dout, n * 2, hipMemcpyDeviceToHost));
    return 0;
}

// ---- monthly / annual aggregation ------------------------------------------
int twx_aggregate_dims(twx_ctx *ctx, int32_t *nyr, int32_t *nmth)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (ctx->ndays <= 0) return fail(ctx, "twx_aggregate_dims: call twx_set_days first");
    if (nyr) *nyr = ctx->agg.nyr;
    if (nmth) *nmth = ctx->agg.nmth;
    return 0;
}

int twx_aggregate(twx_ctx *ctx, const void *daily, int dtype, int64_t ncell, int on_device, double *mthly,
                  int16_t *mthly_i16, double *ann, void *hip_stream, float *kernel_ms)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (!daily || ncell <= 0 || dtype < TWX_DT_I16 || dtype > TWX_DT_F64 || (!mthly && !mthly_i16 && !ann))
        return fail(ctx, "twx_aggregate: bad arguments");
    if (ctx->ndays <= 0) return fail(ctx, "twx_aggregate: call twx_set_days first");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    const size_t esz = dtype == TWX_DT_I16 ? 2 : dtype == TWX_DT_F32 ? 4 : 8;
    const size_t nd = (size_t)ctx->ndays, nc = (size_t)ncell, ng = (size_t)ctx->agg.ng, ny = (size_t)ctx->agg.nyr;
    const void *d_in = daily;
    double *d_m = mthly, *d_a = ann;
    int16_t *d_i = mthly_i16;
    if (!on_device) {
        HIPCHK(ctx->agg_in.ensure(nd * nc * esz));
        HIPCHK(ctx->agg_out.ensure(ng * nc * 10 + ny * nc * 8 + 1024));
        HIPCHK(hipMemcpyAsync(ctx->agg_in.p, daily, nd * nc * esz, hipMemcpyHostToDevice, stream));
        d_in = ctx->agg_in.p;
        char *cur = ctx->agg_out.as<char>();
        d_m = mthly ? carve<double>(cur, ng * nc) : nullptr;
        d_a = ann ? carve<double>(cur, ny * nc) : nullptr;
        d_i = mthly_i16 ? carve<int16_t>(cur, ng * nc) : nullptr;
    }
    if (!ctx->ev_agg_a) { HIPCHK(hipEventCreate(&ctx->ev_agg_a)); HIPCHK(hipEventCreate(&ctx->ev_agg_b)); }
    // widest vector whose loads stay aligned for every day row
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_in);
    HIPCHK(hipEventRecord(ctx->ev_agg_a, stream));
    if (dtype == TWX_DT_I16) {
        if (ncell % 8 == 0 && addr % 16 == 0) launch_agg<int16_t, 8>(d_in, ncell, ctx->agg, d_m, d_i, d_a, stream);
        else if (ncell % 4 == 0 && addr % 8 == 0) launch_agg<int16_t, 4>(d_in, ncell, ctx->agg, d_m, d_i, d_a, stream);
        else launch_agg<int16_t, 1>(d_in, ncell, ctx->agg, d_m, d_i, d_a, stream);
    } else if (dtype == TWX_DT_F32) {
        if (ncell % 4 == 0 && addr % 16 == 0) launch_agg<float, 4>(d_in, ncell, ctx->agg, d_m, d_i, d_a, stream);
        else launch_agg<float, 1>(d_in, ncell, ctx->agg, d_m, d_i, d_a, stream);
    } else {
        if (ncell % 2 == 0 && addr % 16 == 0) launch_agg<double, 2>(d_in, ncell, ctx->agg, d_m, d_i, d_a, stream);
        else launch_agg<double, 1>(d_in, ncell, ctx->agg, d_m, d_i, d_a, stream);
    }
    HIPCHK(hipEventRecord(ctx->ev_agg_b, stream));
    HIPCHK(hipGetLastError());
    if (!on_device) {
        if (mthly) HIPCHK(hipMemcpyAsync(mthly, d_m, ng * nc * 8, hipMemcpyDeviceToHost, stream));
        if (ann) HIPCHK(hipMemcpyAsync(ann, d_a, ny * nc * 8, hipMemcpyDeviceToHost, stream));
        if (mthly_i16) HIPCHK(hipMemcpyAsync(mthly_i16, d_i, ng * nc * 2, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
    }
    if (kernel_ms) {
        HIPCHK(hipEventSynchronize(ctx->ev_agg_b));
        HIPCHK(hipEventElapsedTime(kernel_ms, ctx->ev_agg_a, ctx->ev_agg_b));
    }
    return 0;
}

// ---- point-mode predictor sampling ------------------------------------------
int twx_sample_points(twx_ctx *ctx, const twx_raster *r, int64_t npts, const double *lon, const double *lat,
                      int order, double missing, double *val, int32_t *row, int32_t *col, int32_t *status)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (!r || r->nrows < 2 || r->ncols < 2 || !r->lon || !r->lat || !r->data || npts <= 0 || !lon || !lat || !val ||
        order < 0 || order > 1)
        return fail(ctx, "twx_sample_points: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nr = (size_t)r->nrows, nc = (size_t)r->ncols, n = (size_t)npts;
    HIPCHK(ctx->agg_in.ensure((nr + nc) * 8 + nr * nc * 4 + n * 16 + 2048));
    HIPCHK(ctx->agg_out.ensure(n * 20 + 2048));
    char *cur = ctx->agg_in.as<char>();
    double *d_lon = carve<double>(cur, nc), *d_lat = carve<double>(cur, nr);
    float *d_data = carve<float>(cur, nr * nc);
    double *d_qx = carve<double>(cur, n), *d_qy = carve<double>(cur, n);
    char *oc = ctx->agg_out.as<char>();
    double *d_val = carve<double>(oc, n);
    int32_t *d_row = carve<int32_t>(oc, n), *d_col = carve<int32_t>(oc, n), *d_st = carve<int32_t>(oc, n);
    HIPCHK(hipMemcpy(d_lon, r->lon, nc * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_lat, r->lat, nr * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_data, r->data, nr * nc * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_qx, lon, n * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_qy, lat, n * 8, hipMemcpyHostToDevice));
    RasterDev rd{r->nrows, r->ncols, d_lon, d_lat, d_data};
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, nullptr, rd, npts, d_qx, d_qy,
                       order, missing, d_val, d_row, d_col, d_st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(val, d_val, n * 8, hipMemcpyDeviceToHost));
    if (row) HIPCHK(hipMemcpy(row, d_row, n * 4, hipMemcpyDeviceToHost));
    if (col) HIPCHK(hipMemcpy(col, d_col, n * 4, hipMemcpyDeviceToHost));
    if (status) HIPCHK(hipMemcpy(status, d_st, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ---- grid entries ----------------------------------------------------------------------
int twx_interp_grid_dev(twx_ctx *ctx, const twx_grid *g, const twx_grid_out *o, int vars, void *hip_stream)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (!g || !o || g->Y <= 0 || g->X <= 0 || !g->mask || !g->lat || !g->lon || !g->elev || !g->tdi)
        return fail(ctx, "twx_interp_grid: bad grid");
    const bool has_n = vars & TWX_VAR_TMIN_BIT, has_x = vars & TWX_VAR_TMAX_BIT;
    if (!has_n && !has_x) return fail(ctx, "twx_interp_grid: no variable requested");
    const bool daily = (has_n && o->daily_tmin) || (has_x && o->daily_tmax);
    if (has_n && check_var(ctx, TWX_TMIN, daily)) return -1;
    if (has_x && check_var(ctx, TWX_TMAX, daily)) return -1;
    if (has_n && !g->lst_night) return fail(ctx, "twx_interp_grid: lst_night missing (Tmin predictor)");
    if (has_x && !g->lst_day) return fail(ctx, "twx_interp_grid: lst_day missing (Tmax predictor)");
    if (daily && has_n && has_x && (!o->daily_tmin || !o->daily_tmax))
        return fail(ctx, "twx_interp_grid: with both variables and daily output both daily buffers are needed (fixer)");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    const int ts = ctx->p.tile_cells;
    const int Y = g->Y, X = g->X;
    const int ntx = (X + ts - 1) / ts;
    int64_t batch = ctx->p.batch_cells > 0 ? ctx->p.batch_cells : (daily ? TWX_DAILY_BATCH : 131072);
    // both variables on 8x8 tiles: the daily sums walk per-(tile, month) tables of observation rows staged in LDS
    // (k_daily_tile), and the hat rows are produced in table-row order; the gather flags switch the tables off
    const bool use_table = daily && has_n && has_x && ts * ts <= 64 && !(ctx->p.flags & (TWX_FLAG_OBS_ADDR64 | TWX_FLAG_DAILY_GATHER));
    int band = (int)std::max<int64_t>(ts, batch / X / ts * ts);
    ctx->ev_used = 0; ctx->t_cells = 0;
    // outputs start at the netCDF fill values the reference's worker pre-fills with (step25:68-88): failed and
    // masked cells are never written afterwards
    {
        const size_t yx = (size_t)g->Y * g->X;
        uint32_t f4bits; const float f4 = TWX_FILL_F4; std::memcpy(&f4bits, &f4, 4);
        for (float *pf : {has_n ? o->norm_tmin : nullptr, has_n ? o->se_tmin : nullptr, has_x ? o->norm_tmax : nullptr,
                          has_x ? o->se_tmax : nullptr})
            if (pf) HIPCHK(hipMemsetD32Async(pf, (int)f4bits, yx * 12, stream));
        for (int16_t *pd : {has_n ? o->daily_tmin : nullptr, has_x ? o->daily_tmax : nullptr})
            if (pd) HIPCHK(hipMemsetD16Async(pd, (unsigned short)TWX_FILL_I2, yx * (size_t)ctx->ndays, stream));
        if (o->ninvalid) HIPCHK(hipMemsetD32Async(o->ninvalid, (int)TWX_FILL_I4, yx, stream));
    }
    HIPCHK(ctx->stats.ensure(64));
    HIPCHK(hipMemsetAsync(ctx->stats.p, 0, 64, stream));
    HIPCHK(hipEventRecord(ctx->ev_total_a, stream));
    ctx->have_total = true;
    for (int r0 = 0; r0 < Y; r0 += band) {
        const int r1 = std::min(Y, r0 + band);
        const int64_t cell0 = (int64_t)r0 * X, ncell = (int64_t)(r1 - r0) * X;
        const int64_t tile0 = (int64_t)(r0 / ts) * ntx, ntile = (int64_t)((r1 - r0 + ts - 1) / ts) * ntx;
        CellSrc src[2];
        for (int v = 0; v < 2; ++v) {
            if (!(v == 0 ? has_n : has_x)) continue;
            CellSrc &s = src[v];
            s = CellSrc{};
            s.mode = 0; s.Y = Y; s.X = X; s.ts = ts; s.ntx = ntx;
            s.mask = g->mask; s.lat = g->lat; s.lon = g->lon; s.elev = g->elev; s.tdi = g->tdi;
            s.lst = v == 0 ? g->lst_night : g->lst_day;
            s.do_krig = 1; s.do_anom = daily ? 1 : 0; s.do_vario = 1;
            if (run_select_uk(ctx, v, s, cell0, ncell, tile0, ntile, pick_ksel(ctx, v, 0), daily, stream)) return -1;
        }
        // (a batch that had to be re-run with longer candidate lists than the per-tile LDS tables index takes the gather paths)
        const bool long_lists = (has_n && ctx->work[0].cmax > TWX_CAND_MAX) || (has_x && ctx->work[1].cmax > TWX_CAND_MAX);
        const bool use_table_b = use_table && !long_lists;
        for (int v = 0; v < 2; ++v) ctx->work[v].gw.use_table = use_table_b ? 1 : 0;
        if (daily) {
            // both variables' tables are numbered before any hat row is computed: a (tile, month) goes through the
            // tables only when BOTH unions fit (k_daily_tile takes both variables of a block or neither)
            for (int v = 0; v < 2; ++v)
                if ((v == 0 ? has_n : has_x) && run_gwr_prep(ctx, v, src[v], stream)) return -1;
            for (int v = 0; v < 2; ++v) {
                if (!(v == 0 ? has_n : has_x)) continue;
                ctx->work[v].gw.nurow2 = use_table_b ? ctx->work[1 - v].gw.nurow : nullptr;
                if (run_gwr(ctx, v, src[v], nullptr, stream, false)) return -1;
            }
        }
        const CellSrc &s0 = has_n ? src[0] : src[1];
        hipLaunchKernelGGL(k_finalize_grid, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, s0,
                           ctx->work[0].ws, ctx->work[1].ws, (int)has_n, (int)has_x,
                           (daily && has_n) ? ctx->work[0].gw.gstat : nullptr,
                           (daily && has_x) ? ctx->work[1].gw.gstat : nullptr, *o, 1, (const int32_t *)nullptr);
        ctx->t_cells += ncell;
        if (daily) {
            int maxd = 0;
            for (int m = 0; m < 12; ++m) maxd = std::max(maxd, ctx->da.moff[m + 1] - ctx->da.moff[m]);
            const int nblk = (maxd + 63) / 64;
            HIPCHK(ctx->flags.ensure((size_t)ncell * 4 + 256));
            HIPCHK(ctx->flag_list.ensure((size_t)ncell * 4 + 256));
            HIPCHK(hipMemsetAsync(ctx->flags.p, 0, (size_t)ncell * 4 + 256, stream));
            HIPCHK(ctx->inv_cnt.ensure((size_t)ncell * 4));
            HIPCHK(ctx->inv_day.ensure((size_t)ncell * TWX_INV_CAP * 4));
            HIPCHK(hipMemsetAsync(ctx->inv_cnt.p, 0, (size_t)ncell * 4, stream));
            int32_t *d_icnt = ctx->inv_cnt.as<int32_t>(), *d_iday = ctx->inv_day.as<int32_t>();
            int32_t *d_flag = ctx->flags.as<int32_t>();
            int32_t *d_count = d_flag + ncell;
            // tie guard (see run_tie_guard): on whenever the fixer runs on fast-build normals
            const bool guard = has_n && has_x && !(ctx->p.flags & (TWX_FLAG_UK_F64_ALL | TWX_FLAG_UK_FAST_ONLY | TWX_FLAG_NO_TIE_GUARD));
            int32_t *d_tie = nullptr;
            if (guard) {
                HIPCHK(ctx->tie.ensure((size_t)ncell * 8 + 256));
                HIPCHK(hipMemsetAsync(ctx->tie.p, 0, (size_t)ncell * 8 + 256, stream));
                d_tie = ctx->tie.as<int32_t>();
            }
            {
                EvScope ev(ctx, stream, EV_DAILY);
                for (int v = 0; v < 2; ++v)
                    if (v == 0 ? has_n : has_x) {
                        const SelWs &wsv = ctx->work[v].ws;
                        hipLaunchKernelGGL(k_row_offsets, dim3((unsigned)((wsv.ncell * wsv.ksel + 255) / 256)), dim3(256), 0,
                                           stream, wsv, ctx->work[v].gw, (int)ctx->ndays);
                    }
                const int addr64 = (ctx->p.flags & TWX_FLAG_OBS_ADDR64) ? 1 : 0;
                if (has_n && has_x && ts * ts <= 64) {
                    // both variables: the tile kernel (observation rows of a tile-month staged in LDS)
                    int32_t *d_okc = ctx->flag_list.as<int32_t>();      // free until k_compact_flags (which runs after)
                    hipLaunchKernelGGL(k_daily_ok, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, ctx->work[0].ws,
                                       ctx->work[1].ws, ctx->work[0].gw, ctx->work[1].gw, d_okc);
                    const int gather = (addr64 || (ctx->p.flags & TWX_FLAG_DAILY_GATHER) || long_lists) ? 1 : 0;
                    DtArgs da{};
                    for (int v = 0; v < 2; ++v) {
                        DtVar &dv = v == 0 ? da.n : da.x;
                        const Work &wk = ctx->work[v];
                        dv.obs = ctx->var[v].dev.obs; dv.ka = wk.ws.ka; dv.z = wk.gw.z; dv.zc = wk.gw.zc; dv.zd = wk.gw.zd;
                        dv.urow = wk.gw.urow; dv.nurow = wk.gw.nurow;
                    }
                    da.okc = d_okc; da.mm2chron = ctx->da.mm2chron; da.out_n = o->daily_tmin; da.out_x = o->daily_tmax;
                    da.flag = d_flag; da.inv_cnt = d_icnt; da.inv_day = d_iday; da.tie = d_tie; da.cell0 = cell0; da.ncell = ncell; da.tile0 = tile0; da.ntile = ntile;
                    da.Y = Y; da.X = X; da.ts = ts; da.ntx = ntx; da.ndays = (int)ctx->ndays; da.nblk_max = nblk; da.gather = gather;
                    for (int m = 0; m < 13; ++m) da.moff[m] = ctx->da.moff[m];
                    if (!gather)
                        // units of (8 tiles, month), rounded up to a multiple of the 8 XCDs, x 8 tiles x nblk blocks each
                        hipLaunchKernelGGL(k_daily_tile, dim3((unsigned)((((ntile + 7) / 8) * 12 + 7) / 8 * 8 * 8 * nblk)), dim3(64 * TWX_DT_WAVES), 0, stream, da);
                    hipLaunchKernelGGL(k_daily_tile_gather, dim3((unsigned)ntile, (unsigned)(12 * nblk)), dim3(256), 0, stream,
                                       ctx->var[0].dev, ctx->var[1].dev, s0, ctx->work[0].ws, ctx->work[1].ws, ctx->work[0].gw,
                                       ctx->work[1].gw, ctx->da, *o, d_flag, d_icnt, d_iday, d_tie, d_okc, nblk, addr64, gather);
                } else {
                    hipLaunchKernelGGL(k_daily_grid, dim3((unsigned)((ncell + 63) / 64), (unsigned)(12 * nblk)), dim3(256), 0,
                                       stream, ctx->var[0].dev, ctx->var[1].dev, s0, ctx->work[0].ws, ctx->work[1].ws,
                                       ctx->work[0].gw, ctx->work[1].gw, (int)has_n, (int)has_x, ctx->da, *o, d_flag, d_icnt, d_iday, d_tie, nblk, addr64);
                }
            }
            if (guard && run_tie_guard(ctx, src, ncell, *o, stream)) return -1;
            if (has_n && has_x) {
                hipLaunchKernelGGL(k_compact_flags, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, d_flag,
                                   ncell, ctx->flag_list.as<int32_t>(), d_count);
                // the number of flagged cells stays on the device: a fixed grid strides over the list
                const int nb = (int)std::min<int64_t>(ncell, 512);
                HIPCHK(ctx->fix_scratch.ensure((size_t)nb * 2 * ctx->ndays * 8));
                HIPCHK(ctx->fix_lists.ensure((size_t)nb * ctx->ndays * 4));
                FixArgs fa{};
                fa.cells = ctx->flag_list.as<int32_t>(); fa.ncells_dev = d_count; fa.inv_cnt = d_icnt; fa.inv_day = d_iday; fa.tie = d_tie;
                fa.scratch = ctx->fix_scratch.as<double>(); fa.lists = ctx->fix_lists.as<int32_t>();
                fa.sparse_ok = (!(ctx->p.flags & TWX_FLAG_FIX_FULL) &&
                                fix_sparse_usable(ctx->var[0].dev.ymsum, ctx->var[1].dev.ymsum, ctx->da.norm_ny, ctx->da.tail)) ? 1 : 0;
                EvScope ev(ctx, stream, EV_FIX);
                // cells with at most TWX_INV_CAP invalid days: fixed from those days' windows; the others (all of them when
                // the sparse path is not usable: one predicate, evaluated here): whole series
                if (fa.sparse_ok)
                    hipLaunchKernelGGL(k_fix_sparse, dim3(nb), dim3(256), 0, stream, ctx->var[0].dev, ctx->var[1].dev, s0, ctx->work[0].ws,
                                       ctx->work[1].ws, ctx->work[0].gw, ctx->work[1].gw, ctx->da, *o, fa);
                hipLaunchKernelGGL(k_fix_cells, dim3(nb), dim3(TWX_FIX_THREADS), 0, stream, ctx->var[0].dev, ctx->var[1].dev, s0,
                                   ctx->work[0].ws, ctx->work[1].ws, ctx->work[0].gw, ctx->work[1].gw, ctx->da, *o, fa);
            }
        }
    }
    HIPCHK(hipEventRecord(ctx->ev_total_b, stream));
    HIPCHK(hipGetLastError());
    return 0;
}

namespace {
// device images of a grid call's inputs / outputs carved out of two persistent buffers
struct GridDev {
    twx_grid g{};
    twx_grid_out o{};
    struct Item { void *host; void *dev; size_t bytes; };
    std::vector<Item> in_items, out_items;
};

size_t grid_in_bytes(int Y, int X) { return (size_t)Y * X * (1 + 4 + 4 + 24 * 4) + (size_t)(Y + X) * 8 + 16 * 256; }
size_t grid_out_bytes(int Y, int X, int64_t ndays, bool dn, bool dx)
{
    return (size_t)Y * X * (4 * 48 + 8) + ((dn ? 1 : 0) + (dx ? 1 : 0)) * (size_t)Y * X * (size_t)ndays * 2 + 16 * 256;
}

// lay the device images out in din / dout; host pointers that are null stay null
void carve_grid(const twx_grid *g, const twx_grid_out *o, int vars, int64_t ndays, char *din, char *dout, GridDev &gd)
{
    const size_t yx = (size_t)g->Y * g->X, nd = (size_t)ndays;
    const bool has_n = vars & TWX_VAR_TMIN_BIT, has_x = vars & TWX_VAR_TMAX_BIT;
    auto in = [&](const void *h, size_t bytes) -> void * {
        if (!h) return nullptr;
        void *d = carve<char>(din, bytes);
        gd.in_items.push_back({const_cast<void *>(h), d, bytes});
        return d;
    };
    auto out = [&](void *h, size_t bytes) -> void * {
        if (!h) return nullptr;
        void *d = carve<char>(dout, bytes);
        gd.out_items.push_back({h, d, bytes});
        return d;
    };
    gd.g = *g;
    gd.g.mask = (const uint8_t *)in(g->mask, yx);
    gd.g.lat = (const double *)in(g->lat, (size_t)g->Y * 8);
    gd.g.lon = (const double *)in(g->lon, (size_t)g->X * 8);
    gd.g.elev = (const float *)in(g->elev, yx * 4);
    gd.g.tdi = (const float *)in(g->tdi, yx * 4);
    gd.g.climdiv = nullptr;
    gd.g.lst_night = has_n ? (const float *)in(g->lst_night, yx * 48) : nullptr;
    gd.g.lst_day = has_x ? (const float *)in(g->lst_day, yx * 48) : nullptr;
    gd.o.norm_tmin = (float *)out(o->norm_tmin, yx * 48); gd.o.se_tmin = (float *)out(o->se_tmin, yx * 48);
    gd.o.norm_tmax = (float *)out(o->norm_tmax, yx * 48); gd.o.se_tmax = (float *)out(o->se_tmax, yx * 48);
    gd.o.daily_tmin = (int16_t *)out(o->daily_tmin, yx * nd * 2); gd.o.daily_tmax = (int16_t *)out(o->daily_tmax, yx * nd * 2);
    gd.o.ninvalid = (int32_t *)out(o->ninvalid, yx * 4); gd.o.status = (int32_t *)out(o->status, yx * 4);
}
}  // namespace

int twx_interp_grid(twx_ctx *ctx, const twx_grid *g, const twx_grid_out *o, int vars)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (!g || !o || g->Y <= 0 || g->X <= 0) return fail(ctx, "twx_interp_grid: bad grid");
    HIPCHK(hipSetDevice(ctx->device));
    // persistent device images (grown on demand, released by twx_destroy): no allocation per call, and the outputs
    // are filled on the device -- nothing but the 61 B / cell of predictors goes up
    HIPCHK(ctx->grid_in.ensure(grid_in_bytes(g->Y, g->X)));
    HIPCHK(ctx->grid_out.ensure(grid_out_bytes(g->Y, g->X, ctx->ndays, o->daily_tmin != nullptr, o->daily_tmax != nullptr)));
    GridDev gd;
    carve_grid(g, o, vars, ctx->ndays, ctx->grid_in.as<char>(), ctx->grid_out.as<char>(), gd);
    for (auto &it : gd.in_items) HIPCHK(hipMemcpyAsync(it.dev, it.host, it.bytes, hipMemcpyHostToDevice, nullptr));
    if (twx_interp_grid_dev(ctx, &gd.g, &gd.o, vars, nullptr)) return -1;
    for (auto &it : gd.out_items) HIPCHK(hipMemcpyAsync(it.host, it.dev, it.bytes, hipMemcpyDeviceToHost, nullptr));
    if (hipStreamSynchronize(nullptr) != hipSuccess) return fail(ctx, "twx_interp_grid: kernel execution failed", hipGetLastError());
    return 0;
}

// ---- streamed tiles (step25:177-185, tiling.py:488-537: every chunk is written as soon as it is finished) ---------
struct twx_stream {
    twx_ctx *ctx = nullptr;
    int device = 0;                               // (kept here: destroy must not reach through a context that may be gone)
    int Y = 0, X = 0, vars = 0, daily = 0, nslots = 0;
    size_t in_bytes = 0, out_bytes = 0;
    hipStream_t s_comp = nullptr, s_copy = nullptr;
    DevBuf din[2], dout[2];                       // double-buffered device images
    std::vector<char *> hin, hout;                // pinned host staging per slot
    std::vector<GridDev> views;                   // per slot: where its outputs live in hout[slot]
    std::vector<hipEvent_t> ev_start, ev_comp, ev_copy0, ev_done;   // per slot
    hipEvent_t ev_free[2] = {nullptr, nullptr};   // device set d has been copied out
    bool used[2] = {false, false};
    int64_t nsub = 0;
    // twx_stream_deflate: the daily outputs leave the device as zlib streams per chunk (twx_deflate.h)
    int df_cy = 0, df_cx = 0, df_nchunk = 0, df_nvar = 0, df_nseg = 0;
    int64_t df_N = 0, df_slot = 0;                // elements per chunk, bytes of a chunk's device slot
    DevBuf df_out[2], df_small[2];                // per device set: chunk slots [var][chunk]; block sizes / offsets / Adler sums / chunk sizes
    std::vector<char *> df_host;                  // per slot, pinned: the tile's streams, one after the other
    std::vector<int64_t *> df_sizes;              // per slot, pinned: [var][chunk] bytes of every stream
    std::vector<std::vector<int64_t>> df_off;     // per slot: [var][nchunk + 1] offsets inside the variable's part of df_host[slot]
    std::vector<GridDev> df_dev;                  // per slot: the device images its deferred copy-out reads
    std::vector<int> df_set;                      // per slot: device set of its tile
    int df_pending[2] = {-1, -1};                 // per device set: slot whose copy-out has not been enqueued yet
    hipStream_t s_defl = nullptr;                 // the deflate kernels' own stream: tile t is deflated while tile t + 1 is interpolated
    std::vector<hipEvent_t> ev_k;                 // per slot: the tile's interpolation kernels are done (on s_comp)
};

void twx_stream_destroy(twx_stream *st);

int twx_stream_create(twx_ctx *ctx, int Y, int X, int vars, int daily, int nslots, twx_stream **out)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (!out || Y <= 0 || X <= 0 || nslots < 1 || nslots > 16 || !(vars & 3)) return fail(ctx, "twx_stream_create: bad arguments");
    if (daily && ctx->ndays <= 0) return fail(ctx, "twx_stream_create: daily output needs the day axis (twx_set_days)");
    HIPCHK(hipSetDevice(ctx->device));
    twx_stream *st = new twx_stream();
    st->ctx = ctx; st->device = ctx->device; st->Y = Y; st->X = X; st->vars = vars; st->daily = daily; st->nslots = nslots;
    const bool has_n = vars & TWX_VAR_TMIN_BIT, has_x = vars & TWX_VAR_TMAX_BIT;
    st->in_bytes = grid_in_bytes(Y, X);
    st->out_bytes = grid_out_bytes(Y, X, ctx->ndays, daily && has_n, daily && has_x);
    bool ok = hipStreamCreateWithFlags(&st->s_comp, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&st->s_copy, hipStreamNonBlocking) == hipSuccess;
    for (int d = 0; ok && d < 2; ++d)
        ok = st->din[d].ensure(st->in_bytes) == hipSuccess && st->dout[d].ensure(st->out_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&st->ev_free[d], hipEventDisableTiming) == hipSuccess;
    st->hin.assign(nslots, nullptr); st->hout.assign(nslots, nullptr);
    st->views.resize(nslots); st->ev_start.assign(nslots, nullptr); st->ev_comp.assign(nslots, nullptr); st->ev_copy0.assign(nslots, nullptr); st->ev_done.assign(nslots, nullptr);
    for (int i = 0; ok && i < nslots; ++i)
        ok = hipHostMalloc((void **)&st->hin[i], st->in_bytes, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void **)&st->hout[i], st->out_bytes, hipHostMallocDefault) == hipSuccess &&
             hipEventCreate(&st->ev_start[i]) == hipSuccess && hipEventCreate(&st->ev_comp[i]) == hipSuccess &&
             hipEventCreate(&st->ev_copy0[i]) == hipSuccess && hipEventCreate(&st->ev_done[i]) == hipSuccess;
    ctx->streams.push_back(st);
    if (!ok) { twx_stream_destroy(st); return fail(ctx, "twx_stream_create: allocation failed (device images / pinned host staging)"); }
    *out = st;
    return 0;
}

void twx_stream_destroy(twx_stream *st)
{
    if (!st) return;
    (void)hipSetDevice(st->device);
    if (st->ctx) {
        auto &v = st->ctx->streams;
        v.erase(std::remove(v.begin(), v.end(), st), v.end());
    }
    if (st->s_comp) (void)hipStreamSynchronize(st->s_comp);
    if (st->s_copy) (void)hipStreamSynchronize(st->s_copy);
    if (st->s_defl) { (void)hipStreamSynchronize(st->s_defl); (void)hipStreamDestroy(st->s_defl); }
    for (hipEvent_t e : st->ev_k) if (e) (void)hipEventDestroy(e);
    for (int d = 0; d < 2; ++d) { st->din[d].release(); st->dout[d].release(); if (st->ev_free[d]) (void)hipEventDestroy(st->ev_free[d]); }
    for (char *p : st->hin) if (p) (void)hipHostFree(p);
    for (char *p : st->hout) if (p) (void)hipHostFree(p);
    for (char *p : st->df_host) if (p) (void)hipHostFree(p);
    for (int64_t *p : st->df_sizes) if (p) (void)hipHostFree(p);
    for (int d = 0; d < 2; ++d) { st->df_out[d].release(); st->df_small[d].release(); }
    for (auto *v : {&st->ev_start, &st->ev_comp, &st->ev_copy0, &st->ev_done})
        for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
    if (st->s_comp) (void)hipStreamDestroy(st->s_comp);
    if (st->s_copy) (void)hipStreamDestroy(st->s_copy);
    delete st;
}

namespace {
// per variable: block sizes + offsets (u32 each), Adler sums (4 u32), piece bits (256 u16) per segment; chunk sizes
size_t df_small_bytes(const twx_stream *st)
{
    return (size_t)st->df_nchunk * st->df_nseg * (24 + 2 * TWX_DF_THREADS) + TWX_DF_NSYM * 4 + sizeof(DfTable) + 7 * 256;
}

// kernels of twx_deflate.h for the tile in device set d (after its daily values are final), sizes -> the slot's pinned table
int df_launch(twx_stream *st, int d, const GridDev &dev, int slot)
{
    twx_ctx *ctx = st->ctx;
    EvScope ev(ctx, st->s_defl, EV_DEFLATE);
    const size_t per_var = (size_t)st->df_nchunk * st->df_nseg;
    char *small = st->df_small[d].as<char>();
    DfArgs args[2];
    int nv = 0;
    for (int var = 0; var < 2; ++var) {
        const int16_t *daily = var == 0 ? dev.o.daily_tmin : dev.o.daily_tmax;
        if (!daily) continue;
        DfArgs a{};
        a.daily = reinterpret_cast<const uint16_t *>(daily);
        a.out = st->df_out[d].as<uint8_t>() + (size_t)nv * st->df_nchunk * st->df_slot;
        char *cur = small + (size_t)nv * df_small_bytes(st);
        a.seg_bytes = carve<uint32_t>(cur, per_var);
        a.seg_off = carve<uint32_t>(cur, per_var);
        a.adl = carve<uint32_t>(cur, per_var * 4);
        a.piece_bits = carve<uint16_t>(cur, per_var * TWX_DF_THREADS);
        a.hist = carve<uint32_t>(cur, TWX_DF_NSYM);
        a.table = carve<DfTable>(cur, 1);
        // the chunk sizes go straight into the slot's pinned table (mapped host memory): a copy command on this stream would queue
        // behind the bulk copy-out of the tile before on the DMA engine and hold this tile's "kernels done" event back
        a.chunk_bytes = st->df_sizes[slot] + (size_t)nv * st->df_nchunk;
        a.N = st->df_N; a.slot_bytes = st->df_slot; a.lo_bytes = df_lo_bytes(st->df_N);
        a.Y = st->Y; a.X = st->X; a.cy = st->df_cy; a.cx = st->df_cx; a.ncx = st->X / st->df_cx; a.nseg = st->df_nseg;
        args[nv++] = a;
    }
    const dim3 grid((unsigned)st->df_nchunk, (unsigned)st->df_nseg);     // (chunks fastest: twx_deflate.h)
    const dim3 sgrid((unsigned)st->df_nchunk, (unsigned)((st->df_nseg + TWX_DF_SAMPLE - 1) / TWX_DF_SAMPLE));
    const dim3 th(TWX_DF_THREADS);
    const bool pairs = st->df_cx % 2 == 0 && st->X % 2 == 0;           // two neighbouring values per load
    // the variables' Huffman codes: token counts of every 16th segment -> code lengths, codes, the block header (one work-group
    // each, mostly one thread: both in ONE launch)
    for (int v = 0; v < nv; ++v) {
        HIPCHK(hipMemsetAsync(args[v].hist, 0, TWX_DF_NSYM * 4, st->s_defl));
        if (pairs) hipLaunchKernelGGL(k_deflate_hist<2>, sgrid, th, 0, st->s_defl, args[v]);
        else hipLaunchKernelGGL(k_deflate_hist<1>, sgrid, th, 0, st->s_defl, args[v]);
    }
    hipLaunchKernelGGL(k_deflate_table, dim3((unsigned)nv), th, 0, st->s_defl, args[0].hist, args[0].table, args[nv - 1].hist, args[nv - 1].table);
    for (int v = 0; v < nv; ++v) {
        if (pairs) hipLaunchKernelGGL(k_deflate_count<2>, grid, th, 0, st->s_defl, args[v]);
        else hipLaunchKernelGGL(k_deflate_count<1>, grid, th, 0, st->s_defl, args[v]);
        hipLaunchKernelGGL(k_deflate_scan, dim3((unsigned)st->df_nchunk), th, 0, st->s_defl, args[v]);
        if (pairs) hipLaunchKernelGGL(k_deflate_emit<2>, grid, th, 0, st->s_defl, args[v]);
        else hipLaunchKernelGGL(k_deflate_emit<1>, grid, th, 0, st->s_defl, args[v]);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// the copy-out of the slot's tile: small outputs into the slot's pinned block, every chunk stream -- its size is known now --
// into the slot's stream block, one after the other
int df_enqueue_copy(twx_stream *st, int slot)
{
    twx_ctx *ctx = st->ctx;
    const int d = st->df_set[slot];
    GridDev &dev = st->df_dev[slot];
    HIPCHK(hipEventSynchronize(st->ev_comp[slot]));                 // kernels done, sizes in df_sizes[slot]
    HIPCHK(hipStreamWaitEvent(st->s_copy, st->ev_comp[slot], 0));
    HIPCHK(hipEventRecord(st->ev_copy0[slot], st->s_copy));
    GridDev &view = st->views[slot];
    view = GridDev{};
    char *hout = st->hout[slot];
    twx_grid_out hv{};
    void **fields_dev[6] = {(void **)&dev.o.norm_tmin, (void **)&dev.o.se_tmin, (void **)&dev.o.norm_tmax, (void **)&dev.o.se_tmax,
                            (void **)&dev.o.ninvalid, (void **)&dev.o.status};
    void **fields_host[6] = {(void **)&hv.norm_tmin, (void **)&hv.se_tmin, (void **)&hv.norm_tmax, (void **)&hv.se_tmax,
                             (void **)&hv.ninvalid, (void **)&hv.status};
    const size_t yx = (size_t)st->Y * st->X;
    const size_t fbytes[6] = {yx * 48, yx * 48, yx * 48, yx * 48, yx * 4, yx * 4};
    for (int f = 0; f < 6; ++f) {
        if (!*fields_dev[f]) continue;
        *fields_host[f] = hout;
        HIPCHK(hipMemcpyAsync(hout, *fields_dev[f], fbytes[f], hipMemcpyDeviceToHost, st->s_copy));
        hout += (fbytes[f] + 255) / 256 * 256;
    }
    view.o = hv;
    std::vector<int64_t> &off = st->df_off[slot];
    off.assign((size_t)st->df_nvar * (st->df_nchunk + 1), 0);
    char *dst = st->df_host[slot];
    for (int v = 0; v < st->df_nvar; ++v) {
        int64_t o = 0;
        for (int c = 0; c < st->df_nchunk; ++c) {
            const int64_t n = st->df_sizes[slot][(size_t)v * st->df_nchunk + c];
            if (n <= 0 || n > st->df_slot) return fail(ctx, "twx_stream: a deflated chunk has an impossible size (device fault?)");
            off[(size_t)v * (st->df_nchunk + 1) + c] = o;
            HIPCHK(hipMemcpyAsync(dst + o, st->df_out[d].as<char>() + ((size_t)v * st->df_nchunk + c) * st->df_slot, (size_t)n,
                                  hipMemcpyDeviceToHost, st->s_copy));
            o += n;
        }
        off[(size_t)v * (st->df_nchunk + 1) + st->df_nchunk] = o;
        dst += (size_t)st->df_nchunk * st->df_slot;                  // (every variable's part can hold the longest streams)
    }
    HIPCHK(hipEventRecord(st->ev_free[d], st->s_copy));
    HIPCHK(hipEventRecord(st->ev_done[slot], st->s_copy));
    st->df_pending[d] = -1;
    return 0;
}
}  // namespace

int twx_stream_submit(twx_stream *st, int slot, const twx_grid *g)
{
    if (!st) return -1;
    twx_ctx *ctx = st->ctx;
    ctx->err.clear();
    if (!g || slot < 0 || slot >= st->nslots || g->Y != st->Y || g->X != st->X) return fail(ctx, "twx_stream_submit: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const int d = (int)(st->nsub & 1);
    const bool has_n = st->vars & TWX_VAR_TMIN_BIT, has_x = st->vars & TWX_VAR_TMAX_BIT;
    // output layout of this slot inside its pinned block (pointers only mark which outputs exist)
    twx_grid_out want{};
    char *const mark = st->hout[slot];
    if (has_n) { want.norm_tmin = (float *)mark; want.se_tmin = (float *)mark; if (st->daily) want.daily_tmin = (int16_t *)mark; }
    if (has_x) { want.norm_tmax = (float *)mark; want.se_tmax = (float *)mark; if (st->daily) want.daily_tmax = (int16_t *)mark; }
    want.ninvalid = (int32_t *)mark; want.status = (int32_t *)mark;
    GridDev dev;
    carve_grid(g, &want, st->vars, ctx->ndays, st->din[d].as<char>(), st->dout[d].as<char>(), dev);
    // stage the predictors in pinned memory (61 B / cell), then everything else is asynchronous
    char *hin = st->hin[slot];
    if (st->df_cy) {
        // Deflated outputs: the copy-out of a tile nobody has waited for yet starts NOW (df_enqueue_copy waits for its kernels) --
        // the tile whose slot / device set this call reuses, and also the tile in the other device set: a pipelined caller gets
        // here right after the copy-out of the tile before that one, and this call's own read-backs return only ~20 ms into this
        // tile's kernels; enqueued after them, the copy engine idled that long per tile (configs[3]: 30.2 s end to end, 26.8 s with
        // a non-blocking test of the event here, which often lost the race by a hair).  Where the kernels are the longer stage
        // this wait costs the host's launch latency per tile, nothing more: the next kernels queue behind them anyway.
        for (int dd = 0; dd < 2; ++dd)
            if (st->df_pending[dd] >= 0 && df_enqueue_copy(st, st->df_pending[dd])) return -1;
    }
    HIPCHK(hipEventSynchronize(st->ev_done[slot]));                 // the slot's previous tile has left the device
    for (auto &it : dev.in_items) {
        std::memcpy(hin, it.host, it.bytes);
        if (st->used[d]) HIPCHK(hipStreamWaitEvent(st->s_comp, st->ev_free[d], 0));   // device set d copied out
        HIPCHK(hipMemcpyAsync(it.dev, hin, it.bytes, hipMemcpyHostToDevice, st->s_comp));
        hin += (it.bytes + 255) / 256 * 256;
    }
    HIPCHK(hipEventRecord(st->ev_start[slot], st->s_comp));
    if (twx_interp_grid_dev(ctx, &dev.g, &dev.o, st->vars, st->s_comp)) return -1;
    if (st->df_cy) {
        // the chunk streams of both variables, and their sizes to the slot's pinned table; the copy-out itself is enqueued by
        // twx_stream_wait_deflated (it needs the sizes) -- nothing goes on the copy stream here, so that the copy-out of tile t
        // is never queued behind a wait for the kernels of tile t + 1
        // (on the deflate kernels' own stream: they wait on memory, the next tile's kriging kernels on the fp64 pipes -- side by
        // side the tile's device time is again what it was without them, and precision="auto" keeps the fp64 build)
        HIPCHK(hipEventRecord(st->ev_k[slot], st->s_comp));
        HIPCHK(hipStreamWaitEvent(st->s_defl, st->ev_k[slot], 0));
        if (df_launch(st, d, dev, slot)) return -1;
        HIPCHK(hipEventRecord(st->ev_comp[slot], st->s_defl));
        st->df_dev[slot] = dev;
        st->df_set[slot] = d;
        st->df_pending[d] = slot;
        st->views[slot] = GridDev{};
        st->used[d] = true;
        st->nsub++;
        return 0;
    }
    HIPCHK(hipEventRecord(st->ev_comp[slot], st->s_comp));
    // copy stream: device set d -> the slot's pinned block, overlapping the next tile's kernels
    HIPCHK(hipStreamWaitEvent(st->s_copy, st->ev_comp[slot], 0));
    HIPCHK(hipEventRecord(st->ev_copy0[slot], st->s_copy));
    GridDev &view = st->views[slot];
    view = GridDev{};
    char *hout = st->hout[slot];
    twx_grid_out hv{};
    void **fields_dev[8] = {(void **)&dev.o.norm_tmin, (void **)&dev.o.se_tmin, (void **)&dev.o.norm_tmax, (void **)&dev.o.se_tmax,
                            (void **)&dev.o.daily_tmin, (void **)&dev.o.daily_tmax, (void **)&dev.o.ninvalid, (void **)&dev.o.status};
    void **fields_host[8] = {(void **)&hv.norm_tmin, (void **)&hv.se_tmin, (void **)&hv.norm_tmax, (void **)&hv.se_tmax,
                             (void **)&hv.daily_tmin, (void **)&hv.daily_tmax, (void **)&hv.ninvalid, (void **)&hv.status};
    size_t k = 0;
    for (int f = 0; f < 8; ++f) {
        if (!*fields_dev[f]) continue;
        const size_t bytes = dev.out_items[k++].bytes;
        *fields_host[f] = hout;
        HIPCHK(hipMemcpyAsync(hout, *fields_dev[f], bytes, hipMemcpyDeviceToHost, st->s_copy));
        hout += (bytes + 255) / 256 * 256;
    }
    view.o = hv;
    HIPCHK(hipEventRecord(st->ev_free[d], st->s_copy));
    HIPCHK(hipEventRecord(st->ev_done[slot], st->s_copy));
    st->used[d] = true;
    st->nsub++;
    return 0;
}

int twx_stream_wait(twx_stream *st, int slot, twx_grid_out *views, float *device_ms)
{
    if (!st) return -1;
    twx_ctx *ctx = st->ctx;
    ctx->err.clear();
    if (slot < 0 || slot >= st->nslots || !views) return fail(ctx, "twx_stream_wait: bad arguments");
    if (st->df_cy) return fail(ctx, "twx_stream_wait: this stream delivers deflated chunks (twx_stream_deflate): call twx_stream_wait_deflated");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipEventSynchronize(st->ev_done[slot]));
    *views = st->views[slot].o;
    if (device_ms) {
        *device_ms = 0.f;
        if (st->views[slot].o.status) HIPCHK(hipEventElapsedTime(device_ms, st->ev_start[slot], st->ev_comp[slot]));
    }
    return 0;
}

int twx_stream_deflate(twx_stream *st, int chunk_y, int chunk_x)
{
    if (!st) return -1;
    twx_ctx *ctx = st->ctx;
    ctx->err.clear();
    if (!st->daily) return fail(ctx, "twx_stream_deflate: the stream has no daily outputs");
    if (st->nsub || st->df_cy) return fail(ctx, "twx_stream_deflate: call it once, before the first twx_stream_submit");
    if (chunk_y <= 0 || chunk_x <= 0 || st->Y % chunk_y || st->X % chunk_x)
        return fail(ctx, "twx_stream_deflate: the chunk shape must divide the tile's");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t N = ctx->ndays * (int64_t)chunk_y * chunk_x;
    if (df_slot_bytes(N) >= (int64_t)1 << 32) return fail(ctx, "twx_stream_deflate: a chunk of more than 2^31 values");
    st->df_N = N; st->df_slot = df_slot_bytes(N); st->df_nseg = df_nseg(N);
    st->df_nchunk = (st->Y / chunk_y) * (st->X / chunk_x);
    st->df_nvar = ((st->vars & TWX_VAR_TMIN_BIT) ? 1 : 0) + ((st->vars & TWX_VAR_TMAX_BIT) ? 1 : 0);
    const size_t per_var = (size_t)st->df_nchunk * st->df_nseg;
    const size_t out_bytes = (size_t)st->df_nvar * st->df_nchunk * st->df_slot;
    bool ok = true;
    for (int d = 0; ok && d < 2; ++d)
        ok = st->df_out[d].ensure(out_bytes) == hipSuccess &&
             st->df_small[d].ensure((size_t)st->df_nvar * df_small_bytes(st)) == hipSuccess;
    // the pinned blocks no longer hold the daily arrays: allocate them again without, and the stream blocks beside them
    const size_t small_bytes = grid_out_bytes(st->Y, st->X, ctx->ndays, false, false);
    st->df_host.assign(st->nslots, nullptr); st->df_sizes.assign(st->nslots, nullptr);
    st->df_off.resize(st->nslots); st->df_dev.resize(st->nslots); st->df_set.assign(st->nslots, 0);
    for (int i = 0; ok && i < st->nslots; ++i) {
        if (st->hout[i]) (void)hipHostFree(st->hout[i]);
        st->hout[i] = nullptr;
        ok = hipHostMalloc((void **)&st->hout[i], small_bytes, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void **)&st->df_host[i], out_bytes, hipHostMallocDefault) == hipSuccess &&
             hipHostMalloc((void **)&st->df_sizes[i], (size_t)st->df_nvar * st->df_nchunk * 8, hipHostMallocDefault) == hipSuccess;
    }
    ok = ok && hipStreamCreateWithFlags(&st->s_defl, hipStreamNonBlocking) == hipSuccess;
    st->ev_k.assign(st->nslots, nullptr);
    for (int i = 0; ok && i < st->nslots; ++i) ok = hipEventCreate(&st->ev_k[i]) == hipSuccess;
    if (!ok) return fail(ctx, "twx_stream_deflate: allocation failed (chunk slots / pinned stream blocks)");
    st->df_cy = chunk_y; st->df_cx = chunk_x;
    return 0;
}

int twx_stream_wait_deflated(twx_stream *st, int slot, twx_grid_out *views, twx_deflated *streams, float *device_ms)
{
    if (!st) return -1;
    twx_ctx *ctx = st->ctx;
    ctx->err.clear();
    if (slot < 0 || slot >= st->nslots || !views || !streams) return fail(ctx, "twx_stream_wait_deflated: bad arguments");
    if (!st->df_cy) return fail(ctx, "twx_stream_wait_deflated: twx_stream_deflate was not called on this stream");
    HIPCHK(hipSetDevice(ctx->device));
    for (int d = 0; d < 2; ++d)
        if (st->df_pending[d] == slot && df_enqueue_copy(st, slot)) return -1;
    HIPCHK(hipEventSynchronize(st->ev_done[slot]));
    *views = st->views[slot].o;
    twx_deflated r{};
    r.nchunks = st->df_nchunk; r.chunk_y = st->df_cy; r.chunk_x = st->df_cx;
    if (st->views[slot].o.status) {
        int v = 0;
        for (int var = 0; var < 2; ++var) {
            if (!(st->vars & (var == 0 ? TWX_VAR_TMIN_BIT : TWX_VAR_TMAX_BIT))) continue;
            r.data[var] = reinterpret_cast<const uint8_t *>(st->df_host[slot]) + (size_t)v * st->df_nchunk * st->df_slot;
            r.offset[var] = st->df_off[slot].data() + (size_t)v * (st->df_nchunk + 1);
            ++v;
        }
    }
    *streams = r;
    if (device_ms) {      // (the tile's interpolation kernels: its deflate kernels run beside the next tile's)
        *device_ms = 0.f;
        if (st->views[slot].o.status) HIPCHK(hipEventElapsedTime(device_ms, st->ev_start[slot], st->ev_k[slot]));
    }
    return 0;
}

int twx_stream_times(twx_stream *st, int slot, float *device_ms, float *copy_ms)
{
    if (!st) return -1;
    twx_ctx *ctx = st->ctx;
    ctx->err.clear();
    if (slot < 0 || slot >= st->nslots) return fail(ctx, "twx_stream_times: bad arguments");
    if (!st->views[slot].o.status) return fail(ctx, "twx_stream_times: nothing submitted in this slot");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipEventSynchronize(st->ev_done[slot]));
    if (device_ms) HIPCHK(hipEventElapsedTime(device_ms, st->ev_start[slot], st->df_cy ? st->ev_k[slot] : st->ev_comp[slot]));
    if (copy_ms) HIPCHK(hipEventElapsedTime(copy_ms, st->ev_copy0[slot], st->ev_done[slot]));
    return 0;
}

int twx_device_memory(twx_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes)
{
    if (!ctx) return -1;
    ctx->err.clear();
    HIPCHK(hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
}

int twx_get_timing(twx_ctx *ctx, twx_timing *t)
{
    if (!ctx || !t) return -1;
    HIPCHK(hipSetDevice(ctx->device));
    float acc[EV_NKIND] = {};
    for (size_t i = 0; i < ctx->ev_used; ++i) {
        HIPCHK(hipEventSynchronize(ctx->ev_pool[i].b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ctx->ev_pool[i].a, ctx->ev_pool[i].b));
        acc[ctx->ev_pool[i].kind] += ms;
    }
    twx_timing r{};
    r.tile_cand_ms = acc[EV_TILE]; r.select_ms = acc[EV_SELECT]; r.uk_ms = acc[EV_UK];
    r.gwr_ms = acc[EV_GWR]; r.daily_ms = acc[EV_DAILY]; r.fix_ms = acc[EV_FIX];
    if (ctx->have_total) {
        HIPCHK(hipEventSynchronize(ctx->ev_total_b));
        HIPCHK(hipEventElapsedTime(&r.total_ms, ctx->ev_total_a, ctx->ev_total_b));
    }
    long long st[5] = {0, 0, 0, 0, 0};
    if (ctx->stats.p) HIPCHK(hipMemcpy(st, ctx->stats.p, sizeof st, hipMemcpyDeviceToHost));
    r.cells = ctx->t_cells; r.uk_solves = st[0]; r.uk_launches = st[1]; r.uk_f64_solves = st[2];
    r.tie_solves = st[3]; r.tie_cells = st[4]; r.tie_ms = acc[EV_TIE]; r.deflate_ms = acc[EV_DEFLATE];
    *t = r;
    return 0;
}

int64_t twx_last_bandwidths(twx_ctx *ctx, int var, int32_t *nnghs, int64_t capacity)
{
    if (!ctx) return -1;
    ctx->err.clear();
    if (var < 0 || var > 1 || !nnghs || capacity < 0) return fail(ctx, "twx_last_bandwidths: bad arguments");
    const SelWs &ws = ctx->work[var].ws;
    if (!ws.kk || ws.ncell <= 0) return fail(ctx, "twx_last_bandwidths: nothing computed yet");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, "hipSetDevice");
    const int64_t n = std::min<int64_t>(capacity, ws.ncell * 12);
    if (hipMemcpy(nnghs, ws.kk, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(ctx, "twx_last_bandwidths: copy failed");
    return ws.ncell;
}

}  // extern "C"
