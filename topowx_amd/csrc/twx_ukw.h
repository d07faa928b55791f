// twx_ukw.h -- universal-kriging kernels for SMALL systems (k <= 96 neighbours): one wavefront per
// (cell, month) item, no work-group barrier anywhere.  Two forms (k_bucket_items picks per 8 neighbours):
//   k_ukw<NBR>      bordered, k + 8 <= 16 NBR rows        (k in the lower half of a block row of 16)
//   k_ukwz<NBR>     border rows held as columns, k <= 16 NBR  (upper half; below in this file)
//
// Same algorithm and the same three-step panel scheme as k_uk (twx_uk.h): bordered matrix
// [[C, B], [B', 0]] held negated in registers, right-looking Cholesky in 4-column panels, Schur
// complement -B'C^-1B in the trailing 7x7 block.  The 64 lanes form a 16 x 4 grid, lane (tr, tc) =
// (lane & 15, lane >> 4), and hold element (16a + tr, 4b + tc) of every 16-row x 4-column block with
// 4b <= 16a + 15.  A panel is exactly one block column, so
//   * all 64 lanes hold a piece of the panel and publish it;
//   * the 16 lanes of a DPP row share their column, i.e. the column factor of the rank-4 update is a
//     row_newbcast DPP operand of v_fmac_f64 (one LDS read per 16 of them), as in k_uk;
//   * a finished block column is dead: updates start at the next one, and block columns that hold only
//     padding (between the last C column and the RHS rows) are skipped at run time.
// The covariance build, the slab layout and the GLS epilogue (k_uk_solve) are shared with k_uk.
#pragma once
#include "twx_uk.h"

// waves per SIMD the register budget is sized for (min == max, see twx_uk.h)
#ifndef TWX_UKW_WV
#define TWX_UKW_WV 2, 3, 4, 5   // NBR = 6, 5, 4, 3 (measured on the C2 bench; NBR = 4: 122 VGPRs since the straight-line build, four waves: -8 %; NBR = 3 at six waves spills: +9 %)
#endif
__host__ __device__ constexpr int twx_ukw_waves(int nbr)
{
    constexpr int w[4] = {TWX_UKW_WV};
    return w[6 - nbr];
}

__device__ __forceinline__ constexpr int widx(int a, int b) { return 2 * a * (a + 1) + b; }   // blocks of rows < a: 4a' + 4 each

// PREC = 1: the fp64 covariance build (ill-conditioned systems, uk_needs_f64; see k_uk): distances from the cell's fp64 slab
// (k_cell_dist64), fp64 exponential, nothing read from the fp32 cache.  Same elimination, same register budget target.
template <int NBR, int PREC = 0>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(twx_ukw_waves(NBR), twx_ukw_waves(NBR))))
void k_ukw(StnDev st, CellSrc src, SelWs ws, const int32_t *item_list, const int32_t *nitems_dev)
{
    constexpr int NP = NBR * 16, NC = NP / 4, NT = 2 * NBR * (NBR + 1);
    constexpr int R0 = 9, RHS0 = 16 * (NBR - 1) + R0;                  // first RHS row / column
    constexpr int PS = 6;       // slab row stride (doubles): 48-byte rows, 16-B aligned
    __shared__ __attribute__((aligned(16))) double s_pan[NP * PS];
    __shared__ __attribute__((aligned(16))) double s_raw[NP * 4];
    __shared__ double s_B[7][NP];
    __shared__ double s_et[PREC ? TWX_EXP_TAB_N : 1];       // PREC: the table of exp_neg_f64 (twx_uk.h)

    const int lane = threadIdx.x, tr = lane & 15, tc = lane >> 4;
    // worst-case grid, device-side item count (see k_uk)
    const int it = xcd_contig(blockIdx.x, *nitems_dev);
    if (it < 0) return;
    const int item = item_list[it];
    const int64_t lc = item / 12;
    const int m0 = item % 12;
    const int64_t c = ws.cell0 + lc;
    const int k = ws.kk[lc * 12 + m0];
    const size_t n = (size_t)st.n;
    const CellVals cv = cell_load(src, c);
    const double plst = cell_lst(src, c, m0);
    const double nug = ws.vario[(lc * 12 + m0) * 3 + 0];
    const double psill = ws.vario[(lc * 12 + m0) * 3 + 1];
    const double rng = ws.vario[(lc * 12 + m0) * 3 + 2];
    const double c00 = nug + psill;
    const double c2 = rng == 0.0 ? 0.0 : -1.4426950408889634 / rng;   // -log2(e) / range
    const float chi = (float)c2;
    const double psill_e = rng == 0.0 ? 0.0 : psill;                  // pure nugget (interp.R:223-231): c(h > 0) = 0
    const float lgp = __builtin_amdgcn_logf((float)psill_e);          // log2 psill (-inf for a pure nugget)
    const int kdup = ws.cdup[lc];                                     // systems larger than this hold coincident neighbours: singular

    // the pair distances of this lane's elements (k_cell_dist's cache; 16x16 blocks, element order [column][row]):
    // every load is issued here, before the staging, so that their latency hides behind it (entries outside the
    // neighbourhood are never used; the slab of a cell always spans TWX_DIST_BLOCKS blocks, so the addresses are valid)
    // (first of all the neighbour indices, see k_uk)
    int jq[2];
    float h0q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = min(lane + 64 * u, ws.ksel - 1);
        jq[u] = __hip_atomic_load(&ws.near_idx[lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        h0q[u] = __hip_atomic_load(&ws.h0[lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    float hd[PREC ? 1 : NT];
    double A[NT];
    if constexpr (PREC) exp_tab_stage<64>(s_et, lane);     // (one wave: its LDS operations execute in order; the barrier after the staging publishes it)
    const double ninv = rng == 0.0 ? 0.0 : -1.0 / rng;       // (PREC)
    const int64_t fs = PREC ? (int64_t)ws.cellf64[lc] - 1 : 0;   // (PREC) the cell's slot in the fp64 slabs
    // PREC: the fp64 pair distances of this lane's elements travel straight into the registers that will hold the matrix
    // (A is not live before the build): every load is in flight before the staging begins, as the fast build's are -- the
    // build then turns each register into its covariance in place.  (Issued inside the build loop, one dependent load per
    // exponential, they cost the fp64 build 8.5 of its 11.4 ms over the fast one on the C2 tile.)  Plain loads, not
    // streaming ones: the other months of the cell read the same slab from the L2.
    if constexpr (PREC) {
        const double *d64 = ws.dist64 + fs * (int64_t)(TWX_DIST_BLOCKS * 256) + (tc * 16 + tr);
        sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                A[widx(a, b)] = __hip_atomic_load(&d64[(tri(a, b / 4) * 16 + 4 * (b % 4)) * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            });
        });
    }
    if constexpr (!PREC) {
        const float *dist = ws.dist + lc * (int64_t)(TWX_DIST_BLOCKS * 256) + (tc * 16 + tr);
        sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                hd[widx(a, b)] = __builtin_nontemporal_load(&dist[(tri(a, b / 4) * 16 + 4 * (b % 4)) * 16]);
            });
        });
    }

    // ---- staging: neighbours t = lane, lane + 64 (NP <= 96) -----------------------------------
    // (trend columns shifted to the cell, not scaled: see k_uk)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u;
        double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0, yv = 0.0, c0v = 0.0;
        if (t < k) {
            const int j = jq[u];
            const double4 sr = st.stat_s[j];                 // station record (see k_uk)
            const double2 mr = st.mon_s[(size_t)j * 12 + m0];
            const double lo = sr.x, la = sr.y;
            x0 = lo - cv.lon; x1 = la - cv.lat; x2 = sr.z - cv.elev;
            x3 = mr.x - plst;
            yv = mr.y;
            const float h0 = h0q[u];                           // cell -> station distance (k_cell_dist)
            const bool same = (lo == cv.lon && la == cv.lat) || h0 == 0.f;
            if constexpr (PREC) c0v = same ? c00 : psill_e * exp_neg_f64(ws.h064[fs * ws.ksel + t] * ninv, twx_exp2_tab);
            else c0v = same ? c00 : (double)cov_exp2(h0, chi, lgp);
        }
        if (t < NP) {
            s_B[0][t] = t < k ? 1.0 : 0.0;
            s_B[1][t] = x0; s_B[2][t] = x1; s_B[3][t] = x2; s_B[4][t] = x3;
            s_B[5][t] = yv; s_B[6][t] = c0v;
        }
    }
    for (int q = lane; q < NP * PS; q += 64) s_pan[q] = 0.0;   // finished rows are never written: keep them finite
    __syncthreads();

    // ---- build this lane's elements (negated: the registers hold N = -M): covariance of the cached pair
    //      distance (k_cell_dist; 16x16 blocks, element order [column][row]) -------------------------------------
    //      Straight-line: fma, v_exp_f32, convert per element (cov_exp2).  A row outside the neighbourhood has
    //      c = -inf (its elements come out 0); j <= i < k makes a column test unnecessary below the diagonal, and
    //      what lies above the diagonal is never read by the elimination.
    const bool rhs_row = tr >= R0 && tr < R0 + 7;             // of the last block row: rows RHS0..RHS0+6
    const double *rhs = &s_B[rhs_row ? tr - R0 : 0][tc];      // (s_B is 0 from column k on)
    sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
        constexpr int a = decltype(a_)::value;
        const int i = 16 * a + tr;
        const float ca = i < k ? chi : -__builtin_inff();
        sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value;
            const int j = 4 * b + tc;
            // (the last block row may lie outside what k_cell_dist has written: stale memory is selected away there,
            // not multiplied by -inf)
            double v;
            if constexpr (PREC) v = (i < k && j < i) ? psill_e * exp_neg_f64(A[widx(a, b)] * ninv, s_et) : 0.0;
            else v = (double)(a == NBR - 1 ? (i < k ? cov_exp2(hd[widx(a, b)], chi, lgp) : 0.f) : cov_exp2(hd[widx(a, b)], ca, lgp));
            // rows / columns k .. RHS0-1 are padding: an identity block there makes every panel a full 4-column
            // panel (pivot 1, factors 0), so the panel step has no special cases
            if (b >= 4 * a && i == j) v = i < k ? c00 : (i < RHS0 ? 1.0 : 0.0);
            if (a == NBR - 1) v = rhs_row ? rhs[4 * b] : v;  // unconditional LDS read: no branch per element
            A[widx(a, b)] = -v;
        });
    });
    // ---- elimination: one panel per block column ----------------------------------------------------------
    double nmax = -1.0;                                      // -(smallest pivot)
    sfor<0, NC>([&](auto b_) __attribute__((always_inline)) {
        constexpr int b = decltype(b_)::value;
        constexpr int a0 = b / 4;                            // first block row holding columns 4b..4b+3
        if (k - 4 * b > 0) {                                 // uniform: the panel holds a C column
            // (1) publish the panel as it is (LDS operations of one wave execute in order: no barrier)
            sfor<a0, NBR>([&](auto a_) __attribute__((always_inline)) {
                constexpr int a = decltype(a_)::value;
                s_raw[tc * NP + 16 * a + tr] = A[widx(a, b)];
            });
            __builtin_amdgcn_wave_barrier();
            // (2) the 4x4 diagonal block (uniform addresses: broadcasts), its Cholesky factor, then one row
            //     solve per lane and round -- the same fma sequence per element as a column-by-column sweep.
            //     A non-positive pivot gives NaN factors that reach the Schur block (k_uk_solve rejects
            //     non-finite results); too small a pivot is caught through nmax at the end.
            // column-major panel image: s_raw[column][row] (publishing lanes write consecutive rows, the row solve
            // reads consecutive rows: no bank conflicts); the diagonal block comes back as broadcasts
            const double *dg = &s_raw[4 * b];
            const double g00 = dg[0];
            const double2 g1 = double2{dg[1], dg[NP + 1]};
            const double2 g2 = double2{dg[2], dg[NP + 2]};
            const double g22 = dg[2 * NP + 2];
            const double2 g3 = double2{dg[3], dg[NP + 3]};
            const double2 g3b = double2{dg[2 * NP + 3], dg[3 * NP + 3]};
            auto pivot = [&](double nd) __attribute__((always_inline)) {
                nmax = max_raw(nmax, nd);
                return -rsqrt_nr(-nd);
            };
            const double r0 = pivot(g00);
            const double l10 = g1.x * r0, l20 = g2.x * r0, l30 = g3.x * r0;
            const double r1 = pivot(fma(l10, l10, g1.y));
            const double l21 = fma(l20, l10, g2.y) * r1, l31 = fma(l30, l10, g3.y) * r1;
            const double r2 = pivot(fma(l21, l21, fma(l20, l20, g22)));
            const double l32 = fma(l31, l21, fma(l30, l20, g3b.x)) * r2;
            const double r3 = pivot(fma(l32, l32, fma(l31, l31, fma(l30, l30, g3b.y))));
            // only the rows below the panel are needed (by the update of live elements); the finished rows keep
            // whatever the slab held, which reaches finished elements only
            constexpr int ROW0 = 4 * b + 4, NROW = NP - ROW0;
#pragma unroll
            for (int u = 0; u < (NROW + 63) / 64; ++u) {
                if (u) __builtin_amdgcn_wave_barrier();      // one round's registers at a time
                const int row = ROW0 + lane + 64 * u;
                if (row < NP) {
                    const double2 n01 = double2{s_raw[row], s_raw[NP + row]};
                    const double2 n23 = double2{s_raw[2 * NP + row], s_raw[3 * NP + row]};
                    const double L0 = n01.x * r0;
                    const double L1 = fma(L0, l10, n01.y) * r1;
                    const double L2 = fma(L1, l21, fma(L0, l20, n23.x)) * r2;
                    const double L3 = fma(L2, l32, fma(L1, l31, fma(L0, l30, n23.y))) * r3;
                    *reinterpret_cast<double2 *>(&s_pan[row * PS]) = double2{L0, L1};
                    *reinterpret_cast<double2 *>(&s_pan[row * PS + 2]) = double2{L2, L3};
                }
            }
            __builtin_amdgcn_wave_barrier();
            // (3) rank-4 update of the block columns b+1 .. (the panel's own block column is finished)
            if constexpr (b + 1 < NC) {
                constexpr int NE = 4 * (NC - b - 1), NR = (NE + 15) / 16;
                double P[NR];
                sfor<0, NR>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    const int e = min(16 * r + tr, NE - 1);
                    P[r] = s_pan[(4 * (b + 1 + (e >> 2)) + tc) * PS + (e & 3)];
                });
                constexpr int a1 = (b + 1) / 4;              // first block row with a block column > b
                sfor<a1, NBR>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value;
                    const double2 u0 = *reinterpret_cast<const double2 *>(&s_pan[(16 * a + tr) * PS]);
                    const double2 u1 = *reinterpret_cast<const double2 *>(&s_pan[(16 * a + tr) * PS + 2]);
                    // the (at most two) padding block columns between the last C column and the RHS columns are
                    // updated like the rest -- a run-time test per block would cost more issue slots than their four fmacs
                    constexpr int BHI = (RHS0 + 6) / 4 + 1;
                    sfor<b + 1, (4 * a + 4 < BHI ? 4 * a + 4 : BHI)>([&](auto bb_) __attribute__((always_inline)) {
                        constexpr int bb = decltype(bb_)::value;
                        constexpr int e = 4 * (bb - b - 1);
                        double acc = A[widx(a, bb)];
                        fmac_rowbcast<(e + 0) % 16>(acc, P[(e + 0) / 16], u0.x);
                        fmac_rowbcast<(e + 1) % 16>(acc, P[(e + 1) / 16], u0.y);
                        fmac_rowbcast<(e + 2) % 16>(acc, P[(e + 2) / 16], u1.x);
                        fmac_rowbcast<(e + 3) % 16>(acc, P[(e + 3) / 16], u1.y);
                        A[widx(a, bb)] = acc;
                    });
                });
            }
            __builtin_amdgcn_wave_barrier();
        }
    });

    // ---- Schur complement out (k_uk_solve finishes): rows / columns RHS0 .. RHS0+6 ------------------------------
    if (tr >= R0 && tr < R0 + 7) {
        const int r = tr - R0;
        constexpr int B0 = RHS0 / 4;                         // the two block columns holding the RHS columns
        sfor<B0, B0 + 2>([&](auto bb_) __attribute__((always_inline)) {
            constexpr int bb = decltype(bb_)::value;
            const int cq = 4 * bb + tc - RHS0;
            if (cq >= 0 && r >= cq)
                ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + r * (r + 1) / 2 + cq] = A[widx(NBR - 1, bb)];
        });
    }
    if (lane == 0) ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + 28] = (-nmax > 1e-9 * c00 && k <= kdup) ? 0.0 : 1.0;   // singular / indefinite
}

// ---------------------------------------------------------------------------------
// k_ukwz<NBR>: the same one-wave elimination with the seven border rows held as COLUMNS.
//
// The bordered form pads C's k rows + 7 border rows to a multiple of 16 ROWS -- and the rows at the end of a lower
// triangle are the long ones.  For k in (16 NBR - 8, 16 NBR] the border would open a block row of its own (16 lanes
// of every instruction for 7 rows); here the matrix holds the C rows only (NP = 16 NBR >= k rows, identity padding)
// and the border B (7 x k) lives transposed in two registers per block row: lane (tr, tc) holds Z[16a + tr][tc] and
// Z[16a + tr][4 + tc] (column 7 is a zero dummy).  Algebraically nothing changes -- an entry of a border row is
// updated with the row factor of ITS COLUMN's matrix row (the u registers of the block row, already loaded) and the
// border row's own panel factor (a DPP broadcast, as the column factors are) -- but every panel issues 8 fmacs per
// block row for the border instead of 4 per block column of a whole extra block row: 15-23 % fewer fmacs at these
// sizes (tests/tools/count_fmac.py), and one block row less of registers (96 rows fit one wave at 2 waves per SIMD:
// the systems of 89..96 neighbours leave the two-wave kernel).  The panel's four border columns are published with
// the panel (rows NP .. NP+6 of the panel image), row-solved like any other row, and the 7 x 7 corner B'C^-1B is
// accumulated from the solved border factors in two registers of the lanes (tr = c', tc = c mod 4).
// ---------------------------------------------------------------------------------
#ifndef TWX_UKWZ_WV
#define TWX_UKWZ_WV 2, 2, 3, 4   // NBR = 6, 5, 4, 3 (measured on the C2 bench)
#endif
__host__ __device__ constexpr int twx_ukwz_waves(int nbr)
{
    constexpr int w[4] = {TWX_UKWZ_WV};
    return w[6 - nbr];
}

template <int NBR, int PREC = 0>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(twx_ukwz_waves(NBR), twx_ukwz_waves(NBR))))
void k_ukwz(StnDev st, CellSrc src, SelWs ws, const int32_t *item_list, const int32_t *nitems_dev)
{
    constexpr int NP = NBR * 16, NC = NP / 4, NT = 2 * NBR * (NBR + 1);
    constexpr int NPX = NP + 8;     // rows of the panel image / slab: C rows + 7 border rows (+ 1 dummy)
    constexpr int PS = 6;           // slab row stride (doubles): 48-byte rows, 16-B aligned
    __shared__ __attribute__((aligned(16))) double s_pan[NPX * PS];
    __shared__ __attribute__((aligned(16))) double s_raw[NPX * 4];
    __shared__ double s_B[7][NP];
    __shared__ double s_et[PREC ? TWX_EXP_TAB_N : 1];       // PREC: the table of exp_neg_f64 (twx_uk.h)

    const int lane = threadIdx.x, tr = lane & 15, tc = lane >> 4;
    const int it = xcd_contig(blockIdx.x, *nitems_dev);
    if (it < 0) return;
    const int item = item_list[it];
    const int64_t lc = item / 12;
    const int m0 = item % 12;
    const int64_t c = ws.cell0 + lc;
    const int k = ws.kk[lc * 12 + m0];
    const size_t n = (size_t)st.n;
    const CellVals cv = cell_load(src, c);
    const double plst = cell_lst(src, c, m0);
    const double nug = ws.vario[(lc * 12 + m0) * 3 + 0];
    const double psill = ws.vario[(lc * 12 + m0) * 3 + 1];
    const double rng = ws.vario[(lc * 12 + m0) * 3 + 2];
    const double c00 = nug + psill;
    const double c2 = rng == 0.0 ? 0.0 : -1.4426950408889634 / rng;   // -log2(e) / range
    const float chi = (float)c2;
    const double psill_e = rng == 0.0 ? 0.0 : psill;                  // pure nugget (interp.R:223-231): c(h > 0) = 0
    const float lgp = __builtin_amdgcn_logf((float)psill_e);          // log2 psill (-inf for a pure nugget)
    const int kdup = ws.cdup[lc];                                     // systems larger than this hold coincident neighbours: singular

    int jq[2];
    float h0q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = min(lane + 64 * u, ws.ksel - 1);
        jq[u] = __hip_atomic_load(&ws.near_idx[lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        h0q[u] = __hip_atomic_load(&ws.h0[lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    float hd[PREC ? 1 : NT];
    double A[NT];
    if constexpr (PREC) exp_tab_stage<64>(s_et, lane);     // (one wave: its LDS operations execute in order; the barrier after the staging publishes it)
    const double ninv = rng == 0.0 ? 0.0 : -1.0 / rng;       // (PREC)
    const int64_t fs = PREC ? (int64_t)ws.cellf64[lc] - 1 : 0;   // (PREC) the cell's slot in the fp64 slabs
    // PREC: the fp64 pair distances of this lane's elements travel straight into the registers that will hold the matrix
    // (A is not live before the build): every load is in flight before the staging begins, as the fast build's are -- the
    // build then turns each register into its covariance in place.  (Issued inside the build loop, one dependent load per
    // exponential, they cost the fp64 build 8.5 of its 11.4 ms over the fast one on the C2 tile.)  Plain loads, not
    // streaming ones: the other months of the cell read the same slab from the L2.
    if constexpr (PREC) {
        const double *d64 = ws.dist64 + fs * (int64_t)(TWX_DIST_BLOCKS * 256) + (tc * 16 + tr);
        sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                A[widx(a, b)] = __hip_atomic_load(&d64[(tri(a, b / 4) * 16 + 4 * (b % 4)) * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            });
        });
    }
    if constexpr (!PREC) {
        const float *dist = ws.dist + lc * (int64_t)(TWX_DIST_BLOCKS * 256) + (tc * 16 + tr);
        sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                hd[widx(a, b)] = __builtin_nontemporal_load(&dist[(tri(a, b / 4) * 16 + 4 * (b % 4)) * 16]);
            });
        });
    }

    // ---- staging: neighbours lane, lane + 64 (NP <= 96); trend columns shifted to the cell, not scaled (see k_uk)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u;
        double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0, yv = 0.0, c0v = 0.0;
        if (t < k) {
            const int j = jq[u];
            const double4 sr = st.stat_s[j];                 // station record (see k_uk)
            const double2 mr = st.mon_s[(size_t)j * 12 + m0];
            const double lo = sr.x, la = sr.y;
            x0 = lo - cv.lon; x1 = la - cv.lat; x2 = sr.z - cv.elev;
            x3 = mr.x - plst;
            yv = mr.y;
            const float h0 = h0q[u];                           // cell -> station distance (k_cell_dist)
            const bool same = (lo == cv.lon && la == cv.lat) || h0 == 0.f;
            if constexpr (PREC) c0v = same ? c00 : psill_e * exp_neg_f64(ws.h064[fs * ws.ksel + t] * ninv, twx_exp2_tab);
            else c0v = same ? c00 : (double)cov_exp2(h0, chi, lgp);
        }
        if (t < NP) {
            s_B[0][t] = t < k ? 1.0 : 0.0;
            s_B[1][t] = x0; s_B[2][t] = x1; s_B[3][t] = x2; s_B[4][t] = x3;
            s_B[5][t] = yv; s_B[6][t] = c0v;
        }
    }
    for (int q = lane; q < NPX * PS; q += 64) s_pan[q] = 0.0;   // finished rows are never written: keep them finite
    __syncthreads();

    // ---- build (negated: the registers hold N = -M): C with an identity block in rows / columns k .. NP-1, and the
    //      border transposed: Z[a][h] = -B[tc + 4h][16a + tr] (0 for the dummy column 7 and from row k on)
    double Z[NBR][2], Sacc[2] = {0.0, 0.0};
    sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
        constexpr int a = decltype(a_)::value;
        const int i = 16 * a + tr;
        const float ca = i < k ? chi : -__builtin_inff();
        sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value;
            const int j = 4 * b + tc;
            // (the last block row may lie outside what k_cell_dist has written: stale memory is selected away there,
            // not multiplied by -inf)
            double v;
            if constexpr (PREC) v = (i < k && j < i) ? psill_e * exp_neg_f64(A[widx(a, b)] * ninv, s_et) : 0.0;
            else v = (double)(a == NBR - 1 ? (i < k ? cov_exp2(hd[widx(a, b)], chi, lgp) : 0.f) : cov_exp2(hd[widx(a, b)], ca, lgp));
            if (b >= 4 * a && i == j) v = i < k ? c00 : 1.0;
            A[widx(a, b)] = -v;
        });
        Z[a][0] = -s_B[tc][i];
        Z[a][1] = tc < 3 ? -s_B[4 + (tc < 3 ? tc : 0)][i] : 0.0;
    });
    // ---- elimination: one panel per block column ----------------------------------------------------------
    double nmax = -1.0;                                      // -(smallest pivot)
    sfor<0, NC>([&](auto b_) __attribute__((always_inline)) {
        constexpr int b = decltype(b_)::value;
        constexpr int a0 = b / 4;                            // block row holding rows / columns 4b..4b+3
        if (k - 4 * b > 0) {                                 // uniform: the panel holds a C column
            // (1) publish the panel as it is: its C rows, and its four border columns as rows NP .. NP+7
            sfor<a0, NBR>([&](auto a_) __attribute__((always_inline)) {
                constexpr int a = decltype(a_)::value;
                s_raw[tc * NPX + 16 * a + tr] = A[widx(a, b)];
            });
            if ((tr >> 2) == (b & 3)) {
                s_raw[(tr & 3) * NPX + NP + tc] = Z[a0][0];
                s_raw[(tr & 3) * NPX + NP + 4 + tc] = Z[a0][1];
            }
            __builtin_amdgcn_wave_barrier();
            // (2) the 4x4 diagonal block, its Cholesky factor, one row solve per lane and round (see k_ukw)
            const double *dg = &s_raw[4 * b];
            const double g00 = dg[0];
            const double2 g1 = double2{dg[1], dg[NPX + 1]};
            const double2 g2 = double2{dg[2], dg[NPX + 2]};
            const double g22 = dg[2 * NPX + 2];
            const double2 g3 = double2{dg[3], dg[NPX + 3]};
            const double2 g3b = double2{dg[2 * NPX + 3], dg[3 * NPX + 3]};
            auto pivot = [&](double nd) __attribute__((always_inline)) {
                nmax = max_raw(nmax, nd);
                return -rsqrt_nr(-nd);
            };
            const double r0 = pivot(g00);
            const double l10 = g1.x * r0, l20 = g2.x * r0, l30 = g3.x * r0;
            const double r1 = pivot(fma(l10, l10, g1.y));
            const double l21 = fma(l20, l10, g2.y) * r1, l31 = fma(l30, l10, g3.y) * r1;
            const double r2 = pivot(fma(l21, l21, fma(l20, l20, g22)));
            const double l32 = fma(l31, l21, fma(l30, l20, g3b.x)) * r2;
            const double r3 = pivot(fma(l32, l32, fma(l31, l31, fma(l30, l30, g3b.y))));
            constexpr int ROW0 = 4 * b + 4, NROW = NPX - ROW0;
#pragma unroll
            for (int u = 0; u < (NROW + 63) / 64; ++u) {
                if (u) __builtin_amdgcn_wave_barrier();      // one round's registers at a time
                const int row = ROW0 + lane + 64 * u;
                if (row < NPX) {
                    const double2 n01 = double2{s_raw[row], s_raw[NPX + row]};
                    const double2 n23 = double2{s_raw[2 * NPX + row], s_raw[3 * NPX + row]};
                    const double L0 = n01.x * r0;
                    const double L1 = fma(L0, l10, n01.y) * r1;
                    const double L2 = fma(L1, l21, fma(L0, l20, n23.x)) * r2;
                    const double L3 = fma(L2, l32, fma(L1, l31, fma(L0, l30, n23.y))) * r3;
                    *reinterpret_cast<double2 *>(&s_pan[row * PS]) = double2{L0, L1};
                    *reinterpret_cast<double2 *>(&s_pan[row * PS + 2]) = double2{L2, L3};
                }
            }
            __builtin_amdgcn_wave_barrier();
            // (3) rank-4 update: C block columns b+1 .., the border entries of the rows below the panel, the corner
            {
                // the panel factors of the border rows tc and 4 + tc (lane n of a DPP row holds factor n & 3) and of
                // border row tr & 7 (the corner's other operand)
                const double pz0 = s_pan[(NP + tc) * PS + (tr & 3)], pz1 = s_pan[(NP + 4 + tc) * PS + (tr & 3)];
                const double2 c0 = *reinterpret_cast<const double2 *>(&s_pan[(NP + (tr & 7)) * PS]);
                const double2 c1 = *reinterpret_cast<const double2 *>(&s_pan[(NP + (tr & 7)) * PS + 2]);
                fmac_rowbcast<0>(Sacc[0], pz0, c0.x); fmac_rowbcast<1>(Sacc[0], pz0, c0.y);
                fmac_rowbcast<2>(Sacc[0], pz0, c1.x); fmac_rowbcast<3>(Sacc[0], pz0, c1.y);
                fmac_rowbcast<0>(Sacc[1], pz1, c0.x); fmac_rowbcast<1>(Sacc[1], pz1, c0.y);
                fmac_rowbcast<2>(Sacc[1], pz1, c1.x); fmac_rowbcast<3>(Sacc[1], pz1, c1.y);
                constexpr int NE = 4 * (NC - b - 1), NR = (NE + 15) / 16;
                double P[NR > 0 ? NR : 1];
                sfor<0, NR>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    const int e = min(16 * r + tr, NE - 1);
                    P[r] = s_pan[(4 * (b + 1 + (e >> 2)) + tc) * PS + (e & 3)];
                });
                constexpr int a1 = (b + 1) / 4;              // first block row with rows below the panel
                sfor<a1, NBR>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value;
                    const double2 u0 = *reinterpret_cast<const double2 *>(&s_pan[(16 * a + tr) * PS]);
                    const double2 u1 = *reinterpret_cast<const double2 *>(&s_pan[(16 * a + tr) * PS + 2]);
                    fmac_rowbcast<0>(Z[a][0], pz0, u0.x); fmac_rowbcast<1>(Z[a][0], pz0, u0.y);
                    fmac_rowbcast<2>(Z[a][0], pz0, u1.x); fmac_rowbcast<3>(Z[a][0], pz0, u1.y);
                    fmac_rowbcast<0>(Z[a][1], pz1, u0.x); fmac_rowbcast<1>(Z[a][1], pz1, u0.y);
                    fmac_rowbcast<2>(Z[a][1], pz1, u1.x); fmac_rowbcast<3>(Z[a][1], pz1, u1.y);
                    sfor<b + 1, 4 * a + 4>([&](auto bb_) __attribute__((always_inline)) {
                        constexpr int bb = decltype(bb_)::value;
                        constexpr int e = 4 * (bb - b - 1);
                        double acc = A[widx(a, bb)];
                        fmac_rowbcast<(e + 0) % 16>(acc, P[(e + 0) / 16], u0.x);
                        fmac_rowbcast<(e + 1) % 16>(acc, P[(e + 1) / 16], u0.y);
                        fmac_rowbcast<(e + 2) % 16>(acc, P[(e + 2) / 16], u1.x);
                        fmac_rowbcast<(e + 3) % 16>(acc, P[(e + 3) / 16], u1.y);
                        A[widx(a, bb)] = acc;
                    });
                });
            }
            __builtin_amdgcn_wave_barrier();
        }
    });

    // ---- Schur complement out (k_uk_solve finishes): lane (tr = c', tc) holds S[tc + 4h][c'] in Sacc[h] -------------
    if (tr < 7) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = tc + 4 * h;
            if (r < 7 && tr <= r) ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + r * (r + 1) / 2 + tr] = Sacc[h];
        }
    }
    if (lane == 0) ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + 28] = (-nmax > 1e-9 * c00 && k <= kdup) ? 0.0 : 1.0;   // singular / indefinite
}

// ---------------------------------------------------------------------------------
// k_ukw2<NBR>: TWO systems per wave (bordered form, k + 8 <= 16 NBR, NBR = 3, 4).
//
// In the one-wave kernels the uniform part of a panel step -- the 4x4 Cholesky chain (~40 instructions whose 64 lanes
// all compute the same ten numbers) and the addressing around it -- costs as much issue time as the fmacs of a 64-row
// system.  Here a wave holds two systems, lanes 0..31 one, lanes 32..63 the other: the chain is issued once for both
// (each half on its own diagonal block), the row solve takes 32 rows per round and system, every LDS instruction serves
// both.  Per system a 16 x 2 lane grid: lane (tr, tc2) holds elements (16a + tr, 4b + 2q + tc2), q = 0, 1, of every
// 16-row x 4-column block -- a DPP row of 16 lanes still shares its column, so the column factor of the update stays a
// row_newbcast operand.  Twice the matrix registers per lane: 2 waves per SIMD = 4 systems per SIMD as before for 64 rows.
// What is per system (item, bandwidth, variogram, cell) lives in vector registers, uniform within a half.
// Measured (same-box A/B, C2 step): 48 rows 112.6 vs 120 us (-6 %: used), 64 rows 518 vs 509 us (203 VGPRs: two waves per
// SIMD instead of four; what the shared chain saves in issue slots the lost occupancy takes back: NOT used, TWX_UKW2 = 2
// in twx_hip.hip reproduces it).  80 rows and more do not fit the register file twice.
// ---------------------------------------------------------------------------------
#ifndef TWX_UKW2_WV
#define TWX_UKW2_WV 2, 3        // NBR = 4, 3
#endif
__host__ __device__ constexpr int twx_ukw2_waves(int nbr)
{
    constexpr int w[2] = {TWX_UKW2_WV};
    return w[4 - nbr];
}

template <int NBR, int PREC = 0>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(twx_ukw2_waves(NBR), twx_ukw2_waves(NBR))))
void k_ukw2(StnDev st, CellSrc src, SelWs ws, const int32_t *item_list, const int32_t *nitems_dev)
{
    constexpr int NP = NBR * 16, NC = NP / 4, NT = 2 * NBR * (NBR + 1);
    constexpr int R0 = 9, RHS0 = 16 * (NBR - 1) + R0;       // first RHS row / column
    constexpr int PS = 6;                                    // slab row stride (doubles)
    constexpr int NPS = NP + 2;                              // column stride of the panel image (the two systems' images in different banks)
    __shared__ __attribute__((aligned(16))) double s_pan[2][NP * PS + 2];
    __shared__ __attribute__((aligned(16))) double s_raw[2][NPS * 4];
    __shared__ double s_B[2][7][NP];
    __shared__ double s_et[PREC ? TWX_EXP_TAB_N : 1];       // PREC: the table of exp_neg_f64 (twx_uk.h)

    const int lane = threadIdx.x, tr = lane & 15, tc2 = (lane >> 4) & 1, l32 = lane & 31, sys = lane >> 5;
    const int nitems = *nitems_dev;
    const int pr = xcd_contig(blockIdx.x, (nitems + 1) >> 1);
    if (pr < 0) return;
    const bool active = 2 * pr + sys < nitems;               // (an odd list: the last wave's second half repeats the first)
    const int item = item_list[min(2 * pr + sys, nitems - 1)];
    const int lc = item / 12;
    const int m0 = item - 12 * lc;
    const int k = ws.kk[(int64_t)lc * 12 + m0];
    const size_t n = (size_t)st.n;
    // the cell's predictors: per system through the scalar path, then selected per half
    CellVals cv;
    double plst;
    {
        const int lc0 = __builtin_amdgcn_readlane(lc, 0), lc1 = __builtin_amdgcn_readlane(lc, 32);
        const int ma = __builtin_amdgcn_readlane(m0, 0), mb = __builtin_amdgcn_readlane(m0, 32);
        const CellVals ca = cell_load(src, ws.cell0 + lc0), cb = cell_load(src, ws.cell0 + lc1);
        const double pa = cell_lst(src, ws.cell0 + lc0, ma), pb = cell_lst(src, ws.cell0 + lc1, mb);
        cv.lon = sys ? cb.lon : ca.lon; cv.lat = sys ? cb.lat : ca.lat; cv.elev = sys ? cb.elev : ca.elev; cv.tdi = 0.0;
        plst = sys ? pb : pa;
    }
    const double *vp = ws.vario + ((int64_t)lc * 12 + m0) * 3;
    const double nug = vp[0], psill = vp[1], rng = vp[2];
    const double c00 = nug + psill;
    const double c2 = rng == 0.0 ? 0.0 : -1.4426950408889634 / rng;   // -log2(e) / range
    const float chi = (float)c2;
    const double psill_e = rng == 0.0 ? 0.0 : psill;                  // pure nugget (interp.R:223-231): c(h > 0) = 0
    const float lgp = __builtin_amdgcn_logf((float)psill_e);          // log2 psill (-inf for a pure nugget)
    const int kdup = ws.cdup[lc];
    const int kmx = max(__builtin_amdgcn_readlane(k, 0), __builtin_amdgcn_readlane(k, 32));   // panels the wave walks

    int jq[2];
    float h0q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = min(l32 + 32 * u, ws.ksel - 1);
        jq[u] = __hip_atomic_load(&ws.near_idx[(int64_t)lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        h0q[u] = __hip_atomic_load(&ws.h0[(int64_t)lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    float hd[PREC ? 1 : 2 * NT];
    double A[2 * NT];
    if constexpr (PREC) exp_tab_stage<64>(s_et, lane);     // (one wave: its LDS operations execute in order; the barrier after the staging publishes it)
    const double ninv = rng == 0.0 ? 0.0 : -1.0 / rng;       // (PREC)
    const int64_t fs = PREC ? (int64_t)ws.cellf64[lc] - 1 : 0;   // (PREC) this half's cell's slot in the fp64 slabs
    if constexpr (PREC) {   // (the fp64 pair distances go straight into the matrix registers: see k_ukw)
        const double *d64 = ws.dist64 + fs * (int64_t)(TWX_DIST_BLOCKS * 256) + (tc2 * 16 + tr);
        sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                A[2 * widx(a, b) + 0] = __hip_atomic_load(&d64[(tri(a, b / 4) * 16 + 4 * (b % 4) + 0) * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                A[2 * widx(a, b) + 1] = __hip_atomic_load(&d64[(tri(a, b / 4) * 16 + 4 * (b % 4) + 2) * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            });
        });
    }
    if constexpr (!PREC) {
        const float *dist = ws.dist + lc * (int64_t)(TWX_DIST_BLOCKS * 256) + (tc2 * 16 + tr);
        sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                hd[2 * widx(a, b) + 0] = __builtin_nontemporal_load(&dist[(tri(a, b / 4) * 16 + 4 * (b % 4) + 0) * 16]);
                hd[2 * widx(a, b) + 1] = __builtin_nontemporal_load(&dist[(tri(a, b / 4) * 16 + 4 * (b % 4) + 2) * 16]);
            });
        });
    }

    // ---- staging: neighbours l32, l32 + 32 of this half's system (NP <= 64) ------------------------------------
    double (*sB)[NP] = s_B[sys];
    double *span = s_pan[sys], *sraw = s_raw[sys];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = l32 + 32 * u;
        double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0, yv = 0.0, c0v = 0.0;
        if (t < k) {
            const int j = jq[u];
            const double4 sr = st.stat_s[j];
            const double2 mr = st.mon_s[(size_t)j * 12 + m0];
            const double lo = sr.x, la = sr.y;
            x0 = lo - cv.lon; x1 = la - cv.lat; x2 = sr.z - cv.elev; x3 = mr.x - plst;
            yv = mr.y;
            const float h0 = h0q[u];
            const bool same = (lo == cv.lon && la == cv.lat) || h0 == 0.f;
            if constexpr (PREC) c0v = same ? c00 : psill_e * exp_neg_f64(ws.h064[fs * ws.ksel + t] * ninv, twx_exp2_tab);
            else c0v = same ? c00 : (double)cov_exp2(h0, chi, lgp);
        }
        if (t < NP) {
            sB[0][t] = t < k ? 1.0 : 0.0;
            sB[1][t] = x0; sB[2][t] = x1; sB[3][t] = x2; sB[4][t] = x3;
            sB[5][t] = yv; sB[6][t] = c0v;
        }
    }
    for (int q = l32; q < NP * PS; q += 32) span[q] = 0.0;   // finished rows are never written: keep them finite
    __syncthreads();

    // ---- build (negated) ----------------------------------------------------------------------------------------
    const bool rhs_row = tr >= R0 && tr < R0 + 7;
    const double *rhs = &sB[rhs_row ? tr - R0 : 0][tc2];
    sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
        constexpr int a = decltype(a_)::value;
        const int i = 16 * a + tr;
        const float ca = i < k ? chi : -__builtin_inff();
        sfor<0, 4 * a + 4>([&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value;
            sfor<0, 2>([&](auto q_) __attribute__((always_inline)) {
                constexpr int q = decltype(q_)::value;
                constexpr int e = 2 * widx(a, b) + q;
                const int j = 4 * b + 2 * q + tc2;
                double v;
                if constexpr (PREC) v = (i < k && j < i) ? psill_e * exp_neg_f64(A[e] * ninv, s_et) : 0.0;
                else v = (double)(a == NBR - 1 ? (i < k ? cov_exp2(hd[e], chi, lgp) : 0.f) : cov_exp2(hd[e], ca, lgp));
                if (b >= 4 * a && i == j) v = i < k ? c00 : (i < RHS0 ? 1.0 : 0.0);
                if (a == NBR - 1) v = rhs_row ? rhs[4 * b + 2 * q] : v;
                A[e] = -v;
            });
        });
    });
    // ---- elimination: one panel per block column, both systems at once -------------------------------------------
    double nmax = -1.0;
    sfor<0, NC>([&](auto b_) __attribute__((always_inline)) {
        constexpr int b = decltype(b_)::value;
        constexpr int a0 = b / 4;
        if (kmx - 4 * b > 0) {                               // uniform: one of the two systems still has a C column here
            sfor<a0, NBR>([&](auto a_) __attribute__((always_inline)) {
                constexpr int a = decltype(a_)::value;
                sraw[tc2 * NPS + 16 * a + tr] = A[2 * widx(a, b)];
                sraw[(2 + tc2) * NPS + 16 * a + tr] = A[2 * widx(a, b) + 1];
            });
            __builtin_amdgcn_wave_barrier();
            const double *dg = &sraw[4 * b];
            const double g00 = dg[0];
            const double2 g1 = double2{dg[1], dg[NPS + 1]};
            const double2 g2 = double2{dg[2], dg[NPS + 2]};
            const double g22 = dg[2 * NPS + 2];
            const double2 g3 = double2{dg[3], dg[NPS + 3]};
            const double2 g3b = double2{dg[2 * NPS + 3], dg[3 * NPS + 3]};
            auto pivot = [&](double nd) __attribute__((always_inline)) {
                nmax = max_raw(nmax, nd);
                return -rsqrt_nr(-nd);
            };
            const double r0 = pivot(g00);
            const double l10 = g1.x * r0, l20 = g2.x * r0, l30 = g3.x * r0;
            const double r1 = pivot(fma(l10, l10, g1.y));
            const double l21 = fma(l20, l10, g2.y) * r1, l31 = fma(l30, l10, g3.y) * r1;
            const double r2 = pivot(fma(l21, l21, fma(l20, l20, g22)));
            const double l32v = fma(l31, l21, fma(l30, l20, g3b.x)) * r2;
            const double r3 = pivot(fma(l32v, l32v, fma(l31, l31, fma(l30, l30, g3b.y))));
            constexpr int ROW0 = 4 * b + 4, NROW = NP - ROW0;
#pragma unroll
            for (int u = 0; u < (NROW + 31) / 32; ++u) {
                if (u) __builtin_amdgcn_wave_barrier();
                const int row = ROW0 + l32 + 32 * u;
                if (row < NP) {
                    const double2 n01 = double2{sraw[row], sraw[NPS + row]};
                    const double2 n23 = double2{sraw[2 * NPS + row], sraw[3 * NPS + row]};
                    const double L0 = n01.x * r0;
                    const double L1 = fma(L0, l10, n01.y) * r1;
                    const double L2 = fma(L1, l21, fma(L0, l20, n23.x)) * r2;
                    const double L3 = fma(L2, l32v, fma(L1, l31, fma(L0, l30, n23.y))) * r3;
                    *reinterpret_cast<double2 *>(&span[row * PS]) = double2{L0, L1};
                    *reinterpret_cast<double2 *>(&span[row * PS + 2]) = double2{L2, L3};
                }
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (b + 1 < NC) {
                // column factors of this lane's two columns per block column: entry e = 4 (bb - b - 1) + factor
                constexpr int NE = 4 * (NC - b - 1), NR = (NE + 15) / 16;
                double P0[NR], P1[NR];
                sfor<0, NR>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    const int e = min(16 * r + tr, NE - 1);
                    P0[r] = span[(4 * (b + 1 + (e >> 2)) + tc2) * PS + (e & 3)];
                    P1[r] = span[(4 * (b + 1 + (e >> 2)) + 2 + tc2) * PS + (e & 3)];
                });
                constexpr int a1 = (b + 1) / 4;
                sfor<a1, NBR>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value;
                    const double2 u0 = *reinterpret_cast<const double2 *>(&span[(16 * a + tr) * PS]);
                    const double2 u1 = *reinterpret_cast<const double2 *>(&span[(16 * a + tr) * PS + 2]);
                    constexpr int BHI = (RHS0 + 6) / 4 + 1;
                    sfor<b + 1, (4 * a + 4 < BHI ? 4 * a + 4 : BHI)>([&](auto bb_) __attribute__((always_inline)) {
                        constexpr int bb = decltype(bb_)::value;
                        constexpr int e = 4 * (bb - b - 1);
                        double acc = A[2 * widx(a, bb)];
                        fmac_rowbcast<(e + 0) % 16>(acc, P0[(e + 0) / 16], u0.x);
                        fmac_rowbcast<(e + 1) % 16>(acc, P0[(e + 1) / 16], u0.y);
                        fmac_rowbcast<(e + 2) % 16>(acc, P0[(e + 2) / 16], u1.x);
                        fmac_rowbcast<(e + 3) % 16>(acc, P0[(e + 3) / 16], u1.y);
                        A[2 * widx(a, bb)] = acc;
                        double acd = A[2 * widx(a, bb) + 1];
                        fmac_rowbcast<(e + 0) % 16>(acd, P1[(e + 0) / 16], u0.x);
                        fmac_rowbcast<(e + 1) % 16>(acd, P1[(e + 1) / 16], u0.y);
                        fmac_rowbcast<(e + 2) % 16>(acd, P1[(e + 2) / 16], u1.x);
                        fmac_rowbcast<(e + 3) % 16>(acd, P1[(e + 3) / 16], u1.y);
                        A[2 * widx(a, bb) + 1] = acd;
                    });
                });
            }
            __builtin_amdgcn_wave_barrier();
        }
    });

    // ---- Schur complement out: rows / columns RHS0 .. RHS0+6 --------------------------------------------------------
    double *Sout = ws.uk_S + ((int64_t)lc * 12 + m0) * TWX_UK_SLEN;
    if (active && tr >= R0 && tr < R0 + 7) {
        const int r = tr - R0;
        constexpr int B0 = RHS0 / 4;
        sfor<B0, B0 + 2>([&](auto bb_) __attribute__((always_inline)) {
            constexpr int bb = decltype(bb_)::value;
            sfor<0, 2>([&](auto q_) __attribute__((always_inline)) {
                constexpr int q = decltype(q_)::value;
                const int cq = 4 * bb + 2 * q + tc2 - RHS0;
                if (cq >= 0 && cq < 7 && r >= cq) Sout[r * (r + 1) / 2 + cq] = A[2 * widx(NBR - 1, bb) + q];
            });
        });
    }
    if (active && l32 == 0) Sout[28] = (-nmax > 1e-9 * c00 && k <= kdup) ? 0.0 : 1.0;   // singular / indefinite
}
