// EXPERIMENT RECORD (round 5) -- NOT part of the library, not compiled by build.sh.
// Measured on the MI355X in one gpurun call against the kernels it would replace (C2 step, rocprofv3 averages):
//   k_ukp<4> 570 us  vs  k_ukw<4> 510 us  (111 488 systems of 49..56 neighbours)
//   k_ukp<3> 121 us  vs  k_ukw2<3> 114 us (40 707 systems of <= 40 neighbours)
// and its results were WRONG on part of the systems (tests/test_gpu_parity.py failed; not debugged, because it is slower anyway).
// Why it is slower although it has no LDS traffic, no panel chain and no barrier in the elimination: v_permlane16_swap_b32
// overwrites BOTH of its operands, so handing a double to the partner DPP row costs 2 swaps + 2-4 register copies + 2 masked
// rotations + hazard nops; the disassembly of k_ukp<4> holds 6 106 instructions per wave of two systems -- 2 179 fmacs, 414 swaps,
// 559 v_mov_b32, 300 v_mov_b32_dpp, 327 s_nop, 323 v_mul_f64 -- i.e. as many per system as the panel kernel, at two waves per SIMD
// (184 VGPRs) instead of four.  To try it again: copy into topowx_amd/csrc/, include after twx_ukw.h, launch like k_ukw2.
//
// twx_ukp.h -- universal kriging for small systems WITHOUT LDS in the elimination: k_ukp<NBR>, two systems per wave.
//
// The one-wave kernels of twx_ukw.h spend as many issue slots on the panel machinery (publish the panel to LDS, read the
// 4x4 block back, four dependent rsqrt that all 64 lanes compute alike, row solve, slab reads of the row factors) as on the
// fmacs of a 64-row system, and keep the CU's one LDS pipe about as busy as a SIMD's VALU (profiles/r5_sq_counters_kriging.txt).
// Here a system lives on 32 lanes = two DPP rows and is eliminated COLUMN BY COLUMN out of registers:
//   * lane (h, t), h = DPP row of the system, t = lane of the row, holds matrix rows 16 a + rho, rho = (t + 8 h) & 15 -- every row
//     block in both DPP rows, the second one rotated by eight lanes -- and the columns j with ((j & 15) >> 3) == h: register
//     (a, b, n) = element (16 a + rho, 16 b + 8 h + n), b <= a, n < 8.  8 (1 + 2 + .. + NBR) doubles per lane.
//   * the column factor of a rank-1 step, l(j, p) for column j = 16 b + 8 h + n, sits in the register that holds the row
//     factors of row block b at lane n OF BOTH DPP ROWS (that is what the rotation is for): one v_fmac_f64 with a
//     row_newbcast:n source updates 16 rows x 1 column in each of the wave's four DPP rows -- two columns of each system.
//   * the pivot column exists in one DPP row only (the owner, h_p = (p & 15) >> 3): its scaled entries -- the row factors -- reach
//     the other row with gfx950's v_permlane16_swap_b32 (one instruction moves a dword between the rows of a pair, both ways)
//     and a row_ror:8 under a row mask that realigns them to the other row's rotation.
//   * the pivot's rsqrt chain is issued once per column for BOTH systems of the wave; no barrier, no slab, no panel image.
// Bordered form (the seven right-hand-side rows are rows RHS0 .. RHS0+6 of the last row block, k + 8 <= 16 NBR): they ride
// in lanes that a matrix of k rows leaves idle, and the Schur complement B'C^-1B is what remains in the trailing 7x7 block.
#pragma once
#include "twx_ukw.h"

#ifndef TWX_UKP_WV
#define TWX_UKP_WV 2, 3         // NBR = 4, 3
#endif
__host__ __device__ constexpr int twx_ukp_waves(int nbr)
{
    constexpr int w[2] = {TWX_UKP_WV};
    return w[4 - nbr];
}

// both dwords of a double through v_permlane16_swap_b32 with the same value on both sides: ev = the even DPP rows' values in
// both rows of each pair, od = the odd rows' values in both
__device__ __forceinline__ void pair_swap(double v, double &ev, double &od)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const u2 a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u2 b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    ev = __hiloint2double((int)b.x, (int)a.x);
    od = __hiloint2double((int)b.y, (int)a.y);
}

// rotate a double by eight lanes inside the DPP rows selected by MASK (bit r = DPP row r of the wave); the others keep theirs
template <int MASK>
__device__ __forceinline__ double ror8_rows(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128 /* row_ror:8 */, MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// lane N of each 16-lane row to all of the row
template <int N>
__device__ __forceinline__ double row_bcast(double v)
{
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
    return r;
}

__device__ __forceinline__ constexpr int pidx(int a, int b, int n) { return 8 * (a * (a + 1) / 2 + b) + n; }

template <int NBR>
__global__ __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(twx_ukp_waves(NBR), twx_ukp_waves(NBR))))
void k_ukp(StnDev st, CellSrc src, SelWs ws, const int32_t *item_list, const int32_t *nitems_dev)
{
    constexpr int NP = NBR * 16, NT = 4 * NBR * (NBR + 1);
    constexpr int R0 = 9, RHS0 = 16 * (NBR - 1) + R0;       // first RHS row / column
    __shared__ double s_B[2][7][NP];

    const int lane = threadIdx.x, t = lane & 15, h = (lane >> 4) & 1, l32 = lane & 31, sys = lane >> 5;
    const int rho = (t + 8 * h) & 15;                        // this lane's row of every 16-row block
    const int nitems = *nitems_dev;
    const int pr = xcd_contig(blockIdx.x, (nitems + 1) >> 1);
    if (pr < 0) return;
    const bool active = 2 * pr + sys < nitems;               // (an odd list: the last wave's second half repeats the first)
    const int item = item_list[min(2 * pr + sys, nitems - 1)];
    const int lc = item / 12;
    const int m0 = item - 12 * lc;
    const int k = ws.kk[(int64_t)lc * 12 + m0];
    const size_t n = (size_t)st.n;
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
    const int kmx = max(__builtin_amdgcn_readlane(k, 0), __builtin_amdgcn_readlane(k, 32));   // columns the wave eliminates

    int jq[2];
    float h0q[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = min(l32 + 32 * u, ws.ksel - 1);
        jq[u] = __hip_atomic_load(&ws.near_idx[(int64_t)lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        h0q[u] = __hip_atomic_load(&ws.h0[(int64_t)lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    // the pair distances of this lane's elements: block (a, b) of the cell's cache, element [column 8 h + n][row rho]
    float hd[NT];
    {
        const float *dist = ws.dist + lc * (int64_t)(TWX_DIST_BLOCKS * 256) + (8 * h * 16 + rho);
        sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, a + 1>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                sfor<0, 8>([&](auto n_) __attribute__((always_inline)) {
                    constexpr int nn = decltype(n_)::value;
                    hd[pidx(a, b, nn)] = __builtin_nontemporal_load(&dist[tri(a, b) * 256 + nn * 16]);
                });
            });
        });
    }

    // ---- staging: neighbours l32, l32 + 32 of this half's system (NP <= 64); trend columns shifted to the cell (see k_uk)
    double (*sB)[NP] = s_B[sys];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int q = l32 + 32 * u;
        double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0, yv = 0.0, c0v = 0.0;
        if (q < k) {
            const int j = jq[u];
            const double4 sr = st.stat_s[j];
            const double2 mr = st.mon_s[(size_t)j * 12 + m0];
            const double lo = sr.x, la = sr.y;
            x0 = lo - cv.lon; x1 = la - cv.lat; x2 = sr.z - cv.elev; x3 = mr.x - plst;
            yv = mr.y;
            const float h0 = h0q[u];
            const bool same = (lo == cv.lon && la == cv.lat) || h0 == 0.f;
            c0v = same ? c00 : (double)cov_exp2(h0, chi, lgp);
        }
        if (q < NP) {
            sB[0][q] = q < k ? 1.0 : 0.0;
            sB[1][q] = x0; sB[2][q] = x1; sB[3][q] = x2; sB[4][q] = x3;
            sB[5][q] = yv; sB[6][q] = c0v;
        }
    }
    __syncthreads();

    // ---- build (negated: the registers hold N = -M) ----------------------------------------------------------------------
    double A[NT];
    const bool rhs_row = rho >= R0;                           // of the last row block: rows RHS0 .. RHS0+6
    const double *rhs = &sB[rhs_row ? rho - R0 : 0][8 * h];   // (s_B is 0 from column k on)
    sfor<0, NBR>([&](auto a_) __attribute__((always_inline)) {
        constexpr int a = decltype(a_)::value;
        const int i = 16 * a + rho;
        const float ca = i < k ? chi : -__builtin_inff();
        sfor<0, a + 1>([&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value;
            sfor<0, 8>([&](auto n_) __attribute__((always_inline)) {
                constexpr int nn = decltype(n_)::value;
                constexpr int e = pidx(a, b, nn);
                const int j = 16 * b + 8 * h + nn;
                // (the last block row may lie outside what k_cell_dist has written: stale memory is selected away there)
                double v = (double)(a == NBR - 1 ? (i < k ? cov_exp2(hd[e], chi, lgp) : 0.f) : cov_exp2(hd[e], ca, lgp));
                // rows / columns k .. RHS0-1 are padding: an identity block (pivot 1, factors 0)
                if (b == a && i == j) v = i < k ? c00 : (i < RHS0 ? 1.0 : 0.0);
                if (a == NBR - 1) v = rhs_row ? rhs[16 * b + nn] : v;
                A[e] = -v;
            });
        });
    });

    // ---- elimination, one column at a time -------------------------------------------------------------------------------
    double nmax = -1.0;                                      // -(smallest pivot)
    sfor<0, RHS0>([&](auto p_) __attribute__((always_inline)) {
        constexpr int p = decltype(p_)::value;
        constexpr int bp = p / 16, hp = (p % 16) / 8, np = p % 8;
        if (p < kmx) {                                       // uniform: one of the two systems still has a C column here
            // the pivot: lane np of the owner row -> the owner row -> both rows of the system
            double ev, od;
            pair_swap(row_bcast<np>(A[pidx(bp, bp, np)]), ev, od);
            const double nd = hp ? od : ev;
            nmax = max_raw(nmax, nd);
            const double r = -rsqrt_nr(-nd);
            // row factors l(i, p) = n(i, p) r of the rows from the pivot's block on: formed in the owner row, handed to the
            // other row (pair_swap) in the owner's lane order, realigned there (ror8 under the other rows' mask)
            double L[NBR];
            sfor<bp, NBR>([&](auto a_) __attribute__((always_inline)) {
                constexpr int a = decltype(a_)::value;
                double e2, o2;
                pair_swap(A[pidx(a, bp, np)] * r, e2, o2);
                L[a] = ror8_rows<hp ? 0x5 : 0xA>(hp ? o2 : e2);
            });
            // rank-1 update of every column right of p: column 16 b + 8 h + n in DPP row h (the instruction is issued when
            // the h = 1 column is live; a finished h = 0 column then takes a harmless update)
            sfor<bp, NBR>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                sfor<0, 8>([&](auto n_) __attribute__((always_inline)) {
                    constexpr int nn = decltype(n_)::value;
                    if constexpr (16 * b + 8 + nn > p) {
                        sfor<b, NBR>([&](auto a_) __attribute__((always_inline)) {
                            constexpr int a = decltype(a_)::value;
                            fmac_rowbcast<nn>(A[pidx(a, b, nn)], L[b], L[a]);
                        });
                    }
                });
            });
        }
    });

    // ---- Schur complement out (k_uk_solve finishes): rows / columns RHS0 .. RHS0+6 = DPP row 1, lanes 1 .. 7, n = 1 .. 7 ----
    double *Sout = ws.uk_S + ((int64_t)lc * 12 + m0) * TWX_UK_SLEN;
    if (active && h == 1 && rho >= R0) {
        const int r = rho - R0;
        sfor<0, 7>([&](auto c_) __attribute__((always_inline)) {
            constexpr int cq = decltype(c_)::value;
            if (r >= cq) Sout[r * (r + 1) / 2 + cq] = A[pidx(NBR - 1, NBR - 1, 1 + cq)];
        });
    }
    if (active && l32 == 0) Sout[28] = (-nmax > 1e-9 * c00 && k <= kdup) ? 0.0 : 1.0;   // singular / indefinite
}
