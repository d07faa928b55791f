// EXPERIMENT RECORD (round 5) -- NOT part of the library, not compiled by build.sh.
// k_ukz<NB, 2>: the two-wave elimination with the seven border rows held as COLUMNS, for systems whose k sits in the upper half of a
// block row (105..112 and 121..128 neighbours: the ladder spikes k = 111 / 122 are of that form), so that they run in 112 / 128 rows
// instead of the bordered kernel one size up.  Measured on the MI355X in same-box A/Bs of the C2 step (rocprofv3 averages, round 5):
// kriging +0.18 ms -- the replicated border registers cost NB = 7 its third wave per SIMD, and every wave repeats the border update.
// It lived in the product headers behind -DTWX_UKZ=1 until round 6.  To try it again: append this file's kernel to
// topowx_amd/csrc/twx_uk.h, give twx_krig_bucket() (twx_select.h) the buckets 9 (105..112) and 11 (121..128)
// (`return b < 13 ? b : 13;`), and launch k_ukz<7, 2> / k_ukz<8, 2> (and their <.., 1> fp64-build instances) for those buckets from
// run_uk_stage (twx_hip.hip) with 128 threads per work-group, as launch_uk does for k_uk.
#pragma once
// ---------------------------------------------------------------------------------
// k_ukz<NB, NW, PREC>: the multi-wave elimination with the seven border rows held as COLUMNS (the one-wave k_ukwz's
// scheme, twx_ukw.h, on k_uk's two-wave layout).
//
// The bordered form needs k + 8 <= 16 NB rows: a system of 105..112 neighbours runs in the 128-row kernel, one of 121..128
// in the 144-row kernel -- a whole block row (and block column) of padding more than its covariance matrix has.  Here the
// matrix holds the C rows only (NP = 16 NB >= k, identity padding) and the border B (7 x k) lives transposed in two
// registers per block row: lane (tr, tcl) of EVERY wave holds Z[16a + tr][tcl] and Z[16a + tr][4 + tcl] (column 7 is a zero
// dummy) -- replicated in the NW waves, because the wave that factorises a panel also row-solves the panel's four border
// columns (published as rows NP .. NP+7 of the panel image) and must hold them whichever wave it is; the redundant update
// is 8 fmacs per block row and panel and wave.  The 7 x 7 corner B'C^-1B is accumulated from the solved border factors by
// the wave of column group 0.  Register cost: 4 NB VGPRs over k_uk -- NB = 7 no longer fits three waves per SIMD, so these
// kernels run two waves per SIMD (4 systems per CU, what the kernels they relieve -- k_uk<8, 4>, k_uk<9, 2> -- have too).
// ---------------------------------------------------------------------------------
#ifndef TWX_UKZ_WV
#define TWX_UKZ_WV 2, 2               // NB = 8, 7
#endif
__host__ __device__ constexpr int twx_ukz_waves(int nb)
{
    constexpr int w[2] = {TWX_UKZ_WV};
    return w[8 - nb];
}

template <int NB, int NW, int PREC = 0>
__global__ __launch_bounds__(64 * NW)
__attribute__((amdgpu_waves_per_eu(twx_ukz_waves(NB), twx_ukz_waves(NB))))
void k_ukz(StnDev st, CellSrc src, SelWs ws, const int32_t *item_list, const int32_t *nitems_dev)
{
    static_assert(PREC == 0 || PREC == 1, "the slab-less fp64 build (PREC = 2) exists for k_uk only");
    constexpr int NP = NB * 16, CB = 4 * NW, NBC = NP / CB, NT = uk_eidx<NW>(NB, 0), NTH = 64 * NW;
    constexpr int NPX = NP + 8;     // rows of the panel image / slab: C rows + 7 border rows (+ 1 dummy)
    constexpr int PS = 6;
    constexpr int RPT = (NP + NTH - 1) / NTH;
    static_assert(RPT <= 2, "at most two neighbours per thread in the staging");
    __shared__ __attribute__((aligned(16))) double s_pan[2][NPX * PS];
    __shared__ __attribute__((aligned(16))) double s_raw[4 * NPX];
    __shared__ double s_B[7][NP];
    __shared__ int s_err;

    const int t = threadIdx.x, tr = t & 15, lane = t & 63, tcl = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int it = xcd_contig(blockIdx.x, *nitems_dev);
    if (it < 0) return;
    const int rot = (int)(((unsigned)it * 2654435761u) >> 13) & (NW - 1);
    const int wvp = (wv + rot) & (NW - 1);       // column group of this wave
    const int tc = 4 * wvp + tcl;
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
    const int kdup = ws.cdup[lc];

    int jq[RPT];
    float h0q[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int q = min(t + NTH * u, ws.ksel - 1);
        jq[u] = __hip_atomic_load(&ws.near_idx[lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        h0q[u] = __hip_atomic_load(&ws.h0[lc * ws.ksel + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    float hd[PREC ? 1 : NT];
    const double ninv = rng == 0.0 ? 0.0 : -1.0 / rng;       // (PREC)
    const int64_t fs = PREC ? (int64_t)ws.cellf64[lc] - 1 : 0;
    if constexpr (!PREC) {
        const float *dist = ws.dist + lc * (int64_t)(TWX_DIST_BLOCKS * 256) + (tc * 16 + tr);
        sfor<0, NB>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            sfor<0, uk_nbc<NW>(a)>([&](auto b_) __attribute__((always_inline)) {
                constexpr int b = decltype(b_)::value;
                constexpr int j0 = CB * b;
                hd[uk_eidx<NW>(a, b)] = __builtin_nontemporal_load(&dist[tri(a, j0 / 16) * 256 + (j0 % 16) * 16]);
            });
        });
    }

    // ---- staging (see k_uk): trend columns shifted to the cell, not scaled ----------------------------------------
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int q = t + NTH * u;
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
            if constexpr (PREC) c0v = same ? c00 : psill_e * exp_neg_f64(ws.h064[fs * ws.ksel + q] * ninv);
            else c0v = same ? c00 : (double)cov_exp2(h0, chi, lgp);
        }
        if (q < NP) {
            s_B[0][q] = q < k ? 1.0 : 0.0;
            s_B[1][q] = x0; s_B[2][q] = x1; s_B[3][q] = x2; s_B[4][q] = x3;
            s_B[5][q] = yv; s_B[6][q] = c0v;
        }
    }
    if (t == 0) s_err = 0;
    for (int q = t; q < 2 * NPX * PS; q += NTH) (&s_pan[0][0])[q] = 0.0;   // finished rows are never written: keep them finite
    __syncthreads();

    // ---- build (negated): C with an identity block in rows / columns k .. NP-1; the border transposed, in every wave:
    //      Z[a][h] = -B[tcl + 4h][16a + tr] (0 for the dummy column 7 and from row k on)
    double A[NT], Z[NB][2], Sacc[2] = {0.0, 0.0};
    sfor<0, NB>([&](auto a_) __attribute__((always_inline)) {
        constexpr int a = decltype(a_)::value;
        const int i = 16 * a + tr;
        const float ca = i < k ? chi : -__builtin_inff();
        sfor<0, uk_nbc<NW>(a)>([&](auto b_) __attribute__((always_inline)) {
            constexpr int b = decltype(b_)::value;
            constexpr int e = uk_eidx<NW>(a, b);
            const int j = CB * b + tc;
            double v;
            if constexpr (PREC) {
                constexpr int j0 = CB * b;
                v = (i < k && j < i) ? psill_e * exp_neg_f64(ws.dist64[fs * (int64_t)(TWX_DIST_BLOCKS * 256) + tri(a, j0 / 16) * 256 +
                                                                      ((j0 % 16) + (tc % 16)) * 16 + tr] * ninv) : 0.0;
            } else v = (double)(a == NB - 1 ? (i < k ? cov_exp2(hd[e], chi, lgp) : 0.f) : cov_exp2(hd[e], ca, lgp));
            if (CB * b / 16 == a && i == j) v = i < k ? c00 : 1.0;
            A[e] = -v;
        });
        Z[a][0] = -s_B[tcl][i];
        Z[a][1] = tcl < 3 ? -s_B[4 + (tcl < 3 ? tcl : 0)][i] : 0.0;
    });
    // ---- elimination: panels of four columns, NW per block column ------------------------------------------------
    int pbuf = 0;
    double nmax = -1.0;
    sfor<0, NBC>([&](auto bc_) __attribute__((always_inline)) {
        constexpr int bc = decltype(bc_)::value;
        constexpr int a0 = CB * bc / 16;                     // the block row holding rows / columns CB bc .. CB bc + CB - 1
        const int ncb = k - CB * bc;                         // C columns left
        if (ncb > 0) {
            const int npan = min(NW, (ncb + 3) >> 2);
#pragma nounroll
            for (int s = 0; s < npan; ++s) {
                if (wvp == s) {
                    // (1) publish the panel: its C rows, and its four border columns (rows CB bc + 4s .. + 3 of Z) as rows
                    //     NP .. NP+7 of the image
                    sfor<a0, NB>([&](auto a_) __attribute__((always_inline)) {
                        constexpr int a = decltype(a_)::value;
                        s_raw[tcl * NPX + 16 * a + tr] = A[uk_eidx<NW>(a, bc)];
                    });
                    if ((tr >> 2) == (((CB * bc) >> 2) + s & 3)) {
                        s_raw[(tr & 3) * NPX + NP + tcl] = Z[a0][0];
                        s_raw[(tr & 3) * NPX + NP + 4 + tcl] = Z[a0][1];
                    }
                    __builtin_amdgcn_wave_barrier();
                    // (2) the 4x4 diagonal block, its factor, the row solves (see k_uk)
                    const double *dg = &s_raw[CB * bc + 4 * s];
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
                    constexpr int NROW = NPX - CB * bc, RPR = 64;
#pragma unroll
                    for (int u = 0; u < (NROW + RPR - 1) / RPR; ++u) {
                        const int rr = 4 * s + 4 + lane + RPR * u;           // row counted from the block column's first
                        if (rr < NROW) {
                            const int myrow = CB * bc + rr;
                            const double2 n01 = double2{s_raw[myrow], s_raw[NPX + myrow]};
                            const double2 n23 = double2{s_raw[2 * NPX + myrow], s_raw[3 * NPX + myrow]};
                            const double L0 = n01.x * r0;
                            const double L1 = fma(L0, l10, n01.y) * r1;
                            const double L2 = fma(L1, l21, fma(L0, l20, n23.x)) * r2;
                            const double L3 = fma(L2, l32, fma(L1, l31, fma(L0, l30, n23.y))) * r3;
                            *reinterpret_cast<double2 *>(&s_pan[pbuf][myrow * PS]) = double2{L0, L1};
                            *reinterpret_cast<double2 *>(&s_pan[pbuf][myrow * PS + 2]) = double2{L2, L3};
                        }
                    }
                }
                __syncthreads();
                // (3) rank-4 update: C elements right of the panel, the border entries of the live block rows, the corner
                const double *pan = s_pan[pbuf];
                pbuf ^= 1;
                constexpr int NE = 4 * (NBC - bc), NR = (NE + 15) / 16;
                const bool own_live = wvp > s;
                // the panel factors of the border rows tcl and 4 + tcl (lane n of a DPP row holds factor n & 3)
                const double pz0 = pan[(NP + tcl) * PS + (tr & 3)], pz1 = pan[(NP + 4 + tcl) * PS + (tr & 3)];
                if (wvp == 0) {
                    const double2 c0 = *reinterpret_cast<const double2 *>(&pan[(NP + (tr & 7)) * PS]);
                    const double2 c1 = *reinterpret_cast<const double2 *>(&pan[(NP + (tr & 7)) * PS + 2]);
                    fmac_rowbcast<0>(Sacc[0], pz0, c0.x); fmac_rowbcast<1>(Sacc[0], pz0, c0.y);
                    fmac_rowbcast<2>(Sacc[0], pz0, c1.x); fmac_rowbcast<3>(Sacc[0], pz0, c1.y);
                    fmac_rowbcast<0>(Sacc[1], pz1, c0.x); fmac_rowbcast<1>(Sacc[1], pz1, c0.y);
                    fmac_rowbcast<2>(Sacc[1], pz1, c1.x); fmac_rowbcast<3>(Sacc[1], pz1, c1.y);
                }
                double P[NR];
                sfor<0, NR>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    const int e = min(16 * r + tr, NE - 1);
                    P[r] = pan[(CB * (bc + (e >> 2)) + tc) * PS + (e & 3)];
                });
                sfor<a0, NB>([&](auto a_) __attribute__((always_inline)) {
                    constexpr int a = decltype(a_)::value;
                    const double2 u0 = *reinterpret_cast<const double2 *>(&pan[(16 * a + tr) * PS]);
                    const double2 u1 = *reinterpret_cast<const double2 *>(&pan[(16 * a + tr) * PS + 2]);
                    fmac_rowbcast<0>(Z[a][0], pz0, u0.x); fmac_rowbcast<1>(Z[a][0], pz0, u0.y);
                    fmac_rowbcast<2>(Z[a][0], pz0, u1.x); fmac_rowbcast<3>(Z[a][0], pz0, u1.y);
                    fmac_rowbcast<0>(Z[a][1], pz1, u0.x); fmac_rowbcast<1>(Z[a][1], pz1, u0.y);
                    fmac_rowbcast<2>(Z[a][1], pz1, u1.x); fmac_rowbcast<3>(Z[a][1], pz1, u1.y);
                    if (own_live) {
                        double acc = A[uk_eidx<NW>(a, bc)];
                        fmac_rowbcast<0>(acc, P[0], u0.x);
                        fmac_rowbcast<1>(acc, P[0], u0.y);
                        fmac_rowbcast<2>(acc, P[0], u1.x);
                        fmac_rowbcast<3>(acc, P[0], u1.y);
                        A[uk_eidx<NW>(a, bc)] = acc;
                    }
                    sfor<bc + 1, uk_nbc<NW>(a)>([&](auto b_) __attribute__((always_inline)) {
                        constexpr int b = decltype(b_)::value;
                        constexpr int e = 4 * (b - bc);
                        double acc = A[uk_eidx<NW>(a, b)];
                        fmac_rowbcast<(e + 0) % 16>(acc, P[(e + 0) / 16], u0.x);
                        fmac_rowbcast<(e + 1) % 16>(acc, P[(e + 1) / 16], u0.y);
                        fmac_rowbcast<(e + 2) % 16>(acc, P[(e + 2) / 16], u1.x);
                        fmac_rowbcast<(e + 3) % 16>(acc, P[(e + 3) / 16], u1.y);
                        A[uk_eidx<NW>(a, b)] = acc;
                    });
                });
            }
        }
    });
    if (!(-nmax > 1e-9 * c00) || k > kdup) s_err = 1;        // singular / indefinite system (benign race: all write 1)

    // ---- Schur complement out (k_uk_solve finishes): in the wave of column group 0, lane (tr = c', tcl) holds S[tcl + 4h][c']
    if (wvp == 0 && tr < 7) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = tcl + 4 * h;
            if (r < 7 && tr <= r) ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + r * (r + 1) / 2 + tr] = Sacc[h];
        }
    }
    __syncthreads();
    if (t == 0) ws.uk_S[(lc * 12 + m0) * TWX_UK_SLEN + 28] = s_err ? 1.0 : 0.0;
}
