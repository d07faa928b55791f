// EXPERIMENT RECORD (round 5) -- not compiled, not part of the library.
//
// k_uk's elimination with the NEXT panel's factorisation inside the current panel's update period ("look-ahead with a
// balanced split"): the wave that holds the next panel's columns applies the current panel to those columns, factorises,
// applies the current panel to its upper ~0.8 of the block rows and goes to the barrier; the lower block rows it owes are
// applied at the start of the next period.  Every element still receives its panels in order: the outputs are BIT-EQUAL to the
// library's (tests/tools/ab_bits.sh: grid, points on every kernel-size boundary, TWX_FLAG_UK_F64_ALL, TWX_FLAG_NO_HOST_SYNC).
// s_memtime stamps of the library's kernels (clocks per panel period: update of the slowest wave U / publish + chain + row
// solve C / period): k_uk<7,2> 1036 / 806 / 2242, <8,4> 1034 / 866 / 2155, <9,2> 1322 / 836 / 2542, <10,2> 1824 / 910 / 3602 --
// the period is U + C + barrier, and the balanced split promises U + C / 2.
// MEASURED (one gpurun call each, C2 bench): slower in every form --
//   run-time row range + flags around every block (one instance of the update):  kriging 9.94 -> 12.7 ms
//   the same with one branch per block row:                                     9.86 -> 12.6 ms
//   compile-time row ranges, separate instances (263-903 spilled VGPRs):        not timed
//   compile-time upper / lower regions, one instance each (this file):          9.8 -> 11.4 ms
//     (k_uk<8,4> 1627 -> 2157 us, <10,2> 1260 -> 1686, <7,2> 1296 -> 1655, <9,2> 836 -> 1044; stamps of <7,2>: period 3396,
//      a non-holder's owed rows + update 2852 clocks, the holder's column + factor + upper rows 1893)
// Why: what a wave waits at the barrier is not idle SIMD time -- three or four other systems' waves share the SIMD and the
// CU's one LDS pipe (0.83-0.93 of a SIMD's VALU time in the library's kernels) -- and the pipelined form ADDS work to both:
// the next panel's columns and the owed rows reload their row and column factors from LDS (+ ~30 % LDS reads), a third
// slab, 13-19 spilled registers in the 112-144-row kernels and 196 in the 160-row one, 1.35 x the code.
// The block below replaces the section "---- elimination" of k_uk (twx_uk.h) and needs s_pan[3][NP * PS] and
//   #define TWX_UK_DEFER_NUM 4 / TWX_UK_DEFER_DEN 5 / TWX_UK_DEFER_MIN 3.

    // ---- elimination: panels of four columns, NW per block column --------------------------------------
    // One barrier per panel, and what a period between two barriers holds is arranged so that the panel chain of the NEXT
    // panel -- publish, 4x4 factor, row solve: ~800-900 clocks of one wave, measured with s_memtime stamps -- runs beside the
    // update of THIS one instead of after it:
    //   the wave that holds the next panel's four columns applies this panel to those columns first (4 fmacs per block
    //   row), factorises, writes the next slab, then applies this panel to its upper block rows only and goes to the barrier;
    //   the lower block rows it still owes are applied at the start of the next period (where another wave is the holder),
    //   before that period's update -- every element receives its panels in order, so the results are the bits of the
    //   unpipelined form (rounds 1-4: update, then chain, the other waves waiting: period U + C; now ~ U + C / 2 with the
    //   split at ~0.8 of the block rows) --;
    //   every other wave applies the whole panel.
    // Three slabs: the panel being applied, the one being written, and the previous one (owed rows).
    double nmax = -1.0;                                      // -(smallest pivot this wave has factorised)
    // publish + chain + row solve of the panel at columns colbase .. colbase + 3 (this wave's registers regs(a)) into slab dst
    auto factor_panel = [&](auto a0_, auto get, const int colbase, double *dst) __attribute__((always_inline)) {
        constexpr int a0 = decltype(a0_)::value;
        sfor<a0, NB>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            s_raw[tcl * NP + 16 * a + tr] = get(a_);
        });
        __builtin_amdgcn_wave_barrier();
        // column-major panel image: s_raw[column][row] (publishing lanes write consecutive rows, the row solve
        // reads consecutive rows: no bank conflicts); the diagonal block comes back as broadcasts
        const double *dg = &s_raw[colbase];
        const double g00 = dg[0];
        const double2 g1 = double2{dg[1], dg[NP + 1]};
        const double2 g2 = double2{dg[2], dg[NP + 2]};
        const double g22 = dg[2 * NP + 2];
        const double2 g3 = double2{dg[3], dg[NP + 3]};
        const double2 g3b = double2{dg[2 * NP + 3], dg[3 * NP + 3]};
        // the registers hold N = -M: pivot d = -n, l = n * (-1/sqrt(d)), updates n += l l.  A non-positive pivot gives NaN
        // factors that reach the Schur block (k_uk_solve rejects non-finite results); too small a pivot is caught through nmax
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
        constexpr int NROWMAX = NP - 16 * a0, RPR = 64;      // (only the rows below the panel are needed)
#pragma unroll
        for (int u = 0; u < (NROWMAX + RPR - 1) / RPR; ++u) {
            const int myrow = colbase + 4 + lane + RPR * u;
            if (myrow < NP) {
                const double2 n01 = double2{s_raw[myrow], s_raw[NP + myrow]};
                const double2 n23 = double2{s_raw[2 * NP + myrow], s_raw[3 * NP + myrow]};
                const double L0 = n01.x * r0;
                const double L1 = fma(L0, l10, n01.y) * r1;
                const double L2 = fma(L1, l21, fma(L0, l20, n23.x)) * r2;
                const double L3 = fma(L2, l32, fma(L1, l31, fma(L0, l30, n23.y))) * r3;
                *reinterpret_cast<double2 *>(&dst[myrow * PS]) = double2{L0, L1};
                *reinterpret_cast<double2 *>(&dst[myrow * PS + 2]) = double2{L2, L3};
            }
        }
    };
    // rank-4 update N(i,j) += l(i,:) . l(j,:) with the panel of block column bc in slab pan, block rows ALO <= a < AHI
    // (compile-time: straight-line code, the row factors of the next block row load while this one's fmacs issue).  own / b1:
    // this wave's columns of the panel's own block column / of the one after it (wave-uniform: scalar branches around
    // one block each); the block columns beyond always.  The column factors l(CB b + tc, 0..3) are common to the 16 lanes of a
    // DPP row: lane n of the row loads entry e = 16r + n (e = 4(b-bc) + column) once per call and every fmac picks its
    // operand with row_newbcast; the row factors l(16a+tr, 0..3) are read once per block row.
    auto update = [&](auto bc_, auto alo_, auto ahi_, const double *pan, const bool own, const bool b1) __attribute__((always_inline)) {
        constexpr int bc = decltype(bc_)::value, ALO = decltype(alo_)::value, AHI = decltype(ahi_)::value;
        constexpr int NE = 4 * (NBC - bc), NR = (NE + 15) / 16;
        if constexpr (ALO < AHI) {
            double P[NR];
            sfor<0, NR>([&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                const int e = min(16 * r + tr, NE - 1);
                P[r] = pan[(CB * (bc + (e >> 2)) + tc) * PS + (e & 3)];
            });
            sfor<ALO, AHI>([&](auto a_) __attribute__((always_inline)) {
                constexpr int a = decltype(a_)::value;
                const double2 u0 = *reinterpret_cast<const double2 *>(&pan[(16 * a + tr) * PS]);
                const double2 u1 = *reinterpret_cast<const double2 *>(&pan[(16 * a + tr) * PS + 2]);
                auto block = [&](auto b_) __attribute__((always_inline)) {
                    constexpr int b = decltype(b_)::value;
                    constexpr int e = 4 * (b - bc);
                    double acc = A[uk_eidx<NW>(a, b)];
                    fmac_rowbcast<(e + 0) % 16>(acc, P[(e + 0) / 16], u0.x);
                    fmac_rowbcast<(e + 1) % 16>(acc, P[(e + 1) / 16], u0.y);
                    fmac_rowbcast<(e + 2) % 16>(acc, P[(e + 2) / 16], u1.x);
                    fmac_rowbcast<(e + 3) % 16>(acc, P[(e + 3) / 16], u1.y);
                    A[uk_eidx<NW>(a, b)] = acc;
                };
                if (own) block(bc_);
                if constexpr (bc + 1 < uk_nbc<NW>(a))
                    if (b1) block(std::integral_constant<int, bc + 1>{});
                sfor<bc + 2, uk_nbc<NW>(a)>(block);
            });
        }
    };
    // the same update for ONE block column bx (bc or bc + 1) only, every block row that reaches it
    auto update_column = [&](auto bc_, auto bx_, const double *pan) __attribute__((always_inline)) {
        constexpr int bc = decltype(bc_)::value, bx = decltype(bx_)::value, e = 4 * (bx - bc);
        const double p = pan[(CB * bx + tc) * PS + tr % 4];   // lane n of a row: entry n % 4 of this wave's column
        sfor<CB * bx / 16, NB>([&](auto a_) __attribute__((always_inline)) {
            constexpr int a = decltype(a_)::value;
            const double2 u0 = *reinterpret_cast<const double2 *>(&pan[(16 * a + tr) * PS]);
            const double2 u1 = *reinterpret_cast<const double2 *>(&pan[(16 * a + tr) * PS + 2]);
            double acc = A[uk_eidx<NW>(a, bx)];
            fmac_rowbcast<0>(acc, p, u0.x);
            fmac_rowbcast<1>(acc, p, u0.y);
            fmac_rowbcast<2>(acc, p, u1.x);
            fmac_rowbcast<3>(acc, p, u1.y);
            A[uk_eidx<NW>(a, bx)] = acc;
        });
        (void)e;
    };
    // first block row the holder of the next panel leaves for the next period (of the rows a0 .. NB - 1 of block column bc)
    constexpr auto split_row = [](int a0) constexpr {
        return (NB - a0 >= TWX_UK_DEFER_MIN) ? a0 + ((NB - a0) * TWX_UK_DEFER_NUM + TWX_UK_DEFER_DEN / 2) / TWX_UK_DEFER_DEN : NB;
    };
    int cur = 0;                                             // slab of the panel being applied; cur + 1: being written; cur + 2: the previous panel's
    int owed = 0;                                            // this wave owes the lower block rows of the previous panel: 1 same block column, 2 the one before
    if (wvp == 0 && k > 0)                                   // the first panel
        factor_panel(std::integral_constant<int, 0>{}, [&](auto a_) { return A[uk_eidx<NW>(decltype(a_)::value, 0)]; }, 0, s_pan[0]);
    __syncthreads();
    sfor<0, NBC>([&](auto bc_) __attribute__((always_inline)) {
        constexpr int bc = decltype(bc_)::value;
        constexpr int a0 = CB * bc / 16;                     // first block row that reaches this block column
        constexpr int asplit = split_row(a0);
        using A0 = std::integral_constant<int, a0>;
        using AS = std::integral_constant<int, asplit>;
        using AN = std::integral_constant<int, NB>;
        const int ncb = k - CB * bc;                         // C columns left
        if (ncb > 0) {
            const int npan = min(NW, (ncb + 3) >> 2);
            const bool more = bc + 1 < NBC && k - CB * (bc + 1) > 0;     // a panel in the next block column
#pragma nounroll
            for (int s = 0; s < npan; ++s) {
                TWX_STAMP(NW * bc + s, 0);
                const double *pan = s_pan[cur];
                double *nxt = s_pan[cur == 2 ? 0 : cur + 1];
                const double *old = s_pan[cur == 0 ? 2 : cur - 1];
                const bool next_here = s + 1 < npan, next_there = !next_here && more;
                const bool holds_next = (next_here && wvp == s + 1) || (next_there && wvp == 0);
                // In the panel's own block column only the waves holding columns right of the panel (wvp > s) still have
                // live elements: the others skip it (a scalar branch).
                const bool own_live = wvp > s;
                const bool owes_here = !holds_next && owed == 1;
                if (holds_next) {
                    // this panel on the next panel's columns (every block row), then the next panel's factor; below: this panel
                    // on the rest of the UPPER block rows only -- the lower ones wait for the next period
                    TWX_STAMP(NW * bc + s, 2);
                    if (next_here) {
                        update_column(bc_, bc_, pan);
                        factor_panel(A0{}, [&](auto a_) { return A[uk_eidx<NW>(decltype(a_)::value, bc)]; }, CB * bc + 4 * (s + 1), nxt);
                    } else if constexpr (bc + 1 < NBC) {
                        update_column(bc_, std::integral_constant<int, bc + 1>{}, pan);
                        factor_panel(std::integral_constant<int, CB * (bc + 1) / 16>{},
                                     [&](auto a_) { return A[uk_eidx<NW>(decltype(a_)::value, bc + 1)]; }, CB * (bc + 1), nxt);
                    }
                    TWX_STAMP(NW * bc + s, 3);
                } else if (owed == 2) {                      // owed rows of the previous block column's last panel: their own instance
                    if constexpr (bc > 0)
                        update(std::integral_constant<int, bc - 1>{}, std::integral_constant<int, split_row(CB * (bc - 1) / 16)>{}, AN{}, old, false, false);
                }
                // lower block rows: what this wave owes of the previous panel (a wave is never the holder twice in a row), then
                // this panel -- ONE instance of the code, twice through it at most
#pragma nounroll
                for (int it = 0; it < 2; ++it)
                    if (it == 0 ? owes_here : !holds_next)
                        update(bc_, AS{}, AN{}, it == 0 ? old : pan, it == 0 ? false : own_live, true);
                // upper block rows: every wave
                update(bc_, A0{}, AS{}, pan, holds_next ? false : own_live, holds_next ? next_here : true);
                owed = (holds_next && asplit < NB) ? (next_here ? 1 : 2) : 0;
                TWX_STAMP(NW * bc + s, 1);
                __syncthreads();
                cur = cur == 2 ? 0 : cur + 1;
            }
        }
    });
