"""CPU: the register / scratch / LDS budget of the built kernels (topowx_amd/libtwxhip.resources.txt, written by build.sh
from hipcc's -Rpass-analysis=kernel-resource-usage remarks).  The kriging and daily kernels are tuned to a number of
resident waves per SIMD (amdgpu_waves_per_eu in twx_uk.h / twx_ukw.h, the LDS footprint of k_daily_tile): a compiler
bump that adds registers or starts spilling would cost 10-20 % without any test noticing.  This one notices."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
RES = os.path.join(ROOT, "topowx_amd", "libtwxhip.resources.txt")

# kernel -> (resident waves per SIMD the tuning assumes, scratch bytes per lane tolerated)
# gfx950: 512 VGPRs per SIMD lane, allocated in granules of 8: waves w fit when VGPRs + AGPRs <= (512 / w) & ~7.
EXPECT = {
    "k_ukw<4, 0>": (4, 0), "k_ukw<5, 0>": (3, 0), "k_ukw<6, 0>": (2, 0),
    "k_ukw2<3, 0>": (3, 0), "k_ukwz<3, 0>": (4, 0), "k_ukwz<4, 0>": (3, 0), "k_ukwz<5, 0>": (2, 0), "k_ukwz<6, 0>": (2, 0),
    "k_uk<7, 2, 0>": (3, 0), "k_uk<8, 4, 0>": (4, 0), "k_uk<9, 2, 0>": (2, 0),
    # 160 rows on two waves: 220 VGPRs of matrix alone; measured faster with 4 systems per CU and a few spilled
    # registers of the build phase than with 3 (profiles/README.md, round 2): a bounded allowance, not a free pass
    "k_uk<10, 2, 0>": (2, 128),
    # fp64-build variants (ill-conditioned systems only): the SAME residency as the fast kernels of their size -- the inline fp64
    # exponential costs a few spilled registers of the build phase in four of them (bounded allowances); <.., 2>: call frames
    # of the out-of-line covariance function
    "k_ukw2<3, 1>": (3, 0), "k_ukwz<3, 1>": (4, 0), "k_ukw<4, 1>": (4, 32), "k_ukwz<4, 1>": (3, 0), "k_ukw<5, 1>": (3, 32),
    "k_ukwz<5, 1>": (2, 0), "k_ukw<6, 1>": (2, 0), "k_ukwz<6, 1>": (2, 32),
    "k_uk<7, 2, 1>": (3, 0), "k_uk<8, 4, 1>": (4, 0), "k_uk<9, 2, 1>": (2, 0), "k_uk<10, 2, 1>": (2, 160),
    "k_uk<7, 2, 2>": (1, 64), "k_uk<10, 2, 2>": (1, 64),
    "k_tile_dist": (4, 0), "k_stn_nn": (4, 0), "k_cell_dist": (4, 0), "k_uk_solve": (4, 0),
    "k_select<4, 0>": (4, 0), "k_select<1, 1>": (2, 0), "k_tile_cand": (4, 0),
    "k_gwr_z": (4, 0), "k_gwr_z_cell": (4, 0), "k_tile_uidx": (4, 0), "k_perm": (4, 0), "k_daily_tile": (6, 0), "k_daily_tile_gather": (4, 0),
    "k_daily_grid": (4, 0), "k_fix_cells": (4, 0), "k_fix_sparse": (2, 0),
}
LDS_PER_CU = 160 * 1024


def budget(waves):
    return (512 // waves) & ~7


@pytest.fixture(scope="module")
def table():
    """Never builds: a rebuild in the middle of a pytest session would replace a library other tests have dlopen'ed
    (and tests/tools/ab_stats.sh swaps A/B builds in on purpose).  Both files are git-ignored build products."""
    import isa_resources
    so = os.path.join(ROOT, "topowx_amd", "libtwxhip.so")
    if not os.path.exists(RES) or not os.path.exists(so):
        pytest.skip("no build in this checkout (run ./build.sh: it writes libtwxhip.so and libtwxhip.resources.txt)")
    if os.path.getmtime(RES) < os.path.getmtime(so) - 120:
        pytest.fail("libtwxhip.resources.txt is older than libtwxhip.so: the library was not built by ./build.sh "
                    "(an A/B copy?) -- run ./build.sh")
    return isa_resources.parse(RES)


def test_every_tuned_kernel_is_listed(table):
    missing = [k for k in EXPECT if k not in table]
    assert not missing, missing


@pytest.mark.parametrize("kernel", sorted(EXPECT))
def test_register_and_scratch_budget(table, kernel):
    waves, scratch_ok = EXPECT[kernel]
    r = table[kernel]
    assert r["scratch"] <= scratch_ok, "%s: %d B/lane of scratch (%d spilled VGPRs)" % (kernel, r["scratch"], r["vgpr_spill"])
    assert r["vgprs"] + r["agprs"] <= budget(waves), "%s: %d VGPRs + %d AGPRs > %d (%d waves per SIMD)" % (
        kernel, r["vgprs"], r["agprs"], budget(waves), waves)
    assert r["occupancy"] >= waves, "%s: occupancy %d < %d waves per SIMD" % (kernel, r["occupancy"], waves)


def test_daily_tile_keeps_three_workgroups_per_cu(table):
    # k_daily_tile stages a tile-month's rows in LDS; its tuning (8 waves x 3 work-groups per CU: six waves per SIMD, 80 VGPRs)
    # needs <= 53.3 KB each
    assert table["k_daily_tile"]["lds"] <= LDS_PER_CU // 3
    # the pair table of k_tile_dist takes nearly all of a CU's LDS by design (one work-group per CU)
    assert table["k_tile_dist"]["lds"] <= LDS_PER_CU


def test_no_kernel_uses_dynamic_scratch_unexpectedly(table):
    spilled = {k: r["scratch"] for k, r in table.items() if r["scratch"] > 0}
    assert set(spilled) <= {"k_uk<10, 2, 0>", "k_uk<10, 2, 1>", "k_uk<7, 2, 2>", "k_uk<10, 2, 2>", "k_ukw<4, 1>", "k_ukw<5, 1>",
                            "k_ukwz<6, 1>",
                            "k_deflate_table"}, spilled       # (one thread per variable and tile: the 19-symbol alphabet's small arrays)
