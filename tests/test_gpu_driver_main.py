"""GPU: ``python -m topowx_amd.driver`` end to end on one rank -- tiles dealt, computed device-resident
(interp_tiles_device), gathered (gather_mosaic_device) -- equals the whole-grid call bit for bit, also when the tile
size does not divide the grid (edge tiles through a scratch image); the streamed writer refuses such a grid up front."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _driver(*args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run([sys.executable, "-m", "topowx_amd.driver", *args], capture_output=True, text=True, env=env,
                          timeout=600, cwd=ROOT)


@pytest.mark.parametrize("tile", [50, 30])
def test_driver_mosaic_equals_whole_grid(tmp_path, tile):
    from topowx_amd import _lib, synth
    out = str(tmp_path / "mosaic.npz")
    p = _driver("--config", "C1", "--tile", str(tile), "--gather", "--out", out)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.load(out)
    grid, tmin, tmax = synth.make_case("C1")
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
    ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
    want = ctx.interp_grid(grid)
    ctx.close()
    for k in ("norm_tmin", "se_tmin", "norm_tmax", "se_tmax"):
        assert np.array_equal(got[k], want[k]), k


def test_driver_streamed_tiles_and_refusal(tmp_path):
    d = str(tmp_path / "tiles")
    p = _driver("--config", "C1", "--tile", "30", "--tile-dir", d)
    assert p.returncode != 0 and "divides evenly" in (p.stderr + p.stdout)
    assert not os.path.exists(d) or not os.listdir(d)                      # nothing was computed and thrown away
    out = str(tmp_path / "m.npz")
    p = _driver("--config", "C1", "--tile", "50", "--tile-dir", d, "--gather", "--out", out)
    assert p.returncode == 0, p.stderr[-2000:]
    assert sorted(os.listdir(d)) == ["tile%05d.npz" % k for k in range(4)]
    t0 = np.load(os.path.join(d, "tile00000.npz"))
    m = np.load(out)
    assert np.array_equal(m["norm_tmin"][:, :50, :50], t0["norm_tmin"]) and np.array_equal(m["se_tmax"][:, :50, :50], t0["se_tmax"])


@pytest.mark.parametrize("deflate", [False, True])
def test_driver_writes_netcdf4_tile_files(tmp_path, deflate):
    """--nc-dir: every tile into the reference's NetCDF-4 tile files through ncio.TileSink; --deflate: the daily variables with
    shuffle + deflate, their chunk bytes formed on the GPU.  Files == the whole-grid call."""
    from topowx_amd import _lib, h5nc, ncio, synth
    if not h5nc.available():
        pytest.skip("libhdf5 not loadable")
    d = str(tmp_path / "nc")
    p = _driver("--config", "C1", "--tile", "50", "--daily", "--nc-dir", d, "--chunk", "25", "--precision", "fast", *(["--deflate"] if deflate else []))
    assert p.returncode == 0, p.stderr[-2000:]
    grid, tmin, tmax = synth.make_case("C1", with_obs=True)
    ctx = _lib.Context()
    ctx.set_stations(_lib.TMIN, tmin)
    ctx.set_stations(_lib.TMAX, tmax)
    want = ctx.interp_grid(grid, daily=True)
    ctx.close()
    tiles = sorted(os.listdir(d))
    assert len(tiles) == 4
    from topowx_amd.interp import Tiler
    info = Tiler(grid, 50, 50, 25, 25, process_tiles=()).build_tile_grid_info()
    for k in range(4):
        tid = info.get_tile_id(k)
        r0, c0 = info.tile_rc[tid] if hasattr(info, "tile_rc") else (50 * (k // 2), 50 * (k % 2))
        for var in ("tmin", "tmax"):
            t = ncio.read_tile(os.path.join(d, tid, "%s_%s.nc" % (tid, var)), var)
            assert np.array_equal(t["daily"], want["daily_" + var][:, r0:r0 + 50, c0:c0 + 50]), (tid, var)
            assert np.array_equal(t["norm"], want["norm_" + var][:, r0:r0 + 50, c0:c0 + 50])
        ds = h5nc.Dataset(os.path.join(d, tid, "%s_tmin.nc" % tid))
        assert ds.variables["tmin"].chunking() == [tmin.days.size, 25, 25] and ds.variables["tmin"].filters()["zlib"] == deflate
        ds.close()
