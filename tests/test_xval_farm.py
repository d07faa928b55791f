"""Config 5 host logic on CPU: the nstns optimisation reduction against the executed reference
(tests/golden/make_golden_xval.py -> optimize.py:268-374) and the station farm over two gloo ranks.

The farm's per-station compute is an oracle-backed stand-in here (tests may use the oracle as the checker); on the
GPU box the same farm code runs on the HIP classes (tests/test_gpu_xval.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cube_from_golden(g, stns):
    """[12, 16, n_xval] MAE cube + xval ids from the per-division cubes the reference's files would hold."""
    from topowx_amd import stationdb as sdb
    div = g["so_climdiv"]
    ids, cols = [], []
    for d in g["so_divs"]:
        members = np.nonzero(div == d)[0]                 # stnids_climdiv order = table order (step21:96-98)
        cube = g["so_mae_%d" % d]
        assert cube.shape == (12, 16, members.size)
        ids.extend(stns[sdb.STN_ID][members])
        cols.append(cube)
    return np.array(ids), np.concatenate(cols, axis=2)


def test_set_optim_nstns_matches_executed_reference(golden_xval, golden_case):
    from topowx_amd import stationdb as sdb, xval
    g = golden_xval
    _, tmin, _ = golden_case
    for namer, key, fn in ((sdb.get_optim_varname, "so_optim", xval.set_optim_nstns_tair_norm),
                           (sdb.get_optim_anom_varname, "so_optim_anom", xval.set_optim_nstns_tair_anom)):
        stns = tmin.stns.copy()
        stns[sdb.CLIMDIV] = g["so_climdiv"]
        for m in range(1, 13):
            stns[namer(m)] = g["so_fill"]                 # add_stn_variable(..., fill_value) (optimize.py:292-296)
        ids, mae = _cube_from_golden(g, stns)
        # shuffle the cross-validated stations: the reduction must not depend on their order
        perm = np.random.default_rng(1).permutation(ids.size)
        da = sdb.StationSerialDataDb(stns, "tmin", tmin.days)
        chosen = fn(da, ids[perm], mae[:, :, perm], g["xa_ladder"])
        got = np.stack([da.stns[namer(m)] for m in range(1, 13)])
        np.testing.assert_array_equal(got, g[key])                                   # bit-exact, incl. untouched fills
        assert sorted(chosen) == [101.0, 102.0, 4407.0]
        mm = g["so_mae_101"][4].mean(axis=1)
        assert mm[3] == mm[9]                                                        # the cube carries an exact tie
        assert chosen[101.0][4] == g["xa_ladder"][int(np.argmin(mm))]
    # a failed station (NaN column) is left out of the division's mean, as the masked fill value is
    stns = tmin.stns.copy()
    stns[sdb.CLIMDIV] = g["so_climdiv"]
    ids, mae = _cube_from_golden(g, stns)
    mae2 = np.concatenate([mae, np.full((12, 16, 1), np.nan)], axis=2)
    extra = stns[sdb.STN_ID][np.nonzero(g["so_climdiv"] == 101)[0][0]]
    a = xval.set_optim_nstns(stns.copy(), ids, mae, g["xa_ladder"], sdb.get_optim_varname)[1]
    b = xval.set_optim_nstns(stns.copy(), np.append(ids, extra), mae2, g["xa_ladder"], sdb.get_optim_varname)[1]
    for d in a:
        np.testing.assert_array_equal(a[d], b[d])



def test_set_optim_nstns_from_mae_files_matches_executed_reference(golden_xval, golden_case, tmp_path):
    """The route the reference takes (optimize.py:285-316): per-division MAE files written by the step21 / step23
    writer rank (step21:83-128), read back by set_optim_nstns_tair_*.  Files are NetCDF-3 here (ncio); the result is
    bit-identical to the executed reference, incl. a never-written (fill-valued) station."""
    from topowx_amd import ncio, stationdb as sdb, xval
    g = golden_xval
    _, tmin, _ = golden_case
    for namer, key in ((sdb.get_optim_varname, "so_optim"), (sdb.get_optim_anom_varname, "so_optim_anom")):
        stns = tmin.stns.copy()
        stns[sdb.CLIMDIV] = g["so_climdiv"]
        for m in range(1, 13):
            stns[namer(m)] = g["so_fill"]
        ids, mae = _cube_from_golden(g, stns)
        da = sdb.StationSerialDataDb(stns, "tmin", tmin.days)
        d = str(tmp_path / key)
        paths = xval.write_optim_nstns_files(d, da, ids, mae, g["xa_ladder"])
        assert sorted(os.path.basename(p) for p in paths) == ["optim_nstns_tmin_climdiv%d.nc" % c for c in (101, 102, 4407)]
        cube, nghs, fids = ncio.read_climdiv_optim_nstns_db(paths[0])
        np.testing.assert_array_equal(nghs, g["xa_ladder"])
        sel = [int(np.nonzero(ids == s)[0][0]) for s in fids]
        np.testing.assert_array_equal(cube, mae[:, :, sel])
        chosen = xval.set_optim_nstns_from_files(da, d, namer)
        got = np.stack([da.stns[namer(m)] for m in range(1, 13)])
        np.testing.assert_array_equal(got, g[key])
        assert sorted(chosen) == [101.0, 102.0, 4407.0]
    # a station whose slot was never written holds the fill value and is masked out of the mean
    mae2 = mae.copy()
    mae2[:, :, 0] = np.nan
    d2 = str(tmp_path / "nan")
    xval.write_optim_nstns_files(d2, da, ids, mae2, g["xa_ladder"])
    stns_a, stns_b = stns.copy(), stns.copy()
    a = xval.set_optim_nstns(stns_a, ids, mae2, g["xa_ladder"], sdb.get_optim_varname)[1]
    b = xval.set_optim_nstns_from_files(sdb.StationSerialDataDb(stns_b, "tmin", tmin.days), d2, sdb.get_optim_varname)
    for dv in a:
        np.testing.assert_array_equal(a[dv], b[dv])
    # a division whose file is missing: the reference fails opening it (optimize.py:302-304) -- so does this, by default
    os.remove(os.path.join(d2, "optim_nstns_tmin_climdiv102.nc"))
    da3 = sdb.StationSerialDataDb(stns.copy(), "tmin", tmin.days)
    with pytest.raises(IOError, match="climate division 102"):
        xval.set_optim_nstns_from_files(da3, d2, sdb.get_optim_varname)
    lax = xval.set_optim_nstns_from_files(da3, d2, sdb.get_optim_varname, strict=False)
    assert lax["missing"] == [102.0] and sorted(k for k in lax if k != "missing") == [101.0, 4407.0]


def test_shard_unshard_roundtrip():
    from topowx_amd import xval
    for n in (0, 1, 7, 16):
        items = np.arange(n)
        for world in (1, 2, 3, 8):
            parts = [xval.shard(items, r, world) for r in range(world)]
            assert sum(len(p) for p in parts) == n
            if n:
                np.testing.assert_array_equal(xval._unshard([p[None, :] for p in parts], n)[0], items)


# ---- two gloo ranks == one process --------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleXvalNorm(object):
    """Stand-in for interp.optimize.XvalTairNorm with the same batched interface, computed by the oracle."""

    def __init__(self, stn_da, tair_var, device=0):
        from oracle import pyoracle as orc
        from topowx_amd import stationdb as sdb
        self.orc, self.db, self.prm = orc, orc.Db(stn_da), orc.params()
        self.idx = {s: i for i, s in enumerate(stn_da.stns[sdb.STN_ID][np.isnan(stn_da.stns[sdb.BAD])])}

    def run_xval_many(self, stn_ids, ladder, raise_on_error=True):
        c = self.db.cols
        err = np.zeros((len(stn_ids), 12, len(ladder)))
        ok = np.ones(len(stn_ids), bool)
        for i, s in enumerate(stn_ids):
            j = self.idx[s]
            pt = self.orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
            for x, k in enumerate(ladder):
                rc, norms, _ = self.orc.krigall(self.db, self.prm, pt, int(k), excl=j, rm_zero_dist=True)
                ok[i] &= rc == 0
                err[i, :, x] = norms - c["norm"][:, j]
        return err, ok

    def close(self):
        pass


class _OracleKrigParams(object):
    """Stand-in for interp.optimize.StationKrigParams (step22) computed by the oracle."""

    def __init__(self, stn_da, tair_var, device=0):
        from oracle import pyoracle as orc
        from topowx_amd import stationdb as sdb
        self.orc, self.db, self.prm = orc, orc.Db(stn_da), orc.params()
        self.idx = {s: i for i, s in enumerate(stn_da.stns[sdb.STN_ID][np.isnan(stn_da.stns[sdb.BAD])])}

    def get_krig_params_many(self, stn_ids, raise_on_error=True):
        c = self.db.cols
        out = np.zeros((3, len(stn_ids), 12))
        ok = np.ones(len(stn_ids), bool)
        for i, s in enumerate(stn_ids):
            j = self.idx[s]
            pt = self.orc.make_pt(c["lon"][j], c["lat"][j], c["elev"][j], c["tdi"][j], c["lst"][:, j])
            for m in range(1, 13):
                rc, v, _ = self.orc.build_krig_params(self.db, self.prm, pt, m)
                ok[i] &= rc == 0
                out[:, i, m - 1] = v
        return out[0], out[1], out[2], ok

    def close(self):
        pass


def _case():
    from topowx_amd import stationdb as sdb, synth
    grid = synth.make_grid("C1", nrows=20, ncols=20)
    stn = synth.make_stations(grid["bbox"], 260, 4, "tmin")
    # a few stations outside the mask / flagged bad: not cross-validated (step21:146-149)
    stn.stns[sdb.BAD][5] = 1.0
    ids = None
    return stn, ids


def _farm(rank, world):
    from topowx_amd import xval
    xval.XvalTairNorm = _OracleXvalNorm
    stn, _ = _case()
    ids = xval.xval_station_ids(stn)[:23]
    ids_out, mae = xval.optim_nstns_norms(stn, "tmin", ladder=[35, 57], stn_ids=ids, rank=rank, world=world, batch=4)
    chosen = xval.set_optim_nstns_tair_norm(stn, ids_out, mae, [35, 57])
    # step22 on the table step21's reduction has just written the bandwidths into
    import topowx_amd.interp.optimize as opt
    opt.StationKrigParams = _OracleKrigParams
    xval.set_stn_variograms(stn, "tmin", stn_ids=ids[:9], rank=rank, world=world, batch=2)
    return ids_out, mae, chosen, stn


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, mae, chosen, stn = _farm(rank, world)
    from topowx_amd import stationdb as sdb
    np.savez(os.path.join(outdir, "r%d.npz" % rank), ids=ids, mae=mae,
             optim=np.stack([stn.stns[sdb.get_optim_varname(m)] for m in range(1, 13)]),
             vario=np.stack([stn.stns[sdb.get_krigparam_varname(m, f)] for m in range(1, 13)
                             for f in (sdb.VARIO_NUG, sdb.VARIO_PSILL, sdb.VARIO_RNG)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_xval_farm_equals_single_process(tmp_path):
    from topowx_amd import stationdb as sdb
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ids, mae, chosen, stn = _farm(0, 1)
    assert mae.shape == (12, 2, 23) and np.isfinite(mae).all() and (mae > 0).all()
    want_optim = np.stack([stn.stns[sdb.get_optim_varname(m)] for m in range(1, 13)])
    want_vario = np.stack([stn.stns[sdb.get_krigparam_varname(m, f)] for m in range(1, 13)
                           for f in (sdb.VARIO_NUG, sdb.VARIO_PSILL, sdb.VARIO_RNG)])
    orig = np.stack([_case()[0].stns[sdb.get_krigparam_varname(m, f)] for m in range(1, 13)
                     for f in (sdb.VARIO_NUG, sdb.VARIO_PSILL, sdb.VARIO_RNG)])
    changed = ~((want_vario == orig) | (np.isnan(want_vario) & np.isnan(orig)))
    assert 0 < changed.any(axis=0).sum() <= 9                           # only the nine stations step22 was asked for
    for r in range(2):                                     # every rank ends with the full result
        got = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        np.testing.assert_array_equal(got["ids"], ids)
        np.testing.assert_array_equal(got["mae"], mae)
        np.testing.assert_array_equal(got["optim"], want_optim)
        np.testing.assert_array_equal(got["vario"], want_vario)          # step22: every rank ends with the same table
    assert len(chosen) >= 1 and all(set(v) <= {35, 57} for v in chosen.values())


def _set_optim_nstns_masked_loop(stns, stn_ids, mae, ladder, namer):
    """optimize.py:268-374 statement by statement (division loop, masked mean, first minimum) -- the checker of the
    batched ``xval.set_optim_nstns``."""
    from topowx_amd import stationdb as sdb
    ladder, ids = np.asarray(ladder), np.asarray(stn_ids)
    pos = {s: i for i, s in enumerate(stns[sdb.STN_ID])}
    div_of_xval = stns[sdb.CLIMDIV][[pos[s] for s in ids]]
    climdiv_stns = stns[sdb.CLIMDIV]
    chosen = {}
    for clim_div in np.unique(climdiv_stns[np.isfinite(climdiv_stns)]):
        cols = np.nonzero(div_of_xval == clim_div)[0]
        if cols.size == 0:
            continue
        climdiv_mask = np.nonzero(climdiv_stns == clim_div)[0]
        pick = np.zeros(12, ladder.dtype)
        for mth in range(1, 13):
            mmae = np.ma.mean(np.ma.masked_invalid(mae[mth - 1][:, cols]), axis=1)
            min_idx = int(np.argmin(mmae))
            stns[namer(mth)][climdiv_mask] = ladder[min_idx]
            pick[mth - 1] = ladder[min_idx]
        chosen[float(clim_div)] = pick
    return stns, chosen


@pytest.mark.parametrize("case", ["plain", "nan_entries", "failed_stations", "ties", "dead_bandwidth"])
def test_set_optim_nstns_equals_masked_division_loop(case):
    """The batched reduction picks the same bandwidth as the reference's loop of masked means for every (division,
    month) -- also when means tie exactly, when stations failed (NaN columns) and when a bandwidth has no finite
    error in a division."""
    from topowx_amd import stationdb as sdb, synth, xval
    stn = synth.make_stations((30.0, 38.0, -100.0, -90.0), 3000, 5, "tmin")
    ids = xval.xval_station_ids(stn)
    rng = np.random.default_rng(3)
    mae = rng.random((12, 16, len(ids)))
    if case != "plain":
        mae[rng.random(mae.shape) < 0.05] = np.nan
    if case == "failed_stations":
        mae[:, :, rng.random(len(ids)) < 0.3] = np.nan
    if case == "ties":
        mae = np.round(mae, 1)
    if case == "dead_bandwidth":
        mae[:, 3, :] = np.nan
        mae[:, 0, : len(ids) // 2] = np.nan
    a, ca = _set_optim_nstns_masked_loop(stn.stns.copy(), ids, mae, xval.DFLT_LADDER, sdb.get_optim_varname)
    b, cb = xval.set_optim_nstns(stn.stns.copy(), ids, mae, xval.DFLT_LADDER, sdb.get_optim_varname)
    assert len(ca) > 50 and set(ca) == set(cb)
    assert all(np.array_equal(ca[k], cb[k]) for k in ca)
    for m in range(1, 13):
        assert np.array_equal(a[sdb.get_optim_varname(m)], b[sdb.get_optim_varname(m)], equal_nan=True)
