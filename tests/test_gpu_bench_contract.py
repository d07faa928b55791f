"""GPU: bench.py prints ONE JSON line with the fields the driver's contract names (task description: metric / value / unit /
n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config, plus `roofline`
and `cpu_baseline`), on a reduced tile so that the test takes seconds.  Also the 2-rank spawn mode (fresh child processes
through torch.distributed.run; on a 1-GPU box the ranks share the GPU over gloo and say so)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run("--size", "64", "--nstns", "2500", "--steps", "2", "--warmup", "1", "--daily-years", "1", "--stream-tiles", "2",
             "--cpu-sample", "16")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["ms_per_step"] * 1e-3 * d["value"] - d["config"]["cells_ok"] * 12) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r
    assert 0 < d["fp64"]["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    assert d["parity_max_abs_degC"] < 1e-4
    dd = d["daily"]
    assert dd["value"] > 0 and dd["packed_int16_vs_oracle"]["max_abs_lsb"] <= 1 and dd["stream"]["tiles"] == 2


SMALL_STRIP = ("--strip-rows", "100", "--strip-cols", "400", "--strip-tile", "50", "--strip-nstns", "2500")


def test_bench_spawns_ranks_itself():
    d = _run("--gpus", "2", "--size", "64", "--nstns", "2500", "--steps", "2", "--warmup", "1", "--strong-steps", "1",
             "--strong-daily-rows", "100", "--strong-daily-cols", "200", "--strong-daily-years", "1", *SMALL_STRIP)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "daily" not in d and "cpu_baseline" not in d          # rank-0, N = 1 records only
    s = d["strong"]                                              # ... and the tile farm on one fixed grid next to it
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["value"] > 0 and len(s["device_ms_per_rank"]) == 2
    assert sum(s["tiles_per_rank"]) == s["tiles"] and s["cells_ok"] == s["cells_valid"] and s["imbalance_max_over_mean"] >= 1.0
    sd = s["daily"]                                              # the daily (streamed) path under the same deal: no gather
    assert sd["scaling"] == "strong" and sd["value"] > 0 and sd["days"] == 365 and len(sd["d2h_bytes_per_rank"]) == 2
    assert 1 <= sum(sd["tiles_per_rank"]) <= 8 and sd["cells_ok"] > 0, sd      # (tiles without a valid cell are not dealt)
    assert sum(sd["d2h_bytes_per_rank"]) >= sd["cells_ok"] * 365 * 4, sd


def test_bench_strong_two_ranks_equal_one_rank(tmp_path):
    """--scaling strong: ONE masked grid, tiles dealt by driver.assign_tiles, every rank's tiles computed device-resident,
    the mosaic gathered on rank 0 (driver.gather_mosaic_device).  Two ranks (fresh children; on a 1-GPU box they share
    the GPU and the collective runs over gloo) give the mosaic of one rank bit for bit."""
    import numpy as np
    outs = []
    for n in (1, 2):
        path = str(tmp_path / ("mosaic%d.npz" % n))
        d = _run("--gpus", str(n), "--scaling", "strong", "--steps", "1", "--warmup", "0", "--dump-mosaic", path, *SMALL_STRIP)
        assert d["scaling"] == "strong" and d["n_gpus"] == n and d["value"] > 0
        s = d["strong"]
        assert s["cells_ok"] == s["cells_valid"] > 0 and len(s["tiles_per_rank"]) == n and s["gather_ms"] >= 0
        if n == 1:
            assert s["spot_check_vs_oracle"]["max_abs_degC"] < 1e-4 and s["spot_check_vs_oracle"]["fills_where_oracle_fails"]
        outs.append(np.load(path))
    assert sorted(outs[0].files) == ["norm_tmax", "norm_tmin", "se_tmax", "se_tmin"]
    for k in outs[0].files:
        assert outs[0][k].shape == (12, 100, 400) and np.array_equal(outs[0][k], outs[1][k]), k
    assert (outs[0]["norm_tmin"] != np.float32(9.969209968386869e36)).mean() > 0.3      # the masked grid has valid cells


def test_bench_other_configs_reduced():
    """The c4_tile / c5 / c3 records of the default line, on reduced sizes (--force-configs; c3_strip = the code path of
    c3 on a small strip instead of the full configs[2] grid)."""
    d = _run("--size", "64", "--nstns", "2500", "--steps", "1", "--warmup", "1", "--no-daily", "--cpu-sample", "16",
             "--force-configs", "--configs", "c4_tile,c5,c3_strip", "--c5-years", "1", "--c5-nstns", "2500",
             "--strong-steps", "1", *SMALL_STRIP)
    assert d["dtype"].startswith("f64") and d["roofline"]["traffic_measured_in_this_run"] is False
    assert d["roofline"]["systems_on_fp64_covariance_build"] == 0        # the synthetic nuggets are >= 0.1: fast build only
    c = d["configs"]
    t = c["c4_tile"]
    assert t["value"] > 0 and t["cell_days_per_step"] == t["cells_ok"] * 25203 * 2 and t["cells_ok"] == 64 * 64
    assert t["packed_int16_vs_oracle"]["max_abs_lsb"] <= 1 and t["packed_int16_vs_oracle"]["ninvalid_equal"]
    x = c["c5"]
    assert x["step21_s"] > 0 and x["step22_s"] > 0 and x["step23_s"] > 0 and x["step24_s"] > 0 and x["stations"] > 2000
    assert x["db"] == "c5" and x["step22_fitted_frac"] > 0.95
    assert x["spot_check_vs_oracle"]["max_abs_degC"] < 1e-4 and x["spot_check_vs_oracle"]["step21_values"] > 0
    s = c["c3_strip"]
    assert s["value"] > 0 and s["cells_ok"] == s["cells_valid"] and s["spot_check_vs_oracle"]["max_abs_degC"] < 1e-4


def test_bench_c2_fitted_and_c4_reduced():
    """The two records added in round 5, on reduced sizes: ``c2_fitted`` (the headline tile under the variograms step21 ->
    step22 fit on its own database: same systems, parity vs the oracle on the fitted table) and ``c4`` (every tile of a
    masked grid streamed with daily output: counts, per-tile times, oracle check of cells from four tiles)."""
    d = _run("--size", "64", "--nstns", "2500", "--steps", "2", "--warmup", "1", "--no-daily", "--cpu-sample", "16",
             "--force-configs", "--configs", "c2_fitted,c4,c3_fitted", "--c4-rows", "150", "--c4-cols", "400", "--c4-years", "2",
             "--strip-tile", "50", "--strip-nstns", "2500", "--strip-rows", "100", "--strip-cols", "400")
    f = d["configs"]["c2_fitted"]
    assert f["cells_ok"] + sum(f["cells_failed_by_status"].values()) == 64 * 64 and f["uk_solves"] > 0
    assert f["value"] > 0 and f["uk_ms"] > 0 and 0 <= f["systems_on_fp64_covariance_build"] <= f["uk_solves"]
    assert f["parity_max_abs_degC"] < 1e-4 and f["status_equal_oracle"] and f["parity_cells"] > 0
    q = f["fitted_variograms"]["nug_over_psill_quantiles_5_25_50_75_95"]
    assert q is not None and q == sorted(q)
    c = d["configs"]["c4"]
    assert c["tiles"] >= 4 and c["cells_ok"] + sum(c["failures_by_status"].values()) == c["cells_valid"]
    assert c["cell_days"] == c["cells_ok"] * 731 * 2 and c["value"] > 0 and c["d2h_bytes"] > 0
    assert c["device_ms_per_tile"]["min"] <= c["device_ms_per_tile"]["median"] <= c["device_ms_per_tile"]["max"]
    pr = c["precision"]                                 # driver.PrecisionPolicy: requested "auto", the decision in words
    assert pr["requested"] == "auto" and pr["precision"] in ("exact", "fast") and pr["tiles_exact"] + pr["tiles_fast"] == c["tiles"]
    k = c["sink"]                                       # ncio.TileSink: tiles into NetCDF-4 tile files, read back through libhdf5
    assert k["netcdf4"]["tiles"] >= 4 and k["netcdf4"]["tiles_read_back_equal"] == 1 and k["netcdf4"]["int16_GBps_end_to_end"] > 0
    assert k["netcdf4_deflate1"]["tiles_read_back_equal"] == 1 and k["netcdf4_deflate1"]["on_disk_GB"] < k["netcdf4_deflate1"]["int16_GB"]
    assert k["host_new_page_rate"]["cold_GBps"] > 0 and "posix_fallocate" in k["limiting_stage"]
    g = k["netcdf4_deflated_on_gpu"]                    # chunk bytes formed on the device (twx_stream_deflate), appended by the sink
    assert g["tiles_read_back_equal"] == 1 and g["on_disk_GB"] < g["int16_GB"] and g["tiles"] == k["netcdf4"]["tiles"]
    z = c["deflated_on_gpu"]                            # the whole run once more, the daily values leaving the GPU deflated
    assert z["tiles"] == c["tiles"] and z["wall_s"] > 0 and 0 < z["stream_bytes_over_int16"] < 1.001 and z["d2h_bytes"] < c["d2h_bytes"]
    iz = z["inflated_by_zlib"]
    assert iz["max_abs_lsb"] <= 1 and iz["int16_values"] == iz["cells"] * 731 * 2 and iz["differing_from_the_int16_run"] <= 1e-3 * iz["int16_values"]
    assert iz["differing_from_the_int16_run"] == 0 or iz["precision_of_the_int16_run"] != iz["precision_of_this_run"]
    assert z["deflate_kernels_ms_last_tile"] > 0
    s = c["spot_check_vs_oracle"]
    assert s["cells"] >= 12 and s["tiles"] >= 3 and s["status_equal"] and s["ninvalid_equal"]
    assert s["normals_max_abs_degC"] < 1e-4 and s["int16_max_abs_lsb"] <= 1
    assert min(s["tile_valid_cells"]) <= max(s["tile_valid_cells"])
    t = d["configs"]["c3_fitted"]                      # (reduced: the strip instead of the full configs[2] grid)
    assert t["cells_ok"] == t["cells_valid"] and t["uk_solves"] == t["cells_ok"] * 24
    assert 0 <= t["systems_on_fp64_covariance_build"] <= t["uk_solves"] and t["value"] > 0
    assert t["spot_check_vs_oracle"]["max_abs_degC"] < 1e-4 and set(t["fitted_variograms"]) == {"tmin", "tmax"}
