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


def test_bench_spawns_ranks_itself():
    d = _run("--gpus", "2", "--size", "64", "--nstns", "2500", "--steps", "2", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "daily" not in d and "cpu_baseline" not in d          # rank-0, N = 1 records only
