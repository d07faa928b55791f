"""RCCL beside libtwxhip.so in one process.

No multi-GPU node is available to this build (SCALE_rNN.json: skipped), so the N > 1 path has only ever run over gloo
(tests/test_driver_gloo.py on CPU, two ranks sharing the GPU in tests/test_gpu_bench_contract.py).  What a real 8-GPU run
hits FIRST, though, can be shown on the 1-GPU box: that RCCL initialises and runs a collective in a process that has
loaded libtwxhip.so under the torch-first HIP-runtime load order (tests/conftest.py), on the very device tensor
``interp_tiles_device`` has filled.  A ONE-RANK ``nccl`` process group is a complete RCCL communicator: ``dist.all_reduce`` and
``dist.gather`` launch RCCL kernels on the GPU.  Still "unmeasured on hardware" as far as xGMI goes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import torch                                   # first: its HIP runtime serves the process (tests/conftest.py)
import torch.distributed as dist
import numpy as np
sys.path.insert(0, os.environ["TWX_ROOT"])
from topowx_amd import _lib, driver, synth
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
grid, tmin, tmax = synth.make_case("C1")
ctx = _lib.Context(device=0)                    # libtwxhip.so's kernels and RCCL's in one process, one device
ctx.set_stations(_lib.TMIN, tmin, with_obs=False)
ctx.set_stations(_lib.TMAX, tmax, with_obs=False)
tiles = driver.tile_list(grid["mask"], 50, 50)
assignment = driver.assign_tiles(tiles, 1)
dgrid = driver.upload_grid(grid, "cuda:0", assignment[0], 50, 50)
buf, stat, ms = driver.interp_tiles_device(ctx, dgrid, assignment[0], 50, 50)
probe = buf[:, :, :, :2, :2].clone()
dist.all_reduce(probe)                          # an RCCL kernel on the tensor the library's kernels wrote
same = bool(torch.equal(probe, buf[:, :, :, :2, :2]))
mosaic = driver.gather_mosaic_device(buf, assignment, grid["mask"].shape, 50, 50, 0, 1, collective=True)   # dist.gather over RCCL
# ... and the library again AFTER RCCL has run, on the same context
buf2, _, _ = driver.interp_tiles_device(ctx, dgrid, assignment[0], 50, 50)
dist.barrier()
whole = ctx.interp_grid(grid)
ok = all(np.array_equal(mosaic[k].cpu().numpy(), whole[k]) for k in driver.NORMAL_KEYS)
print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "all_reduce_identity": same,
                  "mosaic_equals_whole_grid": ok, "second_pass_equal": bool(torch.equal(buf, buf2)), "tiles": len(tiles),
                  "nccl_version": list(torch.cuda.nccl.version())}))
ctx.close()
dist.destroy_process_group()
"""


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_one_rank_group_beside_libtwxhip():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), TWX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stderr[-3000:], p.stdout[-500:])
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])      # (RCCL prints its library path on stdout)
    assert d["backend"] == "nccl" and d["world"] == 1 and d["tiles"] == 4
    assert d["all_reduce_identity"] and d["mosaic_equals_whole_grid"] and d["second_pass_equal"]


def test_bench_strong_line_under_a_one_rank_nccl_group():
    """``bench.py --gpus 1 --scaling strong`` with TWX_BENCH_FORCE_PG=1: process group, barriers, the max-over-ranks reduction
    and the mosaic gather of the N > 1 path all run, over RCCL, at world 1."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), TWX_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong", "--steps", "1", "--warmup", "0",
                        "--strip-rows", "100", "--strip-cols", "400", "--strip-tile", "50", "--strip-nstns", "2500", "--strong-daily-rows", "0"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    s = d["strong"]
    assert d["scaling"] == "strong" and s["process_group"] == "nccl, world 1" and "ONE-RANK RCCL" in s["gather"]
    assert s["cells_ok"] == s["cells_valid"] > 0 and s["gather_ms"] > 0
    assert s["spot_check_vs_oracle"]["max_abs_degC"] < 1e-4
