"""Reduce the rocprofv3 outputs of the deflate kernels (tests/tools/collect_round.sh: one configs[3]-sized tile through a
deflate stream, tests/tools/gpu_deflate_ab.py --child) to profiles/rN_deflate_kernels.json: per kernel the average launch
duration (--kernel-trace --stats), the HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes, KB units x 1024) and, per tile, the
algorithmic bytes -- 2 bytes per value read by each of the two passes, the streams written once -- against the measured ones.
    python3 tests/tools/reduce_deflate.py gpurun_out/prof_round/dfl"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def short(name):
    return name.split("(")[0].replace("void ", "")


def pmc(sub, key):
    per = collections.defaultdict(lambda: [0, 0.0])
    for p in glob.glob(os.path.join(out, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == key and short(r["Kernel_Name"]).startswith("k_deflate"):
                k = short(r["Kernel_Name"])
                per[k][0] += 1
                per[k][1] += float(r["Counter_Value"])
    return per


res = {"kernels": {}}
for p in glob.glob(os.path.join(out, "stats", "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = short(r["Name"])
        if k.startswith("k_deflate"):
            res["kernels"][k] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
for key, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    for k, (n, v) in pmc(sub, key).items():
        res["kernels"].setdefault(k, {})[key + "_bytes_per_launch"] = v * 1024.0 / max(n, 1)
for name in ("stats", "fetch", "write"):
    for ln in open(os.path.join(out, name + ".log")) if os.path.exists(os.path.join(out, name + ".log")) else ():
        if ln.startswith("{") and name == "stats":
            res["run"] = json.loads(ln)
nd, cells = 25203, 250 * 250
int16 = 2 * nd * cells * 2                                     # both variables
ratio = res.get("run", {}).get("bytes_over_int16")
ks = res["kernels"]
per_var = {k: v for k, v in ks.items()}
res["per_tile"] = {
    "int16_bytes": int16,
    "algorithmic_bytes": {"read": (2 + 1.0 / 16) * int16, "written": None if ratio is None else ratio * int16,
                          "note": "every value is read by k_deflate_count and again by k_deflate_emit (2 bytes each time), every 16th segment also by k_deflate_hist; the streams are written once"},
    "measured_bytes": {"read": sum(v.get("FETCH_SIZE_bytes_per_launch", 0.0) * (1 if k == "k_deflate_table" else 2) for k, v in per_var.items()),
                       "written": sum(v.get("WRITE_SIZE_bytes_per_launch", 0.0) * (1 if k == "k_deflate_table" else 2) for k, v in per_var.items()),
                       "note": "two launches of each kernel per tile (Tmin, Tmax); k_deflate_table: one, both variables"},
    "kernel_us": sum(v.get("avg_us", 0.0) * v.get("calls", 0) for v in per_var.values()) / max(1, ks.get("k_deflate_scan", {}).get("calls", 2) // 2)}
alg = res["per_tile"]["algorithmic_bytes"]
if alg["written"] is not None and res["per_tile"]["kernel_us"] > 0:
    res["per_tile"]["achieved_GBps_algorithmic"] = (alg["read"] + alg["written"]) / res["per_tile"]["kernel_us"] / 1e3
    res["per_tile"]["frac_of_8_TBps"] = res["per_tile"]["achieved_GBps_algorithmic"] / 8000.0
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_hash  # noqa: E402
res["kernel_sources_sha16"] = kernel_hash.kernel_sources_sha16()
json.dump(res, open(os.path.join(out, "deflate_kernels.json"), "w"), indent=1)
print(json.dumps(res["per_tile"]))
