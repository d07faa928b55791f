"""Reduce the rocprofv3 outputs of tests/tools/collect_profiles.sh to the small files kept under profiles/:
the kernel-stats tables, per-kernel sums of the PMC passes, and the HBM bytes per launch of the kriging kernels
(quoted by bench.py as roofline.traffic) and of the daily kernels."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
KRIG = ("k_uk<", "k_ukz<", "k_ukw<", "k_ukwz<", "k_ukw2<", "k_cell_dist", "k_tile_dist")   # the kernels behind bench.py's uk_ms
DAILY = ("k_daily_tile", "k_tile_union", "k_tile_uidx", "k_perm", "k_row_offsets", "k_daily_ok", "k_daily_grid")   # ... daily_ms
DAILY_ONLY = DAILY + ("k_gwr_z", "k_gwr_z_cell", "k_fix_cells", "k_fix_sparse", "k_compact_flags")


def short(name):
    return name.split("(")[0].replace("void ", "")


def pmc(sub, pre, key):
    paths = glob.glob(os.path.join(out, sub, "**", pre + "_counter_collection.csv"), recursive=True)
    per = collections.defaultdict(lambda: [0, 0.0])
    if paths:
        for r in csv.DictReader(open(paths[0])):
            if r["Counter_Name"] == key:
                k = short(r["Kernel_Name"])
                per[k][0] += 1
                per[k][1] += float(r["Counter_Value"])
    return per


def group(per, prefixes, exact=False):
    """Dispatches and KB of the kernels whose name starts with one of ``prefixes`` (``exact``: equals one of them --
    "k_daily_tile" must not swallow "k_daily_tile_gather", which halved the per-launch bytes in round 2)."""
    hit = (lambda k: k in prefixes) if exact else (lambda k: k.startswith(prefixes))
    n = sum(c for k, (c, v) in per.items() if hit(k))
    kb = sum(v for k, (c, v) in per.items() if hit(k))
    return n, kb


res = {}
for key, sub, dsub, pre in (("FETCH_SIZE", "fetch", "dfetch", "f"), ("WRITE_SIZE", "write", "dwrite", "w")):
    per = pmc(sub, pre, key)
    with open(os.path.join(out, "pmc_%s.csv" % key), "w") as fh:
        fh.write("kernel,dispatches,%s_sum_KB\n" % key)
        for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            fh.write('"%s",%d,%.1f\n' % (k, n, v))
    n, kb = group(per, KRIG)
    res[key] = {"k_uk_launches": n, "k_uk_total_KB": kb, "k_uk_per_launch_bytes": kb * 1024.0 / max(n, 1),
                "kernels": "k_tile_dist + k_ukw<..> + k_ukwz<..> + k_uk<..> (headline workload only)"}
    dper = pmc(dsub, pre, key)
    if dper:
        with open(os.path.join(out, "pmc_daily_%s.csv" % key), "w") as fh:
            fh.write("kernel,dispatches,%s_sum_KB\n" % key)
            for k, (n, v) in sorted(dper.items(), key=lambda kv: -kv[1][1]):
                if k.startswith(DAILY_ONLY):
                    fh.write('"%s",%d,%.1f\n' % (k, n, v))
        for o in ("k_daily_tile", "k_daily_tile_gather", "k_tile_uidx", "k_perm", "k_gwr_z", "k_gwr_z_cell", "k_fix_cells", "k_fix_sparse"):
            n, kb = group(dper, (o,), exact=True)
            res[key][o] = {"launches": n, "per_launch_bytes": kb * 1024.0 / max(n, 1)}
        # the launches behind bench.py's daily.timing_ms.daily_ms + gwr_ms, per daily step (Tmin + Tmax): the record's
        # measured traffic (steps = dispatches of k_daily_tile)
        steps = max(1, group(dper, ("k_daily_tile",), exact=True)[0])
        n, kb = group(dper, DAILY + ("k_gwr_z",))            # (prefix: k_gwr_z_cell too)
        res[key]["daily_path_per_step_bytes"] = kb * 1024.0 / steps
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_hash  # noqa: E402
res["kernel_sources_sha16"] = kernel_hash.kernel_sources_sha16()     # the kernels these bytes were measured on
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)


def stats(sub, name, only=None):
    for p in glob.glob(os.path.join(out, sub, "**", "s_kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(p)))
        with open(os.path.join(out, name), "w") as fh:
            fh.write("kernel,calls,total_ns,avg_ns,pct\n")
            for r in rows:
                k = short(r["Name"])
                if only is None or k.startswith(only):
                    fh.write('"%s",%s,%s,%s,%s\n' % (k, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))


stats("stats", "kernel_stats.csv")
stats("dstats", "kernel_stats_daily.csv", DAILY_ONLY)
print(json.dumps(res))
