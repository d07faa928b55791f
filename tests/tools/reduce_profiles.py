"""Reduce the rocprofv3 outputs of tests/tools/collect_profiles.sh to the small files kept under profiles/:
the kernel-stats table, per-kernel sums of the PMC passes, and the HBM bytes per launch of the kriging kernels
(quoted by bench.py as roofline.traffic) and of the daily kernels."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
KRIG = ("k_uk<", "k_ukw<", "k_cell_dist")          # the kernels behind bench.py's uk_ms
DAILY = ("k_daily_grid", "k_row_offsets")          # ... daily_ms
OTHER = ("k_gwr_z", "k_fix_cells", "k_select", "k_tile_cand")


def short(name):
    return name.split("(")[0].replace("void ", "")


res = {}
for key, sub, pre in (("FETCH_SIZE", "fetch", "f"), ("WRITE_SIZE", "write", "w")):
    paths = glob.glob(os.path.join(out, sub, "**", pre + "_counter_collection.csv"), recursive=True)
    if not paths:
        continue
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(paths[0])):
        if r["Counter_Name"] != key:
            continue
        k = short(r["Kernel_Name"])
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    with open(os.path.join(out, "pmc_%s.csv" % key), "w") as fh:
        fh.write("kernel,dispatches,%s_sum_KB\n" % key)
        for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            fh.write('"%s",%d,%.1f\n' % (k, n, v))

    def group(prefixes):
        n = sum(c for k, (c, v) in per.items() if k.startswith(prefixes))
        kb = sum(v for k, (c, v) in per.items() if k.startswith(prefixes))
        return n, kb
    n, kb = group(KRIG)
    res[key] = {"k_uk_launches": n, "k_uk_total_KB": kb, "k_uk_per_launch_bytes": kb * 1024.0 / max(n, 1),
                "kernels": "k_cell_dist + k_ukw<..> + k_uk<..>"}
    n, kb = group(DAILY)
    res[key]["daily"] = {"launches": n, "total_KB": kb, "per_launch_bytes": kb * 1024.0 / max(n, 1),
                         "kernels": "k_daily_grid + k_row_offsets"}
    for o in OTHER:
        n, kb = group((o,))
        res[key][o] = {"launches": n, "per_launch_bytes": kb * 1024.0 / max(n, 1)}
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
# kernel stats table (rocprofv3 --stats)
for p in glob.glob(os.path.join(out, "stats", "**", "s_kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(p)))
    with open(os.path.join(out, "kernel_stats.csv"), "w") as fh:
        fh.write("kernel,calls,total_ns,avg_ns,pct\n")
        for r in rows:
            fh.write('"%s",%s,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
print(json.dumps(res))
