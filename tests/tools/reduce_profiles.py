"""Reduce the rocprofv3 outputs of tests/tools/collect_profiles.sh to the small files kept under profiles/:
per-kernel sums of the PMC passes and the HBM bytes per kriging launch that bench.py quotes."""
import collections
import csv
import json
import os
import sys

out = sys.argv[1]
KRIG = ("k_uk<", "k_ukw<", "k_cell_dist")          # the kernels behind bench.py's uk_ms


def short(name):
    return name.split("(")[0].replace("void ", "")


res = {}
for key, sub, pre in (("FETCH_SIZE", "fetch", "f"), ("WRITE_SIZE", "write", "w")):
    path = os.path.join(out, sub, pre + "_counter_collection.csv")
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != key:
            continue
        k = short(r["Kernel_Name"])
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    with open(os.path.join(out, "pmc_%s.csv" % key), "w") as fh:
        fh.write("kernel,dispatches,%s_sum_KB\n" % key)
        for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            fh.write('"%s",%d,%.1f\n' % (k, n, v))
    n = sum(c for k, (c, v) in per.items() if k.startswith(KRIG))
    kb = sum(v for k, (c, v) in per.items() if k.startswith(KRIG))
    res[key] = {"k_uk_launches": n, "k_uk_total_KB": kb, "k_uk_per_launch_bytes": kb * 1024.0 / max(n, 1),
                "kernels": "k_cell_dist + k_ukw<..> + k_uk<..>"}
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(res))
