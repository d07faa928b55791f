"""Average duration per kernel from a rocprofv3 kernel-trace CSV (any pass of collect_sq.sh / collect_profiles.sh)."""
import collections
import csv
import sys

per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0.0
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    if k.startswith("k_"):
        print("%-22s n=%3d avg %.3f ms  vgpr/lds see trace" % (k, len(v), sum(v) / len(v) / 1e6))
        if k.startswith(("k_uk<", "k_ukw<", "k_cell_dist", "k_tile_dist")):
            tot += sum(v) / len(v) / 1e6
print("kriging kernels (k_tile_dist + k_ukw + k_uk) sum of averages: %.3f ms" % tot)
