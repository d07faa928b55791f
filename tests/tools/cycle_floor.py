"""The cycle floor of the kriging kernels' DESIGN, per kernel, as a markdown table.
    python3 tests/tools/cycle_floor.py SQ_DIR KERNEL_STATS_CSV [BENCH_JSON]
SQ_DIR: what tests/tools/collect_sq.sh left (p1..p3 counter passes of bench.py's normals workload + p1.log with the bench line);
KERNEL_STATS_CSV: rocprofv3 --kernel-trace --stats of the same command WITHOUT counters (average durations).
Per kernel and dispatch: needed fp64 FMA wave-instructions (k^3/3 + 7 k^2 of the bucket's systems / 128), issued ones
(SQ_INSTS_VALU_FMA_F64), all other VALU instructions, the time the VALU pipes need to ISSUE all of them (4 cycles per wave
instruction, 1 024 SIMDs, 2.4 GHz), the LDS pipe's busy time per CU, the larger of the two = the floor of this design, and what
the launch takes."""
import collections
import csv
import glob
import json
import os
import sys

sq_dir, stats_csv = sys.argv[1], sys.argv[2]
CLK, SIMDS, CUS = 2.4e9, 1024, 256
per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for p in glob.glob(os.path.join(sq_dir, "p*", "**", "p_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(" ", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
need, nsys = {}, {}
for ln in open(os.path.join(sq_dir, "p1.log")):
    if ln.startswith("{"):
        f = json.loads(ln)["fp64"]
        for kmax, name in f["kernels_by_bucket_kmax"].items():
            if kmax in f["executed_flops_by_bucket_kmax"]:
                need[name.replace(" ", "")] = f["executed_flops_by_bucket_kmax"][kmax] / 128.0
                nsys[name.replace(" ", "")] = f["systems_by_bucket_kmax"][kmax]
dur = {}
for r in csv.DictReader(open(stats_csv)):
    name = (r.get("kernel") or r.get("Name") or r.get("KernelName") or "").split("(")[0].replace("void ", "").replace(" ", "")
    avg = r.get("avg_ns") or r.get("AverageNs") or r.get("Average")
    if name and avg:
        dur[name] = float(avg) / 1e3                       # us
rows = []
for k in sorted(need, key=lambda n: -dur.get(n, 0)):
    v = {c: per[k][c] / max(calls[k][c], 1) for c in per[k]}
    if not v.get("SQ_INSTS_VALU"):
        continue
    fma, valu = v.get("SQ_INSTS_VALU_FMA_F64", 0), v["SQ_INSTS_VALU"]
    t_valu = valu * 4 / SIMDS / CLK * 1e6
    # SQ_ACTIVE_INST_LDS counts, per SIMD, the cycles an LDS instruction of that SIMD is in flight on the CU's ONE LDS pipe:
    # summed over the launch and divided by the CUs = busy time of a CU's pipe
    t_lds = v.get("SQ_ACTIVE_INST_LDS", 0) * 4 / CUS / CLK * 1e6
    floor = max(t_valu, t_lds)
    rows.append((k, nsys.get(k, 0), need[k], fma, valu - fma, t_valu, t_lds, floor, dur.get(k, float("nan"))))
print("| kernel | systems | needed fp64 FMAs (M) | issued (M) | issued / needed | other VALU (M) | VALU issue floor µs | LDS pipe µs | floor µs | measured µs | measured / floor |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
tot_f = tot_m = 0.0
for k, n, nd, fma, oth, tv, tl, fl, m in rows:
    print("| `%s` | %d | %.1f | %.1f | %.2f | %.1f | %.0f | %.0f | %.0f | %.0f | %.2f |" % (k, n, nd / 1e6, fma / 1e6, fma / nd, oth / 1e6, tv, tl, fl, m, m / fl))
    tot_f += fl
    tot_m += m
print("| all kriging buckets | %d | | | | | | | %.0f | %.0f | %.2f |" % (sum(r[1] for r in rows), tot_f, tot_m, tot_m / tot_f))
