"""Per-kernel sums of the SQ counter passes collected by tests/tools/collect_sq.sh (average per dispatch)."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for p in glob.glob(os.path.join(out, "p*", "**", "p_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
names = sorted({c for k in per for c in per[k]})
# the bench line of the first pass: executed flops per kriging bucket -> fp64 FMAs NEEDED per dispatch of its kernel
need = {}
try:
    import json
    for ln in open(os.path.join(out, "p1.log")):
        if ln.startswith("{"):
            f = json.loads(ln)["fp64"]
            for kmax, kern_name in f["kernels_by_bucket_kmax"].items():
                if kmax in f["executed_flops_by_bucket_kmax"]:
                    need[kern_name.replace(" ", "")] = f["executed_flops_by_bucket_kmax"][kmax] / 128.0
except (OSError, KeyError, ValueError):
    pass
kern = sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", 0))
print("counter averages per dispatch (SQ_*_CYCLES in quad-cycles summed over waves / SIMDs as the counter defines)")
for k in kern:
    if not k.startswith("k_"):
        continue
    v = {c: per[k][c] / max(calls[k][c], 1) for c in names}
    print("\n== %s  (dispatches %d)" % (k, max(calls[k].values())))
    for c in names:
        print("  %-28s %16.0f" % (c, v[c]))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("  -- wait_any %.2f  wait_inst %.2f  active_inst %.2f  (of wave cycles)" %
              (v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    iv = v.get("SQ_INSTS_VALU", 0)
    if iv:
        print("  -- of VALU insts: fma64 %.2f mul64 %.2f add64 %.2f trans64 %.3f fma32 %.2f trans32 %.3f int32 %.2f cvt %.3f" %
              tuple(v.get("SQ_INSTS_VALU_" + n, 0) / iv for n in ("FMA_F64", "MUL_F64", "ADD_F64", "TRANS_F64", "FMA_F32",
                                                                 "TRANS_F32", "INT32", "CVT")))
        print("  -- LDS insts / VALU insts %.3f   SALU / VALU %.3f" % (v.get("SQ_INSTS_LDS", 0) / iv, v.get("SQ_INSTS_SALU", 0) / iv))
    nk = need.get(k.replace(" ", ""))
    if nk and v.get("SQ_INSTS_VALU_FMA_F64"):
        print("  -- fp64 FMA wave-instructions issued / needed (k^3/3 + 7k^2 of the bucket's systems, 128 flops each): %.0f / %.0f = %.2f"
              % (v["SQ_INSTS_VALU_FMA_F64"], nk, v["SQ_INSTS_VALU_FMA_F64"] / nk))
    if v.get("SQ_ACTIVE_INST_LDS") and v.get("SQ_ACTIVE_INST_VALU"):
        # one LDS pipe per CU against four SIMDs: busy time of the LDS pipe relative to the busy time of ONE SIMD's VALU
        print("  -- LDS pipe busy / per-SIMD VALU busy: %.2f  (4 x ACTIVE_INST_LDS / ACTIVE_INST_VALU; bank-conflict share of LDS time %.2f)"
              % (4.0 * v["SQ_ACTIVE_INST_LDS"] / v["SQ_ACTIVE_INST_VALU"],
                 v.get("SQ_LDS_BANK_CONFLICT", 0) / (4.0 * v["SQ_ACTIVE_INST_LDS"])))
    if v.get("SQ_BUSY_CYCLES"):
        print("  -- VALU active / busy cycles %.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0) / v["SQ_BUSY_CYCLES"]))
