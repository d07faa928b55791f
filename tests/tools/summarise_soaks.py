"""Reduce the per-case soak records under profiles/ (tests/tools/gpu_soak.py, gpu_soak_points.py) to ONE summary each:
the totals, the seed range, aggregate counters of the case records, and -- in full -- only the records of failed cases and
of cases with near ties.  A soak is reproducible from its seeds (python3 tests/tools/gpu_soak.py N --seed0 S); the per-case
records of 200 ... 1 200 passing cases are not evidence anyone reads.
    python3 tests/tools/summarise_soaks.py            (rewrites profiles/r*_soak_*.json in place)"""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_soak_*.json"))):
    d = json.load(open(p))
    recs = d.get("records")
    if not isinstance(recs, list) or "records_kept" in d:
        continue                                              # already a summary
    out = {k: v for k, v in d.items() if k != "records"}
    seeds = [r["seed"] for r in recs if "seed" in r]
    out["seeds"] = [min(seeds), max(seeds)] if seeds else None
    agg = {}
    for r in recs:
        for k, v in r.items():
            if isinstance(v, bool):
                agg.setdefault(k + "_true", 0)
                agg[k + "_true"] += int(v)
            elif isinstance(v, (int, float)) and k not in ("seed",):
                a = agg.setdefault(k, {"min": v, "max": v, "sum": 0.0})
                a["min"], a["max"], a["sum"] = min(a["min"], v), max(a["max"], v), a["sum"] + v
    out["aggregate_over_cases"] = {k: (v if not isinstance(v, dict) else {"min": v["min"], "max": v["max"], "mean": v["sum"] / len(recs)})
                                   for k, v in agg.items()}
    bad = [r for r in recs if r.get("ok") is False or r.get("pass") is False or r.get("failed") or r.get("near_tie_cells")
           or r.get("ninvalid_mismatch_cells") or r.get("status_equal") is False or r.get("ninvalid_equal") is False]
    out["records_kept"] = "failed cases and cases with near ties only (%d of %d); all cases are reproducible from their seeds" % (len(bad), len(recs))
    out["records"] = bad
    json.dump(out, open(p, "w"), indent=1)
    print("%s: %d records -> %d kept, %d bytes" % (os.path.basename(p), len(recs), len(bad), os.path.getsize(p)))
