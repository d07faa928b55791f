"""CPU: the committed PMC reduction that bench.py quotes as ``roofline.traffic`` / ``daily.traffic`` belongs to the kernels in
the tree.  ``tests/tools/reduce_profiles.py`` stamps a hash of the kernel sources (topowx_amd/csrc/*, include/twx.h) into
``profiles/r*_bench_hbm_traffic.json``; a kernel change without a fresh ``tests/tools/collect_round.sh`` +
``install_profiles.sh`` run on the GPU box fails here instead of letting the bench line quote stale counters."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def test_committed_hbm_traffic_belongs_to_the_kernels_in_the_tree():
    import kernel_hash
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_hbm_traffic.json")))
    assert paths, "no profiles/r*_bench_hbm_traffic.json"
    newest = json.load(open(paths[-1]))
    assert "kernel_sources_sha16" in newest, "%s carries no kernel-source hash (collected before round 4?)" % paths[-1]
    assert newest["kernel_sources_sha16"] == kernel_hash.kernel_sources_sha16(), (
        "%s was collected on other kernel sources than this tree's: run `gpurun -- bash tests/tools/collect_round.sh`, "
        "then `bash tests/tools/install_profiles.sh rN`" % os.path.basename(paths[-1]))
    # and bench.py says so in its line when they do not match
    sys.path.insert(0, ROOT)
    import bench
    traffic, daily, src = bench.latest_traffic()
    assert traffic > 0 and daily > 0 and "STALE" not in src


def test_committed_bench_lines_quote_counters_of_their_own_kernels():
    """The newest committed bench lines were produced after the counter passes of the same collection
    (tests/tools/collect_profiles.sh runs the PMC passes first): none of them carries the STALE marker, and the C4 tile's
    traffic record belongs to the kernels in the tree as well."""
    import kernel_hash
    for pat in ("r*_bench.json", "r*_bench_profiled.json", "r*_bench_daily_profiled.json", "r*_c4_stream_daily.json"):
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        assert paths, pat
        assert "STALE" not in open(paths[-1]).read(), "%s quotes counters of other kernel sources" % os.path.basename(paths[-1])
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c4_daily_traffic.json")))
    assert paths and json.load(open(paths[-1])).get("kernel_sources_sha16") == kernel_hash.kernel_sources_sha16()
