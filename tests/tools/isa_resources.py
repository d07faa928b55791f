#!/usr/bin/env python3
"""Reduce hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of build.sh) to one line per kernel:
name | SGPRs | VGPRs | AGPRs | scratch bytes/lane | occupancy waves/SIMD | VGPR spills | LDS bytes/block.
Used by build.sh (-> topowx_amd/libtwxhip.resources.txt) and by tests/test_isa_resources.py (parse())."""
import re
import subprocess
import sys

FIELDS = (("TotalSGPRs", "sgprs"), ("VGPRs", "vgprs"), ("AGPRs", "agprs"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "occupancy"), ("SGPRs Spill", "sgpr_spill"), ("VGPRs Spill", "vgpr_spill"),
          ("LDS Size [bytes/block]", "lds"))


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout
            return [re.sub(r"\(.*$", "", ln.replace("void ", "")).strip() for ln in out.splitlines()]
        except (OSError, subprocess.CalledProcessError):
            continue
    return list(names)


def reduce_log(text):
    rows, cur = [], None
    for ln in text.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", ln)
        if m:
            cur = {"mangled": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for label, key in FIELDS:
            m = re.search(r"remark: +" + re.escape(label) + r": (\d+)", ln)
            if m:
                cur[key] = int(m.group(1))
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = n
    return rows


def render(rows):
    out = ["# kernel | SGPRs | VGPRs | AGPRs | scratch B/lane | occupancy waves/SIMD | VGPR spills | LDS B/block   (hipcc "
           "-Rpass-analysis=kernel-resource-usage, gfx950; build.sh)"]
    for r in sorted(rows, key=lambda r: r["name"]):
        out.append("%s | %d | %d | %d | %d | %d | %d | %d" % (r["name"], r.get("sgprs", -1), r.get("vgprs", -1), r.get("agprs", -1),
                                                         r.get("scratch", -1), r.get("occupancy", -1), r.get("vgpr_spill", -1),
                                                         r.get("lds", -1)))
    return "\n".join(out) + "\n"


def parse(path):
    """{kernel name: dict(sgprs, vgprs, agprs, scratch, occupancy, vgpr_spill, lds)} of a rendered table."""
    res = {}
    for ln in open(path):
        if ln.startswith("#") or "|" not in ln:
            continue
        p = [x.strip() for x in ln.split("|")]
        res[p[0]] = dict(zip(("sgprs", "vgprs", "agprs", "scratch", "occupancy", "vgpr_spill", "lds"), (int(x) for x in p[1:8])))
    return res


if __name__ == "__main__":
    sys.stdout.write(render(reduce_log(open(sys.argv[1]).read())))
