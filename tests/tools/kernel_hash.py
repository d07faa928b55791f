"""sha256 (first 16 hex digits) over the kernel sources (topowx_amd/csrc/*, include/twx.h): stamped into
profiles/r*_bench_hbm_traffic.json when the PMC passes are reduced (reduce_profiles.py), so that bench.py and
tests/test_profiles_fresh.py can tell whether the committed traffic figures belong to the kernels in the tree."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_sources_sha16(root=ROOT):
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(root, "topowx_amd", "csrc", "*")) + [os.path.join(root, "include", "twx.h")]):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_sources_sha16())
