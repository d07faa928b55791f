"""What does the GPU box's HOST give a writer of new file pages?  (The ceiling of ncio.TileSink: bench.py's c4 sink record.)
Fills fresh files under DIR (default /dev/shm) in several ways and prints GB/s for each:
    python3 tests/tools/host_page_rates.py [DIR] [GB per test]"""
import json
import mmap
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

base = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
GB = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
out = {"dir": base, "cpus": os.cpu_count(), "GB_per_test": GB}
src = np.full(64 << 20, 7, np.uint8)


def paths(n):
    return [os.path.join(base, "twx_rate_%d_%d.bin" % (os.getpid(), i)) for i in range(n)]


def timed(name, fn, nbytes):
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    out[name] = round(nbytes / dt / 1e9, 2)
    print("%-58s %8.2f GB/s" % (name, out[name]), flush=True)


def mmap_fill(nfiles, nthreads, warm=False, falloc=False):
    n = int(GB * 1e9) // (nfiles * nthreads * 4096) * (nfiles * nthreads * 4096)
    ps = paths(nfiles)
    per = n // nfiles
    mms = []
    t_f = 0.0
    for p in ps:
        fd = os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC)
        if falloc:
            t0 = time.perf_counter()
            os.posix_fallocate(fd, 0, per)
            t_f += time.perf_counter() - t0
        else:
            os.ftruncate(fd, per)
        mms.append(np.memmap(p, dtype=np.uint8, mode="r+"))
        os.close(fd)
    part = per // nthreads
    jobs = [(m, i * part, (i + 1) * part) for m in mms for i in range(nthreads)]

    def run():
        with ThreadPoolExecutor(nfiles * nthreads) as pool:
            list(pool.map(lambda j: j[0][j[1]:j[2]].fill(3), jobs))
    tag = "mmap fill, %d file(s) x %d thread(s)" % (nfiles, nthreads)
    if falloc:
        out["posix_fallocate, %d file(s), one thread" % nfiles] = round(n / t_f / 1e9, 2)
        print("%-58s %8.2f GB/s" % ("posix_fallocate, %d file(s), one thread" % nfiles, n / t_f / 1e9), flush=True)
        tag += " after fallocate"
    timed(tag + (" (cold)" if not falloc else ""), run, n)
    if warm:
        timed(tag + " (again: warm)", run, n)
    del mms, jobs
    for p in ps:
        os.remove(p)


def pwrite_fill(nfiles, nthreads):
    n = int(GB * 1e9) // (nfiles * nthreads * src.size) * (nfiles * nthreads * src.size)
    ps = paths(nfiles)
    fds = [os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC) for p in ps]
    per = n // nfiles
    part = per // nthreads
    jobs = [(fd, i * part, (i + 1) * part) for fd in fds for i in range(nthreads)]

    def one(j):
        fd, a, b = j
        for off in range(a, b, src.size):
            os.pwrite(fd, src, off)

    def run():
        with ThreadPoolExecutor(nfiles * nthreads) as pool:
            list(pool.map(one, jobs))
    timed("pwrite 64 MB blocks, %d file(s) x %d thread(s) (cold)" % (nfiles, nthreads), run, n)
    for fd, p in zip(fds, ps):
        os.close(fd)
        os.remove(p)


def anon_fill(nthreads):
    n = int(GB * 1e9) // (nthreads * 4096) * (nthreads * 4096)
    part = n // nthreads
    bufs = [mmap.mmap(-1, part) for _ in range(nthreads)]
    arrs = [np.frombuffer(b, np.uint8) for b in bufs]

    def run():
        with ThreadPoolExecutor(nthreads) as pool:
            list(pool.map(lambda a: a.fill(3), arrs))
    timed("anonymous memory fill, %d thread(s) (cold)" % nthreads, run, n)
    timed("anonymous memory fill, %d thread(s) (warm)" % nthreads, run, n)


mmap_fill(1, 1, warm=True)
mmap_fill(1, 8)
mmap_fill(1, 64)
mmap_fill(8, 1)
mmap_fill(8, 8)
mmap_fill(64, 1)
pwrite_fill(1, 1)
pwrite_fill(1, 8)
pwrite_fill(8, 1)
pwrite_fill(64, 1)
mmap_fill(1, 8, falloc=True)
mmap_fill(8, 8, falloc=True)
anon_fill(1)
anon_fill(64)
try:
    out["thp_shmem_enabled"] = open("/sys/kernel/mm/transparent_hugepage/shmem_enabled").read().strip()
    out["thp_enabled"] = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
except OSError:
    pass
out["meminfo"] = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/meminfo") if l.split(":")[0] in ("MemTotal", "MemFree", "Shmem", "HugePages_Total", "AnonHugePages", "ShmemHugePages")}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/host_page_rates.json", "w"), indent=1)
print(json.dumps(out))
