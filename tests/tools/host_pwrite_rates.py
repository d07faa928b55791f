import os, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
base="/dev/shm"; GB=6.0
src=np.full(int(3.15e9)//4096*4096, 7, np.uint8)      # one variable's chunk region, warm
def run(nfiles, falloc, block):
    ps=[os.path.join(base,"twx_pw_%d.bin"%i) for i in range(nfiles)]
    fds=[os.open(p, os.O_RDWR|os.O_CREAT|os.O_TRUNC) for p in ps]
    tf=0
    if falloc:
        t0=time.perf_counter()
        for fd in fds: os.posix_fallocate(fd,0,src.size)
        tf=time.perf_counter()-t0
    def one(fd):
        mv=memoryview(src)
        for off in range(0, src.size, block): os.pwrite(fd, mv[off:off+block], off)
    t0=time.perf_counter()
    with ThreadPoolExecutor(nfiles) as pool: list(pool.map(one, fds))
    dt=time.perf_counter()-t0
    for fd,p in zip(fds,ps): os.close(fd); os.remove(p)
    print("pwrite %d file(s) x 1 thread, %s, block %d MB: %.2f GB/s (fallocate %.2f GB/s)"%(nfiles, "after fallocate" if falloc else "cold", block>>20, nfiles*src.size/dt/1e9, (nfiles*src.size/tf/1e9) if tf else 0), flush=True)
for nf in (1,2,4,8):
    for fa in (0,1):
        run(nf, fa, 256<<20)
run(2,1,16<<20)
