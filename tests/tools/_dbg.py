import sys, os, json
sys.argv=["x","1","5043"]
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
# re-create the case exactly as gpu_soak does (same RNG order) by importing its code up to the data construction
src=open(os.path.join(ROOT,"tests/tools/gpu_soak.py")).read()
head=src[:src.index("    flags = int(rng.choice(")]
head=head.replace("for seed in range(seed0, seed0 + ncase):","for seed in range(seed0, seed0 + 1):")
exec(compile(head,"soak_head","exec"))
from topowx_amd import _lib
for var,code,db in (("tmin",_lib.TMIN,dbs[0]),("tmax",_lib.TMAX,dbs[1])):
    ctx=_lib.Context(batch_cells=512)
    ctx.set_stations(code, db, with_obs=False)
    got=ctx.interp_grid(grid, variables=(var,))
    ctx.close()
    want=orc.interp_grid(orc.Db(db) if var=="tmin" else None, orc.Db(db) if var=="tmax" else None, orc.params(), grid, nthreads=16)
    neq=got["status"]!=want["status"]
    print(var, "stations", db.stns.size, "mismatch cells", int(neq.sum()), "gpu", np.unique(got["status"],return_counts=True), "orc", np.unique(want["status"],return_counts=True))
    if neq.any():
        rr,cc=np.nonzero(neq); print(list(zip(rr[:10],cc[:10])), got["status"][neq][:10], want["status"][neq][:10])
        r,c=rr[0],cc[0]
        print("cell lon/lat", grid["lon"][c], grid["lat"][r], "elev", grid["elev"][r,c], "tdi", grid["tdi"][r,c], "lst", grid["lst_night"][:,r,c])
