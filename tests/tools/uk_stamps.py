"""Reduce the s_memtime stamps of the diagnostic k_uk<7> build (tests/tools/uk_stamps.sh): per-panel durations of
the holder's chain + row solve, of the rank-4 update per wave, and of the whole panel period (100 MHz ticks? no:
s_memtime counts shader clocks on gfx950, MI355X_MICROARCH.md)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(2048, 40, 4, 4).astype(np.int64)
used = a[:, :, :, 0].max(axis=(1, 2)) > 0
a = a[used]
nw = int((a[:, :, :, 0].max(axis=(0, 1)) > 0).sum())      # waves per work-group of the stamped kernel (2 or 4)
a = a[:, :, :nw]
print("work-groups with stamps:", a.shape[0], " waves per work-group:", nw)
t0 = a[:, :, :, 0]            # barrier passed
t1 = a[:, :, :, 1]            # update done
t2 = a[:, :, :, 2].max(axis=2)   # holder: chain begins (only the holder's entry is non-zero)
t3 = a[:, :, :, 3].max(axis=2)   # holder: slab written
npan = (t0.max(axis=2) > 0).sum(axis=1)
print("panels per system: mean %.1f" % npan.mean())
rows = []
for w in range(a.shape[0]):
    n = npan[w]
    if n < 3:
        continue
    bar = t0[w, :n].max(axis=1)                 # last wave through the barrier
    upd = (t1[w, :n] - t0[w, :n])               # [panel, wave]
    chain = t3[w, :n] - t2[w, :n]
    period = np.diff(bar)
    # wait of the holder of panel p+1 between its update end (panel p) and its chain start: ~0 by construction
    wait_for_holder = bar[1:] - t3[w, 1:n]      # barrier release after the slab is written
    rows.append((period.mean(), chain[1:].mean(), upd.mean(), upd.max(axis=1).mean(), upd.min(axis=1).mean(),
                 wait_for_holder.mean(), (t2[w, 1:n] - bar[:-1]).mean(), (bar[-1] - bar[0]) / max(n - 1, 1), n))
r = np.array(rows)
names = ("panel period", "holder: publish + chain + row solve", "update (mean over waves)", "update (slowest wave)",
         "update (fastest wave)", "slab written -> barrier released", "barrier -> holder of the next panel starts its chain",
         "first..last barrier / panels", "panels")
for i, nme in enumerate(names):
    print("%-58s mean %8.0f   p10 %8.0f   p90 %8.0f  clocks" % (nme, r[:, i].mean(), np.percentile(r[:, i], 10), np.percentile(r[:, i], 90)))
