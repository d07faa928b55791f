"""CPU-only checks of the host side: the C-ABI library loads and exports every
symbol include/twx.h declares, and fails loudly without a GPU (no fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from topowx_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call([os.path.join(ROOT, "build.sh")])
    return _lib


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "twx.h")).read()
    declared = set(re.findall(r"\b(twx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"twx_ctx"}
    assert declared == set(built.EXPORTS)
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.twx_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.twx_version()         # pure host call, no GPU needed


def test_struct_layouts_match_header(built):
    # sizes the C side compiles to (LP64): any drift between twx.h and _lib.py shows here
    assert ctypes.sizeof(built.TwxParams) == 32
    assert ctypes.sizeof(built.TwxStationTable) == 8 * 13
    assert ctypes.sizeof(built.TwxPt) == 8 * 16 == built.PT_DTYPE.itemsize
    assert ctypes.sizeof(built.TwxGrid) == 8 + 8 * 8
    assert ctypes.sizeof(built.TwxGridOut) == 8 * 8
    assert ctypes.sizeof(built.TwxTiming) == 4 * 7 + 4 + 8 * 6 + 4 * 2
    assert ctypes.sizeof(built.TwxDeflated) == 8 * 2 + 8 * 2 + 4 * 4 and [f[0] for f in built.TwxTiming._fields_][-1] == "deflate_ms"


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built.TwxError):
        built.Context()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "topowx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "twx_oracle" not in src and "libtwxoracle" not in src and "import deflate_oracle" not in src, f


def test_days_metadata():
    import datetime as dt
    from topowx_amd.dates import MONTH, YEAR, get_days_metadata, get_mth_metadata
    d = get_days_metadata(dt.date(1948, 1, 1), dt.date(2016, 12, 31))
    assert d.size == 25203
    cnt = [int((d[MONTH] == m).sum()) for m in range(1, 13)]
    assert min(cnt) == 1950 and max(cnt) == 2139
    assert d[YEAR][0] == 1948 and d.YMD[-1] == 20161231
    assert get_mth_metadata(1981, 2010).size == 360


def test_synth_is_deterministic():
    from topowx_amd import synth
    g1, a1, _ = synth.make_case("C1")
    g2, a2, _ = synth.make_case("C1")
    assert np.array_equal(g1["elev"], g2["elev"]) and a1.stns.tobytes() == a2.stns.tobytes()
    assert g1["lat"][0] > g1["lat"][-1]                      # north-up (step25:113-116)
    assert a1.stns.size == 500


def test_xcd_contiguous_order_is_a_bijection():
    """topowx_amd/csrc/twx_device.h::xcd_contig -- work-group -> list position such that each of the 8 XCDs (work-groups
    are dealt round-robin) takes a contiguous eighth of the list: every position exactly once, -1 for the surplus
    work-groups of a launch rounded up to a multiple of 8, and XCD x's positions form the range [x*per, (x+1)*per)."""
    src = open(os.path.join(ROOT, "topowx_amd", "csrc", "twx_device.h")).read()
    assert "const int per = (n + 7) >> 3;" in src and "(int)(wg & 7u) * per + (int)(wg >> 3)" in src   # the map restated below

    def xcd_contig(wg, n):
        per = (n + 7) >> 3
        it = (wg & 7) * per + (wg >> 3)
        return it if (wg >> 3) < per and it < n else -1

    for n in (1, 7, 8, 9, 63, 64, 65, 1000, 128375):
        grid = (n + 7) // 8 * 8
        got = [xcd_contig(w, n) for w in range(grid)]
        assert sorted(g for g in got if g >= 0) == list(range(n))
        per = (n + 7) >> 3
        for x in range(8):
            mine = [g for w, g in enumerate(got) if w % 8 == x and g >= 0]
            assert mine == list(range(x * per, min(n, (x + 1) * per)))


def test_precision_policy_decides_once_on_the_median_of_three_tiles():
    """driver.PrecisionPolicy (host logic, no GPU): "auto" starts exact and keeps it when the MEDIAN device time of the first three
    exact tiles hides behind the median copy-out time -- one slow tile (workspace growth) does not flip the run --, falls back to fast
    otherwise, and never looks again; "fast" / "exact" are taken as they are; the context is handed back in the fast mode."""
    from topowx_amd import driver

    class Ctx(object):
        def __init__(self):
            self.calls = []

        def set_precision(self, mode):
            self.calls.append(mode)

    c = Ctx()
    p = driver.PrecisionPolicy(c, "auto")
    assert p.mode == "exact" and c.calls == ["exact"]
    for k, (dev, cp) in enumerate(((546.0, 114.0), (52.0, 112.0), (51.0, 111.0))):     # the first tile grew the fp64 slabs
        assert p.mode == "exact"
        p.observe("exact", dev, cp, tile=k)
    assert p.decided and p.mode == "exact" and p.decision.startswith("exact throughout")
    p.observe("exact", 900.0, 1.0, tile=3)                       # decided: later tiles change nothing
    assert p.mode == "exact" and p.summary()["tiles_exact"] == 4 and p.summary()["tile_modes"] == {0: "exact", 1: "exact", 2: "exact", 3: "exact"}
    p.close()
    assert c.calls[-1] == "fast"
    c = Ctx()
    p = driver.PrecisionPolicy(c, "auto")                        # normals-only tiles: kernels 2.5 ms, copy-out 0.1 ms
    for k in range(3):
        p.observe("exact", 2.5, 0.1, tile=k)
    assert p.mode == "fast" and c.calls == ["exact", "fast"] and p.decision.startswith("fast after 3 tiles")
    p.observe("fast", 2.0, 0.1, tile=3)
    s = p.summary()
    assert s["precision"] == "fast" and s["tiles_exact"] == 3 and s["tiles_fast"] == 1 and s["requested"] == "auto"
    for req in ("fast", "exact"):
        c = Ctx()
        p = driver.PrecisionPolicy(c, req)
        p.observe(req, 100.0, 1.0)
        assert p.mode == req and p.decided and c.calls == [req] and p.decision == "as requested"
    with pytest.raises(ValueError):
        driver.PrecisionPolicy(Ctx(), "double")


def test_streamed_driver_loop_on_a_fake_stream():
    """driver.interp_tiles_streamed's host logic without a GPU (a fake context / stream): every tile reaches the sink once, in order
    with one writer; the stream is asked for with keep=True and handed back open; two writers get two more slots; deflate_chunks
    and the precision reach the stream / context; a failing sink stops the run early and is reported, the stream survives; a
    failing submit closes the stream (its state is unknown); the trace holds one submit and one wait per tile."""
    import threading
    from topowx_amd import driver

    class Stream(object):
        def __init__(self, ctx, **kw):
            self.ctx, self.kw, self.open, self.inflight, self.fail_at = ctx, kw, True, {}, None
            self.nsub = 0

        def submit(self, slot, grid, rows, cols):
            assert self.open and slot not in self.inflight and 0 <= slot < self.kw["nslots"]
            if self.fail_at is not None and self.nsub == self.fail_at:
                raise RuntimeError("device lost")
            self.inflight[slot] = (rows.start, cols.start, self.ctx.mode)
            self.nsub += 1

        def wait(self, slot):
            r, c, mode = self.inflight.pop(slot)
            out = {"status": np.zeros((2, 2), np.int32), "origin": np.array([r, c]), "device_ms": 5.0}
            if self.kw.get("deflate_chunks"):
                out["deflated_tmin"], out["deflate_chunks"] = [b"x"], self.kw["deflate_chunks"]
            return out

        def times(self, slot):
            return 5.0, 20.0

        def close(self):
            self.open = False

    class Ctx(object):
        def __init__(self):
            self.mode, self.streams, self.asked = None, {}, []

        def set_precision(self, mode):
            self.mode = mode

        def stream(self, Y, X, variables=("tmin", "tmax"), daily=False, nslots=2, deflate_chunks=None, keep=False):
            self.asked.append(keep)
            key = (Y, X, daily, nslots, deflate_chunks)
            if key not in self.streams or not self.streams[key].open:
                self.streams[key] = Stream(self, nslots=nslots, deflate_chunks=deflate_chunks)
            return self.streams[key]

    tiles = [(k, 10 * k, 20 * k, 4) for k in range(9)]
    ctx = Ctx()
    seen, lock = [], threading.Lock()

    def sink(k, arrays):
        with lock:
            seen.append((k, int(arrays["origin"][0]), int(arrays["origin"][1])))
    trace, log = [], {}
    res, secs, dev_ms = driver.interp_tiles_streamed(ctx, None, tiles, 2, 2, daily=True, sink=sink, precision="exact", trace=trace, log=log)
    assert res is None and seen == [(k, 10 * k, 20 * k) for k in range(9)] and dev_ms == 45.0 and secs > 0
    assert ctx.asked == [True] and ctx.mode == "fast" and log["tiles_exact"] == 9          # handed back in the fast mode
    (st,) = ctx.streams.values()
    assert st.open and not st.inflight and st.kw["nslots"] == 3
    assert sorted(t[1] for t in trace) == ["submit"] * 9 + ["wait"] * 9
    # again: the same stream; two writers: two more slots, another stream; the default sink collects copies (deflated lists too)
    driver.interp_tiles_streamed(ctx, None, tiles[:2], 2, 2, daily=True, sink=sink, precision="fast")
    assert len(ctx.streams) == 1 and st.nsub == 11
    got, _, _ = driver.interp_tiles_streamed(ctx, None, tiles[:4], 2, 2, daily=True, precision="fast", writer_threads=2, deflate_chunks=(1, 2))
    assert len(ctx.streams) == 2 and sorted(got) == [0, 1, 2, 3] and got[2]["deflated_tmin"] == [b"x"] and "deflate_chunks" not in got[2]
    st2 = [s for s in ctx.streams.values() if s is not st][0]
    assert st2.kw == {"nslots": 4, "deflate_chunks": (1, 2)} and st2.open
    # a failing sink: reported, the run stops early, the stream is still good
    del seen[:]

    def bad(k, arrays):
        seen.append(k)
        if k == 2:
            raise IOError("disk full")
    with pytest.raises(IOError, match="disk full"):
        driver.interp_tiles_streamed(ctx, None, tiles, 2, 2, daily=True, sink=bad, precision="fast")
    assert seen[:3] == [0, 1, 2] and len(seen) < 9 and st.open and not st.inflight
    # a failing submit: the exception reaches the caller and the stream is closed, the next run gets a new one
    st.fail_at = st.nsub + 3
    with pytest.raises(RuntimeError, match="device lost"):
        driver.interp_tiles_streamed(ctx, None, tiles, 2, 2, daily=True, sink=sink, precision="fast")
    assert not st.open
    driver.interp_tiles_streamed(ctx, None, tiles[:1], 2, 2, daily=True, sink=sink, precision="fast")
    assert [s for s in ctx.streams.values() if s.kw["nslots"] == 3][0] is not st
