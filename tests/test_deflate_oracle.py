"""CPU: the restatement of csrc/twx_deflate.h (oracle/deflate_oracle.py) against zlib -- the decoder inside libhdf5, i.e. what
reads the reference's ``zlib=True`` products (tiling.py:720,894,913,1035) -- and the token codes the device functions emit
(compiled for the host from the same header) against the restatement's RFC 1951 tables."""
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest

from oracle import deflate_oracle as dorc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _field(rng, nd, cy, cx, rough):
    base = rng.normal(0, rough, (1, cy, cx)) + rng.normal(0, 300, (nd, 1, 1)) + 1500 * np.sin(np.arange(nd) / 58.0)[:, None, None]
    d = np.rint(base + rng.normal(0, 20, (nd, cy, cx))).astype(np.int16)
    d[:, :1, :2] = -32767                                    # fill values (masked cells)
    return d


def _one(d):
    """the stream of a one-chunk tile"""
    return dorc.deflate_tile(d, d.shape[1], d.shape[2])[0]


@pytest.mark.parametrize("nd,cy,cx,rough", [(1, 3, 3, 100), (40, 10, 10, 50), (700, 10, 10, 800), (366, 5, 7, 10), (1400, 8, 6, 30)])
def test_streams_inflate_to_the_shuffled_chunk(nd, cy, cx, rough):
    d = _field(np.random.default_rng(nd), nd, cy, cx, rough)
    s = _one(d)
    lo, hi = dorc.shuffled(d)
    assert zlib.decompress(s) == lo.tobytes() + hi.tobytes()         # (zlib verifies the Adler-32)
    assert np.array_equal(dorc.inflate_chunk(s, nd, cy, cx), d)
    assert s[:2] == b"\x78\x01" and s[-9:-4] == b"\x01\x00\x00\xff\xff"
    n = lo.size
    assert len(s) <= 2 + n + 5 * -(-n // 65535) + n + 5 * -(-n // dorc.SEG) + 9      # a segment without runs is stored
    if rough <= 50 and nd >= 40:
        assert len(s) < 0.75 * d.nbytes                              # smooth fields: the high plane is runs
    if nd == 1400:                                                   # zlib level 1 on the same shuffled bytes is no smaller than + 3 %
        assert len(s) < 1.03 * len(zlib.compress(lo.tobytes() + hi.tobytes(), 1))


def test_a_tile_of_several_chunks_shares_one_code():
    d = _field(np.random.default_rng(3), 900, 8, 6, 25)
    blobs = dorc.deflate_tile(d, 4, 3)
    assert len(blobs) == 4
    for blob, chunk in zip(blobs, dorc._chunks(d, 4, 3)):
        assert np.array_equal(dorc.inflate_chunk(blob, 900, 4, 3), chunk)
    table = dorc.tile_table(d, 4, 3)
    assert blobs[2] == dorc.deflate_chunk(dorc._chunks(d, 4, 3)[2], table)
    lens = table[0]
    assert len(lens) == dorc.NSYM and min(lens) >= 1 and max(lens) <= 15 and sum(2.0 ** -l for l in lens) == 1.0    # a complete code


def test_a_constant_chunk_and_a_noise_chunk():
    flat = np.full((300, 10, 10), 1234, np.int16)
    s = _one(flat)
    assert np.array_equal(dorc.inflate_chunk(s, 300, 10, 10), flat) and len(s) < 0.51 * flat.nbytes
    noise = np.random.default_rng(0).integers(-32768, 32767, (300, 10, 10)).astype(np.int16)
    s = _one(noise)
    assert np.array_equal(dorc.inflate_chunk(s, 300, 10, 10), noise) and len(s) < 1.001 * noise.nbytes + 64


def test_huffman_lengths_are_limited_and_complete():
    rng = np.random.default_rng(1)
    for n, limit in ((277, 15), (19, 7), (2, 7), (1, 7), (40, 15)):
        for scale in (1, 1000, 10 ** 8):
            cnt = [int(c) for c in 1 + rng.integers(0, scale, n) * rng.integers(0, 2, n)]
            fib = [1, 1]
            while len(fib) < n:
                fib.append(fib[-1] + fib[-2])                    # the counts that make the deepest tree
            for counts in (cnt, fib[:n]):
                lens = dorc.huff_lengths(counts, limit)
                assert max(lens) <= limit and min(lens) >= 1
                assert n == 1 or sum(2.0 ** -l for l in lens) == 1.0
                codes = dorc.canonical(lens)
                assert len({(c, l) for c, l in zip(codes, lens)}) == n


@pytest.fixture(scope="module")
def codes_exe(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = str(tmp_path_factory.mktemp("dfl") / "deflate_codes")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "topowx_amd", "csrc"),
                    os.path.join(ROOT, "tests", "tools", "deflate_codes.hip"), "-o", exe], check=True, capture_output=True)
    return exe


def test_device_length_symbols_equal_the_rfc_table(codes_exe):
    lines = subprocess.run([codes_exe], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for ln in lines:
        f = ln.split()
        if f and f[0] == "M":
            L = int(f[1])
            k = max(j for j, b in enumerate(dorc._LBASE) if b <= L)
            assert [int(v) for v in f[2:]] == [257 + k, L - dorc._LBASE[k], dorc._LEXTRA[k]], ln
            seen += 1
    assert seen == dorc.PIECE
    geom = [ln.split() for ln in lines if ln.startswith("S ")][0]
    assert [int(v) for v in geom[1:5]] == [dorc.PIECE, dorc.SEG, dorc.NSYM, dorc.SAMPLE]


def _byte_cases():
    rng = np.random.default_rng(7)
    _, hi = dorc.shuffled(_field(rng, 500, 6, 9, 30))
    runs = np.repeat(rng.integers(0, 256, 400), rng.integers(1, 200, 400)).astype(np.uint8)      # runs of 1..199, every byte value
    return {"field": hi[:20000], "zeros": np.zeros(1000, np.uint8), "constant_255": np.full(777, 255, np.uint8),
            "noise": rng.integers(0, 256, 5001).astype(np.uint8), "runs": runs[:30001],
            "pairs": np.repeat(rng.integers(140, 150, 3000), 2).astype(np.uint8)[:5999],                # runs of 2
            "one": np.array([144], np.uint8), "short": np.array([7, 7, 7, 7, 9], np.uint8)}


@pytest.mark.parametrize("case", ["field", "zeros", "constant_255", "noise", "runs", "pairs", "one", "short"])
def test_device_tokenizer_equals_the_restatement(codes_exe, tmp_path, case):
    """df_piece (the function every GPU thread runs on its piece of bytes): the same tokens as the restatement's, on the high plane
    of a field and on bytes chosen against it (runs across pieces, every byte value, a zero first byte with nothing before it,
    lengths that are no multiple of 4 or of the piece)."""
    hi = _byte_cases()[case]
    p = tmp_path / "hi.bin"
    p.write_bytes(hi.tobytes())
    out = subprocess.run([codes_exe, "tokens", str(p)], check=True, capture_output=True, text=True).stdout.split("\n")
    got = [tuple(int(v) for v in ln.split()[1:]) for ln in out if ln.startswith("T ")]
    assert got == dorc.tokens(hi, 0, hi.size)


@pytest.mark.parametrize("kind", ["field", "flat", "empty", "skewed"])
def test_device_table_builder_equals_the_restatement(codes_exe, tmp_path, kind):
    """df_build_table (one GPU thread per variable and tile): code lengths, canonical codes and the block header's bit string
    equal the restatement's, for the counts of a field, of a constant tile, for no counts at all and for counts spanning 2^30
    (the depth limit's halving loop)."""
    rng = np.random.default_rng(11)
    if kind == "field":
        hist = dorc.tile_hist(_field(rng, 800, 8, 6, 25), 4, 3)
    elif kind == "flat":
        hist = dorc.tile_hist(np.full((500, 4, 4), -321, np.int16), 4, 4)
    elif kind == "empty":
        hist = [0] * dorc.NSYM
    else:
        hist = [int(2 ** (30 * rng.random())) for _ in range(dorc.NSYM)]
    p = tmp_path / "hist.txt"
    p.write_text(" ".join(str(h) for h in hist))
    out = subprocess.run([codes_exe, "table", str(p)], check=True, capture_output=True, text=True).stdout.split("\n")
    lens, codes, hdr = dorc.make_table(hist)
    rows = [ln.split() for ln in out if ln.startswith("C ")]
    assert len(rows) == dorc.NSYM
    for f in rows:
        s = int(f[1])
        assert int(f[2]) == lens[s] and int(f[3]) == int(format(codes[s], "0%db" % lens[s])[::-1], 2), f     # (stored bit-reversed)
    nbits = int([ln.split()[1] for ln in out if ln.startswith("H ")][0])
    words = [int(ln.split()[1]) for ln in out if ln.startswith("W ")]
    assert nbits == hdr.nbits()
    val = int.from_bytes(bytes(hdr.out), "little") | (hdr.acc << (8 * len(hdr.out)))
    assert sum(w << (32 * i) for i, w in enumerate(words)) == val


def test_property_any_int16_tile_round_trips():
    """hypothesis: arbitrary int16 tiles (runs, noise, extremes), arbitrary chunk shapes that divide them -> every chunk's stream
    is inflated by zlib to the chunk, never longer than stored + block headers."""
    from hypothesis import given, settings, strategies as stg
    from hypothesis.extra import numpy as hnp

    @settings(max_examples=40, deadline=None)
    @given(stg.data())
    def run(data):
        cy, cx = data.draw(stg.integers(1, 4)), data.draw(stg.integers(1, 5))
        ny, nx, nd = data.draw(stg.integers(1, 3)), data.draw(stg.integers(1, 3)), data.draw(stg.integers(1, 60))
        elems = stg.one_of(stg.integers(-32768, 32767), stg.sampled_from([-32767, 0, 255, 256, -1, 1234]))
        d = data.draw(hnp.arrays(np.int16, (nd, ny * cy, nx * cx), elements=elems))
        if data.draw(stg.booleans()):
            d = np.repeat(d, 3, axis=0)[:nd]                   # runs along the day axis too
        blobs = dorc.deflate_tile(d, cy, cx)
        assert len(blobs) == ny * nx
        n = d.shape[0] * cy * cx
        for blob, chunk in zip(blobs, dorc._chunks(d, cy, cx)):
            assert np.array_equal(dorc.inflate_chunk(blob, d.shape[0], cy, cx), chunk)
            assert len(blob) <= 2 + 2 * n + 5 * (-(-n // 65535) + -(-n // dorc.SEG)) + 9
    run()


@pytest.mark.parametrize("seed", range(6))
def test_device_table_builder_on_random_counts(codes_exe, tmp_path, seed):
    """df_build_table against the restatement on random sparse / dense / heavy-tailed counts (ties, zeros, one dominant symbol)."""
    rng = np.random.default_rng(100 + seed)
    kind = seed % 3
    if kind == 0:
        hist = (rng.integers(0, 50, dorc.NSYM) * (rng.random(dorc.NSYM) < 0.1)).tolist()              # sparse, many ties
    elif kind == 1:
        hist = rng.integers(0, 2 ** 20, dorc.NSYM).tolist()
    else:
        hist = np.floor(rng.pareto(0.4, dorc.NSYM) * 3).clip(0, 2 ** 31 - 2).astype(np.int64).tolist()     # heavy tail: deep trees
    p = tmp_path / "hist.txt"
    p.write_text(" ".join(str(int(h)) for h in hist))
    out = subprocess.run([codes_exe, "table", str(p)], check=True, capture_output=True, text=True).stdout.split("\n")
    lens, codes, hdr = dorc.make_table(hist)
    rows = [ln.split() for ln in out if ln.startswith("C ")]
    assert [int(f[2]) for f in rows] == lens
    assert [int(f[3]) for f in rows] == [int(format(c, "0%db" % l)[::-1], 2) for c, l in zip(codes, lens)]
    assert int([ln.split()[1] for ln in out if ln.startswith("H ")][0]) == hdr.nbits()
