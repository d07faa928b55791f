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


@pytest.mark.parametrize("nd,cy,cx,rough", [(1, 3, 3, 100), (40, 10, 10, 50), (700, 10, 10, 800), (366, 5, 7, 10), (1400, 8, 6, 30)])
def test_streams_inflate_to_the_shuffled_chunk(nd, cy, cx, rough):
    d = _field(np.random.default_rng(nd), nd, cy, cx, rough)
    s = dorc.deflate_chunk(d)
    lo, hi = dorc.shuffled(d)
    assert zlib.decompress(s) == lo.tobytes() + hi.tobytes()         # (zlib verifies the Adler-32)
    assert np.array_equal(dorc.inflate_chunk(s, nd, cy, cx), d)
    assert s[:2] == b"\x78\x01" and s[-9:-4] == b"\x01\x00\x00\xff\xff"
    n = lo.size
    assert len(s) <= 2 + n + 5 * -(-n // 65535) + n + 5 * -(-n // dorc.SEG) + 9      # a segment without runs is stored
    if rough <= 50 and nd >= 40:
        assert len(s) < 0.8 * d.nbytes                               # smooth fields: the high plane is runs


def test_a_constant_chunk_and_a_noise_chunk():
    flat = np.full((300, 10, 10), 1234, np.int16)
    s = dorc.deflate_chunk(flat)
    assert np.array_equal(dorc.inflate_chunk(s, 300, 10, 10), flat) and len(s) < 0.52 * flat.nbytes
    noise = np.random.default_rng(0).integers(-32768, 32767, (300, 10, 10)).astype(np.int16)
    s = dorc.deflate_chunk(noise)
    assert np.array_equal(dorc.inflate_chunk(s, 300, 10, 10), noise) and len(s) < 1.001 * noise.nbytes + 64


def test_every_match_length_decodes():
    """A fixed-Huffman block 'literal c, match(L, 1)' inflates to L + 1 copies of c, for every length of RFC 1951 3.2.5."""
    for L in range(3, 259):
        bits = dorc._Bits()
        bits.put(1, 1)                                          # BFINAL
        bits.put(1, 2)
        dorc._fixed_symbol(bits, 200)
        dorc._match(bits, L)
        dorc._fixed_symbol(bits, 256)
        bits.align()
        assert zlib.decompressobj(-15).decompress(bytes(bits.out)) == bytes([200]) * (L + 1), L


@pytest.fixture(scope="module")
def codes_exe(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = str(tmp_path_factory.mktemp("dfl") / "deflate_codes")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "topowx_amd", "csrc"),
                    os.path.join(ROOT, "tests", "tools", "deflate_codes.hip"), "-o", exe], check=True, capture_output=True)
    return exe


def test_device_token_codes_equal_the_rfc_tables(codes_exe):
    lines = subprocess.run([codes_exe], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for ln in lines:
        f = ln.split()
        if not f or f[0] not in "LM":
            continue
        bits = dorc._Bits()
        if f[0] == "L":
            dorc._fixed_symbol(bits, int(f[1]))
        else:
            dorc._match(bits, int(f[1]))
        n = bits.n + 8 * len(bits.out)
        val = int.from_bytes(bytes(bits.out), "little") | (bits.acc << (8 * len(bits.out)))
        assert (int(f[2]), int(f[3])) == (n, val), ln
        seen += 1
    assert seen == 256 + 256
    geom = [ln.split() for ln in lines if ln.startswith("S ")][0]
    assert [int(v) for v in geom[1:4]] == [dorc.PIECE, dorc.SEG, dorc.SEG * 9 // 8 + 8]


def _byte_cases():
    rng = np.random.default_rng(7)
    _, hi = dorc.shuffled(_field(rng, 500, 6, 9, 30))
    runs = np.repeat(rng.integers(0, 256, 400), rng.integers(1, 200, 400)).astype(np.uint8)      # runs of 1..199, every byte value
    return {"field": hi[:20000], "zeros": np.zeros(1000, np.uint8), "constant_255": np.full(777, 255, np.uint8),
            "noise": rng.integers(0, 256, 5001).astype(np.uint8), "runs": runs[:30001],
            "pairs": np.repeat(rng.integers(140, 150, 3000), 2).astype(np.uint8)[:5999],                # runs of 2 around the 8 / 9-bit border
            "one": np.array([144], np.uint8), "short": np.array([7, 7, 7, 7, 9], np.uint8)}


@pytest.mark.parametrize("case", ["field", "zeros", "constant_255", "noise", "runs", "pairs", "one", "short"])
def test_device_tokenizer_equals_the_restatement(codes_exe, tmp_path, case):
    """df_piece (the function every GPU thread runs on its piece of bytes): the same bit stream as the restatement's block body, on
    the high plane of a field and on bytes chosen against its masks (runs across pieces and 64-bit windows, every byte value,
    a zero first byte with nothing before it, lengths that are no multiple of 4 or of the piece)."""
    hi = _byte_cases()[case]
    p = tmp_path / "hi.bin"
    p.write_bytes(hi.tobytes())
    out = subprocess.run([codes_exe, str(p)], check=True, capture_output=True, text=True).stdout.split("\n")
    bits = dorc._Bits()
    bits.put(0, 1)
    bits.put(1, 2)
    total = 0
    for ln in out:
        f = ln.split()
        if f and f[0] == "T":
            bits.put(int(f[2]), int(f[1]))
        elif f and f[0] == "P":
            total += int(f[1])
            assert f[1] == f[2], ln                             # the counting pass (masks only) == the emitting pass
    dorc._fixed_symbol(bits, 256)
    bits.put(0, 3)
    bits.align()
    assert bytes(bits.out) + b"\x00\x00\xff\xff" == dorc._huffman_block(hi, 0, hi.size)
    assert (3 + total + 7 + 3 + 7) // 8 + 4 == len(dorc._huffman_block(hi, 0, hi.size))      # df_huff_bytes
