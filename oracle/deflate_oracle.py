"""CPU restatement of topowx_amd/csrc/twx_deflate.h -- TEST INFRASTRUCTURE ONLY (tests/, bench.py's checks): the chunk bytes of an
HDF5 dataset with the shuffle + deflate filters as the GPU forms them, byte for byte.

The reference reaches this format through netCDF4-python's ``createVariable(..., zlib=True)`` (twx/interp/tiling.py:720,894,913,
1035: HDF5 shuffle, then zlib's deflate); which deflate stream a writer emits is its own choice -- any RFC 1950 / 1951 stream that
inflates to the shuffled chunk is the same file content.  Parity is therefore pinned twice: (1) ``zlib.decompress`` (the
reference's own decoder, inside libhdf5) of every stream equals the shuffled chunk -- no restatement involved; (2) the GPU's
bytes equal this restatement's, so a regression of the encoder shows as a diff, not only as a corrupt file.

Stream: ``78 01``; the low-byte plane in stored blocks of <= 65535 bytes; the high-byte plane in fixed-Huffman blocks of 16 384
input bytes, encoded in pieces of 64 bytes (literal / match(length 3..64, distance 1)), each block closed by end-of-block and
an empty stored block (byte alignment) -- or stored, when that is shorter; ``01 00 00 FF FF``; Adler-32, big endian.  Pure-Python loops: small cases only."""
import zlib

import numpy as np

PIECE, SEG, STORED = 64, 64 * 256, 65535

_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]      # RFC 1951, 3.2.5


class _Bits(object):
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):                   # value's bit 0 first (data elements, RFC 1951 3.1.1)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def huff(self, code, nbits):                   # Huffman codes go most significant bit first
        self.put(int(format(code, "0%db" % nbits)[::-1], 2), nbits)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def _fixed_symbol(bits, sym):                      # RFC 1951, 3.2.6
    if sym < 144:
        bits.huff(0x30 + sym, 8)
    elif sym < 256:
        bits.huff(0x190 + sym - 144, 9)
    elif sym < 280:
        bits.huff(sym - 256, 7)
    else:
        bits.huff(0xC0 + sym - 280, 8)


def _match(bits, length):
    k = max(i for i, b in enumerate(_LBASE) if b <= length)
    if length == 258:
        k = 28
    _fixed_symbol(bits, 257 + k)
    if _LEXTRA[k]:
        bits.put(length - _LBASE[k], _LEXTRA[k])
    bits.put(0, 5)                                 # distance code 0 = distance 1, no extra bits


def _huffman_block(hi, start, stop):
    """One fixed-Huffman block over hi[start:stop], pieces of 64 bytes, then end-of-block + empty stored block."""
    bits = _Bits()
    bits.put(0, 1)                                 # BFINAL
    bits.put(1, 2)                                 # BTYPE = 01
    for p0 in range(start, stop, PIECE):
        p1 = min(stop, p0 + PIECE)
        prev = int(hi[p0 - 1]) if p0 > 0 else 256
        i = p0
        while i < p1:
            c = int(hi[i])
            if c == prev:
                r = 1
                while i + r < p1 and r < 258 and hi[i + r] == c:
                    r += 1
                if r >= 3:
                    _match(bits, r)
                    i += r
                    continue
            _fixed_symbol(bits, c)
            prev = c
            i += 1
    _fixed_symbol(bits, 256)                       # end of block
    bits.put(0, 3)                                 # BFINAL = 0, BTYPE = 00: an empty stored block aligns the stream
    bits.align()
    return bytes(bits.out) + b"\x00\x00\xff\xff"


def shuffled(chunk):
    """HDF5's shuffle filter on an int16 chunk: all first (low) bytes, then all second (high) bytes, element order as stored."""
    b = np.ascontiguousarray(chunk, "<i2").reshape(-1).view(np.uint8).reshape(-1, 2)
    return np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])


def deflate_chunk(chunk):
    """int16 [ndays, cy, cx] -> the zlib stream twx_deflate.h emits for it."""
    lo, hi = shuffled(chunk)
    n = lo.size
    out = bytearray(b"\x78\x01")
    for b0 in range(0, n, STORED):
        blk = lo[b0:b0 + STORED].tobytes()
        out += bytes([0, len(blk) & 255, len(blk) >> 8, ~len(blk) & 255, (~len(blk) >> 8) & 255]) + blk
    for s0 in range(0, n, SEG):
        s1 = min(n, s0 + SEG)
        blk = _huffman_block(hi, s0, s1)
        if len(blk) > s1 - s0 + 5:                 # no runs worth coding: the segment as a stored block
            m = s1 - s0
            blk = bytes([0, m & 255, m >> 8, ~m & 255, (~m >> 8) & 255]) + hi[s0:s1].tobytes()
        out += blk
    out += b"\x01\x00\x00\xff\xff"
    out += zlib.adler32(lo.tobytes() + hi.tobytes()).to_bytes(4, "big")
    return bytes(out)


def deflate_tile(daily, cy, cx):
    """int16 [ndays, Y, X] -> list of streams, chunks in row-major order (the order H5Dwrite_chunk is called in)."""
    nd, Y, X = daily.shape
    return [deflate_chunk(daily[:, r0:r0 + cy, c0:c0 + cx]) for r0 in range(0, Y, cy) for c0 in range(0, X, cx)]


def inflate_chunk(stream, nd, cy, cx):
    """The int16 chunk a reader gets: inflate, unshuffle."""
    raw = np.frombuffer(zlib.decompress(stream), np.uint8)
    n = nd * cy * cx
    assert raw.size == 2 * n
    return np.stack([raw[:n], raw[n:]], axis=1).reshape(-1).view("<i2").reshape(nd, cy, cx)
