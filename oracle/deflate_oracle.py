"""CPU restatement of topowx_amd/csrc/twx_deflate.h -- TEST INFRASTRUCTURE ONLY (tests/, bench.py's checks): the chunk bytes of an
HDF5 dataset with the shuffle + deflate filters as the GPU forms them, byte for byte.

The reference reaches this format through netCDF4-python's ``createVariable(..., zlib=True)`` (twx/interp/tiling.py:720,894,913,
1035: HDF5 shuffle, then zlib's deflate); which deflate stream a writer emits is its own choice -- any RFC 1950 / 1951 stream that
inflates to the shuffled chunk is the same file content.  Parity is therefore pinned twice: (1) ``zlib.decompress`` (the
reference's own decoder, inside libhdf5) of every stream equals the shuffled chunk -- no restatement involved; (2) the GPU's
bytes equal this restatement's, so a regression of the encoder shows as a diff, not only as a corrupt file.

Stream: ``78 01``; the low-byte plane in stored blocks of <= 65535 bytes; the high-byte plane in dynamic-Huffman blocks of 16 384
input bytes, tokenized in pieces of 64 bytes (literal / match(length 3..64, distance 1)) and coded with ONE Huffman code per
(variable, tile) -- built from the token counts of every 16th block of every chunk, + 1 per symbol --, each block closed by
end-of-block and an empty stored block (byte alignment) -- or stored, when that is shorter; ``01 00 00 FF FF``; Adler-32, big
endian.  Pure-Python loops: small cases only."""
import zlib

import numpy as np

PIECE, SEG, STORED, SAMPLE, NSYM = 64, 64 * 256, 65535, 16, 277

_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]      # RFC 1951, 3.2.5
_CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]                           # 3.2.7


class _Bits(object):
    def __init__(self, other=None):
        self.acc, self.n, self.out = (0, 0, bytearray()) if other is None else (other.acc, other.n, bytearray(other.out))

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

    def nbits(self):
        return self.n + 8 * len(self.out)


def tokens(hi, start, stop):
    """[(symbol, extra, number of extra bits)] of hi[start:stop], pieces of PIECE bytes: a byte equal to its predecessor opens a
    run; runs of 3 and more (cut at the end of the piece) are one match with distance 1, everything else literals."""
    out = []
    for p0 in range(start, stop, PIECE):
        p1 = min(stop, p0 + PIECE)
        prev = int(hi[p0 - 1]) if p0 > 0 else 256
        i = p0
        while i < p1:
            c = int(hi[i])
            if c == prev:
                r = 1
                while i + r < p1 and hi[i + r] == c:
                    r += 1
                if r >= 3:
                    k = max(j for j, b in enumerate(_LBASE) if b <= r)
                    out.append((257 + k, r - _LBASE[k], _LEXTRA[k]))
                    i += r
                    continue
            out.append((c, 0, 0))
            prev = c
            i += 1
    return out


def huff_lengths(count, limit):
    """Code lengths (<= limit) of a Huffman code for count[] (all > 0): leaves in (count, symbol) order, two queues, a leaf
    before an internal node of the same weight; a code deeper than ``limit``: every count halved (rounding up), once more."""
    n = len(count)
    count = list(count)
    if n == 1:
        return [1]
    while True:
        order = sorted(range(n), key=lambda s: (count[s], s))
        wt = [count[s] for s in order] + [0] * n
        parent = [0] * (2 * n)
        li, qi, nn = 0, n, n
        while (n - li) + (nn - qi) > 1:
            pick = []
            for _ in range(2):
                if li < n and (qi >= nn or wt[li] <= wt[qi]):
                    pick.append(li)
                    li += 1
                else:
                    pick.append(qi)
                    qi += 1
            wt[nn] = wt[pick[0]] + wt[pick[1]]
            parent[pick[0]] = parent[pick[1]] = nn
            nn += 1
        depth = [0] * nn
        for v in range(nn - 2, -1, -1):
            depth[v] = depth[parent[v]] + 1
        if max(depth[:n]) <= limit:
            lens = [0] * n
            for pos, s in enumerate(order):
                lens[s] = depth[pos]
            return lens
        count = [(c + 1) >> 1 for c in count]


def canonical(lens):
    """RFC 1951 3.2.2"""
    maxl = max(lens)
    bl = [0] * (maxl + 2)
    for ln in lens:
        if ln:
            bl[ln] += 1
    code, nxt = 0, [0] * (maxl + 2)
    for b in range(1, maxl + 1):
        code = (code + bl[b - 1]) << 1
        nxt[b] = code
    out = [0] * len(lens)
    for s, ln in enumerate(lens):
        if ln:
            out[s] = nxt[ln]
            nxt[ln] += 1
    return out


def rle_lengths(seq):
    """RFC 1951 3.2.7: (symbol, extra, number of extra bits) of the code-length alphabet for a sequence of code lengths"""
    out, i, n = [], 0, len(seq)
    while i < n:
        v, r = seq[i], 1
        while i + r < n and seq[i + r] == v:
            r += 1
        left = r
        if v == 0:
            while left >= 11:
                t = min(left, 138)
                out.append((18, t - 11, 7))
                left -= t
            if left >= 3:
                out.append((17, left - 3, 3))
                left = 0
        else:
            out.append((v, 0, 0))
            left -= 1
            while left >= 3:
                t = min(left, 6)
                out.append((16, t - 3, 2))
                left -= t
        out += [(v, 0, 0)] * left
        i += r
    return out


def make_table(hist):
    """Sampled token counts -> (code lengths, canonical codes, the block header's bits)."""
    lens = huff_lengths([int(h) + 1 for h in hist], 15)
    codes = canonical(lens)
    r = rle_lengths(lens + [1])                     # literal / length code lengths, then the one distance code (one bit)
    clc = [0] * 19
    for s, _, _ in r:
        clc[s] += 1
    used = [s for s in range(19) if clc[s]]
    cll = [0] * 19
    for s, ln in zip(used, huff_lengths([clc[s] for s in used], 7)):
        cll[s] = ln
    clcode = canonical(cll)
    ncl = 19
    while ncl > 4 and cll[_CL_ORDER[ncl - 1]] == 0:
        ncl -= 1
    b = _Bits()
    b.put(0, 1)                                     # BFINAL
    b.put(2, 2)                                     # BTYPE = 10
    b.put(NSYM - 257, 5)
    b.put(0, 5)
    b.put(ncl - 4, 4)
    for i in range(ncl):
        b.put(cll[_CL_ORDER[i]], 3)
    for s, e, ne in r:
        b.huff(clcode[s], cll[s])
        if ne:
            b.put(e, ne)
    return lens, codes, b


def shuffled(chunk):
    """HDF5's shuffle filter on an int16 chunk: all first (low) bytes, then all second (high) bytes, element order as stored."""
    b = np.ascontiguousarray(chunk, "<i2").reshape(-1).view(np.uint8).reshape(-1, 2)
    return np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])


def _chunks(daily, cy, cx):
    return [np.ascontiguousarray(daily[:, r0:r0 + cy, c0:c0 + cx]) for r0 in range(0, daily.shape[1], cy)
            for c0 in range(0, daily.shape[2], cx)]


def tile_hist(daily, cy, cx):
    """Token counts of every SAMPLE-th segment of every chunk of the tile (what k_deflate_hist counts)."""
    hist = [0] * NSYM
    for chunk in _chunks(daily, cy, cx):
        _, hi = shuffled(chunk)
        for s0 in range(0, hi.size, SEG * SAMPLE):
            for sym, _, _ in tokens(hi, s0, min(hi.size, s0 + SEG)):
                hist[sym] += 1
    return hist


def tile_table(daily, cy, cx):
    return make_table(tile_hist(daily, cy, cx))


def _huffman_block(hi, start, stop, table):
    lens, codes, hdr = table
    b = _Bits(hdr)
    for sym, e, ne in tokens(hi, start, stop):
        b.huff(codes[sym], lens[sym])
        if ne:
            b.put(e, ne)
        if sym > 256:
            b.put(0, 1)                            # the distance code: one code, one bit
    b.huff(codes[256], lens[256])                  # end of block
    b.put(0, 3)                                    # BFINAL = 0, BTYPE = 00: an empty stored block aligns the stream
    b.align()
    return bytes(b.out) + b"\x00\x00\xff\xff"


def deflate_chunk(chunk, table):
    """int16 [ndays, cy, cx] + the tile's table -> the zlib stream twx_deflate.h emits for the chunk."""
    lo, hi = shuffled(chunk)
    n = lo.size
    out = bytearray(b"\x78\x01")
    for b0 in range(0, n, STORED):
        blk = lo[b0:b0 + STORED].tobytes()
        out += bytes([0, len(blk) & 255, len(blk) >> 8, ~len(blk) & 255, (~len(blk) >> 8) & 255]) + blk
    for s0 in range(0, n, SEG):
        s1 = min(n, s0 + SEG)
        blk = _huffman_block(hi, s0, s1, table)
        if len(blk) > s1 - s0 + 5:                 # no runs worth coding: the segment as a stored block
            m = s1 - s0
            blk = bytes([0, m & 255, m >> 8, ~m & 255, (~m >> 8) & 255]) + hi[s0:s1].tobytes()
        out += blk
    out += b"\x01\x00\x00\xff\xff"
    out += zlib.adler32(lo.tobytes() + hi.tobytes()).to_bytes(4, "big")
    return bytes(out)


def deflate_tile(daily, cy, cx):
    """int16 [ndays, Y, X] -> list of streams, chunks in row-major order (the order H5Dwrite_chunk is called in)."""
    table = tile_table(daily, cy, cx)
    return [deflate_chunk(c, table) for c in _chunks(daily, cy, cx)]


def inflate_chunk(stream, nd, cy, cx):
    """The int16 chunk a reader gets: inflate, unshuffle."""
    raw = np.frombuffer(zlib.decompress(stream), np.uint8)
    n = nd * cy * cx
    assert raw.size == 2 * n
    return np.stack([raw[:n], raw[n:]], axis=1).reshape(-1).view("<i2").reshape(nd, cy, cx)
