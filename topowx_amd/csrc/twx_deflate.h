// twx_deflate.h -- the daily int16 outputs of a streamed tile as the chunk bytes of an HDF5 dataset with the shuffle + deflate
// filters (what netCDF4-python writes for ``zlib=True``: the reference's mosaics, tiling.py:720,894,913,1035), formed ON THE
// GPU: the tile leaves the device at ~0.55 of its size and a writer appends the bytes with H5Dwrite_chunk -- no CPU deflate
// (one core deflates ~90 MB/s at level 1; a GPU produces 50 GB/s of int16).
//
// Per (variable, chunk of cy x cx cells x all days) one zlib stream (RFC 1950 / 1951) of the SHUFFLED chunk -- N low bytes, then
// N high bytes, N = ndays cy cx, element order (day, row, column) --:
//     78 01                                          zlib header (deflate, 32 K window, no dictionary, check bits)
//     low plane, ceil(N / 65535) STORED blocks       00 LEN ~LEN data: the low byte of a temperature in 1/100 degC is noise
//     high plane, ceil(N / 16384) DYNAMIC-HUFFMAN    the high byte changes every 2.56 degC: long runs along a row of cells.
//       blocks of 16 384 input bytes each (or stored,  Tokens: literal, or match(length 3..64, distance 1) = "repeat the previous
//       whichever is shorter)                        byte", in ONE Huffman code per (variable, tile) built from the tokens of every
//                                                    16th block (literals cost ~3 bits instead of the fixed code's 8: -12 % of the
//                                                    stream; round 6 started with fixed-Huffman blocks); each block carries the
//                                                    code's description (~40 bytes) and ends byte-aligned (end-of-block, then an
//                                                    empty stored block 00 00 FF FF): blocks are encoded independently
//     01 00 00 FF FF                                 final (empty stored) block
//     Adler-32 of the 2 N shuffled bytes, big endian
// Any inflate reads this (tests: zlib.decompress, libhdf5's filter pipeline, h5py): zlib level 1's size on the same chunks.
// Bit-exact restatement for the tests: oracle/deflate_oracle.py.
//
// Kernels (per variable; grid = chunks x segments, chunks fastest: the chunks side by side in a tile row read the same 128-byte
// lines of the [day][Y][X] image at the same segment number, so they should run together):
//   k_deflate_hist   tokens of every 16th segment -> symbol counts (LDS, then global atomics)
//   k_deflate_table  one thread: Huffman code lengths (<= 15 bits; two-queue construction, ties by symbol; too deep -> counts
//                    halved), canonical codes, the block header's bit string (code lengths run-length coded, RFC 1951 3.2.7)
//   k_deflate_count  stages a segment (16 384 elements of the chunk, gathered from the [day][Y][X] image), writes its low bytes
//                    into their stored blocks (positions are known a priori), counts the bits of its high-plane block, and leaves
//                    the segment's Adler partial sums (sum d, sum (len - j) d of both planes)
//   k_deflate_scan   per chunk: byte offsets of the high-plane blocks (exclusive scan), header, final block, Adler-32, total size
//   k_deflate_emit   stages the segment again, assembles its bit stream in LDS (every thread a 64-byte piece at its scanned bit
//                    offset), copies it to its place
// Roofline: HBM.  Algorithmic bytes per cell-day and variable: 2 read twice (+ 1/16 for the counts) + ~1.1 written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef TWX_DF_PIECE
#define TWX_DF_PIECE 64                                    // input bytes per thread (128: half the work-groups per CU -- LDS --, emit 1.6 x slower)
#endif
#define TWX_DF_THREADS 256
#define TWX_DF_SEG (TWX_DF_PIECE * TWX_DF_THREADS)         // input bytes per Huffman block
#define TWX_DF_SEG_OUT (TWX_DF_SEG + 5)                    // its longest encoding: stored (a longer Huffman block is not emitted)
#define TWX_DF_STORED 65535
#define TWX_DF_PSTRIDE (TWX_DF_PIECE + 4)                   // LDS stride of a piece (an odd number of dwords: conflict-free piece walks, 4-byte aligned)
#define TWX_ADLER_P 65521u

__host__ __device__ __forceinline__ int64_t df_lo_bytes(int64_t N) { return N + 5 * ((N + TWX_DF_STORED - 1) / TWX_DF_STORED); }
__host__ __device__ __forceinline__ int df_nseg(int64_t N) { return (int)((N + TWX_DF_SEG - 1) / TWX_DF_SEG); }
// bytes of a chunk's slot on the device: the longest stream a chunk of N elements can give, rounded to 256
__host__ __device__ __forceinline__ int64_t df_slot_bytes(int64_t N)
{
    return (2 + df_lo_bytes(N) + (int64_t)df_nseg(N) * TWX_DF_SEG_OUT + 9 + 255) / 256 * 256;
}

__host__ __device__ __forceinline__ int64_t df_min64(int64_t a, int64_t b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ unsigned df_rev(unsigned code, unsigned n) { return __builtin_bitreverse32(code) >> (32u - n); }

#define TWX_DF_NSYM 277            // literals 0..255, end of block 256, match lengths 257..276 (3..66: a piece holds 64 bytes)
#define TWX_DF_HDR_WORDS 80         // the block header's bit string: <= 17 + 19 x 3 + 278 x 7 bits

// the length symbol of a match of L bytes (3..66), its extra bits (RFC 1951 3.2.5)
__host__ __device__ __forceinline__ int df_len_sym(int L, unsigned &extra, unsigned &ne)
{
    const int l = L - 3;
    extra = 0; ne = 0;
    if (l < 8) return 257 + l;
    ne = (unsigned)(29 - __builtin_clz((unsigned)l));        // floor(log2 l) - 2 extra bits
    extra = (unsigned)l & ((1u << ne) - 1u);
    return 257 + 4 * ((int)ne + 1) + ((l >> ne) & 3);
}

// The steps of one piece: b[0 .. len) with the byte before it (prev; 256 = none).  A byte equal to its predecessor extends a
// run; every other byte c -- and the end of the piece, c = -1 -- is one STEP: step(run, prev, c) = "the run of `run` more bytes
// equal to prev is closed, then comes c".  The tokens of a step (df_step_tokens): the run as one match (length run >= 3, distance
// 1) or as 1-2 literals of prev, then the literal c.  One call site: a wave executes every divergent call site once per step.
// ALIGNED: b is 4-byte aligned and readable up to the next multiple of 4 (the kernels' LDS pieces).
// (A variant that builds 128-bit "equals its predecessor" / ">= 144" masks by word arithmetic and then steps from run to run
// with count-trailing-zeros gave the same streams 8 % SLOWER, same-box A/B, and was dropped: this loop is not what costs.)
static_assert(TWX_DF_PIECE <= 66 && TWX_DF_PIECE % 4 == 0, "a run inside a piece must fit the longest length symbol");
template <bool ALIGNED, class Step>
__host__ __device__ __forceinline__ void df_piece(const uint8_t *b, int len, int prev, Step step)
{
    int run = 0;
    for (int i = 0; i <= len; i += 4) {
        uint32_t w = 0;
        if (i < len) {
            if (ALIGNED) w = *reinterpret_cast<const uint32_t *>(b + i);
            else for (int k = 0; k < 4 && i + k < len; ++k) w |= (uint32_t)b[i + k] << (8 * k);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k > len) break;
            const int c = i + k == len ? -1 : (int)((w >> (8 * k)) & 255u);     // one step past the last byte: the open run is closed
            if (c == prev) { ++run; continue; }
            if (run > 0 || c >= 0) step(run, prev, c);
            run = 0;
            prev = c;
        }
    }
}

// tok(symbol, extra bits, their number) for the tokens of a step, in stream order
template <class Tok>
__host__ __device__ __forceinline__ void df_step_tokens(int run, int prev, int c, Tok tok)
{
    if (run >= 3) { unsigned ex, ne; const int sym = df_len_sym(run, ex, ne); tok(sym, ex, ne); }
    else { if (run >= 1) tok(prev, 0u, 0u); if (run == 2) tok(prev, 0u, 0u); }
    if (c >= 0) tok(c, 0u, 0u);
}

// ---- the Huffman code of a (variable, tile) ---------------------------------------------------------------------------------
struct DfTable {
    uint16_t code[TWX_DF_NSYM + 3];     // canonical code of every symbol, bit-reversed: as it goes into the stream (LSB first)
    uint8_t len[TWX_DF_NSYM + 3];       // its length, 1..15
    uint32_t hdr_bits;                  // bits of the block header (BFINAL, BTYPE = 10, HLIT, HDIST, HCLEN, the code lengths)
    uint32_t hdr[TWX_DF_HDR_WORDS];     // the header's bit string, LSB first
};

// Code lengths (<= limit) of a Huffman code for count[0 .. n) (all > 0).  Leaves in (count, symbol) order, two queues, a leaf
// before an internal node of the same weight; a code deeper than `limit`: every count halved (rounding up), once more.
// Deterministic: oracle/deflate_oracle.py restates it step by step.  In three parts, because the kernel sorts with all its
// threads and keeps the arrays in LDS (one thread walking 277^2 / 2 insertion steps through scratch memory took 14 ms):
//   df_sort_order       order[] = the symbols by (count, symbol)                        (host: insertion sort)
//   df_huff_sorted      the tree over that order; false when it is deeper than `limit`  (wt / parent / depth: 2 n entries each)
//   df_huff_lengths     both in the halving loop, arrays on the stack                   (host; the 19-symbol alphabet everywhere)
__host__ __device__ inline void df_sort_order(const uint32_t *count, int n, uint16_t *order)
{
    for (int i = 0; i < n; ++i) {
        int j = i;
        while (j > 0 && count[order[j - 1]] > count[i]) { order[j] = order[j - 1]; --j; }
        order[j] = (uint16_t)i;
    }
}

__host__ __device__ inline bool df_huff_sorted(const uint32_t *count, const uint16_t *order, int n, int limit, uint32_t *wt, uint16_t *parent,
                                               uint16_t *depth, uint8_t *len)
{
    if (n == 1) { len[0] = 1; return true; }
    for (int i = 0; i < n; ++i) wt[i] = count[order[i]];
    int li = 0, qi = n, nn = n;
    while ((n - li) + (nn - qi) > 1) {
        int pick[2];
        for (int r = 0; r < 2; ++r) {
            if (li < n && (qi >= nn || wt[li] <= wt[qi])) pick[r] = li++;
            else pick[r] = qi++;
        }
        wt[nn] = wt[pick[0]] + wt[pick[1]];
        parent[pick[0]] = (uint16_t)nn; parent[pick[1]] = (uint16_t)nn;
        ++nn;
    }
    depth[nn - 1] = 0;
    int deepest = 0;
    for (int v = nn - 2; v >= 0; --v) { depth[v] = (uint16_t)(depth[parent[v]] + 1); if (v < n && depth[v] > deepest) deepest = depth[v]; }
    if (deepest > limit) return false;
    for (int i = 0; i < n; ++i) len[order[i]] = (uint8_t)depth[i];
    return true;
}

template <int MAXN>
__host__ __device__ inline void df_huff_lengths(const uint32_t *count_in, int n, int limit, uint8_t *len)
{
    uint32_t count[MAXN], wt[2 * MAXN];
    uint16_t order[MAXN], parent[2 * MAXN], depth[2 * MAXN];
    for (int i = 0; i < n; ++i) count[i] = count_in[i];
    for (;;) {
        df_sort_order(count, n, order);
        if (df_huff_sorted(count, order, n, limit, wt, parent, depth, len)) return;
        for (int i = 0; i < n; ++i) count[i] = (count[i] + 1u) >> 1;
    }
}

// canonical codes (RFC 1951 3.2.2) of lengths len[0 .. n) (0 = unused), bit-reversed
__host__ __device__ inline void df_canonical(const uint8_t *len, int n, uint16_t *code)
{
    uint32_t bl[17] = {0}, next[17] = {0};
    for (int i = 0; i < n; ++i) bl[len[i]]++;
    bl[0] = 0;
    uint32_t c = 0;
    for (int b = 1; b <= 15; ++b) { c = (c + bl[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; ++i) code[i] = len[i] ? (uint16_t)df_rev(next[len[i]]++, len[i]) : (uint16_t)0;
}

struct DfBits {                        // a bit string, LSB first
    uint32_t *w; uint32_t n;
    __host__ __device__ void put(uint32_t v, uint32_t nb)
    {
        if (!nb) return;
        const uint32_t i = n >> 5, sh = n & 31u;
        w[i] |= v << sh;
        if (sh + nb > 32u) w[i + 1] |= v >> (32u - sh);
        n += nb;
    }
};

struct DfRle { uint8_t rs[TWX_DF_NSYM + 1], re[TWX_DF_NSYM + 1]; };     // workspace of df_finish_table

// t->len[] is set: the canonical codes and the block header.  The code lengths of the literal / length code and of the distance
// code (one code, one bit) are run-length coded (3.2.7): symbols 0..15 a length, 16 repeat the previous 3..6 times (2 bits),
// 17 / 18 zeros 3..10 (3 bits) / 11..138 (7 bits); that 19-symbol alphabet has a Huffman code of its own (<= 7 bits).
__host__ __device__ inline void df_finish_table(DfTable *t, DfRle *r)
{
    df_canonical(t->len, TWX_DF_NSYM, t->code);
    uint8_t *rs = r->rs, *re = r->re;
    int nr = 0;
    auto seq = [&](int i) -> int { return i < TWX_DF_NSYM ? (int)t->len[i] : 1; };
    for (int i = 0; i <= TWX_DF_NSYM;) {
        const int v = seq(i);
        int run = 1;
        while (i + run <= TWX_DF_NSYM && seq(i + run) == v) ++run;
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int k = left < 138 ? left : 138; rs[nr] = 18; re[nr++] = (uint8_t)(k - 11); left -= k; }
            if (left >= 3) { rs[nr] = 17; re[nr++] = (uint8_t)(left - 3); left = 0; }
        } else {
            rs[nr] = (uint8_t)v; re[nr++] = 0; --left;
            while (left >= 3) { const int k = left < 6 ? left : 6; rs[nr] = 16; re[nr++] = (uint8_t)(k - 3); left -= k; }
        }
        for (; left > 0; --left) { rs[nr] = (uint8_t)v; re[nr++] = 0; }
        i += run;
    }
    uint32_t clc[19] = {0}, used_cnt[19];
    uint8_t cll[19] = {0}, used[19], ul[19];
    uint16_t clcode[19];
    for (int i = 0; i < nr; ++i) clc[rs[i]]++;
    int nu = 0;
    for (int q = 0; q < 19; ++q) if (clc[q]) { used[nu] = (uint8_t)q; used_cnt[nu++] = clc[q]; }
    df_huff_lengths<19>(used_cnt, nu, 7, ul);
    for (int q = 0; q < nu; ++q) cll[used[q]] = ul[q];
    df_canonical(cll, 19, clcode);
    const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int ncl = 19;
    while (ncl > 4 && cll[ord[ncl - 1]] == 0) --ncl;
    for (int i = 0; i < TWX_DF_HDR_WORDS; ++i) t->hdr[i] = 0u;
    DfBits b{t->hdr, 0u};
    b.put(0u, 1); b.put(2u, 2);                              // BFINAL = 0, BTYPE = 10
    b.put(TWX_DF_NSYM - 257, 5); b.put(0u, 5); b.put((uint32_t)(ncl - 4), 4);
    for (int i = 0; i < ncl; ++i) b.put(cll[ord[i]], 3);
    for (int i = 0; i < nr; ++i) {
        b.put(clcode[rs[i]], cll[rs[i]]);
        if (rs[i] == 16) b.put(re[i], 2); else if (rs[i] == 17) b.put(re[i], 3); else if (rs[i] == 18) b.put(re[i], 7);
    }
    t->hdr_bits = b.n;
}

// hist[TWX_DF_NSYM] (sampled token counts; every symbol gets + 1: what was not seen must still have a code) -> the table.
// The host's form (tests/tools/deflate_codes.hip); k_deflate_table is the same with a parallel sort and LDS arrays.
__host__ inline void df_build_table(const uint32_t *hist, DfTable *t)
{
    uint32_t cnt[TWX_DF_NSYM];
    for (int i = 0; i < TWX_DF_NSYM; ++i) cnt[i] = hist[i] + 1u;
    df_huff_lengths<TWX_DF_NSYM>(cnt, TWX_DF_NSYM, 15, t->len);
    DfRle r;
    df_finish_table(t, &r);
}

// bytes of a dynamic block with `bits` token bits: header, tokens, end-of-block, empty stored block (3 bits, padding, 4 bytes)
__host__ __device__ __forceinline__ uint32_t df_huff_bytes(uint32_t bits, uint32_t hdr_bits, uint32_t eob_len)
{
    return (hdr_bits + bits + eob_len + 3u + 7u) / 8u + 4u;
}
// a segment whose Huffman block would be longer than its bytes stored is a stored block instead
__host__ __device__ __forceinline__ bool df_seg_stored(uint32_t huff_bytes, int len) { return huff_bytes > (uint32_t)len + 5u; }

struct DfArgs {
    const uint16_t *daily;      // [ndays][Y][X] packed values of one variable (device image of the tile)
    uint8_t *out;               // [nchunk][slot_bytes] chunk streams
    uint32_t *seg_bytes;        // [nchunk][nseg] bytes of every high-plane block
    uint32_t *seg_off;          // [nchunk][nseg] its offset inside the high-plane region
    uint32_t *adl;              // [nchunk][2 planes][nseg][2]: sum d, sum (len - j) d  (mod 65521)
    uint16_t *piece_bits;       // [nchunk][nseg][256] token bits of every thread's piece (k_deflate_count -> k_deflate_emit)
    int64_t *chunk_bytes;       // [nchunk] bytes of every chunk's stream
    uint32_t *hist;             // [TWX_DF_NSYM] sampled token counts (zeroed before k_deflate_hist)
    DfTable *table;             // the (variable, tile)'s Huffman code (k_deflate_table)
    int64_t N, slot_bytes, lo_bytes;
    int32_t Y, X, cy, cx, ncx, nseg;
};

#if defined(__HIPCC__)
namespace dfl {

__device__ __forceinline__ uint64_t wave_sum(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive scan over the work-group's 256 threads (s_w: 4 words of LDS); returns the inclusive value, total in *tot
__device__ __forceinline__ uint32_t block_scan(uint32_t v, uint32_t *s_w, uint32_t *tot)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();                                         // (s_w may still be read from a previous call)
    if (lane == 63) s_w[wv] = x;
    __syncthreads();
    uint32_t base = 0, t = 0;
#pragma unroll
    for (int w = 0; w < TWX_DF_THREADS / 64; ++w) { if (w < wv) base += s_w[w]; t += s_w[w]; }
    *tot = t;
    return x + base;
}

// Stage segment `seg` of chunk `ch`: high bytes into s_hi (piece p at p * TWX_DF_PSTRIDE), the byte before the segment into *s_prev
// (256: none).  LOW: also write the low bytes (and stored-block headers) to the stream and accumulate the Adler partial sums.
// VEC = 2 (cx and X even): a thread loads two neighbouring values at once.  Eight loads are in flight per thread before the first
// is used (the loop is bound by the latency of these gathers: rows of cx values, 2 cx bytes each, X values apart).
template <bool LOW, int VEC>
__device__ __forceinline__ int stage(const DfArgs &a, int ch, int seg, uint8_t *s_hi, int *s_prev, uint64_t (&sums)[4])
{
    constexpr int STEP = TWX_DF_THREADS * VEC, U = 8;
    const int t = threadIdx.x;
    const int64_t q0 = (int64_t)seg * TWX_DF_SEG;
    const int len = (int)df_min64(TWX_DF_SEG, a.N - q0);
    const int y0 = (ch / a.ncx) * a.cy, x0 = (ch % a.ncx) * a.cx, cc = a.cy * a.cx;
    // element q of the chunk = (day, row, column) = (q / cc, (q / cx) % cy, q % cx); this thread walks q0 + VEC t + STEP j
    const int64_t q = q0 + VEC * t;
    int64_t day = q / cc;
    const int r = (int)(q - day * cc);
    int y = r / a.cx, x = r - y * a.cx;
    const int dx = STEP % a.cx, dyq = STEP / a.cx, dy = dyq % a.cy, dd = dyq / a.cy;
    int64_t blk = q / TWX_DF_STORED;                         // stored block of low byte q
    int rb = (int)(q - blk * TWX_DF_STORED);
    uint8_t *o = a.out + (int64_t)ch * a.slot_bytes + 2;
    if (t == 0) {
        int pv = 256;
        if (q0 > 0) {
            const int64_t qp = q0 - 1, dp = qp / cc;
            const int rp = (int)(qp - dp * cc);
            pv = a.daily[(dp * a.Y + y0 + rp / a.cx) * a.X + x0 + rp % a.cx] >> 8;
        }
        *s_prev = pv;
    }
    uint64_t s_lo = 0, w_lo = 0, s_hi_sum = 0, w_hi = 0;
    auto low = [&](int64_t b_, int r_, unsigned lo, int ql) __attribute__((always_inline)) {
        uint8_t *p = o + b_ * (TWX_DF_STORED + 5);
        if (r_ == 0) {                                       // first byte of a stored block: its header
            const unsigned bl = (unsigned)df_min64(TWX_DF_STORED, a.N - (q0 + ql));
            p[0] = 0; p[1] = (uint8_t)bl; p[2] = (uint8_t)(bl >> 8); p[3] = (uint8_t)~bl; p[4] = (uint8_t)(~bl >> 8);
        }
        p[5 + r_] = (uint8_t)lo;
    };
    for (int j0 = 0; j0 * STEP < len; j0 += U) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ql = (j0 + u) * STEP + VEC * t;
            v[u] = 0u;
            if (ql < len) {
                const uint16_t *src = a.daily + ((day * a.Y + y0 + y) * a.X + x0 + x);
                v[u] = VEC == 2 ? *reinterpret_cast<const uint32_t *>(src) : (uint32_t)*src;
            }
            x += dx; if (x >= a.cx) { x -= a.cx; ++y; }
            y += dy; if (y >= a.cy) { y -= a.cy; ++day; }
            day += dd;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ql = (j0 + u) * STEP + VEC * t;
            if (ql < len) {
                const unsigned lo0 = v[u] & 255u, hi0 = (v[u] >> 8) & 255u, lo1 = (v[u] >> 16) & 255u, hi1 = v[u] >> 24;
                uint8_t *h = &s_hi[(ql / TWX_DF_PIECE) * TWX_DF_PSTRIDE + (ql % TWX_DF_PIECE)];
                if (VEC == 2) *reinterpret_cast<uint16_t *>(h) = (uint16_t)(hi0 | (hi1 << 8));
                else *h = (uint8_t)hi0;
                if (LOW) {
                    low(blk, rb, lo0, ql);
                    const uint64_t wgt = (uint64_t)(len - ql);
                    s_lo += lo0; w_lo += wgt * lo0; s_hi_sum += hi0; w_hi += wgt * hi0;
                    if (VEC == 2) {
                        const bool wrap = rb + 1 == TWX_DF_STORED;
                        low(wrap ? blk + 1 : blk, wrap ? 0 : rb + 1, lo1, ql + 1);
                        s_lo += lo1; w_lo += (wgt - 1) * lo1; s_hi_sum += hi1; w_hi += (wgt - 1) * hi1;
                    }
                }
            }
            if (LOW) { rb += STEP; if (rb >= TWX_DF_STORED) { rb -= TWX_DF_STORED; ++blk; } }
        }
    }
    sums[0] = s_lo; sums[1] = w_lo; sums[2] = s_hi_sum; sums[3] = w_hi;
    return len;
}

}  // namespace dfl

#define TWX_DF_SAMPLE 16            // k_deflate_hist reads every 16th segment of a chunk

// symbol counts of the sampled segments
template <int VEC>
__global__ __launch_bounds__(TWX_DF_THREADS) void k_deflate_hist(DfArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_hi[TWX_DF_THREADS * TWX_DF_PSTRIDE];
    __shared__ int s_prev;
    __shared__ uint32_t s_h[TWX_DF_NSYM];
    const int ch = blockIdx.x, seg = blockIdx.y * TWX_DF_SAMPLE, t = threadIdx.x;
    for (int i = t; i < TWX_DF_NSYM; i += TWX_DF_THREADS) s_h[i] = 0u;
    uint64_t sums[4];
    const int len = dfl::stage<false, VEC>(a, ch, seg, s_hi, &s_prev, sums);
    __syncthreads();
    const int plen = max(0, min(TWX_DF_PIECE, len - t * TWX_DF_PIECE));
    const int prev = t == 0 ? s_prev : (int)s_hi[(t - 1) * TWX_DF_PSTRIDE + TWX_DF_PIECE - 1];
    df_piece<true>(&s_hi[t * TWX_DF_PSTRIDE], plen, prev, [&](int run, int pv, int c) {
        df_step_tokens(run, pv, c, [&](int sym, unsigned, unsigned) { atomicAdd(&s_h[sym], 1u); });
    });
    __syncthreads();
    for (int i = t; i < TWX_DF_NSYM; i += TWX_DF_THREADS)
        if (s_h[i]) atomicAdd(&a.hist[i], s_h[i]);
}

// the (variable, tile)'s Huffman code from the sampled counts: df_build_table with the sort done by all threads (every symbol's
// rank in (count, symbol) order: 277 comparisons per symbol) and the arrays in LDS; thread 0 walks the tree and writes the header
__global__ __launch_bounds__(TWX_DF_THREADS) void k_deflate_table(const uint32_t *hist0, DfTable *table0, const uint32_t *hist1, DfTable *table1)
{
    const uint32_t *hist = blockIdx.x ? hist1 : hist0;      // one work-group per variable: the two codes are built side by side
    DfTable *table = blockIdx.x ? table1 : table0;
    __shared__ uint32_t s_cnt[TWX_DF_NSYM], s_wt[2 * TWX_DF_NSYM];
    __shared__ uint16_t s_order[TWX_DF_NSYM], s_parent[2 * TWX_DF_NSYM], s_depth[2 * TWX_DF_NSYM];
    __shared__ DfTable s_t;
    __shared__ DfRle s_r;
    __shared__ int s_ok;
    const int t = threadIdx.x;
    for (int i = t; i < TWX_DF_NSYM; i += TWX_DF_THREADS) s_cnt[i] = hist[i] + 1u;
    __syncthreads();
    for (;;) {
        for (int i = t; i < TWX_DF_NSYM; i += TWX_DF_THREADS) {
            const uint32_t c = s_cnt[i];
            int r = 0;
            for (int j = 0; j < TWX_DF_NSYM; ++j) { const uint32_t cj = s_cnt[j]; r += (cj < c || (cj == c && j < i)) ? 1 : 0; }
            s_order[r] = (uint16_t)i;
        }
        __syncthreads();
        if (t == 0) s_ok = df_huff_sorted(s_cnt, s_order, TWX_DF_NSYM, 15, s_wt, s_parent, s_depth, s_t.len) ? 1 : 0;
        __syncthreads();
        if (s_ok) break;
        for (int i = t; i < TWX_DF_NSYM; i += TWX_DF_THREADS) s_cnt[i] = (s_cnt[i] + 1u) >> 1;
        __syncthreads();
    }
    if (t == 0) df_finish_table(&s_t, &s_r);
    __syncthreads();
    uint32_t *dst = reinterpret_cast<uint32_t *>(table);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&s_t);
    for (int i = t; i < (int)(sizeof(DfTable) / 4); i += TWX_DF_THREADS) dst[i] = src[i];
}

template <int VEC>
__global__ __launch_bounds__(TWX_DF_THREADS) void k_deflate_count(DfArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_hi[TWX_DF_THREADS * TWX_DF_PSTRIDE];
    __shared__ int s_prev;
    __shared__ uint32_t s_w[4];
    __shared__ unsigned long long s_sum[4];
    __shared__ uint8_t s_len[TWX_DF_NSYM + 3];
    const int ch = blockIdx.x, seg = blockIdx.y, t = threadIdx.x;     // chunks fastest: see the header
    if (t < 4) s_sum[t] = 0ull;
    for (int i = t; i < TWX_DF_NSYM; i += TWX_DF_THREADS) s_len[i] = a.table->len[i];
    uint64_t sums[4];
    const int len = dfl::stage<true, VEC>(a, ch, seg, s_hi, &s_prev, sums);
    __syncthreads();
    const int plen = max(0, min(TWX_DF_PIECE, len - t * TWX_DF_PIECE));
    const int prev = t == 0 ? s_prev : (int)s_hi[(t - 1) * TWX_DF_PSTRIDE + TWX_DF_PIECE - 1];
    uint32_t bits = 0;
    df_piece<true>(&s_hi[t * TWX_DF_PSTRIDE], plen, prev, [&](int run, int pv, int c) {
        if (run >= 3) { unsigned ex, ne; const int sym = df_len_sym(run, ex, ne); bits += (uint32_t)s_len[sym] + ne + 1u; }     // (+ the one-bit distance code)
        else bits += (uint32_t)run * s_len[pv < 256 ? pv : 0];
        if (c >= 0) bits += s_len[c];
    });
    a.piece_bits[((int64_t)ch * a.nseg + seg) * TWX_DF_THREADS + t] = (uint16_t)bits;
    uint32_t tot;
    dfl::block_scan(bits, s_w, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t v = dfl::wave_sum(sums[k]);
        if ((t & 63) == 0) atomicAdd(&s_sum[k], (unsigned long long)v);
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t hb = df_huff_bytes(tot, a.table->hdr_bits, s_len[256]);
        a.seg_bytes[(int64_t)ch * a.nseg + seg] = df_seg_stored(hb, len) ? (uint32_t)len + 5u : hb;
        uint32_t *ad = a.adl + ((int64_t)ch * 2 * a.nseg + seg) * 2;
        ad[0] = (uint32_t)(s_sum[0] % TWX_ADLER_P); ad[1] = (uint32_t)(s_sum[1] % TWX_ADLER_P);
        ad += (int64_t)a.nseg * 2;
        ad[0] = (uint32_t)(s_sum[2] % TWX_ADLER_P); ad[1] = (uint32_t)(s_sum[3] % TWX_ADLER_P);
    }
}

__global__ __launch_bounds__(TWX_DF_THREADS) void k_deflate_scan(DfArgs a)
{
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_run;
    const int ch = blockIdx.x, t = threadIdx.x;
    if (t == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < a.nseg; base += TWX_DF_THREADS) {
        const int s = base + t;
        const uint32_t v = s < a.nseg ? a.seg_bytes[(int64_t)ch * a.nseg + s] : 0u;
        uint32_t tot;
        const uint32_t inc = dfl::block_scan(v, s_w, &tot);
        const uint32_t run = s_run;
        if (s < a.nseg) a.seg_off[(int64_t)ch * a.nseg + s] = run + inc - v;
        __syncthreads();
        if (t == 0) s_run = run + tot;
        __syncthreads();
    }
    // Adler-32 of the shuffled bytes.  Appending a piece of length L with S = sum d, W = sum (L - j) d_j to (A, B) gives
    // (A + S, B + L A + W): with P_k = 1 + sum of the S before piece k, A = P_end and B = sum_k (L_k P_k + W_k) -- a scan of S
    // and a sum, over the 2 nseg pieces in stream order (low plane, then high plane)
    __shared__ unsigned long long s_b;
    __shared__ uint32_t s_a;
    if (t == 0) { s_b = 0ull; s_a = 1u; }
    __syncthreads();
    for (int base = 0; base < 2 * a.nseg; base += TWX_DF_THREADS) {
        const int k = base + t;
        uint32_t S = 0, W = 0;
        uint64_t L = 0;
        if (k < 2 * a.nseg) {
            const int s = k < a.nseg ? k : k - a.nseg;
            const uint32_t *ad = a.adl + ((int64_t)ch * 2 * a.nseg + k) * 2;
            S = ad[0]; W = ad[1];
            L = (uint64_t)df_min64(TWX_DF_SEG, a.N - (int64_t)s * TWX_DF_SEG);
        }
        uint32_t tot;
        const uint32_t inc = dfl::block_scan(S, s_w, &tot);      // (S < 65521, 256 of them: no overflow)
        const uint32_t before = s_a;
        const uint64_t P = ((uint64_t)before + inc - S) % TWX_ADLER_P;
        const uint64_t term = dfl::wave_sum((L * P + W) % TWX_ADLER_P);
        __syncthreads();
        if ((t & 63) == 0) atomicAdd(&s_b, (unsigned long long)term);
        if (t == 0) s_a = (uint32_t)(((uint64_t)before + tot) % TWX_ADLER_P);
        __syncthreads();
    }
    if (t == 0) {
        uint8_t *o = a.out + (int64_t)ch * a.slot_bytes;
        o[0] = 0x78; o[1] = 0x01;
        uint8_t *tail = o + 2 + a.lo_bytes + s_run;
        tail[0] = 1; tail[1] = 0; tail[2] = 0; tail[3] = 0xFF; tail[4] = 0xFF;
        const uint32_t A = s_a, B = (uint32_t)(s_b % TWX_ADLER_P);
        tail[5] = (uint8_t)(B >> 8); tail[6] = (uint8_t)B; tail[7] = (uint8_t)(A >> 8); tail[8] = (uint8_t)A;
        a.chunk_bytes[ch] = 2 + a.lo_bytes + (int64_t)s_run + 9;
    }
}

template <int VEC>
__global__ __launch_bounds__(TWX_DF_THREADS) void k_deflate_emit(DfArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_hi[TWX_DF_THREADS * TWX_DF_PSTRIDE];
    __shared__ uint32_t s_out[(TWX_DF_SEG + 5 + 3) / 4 + 2];          // (a block longer than its bytes stored is not assembled)
    __shared__ int s_prev;
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_cl[TWX_DF_NSYM + 3];                        // code | length << 16 | (a match: the distance bit) << 24
    const int ch = blockIdx.x, seg = blockIdx.y, t = threadIdx.x;     // chunks fastest: see the header
    uint64_t sums[4];
    const uint32_t bits = a.piece_bits[((int64_t)ch * a.nseg + seg) * TWX_DF_THREADS + t];      // (counted by k_deflate_count)
    for (int i = t; i < TWX_DF_NSYM; i += TWX_DF_THREADS)
        s_cl[i] = (uint32_t)a.table->code[i] | ((uint32_t)a.table->len[i] << 16) | (i > 256 ? 1u << 24 : 0u);
    const uint32_t hdr_bits = a.table->hdr_bits;
    const int len = dfl::stage<false, VEC>(a, ch, seg, s_hi, &s_prev, sums);
    for (int i = t; i < (int)(sizeof(s_out) / 4); i += TWX_DF_THREADS) s_out[i] = 0u;
    __syncthreads();
    const int plen = max(0, min(TWX_DF_PIECE, len - t * TWX_DF_PIECE));
    const int prev = t == 0 ? s_prev : (int)s_hi[(t - 1) * TWX_DF_PSTRIDE + TWX_DF_PIECE - 1];
    const uint8_t *mine = &s_hi[t * TWX_DF_PSTRIDE];
    uint32_t tot;
    const uint32_t inc = dfl::block_scan(bits, s_w, &tot);
    uint8_t *dst = a.out + (int64_t)ch * a.slot_bytes + 2 + a.lo_bytes + a.seg_off[(int64_t)ch * a.nseg + seg];
    const uint32_t eob = s_cl[256];
    const uint32_t nbytes = df_huff_bytes(tot, hdr_bits, (eob >> 16) & 255u);
    if (df_seg_stored(nbytes, len)) {                        // (work-group uniform) no runs worth coding: 00 LEN ~LEN and the bytes
        const unsigned bl = (unsigned)len;
        if (t == 0) { dst[0] = 0; dst[1] = (uint8_t)bl; dst[2] = (uint8_t)(bl >> 8); dst[3] = (uint8_t)~bl; dst[4] = (uint8_t)(~bl >> 8); }
        for (int i = t; i < len; i += TWX_DF_THREADS) dst[5 + i] = s_hi[(i / TWX_DF_PIECE) * TWX_DF_PSTRIDE + (i % TWX_DF_PIECE)];
        return;
    }
    // the block header (the same bit string in every block of the tile's variable), then every thread's tokens at bit
    // hdr_bits + (bits of the threads before it): gathered in a 64-bit register, whole words stored as they fill -- only the first
    // and the last word of a piece are shared with a neighbour (atomic or).  (One atomic per token: 11 of the 19.4 ms of a
    // configs[3] tile, measured with the emission compiled out.)
    for (uint32_t i = t; i < (hdr_bits + 31u) / 32u; i += TWX_DF_THREADS) atomicOr(&s_out[i], a.table->hdr[i]);
    const uint32_t pos = hdr_bits + inc - bits;
    uint64_t acc = 0;
    uint32_t nacc = pos & 31u, w = pos >> 5;
    bool first = true;
    auto flush = [&]() __attribute__((always_inline)) {
        if (nacc >= 32u) {
            if (first) atomicOr(&s_out[w], (uint32_t)acc); else s_out[w] = (uint32_t)acc;
            first = false;
            ++w;
            acc >>= 32;
            nacc -= 32u;
        }
    };
    df_piece<true>(mine, plen, prev, [&](int run, int pv, int c) {
        // the step's bits (<= 19 + 15, or 3 x 15): the closed run, then the literal -- one table word per distinct symbol
        uint64_t sv = 0;
        uint32_t sn = 0;
        if (run >= 3) {
            unsigned ex, ne;
            const uint32_t cl = s_cl[df_len_sym(run, ex, ne)], l = (cl >> 16) & 255u;
            sv = (cl & 0xFFFFu) | (ex << l);                 // code, extra bits, the distance code (one 0 bit)
            sn = l + ne + 1u;
        } else if (run >= 1) {
            const uint32_t cl = s_cl[pv], l = (cl >> 16) & 255u;
            sv = cl & 0xFFFFu;
            sn = l;
            if (run == 2) { sv |= (uint64_t)(cl & 0xFFFFu) << l; sn += l; }
        }
        if (c >= 0) {
            const uint32_t cl = s_cl[c];
            sv |= (uint64_t)(cl & 0xFFFFu) << sn;
            sn += (cl >> 16) & 255u;
        }
        acc |= (sv & 0xFFFFFFFFull) << nacc;                 // (nacc < 32)
        nacc += sn < 32u ? sn : 32u;
        flush();
        if (sn > 32u) { acc |= (sv >> 32) << nacc; nacc += sn - 32u; flush(); }
    });
    if (nacc) atomicOr(&s_out[w], (uint32_t)acc);
    if (t == 0) {                                            // end of block, at bit hdr_bits + tot
        const uint32_t p = hdr_bits + tot, sh = p & 31u, c = eob & 0xFFFFu;
        atomicOr(&s_out[p >> 5], c << sh);
        if (sh + ((eob >> 16) & 255u) > 32u) atomicOr(&s_out[(p >> 5) + 1], c >> (32u - sh));
    }
    __syncthreads();
    uint8_t *sb = reinterpret_cast<uint8_t *>(s_out);        // the empty stored block after it is zeros but for FF FF
    if (t == 0) { sb[nbytes - 2] = 0xFF; sb[nbytes - 1] = 0xFF; }
    __syncthreads();
    for (uint32_t i = t; i < nbytes; i += TWX_DF_THREADS) dst[i] = sb[i];
}
#endif  // __HIPCC__
