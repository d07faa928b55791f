// twx_deflate.h -- the daily int16 outputs of a streamed tile as the chunk bytes of an HDF5 dataset with the shuffle + deflate
// filters (what netCDF4-python writes for ``zlib=True``: the reference's mosaics, tiling.py:720,894,913,1035), formed ON THE
// GPU: the tile leaves the device at about half its size and a writer appends the bytes with H5Dwrite_chunk -- no CPU deflate
// (one core deflates ~60 MB/s at level 1; a GPU produces 50 GB/s of int16).
//
// Per (variable, chunk of cy x cx cells x all days) one zlib stream (RFC 1950 / 1951) of the SHUFFLED chunk -- N low bytes, then
// N high bytes, N = ndays cy cx, element order (day, row, column) --:
//     78 01                                          zlib header (deflate, 32 K window, no dictionary, check bits)
//     low plane, ceil(N / 65535) STORED blocks       00 LEN ~LEN data: the low byte of a temperature in 1/100 degC is noise
//     high plane, ceil(N / 16384) FIXED-HUFFMAN      the high byte changes every 2.56 degC: long runs along a row of cells.
//       blocks of 16 384 input bytes each (or stored,  Tokens: literal, or match(length 3..64, distance 1) = "repeat the previous
//       whichever is shorter)                        byte"; each block ends byte-aligned (end-of-block, then an empty stored block
//                                                    00 00 FF FF), so blocks are encoded independently and concatenated by bytes
//     01 00 00 FF FF                                 final (empty stored) block
//     Adler-32 of the 2 N shuffled bytes, big endian
// Any inflate reads this (tests: zlib.decompress, libhdf5's filter pipeline, h5py).  Bit-exact restatement for the tests:
// oracle/deflate_oracle.py.
//
// Kernels (one launch each per variable; grid = chunks x segments, chunks fastest: the chunks side by side in a tile row read
// the same 128-byte lines of the [day][Y][X] image at the same segment number, so they should run together):
//   k_deflate_count   stages a segment (16 384 elements of the chunk, gathered from the [day][Y][X] image), writes its low bytes
//                  into their stored blocks (positions are known a priori), counts the bits of its high-plane block, and leaves
//                  the segment's Adler partial sums (sum d, sum (len - j) d of both planes)
//   k_deflate_scan    per chunk: byte offsets of the high-plane blocks (exclusive scan), header, final block, Adler-32, total size
//   k_deflate_emit    stages the segment again, assembles its bit stream in LDS (every thread a 64-byte piece at its scanned bit
//                  offset), copies it to its place
// Roofline: HBM.  Algorithmic bytes per cell-day and variable: 2 read twice + ~1.05 written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef TWX_DF_PIECE
#define TWX_DF_PIECE 64                                    // input bytes per thread (128: half the work-groups per CU -- LDS --, emit 1.6 x slower)
#endif
#define TWX_DF_THREADS 256
#define TWX_DF_SEG (TWX_DF_PIECE * TWX_DF_THREADS)         // input bytes per fixed-Huffman block
#define TWX_DF_SEG_OUT (TWX_DF_SEG * 9 / 8 + 8)            // its longest encoding: 9 bits per literal, 3 + 7 + 3 bits, padding, 00 00 FF FF
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

// match(length L in 3..258, distance 1) in the fixed Huffman code, as the bits go into the stream (LSB first); returns their number
__host__ __device__ __forceinline__ unsigned df_match(int L, unsigned &val)
{
    int sym, e = 0, extra = 0;
    if (L == 258) sym = 285;
    else {
        const int l = L - 3;
        if (l < 8) sym = 257 + l;
        else {
            e = 29 - __builtin_clz((unsigned)l);             // floor(log2 l) - 2 extra bits
            sym = 257 + 4 * (e + 1) + ((l >> e) & 3);
            extra = l & ((1 << e) - 1);
        }
    }
    unsigned n1, code;
    if (sym < 280) { n1 = 7; code = (unsigned)(sym - 256); } else { n1 = 8; code = 0xC0u + (unsigned)(sym - 280); }
    val = df_rev(code, n1) | ((unsigned)extra << n1);        // (+ 5 zero bits: distance code 0 = distance 1)
    return n1 + (unsigned)e + 5u;
}

__host__ __device__ __forceinline__ unsigned df_literal(int c, unsigned &val)
{
    if (c < 144) { val = df_rev(0x30u + (unsigned)c, 8); return 8u; }
    val = df_rev(0x190u + (unsigned)(c - 144), 9);
    return 9u;
}

// The tokens of one piece: b[0 .. len) with the byte before it (prev; 256 = none).  A byte equal to its predecessor extends a
// run; a run of 3 and more (cut at the end of the piece) is one match, shorter ones are literals.  put(bits, n) receives the
// token bits in stream order, n <= 27 per call: the tokens a byte closes -- the run before it, then its own literal -- go in
// ONE call from ONE place (a wave executes every divergent call site once per step, and the emitting put is the costly part).
// ALIGNED: b is 4-byte aligned and readable up to the next multiple of 4 (the kernels' LDS pieces).
// (A variant that builds 128-bit "equals its predecessor" / ">= 144" masks by word arithmetic and then steps from run to run
// with count-trailing-zeros gave the same streams 8 % SLOWER, same-box A/B, and was dropped: this loop is not what costs.)
static_assert(TWX_DF_PIECE <= 258 && TWX_DF_PIECE % 4 == 0, "a run inside a piece must fit one match");
__host__ __device__ __forceinline__ unsigned df_run(int run, int prev, unsigned &val)
{
    if (run >= 3) return df_match(run, val);
    val = 0;
    if (run == 0) return 0u;
    unsigned lv;
    const unsigned ln = df_literal(prev, lv);
    val = run == 2 ? (lv | (lv << ln)) : lv;
    return run == 2 ? 2u * ln : ln;
}

template <bool ALIGNED, class Put>
__host__ __device__ __forceinline__ unsigned df_piece(const uint8_t *b, int len, int prev, Put put)
{
    unsigned bits = 0, v, n;
    int run = 0;
    for (int i = 0; i < len; i += 4) {
        uint32_t w = 0;
        if (ALIGNED) w = *reinterpret_cast<const uint32_t *>(b + i);
        else for (int k = 0; k < 4 && i + k < len; ++k) w |= (uint32_t)b[i + k] << (8 * k);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k >= len) break;
            const int c = (int)((w >> (8 * k)) & 255u);
            if (c == prev) { ++run; continue; }
            n = df_run(run, prev, v);                        // <= 18 bits
            unsigned lv;
            const unsigned ln = df_literal(c, lv);
            v |= lv << n;
            n += ln;
            put(v, n);
            bits += n;
            run = 0;
            prev = c;
        }
    }
    n = df_run(run, prev, v);
    if (n) { put(v, n); bits += n; }
    return bits;
}

// bytes of a fixed-Huffman block with `bits` token bits: header (3), tokens, end-of-block (7), empty stored block (3, pad, 4 bytes)
__host__ __device__ __forceinline__ uint32_t df_huff_bytes(uint32_t bits) { return (3u + bits + 7u + 3u + 7u) / 8u + 4u; }
// a segment whose Huffman block would be longer than its bytes stored (no runs: 8-9 bits per literal) is a stored block instead
__host__ __device__ __forceinline__ bool df_seg_stored(uint32_t bits, int len) { return df_huff_bytes(bits) > (uint32_t)len + 5u; }
__host__ __device__ __forceinline__ uint32_t df_seg_bytes(uint32_t bits, int len)
{
    return df_seg_stored(bits, len) ? (uint32_t)len + 5u : df_huff_bytes(bits);
}

struct DfArgs {
    const uint16_t *daily;      // [ndays][Y][X] packed values of one variable (device image of the tile)
    uint8_t *out;               // [nchunk][slot_bytes] chunk streams
    uint32_t *seg_bytes;        // [nchunk][nseg] bytes of every high-plane block
    uint32_t *seg_off;          // [nchunk][nseg] its offset inside the high-plane region
    uint32_t *adl;              // [nchunk][2 planes][nseg][2]: sum d, sum (len - j) d  (mod 65521)
    uint16_t *piece_bits;       // [nchunk][nseg][256] token bits of every thread's piece (k_deflate_count -> k_deflate_emit)
    int64_t *chunk_bytes;       // [nchunk] bytes of every chunk's stream
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

template <int VEC>
__global__ __launch_bounds__(TWX_DF_THREADS) void k_deflate_count(DfArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_hi[TWX_DF_THREADS * TWX_DF_PSTRIDE];
    __shared__ int s_prev;
    __shared__ uint32_t s_w[4];
    __shared__ unsigned long long s_sum[4];
    const int ch = blockIdx.x, seg = blockIdx.y, t = threadIdx.x;     // chunks fastest: see the header
    if (t < 4) s_sum[t] = 0ull;
    uint64_t sums[4];
    const int len = dfl::stage<true, VEC>(a, ch, seg, s_hi, &s_prev, sums);
    __syncthreads();
    const int plen = max(0, min(TWX_DF_PIECE, len - t * TWX_DF_PIECE));
    const int prev = t == 0 ? s_prev : (int)s_hi[(t - 1) * TWX_DF_PSTRIDE + TWX_DF_PIECE - 1];
    const uint32_t bits = df_piece<true>(&s_hi[t * TWX_DF_PSTRIDE], plen, prev, [](unsigned, unsigned) {});
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
        a.seg_bytes[(int64_t)ch * a.nseg + seg] = df_seg_bytes(tot, len);
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
    const int ch = blockIdx.x, seg = blockIdx.y, t = threadIdx.x;     // chunks fastest: see the header
    uint64_t sums[4];
    const uint32_t bits = a.piece_bits[((int64_t)ch * a.nseg + seg) * TWX_DF_THREADS + t];      // (counted by k_deflate_count)
    const int len = dfl::stage<false, VEC>(a, ch, seg, s_hi, &s_prev, sums);
    for (int i = t; i < (int)(sizeof(s_out) / 4); i += TWX_DF_THREADS) s_out[i] = 0u;
    __syncthreads();
    const int plen = max(0, min(TWX_DF_PIECE, len - t * TWX_DF_PIECE));
    const int prev = t == 0 ? s_prev : (int)s_hi[(t - 1) * TWX_DF_PSTRIDE + TWX_DF_PIECE - 1];
    const uint8_t *mine = &s_hi[t * TWX_DF_PSTRIDE];
    uint32_t tot;
    const uint32_t inc = dfl::block_scan(bits, s_w, &tot);
    uint8_t *dst = a.out + (int64_t)ch * a.slot_bytes + 2 + a.lo_bytes + a.seg_off[(int64_t)ch * a.nseg + seg];
    if (df_seg_stored(tot, len)) {                           // (work-group uniform) no runs worth coding: 00 LEN ~LEN and the bytes
        const unsigned bl = (unsigned)len;
        if (t == 0) { dst[0] = 0; dst[1] = (uint8_t)bl; dst[2] = (uint8_t)(bl >> 8); dst[3] = (uint8_t)~bl; dst[4] = (uint8_t)(~bl >> 8); }
        for (int i = t; i < len; i += TWX_DF_THREADS) dst[5 + i] = s_hi[(i / TWX_DF_PIECE) * TWX_DF_PSTRIDE + (i % TWX_DF_PIECE)];
        return;
    }
    // this thread's bits start at bit 3 + (bits of the threads before it): gathered in a 64-bit register, whole words stored as
    // they fill -- only the first and the last word of a piece are shared with a neighbour (atomic or).  (One atomic per token:
    // 11 of the 19.4 ms of a configs[3] tile, measured with the emission compiled out.)
    const uint32_t pos = 3u + inc - bits;
    if (t == 0) atomicOr(&s_out[0], 2u);                      // BFINAL = 0, BTYPE = 01 (fixed Huffman): bits 0 1 0
    uint64_t acc = 0;
    uint32_t nacc = pos & 31u, w = pos >> 5;
    bool first = true;
    df_piece<true>(mine, plen, prev, [&](unsigned v, unsigned n) {
        acc |= (uint64_t)v << nacc;
        nacc += n;
        if (nacc >= 32u) {
            if (first) atomicOr(&s_out[w], (uint32_t)acc); else s_out[w] = (uint32_t)acc;
            first = false;
            ++w;
            acc >>= 32;
            nacc -= 32u;
        }
    });
    if (nacc) atomicOr(&s_out[w], (uint32_t)acc);
    __syncthreads();
    const uint32_t nbytes = df_huff_bytes(tot);              // end-of-block and the empty stored block are zeros but for FF FF
    uint8_t *sb = reinterpret_cast<uint8_t *>(s_out);
    if (t == 0) { sb[nbytes - 2] = 0xFF; sb[nbytes - 1] = 0xFF; }
    __syncthreads();
    for (uint32_t i = t; i < nbytes; i += TWX_DF_THREADS) dst[i] = sb[i];
}
#endif  // __HIPCC__
