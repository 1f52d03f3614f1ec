// grm_deflate.hip -- zlib (RFC 1950 / 1951) encoder on the device for the two large datasets dsk2kover appends to the Kover
// HDF5 file (bin/kover/core/kover/dataset/tools/kmer_pack.py:28-36; schema dataset/create.py:214-238): kmer_matrix chunks
// (1, chunk_cols) of uint64 and kmer_sequences chunks of S<k> strings.  The host hands the finished streams to H5Dwrite_chunk.
//
// ONE WAVE PER CHUNK, three phases inside one launch:
//   1  tokens: the chunk is walked 64 elements at a time; a lane decides its element's token (grm_deflate_fns.h: runs and far
//      matches of whole words for matrix rows -- the far matches through a ring of the last 4096 words and a 4096-slot
//      "latest position" table in LDS, looked up BEFORE the step's own words are entered, entered with ds_max so that the
//      result does not depend on lane timing --, common prefix with the previous string for k-mers), adds its symbols to the
//      LDS histograms, its bytes to the Adler-32 sums;
//   2  codes: the used symbols are ranked by the whole wave, lane 0 builds the two length-limited Huffman codes
//      (<= 286 + 30 symbols: microseconds), the wave adds up the stream's length; a chunk the code would expand is STORED;
//   3  bits: header, then the elements again 64 at a time: a lane's bit count, a wave prefix sum, the lane ORs its bits into
//      an LDS stage at its offset, whole words leave for HBM.
// A wave is its own workgroup, so the __syncthreads() between LDS phases cost nothing and no other wave is waited for.
// Integer / byte work: no MFMA; the launch is latency-bound per wave and throughput comes from the chunk count
// (1600 chunks for 1000 genomes x 10 M columns: 1 GB in a few ms against 0.8 s of zlib on the host's cores).
#include <hip/hip_runtime.h>

#include "grm_deflate_fns.h"
#include "grm_internal.h"

namespace grm {
using namespace dfl;

namespace {

__device__ inline uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ inline uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ inline uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// A lane's writer into the wave's LDS stage: bits are ORed in at the lane's own offset (the ranges of different lanes are
// disjoint, the stage is zero where nothing was written yet)
struct LaneWriter {
    uint32_t *stage;
    uint32_t pos;          // bit position of the next flush
    uint64_t acc = 0;
    uint32_t accn = 0;
    __device__ LaneWriter(uint32_t *s, uint32_t p) : stage(s), pos(p) {}
    __device__ void or_bits(uint32_t v, uint32_t at)
    {
        if (!v) return;
        const uint32_t sh = at & 31, idx = at >> 5;
        atomicOr(&stage[idx], v << sh);
        if (sh && (v >> (32 - sh))) atomicOr(&stage[idx + 1], v >> (32 - sh));
    }
    __device__ void put(uint32_t value, uint32_t nbits)
    {
        acc |= (uint64_t)value << accn;
        accn += nbits;
        if (accn >= 32) {
            or_bits((uint32_t)acc, pos);
            pos += 32;
            acc >>= 32;
            accn -= 32;
        }
    }
    __device__ void finish()
    {
        if (accn) or_bits((uint32_t)acc & (uint32_t)((1ull << accn) - 1), pos);
        pos += accn;
        accn = 0;
        acc = 0;
    }
};

// the wave's output: `cur` bits (< 32) wait in stage[0]; whole words go to out32[wpos ..]
struct WaveSink {
    uint32_t *stage;
    uint32_t *out32;
    uint32_t cur = 0;
    uint32_t wpos = 0;
    __device__ uint64_t bits() const { return (uint64_t)wpos * 32 + cur; }
    // after every lane has written its n bits at cur + (exclusive prefix): total = sum over the lanes
    __device__ void flush(uint32_t total, int lane)
    {
        __syncthreads();
        const uint32_t end = cur + total, nw = end >> 5;
        for (uint32_t t = lane; t < nw; t += 64) out32[wpos + t] = stage[t];
        const uint32_t rem = stage[nw];
        __syncthreads();
        for (uint32_t t = lane; t <= nw + 1; t += 64) stage[t] = 0;
        __syncthreads();
        if (lane == 0) stage[0] = rem;
        __syncthreads();
        cur = end & 31;
        wpos += nw;
    }
};

// rank the used symbols of freq[0..n) ascending by (freq, symbol) with the whole wave; returns their number
__device__ inline int rank_symbols(const uint32_t *freq, int n, uint16_t *order, int lane)
{
    int used = 0;
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        const uint32_t f = s < n ? freq[s] : 0;
        if (f) {
            int r = 0;
            for (int t = 0; t < n; t++) {
                const uint32_t g = freq[t];
                r += (g && (g < f || (g == f && t < s))) ? 1 : 0;
            }
            order[r] = (uint16_t)s;
        }
        used += __popcll(__ballot(f != 0));
    }
    return used;
}

struct Codes {
    uint32_t *freq_ll, *freq_d;      // LL_PAD / D_PAD
    uint32_t *code_ll, *code_d;      // packed code << 4 | length
    uint8_t *len_ll, *len_d;
    uint16_t *order;                 // LL_PAD
    uint32_t *node_freq;             // 2 * LL_PAD
    uint16_t *parent;                // 2 * LL_PAD
    uint8_t *depth;                  // 2 * LL_PAD
};

// phase 2: both codes from the histograms; returns the bits the symbols take (without extra bits and header)
__device__ inline uint64_t build_codes(const Codes &c, int lane)
{
    __syncthreads();
    const int m_ll = rank_symbols(c.freq_ll, LL_SYMS, c.order, lane);
    __syncthreads();
    if (lane == 0) {
        huff_lengths(c.freq_ll, LL_SYMS, c.order, m_ll, MAX_BITS, c.len_ll, c.node_freq, c.parent, c.depth);
        huff_codes(c.len_ll, LL_SYMS, c.code_ll);
    }
    __syncthreads();
    const int m_d = rank_symbols(c.freq_d, D_SYMS, c.order, lane);
    __syncthreads();
    if (lane == 0) {
        huff_lengths(c.freq_d, D_SYMS, c.order, m_d, MAX_BITS, c.len_d, c.node_freq, c.parent, c.depth);
        huff_codes(c.len_d, D_SYMS, c.code_d);
    }
    __syncthreads();
    uint64_t bits = 0;
    for (int s = lane; s < LL_SYMS; s += 64) bits += (uint64_t)c.freq_ll[s] * c.len_ll[s];
    if (lane < D_SYMS) bits += (uint64_t)c.freq_d[lane] * c.len_d[lane];
    return wave_sum_u64(bits);
}

// zlib header + dynamic block header
__device__ inline void emit_header(WaveSink &sink, const Codes &c, int lane)
{
    {
        uint32_t n = 0;
        if (lane == 0) {
            const Bits b = header_piece0();
            LaneWriter w(sink.stage, sink.cur);
            w.put((uint32_t)b.lo, 32);
            w.put((uint32_t)(b.lo >> 32), 32);
            w.put((uint32_t)b.hi, b.n - 64);
            w.finish();
            n = b.n;
        }
        sink.flush(__shfl(n, 0), lane);
    }
    // the 286 + 30 code lengths, 4 bits each: five per lane
    uint32_t n = 0;
    {
        LaneWriter w(sink.stage, sink.cur + 20 * lane);
        for (int q = 0; q < 5; q++) {
            const int s = 5 * lane + q;
            if (s < LL_SYMS + D_SYMS) {
                w.put(header_len_code(s < LL_SYMS ? c.len_ll[s] : c.len_d[s - LL_SYMS]), 4);
                n += 4;
            }
        }
        w.finish();
    }
    sink.flush(wave_sum_u32(n), lane);
}

// end of block, zero bits up to the next byte, Adler-32 most significant byte first; the stream's length in bytes
__device__ inline uint32_t emit_trailer(WaveSink &sink, const Codes &c, uint32_t adler, int lane, uint8_t *out8)
{
    uint32_t n = 0;
    if (lane == 0) {
        LaneWriter w(sink.stage, sink.cur);
        put_code(w, c.code_ll[256]);
        const uint32_t eob = c.code_ll[256] & 15;
        const uint32_t pad = (8 - (uint32_t)((sink.bits() + eob) & 7)) & 7;
        w.put(0, pad);
        w.put(__builtin_bswap32(adler), 32);
        w.finish();
        n = eob + pad + 32;
    }
    sink.flush(__shfl(n, 0), lane);
    // what is left in stage[0] is a whole number of bytes
    const uint32_t rest = sink.cur >> 3;
    if (lane < (int)rest) out8[(size_t)sink.wpos * 4 + lane] = (uint8_t)(sink.stage[0] >> (8 * lane));
    return sink.wpos * 4 + rest;
}

__device__ inline uint32_t adler_finish(uint64_t a_sum, uint64_t b_sum, uint64_t n_bytes)
{
    const uint32_t a = (uint32_t)((1 + a_sum) % ADLER_MOD);
    const uint32_t b = (uint32_t)((n_bytes % ADLER_MOD + b_sum) % ADLER_MOD);
    return b << 16 | a;
}

// stored form of a chunk: blocks of <= 65535 bytes.  byte_at(g) = byte g of the raw chunk
template <class ByteAt>
__device__ inline uint32_t emit_stored(uint8_t *out8, uint64_t n_bytes, uint32_t adler, int lane, ByteAt byte_at)
{
    const uint64_t n_blocks = n_bytes ? (n_bytes + 65534) / 65535 : 1;
    if (lane == 0) { out8[0] = 0x78; out8[1] = 0x01; }
    for (uint64_t b = 0; b < n_blocks; b++) {
        const uint64_t g0 = b * 65535;
        const uint32_t len = (uint32_t)(n_bytes - g0 < 65535 ? n_bytes - g0 : 65535);
        uint8_t *o = out8 + 2 + b * (65535 + 5);
        if (lane == 0) {
            o[0] = b + 1 == n_blocks ? 1 : 0;
            o[1] = (uint8_t)len; o[2] = (uint8_t)(len >> 8);
            o[3] = (uint8_t)~len; o[4] = (uint8_t)(~len >> 8);
        }
        for (uint32_t q = lane; q < len; q += 64) o[5 + q] = byte_at(g0 + q);
    }
    const uint64_t end = 2 + n_blocks * 5 + n_bytes;
    if (lane < 4) out8[end + lane] = (uint8_t)(adler >> (8 * (3 - lane)));
    return (uint32_t)(end + 4);
}

constexpr int ROW_HIST_COPIES = 4;
constexpr int ROW_STAGE_WORDS = 272;        // 64 lanes x 120 bits + 31 waiting = 241 words, + the word a shifted piece spills into
constexpr size_t ROW_LDS_BYTES = (size_t)RING_WORDS * 8 + TABLE_SLOTS * 4 + ROW_HIST_COPIES * LL_PAD * 4 + D_PAD * 4 + LL_PAD * 4 + D_PAD * 4 +
                                 LL_PAD + D_PAD + LL_PAD * 2 + ROW_STAGE_WORDS * 4;

// ---- kmer_matrix ----
// chunk c of the launch = chunk (first_chunk + c) of the matrix, row-major: word-row r = chunk / chunks_per_row, columns
// [j * cw, j * cw + cw) with j = chunk % chunks_per_row; columns beyond n_cols are zero words (HDF5 stores whole chunks).
// tok: cw uint16 per chunk of the launch; out: cap bytes per chunk (deflate_rows_cap); sizes[c] = the stream's length.
__global__ void __launch_bounds__(64) deflate_rows_kernel(const uint64_t *matrix, uint64_t n_cols, uint32_t cw, uint32_t chunks_per_row,
                                                          uint64_t first_chunk, uint16_t *tok_all, uint8_t *out_all, uint64_t cap,
                                                          uint32_t *sizes)
{
    extern __shared__ __align__(16) unsigned char lds[];
    uint64_t *ring = reinterpret_cast<uint64_t *>(lds);
    uint32_t *table = reinterpret_cast<uint32_t *>(ring + RING_WORDS);
    uint32_t *hist_ll = table + TABLE_SLOTS;                         // [copies][LL_PAD]; copy 0 becomes the summed frequencies
    uint32_t *hist_d = hist_ll + ROW_HIST_COPIES * LL_PAD;
    uint32_t *code_ll = hist_d + D_PAD;
    uint32_t *code_d = code_ll + LL_PAD;
    uint8_t *len_ll = reinterpret_cast<uint8_t *>(code_d + D_PAD);
    uint8_t *len_d = len_ll + LL_PAD;
    uint16_t *order = reinterpret_cast<uint16_t *>(len_d + D_PAD);
    uint32_t *stage = reinterpret_cast<uint32_t *>(order + LL_PAD);
    const int lane = threadIdx.x;
    const uint64_t chunk = first_chunk + blockIdx.x;
    const uint64_t r = chunk / chunks_per_row, c0 = (chunk % chunks_per_row) * cw;
    const uint64_t *src = matrix + r * n_cols + c0;
    const uint32_t n_valid = (uint32_t)(n_cols - c0 < cw ? n_cols - c0 : cw);
    uint16_t *tok = tok_all + (size_t)blockIdx.x * cw;
    uint8_t *out8 = out_all + (size_t)blockIdx.x * cap;
    const uint64_t n_bytes = (uint64_t)cw * 8;

    for (int t = lane; t < TABLE_SLOTS; t += 64) table[t] = 0;
    for (int t = lane; t < ROW_HIST_COPIES * LL_PAD + D_PAD; t += 64) hist_ll[t] = 0;
    for (int t = lane; t < ROW_STAGE_WORDS; t += 64) stage[t] = 0;
    __syncthreads();

    // ---- phase 1 ----
    uint32_t *my_hist = hist_ll + (lane & (ROW_HIST_COPIES - 1)) * LL_PAD;
    uint64_t prev_last = 0, a_sum = 0, b_sum = 0;
    uint32_t extra = 0;
    for (uint32_t base = 0; base < cw; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < cw;
        const uint64_t w = (valid && i < n_valid) ? src[i] : 0;
        if (valid) ring[i & (RING_WORDS - 1)] = w;
        uint64_t up = __shfl_up(w, 1);
        if (lane == 0) up = prev_last;
        prev_last = __shfl(w, 63);
        const bool rep = valid && i > 0 && w == up;
        const uint64_t rmask = __ballot(rep);
        uint32_t t16 = TOK_LITERAL;
        if (rep) {
            const uint32_t gm = (uint32_t)(rmask >> (lane & 32));
            const int p = lane & 31;
            if (p == 0 || !((gm >> (p - 1)) & 1)) {
                const uint32_t rest = ~(gm >> p);
                const uint32_t n = rest ? (uint32_t)__ffs((int)rest) - 1 : 32u;
                uint32_t eb, ev;
                atomicAdd(&my_hist[len_symbol(8 * n, &eb, &ev)], 1u);
                atomicAdd(&hist_d[5], 1u);                       // distance 8 = symbol 5 + one extra bit
                extra += eb + 1;
                t16 = TOK_RUN_HEAD | n;
            } else {
                t16 = TOK_RUN_MORE;
            }
        } else if (valid) {
            const uint32_t slot = word_slot(w);
            const uint32_t seen = table[slot];
            uint32_t d = 0;
            if (seen) {
                d = i - (seen - 1);
                if (d > (uint32_t)WINDOW_WORDS || ring[(seen - 1) & (RING_WORDS - 1)] != w) d = 0;
            }
            atomicMax(&table[slot], i + 1);
            if (d) {
                uint32_t eb, ev;
                atomicAdd(&my_hist[262], 1u);                    // length 8
                atomicAdd(&hist_d[dist_symbol(8 * d, &eb, &ev)], 1u);
                extra += eb;
                t16 = d;
            } else {
#pragma unroll
                for (int t = 0; t < 8; t++) atomicAdd(&my_hist[(uint32_t)(w >> (8 * t)) & 0xffu], 1u);
            }
        }
        if (valid) {
            tok[i] = (uint16_t)t16;
            uint32_t s = 0, tw = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t b = (uint32_t)(w >> (8 * t)) & 0xffu;
                s += b;
                tw += (uint32_t)t * b;
            }
            const uint64_t left = (n_bytes - (uint64_t)i * 8) % ADLER_MOD + ADLER_MOD;      // (n - p) of the word's first byte, kept positive
            a_sum += s;
            b_sum += left * s - tw;
        }
    }
    a_sum = wave_sum_u64(a_sum);
    b_sum = wave_sum_u64(b_sum % ADLER_MOD);
    const uint32_t adler = adler_finish(a_sum, b_sum, n_bytes);
    const uint64_t extra_bits = wave_sum_u32(extra);
    __syncthreads();
    for (int s = lane; s < LL_PAD; s += 64) {
        uint32_t f = 0;
        for (int q = 0; q < ROW_HIST_COPIES; q++) f += hist_ll[q * LL_PAD + s];
        hist_ll[s] = s == 256 ? 1u : (s < LL_SYMS ? f : 0u);
    }

    // ---- phase 2 (tree scratch lies over the ring, which phase 3 does not need) ----
    Codes c;
    c.freq_ll = hist_ll; c.freq_d = hist_d; c.code_ll = code_ll; c.code_d = code_d; c.len_ll = len_ll; c.len_d = len_d; c.order = order;
    c.node_freq = reinterpret_cast<uint32_t *>(ring);
    c.parent = reinterpret_cast<uint16_t *>(c.node_freq + 2 * LL_PAD);
    c.depth = reinterpret_cast<uint8_t *>(c.parent + 2 * LL_PAD);
    const uint64_t code_bits = build_codes(c, lane);
    const uint64_t dyn_bytes = (HEADER_BITS + code_bits + extra_bits + 7) / 8 + 4;
    const uint64_t stored_bytes = 2 + 5 * ((n_bytes + 65534) / 65535) + n_bytes + 4;
    if (dyn_bytes >= stored_bytes) {
        const uint32_t n = emit_stored(out8, n_bytes, adler, lane, [&](uint64_t g) -> uint8_t {
            const uint64_t wi = g >> 3;
            const uint64_t w = wi < n_valid ? src[wi] : 0;
            return (uint8_t)(w >> (8 * (g & 7)));
        });
        if (lane == 0) sizes[blockIdx.x] = n;
        return;
    }

    // ---- phase 3 ----
    WaveSink sink;
    sink.stage = stage;
    sink.out32 = reinterpret_cast<uint32_t *>(out8);
    emit_header(sink, c, lane);
    for (uint32_t base = 0; base < cw; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < cw;
        const uint32_t t16 = valid ? tok[i] : TOK_RUN_MORE;
        const uint64_t w = (valid && t16 == TOK_LITERAL && i < n_valid) ? src[i] : 0;
        BitCounter bc;
        emit_row_token(bc, t16, w, code_ll, code_d);
        const uint32_t incl = wave_incl_scan_u32(bc.n, lane);
        LaneWriter lw(stage, sink.cur + incl - bc.n);
        emit_row_token(lw, t16, w, code_ll, code_d);
        lw.finish();
        sink.flush(__shfl(incl, 63), lane);
    }
    const uint32_t n = emit_trailer(sink, c, adler, lane, out8);
    if (lane == 0) sizes[blockIdx.x] = n;
}

// ---- kmer_sequences ----
constexpr int STR_STAGE_WORDS = 4096;       // 64 elements x (48 + 128 x 15) bits = 3936 words
constexpr size_t STR_LDS_BYTES = (size_t)STR_STAGE_WORDS * 4 + LL_PAD * 4 + D_PAD * 4 + LL_PAD * 4 + D_PAD * 4 + LL_PAD + D_PAD + LL_PAD * 2 +
                                 2 * LL_PAD * 4 + 2 * LL_PAD * 2 + 2 * LL_PAD;

// chunk c = strings [c * ce, c * ce + ce) of the dictionary (n k-mers of `words` uint64, most significant first, ascending);
// strings beyond n are k zero bytes
__global__ void __launch_bounds__(64) deflate_kmer_strings_kernel(const uint64_t *kmers, uint64_t n, int words, int k, uint32_t ce,
                                                                  uint64_t first_chunk, uint8_t *out_all, uint64_t cap, uint32_t *sizes)
{
    extern __shared__ __align__(16) unsigned char lds[];
    uint32_t *stage = reinterpret_cast<uint32_t *>(lds);
    uint32_t *freq_ll = stage + STR_STAGE_WORDS;
    uint32_t *freq_d = freq_ll + LL_PAD;
    uint32_t *code_ll = freq_d + D_PAD;
    uint32_t *code_d = code_ll + LL_PAD;
    uint8_t *len_ll = reinterpret_cast<uint8_t *>(code_d + D_PAD);
    uint8_t *len_d = len_ll + LL_PAD;
    uint16_t *order = reinterpret_cast<uint16_t *>(len_d + D_PAD);
    Codes c;
    c.freq_ll = freq_ll; c.freq_d = freq_d; c.code_ll = code_ll; c.code_d = code_d; c.len_ll = len_ll; c.len_d = len_d; c.order = order;
    c.node_freq = reinterpret_cast<uint32_t *>(order + LL_PAD);
    c.parent = reinterpret_cast<uint16_t *>(c.node_freq + 2 * LL_PAD);
    c.depth = reinterpret_cast<uint8_t *>(c.parent + 2 * LL_PAD);
    const int lane = threadIdx.x;
    const uint64_t chunk = first_chunk + blockIdx.x;
    const uint64_t e0 = chunk * ce;
    const uint64_t n_real = n - e0 < ce ? n - e0 : ce;
    uint8_t *out8 = out_all + (size_t)blockIdx.x * cap;
    const uint64_t n_bytes = (uint64_t)ce * k;

    for (int t = lane; t < STR_STAGE_WORDS; t += 64) stage[t] = 0;
    for (int t = lane; t < LL_PAD + D_PAD; t += 64) freq_ll[t] = 0;
    __syncthreads();

    auto load = [&](uint64_t e, uint64_t *a) {
#pragma unroll
        for (int w = 0; w < 4; w++) a[w] = (w < words && e < n_real) ? kmers[(e0 + e) * words + w] : 0;
    };

    // ---- phase 1 ----
    uint32_t cnt[4] = {0, 0, 0, 0}, cnt_zero = 0, n_match = 0, extra = 0;
    uint64_t a_sum = 0, b_sum = 0;
    uint32_t deb, dev;
    const uint32_t dsym = dist_symbol((uint32_t)k, &deb, &dev);
    for (uint32_t base = 0; base < ce; base += 64) {
        const uint64_t e = base + lane;
        if (e >= ce) continue;
        uint64_t a[4], p[4];
        load(e, a);
        load(e ? e - 1 : 0, p);
        const bool pad = e >= n_real;
        const int lcp = kmer_element_lcp(a, p, words, k, e, n_real);
        if (lcp) {
            uint32_t eb, ev;
            atomicAdd(&freq_ll[len_symbol((uint32_t)lcp, &eb, &ev)], 1u);
            n_match++;
            extra += eb + deb;
        }
        if (pad) {
            cnt_zero += (uint32_t)(k - lcp);
        } else {
            uint32_t s = 0;
            uint64_t tw = 0;
            for (int j = 0; j < k; j++) {
                const int bit = 2 * (k - 1 - j);
                const uint32_t code = (uint32_t)(a[words - 1 - bit / 64] >> (bit & 63)) & 3u;
                const uint32_t letter = (0x47544341u >> (8 * code)) & 0xffu;
                s += letter;
                tw += (uint64_t)j * letter;
                if (j >= lcp) {
                    cnt[0] += code == 0;
                    cnt[1] += code == 1;
                    cnt[2] += code == 2;
                    cnt[3] += code == 3;
                }
            }
            const uint64_t left = (n_bytes - e * (uint64_t)k) % ADLER_MOD + ADLER_MOD;
            a_sum += s;
            b_sum += left * s - tw % ADLER_MOD;          // left * s >= 65521 * 65 > the reduced tw; <= 1.5e9 per element
        }
    }
    a_sum = wave_sum_u64(a_sum);
    b_sum = wave_sum_u64(b_sum % ADLER_MOD);
    const uint32_t adler = adler_finish(a_sum, b_sum, n_bytes);
    const uint64_t extra_bits = wave_sum_u32(extra);
    n_match = wave_sum_u32(n_match);
    cnt_zero = wave_sum_u32(cnt_zero);
#pragma unroll
    for (int q = 0; q < 4; q++) cnt[q] = wave_sum_u32(cnt[q]);
    __syncthreads();
    if (lane == 0) {
        freq_ll['A'] = cnt[0]; freq_ll['C'] = cnt[1]; freq_ll['T'] = cnt[2]; freq_ll['G'] = cnt[3];
        freq_ll[0] = cnt_zero;
        freq_ll[256] = 1;
        freq_d[dsym] = n_match;
    }

    // ---- phase 2 ----
    const uint64_t code_bits = build_codes(c, lane);
    const uint64_t dyn_bytes = (HEADER_BITS + code_bits + extra_bits + 7) / 8 + 4;
    const uint64_t stored_bytes = 2 + 5 * ((n_bytes + 65534) / 65535) + n_bytes + 4;
    if (dyn_bytes >= stored_bytes) {
        const uint32_t nn = emit_stored(out8, n_bytes, adler, lane, [&](uint64_t g) -> uint8_t {
            const uint64_t e = g / (uint64_t)k;
            if (e >= n_real) return 0;
            return (uint8_t)kmer_letter(kmers + (e0 + e) * words, words, k, (int)(g - e * (uint64_t)k));
        });
        if (lane == 0) sizes[blockIdx.x] = nn;
        return;
    }

    // ---- phase 3 ----
    WaveSink sink;
    sink.stage = stage;
    sink.out32 = reinterpret_cast<uint32_t *>(out8);
    emit_header(sink, c, lane);
    for (uint32_t base = 0; base < ce; base += 64) {
        const uint64_t e = base + lane;
        const bool valid = e < ce;
        uint64_t a[4] = {0, 0, 0, 0}, p[4] = {0, 0, 0, 0};
        int lcp = 0;
        if (valid) {
            load(e, a);
            load(e ? e - 1 : 0, p);
            lcp = kmer_element_lcp(a, p, words, k, e, n_real);
        }
        const bool pad = e >= n_real;
        BitCounter bc;
        if (valid) emit_kmer_element(bc, a, words, k, lcp, pad, code_ll, code_d);
        const uint32_t incl = wave_incl_scan_u32(bc.n, lane);
        LaneWriter lw(stage, sink.cur + incl - bc.n);
        if (valid) emit_kmer_element(lw, a, words, k, lcp, pad, code_ll, code_d);
        lw.finish();
        sink.flush(__shfl(incl, 63), lane);
    }
    const uint32_t nn = emit_trailer(sink, c, adler, lane, out8);
    if (lane == 0) sizes[blockIdx.x] = nn;
}

// streams of a launch, each at c * cap, moved back to back: dst[off[c] .. off[c] + sizes[c]); off: 16-byte aligned starts
__global__ void __launch_bounds__(256) deflate_compact_kernel(const uint8_t *src, uint64_t cap, const uint32_t *sizes, const uint64_t *off, uint8_t *dst)
{
    const uint32_t c = blockIdx.x;
    const uint32_t n = sizes[c];
    const uint4 *s = reinterpret_cast<const uint4 *>(src + (size_t)c * cap);
    uint4 *d = reinterpret_cast<uint4 *>(dst + off[c]);
    for (uint32_t t = threadIdx.x; t < (n + 15) / 16; t += 256) d[t] = s[t];
}

}  // namespace

uint64_t deflate_chunk_cap(uint64_t raw_bytes)
{
    // the stored form is the upper bound (a chunk the Huffman code would expand is stored); +64: whole-word stores, 16-byte copies
    const uint64_t cap = 2 + 5 * ((raw_bytes + 65534) / 65535 + 1) + raw_bytes + 4 + 64;
    return (cap + 15) & ~15ull;
}

hipError_t launch_deflate_rows(hipStream_t s, const uint64_t *matrix, uint64_t n_cols, uint32_t cw, uint32_t chunks_per_row, uint64_t first_chunk,
                               uint32_t n_chunks, uint16_t *tok, uint8_t *out, uint64_t cap, uint32_t *sizes)
{
    if (!n_chunks) return hipSuccess;
    static_assert(ROW_LDS_BYTES <= 64 * 1024, "deflate_rows_kernel: default dynamic LDS limit");
    hipLaunchKernelGGL(deflate_rows_kernel, dim3(n_chunks), dim3(64), ROW_LDS_BYTES, s, matrix, n_cols, cw, chunks_per_row, first_chunk, tok, out, cap,
                       sizes);
    return hipGetLastError();
}

hipError_t launch_deflate_kmer_strings(hipStream_t s, const uint64_t *kmers, uint64_t n, int words, int k, uint32_t ce, uint64_t first_chunk,
                                       uint32_t n_chunks, uint8_t *out, uint64_t cap, uint32_t *sizes)
{
    if (!n_chunks) return hipSuccess;
    static_assert(STR_LDS_BYTES <= 64 * 1024, "deflate_kmer_strings_kernel: default dynamic LDS limit");
    hipLaunchKernelGGL(deflate_kmer_strings_kernel, dim3(n_chunks), dim3(64), STR_LDS_BYTES, s, kmers, n, words, k, ce, first_chunk, out, cap, sizes);
    return hipGetLastError();
}

hipError_t launch_deflate_compact(hipStream_t s, const uint8_t *src, uint64_t cap, const uint32_t *sizes, uint32_t n_chunks, uint64_t *off, uint8_t *dst)
{
    if (!n_chunks) return hipSuccess;
    hipLaunchKernelGGL(deflate_compact_kernel, dim3(n_chunks), dim3(256), 0, s, src, cap, sizes, off, dst);
    return hipGetLastError();
}
}  // namespace grm
