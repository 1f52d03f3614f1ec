// grm_deflate_fns.h -- the format side of the device-side zlib encoder (grm_deflate.hip): RFC 1950 / 1951 symbol arithmetic,
// length-limited Huffman codes, the pieces of a dynamic block header, and the token rules of the two encoders.  Everything
// here is __host__ __device__ and free of wave intrinsics, so that tests/host/deflate_emul.cpp runs the very same functions
// on the CPU (sequentially, in the kernels' lockstep order) and hands the streams to zlib's inflate.
//
// Why an encoder of our own: dsk2kover appends gzip-filtered HDF5 chunks (bin/kover/core/kover/dataset/tools/kmer_pack.py:28-36,
// schema dataset/create.py:214-238).  Deflating 1 GB of presence words on the host cores was 0.8 s of a 1.4 s end-to-end run
// while the device idled; any valid zlib stream satisfies HDF5's deflate filter, so the chunks are encoded where the matrix
// already is.  The token rules are chosen for THIS data, not for text:
//   kmer_matrix rows (uint64 words, one column each): a word equal to the word before it continues a RUN (match at distance 8,
//     up to 32 words = 256 bytes per token; core k-mers are runs of all-ones words); else a word equal to the LATEST earlier
//     equal word within 4032 words is a FAR match (length 8; the ~31 k-mers around one SNP and the k-mers of one accessory
//     gene carry one pattern and sort to unrelated places); else eight literal bytes.  Measured on the headline's rows:
//     0.3655 of the raw size against zlib level 4's 0.3661.
//   kmer_sequences (S<k> strings in ascending k-mer order): the common prefix with the previous string (>= 3 letters) is one
//     match at distance k, the rest are literals of a four-letter alphabet: 0.22 against zlib level 4's 0.26.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define GRM_DHD __host__ __device__
#else
#define GRM_DHD
#endif

namespace grm {
namespace dfl {

constexpr int LL_SYMS = 286;          // literal / length alphabet (0..255 literals, 256 end of block, 257..285 lengths)
constexpr int D_SYMS = 30;            // distance alphabet
constexpr int MAX_BITS = 15;
constexpr int LL_PAD = 288;           // array sizes (multiples of 32)
constexpr int D_PAD = 32;
constexpr int RING_WORDS = 4096;      // the words a far match may point back into
constexpr int LANES = 64;             // one wave encodes one chunk; tokens are decided 64 words at a time
constexpr int WINDOW_WORDS = RING_WORDS - LANES;   // 4032 words = 32256 bytes < the 32 KiB deflate window
constexpr int TABLE_SLOTS = 4096;     // "latest position of a word with this hash", position + 1 (0 = none)
constexpr int RUN_GROUP = 32;         // a run token never crosses a 32-word group: <= 256 bytes <= the 258-byte match limit
constexpr int HEADER_BITS = 16 + 3 + 5 + 5 + 4 + 19 * 3 + (LL_SYMS + D_SYMS) * 4;   // zlib header + dynamic block header as written here

// tokens of a matrix row chunk, one uint16 per word
constexpr uint16_t TOK_LITERAL = 0;           // eight literal bytes
constexpr uint16_t TOK_RUN_HEAD = 0x8000;     // | n (1..32): this word and the n - 1 after it repeat the word before (match of 8 n bytes at distance 8)
constexpr uint16_t TOK_RUN_MORE = 0xffff;     // inside a run: nothing to emit
//                  2 .. WINDOW_WORDS          : far match, 8 bytes at distance 8 * value

GRM_DHD inline int top_bit(uint32_t x)        // position of the highest set bit, x > 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 31 - __clz((int)x);
#else
    return 31 - __builtin_clz(x);
#endif
}

// length 3..258 -> symbol, number and value of its extra bits (RFC 1951 3.2.5)
GRM_DHD inline uint32_t len_symbol(uint32_t len, uint32_t *ebits, uint32_t *eval)
{
    if (len == 258) { *ebits = 0; *eval = 0; return 285; }
    const uint32_t x = len - 3;
    if (x < 8) { *ebits = 0; *eval = 0; return 257 + x; }
    const int eb = top_bit(x) - 2;
    *ebits = (uint32_t)eb;
    *eval = x & ((1u << eb) - 1);
    return 257 + 4 * (uint32_t)eb + 4 + ((x >> eb) & 3);
}
// distance 1..32768 -> symbol, extra bits
GRM_DHD inline uint32_t dist_symbol(uint32_t dist, uint32_t *ebits, uint32_t *eval)
{
    const uint32_t x = dist - 1;
    if (x < 4) { *ebits = 0; *eval = 0; return x; }
    const int hb = top_bit(x);
    *ebits = (uint32_t)(hb - 1);
    *eval = x & ((1u << (hb - 1)) - 1);
    return 2 * (uint32_t)hb + ((x >> (hb - 1)) & 1);
}

GRM_DHD inline uint32_t word_slot(uint64_t w)
{
    const uint32_t h = (uint32_t)w * 0x9E3779B1u ^ (uint32_t)(w >> 32) * 0x85EBCA77u;
    return (h ^ (h >> 15)) >> 4 & (TABLE_SLOTS - 1);
}

GRM_DHD inline uint32_t reverse_bits(uint32_t code, int len)      // Huffman codes travel most significant bit first
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// Code lengths (<= max_bits) of a Huffman code for freq[0..n).  order[0..m): the used symbols ascending by (freq, symbol) --
// sorted by the caller (the kernel ranks them with the whole wave).  Scratch: node_freq[2 m], parent[2 m] (uint16), depth[2 m] (uint8).
// Two-queue construction, then the depths are folded to max_bits and the Kraft sum is brought back to one by moving the
// cheapest leaves down (as miniz does), and lengths go to the symbols in order: rarest symbols, longest codes.
GRM_DHD inline void huff_lengths(const uint32_t *freq, int n, const uint16_t *order, int m, int max_bits, uint8_t *len,
                                 uint32_t *node_freq, uint16_t *parent, uint8_t *depth)
{
    for (int s = 0; s < n; s++) len[s] = 0;
    if (m == 0) return;
    if (m == 1) { len[order[0]] = 1; return; }
    for (int i = 0; i < m; i++) node_freq[i] = freq[order[i]];
    int li = 0, ii = m, made = m;                 // next unused leaf, next unused internal node, next node to make
    for (int t = 0; t < m - 1; t++) {
        int pick[2];
        for (int q = 0; q < 2; q++) {
            if (li < m && (ii >= made || node_freq[li] <= node_freq[ii])) pick[q] = li++;
            else pick[q] = ii++;
        }
        node_freq[made] = node_freq[pick[0]] + node_freq[pick[1]];
        parent[pick[0]] = (uint16_t)made;
        parent[pick[1]] = (uint16_t)made;
        made++;
    }
    const int root = 2 * m - 2;
    depth[root] = 0;
    uint32_t count[MAX_BITS + 1];
    for (int l = 0; l <= max_bits; l++) count[l] = 0;
    for (int v = root - 1; v >= 0; v--) {          // a parent is made after its children: larger index, already done
        const int d = depth[parent[v]] + 1;
        depth[v] = (uint8_t)(d > 255 ? 255 : d);
        if (v < m) count[d > max_bits ? max_bits : d]++;
    }
    uint32_t total = 0;
    for (int l = 1; l <= max_bits; l++) total += count[l] << (max_bits - l);
    while (total > (1u << max_bits)) {
        count[max_bits]--;
        for (int l = max_bits - 1; l >= 1; l--)
            if (count[l]) { count[l]--; count[l + 1] += 2; break; }
        total--;
    }
    int at = 0;
    for (int l = max_bits; l >= 1; l--)
        for (uint32_t c = 0; c < count[l]; c++) len[order[at++]] = (uint8_t)l;
}

// canonical codes of the lengths, already bit-reversed for an LSB-first bit stream: packed[s] = code << 4 | length
GRM_DHD inline void huff_codes(const uint8_t *len, int n, uint32_t *packed)
{
    uint32_t count[MAX_BITS + 2], next[MAX_BITS + 2];
    for (int l = 0; l <= MAX_BITS + 1; l++) count[l] = 0;
    for (int s = 0; s < n; s++) count[len[s]]++;
    count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= MAX_BITS; l++) {
        code = (code + count[l - 1]) << 1;
        next[l] = code;
    }
    for (int s = 0; s < n; s++) {
        const int l = len[s];
        packed[s] = l ? (reverse_bits(next[l]++, l) << 4 | (uint32_t)l) : 0u;
    }
}

// ---- a piece of the bit stream: up to 128 bits, least significant first ----
struct Bits {
    uint64_t lo, hi;
    uint32_t n;
};
GRM_DHD inline void bits_put(Bits &b, uint32_t value, uint32_t nbits)       // nbits <= 32, value < 2^nbits
{
    if (!nbits) return;
    if (b.n < 64) {
        b.lo |= (uint64_t)value << b.n;
        if (b.n + nbits > 64) b.hi |= (uint64_t)value >> (64 - b.n);
    } else {
        b.hi |= (uint64_t)value << (b.n - 64);
    }
    b.n += nbits;
}
GRM_DHD inline void bits_put_code(Bits &b, uint32_t packed) { bits_put(b, packed >> 4, packed & 15); }

// The dynamic block header as this encoder writes it: the code-length alphabet is NOT compressed -- symbols 0..15 each get
// a 4-bit code (a complete code: 16 x 2^-4), 16 / 17 / 18 are unused -- so every one of the 286 + 30 lengths is its own
// 4-bit code: 1338 bits, 0.04 % of an 800 KB chunk.  piece 0: zlib header (0x78 0x9c: deflate, 32 KiB window, check bits;
// "default compression" as the level hint), BFINAL = 1, BTYPE = 10, HLIT = 29, HDIST = 29, HCLEN = 15, the 19 code length code lengths
// in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15.
GRM_DHD inline Bits header_piece0()
{
    Bits b = {0, 0, 0};
    bits_put(b, 0x78, 8);
    bits_put(b, 0x9c, 8);
    bits_put(b, 1, 1);
    bits_put(b, 2, 2);
    bits_put(b, LL_SYMS - 257, 5);
    bits_put(b, D_SYMS - 1, 5);
    bits_put(b, 19 - 4, 4);
    for (int i = 0; i < 19; i++) bits_put(b, i < 3 ? 0 : 4, 3);
    return b;
}
// the 4-bit code of a code length (symbol l of the code-length alphabet = code l, sent most significant bit first)
GRM_DHD inline uint32_t header_len_code(uint32_t l) { return reverse_bits(l, 4); }

// ---- adler32 of a chunk whose bytes are visited out of order: B = n + sum over byte positions p of (n - p) d_p ----
constexpr uint32_t ADLER_MOD = 65521;

// ---- k-mer strings ----
// number of leading bases two k-mers share; a, b: `words` uint64 each, most significant first, the k-mer right-aligned
GRM_DHD inline int kmer_lcp(const uint64_t *a, const uint64_t *b, int words, int k)
{
    int lead = 0;
    for (int w = 0; w < words; w++) {
        const uint64_t x = a[w] ^ b[w];
        if (x) {
#if defined(__HIP_DEVICE_COMPILE__)
            lead += __clzll((long long)x);
#else
            lead += __builtin_clzll(x);
#endif
            const int pad = 64 * words - 2 * k;
            return (lead - pad) / 2;
        }
        lead += 64;
    }
    return k;
}
GRM_DHD inline uint32_t kmer_letter(const uint64_t *a, int words, int k, int j)      // letter j of the string (GATB code A C T G)
{
    const int bit = 2 * (k - 1 - j);
    const uint32_t c = (uint32_t)(a[words - 1 - bit / 64] >> (bit & 63)) & 3u;
    return (0x47544341u >> (8 * c)) & 0xffu;       // "ACTG"
}


// ---- token -> bits, shared by the kernels and the host emulation.  W: anything with put(uint32_t value, uint32_t nbits <= 25) ----
struct BitCounter {                       // a writer that only counts
    uint32_t n = 0;
    GRM_DHD void put(uint32_t, uint32_t nbits) { n += nbits; }
};
template <class W> GRM_DHD inline void put_code(W &w, uint32_t packed) { w.put(packed >> 4, packed & 15); }

template <class W> GRM_DHD inline void emit_match(W &w, uint32_t len, uint32_t dist, const uint32_t *code_ll, const uint32_t *code_d)
{
    uint32_t eb, ev;
    const uint32_t ls = len_symbol(len, &eb, &ev);
    put_code(w, code_ll[ls]);
    w.put(ev, eb);
    const uint32_t ds = dist_symbol(dist, &eb, &ev);
    put_code(w, code_d[ds]);
    w.put(ev, eb);
}
// one word of a matrix row chunk
template <class W> GRM_DHD inline void emit_row_token(W &w, uint32_t tok, uint64_t word, const uint32_t *code_ll, const uint32_t *code_d)
{
    if (tok == TOK_RUN_MORE) return;
    if (tok & TOK_RUN_HEAD) { emit_match(w, 8 * (tok & 0x7fffu), 8, code_ll, code_d); return; }
    if (tok) { emit_match(w, 8, 8 * tok, code_ll, code_d); return; }
    for (int t = 0; t < 8; t++) put_code(w, code_ll[(uint32_t)(word >> (8 * t)) & 0xffu]);
}
// one element of a k-mer string chunk: lcp letters shared with the element before it (0 or >= 3), then literals; a pad element
// (behind the last k-mer of the dataset: HDF5 stores whole chunks) is k zero bytes
template <class W> GRM_DHD inline void emit_kmer_element(W &w, const uint64_t *a, int words, int k, int lcp, bool pad, const uint32_t *code_ll,
                                                         const uint32_t *code_d)
{
    if (lcp) emit_match(w, (uint32_t)lcp, (uint32_t)k, code_ll, code_d);
    for (int j = lcp; j < k; j++) put_code(w, code_ll[pad ? 0u : kmer_letter(a, words, k, j)]);
}
// the match a string element starts with: 0 (none) or 3..k
GRM_DHD inline int kmer_element_lcp(const uint64_t *a, const uint64_t *prev, int words, int k, uint64_t e, uint64_t n_real)
{
    int lcp;
    if (e == 0) lcp = 0;
    else if (e < n_real) lcp = kmer_lcp(a, prev, words, k);
    else lcp = e == n_real ? 0 : k;          // first pad element: zeros against letters; later ones repeat the pad before them
    return lcp < 3 ? 0 : lcp;
}

}  // namespace dfl
}  // namespace grm
