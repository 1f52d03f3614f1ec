// grm_device_fns.h -- per-lane primitives of the gfx950 k-mer engine.
//
// Everything here is a pure function of its arguments and is marked
// __host__ __device__ so that tests/host_check.cpp can run the exact same code on
// the CPU (no GPU in the build container) against the oracle.  Cooperative parts
// (LDS, ballots, scans) live in grm_kernels.hip.
//
// Conventions ([EXT] GATB-core 1.4.2):
//   2-bit code  = (ascii >> 1) & 3      A/a=0 C/c=1 T/t=2 G/g=3
//   bad symbol  = (ascii >> 3) & 1      N/n and the IUPAC letters with that bit
//   complement  = code ^ 2
//   k-mer value = first base most significant; canonical = min(fwd, revcomp)
//
// Packed symbol stream ("sym2"/"inv"), produced by parse_pack_kernel:
//   group G covers symbols [64G, 64G+64)
//   sym2[2G]   : symbols 64G+0..31, MSB-first  (symbol s at bits 63-2s..62-2s)
//   sym2[2G+1] : symbols 64G+32..63
//   inv[G]     : bit s set  <=>  symbol 64G+s is a separator / bad base
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GRM_HD __host__ __device__ __forceinline__
#else
#define GRM_HD inline
#endif

namespace grm {

constexpr uint64_t EMPTY_KEY = ~0ull;   // never a canonical k-mer for k<=32 (canon(GG..G)=CC..C)

constexpr int T_NONE = 0, T_SEQ = 1, T_HDR = 2;   // line type carried across chunks

GRM_HD uint64_t brev64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
    return (x >> 32) | (x << 32);
#endif
}

GRM_HD uint32_t brev32(uint32_t x) { return (uint32_t)(brev64((uint64_t)x) >> 32); }

// reverse the order of the 32 two-bit groups of x
GRM_HD uint64_t rev_groups64(uint64_t x)
{
    uint64_t y = brev64(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}

// reverse complement of an m-symbol word (first symbol most significant), m in 0..32
GRM_HD uint64_t revcomp_m(uint64_t v, int m)
{
    if (m == 0) return 0;
    uint64_t r = rev_groups64(v) >> (64 - 2 * m);
    uint64_t mask = m == 32 ? ~0ull : ((1ull << (2 * m)) - 1);
    return (r ^ 0xAAAAAAAAAAAAAAAAull) & mask;
}

// spread the 32 bits of v to the even bit positions of a 64-bit word
GRM_HD uint64_t spread32(uint32_t v32)
{
    uint64_t v = v32;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// b0/b1: bit s = low/high code bit of symbol s (s=0..31) -> MSB-first packed word
GRM_HD uint64_t pack32_msb_first(uint32_t b0, uint32_t b1)
{
    return spread32(brev32(b0)) | (spread32(brev32(b1)) << 1);
}

// 64-bit mixer used for bucket selection and LDS slots.  Bijective (odd multiplies,
// xor-shift), so distinct k-mers never alias before the final masking.
GRM_HD uint64_t mix64(uint64_t x)
{
    x *= 0x9E3779B97F4A7C15ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    return x;
}
// radix bucket: top `bb` bits; sub-bucket: the next `sb` bits; slot bits: folded rest
GRM_HD uint32_t hash_bucket(uint64_t h, int bb) { return bb ? (uint32_t)(h >> (64 - bb)) : 0u; }
GRM_HD uint32_t hash_sub(uint64_t h, int bb, int sb)
{
    return sb ? (uint32_t)((h << bb) >> (64 - sb)) : 0u;
}
GRM_HD uint32_t hash_slot(uint64_t h, uint32_t cap_mask) { return (uint32_t)(h ^ (h >> 29)) & cap_mask; }

// ---- FASTA byte classification -------------------------------------------------------
// masks over one 16-byte chunk: bit j <=> byte j
GRM_HD void chunk_masks(const uint32_t w[4], uint32_t &nl, uint32_t &gt, uint32_t &cr)
{
    nl = gt = cr = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
        nl |= (uint32_t)(b == '\n') << i;
        gt |= (uint32_t)(b == '>') << i;
        cr |= (uint32_t)(b == '\r') << i;
    }
}

// type of the LAST line that starts inside the chunk (T_NONE if no line starts here).
// ls = line-start mask = ((nl << 1) | prev_byte_is_nl) & 0xffff
GRM_HD int chunk_last_event(uint32_t ls, uint32_t gt)
{
    if (!ls) return T_NONE;
    int pos = 31 - __builtin_clz(ls);
    return ((gt >> pos) & 1u) ? T_HDR : T_SEQ;
}

// Walk the 16 bytes with incoming line type `cur`.
//   emit  : bit j <=> byte j yields a symbol (a base of a sequence line, or the '>' that
//           opens a header line, which yields one separator)
//   sep   : subset of emit that are separators
//   unk   : bytes that would be symbols if the (still unknown) incoming type were T_SEQ
GRM_HD void chunk_classify(uint32_t nl, uint32_t gt, uint32_t cr, uint32_t ls, int cur,
                           uint32_t &emit, uint32_t &sep, uint32_t &unk)
{
    emit = sep = unk = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t bit = 1u << j;
        if (ls & bit) cur = (gt & bit) ? T_HDR : T_SEQ;
        bool plain = !((nl | cr) & bit);
        if ((ls & gt) & bit) { emit |= bit; sep |= bit; }
        else if (plain && cur == T_SEQ) emit |= bit;
        else if (plain && cur == T_NONE) unk |= bit;
    }
}

// ---- k-mer windows of one 64-symbol group ------------------------------------------
// valid-start mask for the 64 positions of a group: position p is valid iff none of the
// k symbols p..p+k-1 is flagged in the 128-bit window (i1:i0).  k in 1..64.
GRM_HD uint64_t valid_starts(uint64_t i0, uint64_t i1, int k)
{
    uint64_t lo = i0, hi = i1;
    int covered = 1;
    while (covered * 2 <= k) {
        int s = covered;                       // 1..32
        lo |= (lo >> s) | (hi << (64 - s));
        hi |= hi >> s;
        covered *= 2;
    }
    int r = k - covered;                       // 0..covered-1 (<64)
    if (r) {
        lo |= (lo >> r) | (hi << (64 - r));
    }
    return ~lo;
}

GRM_HD uint32_t sym_at(uint64_t a, uint64_t b, uint64_t c, int j)   // j in 0..95
{
    uint64_t w = j < 32 ? a : (j < 64 ? b : c);
    return (uint32_t)(w >> (62 - 2 * (j & 31))) & 3u;
}

// Calls f(i, canonical) for every valid start position i (0..63) of the group.
// a,b,c = sym2[2G], sym2[2G+1], sym2[2G+2]; valid = valid_starts() already clipped to the
// stream end.  k in 1..32.
template <typename F>
GRM_HD void for_each_kmer(uint64_t a, uint64_t b, uint64_t c, uint64_t valid, int k, F &&f)
{
    if (!valid) return;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rcshift = 2 * (k - 1);
    const int m = k - 1;
    uint64_t fwd = m ? (a >> (64 - 2 * m)) : 0;
    uint64_t rc = m ? (revcomp_m(fwd, m) << 2) : 0;
    for (int i = 0; i < 64; i++) {
        uint64_t s = sym_at(a, b, c, i + k - 1);
        fwd = ((fwd << 2) | s) & mask;
        rc = (rc >> 2) | ((s ^ 2) << rcshift);
        if ((valid >> i) & 1) f(i, fwd < rc ? fwd : rc);
    }
}

// ---- 32-position variant (one packed word of start positions + the next word) --------
// valid-start mask of the 32 positions p0..p0+31 where p0 = 64G + 32*half; k in 1..32
GRM_HD uint32_t valid_starts32(uint64_t i0, uint64_t i1, int half, int k)
{
    uint64_t acc = half ? ((i0 >> 32) | (i1 << 32)) : i0;   // inv bits of symbols p0 .. p0+63
    int covered = 1;
    while (covered * 2 <= k) {
        acc |= acc >> covered;
        covered *= 2;
    }
    const int r = k - covered;
    if (r) acc |= acc >> r;
    return ~(uint32_t)acc;          // positions 0..31 only look at bits 0..62
}

// a = packed word of symbols p0..p0+31, b = the next word; calls f(i, canonical) for the
// valid start positions i in 0..31.  The loop is meant to be fully unrolled so that a caller
// may keep the k-mers in registers (statically indexed array).
template <typename F>
GRM_HD void for_each_kmer32(uint64_t a, uint64_t b, uint32_t valid, int k, F &&f)
{
    if (!valid) return;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rcshift = 2 * (k - 1);
    const int m = k - 1;
    uint64_t fwd = m ? (a >> (64 - 2 * m)) : 0;
    uint64_t rc = m ? (revcomp_m(fwd, m) << 2) : 0;
    // stream = the symbols after the first m (m <= 31), MSB-aligned: the next symbol is
    // always the top 2 bits of hi
    uint64_t hi = m ? ((a << (2 * m)) | (b >> (64 - 2 * m))) : a;
    uint64_t lo = m ? (b << (2 * m)) : b;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint64_t s = hi >> 62;
        hi = (hi << 2) | (lo >> 62);
        lo <<= 2;
        fwd = ((fwd << 2) | s) & mask;
        rc = (rc >> 2) | ((s ^ 2) << rcshift);
        if ((valid >> i) & 1u) f(i, fwd < rc ? fwd : rc);
    }
}

}  // namespace grm
