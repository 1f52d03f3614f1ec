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

// Hash used for bucket selection and LDS slots.  It only has to spread k-mers evenly (tables
// compare full keys), so it is built from FULL-RATE 24-bit multiplies (v_mul_u32_u24): a 64-bit
// multiply costs ~4 quarter-rate 32-bit multiplies on CDNA and the hash is evaluated for every
// k-mer occurrence in five kernels.  Product bit j depends on operand bits 0..j, so the TOP bits
// of each 24x24 product depend on all 24 input bits: the bucket (top bits of h32) sees the whole key.
GRM_HD uint32_t mul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)((uint64_t)(a & 0xffffffu) * (uint64_t)(b & 0xffffffu));
#endif
}
// returns (h32 << 32) | slot_bits : bucket / sub-bucket come from the top of h32
GRM_HD uint64_t mix64(uint64_t x)
{
    const uint32_t a = (uint32_t)x & 0xffffffu, b = (uint32_t)(x >> 24) & 0xffffffu, c = (uint32_t)(x >> 48);
    // (no "t + (t << 16)" style term: the compiler folds it into a quarter-rate v_mul_lo_u32)
    const uint32_t h32 = mul24(a, 0x9E3779u) + mul24(b, 0x85EBCBu) + mul24(c, 0xC2B2AFu);
    const uint32_t low = mul24(h32 ^ (h32 >> 12), 0xD6E8FFu) ^ (h32 >> 7);
    return ((uint64_t)h32 << 32) | low;
}
// radix bucket: top `bb` bits; sub-bucket: the next `sb` bits (bb + sb <= 24); slot: low word
// Both read h32 only (bb + sb <= 24), so a caller that needs no slot never computes the low word;
// (x >> 1) >> (31 - n) is x >> (32 - n) that also holds for n = 0, without a select.
GRM_HD uint32_t hash_bucket(uint64_t h, int bb) { return ((uint32_t)(h >> 32) >> 1) >> (31 - bb); }
GRM_HD uint32_t hash_sub(uint64_t h, int bb, int sb)
{
    return (((uint32_t)(h >> 32) << bb) >> 1) >> (31 - sb);
}
GRM_HD uint32_t hash_slot(uint64_t h, uint32_t cap_mask) { return (uint32_t)h & cap_mask; }

// ---- minimizers (record form of the partition, grm_superkmer.hip) ----------------------------
// Order of the canonical m-mers (x < 2^22): the m-mer with the smallest value inside a k-mer is its minimizer.  The order
// only has to look random, and it is evaluated for every symbol: ONE full-rate multiply-add, of which the top 24 bits
// count (the constant term keeps poly-A, x = 0, from being the smallest m-mer of every genome).
constexpr int MINIMIZER_ORDER_BITS = 24;
// The multiplier is one of those for which x -> (x * C + K) mod 2^32 >> 8 is INJECTIVE on x < 2^22 (0x9E3779 collides for two thirds
// of the m-mers: different m-mers with one order value share a bucket, and the buckets get as uneven as with half the minimizers).
GRM_HD uint32_t minimizer_hash(uint32_t x) { return mul24(x, 0xEC0C71u) + 0x7F4A7C15u; }      // order = result >> 8
// bucket of a k-mer from the order value (24 bits) of its minimizer.  The minimum of k - 10 values crowds towards 0, so
// the bucket is NOT its top bits: the value is hashed once more (per run, not per position).
GRM_HD uint32_t minimizer_bucket(uint32_t order, int bb)
{
    const uint32_t t = mul24(order ^ (order >> 11), 0xC2B2AFu) + mul24(order >> 5, 0x85EBCBu);
    return (t >> 1) >> (31 - bb);
}

// the same bucket from the k-mer itself (canonical or not: both strands hold the same canonical m-mers), k >= m_len
GRM_HD uint32_t minimizer_bucket_of_kmer(uint64_t key, int k, int bb, int m_len = 11)
{
    const uint32_t mmask = (1u << (2 * m_len)) - 1;
    uint32_t f = 0, r = 0, omin = ~0u;
    for (int j = 0; j < k; j++) {
        const uint32_t s = (uint32_t)(key >> (2 * (k - 1 - j))) & 3u;
        f = ((f << 2) | s) & mmask;
        r = (r >> 2) | ((s ^ 2u) << (2 * (m_len - 1)));
        if (j >= m_len - 1) {
            const uint32_t o = minimizer_hash(f < r ? f : r) >> (32 - MINIMIZER_ORDER_BITS);
            omin = o < omin ? o : omin;
        }
    }
    return minimizer_bucket(omin, bb);
}

// ---- run records (record form of the partition): per-lane logic of grm_superkmer.hip and of dict_build's decoder ----
// A RUN is a maximal stretch of consecutive valid k-mer starts that share their minimizer OCCURRENCE (the leftmost m-mer
// with the smallest order value among the W = k - M + 1 the k-mer contains).  Where a run starts and ends is a property of
// the sequence around it (2 W - 1 m-mers), not of where a thread's window or a contig begins: the same sequence gives the
// same runs in every genome, however its assembly is cut, ordered or shifted by an indel upstream.  A run holds at most W
// k-mers (all contain the one m-mer), i.e. at most 2 k - M = 53 bases at k = 32, and travels as ONE 16-byte record
//     x            bases 0..31 of the run, MSB-first
//     y[63..22]    bases 32..52; bits behind the run's last base are 0
//     y[21..12]    0 (bit 12 between level 1 and level 2: "store me on the other strand", run_flip)
//     y[11..5]     7 bucket bits below the coarse ones
//     y[4..0]      k-mers in the run (1..22)
// stored STRAND-CANONICALLY: in the orientation in which the minimizer m-mer is its own canonical form (M is odd: an
// m-mer is never its own reverse complement).  A contig and its reverse complement therefore give the same records; the
// decoder emits min(forward, reverse complement) of every k-mer, so the orientation needs no flag.
constexpr int RUN_PPT = 32;          // k-mer start positions a thread takes per step: one packed word
constexpr int RUN_LMAX = 22;         // k-mers per record: W at k = 32
constexpr int RUN_FINE_BITS = 7;     // bucket bits a record carries below the coarse ones
constexpr int RUN_LEN_BITS = 5;
GRM_HD uint32_t run_len(uint64_t y) { return (uint32_t)y & ((1u << RUN_LEN_BITS) - 1u); }
GRM_HD uint32_t run_fine(uint64_t y) { return ((uint32_t)y >> RUN_LEN_BITS) & ((1u << RUN_FINE_BITS) - 1u); }

// the 16 symbols from symbol offset `off` (0..63) of the 64 symbols held MSB-first in the four words d[0..3], as one word (zeros behind
// the 64th); with off known at compile time (unrolled callers) this is one v_alignbit_b32, or nothing
GRM_HD uint32_t sym_window(const uint32_t (&d)[4], int off)
{
    const int b = 2 * off, j = b >> 5, sh = b & 31;
    const uint32_t hi = d[j], lo = j + 1 < 4 ? d[j + 1] : 0u;
    return sh ? (hi << sh) | (lo >> (32 - sh)) : hi;
}
GRM_HD uint32_t min3u(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t ab = a < b ? a : b;          // (the backend fuses the pair into v_min3_u32)
    return ab < c ? ab : c;
}
// Minimizer words of a window.  w0 = the packed word of positions 0..31, w1 the next one, prev2 = the 2-bit code of position -1.
// The word of the M-mer at position q (index q + 1 of h):
//     bits 31..8   its order value
//     bits  6..1   q + 1 (0 .. 31 + W)
//     bit   0      1 = the reverse complement of the m-mer is the canonical one
// An m-mer and its reverse complement are read as the TOP 22 bits of two 16-symbol words at fixed offsets of (w0 : w1) and of its
// reverse complement rc(w1) : rc(w0) -- no rolling words; the 10 bits below only break ties that cannot occur (M is odd: an m-mer
// differs from its reverse complement), so min / compare act on the m-mers themselves.  run_hashes fills h[0 .. n): 8 instructions per
// m-mer (two v_alignbit, min, compare, shift, multiply-add, and-or, add-with-carry).
// base + (a < b) / 2 acc + (a != b): a compare and an add-with-carry (the compiler makes three or four instructions of either)
GRM_HD uint32_t add_is_less(uint32_t base, uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t out;
    __asm__("v_cmp_lt_u32 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, 0, %1, vcc" : "=v"(out) : "v"(base), "v"(a), "v"(b) : "vcc");
    return out;
#else
    return base + (uint32_t)(a < b);
#endif
}
GRM_HD uint32_t shift_in_differs(uint32_t acc, uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __asm__("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    return acc;
#else
    return acc + acc + (uint32_t)(a != b);
#endif
}
template <int M = 11>
GRM_HD uint32_t run_hash_word(uint32_t f32, uint32_t r32, uint32_t pos_field)
{
    constexpr uint32_t KEEP = ~0u << (32 - MINIMIZER_ORDER_BITS);
    const uint32_t m = (f32 < r32 ? f32 : r32) >> (32 - 2 * M);
    return add_is_less((minimizer_hash(m) & KEEP) | pos_field, r32, f32);
}
template <int N, int M = 11>
GRM_HD void run_hashes(uint64_t w0, uint64_t w1, uint32_t prev2, uint32_t (&h)[N])
{
    static_assert(N - 2 + M <= 64, "the m-mers lie inside the two words");
    const uint64_t r1 = revcomp_m(w1, 32), r0 = revcomp_m(w0, 32);
    const uint32_t df[4] = {(uint32_t)(w0 >> 32), (uint32_t)w0, (uint32_t)(w1 >> 32), (uint32_t)w1};
    const uint32_t dr[4] = {(uint32_t)(r1 >> 32), (uint32_t)r1, (uint32_t)(r0 >> 32), (uint32_t)r0};
    // position -1: the symbol before the window, then the first M - 1 of w0; its reverse complement: rc of those, then rc of the symbol
    h[0] = run_hash_word<M>((prev2 << 30) | (df[0] >> 2), sym_window(dr, 64 - (M - 1)) | ((prev2 ^ 2u) << (32 - 2 * M)), 0u);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < N - 1; q++) h[q + 1] = run_hash_word<M>(sym_window(df, q), sym_window(dr, 64 - M - q), (uint32_t)(q + 1) << 1);
}
// val[i + 1] = the smallest word among the W m-mers of the k-mer at position i (-1 .. 31): its minimizer, the LEFTMOST one of equal
// order (the position is part of the compared word).  h: the words of the m-mers at positions -1 .. 31 + W - 1; window minimum by
// spans of 3 and 9 (v_min3_u32), in place.
template <int W>
GRM_HD void run_window_min(uint32_t (&h)[RUN_PPT + W], uint32_t (&val)[RUN_PPT + 1])
{
    constexpr int NM = RUN_PPT + W;
    static_assert(W >= 1 && W <= 27, "window minimum by spans of 1, 3 or 9");
    constexpr int S = W <= 3 ? 1 : W <= 9 ? 3 : 9;
    if (S >= 3) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i + 2 < NM; i++) h[i] = min3u(h[i], h[i + 1], h[i + 2]);             // [i, i + 3)
    }
    if (S >= 9) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i + 8 < NM; i++) h[i] = min3u(h[i], h[i + 3], h[i + 6]);             // [i, i + 9)
    }
    constexpr int MID = S < W - S ? S : W - S;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i <= RUN_PPT; i++) val[i] = min3u(h[i], h[i + MID], h[i + W - S]);     // [i, i + W)
}
// all of it by one thread (host emulation; the kernel shares the hashes of neighbouring windows across lanes, grm_superkmer.hip)
template <int W, int M = 11>
GRM_HD void run_minimizers(uint64_t w0, uint64_t w1, uint32_t prev2, uint32_t (&val)[RUN_PPT + 1])
{
    uint32_t h[RUN_PPT + W];
    run_hashes<RUN_PPT + W, M>(w0, w1, prev2, h);
    run_window_min<W>(h, val);
}
// a word of the window to the right, seen from this window: the m-mer lies RUN_PPT positions further on
GRM_HD uint32_t run_hash_from_right(uint32_t h_right) { return h_right + ((uint32_t)RUN_PPT << 1); }
// first positions of the runs of a window: a valid start whose predecessor is invalid or has another minimizer occurrence.
// valid: bit i = position i is a valid k-mer start; prev_valid: position -1 is one (of the same genome)
GRM_HD uint32_t run_heads(uint32_t valid, bool prev_valid, const uint32_t (&val)[RUN_PPT + 1])
{
    uint32_t differs = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = RUN_PPT - 1; i >= 0; i--) differs = shift_in_differs(differs, val[i + 1], val[i]);
    return valid & (differs | ~((valid << 1) | (uint32_t)prev_valid));
}
// positions at the front of a window that continue a run begun before it (0 when position 0 starts a run or is no k-mer start)
GRM_HD uint32_t run_lead(uint32_t valid, uint32_t heads)
{
    return (uint32_t)__builtin_ctz((heads | ~valid) | 0x80000000u);
}
// k-mers of the run that starts at position i, inside the window: up to the next head, the next invalid start, or the end of the
// window -- a run that reaches the end goes on for run_lead() of the next window
GRM_HD uint32_t run_length(uint32_t heads, uint32_t valid, int i)
{
    const uint64_t bnd = (uint64_t)(heads | ~valid) | (1ull << RUN_PPT);
    const uint64_t rest = bnd >> (i + 1);
    return (uint32_t)__builtin_ctzll(rest) + 1u;
}
// The 16-byte record of the run of `len` k-mers that starts at position i of the window (w0, w1, w2: the window's word and
// the two after it): its len + k - 1 bases as they stand in the stream; fine = the 7 bucket bits below the coarse ones.
// flip: the run is to be stored on the other strand -- level 1 only SAYS so (bit 12 of y), level 2 turns the record over
// (run_flip) on its way through: there every lane holds a record, here the lanes with most runs set the pace.
constexpr uint64_t RUN_FLIP_BIT = 1ull << (RUN_LEN_BITS + RUN_FINE_BITS);
GRM_HD void run_record(uint64_t w0, uint64_t w1, uint64_t w2, int i, uint32_t len, int k, bool flip, uint32_t fine, uint64_t &x, uint64_t &y)
{
    const int span = (int)len + k - 1;              // 11 .. 53 bases
    const uint64_t a = i ? ((w0 << (2 * i)) | (w1 >> (64 - 2 * i))) : w0;
    const uint64_t b = i ? ((w1 << (2 * i)) | (w2 >> (64 - 2 * i))) : w1;
    // the first `span` bases: masks of 2 span bits from the top of the 128
    const int up = 128 - 2 * span;                  // 22 .. 106
    const uint64_t ma = up >= 64 ? ~0ull << (up - 64) : ~0ull, mb = up >= 64 ? 0ull : ~0ull << up;
    x = a & ma;
    y = (b & mb) | (flip ? RUN_FLIP_BIT : 0ull) | ((uint64_t)(fine & ((1u << RUN_FINE_BITS) - 1u)) << RUN_LEN_BITS) | len;
}
// a record marked by level 1: reverse complement of its len + k - 1 bases, left-aligned again; the mark goes
GRM_HD void run_flip(uint64_t &x, uint64_t &y, int k)
{
    if (!(y & RUN_FLIP_BIT)) return;
    const uint64_t low = y & (RUN_FLIP_BIT - 1);
    const int span = (int)run_len(y) + k - 1;
    const uint64_t a = x, b = y & ~(2 * RUN_FLIP_BIT - 1);
    // reverse complement of the 64 bases a : b = rc(b) : rc(a); the run's bases are its LAST `span` ones
    const uint64_t hi = revcomp_m(b, 32), lo = revcomp_m(a, 32);
    const int sh = 2 * (64 - span);                  // 22 .. 106
    if (sh >= 64) {
        x = lo << (sh - 64);
        y = low;
    } else {
        x = (hi << sh) | (lo >> (64 - sh));
        y = (lo << sh) | low;
    }
}
// decoder state of a record: forward / reverse-complement words of the current k-mer, and the bases after it, MSB-aligned
struct RunDecoder {
    uint64_t fwd, rc, rest;
};
GRM_HD RunDecoder run_open(uint64_t x, uint64_t y, int k)
{
    RunDecoder d;
    const int up = 64 - 2 * k;
    d.fwd = x >> up;
    d.rc = revcomp_m(d.fwd, k);
    d.rest = k < 32 ? ((x << (2 * k)) | (y >> up)) : y;       // (at most W - 1 = k - 11 of its bases are read: all above y's low fields)
    return d;
}
GRM_HD uint64_t run_canonical(const RunDecoder &d) { return d.fwd < d.rc ? d.fwd : d.rc; }
GRM_HD void run_next(RunDecoder &d, uint64_t kmask, int rcshift)
{
    const uint64_t sy = d.rest >> 62;
    d.rest <<= 2;
    d.fwd = ((d.fwd << 2) | sy) & kmask;
    d.rc = (d.rc >> 2) | ((sy ^ 2ull) << rcshift);
}
// canonical k-mer number t of a record without rolling up to it: bases t .. t + k - 1 of the run (x: bases 0..31, y's top bits: the
// bases from 32 on; t <= RUN_LMAX - 1 = 21 and t + k <= 53, so y's low fields are never reached)
GRM_HD uint64_t run_kmer_at(uint64_t x, uint64_t y, int k, uint32_t t)
{
    const uint64_t hi = t ? ((x << (2 * t)) | (y >> (64 - 2 * t))) : x;
    const uint64_t fwd = hi >> (64 - 2 * k);
    const uint64_t rc = revcomp_m(fwd, k);
    return fwd < rc ? fwd : rc;
}

// ---- FASTA byte classification -------------------------------------------------------
// 4-bit mask of the bytes of x that equal c (SWAR exact zero-byte test, then bit gather)
GRM_HD uint32_t byte_eq_mask4(uint32_t x, uint32_t c)
{
    const uint32_t y = x ^ (c * 0x01010101u);
    const uint32_t z = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);   // 0x80 in every zero byte
    return (((z >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xfu;
}
// masks over one 16-byte chunk: bit j <=> byte j
// nonzero iff a byte of x equals c (the classic zero-byte test: exact as a yes / no, four instructions)
GRM_HD uint32_t byte_eq_any4(uint32_t x, uint32_t c)
{
    const uint32_t y = x ^ (c * 0x01010101u);
    return (y - 0x01010101u) & ~y & 0x80808080u;
}
// (need_gt false: FASTQ -- '>' is a common quality value there and no line type depends on it)
GRM_HD void chunk_masks(const uint32_t w[4], uint32_t &nl, uint32_t &gt, uint32_t &cr, bool need_gt = true)
{
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    nl = byte_eq_mask4(w0, '\n') | (byte_eq_mask4(w1, '\n') << 4) | (byte_eq_mask4(w2, '\n') << 8) | (byte_eq_mask4(w3, '\n') << 12);
#if defined(__HIP_DEVICE_COMPILE__)
    // '>' and CR are rare (a header per contig; no CR at all in most files): a wave whose 1 KiB holds neither -- nearly every wave --
    // skips their masks (2 x 36 of the ~125 instructions of this function; the parse kernels are bound by instruction issue).
    // Called with every lane active (tile_round / tile_round_fq / tile_rounds_fq).
    uint32_t rare = byte_eq_any4(w0, '\r') | byte_eq_any4(w1, '\r') | byte_eq_any4(w2, '\r') | byte_eq_any4(w3, '\r');
    if (need_gt) rare |= byte_eq_any4(w0, '>') | byte_eq_any4(w1, '>') | byte_eq_any4(w2, '>') | byte_eq_any4(w3, '>');
    if (!__any(rare != 0u)) {
        gt = 0;
        cr = 0;
        return;
    }
#endif
    gt = need_gt ? byte_eq_mask4(w0, '>') | (byte_eq_mask4(w1, '>') << 4) | (byte_eq_mask4(w2, '>') << 8) | (byte_eq_mask4(w3, '>') << 12) : 0u;
    cr = byte_eq_mask4(w0, '\r') | (byte_eq_mask4(w1, '\r') << 4) | (byte_eq_mask4(w2, '\r') << 8) | (byte_eq_mask4(w3, '\r') << 12);
}

// type of the LAST line that starts inside the chunk (T_NONE if no line starts here).
// ls = line-start mask = ((nl << 1) | prev_byte_is_nl) & 0xffff
GRM_HD int chunk_last_event(uint32_t ls, uint32_t gt)
{
    if (!ls) return T_NONE;
    int pos = 31 - __builtin_clz(ls);
    return ((gt >> pos) & 1u) ? T_HDR : T_SEQ;
}

// Classify the 16 bytes given the incoming line type `cur` (bit-parallel, branch-free).
//   emit  : bit j <=> byte j yields a symbol (a base of a sequence line, or the '>' that
//           opens a header line, which yields one separator)
//   sep   : subset of emit that are separators
//   unk   : bytes that would be symbols if the (still unknown) incoming type were T_SEQ
GRM_HD void chunk_classify(uint32_t nl, uint32_t gt, uint32_t cr, uint32_t ls, int cur,
                           uint32_t &emit, uint32_t &sep, uint32_t &unk)
{
    const uint32_t hs = ls & gt;                    // header-line starts
    const uint32_t ss = ls & ~gt;                   // sequence-line starts
    // flood every sequence-line start upward until the next line start (Kogge-Stone fill)
    uint32_t fill = ss, prop = ~ls;
    fill |= (fill << 1) & prop; prop &= prop << 1;
    fill |= (fill << 2) & prop; prop &= prop << 2;
    fill |= (fill << 4) & prop; prop &= prop << 4;
    fill |= (fill << 8) & prop;
    const uint32_t first = ls ? (ls & (0u - ls)) : 0x10000u;
    const uint32_t prefix = first - 1;              // bytes before the first line start of the chunk
    const uint32_t plain = ~(nl | cr) & 0xffffu;
    sep = hs;
    emit = (hs | (plain & fill) | (cur == T_SEQ ? (plain & prefix) : 0u)) & 0xffffu;
    unk = cur == T_NONE ? (plain & prefix) : 0u;
}

// ---- associative summary of a byte range (one scan per tile) ---------------------------
// element = (ev, cs, ch): type of the last line starting in the range (0 = none), symbols the
// range emits when the line running into it is a sequence line (cs) / a header line (ch).
// Packed: ev bits 60-61, cs bits 30-59, ch bits 0-29.  combine(a, b) = summary of "a then b".
GRM_HD uint64_t pelem_make(int ev, uint32_t cs, uint32_t ch)
{
    return ((uint64_t)ev << 60) | ((uint64_t)cs << 30) | (uint64_t)ch;
}
GRM_HD int pelem_ev(uint64_t e) { return (int)(e >> 60); }
GRM_HD uint32_t pelem_cs(uint64_t e) { return (uint32_t)(e >> 30) & 0x3fffffffu; }
GRM_HD uint32_t pelem_ch(uint64_t e) { return (uint32_t)e & 0x3fffffffu; }
GRM_HD uint64_t pelem_combine(uint64_t a, uint64_t b)
{
    const int ea = pelem_ev(a), eb = pelem_ev(b);
    const uint32_t b_after_seq = pelem_cs(b), b_after_hdr = pelem_ch(b);
    // line type running into b: a's last line start if it has one, else what ran into a
    const uint32_t cs = pelem_cs(a) + ((ea ? ea : T_SEQ) == T_SEQ ? b_after_seq : b_after_hdr);
    const uint32_t ch = pelem_ch(a) + ((ea ? ea : T_HDR) == T_SEQ ? b_after_seq : b_after_hdr);
    return pelem_make(eb ? eb : ea, cs, ch);
}
// the same element in 32 bits for a scan inside ONE 16 KiB tile (counts <= 16384 < 2^15): half the cross-lane moves
GRM_HD uint32_t pelem32_make(int ev, uint32_t cs, uint32_t ch) { return ((uint32_t)ev << 30) | (cs << 15) | ch; }
GRM_HD int pelem32_ev(uint32_t e) { return (int)(e >> 30); }
GRM_HD uint32_t pelem32_cs(uint32_t e) { return (e >> 15) & 0x7fffu; }
GRM_HD uint32_t pelem32_ch(uint32_t e) { return e & 0x7fffu; }
GRM_HD uint32_t pelem32_combine(uint32_t a, uint32_t b)
{
    const int ea = pelem32_ev(a), eb = pelem32_ev(b);
    const uint32_t b_after_seq = pelem32_cs(b), b_after_hdr = pelem32_ch(b);
    const uint32_t cs = pelem32_cs(a) + ((ea ? ea : T_SEQ) == T_SEQ ? b_after_seq : b_after_hdr);
    const uint32_t ch = pelem32_ch(a) + ((ea ? ea : T_HDR) == T_SEQ ? b_after_seq : b_after_hdr);
    return pelem32_make(eb ? eb : ea, cs, ch);
}

// ---- FASTQ (4-line records: header, sequence, '+', quality) -------------------------------
// A byte's role depends on its line index mod 4 ("phase").  Header lines (phase 0) emit one
// separator at their first byte, sequence lines (phase 1) emit every byte except \n / \r.
// exclusive prefix parity of a 16-bit mask
GRM_HD uint32_t excl_prefix_xor16(uint32_t x)
{
    x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8;
    return (x << 1) & 0xffffu;
}
// masks of the chunk's bytes by (number of newlines before the byte inside the chunk) mod 4
GRM_HD void fq_phase_masks(uint32_t nl, uint32_t m[4])
{
    const uint32_t e0 = excl_prefix_xor16(nl);           // bit 0 of the running newline count
    const uint32_t e1 = excl_prefix_xor16(nl & e0);      // bit 1: parity of the carries
    m[0] = ~e0 & ~e1 & 0xffffu;
    m[1] = e0 & ~e1 & 0xffffu;
    m[2] = ~e0 & e1 & 0xffffu;
    m[3] = e0 & e1 & 0xffffu;
}
// emit / separator masks of a chunk that starts in line phase s (0..3)
GRM_HD void fq_classify(uint32_t nl, uint32_t cr, uint32_t ls, const uint32_t m[4], int s, uint32_t &emit, uint32_t &sep)
{
    const uint32_t plain = ~(nl | cr) & 0xffffu;
    sep = ls & m[(0 - s) & 3];                          // first byte of a header line
    emit = sep | (plain & m[(1 - s) & 3]);              // + the bytes of a sequence line
}
// FASTQ scan element: newline count mod 4 (bits 60-61) and, for each starting phase s, the
// symbols the range emits (15 bits each, bits 15s..15s+14; a tile holds <= 16384 bytes).
GRM_HD uint64_t fq_elem_make(uint32_t nl_mod4, const uint32_t cnt[4])
{
    return ((uint64_t)nl_mod4 << 60) | (uint64_t)cnt[0] | ((uint64_t)cnt[1] << 15) | ((uint64_t)cnt[2] << 30) | ((uint64_t)cnt[3] << 45);
}
GRM_HD uint32_t fq_elem_nl(uint64_t e) { return (uint32_t)(e >> 60) & 3u; }
GRM_HD uint32_t fq_elem_cnt(uint64_t e, int s) { return (uint32_t)(e >> (15 * s)) & 0x7fffu; }
GRM_HD uint64_t fq_elem_combine(uint64_t a, uint64_t b)
{
    const uint32_t na = fq_elem_nl(a);
    uint32_t c[4];
#pragma unroll
    for (int s = 0; s < 4; s++) c[s] = fq_elem_cnt(a, s) + fq_elem_cnt(b, (s + (int)na) & 3);
    return fq_elem_make((na + fq_elem_nl(b)) & 3u, c);
}

// the emitted symbols of one 16-byte chunk as bit strings: returns the count (0..16);
// sym: 2 bits per symbol, first symbol most significant, in the low 2*count bits;
// inv: bit i = symbol i is a separator / bad base.
// 2-bit codes of the 4 bytes of x as one byte, first byte most significant; bad-base bits of the 4 bytes, first byte in bit 0
GRM_HD uint32_t codes_of_word(uint32_t x)
{
    uint32_t c = (x >> 1) & 0x03030303u;
    c = (c >> 24) | ((c >> 8) & 0xff00u) | ((c << 8) & 0xff0000u) | (c << 24);         // byte swap: first byte on top
    return (c | (c >> 6) | (c >> 12) | (c >> 18)) & 0xffu;
}
GRM_HD uint32_t bad_of_word(uint32_t x)
{
    const uint32_t b = (x >> 3) & 0x01010101u;
    return (b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xfu;
}
GRM_HD int chunk_pack(const uint32_t w[4], uint32_t emit, uint32_t sep, uint32_t &sym, uint32_t &inv)
{
    emit &= 0xffffu;
    sep &= emit;
    const uint32_t holes = ~emit & 0xffffu;
    if (sep == 0 && (holes & (holes - 1)) == 0) {
        // nothing but bases, or all but ONE byte (the newline of an 80-column line): all 16 codes at once (SWAR), then the one
        // hole is closed -- the byte-by-byte loop below spent ~160 instructions per chunk on what is the case of nearly every chunk
        uint32_t s16 = (codes_of_word(w[0]) << 24) | (codes_of_word(w[1]) << 16) | (codes_of_word(w[2]) << 8) | codes_of_word(w[3]);
        uint32_t b16 = bad_of_word(w[0]) | (bad_of_word(w[1]) << 4) | (bad_of_word(w[2]) << 8) | (bad_of_word(w[3]) << 12);
        if (holes) {
            const int j = __builtin_ctz(holes);                   // byte j is not a symbol: symbols j+1.. move up by one
            const uint32_t low_mask = j == 15 ? 0u : (0x3fffffffu >> (2 * j));         // the symbols after j (2-bit groups, MSB-first)
            const uint32_t high = j == 0 ? 0u : (s16 & ~(0xffffffffu >> (2 * j)));
            s16 = (high >> 2) | (s16 & low_mask);
            b16 = (b16 & ((1u << j) - 1u)) | ((b16 >> (j + 1)) << j);
            sym = s16;                                            // 15 symbols in the low 30 bits
            inv = b16;
            return 15;
        }
        sym = s16;
        inv = b16;
        return 16;
    }
    sym = 0; inv = 0;
    int n = 0;
    uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 16; j++) {
        const uint32_t b = (uint32_t)lo & 0xffu;          // byte j (no indexed access: stays in registers)
        lo = (lo >> 8) | (hi << 56);
        hi >>= 8;
        if ((emit >> j) & 1u) {
            uint32_t code = (b >> 1) & 3u, bad = (b >> 3) & 1u;
            if ((sep >> j) & 1u) { code = 0; bad = 1; }
            sym = (sym << 2) | code;
            inv |= bad << n;
            n++;
        }
    }
    return n;
}

// OR `cnt` symbols (bit strings as produced by chunk_pack) into a packed stream at symbol
// position pos: sym words are MSB-first (32 symbols per uint64), inv words LSB-first (64 per
// uint64).  or_sym(word_index, value) / or_inv(word_index, value) perform the OR (an LDS atomic
// on the device, a plain |= in the host emulation).
template <typename OS, typename OI>
GRM_HD void stream_insert(uint32_t pos, int cnt, uint32_t sym, uint32_t inv, OS &&or_sym, OI &&or_inv)
{
    if (cnt == 0) return;
    const uint32_t w = pos >> 5, o = pos & 31u;
    const int first = (int)(32u - o) < cnt ? (int)(32u - o) : cnt;     // symbols that fit into word w
    const int rest = cnt - first;
    const uint64_t part1 = (uint64_t)(sym >> (2 * rest));              // top `first` symbols
    or_sym(w, part1 << (64 - 2 * (int)o - 2 * first));
    if (rest) {
        const uint64_t part2 = (uint64_t)sym & ((1ull << (2 * rest)) - 1);
        or_sym(w + 1, part2 << (64 - 2 * rest));
    }
    const uint32_t wi = pos >> 6, oi = pos & 63u;
    or_inv(wi, (uint64_t)inv << oi);
    if (oi + (uint32_t)cnt > 64u) or_inv(wi + 1, (uint64_t)inv >> (64 - oi));
}

// ---- "clean" chunks: nothing but bytes of sequence lines and newlines ---------------------------------------------------
// Nearly every 1 KiB of a FASTA holds no '>' and no CR, and then a byte is a symbol unless it is a newline, every line that starts is a
// sequence line, and what a chunk adds to the tile's scan follows from the COUNT of its newlines and the place of the first one -- no
// 16-bit masks, no flood of line types.  The test is conservative: letters have bit 6 set, '\n', '\r', '>' and whatever else a parser
// must look at have not; a chunk with a byte below 0x40 that is not '\n' (or a byte >= 0x80 without bit 6) goes the general way.
GRM_HD uint32_t nl_bytes4(uint32_t x)                    // 0x80 in every byte of x that is '\n' (exact)
{
    const uint32_t y = x ^ 0x0a0a0a0au;
    return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
}
// z[i] = nl_bytes4(w[i]); returns nonzero iff the chunk holds a byte that is neither a newline nor has bit 6 set
GRM_HD uint32_t clean_scan(const uint32_t w[4], uint32_t z[4])
{
    uint32_t odd = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 4; i++) {
        z[i] = nl_bytes4(w[i]);
        odd |= ((~w[i] & 0x40404040u) << 1) ^ z[i];
    }
    return odd;
}
GRM_HD uint32_t clean_nl_count(const uint32_t z[4])
{
    return (uint32_t)(__builtin_popcount(z[0]) + __builtin_popcount(z[1]) + __builtin_popcount(z[2]) + __builtin_popcount(z[3]));
}
GRM_HD uint32_t first_bit_or_max(uint32_t x) { return x ? (uint32_t)__builtin_ctz(x) : 0xffffffffu; }      // (v_ffbl_b32 as it is)
GRM_HD uint32_t clean_first_nl(const uint32_t z[4])      // byte index of the chunk's first newline, 16: none -- no branches
{
    const uint32_t a = first_bit_or_max(z[0]), b = first_bit_or_max(z[1]) | 32u, c = first_bit_or_max(z[2]) | 64u, d = first_bit_or_max(z[3]) | 96u;
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    const uint32_t f = (ab < cd ? ab : cd) >> 3;
    return f < 16u ? f : 16u;
}
GRM_HD uint32_t clean_nl_mask16(const uint32_t z[4])     // bit j <=> byte j is a newline (what chunk_masks calls nl)
{
    auto g = [](uint32_t zz) { return (((zz >> 7) & 0x01010101u) * 0x01020408u) >> 24 & 0xfu; };
    return g(z[0]) | (g(z[1]) << 4) | (g(z[2]) << 8) | (g(z[3]) << 12);
}
// the scan element (pelem32) of a clean chunk == what chunk_classify(.., T_NONE, ..) + chunk_last_event give for it:
// the first line start is byte 0 when the byte before the chunk is a newline, else the byte after the first newline (if that is still
// inside); bytes before it count only if a sequence line runs in (extra), the other non-newline bytes always (ch)
GRM_HD uint32_t clean_elem(const uint32_t z[4], uint32_t prev_nl)
{
    const uint32_t nlc = clean_nl_count(z), f = clean_first_nl(z);
    const uint32_t extra = prev_nl ? 0u : f;             // (f == 15: the line starts in the next chunk; 15 plain bytes before it)
    const uint32_t ch = 16u - nlc - extra;
    return pelem32_make((prev_nl || f < 15u) ? T_SEQ : T_NONE, ch + extra, ch);
}
// the 2-bit codes of all 16 bytes, first byte on top (what chunk_pack's SWAR path builds word by word): on the device a 4 x 4 byte
// transpose (8 v_perm_b32) puts byte k of every word side by side, so that ONE shift and mask per k serves the four words
GRM_HD uint32_t chunk_codes16(const uint32_t w[4])
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t a01 = __builtin_amdgcn_perm(w[0], w[1], 0x04000501u), a23 = __builtin_amdgcn_perm(w[0], w[1], 0x06020703u);
    const uint32_t c01 = __builtin_amdgcn_perm(w[2], w[3], 0x04000501u), c23 = __builtin_amdgcn_perm(w[2], w[3], 0x06020703u);
    const uint32_t t0 = __builtin_amdgcn_perm(a01, c01, 0x07060302u), t1 = __builtin_amdgcn_perm(a01, c01, 0x05040100u);
    const uint32_t t2 = __builtin_amdgcn_perm(a23, c23, 0x07060302u), t3 = __builtin_amdgcn_perm(a23, c23, 0x05040100u);
    return ((t0 << 5) & 0xc0c0c0c0u) | ((t1 << 3) & 0x30303030u) | ((t2 << 1) & 0x0c0c0c0cu) | ((t3 >> 1) & 0x03030303u);
#else
    return (codes_of_word(w[0]) << 24) | (codes_of_word(w[1]) << 16) | (codes_of_word(w[2]) << 8) | codes_of_word(w[3]);
#endif
}
// nonzero iff a byte of the chunk that is not a newline has the bad-base bit ('\n' = 0x0a has it too: nlc of the bytes counted are those)
GRM_HD uint32_t clean_bad_any(const uint32_t w[4], uint32_t nlc)
{
    const uint32_t n = (uint32_t)(__builtin_popcount(w[0] & 0x08080808u) + __builtin_popcount(w[1] & 0x08080808u) +
                                  __builtin_popcount(w[2] & 0x08080808u) + __builtin_popcount(w[3] & 0x08080808u));
    return n ^ nlc;
}
// 16 two-bit symbols, first on top; symbol j (0..15) is taken out and the ones after it move up: 15 symbols in the top 30 bits, zeros
// below.  j == 16: nothing is taken out.
GRM_HD uint32_t close_hole_top(uint32_t s16, uint32_t j)
{
    const uint32_t keep = ~(uint32_t)(0xffffffffull >> (2u * j));     // the 2 j bits of the symbols before j
    return (s16 & keep) | ((s16 << 2) & ~keep);
}
// OR up to 16 symbols (top-aligned in v, ZEROS below the last one) into the stream at symbol position pos -- the count is not needed
template <typename OS>
GRM_HD void stream_insert_top(uint32_t pos, uint32_t v, OS &&or_sym)
{
    const uint32_t w = pos >> 5, o = pos & 31u;
    const uint64_t v64 = (uint64_t)v << 32;
    or_sym(w, v64 >> (2u * o));
    if (o > 16u) {
        const uint64_t lo = v64 << (64u - 2u * o);
        if (lo) or_sym(w + 1, lo);
    }
}
template <typename OI>
GRM_HD void stream_insert_inv(uint32_t pos, uint32_t cnt, uint32_t inv, OI &&or_inv)
{
    if (!inv) return;
    const uint32_t wi = pos >> 6, oi = pos & 63u;
    or_inv(wi, (uint64_t)inv << oi);
    if (oi + cnt > 64u) or_inv(wi + 1, (uint64_t)inv >> (64u - oi));
}
// a clean chunk whose incoming line is a sequence line, packed and inserted: every byte but the newlines is a symbol.  with_inv: some
// chunk of the wave holds a bad base (else the inv words stay as they are: zero)
template <typename OS, typename OI>
GRM_HD void clean_chunk_insert(const uint32_t w[4], const uint32_t z[4], uint32_t nlc, uint32_t pos, bool with_inv, OS &&or_sym, OI &&or_inv)
{
    if (nlc >= 2u) {                                      // blank lines, padding, lines shorter than a chunk: the general packer
        uint32_t cs, ci;
        const int cnt = chunk_pack(w, ~clean_nl_mask16(z) & 0xffffu, 0u, cs, ci);
        stream_insert(pos, cnt, cs, ci, or_sym, or_inv);
        return;
    }
    const uint32_t j = clean_first_nl(z);
    stream_insert_top(pos, close_hole_top(chunk_codes16(w), j), or_sym);
    if (with_inv) {
        uint32_t b16 = bad_of_word(w[0]) | (bad_of_word(w[1]) << 4) | (bad_of_word(w[2]) << 8) | (bad_of_word(w[3]) << 12);
        b16 = (b16 & ((1u << j) - 1u)) | ((b16 >> (j + 1u)) << j);        // (j == 16: all 16 stay)
        stream_insert_inv(pos, 16u - nlc, b16, or_inv);
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

// ---- N-position variant (N = 16 or 32 start positions; N + k - 1 <= 64) ---------------
// valid-start mask for the start positions at offset o (0..63) inside group G: bit i <=>
// position 64G + o + i is valid.  Only bits i with i + k - 1 <= 63 are meaningful.  k in 1..32
GRM_HD uint64_t valid_starts_at(uint64_t i0, uint64_t i1, int o, int k)
{
    uint64_t acc = o ? ((i0 >> o) | (i1 << (64 - o))) : i0;     // inv bits of symbols o .. o+63
    int covered = 1;
    while (covered * 2 <= k) {
        acc |= acc >> covered;
        covered *= 2;
    }
    const int r = k - covered;
    if (r) acc |= acc >> r;
    return ~acc;
}
GRM_HD uint32_t valid_starts32(uint64_t i0, uint64_t i1, int half, int k)
{
    return (uint32_t)valid_starts_at(i0, i1, half * 32, k);
}

// w0 = packed word holding the first start position, at symbol offset `off` (0..31) inside
// it; w1 = the next word.  Calls f(i, canonical) for the valid start positions i in 0..NPOS-1
// (needs off + NPOS + k - 1 <= 64).  Fully unrolled: a caller may keep the k-mers in registers.
template <int NPOS, typename F>
GRM_HD void for_each_kmer_n(uint64_t w0, uint64_t w1, int off, uint32_t valid, int k, F &&f)
{
    if (!valid) return;
    const uint64_t a = off ? ((w0 << (2 * off)) | (w1 >> (64 - 2 * off))) : w0;
    const uint64_t b = off ? (w1 << (2 * off)) : w1;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rcshift = 2 * (k - 1);
    const int m = k - 1;
    uint64_t fwd = m ? (a >> (64 - 2 * m)) : 0;
    uint64_t rc = m ? (revcomp_m(fwd, m) << 2) : 0;
    // stream = the symbols after the first m (m <= 31), MSB-aligned: the next symbol is
    // always the top 2 bits of hi
    uint64_t hi = m ? ((a << (2 * m)) | (b >> (64 - 2 * m))) : a;
    uint64_t lo = m ? (b << (2 * m)) : b;
#if defined(__HIP_DEVICE_COMPILE__)
    // almost every wave sees nothing but valid starts (no contig end, no N nearby): a wave-uniform
    // branch then runs the positions without the per-position exec masking
    if (__all(valid == (NPOS == 32 ? ~0u : (1u << (NPOS & 31)) - 1u))) {
#pragma unroll
        for (int i = 0; i < NPOS; i++) {
            const uint64_t s = hi >> 62;
            hi = (hi << 2) | (lo >> 62);
            lo <<= 2;
            fwd = ((fwd << 2) | s) & mask;
            rc = (rc >> 2) | ((s ^ 2) << rcshift);
            f(i, fwd < rc ? fwd : rc);
        }
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < NPOS; i++) {
        const uint64_t s = hi >> 62;
        hi = (hi << 2) | (lo >> 62);
        lo <<= 2;
        fwd = ((fwd << 2) | s) & mask;
        rc = (rc >> 2) | ((s ^ 2) << rcshift);
        if ((valid >> i) & 1u) f(i, fwd < rc ? fwd : rc);
    }
}
template <typename F>
GRM_HD void for_each_kmer32(uint64_t a, uint64_t b, uint32_t valid, int k, F &&f)
{
    for_each_kmer_n<32>(a, b, 0, valid, k, f);
}

// ---- two-word k-mers (33 <= k <= 64): value = hi:lo, first base most significant -------
struct K128 {
    uint64_t hi, lo;
};
GRM_HD bool k128_less(const K128 &a, const K128 &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

// NPOS start positions beginning at symbol offset `off` (0..31) of w0; needs
// off + NPOS + k - 1 <= 96 (three packed words).  Calls f(i, canonical) for the valid ones.
template <int NPOS, typename F>
GRM_HD void for_each_kmer_wide(uint64_t w0, uint64_t w1, uint64_t w2, int off, uint32_t valid, int k, F &&f)
{
    if (!valid) return;
    // 192-bit stream, MSB-aligned at the first symbol
    uint64_t s0 = w0, s1 = w1, s2 = w2;
    if (off) {
        s0 = (w0 << (2 * off)) | (w1 >> (64 - 2 * off));
        s1 = (w1 << (2 * off)) | (w2 >> (64 - 2 * off));
        s2 = w2 << (2 * off);
    }
    const int top = 2 * (k - 1) - 64;                  // bit of the hi word that receives the complement
    const uint64_t mask_hi = k == 64 ? ~0ull : ((1ull << (2 * k - 64)) - 1);
    K128 fwd = {0, 0}, rc = {0, 0};
    auto step = [&]() {
        const uint64_t sym = s0 >> 62;
        s0 = (s0 << 2) | (s1 >> 62);
        s1 = (s1 << 2) | (s2 >> 62);
        s2 <<= 2;
        fwd.hi = ((fwd.hi << 2) | (fwd.lo >> 62)) & mask_hi;
        fwd.lo = (fwd.lo << 2) | sym;
        rc.lo = (rc.lo >> 2) | (rc.hi << 62);
        rc.hi = (rc.hi >> 2) | ((sym ^ 2) << top);
    };
    {
        // the first m = k - 1 symbols (32 <= m <= 63) at once instead of m rolling steps (620 of the ~800 instructions a
        // thread spent on its 16 positions): forward word = top 2m bits of the stream, reverse-complement word = its
        // 2-bit groups reversed and complemented, << 2 (where m steps of the rolling update leave it)
        const int m = k - 1, sh = 128 - 2 * m;         // sh in 2 .. 64
        K128 r;
        if (sh == 64) {
            fwd.hi = 0; fwd.lo = s0;
            r.hi = 0; r.lo = rev_groups64(s0);         // (groups of lo reversed = the whole 64-bit frame)
        } else {
            fwd.hi = s0 >> sh;
            fwd.lo = (s0 << (64 - sh)) | (s1 >> sh);
            const uint64_t rh = rev_groups64(fwd.lo), rl = rev_groups64(fwd.hi);      // reversed 128-bit frame: symbols left-aligned
            r.hi = rh >> sh;
            r.lo = (rh << (64 - sh)) | (rl >> sh);
        }
        const uint64_t m_hi = m == 32 ? 0ull : ((1ull << (2 * m - 64)) - 1);
        r.hi = (r.hi ^ 0xAAAAAAAAAAAAAAAAull) & m_hi;
        r.lo ^= 0xAAAAAAAAAAAAAAAAull;
        rc.hi = (r.hi << 2) | (r.lo >> 62);
        rc.lo = r.lo << 2;
        const int t = 2 * m - 64;                      // the stream moves on by 2m bits
        s0 = t ? ((s1 << t) | (s2 >> (64 - t))) : s1;
        s1 = t ? (s2 << t) : s2;
        s2 = 0;
    }
#pragma unroll
    for (int i = 0; i < NPOS; i++) {                     // fully unrolled: i is static, callers may index registers with it
        step();
        if ((valid >> i) & 1u) f(i, k128_less(fwd, rc) ? fwd : rc);
    }
}


// ---- run records of two-word k-mers (33 <= k <= 64) -----------------------------------------------------------------
// The minimizer of such a k-mer is taken among the RUNW_W(k) <= 22 m-mers in the MIDDLE of its k - 10: the window minimum,
// the run heads and the 7 + 5 + 1 field bits are then exactly those of one-word k-mers, computed on the stream moved on by
// runw_offset(k) positions.  (The window has the parity of k - 10, so that it is centred: a k-mer and its reverse complement
// look at the same m-mers and land in the same bucket.)  A run of up to 22 k-mers is up to 85 bases: THREE words,
//     r[0], r[1]        bases 0..63
//     r[2][63..22]      bases 64..84          r[2][12..0] as y[12..0] of the 16-byte record (flip mark, fine bucket, length)
// stored strand-canonically as the 16-byte records are (level 1 marks, level 2 turns over).
GRM_HD int runw_window(int k) { return ((k - 10) & 1) ? 21 : 22; }
GRM_HD int runw_offset(int k) { return (k - 10 - runw_window(k)) / 2; }
struct RunW {
    uint64_t r[3];
};
// 192-bit shift to the left by 0 <= sh < 192 bits
GRM_HD void shl192(uint64_t &a, uint64_t &b, uint64_t &c, int sh)
{
    if (sh >= 128) { a = c; b = 0; c = 0; sh -= 128; }
    else if (sh >= 64) { a = b; b = c; c = 0; sh -= 64; }
    if (sh) {
        a = (a << sh) | (b >> (64 - sh));
        b = (b << sh) | (c >> (64 - sh));
        c <<= sh;
    }
}
// the record of the run of `len` k-mers that starts at position i (0..31) of the window whose words are e0..e3
GRM_HD RunW runw_record(uint64_t e0, uint64_t e1, uint64_t e2, uint64_t e3, int i, uint32_t len, int k, bool flip, uint32_t fine)
{
    const int span = (int)len + k - 1;               // 33 .. 85 bases
    uint64_t a = e0, b = e1, c = e2;
    if (i) {
        a = (e0 << (2 * i)) | (e1 >> (64 - 2 * i));
        b = (e1 << (2 * i)) | (e2 >> (64 - 2 * i));
        c = (e2 << (2 * i)) | (e3 >> (64 - 2 * i));
    }
    // the first `span` bases: 2 span bits from the top of the 192
    const uint64_t mb = span >= 64 ? ~0ull : ~0ull << (128 - 2 * span);
    const uint64_t mc = span > 64 ? ~0ull << (192 - 2 * span) : 0ull;
    RunW o;
    o.r[0] = a;
    o.r[1] = b & mb;
    o.r[2] = (c & mc) | (flip ? RUN_FLIP_BIT : 0ull) | ((uint64_t)(fine & ((1u << RUN_FINE_BITS) - 1u)) << RUN_LEN_BITS) | len;
    return o;
}
// a record marked by level 1: reverse complement of its bases, left-aligned again; the mark goes
GRM_HD void runw_flip(RunW &o, int k)
{
    if (!(o.r[2] & RUN_FLIP_BIT)) return;
    const uint64_t low = o.r[2] & (RUN_FLIP_BIT - 1);
    const int span = (int)run_len(o.r[2]) + k - 1;
    // reverse complement of the 96 bases r0 : r1 : r2 = rc(r2) : rc(r1) : rc(r0); the run's bases are its LAST `span` ones
    uint64_t a = revcomp_m(o.r[2] & ~(2 * RUN_FLIP_BIT - 1), 32), b = revcomp_m(o.r[1], 32), c = revcomp_m(o.r[0], 32);
    shl192(a, b, c, 2 * (96 - span));
    o.r[0] = a;
    o.r[1] = b;
    o.r[2] = c | low;
}
// canonical k-mer number t (0 .. 21) of a record
GRM_HD K128 runw_kmer_at(const RunW &o, int k, uint32_t t)
{
    uint64_t a = o.r[0], b = o.r[1], c = o.r[2] & ~(2 * RUN_FLIP_BIT - 1);
    if (t) {
        a = (a << (2 * t)) | (b >> (64 - 2 * t));
        b = (b << (2 * t)) | (c >> (64 - 2 * t));
    }
    // the k-mer = the top 2k bits of a : b, right-aligned in hi : lo
    const int sh = 128 - 2 * k;                      // 0 .. 62
    K128 f, r;
    f.hi = sh ? a >> sh : a;
    f.lo = sh ? (a << (64 - sh)) | (b >> sh) : b;
    // reverse complement: the groups of the left-aligned k-mer reversed = right-aligned, complemented
    // (the 128-bit frame a : b with the k-mer at its top, reversed group by group, has the k-mer's groups at its bottom)
    const uint64_t m_hi = k == 64 ? ~0ull : ((1ull << (2 * k - 64)) - 1);
    r.hi = (rev_groups64(b) ^ 0xAAAAAAAAAAAAAAAAull) & m_hi;
    r.lo = rev_groups64(a) ^ 0xAAAAAAAAAAAAAAAAull;
    return k128_less(r, f) ? r : f;
}
// valid k-mer starts for k up to 64: bit i <=> position o + i (o = 0..63 inside the group of i0) starts k symbols without an invalid
// one; i0, i1, i2: three consecutive words of inv bits; meaningful for i + k - 1 <= 127 - o ... the callers use i <= 32
GRM_HD uint64_t valid_starts_wide(uint64_t i0, uint64_t i1, uint64_t i2, int o, int k)
{
    // inv bits of symbols o .. o + 127 in (lo, hi)
    uint64_t lo = o ? ((i0 >> o) | (i1 << (64 - o))) : i0;
    uint64_t hi = o ? ((i1 >> o) | (i2 << (64 - o))) : i1;
    // OR of the k bits from every position on: doubling
    int covered = 1;
    while (covered * 2 <= k) {
        const uint64_t nlo = covered >= 64 ? hi : (lo >> covered) | (hi << (64 - covered));
        const uint64_t nhi = covered >= 64 ? 0ull : hi >> covered;
        lo |= nlo; hi |= nhi;
        covered *= 2;
    }
    const int r = k - covered;
    if (r) lo |= (lo >> r) | (hi << (64 - r));
    return ~lo;
}

}  // namespace grm
