// grm_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the k-mer-matrix engine.
//
// Replaces the compute inside the reference's absent native tools:
//   DSK / multidsk  (call sites bin/kover/core/kover/dataset/tools/kmer_count.py:28-53,
//                    src/app.py:1372)                     -> parse_*, kmer_hist, kmer_scatter,
//                                                             bucket_dedup
//   dsk2kover       (call site .../tools/kmer_pack.py:28-36) -> dict_build, dict_*, matrix_fill
//
// All work is integer / byte work bounded by HBM bandwidth: no MFMA.  Design notes in
// DESIGN.md ("Kernels").  Wave width is hard-coded to 64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "grm_device_fns.h"
#include "grm_internal.h"
#include "grm_coop.h"

namespace grm {

// ------------------------------------------------------------------------------------
// Stage 0: FASTA -> packed symbol stream
//
// raw   : all file images of the batch, each file starting on a TILE_BYTES boundary,
//         gaps filled with '\n'; raw[-1] exists and is '\n' (front pad), so
//         "byte i starts a line" == (raw[i-1] == '\n') everywhere.
// Lines whose first byte is '>' are headers: the '>' emits ONE separator symbol
// (inv bit set) so that no k-mer spans two records; other header bytes emit nothing.
// Every other line emits each byte except '\n' / '\r' as a symbol:
// code (c>>1)&3, inv bit (c>>3)&1  (SURVEY 8(c)(1),(3)).
// ------------------------------------------------------------------------------------
// Per tile: every thread owns one 16-byte chunk in each of the 4 rounds (all four loads are
// issued up front).  A chunk is summarised by an element (last line-start type, symbols if the
// incoming line is a sequence line, symbols if it is a header line); pelem_combine is
// associative, so ONE block scan in (round, thread) order gives every chunk its incoming line
// type and its symbol offset for both possible tile in-states.
struct TileChunks {
    uint32_t w[ROUNDS_PER_TILE][4];
    uint32_t ek[ROUNDS_PER_TILE], sep[ROUNDS_PER_TILE], unk[ROUNDS_PER_TILE];
    uint32_t pre[ROUNDS_PER_TILE];     // exclusive prefix element (pelem32) of the chunk within the tile
    uint32_t total;                    // element of the whole tile
};

template <int R>
__device__ __forceinline__ void tile_round(const uint4 &v, uint32_t edge, int lane, TileChunks &tc, uint32_t &elem)
{
    tc.w[R][0] = v.x; tc.w[R][1] = v.y; tc.w[R][2] = v.z; tc.w[R][3] = v.w;
    uint32_t nl, gt, cr;
    chunk_masks(tc.w[R], nl, gt, cr);
    const uint32_t prev_nl = edge;                       // (every lane holds the byte before its chunk: TileLoad)
    const uint32_t ls = ((nl << 1) | prev_nl) & 0xffffu;
    uint32_t ek, sep, unk;
    chunk_classify(nl, gt, cr, ls, T_NONE, ek, sep, unk);
    tc.ek[R] = ek; tc.sep[R] = sep; tc.unk[R] = unk;
    elem = pelem32_make(chunk_last_event(ls, gt), __popc(ek) + __popc(unk), __popc(ek));
}

// the same element for parse_summarize, which needs nothing else of the chunk: a wave whose 1 KiB is clean (clean_scan: letters and
// newlines only -- nearly every wave of a FASTA) gets it from the count and the place of its newlines, ~60 instructions per chunk
// against ~150 through the masks and the flood of line types; the kernel is bound by instruction issue
// exclusive prefixes of the tile's 16 (round, wave) group totals for this wave's four groups, and the sum of all 16: every row of 16
// lanes scans the same 16 words with four DPP adds and the wave reads its four out with v_readlane -- the loop over 16 LDS words with
// four selects each that this replaces was ~100 instructions per thread
template <int STRIDE = 1>
__device__ __forceinline__ void group_prefixes(const uint32_t *partial /* LDS [16 * STRIDE] */, int lane, int wave, uint32_t (&wp)[4], uint32_t &total)
{
    const uint32_t pv = partial[(lane & 15) * STRIDE];
    uint32_t s = pv;
    s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, false);
    const uint32_t ex = s - pv;
    const int w = __builtin_amdgcn_readfirstlane(wave);
    wp[0] = (uint32_t)__builtin_amdgcn_readlane((int)ex, w);
    wp[1] = (uint32_t)__builtin_amdgcn_readlane((int)ex, 4 + w);
    wp[2] = (uint32_t)__builtin_amdgcn_readlane((int)ex, 8 + w);
    wp[3] = (uint32_t)__builtin_amdgcn_readlane((int)ex, 12 + w);
    total = (uint32_t)__builtin_amdgcn_readlane((int)s, 15);
}

template <int R>
__device__ __forceinline__ void tile_round_sum(const uint4 &v, uint32_t edge, int lane, uint32_t &elem)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t z[4];
    const uint32_t odd = clean_scan(w, z);
    if (!__any(odd != 0u)) {
        elem = clean_elem(z, edge);
    } else {
        TileChunks unused;
        tile_round<R>(v, edge, lane, unused, elem);
    }
}

// The scan of a FASTA tile's 1024 chunks (order: round, thread), giving every chunk its exclusive prefix element (ev, cs, ch) and the
// tile its total.  The element's combine is associative, but a log-step scan of it costs ~14 instructions and a cross-lane move per
// step and round (138 of parse_summarize's 313 instructions per chunk).  The element has more structure than that:
//   ev   = type of the nearest line start BEFORE the chunk: inside a wave a ballot and a count-leading-zeros, across the 16
//          (round, wave) groups two bits per group in one LDS word;
//   ch   = symbols before the chunk if a header line runs into the tile = sum of known_i = ch_i + (ev_i == SEQ ? extra_i : 0);
//   cs   = the same if a sequence line does = ch + sum of pend_i = (ev_i == NONE ? extra_i : 0), extra = the chunk's bytes before
//          its own first line start;
// i.e. ONE integer prefix sum of the packed pair (known | pend << 16; a tile holds 16384 bytes): six DPP adds per round.
// a thread's four chunks of a tile, and the byte before each (is it a newline?).  The
// parse kernels issue these loads FIRST, before they read what kind of tile it is: a workgroup lives for one tile, and every dependent
// load in front of these is time it holds its registers with nothing in flight
struct TileLoad {
    uint4 v[ROUNDS_PER_TILE];
    uint32_t edge[ROUNDS_PER_TILE];
};
__device__ __forceinline__ void tile_load(const uint8_t *__restrict__ raw, uint32_t tile, TileLoad &tl)
{
#pragma unroll
    for (int r = 0; r < ROUNDS_PER_TILE; r++) {
        const uint64_t base = (uint64_t)tile * TILE_BYTES + (uint64_t)r * ROUND_BYTES + threadIdx.x * 16u;
        tl.v[r] = *reinterpret_cast<const uint4 *>(raw + base);
        // (EVERY lane its own byte, and the byte as it is: a load under `lane == 0` needs a move or a compare behind it, which makes the
        // wave wait for the load -- and, loads returning in order, for the 64 bytes asked for before it -- in front of the next round's
        // request: until round 4's second half the four requests of a tile went out one round trip after the other)
        tl.edge[r] = raw[(int64_t)base - 1];
    }
}

// SUM_ONLY: the caller wants tc.pre[] and tc.total only (parse_summarize), not the chunks' words and masks
template <bool SUM_ONLY = false>
__device__ __forceinline__ void tile_scan(const TileLoad &tl, uint64_t *partial64 /* LDS [16] */, TileChunks &tc)
{
    uint32_t *partial = reinterpret_cast<uint32_t *>(partial64);        // [0..15]: group totals, [16]: last line-start type of every group, 2 bits each
    const int lane = lane_id(), wave = wave_id();
    static_assert(ROUNDS_PER_TILE == 4 && PARSE_THREADS == 256, "tile_scan assumes 4 rounds x 4 waves");
    const uint4 (&v)[ROUNDS_PER_TILE] = tl.v;
    const uint32_t (&edge)[ROUNDS_PER_TILE] = tl.edge;
    if (threadIdx.x == 0) partial[16] = 0;
    uint32_t el[ROUNDS_PER_TILE];
    if (SUM_ONLY) {
        tile_round_sum<0>(v[0], (uint32_t)(edge[0] == '\n'), lane, el[0]);
        tile_round_sum<1>(v[1], (uint32_t)(edge[1] == '\n'), lane, el[1]);
        tile_round_sum<2>(v[2], (uint32_t)(edge[2] == '\n'), lane, el[2]);
        tile_round_sum<3>(v[3], (uint32_t)(edge[3] == '\n'), lane, el[3]);
    } else {
        tile_round<0>(v[0], (uint32_t)(edge[0] == '\n'), lane, tc, el[0]);
        tile_round<1>(v[1], (uint32_t)(edge[1] == '\n'), lane, tc, el[1]);
        tile_round<2>(v[2], (uint32_t)(edge[2] == '\n'), lane, tc, el[2]);
        tile_round<3>(v[3], (uint32_t)(edge[3] == '\n'), lane, tc, el[3]);
    }
    __syncthreads();                                     // (partial[16] zeroed; also orders a caller's LDS writes before its use of them)
    // nearest line start before the chunk inside its group, and the group's last one
    uint32_t t_in[ROUNDS_PER_TILE];                      // 0: no line start before the chunk inside its group
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ROUNDS_PER_TILE; r++) {
        const uint32_t ev = (uint32_t)pelem32_ev(el[r]);
        const unsigned long long any = __ballot(ev != 0), hdr = __ballot(ev == (uint32_t)T_HDR);
        if (hdr == 0ull) {                               // (a scalar branch) no header line starts in these 1 KiB: any line start is a sequence line's
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(any >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)any, 0u));
            t_in[r] = before ? (uint32_t)T_SEQ : 0u;
            if (lane == 0 && any) atomicOr(&partial[16], (uint32_t)T_SEQ << (2 * (r * (PARSE_THREADS / 64) + wave)));
            continue;
        }
        const unsigned long long m = any & below;
        const int src = 63 - (int)__builtin_clzll(m | 1ull);
        t_in[r] = m ? (((hdr >> src) & 1ull) ? (uint32_t)T_HDR : (uint32_t)T_SEQ) : 0u;
        if (lane == 0 && any) {
            const int top = 63 - (int)__builtin_clzll(any);
            atomicOr(&partial[16], (((hdr >> top) & 1ull) ? (uint32_t)T_HDR : (uint32_t)T_SEQ) << (2 * (r * (PARSE_THREADS / 64) + wave)));
        }
    }
    __syncthreads();
    const uint32_t groups = (uint32_t)__builtin_amdgcn_readfirstlane((int)partial[16]);          // (scalar: what follows of it is SALU work)
    uint32_t inc[ROUNDS_PER_TILE], own[ROUNDS_PER_TILE];
#pragma unroll
    for (int r = 0; r < ROUNDS_PER_TILE; r++) {
        const int g = r * (PARSE_THREADS / 64) + __builtin_amdgcn_readfirstlane(wave);
        const uint32_t lower = groups & ((1u << (2 * g)) - 1u);
        const uint32_t t_prev = lower ? (groups >> (2 * ((31 - (int)__builtin_clz(lower)) >> 1))) & 3u : 0u;
        const uint32_t t = t_in[r] ? t_in[r] : t_prev;          // type of the nearest line start before the chunk, 0 = none in this tile
        const uint32_t ch = pelem32_ch(el[r]), extra = pelem32_cs(el[r]) - ch;
        own[r] = (ch + (t == (uint32_t)T_SEQ ? extra : 0u)) | ((t == 0u ? extra : 0u) << 16);
        t_in[r] = t;
        inc[r] = wave_scan_incl_dpp(own[r]);
    }
    if (lane == 63) {
#pragma unroll
        for (int r = 0; r < ROUNDS_PER_TILE; r++) partial[r * (PARSE_THREADS / 64) + wave] = inc[r];
    }
    __syncthreads();
    // prefix of this (round, wave) over the 16 group totals
    uint32_t wp[4], acc;
    group_prefixes(partial, lane, wave, wp, acc);
    auto elem = [](uint32_t type, uint32_t packed) { return pelem32_make((int)type, (packed & 0xffffu) + (packed >> 16), packed & 0xffffu); };
    // the tile's total: its last line start is the last group's with one
    tc.total = elem(groups ? (groups >> (2 * ((31 - (int)__builtin_clz(groups)) >> 1))) & 3u : 0u, acc);
    tc.pre[0] = elem(t_in[0], wp[0] + inc[0] - own[0]);
    tc.pre[1] = elem(t_in[1], wp[1] + inc[1] - own[1]);
    tc.pre[2] = elem(t_in[2], wp[2] + inc[2] - own[2]);
    tc.pre[3] = elem(t_in[3], wp[3] + inc[3] - own[3]);
}

// FASTQ variant of the tile scan: the element carries the newline count mod 4 and the symbol
// count for each of the 4 possible starting line phases (fq_elem_* in grm_device_fns.h).
struct TileChunksFq {
    uint32_t w[ROUNDS_PER_TILE][4];
    uint32_t nl[ROUNDS_PER_TILE], cr[ROUNDS_PER_TILE], ls[ROUNDS_PER_TILE];
    uint64_t pre[ROUNDS_PER_TILE];
    uint64_t total;
};

template <int R>
__device__ __forceinline__ void tile_round_fq(const uint4 &v, uint32_t edge, int lane, TileChunksFq &tc, uint64_t &elem)
{
    tc.w[R][0] = v.x; tc.w[R][1] = v.y; tc.w[R][2] = v.z; tc.w[R][3] = v.w;
    uint32_t nl, gt, cr;
    chunk_masks(tc.w[R], nl, gt, cr, false);
    const uint32_t prev_nl = edge;                       // (every lane holds the byte before its chunk: TileLoad)
    const uint32_t ls = ((nl << 1) | prev_nl) & 0xffffu;
    tc.nl[R] = nl; tc.cr[R] = cr; tc.ls[R] = ls;
    uint32_t m[4];
    fq_phase_masks(nl, m);
    uint32_t c0, c1, c2, c3, em, sp;
    fq_classify(nl, cr, ls, m, 0, em, sp); c0 = __popc(em);
    fq_classify(nl, cr, ls, m, 1, em, sp); c1 = __popc(em);
    fq_classify(nl, cr, ls, m, 2, em, sp); c2 = __popc(em);
    fq_classify(nl, cr, ls, m, 3, em, sp); c3 = __popc(em);
    const uint32_t c[4] = {c0, c1, c2, c3};
    elem = fq_elem_make(__popc(nl) & 3u, c);
}

// FASTQ tiles, the same way: the element (newlines mod 4; symbols for each of the 4 line phases the range may start in) needs no
// associative scan either.  A chunk that starts p newlines into the tile is in phase (s + p) & 3 when the tile starts in phase s, so
// with P = the integer prefix sum of the chunks' newline counts the chunk's four counts, ROTATED by P, are what it adds to the four
// hypotheses -- and those add up as plain integers: one more prefix sum of four packed 16-bit counts (two DPP scans).
__device__ __forceinline__ uint64_t fq_rotate_counts(uint64_t c16, uint32_t p)      // fields of 16 bits: out[s] = in[(s + p) & 3]
{
    const uint32_t sh = 16u * (p & 3u);
    return sh ? (c16 >> sh) | (c16 << (64u - sh)) : c16;
}
__device__ __forceinline__ uint64_t fq_elem_from16(uint32_t nl_mod4, uint64_t c16)    // 16-bit fields -> the element's 15-bit fields
{
    return ((uint64_t)(nl_mod4 & 3u) << 60) | (c16 & 0x7fffull) | (((c16 >> 16) & 0x7fffull) << 15) | (((c16 >> 32) & 0x7fffull) << 30) | (((c16 >> 48) & 0x7fffull) << 45);
}
__device__ __forceinline__ void tile_scan_fq(const TileLoad &tl, uint64_t *partial /* LDS [16] */, TileChunksFq &tc)
{
    const int lane = lane_id(), wave = wave_id();
    const uint4 (&v)[ROUNDS_PER_TILE] = tl.v;
    const uint32_t (&edge)[ROUNDS_PER_TILE] = tl.edge;
    uint64_t el[ROUNDS_PER_TILE];
    tile_round_fq<0>(v[0], (uint32_t)(edge[0] == '\n'), lane, tc, el[0]);
    tile_round_fq<1>(v[1], (uint32_t)(edge[1] == '\n'), lane, tc, el[1]);
    tile_round_fq<2>(v[2], (uint32_t)(edge[2] == '\n'), lane, tc, el[2]);
    tile_round_fq<3>(v[3], (uint32_t)(edge[3] == '\n'), lane, tc, el[3]);
    // newlines before every chunk
    uint32_t nl_own[ROUNDS_PER_TILE], nl_inc[ROUNDS_PER_TILE];
#pragma unroll
    for (int r = 0; r < ROUNDS_PER_TILE; r++) {
        nl_own[r] = (uint32_t)__popc(tc.nl[r]);
        nl_inc[r] = wave_scan_incl_dpp(nl_own[r]);
    }
    uint32_t *p32 = reinterpret_cast<uint32_t *>(partial);
    __syncthreads();                                     // (a caller's LDS writes before this; the words below are free)
    if (lane == 63) {
#pragma unroll
        for (int r = 0; r < ROUNDS_PER_TILE; r++) p32[r * (PARSE_THREADS / 64) + wave] = nl_inc[r];
    }
    __syncthreads();
    uint32_t np[4], nl_total;
    group_prefixes(p32, lane, wave, np, nl_total);
    const uint32_t nl_pre[ROUNDS_PER_TILE] = {np[0] + nl_inc[0] - nl_own[0], np[1] + nl_inc[1] - nl_own[1], np[2] + nl_inc[2] - nl_own[2], np[3] + nl_inc[3] - nl_own[3]};
    // the chunks' counts, rotated to the tile's phases, summed
    uint64_t own[ROUNDS_PER_TILE], inc[ROUNDS_PER_TILE];
#pragma unroll
    for (int r = 0; r < ROUNDS_PER_TILE; r++) {
        const uint64_t c16 = (uint64_t)fq_elem_cnt(el[r], 0) | ((uint64_t)fq_elem_cnt(el[r], 1) << 16) | ((uint64_t)fq_elem_cnt(el[r], 2) << 32) |
                             ((uint64_t)fq_elem_cnt(el[r], 3) << 48);
        own[r] = fq_rotate_counts(c16, nl_pre[r]);
        inc[r] = (uint64_t)wave_scan_incl_dpp((uint32_t)own[r]) | ((uint64_t)wave_scan_incl_dpp((uint32_t)(own[r] >> 32)) << 32);
    }
    __syncthreads();                                     // (everybody has read the newline totals)
    if (lane == 63) {
#pragma unroll
        for (int r = 0; r < ROUNDS_PER_TILE; r++) partial[r * (PARSE_THREADS / 64) + wave] = inc[r];
    }
    __syncthreads();
    // (four 16-bit counts per word, none above 16384 in a whole tile: the two halves add up on their own)
    uint32_t wlo[4], whi[4], tlo, thi;
    group_prefixes<2>(p32, lane, wave, wlo, tlo);
    group_prefixes<2>(p32 + 1, lane, wave, whi, thi);
    const uint64_t acc64 = (uint64_t)tlo | ((uint64_t)thi << 32);
    const uint64_t wp0 = (uint64_t)wlo[0] | ((uint64_t)whi[0] << 32), wp1 = (uint64_t)wlo[1] | ((uint64_t)whi[1] << 32);
    const uint64_t wp2 = (uint64_t)wlo[2] | ((uint64_t)whi[2] << 32), wp3 = (uint64_t)wlo[3] | ((uint64_t)whi[3] << 32);
    tc.total = fq_elem_from16(nl_total, acc64);
    tc.pre[0] = fq_elem_from16(nl_pre[0], wp0 + inc[0] - own[0]);
    tc.pre[1] = fq_elem_from16(nl_pre[1], wp1 + inc[1] - own[1]);
    tc.pre[2] = fq_elem_from16(nl_pre[2], wp2 + inc[2] - own[2]);
    tc.pre[3] = fq_elem_from16(nl_pre[3], wp3 + inc[3] - own[3]);
}

// the FASTQ tile's chunks classified, without the scan (parse_pack: the prefix elements come from parse_summarize's chunk_pre64)
__device__ __forceinline__ void tile_rounds_fq(const TileLoad &tl, TileChunksFq &tc)
{
    const uint4 (&v)[ROUNDS_PER_TILE] = tl.v;
    const uint32_t (&edge)[ROUNDS_PER_TILE] = tl.edge;
#pragma unroll
    for (int r = 0; r < ROUNDS_PER_TILE; r++) {
        tc.w[r][0] = v[r].x; tc.w[r][1] = v[r].y; tc.w[r][2] = v[r].z; tc.w[r][3] = v[r].w;
        uint32_t nl, gt, cr;
        chunk_masks(tc.w[r], nl, gt, cr, false);
        const uint32_t prev_nl = (uint32_t)(edge[r] == '\n');
        tc.nl[r] = nl; tc.cr[r] = cr; tc.ls[r] = ((nl << 1) | prev_nl) & 0xffffu;
    }
}

// P1: per tile -> summary.  FASTA: {v0 = symbols emitted whatever runs into the tile, v1 = extra
// symbols if a sequence line runs into it, tag = type of the last line start}.
// FASTQ: {v[s] = symbols when the tile starts in line phase s, tag = 4 | newlines mod 4}.
__global__ __launch_bounds__(PARSE_THREADS) void parse_summarize_kernel(
    const uint8_t *__restrict__ raw, uint32_t n_tiles, const uint8_t *__restrict__ tile_meta,
    TileSummary *__restrict__ sums, uint32_t *__restrict__ chunk_pre, uint64_t *__restrict__ chunk_pre64)
{
    __shared__ uint64_t partial[ROUNDS_PER_TILE * (PARSE_THREADS / 64)];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    TileLoad tl;
    tile_load(raw, tile, tl);
    TileSummary s;
    if (tile_meta[tile] & TILE_META_FASTQ) {
        TileChunksFq tc;
        tile_scan_fq(tl, partial, tc);
        if (chunk_pre64) {
#pragma unroll
            for (int r = 0; r < ROUNDS_PER_TILE; r++) chunk_pre64[((uint64_t)tile * ROUNDS_PER_TILE + r) * PARSE_THREADS + threadIdx.x] = tc.pre[r];
        }
        s.v[0] = fq_elem_cnt(tc.total, 0); s.v[1] = fq_elem_cnt(tc.total, 1);
        s.v[2] = fq_elem_cnt(tc.total, 2); s.v[3] = fq_elem_cnt(tc.total, 3);
        s.tag = 4u | fq_elem_nl(tc.total);
    } else {
        TileChunks tc;
        tile_scan<true>(tl, partial, tc);
        // the exclusive prefix element of every 16-byte chunk: parse_pack needs exactly these and would otherwise repeat the
        // whole scan (4 bytes per 16 of input, against ~240 of its ~450 instructions per chunk)
        // In 16 bits: the type of the last line start before the chunk and cs, the symbols before it when a sequence line runs into
        // the tile (< 16384).  ch, the count when a header line does, follows: nothing before the tile's first line start counts then,
        // everything after it does, so ch = (a line start before the chunk ? cs - v[1] : 0) with the tile's v[1] (tests/host: emul_parse3).
        if (chunk_pre) {
            uint16_t *pre16 = reinterpret_cast<uint16_t *>(chunk_pre);
#pragma unroll
            for (int r = 0; r < ROUNDS_PER_TILE; r++)
                pre16[((uint64_t)tile * ROUNDS_PER_TILE + r) * PARSE_THREADS + threadIdx.x] = (uint16_t)(((uint32_t)pelem32_ev(tc.pre[r]) << 14) | pelem32_cs(tc.pre[r]));
        }
        s.v[0] = pelem32_ch(tc.total);
        s.v[1] = pelem32_cs(tc.total) - pelem32_ch(tc.total);
        s.v[2] = s.v[3] = 0;
        s.tag = (uint32_t)pelem32_ev(tc.total);
    }
    if (threadIdx.x == 0) sums[tile] = s;
}

// P-scan: ONE workgroup walks all tile summaries.  The parser state carried from tile to tile
// is a value in 0..3 (FASTA: 0 none / 1 sequence line / 2 header line; FASTQ: line index mod 4);
// each tile is a function on it, encoded as a 4-entry table (2 bits per entry).  Function
// composition is associative, so a block scan of the tables gives every thread the state running
// into its first tile; the first tile of every file restarts from state 0.
__device__ __forceinline__ uint32_t tbl_apply(uint32_t t, uint32_t s) { return (t >> (2 * s)) & 3u; }
__device__ __forceinline__ uint32_t tbl_compose(uint32_t f, uint32_t g)     // "f then g"
{
    return tbl_apply(g, tbl_apply(f, 0)) | (tbl_apply(g, tbl_apply(f, 1)) << 2) | (tbl_apply(g, tbl_apply(f, 2)) << 4) |
           (tbl_apply(g, tbl_apply(f, 3)) << 6);
}
__device__ __forceinline__ uint32_t tile_table(const TileSummary &s, uint8_t meta)
{
    uint32_t t;
    if (s.tag & 4u) {
        const uint32_t n = s.tag & 3u;
        t = (n & 3u) | (((1 + n) & 3u) << 2) | (((2 + n) & 3u) << 4) | (((3 + n) & 3u) << 6);
    } else {
        t = s.tag ? (s.tag * 0x55u) : 0xE4u;          // constant last_event, or identity {0,1,2,3}
    }
    if (meta & TILE_META_FIRST) t = tbl_apply(t, 0) * 0x55u;   // restart from state 0: constant function
    return t;
}
__device__ __forceinline__ uint64_t tile_count(const TileSummary &s, uint32_t state)
{
    if (s.tag & 4u) return s.v[state];
    return (uint64_t)s.v[0] + (state != (uint32_t)T_HDR ? s.v[1] : 0u);
}

// The scan element of a run of tiles: its state table plus, for each incoming state, the
// symbols it emits.  combine(a, b) = "a then b"; associative.
template <typename C>
struct ScanElem {
    uint32_t tbl;
    C c[4];
};
template <typename C>
__device__ __forceinline__ ScanElem<C> selem_identity()
{
    ScanElem<C> e;
    e.tbl = 0xE4u;
    e.c[0] = e.c[1] = e.c[2] = e.c[3] = 0;
    return e;
}
template <typename C>
__device__ __forceinline__ ScanElem<C> selem_combine(const ScanElem<C> &a, const ScanElem<C> &b)
{
    ScanElem<C> r;
    r.tbl = tbl_compose(a.tbl, b.tbl);
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint32_t mid = tbl_apply(a.tbl, s);
        r.c[s] = a.c[s] + (mid == 0 ? b.c[0] : mid == 1 ? b.c[1] : mid == 2 ? b.c[2] : b.c[3]);
    }
    return r;
}
template <typename C>
__device__ __forceinline__ ScanElem<C> selem_shfl_up(const ScanElem<C> &e, int d)
{
    ScanElem<C> r;
    r.tbl = __shfl_up(e.tbl, d);
#pragma unroll
    for (int s = 0; s < 4; s++) r.c[s] = __shfl_up(e.c[s], d);
    return r;
}
template <typename C>
__device__ __forceinline__ ScanElem<C> selem_shfl_down(const ScanElem<C> &e, int d)
{
    ScanElem<C> r;
    r.tbl = __shfl_down(e.tbl, d);
#pragma unroll
    for (int s = 0; s < 4; s++) r.c[s] = __shfl_down(e.c[s], d);
    return r;
}
// block-wide scan of elements in thread order.  Returns the exclusive prefix of this thread;
// *total = combination of all.  lds: >= 16 elements.  Two barriers.
template <typename C>
__device__ __forceinline__ ScanElem<C> selem_block_scan(ScanElem<C> v, ScanElem<C> *lds, ScanElem<C> *total)
{
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
    ScanElem<C> inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const ScanElem<C> o = selem_shfl_up(inc, d);
        if (lane >= d) inc = selem_combine(o, inc);
    }
    ScanElem<C> exc = selem_shfl_up(inc, 1);
    if (lane == 0) exc = selem_identity<C>();
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    ScanElem<C> prefix = selem_identity<C>(), all = selem_identity<C>();
    for (int w = 0; w < nw; w++) {
        const ScanElem<C> t = lds[w];
        if (w < wave) prefix = selem_combine(prefix, t);
        all = selem_combine(all, t);
    }
    __syncthreads();
    *total = all;
    return selem_combine(prefix, exc);
}
__device__ __forceinline__ ScanElem<uint32_t> tile_elem(const TileSummary &s, uint8_t meta)
{
    ScanElem<uint32_t> e;
    e.tbl = tile_table(s, meta);
#pragma unroll
    for (int st = 0; st < 4; st++) e.c[st] = (uint32_t)tile_count(s, (meta & TILE_META_FIRST) ? 0u : (uint32_t)st);
    return e;
}

// scan phase A: one tile per thread; exclusive prefix inside the 256-tile block + block totals
__global__ __launch_bounds__(256) void parse_scan_a_kernel(const TileSummary *__restrict__ sums, uint32_t n_tiles,
                                                           const uint8_t *__restrict__ tile_meta,
                                                           ScanElem<uint32_t> *__restrict__ tile_pre,
                                                           ScanElem<uint32_t> *__restrict__ block_tot)
{
    __shared__ ScanElem<uint32_t> lds[16];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    ScanElem<uint32_t> e = selem_identity<uint32_t>();
    if (t < n_tiles) e = tile_elem(sums[t], tile_meta[t]);
    ScanElem<uint32_t> total;
    const ScanElem<uint32_t> pre = selem_block_scan(e, lds, &total);
    if (t < n_tiles) tile_pre[t] = pre;
    if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}
// scan phase B: ONE workgroup over the block totals; the buffer starts in state 0, so only the
// state and symbol offset running into each block are kept.
__global__ __launch_bounds__(1024) void parse_scan_b_kernel(const ScanElem<uint32_t> *__restrict__ block_tot, uint32_t n_blocks,
                                                            uint8_t *__restrict__ block_state, uint64_t *__restrict__ block_off,
                                                            uint64_t *__restrict__ total_out)
{
    __shared__ ScanElem<uint64_t> lds[16];
    const uint32_t per = (n_blocks + blockDim.x - 1) / blockDim.x;
    const uint32_t b0 = min((uint64_t)threadIdx.x * per, (uint64_t)n_blocks), b1 = min((uint64_t)b0 + per, (uint64_t)n_blocks);
    ScanElem<uint64_t> acc = selem_identity<uint64_t>();
    for (uint32_t b = b0; b < b1; b++) {
        const ScanElem<uint32_t> t = block_tot[b];
        ScanElem<uint64_t> w;
        w.tbl = t.tbl;
        w.c[0] = t.c[0]; w.c[1] = t.c[1]; w.c[2] = t.c[2]; w.c[3] = t.c[3];
        acc = selem_combine(acc, w);
    }
    ScanElem<uint64_t> total;
    const ScanElem<uint64_t> pre = selem_block_scan(acc, lds, &total);
    uint32_t st = tbl_apply(pre.tbl, 0);
    uint64_t off = pre.c[0];
    for (uint32_t b = b0; b < b1; b++) {
        const ScanElem<uint32_t> t = block_tot[b];
        block_state[b] = (uint8_t)st;
        block_off[b] = off;
        off += st == 0 ? t.c[0] : st == 1 ? t.c[1] : st == 2 ? t.c[2] : t.c[3];
        st = tbl_apply(t.tbl, st);
    }
    if (threadIdx.x == 0) *total_out = total.c[0];
}
// scan phase C: every tile gets its incoming state and first symbol index
__global__ __launch_bounds__(256) void parse_scan_c_kernel(const ScanElem<uint32_t> *__restrict__ tile_pre, uint32_t n_tiles,
                                                           const uint8_t *__restrict__ tile_meta,
                                                           const uint8_t *__restrict__ block_state,
                                                           const uint64_t *__restrict__ block_off, const uint64_t *__restrict__ total,
                                                           uint64_t *__restrict__ tile_off, uint8_t *__restrict__ tile_state)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < n_tiles) {
        const ScanElem<uint32_t> pre = tile_pre[t];
        const uint32_t bs = block_state[blockIdx.x];
        const uint32_t st = tbl_apply(pre.tbl, bs);
        tile_off[t] = block_off[blockIdx.x] + (bs == 0 ? pre.c[0] : bs == 1 ? pre.c[1] : bs == 2 ? pre.c[2] : pre.c[3]);
        tile_state[t] = (uint8_t)((tile_meta[t] & TILE_META_FIRST) ? 0u : st);
    }
    if (t == 0) tile_off[n_tiles] = *total;
}
__global__ void genome_offsets_kernel(const uint64_t *__restrict__ tile_off, const uint32_t *__restrict__ genome_tile_off,
                                      uint32_t n_genomes, uint64_t *__restrict__ genome_sym_off)
{
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g <= n_genomes; g += gridDim.x * blockDim.x)
        genome_sym_off[g] = tile_off[genome_tile_off[g]];
}

// The packed stream is written group by group (64 symbols) by the tile that holds it: plain stores, except for the
// groups a tile boundary falls into, which two (or more) tiles OR their parts into.  Only those groups -- one per tile
// boundary -- and a few groups past the end of the stream (read by the last k-mer windows) must start out as zero.
__global__ void parse_prezero_kernel(const uint64_t *__restrict__ tile_off, uint32_t n_tiles, uint64_t *__restrict__ sym2,
                                     uint64_t *__restrict__ inv)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t <= n_tiles; t += gridDim.x * blockDim.x) {
        const uint64_t G = tile_off[t] >> 6;
        const int extra = t == n_tiles ? 4 : 1;
        for (int e = 0; e < extra; e++) {
            sym2[2 * (G + e)] = 0;
            sym2[2 * (G + e) + 1] = 0;
            inv[G + e] = 0;
        }
    }
}

// P2: same tile scan, now with the tile's incoming line type and first symbol index known.
// Every chunk packs its symbols into two small bit strings and ORs them into the tile's
// LDS image of the packed stream (ds_or_b64); the image is then stored with coalesced writes.
// Groups shared with a neighbouring tile go out through global atomicOr (buffers pre-zeroed).
// (One tile per workgroup, on purpose: a STREAMING form -- as many workgroups as the device holds, each asking for its next tile's bytes,
// chunk prefixes and totals while it packs the one it holds; two register sets taking turns, unconditional requests, LDS-only barriers;
// 92 VGPRs, counted waits as intended in the ISA -- was built and measured on the same box: 1.89-1.94 ms against 1.68-1.69 for this
// kernel.  Four waves per SIMD that each go through load / pack / store in turn use the vector unit worse than seven that are at
// different points of it.)
__global__ __launch_bounds__(PARSE_THREADS, 6) void parse_pack_kernel(
    const uint8_t *__restrict__ raw, uint32_t n_tiles, const uint8_t *__restrict__ tile_meta,
    const uint64_t *__restrict__ tile_off, const uint8_t *__restrict__ tile_state, uint64_t *__restrict__ sym2,
    uint64_t *__restrict__ inv, const TileSummary *__restrict__ sums, const uint32_t *__restrict__ chunk_pre, const uint64_t *__restrict__ chunk_pre64)
{
    constexpr int MAX_GROUPS = TILE_BYTES / 64 + 2;
    __shared__ uint64_t partial[ROUNDS_PER_TILE * (PARSE_THREADS / 64)];
    __shared__ uint64_t w2[2 * MAX_GROUPS];
    __shared__ uint64_t wi[MAX_GROUPS];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    TileLoad tl;
    tile_load(raw, tile, tl);
    for (int i = threadIdx.x; i < 2 * MAX_GROUPS; i += PARSE_THREADS) w2[i] = 0;
    for (int i = threadIdx.x; i < MAX_GROUPS; i += PARSE_THREADS) wi[i] = 0;
    const uint64_t sym_base = tile_off[tile];
    const uint32_t lead = (uint32_t)(sym_base & 63ull);
    const int state = (int)tile_state[tile];
    uint32_t n_tile;
    auto or_sym = [&](uint32_t i, uint64_t val) { atomicOr((unsigned long long *)&w2[i], (unsigned long long)val); };
    auto or_inv = [&](uint32_t i, uint64_t val) { atomicOr((unsigned long long *)&wi[i], (unsigned long long)val); };
    if (tile_meta[tile] & TILE_META_FASTQ) {
        TileChunksFq tc;
        if (chunk_pre64) {
            tile_rounds_fq(tl, tc);
#pragma unroll
            for (int r = 0; r < ROUNDS_PER_TILE; r++) tc.pre[r] = chunk_pre64[((uint64_t)tile * ROUNDS_PER_TILE + r) * PARSE_THREADS + threadIdx.x];
            n_tile = sums[tile].v[state];
            __syncthreads();                              // (orders the zeroing above, as the scan's barriers do)
        } else {
            tile_scan_fq(tl, partial, tc);                  // its barriers also order the zeroing above
            n_tile = fq_elem_cnt(tc.total, state);
        }
#pragma unroll
        for (int r = 0; r < ROUNDS_PER_TILE; r++) {
            uint32_t m[4], emit, sep, cs, ci;
            fq_phase_masks(tc.nl[r], m);
            fq_classify(tc.nl[r], tc.cr[r], tc.ls[r], m, (state + (int)fq_elem_nl(tc.pre[r])) & 3, emit, sep);
            const int cnt = chunk_pack(tc.w[r], emit, sep, cs, ci);
            stream_insert(lead + fq_elem_cnt(tc.pre[r], state), cnt, cs, ci, or_sym, or_inv);
        }
    } else {
        TileChunks tc;
        const int st = state == T_NONE ? T_SEQ : state;   // only before the first line start of a file
        if (chunk_pre) {
            // the scan of this tile was done by parse_summarize: its per-chunk prefixes and the tile's totals are read back.  A wave
            // whose 1 KiB is clean and inside sequence lines -- nearly every one -- packs its chunks without masks or line types
            // (clean_chunk_insert: ~110 instructions per chunk against ~240)
            const uint4 (&v)[ROUNDS_PER_TILE] = tl.v;
            const uint32_t (&edge)[ROUNDS_PER_TILE] = tl.edge;
            uint32_t pre[ROUNDS_PER_TILE];
#pragma unroll
            for (int r = 0; r < ROUNDS_PER_TILE; r++) {
                pre[r] = reinterpret_cast<const uint16_t *>(chunk_pre)[((uint64_t)tile * ROUNDS_PER_TILE + r) * PARSE_THREADS + threadIdx.x];
            }
            const TileSummary ts = sums[tile];
            n_tile = st == T_SEQ ? ts.v[0] + ts.v[1] : ts.v[0];
            __syncthreads();                              // (orders the zeroing above, as the scan's barrier does)
#pragma unroll
            for (int r = 0; r < ROUNDS_PER_TILE; r++) {
                const uint32_t w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
                uint32_t z[4];
                const uint32_t odd = clean_scan(w, z);
                const int ev = (int)(pre[r] >> 14);
                const int cin = ev ? ev : st;
                const uint32_t cs_before = pre[r] & 0x3fffu;
                const uint32_t pos = lead + (st == T_SEQ ? cs_before : (ev ? cs_before - ts.v[1] : 0u));
                if (!__any(odd != 0u || cin != T_SEQ)) {
                    const uint32_t nlc = clean_nl_count(z);
                    const bool with_inv = __any(clean_bad_any(w, nlc) != 0u);
                    clean_chunk_insert(w, z, nlc, pos, with_inv, or_sym, or_inv);
                } else {
                    uint32_t nl, gt, cr, ek, sep, unk, cs, ci;
                    chunk_masks(w, nl, gt, cr);
                    const uint32_t ls = ((nl << 1) | (uint32_t)(edge[r] == '\n')) & 0xffffu;
                    chunk_classify(nl, gt, cr, ls, T_NONE, ek, sep, unk);
                    const int cnt = chunk_pack(w, ek | (cin == T_SEQ ? unk : 0u), sep, cs, ci);
                    stream_insert(pos, cnt, cs, ci, or_sym, or_inv);
                }
            }
        } else {
            tile_scan(tl, partial, tc);
            n_tile = st == T_SEQ ? pelem32_cs(tc.total) : pelem32_ch(tc.total);
#pragma unroll
            for (int r = 0; r < ROUNDS_PER_TILE; r++) {
                const int ev = pelem32_ev(tc.pre[r]);
                const int cin = ev ? ev : st;
                const uint32_t emit = tc.ek[r] | (cin == T_SEQ ? tc.unk[r] : 0u);
                uint32_t cs, ci;
                const int cnt = chunk_pack(tc.w[r], emit, tc.sep[r], cs, ci);
                stream_insert(lead + (st == T_SEQ ? pelem32_cs(tc.pre[r]) : pelem32_ch(tc.pre[r])), cnt, cs, ci, or_sym, or_inv);
            }
        }
    }
    __syncthreads();
    const uint32_t span = lead + n_tile;
    const uint32_t n_groups = (span + 63) >> 6;
    const uint64_t g_base = sym_base >> 6;
    for (uint32_t g = threadIdx.x; g < n_groups; g += PARSE_THREADS) {
        const uint64_t a0 = w2[2 * g], a1 = w2[2 * g + 1], bi = wi[g];
        const uint64_t G = g_base + g;
        const bool full = (g > 0 || lead == 0) && ((g + 1) * 64 <= span);
        if (full) {
            *reinterpret_cast<ulonglong2 *>(&sym2[2 * G]) = make_ulonglong2(a0, a1);
            inv[G] = bi;
        } else {
            if (a0) atomicOr((unsigned long long *)&sym2[2 * G], (unsigned long long)a0);
            if (a1) atomicOr((unsigned long long *)&sym2[2 * G + 1], (unsigned long long)a1);
            if (bi) atomicOr((unsigned long long *)&inv[G], (unsigned long long)bi);
        }
    }
}

// ---- single-pass parse (option "parse_fused"; NOT the default: measured slower, see the end of this comment) ----------
// parse_summarize + the scan over tile summaries + parse_pack read the 5.6 GB of a 1000-genome batch twice (and 1.4 GB of
// per-chunk scan prefixes on top).  Here a tile is read ONCE: classified and scanned in registers as before, then the tile
// learns the parser state and the symbol offset running into it by a decoupled look-back over the tiles before it, and
// packs what it still holds in registers.  Tile numbers come from a ticket counter, so every tile a workgroup waits for
// has been started by a workgroup that is resident or done: the wait always ends.
//   desc[t]   one 64-bit word per tile, written with ONE agent-scope atomic store each time (no flag beside the data):
//             status (bits 63..62) 0 = nothing yet, 1 = AGGREGATE, 2 = PREFIX
//             AGGREGATE  what the tile does to the state and how many symbols it emits for every incoming state --
//                        FASTA: v0 | v1 << 15 | last-line-type << 30 (TileSummary); FASTQ: four 15-bit counts | newlines mod 4 << 60
//             PREFIX     state running OUT of the tile << 60 | first symbol index AFTER the tile
// The groups (64 symbols) a tile boundary falls into cannot be stored by either tile alone: both leave their part in
// pieces[tile][head / tail] and parse_stitch, one thread per tile, puts every such group together with plain stores --
// no atomics on the stream, nothing to zero beforehand.
// Measured (1000 x 5 Mbp, 341 000 tiles): 7.1 ms against 2.4 + 2.6 for the two kernels it replaces.  By ablation: load +
// classify + scan alone 4.1 ms here (2.4 in parse_summarize: that kernel needs 38 VGPRs, this one 101 -- the scan is a chain of
// cross-lane moves that lives on occupancy), + packing and stores 5.3, + the look-back 7.1-8.3.  The two-pass form is bound by
// instruction issue and latency, not by the 5.6 GB it reads twice, so reading once buys nothing and the look-back costs.
constexpr uint64_t PD_AGG = 1ull << 62, PD_PREFIX = 2ull << 62;
__device__ __forceinline__ uint64_t pd_pack(const TileSummary &s)
{
    if (s.tag & 4u) return (uint64_t)s.v[0] | ((uint64_t)s.v[1] << 15) | ((uint64_t)s.v[2] << 30) | ((uint64_t)s.v[3] << 45) | ((uint64_t)(s.tag & 3u) << 60);
    return (uint64_t)s.v[0] | ((uint64_t)s.v[1] << 15) | ((uint64_t)s.tag << 30);
}
__device__ __forceinline__ TileSummary pd_unpack(uint64_t d, uint8_t meta)
{
    TileSummary s;
    if (meta & TILE_META_FASTQ) {
        s.v[0] = (uint32_t)d & 0x7fffu; s.v[1] = (uint32_t)(d >> 15) & 0x7fffu; s.v[2] = (uint32_t)(d >> 30) & 0x7fffu; s.v[3] = (uint32_t)(d >> 45) & 0x7fffu;
        s.tag = 4u | ((uint32_t)(d >> 60) & 3u);
    } else {
        s.v[0] = (uint32_t)d & 0x7fffu; s.v[1] = (uint32_t)(d >> 15) & 0x7fffu; s.v[2] = s.v[3] = 0;
        s.tag = (uint32_t)(d >> 30) & 3u;
    }
    return s;
}
// the whole workgroup: state and symbol offset running into `tile` (> 0).  Thread j of a round looks at tile (tile - 1 - base - j): wave
// by wave the aggregates in front of the nearest known prefix are folded (older tiles first), then the waves' results in turn.
// A look-back must cover tiles faster than they are started, or every tile ends up walking over all tiles in flight: one every ~10 ns
// at 1000 x 5 Mbp against ~2 us per round of descriptor loads -- 64 descriptors per round (one wave, the others waiting at the
// barrier) measured 7.5 ms for the kernel, 256 per round (all of the workgroup) what the table in DESIGN.md says.
struct LookbackShare {
    uint32_t tbl[PARSE_THREADS / 64], c[PARSE_THREADS / 64][4], p[PARSE_THREADS / 64], d_lo[PARSE_THREADS / 64], d_hi[PARSE_THREADS / 64];
};
__device__ __forceinline__ void parse_lookback(const uint64_t *__restrict__ desc, const uint8_t *__restrict__ tile_meta, uint32_t tile, LookbackShare &sh,
                                               uint32_t &state_in, uint64_t &off_in)
{
    const int lane = lane_id(), wave = wave_id();
    constexpr int NW = PARSE_THREADS / 64;
    ScanElem<uint64_t> acc = selem_identity<uint64_t>();          // the tiles between the round looked at and `tile`, as one function
    for (int64_t idx = (int64_t)tile - 1;; idx -= PARSE_THREADS) {
        const int64_t mine = idx - (int64_t)threadIdx.x;
        uint64_t d = PD_PREFIX;                        // before the first tile: state 0, offset 0
        if (mine >= 0) {
            do {
                d = __hip_atomic_load(&desc[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((d >> 62) == 0);
        }
        const unsigned long long is_p = __ballot((d >> 62) == 2);
        const int p = is_p ? (int)__builtin_ctzll(is_p) : 64;        // the nearest tile of this wave's 64 whose prefix is known
        ScanElem<uint32_t> x = selem_identity<uint32_t>();
        if (lane < p) x = tile_elem(pd_unpack(d, tile_meta[mine]), tile_meta[mine]);
        // lane 0 <- (tile of lane p - 1) then ... then (tile of lane 0): older tiles first
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const ScanElem<uint32_t> y = selem_shfl_down(x, dd);
            if (lane + dd < 64) x = selem_combine(y, x);
        }
        const uint32_t p_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)d, p & 63), p_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(d >> 32), p & 63);
        if (lane == 0) {
            sh.tbl[wave] = x.tbl;
#pragma unroll
            for (int q = 0; q < 4; q++) sh.c[wave][q] = x.c[q];
            sh.p[wave] = (uint32_t)p;
            sh.d_lo[wave] = p_lo;
            sh.d_hi[wave] = p_hi;
        }
        __syncthreads();
        bool found = false;
#pragma unroll
        for (int w = 0; w < NW && !found; w++) {
            ScanElem<uint64_t> e;
            e.tbl = sh.tbl[w];
#pragma unroll
            for (int q = 0; q < 4; q++) e.c[q] = sh.c[w][q];
            acc = selem_combine(e, acc);
            if (sh.p[w] < 64u) {
                const uint64_t dp = ((uint64_t)sh.d_hi[w] << 32) | sh.d_lo[w];
                const uint32_t sp = (uint32_t)(dp >> 60) & 3u;
                state_in = tbl_apply(acc.tbl, sp);
                off_in = (dp & ((1ull << 60) - 1)) + (sp == 0 ? acc.c[0] : sp == 1 ? acc.c[1] : sp == 2 ? acc.c[2] : acc.c[3]);
                found = true;
            }
        }
        __syncthreads();                                // (the shared words are written again in the next round)
        if (found) return;
    }
}

__global__ __launch_bounds__(PARSE_THREADS) void parse_fused_kernel(
    const uint8_t *__restrict__ raw, uint32_t n_tiles, const uint8_t *__restrict__ tile_meta, uint64_t *__restrict__ desc,
    uint32_t *__restrict__ ticket, uint64_t *__restrict__ tile_off, uint64_t *__restrict__ sym2, uint64_t *__restrict__ inv,
    uint64_t *__restrict__ pieces, int ablate)
{
    constexpr int MAX_GROUPS = TILE_BYTES / 64 + 2;
    __shared__ uint64_t partial[ROUNDS_PER_TILE * (PARSE_THREADS / 64)];
    __shared__ uint64_t w2[2 * MAX_GROUPS];
    __shared__ uint64_t wi[MAX_GROUPS];
    __shared__ uint32_t s_tile;
    __shared__ LookbackShare s_look;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = threadIdx.x; i < 2 * MAX_GROUPS; i += PARSE_THREADS) w2[i] = 0;
    for (int i = threadIdx.x; i < MAX_GROUPS; i += PARSE_THREADS) wi[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= n_tiles) return;
    const uint8_t meta = tile_meta[tile];
    const bool fastq = (meta & TILE_META_FASTQ) != 0;
    TileChunks tc;
    TileChunksFq tq;
    TileSummary sum;
    TileLoad tl;
    tile_load(raw, tile, tl);
    if (fastq) {
        tile_scan_fq(tl, partial, tq);
        sum.v[0] = fq_elem_cnt(tq.total, 0); sum.v[1] = fq_elem_cnt(tq.total, 1);
        sum.v[2] = fq_elem_cnt(tq.total, 2); sum.v[3] = fq_elem_cnt(tq.total, 3);
        sum.tag = 4u | fq_elem_nl(tq.total);
    } else {
        tile_scan(tl, partial, tc);
        sum.v[0] = pelem32_ch(tc.total);
        sum.v[1] = pelem32_cs(tc.total) - pelem32_ch(tc.total);
        sum.v[2] = sum.v[3] = 0;
        sum.tag = (uint32_t)pelem32_ev(tc.total);
    }
    uint32_t st_in = 0;
    uint64_t off_in = 0;
    if (ablate) {                 // (timing experiments only: wrong offsets)
        st_in = 1;
        off_in = (uint64_t)tile * 16000;
        if (ablate == 2) return;
    } else
    if (tile) {
        if (threadIdx.x == 0) __hip_atomic_store(&desc[tile], PD_AGG | pd_pack(sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        parse_lookback(desc, tile_meta, tile, s_look, st_in, off_in);          // (every thread ends with the same two values)
    }
    if (threadIdx.x == 0) {
        const ScanElem<uint32_t> e = tile_elem(sum, meta);
        const uint32_t n_mine = st_in == 0 ? e.c[0] : st_in == 1 ? e.c[1] : st_in == 2 ? e.c[2] : e.c[3];
        __hip_atomic_store(&desc[tile], PD_PREFIX | ((uint64_t)tbl_apply(e.tbl, st_in) << 60) | (off_in + n_mine), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        tile_off[tile] = off_in;
        if (tile == n_tiles - 1) tile_off[n_tiles] = off_in + n_mine;
    }
    const uint64_t sym_base = off_in;
    const uint32_t lead = (uint32_t)(sym_base & 63ull);
    const int state = (meta & TILE_META_FIRST) ? 0 : (int)st_in;
    uint32_t n_tile;
    auto or_sym = [&](uint32_t i, uint64_t val) { atomicOr((unsigned long long *)&w2[i], (unsigned long long)val); };
    auto or_inv = [&](uint32_t i, uint64_t val) { atomicOr((unsigned long long *)&wi[i], (unsigned long long)val); };
    if (fastq) {
        n_tile = fq_elem_cnt(tq.total, state);
#pragma unroll
        for (int r = 0; r < ROUNDS_PER_TILE; r++) {
            uint32_t m[4], emit, sep, cs, ci;
            fq_phase_masks(tq.nl[r], m);
            fq_classify(tq.nl[r], tq.cr[r], tq.ls[r], m, (state + (int)fq_elem_nl(tq.pre[r])) & 3, emit, sep);
            const int cnt = chunk_pack(tq.w[r], emit, sep, cs, ci);
            stream_insert(lead + fq_elem_cnt(tq.pre[r], state), cnt, cs, ci, or_sym, or_inv);
        }
    } else {
        const int st = state == T_NONE ? T_SEQ : state;   // only before the first line start of a file
        n_tile = st == T_SEQ ? pelem32_cs(tc.total) : pelem32_ch(tc.total);
#pragma unroll
        for (int r = 0; r < ROUNDS_PER_TILE; r++) {
            const int ev = pelem32_ev(tc.pre[r]);
            const int cin = ev ? ev : st;
            const uint32_t emit = tc.ek[r] | (cin == T_SEQ ? tc.unk[r] : 0u);
            uint32_t cs, ci;
            const int cnt = chunk_pack(tc.w[r], emit, tc.sep[r], cs, ci);
            stream_insert(lead + (st == T_SEQ ? pelem32_cs(tc.pre[r]) : pelem32_ch(tc.pre[r])), cnt, cs, ci, or_sym, or_inv);
        }
    }
    __syncthreads();
    const uint32_t span = lead + n_tile;
    const uint32_t n_groups = (span + 63) >> 6;
    const uint64_t g_base = sym_base >> 6;
    for (uint32_t g = threadIdx.x; g < n_groups; g += PARSE_THREADS) {
        const uint64_t a0 = w2[2 * g], a1 = w2[2 * g + 1], bi = wi[g];
        const uint64_t G = g_base + g;
        const bool full = (g > 0 || lead == 0) && ((g + 1) * 64 <= span);
        if (full) {
            *reinterpret_cast<ulonglong2 *>(&sym2[2 * G]) = make_ulonglong2(a0, a1);
            inv[G] = bi;
        } else {
            // head piece: the group the tile starts inside (its first symbols belong to a tile before it); tail piece: a group the tile
            // opens but does not fill
            uint64_t *pc = pieces + ((uint64_t)tile * 2 + ((g == 0 && lead != 0) ? 0 : 1)) * 3;
            pc[0] = a0; pc[1] = a1; pc[2] = bi;
        }
    }
}

// every group that holds symbols of more than one tile (or ends the stream): the tile with its first symbol ORs the pieces together
__global__ void parse_stitch_kernel(const uint64_t *__restrict__ tile_off, uint32_t n_tiles, const uint64_t *__restrict__ pieces,
                                    uint64_t *__restrict__ sym2, uint64_t *__restrict__ inv)
{
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t <= n_tiles; t += gridDim.x * blockDim.x) {
        if (t == n_tiles) {
            // a few groups behind the end of the stream are read by the last k-mer windows: zero
            const uint64_t G = (tile_off[n_tiles] + 63) >> 6;
            for (int e = 0; e < 4; e++) { sym2[2 * (G + e)] = 0; sym2[2 * (G + e) + 1] = 0; inv[G + e] = 0; }
            continue;
        }
        const uint64_t off = tile_off[t], end = tile_off[t + 1];
        if (end == off || !(end & 63ull)) continue;
        const uint64_t g1 = (end - 1) >> 6;
        if ((g1 << 6) < off) continue;                  // the group began in an earlier tile: that one's
        const uint64_t *pc = pieces + ((uint64_t)t * 2 + 1) * 3;
        uint64_t a0 = pc[0], a1 = pc[1], bi = pc[2];
        const uint64_t g_end = (g1 + 1) << 6;
        for (uint32_t u = t + 1; u < n_tiles && tile_off[u] < g_end; u++) {
            const uint64_t eu = tile_off[u + 1];
            if (eu > tile_off[u]) {
                const uint64_t *pu = pieces + (uint64_t)u * 2 * 3;
                a0 |= pu[0]; a1 |= pu[1]; bi |= pu[2];
            }
            if (eu >= g_end) break;
        }
        sym2[2 * g1] = a0;
        sym2[2 * g1 + 1] = a1;
        inv[g1] = bi;
    }
}

// ------------------------------------------------------------------------------------
// Stage 1: canonical k-mers -> radix partition by hash bucket, per genome
// ------------------------------------------------------------------------------------
struct KmerArgs {
    const uint64_t *sym2;
    const uint64_t *inv;
    uint64_t total_syms;
    uint64_t n_groups;
    const uint64_t *genome_sym_off;
    uint32_t n_genomes;
    int k;
    int bb;                     // log2(buckets per genome)
    uint32_t groups_per_thread;
};

template <typename F>
__device__ __forceinline__ void group_kmers(const KmerArgs &a, uint64_t grp, F &&f)
{
    const uint64_t p0 = grp << 6;
    const int64_t nv = (int64_t)a.total_syms - a.k + 1 - (int64_t)p0;
    if (nv <= 0) return;
    const uint64_t A = a.sym2[2 * grp], B = a.sym2[2 * grp + 1], C = a.sym2[2 * grp + 2];
    uint64_t valid = valid_starts(a.inv[grp], a.inv[grp + 1], a.k);
    if (nv < 64) valid &= (1ull << nv) - 1;
    for_each_kmer_n<32>(A, B, 0, (uint32_t)valid, a.k, f);
    for_each_kmer_n<32>(B, C, 0, (uint32_t)(valid >> 32), a.k, [&](int i, uint64_t canon) { f(i + 32, canon); });
}

// K1: per-(genome,bucket) occurrence histogram.
// LDS path when the whole span lies in one genome; global atomics otherwise.
__global__ __launch_bounds__(KMER_THREADS) void kmer_hist_kernel(KmerArgs a, uint32_t n_spans,
                                                                 uint32_t *__restrict__ counts)
{
    extern __shared__ uint32_t lds_hist[];
    const uint64_t span = xcd_span(blockIdx.x, n_spans);
    if (span >= n_spans) return;
    const uint32_t B = 1u << a.bb;
    const uint64_t span_groups = (uint64_t)KMER_THREADS * a.groups_per_thread;
    const uint64_t g_first = span * span_groups;
    if (g_first >= a.n_groups) return;
    const uint64_t g_last = min(g_first + span_groups, a.n_groups) - 1;
    const uint64_t p_first = g_first << 6;
    const uint64_t p_last = min((g_last << 6) + 63, a.total_syms - 1);
    const uint32_t gen0 = genome_of(a.genome_sym_off, a.n_genomes, p_first);
    const bool uniform = a.genome_sym_off[gen0 + 1] > p_last;

    if (uniform) {
        for (uint32_t i = threadIdx.x; i < B; i += KMER_THREADS) lds_hist[i] = 0;
        __syncthreads();
        for (uint32_t it = 0; it < a.groups_per_thread; it++) {
            const uint64_t grp = g_first + (uint64_t)it * KMER_THREADS + threadIdx.x;
            if (grp > g_last) break;
            group_kmers(a, grp, [&](int, uint64_t canon) {
                atomicAdd(&lds_hist[hash_bucket(mix64(canon), a.bb)], 1u);
            });
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < B; i += KMER_THREADS) {
            const uint32_t c = lds_hist[i];
            if (c) atomicAdd(&counts[(uint64_t)gen0 * B + i], c);
        }
    } else {
        for (uint32_t it = 0; it < a.groups_per_thread; it++) {
            const uint64_t grp = g_first + (uint64_t)it * KMER_THREADS + threadIdx.x;
            if (grp > g_last) break;
            const uint64_t p0 = grp << 6;
            uint32_t gen = genome_of(a.genome_sym_off, a.n_genomes, p0);
            uint64_t gend = a.genome_sym_off[gen + 1];
            group_kmers(a, grp, [&](int i, uint64_t canon) {
                while (p0 + (uint64_t)i >= gend) { gen++; gend = a.genome_sym_off[gen + 1]; }
                atomicAdd(&counts[(uint64_t)gen * B + hash_bucket(mix64(canon), a.bb)], 1u);
            });
        }
    }
}

// K3: two-level LDS-staged radix partition of the canonical k-mers.
//
// A single-pass scatter of 8-byte keys into 2^13 buckets leaves L2 as partial lines (measured
// 4x HBM write amplification, profiles/r01/baseline_*).  Instead:
//   level 1  tile of 8192 positions -> k-mers kept in registers -> counting sort by the top
//            b1 (<= 8) hash bits in LDS -> every coarse bucket's run is copied out contiguously
//            (32 keys = 256 B on average): full-line writes.
//   level 2  each (genome, coarse bucket) region is re-read in tiles of 8192 keys and split by
//            the remaining b2 (<= 5) bits the same way (runs of ~256 keys).
// Fine bucket ids are contiguous inside a coarse bucket (bucket = top bits of the hash), so the
// coarse region of level 1 IS the union of its fine segments and `off` (exclusive scan of the
// fine histogram) serves both levels.
// Region of (genome, coarse bucket) in keys1, three ways:
//   region_stride != 0  fixed-capacity regions, region c at c * region_stride ("slack" layout: no histogram pass at
//                       all; a region that would overflow raises *overflow and the host falls back to the dense layout)
//   coarse_off          dense, from the coarse histogram (deep mode)
//   off                 dense, from the fine histogram (fine ids are nested inside the coarse bucket)
__global__ __launch_bounds__(L1_THREADS) void kmer_scatter_l1_kernel(
    KmerArgs a, int b1bits, uint32_t n_tiles, const uint64_t *__restrict__ off, const uint64_t *__restrict__ coarse_off,
    uint32_t *__restrict__ cursor1, uint64_t *__restrict__ keys1, uint64_t region_stride, int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint64_t *skeys = reinterpret_cast<uint64_t *>(lds_raw);                             // [L1_TILE]
    uint64_t *gbase = reinterpret_cast<uint64_t *>(lds_raw + (size_t)L1_TILE * 8);        // [256]
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds_raw + (size_t)L1_TILE * 8 + 2048);  // [256]
    uint32_t *start = hist + 256;                                                        // [256]
    uint32_t *scratch = start + 256;                                                     // [16]
    uint8_t *sbkt = reinterpret_cast<uint8_t *>(scratch + 16);                           // [L1_TILE] coarse bucket of skeys[i]
    const uint64_t tile = xcd_span(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int b2bits = a.bb - b1bits;
    const uint32_t B1 = 1u << b1bits;
    const uint64_t p0 = tile * L1_TILE + (uint64_t)threadIdx.x * L1_PPT;      // L1_PPT start positions per thread
    const uint64_t p_first = tile * L1_TILE;
    if (p_first >= a.total_syms) return;
    const uint64_t p_last = min(p_first + L1_TILE, a.total_syms) - 1;
    const uint32_t gen0 = genome_of(a.genome_sym_off, a.n_genomes, p_first);
    const bool uniform = a.genome_sym_off[gen0 + 1] > p_last;

    uint32_t valid = 0;
    uint64_t w0 = 0, w1 = 0;
    const int64_t nv = (int64_t)a.total_syms - a.k + 1 - (int64_t)p0;
    if (nv > 0) {
        const uint64_t grp = p0 >> 6;
        w0 = a.sym2[p0 >> 5];
        w1 = a.sym2[(p0 >> 5) + 1];
        valid = (uint32_t)valid_starts_at(a.inv[grp], a.inv[grp + 1], (int)(p0 & 63), a.k) & ((1u << L1_PPT) - 1);
        if (nv < L1_PPT) valid &= (1u << nv) - 1;
    }
    const int off_in_word = (int)(p0 & 31);

    if (uniform) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        uint64_t kv[L1_PPT];
        uint32_t bk[L1_PPT], rk[L1_PPT];  // coarse bucket, rank inside the tile's bucket.  The rank is
                                          // the RETURN value of an LDS atomic: it is not touched until the
                                          // placement loop, so the 16 atomics stay in flight instead of
                                          // costing one LDS round trip each.
        for_each_kmer_n<L1_PPT>(w0, w1, off_in_word, valid, a.k, [&](int i, uint64_t canon) {
            kv[i] = canon;
            bk[i] = hash_bucket(mix64(canon), b1bits);
            rk[i] = atomicAdd(&hist[bk[i]], 1u);
        });
        __syncthreads();
        const uint32_t c = threadIdx.x < B1 ? hist[threadIdx.x] : 0u;
        uint32_t n_tile;
        const uint32_t st = block_scan_sum(c, scratch, &n_tile);
        if (threadIdx.x < 256) start[threadIdx.x] = st;
        uint64_t region0 = 0;
        uint32_t reserved = 0;      // not touched before the placement is done: see lds_barrier()
        if (c) {
            // start of the coarse region: from the coarse scan (deep mode: fine offsets do not exist
            // yet) or from the fine scan (fine ids are nested inside the coarse bucket)
            const uint64_t cidx = (uint64_t)gen0 * B1 + threadIdx.x;
            region0 = region_stride ? cidx * region_stride
                      : coarse_off  ? coarse_off[cidx]
                                    : off[(uint64_t)gen0 * (1ull << a.bb) + ((uint64_t)threadIdx.x << b2bits)];
            reserved = atomicAdd(&cursor1[cidx], c);
        }
        lds_barrier();      // start[] is visible; the reservation (a global round trip) lands during the placement
#pragma unroll
        for (int i = 0; i < L1_PPT; i++)
            if ((valid >> i) & 1u) {
                const uint32_t at = start[bk[i]] + rk[i];
                skeys[at] = kv[i];
                sbkt[at] = (uint8_t)bk[i];        // the copy-out then needs no hash
            }
        if (c) {
            const bool fits = !region_stride || (uint64_t)reserved + c <= region_stride;
            if (!fits) atomicExch(overflow, 1);
            gbase[threadIdx.x] = fits ? region0 + reserved : ~0ull;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_tile; i += L1_THREADS) {
            const uint32_t b1 = sbkt[i];
            const uint64_t gb = gbase[b1];
            if (gb != ~0ull) keys1[gb + (i - start[b1])] = skeys[i];
        }
    } else {
        uint32_t gen = genome_of(a.genome_sym_off, a.n_genomes, min(p0, a.total_syms - 1));
        uint64_t gend = a.genome_sym_off[gen + 1];
        for_each_kmer_n<L1_PPT>(w0, w1, off_in_word, valid, a.k, [&](int i, uint64_t canon) {
            while (p0 + (uint64_t)i >= gend) { gen++; gend = a.genome_sym_off[gen + 1]; }
            const uint32_t b1 = hash_bucket(mix64(canon), b1bits);
            const uint64_t cidx = (uint64_t)gen * B1 + b1;
            const uint64_t region0 = region_stride ? cidx * region_stride
                                     : coarse_off  ? coarse_off[cidx]
                                                   : off[(uint64_t)gen * (1ull << a.bb) + ((uint64_t)b1 << b2bits)];
            const uint32_t at = atomicAdd(&cursor1[cidx], 1u);
            if (region_stride && at >= region_stride) atomicExch(overflow, 1);
            else keys1[region0 + at] = canon;
        });
    }
}

// deep mode (more than 2^13 buckets per genome, e.g. read sets at high coverage): the fine
// histogram does not fit an LDS histogram in the k-mer pass, so it is taken from the level-1
// output: one workgroup per (genome, coarse bucket) region counts its <= 256 fine buckets.
__global__ __launch_bounds__(256) void region_hist_kernel(const uint64_t *__restrict__ keys1,
                                                          const uint64_t *__restrict__ coarse_off, uint64_t n_regions, int bb,
                                                          int b1bits, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t hist[256];
    const uint32_t B2 = 1u << (bb - b1bits);
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t r0 = coarse_off[region], r1 = coarse_off[region + 1];
        for (uint64_t i = r0 + threadIdx.x; i < r1; i += 256) atomicAdd(&hist[hash_bucket(mix64(keys1[i]), bb) & (B2 - 1)], 1u);
        __syncthreads();
        if (threadIdx.x < B2) counts[region * B2 + threadIdx.x] = hist[threadIdx.x];
        __syncthreads();
    }
}

// Level 2.  Dense layout: region and fine segments from `off`.  Slack layout (region_stride != 0): region c of keys1
// holds cursor1[c] keys at c * region_stride; fine segment f is written at f * fine_cap and its length goes to
// len_out[f]; a segment that would exceed fine_cap raises *overflow (the host then redoes the partition densely).
__global__ __launch_bounds__(L2_THREADS) void kmer_scatter_l2_kernel(
    const uint64_t *__restrict__ keys1, uint64_t *__restrict__ keys, const uint64_t *__restrict__ off,
    uint64_t n_regions, int bb, int b1bits, uint64_t region_stride, uint32_t fine_cap, const uint32_t *__restrict__ cursor1,
    uint32_t *__restrict__ len_out, int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint64_t *skeys = reinterpret_cast<uint64_t *>(lds_raw);
    uint64_t *gbase = reinterpret_cast<uint64_t *>(lds_raw + (size_t)L2_TILE * 8);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds_raw + (size_t)L2_TILE * 8 + 2048);
    uint32_t *start = hist + 256;
    uint32_t *scratch = start + 256;
    const int b2bits = bb - b1bits;
    const uint32_t B2 = 1u << b2bits;          // <= 32 normally, up to 256 in deep mode
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        const uint64_t g = region >> b1bits, c1 = region & ((1u << b1bits) - 1);
        const uint64_t fine0 = (g << bb) + (c1 << b2bits);
        uint64_t r0, r1, my_first = 0;
        if (region_stride) {
            r0 = region * region_stride;
            r1 = r0 + min((uint64_t)cursor1[region], region_stride);
            if (threadIdx.x < B2) my_first = (fine0 + threadIdx.x) * (uint64_t)fine_cap;
        } else {
            r0 = off[fine0];
            r1 = off[fine0 + B2];
            if (threadIdx.x < B2) my_first = off[fine0 + threadIdx.x];
        }
        // a region is written by this workgroup alone, tile after tile: thread t keeps the running
        // output position of fine bucket t in a register (no global cursor, no atomic round trip)
        uint64_t my_next = my_first;
        for (uint64_t base = r0; base < r1; base += L2_TILE) {
            const uint32_t n = (uint32_t)min((uint64_t)L2_TILE, r1 - base);
            if (threadIdx.x < B2) hist[threadIdx.x] = 0;
            __syncthreads();
            uint64_t kv[L2_PPT];
            uint32_t bk[L2_PPT], rk[L2_PPT];      // fine sub-bucket, returned rank (left in flight, see level 1)
#pragma unroll
            for (int j = 0; j < L2_PPT; j++) {
                const uint32_t i = (uint32_t)j * L2_THREADS + threadIdx.x;
                kv[j] = i < n ? keys1[base + i] : EMPTY_KEY;
            }
#pragma unroll
            for (int j = 0; j < L2_PPT; j++) {
                if (kv[j] != EMPTY_KEY) {
                    bk[j] = hash_bucket(mix64(kv[j]), bb) & (B2 - 1);
                    rk[j] = atomicAdd(&hist[bk[j]], 1u);
                }
            }
            __syncthreads();
            const uint32_t c = threadIdx.x < B2 ? hist[threadIdx.x] : 0u;
            uint32_t n_tile;
            const uint32_t st = block_scan_sum(c, scratch, &n_tile);
            if (threadIdx.x < B2) {
                start[threadIdx.x] = st;
                const bool fits = !region_stride || my_next + c <= my_first + fine_cap;
                if (!fits) atomicExch(overflow, 1);
                gbase[threadIdx.x] = fits ? my_next : ~0ull;
                if (fits) my_next += c;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < L2_PPT; j++)
                if (kv[j] != EMPTY_KEY) skeys[start[bk[j]] + rk[j]] = kv[j];
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += L2_THREADS) {
                const uint64_t key = skeys[i];
                const uint32_t b2 = hash_bucket(mix64(key), bb) & (B2 - 1);
                const uint64_t gb = gbase[b2];
                if (gb != ~0ull) keys[gb + (i - start[b2])] = key;
            }
            __syncthreads();
        }
        if (region_stride && threadIdx.x < B2) len_out[fine0 + threadIdx.x] = (uint32_t)(my_next - my_first);
    }
}

// sum of n uint32 added to *out (uint64, zeroed by the caller)
__global__ __launch_bounds__(256) void sum_u32_kernel(const uint32_t *__restrict__ in, uint64_t n, unsigned long long *__restrict__ out)
{
    __shared__ uint64_t scratch[16];
    uint64_t s = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += in[i];
    uint64_t total;
    (void)block_scan_sum64(s, scratch, &total);
    if (threadIdx.x == 0 && total) atomicAdd(out, (unsigned long long)total);
}

// partition an explicit key list (grm_build_matrix from host-side sets): hist + scatter
__global__ void keys_hist_kernel(const uint64_t *__restrict__ in, uint64_t n,
                                 const uint64_t *__restrict__ genome_key_off, uint32_t n_genomes, int bb,
                                 uint32_t *__restrict__ counts)
{
    const uint32_t B = 1u << bb;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = genome_of(genome_key_off, n_genomes, i);
        atomicAdd(&counts[(uint64_t)g * B + hash_bucket(mix64(in[i]), bb)], 1u);
    }
}
__global__ void keys_scatter_kernel(const uint64_t *__restrict__ in, uint64_t n,
                                    const uint64_t *__restrict__ genome_key_off, uint32_t n_genomes, int bb,
                                    const uint64_t *__restrict__ off, uint32_t *__restrict__ cursor,
                                    uint64_t *__restrict__ keys)
{
    const uint32_t B = 1u << bb;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = genome_of(genome_key_off, n_genomes, i);
        const uint64_t key = in[i];
        const uint64_t idx = (uint64_t)g * B + hash_bucket(mix64(key), bb);
        keys[off[idx] + atomicAdd(&cursor[idx], 1u)] = key;
    }
}

// single-workgroup exclusive scan u32 -> u64 (n up to a few million entries); out[n] = total
__global__ __launch_bounds__(1024) void scan_u32_kernel(const uint32_t *__restrict__ in, uint64_t n,
                                                        uint64_t *__restrict__ out)
{
    __shared__ uint64_t scratch[16];
    const uint64_t per = (n + blockDim.x - 1) / blockDim.x;
    const uint64_t i0 = min((uint64_t)threadIdx.x * per, n), i1 = min(i0 + per, n);
    uint64_t s = 0;
    for (uint64_t i = i0; i < i1; i++) s += in[i];
    uint64_t total;
    uint64_t o = block_scan_sum64(s, scratch, &total);
    for (uint64_t i = i0; i < i1; i++) { out[i] = o; o += in[i]; }
    if (threadIdx.x == 0) out[n] = total;
}

// ------------------------------------------------------------------------------------
// LDS open-addressing table helpers (linear probing, 64-bit CAS = ds_cmpst_rtn_b64)
// ------------------------------------------------------------------------------------
// "this workgroup's table overflowed": a flag in LDS read inside loops.  (A volatile int would be read through the FLAT
// path -- flat_load + s_waitcnt vmcnt(0), which also drains the global loads in flight; a relaxed atomic stays a ds_read.)
struct LdsFlag {
    int *p;
    __device__ __forceinline__ operator bool() const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; }
    __device__ __forceinline__ void set(int v) const { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
};
// returns slot of key (inserting it if absent), or 0xffffffff when the table is full
__device__ __forceinline__ uint32_t lds_find_or_insert(uint64_t *tkeys, uint32_t cap_mask, uint64_t key,
                                                       uint64_t h, bool *inserted)
{
    uint32_t slot = hash_slot(h, cap_mask);
    for (uint32_t probe = 0; probe <= cap_mask; probe++) {
        // plain read first: almost every probe of a pan-genome finds its key already present,
        // and a ds_read_b64 is cheaper than a returning ds_cmpst_b64
        uint64_t cur = lds_peek(&tkeys[slot]);
        if (cur == EMPTY_KEY)
            cur = atomicCAS((unsigned long long *)&tkeys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (cur == EMPTY_KEY) { *inserted = true; return slot; }
        if (cur == key) { *inserted = false; return slot; }
        slot = (slot + 1) & cap_mask;
    }
    return 0xffffffffu;
}
__device__ __forceinline__ uint32_t lds_find(const uint64_t *tkeys, uint32_t cap_mask, uint64_t key, uint64_t h)
{
    uint32_t slot = hash_slot(h, cap_mask);
    for (uint32_t probe = 0; probe <= cap_mask; probe++) {
        const uint64_t cur = tkeys[slot];
        if (cur == key) return slot;
        if (cur == EMPTY_KEY) return 0xffffffffu;
        slot = (slot + 1) & cap_mask;
    }
    return 0xffffffffu;
}

__device__ __forceinline__ void seg_bounds(const SegLayout &L, uint64_t idx, uint64_t &s0, uint64_t &n)
{
    if (L.off) {
        s0 = L.off[idx];
        n = L.len ? (uint64_t)L.len[idx] : L.off[idx + 1] - s0;
    } else {
        s0 = idx * L.stride;
        n = L.len[idx];
    }
}

// K4: per-(genome,bucket) dedup + count + abundance filter, in place.
// After the kernel the segment [off[i], off[i]+len_out[i]) holds the distinct k-mers whose
// count >= abundance_min (order = table slot order); counts_out (optional) is parallel to keys.
__global__ __launch_bounds__(TABLE_THREADS) void bucket_dedup_kernel(
    uint64_t *__restrict__ keys, const SegLayout seg_layout, uint64_t n_segments, uint32_t cap_log2,
    uint32_t abundance_min, uint32_t *__restrict__ len_out, const uint32_t *__restrict__ marks, uint32_t *__restrict__ counts_out,
    int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t cap = 1u << cap_log2, cap_mask = cap - 1;
    uint64_t *tkeys = reinterpret_cast<uint64_t *>(lds_raw);
    uint32_t *tcnt = reinterpret_cast<uint32_t *>(lds_raw + (size_t)cap * 8);
    uint32_t *scratch = reinterpret_cast<uint32_t *>(lds_raw + (size_t)cap * 12);   // [16] + flags
    const LdsFlag full = {reinterpret_cast<int *>(scratch + 16)};
    for (uint64_t seg = blockIdx.x; seg < n_segments; seg += gridDim.x) {
        if (marks && !((marks[seg >> 5] >> (seg & 31)) & 1u)) continue;      // second pass: what the wave form left
        uint64_t s0, n;
        seg_bounds(seg_layout, seg, s0, n);
        for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) { tkeys[i] = EMPTY_KEY; tcnt[i] = 0; }
        if (threadIdx.x == 0) full.set(0);
        __syncthreads();
        for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint64_t key = keys[s0 + i];
            bool ins;
            const uint32_t slot = lds_find_or_insert(tkeys, cap_mask, key, mix64(key), &ins);
            if (slot == 0xffffffffu) full.set(1);
            else atomicAdd(&tcnt[slot], 1u);
        }
        __syncthreads();
        if (full) {
            if (threadIdx.x == 0) { atomicExch(overflow, 1); len_out[seg] = 0; }
            __syncthreads();
            continue;
        }
        uint32_t base = 0;
        for (uint32_t s = 0; s < cap; s += blockDim.x) {
            const uint32_t slot = s + threadIdx.x;
            const uint64_t key = slot < cap ? tkeys[slot] : EMPTY_KEY;
            const uint32_t c = slot < cap ? tcnt[slot] : 0u;
            const bool keep = key != EMPTY_KEY && c >= abundance_min;
            uint32_t sweep_total;
            const uint32_t pos = sweep_compact(keep, scratch, &sweep_total);
            if (keep) {
                keys[s0 + base + pos] = key;
                if (counts_out) counts_out[s0 + base + pos] = c;
            }
            base += sweep_total;
        }
        if (threadIdx.x == 0) len_out[seg] = base;
        __syncthreads();
    }
}

// K4, wave form: one WAVE per (genome, bucket) segment, a table of its own in LDS, no workgroup barrier anywhere.
// The workgroup form above spends a 4096-slot clear, a 4096-slot sweep and ~20 barriers on a segment of ~600
// keys; here the table is sized to the segment (CAP = 512 / 1024 / 2048 slots) and the three phases -- clear,
// insert + count, ballot-compaction in place -- run back to back inside the wave (LDS operations of one wave
// execute in order).  A segment with more distinct k-mers than 7/8 of the table is left untouched and marked
// (len_out = 0xffffffff, *overflow = 1) for the workgroup form.
__device__ __forceinline__ void wave_lds_fence()
{
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): this wave's LDS operations have completed
    __asm__ volatile("" ::: "memory");
}
// insert four keys per lane: their first-probe reads, then the claims of empty home slots, are in flight together;
// only a key that finds someone else's key at home walks the probe sequence.  Returns false when the table is full.
template <uint32_t MASK>
__device__ __forceinline__ bool dedup_insert4(uint64_t *tk, uint32_t *tc, const uint64_t kv[4], uint32_t &n_ins)
{
    uint32_t sl[4];
    uint64_t cur[4];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        sl[j] = hash_slot(mix64(kv[j]), MASK);
        cur[j] = lds_peek(&tk[sl[j]]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (kv[j] != EMPTY_KEY && cur[j] == EMPTY_KEY)
            cur[j] = atomicCAS((unsigned long long *)&tk[sl[j]], (unsigned long long)EMPTY_KEY, (unsigned long long)kv[j]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        bool ins = false;
        if (kv[j] != EMPTY_KEY) {
            uint32_t slot = sl[j];
            if (cur[j] == EMPTY_KEY) ins = true;                       // the CAS claimed the home slot
            else if (cur[j] != kv[j]) {                                // someone else's key there: probe on
                slot = 0xffffffffu;
                uint32_t at = (sl[j] + 1) & MASK;
                for (uint32_t probe = 0; probe < MASK; probe++, at = (at + 1) & MASK) {
                    uint64_t c2 = lds_peek(&tk[at]);
                    if (c2 == EMPTY_KEY)
                        c2 = atomicCAS((unsigned long long *)&tk[at], (unsigned long long)EMPTY_KEY, (unsigned long long)kv[j]);
                    if (c2 == EMPTY_KEY) { ins = true; slot = at; break; }
                    if (c2 == kv[j]) { slot = at; break; }
                }
            }
            if (slot != 0xffffffffu) atomicAdd(&tc[slot], 1u);
            else ok = false;
        }
        n_ins += __popcll(__ballot(ins));
    }
    return ok;
}

template <int CAP_LOG2, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bucket_dedup_wave_kernel(
    uint64_t *__restrict__ keys, const SegLayout seg_layout, uint64_t n_segments, uint32_t abundance_min,
    uint32_t *__restrict__ len_out, uint32_t *__restrict__ marks, uint32_t *__restrict__ counts_out, int *__restrict__ overflow)
{
    constexpr uint32_t CAP = 1u << CAP_LOG2, MASK = CAP - 1, MAX_FILL = CAP - CAP / 8;
    // A wave has few companions on its CU (the tables fill the LDS), so it must keep many bytes in flight by
    // itself: the first 64 x KD keys of a segment -- the whole segment in the design case -- are loaded in one go, and
    // the NEXT segment's while the current one is compacted.
    constexpr int KD = 12;
    __shared__ uint64_t tk_all[WAVES][CAP];
    __shared__ uint32_t tc_all[WAVES][CAP];
    const int lane = lane_id(), wave = wave_id();
    uint64_t *tk = tk_all[wave];
    uint32_t *tc = tc_all[wave];
    const uint64_t lane_lt = (1ull << lane) - 1;
    const uint64_t stride = (uint64_t)gridDim.x * WAVES;
    uint64_t seg = (uint64_t)blockIdx.x * WAVES + wave;
    if (seg >= n_segments) return;
    uint64_t s0, n;
    seg_bounds(seg_layout, seg, s0, n);
    uint64_t kv[KD];
#pragma unroll
    for (int j = 0; j < KD; j++) {
        const uint64_t i = 64u * j + lane;
        kv[j] = i < n ? keys[s0 + i] : EMPTY_KEY;
    }
    for (;;) {
        const uint64_t seg_next = seg + stride;
        uint64_t s0n = 0, nn = 0;
        if (seg_next < n_segments) seg_bounds(seg_layout, seg_next, s0n, nn);     // needed only after the inserts
        for (uint32_t i = lane; i < CAP; i += 64) { tk[i] = EMPTY_KEY; tc[i] = 0; }
        wave_lds_fence();
        uint32_t nd = 0;
        bool over = false;
#pragma unroll
        for (int j0 = 0; j0 < KD; j0 += 4) {
            if (64u * j0 < n) over |= !dedup_insert4<MASK>(tk, tc, kv + j0, nd);
        }
        over = __any(over) || nd > MAX_FILL;
        // the rest of a segment longer than the register window
        for (uint64_t i0 = 64u * KD; i0 < n && !over; i0 += 64 * 4) {
            uint64_t kx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint64_t i = i0 + 64u * j + lane;
                kx[j] = i < n ? keys[s0 + i] : EMPTY_KEY;
            }
            over = !dedup_insert4<MASK>(tk, tc, kx, nd);
            over = __any(over) || nd > MAX_FILL;
        }
        // next segment's keys: in flight during the compaction below
#pragma unroll
        for (int j = 0; j < KD; j++) {
            const uint64_t i = 64u * j + lane;
            kv[j] = i < nn ? keys[s0n + i] : EMPTY_KEY;
        }
        if (over) {
            if (lane == 0) { atomicOr(&marks[seg >> 5], 1u << (seg & 31)); atomicExch(overflow, 1); }
        } else {
            wave_lds_fence();
            uint32_t base = 0;
#pragma unroll 4
            for (uint32_t s = 0; s < CAP; s += 64) {
                const uint64_t key = tk[s + lane];
                const uint32_t c = tc[s + lane];
                const bool keep = key != EMPTY_KEY && c >= abundance_min;
                const uint64_t m = __ballot(keep);
                if (keep) {
                    const uint32_t pos = base + __popcll(m & lane_lt);
                    keys[s0 + pos] = key;
                    if (counts_out) counts_out[s0 + pos] = c;
                }
                base += __popcll(m);
            }
            if (lane == 0) len_out[seg] = base;
        }
        wave_lds_fence();
        if (seg_next >= n_segments) break;
        seg = seg_next; s0 = s0n; n = nn;
    }
}

// K4 from the record form (counting stage; one part per genome): the (genome, minimizer bucket) segments are runs of level-2
// RECORDS -- 16 bytes for ~11 k-mers -- and the distinct k-mers of a segment with their counts leave as key segments in the
// region's key space (region * kstride; a segment gets room for all its k-mers, koff / klen say where the distinct ones stand).
// Two launches: record_count_kernel (below) takes every segment a wave can hold, record_dedup_kernel what that leaves -- segments
// of more than 7/8 x 512 k-mers or more than 64 records; minimizer buckets are uneven, a few per thousand hold three times the mean.
// Common to both: one workgroup per region at a time.  Pass A adds up the records and k-mers per fine bucket (level 2 left the
// region's records sorted by fine bucket) and lays the key segments out.  A chunk of 64 records goes to LDS with the prefix sum of
// the run lengths and a bitmap of the positions where a record's k-mers start, so that a LANE takes a K-MER (lanes that rolled
// through records of 1..21 k-mers would idle two thirds of the time): its record is the popcount of the bitmap below its
// position, the k-mer is cut out of the record's words directly.
// record_dedup_kernel: the workgroup's waves share ONE table of 64-bit keys and counts; the slots a segment claims are listed as
// they are claimed, and the pass that writes the segment out walks that list and leaves every slot empty again (nothing is
// proportional to the table).  Only a segment whose DISTINCT k-mers exceed that table raises *overflow.
__device__ __forceinline__ uint32_t record_slot_hash(uint64_t key)
{
    // three 24-bit multiplies (full rate) instead of mix64's 64-bit ones; the table index is taken from the top
    const uint32_t a = (uint32_t)key & 0xffffffu, b = (uint32_t)(key >> 24) & 0xffffffu, c = (uint32_t)(key >> 48);
    return __umul24(a, 0xC2B2AFu) + __umul24(b, 0x85EBCBu) + (__umul24(c, 0x9E3779u) << 7);
}
// records of a chunk -> LDS; returns the k-mers they hold
__device__ __forceinline__ uint32_t record_chunk_stage(ulonglong2 rec, uint32_t n_chunk, ulonglong2 *srec, uint16_t *sstart, uint64_t *starts,
                                                       uint8_t *firstrec)
{
    const int lane = lane_id();
    const uint32_t ln = (uint32_t)lane < n_chunk ? run_len(rec.y) : 0u;
    const uint32_t incl = wave_scan_incl_dpp(ln), st = incl - ln;
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    srec[lane] = rec;
    sstart[lane] = (uint16_t)st;
    // (a block of 64 positions in which no record starts -- the chunk's last -- belongs to the last record: first record "n_chunk")
    if (lane < (int)((64u * RUN_LMAX + 63) / 64)) { starts[lane] = 0; firstrec[lane] = (uint8_t)n_chunk; }
    wave_lds_fence();
    const uint32_t w = st >> 6;
    const uint32_t w_prev = __shfl_up(w, 1);
    if (ln) {
        atomicOr((unsigned long long *)&starts[w], 1ull << (st & 63u));
        if (lane == 0 || w_prev != w) firstrec[w] = (uint8_t)lane;
    }
    wave_lds_fence();
    return tot;
}
// canonical k-mer number q of the staged chunk
__device__ __forceinline__ uint64_t record_chunk_kmer(uint32_t q, int k, const ulonglong2 *srec, const uint16_t *sstart, const uint64_t *starts,
                                                      const uint8_t *firstrec)
{
    const uint32_t w = q >> 6;
    const uint64_t word = starts[w];
    const uint32_t o = (uint32_t)firstrec[w] + (uint32_t)__popcll(word & ((2ull << (q & 63u)) - 1ull)) - 1u;
    const ulonglong2 r = srec[o];
    return run_kmer_at(r.x, r.y, k, q - sstart[o]);
}
// up to four keys per lane (EMPTY_KEY = none) into the table, counted; slots claimed are appended to the list dl.
// As dedup_insert4: the first-probe reads, then the claims of empty home slots, are in flight together; only a key that finds
// someone else's key at home walks the probe sequence.  (Probing the four keys side by side until the last lane's last key has a
// slot was tried: every round costs the whole wave four keys' worth of instructions, 4.5 against 3.5 wave instructions per k-mer.)
// SHARED: several waves work on the table (the list's end is an LDS counter).  Returns false when the table is full.
// nj: how many of the four hold keys in ANY lane (uniform): the others are skipped.
template <uint32_t CAP_LOG2, bool SHARED>
__device__ __forceinline__ bool record_insert4(uint64_t *tk, uint32_t *tc, uint16_t *dl, uint32_t *nd_shared, const uint64_t kv[4], uint32_t nj, uint32_t &nd)
{
    constexpr uint32_t MASK = (1u << CAP_LOG2) - 1;
    uint32_t sl[4];
    uint64_t cur[4];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if ((uint32_t)j >= nj) break;
        sl[j] = record_slot_hash(kv[j]) >> (32 - CAP_LOG2);
        cur[j] = lds_peek(&tk[sl[j]]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if ((uint32_t)j >= nj) break;
        if (kv[j] != EMPTY_KEY && cur[j] == EMPTY_KEY)
            cur[j] = atomicCAS((unsigned long long *)&tk[sl[j]], (unsigned long long)EMPTY_KEY, (unsigned long long)kv[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if ((uint32_t)j >= nj) break;
        bool ins = false;
        uint32_t slot = sl[j];
        if (kv[j] != EMPTY_KEY) {
            if (cur[j] == EMPTY_KEY) ins = true;                       // the CAS claimed the home slot
            else if (cur[j] != kv[j]) {                                // someone else's key there: probe on
                slot = 0xffffffffu;
                uint32_t at = (sl[j] + 1) & MASK;
                for (uint32_t probe = 0; probe < MASK; probe++, at = (at + 1) & MASK) {
                    uint64_t c2 = lds_peek(&tk[at]);
                    if (c2 == EMPTY_KEY)
                        c2 = atomicCAS((unsigned long long *)&tk[at], (unsigned long long)EMPTY_KEY, (unsigned long long)kv[j]);
                    if (c2 == EMPTY_KEY) { ins = true; slot = at; break; }
                    if (c2 == kv[j]) { slot = at; break; }
                }
            }
            if (slot != 0xffffffffu) atomicAdd(&tc[slot], 1u);
            else ok = false;
        }
        const uint64_t m = __ballot(ins);
        if (m) {                                            // uniform
            uint32_t base = nd;
            if (SHARED) {
                uint32_t got = 0;
                if (lane_id() == 0) got = atomicAdd(nd_shared, (uint32_t)__popcll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            }
            if (ins) dl[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)slot;
            nd += (uint32_t)__popcll(m);
        }
    }
    return !__any(!ok);
}

template <int CAP_LOG2, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void record_dedup_kernel(
    const ulonglong2 *__restrict__ recs, uint32_t rstride, const uint32_t *__restrict__ rcount, uint64_t n_regions, int k, int b2,
    uint64_t kstride, uint32_t abundance_min,
    uint64_t *__restrict__ keys, uint32_t *__restrict__ counts_out, uint64_t *__restrict__ koff, uint32_t *__restrict__ klen,
    int *__restrict__ overflow, const uint8_t *__restrict__ region_big)
{
    // (CAP: the slots of a wave's table in record_count_kernel<CAP_LOG2>, whose leavings these are)
    constexpr uint32_t CAP = 1u << CAP_LOG2, MAX_FILL = CAP - CAP / 8;
    constexpr uint32_t BIG = CAP * WAVES;
    constexpr int BIG_LOG2 = CAP_LOG2 + (WAVES == 4 ? 2 : WAVES == 2 ? 1 : 3);
    static_assert((1u << BIG_LOG2) == BIG, "WAVES is 2, 4 or 8");
    constexpr int NF = 1 << RUN_FINE_BITS;
    constexpr uint32_t CH = 64, CH_WORDS = (CH * RUN_LMAX + 63) / 64;
    constexpr uint32_t THREADS = WAVES * 64;
    __shared__ uint64_t tk_all[BIG];                    // the waves' tables; ONE table of the workgroup for a segment too large for a wave's
    __shared__ uint32_t tc_all[BIG];
    __shared__ uint16_t dl_all[BIG];                    // slots claimed, in the order of their claims
    __shared__ ulonglong2 srec_all[WAVES][CH];
    __shared__ uint16_t sstart_all[WAVES][CH];
    __shared__ uint64_t starts_all[WAVES][CH_WORDS];
    __shared__ uint8_t firstrec_all[WAVES][CH_WORDS + 2];
    __shared__ uint32_t kcount[NF], kstart[NF], rcnt[NF], rstart[NF];       // k-mers / records per fine bucket, and where they start
    __shared__ uint16_t big_list[NF];
    __shared__ uint32_t scratch[32];
    __shared__ uint64_t scratch64[32];
    __shared__ uint32_t big_fail, big_nd;
    const int lane = lane_id(), wave = wave_id();
    const uint32_t B2 = 1u << b2;
    ulonglong2 *srec = srec_all[wave];
    uint16_t *sstart = sstart_all[wave];
    uint64_t *starts = starts_all[wave];
    uint8_t *firstrec = firstrec_all[wave];
    // the tables start empty and every segment leaves them so
    for (uint32_t i = threadIdx.x; i < BIG; i += THREADS) { tk_all[i] = EMPTY_KEY; tc_all[i] = 0; }
    __syncthreads();
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        if (!region_big[region]) continue;                      // only what record_count_kernel left
        const uint32_t n = min(rcount[region], rstride);
        const ulonglong2 *rr = recs + region * rstride;
        const uint64_t seg0 = region << b2;
        // ---- pass A: records and k-mers per fine bucket -> record segments, key segments ----
        if (threadIdx.x < NF) { kcount[threadIdx.x] = 0; rcnt[threadIdx.x] = 0; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
            const uint64_t y = rr[i].y;
            const uint32_t f = run_fine(y) >> (RUN_FINE_BITS - b2);
            atomicAdd(&kcount[f], run_len(y));
            atomicAdd(&rcnt[f], 1u);
        }
        __syncthreads();
        const uint32_t cnt = threadIdx.x < B2 ? kcount[threadIdx.x] : 0u;
        const uint32_t rc = threadIdx.x < B2 ? rcnt[threadIdx.x] : 0u;
        uint64_t both_total;
        const uint64_t both = block_scan_sum64(((uint64_t)rc << 32) | cnt, scratch64, &both_total);
        const uint32_t pre = (uint32_t)both, region_keys = (uint32_t)both_total;
        const bool fits = region_keys <= kstride;                 // uniform
        // ("small" is what record_count_kernel<CAP_LOG2> took -- up to MAX_FILL k-mers in up to 64 records)
        const bool is_small = threadIdx.x < B2 && cnt <= MAX_FILL && rc <= 64u;
        const bool is_big = threadIdx.x < B2 && !is_small;
        uint32_t n_big;
        const uint32_t pb = block_scan_sum(is_big ? 1u : 0u, scratch, &n_big);
        if (threadIdx.x < B2) {
            kstart[threadIdx.x] = pre;
            rstart[threadIdx.x] = (uint32_t)(both >> 32);
            koff[seg0 + threadIdx.x] = region * kstride + pre;
            if (!fits) klen[seg0 + threadIdx.x] = 0;
            if (is_big) big_list[pb] = (uint16_t)threadIdx.x;
        }
        if (threadIdx.x == 0) { big_fail = 0; big_nd = 0; }
        __syncthreads();
        if (!fits) {
            if (threadIdx.x == 0) atomicExch(overflow, 1);
            continue;
        }
        // ---- segments too large for a wave's table: the workgroup's waves share one table ----
        for (uint32_t bi = 0; bi < n_big; bi++) {
            const uint32_t f = big_list[bi];
            const uint32_t nr = rcnt[f], r0 = rstart[f];
            uint32_t nd = 0;
            bool ok = true;
            for (uint32_t c0 = (uint32_t)wave * CH; c0 < nr; c0 += WAVES * CH) {
                const uint32_t nc = min(CH, nr - c0);
                const ulonglong2 rec = (uint32_t)lane < nc ? rr[r0 + c0 + lane] : make_ulonglong2(0, 0);
                const uint32_t tot = record_chunk_stage(rec, nc, srec, sstart, starts, firstrec);
                for (uint32_t q0 = 0; q0 < tot; q0 += 256) {
                    uint64_t kv[4];
                    const uint32_t nj = min(4u, (tot - q0 + 63u) >> 6);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if ((uint32_t)j >= nj) break;
                        const uint32_t q = q0 + 64u * j + lane;
                        kv[j] = q < tot ? record_chunk_kmer(q, k, srec, sstart, starts, firstrec) : EMPTY_KEY;
                    }
                    ok &= record_insert4<BIG_LOG2, true>(tk_all, tc_all, dl_all, &big_nd, kv, nj, nd);
                }
                wave_lds_fence();
            }
            if (!ok && lane == 0) atomicExch(&big_fail, 1u);
            __syncthreads();
            const bool failed = big_fail != 0;                  // uniform
            const uint32_t nd_all = min(big_nd, BIG);
            const uint64_t dst = region * kstride + kstart[f];
            uint32_t base = 0;
            for (uint32_t p0 = 0; p0 < nd_all; p0 += THREADS) {
                const uint32_t p = p0 + threadIdx.x;
                const uint32_t slot = p < nd_all ? dl_all[p] : 0u;
                const uint64_t key = tk_all[slot];
                const uint32_t c = tc_all[slot];
                const bool keep = !failed && p < nd_all && c >= abundance_min;
                uint32_t sweep_total;
                const uint32_t pos = sweep_compact(keep, scratch, &sweep_total);
                if (keep) {
                    keys[dst + base + pos] = key;
                    if (counts_out) counts_out[dst + base + pos] = c;
                }
                base += sweep_total;
            }
            __syncthreads();
            for (uint32_t p = threadIdx.x; p < nd_all; p += THREADS) {
                const uint32_t slot = dl_all[p];
                tk_all[slot] = EMPTY_KEY;
                tc_all[slot] = 0;
            }
            if (threadIdx.x == 0) {
                klen[seg0 + f] = base;
                if (failed) atomicExch(overflow, 1);
                big_fail = 0;
                big_nd = 0;
            }
            __syncthreads();
        }
    }
}

// K4 from the record form when genomes are cut into PARTS (few large genomes, read sets): level 1 / level 2 left every part its own
// regions, a genome's k-mer counts are sums over its parts.  One workgroup of 8 waves per (genome, coarse bucket) at a time: pass A
// adds up the k-mers per fine bucket over the parts' regions and lays the GENOME's key segments out; then, fine bucket by fine
// bucket, the waves go through the parts' record segments (level 2's off / len) chunk by chunk -- lane = k-mer as above -- into ONE
// table of 2^BIG_LOG2 64-bit keys and counts; the claimed slots leave in the order of their claims and are emptied again.  A read
// set at 100x is mostly repeats of k-mers already in the table (read, compare, count); the table must hold the bucket's DISTINCT
// k-mers (solid + erroneous): *overflow otherwise, and the host tries the larger table, then the key form.
template <int BIG_LOG2>
__global__ __launch_bounds__(512) void record_merge_kernel(
    const ulonglong2 *__restrict__ recs, uint32_t rstride, const uint32_t *__restrict__ rcount, const uint64_t *__restrict__ roff,
    const uint32_t *__restrict__ rlen, uint32_t n_genomes, int part_bits, int k, int b1, int b2, uint64_t kstride, uint32_t abundance_min,
    uint64_t *__restrict__ keys, uint32_t *__restrict__ counts_out, uint64_t *__restrict__ koff, uint32_t *__restrict__ klen,
    int *__restrict__ overflow)
{
    constexpr uint32_t BIG = 1u << BIG_LOG2, WAVES = 8, THREADS = 512;
    constexpr int NF = 1 << RUN_FINE_BITS;
    constexpr uint32_t CH = 64, CH_WORDS = (CH * RUN_LMAX + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint64_t *tk_all = reinterpret_cast<uint64_t *>(lds_raw);
    uint32_t *tc_all = reinterpret_cast<uint32_t *>(lds_raw + (size_t)BIG * 8);
    uint16_t *dl_all = reinterpret_cast<uint16_t *>(lds_raw + (size_t)BIG * 12);
    uint8_t *wave_raw = lds_raw + (size_t)BIG * 14;                     // per wave: records 1024, starts bitmap 176, first record 24, run starts 128
    __shared__ uint32_t kcount[NF], kstart[NF];
    __shared__ uint32_t scratch[32];
    __shared__ uint32_t big_fail, big_nd;
    const int lane = lane_id(), wave = wave_id();
    const uint32_t B1 = 1u << b1, B2 = 1u << b2, P = 1u << part_bits;
    const int bb = b1 + b2;
    ulonglong2 *srec = reinterpret_cast<ulonglong2 *>(wave_raw + (size_t)wave * 1360);
    uint64_t *starts = reinterpret_cast<uint64_t *>(wave_raw + (size_t)wave * 1360 + 1024);
    uint8_t *firstrec = wave_raw + (size_t)wave * 1360 + 1024 + 176;
    uint16_t *sstart = reinterpret_cast<uint16_t *>(wave_raw + (size_t)wave * 1360 + 1024 + 176 + 32);
    static_assert(CH_WORDS * 8 <= 176 && CH_WORDS <= 32, "staging layout");
    for (uint32_t i = threadIdx.x; i < BIG; i += THREADS) { tk_all[i] = EMPTY_KEY; tc_all[i] = 0; }
    if (threadIdx.x == 0) { big_fail = 0; big_nd = 0; }
    __syncthreads();
    const uint64_t n_regions = (uint64_t)n_genomes << b1;
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        const uint32_t g = (uint32_t)(region >> b1), c = (uint32_t)region & (B1 - 1);
        const uint64_t seg0 = region << b2;                             // the genome's segments of this coarse bucket
        // ---- pass A: k-mers per fine bucket over the parts -> key segments ----
        if (threadIdx.x < NF) kcount[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t p = 0; p < P; p++) {
            const uint64_t rp = ((((uint64_t)g << part_bits) + p) << b1) + c;
            const uint32_t n = min(rcount[rp], rstride);
            const ulonglong2 *rr = recs + rp * rstride;
            for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
                const uint64_t y = rr[i].y;
                atomicAdd(&kcount[run_fine(y) >> (RUN_FINE_BITS - b2)], run_len(y));
            }
        }
        __syncthreads();
        const uint32_t cnt = threadIdx.x < B2 ? kcount[threadIdx.x] : 0u;
        uint32_t region_keys;
        const uint32_t pre = block_scan_sum(cnt, scratch, &region_keys);
        const bool fits = region_keys <= kstride;                 // uniform
        if (threadIdx.x < B2) {
            kstart[threadIdx.x] = pre;
            koff[seg0 + threadIdx.x] = region * kstride + pre;
            if (!fits || !cnt) klen[seg0 + threadIdx.x] = 0;
        }
        __syncthreads();
        if (!fits) {
            if (threadIdx.x == 0) atomicExch(overflow, 1);
            continue;
        }
        for (uint32_t f = 0; f < B2; f++) {
            if (!kcount[f]) continue;                              // uniform
            uint32_t nd = 0;
            bool ok = true;
            // a wave takes whole parts: their record segments (level 2's bounds).  The bounds of the wave's part AFTER the next one are asked
            // for when a part is begun and not touched until the part before theirs is (an `& 0xffff` on the spot would wait for the load)
            uint32_t p = (uint32_t)wave;
            auto bounds_at = [&](uint32_t pp, uint64_t &ro, uint32_t &rl) {
                const uint64_t sp = ((((uint64_t)g << part_bits) + min(pp, P - 1u)) << bb) + ((uint64_t)c << b2) + f;      // (clamped: every wave asks)
                ro = roff[sp];
                rl = rlen[sp];
            };
            uint64_t r0, r0_n, r0_f;
            uint32_t rl, rl_n, rl_f;
            bounds_at(p, r0, rl);
            bounds_at(p + WAVES, r0_n, rl_n);
            uint32_t nr = p < P ? rl & 0xffffu : 0u;
            // A segment is one or two chunks of 64 records, and a chunk asked for where it is needed is a round trip in front of ~1 us of
            // work: the NEXT chunk -- of this part, or the first of the wave's next part -- is asked for before the current one is staged.
            // The request is unconditional (index clamped; which lanes hold a record is decided at use): a load under `lane < nc` is
            // followed by a select, and the select by a wait for the load it has just issued.
            ulonglong2 ahead = recs[r0 + min((uint32_t)lane, nr ? nr - 1u : 0u)];
            uint32_t ahead_p = p, ahead_c0 = 0;                    // what `ahead` holds: chunk ahead_c0 of part ahead_p
            while (p < P) {
                bounds_at(p + 2 * WAVES, r0_f, rl_f);
                const uint32_t nr_n = p + WAVES < P ? rl_n & 0xffffu : 0u;
                for (uint32_t c0 = 0; c0 < nr; c0 += CH) {
                    const uint32_t nc = min(CH, nr - c0);
                    ulonglong2 rec = ahead;
                    if (ahead_p != p || ahead_c0 != c0) rec = recs[r0 + c0 + min((uint32_t)lane, nc - 1u)];      // (not asked for ahead: after an empty part)
                    if (c0 + CH < nr) {
                        ahead = recs[r0 + c0 + CH + min((uint32_t)lane, nr - c0 - CH - 1u)];
                        ahead_p = p; ahead_c0 = c0 + CH;
                    } else {
                        ahead = recs[r0_n + min((uint32_t)lane, nr_n ? nr_n - 1u : 0u)];
                        ahead_p = p + WAVES; ahead_c0 = 0;
                    }
                    if ((uint32_t)lane >= nc) rec = make_ulonglong2(0, 0);
                    const uint32_t tot = record_chunk_stage(rec, nc, srec, sstart, starts, firstrec);
                    for (uint32_t q0 = 0; q0 < tot; q0 += 256) {
                        uint64_t kv[4];
                        const uint32_t nj = min(4u, (tot - q0 + 63u) >> 6);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if ((uint32_t)j >= nj) break;
                            const uint32_t q = q0 + 64u * j + lane;
                            kv[j] = q < tot ? record_chunk_kmer(q, k, srec, sstart, starts, firstrec) : EMPTY_KEY;
                        }
                        ok &= record_insert4<BIG_LOG2, true>(tk_all, tc_all, dl_all, &big_nd, kv, nj, nd);
                    }
                    wave_lds_fence();
                }
                p += WAVES; r0 = r0_n; nr = nr_n; r0_n = r0_f; rl_n = rl_f;
            }
            if (!ok && lane == 0) atomicExch(&big_fail, 1u);
            __syncthreads();
            const bool failed = big_fail != 0;                  // uniform
            const uint32_t nd_all = min(big_nd, BIG);
            const uint64_t dst = region * kstride + kstart[f];
            uint32_t base = 0;
            for (uint32_t p0 = 0; p0 < nd_all; p0 += THREADS) {
                const uint32_t pp = p0 + threadIdx.x;
                const uint32_t slot = pp < nd_all ? dl_all[pp] : 0u;
                const uint64_t key = tk_all[slot];
                const uint32_t cn = tc_all[slot];
                const bool keep = !failed && pp < nd_all && cn >= abundance_min;
                uint32_t sweep_total;
                const uint32_t pos = sweep_compact(keep, scratch, &sweep_total);
                if (keep) {
                    keys[dst + base + pos] = key;
                    if (counts_out) counts_out[dst + base + pos] = cn;
                }
                base += sweep_total;
            }
            __syncthreads();
            for (uint32_t pp = threadIdx.x; pp < nd_all; pp += THREADS) {
                const uint32_t slot = dl_all[pp];
                tk_all[slot] = EMPTY_KEY;
                tc_all[slot] = 0;
            }
            if (threadIdx.x == 0) {
                klen[seg0 + f] = base;
                if (failed) atomicExch(overflow, 1);
                big_fail = 0;
                big_nd = 0;
            }
            __syncthreads();
        }
    }
}

// The counting stage's main launch over the record form: as record_dedup_kernel (which now takes what this one leaves), with the
// table cut down to ONE 32-bit word per slot.  The insert chain of the 64-bit table -- read the key, claim it, walk on, count,
// list the slot, read key and count back -- ran at 33 of record_dedup's 50 ms with neither the VALU (56 %) nor the LDS (40 %)
// busy: 16 waves per CU (37 KB of LDS per workgroup) waiting for one another's LDS round trips.  Here a wave's table is 2 KB:
//   slot = count << 20 | fingerprint << 9 | (number of the k-mer inside its segment + 1)          0 = empty
// A k-mer claims its home slot with ONE 32-bit compare-and-swap; a slot held by another k-mer with another fingerprint (11 more
// bits of the hash) is passed by, one with the same fingerprint names its k-mer, which is cut out of the staged records again and
// compared (a true repeat, or one in 2048 of the others).  The keys stay in registers (up to 7 per lane: a wave's segments have up
// to 448 k-mers in up to 64 records), so the pass that writes the segment out reads one word per claimed slot and zeroes it.
// Segments beyond that are left to record_dedup_kernel (region_big).
template <int NW>
__device__ __forceinline__ uint32_t record_stage_small(ulonglong2 rec, uint32_t nr, ulonglong2 *srec, uint16_t *sstart, uint64_t *starts, uint8_t *firstrec)
{
    const int lane = lane_id();
    const uint32_t ln = (uint32_t)lane < nr ? run_len(rec.y) : 0u;
    const uint32_t incl = wave_scan_incl_dpp(ln), st = incl - ln;
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    srec[lane] = rec;
    sstart[lane] = (uint16_t)st;
    if (lane < NW) { starts[lane] = 0; firstrec[lane] = (uint8_t)nr; }
    wave_lds_fence();
    const uint32_t w = st >> 6;
    const uint32_t w_prev = __shfl_up(w, 1);
    if (ln) {
        atomicOr((unsigned long long *)&starts[w], 1ull << (st & 63u));
        if (lane == 0 || w_prev != w) firstrec[w] = (uint8_t)lane;
    }
    wave_lds_fence();
    return tot;
}

template <int CAP_LOG2, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 5) void record_count_kernel(
    const ulonglong2 *__restrict__ recs, uint32_t rstride, const uint32_t *__restrict__ rcount, uint64_t n_regions, int k, int b2,
    uint64_t kstride, uint32_t abundance_min,
    uint64_t *__restrict__ keys, uint32_t *__restrict__ counts_out, uint64_t *__restrict__ koff, uint32_t *__restrict__ klen,
    int *__restrict__ overflow, uint8_t *__restrict__ region_big, int *__restrict__ any_big)
{
    constexpr uint32_t CAP = 1u << CAP_LOG2, MASK = CAP - 1, MAX_FILL = CAP - CAP / 8;
    constexpr int KJ = (int)((MAX_FILL + 63) / 64);                 // keys a lane holds
    constexpr int NW = KJ + 1;                                      // bitmap words of a segment's k-mer positions
    constexpr uint32_t ID_BITS = 9, FP_BITS = 11, FP_MASK = ((1u << FP_BITS) - 1) << ID_BITS, ONE = 1u << (ID_BITS + FP_BITS);
    static_assert(MAX_FILL + 1 <= (1u << ID_BITS), "a k-mer's number inside its segment fits the slot word");
    constexpr int NF = 1 << RUN_FINE_BITS;
    constexpr uint32_t THREADS = WAVES * 64;
    __shared__ uint32_t tab_all[WAVES][CAP];
    __shared__ ulonglong2 srec_all[WAVES][64];
    __shared__ uint16_t sstart_all[WAVES][64];
    __shared__ uint64_t starts_all[WAVES][NW];
    __shared__ uint8_t firstrec_all[WAVES][NW + (8 - NW % 8)];
    __shared__ uint32_t kcount[NF], kstart[NF], rcnt[NF], rstart[NF];
    __shared__ uint16_t small_list[NF];
    __shared__ uint32_t scratch[32];
    __shared__ uint64_t scratch64[32];
    const int lane = lane_id(), wave = wave_id();
    const uint32_t B2 = 1u << b2;
    uint32_t *tab = tab_all[wave];
    ulonglong2 *srec = srec_all[wave];
    uint16_t *sstart = sstart_all[wave];
    uint64_t *starts = starts_all[wave];
    uint8_t *firstrec = firstrec_all[wave];
    for (uint32_t i = threadIdx.x; i < WAVES * CAP; i += THREADS) (&tab_all[0][0])[i] = 0;
    __syncthreads();
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        const uint32_t n = min(rcount[region], rstride);
        const ulonglong2 *rr = recs + region * rstride;
        const uint64_t seg0 = region << b2;
        // ---- pass A: records and k-mers per fine bucket -> record segments, key segments ----
        if (threadIdx.x < NF) { kcount[threadIdx.x] = 0; rcnt[threadIdx.x] = 0; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
            const uint64_t y = rr[i].y;
            const uint32_t f = run_fine(y) >> (RUN_FINE_BITS - b2);
            atomicAdd(&kcount[f], run_len(y));
            atomicAdd(&rcnt[f], 1u);
        }
        __syncthreads();
        const uint32_t cnt = threadIdx.x < B2 ? kcount[threadIdx.x] : 0u;
        const uint32_t rc = threadIdx.x < B2 ? rcnt[threadIdx.x] : 0u;
        uint64_t both_total;
        const uint64_t both = block_scan_sum64(((uint64_t)rc << 32) | cnt, scratch64, &both_total);
        const uint32_t pre = (uint32_t)both, region_keys = (uint32_t)both_total;
        const bool fits = region_keys <= kstride;                 // uniform
        const bool is_small = threadIdx.x < B2 && cnt <= MAX_FILL && rc <= 64u;
        uint32_t n_small;
        const uint32_t ps = block_scan_sum(is_small ? 1u : 0u, scratch, &n_small);
        if (threadIdx.x < B2) {
            kstart[threadIdx.x] = pre;
            rstart[threadIdx.x] = (uint32_t)(both >> 32);
            koff[seg0 + threadIdx.x] = region * kstride + pre;
            if (!fits || !is_small) klen[seg0 + threadIdx.x] = 0;
            if (is_small) small_list[ps] = (uint16_t)threadIdx.x;
        }
        if (threadIdx.x == 0) {
            const bool big = fits && n_small < B2;
            region_big[region] = big ? 1 : 0;
            if (big) atomicExch(any_big, 1);
            if (!fits) atomicExch(overflow, 1);
        }
        __syncthreads();
        if (fits) {
            // (a segment's records are asked for two segments ahead)
            uint32_t idx = (uint32_t)wave;
            uint32_t f = idx < n_small ? small_list[idx] : 0u;
            uint32_t nr = idx < n_small ? rcnt[f] : 0u;
            ulonglong2 first = (uint32_t)lane < nr ? rr[rstart[f] + lane] : make_ulonglong2(0, 0);
            uint32_t f_n = idx + WAVES < n_small ? small_list[idx + WAVES] : 0u;
            uint32_t nr_n = idx + WAVES < n_small ? rcnt[f_n] : 0u;
            ulonglong2 second = (uint32_t)lane < nr_n ? rr[rstart[f_n] + lane] : make_ulonglong2(0, 0);
            while (idx < n_small) {
                const uint32_t tot = record_stage_small<NW>(first, nr, srec, sstart, starts, firstrec);
                const uint32_t nj = (tot + 63u) >> 6;
                const uint32_t idx_2 = idx + 2 * WAVES;
                const uint32_t f_2 = idx_2 < n_small ? small_list[idx_2] : 0u;
                const uint32_t nr_2 = idx_2 < n_small ? rcnt[f_2] : 0u;
                first = second;
                second = (uint32_t)lane < nr_2 ? rr[rstart[f_2] + lane] : make_ulonglong2(0, 0);
                uint64_t kv[KJ];
                uint32_t sl[KJ], old[KJ];
                uint32_t fresh = 0;
                // the claims of the home slots: all in flight together
#pragma unroll
                for (int j = 0; j < KJ; j++) {
                    if ((uint32_t)j >= nj) break;
                    const uint32_t q = 64u * j + lane;
                    kv[j] = q < tot ? record_chunk_kmer(q, k, srec, sstart, starts, firstrec) : EMPTY_KEY;
                    const uint32_t h = record_slot_hash(kv[j]);
                    sl[j] = h >> (32 - CAP_LOG2);
                    const uint32_t nw = ONE | ((h >> (32 - CAP_LOG2 - FP_BITS)) << ID_BITS & FP_MASK) | (q + 1u);
                    old[j] = q < tot ? atomicCAS(&tab[sl[j]], 0u, nw) : 0xffffffffu;
                }
#pragma unroll
                for (int j = 0; j < KJ; j++) {
                    if ((uint32_t)j >= nj) break;
                    if (old[j] != 0xffffffffu) {
                        uint32_t o = old[j], at = sl[j];
                        bool mine = true;
                        // (the slot word again -- only a k-mer that found its home slot taken comes here)
                        const uint32_t nw = o == 0u ? 0u
                                                    : ONE | ((record_slot_hash(kv[j]) >> (32 - CAP_LOG2 - FP_BITS)) << ID_BITS & FP_MASK) | (64u * j + lane + 1u);
                        while (o != 0u) {
                            if (((o ^ nw) & FP_MASK) == 0u) {             // the same fingerprint: the same k-mer?
                                const uint64_t other = record_chunk_kmer((o & ((1u << ID_BITS) - 1u)) - 1u, k, srec, sstart, starts, firstrec);
                                if (other == kv[j]) {
                                    atomicAdd(&tab[at], ONE);       // (a count stays below the segment's <= 448 k-mers: 12 bits hold it)
                                    mine = false;
                                    break;
                                }
                            }
                            at = (at + 1) & MASK;
                            o = atomicCAS(&tab[at], 0u, nw);
                        }
                        if (mine) { fresh |= 1u << j; sl[j] = at; }
                    }
                }
                wave_lds_fence();
                // the segment leaves: claimed slots in the order of the k-mers, each left empty
                const uint64_t dst = region * kstride + kstart[f];
                uint32_t base = 0;
#pragma unroll
                for (int j = 0; j < KJ; j++) {
                    if ((uint32_t)j >= nj) break;
                    const bool fr = (fresh >> j) & 1u;
                    uint32_t c = 0;
                    if (fr) {
                        c = tab[sl[j]] >> (ID_BITS + FP_BITS);
                        tab[sl[j]] = 0;
                    }
                    const bool keep = fr && c >= abundance_min;
                    const uint64_t m = __ballot(keep);
                    if (keep) {
                        const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        keys[dst + pos] = kv[j];
                        if (counts_out) counts_out[dst + pos] = c;
                    }
                    base += (uint32_t)__popcll(m);
                }
                if (lane == 0) klen[seg0 + f] = base;
                wave_lds_fence();
                idx += WAVES;
                f = f_n; nr = nr_n;
                f_n = f_2; nr_n = nr_2;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// Stage 3a (K5): per-bucket dictionary AND presence bits.  One workgroup per (bucket, sub-bucket)
// unions the bucket's segment of EVERY genome in an LDS table; a wave takes one genome at a time,
// all waves stay inside one word-row (64 genomes) between two barriers.
//   tkeys[slot]  the k-mer (linear probing, 64-bit ds_cmpst)
//   words[slot]  presence bits of the CURRENT word-row: a key that is found ORs its genome's bit in
//                (ds_or_b64, no return) -- the whole per-occurrence work of a pan-genome
//   meta[slot]   entry id (insertion order inside the workgroup) | SEEN (had a bit in an earlier row)
//                | MULTI (carried by several genomes)
// At the end of a row the words of all occupied slots go to matrix_s[wg][row][entry id] (a dense,
// contiguous run per workgroup and row) and are cleared.  The per-occurrence 2-byte slot ids of the
// earlier design (11 GB written here, 11 GB read by the fill, per 1000 genomes) no longer exist:
// what leaves the kernel is U-sized.  "one genome / several" comes from the bits themselves.
// ------------------------------------------------------------------------------------
constexpr uint32_t META_ID = 0x1fffu, META_SEEN = 0x4000u, META_MULTI = 0x8000u;

// Pair-aligned linear probing: a key's probe sequence starts at the EVEN slot (h & mask & ~1), so that one
// 16-byte LDS read (ds_read_b128) returns two consecutive slots of it and two reads cover a displacement of
// up to 3 -- at the table's load that settles all but a few keys per thousand in the straight-line part.
// returns the slot of key (inserting it if absent), or 0xffffffff when the table is full
__device__ __forceinline__ uint32_t lds_pair_find_or_insert(uint64_t *tkeys, uint32_t cap_mask, uint64_t key, uint64_t h,
                                                            bool *inserted)
{
    uint32_t slot = hash_slot(h, cap_mask) & ~1u;
    for (uint32_t probe = 0; probe <= cap_mask; probe++) {
        uint64_t cur = lds_peek(&tkeys[slot]);
        if (cur == EMPTY_KEY)
            cur = atomicCAS((unsigned long long *)&tkeys[slot], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (cur == EMPTY_KEY) { *inserted = true; return slot; }
        if (cur == key) { *inserted = false; return slot; }
        slot = (slot + 1) & cap_mask;
    }
    return 0xffffffffu;
}

struct DictWave {
    uint64_t *tkeys;
    unsigned long long *words;
    uint16_t *meta;
    LdsFlag full;
    uint32_t *n_distinct;
    uint32_t cap_mask, max_fill, cap_log2;
    uint32_t wg, sub, G;
    int bb, sb;
    uint16_t *birth;
    uint32_t *need;
    uint16_t *sid;          // table slot of entry id (record form with the memo: the word-row's end goes by id); nullptr: not kept
    uint32_t *track;        // rank union: table slot of every key of the tracked rank's list, by list position (DictArgs::track); nullptr: none
};

// J keys per lane (EMPTY_KEY = none) against the table: the table is consulted four keys at a time (the 16-byte reads
// of more keys than that at once cost more registers than the LDS latency they would hide).
// several (FLAGS): bit j = key j is carried by several genomes inside its rank
// LIVE: the caller says which of the J keys exist (bit j of `live`) instead of marking the others with EMPTY_KEY
// tpos (FLAGS, rank union): list position of key 0 of a lane of the TRACKED rank (key j: tpos + 64 j), 0xffffffff otherwise -- the
// slot every such key ends in is noted, so that the rank's entries learn their union entry without a search afterwards
template <int J, bool FLAGS, bool LIVE = false>
__device__ __forceinline__ void dict_probe(const DictWave &w, const uint64_t (&kv)[J], uint32_t several, uint32_t g, uint32_t r,
                                           unsigned long long bit, uint32_t live = 0, uint32_t tpos = 0xffffffffu)
{
    uint32_t sl[J];
    uint32_t todo = 0;
    constexpr int GRP = J < 4 ? J : 4;
#pragma unroll
    for (int j0 = 0; j0 < J; j0 += GRP) {
        ulonglong2 p0[GRP], p1[GRP];
        uint32_t hs[GRP];
#pragma unroll
        for (int q = 0; q < GRP; q++) {
            const uint64_t h = mix64(kv[j0 + q]);
            sl[j0 + q] = hash_slot(h, w.cap_mask) & ~1u;
            hs[q] = w.sb ? hash_sub(h, w.bb, w.sb) : 0u;
        }
#pragma unroll
        for (int q = 0; q < GRP; q++) {
            p0[q] = *reinterpret_cast<const ulonglong2 *>(&w.tkeys[sl[j0 + q]]);
            p1[q] = *reinterpret_cast<const ulonglong2 *>(&w.tkeys[(sl[j0 + q] + 2) & w.cap_mask]);
        }
#pragma unroll
        for (int q = 0; q < GRP; q++) {
            const uint64_t key = kv[j0 + q];
            const bool active = (LIVE ? ((live >> (j0 + q)) & 1u) != 0 : key != EMPTY_KEY) && hs[q] == w.sub;
            uint32_t at = sl[j0 + q];
            const bool h0 = p0[q].x == key, h1 = p0[q].y == key, h2 = p1[q].x == key, h3 = p1[q].y == key;
            if (h1) at = sl[j0 + q] + 1;
            if (h2) at = (sl[j0 + q] + 2) & w.cap_mask;
            if (h3) at = (sl[j0 + q] + 3) & w.cap_mask;
            const bool done = h0 | h1 | h2 | h3;
            if (active && done) atomicOr(&w.words[at], (FLAGS && ((several >> (j0 + q)) & 1u)) ? (bit | 1ull) : bit);
            if (FLAGS && active && done && tpos != 0xffffffffu) w.track[tpos + 64u * (uint32_t)(j0 + q)] = at;
            todo |= (uint32_t)(active && !done) << (j0 + q);
        }
    }
    while (todo) {
        if (w.full) break;           // LDS flag: once the table is given up, stop probing it
        const int j = __ffs(todo) - 1;
        todo &= todo - 1;
        uint64_t key = kv[0];
#pragma unroll
        for (int q = 1; q < J; q++) { if (j == q) key = kv[q]; }
        bool ins;
        const uint32_t slot = lds_pair_find_or_insert(w.tkeys, w.cap_mask, key, mix64(key), &ins);
        bool over = slot == 0xffffffffu;
        if (!over) {
            if (ins) {
                const uint32_t id = atomicAdd(w.n_distinct, 1u);
                w.meta[slot] = (uint16_t)(id & META_ID);
                if (w.sid && id <= w.cap_mask) w.sid[id] = (uint16_t)slot;
                if (w.birth && id <= w.cap_mask) w.birth[((uint64_t)w.wg << w.cap_log2) + id] = (uint16_t)r;
                over = id >= w.max_fill;
            }
            atomicOr(&w.words[slot], (FLAGS && ((several >> j) & 1u)) ? (bit | 1ull) : bit);
            if (FLAGS && tpos != 0xffffffffu) w.track[tpos + 64u * (uint32_t)j] = slot;
        }
        if (over) {
            w.full.set(1);       // (what the workgroup would have needed is counted after the word-row loop: dict_build_kernel)
            break;
        }
    }
}

// J x 64 keys of one genome's segment: keys[i0 + 64 j] for j < J (i0 includes the lane), n = segment length
template <int J, bool FLAGS>
__device__ __forceinline__ void dict_chunk(const DictWave &w, const uint64_t *__restrict__ keys, const uint8_t *__restrict__ flg, uint64_t i0,
                                           uint64_t n, uint32_t g, uint32_t r, unsigned long long bit, uint32_t tpos = 0xffffffffu)
{
    uint64_t kv[J];
    uint32_t several = 0;
#pragma unroll
    for (int j = 0; j < J; j++) {
        const uint64_t i = i0 + 64u * j;
        kv[j] = i < n ? keys[i] : EMPTY_KEY;
        if (FLAGS && i < n && flg[i] >= 2) several |= 1u << j;
    }
    // all J global loads are in flight
    dict_probe<J, FLAGS>(w, kv, several, g, r, bit, 0u, tpos == 0xffffffffu ? tpos : tpos + (uint32_t)i0);
}

// ---- record memo (record form) --------------------------------------------------------------------------------------
// Genomes of one species share most of their sequence, and with it most of their run records: the SAME 16 bytes come
// in from most genomes of a word-row.  The workgroup therefore keeps a small table of the distinct records it has met
// (id = order of arrival) with one presence word per record; an occurrence of a held record costs one hash of 16 bytes,
// one 16-byte read of its bucket of four slots, one 16-byte compare and one OR -- instead of decoding its up to 16 k-mers
// and probing the key table for each.  At the end of a word-row the word of every held record goes to its k-mers' words.
// A record the table does not hold (table or bucket full, slot being written) goes the direct way, so the memo is an
// accelerator and never a point of failure; a workgroup whose memo is full and rarely hit (unrelated genomes) switches
// it off.
constexpr uint32_t MEMO_NONE = 0xffffffffu, MEMO_LOCK = 0x0000ffffu;
struct DictMemo {
    uint32_t *slot;                 // buckets of 4 slots: 0 = empty, MEMO_LOCK = being written, else tag << 16 | id + 1
    ulonglong2 *rec;                // [n_ent] the records, by id
    unsigned long long *words;      // [n_ent] presence word of the current word-row
    uint32_t *kslot;                // [12 n_ent] table slots of the record's up to 22 k-mers (16 bits each, 0xffff = none; 24 per record), once resolved
    uint32_t *ctl;                  // [0] records held, [1] memo in use, [2] occurrences that went the direct way, [3] occurrences (of
                                    // the row), [4] records whose k-mer slots are resolved (ids below it)
    uint32_t n_ent, bmask;          // bmask: buckets - 1
};
__device__ __forceinline__ uint32_t memo_hash(uint64_t x, uint64_t y)
{
    const uint32_t h = mul24((uint32_t)x, 0x9E3779u) + mul24((uint32_t)(x >> 24), 0x85EBCBu) + mul24((uint32_t)(x >> 48), 0xC2B2AFu) +
                       mul24((uint32_t)(y >> 40), 0xD6E8FFu) + mul24((uint32_t)(y >> 16), 0xA54FF5u) + mul24((uint32_t)y & 0xffffu, 0x3C6EF3u);
    return h ^ (h >> 13);
}
// The common case, straight-line: the record's bucket holds it.  ORs `bit` into its word and returns true then.
__device__ __forceinline__ bool memo_hit(const DictMemo &M, uint64_t x, uint64_t y, unsigned long long bit)
{
    const uint32_t h = memo_hash(x, y);
    const uint32_t tag = h >> 16;
    const uint4 v = *reinterpret_cast<const uint4 *>(&M.slot[(h & M.bmask) << 2]);
    uint32_t e = 0;
    e = (v.w >> 16) == tag ? v.w : e;
    e = (v.z >> 16) == tag ? v.z : e;
    e = (v.y >> 16) == tag ? v.y : e;
    e = (v.x >> 16) == tag ? v.x : e;
    const uint32_t idp = e & 0xffffu;                   // id + 1 of the first slot with the tag (0: an empty slot or none; 0xffff: being written)
    bool ok = idp != 0u && idp != 0xffffu;
    const uint32_t id = ok ? idp - 1u : 0u;
    const ulonglong2 held = M.rec[id];
    ok = ok && held.x == x && held.y == y;
    if (ok) atomicOr(&M.words[id], bit);
    return ok;
}
// The rest (first occurrence of a record, a slot being written, a second slot with the same tag): id of the record, entering
// it when bucket and memo have room; MEMO_NONE: not held, take the direct way
__device__ __forceinline__ uint32_t memo_find_or_insert(const DictMemo &M, uint64_t x, uint64_t y)
{
    const uint32_t h = memo_hash(x, y);
    const uint32_t tag = h >> 16;
    // (its bucket, then the next one: a record that lives there is found here at every occurrence -- rare, and far cheaper
    // than the direct way)
    uint32_t *bucket = &M.slot[(h & M.bmask) << 2];
    uint32_t *const bucket2 = &M.slot[((h + 1u) & M.bmask) << 2];
    for (int p = 0; p < 8; p++) {
        bool again = false;
        for (int q = 0; q < 4; q++) {
            // (relaxed: the record is read through the id the slot word carries, and an acquire would also wait for the global
            // loads in flight -- the next records)
            uint32_t v = lds_peek(&bucket[q]);
            if (v == 0) {
                if (lds_peek(&M.ctl[0]) >= M.n_ent) return MEMO_NONE;
                v = atomicCAS(&bucket[q], 0u, MEMO_LOCK);
                if (v == 0) {
                    const uint32_t id = atomicAdd(&M.ctl[0], 1u);
                    if (id >= M.n_ent) return MEMO_NONE;          // (the slot stays locked: whoever reaches it goes the direct way)
                    M.rec[id] = make_ulonglong2(x, y);
                    // the record before the slot word that publishes it: LDS only (a plain release would drain the global loads too)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __hip_atomic_store(&bucket[q], (tag << 16) | (id + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    return id;
                }
            }
            if (v == MEMO_LOCK) {           // being written (often by a lane of this wave with the same record): look again
                again = true;
                break;
            }
            if ((v >> 16) == tag) {
                const uint32_t id = (v & 0xffffu) - 1u;
                const ulonglong2 e = M.rec[id];
                if (e.x == x && e.y == y) return id;
            }
        }
        if (!again) {
            if (bucket == bucket2) return MEMO_NONE;       // eight other records
            bucket = bucket2;
        }
    }
    return MEMO_NONE;
}
// slot of `key` in the workgroup's table, entering it if absent (entry id, birth row, fill limit as in dict_probe);
// 0xffffffff: the table is full (the flag is set)
__device__ __forceinline__ uint32_t dict_slot_of(const DictWave &w, uint64_t key, uint32_t r)
{
    bool ins;
    const uint32_t slot = lds_pair_find_or_insert(w.tkeys, w.cap_mask, key, mix64(key), &ins);
    bool over = slot == 0xffffffffu;
    if (!over && ins) {
        const uint32_t id = atomicAdd(w.n_distinct, 1u);
        w.meta[slot] = (uint16_t)(id & META_ID);
        if (w.sid && id <= w.cap_mask) w.sid[id] = (uint16_t)slot;
        if (w.birth && id <= w.cap_mask) w.birth[((uint64_t)w.wg << w.cap_log2) + id] = (uint16_t)r;
        over = id >= w.max_fill;
    }
    if (over) w.full.set(1);
    return slot;
}
// End of a word-row: the word of every held record goes to the words of its k-mers.  The table slots of a record's k-mers
// are resolved when the record is met here for the first time (decoded, every k-mer of this sub-bucket found or entered)
// and kept, 16 bits each: afterwards a record costs three 4-byte reads and up to 6 ORs per lane and word-row.
__device__ __forceinline__ void memo_flush_row(const DictWave &w, const DictMemo &M, uint32_t r, int kk, uint64_t kmask, int rcshift)
{
    const uint32_t held = min(M.ctl[0], M.n_ent), known = M.ctl[4];
    // records met for the first time in this word-row: one lane resolves one record
    for (uint32_t t = known + threadIdx.x; t < held; t += blockDim.x) {
        unsigned long long k0 = ~0ull, k1 = ~0ull, k2 = ~0ull, k3 = ~0ull, k4 = ~0ull, k5 = ~0ull;   // (scalars: an indexed array would live in scratch)
        const ulonglong2 rec = M.rec[t];
        const uint32_t len = run_len(rec.y);
        RunDecoder dec = run_open(rec.x, rec.y, kk);
        for (uint32_t tt = 0; tt < len; tt++) {
            if (w.full) break;
            const uint64_t key = run_canonical(dec);
            run_next(dec, kmask, rcshift);
            if (w.sb && hash_sub(mix64(key), w.bb, w.sb) != w.sub) continue;
            const uint32_t slot = dict_slot_of(w, key, r);
            if (slot == 0xffffffffu) break;
            const unsigned long long put = ~((unsigned long long)(0xffffu ^ slot) << (16 * (tt & 3)));
            const uint32_t qd = tt >> 2;
            k0 &= qd == 0u ? put : ~0ull;
            k1 &= qd == 1u ? put : ~0ull;
            k2 &= qd == 2u ? put : ~0ull;
            k3 &= qd == 3u ? put : ~0ull;
            k4 &= qd == 4u ? put : ~0ull;
            k5 &= qd == 5u ? put : ~0ull;
        }
        ulonglong2 *ks = reinterpret_cast<ulonglong2 *>(M.kslot + 12 * t);
        ks[0] = make_ulonglong2(k0, k1);
        ks[1] = make_ulonglong2(k2, k3);
        ks[2] = make_ulonglong2(k4, k5);
    }
    if (held > known) lds_barrier();         // (uniform: both read between the caller's barriers)
    // every held record: four lanes (of one wave) take its word to the slots of six of its k-mers each
    for (uint32_t t4 = threadIdx.x; t4 < 4u * held; t4 += blockDim.x) {
        const uint32_t t = t4 >> 2, qr = t4 & 3u;
        const unsigned long long wd = M.words[t];
        if (!wd) continue;
        const uint32_t *ks = M.kslot + 12 * t + 3 * qr;
        const uint32_t s0 = ks[0], s1 = ks[1], s2 = ks[2];
        if (qr == 0) M.words[t] = 0;             // (its partner lanes have read the word: same instruction, same wave)
#pragma unroll
        for (int tt = 0; tt < 6; tt++) {
            const uint32_t slot = ((tt < 2 ? s0 : tt < 4 ? s1 : s2) >> (16 * (tt & 1))) & 0xffffu;
            if (slot != 0xffffu) atomicOr(&w.words[slot], wd);
        }
    }
}

// one record per lane (rec.y == 0: none), genome bit `bit` of word-row r: through the memo, or J keys to the table
template <int J>
__device__ __forceinline__ void dict_take_record(const DictWave &w, const DictMemo &M, bool memo_on, const ulonglong2 rec, unsigned long long bit,
                                                 uint32_t r, int kk, uint64_t kmask, int rcshift)
{
    bool direct = rec.y != 0;
    if (memo_on) {
        if (direct && memo_hit(M, rec.x, rec.y, bit)) direct = false;
        if (!__ballot(direct)) return;
        const uint32_t id = direct ? memo_find_or_insert(M, rec.x, rec.y) : MEMO_NONE;
        if (id != MEMO_NONE) {
            atomicOr(&M.words[id], bit);
            direct = false;
        }
        const unsigned long long left = __ballot(direct);
        if (!left) return;
        if (lane_id() == 0) atomicAdd(&M.ctl[2], (uint32_t)__popcll(left));
    }
    const uint32_t len = run_len(rec.y);
    RunDecoder dec = run_open(rec.x, rec.y, kk);
    uint64_t kv[J];
    kv[0] = run_canonical(dec);
#pragma unroll
    for (int t = 1; t < J; t++) {
        run_next(dec, kmask, rcshift);
        kv[t] = run_canonical(dec);          // (past the record's last k-mer: not live)
    }
    dict_probe<J, false, true>(w, kv, 0u, 0u, r, bit, direct ? (1u << len) - 1u : 0u);
    if (J == 8) {
        // a record holds up to 22 k-mers: the second and third eight, where a lane of the wave has them
#pragma unroll 1
        for (uint32_t done = 8; done < (uint32_t)RUN_LMAX; done += 8) {
            const uint32_t live2 = direct ? ((1u << len) - 1u) >> done : 0u;
            if (!__ballot(live2 != 0)) return;
            uint64_t kv2[J];
#pragma unroll
            for (int t = 0; t < J; t++) {
                run_next(dec, kmask, rcshift);
                kv2[t] = run_canonical(dec);
            }
            dict_probe<J, false, true>(w, kv2, 0u, 0u, r, bit, live2 & 0xffu);
        }
    }
}

// MAXT: largest workgroup the instance is launched with (the 8-deep form needs more than the 128 VGPRs a
// 1024-thread workgroup leaves per lane)
template <int KIF, int MAXT, bool FLAGS, bool REC>
__global__ __launch_bounds__(MAXT, 4) void dict_build_kernel(const DictArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t cap = 1u << a.cap_log2, cap_mask = cap - 1;
    uint64_t *tkeys = reinterpret_cast<uint64_t *>(lds_raw);
    unsigned long long *words = reinterpret_cast<unsigned long long *>(lds_raw + (size_t)cap * 8);
    uint16_t *meta = reinterpret_cast<uint16_t *>(lds_raw + (size_t)cap * 16);
    uint32_t *scratch = reinterpret_cast<uint32_t *>(lds_raw + (size_t)cap * 18);   // [16] + flags
    const LdsFlag full = {reinterpret_cast<int *>(scratch + 16)};
    uint32_t &n_distinct = scratch[17];
    uint64_t &out_base = *reinterpret_cast<uint64_t *>(scratch + 18);
    const uint32_t wg = blockIdx.x;
    const uint32_t B = 1u << a.bb;
    const int sb = a.sb;
    const uint32_t b = wg >> sb;
    const uint32_t G = a.n_genomes, n_rows = (G + 63) >> 6;
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) { tkeys[i] = EMPTY_KEY; words[i] = 0; meta[i] = 0; }
    if (threadIdx.x == 0) { full.set(0); n_distinct = 0; }
    __syncthreads();
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;     // nw divides 64 (launcher)
    // a genome may come in 2^part_bits parts ("virtual genomes" g, real genome g >> pb; record form of the partition)
    const int pb = a.part_bits;
    const uint32_t GV = G << pb;
    const uint32_t per_row = (64u << pb) / (uint32_t)nw;    // (virtual) genomes per wave per word-row
    DictWave w;
    w.tkeys = tkeys; w.words = words; w.meta = meta; w.full = full; w.n_distinct = &n_distinct;
    w.cap_mask = cap_mask; w.max_fill = cap - (cap >> 3); w.cap_log2 = a.cap_log2;
    w.wg = wg; w.sub = wg & ((1u << sb) - 1); w.G = G; w.bb = a.bb; w.sb = sb; w.birth = a.birth; w.need = a.need; w.sid = nullptr;
    w.track = FLAGS ? a.track : nullptr;
    // segment of the NEXT genome is fetched while the current one is processed (the two dependent
    // global round trips -- bounds, then keys -- would otherwise serialise per genome).  Wave w takes
    // genomes w, w + nw, ...: since nw divides 64 that sequence walks the word-rows in step with the
    // other waves.
    uint32_t g = (uint32_t)wave;
    uint64_t s0 = 0, n = 0, f0 = 0;
    if (g < GV) {
        seg_bounds(a.seg, (uint64_t)g * B + b, s0, n);
        if (FLAGS) f0 = a.in_flag_off[(uint64_t)g * B + b];
    }
    // record form: decoding constants, and a wave-private table of the 16 sub-segments (8 genomes x 2 length classes) a wave pools
    const int kk = REC ? a.k : 1;
    const int rcshift = 2 * (kk - 1);
    const uint64_t kmask = kk == 32 ? ~0ull : ((1ull << (2 * kk)) - 1);
    uint64_t *tabD = reinterpret_cast<uint64_t *>(lds_raw + (size_t)cap * 18 + TABLE_SCRATCH_BYTES) + 16 * wave;
    uint32_t *tabS = reinterpret_cast<uint32_t *>(lds_raw + (size_t)cap * 18 + TABLE_SCRATCH_BYTES + (size_t)nw * 128) + 16 * wave;
    // record memo (REC, a.memo_log2 > 0): 15/32 * 2^memo_log2 records, twice 2^memo_log2 slots, behind the waves' pool tables
    DictMemo M;
    M.bmask = REC && a.memo_log2 ? (1u << (a.memo_log2 - 1)) - 1u : 0u;          // 2^(memo_log2 + 1) slots in buckets of 4
    M.n_ent = REC && a.memo_log2 ? dict_memo_entries(a.memo_log2) : 0u;
    {
        uint8_t *mb = lds_raw + (size_t)cap * 18 + TABLE_SCRATCH_BYTES + (size_t)nw * 192;
        M.rec = reinterpret_cast<ulonglong2 *>(mb);
        M.kslot = reinterpret_cast<uint32_t *>(mb + (size_t)M.n_ent * 16);
        M.words = reinterpret_cast<unsigned long long *>(mb + (size_t)M.n_ent * 64);
        M.slot = reinterpret_cast<uint32_t *>(mb + (size_t)M.n_ent * 72);
        M.ctl = M.slot + ((size_t)(M.bmask + 1) << 2);
    }
    w.sid = REC && a.memo_log2 ? reinterpret_cast<uint16_t *>(M.ctl + 8) : nullptr;          // [cap]
    if (REC && a.memo_log2) {
        for (uint32_t i = threadIdx.x; i < (M.bmask + 1) << 2; i += blockDim.x) M.slot[i] = 0;
        for (uint32_t i = threadIdx.x; i < M.n_ent; i += blockDim.x) M.words[i] = 0;
        if (threadIdx.x == 0) { M.ctl[0] = 0; M.ctl[1] = 1; M.ctl[2] = 0; M.ctl[3] = 0; M.ctl[4] = 0; }
        __syncthreads();
    }
    // record form: segment bounds of a wave's group of 8 (virtual) genomes -- lane e holds those of genome e & 7
    // (record segments come with offsets AND lengths -- regions leave gaps; read here without seg_bounds' cases, whose merges
    // would make the compiler wait for the loads on the spot)
    auto group_bounds = [&](uint32_t rr, uint32_t j0, uint64_t &sj, uint32_t &nj) {
        const uint32_t jj = (uint32_t)lane & 7u;
        const uint32_t vgj = ((rr * 64u) << pb) + (uint32_t)wave + (j0 + jj) * (uint32_t)nw;
        const bool mine = rr < n_rows && j0 + jj < per_row && vgj < GV;          // (every lane: those of genome lane & 7)
        const uint64_t at = mine ? (uint64_t)vgj * B + b : 0;
        sj = mine ? a.seg.off[at] : 0ull;
        nj = mine ? a.seg.len[at] : 0u;
    };
    uint64_t sj_n = 0;
    uint32_t nj_n = 0;
    if (REC) group_bounds(0, 0, sj_n, nj_n);
    for (uint32_t r = 0; r < n_rows; r++) {
        if (REC) {
            // Record form.  A record holds a run of 1..22 consecutive k-mers with their bases; ONE LANE decodes one record
            // (first k-mer by a shift and a reverse complement, the others by rolling both words) and takes its keys to
            // the table.  A segment holds its records of at most 4 k-mers first: those go four keys at a time, the others
            // eight at a time (one size for all would leave a third of the key slots empty).  The records of the wave's
            // genomes of this word-row are pooled, 8 genomes at a time and class by class, so that the lanes of the last
            // instruction of a genome are not left idle: lane -> (class, genome, record) through the running totals of
            // the 16 pooled sub-segments.
            const bool memo_on = a.memo_log2 && M.ctl[1];          // (changes between the barriers of a row's end only)
            for (uint32_t j0 = 0; j0 < per_row && !full; j0 += 8) {
                const uint32_t jj = (uint32_t)lane & 7u, cls = ((uint32_t)lane >> 3) & 1u;
                const uint32_t vgj = ((r * 64u) << pb) + (uint32_t)wave + (j0 + jj) * (uint32_t)nw;
                // this group's bounds were asked for a group ago; the next group's (the next word-row's first after this row's
                // last) are asked for now.  (The asm makes the wait for sj / nj stand BEFORE the new loads are issued: the compiler
                // waits with vmcnt(0) at the first use, which would otherwise take the new loads with it.)
                const uint64_t sj = sj_n;
                const uint32_t nj = nj_n;
                __asm__ volatile("" ::"v"(sj), "v"(nj));
                if (j0 + 8 < per_row) group_bounds(r, j0 + 8, sj_n, nj_n);
                else group_bounds(r + 1, 0, sj_n, nj_n);
                if (memo_on) {
                    // With the memo an occurrence is a lookup whatever its length: no classes, no pooling.  The 8 lanes
                    // l & 7 == e take the records of genome e, 8 at a time (128 contiguous bytes); four batches are in flight (below).
                    const uint32_t n_mine = nj & 0xffffu;
                    uint32_t n_max = 0;
#pragma unroll
                    for (int e = 0; e < 8; e++) n_max = max(n_max, (uint32_t)__builtin_amdgcn_readlane((int)n_mine, e));
                    const unsigned long long bit = 1ull << (63u - ((vgj >> pb) & 63u));
                    {
                        uint32_t n_sum = 0;
#pragma unroll
                        for (int e = 0; e < 8; e++) n_sum += (uint32_t)__builtin_amdgcn_readlane((int)n_mine, e);
                        if (lane == 0) atomicAdd(&M.ctl[3], n_sum);
                    }
                    // Four batches of 64 records are in flight per wave (the reads are 128-byte pieces of segments that lie
                    // megabytes apart: latency, not bandwidth, is what they cost).  The loads are unconditional (index clamped,
                    // validity applied at use) and the batches sit in four fixed registers, so that the wait for one batch
                    // is a counted s_waitcnt that leaves the younger three in flight.
                    const ulonglong2 *seg = a.recs + sj;
                    const uint32_t i_mine = (uint32_t)lane >> 3, n_last = n_mine ? n_mine - 1u : 0u;
                    ulonglong2 q[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) q[u] = seg[min(i_mine + 8u * u, n_last)];
                    // (the four steps written out: left to the unroller, a body this size stays a loop and q[] goes to scratch)
                    auto step = [&](ulonglong2 &qu, uint32_t at) {
                        ulonglong2 rec = qu;
                        __asm__ volatile("" ::"v"(rec.x), "v"(rec.y));      // (the wait stands here, before the next load is issued)
                        qu = seg[min(i_mine + at + 32u, n_last)];
                        if (i_mine + at >= n_mine) rec.y = 0;
                        dict_take_record<8>(w, M, true, rec, bit, r, kk, kmask, rcshift);
                    };
                    for (uint32_t i = 0; i < n_max && !full; i += 32) {
                        step(q[0], i);
                        if (i + 8u < n_max) step(q[1], i + 8u);
                        if (i + 16u < n_max) step(q[2], i + 16u);
                        if (i + 24u < n_max) step(q[3], i + 24u);
                    }
                    continue;
                }
                const uint32_t n_all = lane < 16 ? nj & 0xffffu : 0u, n_short = lane < 16 ? nj >> 16 : 0u;
                const uint32_t n32 = cls ? n_all - n_short : n_short;
                const uint32_t inc = wave_scan_incl_dpp(n32);
                const uint32_t cume = inc - n32;
                const uint32_t N = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                if (lane < 16) {
                    tabD[lane] = sj + (cls ? n_short : 0u) - cume;
                    tabS[lane] = 63u - ((vgj >> pb) & 63u);
                }
                uint32_t th[16];
#pragma unroll
                for (int e = 1; e < 16; e++) th[e] = (uint32_t)__builtin_amdgcn_readlane((int)cume, e);
                const uint32_t NA = th[8];           // records of the short class in the pool
                auto fetch = [&](uint32_t q, uint32_t q_end, int e0, ulonglong2 &rec, uint32_t &sh) {
                    uint32_t e = (uint32_t)e0;
#pragma unroll
                    for (int i = 1; i < 8; i++) e += (uint32_t)(q >= th[e0 + i]);
                    rec = q < q_end ? a.recs[tabD[e] + q] : make_ulonglong2(0, 0);
                    sh = tabS[e];
                };
                // One batch of 64 records is always in flight while the batch before it is worked on -- across the two classes
                // as well: the first long records are asked for before the short ones are gone through.
                ulonglong2 rec_n, rec_l;
                uint32_t sh_n, sh_l;
                fetch((uint32_t)lane, NA, 0, rec_n, sh_n);
                fetch(NA + (uint32_t)lane, N, 8, rec_l, sh_l);
                // ---- the short records: four keys per lane ----
                for (uint32_t q0 = 0; q0 < NA && !full; q0 += 64) {
                    const ulonglong2 rec = rec_n;
                    const uint32_t sh = sh_n;
                    __asm__ volatile("" ::"v"(rec.x), "v"(rec.y));      // (as above: wait for this batch, then ask for the next)
                    fetch(q0 + 64u + (uint32_t)lane, NA, 0, rec_n, sh_n);
                    dict_take_record<4>(w, M, false, rec, 1ull << sh, r, kk, kmask, rcshift);
                }
                // ---- the long ones: eight ----
                rec_n = rec_l;
                sh_n = sh_l;
                for (uint32_t q0 = NA; q0 < N && !full; q0 += 64) {
                    const ulonglong2 rec = rec_n;
                    const uint32_t sh = sh_n;
                    __asm__ volatile("" ::"v"(rec.x), "v"(rec.y));
                    fetch(q0 + 64u + (uint32_t)lane, N, 8, rec_n, sh_n);
                    dict_take_record<8>(w, M, false, rec, 1ull << sh, r, kk, kmask, rcshift);
                }
            }
        } else {
        for (uint32_t jr = 0; jr < per_row; jr++, g += nw) {
                uint64_t s0_next = 0, n_next = 0, f0_next = 0;
                if (g + nw < GV) {
                    seg_bounds(a.seg, (uint64_t)(g + nw) * B + b, s0_next, n_next);
                    if (FLAGS) f0_next = a.in_flag_off[(uint64_t)(g + nw) * B + b];
                }
                if (g < GV && !full) {
                    const uint32_t gr = g >> pb;
                    const unsigned long long bit = 1ull << (63 - (gr & 63));
                    const uint64_t *seg = a.keys + s0;
                    const uint8_t *flg = FLAGS ? a.in_flags + f0 : nullptr;
                    // The kernel is bound by latency and instruction issue, so the common case is straight-line and
                    // wide: KIF keys per lane in flight, their hashes, two 16-byte table reads each; a key found there
                    // -- almost every key of a pan-genome after the first few genomes -- costs one more LDS OR.  The
                    // tail of the segment goes through a copy of the same code that is exactly as deep as it needs.
                    // (rank union: the list position of this segment's first key when this is the tracked rank)
                    const uint32_t tp = (FLAGS && a.track && g == a.track_g) ? (uint32_t)(f0 - a.track_flag_base) : 0xffffffffu;
                    uint64_t i0 = lane;
                    for (; i0 + 64u * (KIF - 1) < n; i0 += 64 * KIF) dict_chunk<KIF, FLAGS>(w, seg, flg, i0, n, gr, r, bit, tp);
                    if (i0 - lane < n && !full) {
                        const uint32_t nj = (uint32_t)((n - (i0 - lane) + 63) >> 6);        // wave-uniform: 1 .. KIF - 1
                        if (KIF > 4 && nj > 4) dict_chunk<(KIF > 4 ? KIF : 1), FLAGS>(w, seg, flg, i0, n, gr, r, bit, tp);
                        else if (KIF > 2 && nj > 2) dict_chunk<(KIF > 2 ? 4 : 1), FLAGS>(w, seg, flg, i0, n, gr, r, bit, tp);
                        else if (KIF > 1 && nj == 2) dict_chunk<(KIF > 1 ? 2 : 1), FLAGS>(w, seg, flg, i0, n, gr, r, bit, tp);
                        else dict_chunk<1, FLAGS>(w, seg, flg, i0, n, gr, r, bit, tp);
                    }
                }
                s0 = s0_next;
                n = n_next;
                f0 = f0_next;
            }
        }
        __syncthreads();
        if (full) break;         // read between two barriers: uniform
        if (REC && a.memo_log2 && M.ctl[1]) {
            const uint32_t held = min(M.ctl[0], M.n_ent), asked = M.ctl[3], found = asked - min(asked, M.ctl[2]);
            memo_flush_row(w, M, r, kk, kmask, rcshift);
            __syncthreads();
            // full and hit by less than a quarter of the row's records: unrelated genomes, the memo only costs
            if (threadIdx.x == 0) {
                if (held >= M.n_ent && found * 4u < asked) M.ctl[1] = 0;
                M.ctl[2] = 0;
                M.ctl[3] = 0;
                M.ctl[4] = held;
                if (a.memo_stats) {         // (diagnostics, GRM_MEMO_STATS: records held, occurrences asked / found, rows with the memo on)
                    atomicAdd(&a.memo_stats[0], (unsigned long long)held);
                    atomicAdd(&a.memo_stats[1], (unsigned long long)asked);
                    atomicAdd(&a.memo_stats[2], (unsigned long long)found);
                    atomicAdd(&a.memo_stats[3], 1ull);
                }
            }
            if (full) break;
        }
        // end of the word-row: publish and clear the words of every occupied slot (by entry id where the slots of the ids are kept)
        if (w.sid) {
            const uint32_t nd = min(n_distinct, cap);
            for (uint32_t id = threadIdx.x; id < nd; id += blockDim.x) {
                const uint32_t slot = w.sid[id];
                const unsigned long long wd = words[slot];
                if (a.matrix_s) a.matrix_s[(((uint64_t)wg * n_rows + r) << a.cap_log2) + id] = wd;
                if (wd) {
                    const uint32_t m = meta[slot];
                    const bool multi = (m & META_SEEN) || (wd & (wd - 1));
                    meta[slot] = (uint16_t)(m | META_SEEN | (multi ? META_MULTI : 0u));
                    words[slot] = 0;
                }
            }
        } else
        for (uint32_t slot = threadIdx.x; slot < cap; slot += blockDim.x) {
            if (tkeys[slot] == EMPTY_KEY) continue;
            const unsigned long long wd = words[slot];
            const uint32_t m = meta[slot];
            if (a.matrix_s) a.matrix_s[(((uint64_t)wg * n_rows + r) << a.cap_log2) + (m & META_ID)] = wd;
            if (wd) {
                const bool multi = (m & META_SEEN) || (wd & (wd - 1));
                meta[slot] = (uint16_t)(m | META_SEEN | (multi ? META_MULTI : 0u));
                words[slot] = 0;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (full) {
        // The table overflowed.  What would this (bucket, sub-bucket) have needed?  Extrapolating the fill over the genomes
        // is exact for unrelated genomes and 10-50x too much for a pan-genome, whose distinct k-mers saturate -- and every
        // doubling of that estimate doubles the sub-bucket workgroups that re-read (record form: re-decode) the bucket.  So
        // the workgroup COUNTS: all its keys once more into a HyperLogLog sketch (up to 4096 registers in the table's LDS,
        // ~2 % error), and the host sizes the retry from the fullest workgroup's count.
        const uint32_t HLL_M = cap < 4096u ? cap : 4096u;          // (a power of two >= 64; 4 of the table's 18 bytes per slot)
        uint32_t *hll = reinterpret_cast<uint32_t *>(lds_raw);
        uint64_t *scr64 = reinterpret_cast<uint64_t *>(lds_raw + (size_t)HLL_M * 4);
        for (uint32_t i = threadIdx.x; i < HLL_M; i += blockDim.x) hll[i] = 0;
        __syncthreads();
        auto sketch = [&](uint64_t key) {
            const uint64_t h = mix64(key);
            if (sb && hash_sub(h, a.bb, sb) != w.sub) return;
            const uint32_t lo = (uint32_t)h;
            atomicMax(&hll[lo & (HLL_M - 1)], (uint32_t)__clz((lo >> 12) | 1u) - 11u);      // rank of the upper 20 bits: 1 .. 21
        };
        for (uint32_t vg = (uint32_t)wave; vg < GV; vg += (uint32_t)nw) {
            uint64_t sv = 0, nv = 0;
            seg_bounds(a.seg, (uint64_t)vg * B + b, sv, nv);
            if (REC) {
                nv &= 0xffffu;
                for (uint64_t q0 = 0; q0 < nv; q0 += 64) {
                    const ulonglong2 rec = q0 + lane < nv ? a.recs[sv + q0 + lane] : make_ulonglong2(0, 0);
                    const uint32_t len = run_len(rec.y);
                    RunDecoder dec = run_open(rec.x, rec.y, kk);
                    for (uint32_t t = 0; t < len; t++) {
                        sketch(run_canonical(dec));
                        run_next(dec, kmask, rcshift);
                    }
                }
            } else {
                for (uint64_t i = lane; i < nv; i += 64) sketch(a.keys[sv + i]);
            }
        }
        __syncthreads();
        // sum of 2^-register (in units of 2^-32) and the number of empty registers
        uint64_t part = 0, zeros = 0;
        for (uint32_t i = threadIdx.x; i < HLL_M; i += blockDim.x) {
            const uint32_t m = hll[i];
            part += 1ull << (32 - m);
            zeros += m == 0;
        }
        uint64_t sum = 0, nz = 0;
        (void)block_scan_sum64(part, scr64, &sum);
        (void)block_scan_sum64(zeros, scr64, &nz);
        if (threadIdx.x == 0) {
            const double m = (double)HLL_M;
            double est = 0.7213 / (1.0 + 1.079 / m) * m * m / ((double)sum / 4294967296.0);
            if (est <= 2.5 * m && nz) est = m * log(m / (double)nz);             // small range: linear counting
            atomicMax(a.need, (uint32_t)min(est * 1.05, 4.0e9));
            atomicMax(a.overflow, 1);
            a.wg_cnt[wg] = 0;
            a.wg_base[wg] = 0;
        }
        return;
    }
    if (threadIdx.x == 0) {
        const uint32_t nd = n_distinct;
        const uint64_t base = atomicAdd(a.n_out, (unsigned long long)nd);
        const bool fits = base + nd <= a.out_cap;
        if (!fits) atomicMax(a.overflow, 2);
        a.wg_base[wg] = base;
        a.wg_cnt[wg] = fits ? nd : 0u;
        out_base = fits ? base : ~0ull;
    }
    __syncthreads();
    const uint64_t base = out_base;
    if (base == ~0ull) return;
    for (uint32_t slot = threadIdx.x; slot < cap; slot += blockDim.x) {
        const uint64_t key = tkeys[slot];
        if (key == EMPTY_KEY) continue;
        const uint32_t m = meta[slot];
        a.out_keys[base + (m & META_ID)] = key;
        a.out_flags[base + (m & META_ID)] = (m & META_MULTI) ? 2 : 1;
    }
    if (FLAGS && a.track && a.track_g < G) {
        // the tracked rank's keys of this bucket: slot noted while probing -> index of the union entry (the slots' ids are final now)
        uint64_t ts = 0, tn = 0;
        seg_bounds(a.seg, (uint64_t)a.track_g * B + b, ts, tn);
        const uint32_t tp = (uint32_t)(a.in_flag_off[(uint64_t)a.track_g * B + b] - a.track_flag_base);
        for (uint64_t i = threadIdx.x; i < tn; i += blockDim.x) {
            if (sb && hash_sub(mix64(a.keys[ts + i]), a.bb, sb) != w.sub) continue;          // a sibling sub-bucket workgroup's key
            const uint32_t slot = a.track[tp + i];
            a.track[tp + i] = (uint32_t)(base + (meta[slot & cap_mask] & META_ID));
        }
    }
}

// local dictionary in bucket order (multi-GPU exchange): entries of workgroup wg -> [ord_off[wg], ord_off[wg + 1])
__global__ void dict_export_ordered_kernel(const uint64_t *__restrict__ keys, const uint8_t *__restrict__ flags,
                                           const uint64_t *__restrict__ wg_base, const uint32_t *__restrict__ wg_cnt,
                                           const uint64_t *__restrict__ ord_off, uint32_t n_wg, uint64_t *__restrict__ out_keys,
                                           uint8_t *__restrict__ out_flags)
{
    for (uint32_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
        const uint32_t n = wg_cnt[wg];
        const uint64_t src = wg_base[wg], dst = ord_off[wg];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            out_keys[dst + i] = keys[src + i];
            out_flags[dst + i] = flags[src + i];
        }
    }
}
__global__ void dict_bucket_offsets_kernel(const uint64_t *__restrict__ ord_off, int sb, uint32_t n_buckets, uint32_t *__restrict__ bucket_off)
{
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= n_buckets; b += gridDim.x * blockDim.x)
        bucket_off[b] = (uint32_t)ord_off[(uint64_t)b << sb];
}
__global__ void union_segments_kernel(const uint8_t *__restrict__ payload, uint32_t n_ranks, uint64_t stride, uint64_t flags_off,
                                      uint64_t boff_off, uint32_t n_buckets, RankShifts shifts, uint64_t *__restrict__ off,
                                      uint32_t *__restrict__ len, uint64_t *__restrict__ flag_off)
{
    const uint64_t total = (uint64_t)n_ranks * n_buckets;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = idx / n_buckets, b = idx % n_buckets;
        const uint32_t *boff = reinterpret_cast<const uint32_t *>(payload + r * stride + boff_off);
        // a rank that used 2^d times as many buckets: its buckets b * 2^d .. (b + 1) * 2^d - 1 are this bucket (bucket = TOP bits)
        const int d = shifts.d[r];
        const uint32_t o0 = boff[b << d], o1 = boff[(b + 1) << d];
        off[idx] = r * (stride / 8) + o0;
        len[idx] = o1 - o0;
        flag_off[idx] = r * stride + flags_off + o0;
    }
}

// column of every local entry = rank of its key in the sorted global dictionary.  A table of the first dictionary position of every
// PB-bit key prefix (2^22 prefixes: ~2.5 entries each at 10 M columns) narrows the search to one or two steps.  The table is filled from
// one sweep of the sorted dictionary (entry i opens every prefix between its predecessor's and its own) -- a binary search per prefix, as
// before, was 2^20 x 23 dependent loads: most of the 0.53 ms this step took.
// rank union with tracking: local entry (workgroup wg, id) stands at position ord_off[wg] + id of the rank's exported list;
// track[] gives that position's union entry, union_col[] the union entry's column (0xffffffff: filtered out)
__global__ void entry_cols_from_union_kernel(const uint64_t *__restrict__ wg_base, const uint32_t *__restrict__ wg_cnt, const uint64_t *__restrict__ ord_off,
                                             uint32_t n_wg, const uint32_t *__restrict__ track, const uint32_t *__restrict__ union_col,
                                             uint32_t *__restrict__ entry_col)
{
    for (uint32_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
        const uint64_t base = wg_base[wg], o = ord_off[wg];
        const uint32_t n = wg_cnt[wg];
        for (uint32_t id = threadIdx.x; id < n; id += blockDim.x) entry_col[base + id] = union_col[track[o + id]];
    }
}

__global__ void dict_prefix_index_kernel(const uint64_t *__restrict__ dict, uint64_t n_dict, int shift, uint32_t n_prefix,
                                         uint32_t *__restrict__ first)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_dict; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t lo = i ? (dict[i - 1] >> shift) + 1 : 0;                    // prefixes after the predecessor's ...
        const uint64_t hi = i < n_dict ? (dict[i] >> shift) : (uint64_t)n_prefix;  // ... up to this entry's (the end: all that are left, and n_prefix itself)
        for (uint64_t p = lo; p <= hi && p <= n_prefix; p++) first[p] = (uint32_t)i;
    }
}
__global__ void dict_entry_cols_kernel(const uint64_t *__restrict__ dict, uint64_t n_dict, const uint64_t *__restrict__ entry_keys,
                                       uint64_t n_entries, int shift, uint32_t n_prefix, const uint32_t *__restrict__ first,
                                       uint32_t *__restrict__ entry_col)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_entries; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = entry_keys[i];
        const uint64_t p = key >> shift;
        uint32_t c = 0xffffffffu;
        if (p < n_prefix) {
            uint64_t lo = first[p], hi = first[p + 1];
            while (lo < hi) {
                const uint64_t m = (lo + hi) >> 1;
                if (dict[m] < key) lo = m + 1; else hi = m;
            }
            if (lo < n_dict && dict[lo] == key) c = (uint32_t)lo;
        }
        entry_col[i] = c;
    }
}

// Stage 3b: presence words from (workgroup, row, entry) order to matrix[row][column].
// Bit layout: genome i -> row i/64, bit 63-(i%64)  (bin/kover/core/kover/utils.py:133-156).
// An entry's column is its key's rank by VALUE while entries are grouped by HASH, so consecutive entries go to
// unrelated columns: written straight into the row-major matrix that is one scattered 8-byte store per
// (entry, row) -- measured 4x HBM write amplification.  Two steps instead, both with full-line accesses:
//   matrix_entry_rows   [wg][row][id] -> entry-major lines  me[column][row]  (n_rows x 8 B contiguous per column)
//   matrix_transpose    me[column][row] -> matrix[row][column] through LDS tiles
// The direct form stays for matrices of fewer than 4 word-rows, where a line would be a partial one anyway.
__global__ __launch_bounds__(256) void matrix_permute_kernel(
    const uint64_t *__restrict__ matrix_s, const uint16_t *__restrict__ birth, const uint64_t *__restrict__ wg_base,
    const uint32_t *__restrict__ wg_cnt, const uint32_t *__restrict__ entry_col, uint32_t n_wg, uint32_t n_rows, uint32_t cap_log2,
    uint64_t *__restrict__ matrix, uint64_t n_cols)
{
    for (uint32_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
        const uint32_t n = wg_cnt[wg];
        const uint64_t base = wg_base[wg];
        for (uint32_t id = threadIdx.x; id < n; id += blockDim.x) {
            const uint32_t c = entry_col[base + id];
            if (c == 0xffffffffu) continue;
            const uint32_t b0 = birth[((uint64_t)wg << cap_log2) + id];
            for (uint32_t r = 0; r < n_rows; r++) {
                const uint64_t v = r >= b0 ? matrix_s[(((uint64_t)wg * n_rows + r) << cap_log2) + id] : 0ull;
                matrix[(uint64_t)r * n_cols + c] = v;
            }
        }
    }
}

constexpr int ER_IDS = 256, ER_ROWS = 16;       // tile of matrix_entry_rows: 256 entries x 16 word-rows
__global__ __launch_bounds__(ER_IDS) void matrix_entry_rows_kernel(
    const uint64_t *__restrict__ matrix_s, const uint16_t *__restrict__ birth, const uint64_t *__restrict__ wg_base,
    const uint32_t *__restrict__ wg_cnt, const uint32_t *__restrict__ entry_col, uint32_t n_wg, uint32_t n_rows, uint32_t cap_log2,
    uint64_t *__restrict__ me)
{
    __shared__ uint64_t tile[ER_IDS][ER_ROWS + 1];
    __shared__ uint32_t cols[ER_IDS];
    for (uint32_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
        const uint32_t n = wg_cnt[wg];
        const uint64_t base = wg_base[wg];
        for (uint32_t id0 = 0; id0 < n; id0 += ER_IDS) {
            const uint32_t id = id0 + threadIdx.x;
            const bool have = id < n;
            const uint32_t c = have ? entry_col[base + id] : 0xffffffffu;
            const uint32_t b0 = have ? birth[((uint64_t)wg << cap_log2) + id] : 0u;
            const uint32_t n_here = min((uint32_t)ER_IDS, n - id0);
            for (uint32_t r0 = 0; r0 < n_rows; r0 += ER_ROWS) {
                __syncthreads();            // the previous tile has been written out
                if (r0 == 0) cols[threadIdx.x] = c;
                // all sixteen words asked for before the first goes to LDS, from an address that is valid whatever the word's condition says
                // (the condition picks the word or zero afterwards): a load under a condition, stored at once, is a round trip of its own
                uint64_t got[ER_ROWS];
#pragma unroll
                for (int rr = 0; rr < ER_ROWS; rr++) {
                    const uint32_t r = min(r0 + (uint32_t)rr, n_rows - 1u);
                    got[rr] = matrix_s[(((uint64_t)wg * n_rows + r) << cap_log2) + (have ? id : 0u)];
                }
#pragma unroll
                for (int rr = 0; rr < ER_ROWS; rr++) {
                    const uint32_t r = r0 + rr;
                    tile[threadIdx.x][rr] = (have && r < n_rows && r >= b0) ? got[rr] : 0ull;
                }
                __syncthreads();
                const uint32_t rows_here = min((uint32_t)ER_ROWS, n_rows - r0);
                for (uint32_t j = threadIdx.x; j < n_here * ER_ROWS; j += ER_IDS) {
                    const uint32_t il = j / ER_ROWS, rr = j % ER_ROWS;
                    const uint32_t cc = cols[il];
                    if (cc != 0xffffffffu && rr < rows_here) me[(uint64_t)cc * n_rows + r0 + rr] = tile[il][rr];
                }
            }
            __syncthreads();                // cols[] is rewritten by the next chunk
        }
    }
}

constexpr int TR_COLS = 64, TR_ROWS = 16;       // tile of matrix_transpose
__global__ __launch_bounds__(256) void matrix_transpose_kernel(const uint64_t *__restrict__ me, uint64_t n_cols, uint32_t n_rows,
                                                               uint64_t *__restrict__ matrix)
{
    __shared__ uint64_t tile[TR_ROWS][TR_COLS + 1];
    const uint64_t n_ct = (n_cols + TR_COLS - 1) / TR_COLS;
    const uint32_t n_rt = (n_rows + TR_ROWS - 1) / TR_ROWS;
    for (uint64_t t = blockIdx.x; t < n_ct * n_rt; t += gridDim.x) {
        const uint64_t c0 = (t / n_rt) * TR_COLS;
        const uint32_t r0 = (uint32_t)(t % n_rt) * TR_ROWS;
        __syncthreads();
        // read: 16 consecutive threads take the 16 rows (128 B) of one column
        uint64_t got[TR_COLS * TR_ROWS / 256];               // (asked for together, from clamped addresses: see matrix_entry_rows)
#pragma unroll
        for (int pass = 0; pass < TR_COLS * TR_ROWS / 256; pass++) {
            const uint32_t j = pass * 256 + threadIdx.x;
            const uint32_t cl = j / TR_ROWS, rr = j % TR_ROWS;
            got[pass] = me[min(c0 + cl, n_cols - 1) * n_rows + min(r0 + rr, n_rows - 1u)];
        }
#pragma unroll
        for (int pass = 0; pass < TR_COLS * TR_ROWS / 256; pass++) {
            const uint32_t j = pass * 256 + threadIdx.x;
            const uint32_t cl = j / TR_ROWS, rr = j % TR_ROWS;
            const uint64_t c = c0 + cl;
            tile[rr][cl] = (c < n_cols && r0 + rr < n_rows) ? got[pass] : 0ull;
        }
        __syncthreads();
        // write: 64 consecutive threads take 64 consecutive columns (512 B) of one row
#pragma unroll
        for (int pass = 0; pass < TR_COLS * TR_ROWS / 256; pass++) {
            const uint32_t j = pass * 256 + threadIdx.x;
            const uint32_t rr = j / TR_COLS, cl = j % TR_COLS;
            const uint64_t c = c0 + cl;
            if (c < n_cols && r0 + rr < n_rows) matrix[(uint64_t)(r0 + rr) * n_cols + c] = tile[rr][cl];
        }
    }
}

// After sorting the (possibly multi-rank) concatenated dictionaries by key: one thread per
// element; the head of each run of equal keys decides keep/drop.
// multi = run longer than 1 (k-mer seen on several ranks) or any member flagged 2.
__global__ void dict_mark_kernel(const uint64_t *__restrict__ skeys, const uint8_t *__restrict__ sflags, uint64_t n,
                                 int filter_singleton, uint32_t *__restrict__ keep)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = skeys[i];
        uint32_t k = 0;
        if (i == 0 || skeys[i - 1] != key) {
            bool multi = sflags[i] >= 2;
            for (uint64_t j = i + 1; j < n && skeys[j] == key; j++) multi = true;
            k = (!filter_singleton || multi) ? 1u : 0u;
        }
        keep[i] = k;
    }
}
__global__ void dict_select_kernel(const uint64_t *__restrict__ skeys, const uint32_t *__restrict__ keep,
                                   const uint64_t *__restrict__ pos, uint64_t n, uint64_t *__restrict__ dict)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (keep[i]) dict[pos[i]] = skeys[i];
}
// the same for a list of DISTINCT keys sorted together with their entry index (one GPU: the dictionary is the batch's
// own): flags are read through the index, and every entry learns its column (0xffffffff: filtered out) right here --
// no search of the dictionary afterwards
__global__ void dict_mark_idx_kernel(const uint8_t *__restrict__ flags, const uint32_t *__restrict__ sidx, uint64_t n, int filter_singleton,
                                     uint32_t *__restrict__ keep)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        keep[i] = (!filter_singleton || flags[sidx[i]] >= 2) ? 1u : 0u;
}
__global__ void dict_select_idx_kernel(const uint64_t *__restrict__ skeys, const uint32_t *__restrict__ keep,
                                       const uint64_t *__restrict__ pos, const uint32_t *__restrict__ sidx, uint64_t n,
                                       uint64_t *__restrict__ dict, uint32_t *__restrict__ entry_col)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const bool k = keep[i] != 0;
        if (k) dict[pos[i]] = skeys[i];
        entry_col[sidx[i]] = k ? (uint32_t)pos[i] : 0xffffffffu;
    }
}
// bucket id ((bucket << sb) | sub) and identity column index of every dictionary entry
__global__ void dict_bucket_ids_kernel(const uint64_t *__restrict__ dict, uint64_t n, int bb, int sb,
                                       uint32_t *__restrict__ bucket_of, uint32_t *__restrict__ col_of)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64(dict[i]);
        bucket_of[i] = (hash_bucket(h, bb) << sb) | hash_sub(h, bb, sb);
        col_of[i] = (uint32_t)i;
    }
}
// copy variable-length segments to dense destinations: segment j = src[src_off[j] .. +len[j]) -> dst[dst_off[j]..)
__global__ void segments_compact_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ src_cnt,
                                        const uint64_t *__restrict__ src_off, const uint32_t *__restrict__ len,
                                        const uint64_t *__restrict__ dst_off, uint32_t n_seg,
                                        uint64_t *__restrict__ dst, uint32_t *__restrict__ dst_cnt)
{
    for (uint32_t j = blockIdx.x; j < n_seg; j += gridDim.x) {
        const uint64_t s0 = src_off[j], d0 = dst_off[j];
        const uint32_t n = len[j];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            dst[d0 + i] = src[s0 + i];
            if (dst_cnt) dst_cnt[d0 + i] = src_cnt ? src_cnt[s0 + i] : 1u;
        }
    }
}
// segment starts of a sorted u32 id list: start[v] = first index with ids[i] >= v, v in [0, n_ids]
__global__ void segment_starts_kernel(const uint32_t *__restrict__ ids, uint64_t n, uint32_t n_ids,
                                      uint64_t *__restrict__ start)
{
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v <= n_ids; v += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t m = (lo + hi) >> 1;
            if (ids[m] < v) lo = m + 1; else hi = m;
        }
        start[v] = lo;
    }
}
__global__ void gather_u64_kernel(const uint64_t *__restrict__ src, const uint32_t *__restrict__ index, uint64_t n,
                                  uint64_t *__restrict__ dst)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = src[index[i]];
}

// ------------------------------------------------------------------------------------
// Stage 3b (K7): presence bits.  One workgroup per (bucket, sub-bucket): the bucket's slice
// of the global dictionary sits in an LDS table; for each word-row (64 genomes) the waves
// stream the genomes' segments, look every k-mer up and OR the genome's bit into
// words[slot]; the finished 64-bit words go to matrix[row][column].
// Bit layout: genome i -> row i/64, bit 63-(i%64)  (bin/kover/core/kover/utils.py:133-156).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(TABLE_THREADS) void matrix_fill_kernel(
    const uint64_t *__restrict__ keys, const SegLayout seg_layout,
    uint32_t n_genomes, int bb, int sb, uint32_t cap_log2, const uint64_t *__restrict__ dkeys,
    const uint32_t *__restrict__ dcol, const uint64_t *__restrict__ seg_start, uint64_t *__restrict__ matrix,
    uint64_t n_cols, int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t cap = 1u << cap_log2, cap_mask = cap - 1;
    uint64_t *tkeys = reinterpret_cast<uint64_t *>(lds_raw);
    uint64_t *words = reinterpret_cast<uint64_t *>(lds_raw + (size_t)cap * 8);
    const uint32_t wg = blockIdx.x;
    const uint32_t B = 1u << bb;
    const uint32_t b = wg >> sb, sub = wg & ((1u << sb) - 1);
    const uint64_t d0 = seg_start[wg];
    const uint32_t nd = (uint32_t)(seg_start[wg + 1] - d0);
    if (nd == 0) return;
    if (nd > cap - (cap >> 3)) {
        if (threadIdx.x == 0) atomicExch(overflow, 1);
        return;
    }
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) tkeys[i] = EMPTY_KEY;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) {
        const uint64_t key = dkeys[d0 + i];
        bool ins;
        lds_find_or_insert(tkeys, cap_mask, key, mix64(key), &ins);
    }
    __syncthreads();
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;
    const uint32_t n_rows = (n_genomes + 63) >> 6;
    for (uint32_t r = 0; r < n_rows; r++) {
        for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) words[i] = 0;
        __syncthreads();
        const uint32_t g_end = min(r * 64 + 64, n_genomes);
        for (uint32_t g = r * 64 + wave; g < g_end; g += nw) {
            const unsigned long long bit = 1ull << (63 - (g & 63));
            uint64_t s0, n;
            seg_bounds(seg_layout, (uint64_t)g * B + b, s0, n);
            for (uint64_t i0 = lane; i0 < n; i0 += 64 * KEYS_IN_FLIGHT) {
                uint64_t kv[KEYS_IN_FLIGHT];
#pragma unroll
                for (int j = 0; j < KEYS_IN_FLIGHT; j++) {
                    const uint64_t i = i0 + 64u * j;
                    kv[j] = i < n ? keys[s0 + i] : EMPTY_KEY;
                }
#pragma unroll
                for (int j = 0; j < KEYS_IN_FLIGHT; j++) {
                    const uint64_t key = kv[j];
                    if (key == EMPTY_KEY) continue;
                    const uint64_t h = mix64(key);
                    if (sb && hash_sub(h, bb, sb) != sub) continue;
                    const uint32_t slot = lds_find(tkeys, cap_mask, key, h);
                    if (slot != 0xffffffffu) atomicOr((unsigned long long *)&words[slot], bit);
                }
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nd; i += blockDim.x) {
            const uint64_t key = dkeys[d0 + i];
            const uint32_t slot = lds_find(tkeys, cap_mask, key, mix64(key));
            matrix[(uint64_t)r * n_cols + dcol[d0 + i]] = words[slot];
        }
        __syncthreads();
    }
}

// masked popcount over the packed matrix: out[c] = popcount(matrix[r][c] & mask[r]) summed
// over rows -- the learner-side inner loop (learning/common/popcount.pyx:76-95 with
// learning/common/rules.py:243-262), used here for self-checks (per-column carrier count).
__global__ void column_popcount_kernel(const uint64_t *__restrict__ matrix, uint64_t n_rows, uint64_t n_cols,
                                       const uint64_t *__restrict__ row_mask, uint32_t *__restrict__ out)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t s = 0;
        for (uint64_t r = 0; r < n_rows; r++) s += __popcll(matrix[r * n_cols + c] & (row_mask ? row_mask[r] : ~0ull));
        out[c] = s;
    }
}

// `kover dataset split` risk tables (dataset/split.py:171-188), device part: ONE sweep of the matrix with the masks of
// the positive and the negative training genomes gives every k-mer's error count as a presence rule,
//     errors[c] = (n_pos - popcount(col & pos)) + popcount(col & neg)        (an integer in 0 .. n_train)
// and the histogram of those counts (LDS per workgroup when n_train < 8192, flushed with one global add per
// occupied bin).  The reference's rounding to 5 decimals and unique-indexing then act on at most n_train + 1
// distinct values; the per-k-mer index tables come from two small look-up tables (risk_index_kernel).
constexpr uint32_t ERR_LDS_BINS = 8192;
__global__ __launch_bounds__(256) void column_errors_kernel(const uint64_t *__restrict__ matrix, uint64_t n_rows, uint64_t n_cols,
                                                            const uint64_t *__restrict__ pos_mask, const uint64_t *__restrict__ neg_mask,
                                                            uint32_t n_pos, uint32_t n_train, uint32_t *__restrict__ errors,
                                                            unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t lh[ERR_LDS_BINS];
    const bool lds = n_train < ERR_LDS_BINS;
    if (lds) {
        for (uint32_t i = threadIdx.x; i <= n_train; i += blockDim.x) lh[i] = 0;
        __syncthreads();
    }
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t p = 0, q = 0;
        for (uint64_t r = 0; r < n_rows; r++) {
            const uint64_t w = matrix[r * n_cols + c];
            p += __popcll(w & pos_mask[r]);
            q += __popcll(w & neg_mask[r]);
        }
        const uint32_t e = n_pos - p + q;
        errors[c] = e;
        if (lds) atomicAdd(&lh[e], 1u);
        else atomicAdd(&hist[e], 1ull);
    }
    if (lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i <= n_train; i += blockDim.x) {
            const uint32_t v = lh[i];
            if (v) atomicAdd(&hist[i], (unsigned long long)v);
        }
    }
}
__global__ void risk_index_kernel(const uint32_t *__restrict__ errors, uint64_t n_cols, const uint32_t *__restrict__ lut_presence,
                                  const uint32_t *__restrict__ lut_absence, uint32_t *__restrict__ by_kmer, uint32_t *__restrict__ by_anti)
{
    for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t e = errors[c];
        by_kmer[c] = lut_presence[e];
        by_anti[c] = lut_absence[e];
    }
}

__global__ void iota_u32_kernel(uint32_t *p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

// ------------------------------------------------------------------------------------
// launchers (called from grm_api.cpp)
// ------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap = 256u * 8u)
{
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (uint32_t)(g > cap ? cap : g);
}

void launch_parse_summarize(hipStream_t s, const uint8_t *raw, uint32_t n_tiles, const uint8_t *tile_meta, TileSummary *sums, uint32_t *chunk_pre,
                            uint64_t *chunk_pre64)
{
    hipLaunchKernelGGL(parse_summarize_kernel, dim3(n_tiles), dim3(PARSE_THREADS), 0, s, raw, n_tiles, tile_meta, sums, chunk_pre, chunk_pre64);
}
// 2 bytes per 16-byte chunk (FASTA tiles; the buffer keeps 4); with FASTQ tiles in the batch 8 more per chunk behind them (chunk_pre64 = chunk_pre + the 4-byte part)
size_t parse_chunk_pre_bytes(uint32_t n_tiles, bool with_fastq) { return (size_t)n_tiles * ROUNDS_PER_TILE * PARSE_THREADS * (with_fastq ? 12 : 4); }
// scratch: n_tiles * 20 B (tile prefixes) + n_blocks * (20 + 1 + 8) B + 8 B
size_t parse_scan_scratch_bytes(uint32_t n_tiles)
{
    const size_t n_blocks = ((size_t)n_tiles + 255) / 256;
    return (size_t)n_tiles * sizeof(ScanElem<uint32_t>) + n_blocks * (sizeof(ScanElem<uint32_t>) + 16) + 64;
}
void launch_parse_scan(hipStream_t s, const TileSummary *sums, uint32_t n_tiles, const uint8_t *tile_meta, uint64_t *tile_off,
                       uint8_t *tile_state, const uint32_t *genome_tile_off, uint32_t n_genomes,
                       uint64_t *genome_sym_off, void *scratch)
{
    const uint32_t n_blocks = (n_tiles + 255) / 256;
    uint8_t *p = reinterpret_cast<uint8_t *>(scratch);
    uint64_t *total = reinterpret_cast<uint64_t *>(p); p += 16;
    uint64_t *block_off = reinterpret_cast<uint64_t *>(p); p += (size_t)n_blocks * 8;
    ScanElem<uint32_t> *tile_pre = reinterpret_cast<ScanElem<uint32_t> *>(p); p += (size_t)n_tiles * sizeof(ScanElem<uint32_t>);
    ScanElem<uint32_t> *block_tot = reinterpret_cast<ScanElem<uint32_t> *>(p); p += (size_t)n_blocks * sizeof(ScanElem<uint32_t>);
    uint8_t *block_state = p;
    hipLaunchKernelGGL(parse_scan_a_kernel, dim3(n_blocks), dim3(256), 0, s, sums, n_tiles, tile_meta, tile_pre, block_tot);
    hipLaunchKernelGGL(parse_scan_b_kernel, dim3(1), dim3(1024), 0, s, block_tot, n_blocks, block_state, block_off, total);
    hipLaunchKernelGGL(parse_scan_c_kernel, dim3(n_blocks), dim3(256), 0, s, tile_pre, n_tiles, tile_meta, block_state, block_off,
                       total, tile_off, tile_state);
    hipLaunchKernelGGL(genome_offsets_kernel, dim3((n_genomes + 256) / 256), dim3(256), 0, s, tile_off, genome_tile_off, n_genomes,
                       genome_sym_off);
}
void launch_parse_pack(hipStream_t s, const uint8_t *raw, uint32_t n_tiles, const uint8_t *tile_meta, const uint64_t *tile_off,
                       const uint8_t *tile_state, uint64_t *sym2, uint64_t *inv, const TileSummary *sums, const uint32_t *chunk_pre,
                       const uint64_t *chunk_pre64)
{
    hipLaunchKernelGGL(parse_prezero_kernel, dim3(((uint64_t)n_tiles + 256) / 256), dim3(256), 0, s, tile_off, n_tiles, sym2, inv);
    hipLaunchKernelGGL(parse_pack_kernel, dim3(n_tiles), dim3(PARSE_THREADS), 0, s, raw, n_tiles, tile_meta, tile_off,
                       tile_state, sym2, inv, sums, chunk_pre, chunk_pre64);
}

// desc: n_tiles words + the ticket behind them (all zeroed here); pieces: n_tiles x 2 x 3 words
size_t parse_fused_desc_bytes(uint32_t n_tiles) { return ((size_t)n_tiles + 2) * 8; }
size_t parse_fused_piece_bytes(uint32_t n_tiles) { return ((size_t)n_tiles + 1) * 48; }
hipError_t launch_parse_fused(hipStream_t s, const uint8_t *raw, uint32_t n_tiles, const uint8_t *tile_meta, uint64_t *desc, uint64_t *pieces,
                              uint64_t *tile_off, uint64_t *sym2, uint64_t *inv, const uint32_t *genome_tile_off, uint32_t n_genomes,
                              uint64_t *genome_sym_off)
{
    hipError_t e = hipMemsetAsync(desc, 0, parse_fused_desc_bytes(n_tiles), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(parse_fused_kernel, dim3(n_tiles), dim3(PARSE_THREADS), 0, s, raw, n_tiles, tile_meta, desc, reinterpret_cast<uint32_t *>(desc + n_tiles),
                       tile_off, sym2, inv, pieces, getenv("GRM_PARSE_ABLATE") ? atoi(getenv("GRM_PARSE_ABLATE")) : 0);
    hipLaunchKernelGGL(parse_stitch_kernel, dim3(((uint64_t)n_tiles + 256) / 256), dim3(256), 0, s, tile_off, n_tiles, pieces, sym2, inv);
    hipLaunchKernelGGL(genome_offsets_kernel, dim3((n_genomes + 256) / 256), dim3(256), 0, s, tile_off, genome_tile_off, n_genomes, genome_sym_off);
    return hipGetLastError();
}

static KmerArgs make_args(const KmerLaunch &L)
{
    KmerArgs a;
    a.sym2 = L.sym2; a.inv = L.inv; a.total_syms = L.total_syms;
    a.n_groups = (L.total_syms + 63) / 64;
    a.genome_sym_off = L.genome_sym_off; a.n_genomes = L.n_genomes; a.k = L.k; a.bb = L.bb;
    a.groups_per_thread = L.groups_per_thread;
    return a;
}
static uint32_t n_spans_of(const KmerArgs &a)
{
    const uint64_t span_groups = (uint64_t)KMER_THREADS * a.groups_per_thread;
    return (uint32_t)((a.n_groups + span_groups - 1) / span_groups);
}
void launch_kmer_hist(hipStream_t s, const KmerLaunch &L, uint32_t *counts)
{
    KmerArgs a = make_args(L);
    if (a.total_syms == 0) return;
    const uint32_t n_spans = n_spans_of(a);
    const uint32_t grid = ((n_spans + 7) / 8) * 8;     // xcd_span() needs the full 8 x per layout
    hipLaunchKernelGGL(kmer_hist_kernel, dim3(grid), dim3(KMER_THREADS), (size_t)4 << L.bb, s, a, n_spans, counts);
}
int scatter_b1_bits(int bb) { return bb < L1_MAX_BITS ? bb : L1_MAX_BITS; }

// level 1 writes `out` = keys1 when a second level follows, else the final keys
void launch_kmer_scatter_l1(hipStream_t s, const KmerLaunch &L, const uint64_t *off, const uint64_t *coarse_off,
                            uint32_t *cursor1, uint64_t *out, uint64_t region_stride, int *overflow)
{
    KmerArgs a = make_args(L);
    if (a.total_syms == 0) return;
    const uint32_t n_tiles = (uint32_t)((a.total_syms + L1_TILE - 1) / L1_TILE);
    const uint32_t grid = ((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(kmer_scatter_l1_kernel, dim3(grid), dim3(L1_THREADS), L1_LDS_BYTES, s, a, scatter_b1_bits(L.bb), n_tiles,
                       off, coarse_off, cursor1, out, region_stride, overflow);
}
void launch_region_hist(hipStream_t s, const uint64_t *keys1, const uint64_t *coarse_off, uint64_t n_regions, int bb,
                        uint32_t *counts)
{
    if (!n_regions) return;
    const uint32_t grid = (uint32_t)(n_regions < 256u * 32u ? n_regions : 256u * 32u);
    hipLaunchKernelGGL(region_hist_kernel, dim3(grid), dim3(256), 0, s, keys1, coarse_off, n_regions, bb, scatter_b1_bits(bb), counts);
}
void launch_sum_u32(hipStream_t s, const uint32_t *in, uint64_t n, uint64_t *out)
{
    hipLaunchKernelGGL(sum_u32_kernel, dim3(grid_for(n, 256, 256)), dim3(256), 0, s, in, n, reinterpret_cast<unsigned long long *>(out));
}
void launch_kmer_scatter_l2(hipStream_t s, const KmerLaunch &L, const uint64_t *off, const uint64_t *keys1,
                            uint64_t *keys, uint64_t region_stride, uint32_t fine_cap, const uint32_t *cursor1, uint32_t *len_out,
                            int *overflow)
{
    const int b1 = scatter_b1_bits(L.bb);
    if (L.total_syms == 0 || L.bb <= b1) return;
    const uint64_t n_regions = (uint64_t)L.n_genomes << b1;
    const uint32_t grid2 = (uint32_t)(n_regions < 256u * 16u ? n_regions : 256u * 16u);
    hipLaunchKernelGGL(kmer_scatter_l2_kernel, dim3(grid2), dim3(L2_THREADS), L2_LDS_BYTES, s, keys1, keys, off,
                       n_regions, L.bb, b1, region_stride, fine_cap, cursor1, len_out, overflow);
}
void launch_keys_partition_hist(hipStream_t s, const uint64_t *in, uint64_t n, const uint64_t *genome_key_off,
                                uint32_t n_genomes, int bb, uint32_t *counts)
{
    if (!n) return;
    hipLaunchKernelGGL(keys_hist_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, in, n, genome_key_off, n_genomes, bb, counts);
}
void launch_keys_partition_scatter(hipStream_t s, const uint64_t *in, uint64_t n, const uint64_t *genome_key_off,
                                   uint32_t n_genomes, int bb, const uint64_t *off, uint32_t *cursor, uint64_t *keys)
{
    if (!n) return;
    hipLaunchKernelGGL(keys_scatter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, in, n, genome_key_off, n_genomes, bb,
                       off, cursor, keys);
}
void launch_scan_u32(hipStream_t s, const uint32_t *in, uint64_t n, uint64_t *out)
{
    hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, s, in, n, out);
}
void launch_bucket_dedup(hipStream_t s, uint64_t *keys, const SegLayout &seg, uint64_t n_segments, uint32_t cap_log2,
                         uint32_t abundance_min, uint32_t *len_out, const uint32_t *marks, uint32_t *counts_out, int *overflow)
{
    if (!n_segments) return;
    const size_t lds = (((size_t)12) << cap_log2) + TABLE_SCRATCH_BYTES;
    const uint32_t grid = (uint32_t)(n_segments < 256u * 16u ? n_segments : 256u * 16u);
    hipLaunchKernelGGL(bucket_dedup_kernel, dim3(grid), dim3(TABLE_THREADS), lds, s, keys, seg, n_segments, cap_log2,
                       abundance_min, len_out, marks, counts_out, overflow);
}
// wave form: wave_cap_log2 in {9, 10, 11}; marks: one bit per segment (zeroed by the caller), set for segments left untouched
void launch_bucket_dedup_wave(hipStream_t s, uint64_t *keys, const SegLayout &seg, uint64_t n_segments, int wave_cap_log2,
                              uint32_t abundance_min, uint32_t *len_out, uint32_t *marks, uint32_t *counts_out, int *overflow)
{
    if (!n_segments) return;
    // 48 KiB of LDS per workgroup in every form: 3 workgroups per CU
#define GRM_LAUNCH_DW(C, W)                                                                                               \
    {                                                                                                                     \
        const uint64_t wgs = (n_segments + (W) - 1) / (W);                                                                \
        const uint32_t grid = (uint32_t)(wgs < 256u * 24u ? wgs : 256u * 24u);                                            \
        hipLaunchKernelGGL((bucket_dedup_wave_kernel<C, W>), dim3(grid), dim3((W) * 64), 0, s, keys, seg, n_segments,     \
                           abundance_min, len_out, marks, counts_out, overflow);                                          \
    }
    if (wave_cap_log2 <= 9) GRM_LAUNCH_DW(9, 8)
    else if (wave_cap_log2 == 10) GRM_LAUNCH_DW(10, 4)
    else GRM_LAUNCH_DW(11, 2)
#undef GRM_LAUNCH_DW
}
// counting stage over the record form: the wave-table launch (2^cap_log2 one-word slots per wave; 8: tests), then -- when it left
// segments behind (too many k-mers or records for a wave; *any_big, read back by the caller) -- launch_record_dedup_rest
void launch_record_count(hipStream_t s, const void *recs, uint32_t rstride, const uint32_t *rcount, uint64_t n_regions, int k, int bb, int b1,
                         uint64_t kstride, int cap_log2, uint32_t abundance_min, uint64_t *keys, uint32_t *counts_out, uint64_t *koff,
                         uint32_t *klen, int *overflow, uint8_t *region_big, int *any_big)
{
    if (!n_regions) return;
    const ulonglong2 *r = reinterpret_cast<const ulonglong2 *>(recs);
    const int b2 = bb - b1;
    const uint32_t grid = (uint32_t)(n_regions < 256u * 32u ? n_regions : 256u * 32u);
    if (cap_log2 <= 8)
        hipLaunchKernelGGL((record_count_kernel<8, 4>), dim3(grid), dim3(256), 0, s, r, rstride, rcount, n_regions, k, b2, kstride, abundance_min, keys,
                           counts_out, koff, klen, overflow, region_big, any_big);
    else
        hipLaunchKernelGGL((record_count_kernel<9, 4>), dim3(grid), dim3(256), 0, s, r, rstride, rcount, n_regions, k, b2, kstride, abundance_min, keys,
                           counts_out, koff, klen, overflow, region_big, any_big);
}
void launch_record_dedup_rest(hipStream_t s, const void *recs, uint32_t rstride, const uint32_t *rcount, uint64_t n_regions, int k, int bb, int b1,
                              uint64_t kstride, int cap_log2, uint32_t abundance_min, uint64_t *keys, uint32_t *counts_out, uint64_t *koff,
                              uint32_t *klen, int *overflow, const uint8_t *region_big)
{
    if (!n_regions) return;
    const ulonglong2 *r = reinterpret_cast<const ulonglong2 *>(recs);
    const int b2 = bb - b1;
    const uint32_t grid = (uint32_t)(n_regions < 256u * 16u ? n_regions : 256u * 16u);
    if (cap_log2 <= 8)
        hipLaunchKernelGGL((record_dedup_kernel<8, 4>), dim3(grid), dim3(256), 0, s, r, rstride, rcount, n_regions, k, b2, kstride, abundance_min, keys,
                           counts_out, koff, klen, overflow, region_big);
    else
        hipLaunchKernelGGL((record_dedup_kernel<9, 4>), dim3(grid), dim3(256), 0, s, r, rstride, rcount, n_regions, k, b2, kstride, abundance_min, keys,
                           counts_out, koff, klen, overflow, region_big);
}
// counting stage over the record form, genomes in parts: big_log2 in {12, 13} slots of the workgroup's table
hipError_t launch_record_merge(hipStream_t s, const void *recs, uint32_t rstride, const uint32_t *rcount, const uint64_t *roff, const uint32_t *rlen,
                               uint32_t n_genomes, int part_bits, int k, int bb, int b1, uint64_t kstride, int big_log2, uint32_t abundance_min,
                               uint64_t *keys, uint32_t *counts_out, uint64_t *koff, uint32_t *klen, int *overflow)
{
    const uint64_t n_regions = (uint64_t)n_genomes << b1;
    if (!n_regions) return hipSuccess;
    const ulonglong2 *r = reinterpret_cast<const ulonglong2 *>(recs);
    const int b2 = bb - b1;
    static std::atomic<uint64_t> lds_set12{0}, lds_set13{0};
    hipError_t ea = ensure_dynamic_lds(reinterpret_cast<const void *>(record_merge_kernel<12>), 150 * 1024, lds_set12);
    if (ea == hipSuccess) ea = ensure_dynamic_lds(reinterpret_cast<const void *>(record_merge_kernel<13>), 150 * 1024, lds_set13);
    if (ea != hipSuccess) return ea;
    const size_t lds = ((size_t)14 << (big_log2 >= 13 ? 13 : 12)) + 8 * 1360;
    const uint32_t grid = (uint32_t)(n_regions < 256u * 8u ? n_regions : 256u * 8u);
    if (big_log2 >= 13)
        hipLaunchKernelGGL(record_merge_kernel<13>, dim3(grid), dim3(512), lds, s, r, rstride, rcount, roff, rlen, n_genomes, part_bits, k, b1, b2, kstride,
                           abundance_min, keys, counts_out, koff, klen, overflow);
    else
        hipLaunchKernelGGL(record_merge_kernel<12>, dim3(grid), dim3(512), lds, s, r, rstride, rcount, roff, rlen, n_genomes, part_bits, k, b1, b2, kstride,
                           abundance_min, keys, counts_out, koff, klen, overflow);
    return hipSuccess;
}
static int g_dict_kif = 8, g_table_threads = 0;      // 0: every kernel's own default (key form: TABLE_THREADS, record form: 256)
void set_table_tuning(int kif, int threads)
{
    g_dict_kif = (kif == 1 || kif == 2 || kif == 4) ? kif : 8;
    g_table_threads = (threads == 256 || threads == 512 || threads == 1024) ? threads : 0;
}
void launch_dict_build(hipStream_t s, const DictArgs &a)
{
    const size_t lds = (((size_t)18) << a.cap_log2) + TABLE_SCRATCH_BYTES;
    const int key_threads = g_table_threads ? g_table_threads : TABLE_THREADS;
    const dim3 grid(1u << (a.bb + a.sb)), block(key_threads);     // 256 / 512 / 1024 threads: the wave count divides 64
#define GRM_LAUNCH_DICT(K, T, F, R) hipLaunchKernelGGL((dict_build_kernel<K, T, F, R>), grid, block, lds, s, a)
    if (a.in_flags) {                   // union over ranks: one instance; one word-row, i.e. all set-up: 256 threads (0.51 ms against 0.64 with 512)
        hipLaunchKernelGGL((dict_build_kernel<4, 1024, true, false>), grid, dim3(g_table_threads ? g_table_threads : 256), lds, s, a);
        return;
    }
    if (a.recs) {                       // one record (up to 16 keys) per lane; 8 waves (the instance's launch bound)
        const size_t lds_t = (((size_t)18) << a.cap_log2) + TABLE_SCRATCH_BYTES + (size_t)(TABLE_THREADS / 64) * 192;      // + the pool table of every wave
        DictArgs b = a;
        // (a caller-forced table of 2^13 slots leaves no room for the record memo inside the 159 KB a workgroup may allocate: none then)
        if (lds_t + dict_memo_bytes(b.memo_log2, b.cap_log2) > (size_t)159 * 1024) b.memo_log2 = 0;
        const size_t lds_r = lds_t + dict_memo_bytes(b.memo_log2, b.cap_log2);                                              // + the record memo
        // 256 threads unless told otherwise: with the 104-record memo (52 KB of LDS) three workgroups share a CU, and what a workgroup
        // spends outside its word-rows (table set-up, the first bounds -> records round trips, the entries' way out) is latency that
        // only more workgroups in flight hide -- 1000 x 5 Mbp: 4.0 ms against 4.3 with 512 threads and 208 records; a rank's 128
        // genomes (two word-rows, nearly all of it set-up): 1.2 against 1.45
        hipLaunchKernelGGL((dict_build_kernel<8, 512, false, true>), grid, dim3(g_table_threads == 512 || g_table_threads == 1024 ? 512 : 256), lds_r, s, b);
        return;
    }
    const int kif = (key_threads > 512 && g_dict_kif > 4) ? 4 : g_dict_kif;
    switch (kif) {
    case 1: GRM_LAUNCH_DICT(1, 1024, false, false); break;
    case 2: GRM_LAUNCH_DICT(2, 1024, false, false); break;
    case 4: GRM_LAUNCH_DICT(4, 1024, false, false); break;
    default: GRM_LAUNCH_DICT(8, 512, false, false); break;
    }
#undef GRM_LAUNCH_DICT
}
void launch_bucket_offsets(hipStream_t s, const uint64_t *ord_off, int sb, uint32_t n_buckets, uint32_t *bucket_off)
{
    hipLaunchKernelGGL(dict_bucket_offsets_kernel, dim3(grid_for((uint64_t)n_buckets + 1, 256)), dim3(256), 0, s, ord_off, sb, n_buckets,
                       bucket_off);
}
void launch_dict_export_ordered(hipStream_t s, const uint64_t *keys, const uint8_t *flags, const uint64_t *wg_base, const uint32_t *wg_cnt,
                                const uint64_t *ord_off, uint32_t n_wg, int sb, uint64_t *out_keys, uint8_t *out_flags, uint32_t *bucket_off)
{
    hipLaunchKernelGGL(dict_export_ordered_kernel, dim3(n_wg < 256u * 32u ? n_wg : 256u * 32u), dim3(256), 0, s, keys, flags, wg_base, wg_cnt,
                       ord_off, n_wg, out_keys, out_flags);
    launch_bucket_offsets(s, ord_off, sb, n_wg >> sb, bucket_off);
}
void launch_union_segments(hipStream_t s, const uint8_t *payload, uint32_t n_ranks, uint64_t stride, uint64_t flags_off, uint64_t boff_off,
                           uint32_t n_buckets, const RankShifts &shifts, uint64_t *off, uint32_t *len, uint64_t *flag_off)
{
    hipLaunchKernelGGL(union_segments_kernel, dim3(grid_for((uint64_t)n_ranks * n_buckets, 256)), dim3(256), 0, s, payload, n_ranks, stride,
                       flags_off, boff_off, n_buckets, shifts, off, len, flag_off);
}
void launch_entry_cols_from_union(hipStream_t s, const uint64_t *wg_base, const uint32_t *wg_cnt, const uint64_t *ord_off, uint32_t n_wg,
                                  const uint32_t *track, const uint32_t *union_col, uint32_t *entry_col)
{
    if (!n_wg) return;
    hipLaunchKernelGGL(entry_cols_from_union_kernel, dim3(n_wg < (1u << 16) ? n_wg : (1u << 16)), dim3(64), 0, s, wg_base, wg_cnt, ord_off, n_wg, track,
                       union_col, entry_col);
}

void launch_dict_entry_cols(hipStream_t s, const uint64_t *dict, uint64_t n_dict, const uint64_t *entry_keys, uint64_t n_entries, int k,
                            uint32_t *prefix_first /* 2^22 + 2 entries */, uint32_t *entry_col)
{
    if (!n_entries) return;
    const int pb = 2 * k < 22 ? 2 * k : 22;
    const int shift = 2 * k - pb;
    const uint32_t n_prefix = 1u << pb;
    hipLaunchKernelGGL(dict_prefix_index_kernel, dim3(grid_for(n_dict + 1, 256, 256u * 32u)), dim3(256), 0, s, dict, n_dict, shift, n_prefix,
                       prefix_first);
    hipLaunchKernelGGL(dict_entry_cols_kernel, dim3(grid_for(n_entries, 256, 256u * 32u)), dim3(256), 0, s, dict, n_dict, entry_keys,
                       n_entries, shift, n_prefix, prefix_first, entry_col);
}
void launch_matrix_permute(hipStream_t s, const uint64_t *matrix_s, const uint16_t *birth, const uint64_t *wg_base,
                           const uint32_t *wg_cnt, const uint32_t *entry_col, uint32_t n_wg, uint32_t n_rows, uint32_t cap_log2,
                           uint64_t *matrix, uint64_t n_cols, uint64_t *entry_major /* n_cols * n_rows words, zeroed where columns may lack an entry; nullptr: direct form */)
{
    if (!n_wg || !n_rows || !n_cols) return;
    const uint32_t grid = n_wg < 256u * 32u ? n_wg : 256u * 32u;
    if (!entry_major || n_rows < 4) {
        hipLaunchKernelGGL(matrix_permute_kernel, dim3(grid), dim3(256), 0, s, matrix_s, birth, wg_base, wg_cnt, entry_col, n_wg, n_rows,
                           cap_log2, matrix, n_cols);
        return;
    }
    hipLaunchKernelGGL(matrix_entry_rows_kernel, dim3(grid), dim3(ER_IDS), 0, s, matrix_s, birth, wg_base, wg_cnt, entry_col, n_wg, n_rows,
                       cap_log2, entry_major);
    const uint64_t n_tiles = ((n_cols + TR_COLS - 1) / TR_COLS) * ((n_rows + TR_ROWS - 1) / TR_ROWS);
    hipLaunchKernelGGL(matrix_transpose_kernel, dim3((uint32_t)(n_tiles < 256u * 64u ? n_tiles : 256u * 64u)), dim3(256), 0, s, entry_major,
                       n_cols, n_rows, matrix);
}
void launch_dict_mark(hipStream_t s, const uint64_t *skeys, const uint8_t *sflags, uint64_t n, int filter_singleton,
                      uint32_t *keep)
{
    if (!n) return;
    hipLaunchKernelGGL(dict_mark_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, skeys, sflags, n, filter_singleton, keep);
}
void launch_dict_select(hipStream_t s, const uint64_t *skeys, const uint32_t *keep, const uint64_t *pos, uint64_t n,
                        uint64_t *dict)
{
    if (!n) return;
    hipLaunchKernelGGL(dict_select_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, skeys, keep, pos, n, dict);
}
__global__ void gather_u8_kernel(const uint8_t *__restrict__ src, const uint32_t *__restrict__ index, uint64_t n, uint8_t *__restrict__ dst)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[index[i]];
}
void launch_gather_u8(hipStream_t s, const uint8_t *src, const uint32_t *index, uint64_t n, uint8_t *dst)
{
    if (!n) return;
    hipLaunchKernelGGL(gather_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, index, n, dst);
}
void launch_dict_mark_idx(hipStream_t s, const uint8_t *flags, const uint32_t *sidx, uint64_t n, int filter_singleton, uint32_t *keep)
{
    if (!n) return;
    hipLaunchKernelGGL(dict_mark_idx_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, flags, sidx, n, filter_singleton, keep);
}
void launch_dict_select_idx(hipStream_t s, const uint64_t *skeys, const uint32_t *keep, const uint64_t *pos, const uint32_t *sidx, uint64_t n,
                            uint64_t *dict, uint32_t *entry_col)
{
    if (!n) return;
    hipLaunchKernelGGL(dict_select_idx_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, skeys, keep, pos, sidx, n, dict, entry_col);
}
void launch_dict_bucket_ids(hipStream_t s, const uint64_t *dict, uint64_t n, int bb, int sb, uint32_t *bucket_of,
                            uint32_t *col_of)
{
    if (!n) return;
    hipLaunchKernelGGL(dict_bucket_ids_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, dict, n, bb, sb, bucket_of, col_of);
}
void launch_segments_compact(hipStream_t s, const uint64_t *src, const uint32_t *src_cnt, const uint64_t *src_off,
                             const uint32_t *len, const uint64_t *dst_off, uint32_t n_seg, uint64_t *dst,
                             uint32_t *dst_cnt)
{
    if (!n_seg) return;
    hipLaunchKernelGGL(segments_compact_kernel, dim3(n_seg < 4096u ? n_seg : 4096u), dim3(256), 0, s, src, src_cnt,
                       src_off, len, dst_off, n_seg, dst, dst_cnt);
}
void launch_segment_starts(hipStream_t s, const uint32_t *ids, uint64_t n, uint32_t n_ids, uint64_t *start)
{
    hipLaunchKernelGGL(segment_starts_kernel, dim3(grid_for((uint64_t)n_ids + 1, 256)), dim3(256), 0, s, ids, n, n_ids, start);
}
void launch_gather_u64(hipStream_t s, const uint64_t *src, const uint32_t *index, uint64_t n, uint64_t *dst)
{
    if (!n) return;
    hipLaunchKernelGGL(gather_u64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, index, n, dst);
}
void launch_matrix_fill(hipStream_t s, const uint64_t *keys, const SegLayout &seg,
                        uint32_t n_genomes, int bb, int sb, uint32_t cap_log2, const uint64_t *dkeys,
                        const uint32_t *dcol, const uint64_t *seg_start, uint64_t *matrix, uint64_t n_cols,
                        int *overflow)
{
    const size_t lds = (((size_t)16) << cap_log2) + TABLE_SCRATCH_BYTES;
    hipLaunchKernelGGL(matrix_fill_kernel, dim3(1u << (bb + sb)), dim3(TABLE_THREADS), lds, s, keys, seg, n_genomes,
                       bb, sb, cap_log2, dkeys, dcol, seg_start, matrix, n_cols, overflow);
}
void launch_column_popcount(hipStream_t s, const uint64_t *matrix, uint64_t n_rows, uint64_t n_cols,
                            const uint64_t *row_mask, uint32_t *out)
{
    if (!n_cols) return;
    hipLaunchKernelGGL(column_popcount_kernel, dim3(grid_for(n_cols, 256)), dim3(256), 0, s, matrix, n_rows, n_cols,
                       row_mask, out);
}
// (hi, lo) pairs of a two-word k-mer set -> separate hi / lo arrays
__global__ void split_pairs_u64_kernel(const uint64_t *__restrict__ pairs, uint64_t n, uint64_t *__restrict__ hi,
                                       uint64_t *__restrict__ lo)
{
    // grid_for() caps the grid: stride over the rest
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(pairs)[i];
        hi[i] = v.x;
        lo[i] = v.y;
    }
}
void launch_split_pairs_u64(hipStream_t s, const uint64_t *pairs, uint64_t n, uint64_t *hi, uint64_t *lo)
{
    if (!n) return;
    hipLaunchKernelGGL(split_pairs_u64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, pairs, n, hi, lo);
}
__global__ void join_pairs_u64_kernel(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n,
                                      uint64_t *__restrict__ pairs)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        reinterpret_cast<ulonglong2 *>(pairs)[i] = make_ulonglong2(hi[i], lo[i]);
}
void launch_join_pairs_u64(hipStream_t s, const uint64_t *hi, const uint64_t *lo, uint64_t n, uint64_t *pairs)
{
    if (!n) return;
    hipLaunchKernelGGL(join_pairs_u64_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, hi, lo, n, pairs);
}
// ---- pooled merge of counted sets (dsk over more symbols than one pooled batch holds) ----
// sorted keys: head[i] = 1 where a run of equal keys starts
__global__ void runs_mark_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t *__restrict__ head)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// run id of entry i = incl[i] - 1 (inclusive scan of head): key to out_keys[run], count summed into out_counts[run]
// (saturating at 2^32 - 1 is not needed: a count is bounded by the symbols of the inputs, < 2^32 per chunk, and
// the sum is taken in 64 bits and clamped)
__global__ void runs_reduce_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ counts,
                                   const uint32_t *__restrict__ head, const uint32_t *__restrict__ incl, uint64_t n,
                                   uint64_t *__restrict__ out_keys, unsigned long long *__restrict__ out_counts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t run = incl[i] - 1;
        if (head[i]) out_keys[run] = keys[i];
        atomicAdd(&out_counts[run], (unsigned long long)counts[i]);
    }
}
__global__ void runs_keep_kernel(const unsigned long long *__restrict__ sums, uint64_t n_runs, uint32_t abundance_min,
                                 uint32_t *__restrict__ keep)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (uint64_t)gridDim.x * blockDim.x)
        keep[i] = sums[i] >= abundance_min ? 1u : 0u;
}
__global__ void runs_emit_kernel(const uint64_t *__restrict__ keys, const unsigned long long *__restrict__ sums,
                                 const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint64_t n_runs,
                                 uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_counts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (uint64_t)gridDim.x * blockDim.x)
        if (keep[i]) {
            out_keys[pos[i]] = keys[i];
            out_counts[pos[i]] = sums[i] > 0xffffffffull ? 0xffffffffu : (uint32_t)sums[i];
        }
}
void launch_runs_mark(hipStream_t s, const uint64_t *keys, uint64_t n, uint32_t *head)
{
    if (n) hipLaunchKernelGGL(runs_mark_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, keys, n, head);
}
void launch_runs_reduce(hipStream_t s, const uint64_t *keys, const uint32_t *counts, const uint32_t *head, const uint32_t *incl,
                        uint64_t n, uint64_t *out_keys, unsigned long long *out_counts)
{
    if (n) hipLaunchKernelGGL(runs_reduce_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, keys, counts, head, incl, n, out_keys, out_counts);
}
void launch_runs_keep(hipStream_t s, const unsigned long long *sums, uint64_t n_runs, uint32_t abundance_min, uint32_t *keep)
{
    if (n_runs) hipLaunchKernelGGL(runs_keep_kernel, dim3(grid_for(n_runs, 256)), dim3(256), 0, s, sums, n_runs, abundance_min, keep);
}
void launch_runs_emit(hipStream_t s, const uint64_t *keys, const unsigned long long *sums, const uint32_t *keep, const uint32_t *pos,
                      uint64_t n_runs, uint64_t *out_keys, uint32_t *out_counts)
{
    if (n_runs) hipLaunchKernelGGL(runs_emit_kernel, dim3(grid_for(n_runs, 256)), dim3(256), 0, s, keys, sums, keep, pos, n_runs, out_keys, out_counts);
}
void launch_column_errors(hipStream_t s, const uint64_t *matrix, uint64_t n_rows, uint64_t n_cols, const uint64_t *pos_mask,
                          const uint64_t *neg_mask, uint32_t n_pos, uint32_t n_train, uint32_t *errors, unsigned long long *hist)
{
    if (!n_cols) return;
    hipLaunchKernelGGL(column_errors_kernel, dim3(grid_for(n_cols, 256)), dim3(256), 0, s, matrix, n_rows, n_cols, pos_mask, neg_mask, n_pos,
                       n_train, errors, hist);
}
void launch_risk_index(hipStream_t s, const uint32_t *errors, uint64_t n_cols, const uint32_t *lut_presence, const uint32_t *lut_absence,
                       uint32_t *by_kmer, uint32_t *by_anti)
{
    if (!n_cols) return;
    hipLaunchKernelGGL(risk_index_kernel, dim3(grid_for(n_cols, 256)), dim3(256), 0, s, errors, n_cols, lut_presence, lut_absence, by_kmer,
                       by_anti);
}
void launch_iota_u32(hipStream_t s, uint32_t *p, uint64_t n)
{
    if (!n) return;
    hipLaunchKernelGGL(iota_u32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, n);
}

hipError_t set_max_dynamic_lds()
{
    // kernels that may ask for more than the default 64 KiB of dynamic LDS
    hipError_t e;
    const int max_lds = 160 * 1024;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kmer_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kmer_scatter_l1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kmer_scatter_l2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(bucket_dedup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(dict_build_kernel<1, 1024, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(dict_build_kernel<2, 1024, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(dict_build_kernel<4, 1024, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(dict_build_kernel<8, 512, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(dict_build_kernel<4, 1024, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(dict_build_kernel<8, 512, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(matrix_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds - 1024);
    if (e != hipSuccess) return e;
    return e;
}

}  // namespace grm
