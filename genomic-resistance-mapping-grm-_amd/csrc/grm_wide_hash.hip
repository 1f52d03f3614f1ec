// grm_wide_hash.hip -- the hash-partition pipeline for two-word k-mers (33 <= k <= 64).
//
// Same stages as grm_kernels.hip (two-level LDS-staged partition into fixed-capacity segments, or into histogram-sized
// ones as the fallback -> per-bucket LDS dictionary that also sets the presence bits -> the shared permutation fill),
// with 16-byte keys (hi, lo).  Differences that matter:
//   * tiles hold 4096 keys (64 KiB of LDS) instead of 8192;
//   * LDS tables are {lo[cap], hi[cap], state[cap]} and there is no 128-bit LDS compare-and-swap:
//     a slot is claimed with a 64-bit CAS on `lo`, then `hi` is published; a prober that meets a
//     claimed slot whose `hi` is not published yet simply goes round its loop again.  Lanes of one
//     wave reconverge every iteration, so the claimer always gets to publish: no spinning on a
//     lane of the same wave.
// Scope of this path: grm_batch_run with abundance-min 1 on one GPU (BASELINE config C5);
// counted sets, abundance filters and anything that overflows fall back to the sort-based path
// in grm_wide.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grm_device_fns.h"
#include "grm_internal.h"
#include "grm_coop.h"

namespace grm {

constexpr int WH_PPT = 8;                        // start positions / keys per thread
constexpr int WH_THREADS = 512;                  // 8 waves per tile, two tiles (64 KiB of keys each) per CU
constexpr int WH_TILE = WH_THREADS * WH_PPT;     // 4096 keys staged in LDS
// thread t owns bucket t of a tile (grm_internal.h, "Bucket ownership by thread id"); this path stays <= 2^13 buckets
static_assert(WH_THREADS >= (1 << L1_MAX_BITS) && WH_THREADS >= (1 << (MAX_HIST_BITS - L1_MAX_BITS)), "one thread per bucket of a level");
constexpr int WH_LDS_BYTES = WH_TILE * 16 + 2048 + 1024 + 1024 + 64;
constexpr uint64_t WH_EMPTY = ~0ull;             // lo == hi == ~0 is never a canonical k-mer
constexpr uint64_t WH_PENDING = ~0ull;           // hi of a slot that is claimed but not yet published

__device__ __forceinline__ uint64_t mix128(uint64_t hi, uint64_t lo)
{
    const uint64_t a = mix64(lo), b = mix64(hi ^ 0x5bd1e9955bd1e995ull);
    const uint32_t top = (uint32_t)(a >> 32) + (uint32_t)(b >> 32) * 0x9E3779B1u;
    const uint32_t low = (uint32_t)a ^ ((uint32_t)b << 7) ^ ((uint32_t)b >> 11);
    return ((uint64_t)top << 32) | low;
}

struct WideArgs {
    const uint64_t *sym2;
    const uint64_t *inv;
    uint64_t total_syms;
    const uint64_t *genome_sym_off;
    uint32_t n_genomes;
    int k;
    int bb;
};

// the WH_PPT start positions p0 .. of a thread; calls f(i, key) for the valid ones
template <typename F>
__device__ __forceinline__ uint32_t wide_positions(const WideArgs &a, uint64_t p0, F &&f)
{
    const int64_t nv = (int64_t)a.total_syms - a.k + 1 - (int64_t)p0;
    if (nv <= 0) return 0;
    const uint64_t grp = p0 >> 6;
    uint32_t valid = (uint32_t)(valid_starts(a.inv[grp], a.inv[grp + 1], a.k) >> (p0 & 63)) & ((1u << WH_PPT) - 1u);
    if (nv < WH_PPT) valid &= (1u << nv) - 1;
    if (valid) {
        const uint64_t wi = p0 >> 5;
        for_each_kmer_wide<WH_PPT>(a.sym2[wi], a.sym2[wi + 1], a.sym2[wi + 2], (int)(p0 & 31), valid, a.k, f);
    }
    return valid;
}

// ---- histogram --------------------------------------------------------------------------
__global__ __launch_bounds__(WH_THREADS) void wide_hist_kernel(WideArgs a, uint32_t n_tiles, uint32_t tiles_per_block,
                                                               uint32_t *__restrict__ counts)
{
    extern __shared__ uint32_t lds_hist[];
    const uint32_t B = 1u << a.bb;
    const uint64_t t_first = (uint64_t)blockIdx.x * tiles_per_block;
    if (t_first >= n_tiles) return;
    const uint64_t t_last = min(t_first + tiles_per_block, (uint64_t)n_tiles) - 1;
    const uint64_t p_first = t_first * WH_TILE;
    const uint64_t p_last = min((t_last + 1) * WH_TILE, a.total_syms) - 1;
    const uint32_t gen0 = genome_of(a.genome_sym_off, a.n_genomes, p_first);
    const bool uniform = a.genome_sym_off[gen0 + 1] > p_last;
    if (uniform) {
        for (uint32_t i = threadIdx.x; i < B; i += WH_THREADS) lds_hist[i] = 0;
        __syncthreads();
        for (uint64_t t = t_first; t <= t_last; t++)
            wide_positions(a, t * WH_TILE + (uint64_t)threadIdx.x * WH_PPT,
                           [&](int, K128 c) { atomicAdd(&lds_hist[hash_bucket(mix128(c.hi, c.lo), a.bb)], 1u); });
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < B; i += WH_THREADS) {
            const uint32_t c = lds_hist[i];
            if (c) atomicAdd(&counts[(uint64_t)gen0 * B + i], c);
        }
    } else {
        for (uint64_t t = t_first; t <= t_last; t++) {
            const uint64_t p0 = t * WH_TILE + (uint64_t)threadIdx.x * WH_PPT;
            if (p0 >= a.total_syms) continue;
            uint32_t gen = genome_of(a.genome_sym_off, a.n_genomes, p0);
            uint64_t gend = a.genome_sym_off[gen + 1];
            wide_positions(a, p0, [&](int i, K128 c) {
                while (p0 + (uint64_t)i >= gend) { gen++; gend = a.genome_sym_off[gen + 1]; }
                atomicAdd(&counts[(uint64_t)gen * B + hash_bucket(mix128(c.hi, c.lo), a.bb)], 1u);
            });
        }
    }
}

// ---- level 1: tile of 4096 positions -> coarse buckets ---------------------------------------
// region_stride != 0: fixed-capacity regions (slack layout, see kmer_scatter_l1_kernel), else regions from `off`
__global__ __launch_bounds__(WH_THREADS) void wide_l1_kernel(WideArgs a, int b1bits, uint32_t n_tiles,
                                                             const uint64_t *__restrict__ off, uint32_t *__restrict__ cursor1,
                                                             ulonglong2 *__restrict__ keys1, uint64_t region_stride,
                                                             int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    ulonglong2 *skeys = reinterpret_cast<ulonglong2 *>(lds_raw);                          // [WH_TILE]
    uint64_t *gbase = reinterpret_cast<uint64_t *>(lds_raw + (size_t)WH_TILE * 16);       // [256]
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds_raw + (size_t)WH_TILE * 16 + 2048); // [256]
    uint32_t *start = hist + 256;
    uint32_t *scratch = start + 256;
    const uint64_t tile = xcd_span(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int b2bits = a.bb - b1bits;
    const uint32_t B1 = 1u << b1bits;
    const uint64_t p_first = tile * WH_TILE;
    if (p_first >= a.total_syms) return;
    const uint64_t p_last = min(p_first + WH_TILE, a.total_syms) - 1;
    const uint64_t p0 = p_first + (uint64_t)threadIdx.x * WH_PPT;
    const uint32_t gen0 = genome_of(a.genome_sym_off, a.n_genomes, p_first);
    const bool uniform = a.genome_sym_off[gen0 + 1] > p_last;
    if (uniform) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        K128 kv[WH_PPT];
        uint32_t bk[WH_PPT], rk[WH_PPT];
        const uint32_t valid = wide_positions(a, p0, [&](int i, K128 c) {
            kv[i] = c;
            bk[i] = hash_bucket(mix128(c.hi, c.lo), b1bits);
            rk[i] = atomicAdd(&hist[bk[i]], 1u);
        });
        __syncthreads();
        const uint32_t c = threadIdx.x < B1 ? hist[threadIdx.x] : 0u;
        uint32_t n_tile;
        const uint32_t st = block_scan_sum(c, scratch, &n_tile);
        if (threadIdx.x < 256) start[threadIdx.x] = st;
        if (c) {
            const uint64_t cidx = (uint64_t)gen0 * B1 + threadIdx.x;
            const uint64_t region0 = region_stride ? cidx * region_stride
                                                   : off[(uint64_t)gen0 * (1ull << a.bb) + ((uint64_t)threadIdx.x << b2bits)];
            const uint32_t reserved = atomicAdd(&cursor1[cidx], c);
            const bool fits = !region_stride || (uint64_t)reserved + c <= region_stride;
            if (!fits) atomicExch(overflow, 1);
            gbase[threadIdx.x] = fits ? region0 + reserved : ~0ull;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WH_PPT; i++)
            if ((valid >> i) & 1u) skeys[start[bk[i]] + rk[i]] = make_ulonglong2(kv[i].lo, kv[i].hi);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_tile; i += WH_THREADS) {
            const ulonglong2 key = skeys[i];
            const uint32_t b1 = hash_bucket(mix128(key.y, key.x), b1bits);
            const uint64_t gb = gbase[b1];
            if (gb != ~0ull) keys1[gb + (i - start[b1])] = key;
        }
    } else {
        if (p0 >= a.total_syms) return;
        uint32_t gen = genome_of(a.genome_sym_off, a.n_genomes, p0);
        uint64_t gend = a.genome_sym_off[gen + 1];
        wide_positions(a, p0, [&](int i, K128 c) {
            while (p0 + (uint64_t)i >= gend) { gen++; gend = a.genome_sym_off[gen + 1]; }
            const uint32_t b1 = hash_bucket(mix128(c.hi, c.lo), b1bits);
            const uint64_t cidx = (uint64_t)gen * B1 + b1;
            const uint64_t region0 = region_stride ? cidx * region_stride : off[(uint64_t)gen * (1ull << a.bb) + ((uint64_t)b1 << b2bits)];
            const uint32_t at = atomicAdd(&cursor1[cidx], 1u);
            if (region_stride && at >= region_stride) atomicExch(overflow, 1);
            else keys1[region0 + at] = make_ulonglong2(c.lo, c.hi);
        });
    }
}

// ---- level 2: (genome, coarse bucket) region -> fine buckets ------------------------------------
__global__ __launch_bounds__(WH_THREADS) void wide_l2_kernel(const ulonglong2 *__restrict__ keys1, ulonglong2 *__restrict__ keys,
                                                             const uint64_t *__restrict__ off, uint64_t n_regions, int bb, int b1bits,
                                                             uint64_t region_stride, uint32_t fine_cap, const uint32_t *__restrict__ cursor1,
                                                             uint32_t *__restrict__ len_out, int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    ulonglong2 *skeys = reinterpret_cast<ulonglong2 *>(lds_raw);
    uint64_t *gbase = reinterpret_cast<uint64_t *>(lds_raw + (size_t)WH_TILE * 16);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds_raw + (size_t)WH_TILE * 16 + 2048);
    uint32_t *start = hist + 256;
    uint32_t *scratch = start + 256;
    const int b2bits = bb - b1bits;
    const uint32_t B2 = 1u << b2bits;
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
        uint64_t my_next = my_first;   // running output position of fine bucket t
        for (uint64_t base = r0; base < r1; base += WH_TILE) {
            const uint32_t n = (uint32_t)min((uint64_t)WH_TILE, r1 - base);
            if (threadIdx.x < B2) hist[threadIdx.x] = 0;
            __syncthreads();
            ulonglong2 kv[WH_PPT];
            uint32_t bk[WH_PPT], rk[WH_PPT];
#pragma unroll
            for (int j = 0; j < WH_PPT; j++) {
                const uint32_t i = (uint32_t)j * WH_THREADS + threadIdx.x;
                kv[j] = i < n ? keys1[base + i] : make_ulonglong2(WH_EMPTY, WH_EMPTY);
            }
#pragma unroll
            for (int j = 0; j < WH_PPT; j++) {
                if (!(kv[j].x == WH_EMPTY && kv[j].y == WH_EMPTY)) {
                    bk[j] = hash_bucket(mix128(kv[j].y, kv[j].x), bb) & (B2 - 1);
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
            for (int j = 0; j < WH_PPT; j++)
                if (!(kv[j].x == WH_EMPTY && kv[j].y == WH_EMPTY)) skeys[start[bk[j]] + rk[j]] = kv[j];
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += WH_THREADS) {
                const ulonglong2 key = skeys[i];
                const uint32_t b2 = hash_bucket(mix128(key.y, key.x), bb) & (B2 - 1);
                const uint64_t gb = gbase[b2];
                if (gb != ~0ull) keys[gb + (i - start[b2])] = key;
            }
            __syncthreads();
        }
        if (region_stride && threadIdx.x < B2) len_out[fine0 + threadIdx.x] = (uint32_t)(my_next - my_first);
    }
}

// ---- per-bucket dictionary with 16-byte keys ---------------------------------------------------
// returns the slot of (hi, lo), inserting it if absent; 0xffffffff when the table is full
__device__ __forceinline__ uint32_t wide_find_or_insert(ulonglong2 *tkey, uint32_t cap_mask, uint64_t hi, uint64_t lo, uint64_t h, bool *inserted)
{
    uint32_t slot = hash_slot(h, cap_mask);
    uint32_t probes = 0;
    *inserted = false;
    // every lane goes round this loop until it is done; a lane that meets a claimed-but-unpublished
    // slot re-reads it next time round (the claimer published in the meantime or will soon)
    for (uint32_t guard = 0; guard < 64u * (cap_mask + 1); guard++) {
        unsigned long long *plo = reinterpret_cast<unsigned long long *>(&tkey[slot].x);
        unsigned long long *phi = reinterpret_cast<unsigned long long *>(&tkey[slot].y);
        uint64_t cur = __hip_atomic_load(plo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == WH_EMPTY) {
            cur = atomicCAS(plo, (unsigned long long)WH_EMPTY, (unsigned long long)lo);
            if (cur == WH_EMPTY) {                                   // claimed: publish hi
                __hip_atomic_store(phi, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                *inserted = true;
                return slot;
            }
        }
        if (cur == lo) {
            const uint64_t ch = __hip_atomic_load(phi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (ch == hi) return slot;
            if (ch == WH_PENDING) continue;                          // not published yet: look again
        }
        slot = (slot + 1) & cap_mask;
        if (++probes > cap_mask) return 0xffffffffu;
    }
    return 0xffffffffu;
}

// NOTE: hi == WH_PENDING (all ones) together with a real lo is impossible for k <= 63 (hi has
// at most 62 significant bits); for k == 64 a canonical k-mer cannot start with 32 G's
// (its reverse complement would start with C's and be smaller), so hi is never all ones either.
//
// As dict_build_kernel (grm_kernels.hip): the union of a bucket over all genomes AND the presence bits, one word-row
// (64 genomes) between two barriers; words[slot] collects the bits of the current row, meta[slot] = entry id | SEEN |
// MULTI.  Entries leave in entry-id order at wg * cap (staged; the host gathers them densely in workgroup order, which
// is also the order of the exchange records), the words at matrix_s[wg][row][entry id].
// RECS (record form of the partition, grm_superkmer.hip): `keys` are the 24-byte run records level 2 left sorted by minimizer
// bucket, a segment = seg.off / seg.len in records.  A wave puts a chunk of 64 records into LDS with the prefix sum of their
// lengths and the bitmap of the positions where a record's k-mers start; a lane then takes a K-MER (its record = the popcount
// of the bitmap below its position), cut out of the record's three words (runw_kmer_at) -- as the counting stage does for
// one-word k-mers (record_count_kernel, grm_kernels.hip).
constexpr uint32_t WMETA_ID = 0x1fffu, WMETA_SEEN = 0x4000u, WMETA_MULTI = 0x8000u;
constexpr uint32_t WREC_WORDS = (64 * RUN_LMAX + 63) / 64;                  // bitmap words of a chunk's k-mer positions
// per wave: records, their memo ids, starts, bitmap, first record of every bitmap word
constexpr uint32_t WREC_SEGS = 16;                                           // segments a wave packs its chunks from at a time
constexpr uint32_t WREC_WAVE_BYTES = 64 * 24 + 64 * 2 + 64 * 2 + WREC_WORDS * 8 + WREC_WORDS * 4 + 8 + 64 + WREC_SEGS * 13 + 8;
struct WideRecStage {
    uint64_t *srec;         // [64][3]
    uint16_t *smid;         // [64] memo id of the record when this wave entered it (its k-mers' slots are noted), else 0xffff
    uint16_t *sstart;       // [64]
    uint64_t *starts;       // [WREC_WORDS]
    uint32_t *firstrec;     // [WREC_WORDS]
    uint8_t *sgb;           // [64] the record's genome inside the word-row (bit 63 - sgb)
    // the group of segments the chunks are packed from: records before the end of segment i (running), where record f of the group
    // stands (seg_base[i] + f), the segment's genome inside the word-row
    uint64_t *seg_base;     // [WREC_SEGS]
    uint32_t *seg_end;      // [WREC_SEGS]
    uint8_t *seg_gb;        // [WREC_SEGS]
};
// The RECORD MEMO, as dict_build's (grm_kernels.hip): genomes of a pan-genome hold the same runs, so the bucket's records are kept
// in a small LDS table with the table slots of their k-mers.  A record that is held costs ONE atomic OR into the record's own presence
// word -- no k-mer is cut out, hashed or looked up; at the end of a word-row the word of every held record goes to its k-mers'
// words.  The first occurrence of a record goes the direct way and notes the slots.  A record the memo has no room for (or whose
// bucket is being written) goes the direct way too: an accelerator, never a point of failure.
constexpr uint32_t WMEMO_ENT = 96, WMEMO_BUCKETS = 64, WMEMO_KS = 24, WMEMO_NONE = 0xffffu, WMEMO_LOCK = 0x0000ffffu;
constexpr uint32_t WMEMO_BYTES = WMEMO_BUCKETS * 16 + WMEMO_ENT * 24 + WMEMO_ENT * 8 + WMEMO_ENT * WMEMO_KS * 2 + 16;
struct WideMemo {
    uint32_t *slot;                 // buckets of 4: 0 = empty, WMEMO_LOCK = being written, else tag << 16 | id + 1
    uint64_t *rec;                  // [3 * WMEMO_ENT]
    unsigned long long *words;      // [WMEMO_ENT] presence word of the current word-row
    uint16_t *kslot;                // [WMEMO_KS * WMEMO_ENT] table slot of the record's k-mer t, 0xffff = none (not this sub-bucket's)
    uint32_t *ctl;                  // [0] records held
};
__device__ __forceinline__ uint32_t wmemo_hash(uint64_t r0, uint64_t r1, uint64_t r2)
{
    const uint32_t h = __umul24((uint32_t)r0, 0x9E3779u) + __umul24((uint32_t)(r0 >> 24), 0x85EBCBu) + __umul24((uint32_t)(r0 >> 48), 0xC2B2AFu) +
                       __umul24((uint32_t)r1, 0xD6E8FFu) + __umul24((uint32_t)(r1 >> 24), 0xA54FF5u) + __umul24((uint32_t)(r1 >> 48), 0x3C6EF3u) +
                       __umul24((uint32_t)(r2 >> 40), 0x7F4A7Du) + __umul24((uint32_t)(r2 >> 16), 0x94D049u) + __umul24((uint32_t)r2 & 0xffffu, 0xBF5847u);
    return h ^ (h >> 13);
}
// The common case, straight-line: the record's bucket holds it (the first slot with its tag).  ORs `bit` into its word and returns true then.
__device__ __forceinline__ bool wmemo_hit(const WideMemo &M, uint64_t r0, uint64_t r1, uint64_t r2, unsigned long long bit, uint32_t h)
{
    const uint32_t tag = (h >> 16) | 1u;
    const uint4 v = *reinterpret_cast<const uint4 *>(&M.slot[(h & (WMEMO_BUCKETS - 1)) << 2]);
    uint32_t e = 0;
    e = (v.w >> 16) == tag ? v.w : e;
    e = (v.z >> 16) == tag ? v.z : e;
    e = (v.y >> 16) == tag ? v.y : e;
    e = (v.x >> 16) == tag ? v.x : e;
    const uint32_t idp = e & 0xffffu;                   // id + 1 (0: no slot with the tag; a slot being written has tag 0)
    bool ok = idp != 0u;
    const uint32_t id = ok ? idp - 1u : 0u;
    ok = ok && M.rec[3 * id] == r0 && M.rec[3 * id + 1] == r1 && M.rec[3 * id + 2] == r2;
    if (ok) atomicOr(&M.words[id], bit);
    return ok;
}
// The rest (first occurrence of a record, a slot being written, a second slot with the same tag).
// 0: the record is held and `bit` went into its word; 1: this lane entered it (*id: its k-mers' slots are to be noted); 2: not held
__device__ __forceinline__ int wmemo_take(const WideMemo &M, uint64_t r0, uint64_t r1, uint64_t r2, unsigned long long bit, uint32_t h, uint32_t *id_out)
{
    const uint32_t tag = (h >> 16) | 1u;          // (never 0: an entry word is never 0 or the lock)
    uint32_t *bucket = &M.slot[(h & (WMEMO_BUCKETS - 1)) << 2];
    for (int round = 0; round < 4; round++) {
        bool again = false;
        for (int q = 0; q < 4; q++) {
            uint32_t v = lds_peek(&bucket[q]);
            if (v == 0) {
                if (lds_peek(&M.ctl[0]) >= WMEMO_ENT) return 2;
                v = atomicCAS(&bucket[q], 0u, WMEMO_LOCK);
                if (v == 0) {
                    const uint32_t id = atomicAdd(&M.ctl[0], 1u);
                    if (id >= WMEMO_ENT) return 2;                 // (the slot stays locked: whoever reaches it goes the direct way)
                    M.rec[3 * id] = r0; M.rec[3 * id + 1] = r1; M.rec[3 * id + 2] = r2;
                    for (uint32_t t = 0; t < WMEMO_KS; t += 4) *reinterpret_cast<uint64_t *>(&M.kslot[id * WMEMO_KS + t]) = ~0ull;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    __hip_atomic_store(&bucket[q], (tag << 16) | (id + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    *id_out = id;
                    return 1;
                }
            }
            if (v == WMEMO_LOCK) { again = true; continue; }       // being written (or abandoned): look again, then give up
            if ((v >> 16) == tag) {
                const uint32_t id = (v & 0xffffu) - 1u;
                if (M.rec[3 * id] == r0 && M.rec[3 * id + 1] == r1 && M.rec[3 * id + 2] == r2) {
                    atomicOr(&M.words[id], bit);
                    return 0;
                }
            }
        }
        if (!again) return 2;                                       // a full bucket without the record
    }
    return 2;
}
// A chunk of up to 64 records (one per lane, `have`; gb: the record's genome inside the word-row): those the memo holds are done
// with here; the others are put into LDS (in their order) for the lanes to take their k-mers.  Returns the k-mers staged.
__device__ __forceinline__ uint32_t wide_rec_stage(bool have, uint64_t r0, uint64_t r1, uint64_t r2, uint32_t gb, const WideRecStage &st,
                                                   const WideMemo &M)
{
    const int lane = lane_id();
    const unsigned long long bit = 1ull << (63u - gb);
    uint32_t mid = WMEMO_NONE;
    bool direct = have;
    const uint32_t h = wmemo_hash(r0, r1, r2);
    if (have && !wmemo_hit(M, r0, r1, r2, bit, h)) {
        uint32_t id = 0;
        const int how = wmemo_take(M, r0, r1, r2, bit, h, &id);
        direct = how != 0;
        if (how == 1) mid = id;
    } else {
        direct = false;
    }
    const uint64_t dm = __ballot(direct);
    const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm, 0u));
    const uint32_t n_direct = (uint32_t)__popcll(dm);
    if (!n_direct) return 0;                            // (uniform) the common case in a pan-genome: every record held
    const uint32_t ln = direct ? run_len(r2) : 0u;
    const uint32_t incl = wave_scan_incl_dpp(ln), s0 = incl - ln;
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (lane < (int)WREC_WORDS) { st.starts[lane] = 0; st.firstrec[lane] = n_direct; }
    if (direct) {
        st.srec[3 * pos] = r0; st.srec[3 * pos + 1] = r1; st.srec[3 * pos + 2] = r2;
        st.sstart[pos] = (uint16_t)s0;
        st.smid[pos] = (uint16_t)mid;
        st.sgb[pos] = (uint8_t)gb;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __asm__ volatile("" ::: "memory");
    if (direct) {
        atomicOr((unsigned long long *)&st.starts[s0 >> 6], 1ull << (s0 & 63u));
        atomicMin(&st.firstrec[s0 >> 6], pos);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __asm__ volatile("" ::: "memory");
    return tot;
}
// k-mer q of the staged records: (lo, hi) as the key segments hold them; o / t: its record among the staged ones, its number inside it
__device__ __forceinline__ ulonglong2 wide_rec_kmer(uint32_t q, int k, const WideRecStage &st, uint32_t &o, uint32_t &t)
{
    const uint32_t w = q >> 6;
    const uint64_t word = st.starts[w];
    o = st.firstrec[w] + (uint32_t)__popcll(word & ((2ull << (q & 63u)) - 1ull)) - 1u;
    t = q - st.sstart[o];
    RunW r;
    r.r[0] = st.srec[3 * o]; r.r[1] = st.srec[3 * o + 1]; r.r[2] = st.srec[3 * o + 2];
    const K128 key = runw_kmer_at(r, k, t);
    return make_ulonglong2(key.lo, key.hi);
}
template <bool RECS>
__global__ __launch_bounds__(TABLE_THREADS) void wide_dict_build_kernel(
    const ulonglong2 *__restrict__ keys, const SegLayout seg, int k, int part_bits, uint32_t n_genomes, int bb, int sb, uint32_t cap_log2,
    uint64_t *__restrict__ stage_lo, uint64_t *__restrict__ stage_hi, uint8_t *__restrict__ stage_flags,
    uint32_t *__restrict__ stage_cnt, uint64_t *__restrict__ matrix_s, uint16_t *__restrict__ birth, int *__restrict__ overflow,
    uint32_t *__restrict__ need)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const uint32_t cap = 1u << cap_log2, cap_mask = cap - 1;
    ulonglong2 *tkey = reinterpret_cast<ulonglong2 *>(lds_raw);        // slot = (lo, hi): one 16-byte LDS read per probe
    unsigned long long *words = reinterpret_cast<unsigned long long *>(lds_raw + (size_t)cap * 16);
    uint16_t *meta = reinterpret_cast<uint16_t *>(lds_raw + (size_t)cap * 24);
    uint32_t *scratch = reinterpret_cast<uint32_t *>(lds_raw + (size_t)cap * 26);
    uint8_t *stage_raw = lds_raw + (size_t)cap * 26 + TABLE_SCRATCH_BYTES;             // RECS: WREC_WAVE_BYTES per wave
    int *const full_p = reinterpret_cast<int *>(scratch + 16);        // (read with lds_peek: a volatile int would go through the flat path)
    auto is_full = [&]() { return lds_peek(full_p) != 0; };
    uint32_t &n_distinct = scratch[17];
    const uint32_t wg = blockIdx.x;
    const uint32_t B = 1u << bb;
    const uint32_t b = wg >> sb, sub = wg & ((1u << sb) - 1);
    const uint32_t G = n_genomes, n_rows = (G + 63) >> 6;
    for (uint32_t i = threadIdx.x; i < cap; i += blockDim.x) { tkey[i] = make_ulonglong2(WH_EMPTY, WH_PENDING); words[i] = 0; meta[i] = 0; }
    if (threadIdx.x == 0) { *full_p = 0; n_distinct = 0; }
    __syncthreads();
    const int lane = lane_id(), wave = wave_id(), nw = blockDim.x >> 6;        // nw = 8: divides 64
    const uint32_t max_fill = cap - (cap >> 3);
    const uint32_t per_row = 64u / (uint32_t)nw;
    // (RECS: a genome may be cut into 2^part_bits parts, each with segments of its own: gg counts parts then)
    auto seg_of = [&](uint32_t gg, uint64_t &s0, uint64_t &n) {
        const uint64_t idx = (uint64_t)gg * B + b;
        if (RECS) { s0 = seg.off[idx]; n = seg.len[idx] & 0xffffu; }
        else if (seg.off) { s0 = seg.off[idx]; n = seg.len ? (uint64_t)seg.len[idx] : seg.off[idx + 1] - s0; }
        else { s0 = idx * seg.stride; n = seg.len[idx]; }
    };
    WideRecStage st;
    WideMemo M;
    {
        uint8_t *mine = stage_raw + (size_t)wave_id() * WREC_WAVE_BYTES;
        st.srec = reinterpret_cast<uint64_t *>(mine);
        st.starts = reinterpret_cast<uint64_t *>(mine + 64 * 24);
        st.firstrec = reinterpret_cast<uint32_t *>(mine + 64 * 24 + WREC_WORDS * 8);
        st.sstart = reinterpret_cast<uint16_t *>(mine + 64 * 24 + WREC_WORDS * 12);
        st.smid = reinterpret_cast<uint16_t *>(mine + 64 * 24 + WREC_WORDS * 12 + 64 * 2);
        {
            uint8_t *more = mine + ((64 * 24 + WREC_WORDS * 12 + 64 * 4 + 7) & ~7u);
            st.seg_base = reinterpret_cast<uint64_t *>(more);
            st.seg_end = reinterpret_cast<uint32_t *>(more + WREC_SEGS * 8);
            st.seg_gb = more + WREC_SEGS * 12;
            st.sgb = more + WREC_SEGS * 13;
        }
        uint8_t *mraw = stage_raw + (size_t)(blockDim.x >> 6) * WREC_WAVE_BYTES;
        M.slot = reinterpret_cast<uint32_t *>(mraw);
        M.rec = reinterpret_cast<uint64_t *>(mraw + WMEMO_BUCKETS * 16);
        M.words = reinterpret_cast<unsigned long long *>(mraw + WMEMO_BUCKETS * 16 + WMEMO_ENT * 24);
        M.kslot = reinterpret_cast<uint16_t *>(mraw + WMEMO_BUCKETS * 16 + WMEMO_ENT * 32);
        M.ctl = reinterpret_cast<uint32_t *>(mraw + WMEMO_BUCKETS * 16 + WMEMO_ENT * 32 + WMEMO_ENT * WMEMO_KS * 2);
        if (RECS) {
            for (uint32_t i = threadIdx.x; i < WMEMO_BUCKETS * 4; i += blockDim.x) M.slot[i] = 0;
            for (uint32_t i = threadIdx.x; i < WMEMO_ENT; i += blockDim.x) M.words[i] = 0;
            if (threadIdx.x == 0) M.ctl[0] = 0;
            __syncthreads();
        }
    }
    const uint64_t *recs = reinterpret_cast<const uint64_t *>(keys);
    // bounds of the next genome's segment are requested while the current one is processed
    const int pb = RECS ? part_bits : 0;
    const uint32_t n_parts = 1u << pb;
    // key form: the bounds of the wave's next segments (genome wave + i * nw) are asked for two steps ahead
    auto seg_step = [&](uint32_t i, uint64_t &a, uint64_t &c) {
        const uint32_t gq = (uint32_t)wave + i * (uint32_t)nw;
        a = 0; c = 0;
        if (!RECS && gq < G) seg_of(gq, a, c);
    };
    uint32_t g = (uint32_t)wave, step = 0;
    uint64_t s0, n, s1, n1;
    seg_step(0, s0, n);
    seg_step(1, s1, n1);
    for (uint32_t r = 0; r < n_rows; r++) {
        if constexpr (RECS) {
            // The wave's segments of this word-row -- part p of genome r * 64 + wave + j * nw: t = j << pb | p -- in groups of WREC_SEGS:
            // their records are taken 64 at a time ACROSS the segments (a segment holds ~30 records: taken one segment at a time, half the
            // lanes of every chunk idled and the dictionary was bound by the instructions of its ~8 million chunks).  A lane finds the
            // segment of its record by a search over the group's running record counts.
            const uint32_t T = per_row << pb;
            for (uint32_t t0 = 0; t0 < T && !is_full(); t0 += WREC_SEGS) {
                {
                    const uint32_t t = t0 + (uint32_t)lane;
                    const uint32_t gq = r * 64u + (uint32_t)wave + (t >> pb) * (uint32_t)nw;
                    uint64_t a = 0, c = 0;
                    if ((uint32_t)lane < WREC_SEGS && t < T && gq < G) seg_of((gq << pb) + (t & (n_parts - 1)), a, c);
                    const uint32_t incl = wave_scan_incl_dpp((uint32_t)c);
                    if ((uint32_t)lane < WREC_SEGS) {
                        st.seg_end[lane] = incl;
                        st.seg_base[lane] = a - (uint64_t)(incl - (uint32_t)c);
                        st.seg_gb[lane] = (uint8_t)(gq & 63u);
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __asm__ volatile("" ::: "memory");
                }
                const uint32_t total = st.seg_end[WREC_SEGS - 1];
                // a batch of 64 records is asked for while the batch before it is worked on (the loads are unconditional -- index clamped,
                // validity applied at use -- so that nothing but the wait for THIS batch stands between them and the work)
                auto fetch = [&](uint32_t f0, uint64_t &q0, uint64_t &q1, uint64_t &q2, uint32_t &qgb) {
                    const uint32_t f = min(f0 + (uint32_t)lane, total ? total - 1u : 0u);
                    uint32_t lo = 0, hi = WREC_SEGS - 1;            // first segment whose running count exceeds f
#pragma unroll
                    for (int it = 0; it < 4; it++) {
                        const uint32_t mid = (lo + hi) >> 1;
                        const bool below = st.seg_end[mid] > f;
                        hi = below ? mid : hi;
                        lo = below ? lo : min(mid + 1, WREC_SEGS - 1);
                    }
                    const uint64_t at = total ? st.seg_base[lo] + f : 0;
                    qgb = st.seg_gb[lo];
                    q0 = recs[3 * at]; q1 = recs[3 * at + 1]; q2 = recs[3 * at + 2];
                };
                uint64_t n0 = 0, n1 = 0, n2 = 0;
                uint32_t ngb = 0;
                if (total) fetch(0, n0, n1, n2, ngb);
                for (uint32_t f0 = 0; f0 < total && !is_full(); f0 += 64) {
                    const bool have = f0 + (uint32_t)lane < total;
                    uint64_t r0 = n0, r1 = n1, r2 = n2;
                    const uint32_t gb = ngb;
                    __asm__ volatile("" ::"v"(r0), "v"(r1), "v"(r2));          // (the wait for this batch stands here, before the next one is asked for)
                    if (f0 + 64 < total) fetch(f0 + 64, n0, n1, n2, ngb);
                    if (!have) { r0 = 0; r1 = 0; r2 = 0; }
                    const uint64_t n_keys = wide_rec_stage(have, r0, r1, r2, gb, st, M);
                    constexpr int KJ = 2;
                    for (uint64_t i0 = lane; i0 < n_keys + (uint64_t)lane && !is_full(); i0 += 64 * KJ) {        // (uniform trip count)
                        ulonglong2 kv[KJ];
                        uint64_t hv[KJ];
                        uint32_t sl[KJ];
                        uint32_t note[KJ];               // where in the memo the k-mer's slot is noted (its record entered by this wave), else ~0
                        unsigned long long kbit[KJ];
#pragma unroll
                        for (int j = 0; j < KJ; j++) {
                            const uint64_t i = i0 + 64u * j;
                            note[j] = ~0u;
                            kbit[j] = 0;
                            kv[j] = make_ulonglong2(WH_EMPTY, WH_EMPTY);
                            if (i < n_keys) {
                                uint32_t o, t;
                                kv[j] = wide_rec_kmer((uint32_t)i, k, st, o, t);
                                const uint32_t mid = st.smid[o];
                                if (mid != WMEMO_NONE) note[j] = mid * WMEMO_KS + t;
                                kbit[j] = 1ull << (63u - st.sgb[o]);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < KJ; j++) {
                            hv[j] = mix128(kv[j].y, kv[j].x);
                            sl[j] = hash_slot(hv[j], cap_mask);
                        }
                        ulonglong2 c0[KJ], c1[KJ];
#pragma unroll
                        for (int j = 0; j < KJ; j++) {
                            c0[j] = tkey[sl[j]];
                            c1[j] = tkey[(sl[j] + 1) & cap_mask];
                        }
                        uint32_t todo = 0;
#pragma unroll
                        for (int j = 0; j < KJ; j++) {
                            const bool active = !(kv[j].x == WH_EMPTY && kv[j].y == WH_EMPTY) && (!sb || hash_sub(hv[j], bb, sb) == sub);
                            const bool hit0 = c0[j].x == kv[j].x && c0[j].y == kv[j].y;
                            const bool hit1 = c1[j].x == kv[j].x && c1[j].y == kv[j].y;
                            const uint32_t at2 = hit1 ? ((sl[j] + 1) & cap_mask) : sl[j];
                            if (active && (hit0 | hit1)) {
                                __hip_atomic_fetch_or(&words[at2], kbit[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (note[j] != ~0u) M.kslot[note[j]] = (uint16_t)at2;
                            }
                            todo |= (uint32_t)(active && !(hit0 | hit1)) << j;
                        }
                        while (todo) {
                            if (is_full()) break;
                            const int j = __ffs(todo) - 1;
                            todo &= todo - 1;
                            ulonglong2 key = kv[0];
                            uint64_t h = hv[0];
                            uint32_t nt = note[0];
                            unsigned long long kb = kbit[0];
#pragma unroll
                            for (int q = 1; q < KJ; q++) { if (j == q) { key = kv[q]; h = hv[q]; nt = note[q]; kb = kbit[q]; } }
                            bool ins;
                            const uint32_t slot = wide_find_or_insert(tkey, cap_mask, key.y, key.x, h, &ins);
                            bool over = slot == 0xffffffffu;
                            if (!over) {
                                if (ins) {
                                    const uint32_t id = atomicAdd(&n_distinct, 1u);
                                    meta[slot] = (uint16_t)(id & WMETA_ID);
                                    if (birth && id < cap) birth[((uint64_t)wg << cap_log2) + id] = (uint16_t)r;
                                    over = id >= max_fill;
                                }
                                __hip_atomic_fetch_or(&words[slot], kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (nt != ~0u) M.kslot[nt] = (uint16_t)slot;
                            }
                            if (over) {
                                __hip_atomic_store(full_p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                break;
                            }
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xC07F);      // the staging arrays are free again
                    __asm__ volatile("" ::: "memory");
                }
            }
        } else {
        for (uint32_t jr = 0; jr < per_row; jr++, g += nw) {
            uint64_t s0_next = s1, n_next = n1;
            seg_step(step + 2, s1, n1);
            step++;
            if (g < G && !is_full()) {
                const unsigned long long bit = 1ull << (63 - (g & 63));
                // straight-line and predicated, as dict_build's probe (grm_kernels.hip): both probe slots of every key are
                // read, a key found there ORs its bit in under a predicate; only a key that is in neither slot goes round
                // the insertion loop (a bit mask of the lane's keys still to do, no per-key branches)
                constexpr int KJ = 2;                    // keys per lane in flight (segments hold ~300 keys)
                for (uint64_t i0 = lane; i0 < n && !is_full(); i0 += 64 * KJ) {
                    ulonglong2 kv[KJ];
                    uint64_t hv[KJ];
                    uint32_t sl[KJ];
#pragma unroll
                    for (int j = 0; j < KJ; j++) {
                        const uint64_t i = i0 + 64u * j;
                        kv[j] = i < n ? keys[s0 + i] : make_ulonglong2(WH_EMPTY, WH_EMPTY);
                    }
#pragma unroll
                    for (int j = 0; j < KJ; j++) {
                        hv[j] = mix128(kv[j].y, kv[j].x);
                        sl[j] = hash_slot(hv[j], cap_mask);
                    }
                    ulonglong2 c0[KJ], c1[KJ];
#pragma unroll
                    for (int j = 0; j < KJ; j++) {
                        c0[j] = tkey[sl[j]];
                        c1[j] = tkey[(sl[j] + 1) & cap_mask];
                    }
                    uint32_t todo = 0;
#pragma unroll
                    for (int j = 0; j < KJ; j++) {
                        const bool active = !(kv[j].x == WH_EMPTY && kv[j].y == WH_EMPTY) && (!sb || hash_sub(hv[j], bb, sb) == sub);
                        const bool hit0 = c0[j].x == kv[j].x && c0[j].y == kv[j].y;
                        const bool hit1 = c1[j].x == kv[j].x && c1[j].y == kv[j].y;
                        const uint32_t at = hit1 ? ((sl[j] + 1) & cap_mask) : sl[j];
                        if (active && (hit0 | hit1)) __hip_atomic_fetch_or(&words[at], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        todo |= (uint32_t)(active && !(hit0 | hit1)) << j;
                    }
                    while (todo) {
                        if (is_full()) break;
                        const int j = __ffs(todo) - 1;
                        todo &= todo - 1;
                        ulonglong2 key = kv[0];
                        uint64_t h = hv[0];
#pragma unroll
                        for (int q = 1; q < KJ; q++) { if (j == q) { key = kv[q]; h = hv[q]; } }
                        bool ins;
                        const uint32_t slot = wide_find_or_insert(tkey, cap_mask, key.y, key.x, h, &ins);
                        bool over = slot == 0xffffffffu;
                        if (!over) {
                            if (ins) {
                                const uint32_t id = atomicAdd(&n_distinct, 1u);
                                meta[slot] = (uint16_t)(id & WMETA_ID);
                                if (birth && id < cap) birth[((uint64_t)wg << cap_log2) + id] = (uint16_t)r;
                                over = id >= max_fill;
                            }
                            __hip_atomic_fetch_or(&words[slot], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (over) {
                            __hip_atomic_store(full_p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);    // (what it would have needed is counted after the word-row loop)
                            break;
                        }
                    }
                }
            }
            s0 = s0_next;
            n = n_next;
        }
        }
        __syncthreads();
        if (is_full()) break;    // read between two barriers: uniform
        if (RECS) {
            // the words of the held records go to their k-mers' words
            const uint32_t held = min(M.ctl[0], WMEMO_ENT);
            for (uint32_t e = threadIdx.x; e < held * WMEMO_KS; e += blockDim.x) {
                const uint32_t id = e / WMEMO_KS;
                const unsigned long long w = M.words[id];
                const uint32_t ks = M.kslot[e];
                if (w && ks != 0xffffu) __hip_atomic_fetch_or(&words[ks], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
            for (uint32_t id = threadIdx.x; id < held; id += blockDim.x) M.words[id] = 0;
        }
        for (uint32_t slot = threadIdx.x; slot < cap; slot += blockDim.x) {
            if (tkey[slot].x == WH_EMPTY) continue;
            const unsigned long long wd = words[slot];
            const uint32_t m = meta[slot];
            if (matrix_s) matrix_s[(((uint64_t)wg * n_rows + r) << cap_log2) + (m & WMETA_ID)] = wd;
            if (wd) {
                const bool multi = (m & WMETA_SEEN) || (wd & (wd - 1));
                meta[slot] = (uint16_t)(m | WMETA_SEEN | (multi ? WMETA_MULTI : 0u));
                words[slot] = 0;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (is_full()) {
        // what the workgroup would have needed: all its keys once more through a HyperLogLog sketch in the abandoned table's
        // LDS (see dict_build_kernel, grm_kernels.hip)
        const uint32_t HLL_M = cap < 4096u ? cap : 4096u;
        uint32_t *hll = reinterpret_cast<uint32_t *>(lds_raw);
        uint64_t *scr64 = reinterpret_cast<uint64_t *>(lds_raw + (size_t)HLL_M * 4);
        for (uint32_t i = threadIdx.x; i < HLL_M; i += blockDim.x) hll[i] = 0;
        __syncthreads();
        for (uint32_t gg = (uint32_t)wave; gg < G * n_parts; gg += (uint32_t)nw) {
            uint64_t sv = 0, nv = 0;
            seg_of(gg, sv, nv);
            for (uint64_t i = lane; i < nv; i += 64) {
                if (RECS) {                 // (a lane rolls through its record: this is the rare way out)
                    RunW rw;
                    rw.r[0] = recs[3 * (sv + i)]; rw.r[1] = recs[3 * (sv + i) + 1]; rw.r[2] = recs[3 * (sv + i) + 2];
                    for (uint32_t t = 0; t < run_len(rw.r[2]); t++) {
                        const K128 key = runw_kmer_at(rw, k, t);
                        const uint64_t h = mix128(key.hi, key.lo);
                        if (sb && hash_sub(h, bb, sb) != sub) continue;
                        const uint32_t lo = (uint32_t)h;
                        atomicMax(&hll[lo & (HLL_M - 1)], (uint32_t)__clz((lo >> 12) | 1u) - 11u);
                    }
                    continue;
                }
                const ulonglong2 key = keys[sv + i];
                const uint64_t h = mix128(key.y, key.x);
                if (sb && hash_sub(h, bb, sb) != sub) continue;
                const uint32_t lo = (uint32_t)h;
                atomicMax(&hll[lo & (HLL_M - 1)], (uint32_t)__clz((lo >> 12) | 1u) - 11u);
            }
        }
        __syncthreads();
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
            if (est <= 2.5 * m && nz) est = m * log(m / (double)nz);
            atomicMax(need, (uint32_t)min(est * 1.05, 4.0e9));
            atomicExch(overflow, 1);
            stage_cnt[wg] = 0;
        }
        return;
    }
    const uint64_t out0 = (uint64_t)wg * cap;
    for (uint32_t slot = threadIdx.x; slot < cap; slot += blockDim.x) {
        const ulonglong2 key = tkey[slot];
        if (key.x == WH_EMPTY) continue;
        const uint32_t m = meta[slot];
        stage_lo[out0 + (m & WMETA_ID)] = key.x;
        stage_hi[out0 + (m & WMETA_ID)] = key.y;
        stage_flags[out0 + (m & WMETA_ID)] = (m & WMETA_MULTI) ? 2 : 1;
    }
    if (threadIdx.x == 0) stage_cnt[wg] = n_distinct;
}

// dense (hi, lo, flag) lists from the staged per-workgroup dictionaries
__global__ void wide_dict_gather_kernel(const uint64_t *__restrict__ stage_lo, const uint64_t *__restrict__ stage_hi,
                                        const uint8_t *__restrict__ stage_flags, const uint64_t *__restrict__ stage_off, uint32_t cap,
                                        uint64_t *__restrict__ out_lo, uint64_t *__restrict__ out_hi, uint8_t *__restrict__ out_flags)
{
    const uint32_t wg = blockIdx.x;
    const uint64_t o = stage_off[wg];
    const uint32_t n = (uint32_t)(stage_off[wg + 1] - o);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        out_lo[o + i] = stage_lo[(uint64_t)wg * cap + i];
        out_hi[o + i] = stage_hi[(uint64_t)wg * cap + i];
        out_flags[o + i] = stage_flags[(uint64_t)wg * cap + i];
    }
}
// keep flags of the value-sorted dictionary (single GPU: keys are already unique)
// sorted (hi, lo) with the flag of entry i at flags[order[i]].  A run of equal keys (the same k-mer
// reported by several ranks, whose genomes are disjoint) is ONE column carried by several genomes;
// keep[i] = 1 on the first entry of every run that survives the singleton filter.
__global__ void wide_mark_kernel(const uint64_t *__restrict__ s_hi, const uint64_t *__restrict__ s_lo, const uint8_t *__restrict__ flags,
                                 const uint32_t *__restrict__ order, uint64_t n, int filter_singleton, uint32_t *__restrict__ keep)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t hi = s_hi[i], lo = s_lo[i];
        const bool head = i == 0 || s_hi[i - 1] != hi || s_lo[i - 1] != lo;
        const bool several = flags[order[i]] >= 2 || (i + 1 < n && s_hi[i + 1] == hi && s_lo[i + 1] == lo);
        keep[i] = (head && (!filter_singleton || several)) ? 1u : 0u;
    }
}
// final dictionary (hi, lo interleaved, ascending)
__global__ void wide_select_kernel(const uint64_t *__restrict__ s_hi, const uint64_t *__restrict__ s_lo,
                                   const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint64_t n,
                                   uint64_t *__restrict__ dict)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!keep[i]) continue;
        const uint32_t c = pos[i];
        dict[2ull * c] = s_hi[i];
        dict[2ull * c + 1] = s_lo[i];
    }
}
// one GPU: the sorted entries ARE the local entries, each exactly once, and order[i] is the local index of sorted entry i
__global__ void wide_cols_from_order_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ keep,
                                            const uint32_t *__restrict__ pos, uint64_t n, uint32_t *__restrict__ entry_col)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        entry_col[order[i]] = keep[i] ? pos[i] : 0xffffffffu;
}
// column of every local entry: binary search of its (hi, lo) among the n sorted entries of all ranks; the first entry of
// its run says whether the k-mer was kept and which column it got (0xffffffff: filtered out)
__global__ void wide_entry_cols_kernel(const uint64_t *__restrict__ s_hi, const uint64_t *__restrict__ s_lo,
                                       const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint64_t n,
                                       const uint64_t *__restrict__ e_hi, const uint64_t *__restrict__ e_lo, uint64_t n_entries,
                                       uint32_t *__restrict__ entry_col)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t hi = e_hi[e], lo = e_lo[e];
        uint64_t a = 0, b = n;
        while (a < b) {
            const uint64_t m = (a + b) >> 1;
            const uint64_t mh = s_hi[m];
            if (mh < hi || (mh == hi && s_lo[m] < lo)) a = m + 1; else b = m;
        }
        entry_col[e] = (a < n && s_hi[a] == hi && s_lo[a] == lo && keep[a]) ? pos[a] : 0xffffffffu;
    }
}

// ---- launchers -------------------------------------------------------------------------------
static WideArgs wargs(const KmerLaunch &L)
{
    WideArgs a;
    a.sym2 = L.sym2; a.inv = L.inv; a.total_syms = L.total_syms; a.genome_sym_off = L.genome_sym_off;
    a.n_genomes = L.n_genomes; a.k = L.k; a.bb = L.bb;
    return a;
}
void launch_wh_hist(hipStream_t s, const KmerLaunch &L, uint32_t *counts)
{
    if (!L.total_syms) return;
    const uint32_t n_tiles = (uint32_t)((L.total_syms + WH_TILE - 1) / WH_TILE);
    const uint32_t tpb = 16;       // 64 Ki positions per LDS histogram flush
    hipLaunchKernelGGL(wide_hist_kernel, dim3((n_tiles + tpb - 1) / tpb), dim3(WH_THREADS), (size_t)4 << L.bb, s, wargs(L), n_tiles, tpb, counts);
}
void launch_wh_l1(hipStream_t s, const KmerLaunch &L, const uint64_t *off, uint32_t *cursor1, void *out, uint64_t region_stride, int *overflow)
{
    if (!L.total_syms) return;
    const uint32_t n_tiles = (uint32_t)((L.total_syms + WH_TILE - 1) / WH_TILE);
    const uint32_t grid = ((n_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(wide_l1_kernel, dim3(grid), dim3(WH_THREADS), WH_LDS_BYTES, s, wargs(L), scatter_b1_bits(L.bb), n_tiles, off, cursor1,
                       reinterpret_cast<ulonglong2 *>(out), region_stride, overflow);
}
void launch_wh_l2(hipStream_t s, const KmerLaunch &L, const uint64_t *off, const void *keys1, void *keys, uint64_t region_stride,
                  uint32_t fine_cap, const uint32_t *cursor1, uint32_t *len_out, int *overflow)
{
    const int b1 = scatter_b1_bits(L.bb);
    if (!L.total_syms || L.bb <= b1) return;
    const uint64_t n_regions = (uint64_t)L.n_genomes << b1;
    const uint32_t grid = (uint32_t)(n_regions < 256u * 16u ? n_regions : 256u * 16u);
    hipLaunchKernelGGL(wide_l2_kernel, dim3(grid), dim3(WH_THREADS), WH_LDS_BYTES, s, reinterpret_cast<const ulonglong2 *>(keys1),
                       reinterpret_cast<ulonglong2 *>(keys), off, n_regions, L.bb, b1, region_stride, fine_cap, cursor1, len_out, overflow);
}
// recs_k: 0 = `keys` are 16-byte keys; else k, and `keys` are 24-byte run records (seg.off / seg.len in records)
void launch_wh_dict_build(hipStream_t s, const void *keys, const SegLayout &seg, uint32_t n_genomes, int bb, int sb, uint32_t cap_log2,
                          uint64_t *stage_lo, uint64_t *stage_hi, uint8_t *stage_flags, uint32_t *stage_cnt, uint64_t *matrix_s,
                          uint16_t *birth, int *overflow, uint32_t *need, int recs_k, int part_bits)
{
    const size_t lds = (((size_t)26) << cap_log2) + TABLE_SCRATCH_BYTES + (recs_k ? (size_t)(TABLE_THREADS / 64) * WREC_WAVE_BYTES + WMEMO_BYTES : 0);
    if (recs_k)
        hipLaunchKernelGGL(wide_dict_build_kernel<true>, dim3(1u << (bb + sb)), dim3(TABLE_THREADS), lds, s, reinterpret_cast<const ulonglong2 *>(keys),
                           seg, recs_k, part_bits, n_genomes, bb, sb, cap_log2, stage_lo, stage_hi, stage_flags, stage_cnt, matrix_s, birth, overflow, need);
    else
        hipLaunchKernelGGL(wide_dict_build_kernel<false>, dim3(1u << (bb + sb)), dim3(TABLE_THREADS), lds, s, reinterpret_cast<const ulonglong2 *>(keys),
                           seg, 0, 0, n_genomes, bb, sb, cap_log2, stage_lo, stage_hi, stage_flags, stage_cnt, matrix_s, birth, overflow, need);
}
void launch_wh_dict_gather(hipStream_t s, const uint64_t *stage_lo, const uint64_t *stage_hi, const uint8_t *stage_flags,
                           const uint64_t *stage_off, uint32_t n_wg, uint32_t cap, uint64_t *out_lo, uint64_t *out_hi, uint8_t *out_flags)
{
    hipLaunchKernelGGL(wide_dict_gather_kernel, dim3(n_wg), dim3(256), 0, s, stage_lo, stage_hi, stage_flags, stage_off, cap, out_lo, out_hi,
                       out_flags);
}
// The dictionary of two-word k-mers through the key-range sort of one-word ones (grm_dictsort.hip): sorted by the TOP 64 bits of the
// 2k (the first 32 bases) with the entries' indices; entries that agree there -- the two alleles of a SNP in the k-mer's second half --
// stand together then, and a thread per such group puts its indices in the order of the low words.
__global__ void wide_top64_kernel(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int k, uint64_t *__restrict__ top)
{
    const int up = 128 - 2 * k;              // 0 .. 62
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        top[i] = up ? (hi[i] << up) | (lo[i] >> (64 - up)) : hi[i];
}
// A group of entries that agree in their top 64 bits is put in order by ONE thread (an insertion sort through global memory): fine for
// the pairs and triples a dictionary holds (the two alleles of a SNP in a k-mer's second half; one copy per rank in the multi-rank
// merge), quadratic for long groups -- k = 64 repeats that share their first 32 bases, low-complexity flanks.  A group longer than
// WIDE_TIE_MAX raises *too_long and is left alone: the host then takes the two-pass radix sort.
constexpr uint32_t WIDE_TIE_MAX = 32;
__global__ void wide_ties_kernel(const uint64_t *__restrict__ top_sorted, uint32_t *__restrict__ order, const uint64_t *__restrict__ lo, uint64_t n,
                                 int *__restrict__ too_long)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t t = top_sorted[i];
        if ((i && top_sorted[i - 1] == t) || i + 1 >= n || top_sorted[i + 1] != t) continue;       // not the first of a group of two or more
        uint64_t e = i + 2;
        while (e < n && e - i <= WIDE_TIE_MAX && top_sorted[e] == t) e++;
        if (e - i > WIDE_TIE_MAX) { atomicExch(too_long, 1); continue; }
        for (uint64_t a = i + 1; a < e; a++) {           // insertion sort of the group's indices by the low word
            const uint32_t ia = order[a];
            const uint64_t la = lo[ia];
            uint64_t b = a;
            while (b > i && lo[order[b - 1]] > la) { order[b] = order[b - 1]; b--; }
            order[b] = ia;
        }
    }
}
void launch_wh_top64(hipStream_t s, const uint64_t *hi, const uint64_t *lo, uint64_t n, int k, uint64_t *top)
{
    if (!n) return;
    const uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(wide_top64_kernel, dim3((uint32_t)(g > 8192 ? 8192 : g)), dim3(256), 0, s, hi, lo, n, k, top);
}
void launch_wh_ties(hipStream_t s, const uint64_t *top_sorted, uint32_t *order, const uint64_t *lo, uint64_t n, int *too_long)
{
    if (!n) return;
    const uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(wide_ties_kernel, dim3((uint32_t)(g > 8192 ? 8192 : g)), dim3(256), 0, s, top_sorted, order, lo, n, too_long);
}
void launch_wh_mark(hipStream_t s, const uint64_t *s_hi, const uint64_t *s_lo, const uint8_t *flags, const uint32_t *order, uint64_t n,
                    int filter_singleton, uint32_t *keep)
{
    if (!n) return;
    const uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(wide_mark_kernel, dim3((uint32_t)(g > 8192 ? 8192 : g)), dim3(256), 0, s, s_hi, s_lo, flags, order, n, filter_singleton, keep);
}
void launch_wh_select(hipStream_t s, const uint64_t *s_hi, const uint64_t *s_lo, const uint32_t *keep, const uint32_t *pos, uint64_t n,
                      uint64_t *dict)
{
    if (!n) return;
    uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(wide_select_kernel, dim3((uint32_t)(g > 8192 ? 8192 : g)), dim3(256), 0, s, s_hi, s_lo, keep, pos, n, dict);
}
void launch_wh_cols_from_order(hipStream_t s, const uint32_t *order, const uint32_t *keep, const uint32_t *pos, uint64_t n, uint32_t *entry_col)
{
    if (!n) return;
    const uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(wide_cols_from_order_kernel, dim3((uint32_t)(g > 8192 ? 8192 : g)), dim3(256), 0, s, order, keep, pos, n, entry_col);
}
void launch_wh_entry_cols(hipStream_t s, const uint64_t *s_hi, const uint64_t *s_lo, const uint32_t *keep, const uint32_t *pos, uint64_t n,
                          const uint64_t *e_hi, const uint64_t *e_lo, uint64_t n_entries, uint32_t *entry_col)
{
    if (!n_entries) return;
    uint64_t g = (n_entries + 255) / 256;
    hipLaunchKernelGGL(wide_entry_cols_kernel, dim3((uint32_t)(g > 8192 ? 8192 : g)), dim3(256), 0, s, s_hi, s_lo, keep, pos, n, e_hi, e_lo,
                       n_entries, entry_col);
}
hipError_t wh_set_max_dynamic_lds()
{
    const int max_lds = 159 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wide_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(wide_l1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(wide_l2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(wide_dict_build_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(wide_dict_build_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
}

}  // namespace grm
