// grm_superkmer.hip -- record form of the partition (11 <= k <= 32, abundance-min 1, matrix path).
//
// The key form (grm_kernels.hip, levels 1 and 2) moves every canonical k-mer through HBM as an 8-byte key: written by
// level 1, read and written by level 2, read by the consumer -- 160 GB per 1000 x 5 Mbp.  Here the bucket of a k-mer
// is a function of its MINIMIZER -- the canonical 11-mer with the smallest order value among the k - 10 it contains -- so
// consecutive k-mers of a sequence share their bucket (what DSK itself does with its minimizer partitions [EXT]).  A RUN =
// the consecutive valid k-mer starts that share one minimizer OCCURRENCE (a super-k-mer: at most k - 10 k-mers, 11 on
// average at k = 31) travels as ONE 16-byte record that carries its own bases (layout: grm_device_fns.h, "run records"):
// 8 B per k-mer become ~1.5 B.  Where a run starts and ends depends on the sequence alone, and a record is stored in the
// orientation in which its minimizer is canonical: genomes that share a stretch of sequence share its records whatever the
// contig order, strand or an indel upstream -- which is what dict_build's record memo lives on.
// Level 2 owns a (genome part, coarse bucket) region: it sorts the region's records by fine bucket (and short / long) for
// dict_build's record form, which decodes them itself; for a consumer that needs keys (the probing fill) a second form of
// level 2 EXPANDS them to canonical k-mers on the way out, leaving the same bucket-sorted key segments as the key form.
// A k-mer and its reverse complement contain the same canonical 11-mers, so the bucket is a function of the canonical
// k-mer, on every rank alike (minimizer_bucket_of_kmer re-derives it from a key where a consumer has only the key).
#include "grm_internal.h"
#include "grm_device_fns.h"
#include "grm_coop.h"
#include <algorithm>

namespace grm {

constexpr int SK_THREADS = 1024;
constexpr int SK_PPT = RUN_PPT;                       // k-mer start positions per thread and step: one packed word
constexpr int SK_LMAX = RUN_LMAX;                     // k-mers per record, at most
static_assert(SK_M == 11, "run_minimizers<W> is instantiated with its default m-mer length");
constexpr int SK_FINE_BITS = RUN_FINE_BITS;           // a record carries 7 bucket bits below the coarse ones, whatever the bucket count in use:
                                                      // level 2 takes the top bb - b1 of them, so more buckets only need level 2 again
constexpr int SK_MAX_COARSE = 9;
constexpr int SK_MAX_BITS = SK_MAX_COARSE + SK_FINE_BITS;         // at most 9 coarse bits (cursors) + the fine field
// A wave takes 62 consecutive windows per step, one per lane.  The W - 1 m-mers a window's last k-mers reach into the next
// window are hashed by that window's lane and come over with a whole-wave DPP shift (8 instructions per m-mer otherwise: 40 %
// of them would be hashed twice), and the run that crosses a window's end is finished with what the lane to the right knows.
// Lane 62 works on the window after the 62 (the next wave's, or the next step's, first) only to say how far the run that crosses
// into it goes on; lane 63 on the one after that, only to hash.
constexpr int SK_WAVE_WINDOWS = 62;
constexpr int SK_STEP_WINDOWS = (SK_THREADS / 64) * SK_WAVE_WINDOWS;
// runs a wave can hand round among its lanes per step (a wave with more -- every other position a run's first: k = 11, or junk -- takes
// them lane by lane)
constexpr int SK_WAVE_RUNS = 320;
constexpr int SK_WAVES = SK_THREADS / 64;
constexpr int SK_VAL_PITCH = SK_THREADS + 1;          // minimizer words in LDS: position i of thread t at [i * pitch + t] -- a column per thread when it writes,
                                                      // and no two positions of one thread in one bank when the lanes of its wave read them back
constexpr int SK_WORD_PITCH = 66;                     // window words a wave keeps for handing runs round: its 64 lanes' + the two behind them (two-word k-mers)
constexpr size_t SK_L1_LDS = (size_t)SK_PPT * SK_VAL_PITCH * 4 + ((size_t)4 << SK_MAX_COARSE) + 128 + (size_t)SK_WAVES * (SK_WORD_PITCH * 8 + 64 * 8 + SK_WAVE_RUNS * 2);
static_assert(SK_L1_LDS <= 159 * 1024, "one 1024-thread workgroup per CU: its LDS");
constexpr int SK2_THREADS = 256;
constexpr int SK2R_THREADS = 512;                     // level 2, records only
constexpr int SK2_TILE_KEYS = SK2_THREADS * SK_LMAX;                // one record per thread: at most 256 x 22 keys = 44 KB of LDS per tile

struct SkArgs {
    const uint64_t *sym2;
    const uint64_t *inv;
    uint64_t total_syms;
    const uint64_t *genome_sym_off;
    uint32_t n_genomes;
    int k, bb;
    int b1;                                           // coarse bits of the bucket (level 1); the other bb - b1 are the fine ones
    int part_bits;                                    // a genome is cut into 2^part_bits parts, one workgroup each
};

// ---- level 1 -----------------------------------------------------------------------------------------------------
// One workgroup per (genome, part) = "virtual genome" vg: it owns the part's 2^b1 coarse regions of records1, so the
// place of a record needs no global atomic (a returning global atomic per record measured 20 ms for 6e8 records, and an
// add per wave to ONE global counter 28 ms: same-address atomics serialise in L2).  Per step a thread takes one window
// = one word of the packed stream = 32 k-mer start positions: minimizer of every start (template on W = k - SK_M + 1:
// the window minimum is a fixed pattern of register moves), run heads where the minimizer occurrence changes.  A run
// belongs to the window it STARTS in; the one that crosses the window's end goes on for as many positions as the next
// window's thread (the lane to the right) finds in front of its first head.  Each run leaves as one record, stored at the
// LDS cursor of its coarse region.
// WIDE (two-word k-mers, 33 <= k <= 64): W = runw_window(k) m-mers in the middle of the k-mer -- the same minimizers, heads and leads
// computed on the stream moved on by runw_offset(k) positions; validity over k <= 64 symbols; records of 24 bytes (recs1 counts in
// 8-byte words then: three per record).
template <int W, bool WIDE>
__global__ __launch_bounds__(SK_THREADS) void superkmer_l1_kernel(SkArgs a, ulonglong2 *__restrict__ recs1, uint32_t rstride,
                                                                   uint32_t *__restrict__ rcount1, uint32_t *__restrict__ part_kmers,
                                                                   int *__restrict__ overflow)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint32_t *s_val = reinterpret_cast<uint32_t *>(lds_raw);          // [SK_PPT][SK_VAL_PITCH]: minimizer word of every position of the step
    uint32_t *cursor = s_val + SK_PPT * SK_VAL_PITCH;                 // records written so far to coarse region c
    uint32_t *scratch = cursor + (1u << SK_MAX_COARSE);
    // per wave, for handing the step's runs round among the lanes: every lane's window word, (run boundaries, lead), and the list of runs
    uint64_t *w_word = reinterpret_cast<uint64_t *>(scratch + 32) + (threadIdx.x >> 6) * SK_WORD_PITCH;
    uint2 *w_edge = reinterpret_cast<uint2 *>(reinterpret_cast<uint64_t *>(scratch + 32) + SK_WAVES * SK_WORD_PITCH) + (threadIdx.x >> 6) * 64;
    uint16_t *w_runs = reinterpret_cast<uint16_t *>(reinterpret_cast<uint64_t *>(scratch + 32) + SK_WAVES * (SK_WORD_PITCH + 64)) + (threadIdx.x >> 6) * SK_WAVE_RUNS;
    const int b1 = a.b1;
    const uint32_t B1 = 1u << b1;
    const uint32_t vg = blockIdx.x;
    const uint32_t gen = vg >> a.part_bits, part = vg & ((1u << a.part_bits) - 1u);
    const uint64_t lo = a.genome_sym_off[gen], hi = a.genome_sym_off[gen + 1];
    // positions that may start a k-mer of this genome: [lo, p_end)
    const uint64_t last = a.total_syms >= (uint64_t)a.k ? a.total_syms - a.k + 1 : 0;
    const uint64_t p_end = hi < last ? hi : last;
    // the genome's windows (words of the stream), cut into parts
    const uint64_t jw_lo = lo >> 5, jw_hi = hi > lo ? (hi + 31) >> 5 : jw_lo;
    const uint64_t n_win = jw_hi - jw_lo;
    const uint64_t per_part = (n_win + (1u << a.part_bits) - 1) >> a.part_bits;
    const uint64_t j_a = jw_lo + min((uint64_t)part * per_part, n_win), j_b = min(j_a + per_part, jw_hi);
    for (uint32_t c = threadIdx.x; c < B1; c += SK_THREADS) cursor[c] = 0;
    __syncthreads();
    const int lane = lane_id(), wave = wave_id();
    const uint32_t wave_t0 = threadIdx.x & ~63u;
    ulonglong2 *my_recs = recs1 + (uint64_t)vg * B1 * rstride;           // the part's regions: 2^b1 x rstride records (< 2^32 of them: host)
    uint64_t *my_recs_w = reinterpret_cast<uint64_t *>(recs1) + (uint64_t)vg * B1 * rstride * 3;      // (WIDE: 24-byte records)
    const int woff = WIDE ? runw_offset(a.k) : 0;
    const bool narrow = rstride < (1u << 15);                            // (region index x stride as a full-rate 24-bit multiply)
    uint32_t n_valid = 0;
    bool over = false;
    // Related genomes are the same sequence at about the same window numbers: started together, their workgroups would store
    // to the same places of their (equally laid out) regions at the same time.  Each workgroup starts its round through the
    // part's steps somewhere else.
    const uint32_t n_steps = (uint32_t)((j_b - j_a + SK_STEP_WINDOWS - 1) / SK_STEP_WINDOWS);
    const uint32_t rot = n_steps ? (uint32_t)((vg * 0x9E3779B1u) >> 8) % n_steps : 0u;
    // The words of a step are asked for one step ahead (a wave that waited for them at the top of every step left its SIMD to three others,
    // which wait for theirs: 55 % of the wave cycles were waits): the raw words of the stream and of the flags, used a step later.
    struct StepWords {
        uint64_t w0, w1, w2, w3, pw, i0, i1, i2;
    };
    auto step_window = [&](uint32_t it) -> uint64_t {
        const uint32_t st = it + rot < n_steps ? it + rot : it + rot - n_steps;
        return j_a + (uint64_t)st * SK_STEP_WINDOWS + (uint32_t)(wave * SK_WAVE_WINDOWS + lane);
    };
    auto fetch = [&](uint32_t it, StepWords &f) {
        // EVERY lane asks, at every call, for the same number of words: a window past the part's last ones (or a step past the last step)
        // asks for the last window's again, and what is valid is decided where the words are used.  A request under a condition makes
        // the compiler lose count of the loads in flight at the loop's back edge, and it then waits for ALL of them -- the two steps'
        // worth just asked for included -- at the first use: until round 4's second half the "two steps ahead" above was none.
        // (j_b: only for the run that crosses into it; j_b + 1: only its hashes; WIDE: j_b + 2 for its word -- a run of 85 bases that starts late
        // in the part's last window reaches the fourth word; the stream buffers end with slack words)
        const uint64_t j = min(step_window(it), j_b + (WIDE ? 2 : 1));
        f.w3 = f.i2 = 0;
        // (streamed once: loads marked non-temporal, so that the packed stream does not push the workgroups' half-filled record
        // lines out of L2)
        f.w0 = __builtin_nontemporal_load(&a.sym2[j]);
        f.w1 = __builtin_nontemporal_load(&a.sym2[j + 1]);
        f.w2 = __builtin_nontemporal_load(&a.sym2[j + 2]);
        if (WIDE) f.w3 = __builtin_nontemporal_load(&a.sym2[j + 3]);
        f.pw = __builtin_nontemporal_load(&a.sym2[j ? j - 1 : 0]);             // (j == 0: not used)
        const uint64_t q = j ? (j << 5) - 1 : 0;
        f.i0 = __builtin_nontemporal_load(&a.inv[q >> 6]);
        f.i1 = __builtin_nontemporal_load(&a.inv[(q >> 6) + 1]);
        if (WIDE) f.i2 = __builtin_nontemporal_load(&a.inv[(q >> 6) + 2]);
    };
    StepWords nxt, nxt2;
    fetch(0, nxt);
    fetch(1, nxt2);
    for (uint32_t it = 0; it < n_steps; it++) {
        const uint32_t st = it + rot < n_steps ? it + rot : it + rot - n_steps;
        const uint64_t j = step_window(it);
        const StepWords cur = nxt;
        nxt = nxt2;
        fetch(it + 2, nxt2);
        if (j_a + (uint64_t)st * SK_STEP_WINDOWS + (uint32_t)(wave * SK_WAVE_WINDOWS) >= j_b) continue;        // (the wave as a whole: nothing left)
        uint32_t valid = 0, heads = 0;
        uint64_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, vs = 0;
        uint32_t prev2 = 0;
        if (j <= j_b + (WIDE ? 2 : 1)) {
            const uint64_t p0 = j << 5;
            w0 = cur.w0; w1 = cur.w1; w2 = cur.w2; w3 = cur.w3;
            prev2 = j ? (uint32_t)cur.pw & 3u : 0u;
            // bit t of vs: position p0 - 1 + t starts a k-mer (t = 0 .. 32; k <= 32 keeps all 33 inside the 64 flags read)
            const int qo = p0 ? (int)((p0 - 1) & 63) : 0;
            if (WIDE) vs = valid_starts_wide(cur.i0, cur.i1, cur.i2, qo, a.k);
            else vs = valid_starts_at(cur.i0, cur.i1, qo, a.k);
            if (!p0) vs <<= 1;
            // ... of THIS genome: lo <= p0 - 1 + t < p_end
            const uint64_t t_lo = lo + 1 > p0 ? lo + 1 - p0 : 0, t_hi = p_end + 1 > p0 ? p_end + 1 - p0 : 0;
            const uint64_t keep = (t_hi >= 33 ? (1ull << 33) - 1 : (1ull << t_hi) - 1) & ~(t_lo >= 33 ? (1ull << 33) - 1 : (1ull << t_lo) - 1);
            vs &= keep;
            valid = (uint32_t)(vs >> 1);
        }
        {
            // every lane, whatever it holds (no lane may sit out a DPP move): the words of the m-mers at positions -1 .. 31 of the window,
            // and of those at 32 .. 31 + W - 1 from the lane to the right
            uint32_t h[SK_PPT + W];
            {
                uint32_t own[SK_PPT + 1];
                if (WIDE && woff) {
                    // the m-mers the k-mers of this window look at start woff positions further on
                    run_hashes<SK_PPT + 1>((w0 << (2 * woff)) | (w1 >> (64 - 2 * woff)), (w1 << (2 * woff)) | (w2 >> (64 - 2 * woff)),
                                           (uint32_t)(w0 >> (64 - 2 * woff)) & 3u, own);
                } else {
                    run_hashes<SK_PPT + 1>(w0, w1, prev2, own);
                }
#pragma unroll
                for (int i = 0; i <= SK_PPT; i++) h[i] = own[i];
#pragma unroll
                for (int t = 0; t + 1 < W; t++)
                    h[SK_PPT + 1 + t] = run_hash_from_right((uint32_t)__builtin_amdgcn_update_dpp(0, (int)own[1 + t], 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
            }
            uint32_t val[SK_PPT + 1];
            run_window_min<W>(h, val);
            heads = run_heads(valid, (vs & 1u) != 0, val);
            // the minimizer word of a run's first position is the only per-position value the emission needs, and it is
            // indexed by a run-time position: through LDS (each thread reads back its own column)
#pragma unroll
            for (int i = 0; i < SK_PPT; i++) s_val[i * SK_VAL_PITCH + threadIdx.x] = val[i + 1];
        }
        const bool mine = lane < SK_WAVE_WINDOWS && j < j_b;
        if (mine) n_valid += (uint32_t)__popc(valid);
        // One run = one record: its minimizer word back from LDS, its bucket, its bases out of the window's words, its place from
        // the LDS cursor of its coarse region.  A lane's window starts ~3.9 runs, the fullest lane of a wave ~8: taken lane by lane
        // the wave would go round 8 times with half its lanes idle, so the wave's runs are listed (lane, position) and handed round,
        // 64 at a time, with what a run needs of its window (the words, the run boundaries, the neighbour's lead) read from LDS.
        const uint32_t lead = run_lead(valid, heads);
        const uint32_t n_mine = mine ? (uint32_t)__popc(heads) : 0u;
        const uint32_t upto = wave_scan_incl_dpp(n_mine);
        const uint32_t n_wave = (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
        auto emit = [&](uint64_t e0, uint64_t e1, uint64_t e2, uint64_t e3, uint32_t bnd, uint32_t lead_next, uint32_t v, int i) {
            uint32_t len = (uint32_t)__builtin_ctz((bnd >> 1 >> i) | (0x80000000u >> i)) + 1u;       // run_length(): to the next boundary or the window's end
            if (i + (int)len == SK_PPT) len += lead_next;
            const uint32_t bkt = minimizer_bucket(v >> (32 - MINIMIZER_ORDER_BITS), b1 + SK_FINE_BITS);
            const uint32_t c = bkt >> SK_FINE_BITS;
            const uint32_t slot = atomicAdd(&cursor[c], 1u);
            if (WIDE) {
                const RunW rw = runw_record(e0, e1, e2, e3, i, len, a.k, (v & 1u) != 0, bkt);
                if (slot < rstride) {
                    uint64_t *o = my_recs_w + 3ull * ((narrow ? mul24(c, rstride) : c * rstride) + slot);
                    o[0] = rw.r[0]; o[1] = rw.r[1]; o[2] = rw.r[2];
                } else over = true;
            } else {
                uint64_t rx, ry;
                run_record(e0, e1, e2, i, len, a.k, (v & 1u) != 0, bkt, rx, ry);
                if (slot < rstride) my_recs[(narrow ? mul24(c, rstride) : c * rstride) + slot] = make_ulonglong2(rx, ry);
                else over = true;
            }
        };
        if (n_wave <= (uint32_t)SK_WAVE_RUNS) {
            w_word[lane] = w0;
            if (WIDE && lane == 63) { w_word[64] = w1; w_word[65] = w2; }
            w_edge[lane] = make_uint2(heads | ~valid, lead);
            uint32_t hd = mine ? heads : 0u, at = upto - n_mine;
            while (hd) {
                w_runs[at++] = (uint16_t)(((uint32_t)lane << 5) | (uint32_t)(__ffs(hd) - 1));
                hd &= hd - 1;
            }
            for (uint32_t r0 = 0; r0 < n_wave; r0 += 64) {
                if (r0 + (uint32_t)lane < n_wave) {
                    const uint32_t e = w_runs[r0 + lane], owner = e >> 5;
                    const int i = (int)(e & 31u);
                    emit(w_word[owner], w_word[owner + 1], w_word[owner + 2], WIDE ? w_word[owner + 3] : 0ull, w_edge[owner].x, w_edge[owner + 1].y,
                         s_val[i * SK_VAL_PITCH + wave_t0 + owner], i);
                }
            }
        } else {
            const uint32_t lead_next = __shfl_down(lead, 1);
            uint32_t hd = mine ? heads : 0u;
            while (hd) {
                const int i = __ffs(hd) - 1;
                hd &= hd - 1;
                emit(w0, w1, w2, w3, heads | ~valid, lead_next, s_val[i * SK_VAL_PITCH + threadIdx.x], i);
            }
        }
    }
    __syncthreads();
    if (over) atomicExch(overflow, 1);
    for (uint32_t c = threadIdx.x; c < B1; c += SK_THREADS) rcount1[(uint64_t)vg * B1 + c] = min(cursor[c], rstride);
    uint32_t total;
    (void)block_scan_sum(n_valid, scratch, &total);
    if (threadIdx.x == 0) part_kmers[vg] = total;
}

// fine bucket of a record when 2^b2 fine buckets are in use: the top b2 of its SK_FINE_BITS
__device__ __forceinline__ uint32_t rec_fine(uint64_t y, int b2) { return run_fine(y) >> (SK_FINE_BITS - b2); }

// ---- level 2 -----------------------------------------------------------------------------------------------------
// One workgroup per (virtual genome, coarse bucket) region at a time.  Pass A adds up the k-mers per fine bucket (the
// region is a few ten KB: its second reading comes from L2) and lays the region's fine key segments out back to back
// at region * kstride: off / len of segment vg * 2^bb + bucket.  Pass B takes tiles of 256 records (the next tile's
// records are requested before the current one is worked on).  A record's k-mers go to ONE fine bucket, so the rank of
// the record inside its bucket (one returning LDS atomic per record) places all of them.  A lane rolls through its record
// (forward and reverse-complement words, as the key form's extraction does) and stores the canonical k-mers into the
// tile's LDS image, which leaves as one contiguous run per fine bucket.  Records hold 1..16 k-mers: the tile's records are
// first sorted by length (a counting sort through LDS), so that the lanes of a wave roll for about the same number of
// steps -- unsorted, a third of the lanes idle.  (The rank of a record inside its fine bucket travels beside it in LDS: a record
// has no spare bits, and none that differ between two genomes with the same run.)
__global__ __launch_bounds__(SK2_THREADS) void superkmer_l2_kernel(const ulonglong2 *__restrict__ recs1, uint32_t rstride,
                                                                    const uint32_t *__restrict__ rcount1, uint64_t n_regions, int k, int bb, int b1,
                                                                    uint64_t kstride, uint64_t *__restrict__ keys, uint64_t *__restrict__ off,
                                                                    uint32_t *__restrict__ len_out, int *__restrict__ overflow)
{
    __shared__ uint64_t skeys[SK2_TILE_KEYS];
    __shared__ ulonglong2 srec[SK2_THREADS];
    __shared__ uint32_t srank[SK2_THREADS];
    constexpr int NF = 1 << SK_FINE_BITS;          // fine buckets of a region, at most
    __shared__ uint32_t gbase[NF];                      // relative to the region's first key
    __shared__ uint32_t hist[NF], start[NF], lhist[64], lstart[64];
    __shared__ uint32_t scratch[32];
    const int b2 = bb - b1;
    const uint32_t B2 = 1u << b2;
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const int rcshift = 2 * (k - 1);
    const int lane = lane_id(), wave = wave_id();
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        const uint32_t n = min(rcount1[region], rstride);
        const ulonglong2 *rr = recs1 + region * rstride;
        const uint64_t seg0 = region << b2;            // segment index of the region's fine bucket 0: (vg * B1 + c) * B2
        ulonglong2 rec_next = threadIdx.x < n ? rr[threadIdx.x] : make_ulonglong2(0, 0);
        // ---- pass A: k-mers per fine bucket -> segment offsets ----
        if (threadIdx.x < NF) hist[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += SK2_THREADS) {
            const uint64_t y = rr[i].y;
            atomicAdd(&hist[rec_fine(y, b2)], run_len(y));
        }
        __syncthreads();
        const uint32_t cnt = threadIdx.x < B2 ? hist[threadIdx.x] : 0u;
        uint32_t region_keys;
        const uint32_t pre = block_scan_sum(cnt, scratch, &region_keys);
        const bool region_fits = region_keys <= kstride;           // uniform
        if (!region_fits && threadIdx.x == 0) atomicExch(overflow, 1);
        const uint64_t key0 = region * kstride;
        uint32_t my_next = pre;                                     // thread f: running output position of fine bucket f
        if (threadIdx.x < B2) {
            off[seg0 + threadIdx.x] = key0 + my_next;
            len_out[seg0 + threadIdx.x] = region_fits ? cnt : 0u;
        }
        if (!region_fits) continue;
        // ---- pass B: tiles of SK2_THREADS records ----
        for (uint32_t t0 = 0; t0 < n; t0 += SK2_THREADS) {
            ulonglong2 rec = rec_next;
            {
                const uint32_t i = t0 + SK2_THREADS + threadIdx.x;
                rec_next = i < n ? rr[i] : make_ulonglong2(0, 0);
            }
            if (threadIdx.x < NF) hist[threadIdx.x] = 0;
            if (threadIdx.x < 64) lhist[threadIdx.x] = 0;
            __syncthreads();
            const uint32_t ln = run_len(rec.y);                          // 0 for the padding of the last tile
            const uint32_t fine = rec_fine(rec.y, b2);
            const uint32_t frank = ln ? atomicAdd(&hist[fine], ln) : 0u;  // first k-mer of the record inside its fine bucket
            const uint32_t lrank = atomicAdd(&lhist[ln], 1u);
            __syncthreads();
            if (wave == 0) {
                const uint32_t c = lhist[lane];
                lstart[lane] = wave_scan_incl_dpp(c) - c;
            }
            const uint32_t c2 = threadIdx.x < B2 ? hist[threadIdx.x] : 0u;
            uint32_t n_tile;
            const uint32_t st = block_scan_sum(c2, scratch, &n_tile);      // its barriers also publish lstart
            if (threadIdx.x < B2) {
                start[threadIdx.x] = st;
                gbase[threadIdx.x] = my_next;
                my_next += c2;
            }
            // by length: thread j continues with the j-th shortest record
            srec[lstart[ln] + lrank] = rec;
            srank[lstart[ln] + lrank] = frank;
            __syncthreads();
            {
                const ulonglong2 r2 = srec[threadIdx.x];
                const uint32_t l2 = run_len(r2.y);
                if (l2) {
                    uint32_t at = start[rec_fine(r2.y, b2)] + srank[threadIdx.x];
                    RunDecoder dec = run_open(r2.x, r2.y, k);
                    for (uint32_t t = 0;; t++) {
                        skeys[at++] = run_canonical(dec);
                        if (t + 1 >= l2) break;
                        run_next(dec, mask, rcshift);
                    }
                }
            }
            __syncthreads();
            // one contiguous run per fine bucket: the waves take the buckets in turn
            for (uint32_t f = (uint32_t)wave; f < B2; f += SK2_THREADS / 64) {
                const uint32_t nf = hist[f], src = start[f];
                const uint64_t dst = key0 + gbase[f];
                for (uint32_t i = (uint32_t)lane; i < nf; i += 64) keys[dst + i] = skeys[src + i];
            }
            __syncthreads();
        }
    }
}

// Level 2 without the expansion (dict_build decodes the records itself, record form): the region's records sorted by fine
// bucket, in place of the region in a second buffer.  STAGED: the region is read once into LDS (a returning LDS atomic
// gives every record its rank inside its fine bucket, kept in a 16-bit array beside the records) and written
// from there; regions too large for that are swept twice (count, then place; the second sweep comes from L2).
// Inside a fine bucket the records of at most 4 k-mers stand first: without its memo dict_build takes those four keys at a
// time and the others eight at a time.
// Segment vg * 2^bb + bucket = recs2[off[..] .. + (len[..] & 0xffff)) in RECORDS, the first len[..] >> 16 of them short.
// bin of a record: 2 * fine bucket + (more than 4 k-mers)
// a record level 1 marked for the other strand is turned over on its way through level 2
__device__ __forceinline__ void rec_flip(ulonglong2 &r, int k)
{
    uint64_t x = r.x, y = r.y;
    run_flip(x, y, k);
    r.x = x;
    r.y = y;
}
__device__ __forceinline__ uint32_t rec_bin(uint64_t y, int b2) { return (rec_fine(y, b2) << 1) | (uint32_t)(run_len(y) > 4u); }

template <bool STAGED, int MAXR>
__global__ __launch_bounds__(SK2R_THREADS) void superkmer_l2_records_kernel(const ulonglong2 *__restrict__ recs1, uint32_t rstride,
                                                                            const uint32_t *__restrict__ rcount1, uint64_t n_regions, int k, int bb, int b1,
                                                                            ulonglong2 *__restrict__ recs2, uint64_t *__restrict__ off,
                                                                            uint32_t *__restrict__ len_out, int *__restrict__ overflow)
{
    constexpr int NF = 1 << SK_FINE_BITS, NB = 2 * NF;  // bins: (fine bucket, length class)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    ulonglong2 *srec = reinterpret_cast<ulonglong2 *>(lds_raw);          // [rstride] when STAGED
    uint16_t *srank = reinterpret_cast<uint16_t *>(lds_raw + (size_t)rstride * 16);      // [rstride] (rstride < 2^16 when STAGED)
    __shared__ uint32_t hist[NB], start[NB];
    __shared__ uint32_t scratch[32];
    const int b2 = bb - b1;
    const uint32_t B2 = 1u << b2;
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        const uint32_t n = min(rcount1[region], rstride);
        const ulonglong2 *rr = recs1 + region * rstride;
        ulonglong2 *out = recs2 + region * rstride;
        const uint64_t seg0 = region << b2;
        if (threadIdx.x < NB) hist[threadIdx.x] = 0;
        __syncthreads();
        if (STAGED) {
            // all of a thread's loads are in flight before the first one is used
            ulonglong2 in[MAXR > 0 ? MAXR : 1];
#pragma unroll
            for (int j = 0; j < MAXR; j++) {
                const uint32_t i = (uint32_t)j * SK2R_THREADS + threadIdx.x;
                in[j] = i < n ? rr[i] : make_ulonglong2(0, 0);
            }
#pragma unroll
            for (int j = 0; j < MAXR; j++) {
                const uint32_t i = (uint32_t)j * SK2R_THREADS + threadIdx.x;
                if (i < n) {
                    rec_flip(in[j], k);
                    srank[i] = (uint16_t)atomicAdd(&hist[rec_bin(in[j].y, b2)], 1u);
                    srec[i] = in[j];
                }
            }
        } else {
            // (four records per thread asked for before the first is used, from clamped indices: one load per trip of the loop is one
            // round trip per trip)
            for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 4u * SK2R_THREADS) {
                uint64_t y[4];
#pragma unroll
                for (int u = 0; u < 4; u++) y[u] = rr[min(i0 + (uint32_t)u * SK2R_THREADS, n - 1u)].y;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + (uint32_t)u * SK2R_THREADS < n) atomicAdd(&hist[rec_bin(y[u], b2)], 1u);
            }
        }
        __syncthreads();
        const uint32_t cnt = threadIdx.x < 2 * B2 ? hist[threadIdx.x] : 0u;
        uint32_t total;
        const uint32_t pre = block_scan_sum(cnt, scratch, &total);
        const uint32_t cnt_long = __shfl_down(cnt, 1);         // (of the bin after this one)
        if (threadIdx.x < 2 * B2) {
            start[threadIdx.x] = pre;
            hist[threadIdx.x] = 0;                     // (two-sweep form: now the running rank)
            if (!(threadIdx.x & 1u)) {                 // bin 2f: the short records of fine bucket f, followed by its long ones
                const uint32_t f = threadIdx.x >> 1;
                off[seg0 + f] = region * rstride + pre;
                len_out[seg0 + f] = (cnt + cnt_long) | (cnt << 16);
                // (a segment's record count travels in 16 bits beside the count of its short ones: the host falls back to the key form)
                if (cnt + cnt_long > 0xffffu) atomicExch(overflow, 1);
            }
        }
        __syncthreads();
        if (STAGED) {
            // in-place permutation of the LDS image through registers (a thread holds its <= MAXR records), then the
            // sorted region leaves with full-line stores: scattered 16-byte stores ran into the L2 request rate
            ulonglong2 mine[MAXR > 0 ? MAXR : 1];
            uint32_t rank[MAXR > 0 ? MAXR : 1];
#pragma unroll
            for (int j = 0; j < MAXR; j++) {
                const uint32_t i = (uint32_t)j * SK2R_THREADS + threadIdx.x;
                mine[j] = i < n ? srec[i] : make_ulonglong2(0, 0);
                rank[j] = i < n ? srank[i] : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MAXR; j++) {
                const uint32_t i = (uint32_t)j * SK2R_THREADS + threadIdx.x;
                if (i < n) srec[start[rec_bin(mine[j].y, b2)] + rank[j]] = mine[j];
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < n; i += SK2R_THREADS) out[i] = srec[i];
        } else {
            for (uint32_t i0 = threadIdx.x; i0 < n; i0 += 4u * SK2R_THREADS) {
                ulonglong2 rec[4];
#pragma unroll
                for (int u = 0; u < 4; u++) rec[u] = rr[min(i0 + (uint32_t)u * SK2R_THREADS, n - 1u)];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (i0 + (uint32_t)u * SK2R_THREADS >= n) continue;
                    rec_flip(rec[u], k);
                    const uint32_t f = rec_bin(rec[u].y, b2);
                    out[start[f] + atomicAdd(&hist[f], 1u)] = rec[u];
                }
            }
        }
        __syncthreads();
    }
}

// Level 2 for the 24-byte records of two-word k-mers: a region's records sorted by fine bucket (and turned over where level 1
// marked them), as superkmer_l2_records_kernel does for 16-byte ones.  A thread keeps its <= SKW_MAXR records in registers
// between the counting and the placing; the sorted region is put together in LDS and leaves with full-line stores.  Regions
// of more records than that are swept twice (count, then place straight into global memory).
// Segment vg * 2^bb + bucket = recs2[off[..] .. + len[..]) in RECORDS.
constexpr int SKW_THREADS = 512, SKW_MAXR = 8;
__global__ __launch_bounds__(SKW_THREADS) void superkmer_l2_wide_kernel(const uint64_t *__restrict__ recs1, uint32_t rstride,
                                                                        const uint32_t *__restrict__ rcount1, uint64_t n_regions, int k, int bb, int b1,
                                                                        uint64_t *__restrict__ recs2, uint64_t *__restrict__ off,
                                                                        uint32_t *__restrict__ len_out, int *__restrict__ overflow)
{
    constexpr int NF = 1 << SK_FINE_BITS;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint64_t *sout = reinterpret_cast<uint64_t *>(lds_raw);              // [3 * min(rstride, SKW_THREADS * SKW_MAXR)]
    __shared__ uint32_t hist[NF], start[NF];
    __shared__ uint32_t scratch[32];
    const int b2 = bb - b1;
    const uint32_t B2 = 1u << b2;
    for (uint64_t region = blockIdx.x; region < n_regions; region += gridDim.x) {
        const uint32_t n = min(rcount1[region], rstride);
        const uint64_t *rr = recs1 + region * rstride * 3;
        uint64_t *out = recs2 + region * rstride * 3;
        const uint64_t seg0 = region << b2;
        const bool staged = n <= (uint32_t)(SKW_THREADS * SKW_MAXR);       // uniform
        if (threadIdx.x < NF) hist[threadIdx.x] = 0;
        __syncthreads();
        RunW in[SKW_MAXR];
        uint32_t rank[SKW_MAXR];
        if (staged) {
#pragma unroll
            for (int j = 0; j < SKW_MAXR; j++) {
                const uint32_t i = (uint32_t)j * SKW_THREADS + threadIdx.x;
                if (i < n) { in[j].r[0] = rr[3 * i]; in[j].r[1] = rr[3 * i + 1]; in[j].r[2] = rr[3 * i + 2]; }
                else { in[j].r[0] = in[j].r[1] = in[j].r[2] = 0; }
            }
#pragma unroll
            for (int j = 0; j < SKW_MAXR; j++) {
                const uint32_t i = (uint32_t)j * SKW_THREADS + threadIdx.x;
                rank[j] = 0;
                if (i < n) {
                    runw_flip(in[j], k);
                    rank[j] = atomicAdd(&hist[rec_fine(in[j].r[2], b2)], 1u);
                }
            }
        } else {
            for (uint32_t i = threadIdx.x; i < n; i += SKW_THREADS) atomicAdd(&hist[rec_fine(rr[3 * i + 2], b2)], 1u);
        }
        __syncthreads();
        const uint32_t cnt = threadIdx.x < B2 ? hist[threadIdx.x] : 0u;
        uint32_t total;
        const uint32_t pre = block_scan_sum(cnt, scratch, &total);
        if (threadIdx.x < B2) {
            start[threadIdx.x] = pre;
            hist[threadIdx.x] = 0;
            off[seg0 + threadIdx.x] = region * rstride + pre;
            len_out[seg0 + threadIdx.x] = cnt;
            if (cnt > 0xffffu) atomicExch(overflow, 1);
        }
        __syncthreads();
        if (staged) {
#pragma unroll
            for (int j = 0; j < SKW_MAXR; j++) {
                const uint32_t i = (uint32_t)j * SKW_THREADS + threadIdx.x;
                if (i < n) {
                    const uint32_t at = 3u * (start[rec_fine(in[j].r[2], b2)] + rank[j]);
                    sout[at] = in[j].r[0]; sout[at + 1] = in[j].r[1]; sout[at + 2] = in[j].r[2];
                }
            }
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < 3u * n; i += SKW_THREADS) out[i] = sout[i];
        } else {
            for (uint32_t i = threadIdx.x; i < n; i += SKW_THREADS) {
                RunW r;
                r.r[0] = rr[3 * i]; r.r[1] = rr[3 * i + 1]; r.r[2] = rr[3 * i + 2];
                runw_flip(r, k);
                const uint32_t f = rec_fine(r.r[2], b2);
                const uint64_t at = 3ull * (start[f] + atomicAdd(&hist[f], 1u));
                out[at] = r.r[0]; out[at + 1] = r.r[1]; out[at + 2] = r.r[2];
            }
        }
        __syncthreads();
    }
}

// bucket ((bucket << sb) | sub) of dictionary keys under minimizer buckets (the probing form of the fill builds its
// per-bucket tables from the dictionary): the minimizer is re-derived from the k-mer itself
__global__ void minimizer_bucket_ids_kernel(const uint64_t *__restrict__ dict, uint64_t n, int k, int bb, int sb,
                                            uint32_t *__restrict__ bucket_of, uint32_t *__restrict__ col_of)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = dict[i];
        bucket_of[i] = (minimizer_bucket_of_kmer(key, k, bb, SK_M) << sb) | hash_sub(mix64(key), bb, sb);
        col_of[i] = (uint32_t)i;
    }
}

template <int W, bool WIDE = false>
static void launch_sk1(hipStream_t s, const SkArgs &a, ulonglong2 *recs1, uint32_t rstride, uint32_t *rcount1, uint32_t *part_kmers, int *overflow)
{
    static std::atomic<uint64_t> lds_set{0};
    (void)ensure_dynamic_lds(reinterpret_cast<const void *>(superkmer_l1_kernel<W, WIDE>), (int)SK_L1_LDS, lds_set);
    hipLaunchKernelGGL((superkmer_l1_kernel<W, WIDE>), dim3(a.n_genomes << a.part_bits), dim3(SK_THREADS), SK_L1_LDS, s, a, recs1, rstride, rcount1,
                       part_kmers, overflow);
}

int superkmer_max_bits() { return SK_MAX_BITS; }
// 256 regions per genome part: with 512 a workgroup keeps twice the half-filled 128-byte lines open in L2 and writes 1.8x its records to
// HBM (13.8 GB for 8.0 at 1000 x 5 Mbp; 8.7 with 256), and level 2 still sorts a region's ~2000 records (5 Mbp) inside LDS; with 128 it
// no longer does (5.6 ms instead of 2.8)
int superkmer_coarse_bits(int bb) { return bb < 8 ? bb : 8; }
int superkmer_lmax() { return SK_LMAX; }
int superkmer_wide_window(int k) { return runw_window(k); }

void launch_superkmer_l1(hipStream_t s, const KmerLaunch &L, int b1, int part_bits, void *recs1, uint32_t rstride, uint32_t *rcount1,
                         uint32_t *part_kmers, int *overflow)
{
    if (!L.total_syms || !L.n_genomes) return;
    SkArgs a;
    a.sym2 = L.sym2; a.inv = L.inv; a.total_syms = L.total_syms; a.genome_sym_off = L.genome_sym_off;
    a.n_genomes = L.n_genomes; a.k = L.k; a.bb = L.bb;
    a.b1 = b1;
    a.part_bits = part_bits;
    ulonglong2 *r = reinterpret_cast<ulonglong2 *>(recs1);
    if (L.k > 32) {         // two-word k-mers: 24-byte records, the window in the middle of the k-mer
        if (runw_window(L.k) == 21) launch_sk1<21, true>(s, a, r, rstride, rcount1, part_kmers, overflow);
        else launch_sk1<22, true>(s, a, r, rstride, rcount1, part_kmers, overflow);
        return;
    }
    switch (L.k - SK_M + 1) {
#define GRM_SK_CASE(W) case W: launch_sk1<W>(s, a, r, rstride, rcount1, part_kmers, overflow); break;
        GRM_SK_CASE(1) GRM_SK_CASE(2) GRM_SK_CASE(3) GRM_SK_CASE(4) GRM_SK_CASE(5) GRM_SK_CASE(6) GRM_SK_CASE(7) GRM_SK_CASE(8)
        GRM_SK_CASE(9) GRM_SK_CASE(10) GRM_SK_CASE(11) GRM_SK_CASE(12) GRM_SK_CASE(13) GRM_SK_CASE(14) GRM_SK_CASE(15)
        GRM_SK_CASE(16) GRM_SK_CASE(17) GRM_SK_CASE(18) GRM_SK_CASE(19) GRM_SK_CASE(20) GRM_SK_CASE(21) GRM_SK_CASE(22)
#undef GRM_SK_CASE
        default: break;     // the host only asks for 11 <= k <= 32 here
    }
}

void launch_superkmer_l2(hipStream_t s, const void *recs1, uint32_t rstride, const uint32_t *rcount1, uint64_t n_regions, int k, int bb, int b1,
                         uint64_t kstride, uint64_t *keys, uint64_t *off, uint32_t *len, int *overflow)
{
    if (!n_regions) return;
    const uint32_t grid = (uint32_t)(n_regions < 256u * 32u ? n_regions : 256u * 32u);
    hipLaunchKernelGGL(superkmer_l2_kernel, dim3(grid), dim3(SK2_THREADS), 0, s, reinterpret_cast<const ulonglong2 *>(recs1), rstride, rcount1,
                       n_regions, k, bb, b1, kstride, keys, off, len, overflow);
}

template <int MAXR>
static void launch_l2r_staged(hipStream_t s, uint32_t grid, const ulonglong2 *r1, uint32_t rstride, const uint32_t *rcount1, uint64_t n_regions, int k,
                              int bb, int b1, ulonglong2 *r2, uint64_t *off, uint32_t *len, int *overflow)
{
    static std::atomic<uint64_t> lds_set{0};
    (void)ensure_dynamic_lds(reinterpret_cast<const void *>(superkmer_l2_records_kernel<true, MAXR>), 96 * 1024, lds_set);
    // the records and, beside them, their 16-bit ranks
    hipLaunchKernelGGL((superkmer_l2_records_kernel<true, MAXR>), dim3(grid), dim3(SK2R_THREADS), (size_t)rstride * 18, s, r1, rstride, rcount1,
                       n_regions, k, bb, b1, r2, off, len, overflow);
}

void launch_superkmer_l2_records(hipStream_t s, const void *recs1, uint32_t rstride, const uint32_t *rcount1, uint64_t n_regions, int k, int bb, int b1,
                                 void *recs2, uint64_t *off, uint32_t *len, int *overflow)
{
    if (!n_regions) return;
    const uint32_t grid = (uint32_t)(n_regions < 256u * 32u ? n_regions : 256u * 32u);
    const ulonglong2 *r1 = reinterpret_cast<const ulonglong2 *>(recs1);
    ulonglong2 *r2 = reinterpret_cast<ulonglong2 *>(recs2);
    // the region in LDS: up to 1536 records (27 KB, five workgroups per CU), 3072 (54 KB, two), 5120 (90 KB, one), else two sweeps
    if (rstride <= 3u * SK2R_THREADS) launch_l2r_staged<3>(s, grid, r1, rstride, rcount1, n_regions, k, bb, b1, r2, off, len, overflow);
    else if (rstride <= 6u * SK2R_THREADS) launch_l2r_staged<6>(s, grid, r1, rstride, rcount1, n_regions, k, bb, b1, r2, off, len, overflow);
    else if (rstride <= 10u * SK2R_THREADS) launch_l2r_staged<10>(s, grid, r1, rstride, rcount1, n_regions, k, bb, b1, r2, off, len, overflow);
    else hipLaunchKernelGGL((superkmer_l2_records_kernel<false, 0>), dim3(grid), dim3(SK2R_THREADS), 0, s, r1, rstride, rcount1, n_regions, k, bb, b1, r2, off, len, overflow);
}

void launch_minimizer_bucket_ids(hipStream_t s, const uint64_t *dict, uint64_t n, int k, int bb, int sb, uint32_t *bucket_of, uint32_t *col_of)
{
    if (!n) return;
    const uint64_t g = (n + 255) / 256;
    hipLaunchKernelGGL(minimizer_bucket_ids_kernel, dim3((uint32_t)(g > 8192 ? 8192 : g)), dim3(256), 0, s, dict, n, k, bb, sb, bucket_of, col_of);
}

void launch_superkmer_l2_wide(hipStream_t s, const void *recs1, uint32_t rstride, const uint32_t *rcount1, uint64_t n_regions, int k, int bb, int b1,
                              void *recs2, uint64_t *off, uint32_t *len_out, int *overflow)
{
    if (!n_regions) return;
    const uint32_t cap = rstride < (uint32_t)(SKW_THREADS * SKW_MAXR) ? rstride : (uint32_t)(SKW_THREADS * SKW_MAXR);
    const size_t lds = (size_t)cap * 24;
    static std::atomic<uint64_t> lds_set{0};
    (void)ensure_dynamic_lds(reinterpret_cast<const void *>(superkmer_l2_wide_kernel), SKW_THREADS * SKW_MAXR * 24, lds_set);
    const uint32_t grid = (uint32_t)(n_regions < 256u * 16u ? n_regions : 256u * 16u);
    hipLaunchKernelGGL(superkmer_l2_wide_kernel, dim3(grid), dim3(SKW_THREADS), lds, s, reinterpret_cast<const uint64_t *>(recs1), rstride, rcount1,
                       n_regions, k, bb, b1, reinterpret_cast<uint64_t *>(recs2), off, len_out, overflow);
}

}  // namespace grm
