// grm_internal.h -- shared between grm_kernels.hip (device side) and grm_api.cpp (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace grm {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds per DEVICE: a kernel that asks for more than 64 KiB of dynamic LDS registers
// itself once per device it is launched on (`done`: one bit per device ordinal, a function-local static of the launcher).  A failure
// is returned; a launcher that cannot return it launches anyway and the launch itself fails loudly (hipGetLastError at the call site).
inline hipError_t ensure_dynamic_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// ---- parse geometry ----
constexpr int PARSE_THREADS = 256;
constexpr int ROUND_BYTES = PARSE_THREADS * 16;            // one 16-B load per thread
constexpr int ROUNDS_PER_TILE = 4;
constexpr int TILE_BYTES = ROUND_BYTES * ROUNDS_PER_TILE;  // 16 KiB; every file starts on a tile boundary
constexpr int STAGE_BYTES = ROUND_BYTES + 64;              // <=63 lead symbols + one round
constexpr int RAW_FRONT_PAD = 64;                          // '\n' bytes in front of raw[0]

// ---- k-mer kernels ----
constexpr int KMER_THREADS = 256;
constexpr int MAX_BUCKET_BITS = 16;                        // 8 coarse + up to 8 fine bits (two-level partition)
constexpr int MAX_HIST_BITS = 13;                          // largest LDS histogram of the k-mer pass; deeper: region_hist
constexpr int L1_PPT = 16;                                 // start positions per thread in level 1
constexpr int L1_THREADS = 512;                            // 8 waves per tile: 2 tiles (16 waves) per CU
constexpr int L1_TILE = L1_THREADS * L1_PPT;               // 8192 k-mers staged in LDS per tile
constexpr int L2_THREADS = 256;                            // level 2: L2_PPT keys per thread
constexpr int L2_PPT = 32;
constexpr int L2_TILE = L2_THREADS * L2_PPT;
constexpr int L2_LDS_BYTES = L2_TILE * 8 + 2048 + 1024 + 1024 + 64;   // keys, gbase[256], hist[256], start[256], scratch[16]
constexpr int L1_MAX_BITS = 8;                             // coarse fan-out 256: ~32 keys (256 B) per run
constexpr int L1_LDS_BYTES = L1_TILE * 8 + 2048 + 1024 + 1024 + 64 + L1_TILE;   // keys, gbase[256], hist[256], start[256], scratch[16], bucket bytes
// Bucket ownership by thread id.  In both partition levels thread t owns bucket t of the tile (its hist / start /
// gbase entries, and in level 2 the running output position my_next), so a workgroup needs at least as many
// threads as the level has buckets: 2^L1_MAX_BITS coarse ones, up to 2^(MAX_BUCKET_BITS - L1_MAX_BITS) fine ones
// in deep mode (bucket_bits 16 -> 256 = every thread of a level-2 workgroup).  A level-2 geometry with fewer
// threads leaves the upper fine buckets without an owner: their gbase is never written and the copy-out stores
// through garbage addresses -- only in deep mode, which is how the aborted run of round 1 (bucket_bits 16 failed,
// 14 passed) presented.  The LDS carve-up below reserves 256 entries for each of the three arrays for the same reason.
static_assert(L1_THREADS >= (1 << L1_MAX_BITS), "level 1: one thread per coarse bucket");
static_assert(L2_THREADS >= (1 << (MAX_BUCKET_BITS - L1_MAX_BITS)), "level 2: one thread per fine bucket, deep mode included");
static_assert((1 << L1_MAX_BITS) <= 256 && (1 << (MAX_BUCKET_BITS - L1_MAX_BITS)) <= 256, "gbase / hist / start hold 256 entries");

// m-mer length of the minimizers (record form of the partition): 4^11 / 2 canonical 11-mers order a 5 Mbp genome's
// ~250 000 minimizers finely enough for 2^13..2^14 buckets, and the word fits the 24-bit multiplies
constexpr int SK_M = 11;

// ---- LDS table kernels ----
constexpr int TABLE_THREADS = 512;                         // 8 waves: one genome per wave at a time
constexpr int TABLE_SCRATCH_BYTES = 128;
constexpr int KEYS_IN_FLIGHT = 4;
constexpr int SLOTS_IN_FLIGHT = 8;                         // independent 2-byte slot loads per lane in the slot fill                          // independent key loads per lane in dict/fill

// per-tile flags (host-built): every file starts on a tile boundary
constexpr uint8_t TILE_META_FIRST = 1;   // first tile of a file: the parser state restarts
constexpr uint8_t TILE_META_FASTQ = 2;   // 4-line FASTQ records instead of FASTA

struct TileSummary {
    uint32_t v[4];   // FASTA: v0 = symbols emitted whatever runs into the tile, v1 = extra if a sequence line does
                     // FASTQ: v[s] = symbols when the tile starts in line phase s
    uint32_t tag;    // FASTA: type of the last line start (0 none, 1 seq, 2 header); FASTQ: 4 | newlines mod 4
};

struct KmerLaunch {
    const uint64_t *sym2;
    const uint64_t *inv;
    uint64_t total_syms;
    const uint64_t *genome_sym_off;   // device, n_genomes + 1
    uint32_t n_genomes;
    int k;
    int bb;
    uint32_t groups_per_thread;
};

// chunk_pre (optional): the exclusive prefix element of every 16-byte chunk of every tile, for parse_pack (parse_chunk_pre_bytes)
// (chunk_pre64: the same for FASTQ tiles, 8 bytes per chunk; nullptr: parse_pack scans those tiles again)
void launch_parse_summarize(hipStream_t s, const uint8_t *raw, uint32_t n_tiles, const uint8_t *tile_meta, TileSummary *sums, uint32_t *chunk_pre,
                            uint64_t *chunk_pre64);
size_t parse_chunk_pre_bytes(uint32_t n_tiles, bool with_fastq);
void launch_parse_scan(hipStream_t s, const TileSummary *sums, uint32_t n_tiles, const uint8_t *tile_meta, uint64_t *tile_off,
                       uint8_t *tile_state, const uint32_t *genome_tile_off, uint32_t n_genomes,
                       uint64_t *genome_sym_off, void *scratch);
size_t parse_scan_scratch_bytes(uint32_t n_tiles);
void launch_parse_pack(hipStream_t s, const uint8_t *raw, uint32_t n_tiles, const uint8_t *tile_meta, const uint64_t *tile_off,
                       const uint8_t *tile_state, uint64_t *sym2, uint64_t *inv, const TileSummary *sums, const uint32_t *chunk_pre,
                       const uint64_t *chunk_pre64);
// single-pass parse (decoupled look-back over the tiles): desc / pieces are scratch of parse_fused_desc_bytes / _piece_bytes; fills
// tile_off[0 .. n_tiles], sym2, inv, genome_sym_off
size_t parse_fused_desc_bytes(uint32_t n_tiles);
size_t parse_fused_piece_bytes(uint32_t n_tiles);
hipError_t launch_parse_fused(hipStream_t s, const uint8_t *raw, uint32_t n_tiles, const uint8_t *tile_meta, uint64_t *desc, uint64_t *pieces,
                              uint64_t *tile_off, uint64_t *sym2, uint64_t *inv, const uint32_t *genome_tile_off, uint32_t n_genomes,
                              uint64_t *genome_sym_off);
void launch_kmer_hist(hipStream_t s, const KmerLaunch &L, uint32_t *counts);
int scatter_b1_bits(int bb);
void launch_kmer_scatter_l1(hipStream_t s, const KmerLaunch &L, const uint64_t *off, const uint64_t *coarse_off,
                            uint32_t *cursor1, uint64_t *out, uint64_t region_stride, int *overflow);
void launch_sum_u32(hipStream_t s, const uint32_t *in, uint64_t n, uint64_t *out);
// record form of the partition (grm_superkmer.hip; 11 <= k <= 32): buckets by minimizer.  Genome g is cut into
// 2^part_bits parts ("virtual genomes" vg = (g << part_bits) + part).
//   level 1: runs of consecutive k-mers with one minimizer occurrence -> 16-byte records, sorted by the coarse bucket bits:
//            region (vg, coarse) = recs1[(vg * 2^b1 + coarse) * rstride ..], rcount1[..] records; part_kmers[vg] = k-mer
//            occurrences of the part; *overflow = 1 when a region would exceed rstride
//   level 2: records -> canonical k-mers, sorted by the fine bits: segment vg * 2^bb + bucket = keys[off[..] .. + len[..]),
//            the segments of region r back to back from r * kstride; *overflow = 1 when a region holds more than kstride
void launch_superkmer_l1(hipStream_t s, const KmerLaunch &L, int b1, int part_bits, void *recs1, uint32_t rstride, uint32_t *rcount1,
                         uint32_t *part_kmers, int *overflow);
void launch_superkmer_l2(hipStream_t s, const void *recs1, uint32_t rstride, const uint32_t *rcount1, uint64_t n_regions, int k, int bb, int b1,
                         uint64_t kstride, uint64_t *keys, uint64_t *off, uint32_t *len, int *overflow);
//   level 2, records only: the region's records sorted by the fine bits, region r again at r * rstride of recs2: segment
//            vg * 2^bb + bucket = recs2[off[..] .. + len[..]) (in records; dict_build decodes them, DictArgs::recs)
//            *overflow = 1 when a segment holds more than 65535 records (its count travels in 16 bits)
void launch_superkmer_l2_records(hipStream_t s, const void *recs1, uint32_t rstride, const uint32_t *rcount1, uint64_t n_regions, int k, int bb, int b1,
                                 void *recs2, uint64_t *off, uint32_t *len, int *overflow);
// (bucket << sb) | sub of dictionary keys under minimizer buckets (launch_dict_bucket_ids for the hashed ones)
void launch_minimizer_bucket_ids(hipStream_t s, const uint64_t *dict, uint64_t n, int k, int bb, int sb, uint32_t *bucket_of, uint32_t *col_of);
int superkmer_max_bits();
int superkmer_coarse_bits(int bb);      // b1 of a partition with 2^bb buckets
int superkmer_lmax();
int superkmer_wide_window(int k);            // m-mers a two-word k-mer takes its minimizer from (21 or 22, in its middle)
void launch_region_hist(hipStream_t s, const uint64_t *keys1, const uint64_t *coarse_off, uint64_t n_regions, int bb,
                        uint32_t *counts);
void launch_kmer_scatter_l2(hipStream_t s, const KmerLaunch &L, const uint64_t *off, const uint64_t *keys1,
                            uint64_t *keys, uint64_t region_stride, uint32_t fine_cap, const uint32_t *cursor1, uint32_t *len_out,
                            int *overflow);
void launch_keys_partition_hist(hipStream_t s, const uint64_t *in, uint64_t n, const uint64_t *genome_key_off,
                                uint32_t n_genomes, int bb, uint32_t *counts);
void launch_keys_partition_scatter(hipStream_t s, const uint64_t *in, uint64_t n, const uint64_t *genome_key_off,
                                   uint32_t n_genomes, int bb, const uint64_t *off, uint32_t *cursor, uint64_t *keys);
void launch_scan_u32(hipStream_t s, const uint32_t *in, uint64_t n, uint64_t *out);
// Segment layout of the partitioned keys: segment idx = genome * 2^bb + bucket.
//   off != nullptr : keys[off[idx] .. off[idx] + (len ? len[idx] : off[idx+1] - off[idx]))      (histogram-sized, dense)
//   off == nullptr : keys[idx * stride .. idx * stride + len[idx])                              (fixed-capacity slots)
struct SegLayout {
    const uint64_t *off;
    const uint32_t *len;
    uint64_t stride;
};
void launch_bucket_dedup(hipStream_t s, uint64_t *keys, const SegLayout &seg, uint64_t n_segments, uint32_t cap_log2,
                         uint32_t abundance_min, uint32_t *len_out, const uint32_t *marks, uint32_t *counts_out, int *overflow);
void launch_superkmer_l2_wide(hipStream_t s, const void *recs1, uint32_t rstride, const uint32_t *rcount1, uint64_t n_regions, int k, int bb, int b1,
                              void *recs2, uint64_t *off, uint32_t *len_out, int *overflow);
hipError_t launch_record_merge(hipStream_t s, const void *recs, uint32_t rstride, const uint32_t *rcount, const uint64_t *roff, const uint32_t *rlen,
                               uint32_t n_genomes, int part_bits, int k, int bb, int b1, uint64_t kstride, int big_log2, uint32_t abundance_min,
                               uint64_t *keys, uint32_t *counts_out, uint64_t *koff, uint32_t *klen, int *overflow);
void launch_record_count(hipStream_t s, const void *recs, uint32_t rstride, const uint32_t *rcount, uint64_t n_regions, int k, int bb, int b1,
                         uint64_t kstride, int cap_log2, uint32_t abundance_min, uint64_t *keys, uint32_t *counts_out, uint64_t *koff,
                         uint32_t *klen, int *overflow, uint8_t *region_big, int *any_big);
void launch_record_dedup_rest(hipStream_t s, const void *recs, uint32_t rstride, const uint32_t *rcount, uint64_t n_regions, int k, int bb, int b1,
                              uint64_t kstride, int cap_log2, uint32_t abundance_min, uint64_t *keys, uint32_t *counts_out, uint64_t *koff,
                              uint32_t *klen, int *overflow, const uint8_t *region_big);
void launch_bucket_dedup_wave(hipStream_t s, uint64_t *keys, const SegLayout &seg, uint64_t n_segments, int wave_cap_log2,
                              uint32_t abundance_min, uint32_t *len_out, uint32_t *marks, uint32_t *counts_out, int *overflow);
// dict_build: per-(bucket, sub-bucket) union over all genomes in an LDS table + the presence bits of
// every distinct k-mer ("entry"), one word-row (64 genomes) at a time.
struct DictArgs {
    const uint64_t *keys;
    int part_bits;          // segments of 2^part_bits parts per genome: segment index = ((genome << part_bits) + part) * 2^bb + bucket
    // record form: the segments hold the 16-byte run records of grm_superkmer.hip (1..22 k-mers each) instead of keys,
    // seg counts records; k as given (keys == nullptr then)
    const ulonglong2 *recs;
    int k;
    // record memo of the record form (grm_kernels.hip, "record memo"): 2^memo_log2 slots; 0 = none
    int memo_log2;
    unsigned long long *memo_stats;     // diagnostics (nullptr: none): sums over (workgroup, word-row) of records held, occurrences asked, found, and their number
    SegLayout seg;
    uint32_t n_genomes;
    int bb, sb;
    uint32_t cap_log2;
    // entries: workgroup wg reserves [wg_base[wg], wg_base[wg] + wg_cnt[wg]) of out_* with one atomic add on
    // *n_out; inside the range entries stand in insertion order (entry id).  flag 1 = one genome, 2 = several.
    uint64_t *out_keys;
    uint8_t *out_flags;
    uint64_t out_cap;
    unsigned long long *n_out;
    uint64_t *wg_base;
    uint32_t *wg_cnt;
    // presence words by (workgroup, word-row, entry id): matrix_s[(wg * n_rows + r) << cap_log2 | id], valid for
    // r >= birth[wg << cap_log2 | id] (the row in which the entry was inserted); nullptr: no bits wanted
    uint64_t *matrix_s;
    uint16_t *birth;
    int *overflow;          // set to 1 when a table overflowed (retry with more sub-buckets), 2 when out_cap did
    uint32_t *need;         // max over overflowing workgroups of their estimated distinct k-mers (sizes the retry)
    // union of the dictionaries of several ranks ("genomes" = ranks, one word-row, at most 63 ranks): every key comes
    // with its rank-local flag at in_flags[in_flag_off[segment] + i]; a key flagged 2 (several genomes inside its
    // rank) is marked as carried by several although only one rank holds it.  nullptr otherwise.
    const uint8_t *in_flags;
    const uint64_t *in_flag_off;
    // rank union, optional: the calling rank's own list is "tracked" -- track[p] = index of the union entry (in out_keys) that
    // holds the key at position p of rank track_g's list (p = in_flag_off[segment] - track_flag_base + i), so that the rank's
    // local entries find their columns through the union's sort instead of a search.  nullptr: not wanted
    uint32_t *track;
    uint32_t track_g;
    uint64_t track_flag_base;
};
void launch_dict_build(hipStream_t s, const DictArgs &a);
// record memo of dict_build's record form: 13/32 * 2^memo_log2 records; LDS bytes: 72 per record (the record, the 24 table slots of
// its k-mers, its word), twice 2^memo_log2 slots of 4 bytes, 32 of counters -- and 2 per slot of the key table (slot of every entry
// id).  (2^10: 416 records, 42.3 KB; with the 2048-slot key table 78.9 KB: two workgroups share a CU.)
constexpr uint32_t dict_memo_entries(int memo_log2) { return memo_log2 > 0 ? (13u << memo_log2) >> 5 : 0u; }
constexpr size_t dict_memo_bytes(int memo_log2, uint32_t cap_log2)
{
    return memo_log2 > 0 ? (size_t)dict_memo_entries(memo_log2) * 72 + ((size_t)8 << memo_log2) + 32 + ((size_t)2 << cap_log2) : 0;
}
// local dictionary in bucket order: entries of workgroup wg copied to [ord_off[wg], ord_off[wg+1]) (ord_off = exclusive
// scan of wg_cnt), and the first entry of every hash bucket (2^bb + 1 offsets)
void launch_dict_export_ordered(hipStream_t s, const uint64_t *keys, const uint8_t *flags, const uint64_t *wg_base, const uint32_t *wg_cnt,
                                const uint64_t *ord_off, uint32_t n_wg, int sb, uint64_t *out_keys, uint8_t *out_flags, uint32_t *bucket_off);
void launch_bucket_offsets(hipStream_t s, const uint64_t *ord_off, int sb, uint32_t n_buckets, uint32_t *bucket_off);
// segment arrays of the rank union over a gathered payload (rank r at r * stride bytes: keys, flags at flags_off, bucket
// offsets at boff_off): key index (in uint64 units of the payload), length, byte offset of the flags.  The union runs over
// n_buckets = 2^(smallest bucket bits of any rank); shifts.d[r] = rank r's bucket bits minus that (a bucket is the TOP
// bits of a hash, so a rank's finer buckets nest inside the coarser ones, in order)
struct RankShifts { uint8_t d[64]; };
void launch_union_segments(hipStream_t s, const uint8_t *payload, uint32_t n_ranks, uint64_t stride, uint64_t flags_off, uint64_t boff_off,
                           uint32_t n_buckets, const RankShifts &shifts, uint64_t *off, uint32_t *len, uint64_t *flag_off);
// column of every local entry: position of its key in the sorted global dictionary, 0xffffffff if filtered / absent
// (prefix_first: scratch of 2^22 + 2 uint32)
void launch_dict_entry_cols(hipStream_t s, const uint64_t *dict, uint64_t n_dict, const uint64_t *entry_keys, uint64_t n_entries, int k,
                            uint32_t *prefix_first, uint32_t *entry_col);
void launch_entry_cols_from_union(hipStream_t s, const uint64_t *wg_base, const uint32_t *wg_cnt, const uint64_t *ord_off, uint32_t n_wg,
                                  const uint32_t *track, const uint32_t *union_col, uint32_t *entry_col);
// matrix[r][entry_col[e]] = presence word of entry e in row r.  entry_major: scratch of n_cols * n_rows words for
// the two-step form (zeroed by the caller where a column may have no local entry); nullptr: direct scattered form
void launch_matrix_permute(hipStream_t s, const uint64_t *matrix_s, const uint16_t *birth, const uint64_t *wg_base,
                           const uint32_t *wg_cnt, const uint32_t *entry_col, uint32_t n_wg, uint32_t n_rows, uint32_t cap_log2,
                           uint64_t *matrix, uint64_t n_cols, uint64_t *entry_major);
void launch_dict_mark(hipStream_t s, const uint64_t *skeys, const uint8_t *sflags, uint64_t n, int filter_singleton,
                      uint32_t *keep);
void launch_dict_select(hipStream_t s, const uint64_t *skeys, const uint32_t *keep, const uint64_t *pos, uint64_t n,
                        uint64_t *dict);
void launch_dict_mark_idx(hipStream_t s, const uint8_t *flags, const uint32_t *sidx, uint64_t n, int filter_singleton, uint32_t *keep);
void launch_dict_select_idx(hipStream_t s, const uint64_t *skeys, const uint32_t *keep, const uint64_t *pos, const uint32_t *sidx, uint64_t n,
                            uint64_t *dict, uint32_t *entry_col);
void launch_dict_bucket_ids(hipStream_t s, const uint64_t *dict, uint64_t n, int bb, int sb, uint32_t *bucket_of,
                            uint32_t *col_of);
void launch_segments_compact(hipStream_t s, const uint64_t *src, const uint32_t *src_cnt, const uint64_t *src_off,
                             const uint32_t *len, const uint64_t *dst_off, uint32_t n_seg, uint64_t *dst,
                             uint32_t *dst_cnt);
void launch_segment_starts(hipStream_t s, const uint32_t *ids, uint64_t n, uint32_t n_ids, uint64_t *start);
void launch_gather_u64(hipStream_t s, const uint64_t *src, const uint32_t *index, uint64_t n, uint64_t *dst);
void launch_gather_u8(hipStream_t s, const uint8_t *src, const uint32_t *index, uint64_t n, uint8_t *dst);
void launch_matrix_fill(hipStream_t s, const uint64_t *keys, const SegLayout &seg,
                        uint32_t n_genomes, int bb, int sb, uint32_t cap_log2, const uint64_t *dkeys,
                        const uint32_t *dcol, const uint64_t *seg_start, uint64_t *matrix, uint64_t n_cols,
                        int *overflow);
void launch_column_popcount(hipStream_t s, const uint64_t *matrix, uint64_t n_rows, uint64_t n_cols,
                            const uint64_t *row_mask, uint32_t *out);
void launch_column_errors(hipStream_t s, const uint64_t *matrix, uint64_t n_rows, uint64_t n_cols, const uint64_t *pos_mask,
                          const uint64_t *neg_mask, uint32_t n_pos, uint32_t n_train, uint32_t *errors, unsigned long long *hist);
void launch_risk_index(hipStream_t s, const uint32_t *errors, uint64_t n_cols, const uint32_t *lut_presence, const uint32_t *lut_absence,
                       uint32_t *by_kmer, uint32_t *by_anti);
void launch_iota_u32(hipStream_t s, uint32_t *p, uint64_t n);
void launch_runs_mark(hipStream_t s, const uint64_t *keys, uint64_t n, uint32_t *head);
void launch_runs_reduce(hipStream_t s, const uint64_t *keys, const uint32_t *counts, const uint32_t *head, const uint32_t *incl,
                        uint64_t n, uint64_t *out_keys, unsigned long long *out_counts);
void launch_runs_keep(hipStream_t s, const unsigned long long *sums, uint64_t n_runs, uint32_t abundance_min, uint32_t *keep);
void launch_runs_emit(hipStream_t s, const uint64_t *keys, const unsigned long long *sums, const uint32_t *keep, const uint32_t *pos,
                      uint64_t n_runs, uint64_t *out_keys, uint32_t *out_counts);
void launch_split_pairs_u64(hipStream_t s, const uint64_t *pairs, uint64_t n, uint64_t *hi, uint64_t *lo);
void launch_join_pairs_u64(hipStream_t s, const uint64_t *hi, const uint64_t *lo, uint64_t n, uint64_t *pairs);
// dictionary sort (grm_dictsort.hip): keys below 2^key_bits -> okeys ascending, oidx = every key's position in `keys`; *too_big (device,
// zeroed by the caller) = 1: a key range did not fit LDS and the output is NOT sorted (take sort_pairs_u64_u32)
size_t dict_sort_scratch_bytes(uint64_t n);
hipError_t launch_dict_sort(hipStream_t s, const uint64_t *keys, uint64_t n, int key_bits, uint64_t *okeys, uint32_t *oidx, void *scratch, int *too_big);
hipError_t set_max_dynamic_lds();
void set_table_tuning(int keys_in_flight, int threads);

// rocPRIM-backed plumbing for the (small) dictionary: radix sort of pairs, exclusive scan
hipError_t sort_pairs_u64_u8(hipStream_t s, const uint64_t *kin, uint64_t *kout, const uint8_t *vin, uint8_t *vout,
                             uint64_t n, void *tmp, size_t &tmp_bytes);
hipError_t sort_pairs_u32_u32(hipStream_t s, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                              uint64_t n, int end_bit, void *tmp, size_t &tmp_bytes);
hipError_t sort_pairs_u64_u32(hipStream_t s, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                              uint64_t n, void *tmp, size_t &tmp_bytes);
hipError_t exclusive_scan_u32_u64(hipStream_t s, const uint32_t *in, uint64_t *out, uint64_t n, void *tmp,
                                  size_t &tmp_bytes);

hipError_t exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n, void *tmp, size_t &tmp_bytes);
hipError_t inclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint64_t n, void *tmp, size_t &tmp_bytes);

// ---- two-word k-mers (grm_wide.hip) ----
void launch_wide_extract(hipStream_t s, const uint64_t *sym2, const uint64_t *inv, uint64_t total_syms, int k, uint64_t *khi,
                         uint64_t *klo, unsigned long long *n_valid);
void launch_wide_mark(hipStream_t s, const uint64_t *khi, const uint64_t *klo, const uint32_t *pos, const uint64_t *gso,
                      uint32_t n_genomes, uint32_t n, uint32_t *key_head, uint32_t *kg_head);
void launch_wide_sub_start(hipStream_t s, const uint32_t *kg_head, const uint32_t *sub_id, uint32_t n, uint32_t n_sub,
                           uint32_t *sub_start);
void launch_wide_sub(hipStream_t s, const uint32_t *sub_start, const uint32_t *key_head, uint32_t n_sub, uint32_t abundance_min,
                     uint32_t *sub_key_head, uint32_t *sub_ok);
void launch_wide_key_count(hipStream_t s, const uint32_t *key_incl, const uint32_t *sub_ok, uint32_t n_sub, uint32_t *carriers);
void launch_wide_keep(hipStream_t s, const uint32_t *carriers, uint32_t n_keys, uint32_t min_carriers, uint32_t *keep);
void launch_wide_emit(hipStream_t s, const uint64_t *khi, const uint64_t *klo, const uint32_t *pos, const uint64_t *gso,
                      uint32_t n_genomes, const uint32_t *sub_start, const uint32_t *sub_key_head, const uint32_t *sub_ok,
                      const uint32_t *key_incl, const uint32_t *keep, const uint32_t *col, uint32_t n_sub, uint64_t *dict,
                      uint64_t *matrix, uint64_t n_cols);
void launch_wide_set(hipStream_t s, const uint64_t *khi, const uint64_t *klo, const uint32_t *sub_start, const uint32_t *sub_ok,
                     const uint32_t *out_pos, uint32_t n_sub, uint64_t *kmers, uint32_t *counts);

// ---- three- and four-word k-mers, 65 <= k <= 128 (grm_multi.hip) ----
struct MultiWords { const uint64_t *w[4]; };       // structure of arrays, w[0] most significant
struct MultiWordsOut { uint64_t *w[4]; };
void launch_multi_extract(hipStream_t s, int words, const uint64_t *sym2, const uint64_t *inv, uint64_t total_syms, int k,
                          const MultiWordsOut &out, unsigned long long *n_valid);
void launch_multi_mark(hipStream_t s, int words, const MultiWords &S, const uint32_t *pos, const uint64_t *gso, uint32_t n_genomes, uint32_t n,
                       uint32_t *key_head, uint32_t *kg_head);
void launch_multi_emit(hipStream_t s, int words, const MultiWords &S, const uint32_t *pos, const uint64_t *gso, uint32_t n_genomes,
                       const uint32_t *sub_start, const uint32_t *sub_key_head, const uint32_t *sub_ok, const uint32_t *key_incl,
                       const uint32_t *keep, const uint32_t *col, uint32_t n_sub, uint64_t *dict, uint64_t *matrix, uint64_t n_cols);
void launch_multi_set(hipStream_t s, int words, const MultiWords &S, const uint32_t *sub_start, const uint32_t *sub_ok, const uint32_t *out_pos,
                      uint32_t n_sub, uint64_t *kmers, uint32_t *counts);
void launch_multi_split(hipStream_t s, int words, const uint64_t *keys, uint64_t n, const MultiWordsOut &out, uint64_t at);
// the staged calls over the sort path (k > 64; 33..64 with abundance-min > 1)
void launch_multi_flags(hipStream_t s, const uint64_t *matrix, uint64_t n_rows, uint64_t n_cols, uint8_t *flags);
void launch_multi_flag_mark(hipStream_t s, const uint8_t *flags, uint64_t n, uint32_t *mark);
void launch_multi_split_marked(hipStream_t s, int words, const uint64_t *keys, const uint32_t *mark, const uint32_t *pos, uint64_t n,
                               const MultiWordsOut &out, uint64_t at);
void launch_multi_lookup(hipStream_t s, int words, const uint64_t *mine, uint64_t n, const uint64_t *dict, uint64_t n_dict, uint32_t *col);
void launch_multi_scatter_cols(hipStream_t s, const uint64_t *own, uint64_t n_own, const uint32_t *col, uint64_t n_rows, uint64_t *out, uint64_t n_cols);

// ---- two-word k-mers, hash-partition pipeline (grm_wide_hash.hip) ----
void launch_wh_hist(hipStream_t s, const KmerLaunch &L, uint32_t *counts);
void launch_wh_l1(hipStream_t s, const KmerLaunch &L, const uint64_t *off, uint32_t *cursor1, void *out, uint64_t region_stride, int *overflow);
void launch_wh_l2(hipStream_t s, const KmerLaunch &L, const uint64_t *off, const void *keys1, void *keys, uint64_t region_stride,
                  uint32_t fine_cap, const uint32_t *cursor1, uint32_t *len_out, int *overflow);
// entries leave staged at wg * cap in entry-id order; presence words at matrix_s[wg][row][entry id] (nullptr: no bits)
void launch_wh_top64(hipStream_t s, const uint64_t *hi, const uint64_t *lo, uint64_t n, int k, uint64_t *top);
// *too_long (device, zeroed by the caller) = 1: a group of equal top words was longer than the kernel orders; the output is then NOT sorted
void launch_wh_ties(hipStream_t s, const uint64_t *top_sorted, uint32_t *order, const uint64_t *lo, uint64_t n, int *too_long);
void launch_wh_dict_build(hipStream_t s, const void *keys, const SegLayout &seg, uint32_t n_genomes, int bb, int sb, uint32_t cap_log2,
                          uint64_t *stage_lo, uint64_t *stage_hi, uint8_t *stage_flags, uint32_t *stage_cnt, uint64_t *matrix_s,
                          uint16_t *birth, int *overflow, uint32_t *need, int recs_k = 0, int part_bits = 0);
void launch_wh_dict_gather(hipStream_t s, const uint64_t *stage_lo, const uint64_t *stage_hi, const uint8_t *stage_flags,
                           const uint64_t *stage_off, uint32_t n_wg, uint32_t cap, uint64_t *out_lo, uint64_t *out_hi, uint8_t *out_flags);
void launch_wh_mark(hipStream_t s, const uint64_t *s_hi, const uint64_t *s_lo, const uint8_t *flags, const uint32_t *order, uint64_t n,
                    int filter_singleton, uint32_t *keep);
void launch_wh_select(hipStream_t s, const uint64_t *s_hi, const uint64_t *s_lo, const uint32_t *keep, const uint32_t *pos, uint64_t n,
                      uint64_t *dict);
void launch_wh_entry_cols(hipStream_t s, const uint64_t *s_hi, const uint64_t *s_lo, const uint32_t *keep, const uint32_t *pos, uint64_t n,
                          const uint64_t *e_hi, const uint64_t *e_lo, uint64_t n_entries, uint32_t *entry_col);
// one GPU (the sorted entries are the local ones): entry_col[order[i]] = keep[i] ? pos[i] : 0xffffffff
void launch_wh_cols_from_order(hipStream_t s, const uint32_t *order, const uint32_t *keep, const uint32_t *pos, uint64_t n, uint32_t *entry_col);
hipError_t wh_set_max_dynamic_lds();

// ---- zlib encoder on the device (grm_deflate.hip): the HDF5 chunks of kmer_matrix / kmer_sequences as finished streams ----
// one wave per chunk; out: deflate_chunk_cap(raw bytes of a chunk) bytes per chunk of the launch, sizes[c] = stream length;
// tok: chunk_cols uint16 per chunk of the launch (scratch)
uint64_t deflate_chunk_cap(uint64_t raw_bytes);
hipError_t launch_deflate_rows(hipStream_t s, const uint64_t *matrix, uint64_t n_cols, uint32_t chunk_cols, uint32_t chunks_per_row,
                               uint64_t first_chunk, uint32_t n_chunks, uint16_t *tok, uint8_t *out, uint64_t cap, uint32_t *sizes);
hipError_t launch_deflate_kmer_strings(hipStream_t s, const uint64_t *kmers, uint64_t n, int words, int k, uint32_t chunk_elems,
                                       uint64_t first_chunk, uint32_t n_chunks, uint8_t *out, uint64_t cap, uint32_t *sizes);
// the streams of a launch moved next to each other: dst[off[c] ..), off[c] multiples of 16
hipError_t launch_deflate_compact(hipStream_t s, const uint8_t *src, uint64_t cap, const uint32_t *sizes, uint32_t n_chunks, uint64_t *off, uint8_t *dst);

}  // namespace grm
