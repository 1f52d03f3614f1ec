/*
 * grm_kmer.h -- C ABI of libgrmkmer.so, the MI355X (gfx950) k-mer-matrix engine.
 *
 * Drop-in boundary.  The reference (editnori/Genomic-Resistance-Mapping-GRM-) has no
 * in-process FFI for this path: it shells out to four native executables
 *     bin/dsk/dsk                         (src/app.py:1372)
 *     bin/ray/Ray  under mpiexec -n 4     (src/app.py:1310, conf: src/app.py:3820-3833)
 *     kmer_tools/multidsk                 (bin/kover/core/kover/dataset/tools/kmer_count.py:28-37, :44-53)
 *     kmer_tools/dsk2kover                (bin/kover/core/kover/dataset/tools/kmer_pack.py:28-36)
 * The replacement executables (genomic-resistance-mapping-grm-_amd/cli/) keep that argv
 * surface and call the functions below through ctypes; INTEGRATION.md shows the binding.
 * The first block restates SURVEY.md 8(b)'s operator API verbatim; the "batch" block is
 * the fused, device-resident form of the same path (parse -> partition -> dictionary ->
 * presence bits) that the executables and bench.py actually drive.
 *
 * Conventions
 *   - every function returns 0 on success or a negative grm_status; grm_last_error(ctx)
 *     returns a message owned by the ctx (valid until the next call on that ctx).
 *   - k-mers are 2-bit packed, first base most significant, A=0 C=1 T=2 G=3 (GATB),
 *     canonical = min(forward, reverse complement); `words` = ceil(k / 32) uint64 per k-mer, most significant
 *     first (k = 1 .. 128, the reference's range: bin/kover/kover:114).  Inputs: FASTA or 4-line FASTQ, optionally gzip.
 *   - matrix: uint64 [n_rows][n_kmers] row-major, genome i -> row i/64, bit 63-(i%64)
 *     (bin/kover/core/kover/utils.py:133-156); columns ascending by k-mer value.
 *   - a grm_ctx is single-caller; HIP streams inside are the parallelism.
 *   - lifetime: every handle made from a context (grm_batch, grm_matrix, grm_kmer_set, grm_dict_accum) holds a
 *     reference to it.  grm_destroy drops the owner's reference; the device streams are released when the last
 *     handle has been freed, so handles may be freed after grm_destroy (a caller that unwinds on an error does).
 *     After grm_destroy the grm_ctx pointer is the owner's no longer: do not pass it to anything but grm_destroy,
 *     grm_ctx_live_handles and grm_last_error, which recognise a context that has ceased to exist (no-op / 0 / a fixed text).
 *   - plain pointers and sizes only: no Python / torch types cross this boundary.
 *     Arguments named dev_* are HIP device pointers on the ctx's device.
 */
#ifndef GRM_KMER_H
#define GRM_KMER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GRM_OK = 0,
    GRM_ERR_ARG = -1,          /* bad argument (k out of range, NULL, ...) */
    GRM_ERR_NO_DEVICE = -2,    /* no usable gfx950 device: there is NO CPU fallback */
    GRM_ERR_HIP = -3,          /* HIP runtime error, text in grm_last_error */
    GRM_ERR_IO = -4,
    GRM_ERR_OOM = -5,
    GRM_ERR_UNSUPPORTED = -6,  /* e.g. per-k-mer counts of a whole batch (grm_batch_partition_counts) at k > 32 */
    GRM_ERR_STATE = -7,        /* calls out of order */
    GRM_ERR_OVERFLOW = -8,     /* LDS table overflow that retries could not cure */
    GRM_ERR_HDF5 = -9
} grm_status;

typedef struct grm_ctx grm_ctx;
typedef struct grm_kmer_set grm_kmer_set;
typedef struct grm_matrix grm_matrix;
typedef struct grm_batch grm_batch;

/* ---- context ------------------------------------------------------------------------ */
/* device_ordinal >= 0: HIP device.  There is no CPU mode: -1 is rejected (GRM_ERR_NO_DEVICE). */
grm_ctx    *grm_create(int device_ordinal, int n_streams);
void        grm_destroy(grm_ctx *);
/* handles made from the context that have not been freed yet */
int         grm_ctx_live_handles(const grm_ctx *);
const char *grm_last_error(grm_ctx *);
const char *grm_version(void);
/* tuning knobs (tests and measurements; value < 0 restores the automatic choice).  Geometry: "bucket_bits", "cap_log2" (LDS table
 * slots, log2), "sub_bits", "groups_per_thread", "keys_in_flight", "table_threads" (256 / 512 / 1024).  Paths: "records" (0: never the
 * minimizer-record form of the partition, 1: from k = 11), "rec_keys" (records always expanded to key segments), "no_slots" (probing
 * fill), "direct_permute", "dense_layout", "dedup_wg", "dedup_cap_shift", "wide_sort" (k > 32 through the sort-based path), "no_union"
 * (gathered rank dictionaries sorted as a whole), "parse_fused" (single-pass parse kernel).  Record form: "rec_bucket_shift",
 * "rec_part_bits", "rec_coarse", "rec_memo" (log2 of dict_build's record memo base, 0 = none), "memo_stats" (1: count memo hits, see
 * grm_batch_memo_stats), "rec_count" (0: a counting partition never takes the record form), "rec_count_cap" (8: its wave tables hold
 * 256 slots instead of 512; 12 / 13: log2 slots of the workgroup's table when genomes are counted over their parts).  Host: "upload_slab_kb". */
int         grm_set_option(grm_ctx *, const char *name, int value);
/* per-kernel device timings (HIP events on the engine's stream) */
int         grm_timing_enable(grm_ctx *, int on);
int         grm_timing_reset(grm_ctx *);
int         grm_timing_count(grm_ctx *);
int         grm_timing_get(grm_ctx *, int i, char *name, size_t name_cap, double *ms, uint64_t *units);

/* ---- SURVEY 8(b) operator API ----------------------------------------------------------- */
/* replaces one DSK run = one multidsk list line (kmer_count.py:28-53): all files of ONE
 * genome -> sorted distinct canonical k-mers with count >= abundance_min. */
int  grm_count_genome(grm_ctx *, const char *const *paths, int n_paths, int k, uint32_t abundance_min,
                      grm_kmer_set **out);
int  grm_count_genome_buffers(grm_ctx *, const void *const *bufs, const size_t *lens, int n_bufs, int k,
                              uint32_t abundance_min, grm_kmer_set **out);
/* pooled merge (DSK counts all listed files as ONE pool, src/app.py:1371-1372): sums the counts of
 * equal k-mers over the sets and keeps those whose total reaches abundance_min; k <= 32 */
int  grm_merge_counted_sets(grm_ctx *, grm_kmer_set *const *sets, int n_sets, uint32_t abundance_min, grm_kmer_set **out);
/* build a set from host arrays (used by dsk2kover when it re-loads multidsk's artefacts); for
 * k > 32 `kmers` holds ceil(k/32)*n words (most significant word of each k-mer first) */
int  grm_kmer_set_from_host(grm_ctx *, const uint64_t *kmers, const uint32_t *counts, size_t n, int k,
                            grm_kmer_set **out);
size_t          grm_kmer_set_size(const grm_kmer_set *);
int             grm_kmer_set_k(const grm_kmer_set *);
int             grm_kmer_set_words(const grm_kmer_set *);
uint64_t        grm_kmer_set_occurrences(const grm_kmer_set *);
/* host arrays (size*words / size).  A set counted on the device stays in HBM, where
 * grm_build_matrix consumes it; the first call of either accessor downloads it (NULL on failure). */
const uint64_t *grm_kmer_set_kmers(grm_kmer_set *);
const uint32_t *grm_kmer_set_counts(grm_kmer_set *);
void            grm_kmer_set_free(grm_kmer_set *);

/* replaces dsk2kover's merge (kmer_pack.py:28-36): N per-genome sets -> dictionary + matrix */
int  grm_build_matrix(grm_ctx *, grm_kmer_set *const *sets, int n_genomes, int filter_singleton,
                      grm_matrix **out);
size_t          grm_matrix_n_kmers(const grm_matrix *);
size_t          grm_matrix_n_rows(const grm_matrix *);
int             grm_matrix_n_genomes(const grm_matrix *);
int             grm_matrix_k(const grm_matrix *);
int             grm_matrix_words(const grm_matrix *);
const uint64_t *grm_matrix_kmers(grm_matrix *);              /* host copy (lazy D2H) */
const uint64_t *grm_matrix_data(grm_matrix *);               /* host copy (lazy D2H) */
const void     *grm_matrix_dev_kmers(const grm_matrix *);    /* device pointers */
const void     *grm_matrix_dev_data(const grm_matrix *);
/* per-column carrier count = popcount down the column (the learner's sum_rows with an
 * all-ones mask, learning/common/rules.py:243-262); host array of n_kmers uint32 */
int             grm_matrix_column_counts(grm_matrix *, uint32_t *out);
/* masked form = KmerRuleClassifications.sum_rows (rules.py:201-267, popcount.pyx:76-95):
 * row_mask: n_rows host words (bit 63-(i%64) of word i/64 selects genome i); out: n_kmers uint32 */
int             grm_matrix_sum_rows(grm_matrix *, const uint64_t *row_mask, uint32_t *out);
/* host-only matrix from caller arrays (copied): rows gathered from several ranks, or tests.
 * Only the accessors and the two writers work on it (no device).  kmers: n_kmers words for
 * k <= 32, ceil(k/32)*n_kmers (most significant word first) above. */
int             grm_matrix_from_host(const uint64_t *kmers, const uint64_t *data, size_t n_kmers, int n_genomes, int k,
                                     grm_matrix **out);
/* places a host-only matrix (grm_matrix_from_host, e.g. rows read back from a .kover file) in HBM: afterwards the
 * device entry points (column_counts, sum_rows, risk_errors) work on it */
int             grm_matrix_to_device(grm_ctx *, grm_matrix *);
/* `kover dataset split` risk tables (dataset/split.py:171-188), device part: one sweep of the matrix with the row masks
 * of the positive / negative training genomes: errors[c] = (n_pos - popcount(col & pos)) + popcount(col & neg), kept in
 * HBM, and hist_out[e] = number of k-mers with e errors (n_train + 1 host counters).  The caller rounds and
 * unique-indexes the at most n_train + 1 distinct risks as the reference does and hands two look-up tables back:
 * by_kmer[c] = lut_presence[errors[c]], by_anti[c] = lut_absence[errors[c]] (n_kmers host uint32 each). */
int             grm_matrix_risk_errors(grm_matrix *, const uint64_t *pos_mask, const uint64_t *neg_mask, uint32_t n_pos, uint32_t n_train,
                                       uint64_t *hist_out);
int             grm_matrix_risk_index(grm_matrix *, const uint32_t *lut_presence, const uint32_t *lut_absence, uint32_t n_lut,
                                      uint32_t *by_kmer_out, uint32_t *by_anti_out);
const char     *grm_matrix_last_error(const grm_matrix *);
void            grm_matrix_free(grm_matrix *);

/* Ray-Surveyor-compatible TSV (layout read by dataset/create.py:121-137,241) */
int  grm_write_tsv(grm_matrix *, const char *const *genome_ids, const char *path);
/* rows [first_kmer, first_kmer + n_kmers) of the same file (a TSV row = one k-mer), written in place at their final offsets --
 * every row has the same byte length (create.py:130-137) -- and the header when first_kmer == 0.  The file is opened without
 * truncation and not renamed: several writers (the ranks Ray runs as under mpiexec, src/app.py:1310) fill one file. */
int  grm_write_tsv_slice(grm_matrix *, const char *const *genome_ids, const char *path, uint64_t first_kmer, uint64_t n_kmers);
/* appends kmer_sequences / kmer_matrix / kmer_by_matrix_column to the EXISTING Kover HDF5
 * exactly as dsk2kover does (schema dataset/create.py:214-238); libhdf5 is dlopen()ed. */
int  grm_write_kover_h5(grm_matrix *, const char *existing_h5_path, int gzip_level, int chunk_cols);
/* On failure (GRM_ERR_HDF5) whatever exists of the three datasets is unlinked before the file is closed: the caller
 * (kmer_pack.py:28) ignores the tool's return code, and unwritten chunks would read back as "absent everywhere".
 * A matrix that lives on the device is deflated THERE (one wave per HDF5 chunk, zlib streams handed to H5Dwrite_chunk; any
 * gzip_level >= 1 gives the same streams, the level only goes into the dataset's filter parameters); host-only matrices,
 * GRM_DEFLATE=host and libhdf5 < 1.10.3 take host threads (GRM_WRITER_THREADS; default: the process's CPU share).
 *
 * The same streams for callers that write the file elsewhere -- the ranks of a multi-GPU run deflate the word-rows they
 * filled and rank 0 appends them (replaces the gather of raw rows; create.py:365-390 seen from N GPUs):
 * chunk i of kmer_matrix in row-major order (word-row i / chunks_per_row, columns (i % chunks_per_row) * cw .. + cw with
 * cw = min(n_kmers, chunk_cols), zero-padded) is lens[i] bytes at streams + starts[i]; kmer_sequences likewise in chunks of
 * min(n_kmers, chunk_elems) strings (the writer uses 65536).  The three arrays are malloc'ed: grm_host_free each. */
int  grm_matrix_deflate_rows(grm_matrix *, int chunk_cols, unsigned char **streams, uint64_t **starts, uint32_t **lens, uint64_t *n_chunks);
int  grm_matrix_deflate_kmer_strings(grm_matrix *, int chunk_elems, unsigned char **streams, uint64_t **starts, uint32_t **lens, uint64_t *n_chunks);
void grm_host_free(void *);
/* grm_write_kover_h5 with the kmer_matrix chunks made elsewhere: `dict` gives the dictionary (kmer_sequences,
 * kmer_by_matrix_column; on the device or host-only); part p holds the chunks of word-rows [row0[p], row0[p] + rows[p]) as
 * grm_matrix_deflate_rows returns them; the parts cover rows 0 .. n_rows_total in order.  gzip_level >= 1. */
int  grm_write_kover_h5_parts(grm_matrix *dict, const char *existing_h5_path, int gzip_level, int chunk_cols, uint64_t n_rows_total, int n_parts,
                              const unsigned char *const *streams, const uint64_t *const *starts, const uint32_t *const *lens,
                              const uint64_t *row0, const uint64_t *rows);

/* ---- fused device-resident batch path ------------------------------------------------- */
int  grm_batch_create(grm_ctx *, int n_genomes, grm_batch **out);
/* append one file image (FASTA) to genome `genome_index`; bytes are copied */
int  grm_batch_add(grm_batch *, int genome_index, const void *buf, size_t len);
int  grm_batch_add_file(grm_batch *, int genome_index, const char *path);
/* assemble + host-to-device copy; afterwards the inputs are resident in HBM */
int  grm_batch_upload(grm_batch *);
/* whole hot path on the resident inputs: == partition + local_dict + set_global_dict(own) + fill */
int  grm_batch_run(grm_batch *, int k, uint32_t abundance_min, int filter_singleton, grm_matrix **out);
/* the same path in stages, so that the host can put ONE collective between them when the
 * genomes are sharded over several GPUs.  Dictionary entries are rows of ceil(k / 32) 64-bit words, most significant
 * first: 8-byte keys for k <= 32, (hi, lo) pairs for 33 <= k <= 64, three or four words up to k = 128 (the whole range of
 * grm_batch_run; k > 64, and k > 32 with abundance-min > 1, go through the sort path in stages: the lists they export are sorted, not
 * grouped by hash bucket); flags are one byte each (1 = carried by one local genome, 2 = by several); n counts k-mers, not words. */
int  grm_batch_partition(grm_batch *, int k, uint32_t abundance_min);
/* same, but keeps per-k-mer occurrence counts so that grm_batch_genome_set can return them
 * (multidsk: one batch, one sorted counted set per genome) */
int  grm_batch_partition_counts(grm_batch *, int k, uint32_t abundance_min);
int  grm_batch_local_dict(grm_batch *, uint64_t *n_local);
int  grm_batch_export_dict(grm_batch *, void *dev_keys_out, void *dev_flags_out);
int  grm_batch_set_global_dict(grm_batch *, const void *dev_keys, const void *dev_flags, uint64_t n,
                               int filter_singleton, uint64_t *n_kmers);
int  grm_batch_fill(grm_batch *, grm_matrix **out);
/* The exchange as ONE all-gather of fixed-stride records ("one RCCL all-gather over xGMI").  A rank's
 * record, for n_max = the largest n_local of any rank and the largest bucket_bits of any rank:
 *     [0, n_max * 8 * words)   keys grouped by hash bucket     [flags_off, +n_max)   flags
 *     [boff_off, +4 * (2^bucket_bits + 1))   uint32 index of the first entry of every hash bucket
 * grm_exchange_layout gives the offsets and the record stride (a multiple of 16 bytes); every rank writes its record
 * with grm_batch_export_dict_ordered, the host all-gathers n_ranks * stride bytes, and
 * grm_batch_set_global_dict_gathered builds the global dictionary from the gathered payload (counts / bucket_bits:
 * host arrays with n_local and grm_batch_bucket_bits of every rank).  grm_batch_bucket_bits returns the bucket geometry
 * as ranks compare it: the bucket bits in the low byte, + 0x100 when the buckets are minimizer buckets (the record form
 * of the partition); grm_exchange_layout reads the low byte only, so pass the largest low byte of any rank.  Replaces
 * the k-mer delivery between Ray's MPI ranks (src/app.py:1310). */
void grm_exchange_layout(uint64_t n_max, int words, int bucket_bits, uint64_t *flags_off, uint64_t *boff_off, uint64_t *stride);
/* The last GRM_EXCHANGE_HEADER_BYTES of every record: { uint64 n_local; uint32 bucket-bits code (grm_batch_bucket_bits);
 * uint32 GRM_EXCHANGE_MAGIC }.  With it a step needs ONE collective: the layout (n_cap, bucket_bits) is fixed from what the
 * previous step saw, grm_batch_export_dict_record writes the header always and the lists when they fit (*fits), the host
 * all-gathers, reads the n_ranks headers, and -- only if some rank did not fit -- repeats with the larger layout every rank now
 * knows.  The reference has no such step (Ray's ranks exchange k-mers in many small MPI messages, src/app.py:1310). */
#define GRM_EXCHANGE_HEADER_BYTES 16
#define GRM_EXCHANGE_MAGIC 0x584d5247u /* "GRMX" */
int  grm_batch_export_dict_record(grm_batch *, void *dev_record, uint64_t n_cap, int bucket_bits, int *fits);
int  grm_batch_bucket_bits(const grm_batch *);
int  grm_batch_export_dict_ordered(grm_batch *, void *dev_record, uint64_t flags_off, uint64_t boff_off);
int  grm_batch_set_global_dict_gathered(grm_batch *, const void *dev_payload, int n_ranks, uint64_t n_max, const uint64_t *counts,
                                        const int *bucket_bits, int filter_singleton, uint64_t *n_kmers);
/* the same when the caller knows which record of the payload is this batch's own (my_rank: its index; the record must be the one
 * grm_batch_export_dict_ordered wrote last): the union then notes, for every entry of that list, the union entry it fell into,
 * and the batch's entries take their columns from the union's sort instead of searching the finished dictionary */
int  grm_batch_set_global_dict_gathered_from(grm_batch *, const void *dev_payload, int n_ranks, int my_rank, uint64_t n_max,
                                             const uint64_t *counts, const int *bucket_bits, int filter_singleton, uint64_t *n_kmers);
/* Inputs larger than one device batch (thousands of genomes): two passes over chunks of genomes.
 * Pass 1, per chunk: upload, grm_batch_partition, grm_batch_local_dict, grm_dict_accum_add, free.
 * Pass 2, per chunk: upload, partition, local_dict, grm_batch_set_global_dict_accum, grm_batch_fill.
 * A k-mer seen in several chunks is merged like a k-mer seen on several GPUs; grm_matrix_stack_rows
 * puts the chunks' word-rows together (every chunk but the last: a multiple of 64 genomes).
 * Replaces the multidsk -> dsk2kover hand-over through per-genome files for contig inputs
 * (dataset/create.py:365-390) when they do not fit HBM at once. */
typedef struct grm_dict_accum grm_dict_accum;
int      grm_dict_accum_create(grm_ctx *, grm_dict_accum **out);
int      grm_dict_accum_add(grm_dict_accum *, grm_batch *);
uint64_t grm_dict_accum_size(const grm_dict_accum *);
void     grm_dict_accum_free(grm_dict_accum *);
int      grm_batch_set_global_dict_accum(grm_batch *, const grm_dict_accum *, int filter_singleton, uint64_t *n_kmers);
int      grm_matrix_stack_rows(grm_matrix *const *parts, int n_parts, grm_matrix **out);
/* statistics of the last partition */
uint64_t grm_batch_n_symbols(const grm_batch *);
uint64_t grm_batch_n_occurrences(const grm_batch *);     /* valid k-mer windows */
uint64_t grm_batch_input_bytes(const grm_batch *);
uint64_t grm_batch_n_local(const grm_batch *);           /* entries of the last local dictionary */
/* record memo of the last dictionary launch (option "memo_stats" = 1; zeros when the launch had no memo): sums over
 * (workgroup, word-row) of { records held, record occurrences asked, occurrences found in the memo, (workgroup, word-row) count } */
int      grm_batch_memo_stats(const grm_batch *, uint64_t *out4);
/* per-genome distinct k-mers after the last partition (+dedup): sorted set for genome g */
int  grm_batch_genome_set(grm_batch *, int genome_index, grm_kmer_set **out);
void grm_batch_free(grm_batch *);

#ifdef __cplusplus
}
#endif
#endif
