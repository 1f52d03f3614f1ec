// grm_api.cpp -- C ABI of libgrmkmer.so (include/grm_kmer.h): host orchestration of the
// gfx950 kernels in grm_kernels.hip.  No CPU fallback: every compute entry point needs a
// HIP device and fails loudly without one.
//
// Reference call sites replaced (the reference has no in-process API for this path):
//   multidsk  bin/kover/core/kover/dataset/tools/kmer_count.py:28-37,44-53
//   dsk2kover bin/kover/core/kover/dataset/tools/kmer_pack.py:28-36
//   dsk       src/app.py:1372           Ray Surveyor  src/app.py:1310
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/grm_kmer.h"
#include "grm_internal.h"

using namespace grm;

// --------------------------------------------------------------------------------------
// context
// --------------------------------------------------------------------------------------
struct TimingRec {
    std::string name;
    hipEvent_t e0, e1;
    uint64_t units;
};

struct grm_ctx {
    // Lifetime: the owner (grm_create / grm_destroy) holds one reference and every handle made from the context (batch, matrix,
    // k-mer set, dictionary accumulator) holds another: grm_destroy with handles still alive only drops the owner's, and the
    // streams go when the last handle is freed.  (Python finalises a Batch kept alive by a traceback AFTER its Context was closed:
    // its free then synchronised a stream of a deleted context -- round 2's aborts at process exit.)
    std::atomic<int> refs{1};
    std::atomic<bool> owner_gone{false};
    int device = 0;
    hipStream_t stream = nullptr;
    // uploads run on their own stream: grm_batch_upload of the NEXT batch may be called from a
    // second host thread while this context computes on `stream` (kover_dataset.counted_sets)
    hipStream_t copy_stream = nullptr;
    std::mutex err_mu;
    std::string err;
    bool timing = false;
    std::vector<TimingRec> recs;
    // options (<0: automatic)
    int opt_groups_per_thread = -1;
    int opt_bucket_bits = -1;
    int opt_cap_log2 = -1;
    int opt_sub_bits = -1;
    int opt_no_slots = -1;      // > 0: force the probing form of the fill (tests)
    int opt_keys_in_flight = -1, opt_table_threads = -1;
    int opt_wide_sort = -1;     // > 0: k > 32 always through the sort-based path (tests)
    int opt_upload_slab_kb = -1; // pinned upload slab size in KiB (tests; default 128 MiB)
    int opt_direct_permute = -1; // > 0: scattered single-step form of the fill (tests, measurements)
    int opt_dedup_wg = -1;       // > 0: per-segment dedup in the workgroup form only (tests)
    int opt_dedup_cap_shift = -1; // > 0: wave-form dedup tables 2^shift times larger than the sizing rule asks for (measurements)
    int opt_dense_layout = -1;   // > 0: histogram-sized dense partition layout (tests, measurements)
    int opt_no_union = -1;       // > 0: gathered rank dictionaries are sorted as a whole (tests)
    int opt_records = -1;        // 0: never use the record (minimizer) form of the partition; 1: also for 11 <= k < 19 (tests, measurements)
    int opt_rec_bucket_shift = -1; // record form: bucket bits on top of the key form's choice (default 1)
    int opt_rec_part_bits = -1;  // record form: parts per genome, log2 (tests)
    int opt_rec_keys = -1;       // > 0: record form always expands to key segments in level 2 (tests, measurements)
    int opt_rec_count = -1;      // 0: counting partitions never take the record form
    int opt_rec_count_cap = -1;  // counting over records: log2 slots of a wave's table (8 / 9; tests of the shared-table path), or of record_merge's table (12 / 13)
    int opt_rec_coarse = -1;     // record form: coarse bucket bits of level 1 (tests, measurements); < 0 = min(bucket bits, 9)
    int opt_dict_sort_prim = -1; // > 0: the dictionary is sorted by rocPRIM's radix sort instead of the key-range sort of grm_dictsort.hip (tests)
    int opt_parse_fused = -1;    // > 0: the single-pass parse kernel (decoupled look-back) instead of summarize / scan / pack -- measured SLOWER
                                 // (7.1 against 5.0 ms at 1000 x 5 Mbp): kept for tests and measurements
    int opt_memo_stats = -1;     // > 0: dict_build counts what its record memo held / was asked / found (grm_batch_memo_stats)
    int opt_rec_memo = -1;       // record memo of dict_build: log2 of its size base (8..11; 15/32 of 2^that records), 0 = none, < 0 = default (10, with a 2^11 key table)
};
static void ctx_release(grm_ctx *c);
static inline void ctx_retain(grm_ctx *c) { c->refs.fetch_add(1, std::memory_order_relaxed); }
// The contexts that exist.  The owner's pointer dangles once grm_destroy has been called AND the last handle is gone; the calls
// that take a bare grm_ctx* from the owner (grm_destroy, grm_ctx_live_handles, grm_last_error) look it up here first, so that a
// second grm_destroy or a late query is a no-op instead of a read of freed memory.
static std::mutex g_ctx_mu;
static std::vector<grm_ctx *> g_ctx_live;
static bool ctx_is_live(const grm_ctx *c)
{
    std::lock_guard<std::mutex> g(g_ctx_mu);
    return std::find(g_ctx_live.begin(), g_ctx_live.end(), c) != g_ctx_live.end();
}
// a handle's reference to its context (NULL for host-only objects)
struct CtxRef {
    grm_ctx *p = nullptr;
    CtxRef() = default;
    CtxRef(const CtxRef &) = delete;
    CtxRef &operator=(const CtxRef &) = delete;
    CtxRef &operator=(grm_ctx *c)
    {
        if (c) ctx_retain(c);
        if (p) ctx_release(p);
        p = c;
        return *this;
    }
    ~CtxRef() { if (p) ctx_release(p); }
    operator grm_ctx *() const { return p; }
    grm_ctx *operator->() const { return p; }
};
static inline int c_opt_wide_sort(const grm_ctx *c) { return c->opt_wide_sort; }
// 64-bit words of a k-mer, most significant first: 1 (k <= 32), 2 (<= 64), 3 (<= 96), 4 (<= 128)
static inline int words_of(int k) { return (k + 31) / 32; }
constexpr int GRM_MAX_K = 128;

static int fail(grm_ctx *c, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) {
        std::lock_guard<std::mutex> g(c->err_mu);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((ctx), e_ == hipErrorOutOfMemory ? GRM_ERR_OOM : GRM_ERR_HIP, "%s: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                               \
    } while (0)

// Device allocations are recycled inside the process: hipFree / hipMalloc of the multi-GB working set
// of a batch cost seconds (measured 4 s per 1000-genome chunk of a chunked run), far more than the
// kernels that use it.  Released blocks of >= 1 MiB are parked here and handed to the next request
// they fit (best fit, at most 2x larger); the pool is emptied by grm_destroy and whenever hipMalloc fails.
struct DevPool {
    struct Block { void *p; size_t bytes; int device; };
    std::mutex mu;
    std::vector<Block> blocks;
    void *take(size_t n, size_t *got)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> g(mu);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); i++)
            if (blocks[i].device == dev && blocks[i].bytes >= n && blocks[i].bytes <= 2 * n &&
                (best == blocks.size() || blocks[i].bytes < blocks[best].bytes)) best = i;
        if (best == blocks.size()) return nullptr;
        void *p = blocks[best].p;
        *got = blocks[best].bytes;
        blocks.erase(blocks.begin() + (long)best);
        return p;
    }
    bool give(void *p, size_t bytes)
    {
        if (bytes < ((size_t)1 << 20)) return false;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        std::lock_guard<std::mutex> g(mu);
        blocks.push_back({p, bytes, dev});
        return true;
    }
    void trim()
    {
        std::vector<Block> all;
        {
            std::lock_guard<std::mutex> g(mu);
            all.swap(blocks);
        }
        for (auto &b : all) (void)hipFree(b.p);
    }
};
static DevPool g_pool;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p && !g_pool.give(p, bytes)) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n)
    {
        release();
        if (n == 0) n = 16;
        size_t got = 0;
        if (void *q = g_pool.take(n, &got)) { p = q; bytes = got; return hipSuccess; }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {              // make room: give back everything that is parked, try once more
            (void)hipGetLastError();
            g_pool.trim();
            e = hipMalloc(&p, n);
        }
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    // grow-only: keeps the allocation when it is already large enough (hipMalloc / hipFree of
    // multi-GB buffers costs far more than the kernels that use them)
    hipError_t ensure(size_t n)
    {
        if (p && bytes >= n) return hipSuccess;
        return alloc(n + n / 16);
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct TimeScope {
    grm_ctx *c;
    int idx = -1;
    TimeScope(grm_ctx *ctx, const char *name, uint64_t units) : c(ctx)
    {
        if (!c->timing) return;
        TimingRec r;
        r.name = name;
        r.units = units;
        if (hipEventCreate(&r.e0) != hipSuccess) return;
        if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
        (void)hipEventRecord(r.e0, c->stream);
        c->recs.push_back(r);
        idx = (int)c->recs.size() - 1;
    }
    ~TimeScope()
    {
        if (idx >= 0) (void)hipEventRecord(c->recs[idx].e1, c->stream);
    }
};

extern "C" const char *grm_version(void) { return "grm-kmer-mi355x 0.1 (gfx950)"; }

extern "C" grm_ctx *grm_create(int device_ordinal, int n_streams)
{
    (void)n_streams;
    if (device_ordinal < 0) return nullptr;   // no CPU mode
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_ordinal >= n) return nullptr;
    if (hipSetDevice(device_ordinal) != hipSuccess) return nullptr;
    grm_ctx *c = new grm_ctx();
    c->device = device_ordinal;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        set_max_dynamic_lds() != hipSuccess) {
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_ctx_live.push_back(c);
    }
    return c;
}

static void ctx_release(grm_ctx *c)
{
    if (c->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    {
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_ctx_live.erase(std::remove(g_ctx_live.begin(), g_ctx_live.end(), c), g_ctx_live.end());
    }
    (void)hipSetDevice(c->device);
    for (auto &r : c->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    g_pool.trim();
    delete c;
}

extern "C" void grm_destroy(grm_ctx *c)
{
    if (!c || !ctx_is_live(c)) return;             // (a second grm_destroy after the last handle went: the context no longer exists)
    if (c->owner_gone.exchange(true)) return;      // (a second grm_destroy while handles keep the context alive)
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    ctx_release(c);
}
extern "C" int grm_ctx_live_handles(const grm_ctx *c) { return c && ctx_is_live(c) ? c->refs.load() - (c->owner_gone.load() ? 0 : 1) : 0; }

extern "C" const char *grm_last_error(grm_ctx *c)
{
    if (!c) return "no context (no HIP device?)";
    return ctx_is_live(c) ? c->err.c_str() : "the context no longer exists (grm_destroy, and its last handle was freed)";
}

extern "C" int grm_set_option(grm_ctx *c, const char *name, int value)
{
    if (!c || !name) return GRM_ERR_ARG;
    std::string n(name);
    if (n == "groups_per_thread") c->opt_groups_per_thread = value;
    else if (n == "bucket_bits") c->opt_bucket_bits = value;
    else if (n == "cap_log2") c->opt_cap_log2 = value;
    else if (n == "sub_bits") c->opt_sub_bits = value;
    else if (n == "no_slots") c->opt_no_slots = value;
    else if (n == "wide_sort") c->opt_wide_sort = value;
    else if (n == "upload_slab_kb") c->opt_upload_slab_kb = value;
    else if (n == "direct_permute") c->opt_direct_permute = value;
    else if (n == "dedup_wg") c->opt_dedup_wg = value;
    else if (n == "dedup_cap_shift") c->opt_dedup_cap_shift = value;
    else if (n == "dense_layout") c->opt_dense_layout = value;
    else if (n == "no_union") c->opt_no_union = value;
    else if (n == "records") c->opt_records = value;
    else if (n == "rec_bucket_shift") c->opt_rec_bucket_shift = value;
    else if (n == "rec_part_bits") c->opt_rec_part_bits = value;
    else if (n == "rec_keys") c->opt_rec_keys = value;
    else if (n == "rec_count") c->opt_rec_count = value;
    else if (n == "rec_count_cap") c->opt_rec_count_cap = value;
    else if (n == "rec_memo") c->opt_rec_memo = value;
    else if (n == "memo_stats") c->opt_memo_stats = value;
    else if (n == "parse_fused") c->opt_parse_fused = value;
    else if (n == "dict_sort_prim") c->opt_dict_sort_prim = value;
    else if (n == "rec_coarse") c->opt_rec_coarse = value;
    else if (n == "keys_in_flight") { c->opt_keys_in_flight = value; set_table_tuning(c->opt_keys_in_flight, c->opt_table_threads); }
    else if (n == "table_threads") { c->opt_table_threads = value; set_table_tuning(c->opt_keys_in_flight, c->opt_table_threads); }
    else return fail(c, GRM_ERR_ARG, "unknown option %s", name);
    return GRM_OK;
}

extern "C" int grm_timing_enable(grm_ctx *c, int on)
{
    if (!c) return GRM_ERR_ARG;
    c->timing = on != 0;
    return GRM_OK;
}
extern "C" int grm_timing_reset(grm_ctx *c)
{
    if (!c) return GRM_ERR_ARG;
    (void)hipStreamSynchronize(c->stream);
    for (auto &r : c->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    c->recs.clear();
    return GRM_OK;
}
extern "C" int grm_timing_count(grm_ctx *c) { return c ? (int)c->recs.size() : 0; }
extern "C" int grm_timing_get(grm_ctx *c, int i, char *name, size_t name_cap, double *ms, uint64_t *units)
{
    if (!c || i < 0 || i >= (int)c->recs.size()) return GRM_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->recs[i].e0, c->recs[i].e1));
    if (name && name_cap) snprintf(name, name_cap, "%s", c->recs[i].name.c_str());
    if (ms) *ms = f;
    if (units) *units = c->recs[i].units;
    return GRM_OK;
}

// --------------------------------------------------------------------------------------
// result objects
// --------------------------------------------------------------------------------------
// A counted set lives where it was produced: sets that come out of a device batch stay in HBM
// (grm_build_matrix consumes them there) and reach the host only when the caller asks for the
// arrays; sets built from host arrays stay on the host.
// std::vector whose resize() leaves new elements uninitialised: the host copies of results are
// filled by a device -> host copy right away, and value-initialising 1 GB first costs ~170 ms of
// memset + page faults on the calling thread.
template <typename T> struct default_init_alloc : std::allocator<T> {
    template <typename U> struct rebind { using other = default_init_alloc<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using raw_vector = std::vector<T, default_init_alloc<T>>;

// Device -> pageable host copy on the context's own stream (nothing in the library uses the NULL stream)
static hipError_t d2h(grm_ctx *c, void *dst, const void *src, size_t bytes)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e;
}

struct grm_kmer_set {
    CtxRef ctx;
    int k = 0, words = 1;
    uint64_t occurrences = 0;
    size_t n = 0;
    bool on_device = false, on_host = true;
    DevBuf d_kmers, d_counts;
    raw_vector<uint64_t> kmers;
    raw_vector<uint32_t> counts;
    bool to_host();
};

struct grm_matrix {
    CtxRef ctx;
    int k = 0, words = 1, n_genomes = 0;
    size_t n_rows = 0, n_kmers = 0;
    DevBuf d_kmers, d_data, d_errors;     // d_errors: per-column error counts of the last grm_matrix_risk_errors
    raw_vector<uint64_t> h_kmers, h_data;
    bool have_kmers = false, have_data = false;
};

bool grm_kmer_set::to_host()
{
    if (on_host) return true;
    kmers.resize(n * (size_t)words);
    counts.resize(n);
    if (n) {
        if (hipSetDevice(ctx->device) != hipSuccess) return false;
        if (d2h(ctx, kmers.data(), d_kmers.p, n * 8 * (size_t)words) != hipSuccess ||
            d2h(ctx, counts.data(), d_counts.p, n * 4) != hipSuccess) {
            (void)fail(ctx, GRM_ERR_HIP, "k-mer set download failed");
            kmers.clear(); counts.clear();
            return false;
        }
    }
    on_host = true;
    return true;
}

extern "C" size_t grm_kmer_set_size(const grm_kmer_set *s) { return s ? s->n : 0; }
extern "C" int grm_kmer_set_k(const grm_kmer_set *s) { return s ? s->k : 0; }
extern "C" int grm_kmer_set_words(const grm_kmer_set *s) { return s ? s->words : 0; }
extern "C" uint64_t grm_kmer_set_occurrences(const grm_kmer_set *s) { return s ? s->occurrences : 0; }
extern "C" const uint64_t *grm_kmer_set_kmers(grm_kmer_set *s) { return s && s->to_host() ? s->kmers.data() : nullptr; }
extern "C" const uint32_t *grm_kmer_set_counts(grm_kmer_set *s) { return s && s->to_host() ? s->counts.data() : nullptr; }
extern "C" void grm_kmer_set_free(grm_kmer_set *s) { delete s; }

extern "C" int grm_kmer_set_from_host(grm_ctx *c, const uint64_t *kmers, const uint32_t *counts, size_t n, int k,
                                      grm_kmer_set **out)
{
    if (!c || !out || (n && !kmers)) return fail(c, GRM_ERR_ARG, "grm_kmer_set_from_host: bad argument");
    if (k < 1 || k > GRM_MAX_K) return fail(c, GRM_ERR_ARG, "k=%d unsupported (1..128)", k);
    grm_kmer_set *s = new grm_kmer_set();
    s->k = k;
    s->ctx = c;
    s->n = n;
    s->words = words_of(k);                        // ceil(k / 32) words per k-mer, most significant first
    s->kmers.assign(kmers, kmers + n * (size_t)s->words);
    if (counts) s->counts.assign(counts, counts + n);
    else s->counts.assign(n, 1u);
    *out = s;
    return GRM_OK;
}

extern "C" size_t grm_matrix_n_kmers(const grm_matrix *m) { return m ? m->n_kmers : 0; }
extern "C" size_t grm_matrix_n_rows(const grm_matrix *m) { return m ? m->n_rows : 0; }
extern "C" int grm_matrix_n_genomes(const grm_matrix *m) { return m ? m->n_genomes : 0; }
extern "C" int grm_matrix_k(const grm_matrix *m) { return m ? m->k : 0; }
extern "C" int grm_matrix_words(const grm_matrix *m) { return m ? m->words : 0; }
extern "C" const void *grm_matrix_dev_kmers(const grm_matrix *m) { return m ? m->d_kmers.p : nullptr; }
extern "C" const void *grm_matrix_dev_data(const grm_matrix *m) { return m ? m->d_data.p : nullptr; }

extern "C" const uint64_t *grm_matrix_kmers(grm_matrix *m)
{
    if (!m) return nullptr;
    if (!m->have_kmers) {
        m->h_kmers.resize(m->n_kmers * m->words + 1);
        if (m->n_kmers) {
            (void)hipSetDevice(m->ctx->device);
            if (d2h(m->ctx, m->h_kmers.data(), m->d_kmers.p, m->n_kmers * m->words * 8) != hipSuccess) {
                fail(m->ctx, GRM_ERR_HIP, "D2H of dictionary failed");
                return nullptr;
            }
        }
        m->have_kmers = true;
    }
    return m->h_kmers.data();
}
// Row-wise download for the HDF5 writer (grm_h5.cpp), which deflates the rows that have arrived while
// the next ones are on their way: begin returns the host buffer (nothing copied yet unless the data is
// already on the host), rows copies word-rows [r0, r1), end marks the host copy complete.
extern "C" uint64_t *grm_internal_matrix_download_begin(grm_matrix *m, int *already)
{
    if (!m) return nullptr;
    *already = m->have_data ? 1 : 0;
    if (!m->have_data) m->h_data.resize(m->n_kmers * m->n_rows + 1);
    return m->h_data.data();
}
extern "C" int grm_internal_matrix_download_rows(grm_matrix *m, size_t r0, size_t r1)
{
    if (!m || m->have_data || r1 <= r0 || !m->n_kmers) return GRM_OK;
    if (hipSetDevice(m->ctx->device) != hipSuccess ||
        d2h(m->ctx, m->h_data.data() + r0 * m->n_kmers, m->d_data.as<uint64_t>() + r0 * m->n_kmers, (r1 - r0) * m->n_kmers * 8) != hipSuccess)
        return fail(m->ctx, GRM_ERR_HIP, "D2H of matrix rows failed");
    return GRM_OK;
}
extern "C" void grm_internal_matrix_download_end(grm_matrix *m) { if (m) m->have_data = true; }

extern "C" const uint64_t *grm_matrix_data(grm_matrix *m)
{
    if (!m) return nullptr;
    if (!m->have_data) {
        m->h_data.resize(m->n_kmers * m->n_rows + 1);
        if (m->n_kmers && m->n_rows) {
            (void)hipSetDevice(m->ctx->device);
            if (d2h(m->ctx, m->h_data.data(), m->d_data.p, m->n_kmers * m->n_rows * 8) != hipSuccess) {
                fail(m->ctx, GRM_ERR_HIP, "D2H of matrix failed");
                return nullptr;
            }
        }
        m->have_data = true;
    }
    return m->h_data.data();
}
extern "C" int grm_matrix_column_counts(grm_matrix *m, uint32_t *out)
{
    if (!m || !out) return GRM_ERR_ARG;
    grm_ctx *c = m->ctx;
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!m->n_kmers) return GRM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf d;
    HIPCHK(c, d.alloc(m->n_kmers * 4));
    {
        TimeScope t(c, "column_popcount", m->n_kmers * m->n_rows);
        launch_column_popcount(c->stream, m->d_data.as<uint64_t>(), m->n_rows, m->n_kmers, nullptr, d.as<uint32_t>());
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, d.p, m->n_kmers * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GRM_OK;
}
// KmerRuleClassifications.sum_rows (learning/common/rules.py:201-267 + popcount.pyx:76-95):
// out[c] = sum over word-rows of popcount(matrix[r][c] & row_mask[r]).
extern "C" int grm_matrix_sum_rows(grm_matrix *m, const uint64_t *row_mask, uint32_t *out)
{
    if (!m || !out || !row_mask) return GRM_ERR_ARG;
    grm_ctx *c = m->ctx;
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!m->n_kmers) return GRM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf d, dm;
    HIPCHK(c, d.alloc(m->n_kmers * 4));
    HIPCHK(c, dm.alloc((m->n_rows + 1) * 8));
    HIPCHK(c, hipMemcpyAsync(dm.p, row_mask, m->n_rows * 8, hipMemcpyHostToDevice, c->stream));
    {
        TimeScope t(c, "sum_rows", m->n_kmers * m->n_rows);
        launch_column_popcount(c->stream, m->d_data.as<uint64_t>(), m->n_rows, m->n_kmers, dm.as<uint64_t>(), d.as<uint32_t>());
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, d.p, m->n_kmers * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GRM_OK;
}

// a host-only matrix (grm_matrix_from_host: rows read back from a .kover file) placed in HBM, so that the
// learner-side kernels (sum_rows, risk tables) run on it
extern "C" int grm_matrix_to_device(grm_ctx *c, grm_matrix *m)
{
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!m) return fail(c, GRM_ERR_ARG, "grm_matrix_to_device: NULL matrix");
    if (m->ctx) return m->ctx == c ? GRM_OK : fail(c, GRM_ERR_ARG, "grm_matrix_to_device: matrix of another context");
    if (!m->have_data || !m->have_kmers) return fail(c, GRM_ERR_STATE, "grm_matrix_to_device: no host data");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t cells = m->n_rows * m->n_kmers, kw = m->n_kmers * (size_t)m->words;
    HIPCHK(c, m->d_data.alloc(cells * 8));
    HIPCHK(c, m->d_kmers.alloc((kw + 2) * 8));
    if (cells) HIPCHK(c, hipMemcpyAsync(m->d_data.p, m->h_data.data(), cells * 8, hipMemcpyHostToDevice, c->stream));
    if (kw) HIPCHK(c, hipMemcpyAsync(m->d_kmers.p, m->h_kmers.data(), kw * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    m->ctx = c;
    return GRM_OK;
}

// dataset/split.py:171-188, device part (see column_errors_kernel).  hist_out: n_train + 1 host counters.
extern "C" int grm_matrix_risk_errors(grm_matrix *m, const uint64_t *pos_mask, const uint64_t *neg_mask, uint32_t n_pos, uint32_t n_train,
                                      uint64_t *hist_out)
{
    if (!m || !pos_mask || !neg_mask || !hist_out) return GRM_ERR_ARG;
    grm_ctx *c = m->ctx;
    if (!c) return GRM_ERR_NO_DEVICE;
    if (n_pos > n_train) return fail(c, GRM_ERR_ARG, "grm_matrix_risk_errors: n_pos > n_train");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    DevBuf dm, dh;
    HIPCHK(c, dm.alloc((2 * m->n_rows + 2) * 8));
    HIPCHK(c, dh.alloc(((size_t)n_train + 1) * 8));
    HIPCHK(c, m->d_errors.ensure((m->n_kmers + 1) * 4));
    HIPCHK(c, hipMemcpyAsync(dm.p, pos_mask, m->n_rows * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dm.as<uint64_t>() + m->n_rows, neg_mask, m->n_rows * 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(dh.p, 0, ((size_t)n_train + 1) * 8, s));
    {
        TimeScope t(c, "risk_errors", m->n_kmers * m->n_rows);
        launch_column_errors(s, m->d_data.as<uint64_t>(), m->n_rows, m->n_kmers, dm.as<uint64_t>(), dm.as<uint64_t>() + m->n_rows, n_pos, n_train,
                             m->d_errors.as<uint32_t>(), dh.as<unsigned long long>());
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hist_out, dh.p, ((size_t)n_train + 1) * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return GRM_OK;
}

// by_kmer[c] = lut_presence[errors[c]], by_anti[c] = lut_absence[errors[c]] (errors of the last grm_matrix_risk_errors);
// the two look-up tables hold n_lut = n_train + 1 entries; outputs: n_kmers host uint32 each
extern "C" int grm_matrix_risk_index(grm_matrix *m, const uint32_t *lut_presence, const uint32_t *lut_absence, uint32_t n_lut,
                                     uint32_t *by_kmer_out, uint32_t *by_anti_out)
{
    if (!m || !lut_presence || !lut_absence || !by_kmer_out || !by_anti_out) return GRM_ERR_ARG;
    grm_ctx *c = m->ctx;
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!m->d_errors.p) return fail(c, GRM_ERR_STATE, "grm_matrix_risk_index before grm_matrix_risk_errors");
    if (!m->n_kmers) return GRM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    DevBuf dl, dout;
    HIPCHK(c, dl.alloc(((size_t)2 * n_lut + 2) * 4));
    HIPCHK(c, dout.alloc((size_t)2 * m->n_kmers * 4));
    HIPCHK(c, hipMemcpyAsync(dl.p, lut_presence, (size_t)n_lut * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(dl.as<uint32_t>() + n_lut, lut_absence, (size_t)n_lut * 4, hipMemcpyHostToDevice, s));
    launch_risk_index(s, m->d_errors.as<uint32_t>(), m->n_kmers, dl.as<uint32_t>(), dl.as<uint32_t>() + n_lut, dout.as<uint32_t>(),
                      dout.as<uint32_t>() + m->n_kmers);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(by_kmer_out, dout.p, m->n_kmers * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(by_anti_out, dout.as<uint32_t>() + m->n_kmers, m->n_kmers * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return GRM_OK;
}

// The HDF5 chunks of kmer_matrix (what = 0: chunk_elems columns of one word-row each, row-major) or of kmer_sequences
// (what = 1: chunk_elems strings each) as zlib streams made on the device (grm_deflate.hip).  The chunks go through the
// device in slabs (a few GB of scratch at most; `max_slab` chunks when the caller wants them sooner than that).
//   sink == nullptr: host result (malloc) = ALL streams at starts[i] (multiples of 16), lens[i] bytes each;
//   sink != nullptr: every slab is handed over as soon as it has arrived on the host -- sink(user, first chunk, n, block, starts
//     relative to the block, lens); the block (malloc) is the sink's to free.  The HDF5 writer appends slab k while slab k + 1
//     is encoded and copied (grm_h5.cpp).
// Two pinned buffers between the encoder and a sink that consumes on another thread: a piece (whole chunks, <= cap bytes) is copied
// into a free buffer at PCIe speed (a pageable copy runs at a fifth of it and faults its pages in first), the sink gives the buffer
// back with grm_internal_deflate_release when it has written the piece.
struct DeflateRing {
    unsigned char *buf[2] = {nullptr, nullptr};
    size_t cap = 0;
    bool busy[2] = {false, false};
    std::mutex mu;
    std::condition_variable cv;
    int acquire()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !busy[0] || !busy[1]; });
        const int i = busy[0] ? 1 : 0;
        busy[i] = true;
        return i;
    }
    void release(int i)
    {
        std::lock_guard<std::mutex> lk(mu);
        busy[i] = false;
        cv.notify_all();
    }
    void wait_idle()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !busy[0] && !busy[1]; });
    }
    ~DeflateRing()
    {
        for (auto *b : buf)
            if (b) (void)hipHostFree(b);
    }
};
extern "C" void grm_internal_deflate_release(void *ring, int slot)
{
    if (ring && slot >= 0) static_cast<DeflateRing *>(ring)->release(slot);
}
// sink: block holds the piece's streams at starts[i] (relative), lens[i]; slot >= 0: a buffer of `ring`, to be given back with
// grm_internal_deflate_release(ring, slot) -- also when the sink fails or has lost interest; slot < 0: a malloc block, the sink's to free
typedef int (*grm_deflate_sink)(void *user, uint64_t first_chunk, uint64_t n, unsigned char *block, const uint64_t *starts, const uint32_t *lens, void *ring,
                                int slot);
static int matrix_deflate_device(grm_matrix *m, int what, uint64_t chunk_elems, unsigned char **streams, uint64_t **starts_out, uint32_t **lens_out,
                                 uint64_t *n_chunks_out, grm_deflate_sink sink = nullptr, void *sink_user = nullptr, uint64_t max_slab = 0)
{
    grm_ctx *c = m->ctx;
    if (!c) return GRM_ERR_NO_DEVICE;
    if (chunk_elems == 0 || chunk_elems > 0x7fffffffu) return fail(c, GRM_ERR_ARG, "deflate: chunk of %llu elements", (unsigned long long)chunk_elems);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint64_t U = m->n_kmers, R = m->n_rows;
    const uint64_t per_row = (U + chunk_elems - 1) / chunk_elems;
    const uint64_t n_chunks = what == 0 ? per_row * R : per_row;
    const uint64_t raw = chunk_elems * (what == 0 ? 8ull : (uint64_t)m->k);
    if (raw >= ((uint64_t)1 << 32) - 4096) return fail(c, GRM_ERR_ARG, "deflate: a chunk of %llu bytes is beyond the encoder's 4 GiB", (unsigned long long)raw);
    if (streams) { *streams = nullptr; *starts_out = nullptr; *lens_out = nullptr; }
    *n_chunks_out = n_chunks;
    if (!n_chunks) return GRM_OK;
    const uint64_t cap = deflate_chunk_cap(raw);
    const uint64_t per_chunk = cap + (what == 0 ? 2 * chunk_elems : 0);
    uint64_t slab = std::max<uint64_t>(1, std::min<uint64_t>(n_chunks, ((uint64_t)4 << 30) / per_chunk));
    if (max_slab) slab = std::min(slab, max_slab);
    DevBuf d_out, d_tok, d_sizes, d_off, d_packed;
    HIPCHK(c, d_out.alloc(slab * cap));
    if (what == 0) HIPCHK(c, d_tok.alloc(slab * chunk_elems * 2));
    HIPCHK(c, d_sizes.alloc(slab * 4));
    HIPCHK(c, d_off.alloc((slab + 1) * 8));
    uint64_t *starts = static_cast<uint64_t *>(malloc((sink ? slab : n_chunks) * 8));
    uint32_t *lens = static_cast<uint32_t *>(malloc((sink ? slab : n_chunks) * 4));
    unsigned char *host = nullptr;
    uint64_t host_used = 0, host_cap = 0;
    DeflateRing ring;
    // (sink mode: the ring's buffers die with this frame -- whatever the sink still holds must have come back first)
    auto bail = [&](int rc) { if (sink) ring.wait_idle(); free(starts); free(lens); free(host); return rc; };
    if (!starts || !lens) return bail(fail(c, GRM_ERR_OOM, "deflate: host allocation failed"));
    std::vector<uint64_t> off(slab + 2), rel(slab + 1);
    if (sink) {
        ring.cap = (size_t)64 << 20;
        for (auto *&bp : ring.buf) {
            const hipError_t e = hipHostMalloc((void **)&bp, ring.cap, hipHostMallocDefault);
            if (e != hipSuccess) { bp = nullptr; return bail(fail(c, GRM_ERR_HIP, "deflate: pinned buffers: %s", hipGetErrorString(e))); }
        }
    }
    const bool trace = getenv("GRM_TRACE") && atoi(getenv("GRM_TRACE")) > 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    for (uint64_t c0 = 0; c0 < n_chunks; c0 += slab) {
        const uint32_t n = (uint32_t)std::min<uint64_t>(slab, n_chunks - c0);
        uint32_t *ln = sink ? lens : lens + c0;
        hipError_t e;
        {
            TimeScope t(c, what == 0 ? "deflate_rows" : "deflate_kmer_strings", (uint64_t)n * raw);
            e = what == 0 ? launch_deflate_rows(s, m->d_data.as<uint64_t>(), U, (uint32_t)chunk_elems, (uint32_t)per_row, c0, n, d_tok.as<uint16_t>(),
                                                d_out.as<uint8_t>(), cap, d_sizes.as<uint32_t>())
                          : launch_deflate_kmer_strings(s, m->d_kmers.as<uint64_t>(), U, m->words, m->k, (uint32_t)chunk_elems, c0, n, d_out.as<uint8_t>(), cap,
                                                        d_sizes.as<uint32_t>());
        }
        if (e == hipSuccess) e = hipMemcpyAsync(ln, d_sizes.p, (size_t)n * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "deflate on the device: %s", hipGetErrorString(e)));
        if (trace) fprintf(stderr, "[deflate %s] %u chunks encoded               %8.1f ms\n", what == 0 ? "rows" : "strings", n, (now() - t_last) * 1e3), t_last = now();
        uint64_t total = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (ln[i] == 0 || ln[i] > cap) return bail(fail(c, GRM_ERR_STATE, "deflate: chunk %llu came back with %u bytes", (unsigned long long)(c0 + i), ln[i]));
            off[i] = total;
            total += ((uint64_t)ln[i] + 15) & ~15ull;
        }
        if (!sink && host_used + total > host_cap) {
            // first slab: the whole result at this slab's ratio; later: what is missing
            host_cap = host_used + std::max<uint64_t>(total, c0 == 0 ? total * ((n_chunks + n - 1) / n) + (total >> 4) : total * 2);
            unsigned char *p = static_cast<unsigned char *>(realloc(host, host_cap));
            if (!p) return bail(fail(c, GRM_ERR_OOM, "deflate: host allocation of %llu bytes failed", (unsigned long long)host_cap));
            host = p;
        }
        if ((e = d_packed.ensure(total)) != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "deflate: %s", hipGetErrorString(e)));
        e = hipMemcpyAsync(d_off.p, off.data(), (size_t)n * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = launch_deflate_compact(s, d_out.as<uint8_t>(), cap, d_sizes.as<uint32_t>(), n, d_off.as<uint64_t>(), d_packed.as<uint8_t>());
        if (sink) {
            // pieces of whole chunks through the pinned ring (a chunk larger than a buffer: a malloc block of its own)
            if (e != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "deflate: gathering the streams: %s", hipGetErrorString(e)));
            off[n] = total;
            for (uint32_t i0 = 0; i0 < n;) {
                uint32_t i1 = i0 + 1;
                while (i1 < n && off[i1 + 1] - off[i0] <= ring.cap) i1++;
                const uint64_t bytes = off[i1] - off[i0];
                int slot = -1;
                unsigned char *block;
                if (bytes <= ring.cap) { slot = ring.acquire(); block = ring.buf[slot]; }
                else if (!(block = static_cast<unsigned char *>(malloc(bytes)))) return bail(fail(c, GRM_ERR_OOM, "deflate: host allocation failed"));
                e = hipMemcpyAsync(block, d_packed.as<uint8_t>() + off[i0], bytes, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess) e = hipStreamSynchronize(s);
                if (e != hipSuccess) {
                    if (slot >= 0) ring.release(slot); else free(block);
                    return bail(fail(c, GRM_ERR_HIP, "deflate: copying the streams: %s", hipGetErrorString(e)));
                }
                for (uint32_t i = i0; i < i1; i++) rel[i - i0] = off[i] - off[i0];
                const int rc = sink(sink_user, c0 + i0, i1 - i0, block, rel.data(), ln + i0, &ring, slot);
                if (rc != GRM_OK) return bail(rc);
                i0 = i1;
            }
            if (trace) fprintf(stderr, "[deflate %s] %.1f MB handed to the writer                %6.1f ms\n", what == 0 ? "rows" : "strings", total / 1e6, (now() - t_last) * 1e3), t_last = now();
            continue;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(host + host_used, d_packed.p, total, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "deflate: gathering the streams: %s", hipGetErrorString(e)));
        if (trace) fprintf(stderr, "[deflate %s] %.1f MB gathered and copied to the host %6.1f ms\n", what == 0 ? "rows" : "strings", total / 1e6, (now() - t_last) * 1e3), t_last = now();
        for (uint32_t i = 0; i < n; i++) starts[c0 + i] = host_used + off[i];
        host_used += total;
    }
    if (sink) ring.wait_idle();                 // (the ring's buffers die with this frame: the sink must be through with them)
    if (sink) { free(starts); free(lens); return GRM_OK; }
    *streams = host; *starts_out = starts; *lens_out = lens;
    return GRM_OK;
}
// the slab-wise form for the HDF5 writer (grm_h5.cpp): what = 0 rows / 1 k-mer strings
extern "C" int grm_internal_deflate_stream(grm_matrix *m, int what, uint64_t chunk_elems, uint64_t max_slab, grm_deflate_sink sink, void *user, uint64_t *n_chunks)
{
    if (!m || !sink || !n_chunks) return GRM_ERR_ARG;
    const uint64_t ce = m->n_kmers ? std::min<uint64_t>(m->n_kmers, chunk_elems) : 1;
    return matrix_deflate_device(m, what, ce, nullptr, nullptr, nullptr, n_chunks, sink, user, max_slab);
}
extern "C" int grm_matrix_deflate_rows(grm_matrix *m, int chunk_cols, unsigned char **streams, uint64_t **starts, uint32_t **lens, uint64_t *n_chunks)
{
    if (!m || !streams || !starts || !lens || !n_chunks || chunk_cols <= 0) return GRM_ERR_ARG;
    if (!m->ctx || !m->d_data.p) return m->ctx ? fail(m->ctx, GRM_ERR_STATE, "grm_matrix_deflate_rows: the matrix is not on the device") : GRM_ERR_NO_DEVICE;
    // chunk dimensions as the writer lays them out: (1, min(n_kmers, chunk_cols)) (dataset/create.py:160,230)
    const uint64_t cw = m->n_kmers ? std::min<uint64_t>(m->n_kmers, (uint64_t)chunk_cols) : 1;
    return matrix_deflate_device(m, 0, cw, streams, starts, lens, n_chunks);
}
extern "C" int grm_matrix_deflate_kmer_strings(grm_matrix *m, int chunk_elems, unsigned char **streams, uint64_t **starts, uint32_t **lens, uint64_t *n_chunks)
{
    if (!m || !streams || !starts || !lens || !n_chunks || chunk_elems <= 0) return GRM_ERR_ARG;
    if (!m->ctx || !m->d_kmers.p) return m->ctx ? fail(m->ctx, GRM_ERR_STATE, "grm_matrix_deflate_kmer_strings: the dictionary is not on the device") : GRM_ERR_NO_DEVICE;
    const uint64_t ce = m->n_kmers ? std::min<uint64_t>(m->n_kmers, (uint64_t)chunk_elems) : 1;
    return matrix_deflate_device(m, 1, ce, streams, starts, lens, n_chunks);
}
extern "C" void grm_host_free(void *p) { free(p); }
extern "C" int grm_internal_matrix_on_device(const grm_matrix *m) { return m && m->ctx && m->d_data.p && m->d_kmers.p; }

extern "C" void grm_matrix_free(grm_matrix *m)
{
    if (!m) return;
    if (m->ctx) (void)hipSetDevice(m->ctx->device);
    delete m;
}

// host-only matrix (no device, no ctx): lets the writers run on rows gathered from several
// ranks, and lets the CPU test-suite exercise the TSV / HDF5 writers without a GPU.
extern "C" int grm_matrix_from_host(const uint64_t *kmers, const uint64_t *data, size_t n_kmers, int n_genomes, int k,
                                    grm_matrix **out)
{
    if (!out || n_genomes < 0 || k < 1 || k > GRM_MAX_K || (n_kmers && (!kmers || (n_genomes && !data)))) return GRM_ERR_ARG;
    grm_matrix *m = new grm_matrix();
    m->k = k;
    m->words = words_of(k);                         // kmers: n_kmers * words, most significant word first
    m->n_genomes = n_genomes;
    m->n_rows = ((size_t)n_genomes + 63) / 64;
    m->n_kmers = n_kmers;
    m->h_kmers.assign(kmers, kmers + n_kmers * (size_t)m->words);
    m->h_kmers.push_back(0);
    m->h_data.assign(data, data + n_kmers * m->n_rows);
    m->h_data.push_back(0);
    m->have_kmers = m->have_data = true;
    *out = m;
    return GRM_OK;
}

static std::string g_hostonly_err;
extern "C" int grm_internal_fail(grm_matrix *m, int code, const char *msg)
{
    if (m && m->ctx) m->ctx->err = msg ? msg : "";
    else g_hostonly_err = msg ? msg : "";
    return code;
}
extern "C" const char *grm_matrix_last_error(const grm_matrix *m)
{
    return (m && m->ctx) ? m->ctx->err.c_str() : g_hostonly_err.c_str();
}

// --------------------------------------------------------------------------------------
// batch
// --------------------------------------------------------------------------------------
struct HostFile {
    int genome;
    bool fastq = false;           // 4-line FASTQ (first non-blank byte '@'), else FASTA
    std::vector<uint8_t> bytes;   // inflated image (memory buffers, gzip members, pipes)
    std::string path;             // plain regular file: read straight into the pinned slab at upload
    uint64_t size = 0;            // image bytes (bytes.size() or the file size)
    bool deferred() const { return !path.empty(); }
};

// gzip / zlib streams (magic 1f 8b) are inflated on the host: Kover's from-reads accepts
// .fastq.gz (dataset/create.py:402).  Concatenated members are handled.
static bool inflate_gzip(const uint8_t *src, size_t len, std::vector<uint8_t> &out)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) return false;
    zs.next_in = const_cast<Bytef *>(src);
    zs.avail_in = (uInt)0;
    size_t in_pos = 0;
    std::vector<uint8_t> buf(1 << 20);
    int rc = Z_OK;
    for (;;) {
        if (zs.avail_in == 0 && in_pos < len) {
            const size_t chunk = std::min<size_t>(len - in_pos, 1u << 30);
            zs.next_in = const_cast<Bytef *>(src + in_pos);
            zs.avail_in = (uInt)chunk;
            in_pos += chunk;
        }
        zs.next_out = buf.data();
        zs.avail_out = (uInt)buf.size();
        rc = inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) break;
        out.insert(out.end(), buf.data(), buf.data() + (buf.size() - zs.avail_out));
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0 && in_pos >= len) break;
            if (inflateReset(&zs) != Z_OK) { rc = Z_DATA_ERROR; break; }      // next gzip member
        } else if (zs.avail_in == 0 && in_pos >= len && zs.avail_out != 0) {
            rc = Z_DATA_ERROR;      // truncated stream
            break;
        }
    }
    inflateEnd(&zs);
    return rc == Z_STREAM_END;
}

struct WideSorted;
static void wide_free(WideSorted *w);
struct MultiSorted;
static void multi_free(MultiSorted *w);
struct WideHash;
static void wide_hash_free(WideHash *w);
// staged (multi-GPU) entry points of the two-word hash pipeline, defined next to it
static int wide_stage_local(grm_batch *b, int k);
static int wide_stage_n_local(grm_batch *b, uint64_t *n_local);
static int wide_stage_export(grm_batch *b, void *dev_keys_out, void *dev_flags_out);
static int wide_stage_export_ordered(grm_batch *b, uint8_t *rec, uint64_t flags_off, uint64_t boff_off);
static int wide_stage_global(grm_batch *b, const void *dev_keys, const void *dev_flags, uint64_t n, int filter_singleton, uint64_t *n_kmers);
static int wide_stage_fill(grm_batch *b, grm_matrix **out);
// the same calls over the SORT path (grm_multi.hip): k > 64, and 33 <= k <= 64 with abundance-min > 1 -- what grm_batch_run does for
// those in one go (multi_matrix / wide_matrix), cut into the stages the multi-GPU and chunked routes need
struct SortedStage;
static void sorted_stage_free(SortedStage *);
static int sorted_stage_local(grm_batch *b, int k, uint32_t abundance_min);
static int sorted_stage_export(grm_batch *b, void *dev_keys_out, void *dev_flags_out);
static int sorted_stage_global(grm_batch *b, const void *dev_keys, const void *dev_flags, uint64_t n, int filter_singleton, uint64_t *n_kmers);
static int sorted_stage_fill(grm_batch *b, grm_matrix **out);

struct grm_batch {
    CtxRef ctx;
    WideSorted *wide = nullptr;      // two-word (k > 32) sort path buffers, created on first use
    MultiSorted *multi = nullptr;    // three- / four-word (k > 64) sort path buffers
    WideHash *whash = nullptr;       // two-word hash-partition path buffers
    SortedStage *sorted = nullptr;   // staged calls over the sort path: the batch's own matrix and the global dictionary
    bool sorted_stage = false;       // the current partition went that way
    ~grm_batch() { wide_free(wide); multi_free(multi); wide_hash_free(whash); sorted_stage_free(sorted); }
    int n_genomes = 0;
    std::vector<HostFile> files;
    bool uploaded = false, partitioned = false, have_local = false, have_global = false;
    // inputs
    uint64_t input_bytes = 0;
    uint64_t raw_bytes = 0;   // tile-aligned image size (without front pad)
    uint32_t n_tiles = 0;
    DevBuf d_raw_alloc;       // front pad + image
    DevBuf d_genome_tile_off; // u32[n_genomes+1]
    DevBuf d_tile_meta;       // u8[n_tiles]: TILE_META_FIRST | TILE_META_FASTQ
    // parse products
    DevBuf d_sums, d_tile_off, d_tile_state, d_sym2, d_inv, d_genome_sym_off, d_scan_scratch, d_chunk_pre;
    uint64_t total_syms = 0;
    std::vector<uint64_t> h_genome_sym_off;
    // partition products
    int k = 0, bb = 0;
    uint32_t abundance_min = 1;
    uint64_t total_keys = 0;       // k-mer occurrences
    DevBuf d_counts, d_off, d_cursor, d_cursor1, d_counts1, d_off1, d_keys, d_keys1, d_len, d_kcnt, d_koff, d_klen;
    bool deduped = false;
    uint64_t seg_stride = 0;       // 0: dense layout (d_off); else slack layout: segment i at i * seg_stride, length d_len[i]
    // record form of the partition (grm_superkmer.hip): minimizer buckets; d_recs holds the level-1 records, the key
    // segments are given by d_off AND d_len (regions leave gaps).  rec_failed: a region overflowed once (repeat-rich
    // input) or a later stage could not use the layout -- the batch stays on the key form from then on.
    // rec_dict: the segments (d_off / d_len) are still segments of RECORDS in d_recs2 -- dict_build decodes them; the keys
    // are expanded from d_recs only if a later stage asks for them (batch_expand_keys).
    DevBuf d_recs, d_recs2;
    bool rec_mode = false, rec_failed = false, rec_dict = false;
    int rec_count_pbits = 0;           // counting over records: parts per genome (log2) level 1 cut -- their k-mer counts add up to the genome's
    uint32_t rec_rstride = 0;
    uint64_t rec_kstride = 0, rec_regions = 0;
    int rec_b1 = 0;                // coarse bits of the record regions
    uint32_t rec_need = 0;         // what the dictionary launch that gave the record form up estimated for its fullest bucket ...
    int rec_need_bb = 0;           // ... of 2^rec_need_bb
    int rec_bb_hint = -1, rec_bb_hint_k = 0;   // bucket bits a dictionary of this batch needed last time (more buckets instead of sub-buckets)
    int rec_part_bits = 0;         // genomes are cut into 2^rec_part_bits parts (segment index: virtual genome * 2^bb + bucket)
    int rec_memo_log2 = 0;         // slots of dict_build's record memo (log2), 0 = none
    uint64_t memo_stats[4] = {0, 0, 0, 0};   // of the last dictionary launch with the "memo_stats" option: see grm_batch_memo_stats
    bool slack_failed = false;     // a slack-layout partition of this batch overflowed: dense layout from then on
    DevBuf d_marks;                // one bit per segment: left to the workgroup form of the dedup
    // dictionary
    int sb_dict = 0, sb_fill = 0;
    uint32_t cap_log2 = 12;
    // sub-bucket count that the last successful dict_build of this batch needed (for the same k, abundance-min
    // and bucket geometry): the next run starts there instead of walking up from 0 again
    int sb_hint = -1, sb_hint_k = 0, sb_hint_bb = -1;
    uint32_t sb_hint_amin = 0;
    int dict_launches = 0;         // dict_build launches of the last grm_batch_local_dict (tests)
    DevBuf d_local_keys, d_local_flags;     // entries of the local dictionary: workgroup wg owns [wg_base, wg_base + wg_cnt)
    uint64_t n_local = 0;
    DevBuf d_dict;                 // sorted, filtered global dictionary (U)
    uint64_t n_dict = 0;
    DevBuf d_dkeys, d_dcol, d_seg_start;   // bucketised view for matrix_fill (probing form)
    // fused form: dict_build also leaves the presence words of every entry (by workgroup, word-row, entry id);
    // the fill is then a permutation into column order that needs neither keys nor probes
    DevBuf d_wg_base, d_wg_cnt, d_matrix_s, d_birth, d_entry_col, d_ctrl, d_prefix, d_entry_major;
    bool own_dict = false;         // the global dictionary is this batch's own local one: every column has a local entry
    bool entry_cols_ready = false; // d_entry_col was filled by the sort itself (own dictionary)
    bool have_bits = false, fill_by_bits = false;
    int filter_singleton = 0;
    // scratch that survives between steps (grow-only)
    DevBuf t_flag, t_ord_off;
    DevBuf t_track, t_union_col;   // rank union with tracking: union entry of every position of the exported list; column of every union entry
    bool exported_ordered = false; // t_ord_off describes the list grm_batch_export_dict_ordered wrote for the current local dictionary
    DevBuf t_u_off, t_u_len, t_u_foff, t_u_keys, t_u_flags, t_u_wg_base, t_u_wg_cnt;     // union of gathered rank dictionaries
    int sb_union_hint = 0;
    DevBuf t_sk, t_sf, t_keep, t_pos, t_tmp, t_bid, t_col, t_bid_sorted;
    DevBuf t_set_off, t_set_src, t_set_len, t_set_k, t_set_c, t_set_tmp;     // grm_batch_genome_set
};

extern "C" int grm_batch_create(grm_ctx *c, int n_genomes, grm_batch **out)
{
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!out || n_genomes < 0) return fail(c, GRM_ERR_ARG, "grm_batch_create: bad argument");
    grm_batch *b = new grm_batch();
    b->ctx = c;
    b->n_genomes = n_genomes;
    *out = b;
    return GRM_OK;
}

extern "C" void grm_batch_free(grm_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    delete b;
}

extern "C" int grm_batch_add(grm_batch *b, int genome_index, const void *buf, size_t len)
{
    if (!b) return GRM_ERR_ARG;
    if (genome_index < 0 || genome_index >= b->n_genomes || (len && !buf))
        return fail(b->ctx, GRM_ERR_ARG, "grm_batch_add: bad genome index %d / buffer", genome_index);
    if (b->uploaded) return fail(b->ctx, GRM_ERR_STATE, "grm_batch_add after upload");
    HostFile f;
    f.genome = genome_index;
    const uint8_t *p = (const uint8_t *)buf;
    if (len >= 2 && p[0] == 0x1f && p[1] == 0x8b) {
        if (!inflate_gzip(p, len, f.bytes)) return fail(b->ctx, GRM_ERR_IO, "grm_batch_add: corrupt gzip stream (genome %d)", genome_index);
    } else {
        f.bytes.assign(p, p + len);
    }
    size_t i = 0;
    while (i < f.bytes.size() && (f.bytes[i] == '\n' || f.bytes[i] == '\r' || f.bytes[i] == ' ' || f.bytes[i] == '\t')) i++;
    f.fastq = i < f.bytes.size() && f.bytes[i] == '@';
    f.size = f.bytes.size();
    b->files.push_back(std::move(f));
    return GRM_OK;
}

extern "C" int grm_batch_add_file(grm_batch *b, int genome_index, const char *path)
{
    if (!b || !path) return GRM_ERR_ARG;
    if (genome_index < 0 || genome_index >= b->n_genomes)
        return fail(b->ctx, GRM_ERR_ARG, "grm_batch_add_file: bad genome index %d", genome_index);
    if (b->uploaded) return fail(b->ctx, GRM_ERR_STATE, "grm_batch_add_file after upload");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(b->ctx, GRM_ERR_IO, "cannot open %s", path);
    // a plain regular file is only looked at here (format, size); its bytes go from the page
    // cache straight into the pinned upload slab, in parallel, at grm_batch_upload
    struct stat st;
    uint8_t head[4096];
    if (fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        const size_t got = fread(head, 1, sizeof head, f);
        const bool gz = got >= 2 && head[0] == 0x1f && head[1] == 0x8b;
        size_t i = 0;
        while (i < got && (head[i] == '\n' || head[i] == '\r' || head[i] == ' ' || head[i] == '\t')) i++;
        if (!gz && (i < got || got == (size_t)st.st_size)) {
            HostFile hf;
            hf.genome = genome_index;
            hf.fastq = i < got && head[i] == '@';
            hf.path = path;
            hf.size = (uint64_t)st.st_size;
            fclose(f);
            b->files.push_back(std::move(hf));
            return GRM_OK;
        }
        rewind(f);
    }
    std::vector<uint8_t> bytes;
    if (fseek(f, 0, SEEK_END) == 0) {                 // regular file: one read of the whole image
        const long sz = ftell(f);
        rewind(f);
        if (sz > 0) {
            bytes.resize((size_t)sz);
            const size_t got = fread(bytes.data(), 1, (size_t)sz, f);
            bytes.resize(got);
        }
    }
    uint8_t tmp[1 << 16];                             // pipes / anything left
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) bytes.insert(bytes.end(), tmp, tmp + got);
    const bool err = ferror(f) != 0;
    fclose(f);
    if (err) return fail(b->ctx, GRM_ERR_IO, "read error on %s", path);
    return grm_batch_add(b, genome_index, bytes.data(), bytes.size());
}

static inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

extern "C" int grm_batch_upload(grm_batch *b)
{
    if (!b) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (b->uploaded) return fail(c, GRM_ERR_STATE, "batch already uploaded");
    HIPCHK(c, hipSetDevice(c->device));
    // genome-major, stable in insertion order
    std::stable_sort(b->files.begin(), b->files.end(), [](const HostFile &x, const HostFile &y) { return x.genome < y.genome; });
    // layout: every FASTA file = ">\n" + bytes + "\n", every FASTQ file = bytes + "\n", padded with
    // '\n' to a tile boundary.  The synthetic FASTA header guarantees a separator in front of a
    // file's first base (a FASTQ file opens with its own header line); the trailing newline
    // terminates an unterminated last line.
    std::vector<uint32_t> genome_tile_off(b->n_genomes + 1, 0);
    std::vector<uint8_t> tile_meta;
    uint64_t pos = 0;
    std::vector<uint64_t> file_pos(b->files.size());
    size_t fi = 0;
    b->input_bytes = 0;
    for (int g = 0; g < b->n_genomes; g++) {
        genome_tile_off[g] = (uint32_t)(pos / TILE_BYTES);
        bool any = false;
        while (fi < b->files.size() && b->files[fi].genome == g) {
            file_pos[fi] = pos;
            const uint64_t end = round_up(pos + (b->files[fi].fastq ? 0 : 2) + b->files[fi].size + 1, TILE_BYTES);
            const uint8_t fmt = b->files[fi].fastq ? TILE_META_FASTQ : 0;
            for (uint64_t t = pos; t < end; t += TILE_BYTES) tile_meta.push_back(fmt | (t == pos ? TILE_META_FIRST : 0));
            pos = end;
            b->input_bytes += b->files[fi].size;
            fi++;
            any = true;
        }
        if (!any) { pos += TILE_BYTES; tile_meta.push_back(TILE_META_FIRST); }   // an empty genome owns one all-newline tile
    }
    genome_tile_off[b->n_genomes] = (uint32_t)(pos / TILE_BYTES);
    if (pos / TILE_BYTES >= 0xffffffffull) return fail(c, GRM_ERR_ARG, "batch too large (%llu bytes)", (unsigned long long)pos);
    b->raw_bytes = pos;
    b->n_tiles = (uint32_t)(pos / TILE_BYTES);

    HIPCHK(c, b->d_raw_alloc.alloc(RAW_FRONT_PAD + pos + 64));
    uint8_t *d_raw = b->d_raw_alloc.as<uint8_t>();
    // The image is assembled in two pinned slabs: host threads fill one (memory buffers by memcpy,
    // plain files by pread from the page cache) while the other is on its way over PCIe.
    {
        const uint64_t total = RAW_FRONT_PAD + pos + 64;
        // 32 MiB: a filler's piece of it (2 MiB at 16 threads) stays in its core's L2 between the newline fill and the read; measured
        // at 1000 x 5.6 MB files: 0.165 s for 5.6 GB against 0.222 s with 128 MiB slabs (and 0.33 s with 8 MiB: too little work per hand-over)
        const uint64_t slab_want = c->opt_upload_slab_kb > 0 ? (uint64_t)c->opt_upload_slab_kb << 10 : (uint64_t)32 << 20;
        const size_t slab_bytes = (size_t)std::min<uint64_t>(slab_want, total);
        uint8_t *pinned[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        bool in_flight[2] = {false, false};
        std::string io_err;
        std::mutex io_mu;
        auto cleanup = [&]() {
            (void)hipStreamSynchronize(c->copy_stream);
            for (int i = 0; i < 2; i++) {
                if (pinned[i]) (void)hipHostFree(pinned[i]);
                if (done[i]) (void)hipEventDestroy(done[i]);
            }
        };
        const int n_slabs = total > slab_bytes ? 2 : 1;
        for (int i = 0; i < n_slabs; i++) {
            hipError_t e = hipHostMalloc((void **)&pinned[i], slab_bytes, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
            if (e != hipSuccess) { cleanup(); return fail(c, GRM_ERR_HIP, "upload: pinned slab: %s", hipGetErrorString(e)); }
        }
        // host threads that fill a slab (page cache -> pinned memory): GRM_UPLOAD_THREADS, else 16 (what a one-GPU share of a
        // box gives; more only take turns there)
        const unsigned hw = std::thread::hardware_concurrency();
        int n_thr = (int)std::min<unsigned>(16u, std::max(1u, hw));
        if (const char *e = getenv("GRM_UPLOAD_THREADS")) {
            const int v = atoi(e);
            if (v > 0 && v <= 256) n_thr = v;
        }
        // fills image bytes [a, b) (absolute image offsets) into dst (which maps offset a)
        auto fill_range = [&](uint8_t *dst, uint64_t a, uint64_t bnd) {
            memset(dst, '\n', (size_t)(bnd - a));
            // first file whose image may reach a: file_pos is ascending
            size_t lo_i = 0, hi_i = b->files.size();
            while (lo_i < hi_i) {
                const size_t mid = (lo_i + hi_i) / 2;
                const uint64_t f1 = RAW_FRONT_PAD + file_pos[mid] + (b->files[mid].fastq ? 0 : 2) + b->files[mid].size;
                if (f1 <= a) lo_i = mid + 1; else hi_i = mid;
            }
            for (size_t i = lo_i; i < b->files.size(); i++) {
                const HostFile &hf = b->files[i];
                const size_t head = hf.fastq ? 0 : 2;
                const uint64_t f0 = RAW_FRONT_PAD + file_pos[i];                 // image start (header included)
                const uint64_t f1 = f0 + head + hf.size;                         // image end (the trailing \n is padding)
                if (f0 >= bnd) break;
                if (f1 <= a) continue;
                if (head) {
                    if (f0 >= a && f0 < bnd) dst[f0 - a] = '>';
                    if (f0 + 1 >= a && f0 + 1 < bnd) dst[f0 + 1 - a] = '\n';
                }
                const uint64_t d0 = f0 + head;                                   // data start
                const uint64_t lo = std::max(d0, a), hi = std::min(f1, bnd);
                if (hi <= lo) continue;
                if (!hf.deferred()) {
                    memcpy(dst + (lo - a), hf.bytes.data() + (lo - d0), (size_t)(hi - lo));
                    continue;
                }
                const int fd = open(hf.path.c_str(), O_RDONLY);
                bool ok = fd >= 0;
                uint64_t at = lo;
                while (ok && at < hi) {
                    const ssize_t r = pread(fd, dst + (at - a), (size_t)(hi - at), (off_t)(at - d0));
                    if (r <= 0) ok = false; else at += (uint64_t)r;
                }
                if (fd >= 0) close(fd);
                if (!ok) {
                    std::lock_guard<std::mutex> g(io_mu);
                    if (io_err.empty()) io_err = "read error on " + hf.path + " (changed since it was added?)";
                }
            }
        };
        // the fillers live for the whole upload (a slab is a few milliseconds of work: starting threads per slab cost a fifth of it)
        struct FillPool {
            std::vector<std::thread> th;
            std::mutex mu;
            std::condition_variable go, done;
            uint64_t gen = 0;
            int pending = 0;
            bool stop = false;
            std::function<void(int)> job;
            explicit FillPool(int n)
            {
                for (int t = 0; t < n; t++)
                    th.emplace_back([this, t]() {
                        uint64_t seen = 0;
                        for (;;) {
                            std::function<void(int)> j;
                            {
                                std::unique_lock<std::mutex> lk(mu);
                                go.wait(lk, [&]() { return stop || gen != seen; });
                                if (stop) return;
                                seen = gen;
                                j = job;
                            }
                            j(t);
                            std::lock_guard<std::mutex> lk(mu);
                            if (--pending == 0) done.notify_one();
                        }
                    });
            }
            void run(std::function<void(int)> j)
            {
                std::unique_lock<std::mutex> lk(mu);
                job = std::move(j);
                pending = (int)th.size();
                gen++;
                go.notify_all();
                done.wait(lk, [&]() { return pending == 0; });
            }
            ~FillPool()
            {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    stop = true;
                }
                go.notify_all();
                for (auto &t : th) t.join();
            }
        };
        FillPool pool(n_thr);
        int which = 0;
        for (uint64_t s0 = 0; s0 < total; s0 += slab_bytes, which ^= 1) {
            const uint64_t s1 = std::min<uint64_t>(s0 + slab_bytes, total);
            if (in_flight[which]) {
                hipError_t e = hipEventSynchronize(done[which]);
                if (e != hipSuccess) { cleanup(); return fail(c, GRM_ERR_HIP, "upload: %s", hipGetErrorString(e)); }
                in_flight[which] = false;
            }
            uint8_t *slab = pinned[which];
            const uint64_t len = s1 - s0, piece = round_up((len + n_thr - 1) / n_thr, 4096);
            pool.run([&](int t) {
                const uint64_t a = s0 + (uint64_t)t * piece, bnd = std::min(s1, a + piece);
                if (a < bnd) fill_range(slab + (a - s0), a, bnd);
            });
            if (!io_err.empty()) { cleanup(); return fail(c, GRM_ERR_IO, "%s", io_err.c_str()); }
            hipError_t e = hipMemcpyAsync(d_raw + s0, slab, (size_t)len, hipMemcpyHostToDevice, c->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(done[which], c->copy_stream);
            if (e != hipSuccess) { cleanup(); return fail(c, GRM_ERR_HIP, "upload: %s", hipGetErrorString(e)); }
            in_flight[which] = true;
        }
        hipError_t e = hipStreamSynchronize(c->copy_stream);
        cleanup();
        if (e != hipSuccess) return fail(c, GRM_ERR_HIP, "upload: %s", hipGetErrorString(e));
    }
    for (auto &f : b->files) std::vector<uint8_t>().swap(f.bytes);
    HIPCHK(c, b->d_tile_meta.alloc(tile_meta.size() + 16));
    if (!tile_meta.empty()) HIPCHK(c, hipMemcpyAsync(b->d_tile_meta.p, tile_meta.data(), tile_meta.size(), hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, b->d_genome_tile_off.alloc((b->n_genomes + 1) * 4));
    HIPCHK(c, hipMemcpyAsync(b->d_genome_tile_off.p, genome_tile_off.data(), (b->n_genomes + 1) * 4, hipMemcpyHostToDevice, c->copy_stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    b->uploaded = true;
    return GRM_OK;
}

// presence words kept by dict_build: 2^(bb+sb) workgroups x word-rows x table slots; beyond this many bytes
// the probing form of the fill is used instead (it needs no intermediate)
static const size_t MATRIX_S_LIMIT = (size_t)96 << 30;

static int pick_bucket_bits(grm_ctx *c, uint64_t max_genome_syms)
{
    if (c->opt_bucket_bits >= 0) return std::min(c->opt_bucket_bits, MAX_BUCKET_BITS);
    // aim at ~512 k-mer occurrences per (genome, bucket); past 2^13 buckets (deep mode: one more
    // pass over the keys for the fine histogram) only when segments would exceed 2048 occurrences
    int bb = 0;
    while (bb < MAX_HIST_BITS && (max_genome_syms >> bb) > 512) bb++;
    while (bb < MAX_BUCKET_BITS && (max_genome_syms >> bb) > 2048) bb++;
    return bb;
}

// LDS table size: 12 B (dict) / 16 B (fill) per slot; 2^12 slots = 64 KiB lets two
// workgroups share a CU's 160 KiB.  Clamped to what one workgroup may allocate.
static uint32_t pick_cap_log2(grm_ctx *c)
{
    int v = c->opt_cap_log2 > 0 ? c->opt_cap_log2 : 12;
    return (uint32_t)std::min(13, std::max(6, v));
}

// segment layout of the partitioned keys of a batch: dense (offsets from the histogram; lengths only once a dedup has
// shortened the segments) or slack (fixed-capacity segments + lengths)
static SegLayout batch_segments(const grm_batch *b, bool after_dedup = true)
{
    SegLayout L;
    if (b->seg_stride) {
        L.off = nullptr;
        L.len = b->d_len.as<uint32_t>();
        L.stride = b->seg_stride;
    } else {
        L.off = b->d_off.as<uint64_t>();
        L.len = (b->rec_mode || (after_dedup && b->deduped)) ? b->d_len.as<uint32_t>() : nullptr;     // record form: regions leave gaps
        L.stride = 0;
    }
    return L;
}

// parse + histogram + scan + scatter (+ dedup when abundance_min > 1)
static int batch_partition_impl(grm_batch *b, int k, uint32_t abundance_min, bool want_counts)
{
    grm_ctx *c = b->ctx;
    if (!b->uploaded) return fail(c, GRM_ERR_STATE, "grm_batch_partition before grm_batch_upload");
    if (k < 1 || k > GRM_MAX_K) return fail(c, GRM_ERR_ARG, "k=%d out of range (1..128)", k);
    if (abundance_min < 1) abundance_min = 1;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const uint8_t *raw = b->d_raw_alloc.as<uint8_t>() + RAW_FRONT_PAD;
    const uint32_t G = (uint32_t)b->n_genomes;
    b->partitioned = b->have_local = b->have_global = false;
    b->k = k;
    b->abundance_min = abundance_min;
    b->deduped = false;
    b->rec_mode = b->rec_dict = false;

    if (b->n_tiles == 0 || G == 0) {
        b->total_syms = b->total_keys = 0;
        b->bb = 0;
        b->h_genome_sym_off.assign(G + 1, 0);
        HIPCHK(c, b->d_off.ensure((G + 2) * 8));
        HIPCHK(c, hipMemsetAsync(b->d_off.p, 0, (G + 2) * 8, s));
        HIPCHK(c, b->d_keys.ensure(16));
        HIPCHK(c, hipStreamSynchronize(s));
        b->partitioned = true;
        return GRM_OK;
    }

    // ---- stage 0: parse ----
    const uint64_t max_groups = b->raw_bytes / 64 + 8;
    {
        HIPCHK(c, b->d_sums.ensure((size_t)b->n_tiles * sizeof(TileSummary)));
        HIPCHK(c, b->d_tile_off.ensure(((size_t)b->n_tiles + 1) * 8));
        HIPCHK(c, b->d_tile_state.ensure((size_t)b->n_tiles + 16));
        HIPCHK(c, b->d_sym2.ensure(max_groups * 16));
        HIPCHK(c, b->d_inv.ensure(max_groups * 8));
        HIPCHK(c, b->d_genome_sym_off.ensure(((size_t)G + 1) * 8));
    }
    if (c->opt_parse_fused > 0) {
        // one pass over the input: a tile learns what runs into it by a look-back over the tiles before it (grm_kernels.hip)
        HIPCHK(c, b->d_scan_scratch.ensure(parse_fused_desc_bytes(b->n_tiles)));
        HIPCHK(c, b->d_chunk_pre.ensure(parse_fused_piece_bytes(b->n_tiles)));
        TimeScope t(c, "parse_fused", b->raw_bytes);
        HIPCHK(c, launch_parse_fused(s, raw, b->n_tiles, b->d_tile_meta.as<uint8_t>(), b->d_scan_scratch.as<uint64_t>(), b->d_chunk_pre.as<uint64_t>(),
                                     b->d_tile_off.as<uint64_t>(), b->d_sym2.as<uint64_t>(), b->d_inv.as<uint64_t>(),
                                     b->d_genome_tile_off.as<uint32_t>(), G, b->d_genome_sym_off.as<uint64_t>()));
    } else {
    HIPCHK(c, b->d_scan_scratch.ensure(parse_scan_scratch_bytes(b->n_tiles)));
    bool any_fastq = false;
    for (const auto &f : b->files) any_fastq = any_fastq || f.fastq;
    HIPCHK(c, b->d_chunk_pre.ensure(parse_chunk_pre_bytes(b->n_tiles, any_fastq)));
    uint64_t *chunk_pre64 = any_fastq ? reinterpret_cast<uint64_t *>(b->d_chunk_pre.as<uint8_t>() + parse_chunk_pre_bytes(b->n_tiles, false)) : nullptr;
    // (no memset of the packed stream: parse_pack's companion kernel zeroes the groups that need it)
    {
        TimeScope t(c, "parse_summarize", b->raw_bytes);
        launch_parse_summarize(s, raw, b->n_tiles, b->d_tile_meta.as<uint8_t>(), b->d_sums.as<TileSummary>(), b->d_chunk_pre.as<uint32_t>(), chunk_pre64);
    }
    {
        TimeScope t(c, "parse_scan", b->n_tiles);
        launch_parse_scan(s, b->d_sums.as<TileSummary>(), b->n_tiles, b->d_tile_meta.as<uint8_t>(), b->d_tile_off.as<uint64_t>(),
                          b->d_tile_state.as<uint8_t>(), b->d_genome_tile_off.as<uint32_t>(), G,
                          b->d_genome_sym_off.as<uint64_t>(), b->d_scan_scratch.p);
    }
    {
        TimeScope t(c, "parse_pack", b->raw_bytes);
        launch_parse_pack(s, raw, b->n_tiles, b->d_tile_meta.as<uint8_t>(), b->d_tile_off.as<uint64_t>(), b->d_tile_state.as<uint8_t>(),
                          b->d_sym2.as<uint64_t>(), b->d_inv.as<uint64_t>(), b->d_sums.as<TileSummary>(), b->d_chunk_pre.as<uint32_t>(), chunk_pre64);
    }
    }
    HIPCHK(c, hipGetLastError());
    b->h_genome_sym_off.resize(G + 1);
    HIPCHK(c, hipMemcpyAsync(b->h_genome_sym_off.data(), b->d_genome_sym_off.p, ((size_t)G + 1) * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    b->total_syms = b->h_genome_sym_off[G];
    uint64_t max_g = 0;
    for (uint32_t g = 0; g < G; g++) max_g = std::max(max_g, b->h_genome_sym_off[g + 1] - b->h_genome_sym_off[g]);
    if (max_g >= 0xffffffffull) return fail(c, GRM_ERR_ARG, "a genome has %llu symbols (limit 2^32-1)", (unsigned long long)max_g);
    if (b->total_syms > max_groups * 64 - 256) return fail(c, GRM_ERR_HIP, "internal: symbol count exceeds the packed buffers");
    if (k > 32) return GRM_OK;       // two-word k-mers: the caller continues on grm_wide_hash.hip / grm_wide.hip
    b->bb = pick_bucket_bits(c, max_g);
    b->rec_mode = false;
    b->rec_dict = false;
    b->cap_log2 = pick_cap_log2(c);

    // ---- record form (grm_superkmer.hip): buckets by minimizer; level 1 moves runs of consecutive k-mers as 16-byte
    // records, level 2 expands them into the same bucket-sorted key segments the key form leaves ----
    // Minimizer buckets are less even than hashed k-mers (a fine bucket holds ~30 minimizers of very different weight:
    // sigma ~25 % of the mean at 2^13 buckets), hence one more bucket bit than the key form and exact (not slack)
    // segment sizes inside a region.
    // (k < 19: a k-mer holds fewer than 9 m-mers and runs get short: measured at 300 x 5 Mbp, the record form takes 16.2 ms against
    // 18.0 for the key form at k = 21, but 19.8 against 17.9 at k = 15 and 28.7 against 16.8 at k = 12 -- unless asked for)
    // A counting partition (counts wanted, or an abundance filter) takes the records too: level 2 sorts the records by bucket and
    // record_count / record_dedup (grm_kernels.hip) count the k-mers of a (genome, bucket) segment straight from its records --
    // 1.45 B per k-mer moved twice instead of 8 B moved twice and read a third time.  Few large genomes (read sets) are cut into
    // parts for level 1: record_merge then counts a genome's bucket over its parts' segments in one workgroup-wide table.
    const bool counting = abundance_min > 1 || want_counts;
    if (k >= (c->opt_records > 0 ? SK_M : SK_M + 8) && (!counting || c->opt_rec_count != 0) && c->opt_records != 0 && !b->rec_failed &&
        c->opt_dense_layout <= 0) {
        int bbr = c->opt_bucket_bits >= 0 ? b->bb : b->bb + (c->opt_rec_bucket_shift >= 0 ? c->opt_rec_bucket_shift : 1);
        if (c->opt_bucket_bits < 0 && b->rec_bb_hint > bbr && b->rec_bb_hint_k == k) bbr = b->rec_bb_hint;
        // (counting: a wave's table of 512 slots per segment -- segments of ~150 k-mers, minimizer buckets being uneven)
        if (counting && c->opt_bucket_bits < 0) while (bbr < superkmer_coarse_bits(bbr + 1) + 7 && (max_g >> bbr) > 192) bbr++;
        bbr = std::min(bbr, superkmer_max_bits());
        // (a record carries 7 bucket bits below the coarse ones)
        if (c->opt_rec_coarse <= 0) bbr = std::min(bbr, superkmer_coarse_bits(bbr) + 7);
        const int b1r = c->opt_rec_coarse > 0 ? std::max(bbr - 7, std::min(c->opt_rec_coarse, superkmer_coarse_bits(bbr))) : superkmer_coarse_bits(bbr);
        // one workgroup per genome part owns the part's regions: enough parts to fill the device when genomes are few
        int pbits = 0;
        // (one 1024-thread workgroup per CU and part: 128 genomes of 5 Mbp measured 7.6 / 7.5 / 8.3 / 10.2 ms per pass with 1 / 2 / 4 / 8
        // parts per genome -- level 1 gains, level 2 and dict_build pay for the smaller segments)
        while (((uint64_t)G << pbits) < 256 && pbits < 6 && (max_g >> (pbits + 1)) >= 65536) pbits++;
        if (c->opt_rec_part_bits >= 0) pbits = std::min(c->opt_rec_part_bits, 6);
        // (counting a read set with more, smaller parts -- so that level 2 still sorts a region inside LDS: 3.8 instead of 7.1 ms for C4's
        // 7e8 records -- costs record_merge more than that: 27.1 against 20.2 ms with 128 instead of 32 parts per genome)
        const uint64_t n_parts = (uint64_t)G << pbits;
        const uint64_t n_regions = n_parts << b1r, n_seg_r = n_parts << bbr;
        // records per region: a run ends where the minimizer occurrence changes -- 2 / (w + 1) per position for the w m-mers
        // of a k-mer -- or at a base that is none (0.005: contig ends, N runs); k-mers per region: ~0.05 distinct minimizers per
        // position, sigma of a region = 1.39 x its share / sqrt(its minimizers) (both simulated)
        const int w = k - SK_M + 1;
        const double mean_k = (double)((max_g >> pbits) >> b1r) + 1.0;
        const double mean_r = mean_k * (2.0 / (w + 1) + 0.005);
        const uint64_t rstride64 = (uint64_t)(mean_r * 1.02 + 14.0 * std::sqrt(mean_r) + 32.0 + 15.0) / 16 * 16;     // mean + 7.5 sigma
        const uint64_t kstride = (uint64_t)(mean_k + 50.0 * std::sqrt(mean_k) + 256.0 + 15.0) / 16 * 16;
        // genomes of very different sizes would waste most of a layout sized for the largest one
        // (a segment's record count travels in 16 bits beside the count of its short records: only a forced, tiny bucket count gets
        // near; level 2 raises the overflow flag if one does)
        const double expected_records = (double)b->total_syms * (2.0 / (w + 1) + 0.005);
        bool rec = n_seg_r < 0xffffffffull && (rstride64 << b1r) < 0xffffffffull && kstride < 0xffffffffull && mean_r / (double)(1u << (bbr - b1r)) < 16000.0 &&
                   (double)n_regions * (double)rstride64 <= 4.0 * expected_records + 16.0 * 1024 * 1024;
        // the fill through the presence bits needs no keys at all: dict_build then decodes the records itself and
        // level 2 only sorts them by fine bucket; else (probing fill) level 2 expands them to key segments
        const size_t n_rows_b = ((size_t)G + 63) / 64;
        // (dict_build's record memo takes LDS: with it the key table has 2^11 slots unless the caller says otherwise, so that
        // two workgroups still share a CU)
        b->rec_memo_log2 = c->opt_rec_memo < 0 ? 8 : c->opt_rec_memo == 0 ? 0 : std::min(11, std::max(8, c->opt_rec_memo));
        const uint32_t cap_r = (b->rec_memo_log2 && c->opt_cap_log2 <= 0) ? 11u : b->cap_log2;
        bool by_records = !counting && c->opt_no_slots <= 0 && c->opt_rec_keys <= 0 && n_rows_b <= 0xffffu &&
                          ((size_t)1 << bbr) * n_rows_b * ((size_t)1 << cap_r) * 8 <= MATRIX_S_LIMIT;
        // counting, one part per genome: segments of more than 448 k-mers take the second, workgroup-wide launch, and more than 1792 distinct
        // ones fit no table; level 1 runs one workgroup per part: fewer than 128 of them leave the device idle (the key form then)
        if (counting && pbits == 0 && (max_g >> bbr) > 600) rec = false;
        if (counting && n_parts < (pbits ? 32u : 128u)) rec = false;
        const bool l2_records = by_records || counting;
        if (rec && b->d_recs.ensure((n_regions * rstride64 + 4) * 16) != hipSuccess) {
            (void)hipGetLastError();
            rec = false;
        }
        if (rec && l2_records && b->d_recs2.ensure((n_regions * rstride64 + 4) * 16) != hipSuccess) {
            (void)hipGetLastError();
            if (counting) rec = false;
            by_records = false;
        }
        // (key segments: a layout of kstride keys per region)
        if (rec && !by_records && (double)n_regions * (double)kstride > 3.0 * (double)b->total_syms + 65536.0 * 1024.0) rec = false;
        if (rec && counting) {
            hipError_t e = b->d_keys.ensure((n_regions * kstride + 4) * 8);
            if (e == hipSuccess && want_counts) e = b->d_kcnt.ensure((n_regions * kstride + 4) * 4);
            if (e == hipSuccess) e = b->d_koff.ensure((n_seg_r + 1) * 8);
            if (e == hipSuccess) e = b->d_klen.ensure((n_seg_r + 1) * 4);
            if (e != hipSuccess) { (void)hipGetLastError(); rec = false; }
        }
        if (rec) {
            const uint32_t rstride = (uint32_t)rstride64;
            KmerLaunch Lr;
            Lr.sym2 = b->d_sym2.as<uint64_t>(); Lr.inv = b->d_inv.as<uint64_t>(); Lr.total_syms = b->total_syms;
            Lr.genome_sym_off = b->d_genome_sym_off.as<uint64_t>(); Lr.n_genomes = G; Lr.k = k; Lr.bb = bbr; Lr.groups_per_thread = 1;
            HIPCHK(c, b->d_off.ensure((n_seg_r + 1) * 8));
            HIPCHK(c, b->d_len.ensure((n_seg_r + 1) * 4));
            HIPCHK(c, b->d_counts1.ensure((n_regions + 1) * 4));
            HIPCHK(c, b->d_cursor1.ensure(n_parts * 4));
            HIPCHK(c, b->t_flag.ensure(32));
            HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 32, s));
            {
                TimeScope t(c, "superkmer_l1", b->total_syms);
                launch_superkmer_l1(s, Lr, b1r, pbits, b->d_recs.p, rstride, b->d_counts1.as<uint32_t>(), b->d_cursor1.as<uint32_t>(), b->t_flag.as<int>());
            }
            b->rec_rstride = rstride; b->rec_kstride = kstride; b->rec_regions = n_regions; b->rec_b1 = b1r;
            int l2_idx = -1;
            if (l2_records) {
                TimeScope t(c, "superkmer_l2", b->total_syms);        // (units: the record count, once it is known)
                l2_idx = t.idx;
                launch_superkmer_l2_records(s, b->d_recs.p, rstride, b->d_counts1.as<uint32_t>(), n_regions, k, bbr, b1r, b->d_recs2.p,
                                            b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), b->t_flag.as<int>());
            } else {
                HIPCHK(c, b->d_keys.ensure((n_regions * kstride + 4) * 8));
                TimeScope t(c, "superkmer_l2_keys", b->total_syms);
                launch_superkmer_l2(s, b->d_recs.p, rstride, b->d_counts1.as<uint32_t>(), n_regions, k, bbr, b1r, kstride, b->d_keys.as<uint64_t>(),
                                    b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), b->t_flag.as<int>());
            }
            launch_sum_u32(s, b->d_cursor1.as<uint32_t>(), n_parts, reinterpret_cast<uint64_t *>(b->t_flag.as<uint8_t>() + 8));
            launch_sum_u32(s, b->d_counts1.as<uint32_t>(), n_regions, reinterpret_cast<uint64_t *>(b->t_flag.as<uint8_t>() + 16));
            HIPCHK(c, hipGetLastError());
            struct { int over; int pad; uint64_t total; uint64_t records; uint64_t pad2; } h;
            HIPCHK(c, hipMemcpyAsync(&h, b->t_flag.p, 32, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            if (l2_idx >= 0 && l2_idx < (int)c->recs.size()) c->recs[l2_idx].units = h.records;
            bool merged_parts = false;
            // (record_merge adds a genome's k-mer occurrences per coarse bucket over all its parts in 32 bits and compares the sum with the
            // key capacity kstride << pbits: beyond 2^32 - 1 the sum could wrap and pass the test -- such a read set takes the key form)
            if (!h.over && counting && pbits > 0 && (kstride << pbits) >= (1ull << 32)) h.over = 1;
            if (!h.over && counting && pbits > 0) {
                // genomes in parts: a genome's bucket is counted over its parts' record segments (the table must hold its DISTINCT k-mers:
                // 2^12 slots first, 2^13 if those overflow, the key form after that)
                const uint64_t n_seg_g = (uint64_t)G << bbr;
                HIPCHK(c, b->d_koff.ensure((n_seg_g + 1) * 8));
                HIPCHK(c, b->d_klen.ensure((n_seg_g + 1) * 4));
                int fl = 0;
                for (int big = c->opt_rec_count_cap >= 12 ? std::min(13, c->opt_rec_count_cap) : 12; big <= 13; big++) {
                    HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 8, s));
                    {
                        TimeScope t(c, "record_merge", h.total);
                        HIPCHK(c, launch_record_merge(s, b->d_recs2.p, rstride, b->d_counts1.as<uint32_t>(), b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), G, pbits,
                                                      k, bbr, b1r, kstride << pbits, big, abundance_min, b->d_keys.as<uint64_t>(),
                                                      want_counts ? b->d_kcnt.as<uint32_t>() : nullptr, b->d_koff.as<uint64_t>(), b->d_klen.as<uint32_t>(),
                                                      b->t_flag.as<int>()));
                    }
                    HIPCHK(c, hipGetLastError());
                    HIPCHK(c, hipMemcpyAsync(&fl, b->t_flag.p, 4, hipMemcpyDeviceToHost, s));
                    HIPCHK(c, hipStreamSynchronize(s));
                    if (!fl) break;
                }
                h.over = fl;
                if (!h.over) {
                    std::swap(b->d_off.p, b->d_koff.p); std::swap(b->d_off.bytes, b->d_koff.bytes);
                    std::swap(b->d_len.p, b->d_klen.p); std::swap(b->d_len.bytes, b->d_klen.bytes);
                    b->deduped = true;
                    merged_parts = true;
                }
            } else if (!h.over && counting) {
                // distinct k-mers (+ counts, abundance filter) of every (genome, bucket) segment, from its records
                const int cap_w = c->opt_rec_count_cap == 8 ? 8 : 9;
                HIPCHK(c, b->d_marks.ensure(n_regions + 16));
                HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 8, s));
                {
                    TimeScope t(c, "record_count", h.total);
                    launch_record_count(s, b->d_recs2.p, rstride, b->d_counts1.as<uint32_t>(), n_regions, k, bbr, b1r, kstride, cap_w, abundance_min,
                                        b->d_keys.as<uint64_t>(), want_counts ? b->d_kcnt.as<uint32_t>() : nullptr, b->d_koff.as<uint64_t>(),
                                        b->d_klen.as<uint32_t>(), b->t_flag.as<int>(), b->d_marks.as<uint8_t>(), b->t_flag.as<int>() + 1);
                }
                HIPCHK(c, hipGetLastError());
                int fl[2] = {0, 0};
                HIPCHK(c, hipMemcpyAsync(fl, b->t_flag.p, 8, hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
                if (!fl[0] && fl[1]) {
                    // segments with more k-mers or records than a wave takes: the workgroup form over the regions that hold one
                    TimeScope t(c, "record_dedup", h.total);
                    launch_record_dedup_rest(s, b->d_recs2.p, rstride, b->d_counts1.as<uint32_t>(), n_regions, k, bbr, b1r, kstride, cap_w, abundance_min,
                                             b->d_keys.as<uint64_t>(), want_counts ? b->d_kcnt.as<uint32_t>() : nullptr, b->d_koff.as<uint64_t>(),
                                             b->d_klen.as<uint32_t>(), b->t_flag.as<int>(), b->d_marks.as<uint8_t>());
                    HIPCHK(c, hipGetLastError());
                    HIPCHK(c, hipMemcpyAsync(fl, b->t_flag.p, 4, hipMemcpyDeviceToHost, s));
                    HIPCHK(c, hipStreamSynchronize(s));
                }
                h.over = fl[0];
                if (!h.over) {
                    // the key segments take the place of the record segments
                    std::swap(b->d_off.p, b->d_koff.p); std::swap(b->d_off.bytes, b->d_koff.bytes);
                    std::swap(b->d_len.p, b->d_klen.p); std::swap(b->d_len.bytes, b->d_klen.bytes);
                    b->deduped = true;
                }
            }
            if (!h.over) {
                b->bb = bbr;
                b->total_keys = h.total;
                b->seg_stride = 0;
                b->rec_part_bits = merged_parts ? 0 : pbits;         // (merged: the segments are the genomes' again)
                b->rec_count_pbits = counting ? pbits : 0;
                b->rec_mode = true;
                b->rec_dict = by_records;
                if (by_records) b->cap_log2 = cap_r;
                b->partitioned = true;
                return GRM_OK;
            }
            b->rec_failed = true;            // repeat-rich input: the key form from now on
        }
    }
    const uint64_t B = 1ull << b->bb;
    const uint64_t n_seg = (uint64_t)G * B;

    // ---- stage 1: histogram, scan, scatter ----
    KmerLaunch L;
    L.sym2 = b->d_sym2.as<uint64_t>();
    L.inv = b->d_inv.as<uint64_t>();
    L.total_syms = b->total_syms;
    L.genome_sym_off = b->d_genome_sym_off.as<uint64_t>();
    L.n_genomes = G;
    L.k = k;
    L.bb = b->bb;
    L.groups_per_thread = c->opt_groups_per_thread > 0 ? (uint32_t)c->opt_groups_per_thread : 4u;

    const int b1 = scatter_b1_bits(b->bb);
    const bool deep = b->bb > MAX_HIST_BITS;
    const uint64_t n_coarse = (uint64_t)G << b1;
    HIPCHK(c, b->d_cursor1.ensure(n_coarse * 4));
    HIPCHK(c, b->t_flag.ensure(16));

    // ---- slack layout: no histogram pass.  The hash spreads a genome's k-mers evenly, so a (genome, fine bucket)
    // segment holds mean m = symbols / 2^bb keys with a standard deviation of sqrt(m): segments get a fixed capacity
    // of m + 6 sqrt(m) + 32 keys (m taken from the largest genome) at computed offsets, level 1 reserves space in
    // fixed-capacity coarse regions, level 2 writes the fine segments and their lengths.  A region or segment that
    // would overflow (heavily repeated k-mers) raises a flag: the partition is then redone with the histogram-sized
    // dense layout below, and the batch remembers that for later runs.
    b->seg_stride = 0;
    bool slack = !deep && c->opt_dense_layout <= 0 && !b->slack_failed;
    uint32_t fine_cap = 0;
    uint64_t region_stride = 0;
    if (slack) {
        const uint64_t m = max_g >> b->bb;
        fine_cap = (uint32_t)((m + (uint64_t)(6.0 * std::sqrt((double)m + 1.0)) + 32 + 15) / 16 * 16);
        region_stride = (uint64_t)fine_cap << (b->bb - b1);
        // genomes of very different sizes would waste most of a layout sized for the largest one
        if ((double)n_seg * fine_cap > 1.75 * (double)b->total_syms + 65536.0) slack = false;
    }
    if (slack) {
        const uint64_t layout_keys = n_seg * (uint64_t)fine_cap;
        HIPCHK(c, b->d_keys.ensure((layout_keys + 2) * 8));
        if (b->bb > b1) HIPCHK(c, b->d_keys1.ensure((layout_keys + 2) * 8));
        HIPCHK(c, b->d_len.ensure((n_seg + 1) * 4));
        HIPCHK(c, hipMemsetAsync(b->d_cursor1.p, 0, n_coarse * 4, s));
        HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 16, s));
        {
            TimeScope t(c, "kmer_scatter_l1", b->total_syms);
            launch_kmer_scatter_l1(s, L, nullptr, nullptr, b->d_cursor1.as<uint32_t>(),
                                   b->bb > b1 ? b->d_keys1.as<uint64_t>() : b->d_keys.as<uint64_t>(), region_stride, b->t_flag.as<int>());
        }
        if (b->bb > b1) {
            TimeScope t(c, "kmer_scatter_l2", b->total_syms);
            launch_kmer_scatter_l2(s, L, nullptr, b->d_keys1.as<uint64_t>(), b->d_keys.as<uint64_t>(), region_stride, fine_cap,
                                   b->d_cursor1.as<uint32_t>(), b->d_len.as<uint32_t>(), b->t_flag.as<int>());
        } else {
            // one level: the coarse regions ARE the final segments, their fill the segment lengths
            HIPCHK(c, hipMemcpyAsync(b->d_len.p, b->d_cursor1.p, n_seg * 4, hipMemcpyDeviceToDevice, s));
        }
        launch_sum_u32(s, b->d_cursor1.as<uint32_t>(), n_coarse, reinterpret_cast<uint64_t *>(b->t_flag.as<uint8_t>() + 8));
        HIPCHK(c, hipGetLastError());
        struct { int over; int pad; uint64_t total; } h;
        HIPCHK(c, hipMemcpyAsync(&h, b->t_flag.p, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (h.over) {
            b->slack_failed = true;          // skewed input: dense layout from now on
            slack = false;
        } else {
            b->total_keys = h.total;
            b->seg_stride = fine_cap;
        }
    }
    if (!slack) {
    HIPCHK(c, b->d_counts.ensure((n_seg + 1) * 4));
    HIPCHK(c, b->d_off.ensure((n_seg + 1) * 8));
    HIPCHK(c, hipMemsetAsync(b->d_counts.p, 0, (n_seg + 1) * 4, s));
    HIPCHK(c, hipMemsetAsync(b->d_cursor1.p, 0, n_coarse * 4, s));
    auto scan_counts = [&](DevBuf &counts, DevBuf &off, uint64_t n) -> int {
        // exclusive scan over n+1 entries (the extra, zeroed entry yields the total)
        size_t tb = 0;
        HIPCHK(c, exclusive_scan_u32_u64(s, counts.as<uint32_t>(), off.as<uint64_t>(), n + 1, nullptr, tb));
        HIPCHK(c, b->t_tmp.ensure(tb));
        HIPCHK(c, exclusive_scan_u32_u64(s, counts.as<uint32_t>(), off.as<uint64_t>(), n + 1, b->t_tmp.p, tb));
        return GRM_OK;
    };
    if (!deep) {
        {
            TimeScope t(c, "kmer_hist", b->total_syms);
            launch_kmer_hist(s, L, b->d_counts.as<uint32_t>());
        }
        TimeScope t(c, "bucket_scan", n_seg);
        int r = scan_counts(b->d_counts, b->d_off, n_seg);
        if (r) return r;
    } else {
        // coarse histogram in the k-mer pass; the fine one is taken from the level-1 output
        HIPCHK(c, b->d_counts1.ensure((n_coarse + 1) * 4));
        HIPCHK(c, b->d_off1.ensure((n_coarse + 1) * 8));
        HIPCHK(c, hipMemsetAsync(b->d_counts1.p, 0, (n_coarse + 1) * 4, s));
        KmerLaunch Lc = L;
        Lc.bb = b1;
        {
            TimeScope t(c, "kmer_hist", b->total_syms);
            launch_kmer_hist(s, Lc, b->d_counts1.as<uint32_t>());
        }
        TimeScope t(c, "bucket_scan", n_coarse);
        int r = scan_counts(b->d_counts1, b->d_off1, n_coarse);
        if (r) return r;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&b->total_keys, (deep ? b->d_off1.as<uint64_t>() + n_coarse : b->d_off.as<uint64_t>() + n_seg), 8,
                             hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, b->d_keys.ensure((b->total_keys + 2) * 8));
    if (b->bb > b1) HIPCHK(c, b->d_keys1.ensure((b->total_keys + 2) * 8));
    {
        TimeScope t(c, "kmer_scatter_l1", b->total_keys);
        launch_kmer_scatter_l1(s, L, b->d_off.as<uint64_t>(), deep ? b->d_off1.as<uint64_t>() : nullptr, b->d_cursor1.as<uint32_t>(),
                               b->bb > b1 ? b->d_keys1.as<uint64_t>() : b->d_keys.as<uint64_t>(), 0, nullptr);
    }
    if (deep) {
        {
            TimeScope t(c, "region_hist", b->total_keys);
            launch_region_hist(s, b->d_keys1.as<uint64_t>(), b->d_off1.as<uint64_t>(), n_coarse, b->bb, b->d_counts.as<uint32_t>());
        }
        int r = scan_counts(b->d_counts, b->d_off, n_seg);
        if (r) return r;
    }
    if (b->bb > b1) {
        TimeScope t(c, "kmer_scatter_l2", b->total_keys);
        launch_kmer_scatter_l2(s, L, b->d_off.as<uint64_t>(), b->d_keys1.as<uint64_t>(),
                               b->d_keys.as<uint64_t>(), 0, 0, nullptr, nullptr, nullptr);
    }
    HIPCHK(c, hipGetLastError());
    }
    // key positions of the layout (dense: the keys themselves; slack: the segment slots): per-key side arrays are this long
    const uint64_t layout_len = b->seg_stride ? n_seg * b->seg_stride : b->total_keys;

    // ---- stage 2 (optional): per-bucket dedup / count / abundance filter ----
    b->cap_log2 = pick_cap_log2(c);
    if (abundance_min > 1 || want_counts) {
        DevBuf &d_flag = b->t_flag;
        HIPCHK(c, d_flag.ensure(4));
        HIPCHK(c, hipMemsetAsync(d_flag.p, 0, 4, s));
        HIPCHK(c, b->d_len.ensure((n_seg + 1) * 4));
        if (want_counts) HIPCHK(c, b->d_kcnt.ensure((layout_len + 2) * 4));
        HIPCHK(c, b->d_marks.ensure((n_seg / 32 + 2) * 4));
        HIPCHK(c, hipMemsetAsync(b->d_marks.p, 0, (n_seg / 32 + 2) * 4, s));
        const SegLayout seg = batch_segments(b, false);        // the lengths the partition left (dense: from the offsets)
        // wave form first (table sized to the expected segment, no barriers); what it marks as too dense goes
        // through the workgroup form with the full-size table
        const uint64_t mean_seg = max_g >> b->bb;
        const uint64_t want = (uint64_t)((double)(mean_seg + 16) + 6.0 * std::sqrt((double)mean_seg + 1.0)) * 5 / 4;
        int wave_cap = 9;
        while (wave_cap < 11 && (1ull << wave_cap) < want) wave_cap++;
        if (c->opt_dedup_cap_shift > 0) wave_cap = std::min(11, wave_cap + c->opt_dedup_cap_shift);
        const bool wave_form = c->opt_dedup_wg <= 0;
        int ov = 0;
        if (wave_form) {
            {
                TimeScope t(c, "bucket_dedup", b->total_keys);
                launch_bucket_dedup_wave(s, b->d_keys.as<uint64_t>(), seg, n_seg, wave_cap, abundance_min, b->d_len.as<uint32_t>(),
                                         b->d_marks.as<uint32_t>(), want_counts ? b->d_kcnt.as<uint32_t>() : nullptr, d_flag.as<int>());
            }
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(&ov, d_flag.p, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
        }
        if (!wave_form || ov) {
            HIPCHK(c, hipMemsetAsync(d_flag.p, 0, 4, s));
            TimeScope t(c, wave_form ? "bucket_dedup_dense" : "bucket_dedup", b->total_keys);
            launch_bucket_dedup(s, b->d_keys.as<uint64_t>(), seg, n_seg, b->cap_log2, abundance_min, b->d_len.as<uint32_t>(),
                                wave_form ? b->d_marks.as<uint32_t>() : nullptr, want_counts ? b->d_kcnt.as<uint32_t>() : nullptr, d_flag.as<int>());
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&ov, d_flag.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (ov) return fail(c, GRM_ERR_OVERFLOW, "bucket_dedup: a (genome,bucket) segment holds more distinct k-mers than the LDS table (cap 2^%u); raise bucket_bits", b->cap_log2);
        b->deduped = true;
    }
    HIPCHK(c, hipStreamSynchronize(s));
    b->partitioned = true;
    return GRM_OK;
}

extern "C" int grm_batch_partition(grm_batch *b, int k, uint32_t abundance_min)
{
    if (!b) return GRM_ERR_ARG;
    b->sorted_stage = false;
    if (k > 64 || (k > 32 && abundance_min > 1)) {
        // three- / four-word k-mers, and two-word ones with an abundance filter: the sort path, which leaves the batch's own
        // dictionary AND its presence bits behind
        int rc = batch_partition_impl(b, k, abundance_min, false);
        if (rc) return rc;
        return sorted_stage_local(b, k, abundance_min < 1 ? 1 : abundance_min);
    }
    if (k > 32 && k <= 64) {
        // two-word k-mers: the hash-partition pipeline, which also leaves the local dictionary behind
        int rc = batch_partition_impl(b, k, abundance_min, false);
        if (rc) return rc;
        return wide_stage_local(b, k);
    }
    return batch_partition_impl(b, k, abundance_min, false);
}

extern "C" int grm_batch_partition_counts(grm_batch *b, int k, uint32_t abundance_min)
{
    if (!b) return GRM_ERR_ARG;
    if (k > 32) return fail(b->ctx, GRM_ERR_UNSUPPORTED, "k=%d: counted batches handle k <= 32; use grm_count_genome per genome", k);
    return batch_partition_impl(b, k, abundance_min, true);
}

extern "C" uint64_t grm_batch_n_symbols(const grm_batch *b) { return b ? b->total_syms : 0; }
extern "C" uint64_t grm_batch_n_occurrences(const grm_batch *b) { return b ? b->total_keys : 0; }
extern "C" uint64_t grm_batch_input_bytes(const grm_batch *b) { return b ? b->input_bytes : 0; }
extern "C" uint64_t grm_batch_n_local(const grm_batch *b) { return b ? b->n_local : 0; }
extern "C" int grm_batch_memo_stats(const grm_batch *b, uint64_t *out4)
{
    if (!b || !out4) return GRM_ERR_ARG;
    for (int i = 0; i < 4; i++) out4[i] = b->memo_stats[i];
    return GRM_OK;
}

// ---- dictionary of the local genomes ------------------------------------------------------
// device control block of dict_build: { n_out u64, overflow i32, need u32 }
struct DictCtrl {
    unsigned long long n_out;
    int overflow;
    uint32_t need;
    unsigned long long memo_stats[4];
};


// One dictionary build with the sizing ladder.  `a` carries the input side (keys, segments, genomes, bb, cap_log2,
// optional rank flags); the output buffers are sized here for every attempt.  want_bits: also keep the presence words
// (matrix_s / birth of the batch).  On success *sb_out / *n_out / *bits_out describe what was built.
struct DictOut {
    DevBuf *keys, *flags, *wg_base, *wg_cnt;
};
// a launch holds fewer than 2^32 threads: at most 2^22 workgroups of 512
static const int DICT_MAX_WG_BITS = 22;

// run_dict_ladder -> grm_batch_local_dict: the record form would need many sub-buckets, each of which decodes every record
// of its bucket again; the key form (sub-bucket workgroups only re-read and skip keys) is the cheaper way then
static const int GRM_INTERNAL_WANT_KEY_FORM = -1000;

// record form: the same records in 2^new_bb buckets -- level 2 again (the records carry more bucket bits than are in use)
static int batch_rebucket(grm_batch *b, int new_bb)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    const uint64_t n_seg = (((uint64_t)b->n_genomes << b->rec_part_bits)) << new_bb;
    // launch contract of level 2: it writes off / len for EVERY segment of the new bucket count and the records of every region.
    // (Round 2's memory fault of 10:25 -- DESIGN.md 7b -- was this launch in its first, uncommitted form: level 2 run again for
    // 2^new_bb buckets into off / len arrays still sized for the old count, a linear overrun that faulted at the first unmapped
    // 2 MiB boundary behind them.)
    HIPCHK(c, b->d_off.ensure((n_seg + 1) * 8));
    HIPCHK(c, b->d_len.ensure((n_seg + 1) * 4));
    HIPCHK(c, b->t_flag.ensure(32));
    if (new_bb < b->rec_b1 || new_bb > b->rec_b1 + 7 || b->d_off.bytes < (n_seg + 1) * 8 || b->d_len.bytes < (n_seg + 1) * 4 ||
        b->d_recs2.bytes < (b->rec_regions * (uint64_t)b->rec_rstride + 4) * 16 || b->d_counts1.bytes < (b->rec_regions + 1) * 4)
        return fail(c, GRM_ERR_STATE, "internal: level 2 asked for 2^%d buckets over buffers sized for fewer", new_bb);
    HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 4, s));
    {
        TimeScope t(c, "superkmer_l2_again", 0);
        launch_superkmer_l2_records(s, b->d_recs.p, b->rec_rstride, b->d_counts1.as<uint32_t>(), b->rec_regions, b->k, new_bb, b->rec_b1, b->d_recs2.p,
                                    b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), b->t_flag.as<int>());
    }
    HIPCHK(c, hipGetLastError());
    b->bb = new_bb;
    return GRM_OK;
}
static int run_dict_ladder(grm_batch *b, DictArgs a, uint64_t total_keys, int sb, bool want_bits, const DictOut &out, const char *tname,
                           int *sb_out, uint64_t *n_out, bool *bits_out, int *launches)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    const uint32_t cap = 1u << a.cap_log2, max_fill = cap - (cap >> 3);
    const size_t n_rows = ((size_t)a.n_genomes + 63) / 64;
    HIPCHK(c, b->d_ctrl.ensure(sizeof(DictCtrl)));
    for (int attempt = 0;; attempt++) {
        if (a.bb + sb > DICT_MAX_WG_BITS) return fail(c, GRM_ERR_OVERFLOW, "%s: bucket union does not fit the LDS table even with 2^%d sub-buckets", tname, sb);
        const uint32_t n_wg = 1u << (a.bb + sb);
        // every workgroup holds at most max_fill entries, and there are no more entries than keys
        const uint64_t out_cap = std::min<uint64_t>(total_keys, (uint64_t)n_wg * max_fill) + 64;
        HIPCHK(c, out.keys->ensure(out_cap * 8));
        HIPCHK(c, out.flags->ensure(out_cap));
        HIPCHK(c, out.wg_base->ensure((size_t)n_wg * 8));
        HIPCHK(c, out.wg_cnt->ensure(((size_t)n_wg + 1) * 4));
        const size_t ms_bytes = (size_t)n_wg * n_rows * cap * 8;
        const bool bits = want_bits && ms_bytes <= MATRIX_S_LIMIT;
        if (bits) {
            HIPCHK(c, b->d_matrix_s.ensure(ms_bytes));
            HIPCHK(c, b->d_birth.ensure((size_t)n_wg * cap * 2));
        }
        // launch contract: one workgroup per (bucket, sub-bucket) and fewer than 2^32 threads per launch; every array the kernel
        // indexes by workgroup or by (virtual genome, bucket) covers THIS bucket count
        const uint64_t n_seg_in = ((uint64_t)a.n_genomes << a.part_bits) << a.bb;
        if (a.bb + sb > DICT_MAX_WG_BITS || out.wg_base->bytes < (size_t)n_wg * 8 || out.wg_cnt->bytes < ((size_t)n_wg + 1) * 4 ||
            out.keys->bytes < out_cap * 8 || out.flags->bytes < out_cap ||
            (bits && (b->d_matrix_s.bytes < ms_bytes || b->d_birth.bytes < (size_t)n_wg * cap * 2)) ||
            (a.seg.off == b->d_off.as<uint64_t>() && a.seg.off && b->d_off.bytes < n_seg_in * 8) ||
            (a.seg.len == b->d_len.as<uint32_t>() && a.seg.len && b->d_len.bytes < n_seg_in * 4))
            return fail(c, GRM_ERR_STATE, "internal: %s launch of 2^%d workgroups over buffers sized for fewer", tname, a.bb + sb);
        HIPCHK(c, hipMemsetAsync(b->d_ctrl.p, 0, sizeof(DictCtrl), s));
        a.sb = sb;
        a.out_keys = out.keys->as<uint64_t>(); a.out_flags = out.flags->as<uint8_t>();
        a.out_cap = out_cap;
        DictCtrl *ctrl = b->d_ctrl.as<DictCtrl>();
        a.n_out = &ctrl->n_out; a.overflow = &ctrl->overflow; a.need = &ctrl->need;
        static const bool memo_env = getenv("GRM_MEMO_STATS") != nullptr;
        const bool memo_diag = memo_env || c->opt_memo_stats > 0;
        a.memo_stats = memo_diag ? ctrl->memo_stats : nullptr;
        a.wg_base = out.wg_base->as<uint64_t>(); a.wg_cnt = out.wg_cnt->as<uint32_t>();
        a.matrix_s = bits ? b->d_matrix_s.as<uint64_t>() : nullptr;
        a.birth = bits ? b->d_birth.as<uint16_t>() : nullptr;
        {
            TimeScope t(c, tname, total_keys);
            launch_dict_build(s, a);
        }
        if (launches) (*launches)++;
        HIPCHK(c, hipGetLastError());
        DictCtrl h;
        HIPCHK(c, hipMemcpyAsync(&h, b->d_ctrl.p, sizeof h, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (memo_diag) for (int i = 0; i < 4; i++) b->memo_stats[i] = a.memo_log2 ? h.memo_stats[i] : 0;
        if (memo_env) fprintf(stderr, "[grm] %s launch %d: 2^%d buckets, 2^%d sub-buckets, overflow %d, need %u (usable %u)\n", tname, attempt, a.bb, sb, h.overflow, h.need, max_fill);
        if (memo_env && a.memo_log2)
            fprintf(stderr, "[grm] %s memo 2^%d: %llu (workgroup, row) ends with the memo on, mean records held %.1f, occurrences asked %llu found %llu; overflow %d\n",
                    tname, a.memo_log2, h.memo_stats[3], h.memo_stats[3] ? (double)h.memo_stats[0] / (double)h.memo_stats[3] : 0.0, h.memo_stats[1],
                    h.memo_stats[2], h.overflow);
        if (!h.overflow) {
            *sb_out = sb;
            *n_out = h.n_out;
            *bits_out = bits;
            return GRM_OK;
        }
        if (h.overflow >= 2 && h.need == 0) return fail(c, GRM_ERR_HIP, "internal: %s output capacity exceeded (%llu entries)", tname, (unsigned long long)h.n_out);
        if (attempt >= 8 || (c->opt_sub_bits >= 0 && sb >= c->opt_sub_bits + 8)) return fail(c, GRM_ERR_OVERFLOW, "%s: overflow persists", tname);
        // jump to the sub-bucket count the failed launch asks for: its fullest workgroup estimated `need`
        // distinct k-mers; aim at 70 % of the usable table so that the estimate's error does not cost a third launch
        int step = 1;
        while (step < 24 && ((uint64_t)h.need >> step) > (uint64_t)max_fill * 7 / 10) step++;
        bool more_buckets = false;
        if (a.recs && b->rec_dict && c->opt_sub_bits < 0 && c->opt_bucket_bits < 0) {
            // record form: every sub-bucket workgroup would decode ALL records of its bucket, so more BUCKETS come first
            // (level 2 again, 6 ms) as far as the records' bucket bits and the presence words allow
            // (a bucket's k-mers follow their minimizers into the finer buckets, a few dozen minimizers of unequal weight per bucket:
            // the fullest of the new buckets gets more than its even share, so aim at half the usable table, not at 70 %)
            int step_b = 1;
            while (step_b < 24 && ((uint64_t)h.need >> step_b) > (uint64_t)max_fill / 2) step_b++;
            int more = std::min(step_b, std::min(superkmer_max_bits(), b->rec_b1 + 7) - a.bb);
            if (step > std::min(superkmer_max_bits(), b->rec_b1 + 7) - a.bb + 2) more = 0;      // hopeless: straight to the key form
            while (more > 0 && ((size_t)1 << (a.bb + more + sb)) * n_rows * cap * 8 > MATRIX_S_LIMIT) more--;
            if (more > 0) {
                int rc = batch_rebucket(b, a.bb + more);
                if (rc) return rc;
                a.bb = b->bb;
                a.seg.off = b->d_off.as<uint64_t>();
                a.seg.len = b->d_len.as<uint32_t>();
                // `need` extrapolates the fill linearly over the genomes, far too much for a pan-genome whose distinct
                // k-mers saturate early: the new buckets are tried before any sub-bucket is added
                step = 0;
                more_buckets = true;
            } else if (sb + step > 2) {
                b->rec_need = h.need;                    // (of the fullest of 2^a.bb minimizer buckets)
                b->rec_need_bb = a.bb;
                return GRM_INTERNAL_WANT_KEY_FORM;       // no bucket bits left and more than 4 sub-buckets asked for: see the caller
            }
        }
        if (a.bb + sb + step > DICT_MAX_WG_BITS) step = DICT_MAX_WG_BITS - a.bb - sb;      // as many sub-buckets as a launch allows before giving up
        if (step <= 0 && !more_buckets) return fail(c, GRM_ERR_OVERFLOW, "%s: bucket union does not fit the LDS table even with 2^%d sub-buckets", tname, sb);
        sb += step;
    }
}

// record form: key segments from the level-1 records, for a consumer that cannot decode records
static int batch_expand_keys(grm_batch *b)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    HIPCHK(c, b->d_keys.ensure((b->rec_regions * b->rec_kstride + 4) * 8));
    HIPCHK(c, b->t_flag.ensure(16));
    HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 16, s));
    {
        TimeScope t(c, "superkmer_l2_keys", b->total_syms);
        launch_superkmer_l2(s, b->d_recs.p, b->rec_rstride, b->d_counts1.as<uint32_t>(), b->rec_regions, b->k, b->bb, b->rec_b1, b->rec_kstride,
                            b->d_keys.as<uint64_t>(), b->d_off.as<uint64_t>(), b->d_len.as<uint32_t>(), b->t_flag.as<int>());
    }
    HIPCHK(c, hipGetLastError());
    int over = 0;
    HIPCHK(c, hipMemcpyAsync(&over, b->t_flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (over) return fail(c, GRM_ERR_OVERFLOW, "record form: a region holds more k-mers than its key capacity");
    b->rec_dict = false;
    return GRM_OK;
}

extern "C" int grm_batch_local_dict(grm_batch *b, uint64_t *n_local)
{
    if (!b) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->partitioned) return fail(c, GRM_ERR_STATE, "grm_batch_local_dict before grm_batch_partition");
    if (b->sorted_stage) {
        if (n_local) *n_local = b->n_local;
        return GRM_OK;
    }
    if (b->k > 32) return wide_stage_n_local(b, n_local);
    HIPCHK(c, hipSetDevice(c->device));
    b->have_local = b->have_global = false;
    b->exported_ordered = false;
    b->have_bits = false;
    b->dict_launches = 0;
    const uint32_t G = (uint32_t)b->n_genomes;
    if (b->total_keys == 0) {
        b->n_local = 0;
        HIPCHK(c, b->d_local_keys.ensure(16));
        HIPCHK(c, b->d_local_flags.ensure(16));
        b->have_local = true;
        if (n_local) *n_local = 0;
        return GRM_OK;
    }
    if (((size_t)G + 63) / 64 > 0xffffu) return fail(c, GRM_ERR_UNSUPPORTED, "%u genomes in one batch (limit 65535 word-rows); use the chunked flow", G);
    // sub-buckets: what the caller forces, else what this batch needed last time for the same geometry, else none
    int sb = c->opt_sub_bits >= 0 ? c->opt_sub_bits : 0;
    if (c->opt_sub_bits < 0 && b->sb_hint >= 0 && b->sb_hint_k == b->k && b->sb_hint_bb == b->bb && b->sb_hint_amin == b->abundance_min)
        sb = b->sb_hint;
    DictArgs a;
    memset(&a, 0, sizeof a);
    a.keys = b->rec_dict ? nullptr : b->d_keys.as<uint64_t>();
    a.recs = b->rec_dict ? reinterpret_cast<const ulonglong2 *>(b->d_recs2.p) : nullptr;
    a.k = b->k;
    a.memo_log2 = b->rec_dict ? b->rec_memo_log2 : 0;
    a.part_bits = b->rec_mode ? b->rec_part_bits : 0;
    a.seg = batch_segments(b);
    a.n_genomes = G; a.bb = b->bb; a.cap_log2 = b->cap_log2;
    const DictOut out = {&b->d_local_keys, &b->d_local_flags, &b->d_wg_base, &b->d_wg_cnt};
    int rc = run_dict_ladder(b, a, b->total_keys, sb, c->opt_no_slots <= 0, out, "dict_build", &sb, &b->n_local, &b->have_bits, &b->dict_launches);
    if (rc == GRM_INTERNAL_WANT_KEY_FORM) {
        b->rec_failed = true;                // this batch holds too many distinct k-mers per minimizer bucket: key form from now on
        rc = batch_partition_impl(b, b->k, b->abundance_min, false);
        if (rc) return rc;
        if (c->opt_sub_bits < 0) {
            // what the failed launch learnt sizes the key form's first attempt: its fullest minimizer bucket held about twice an
            // average one, and hashed buckets are even
            const uint64_t per_bucket = b->rec_need_bb >= b->bb ? ((uint64_t)b->rec_need << (b->rec_need_bb - b->bb)) / 2
                                                                : ((uint64_t)b->rec_need >> (b->bb - b->rec_need_bb)) / 2;
            const uint32_t cap = 1u << b->cap_log2, max_fill = cap - (cap >> 3);
            int s0 = 0;
            while (s0 < DICT_MAX_WG_BITS - b->bb && (per_bucket >> s0) > (uint64_t)max_fill * 7 / 10) s0++;
            b->sb_hint = s0; b->sb_hint_k = b->k; b->sb_hint_bb = b->bb; b->sb_hint_amin = b->abundance_min;
        }
        return grm_batch_local_dict(b, n_local);
    }
    if (rc) return rc;
    if (b->rec_mode && !b->have_bits) {
        // the probing form of the fill reads key segments, genome by genome
        if (b->rec_part_bits > 0) {
            b->rec_failed = true;            // ... of whole genomes: partition again in the key form and stay there
            rc = batch_partition_impl(b, b->k, b->abundance_min, false);
            if (rc) return rc;
            return grm_batch_local_dict(b, n_local);
        }
        if (b->rec_dict) {
            rc = batch_expand_keys(b);
            if (rc) return rc;
        }
    }
    b->sb_dict = sb;
    if (b->rec_mode && c->opt_bucket_bits < 0) { b->rec_bb_hint = b->bb; b->rec_bb_hint_k = b->k; }
    if (c->opt_sub_bits < 0) { b->sb_hint = sb; b->sb_hint_k = b->k; b->sb_hint_bb = b->bb; b->sb_hint_amin = b->abundance_min; }
    b->have_local = true;
    if (n_local) *n_local = b->n_local;
    return GRM_OK;
}

extern "C" int grm_batch_export_dict(grm_batch *b, void *dev_keys_out, void *dev_flags_out)
{
    if (!b) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->have_local) return fail(c, GRM_ERR_STATE, "grm_batch_export_dict before grm_batch_local_dict");
    HIPCHK(c, hipSetDevice(c->device));
    if (b->sorted_stage) return sorted_stage_export(b, dev_keys_out, dev_flags_out);
    if (b->k > 32) return wide_stage_export(b, dev_keys_out, dev_flags_out);
    if (b->n_local) {
        HIPCHK(c, hipMemcpyAsync(dev_keys_out, b->d_local_keys.p, b->n_local * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dev_flags_out, b->d_local_flags.p, b->n_local, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GRM_OK;
}

// bucketise the global dictionary for matrix_fill with 2^sb sub-buckets
static int bucketise_dict(grm_batch *b, int sb)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    const uint64_t U = b->n_dict;
    const uint32_t n_wg = 1u << (b->bb + sb);
    HIPCHK(c, b->d_seg_start.ensure(((size_t)n_wg + 2) * 8));
    HIPCHK(c, b->d_dkeys.ensure((U + 2) * 8));
    HIPCHK(c, b->d_dcol.ensure((U + 2) * 4));
    if (U == 0) {
        HIPCHK(c, hipMemsetAsync(b->d_seg_start.p, 0, ((size_t)n_wg + 2) * 8, s));
        b->sb_fill = sb;
        return GRM_OK;
    }
    DevBuf &d_bid = b->t_bid, &d_col = b->t_col, &d_bid_sorted = b->t_bid_sorted, &d_tmp = b->t_tmp;
    HIPCHK(c, d_bid.ensure(U * 4));
    HIPCHK(c, d_col.ensure(U * 4));
    HIPCHK(c, d_bid_sorted.ensure(U * 4));
    TimeScope t(c, "dict_bucketise", U);
    if (b->rec_mode) launch_minimizer_bucket_ids(s, b->d_dict.as<uint64_t>(), U, b->k, b->bb, sb, d_bid.as<uint32_t>(), d_col.as<uint32_t>());
    else launch_dict_bucket_ids(s, b->d_dict.as<uint64_t>(), U, b->bb, sb, d_bid.as<uint32_t>(), d_col.as<uint32_t>());
    size_t tmp_bytes = 0;
    HIPCHK(c, sort_pairs_u32_u32(s, d_bid.as<uint32_t>(), d_bid_sorted.as<uint32_t>(), d_col.as<uint32_t>(),
                                 b->d_dcol.as<uint32_t>(), U, b->bb + sb, nullptr, tmp_bytes));
    HIPCHK(c, d_tmp.ensure(tmp_bytes));
    HIPCHK(c, sort_pairs_u32_u32(s, d_bid.as<uint32_t>(), d_bid_sorted.as<uint32_t>(), d_col.as<uint32_t>(),
                                 b->d_dcol.as<uint32_t>(), U, b->bb + sb, d_tmp.p, tmp_bytes));
    launch_gather_u64(s, b->d_dict.as<uint64_t>(), b->d_dcol.as<uint32_t>(), U, b->d_dkeys.as<uint64_t>());
    launch_segment_starts(s, d_bid_sorted.as<uint32_t>(), U, n_wg, b->d_seg_start.as<uint64_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    b->sb_fill = sb;
    return GRM_OK;
}

// (key, flag) entries -- several per k-mer when it was reported by several ranks or chunks -- -> the sorted,
// merged, filtered global dictionary b->d_dict / b->n_dict
static int dict_from_entries(grm_batch *b, const uint64_t *keys, const uint8_t *flags, uint64_t n, int filter_singleton)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    b->n_dict = 0;
    if (!n) return b->d_dict.ensure(16) == hipSuccess ? GRM_OK : fail(c, GRM_ERR_OOM, "alloc");
    if (n >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "dictionary of %llu k-mers exceeds 2^32-1 columns", (unsigned long long)n);
    DevBuf &d_sk = b->t_sk, &d_sf = b->t_sf, &d_keep = b->t_keep, &d_pos = b->t_pos, &d_tmp = b->t_tmp;
    HIPCHK(c, d_sk.ensure(n * 8));
    HIPCHK(c, d_sf.ensure(n));
    HIPCHK(c, d_keep.ensure((n + 1) * 4));
    HIPCHK(c, d_pos.ensure((n + 1) * 8));
    for (int attempt = 0; attempt < 2; attempt++) {
        // key-range sort with the entries' indices (the flags follow through them), or the general sort of (key, flag) pairs
        const bool by_ranges = attempt == 0 && c->opt_dict_sort_prim <= 0;
        if (attempt == 0 && !by_ranges) continue;
        TimeScope t(c, "dict_sort", n);
        HIPCHK(c, b->t_flag.ensure(32));
        HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 4, s));
        if (by_ranges) {
            HIPCHK(c, b->t_col.ensure(n * 4));
            HIPCHK(c, d_tmp.ensure(dict_sort_scratch_bytes(n)));
            HIPCHK(c, launch_dict_sort(s, keys, n, 2 * b->k, d_sk.as<uint64_t>(), b->t_col.as<uint32_t>(), d_tmp.p, b->t_flag.as<int>()));
            launch_gather_u8(s, flags, b->t_col.as<uint32_t>(), n, d_sf.as<uint8_t>());
        } else {
            size_t tmp_bytes = 0;
            HIPCHK(c, sort_pairs_u64_u8(s, keys, d_sk.as<uint64_t>(), flags, d_sf.as<uint8_t>(), n, nullptr, tmp_bytes));
            HIPCHK(c, d_tmp.ensure(tmp_bytes));
            HIPCHK(c, sort_pairs_u64_u8(s, keys, d_sk.as<uint64_t>(), flags, d_sf.as<uint8_t>(), n, d_tmp.p, tmp_bytes));
        }
        HIPCHK(c, hipMemsetAsync(d_keep.as<uint32_t>() + n, 0, 4, s));
        launch_dict_mark(s, d_sk.as<uint64_t>(), d_sf.as<uint8_t>(), n, filter_singleton, d_keep.as<uint32_t>());
        size_t tmp2 = 0;
        HIPCHK(c, exclusive_scan_u32_u64(s, d_keep.as<uint32_t>(), d_pos.as<uint64_t>(), n + 1, nullptr, tmp2));
        HIPCHK(c, b->t_set_tmp.ensure(tmp2));
        HIPCHK(c, exclusive_scan_u32_u64(s, d_keep.as<uint32_t>(), d_pos.as<uint64_t>(), n + 1, b->t_set_tmp.p, tmp2));
        int too_big = 0;
        HIPCHK(c, hipMemcpyAsync(&b->n_dict, d_pos.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&too_big, b->t_flag.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (by_ranges && too_big) continue;
        HIPCHK(c, b->d_dict.ensure((b->n_dict + 2) * 8));
        launch_dict_select(s, d_sk.as<uint64_t>(), d_keep.as<uint32_t>(), d_pos.as<uint64_t>(), n, b->d_dict.as<uint64_t>());
        break;
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    return GRM_OK;
}

// DISTINCT entries (one GPU: the batch's own local dictionary; several ranks: their union out of the LDS tables): they are
// sorted together with their entry index and every entry's column falls out of the select step -- col_of_entry[i] = column
// of entry i, 0xffffffff when filtered out -- no search afterwards
static int dict_from_distinct_entries(grm_batch *b, const uint64_t *keys, const uint8_t *flags, uint64_t n, int filter_singleton, uint32_t *col_of_entry)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    b->n_dict = 0;
    if (n >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "dictionary of %llu k-mers exceeds 2^32-1 columns", (unsigned long long)n);
    if (!n) return b->d_dict.ensure(16) == hipSuccess ? GRM_OK : fail(c, GRM_ERR_OOM, "alloc");
    DevBuf &d_sk = b->t_sk, &d_keep = b->t_keep, &d_pos = b->t_pos, &d_tmp = b->t_tmp, &d_i0 = b->t_bid, &d_i1 = b->t_col;
    HIPCHK(c, d_sk.ensure(n * 8));
    HIPCHK(c, d_keep.ensure((n + 1) * 4));
    HIPCHK(c, d_pos.ensure((n + 1) * 8));
    HIPCHK(c, d_i0.ensure(n * 4));
    HIPCHK(c, d_i1.ensure(n * 4));
    for (int attempt = 0; attempt < 2; attempt++) {
        // key-range sort (grm_dictsort.hip); the general radix sort when asked for, or when a key range did not fit LDS
        const bool by_ranges = attempt == 0 && c->opt_dict_sort_prim <= 0;
        if (attempt == 0 && !by_ranges) continue;
        TimeScope t(c, "dict_sort", n);
        HIPCHK(c, b->t_flag.ensure(32));
        HIPCHK(c, hipMemsetAsync(b->t_flag.p, 0, 4, s));
        if (by_ranges) {
            HIPCHK(c, d_tmp.ensure(dict_sort_scratch_bytes(n)));
            HIPCHK(c, launch_dict_sort(s, keys, n, 2 * b->k, d_sk.as<uint64_t>(), d_i1.as<uint32_t>(), d_tmp.p, b->t_flag.as<int>()));
        } else {
            launch_iota_u32(s, d_i0.as<uint32_t>(), n);
            size_t tmp_bytes = 0;
            HIPCHK(c, sort_pairs_u64_u32(s, keys, d_sk.as<uint64_t>(), d_i0.as<uint32_t>(), d_i1.as<uint32_t>(), n, nullptr, tmp_bytes));
            HIPCHK(c, d_tmp.ensure(tmp_bytes));
            HIPCHK(c, sort_pairs_u64_u32(s, keys, d_sk.as<uint64_t>(), d_i0.as<uint32_t>(), d_i1.as<uint32_t>(), n, d_tmp.p, tmp_bytes));
        }
        HIPCHK(c, hipMemsetAsync(d_keep.as<uint32_t>() + n, 0, 4, s));
        launch_dict_mark_idx(s, flags, d_i1.as<uint32_t>(), n, filter_singleton, d_keep.as<uint32_t>());
        size_t tmp2 = 0;
        HIPCHK(c, exclusive_scan_u32_u64(s, d_keep.as<uint32_t>(), d_pos.as<uint64_t>(), n + 1, nullptr, tmp2));
        // (the scan's scratch must not be the sort's: d_tmp may be re-allocated here only after the sort has run -- same stream, in order)
        HIPCHK(c, b->t_set_tmp.ensure(tmp2));
        HIPCHK(c, exclusive_scan_u32_u64(s, d_keep.as<uint32_t>(), d_pos.as<uint64_t>(), n + 1, b->t_set_tmp.p, tmp2));
        int too_big = 0;
        HIPCHK(c, hipMemcpyAsync(&b->n_dict, d_pos.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&too_big, b->t_flag.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (by_ranges && too_big) continue;          // (keys crowding under one prefix: a degenerate input)
        HIPCHK(c, b->d_dict.ensure((b->n_dict + 2) * 8));
        launch_dict_select_idx(s, d_sk.as<uint64_t>(), d_keep.as<uint32_t>(), d_pos.as<uint64_t>(), d_i1.as<uint32_t>(), n, b->d_dict.as<uint64_t>(),
                               col_of_entry);
        break;
    }
    HIPCHK(c, hipGetLastError());
    return GRM_OK;
}

static int dict_from_own_entries(grm_batch *b, int filter_singleton)
{
    HIPCHK(b->ctx, b->d_entry_col.ensure((b->n_local + 1) * 4));
    int rc = dict_from_distinct_entries(b, b->d_local_keys.as<uint64_t>(), b->d_local_flags.as<uint8_t>(), b->n_local, filter_singleton,
                                        b->d_entry_col.as<uint32_t>());
    if (rc == GRM_OK) b->entry_cols_ready = true;
    return rc;
}

// the batch's own structures against b->d_dict: columns of its entries (fused form) or the bucketised dictionary
static int dict_attach(grm_batch *b, uint64_t *n_kmers)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    b->fill_by_bits = false;
    if (b->have_local && b->have_bits && b->total_keys) {
        // fused form: every local entry learns its global column (or that it was filtered out)
        HIPCHK(c, b->d_entry_col.ensure((b->n_local + 1) * 4));
        if (!b->entry_cols_ready) {
            TimeScope t(c, "dict_entry_cols", b->n_local);
            HIPCHK(c, b->d_prefix.ensure(((size_t)1 << 22) * 4 + 16));
            launch_dict_entry_cols(s, b->d_dict.as<uint64_t>(), b->n_dict, b->d_local_keys.as<uint64_t>(), b->n_local, b->k,
                                   b->d_prefix.as<uint32_t>(), b->d_entry_col.as<uint32_t>());
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(s));
        b->fill_by_bits = true;
        b->have_global = true;
        if (n_kmers) *n_kmers = b->n_dict;
        return GRM_OK;
    }
    // probing form.  sub-bucket count for the fill: keep the mean dictionary slice under 1/4 of the table
    int sb = c->opt_sub_bits >= 0 ? c->opt_sub_bits : 0;
    const uint64_t cap = 1ull << b->cap_log2;
    while (b->bb + sb < DICT_MAX_WG_BITS && (b->n_dict >> (b->bb + sb)) > cap / 4) sb++;
    int rc = bucketise_dict(b, sb);
    if (rc) return rc;
    b->have_global = true;
    if (n_kmers) *n_kmers = b->n_dict;
    return GRM_OK;
}

extern "C" int grm_batch_set_global_dict(grm_batch *b, const void *dev_keys, const void *dev_flags, uint64_t n,
                                         int filter_singleton, uint64_t *n_kmers)
{
    if (!b) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->partitioned) return fail(c, GRM_ERR_STATE, "grm_batch_set_global_dict before grm_batch_partition");
    if (n && (!dev_keys || !dev_flags)) return fail(c, GRM_ERR_ARG, "grm_batch_set_global_dict: NULL buffers");
    if (n >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "dictionary of %llu k-mers exceeds 2^32-1 columns", (unsigned long long)n);
    HIPCHK(c, hipSetDevice(c->device));
    if (b->sorted_stage) return sorted_stage_global(b, dev_keys, dev_flags, n, filter_singleton, n_kmers);
    if (b->k > 32) return wide_stage_global(b, dev_keys, dev_flags, n, filter_singleton, n_kmers);
    b->have_global = false;
    b->filter_singleton = filter_singleton;
    b->own_dict = n && dev_keys == b->d_local_keys.p && n == b->n_local;     // filtered or not: every COLUMN stems from a local entry
    b->entry_cols_ready = false;
    int rc;
    if (b->own_dict && b->have_bits && dev_flags == b->d_local_flags.p) rc = dict_from_own_entries(b, filter_singleton);
    else rc = dict_from_entries(b, (const uint64_t *)dev_keys, (const uint8_t *)dev_flags, n, filter_singleton);
    if (rc) return rc;
    return dict_attach(b, n_kmers);
}

// ---- multi-GPU exchange: ONE all-gather of fixed-stride records ---------------------------------------------
// record of a rank (n_max = largest n_local of any rank, B = 2^bucket_bits):
//   [0, n_max * 8 * words)            keys, grouped by hash bucket (ascending), (hi, lo) pairs for words = 2
//   [flags_off, flags_off + n_max)    flags
//   [boff_off, boff_off + 4 (B + 1))  first entry of every hash bucket (uint32), boff[B] = n_local
//   [stride - 16, stride)             header, written by grm_batch_export_dict_record: n_local (uint64), bucket-bits code (uint32,
//                                     grm_batch_bucket_bits), GRM_EXCHANGE_MAGIC (uint32) -- what the ranks used to tell each other in
//                                     a collective of its own before the layout could be fixed
extern "C" void grm_exchange_layout(uint64_t n_max, int words, int bucket_bits, uint64_t *flags_off, uint64_t *boff_off, uint64_t *stride)
{
    const uint64_t fo = n_max * 8 * (uint64_t)(words < 1 ? 1 : words);
    const uint64_t bo = (fo + n_max + 15) / 16 * 16;
    const uint64_t st = (bo + (((uint64_t)1 << (bucket_bits & 0xff)) + 1) * 4 + 15) / 16 * 16 + GRM_EXCHANGE_HEADER_BYTES;
    if (flags_off) *flags_off = fo;
    if (boff_off) *boff_off = bo;
    if (stride) *stride = st;
}
// bucket geometry of the batch as ranks compare it: bucket bits, + 0x100 when the buckets are minimizer buckets
// (record form) -- lists of ranks with different codes cannot be united bucket by bucket
extern "C" int grm_batch_bucket_bits(const grm_batch *b) { return !b || b->sorted_stage ? 0 : (b->bb | (b->rec_mode ? 0x100 : 0)); }

extern "C" int grm_batch_export_dict_ordered(grm_batch *b, void *dev_record, uint64_t flags_off, uint64_t boff_off)
{
    if (!b || !dev_record) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->have_local) return fail(c, GRM_ERR_STATE, "grm_batch_export_dict_ordered before grm_batch_local_dict");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    uint8_t *rec = (uint8_t *)dev_record;
    if (b->sorted_stage) {
        // a sorted list has no hash buckets: one bucket holds it all (bucket bits 0; lists of several words are never united bucket by bucket)
        const uint32_t boff[2] = {0u, (uint32_t)b->n_local};
        int rc = sorted_stage_export(b, rec, rec + flags_off);
        if (rc) return rc;
        HIPCHK(c, hipMemcpy(rec + boff_off, boff, sizeof boff, hipMemcpyHostToDevice));
        return GRM_OK;
    }
    if (b->k > 32) return wide_stage_export_ordered(b, rec, flags_off, boff_off);
    const uint32_t B = 1u << b->bb;
    if (b->total_keys == 0 || b->n_local == 0) {
        HIPCHK(c, hipMemsetAsync(rec + boff_off, 0, ((size_t)B + 1) * 4, s));
        HIPCHK(c, hipStreamSynchronize(s));
        return GRM_OK;
    }
    const uint32_t n_wg = 1u << (b->bb + b->sb_dict);
    HIPCHK(c, b->t_ord_off.ensure(((size_t)n_wg + 2) * 8));
    HIPCHK(c, hipMemsetAsync(b->d_wg_cnt.as<uint32_t>() + n_wg, 0, 4, s));
    size_t tb = 0;
    HIPCHK(c, exclusive_scan_u32_u64(s, b->d_wg_cnt.as<uint32_t>(), b->t_ord_off.as<uint64_t>(), (uint64_t)n_wg + 1, nullptr, tb));
    HIPCHK(c, b->t_tmp.ensure(tb));
    HIPCHK(c, exclusive_scan_u32_u64(s, b->d_wg_cnt.as<uint32_t>(), b->t_ord_off.as<uint64_t>(), (uint64_t)n_wg + 1, b->t_tmp.p, tb));
    launch_dict_export_ordered(s, b->d_local_keys.as<uint64_t>(), b->d_local_flags.as<uint8_t>(), b->d_wg_base.as<uint64_t>(),
                               b->d_wg_cnt.as<uint32_t>(), b->t_ord_off.as<uint64_t>(), n_wg, b->sb_dict, (uint64_t *)rec, rec + flags_off,
                               (uint32_t *)(rec + boff_off));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    b->exported_ordered = true;          // t_ord_off describes the list just written (tracked through the union by set_global_dict_gathered_from)
    return GRM_OK;
}

// The record of a step whose layout was fixed BEFORE the ranks knew each other's sizes (n_cap, bucket_bits: what the previous step
// saw, with some slack): the header always goes out; the lists only when they fit -- a rank that overflows says so through its header,
// every rank reads every header after the all-gather, and all of them repeat the step with the larger layout.
extern "C" int grm_batch_export_dict_record(grm_batch *b, void *dev_record, uint64_t n_cap, int bucket_bits, int *fits)
{
    if (!b || !dev_record) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->have_local) return fail(c, GRM_ERR_STATE, "grm_batch_export_dict_record before grm_batch_local_dict");
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t flags_off, boff_off, stride;
    grm_exchange_layout(n_cap, words_of(b->k), bucket_bits, &flags_off, &boff_off, &stride);
    const bool ok = b->n_local <= n_cap && (grm_batch_bucket_bits(b) & 0xff) <= (bucket_bits & 0xff);
    if (fits) *fits = ok ? 1 : 0;
    struct { uint64_t n_local; uint32_t code, magic; } head = {b->n_local, (uint32_t)grm_batch_bucket_bits(b), GRM_EXCHANGE_MAGIC};
    static_assert(sizeof head == GRM_EXCHANGE_HEADER_BYTES, "exchange header");
    HIPCHK(c, hipMemcpyAsync((uint8_t *)dev_record + stride - GRM_EXCHANGE_HEADER_BYTES, &head, sizeof head, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // (head is a stack variable; the export below ends with a synchronize of its own)
    if (!ok) return GRM_OK;
    return grm_batch_export_dict_ordered(b, dev_record, flags_off, boff_off);
}

// dev_payload: the records of all ranks (rank r at r * stride, grm_exchange_layout(n_max, words, max bucket_bits)).
// counts / bucket_bits: host arrays, one entry per rank.  When every rank used the same bucket geometry the gathered
// lists are first united bucket by bucket in LDS tables (the ranks of a pan-genome hold nearly the same k-mers, so
// what has to be sorted shrinks from the sum of the rank dictionaries to their union); otherwise they are sorted as a whole.
static int set_global_dict_gathered(grm_batch *b, const void *dev_payload, int n_ranks, int my_rank, uint64_t n_max, const uint64_t *counts,
                                    const int *bucket_bits, int filter_singleton, uint64_t *n_kmers);
extern "C" int grm_batch_set_global_dict_gathered(grm_batch *b, const void *dev_payload, int n_ranks, uint64_t n_max,
                                                  const uint64_t *counts, const int *bucket_bits, int filter_singleton, uint64_t *n_kmers)
{
    return set_global_dict_gathered(b, dev_payload, n_ranks, -1, n_max, counts, bucket_bits, filter_singleton, n_kmers);
}
extern "C" int grm_batch_set_global_dict_gathered_from(grm_batch *b, const void *dev_payload, int n_ranks, int my_rank, uint64_t n_max,
                                                       const uint64_t *counts, const int *bucket_bits, int filter_singleton, uint64_t *n_kmers)
{
    return set_global_dict_gathered(b, dev_payload, n_ranks, my_rank, n_max, counts, bucket_bits, filter_singleton, n_kmers);
}
static int set_global_dict_gathered(grm_batch *b, const void *dev_payload, int n_ranks, int my_rank, uint64_t n_max, const uint64_t *counts,
                                    const int *bucket_bits, int filter_singleton, uint64_t *n_kmers)
{
    if (!b || n_ranks < 1 || !counts || !bucket_bits) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->partitioned) return fail(c, GRM_ERR_STATE, "grm_batch_set_global_dict_gathered before grm_batch_partition");
    if (!dev_payload) return fail(c, GRM_ERR_ARG, "grm_batch_set_global_dict_gathered: NULL payload");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int words = words_of(b->k);
    int bb_max = 0, bb_min = 255;
    bool same_kind = true;                 // hashed buckets or minimizer buckets, the same on every rank
    uint64_t total = 0;
    for (int r = 0; r < n_ranks; r++) {
        bb_max = std::max(bb_max, bucket_bits[r] & 0xff);
        bb_min = std::min(bb_min, bucket_bits[r] & 0xff);
        same_kind = same_kind && (bucket_bits[r] & ~0xff) == (bucket_bits[0] & ~0xff);
        if (counts[r] > n_max) return fail(c, GRM_ERR_ARG, "grm_batch_set_global_dict_gathered: rank %d holds more than n_max entries", r);
        total += counts[r];
    }
    uint64_t flags_off, boff_off, stride;
    grm_exchange_layout(n_max, words, bb_max, &flags_off, &boff_off, &stride);
    const uint8_t *payload = (const uint8_t *)dev_payload;
    if (words == 1 && same_kind && n_ranks <= 63 && total && c->opt_no_union <= 0) {
        // ranks may have ended with different bucket counts (a rank whose tables overflowed took more): the union runs over
        // the coarsest, in which every rank's buckets nest
        RankShifts shifts;
        memset(&shifts, 0, sizeof shifts);
        for (int r = 0; r < n_ranks; r++) shifts.d[r] = (uint8_t)((bucket_bits[r] & 0xff) - bb_min);
        const uint32_t B = 1u << bb_min;
        const uint64_t n_seg = (uint64_t)n_ranks * B;
        HIPCHK(c, b->t_u_off.ensure(n_seg * 8));
        HIPCHK(c, b->t_u_len.ensure(n_seg * 4));
        HIPCHK(c, b->t_u_foff.ensure(n_seg * 8));
        launch_union_segments(s, payload, (uint32_t)n_ranks, stride, flags_off, boff_off, B, shifts, b->t_u_off.as<uint64_t>(),
                              b->t_u_len.as<uint32_t>(), b->t_u_foff.as<uint64_t>());
        HIPCHK(c, hipGetLastError());
        DictArgs a;
        memset(&a, 0, sizeof a);
        a.keys = (const uint64_t *)payload;
        a.seg.off = b->t_u_off.as<uint64_t>(); a.seg.len = b->t_u_len.as<uint32_t>(); a.seg.stride = 0;
        a.n_genomes = (uint32_t)n_ranks; a.bb = bb_min; a.cap_log2 = b->cap_log2;
        a.in_flags = payload; a.in_flag_off = b->t_u_foff.as<uint64_t>();
        const DictOut out = {&b->t_u_keys, &b->t_u_flags, &b->t_u_wg_base, &b->t_u_wg_cnt};
        // The calling rank's own list inside the payload is TRACKED through the union when the caller says which one it is (and it is
        // the list this batch exported): every local entry then learns its union entry in the union kernel and its column from the
        // union's sort -- no search of the dictionary afterwards.
        const bool track = my_rank >= 0 && my_rank < n_ranks && b->have_local && b->have_bits && b->total_keys && b->exported_ordered &&
                           counts[my_rank] == b->n_local && (bucket_bits[my_rank] & 0xff) == b->bb && b->n_local;
        if (track) {
            HIPCHK(c, b->t_track.ensure((b->n_local + 1) * 4));
            a.track = b->t_track.as<uint32_t>();
            a.track_g = (uint32_t)my_rank;
            a.track_flag_base = (uint64_t)my_rank * stride + flags_off;
        }
        int sb = 0;
        uint64_t n_union = 0;
        bool bits = false;
        int rc = run_dict_ladder(b, a, total, b->sb_union_hint > 0 ? b->sb_union_hint : 0, false, out, "dict_union", &sb, &n_union, &bits, nullptr);
        if (rc == GRM_OK) {
            b->sb_union_hint = sb;
            b->have_global = false;
            b->filter_singleton = filter_singleton;
            b->own_dict = false;
            b->entry_cols_ready = false;
            // the union's entries are distinct: sorted with their index, each learns its column in the select step
            HIPCHK(c, b->t_union_col.ensure((n_union + 1) * 4));
            rc = dict_from_distinct_entries(b, b->t_u_keys.as<uint64_t>(), b->t_u_flags.as<uint8_t>(), n_union, filter_singleton, b->t_union_col.as<uint32_t>());
            if (rc) return rc;
            if (track) {
                HIPCHK(c, b->d_entry_col.ensure((b->n_local + 1) * 4));
                TimeScope t(c, "dict_entry_cols", b->n_local);
                launch_entry_cols_from_union(s, b->d_wg_base.as<uint64_t>(), b->d_wg_cnt.as<uint32_t>(), b->t_ord_off.as<uint64_t>(), 1u << (b->bb + b->sb_dict),
                                             b->t_track.as<uint32_t>(), b->t_union_col.as<uint32_t>(), b->d_entry_col.as<uint32_t>());
                HIPCHK(c, hipGetLastError());
                b->entry_cols_ready = true;
            }
            return dict_attach(b, n_kmers);
        }
        if (rc != GRM_ERR_OVERFLOW) return rc;         // a union too large for the tables: sort everything instead
    }
    // general form: the ranks' entries back to back, then the sort-based merge
    HIPCHK(c, b->t_u_keys.ensure((total + 2) * 8 * (size_t)words));
    HIPCHK(c, b->t_u_flags.ensure(total + 16));
    uint64_t at = 0;
    for (int r = 0; r < n_ranks; r++) {
        if (!counts[r]) continue;
        HIPCHK(c, hipMemcpyAsync(b->t_u_keys.as<uint8_t>() + at * 8 * (size_t)words, payload + (uint64_t)r * stride, counts[r] * 8 * (size_t)words,
                                 hipMemcpyDeviceToDevice, s));
        HIPCHK(c, hipMemcpyAsync(b->t_u_flags.as<uint8_t>() + at, payload + (uint64_t)r * stride + flags_off, counts[r], hipMemcpyDeviceToDevice, s));
        at += counts[r];
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return grm_batch_set_global_dict(b, b->t_u_keys.p, b->t_u_flags.p, total, filter_singleton, n_kmers);
}

extern "C" int grm_batch_fill(grm_batch *b, grm_matrix **out)
{
    if (!b || !out) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->have_global) return fail(c, GRM_ERR_STATE, "grm_batch_fill before grm_batch_set_global_dict");
    HIPCHK(c, hipSetDevice(c->device));
    if (b->sorted_stage) return sorted_stage_fill(b, out);
    if (b->k > 32) return wide_stage_fill(b, out);
    hipStream_t s = c->stream;
    grm_matrix *m = new grm_matrix();
    m->ctx = c;
    m->k = b->k;
    m->words = 1;
    m->n_genomes = b->n_genomes;
    m->n_rows = ((size_t)b->n_genomes + 63) / 64;
    m->n_kmers = b->n_dict;
    const size_t cells = m->n_rows * m->n_kmers;
    hipError_t e = m->d_data.alloc(cells * 8);
    if (e == hipSuccess) e = m->d_kmers.alloc((m->n_kmers + 2) * 8);
    if (e != hipSuccess) { delete m; return fail(c, GRM_ERR_OOM, "matrix allocation failed (%zu cells)", cells); }
    int rc = GRM_OK;
    DevBuf &d_flag = b->t_flag;
    if (d_flag.ensure(4) != hipSuccess) { delete m; return fail(c, GRM_ERR_OOM, "alloc"); }
    for (;;) {
        const bool two_step = b->fill_by_bits && b->total_keys && m->n_rows >= 4 && c->opt_direct_permute <= 0;     // writes every cell
        if (cells && !two_step) (void)hipMemsetAsync(m->d_data.p, 0, cells * 8, s);
        (void)hipMemsetAsync(d_flag.p, 0, 4, s);
        if (cells && b->total_keys && b->fill_by_bits) {
            // two-step form (entry-major lines, then a transpose) from 4 word-rows up; the intermediate is zeroed
            // only when some columns may have no entry here (dictionary gathered from several ranks / chunks)
            uint64_t *em = nullptr;
            if (m->n_rows >= 4 && c->opt_direct_permute <= 0) {
                if (b->d_entry_major.ensure(cells * 8) != hipSuccess) { rc = fail(c, GRM_ERR_OOM, "matrix_fill: entry-major scratch"); break; }
                em = b->d_entry_major.as<uint64_t>();
                if (!b->own_dict) (void)hipMemsetAsync(em, 0, cells * 8, s);
            }
            TimeScope t(c, "matrix_fill", (uint64_t)b->n_local * m->n_rows);
            launch_matrix_permute(s, b->d_matrix_s.as<uint64_t>(), b->d_birth.as<uint16_t>(), b->d_wg_base.as<uint64_t>(),
                                  b->d_wg_cnt.as<uint32_t>(), b->d_entry_col.as<uint32_t>(), 1u << (b->bb + b->sb_dict),
                                  (uint32_t)m->n_rows, b->cap_log2, m->d_data.as<uint64_t>(), m->n_kmers, em);
        } else if (cells && b->total_keys) {
            TimeScope t(c, "matrix_fill", b->total_keys);
            launch_matrix_fill(s, b->d_keys.as<uint64_t>(), batch_segments(b),
                               (uint32_t)b->n_genomes, b->bb, b->sb_fill, b->cap_log2, b->d_dkeys.as<uint64_t>(),
                               b->d_dcol.as<uint32_t>(), b->d_seg_start.as<uint64_t>(), m->d_data.as<uint64_t>(), m->n_kmers,
                               d_flag.as<int>());
        }
        int ov = 0;
        hipError_t e2 = hipGetLastError();
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&ov, d_flag.p, 4, hipMemcpyDeviceToHost, s);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);
        if (e2 != hipSuccess) { rc = fail(c, GRM_ERR_HIP, "matrix_fill: %s", hipGetErrorString(e2)); break; }
        if (!ov) break;
        if (b->bb + b->sb_fill >= DICT_MAX_WG_BITS) { rc = fail(c, GRM_ERR_OVERFLOW, "matrix_fill: dictionary slice does not fit the LDS table"); break; }
        rc = bucketise_dict(b, b->sb_fill + 1);
        if (rc) break;
    }
    if (rc == GRM_OK && m->n_kmers) {
        hipError_t e3 = hipMemcpyAsync(m->d_kmers.p, b->d_dict.p, m->n_kmers * 8, hipMemcpyDeviceToDevice, s);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(s);
        if (e3 != hipSuccess) rc = fail(c, GRM_ERR_HIP, "dictionary copy: %s", hipGetErrorString(e3));
    }
    if (rc != GRM_OK) { delete m; return rc; }
    *out = m;
    return GRM_OK;
}

static int wide_scan(grm_ctx *c, DevBuf &tmp, bool inclusive, const uint32_t *in, uint32_t *out, uint64_t n);

// ---- pooled merge of counted sets (one-word k-mers): the sum of the counts of equal k-mers, filtered ----
// DSK counts every listed file as ONE pool (src/app.py:1371-1372).  A pool of more than 2^32 - 1 symbols is
// counted in chunks with abundance-min 1 and the chunk sets are merged here: sort by key, sum runs, filter.
extern "C" int grm_merge_counted_sets(grm_ctx *c, grm_kmer_set *const *sets, int n_sets, uint32_t abundance_min, grm_kmer_set **out)
{
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!out || n_sets < 0 || (n_sets && !sets)) return fail(c, GRM_ERR_ARG, "grm_merge_counted_sets: bad argument");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (abundance_min < 1) abundance_min = 1;
    uint64_t n = 0, occ = 0;
    int k = n_sets ? sets[0]->k : 1;
    for (int i = 0; i < n_sets; i++) {
        if (!sets[i] || sets[i]->k != k) return fail(c, GRM_ERR_ARG, "grm_merge_counted_sets: sets with different k");
        if (sets[i]->words != 1) return fail(c, GRM_ERR_UNSUPPORTED, "grm_merge_counted_sets: k=%d (two-word k-mers) is not supported", k);
        n += sets[i]->n;
        occ += sets[i]->occurrences;
    }
    if (n >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "grm_merge_counted_sets: %llu entries exceed 2^32-1", (unsigned long long)n);
    grm_kmer_set *r = new grm_kmer_set();
    r->ctx = c; r->k = k; r->words = 1; r->occurrences = occ;
    auto bail = [&](int code) { delete r; return code; };
    if (n == 0) { *out = r; return GRM_OK; }
    DevBuf k0, k1, c0, c1, head, incl, sums, rkeys, keep, pos, tmp;
    hipError_t e = hipSuccess;
    auto need = [&](DevBuf &d, size_t bytes) { if (e == hipSuccess) e = d.alloc(bytes); };
    need(k0, n * 8); need(k1, n * 8); need(c0, n * 4); need(c1, n * 4); need(head, n * 4); need(incl, n * 4);
    if (e != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "grm_merge_counted_sets: %s", hipGetErrorString(e)));
    uint64_t at = 0;
    for (int i = 0; i < n_sets && e == hipSuccess; i++) {
        const size_t m = sets[i]->n;
        if (!m) continue;
        if (sets[i]->on_device) {
            e = hipMemcpyAsync(k0.as<uint64_t>() + at, sets[i]->d_kmers.p, m * 8, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(c0.as<uint32_t>() + at, sets[i]->d_counts.p, m * 4, hipMemcpyDeviceToDevice, s);
        } else {
            e = hipMemcpy(k0.as<uint64_t>() + at, sets[i]->kmers.data(), m * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(c0.as<uint32_t>() + at, sets[i]->counts.data(), m * 4, hipMemcpyHostToDevice);
        }
        at += m;
    }
    size_t tb = 0;
    if (e == hipSuccess) e = sort_pairs_u64_u32(s, k0.as<uint64_t>(), k1.as<uint64_t>(), c0.as<uint32_t>(), c1.as<uint32_t>(), n, nullptr, tb);
    if (e == hipSuccess) e = tmp.alloc(tb);
    if (e == hipSuccess) e = sort_pairs_u64_u32(s, k0.as<uint64_t>(), k1.as<uint64_t>(), c0.as<uint32_t>(), c1.as<uint32_t>(), n, tmp.p, tb);
    if (e != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "grm_merge_counted_sets: %s", hipGetErrorString(e)));
    launch_runs_mark(s, k1.as<uint64_t>(), n, head.as<uint32_t>());
    int rc = wide_scan(c, tmp, true, head.as<uint32_t>(), incl.as<uint32_t>(), n);
    if (rc) return bail(rc);
    uint32_t n_runs = 0;
    if (hipMemcpyAsync(&n_runs, incl.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return bail(fail(c, GRM_ERR_HIP, "grm_merge_counted_sets: D2H"));
    need(sums, (size_t)n_runs * 8); need(rkeys, (size_t)n_runs * 8); need(keep, ((size_t)n_runs + 1) * 4); need(pos, ((size_t)n_runs + 1) * 4);
    if (e != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "grm_merge_counted_sets: %s", hipGetErrorString(e)));
    (void)hipMemsetAsync(sums.p, 0, (size_t)n_runs * 8, s);
    (void)hipMemsetAsync(keep.as<uint32_t>() + n_runs, 0, 4, s);
    launch_runs_reduce(s, k1.as<uint64_t>(), c1.as<uint32_t>(), head.as<uint32_t>(), incl.as<uint32_t>(), n, rkeys.as<uint64_t>(),
                       sums.as<unsigned long long>());
    launch_runs_keep(s, sums.as<unsigned long long>(), n_runs, abundance_min, keep.as<uint32_t>());
    rc = wide_scan(c, tmp, false, keep.as<uint32_t>(), pos.as<uint32_t>(), (uint64_t)n_runs + 1);
    if (rc) return bail(rc);
    uint32_t n_out = 0;
    if (hipMemcpyAsync(&n_out, pos.as<uint32_t>() + n_runs, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return bail(fail(c, GRM_ERR_HIP, "grm_merge_counted_sets: D2H"));
    if (r->d_kmers.alloc((size_t)n_out * 8 + 16) != hipSuccess || r->d_counts.alloc((size_t)n_out * 4 + 16) != hipSuccess)
        return bail(fail(c, GRM_ERR_OOM, "grm_merge_counted_sets: result allocation failed"));
    launch_runs_emit(s, rkeys.as<uint64_t>(), sums.as<unsigned long long>(), keep.as<uint32_t>(), pos.as<uint32_t>(), n_runs,
                     r->d_kmers.as<uint64_t>(), r->d_counts.as<uint32_t>());
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "grm_merge_counted_sets failed"));
    r->n = n_out;
    r->on_device = true;
    r->on_host = false;
    *out = r;
    return GRM_OK;
}

// ---- inputs larger than one device batch: two passes over chunks of genomes -------------------
// Pass 1 pushes every chunk through partition + local dictionary and keeps only (key, flag) of its
// distinct k-mers in an accumulator; pass 2 pushes the chunks through again, hands the accumulated
// entries to grm_batch_set_global_dict (a k-mer seen in several chunks is a run of equal keys: one
// column carried by several genomes -- the multi-GPU merge, applied over time instead of over ranks)
// and fills that chunk's word-rows.  grm_matrix_stack_rows puts the row blocks together.
struct grm_dict_accum {
    CtxRef ctx;
    int words = 0;                 // 0 until the first batch is added
    uint64_t n = 0, cap = 0;
    DevBuf keys, flags;
    // the merged dictionary, computed by the first pass-2 chunk and reused by the others (one-word k-mers)
    DevBuf dict;
    uint64_t n_dict = 0;
    int dict_filter = -1;          // -1: not computed for the current contents
};

extern "C" int grm_dict_accum_create(grm_ctx *c, grm_dict_accum **out)
{
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!out) return fail(c, GRM_ERR_ARG, "grm_dict_accum_create: NULL out");
    grm_dict_accum *a = new grm_dict_accum();
    a->ctx = c;
    *out = a;
    return GRM_OK;
}
extern "C" void grm_dict_accum_free(grm_dict_accum *a)
{
    if (!a) return;
    (void)hipSetDevice(a->ctx->device);
    delete a;
}
extern "C" uint64_t grm_dict_accum_size(const grm_dict_accum *a) { return a ? a->n : 0; }

// appends the local dictionary of `b` (grm_batch_local_dict must have run)
extern "C" int grm_dict_accum_add(grm_dict_accum *a, grm_batch *b)
{
    if (!a || !b) return GRM_ERR_ARG;
    grm_ctx *c = a->ctx;
    if (b->ctx != c) return fail(c, GRM_ERR_ARG, "grm_dict_accum_add: batch of another context");
    if (!b->have_local) return fail(c, GRM_ERR_STATE, "grm_dict_accum_add before grm_batch_local_dict");
    HIPCHK(c, hipSetDevice(c->device));
    const int words = words_of(b->k);
    if (a->words && a->words != words) return fail(c, GRM_ERR_ARG, "grm_dict_accum_add: batches with different k");
    a->words = words;
    const uint64_t add = b->n_local, need = a->n + add;
    if (need > a->cap) {                                   // grow geometrically, keep what is there
        const uint64_t cap = std::max<uint64_t>(need + need / 2, 1u << 20);
        DevBuf nk, nf;
        HIPCHK(c, nk.alloc(cap * 8 * (size_t)words));
        HIPCHK(c, nf.alloc(cap));
        if (a->n) {
            HIPCHK(c, hipMemcpyAsync(nk.p, a->keys.p, a->n * 8 * (size_t)words, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(nf.p, a->flags.p, a->n, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        std::swap(a->keys.p, nk.p); std::swap(a->keys.bytes, nk.bytes);
        std::swap(a->flags.p, nf.p); std::swap(a->flags.bytes, nf.bytes);
        a->cap = cap;
    }
    if (add) {
        int rc = grm_batch_export_dict(b, a->keys.as<uint8_t>() + a->n * 8 * (size_t)words, a->flags.as<uint8_t>() + a->n);
        if (rc) return rc;
    }
    a->n = need;
    a->dict_filter = -1;
    return GRM_OK;
}

extern "C" int grm_batch_set_global_dict_accum(grm_batch *b, const grm_dict_accum *a_, int filter_singleton, uint64_t *n_kmers)
{
    if (!b || !a_) return GRM_ERR_ARG;
    grm_dict_accum *a = const_cast<grm_dict_accum *>(a_);      // the cached dictionary is not part of its visible state
    grm_ctx *c = b->ctx;
    if (a->words && a->words != words_of(b->k)) return fail(c, GRM_ERR_ARG, "accumulator holds k-mers of another width");
    if (b->k > 32 || !a->n) return grm_batch_set_global_dict(b, a->keys.p, a->flags.p, a->n, filter_singleton, n_kmers);
    if (!b->partitioned) return fail(c, GRM_ERR_STATE, "grm_batch_set_global_dict_accum before grm_batch_partition");
    HIPCHK(c, hipSetDevice(c->device));
    b->have_global = false;
    b->filter_singleton = filter_singleton;
    b->own_dict = false;
    b->entry_cols_ready = false;
    if (a->dict_filter != (filter_singleton ? 1 : 0)) {
        // sort / merge / filter the accumulated entries ONCE; every chunk of pass 2 then only looks its own entries up
        int rc = dict_from_entries(b, a->keys.as<uint64_t>(), a->flags.as<uint8_t>(), a->n, filter_singleton);
        if (rc) return rc;
        HIPCHK(c, a->dict.ensure((b->n_dict + 2) * 8));
        if (b->n_dict) HIPCHK(c, hipMemcpyAsync(a->dict.p, b->d_dict.p, b->n_dict * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        a->n_dict = b->n_dict;
        a->dict_filter = filter_singleton ? 1 : 0;
    } else {
        b->n_dict = a->n_dict;
        HIPCHK(c, b->d_dict.ensure((b->n_dict + 2) * 8));
        if (b->n_dict) HIPCHK(c, hipMemcpyAsync(b->d_dict.p, a->dict.p, b->n_dict * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return dict_attach(b, n_kmers);
}

// parts[i]: word-rows of consecutive genome blocks against the SAME dictionary; every part but the last
// must hold a multiple of 64 genomes.  -> one matrix, rows stacked.
extern "C" int grm_matrix_stack_rows(grm_matrix *const *parts, int n_parts, grm_matrix **out)
{
    if (!parts || n_parts < 1 || !out || !parts[0]) return GRM_ERR_ARG;
    grm_ctx *c = parts[0]->ctx;
    if (!c) return GRM_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t U = parts[0]->n_kmers;
    size_t rows = 0;
    long genomes = 0;
    for (int i = 0; i < n_parts; i++) {
        const grm_matrix *p = parts[i];
        if (!p || p->ctx != c || p->n_kmers != U || p->k != parts[0]->k || p->words != parts[0]->words)
            return fail(c, GRM_ERR_ARG, "grm_matrix_stack_rows: part %d does not share the dictionary of part 0", i);
        if (i + 1 < n_parts && p->n_genomes % 64) return fail(c, GRM_ERR_ARG, "grm_matrix_stack_rows: part %d holds %d genomes (not a multiple of 64)", i, p->n_genomes);
        rows += p->n_rows;
        genomes += p->n_genomes;
    }
    grm_matrix *m = new grm_matrix();
    m->ctx = c; m->k = parts[0]->k; m->words = parts[0]->words;
    m->n_genomes = (int)genomes; m->n_rows = rows; m->n_kmers = U;
    if (m->d_data.alloc(rows * U * 8) != hipSuccess || m->d_kmers.alloc((U + 2) * 8 * (size_t)m->words) != hipSuccess) {
        delete m;
        return fail(c, GRM_ERR_OOM, "grm_matrix_stack_rows: matrix allocation failed (%zu x %zu)", rows, U);
    }
    size_t r0 = 0;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n_parts && e == hipSuccess; i++) {
        if (parts[i]->n_rows && U)
            e = hipMemcpyAsync(m->d_data.as<uint64_t>() + r0 * U, parts[i]->d_data.p, parts[i]->n_rows * U * 8, hipMemcpyDeviceToDevice, c->stream);
        r0 += parts[i]->n_rows;
    }
    if (e == hipSuccess && U) e = hipMemcpyAsync(m->d_kmers.p, parts[0]->d_kmers.p, U * 8 * (size_t)m->words, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { delete m; return fail(c, GRM_ERR_HIP, "grm_matrix_stack_rows: %s", hipGetErrorString(e)); }
    *out = m;
    return GRM_OK;
}

// ---- two-word k-mers (33..64): sort-based path ------------------------------------------------
// Buffers live in the batch and are reused across runs (grow-only).  Four N-sized uint64 and two
// N-sized uint32 arrays are ping-ponged through extract / sort / gather; the head flags alias the
// dead extract buffers:  A0=hi0 A1=lo0 | sort1 (A1,I0)->(A2,I1) | A2=hi0[I1] | sort2 (A2,I1)->(A3,I0)
// | A2=lo0[I0]  =>  khi=A3, klo=A2, pos=I0 ; key_head/kg_head in A0, sub_id in A1.
struct WideSorted {
    DevBuf A[4], I[2];
    DevBuf sub_start, sub_key_head, sub_ok, key_incl, carriers, keep, col, opos, tmp, n_valid;
    uint32_t n = 0, n_sub = 0;
    uint64_t *khi() const { return A[3].as<uint64_t>(); }
    uint64_t *klo() const { return A[2].as<uint64_t>(); }
    uint32_t *pos() const { return I[0].as<uint32_t>(); }
};

static int wide_scan(grm_ctx *c, DevBuf &tmp, bool inclusive, const uint32_t *in, uint32_t *out, uint64_t n)
{
    size_t tb = 0;
    hipError_t e = inclusive ? inclusive_scan_u32(c->stream, in, out, n, nullptr, tb) : exclusive_scan_u32(c->stream, in, out, n, nullptr, tb);
    if (e == hipSuccess) e = tmp.ensure(tb);
    if (e == hipSuccess)
        e = inclusive ? inclusive_scan_u32(c->stream, in, out, n, tmp.p, tb) : exclusive_scan_u32(c->stream, in, out, n, tmp.p, tb);
    if (e != hipSuccess) return fail(c, GRM_ERR_HIP, "scan: %s", hipGetErrorString(e));
    return GRM_OK;
}

// parse must have run (batch_partition_impl with k > 32).  Extract, sort, mark runs.
// preloaded: A[0] / A[1] already hold total_syms (hi, lo) keys, all valid, and d_genome_sym_off
// their per-genome offsets (grm_build_matrix on two-word sets).
static int wide_sort_and_mark(grm_batch *b, int k, uint32_t abundance_min, WideSorted &W, bool preloaded = false)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    const uint64_t N = b->total_syms;
    if (N >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "k > 32 path is limited to 2^32-1 symbols per batch (got %llu)", (unsigned long long)N);
    W.n = W.n_sub = 0;
    b->total_keys = 0;
    if (N == 0) return GRM_OK;
    for (int i = 0; i < 4; i++) HIPCHK(c, W.A[i].ensure((N + 2) * 8));
    for (int i = 0; i < 2; i++) HIPCHK(c, W.I[i].ensure((N + 2) * 4));
    HIPCHK(c, W.n_valid.ensure(8));
    uint64_t *hi0 = W.A[0].as<uint64_t>(), *lo0 = W.A[1].as<uint64_t>(), *t2 = W.A[2].as<uint64_t>(), *t3 = W.A[3].as<uint64_t>();
    uint32_t *i0 = W.I[0].as<uint32_t>(), *i1 = W.I[1].as<uint32_t>();
    HIPCHK(c, hipMemsetAsync(W.n_valid.p, 0, 8, s));
    unsigned long long nv = N;
    if (!preloaded) {
        {
            TimeScope t(c, "wide_extract", N);
            launch_wide_extract(s, b->d_sym2.as<uint64_t>(), b->d_inv.as<uint64_t>(), N, k, hi0, lo0, W.n_valid.as<unsigned long long>());
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&nv, W.n_valid.p, 8, hipMemcpyDeviceToHost, s));
    }
    {
        TimeScope t(c, "wide_sort", N);
        launch_iota_u32(s, i0, N);
        size_t tb = 0;
        HIPCHK(c, sort_pairs_u64_u32(s, lo0, t2, i0, i1, N, nullptr, tb));
        HIPCHK(c, W.tmp.ensure(tb));
        HIPCHK(c, sort_pairs_u64_u32(s, lo0, t2, i0, i1, N, W.tmp.p, tb));           // by lo
        launch_gather_u64(s, hi0, i1, N, t2);                                          // hi in lo-order
        tb = 0;
        HIPCHK(c, sort_pairs_u64_u32(s, t2, t3, i1, i0, N, nullptr, tb));
        HIPCHK(c, W.tmp.ensure(tb));
        HIPCHK(c, sort_pairs_u64_u32(s, t2, t3, i1, i0, N, W.tmp.p, tb));              // stable by hi: khi = A3, pos = I0
        launch_gather_u64(s, lo0, i0, N, t2);                                          // klo = A2
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(s));
    b->total_keys = nv;
    W.n = (uint32_t)nv;
    const uint32_t n = W.n;
    if (n == 0) return GRM_OK;
    TimeScope t(c, "wide_mark", n);
    uint32_t *key_head = W.A[0].as<uint32_t>(), *kg_head = W.A[0].as<uint32_t>() + N, *sub_id = W.A[1].as<uint32_t>();
    launch_wide_mark(s, W.khi(), W.klo(), W.pos(), b->d_genome_sym_off.as<uint64_t>(), (uint32_t)b->n_genomes, n, key_head, kg_head);
    int rc = wide_scan(c, W.tmp, false, kg_head, sub_id, n);
    if (rc) return rc;
    uint32_t last_id = 0, last_flag = 0;
    HIPCHK(c, hipMemcpyAsync(&last_id, sub_id + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&last_flag, kg_head + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    W.n_sub = last_id + last_flag;
    HIPCHK(c, W.sub_start.ensure(((size_t)W.n_sub + 1) * 4));
    HIPCHK(c, W.sub_key_head.ensure((size_t)W.n_sub * 4 + 4));
    HIPCHK(c, W.sub_ok.ensure((size_t)W.n_sub * 4 + 4));
    launch_wide_sub_start(s, kg_head, sub_id, n, W.n_sub, W.sub_start.as<uint32_t>());
    launch_wide_sub(s, W.sub_start.as<uint32_t>(), key_head, W.n_sub, abundance_min, W.sub_key_head.as<uint32_t>(),
                    W.sub_ok.as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    return GRM_OK;
}

static void wide_free(WideSorted *w) { delete w; }

// ---- three- and four-word k-mers (65..128): sort-based path, grm_multi.hip --------------------------
// K[w]: the extracted words by symbol position, S[w]: the same in sorted order (W words each, word 0 most
// significant); pos = position of every sorted entry.  Stable LSD radix sort, one pass per word.
struct MultiSorted {
    DevBuf K[4], S[4], I[2], tk;
    DevBuf key_head, kg_head, sub_id, sub_start, sub_key_head, sub_ok, key_incl, carriers, keep, col, opos, tmp, n_valid;
    uint32_t n = 0, n_sub = 0;
    int words = 0;
    MultiWords sorted() const
    {
        MultiWords m;
        for (int j = 0; j < 4; j++) m.w[j] = S[j].as<uint64_t>();
        return m;
    }
    uint32_t *pos() const { return I[0].as<uint32_t>(); }
};
static void multi_free(MultiSorted *w) { delete w; }

// parse must have run (batch_partition_impl with k > 64), or K[] is preloaded with total_syms valid keys
// lists: the preloaded keys are not the batch's genomes but lists of their own (the staged calls' gathered dictionaries): n_lists
// of them, bounds in the device array list_off, n_keys keys in all
struct MultiLists {
    const uint64_t *list_off = nullptr;
    uint32_t n_lists = 0;
    uint64_t n_keys = 0;
};
static int multi_sort_and_mark(grm_batch *b, int k, uint32_t abundance_min, MultiSorted &M, bool preloaded = false, const MultiLists *lists = nullptr)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    const int W = words_of(k);
    const uint64_t N = lists ? lists->n_keys : b->total_syms;
    const uint64_t *gso = lists ? lists->list_off : b->d_genome_sym_off.as<uint64_t>();
    const uint32_t n_gen = lists ? lists->n_lists : (uint32_t)b->n_genomes;
    if (N >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "k > 64 path is limited to 2^32-1 symbols per batch (got %llu)", (unsigned long long)N);
    M.n = M.n_sub = 0;
    M.words = W;
    b->total_keys = 0;
    if (N == 0) return GRM_OK;
    for (int j = 0; j < W; j++) { HIPCHK(c, M.K[j].ensure((N + 2) * 8)); HIPCHK(c, M.S[j].ensure((N + 2) * 8)); }
    for (int i = 0; i < 2; i++) HIPCHK(c, M.I[i].ensure((N + 2) * 4));
    HIPCHK(c, M.tk.ensure((N + 2) * 8));
    HIPCHK(c, M.n_valid.ensure(8));
    HIPCHK(c, hipMemsetAsync(M.n_valid.p, 0, 8, s));
    unsigned long long nv = N;
    if (!preloaded) {
        MultiWordsOut out;
        for (int j = 0; j < 4; j++) out.w[j] = M.K[j].as<uint64_t>();
        {
            TimeScope t(c, "multi_extract", N);
            launch_multi_extract(s, W, b->d_sym2.as<uint64_t>(), b->d_inv.as<uint64_t>(), N, k, out, M.n_valid.as<unsigned long long>());
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&nv, M.n_valid.p, 8, hipMemcpyDeviceToHost, s));
    }
    {
        TimeScope t(c, "multi_sort", N);
        uint32_t *ia = M.I[0].as<uint32_t>(), *ib = M.I[1].as<uint32_t>();
        launch_iota_u32(s, ia, N);
        for (int j = W - 1; j >= 0; j--) {                       // least significant word first
            const uint64_t *src = M.K[j].as<uint64_t>();
            if (j != W - 1) {                                      // word j in the order reached so far
                launch_gather_u64(s, M.K[j].as<uint64_t>(), ia, N, M.tk.as<uint64_t>());
                src = M.tk.as<uint64_t>();
            }
            size_t tb = 0;
            HIPCHK(c, sort_pairs_u64_u32(s, src, M.S[0].as<uint64_t>(), ia, ib, N, nullptr, tb));
            HIPCHK(c, M.tmp.ensure(tb));
            HIPCHK(c, sort_pairs_u64_u32(s, src, M.S[0].as<uint64_t>(), ia, ib, N, M.tmp.p, tb));
            std::swap(ia, ib);
        }
        // ia = final order; S[0] already holds word 0 sorted; the other words follow the order
        if (ia != M.I[0].as<uint32_t>()) HIPCHK(c, hipMemcpyAsync(M.I[0].p, ia, N * 4, hipMemcpyDeviceToDevice, s));
        for (int j = 1; j < W; j++) launch_gather_u64(s, M.K[j].as<uint64_t>(), M.I[0].as<uint32_t>(), N, M.S[j].as<uint64_t>());
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(s));
    b->total_keys = nv;
    M.n = (uint32_t)nv;
    const uint32_t n = M.n;
    if (n == 0) return GRM_OK;
    TimeScope t(c, "multi_mark", n);
    HIPCHK(c, M.key_head.ensure((size_t)n * 4)); HIPCHK(c, M.kg_head.ensure((size_t)n * 4)); HIPCHK(c, M.sub_id.ensure((size_t)n * 4));
    launch_multi_mark(s, W, M.sorted(), M.pos(), gso, n_gen, n, M.key_head.as<uint32_t>(), M.kg_head.as<uint32_t>());
    int rc = wide_scan(c, M.tmp, false, M.kg_head.as<uint32_t>(), M.sub_id.as<uint32_t>(), n);
    if (rc) return rc;
    uint32_t last_id = 0, last_flag = 0;
    HIPCHK(c, hipMemcpyAsync(&last_id, M.sub_id.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&last_flag, M.kg_head.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    M.n_sub = last_id + last_flag;
    HIPCHK(c, M.sub_start.ensure(((size_t)M.n_sub + 1) * 4));
    HIPCHK(c, M.sub_key_head.ensure((size_t)M.n_sub * 4 + 4));
    HIPCHK(c, M.sub_ok.ensure((size_t)M.n_sub * 4 + 4));
    launch_wide_sub_start(s, M.kg_head.as<uint32_t>(), M.sub_id.as<uint32_t>(), n, M.n_sub, M.sub_start.as<uint32_t>());
    launch_wide_sub(s, M.sub_start.as<uint32_t>(), M.key_head.as<uint32_t>(), M.n_sub, abundance_min, M.sub_key_head.as<uint32_t>(),
                    M.sub_ok.as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    return GRM_OK;
}

static int multi_matrix(grm_batch *b, int k, uint32_t abundance_min, int filter_singleton, grm_matrix **out, bool preloaded = false)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    if (!b->multi) b->multi = new MultiSorted();
    MultiSorted &M = *b->multi;
    int rc = multi_sort_and_mark(b, k, abundance_min, M, preloaded);
    if (rc) return rc;
    const int W = words_of(k);
    grm_matrix *m = new grm_matrix();
    m->ctx = c; m->k = k; m->words = W; m->n_genomes = b->n_genomes;
    m->n_rows = ((size_t)b->n_genomes + 63) / 64;
    auto bail = [&](int code) { delete m; return code; };
    uint32_t U = 0;
    if (M.n_sub) {
        TimeScope t(c, "multi_reduce", M.n_sub);
        if (M.key_incl.ensure((size_t)M.n_sub * 4) != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "alloc"));
        rc = wide_scan(c, M.tmp, true, M.sub_key_head.as<uint32_t>(), M.key_incl.as<uint32_t>(), M.n_sub);
        if (rc) return bail(rc);
        uint32_t n_keys = 0;
        if (hipMemcpyAsync(&n_keys, M.key_incl.as<uint32_t>() + (M.n_sub - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "D2H"));
        if (M.carriers.ensure((size_t)n_keys * 4 + 4) != hipSuccess || M.keep.ensure((size_t)n_keys * 4 + 4) != hipSuccess ||
            M.col.ensure((size_t)n_keys * 4 + 4) != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "alloc"));
        (void)hipMemsetAsync(M.carriers.p, 0, (size_t)n_keys * 4 + 4, s);
        launch_wide_key_count(s, M.key_incl.as<uint32_t>(), M.sub_ok.as<uint32_t>(), M.n_sub, M.carriers.as<uint32_t>());
        launch_wide_keep(s, M.carriers.as<uint32_t>(), n_keys, filter_singleton ? 2u : 1u, M.keep.as<uint32_t>());
        rc = wide_scan(c, M.tmp, false, M.keep.as<uint32_t>(), M.col.as<uint32_t>(), n_keys);
        if (rc) return bail(rc);
        uint32_t last_col = 0, last_keep = 0;
        if (hipMemcpyAsync(&last_col, M.col.as<uint32_t>() + (n_keys - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&last_keep, M.keep.as<uint32_t>() + (n_keys - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "D2H"));
        U = last_col + last_keep;
    }
    m->n_kmers = U;
    const size_t cells = m->n_rows * (size_t)U;
    if (m->d_data.alloc(cells * 8) != hipSuccess || m->d_kmers.alloc(((size_t)U + 1) * 8 * (size_t)W) != hipSuccess)
        return bail(fail(c, GRM_ERR_OOM, "matrix allocation failed"));
    if (cells) (void)hipMemsetAsync(m->d_data.p, 0, cells * 8, s);
    if (U) {
        TimeScope t(c, "multi_emit", M.n_sub);
        launch_multi_emit(s, W, M.sorted(), M.pos(), b->d_genome_sym_off.as<uint64_t>(), (uint32_t)b->n_genomes, M.sub_start.as<uint32_t>(),
                          M.sub_key_head.as<uint32_t>(), M.sub_ok.as<uint32_t>(), M.key_incl.as<uint32_t>(), M.keep.as<uint32_t>(),
                          M.col.as<uint32_t>(), M.n_sub, m->d_kmers.as<uint64_t>(), m->d_data.as<uint64_t>(), U);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "multi_emit failed"));
    *out = m;
    return GRM_OK;
}

// single-genome batch -> sorted counted set (W words per k-mer)
static int multi_set(grm_batch *b, int k, uint32_t abundance_min, grm_kmer_set **out, const MultiLists *lists = nullptr)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    if (!b->multi) b->multi = new MultiSorted();
    MultiSorted &M = *b->multi;
    int rc = multi_sort_and_mark(b, k, abundance_min, M, lists != nullptr, lists);
    if (rc) return rc;
    const int W = words_of(k);
    grm_kmer_set *set = new grm_kmer_set();
    set->ctx = c; set->k = k; set->words = W; set->occurrences = b->total_keys;
    *out = set;
    if (!M.n_sub) return GRM_OK;
    HIPCHK(c, M.opos.ensure((size_t)M.n_sub * 4));
    rc = wide_scan(c, M.tmp, false, M.sub_ok.as<uint32_t>(), M.opos.as<uint32_t>(), M.n_sub);
    if (rc) return rc;
    uint32_t last_pos = 0, last_ok = 0;
    HIPCHK(c, hipMemcpyAsync(&last_pos, M.opos.as<uint32_t>() + (M.n_sub - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&last_ok, M.sub_ok.as<uint32_t>() + (M.n_sub - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const size_t n_out = (size_t)last_pos + last_ok;
    if (!n_out) return GRM_OK;
    HIPCHK(c, set->d_kmers.alloc(n_out * 8 * (size_t)W)); HIPCHK(c, set->d_counts.alloc(n_out * 4));
    launch_multi_set(s, W, M.sorted(), M.sub_start.as<uint32_t>(), M.sub_ok.as<uint32_t>(), M.opos.as<uint32_t>(), M.n_sub,
                     set->d_kmers.as<uint64_t>(), set->d_counts.as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    set->n = n_out;
    set->on_device = true;
    set->on_host = false;
    return GRM_OK;
}

// dsk2kover's merge for three- / four-word k-mers: the sets' keys laid out genome after genome, then the same reduction
static int build_matrix_multi(grm_ctx *c, grm_kmer_set *const *sets, int n_genomes, int filter_singleton, grm_matrix **out)
{
    const int k = sets[0]->k, W = sets[0]->words;
    std::vector<uint64_t> gko(n_genomes + 1, 0);
    for (int g = 0; g < n_genomes; g++) {
        if (!sets[g] || sets[g]->k != k || sets[g]->words != W) return fail(c, GRM_ERR_ARG, "grm_build_matrix: sets with different k");
        gko[g + 1] = gko[g] + sets[g]->n;
    }
    const uint64_t n = gko[n_genomes];
    grm_batch *b = nullptr;
    int rc = grm_batch_create(c, n_genomes, &b);
    if (rc) return rc;
    auto body = [&]() -> int {
        b->uploaded = true;
        b->k = k;
        b->abundance_min = 1;
        b->total_syms = n;
        b->multi = new MultiSorted();
        MultiSorted &M = *b->multi;
        if (n >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "k > 64 merge is limited to 2^32-1 set entries (got %llu)", (unsigned long long)n);
        MultiWordsOut dst;
        for (int j = 0; j < 4; j++) dst.w[j] = nullptr;
        for (int j = 0; j < W; j++) { HIPCHK(c, M.K[j].ensure((n + 2) * 8)); dst.w[j] = M.K[j].as<uint64_t>(); }
        HIPCHK(c, b->d_genome_sym_off.ensure(((size_t)n_genomes + 1) * 8));
        HIPCHK(c, hipMemcpy(b->d_genome_sym_off.p, gko.data(), ((size_t)n_genomes + 1) * 8, hipMemcpyHostToDevice));
        DevBuf stage;
        for (int g = 0; g < n_genomes; g++) {
            const size_t m = sets[g]->n;
            if (!m) continue;
            const uint64_t *src = sets[g]->d_kmers.as<uint64_t>();
            if (!sets[g]->on_device) {
                HIPCHK(c, stage.ensure(m * 8 * (size_t)W));
                HIPCHK(c, hipMemcpy(stage.p, sets[g]->kmers.data(), m * 8 * (size_t)W, hipMemcpyHostToDevice));
                src = stage.as<uint64_t>();
            }
            launch_multi_split(c->stream, W, src, m, dst, gko[g]);
            if (!sets[g]->on_device) HIPCHK(c, hipStreamSynchronize(c->stream));       // the staging buffer is reused
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return multi_matrix(b, k, 1, filter_singleton, out, true);
    };
    rc = body();
    grm_batch_free(b);
    return rc;
}

// ---- the staged calls over the sort path -----------------------------------------------------------------------
// grm_batch_partition    parse, then multi_matrix WITHOUT the singleton filter: the batch's own matrix = its sorted dictionary
//                        (n_local x W words) and its presence bits; flag of a column = 1 / 2 for one / several carriers
// export_dict            copies of the two
// set_global_dict        the gathered lists (each holds distinct keys), with the entries flagged 2 laid out twice: sorted as ONE list
//                        (multi_set, the counted set of a single-genome batch) a key that is "in two lists, or flagged in one" is a
//                        run of length >= 2 -- the singleton filter is that call's abundance filter
// fill                   every own column looks its key up in the global dictionary (binary search) and moves there
struct SortedStage {
    grm_matrix *own = nullptr;
    DevBuf flags, gkeys, col, list_off, mark, pos;
    uint64_t n_global = 0;
    int words = 0;
};
static void sorted_stage_free(SortedStage *s)
{
    if (!s) return;
    delete s->own;
    delete s;
}
static int sorted_stage_local(grm_batch *b, int k, uint32_t abundance_min)
{
    grm_ctx *c = b->ctx;
    if (!b->sorted) b->sorted = new SortedStage();
    SortedStage &S = *b->sorted;
    delete S.own;
    S.own = nullptr;
    b->have_local = b->have_global = false;
    b->exported_ordered = false;
    grm_matrix *m = nullptr;
    int rc = multi_matrix(b, k, abundance_min, 0, &m);
    if (rc) return rc;
    S.own = m;
    S.words = m->words;
    b->n_local = m->n_kmers;
    HIPCHK(c, S.flags.ensure(m->n_kmers + 16));
    launch_multi_flags(c->stream, m->d_data.as<uint64_t>(), m->n_rows, m->n_kmers, S.flags.as<uint8_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b->sorted_stage = true;
    b->partitioned = true;
    b->have_local = true;
    return GRM_OK;
}
static int sorted_stage_export(grm_batch *b, void *dev_keys_out, void *dev_flags_out)
{
    grm_ctx *c = b->ctx;
    SortedStage &S = *b->sorted;
    if (b->n_local) {
        HIPCHK(c, hipMemcpyAsync(dev_keys_out, S.own->d_kmers.p, b->n_local * 8 * (size_t)S.words, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(dev_flags_out, S.flags.p, b->n_local, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GRM_OK;
}
static int sorted_stage_global(grm_batch *b, const void *dev_keys, const void *dev_flags, uint64_t n, int filter_singleton, uint64_t *n_kmers)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    if (!b->sorted || !b->sorted->own) return fail(c, GRM_ERR_STATE, "grm_batch_set_global_dict before grm_batch_partition");
    SortedStage &S = *b->sorted;
    const int W = S.words;
    b->have_global = false;
    b->filter_singleton = filter_singleton;
    S.n_global = 0;
    if (n) {
        MultiSorted &M = *b->multi;
        HIPCHK(c, S.mark.ensure((n + 1) * 4));
        HIPCHK(c, S.pos.ensure((n + 1) * 4));
        launch_multi_flag_mark(s, (const uint8_t *)dev_flags, n, S.mark.as<uint32_t>());
        int rc = wide_scan(c, M.tmp, false, S.mark.as<uint32_t>(), S.pos.as<uint32_t>(), n);
        if (rc) return rc;
        uint32_t last_pos = 0, last_mark = 0;
        HIPCHK(c, hipMemcpyAsync(&last_pos, S.pos.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(&last_mark, S.mark.as<uint32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const uint64_t total = n + last_pos + last_mark;
        if (total >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "gathered dictionaries of %llu entries exceed 2^32-1", (unsigned long long)total);
        MultiWordsOut dst;
        for (int j = 0; j < 4; j++) dst.w[j] = nullptr;
        for (int j = 0; j < W; j++) { HIPCHK(c, M.K[j].ensure((total + 2) * 8)); dst.w[j] = M.K[j].as<uint64_t>(); }
        launch_multi_split(s, W, (const uint64_t *)dev_keys, n, dst, 0);
        launch_multi_split_marked(s, W, (const uint64_t *)dev_keys, S.mark.as<uint32_t>(), S.pos.as<uint32_t>(), n, dst, n);
        HIPCHK(c, hipGetLastError());
        const uint64_t bounds[2] = {0, total};
        HIPCHK(c, S.list_off.ensure(sizeof bounds));
        HIPCHK(c, hipMemcpyAsync(S.list_off.p, bounds, sizeof bounds, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));
        MultiLists lists;
        lists.list_off = S.list_off.as<uint64_t>(); lists.n_lists = 1; lists.n_keys = total;
        const uint64_t keys_before = b->total_keys;
        grm_kmer_set *set = nullptr;
        rc = multi_set(b, b->k, filter_singleton ? 2u : 1u, &set, &lists);
        b->total_keys = keys_before;
        if (rc) { delete set; return rc; }
        S.n_global = set->n;
        hipError_t e = S.gkeys.ensure((set->n + 1) * 8 * (size_t)W);
        if (e == hipSuccess && set->n) e = hipMemcpyAsync(S.gkeys.p, set->d_kmers.p, set->n * 8 * (size_t)W, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        delete set;
        if (e != hipSuccess) return fail(c, GRM_ERR_HIP, "global dictionary copy: %s", hipGetErrorString(e));
    }
    b->have_global = true;
    if (n_kmers) *n_kmers = S.n_global;
    return GRM_OK;
}
static int sorted_stage_fill(grm_batch *b, grm_matrix **out)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    SortedStage &S = *b->sorted;
    const int W = S.words;
    grm_matrix *m = new grm_matrix();
    m->ctx = c; m->k = b->k; m->words = W; m->n_genomes = b->n_genomes;
    m->n_rows = ((size_t)b->n_genomes + 63) / 64;
    m->n_kmers = S.n_global;
    const size_t cells = m->n_rows * m->n_kmers;
    auto bail = [&](int code) { delete m; return code; };
    if (m->d_data.alloc(cells * 8) != hipSuccess || m->d_kmers.alloc((m->n_kmers + 1) * 8 * (size_t)W) != hipSuccess)
        return bail(fail(c, GRM_ERR_OOM, "matrix allocation failed (%zu cells)", cells));
    if (S.col.ensure(((size_t)b->n_local + 1) * 4) != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "alloc"));
    if (cells) (void)hipMemsetAsync(m->d_data.p, 0, cells * 8, s);
    if (m->n_kmers) (void)hipMemcpyAsync(m->d_kmers.p, S.gkeys.p, m->n_kmers * 8 * (size_t)W, hipMemcpyDeviceToDevice, s);
    if (cells && b->n_local) {
        TimeScope t(c, "multi_fill", (uint64_t)b->n_local * m->n_rows);
        launch_multi_lookup(s, W, S.own->d_kmers.as<uint64_t>(), b->n_local, S.gkeys.as<uint64_t>(), S.n_global, S.col.as<uint32_t>());
        launch_multi_scatter_cols(s, S.own->d_data.as<uint64_t>(), b->n_local, S.col.as<uint32_t>(), m->n_rows, m->d_data.as<uint64_t>(), m->n_kmers);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "multi_fill failed"));
    *out = m;
    return GRM_OK;
}

// ---- two-word k-mers: hash-partition pipeline (grm_wide_hash.hip) ------------------------------
// Three stages, as for one-word k-mers: local (partition + per-bucket dictionary of this batch's
// genomes), global (sort / merge / filter of the k-mers of every rank), fill.
struct WideHash {
    DevBuf counts, off, cursor1, keys, keys1, len, recs, recs2;
    DevBuf stage_lo, stage_hi, stage_flags, stage_cnt, stage_off, matrix_s, birth, entry_col, entry_major, flag;
    DevBuf loc_lo, loc_hi, loc_flags, idx0, idx1, idx2, t_a, t_b, keep, pos, tmp, g_hi, g_lo;
    int sb = 0, sb_hint = -1, sb_hint_k = 0, sb_hint_bb = -1;
    uint32_t cap_log2 = 0, n_wg = 0, U = 0;
    uint64_t n_local = 0, n_sorted = 0;
    uint64_t seg_stride = 0;       // 0: dense layout (off); else slack layout (segment i at i * seg_stride, length len[i])
    bool slack_failed = false, have_bits = false, own_dict = false, rec_failed = false;
    bool have_local = false, have_global = false;
};
static void wide_hash_free(WideHash *w) { delete w; }

// returns GRM_OK with *fallback = true when the input does not suit this path (the caller then
// uses the sort-based path); parse must have run.
static int wide_hash_local(grm_batch *b, int k, bool *fallback)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    *fallback = false;
    const uint32_t G = (uint32_t)b->n_genomes;
    if (!b->whash) b->whash = new WideHash();
    WideHash &W = *b->whash;
    W.have_local = W.have_global = false;
    if (b->total_syms == 0 || G == 0) { *fallback = true; return GRM_OK; }
    static bool lds_attr_set = false;
    if (!lds_attr_set) { HIPCHK(c, wh_set_max_dynamic_lds()); lds_attr_set = true; }
    uint64_t max_g = 0;
    for (uint32_t g = 0; g < G; g++) max_g = std::max(max_g, b->h_genome_sym_off[g + 1] - b->h_genome_sym_off[g]);
    int bb = 0;
    while (bb < MAX_HIST_BITS && (max_g >> bb) > 512) bb++;
    if (c->opt_bucket_bits >= 0) bb = std::min(c->opt_bucket_bits, MAX_HIST_BITS);
    if ((max_g >> bb) > 4096) { *fallback = true; return GRM_OK; }        // deep inputs: not on this path yet
    const uint32_t cap_log2 = c->opt_cap_log2 > 0 ? (uint32_t)std::min(12, std::max(6, c->opt_cap_log2)) : 11u;
    const uint32_t cap = 1u << cap_log2;
    const uint64_t B = 1ull << bb, n_seg = (uint64_t)G * B;
    const int b1 = scatter_b1_bits(bb);
    const uint64_t n_coarse = (uint64_t)G << b1;
    b->bb = bb;
    b->k = k;
    W.cap_log2 = cap_log2;

    KmerLaunch L;
    L.sym2 = b->d_sym2.as<uint64_t>(); L.inv = b->d_inv.as<uint64_t>(); L.total_syms = b->total_syms;
    L.genome_sym_off = b->d_genome_sym_off.as<uint64_t>(); L.n_genomes = G; L.k = k; L.bb = bb; L.groups_per_thread = 1;

    HIPCHK(c, W.cursor1.ensure(std::max<uint64_t>(n_coarse, (uint64_t)G << 6) * 4)); HIPCHK(c, W.flag.ensure(32));
    // ---- record form (grm_superkmer.hip): the k-mers travel as 24-byte run records --
    // runs of up to 22 k-mers that share the minimizer among the m-mers in the middle of the k-mer: 2.1 B per k-mer through level 1
    // and level 2 instead of 16 B through wh_scatter_l1 / l2, and wh_dict_build cuts the k-mers out of the records ----
    bool by_recs = false;
    int rec_pbits = 0;
    if (c->opt_records != 0 && c->opt_dense_layout <= 0 && !W.rec_failed && k <= 64) {
        // minimizer buckets are less even than hashed k-mers, and a bucket's DISTINCT records must fit wh_dict_build's memo (96 records of up
        // to 22 k-mers): two more bits than the key form -- at 2^14 buckets a third of them held more records than the memo (78 on
        // average at 500 x 5 Mbp) and every occurrence of the rest went the direct way: 7.85 ms against 6.55 ms at 2^15
        int bbr = c->opt_bucket_bits >= 0 ? bb : bb + (c->opt_rec_bucket_shift >= 0 ? c->opt_rec_bucket_shift : 2);
        bbr = std::min(bbr, superkmer_coarse_bits(bbr) + 7);
        const int b1r = superkmer_coarse_bits(bbr);
        // one workgroup of level 1 per genome part: enough parts to fill the device when genomes are few (as batch_partition_impl)
        int pbits = 0;
        while (((uint64_t)G << pbits) < 256 && pbits < 6 && (max_g >> (pbits + 1)) >= 65536) pbits++;
        if (c->opt_rec_part_bits >= 0) pbits = std::min(c->opt_rec_part_bits, 6);
        const uint64_t n_parts = (uint64_t)G << pbits;
        const uint64_t n_regions = n_parts << b1r, n_seg_r = n_parts << bbr;
        const int w = superkmer_wide_window(k);
        const double mean_k = (double)((max_g >> pbits) >> b1r) + 1.0;
        const double mean_r = mean_k * (2.0 / (w + 1) + 0.005);
        const uint64_t rstride64 = (uint64_t)(mean_r * 1.02 + 14.0 * std::sqrt(mean_r) + 32.0 + 15.0) / 16 * 16;
        const double expected_records = (double)b->total_syms * (2.0 / (w + 1) + 0.005);
        bool rec = n_seg_r < 0xffffffffull && (rstride64 << b1r) < 0xffffffffull && mean_r / (double)(1u << (bbr - b1r)) < 16000.0 &&
                   (double)n_regions * (double)rstride64 <= 4.0 * expected_records + 16.0 * 1024 * 1024 &&
                   ((size_t)1 << bbr) * (((size_t)G + 63) / 64) * cap * 8 <= MATRIX_S_LIMIT;
        if (rec) {
            hipError_t e = W.recs.ensure((n_regions * rstride64 + 4) * 24);
            if (e == hipSuccess) e = W.recs2.ensure((n_regions * rstride64 + 4) * 24);
            if (e == hipSuccess) e = W.off.ensure((n_seg_r + 1) * 8);
            if (e == hipSuccess) e = W.len.ensure((n_seg_r + 1) * 4);
            if (e == hipSuccess) e = W.counts.ensure((n_regions + 1) * 4);
            if (e != hipSuccess) { (void)hipGetLastError(); rec = false; }
        }
        if (rec) {
            const uint32_t rstride = (uint32_t)rstride64;
            KmerLaunch Lr = L;
            Lr.bb = bbr;
            HIPCHK(c, hipMemsetAsync(W.flag.p, 0, 32, s));
            {
                TimeScope t(c, "superkmer_l1", b->total_syms);
                launch_superkmer_l1(s, Lr, b1r, pbits, W.recs.p, rstride, W.counts.as<uint32_t>(), W.cursor1.as<uint32_t>(), W.flag.as<int>());
            }
            int l2_idx = -1;
            {
                TimeScope t(c, "superkmer_l2", b->total_syms);
                l2_idx = t.idx;
                launch_superkmer_l2_wide(s, W.recs.p, rstride, W.counts.as<uint32_t>(), n_regions, k, bbr, b1r, W.recs2.p, W.off.as<uint64_t>(),
                                         W.len.as<uint32_t>(), W.flag.as<int>());
            }
            launch_sum_u32(s, W.cursor1.as<uint32_t>(), n_parts, reinterpret_cast<uint64_t *>(W.flag.as<uint8_t>() + 8));
            launch_sum_u32(s, W.counts.as<uint32_t>(), n_regions, reinterpret_cast<uint64_t *>(W.flag.as<uint8_t>() + 16));
            HIPCHK(c, hipGetLastError());
            struct { int over; int pad; uint64_t total; uint64_t records; uint64_t pad2; } h;
            HIPCHK(c, hipMemcpyAsync(&h, W.flag.p, 32, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            if (l2_idx >= 0 && l2_idx < (int)c->recs.size()) c->recs[l2_idx].units = h.records;
            if (!h.over) {
                by_recs = true;
                rec_pbits = pbits;
                bb = bbr;
                b->bb = bbr;
                b->total_keys = h.total;
            } else {
                W.rec_failed = true;         // repeat-rich input: the key form from now on
            }
        }
    }
    // ---- partition: slack layout (no histogram pass; see batch_partition_impl), dense layout as the fallback ----
    W.seg_stride = 0;
    bool slack = !by_recs && c->opt_dense_layout <= 0 && !W.slack_failed;
    uint32_t fine_cap = 0;
    uint64_t region_stride = 0;
    if (slack) {
        const uint64_t m = max_g >> bb;
        fine_cap = (uint32_t)((m + (uint64_t)(6.0 * std::sqrt((double)m + 1.0)) + 32 + 15) / 16 * 16);
        region_stride = (uint64_t)fine_cap << (bb - b1);
        if ((double)n_seg * fine_cap > 1.75 * (double)b->total_syms + 65536.0) slack = false;
    }
    if (slack) {
        const uint64_t layout_keys = n_seg * (uint64_t)fine_cap;
        HIPCHK(c, W.keys.ensure((layout_keys + 4) * 16));
        if (bb > b1) HIPCHK(c, W.keys1.ensure((layout_keys + 4) * 16));
        HIPCHK(c, W.len.ensure((n_seg + 1) * 4));
        HIPCHK(c, hipMemsetAsync(W.cursor1.p, 0, n_coarse * 4, s));
        HIPCHK(c, hipMemsetAsync(W.flag.p, 0, 16, s));
        {
            TimeScope t(c, "wh_scatter_l1", b->total_syms);
            launch_wh_l1(s, L, nullptr, W.cursor1.as<uint32_t>(), bb > b1 ? W.keys1.p : W.keys.p, region_stride, W.flag.as<int>());
        }
        if (bb > b1) {
            TimeScope t(c, "wh_scatter_l2", b->total_syms);
            launch_wh_l2(s, L, nullptr, W.keys1.p, W.keys.p, region_stride, fine_cap, W.cursor1.as<uint32_t>(), W.len.as<uint32_t>(), W.flag.as<int>());
        } else {
            HIPCHK(c, hipMemcpyAsync(W.len.p, W.cursor1.p, n_seg * 4, hipMemcpyDeviceToDevice, s));
        }
        launch_sum_u32(s, W.cursor1.as<uint32_t>(), n_coarse, reinterpret_cast<uint64_t *>(W.flag.as<uint8_t>() + 8));
        HIPCHK(c, hipGetLastError());
        struct { int over; int pad; uint64_t total; } h;
        HIPCHK(c, hipMemcpyAsync(&h, W.flag.p, 16, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (h.over) { W.slack_failed = true; slack = false; }
        else { b->total_keys = h.total; W.seg_stride = fine_cap; }
    }
    if (!slack && !by_recs) {
        HIPCHK(c, W.counts.ensure((n_seg + 1) * 4)); HIPCHK(c, W.off.ensure((n_seg + 1) * 8));
        HIPCHK(c, hipMemsetAsync(W.counts.p, 0, (n_seg + 1) * 4, s));
        HIPCHK(c, hipMemsetAsync(W.cursor1.p, 0, n_coarse * 4, s));
        {
            TimeScope t(c, "wh_hist", b->total_syms);
            launch_wh_hist(s, L, W.counts.as<uint32_t>());
        }
        {
            size_t tb = 0;
            HIPCHK(c, exclusive_scan_u32_u64(s, W.counts.as<uint32_t>(), W.off.as<uint64_t>(), n_seg + 1, nullptr, tb));
            HIPCHK(c, W.tmp.ensure(tb));
            HIPCHK(c, exclusive_scan_u32_u64(s, W.counts.as<uint32_t>(), W.off.as<uint64_t>(), n_seg + 1, W.tmp.p, tb));
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&b->total_keys, W.off.as<uint64_t>() + n_seg, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        const uint64_t NKd = b->total_keys;
        HIPCHK(c, W.keys.ensure((NKd + 4) * 16));
        if (bb > b1) HIPCHK(c, W.keys1.ensure((NKd + 4) * 16));
        {
            TimeScope t(c, "wh_scatter_l1", NKd);
            launch_wh_l1(s, L, W.off.as<uint64_t>(), W.cursor1.as<uint32_t>(), bb > b1 ? W.keys1.p : W.keys.p, 0, nullptr);
        }
        if (bb > b1) {
            TimeScope t(c, "wh_scatter_l2", NKd);
            launch_wh_l2(s, L, W.off.as<uint64_t>(), W.keys1.p, W.keys.p, 0, 0, nullptr, nullptr, nullptr);
        }
        HIPCHK(c, hipGetLastError());
    }
    const uint64_t NK = b->total_keys;
    SegLayout seg;
    if (by_recs) { seg.off = W.off.as<uint64_t>(); seg.len = W.len.as<uint32_t>(); seg.stride = 0; }         // (in records)
    else if (W.seg_stride) { seg.off = nullptr; seg.len = W.len.as<uint32_t>(); seg.stride = W.seg_stride; }
    else { seg.off = W.off.as<uint64_t>(); seg.len = nullptr; seg.stride = 0; }

    // ---- per-bucket dictionary + presence bits; the sub-bucket count jumps to what a failed launch asked for ----
    const size_t n_rows = ((size_t)G + 63) / 64;
    if (n_rows > 0xffffu) { *fallback = true; return GRM_OK; }
    const uint32_t max_fill = cap - (cap >> 3);
    int sb = c->opt_sub_bits >= 0 ? c->opt_sub_bits : 0;
    if (c->opt_sub_bits < 0 && W.sb_hint >= 0 && W.sb_hint_k == k && W.sb_hint_bb == bb) sb = W.sb_hint;
    uint32_t n_wg = 0;
    bool bits = false;
    for (int attempt = 0;; attempt++) {
        if (bb + sb > 22 || attempt > 8) { *fallback = true; return GRM_OK; }
        n_wg = 1u << (bb + sb);
        const size_t slots = (size_t)n_wg * cap;
        HIPCHK(c, W.stage_lo.ensure(slots * 8)); HIPCHK(c, W.stage_hi.ensure(slots * 8)); HIPCHK(c, W.stage_flags.ensure(slots));
        HIPCHK(c, W.stage_cnt.ensure((size_t)n_wg * 4 + 4)); HIPCHK(c, W.stage_off.ensure(((size_t)n_wg + 1) * 8));
        const size_t ms_bytes = slots * n_rows * 8;
        bits = ms_bytes <= MATRIX_S_LIMIT;
        if (!bits) { *fallback = true; return GRM_OK; }            // no probing fill on this path: the sort-based one takes over
        HIPCHK(c, W.matrix_s.ensure(ms_bytes));
        HIPCHK(c, W.birth.ensure(slots * 2));
        HIPCHK(c, hipMemsetAsync(W.flag.p, 0, 16, s));
        {
            TimeScope t(c, "wh_dict_build", NK);
            launch_wh_dict_build(s, by_recs ? W.recs2.p : W.keys.p, seg, G, bb, sb, cap_log2, W.stage_lo.as<uint64_t>(), W.stage_hi.as<uint64_t>(),
                                 W.stage_flags.as<uint8_t>(), W.stage_cnt.as<uint32_t>(), W.matrix_s.as<uint64_t>(), W.birth.as<uint16_t>(),
                                 W.flag.as<int>(), W.flag.as<uint32_t>() + 1, by_recs ? k : 0, rec_pbits);
        }
        HIPCHK(c, hipGetLastError());
        struct { int over; uint32_t need; } h;
        HIPCHK(c, hipMemcpyAsync(&h, W.flag.p, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (!h.over) break;
        int step = 1;
        while (step < 22 && ((uint64_t)h.need >> step) > (uint64_t)max_fill * 7 / 10) step++;
        sb += step;
    }
    if (c->opt_sub_bits < 0) { W.sb_hint = sb; W.sb_hint_k = k; W.sb_hint_bb = bb; }
    W.have_bits = bits;
    uint64_t n_local = 0;
    launch_scan_u32(s, W.stage_cnt.as<uint32_t>(), n_wg, W.stage_off.as<uint64_t>());
    HIPCHK(c, hipMemcpyAsync(&n_local, W.stage_off.as<uint64_t>() + n_wg, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (n_local >= 0xffffffffull) { *fallback = true; return GRM_OK; }
    W.sb = sb;
    W.n_wg = n_wg;
    W.n_local = n_local;
    HIPCHK(c, W.loc_lo.ensure((n_local + 2) * 8)); HIPCHK(c, W.loc_hi.ensure((n_local + 2) * 8)); HIPCHK(c, W.loc_flags.ensure(n_local + 16));
    if (n_local) {
        TimeScope t(c, "wh_dict_gather", n_local);
        launch_wh_dict_gather(s, W.stage_lo.as<uint64_t>(), W.stage_hi.as<uint64_t>(), W.stage_flags.as<uint8_t>(), W.stage_off.as<uint64_t>(),
                              n_wg, cap, W.loc_lo.as<uint64_t>(), W.loc_hi.as<uint64_t>(), W.loc_flags.as<uint8_t>());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(s));
    }
    W.have_local = true;
    return GRM_OK;
}

// (hi, lo, flag) of every rank (or of this batch alone) -> sorted, merged, filtered dictionary:
// sorted keys stay in t_b (hi) / t_a (lo), keep / pos give the columns, W.U their number.
static int wide_hash_global(grm_batch *b, const uint64_t *hi, const uint64_t *lo, const uint8_t *flags, uint64_t n, int filter_singleton)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    WideHash &W = *b->whash;
    W.have_global = false;
    W.U = 0;
    W.n_sorted = n;
    if (n >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "dictionary of %llu k-mers exceeds 2^32-1 entries", (unsigned long long)n);
    if (n) {
        TimeScope t(c, "wh_dict_sort", n);
        hipError_t e = hipSuccess;
        auto need = [&](DevBuf &d, size_t bytes) { if (e == hipSuccess) e = d.ensure(bytes); };
        need(W.idx0, n * 4); need(W.idx1, n * 4); need(W.idx2, n * 4); need(W.t_a, n * 8); need(W.t_b, n * 8);
        need(W.keep, (n + 1) * 4); need(W.pos, (n + 1) * 4);
        if (e != hipSuccess) return fail(c, GRM_ERR_OOM, "dictionary buffers: %s", hipGetErrorString(e));
        // by key ranges of the top 64 bits, ties by the low word (grm_wide_hash.hip); the two-pass radix sort below when a range is too
        // crowded for its LDS sort (or asked for: "dict_sort_prim")
        bool sorted = false;
        if (c->opt_dict_sort_prim <= 0) {
            need(W.tmp, dict_sort_scratch_bytes(n));
            need(W.flag, 32);
            if (e != hipSuccess) return fail(c, GRM_ERR_OOM, "dictionary buffers: %s", hipGetErrorString(e));
            HIPCHK(c, hipMemsetAsync(W.flag.p, 0, 4, s));
            launch_wh_top64(s, hi, lo, n, b->k, W.t_a.as<uint64_t>());
            HIPCHK(c, launch_dict_sort(s, W.t_a.as<uint64_t>(), n, 64, W.t_b.as<uint64_t>(), W.idx2.as<uint32_t>(), W.tmp.p, W.flag.as<int>()));
            int too_big = 0;
            HIPCHK(c, hipMemcpyAsync(&too_big, W.flag.p, 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipStreamSynchronize(s));
            if (!too_big) {
                // (W.flag is zero here; a tie group too long for the one-thread ordering raises it: the radix sort below then)
                launch_wh_ties(s, W.t_b.as<uint64_t>(), W.idx2.as<uint32_t>(), lo, n, W.flag.as<int>());
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipMemcpyAsync(&too_big, W.flag.p, 4, hipMemcpyDeviceToHost, s));
                HIPCHK(c, hipStreamSynchronize(s));
            }
            if (!too_big) {
                launch_gather_u64(s, hi, W.idx2.as<uint32_t>(), n, W.t_b.as<uint64_t>());
                launch_gather_u64(s, lo, W.idx2.as<uint32_t>(), n, W.t_a.as<uint64_t>());     // sorted: hi = t_b, lo = t_a
                HIPCHK(c, hipGetLastError());
                sorted = true;
            }
        }
        if (!sorted) {
        launch_iota_u32(s, W.idx0.as<uint32_t>(), n);
        size_t tb = 0;
        // stable LSD sort by (hi, lo): by lo, then by hi
        e = sort_pairs_u64_u32(s, lo, W.t_a.as<uint64_t>(), W.idx0.as<uint32_t>(), W.idx1.as<uint32_t>(), n, nullptr, tb);
        if (e == hipSuccess) e = W.tmp.ensure(tb);
        if (e == hipSuccess) e = sort_pairs_u64_u32(s, lo, W.t_a.as<uint64_t>(), W.idx0.as<uint32_t>(), W.idx1.as<uint32_t>(), n, W.tmp.p, tb);
        launch_gather_u64(s, hi, W.idx1.as<uint32_t>(), n, W.t_a.as<uint64_t>());
        tb = 0;
        if (e == hipSuccess) e = sort_pairs_u64_u32(s, W.t_a.as<uint64_t>(), W.t_b.as<uint64_t>(), W.idx1.as<uint32_t>(), W.idx2.as<uint32_t>(), n, nullptr, tb);
        if (e == hipSuccess) e = W.tmp.ensure(tb);
        if (e == hipSuccess) e = sort_pairs_u64_u32(s, W.t_a.as<uint64_t>(), W.t_b.as<uint64_t>(), W.idx1.as<uint32_t>(), W.idx2.as<uint32_t>(), n, W.tmp.p, tb);
        launch_gather_u64(s, lo, W.idx2.as<uint32_t>(), n, W.t_a.as<uint64_t>());     // sorted: hi = t_b, lo = t_a
        if (e != hipSuccess) return fail(c, GRM_ERR_HIP, "dictionary sort: %s", hipGetErrorString(e));
        }
        HIPCHK(c, hipMemsetAsync(W.keep.as<uint32_t>() + n, 0, 4, s));
        // a k-mer held by several ranks shows up as a run of equal keys: one column, carried by >= 2 genomes
        launch_wh_mark(s, W.t_b.as<uint64_t>(), W.t_a.as<uint64_t>(), flags, W.idx2.as<uint32_t>(), n, filter_singleton, W.keep.as<uint32_t>());
        int rc = wide_scan(c, W.tmp, false, W.keep.as<uint32_t>(), W.pos.as<uint32_t>(), n + 1);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(&W.U, W.pos.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    // every local entry learns its column: from the sort's order when the sorted entries are the local ones (one GPU),
    // else by a binary search of its (hi, lo) among the sorted entries of all ranks
    W.own_dict = n && hi == W.loc_hi.as<uint64_t>() && lo == W.loc_lo.as<uint64_t>() && n == W.n_local;
    HIPCHK(c, W.entry_col.ensure((W.n_local + 1) * 4));
    if (W.n_local && W.own_dict) {
        TimeScope t(c, "wh_entry_cols", W.n_local);
        launch_wh_cols_from_order(s, W.idx2.as<uint32_t>(), W.keep.as<uint32_t>(), W.pos.as<uint32_t>(), n, W.entry_col.as<uint32_t>());
        HIPCHK(c, hipGetLastError());
    } else if (W.n_local) {
        TimeScope t(c, "wh_entry_cols", W.n_local);
        if (n) launch_wh_entry_cols(s, W.t_b.as<uint64_t>(), W.t_a.as<uint64_t>(), W.keep.as<uint32_t>(), W.pos.as<uint32_t>(), n,
                                    W.loc_hi.as<uint64_t>(), W.loc_lo.as<uint64_t>(), W.n_local, W.entry_col.as<uint32_t>());
        else HIPCHK(c, hipMemsetAsync(W.entry_col.p, 0xff, W.n_local * 4, s));
        HIPCHK(c, hipGetLastError());
    }
    W.have_global = true;
    return GRM_OK;
}

static int wide_hash_fill(grm_batch *b, grm_matrix **out)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    WideHash &W = *b->whash;
    grm_matrix *m = new grm_matrix();
    m->ctx = c; m->k = b->k; m->words = 2; m->n_genomes = b->n_genomes; m->n_rows = ((size_t)b->n_genomes + 63) / 64;
    auto bail = [&](int code) { delete m; return code; };
    const uint32_t U = W.U;
    m->n_kmers = U;
    const size_t cells = m->n_rows * (size_t)U;
    if (m->d_data.alloc(cells * 8) != hipSuccess || m->d_kmers.alloc(((size_t)U + 1) * 16) != hipSuccess)
        return bail(fail(c, GRM_ERR_OOM, "matrix allocation failed"));
    if (U) {
        launch_wh_select(s, W.t_b.as<uint64_t>(), W.t_a.as<uint64_t>(), W.keep.as<uint32_t>(), W.pos.as<uint32_t>(), W.n_sorted, m->d_kmers.as<uint64_t>());
        if (cells && b->total_keys && W.have_bits) {
            // the shared fill: entry-major lines + transpose from 4 word-rows up (zeroed when columns may lack a local entry)
            uint64_t *em = nullptr;
            const bool two_step = m->n_rows >= 4 && c->opt_direct_permute <= 0;
            if (two_step) {
                if (W.entry_major.ensure(cells * 8) != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "matrix_fill: entry-major scratch"));
                em = W.entry_major.as<uint64_t>();
                if (!W.own_dict) (void)hipMemsetAsync(em, 0, cells * 8, s);
            } else {
                (void)hipMemsetAsync(m->d_data.p, 0, cells * 8, s);
            }
            TimeScope t(c, "matrix_fill", (uint64_t)W.n_local * m->n_rows);
            launch_matrix_permute(s, W.matrix_s.as<uint64_t>(), W.birth.as<uint16_t>(), W.stage_off.as<uint64_t>(), W.stage_cnt.as<uint32_t>(),
                                  W.entry_col.as<uint32_t>(), W.n_wg, (uint32_t)m->n_rows, W.cap_log2, m->d_data.as<uint64_t>(), U, em);
        } else if (cells) {
            (void)hipMemsetAsync(m->d_data.p, 0, cells * 8, s);
        }
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "wide hash pipeline failed"));
    *out = m;
    return GRM_OK;
}

static int wide_hash_matrix(grm_batch *b, int k, int filter_singleton, grm_matrix **out, bool *fallback)
{
    int rc = wide_hash_local(b, k, fallback);
    if (rc || *fallback) return rc;
    WideHash &W = *b->whash;
    rc = wide_hash_global(b, W.loc_hi.as<uint64_t>(), W.loc_lo.as<uint64_t>(), W.loc_flags.as<uint8_t>(), W.n_local, filter_singleton);
    if (rc) return rc;
    return wide_hash_fill(b, out);
}

// ---- staged form (grm_batch_partition / local_dict / export_dict / set_global_dict / fill) ----
static int wide_stage_local(grm_batch *b, int k)
{
    grm_ctx *c = b->ctx;
    bool fallback = false;
    b->partitioned = b->have_local = b->have_global = false;
    int rc = wide_hash_local(b, k, &fallback);
    if (rc) return rc;
    if (fallback && (b->total_syms == 0 || b->n_genomes == 0)) {       // nothing to count: an empty local dictionary
        WideHash &W = *b->whash;
        W.n_local = 0; W.n_wg = 1; W.sb = 0; W.cap_log2 = 6;
        b->total_keys = 0;
        b->k = k;
        b->bb = 0;
        HIPCHK(c, W.loc_lo.ensure(16)); HIPCHK(c, W.loc_hi.ensure(16)); HIPCHK(c, W.loc_flags.ensure(16));
        HIPCHK(c, W.stage_off.ensure(16)); HIPCHK(c, W.stage_cnt.ensure(16));
        HIPCHK(c, hipMemsetAsync(W.stage_off.p, 0, 16, c->stream));
        HIPCHK(c, hipMemsetAsync(W.stage_cnt.p, 0, 16, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        W.have_bits = false;
        W.have_local = true;
    } else if (fallback) {
        return fail(c, GRM_ERR_UNSUPPORTED, "k=%d: this input is too deep for the staged two-word pipeline (use grm_batch_run)", k);
    }
    b->partitioned = true;
    b->have_local = true;
    return GRM_OK;
}
static int wide_stage_n_local(grm_batch *b, uint64_t *n_local)
{
    if (!b->whash || !b->whash->have_local) return fail(b->ctx, GRM_ERR_STATE, "grm_batch_local_dict before grm_batch_partition");
    b->n_local = b->whash->n_local;
    if (n_local) *n_local = b->n_local;
    return GRM_OK;
}
// keys leave as (hi, lo) pairs, 16 bytes per k-mer
static int wide_stage_export(grm_batch *b, void *dev_keys_out, void *dev_flags_out)
{
    grm_ctx *c = b->ctx;
    WideHash &W = *b->whash;
    if (W.n_local) {
        launch_join_pairs_u64(c->stream, W.loc_hi.as<uint64_t>(), W.loc_lo.as<uint64_t>(), W.n_local, (uint64_t *)dev_keys_out);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(dev_flags_out, W.loc_flags.p, W.n_local, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GRM_OK;
}
// the local (hi, lo) lists are already in workgroup (= bucket) order; bucket offsets from the staging scan
static int wide_stage_export_ordered(grm_batch *b, uint8_t *rec, uint64_t flags_off, uint64_t boff_off)
{
    grm_ctx *c = b->ctx;
    WideHash &W = *b->whash;
    int rc = wide_stage_export(b, rec, rec + flags_off);
    if (rc) return rc;
    const uint32_t B = 1u << b->bb;
    if (!W.n_local) {
        HIPCHK(c, hipMemsetAsync(rec + boff_off, 0, ((size_t)B + 1) * 4, c->stream));
    } else {
        launch_bucket_offsets(c->stream, W.stage_off.as<uint64_t>(), W.sb, B, (uint32_t *)(rec + boff_off));
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GRM_OK;
}
static int wide_stage_global(grm_batch *b, const void *dev_keys, const void *dev_flags, uint64_t n, int filter_singleton, uint64_t *n_kmers)
{
    grm_ctx *c = b->ctx;
    if (!b->whash || !b->whash->have_local) return fail(c, GRM_ERR_STATE, "grm_batch_set_global_dict before grm_batch_partition");
    WideHash &W = *b->whash;
    b->have_global = false;
    b->filter_singleton = filter_singleton;
    HIPCHK(c, W.g_hi.ensure((n + 2) * 8)); HIPCHK(c, W.g_lo.ensure((n + 2) * 8));
    if (n) {
        launch_split_pairs_u64(c->stream, (const uint64_t *)dev_keys, n, W.g_hi.as<uint64_t>(), W.g_lo.as<uint64_t>());
        HIPCHK(c, hipGetLastError());
    }
    int rc = wide_hash_global(b, W.g_hi.as<uint64_t>(), W.g_lo.as<uint64_t>(), (const uint8_t *)dev_flags, n, filter_singleton);
    if (rc) return rc;
    b->n_dict = W.U;
    b->have_global = true;
    if (n_kmers) *n_kmers = W.U;
    return GRM_OK;
}
static int wide_stage_fill(grm_batch *b, grm_matrix **out)
{
    if (!b->whash || !b->whash->have_global) return fail(b->ctx, GRM_ERR_STATE, "grm_batch_fill before grm_batch_set_global_dict");
    return wide_hash_fill(b, out);
}

static int wide_matrix(grm_batch *b, int k, uint32_t abundance_min, int filter_singleton, grm_matrix **out, bool preloaded = false)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    if (!b->wide) b->wide = new WideSorted();
    WideSorted &W = *b->wide;
    int rc = wide_sort_and_mark(b, k, abundance_min, W, preloaded);
    if (rc) return rc;
    grm_matrix *m = new grm_matrix();
    m->ctx = c; m->k = k; m->words = 2; m->n_genomes = b->n_genomes;
    m->n_rows = ((size_t)b->n_genomes + 63) / 64;
    auto bail = [&](int code) { delete m; return code; };
    uint32_t U = 0;
    DevBuf &key_incl = W.key_incl, &carriers = W.carriers, &keep = W.keep, &col = W.col;
    if (W.n_sub) {
        TimeScope t(c, "wide_reduce", W.n_sub);
        if (key_incl.ensure((size_t)W.n_sub * 4) != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "alloc"));
        rc = wide_scan(c, W.tmp, true, W.sub_key_head.as<uint32_t>(), key_incl.as<uint32_t>(), W.n_sub);
        if (rc) return bail(rc);
        uint32_t n_keys = 0;
        if (hipMemcpyAsync(&n_keys, key_incl.as<uint32_t>() + (W.n_sub - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "D2H"));
        if (carriers.ensure((size_t)n_keys * 4 + 4) != hipSuccess || keep.ensure((size_t)n_keys * 4 + 4) != hipSuccess ||
            col.ensure((size_t)n_keys * 4 + 4) != hipSuccess) return bail(fail(c, GRM_ERR_OOM, "alloc"));
        (void)hipMemsetAsync(carriers.p, 0, (size_t)n_keys * 4 + 4, s);
        launch_wide_key_count(s, key_incl.as<uint32_t>(), W.sub_ok.as<uint32_t>(), W.n_sub, carriers.as<uint32_t>());
        launch_wide_keep(s, carriers.as<uint32_t>(), n_keys, filter_singleton ? 2u : 1u, keep.as<uint32_t>());
        rc = wide_scan(c, W.tmp, false, keep.as<uint32_t>(), col.as<uint32_t>(), n_keys);
        if (rc) return bail(rc);
        uint32_t last_col = 0, last_keep = 0;
        if (hipMemcpyAsync(&last_col, col.as<uint32_t>() + (n_keys - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&last_keep, keep.as<uint32_t>() + (n_keys - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "D2H"));
        U = last_col + last_keep;
    }
    m->n_kmers = U;
    const size_t cells = m->n_rows * (size_t)U;
    if (m->d_data.alloc(cells * 8) != hipSuccess || m->d_kmers.alloc(((size_t)U + 1) * 16) != hipSuccess)
        return bail(fail(c, GRM_ERR_OOM, "matrix allocation failed"));
    if (cells) (void)hipMemsetAsync(m->d_data.p, 0, cells * 8, s);
    if (U) {
        TimeScope t(c, "wide_emit", W.n_sub);
        launch_wide_emit(s, W.khi(), W.klo(), W.pos(), b->d_genome_sym_off.as<uint64_t>(),
                         (uint32_t)b->n_genomes, W.sub_start.as<uint32_t>(), W.sub_key_head.as<uint32_t>(), W.sub_ok.as<uint32_t>(),
                         key_incl.as<uint32_t>(), keep.as<uint32_t>(), col.as<uint32_t>(), W.n_sub, m->d_kmers.as<uint64_t>(),
                         m->d_data.as<uint64_t>(), U);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bail(fail(c, GRM_ERR_HIP, "wide_emit failed"));
    *out = m;
    return GRM_OK;
}

// single-genome batch -> sorted counted set (two words per k-mer)
static int wide_set(grm_batch *b, int k, uint32_t abundance_min, grm_kmer_set **out)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    if (!b->wide) b->wide = new WideSorted();
    WideSorted &W = *b->wide;
    int rc = wide_sort_and_mark(b, k, abundance_min, W);
    if (rc) return rc;
    grm_kmer_set *set = new grm_kmer_set();
    set->ctx = c; set->k = k; set->words = 2; set->occurrences = b->total_keys;
    *out = set;
    if (!W.n_sub) return GRM_OK;
    DevBuf &opos = W.opos;
    DevBuf &dk = set->d_kmers, &dc = set->d_counts;
    HIPCHK(c, opos.ensure((size_t)W.n_sub * 4));
    rc = wide_scan(c, W.tmp, false, W.sub_ok.as<uint32_t>(), opos.as<uint32_t>(), W.n_sub);
    if (rc) return rc;
    uint32_t last_pos = 0, last_ok = 0;
    HIPCHK(c, hipMemcpyAsync(&last_pos, opos.as<uint32_t>() + (W.n_sub - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&last_ok, W.sub_ok.as<uint32_t>() + (W.n_sub - 1), 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const size_t n_out = (size_t)last_pos + last_ok;
    if (!n_out) return GRM_OK;
    HIPCHK(c, dk.alloc(n_out * 16)); HIPCHK(c, dc.alloc(n_out * 4));
    launch_wide_set(s, W.khi(), W.klo(), W.sub_start.as<uint32_t>(), W.sub_ok.as<uint32_t>(),
                    opos.as<uint32_t>(), W.n_sub, dk.as<uint64_t>(), dc.as<uint32_t>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    set->n = n_out;
    set->on_device = true;
    set->on_host = false;
    return GRM_OK;
}

extern "C" int grm_batch_run(grm_batch *b, int k, uint32_t abundance_min, int filter_singleton, grm_matrix **out)
{
    if (!b || !out) return GRM_ERR_ARG;
    int rc = batch_partition_impl(b, k, abundance_min, false);
    if (rc) return rc;
    if (k > 64) return multi_matrix(b, k, abundance_min < 1 ? 1 : abundance_min, filter_singleton, out);
    if (k > 32) {
        // hash-partition pipeline when it applies (abundance-min 1, moderate depth); the sort-based path
        // is the general fallback (and can be forced with the "wide_sort" option, for tests)
        if (abundance_min <= 1 && c_opt_wide_sort(b->ctx) <= 0) {
            bool fallback = false;
            rc = wide_hash_matrix(b, k, filter_singleton, out, &fallback);
            if (rc || !fallback) return rc;
        }
        return wide_matrix(b, k, abundance_min < 1 ? 1 : abundance_min, filter_singleton, out);
    }
    uint64_t n_local = 0, n_kmers = 0;
    rc = grm_batch_local_dict(b, &n_local);
    if (rc) return rc;
    rc = grm_batch_set_global_dict(b, b->d_local_keys.p, b->d_local_flags.p, n_local, filter_singleton, &n_kmers);
    if (rc) return rc;
    return grm_batch_fill(b, out);
}

// sorted (k-mer, count) list of one genome after partition(+dedup)
static int genome_set_impl(grm_batch *b, int g, bool have_counts, grm_kmer_set **out)
{
    grm_ctx *c = b->ctx;
    hipStream_t s = c->stream;
    const uint64_t B = 1ull << b->bb;
    grm_kmer_set *set = new grm_kmer_set();
    set->ctx = c;
    set->k = b->k;
    set->words = 1;
    *out = set;
    if (b->total_keys == 0) return GRM_OK;
    std::vector<uint64_t> off(B + 1), dst(B + 1);
    std::vector<uint32_t> len(B);
    if (b->seg_stride) {
        // slack layout: computed offsets, lengths from the partition (or the dedup)
        for (uint64_t i = 0; i <= B; i++) off[i] = ((uint64_t)g * B + i) * b->seg_stride;
        HIPCHK(c, hipMemcpy(len.data(), b->d_len.as<uint32_t>() + (uint64_t)g * B, B * 4, hipMemcpyDeviceToHost));
        set->occurrences = 0;        // filled below when the batch kept the pre-dedup counts
    } else if (b->rec_mode) {
        // key segments expanded from records: regions leave gaps (lengths from level 2 or the dedup); one part per genome
        if (b->rec_part_bits != 0 || !b->deduped) return fail(c, GRM_ERR_STATE, "internal: a genome's set from record segments needs one part per genome and a dedup");
        HIPCHK(c, hipMemcpy(off.data(), b->d_off.as<uint64_t>() + (uint64_t)g * B, B * 8, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(len.data(), b->d_len.as<uint32_t>() + (uint64_t)g * B, B * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> occ((size_t)1 << b->rec_count_pbits);
        HIPCHK(c, hipMemcpy(occ.data(), b->d_cursor1.as<uint32_t>() + ((size_t)g << b->rec_count_pbits), occ.size() * 4, hipMemcpyDeviceToHost));   // level 1: k-mers of the parts
        set->occurrences = 0;
        for (uint32_t v : occ) set->occurrences += v;
    } else {
        HIPCHK(c, hipMemcpy(off.data(), b->d_off.as<uint64_t>() + (uint64_t)g * B, (B + 1) * 8, hipMemcpyDeviceToHost));
        set->occurrences = off[B] - off[0];
        if (b->deduped) HIPCHK(c, hipMemcpy(len.data(), b->d_len.as<uint32_t>() + (uint64_t)g * B, B * 4, hipMemcpyDeviceToHost));
        else for (uint64_t i = 0; i < B; i++) len[i] = (uint32_t)(off[i + 1] - off[i]);
    }
    uint64_t n = 0;
    for (uint64_t i = 0; i < B; i++) { dst[i] = n; n += len[i]; }
    if (b->seg_stride) {
        // occurrences of this genome = what level 1 put into its coarse regions
        const int b1 = scatter_b1_bits(b->bb);
        std::vector<uint32_t> cur((size_t)1 << b1);
        HIPCHK(c, hipMemcpy(cur.data(), b->d_cursor1.as<uint32_t>() + ((uint64_t)g << b1), cur.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t v : cur) set->occurrences += v;
    }
    if (n == 0) return GRM_OK;
    // scratch is the batch's (grow-only); the sorted result goes straight into the set's own buffers
    DevBuf &d_dst_off = b->t_set_off, &d_len = b->t_set_len, &d_k = b->t_set_k, &d_c = b->t_set_c, &d_tmp = b->t_set_tmp;
    DevBuf &d_src_off = b->t_set_src;
    DevBuf &d_k2 = set->d_kmers, &d_c2 = set->d_counts;
    HIPCHK(c, d_dst_off.ensure(B * 8));
    HIPCHK(c, d_src_off.ensure(B * 8));
    HIPCHK(c, d_len.ensure(B * 4));
    HIPCHK(c, d_k.ensure(n * 8)); HIPCHK(c, d_c.ensure(n * 4));
    HIPCHK(c, d_k2.alloc(n * 8)); HIPCHK(c, d_c2.alloc(n * 4));
    HIPCHK(c, hipMemcpy(d_dst_off.p, dst.data(), B * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_src_off.p, off.data(), B * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_len.p, len.data(), B * 4, hipMemcpyHostToDevice));
    launch_segments_compact(s, b->d_keys.as<uint64_t>(), have_counts ? b->d_kcnt.as<uint32_t>() : nullptr,
                            d_src_off.as<uint64_t>(), d_len.as<uint32_t>(), d_dst_off.as<uint64_t>(),
                            (uint32_t)B, d_k.as<uint64_t>(), d_c.as<uint32_t>());
    size_t tb = 0;
    HIPCHK(c, sort_pairs_u64_u32(s, d_k.as<uint64_t>(), d_k2.as<uint64_t>(), d_c.as<uint32_t>(), d_c2.as<uint32_t>(), n, nullptr, tb));
    HIPCHK(c, d_tmp.ensure(tb));
    HIPCHK(c, sort_pairs_u64_u32(s, d_k.as<uint64_t>(), d_k2.as<uint64_t>(), d_c.as<uint32_t>(), d_c2.as<uint32_t>(), n, d_tmp.p, tb));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));
    set->n = n;
    set->on_device = true;
    set->on_host = false;
    return GRM_OK;
}

extern "C" int grm_batch_genome_set(grm_batch *b, int genome_index, grm_kmer_set **out)
{
    if (!b || !out) return GRM_ERR_ARG;
    grm_ctx *c = b->ctx;
    if (!b->partitioned) return fail(c, GRM_ERR_STATE, "grm_batch_genome_set before partition");
    if (genome_index < 0 || genome_index >= b->n_genomes) return fail(c, GRM_ERR_ARG, "bad genome index");
    if (!b->deduped || !b->d_kcnt.p) return fail(c, GRM_ERR_STATE, "grm_batch_genome_set needs a counting partition (use grm_count_genome)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = genome_set_impl(b, genome_index, true, out);
    if (rc) { grm_kmer_set_free(*out); *out = nullptr; }
    return rc;
}

extern "C" int grm_count_genome_buffers(grm_ctx *c, const void *const *bufs, const size_t *lens, int n_bufs, int k,
                                        uint32_t abundance_min, grm_kmer_set **out)
{
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!out || n_bufs < 0) return fail(c, GRM_ERR_ARG, "grm_count_genome_buffers: bad argument");
    grm_batch *b = nullptr;
    int rc = grm_batch_create(c, 1, &b);
    if (rc) return rc;
    for (int i = 0; i < n_bufs && !rc; i++) rc = grm_batch_add(b, 0, bufs[i], lens[i]);
    if (!rc) rc = grm_batch_upload(b);
    if (!rc) rc = batch_partition_impl(b, k, abundance_min, true);
    if (!rc) rc = k > 64 ? multi_set(b, k, abundance_min < 1 ? 1 : abundance_min, out)
                : k > 32 ? wide_set(b, k, abundance_min < 1 ? 1 : abundance_min, out) : genome_set_impl(b, 0, true, out);
    if (rc && *out) { grm_kmer_set_free(*out); *out = nullptr; }
    grm_batch_free(b);
    return rc;
}

extern "C" int grm_count_genome(grm_ctx *c, const char *const *paths, int n_paths, int k, uint32_t abundance_min,
                                grm_kmer_set **out)
{
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!out || n_paths < 0 || (n_paths && !paths)) return fail(c, GRM_ERR_ARG, "grm_count_genome: bad argument");
    *out = nullptr;
    grm_batch *b = nullptr;
    int rc = grm_batch_create(c, 1, &b);
    if (rc) return rc;
    for (int i = 0; i < n_paths && !rc; i++) rc = grm_batch_add_file(b, 0, paths[i]);
    if (!rc) rc = grm_batch_upload(b);
    if (!rc) rc = batch_partition_impl(b, k, abundance_min, true);
    if (!rc) rc = k > 64 ? multi_set(b, k, abundance_min < 1 ? 1 : abundance_min, out)
                : k > 32 ? wide_set(b, k, abundance_min < 1 ? 1 : abundance_min, out) : genome_set_impl(b, 0, true, out);
    if (rc && *out) { grm_kmer_set_free(*out); *out = nullptr; }
    grm_batch_free(b);
    return rc;
}

// dsk2kover's merge for two-word k-mers: the sets' keys are laid out genome after genome and go
// through the sort-based reduction (sort by 128-bit key, runs -> columns, carriers -> filter).
static int build_matrix_wide(grm_ctx *c, grm_kmer_set *const *sets, int n_genomes, int filter_singleton, grm_matrix **out)
{
    const int k = sets[0]->k;
    std::vector<uint64_t> gko(n_genomes + 1, 0);
    size_t max_g = 0;
    for (int g = 0; g < n_genomes; g++) {
        if (!sets[g] || sets[g]->k != k || sets[g]->words != 2) return fail(c, GRM_ERR_ARG, "grm_build_matrix: sets with different k");
        gko[g + 1] = gko[g] + sets[g]->n;
        if (!sets[g]->on_device) max_g = std::max(max_g, sets[g]->n);
    }
    const uint64_t n = gko[n_genomes];
    grm_batch *b = nullptr;
    int rc = grm_batch_create(c, n_genomes, &b);
    if (rc) return rc;
    auto body = [&]() -> int {
        b->uploaded = true;
        b->k = k;
        b->abundance_min = 1;
        b->total_syms = n;
        b->wide = new WideSorted();
        WideSorted &W = *b->wide;
        if (n >= 0xffffffffull) return fail(c, GRM_ERR_UNSUPPORTED, "k > 32 merge is limited to 2^32-1 set entries (got %llu)", (unsigned long long)n);
        for (int i = 0; i < 2; i++) HIPCHK(c, W.A[i].ensure((n + 2) * 8));
        HIPCHK(c, b->d_genome_sym_off.ensure(((size_t)n_genomes + 1) * 8));
        HIPCHK(c, hipMemcpy(b->d_genome_sym_off.p, gko.data(), ((size_t)n_genomes + 1) * 8, hipMemcpyHostToDevice));
        std::vector<uint64_t> hi(max_g), lo(max_g);
        for (int g = 0; g < n_genomes; g++) {
            const size_t m = sets[g]->n;
            if (!m) continue;
            if (sets[g]->on_device) {          // (hi, lo) pairs in HBM: split on the device
                launch_split_pairs_u64(c->stream, sets[g]->d_kmers.as<uint64_t>(), m, W.A[0].as<uint64_t>() + gko[g], W.A[1].as<uint64_t>() + gko[g]);
                continue;
            }
            const uint64_t *w = sets[g]->kmers.data();
            for (size_t i = 0; i < m; i++) { hi[i] = w[2 * i]; lo[i] = w[2 * i + 1]; }
            HIPCHK(c, hipMemcpy(W.A[0].as<uint64_t>() + gko[g], hi.data(), m * 8, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(W.A[1].as<uint64_t>() + gko[g], lo.data(), m * 8, hipMemcpyHostToDevice));
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return wide_matrix(b, k, 1, filter_singleton, out, true);
    };
    rc = body();
    grm_batch_free(b);
    return rc;
}

// dsk2kover's merge: per-genome sets (host) -> partition the explicit key lists -> same
// dictionary / fill kernels as the fused path.
extern "C" int grm_build_matrix(grm_ctx *c, grm_kmer_set *const *sets, int n_genomes, int filter_singleton,
                                grm_matrix **out)
{
    if (!c) return GRM_ERR_NO_DEVICE;
    if (!out || n_genomes < 0 || (n_genomes && !sets)) return fail(c, GRM_ERR_ARG, "grm_build_matrix: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (n_genomes && sets[0] && sets[0]->words > 2) return build_matrix_multi(c, sets, n_genomes, filter_singleton, out);
    if (n_genomes && sets[0] && sets[0]->words == 2) return build_matrix_wide(c, sets, n_genomes, filter_singleton, out);
    int k = n_genomes ? sets[0]->k : 1;
    std::vector<uint64_t> gko(n_genomes + 1, 0);
    uint64_t max_g = 0;
    for (int g = 0; g < n_genomes; g++) {
        if (!sets[g] || sets[g]->k != k || sets[g]->words != 1) return fail(c, GRM_ERR_ARG, "grm_build_matrix: sets with different k");
        gko[g + 1] = gko[g] + sets[g]->n;
        max_g = std::max<uint64_t>(max_g, sets[g]->n);
    }
    const uint64_t n = gko[n_genomes];
    grm_batch *b = nullptr;
    int rc = grm_batch_create(c, n_genomes, &b);
    if (rc) return rc;
    auto body = [&]() -> int {
        b->uploaded = true;
        b->k = k;
        b->abundance_min = 1;
        b->total_keys = n;
        b->total_syms = n;
        b->bb = pick_bucket_bits(c, max_g);
        b->cap_log2 = pick_cap_log2(c);
        const uint64_t B = 1ull << b->bb, n_seg = (uint64_t)n_genomes * B;
        DevBuf d_in, d_gko;
        HIPCHK(c, d_in.alloc((n + 2) * 8));
        HIPCHK(c, d_gko.alloc((n_genomes + 1) * 8));
        for (int g = 0; g < n_genomes; g++)
            if (sets[g]->n)
                HIPCHK(c, sets[g]->on_device
                              ? hipMemcpyAsync(d_in.as<uint64_t>() + gko[g], sets[g]->d_kmers.p, sets[g]->n * 8, hipMemcpyDeviceToDevice, s)
                              : hipMemcpy(d_in.as<uint64_t>() + gko[g], sets[g]->kmers.data(), sets[g]->n * 8, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(d_gko.p, gko.data(), (n_genomes + 1) * 8, hipMemcpyHostToDevice));
        HIPCHK(c, b->d_counts.ensure((n_seg + 1) * 4));
        HIPCHK(c, b->d_cursor.ensure((n_seg + 1) * 4));
        HIPCHK(c, b->d_off.ensure((n_seg + 2) * 8));
        HIPCHK(c, b->d_keys.ensure((n + 2) * 8));
        HIPCHK(c, hipMemsetAsync(b->d_counts.p, 0, (n_seg + 1) * 4, s));
        HIPCHK(c, hipMemsetAsync(b->d_cursor.p, 0, (n_seg + 1) * 4, s));
        launch_keys_partition_hist(s, d_in.as<uint64_t>(), n, d_gko.as<uint64_t>(), (uint32_t)n_genomes, b->bb, b->d_counts.as<uint32_t>());
        launch_scan_u32(s, b->d_counts.as<uint32_t>(), n_seg, b->d_off.as<uint64_t>());
        launch_keys_partition_scatter(s, d_in.as<uint64_t>(), n, d_gko.as<uint64_t>(), (uint32_t)n_genomes, b->bb,
                                      b->d_off.as<uint64_t>(), b->d_cursor.as<uint32_t>(), b->d_keys.as<uint64_t>());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(s));
        b->partitioned = true;
        uint64_t n_local = 0, n_kmers = 0;
        int r = grm_batch_local_dict(b, &n_local);
        if (r) return r;
        r = grm_batch_set_global_dict(b, b->d_local_keys.p, b->d_local_flags.p, n_local, filter_singleton, &n_kmers);
        if (r) return r;
        return grm_batch_fill(b, out);
    };
    rc = body();
    grm_batch_free(b);
    return rc;
}

// --------------------------------------------------------------------------------------
// writers
// --------------------------------------------------------------------------------------
// w: `words` uint64, most significant first (1 word for k <= 32, 2 for k <= 64)
static inline void decode_kmer(const uint64_t *w, int words, int k, char *out)
{
    static const char L[4] = {'A', 'C', 'T', 'G'};
    for (int i = 0; i < k; i++) {
        const int bit = 2 * (k - 1 - i);                  // position from the least significant end
        const uint64_t word = w[words - 1 - bit / 64];
        out[i] = L[(word >> (bit & 63)) & 3];
    }
}

// rows [col0, col1) of the TSV (a row = one k-mer = one matrix column) into `path` at their final offsets; the header when
// col0 == 0.  whole = true: path is written as <path>.tmp and renamed (grm_write_tsv); false: the file is shared with other
// writers (the ranks of a multi-GPU run each format a slice), opened without truncation, no rename
static int write_tsv_rows(grm_matrix *m, const char *const *genome_ids, const char *path, size_t col0, size_t col1, bool whole)
{
    if (!m || !path || (m->n_genomes && !genome_ids)) return GRM_ERR_ARG;
    grm_ctx *c = m->ctx;
    const uint64_t *kmers = grm_matrix_kmers(m);
    const uint64_t *data = grm_matrix_data(m);
    if (!kmers || !data) return GRM_ERR_HIP;
    std::string tmp = whole ? std::string(path) + ".tmp" : std::string(path);
    int fd = open(tmp.c_str(), O_CREAT | (whole ? O_TRUNC : 0) | O_WRONLY, 0644);
    if (fd < 0) return fail(c, GRM_ERR_IO, "cannot create %s", tmp.c_str());
    std::string header = "kmers";                      // first header cell is forced by dataset/create.py:241
    for (int g = 0; g < m->n_genomes; g++) { header += '\t'; header += genome_ids[g]; }
    header += '\n';
    // every row has the same byte length (create.py:130-137 relies on it), so rows can be formatted
    // and written by all host cores at computed file offsets
    const size_t line_len = (size_t)m->k + 2 * (size_t)m->n_genomes + 1;
    const size_t U = m->n_kmers;
    col1 = std::min(col1, U);
    col0 = std::min(col0, col1);
    std::atomic<int> bad(0);
    if (col0 == 0 && pwrite(fd, header.data(), header.size(), 0) != (ssize_t)header.size()) bad = 1;
    const size_t rows_per_block = std::max<size_t>(1, ((size_t)8 << 20) / line_len);
    const size_t n_blocks = (col1 - col0 + rows_per_block - 1) / rows_per_block;
    std::atomic<size_t> next(0);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    if (nt > n_blocks) nt = (unsigned)std::max<size_t>(1, n_blocks);
    auto work = [&]() {
        std::vector<char> buf(rows_per_block * line_len);
        for (;;) {
            const size_t blk = next.fetch_add(1);
            if (blk >= n_blocks || bad) break;
            const size_t c0 = col0 + blk * rows_per_block, c1 = std::min(col1, c0 + rows_per_block);
            char *p = buf.data();
            for (size_t col = c0; col < c1; col++) {
                decode_kmer(kmers + col * (size_t)m->words, m->words, m->k, p);
                p += m->k;
                for (int g = 0; g < m->n_genomes; g++) {
                    *p++ = '\t';
                    *p++ = ((data[(size_t)(g >> 6) * U + col] >> (63 - (g & 63))) & 1) ? '1' : '0';
                }
                *p++ = '\n';
            }
            const size_t bytes = (size_t)(p - buf.data());
            size_t done = 0;
            while (done < bytes) {
                const ssize_t w = pwrite(fd, buf.data() + done, bytes - done, (off_t)(header.size() + c0 * line_len + done));
                if (w <= 0) { bad = 1; break; }
                done += (size_t)w;
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work);
    for (auto &t : th) t.join();
    if (close(fd) != 0) bad = 1;
    if (bad) { if (whole) remove(tmp.c_str()); return fail(c, GRM_ERR_IO, "write to %s failed", tmp.c_str()); }
    if (whole && rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return fail(c, GRM_ERR_IO, "rename to %s failed", path); }
    return GRM_OK;
}

extern "C" int grm_write_tsv(grm_matrix *m, const char *const *genome_ids, const char *path)
{
    return write_tsv_rows(m, genome_ids, path, 0, (size_t)-1, true);
}
extern "C" int grm_write_tsv_slice(grm_matrix *m, const char *const *genome_ids, const char *path, uint64_t first_kmer, uint64_t n_kmers)
{
    return write_tsv_rows(m, genome_ids, path, (size_t)first_kmer, (size_t)(first_kmer + n_kmers), false);
}
