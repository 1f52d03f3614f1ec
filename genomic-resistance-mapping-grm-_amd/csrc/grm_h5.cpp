// grm_h5.cpp -- Kover HDF5 writer: the output side of dsk2kover
// (invoked at bin/kover/core/kover/dataset/tools/kmer_pack.py:28-36).
//
// dsk2kover opens the HDF5 file Kover has ALREADY created (attrs, phenotype,
// genome_identifiers, phenotype_tags; closed at dataset/create.py:356) and appends three
// datasets whose schema is fixed by dataset/create.py:214-238 and read back by
// dataset/ds.py:72-94 and learning/common/rules.py:104-131:
//     kmer_sequences         S<k>   [U]            gzip G
//     kmer_matrix            uint64 [ceil(N/64)][U] chunks (1, chunk_cols), gzip G
//     kmer_by_matrix_column  min-uint[U]           gzip G   (identity here: column c <-> k-mer c)
//
// libhdf5 is dlopen()ed at run time (h5py is not a dependency; the C library may live in a
// conda prefix).  The chunks reach the file as finished zlib streams through H5Dwrite_chunk:
//   * a matrix that lives on the device is deflated THERE (grm_deflate.hip: one wave per chunk; kmer_matrix and
//     kmer_sequences), only the compressed bytes cross PCIe -- deflating 1 GB of presence words on the host's CPU share
//     (16 cores on the measured box) was 0.8 s of a 1.4 s end-to-end run with the device idle;
//   * a host-only matrix (rows gathered from other ranks, tests without a GPU), GRM_DEFLATE=host, or a libhdf5 without
//     H5Dwrite_chunk: chunks deflated on the host threads (libdeflate when it can be dlopen()ed, else zlib), the rows of a
//     device matrix arriving while earlier ones are deflated.
// A failed append never leaves a plausible file: whatever of the three datasets exists is unlinked before the file is
// closed (kmer_pack.py:28 ignores the tool's return code; unwritten chunks would read back as "absent everywhere").
#include <dlfcn.h>
#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/grm_kmer.h"

extern "C" void grm_internal_deflate_release(void *ring, int slot);

typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;

namespace {

struct H5 {
    void *h = nullptr;
    herr_t (*open)() = nullptr;
    hid_t (*Fopen)(const char *, unsigned, hid_t) = nullptr;
    herr_t (*Fclose)(hid_t) = nullptr;
    hid_t (*Screate_simple)(int, const hsize_t *, const hsize_t *) = nullptr;
    herr_t (*Sclose)(hid_t) = nullptr;
    hid_t (*Pcreate)(hid_t) = nullptr;
    herr_t (*Pset_chunk)(hid_t, int, const hsize_t *) = nullptr;
    herr_t (*Pset_deflate)(hid_t, unsigned) = nullptr;
    herr_t (*Pclose)(hid_t) = nullptr;
    hid_t (*Tcopy)(hid_t) = nullptr;
    herr_t (*Tset_size)(hid_t, size_t) = nullptr;
    herr_t (*Tset_strpad)(hid_t, int) = nullptr;
    herr_t (*Tclose)(hid_t) = nullptr;
    hid_t (*Dcreate2)(hid_t, const char *, hid_t, hid_t, hid_t, hid_t, hid_t) = nullptr;
    herr_t (*Dwrite)(hid_t, hid_t, hid_t, hid_t, hid_t, const void *) = nullptr;
    herr_t (*Dwrite_chunk)(hid_t, hid_t, uint32_t, const hsize_t *, size_t, const void *) = nullptr;
    herr_t (*Dclose)(hid_t) = nullptr;
    int (*Lexists)(hid_t, const char *, hid_t) = nullptr;
    herr_t (*Ldelete)(hid_t, const char *, hid_t) = nullptr;
    herr_t (*Eset_auto2)(hid_t, void *, void *) = nullptr;
    hid_t C_S1 = -1, U8 = -1, U16 = -1, U32 = -1, U64 = -1, DCPL = -1;
    std::string err;

    template <typename T> bool sym(T &fn, const char *name)
    {
        fn = reinterpret_cast<T>(dlsym(h, name));
        if (!fn) { err = std::string("libhdf5: missing symbol ") + name; return false; }
        return true;
    }
    bool global(hid_t &v, const char *name)
    {
        hid_t *p = reinterpret_cast<hid_t *>(dlsym(h, name));
        if (!p) { err = std::string("libhdf5: missing global ") + name; return false; }
        v = *p;
        return true;
    }
    bool load()
    {
        if (h) return true;
        std::vector<std::string> cands;
        if (const char *e = getenv("GRM_HDF5_LIB")) cands.push_back(e);
        for (const char *n : {"libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so.103",
                              "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"})
            cands.push_back(n);
        for (auto &c : cands) {
            h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) { err = "libhdf5 not found (set GRM_HDF5_LIB=/path/to/libhdf5.so); the TSV writer does not need it"; return false; }
        bool ok = sym(open, "H5open") && sym(Fopen, "H5Fopen") && sym(Fclose, "H5Fclose") &&
                  sym(Screate_simple, "H5Screate_simple") && sym(Sclose, "H5Sclose") && sym(Pcreate, "H5Pcreate") &&
                  sym(Pset_chunk, "H5Pset_chunk") && sym(Pset_deflate, "H5Pset_deflate") && sym(Pclose, "H5Pclose") &&
                  sym(Tcopy, "H5Tcopy") && sym(Tset_size, "H5Tset_size") && sym(Tset_strpad, "H5Tset_strpad") && sym(Tclose, "H5Tclose") &&
                  sym(Dcreate2, "H5Dcreate2") && sym(Dwrite, "H5Dwrite") && sym(Dclose, "H5Dclose") &&
                  sym(Lexists, "H5Lexists") && sym(Ldelete, "H5Ldelete") && sym(Eset_auto2, "H5Eset_auto2");
        if (!ok) { dlclose(h); h = nullptr; return false; }
        Dwrite_chunk = reinterpret_cast<decltype(Dwrite_chunk)>(dlsym(h, "H5Dwrite_chunk"));   // >= 1.10.3, optional
        if (open() < 0) { err = "H5open failed"; return false; }
        ok = global(C_S1, "H5T_C_S1_g") && global(U8, "H5T_NATIVE_UINT8_g") && global(U16, "H5T_NATIVE_UINT16_g") &&
             global(U32, "H5T_NATIVE_UINT32_g") && global(U64, "H5T_NATIVE_UINT64_g") &&
             global(DCPL, "H5P_CLS_DATASET_CREATE_ID_g");
        return ok;
    }
};

H5 g_h5;

inline void decode_kmer(const uint64_t *w, int words, int k, char *out)
{
    static const char L[4] = {'A', 'C', 'T', 'G'};
    for (int i = 0; i < k; i++) {
        const int bit = 2 * (k - 1 - i);
        const uint64_t word = w[words - 1 - bit / 64];
        out[i] = L[(word >> (bit & 63)) & 3];
    }
}

// chunked (optionally deflated) dataset creation property list
hid_t make_dcpl(H5 &H, int rank, const hsize_t *chunk, int gzip)
{
    hid_t p = H.Pcreate(H.DCPL);
    if (p < 0) return p;
    if (H.Pset_chunk(p, rank, chunk) < 0) { H.Pclose(p); return -1; }
    if (gzip > 0 && H.Pset_deflate(p, (unsigned)gzip) < 0) { H.Pclose(p); return -1; }
    return p;
}

// Threads for the host-side work: GRM_WRITER_THREADS, else what this process may actually use -- its CPU affinity, cut down to
// the cgroup's CPU quota when there is one (a 256-thread box may hand a job 16 CPUs: more threads than that only take turns).
unsigned host_threads()
{
    if (const char *e = getenv("GRM_WRITER_THREADS")) {
        const int v = atoi(e);
        if (v > 0) return (unsigned)v;
    }
    unsigned nt = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) nt = (unsigned)CPU_COUNT(&set);
    if (nt == 0) nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {            // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64];
        long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) {
            const long long quota = atoll(q);
            if (quota > 0) nt = std::min<unsigned>(nt, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    }
    return nt;
}

// body(i0, i1) over [0, n) split across the host threads
template <typename Body>
void parallel_for(size_t n, Body &&body)
{
    unsigned nt = host_threads();
    if (n < (size_t)1 << 16) nt = 1;
    const size_t per = (n + nt - 1) / nt;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) {
        const size_t a = std::min(n, t * per), b = std::min(n, a + per);
        if (a < b) th.emplace_back([&body, a, b]() { body(a, b); });
    }
    body(0, std::min(n, per));
    for (auto &t : th) t.join();
}

// libdeflate, when the box has it: the same zlib container from a faster encoder (any valid stream satisfies HDF5's filter)
struct LibDeflate {
    void *h = nullptr;
    bool tried = false;
    void *(*alloc)(int) = nullptr;
    size_t (*zlib_compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    size_t (*zlib_bound)(void *, size_t) = nullptr;
    void (*free_compressor)(void *) = nullptr;
    bool load()
    {
        if (tried) return h != nullptr;
        tried = true;
        if (const char *e = getenv("GRM_DEFLATE_LIB"))
            if (!strcmp(e, "zlib")) return false;
        for (const char *n : {"libdeflate.so.0", "libdeflate.so", "/opt/conda/lib/libdeflate.so"}) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return false;
        alloc = reinterpret_cast<decltype(alloc)>(dlsym(h, "libdeflate_alloc_compressor"));
        zlib_compress = reinterpret_cast<decltype(zlib_compress)>(dlsym(h, "libdeflate_zlib_compress"));
        zlib_bound = reinterpret_cast<decltype(zlib_bound)>(dlsym(h, "libdeflate_zlib_compress_bound"));
        free_compressor = reinterpret_cast<decltype(free_compressor)>(dlsym(h, "libdeflate_free_compressor"));
        if (!alloc || !zlib_compress || !zlib_bound || !free_compressor) { dlclose(h); h = nullptr; }
        return h != nullptr;
    }
};
LibDeflate g_ld;

// the finished streams of a dataset's chunks, wherever they were made
struct Streams {
    std::vector<std::vector<unsigned char>> z;      // host-made: one vector per chunk
    unsigned char *buf = nullptr;                   // device-made (grm_matrix_deflate_*): malloc'ed block + starts + lens
    uint64_t *starts = nullptr;
    uint32_t *lens = nullptr;
    uint64_t n = 0;
    Streams() = default;
    Streams(const Streams &) = delete;
    Streams &operator=(const Streams &) = delete;
    ~Streams() { grm_host_free(buf); grm_host_free(starts); grm_host_free(lens); }
    const unsigned char *at(size_t i) const { return buf ? buf + starts[i] : z[i].data(); }
    size_t len(size_t i) const { return buf ? lens[i] : z[i].size(); }
};

// deflate `n_chunks` equally sized raw chunks on the host threads; get_chunk(i, buf) must leave the
// (padded) chunk bytes in buf and return a pointer to them
template <typename GetChunk>
bool deflate_chunks(size_t n_chunks, size_t chunk_bytes, int gzip, GetChunk &&get_chunk, Streams &out)
{
    out.z.assign(n_chunks, {});
    out.n = n_chunks;
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    unsigned nt = host_threads();
    if (nt > n_chunks) nt = (unsigned)n_chunks;
    const bool use_ld = g_ld.load();
    auto work = [&]() {
        std::vector<unsigned char> raw(chunk_bytes);
        void *comp = use_ld ? g_ld.alloc(gzip) : nullptr;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_chunks) break;
            const unsigned char *src = get_chunk(i, raw.data());
            auto &z = out.z[i];
            if (comp) {
                z.resize(g_ld.zlib_bound(comp, chunk_bytes));
                const size_t zl = g_ld.zlib_compress(comp, src, chunk_bytes, z.data(), z.size());
                if (!zl) bad = 1;
                z.resize(zl);
            } else {
                uLongf zl = compressBound(chunk_bytes);
                z.resize(zl);
                if (compress2(z.data(), &zl, src, chunk_bytes, gzip) != Z_OK) bad = 1;
                z.resize(zl);
            }
        }
        if (comp) g_ld.free_compressor(comp);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return !bad;
}

// GRM_FAULT_H5_CHUNK=<i> (tests): the i-th chunk handed to HDF5 in this call fails, as a full disk would make it
struct Fault {
    long at = -1, seen = 0;
    Fault() { if (const char *e = getenv("GRM_FAULT_H5_CHUNK")) at = atol(e); }
    bool hit() { return at >= 0 && seen++ == at; }
};

// slabs of finished chunks on their way from the device (grm_internal_deflate_stream, a thread of its own) to the HDF5 calls of the
// writing thread: the k-mer strings' slabs first, then the matrix rows', each in chunk order
struct SlabQueue {
    struct Slab {
        int what = 0;
        uint64_t first = 0, n = 0;
        unsigned char *block = nullptr;
        void *ring = nullptr;        // block is buffer `slot` of the producer's pinned ring (given back when done), or a malloc block (slot < 0)
        int slot = -1;
        std::vector<uint64_t> starts;
        std::vector<uint32_t> lens;
        void done()
        {
            if (block) { if (slot >= 0) grm_internal_deflate_release(ring, slot); else free(block); }
            block = nullptr;
        }
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Slab> q;
    bool finished = false, abandoned = false;
    int producing = 0;           // dataset the producer is on (sink calls carry no tag of their own)
    Slab cur;
    ~SlabQueue() { drop_all(); }
    void drop_all()
    {
        cur.done();
        for (auto &sl : q) sl.done();
        q.clear();
    }
    static int sink(void *user, uint64_t first, uint64_t n, unsigned char *block, const uint64_t *starts, const uint32_t *lens, void *ring, int slot)
    {
        SlabQueue *Q = static_cast<SlabQueue *>(user);
        Slab sl;
        sl.what = Q->producing; sl.first = first; sl.n = n; sl.block = block; sl.ring = ring; sl.slot = slot;
        sl.starts.assign(starts, starts + n);
        sl.lens.assign(lens, lens + n);
        std::lock_guard<std::mutex> g(Q->mu);
        if (Q->abandoned) { sl.done(); return GRM_ERR_STATE; }       // the writer gave up: stop producing
        Q->q.push_back(std::move(sl));
        Q->cv.notify_all();
        return GRM_OK;
    }
    void finish()
    {
        std::lock_guard<std::mutex> g(mu);
        finished = true;
        cv.notify_all();
    }
    // the writer is through (done, or failed): nothing more is wanted, and the producer gets its buffers back at once
    void abandon()
    {
        std::lock_guard<std::mutex> g(mu);
        abandoned = true;
        drop_all();
    }
    // chunk i of dataset `what` (asked for in order); false: the producer ended without delivering it
    bool get(int what, uint64_t i, const unsigned char *&p, size_t &len)
    {
        while (!(cur.block && cur.what == what && i >= cur.first && i < cur.first + cur.n)) {
            cur.done();
            cur = Slab();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return !q.empty() || finished; });
            if (q.empty()) return false;
            cur = std::move(q.front());
            q.pop_front();
        }
        p = cur.block + cur.starts[i - cur.first];
        len = cur.lens[i - cur.first];
        return true;
    }
};

// 1-D dataset with `ce_want` elements per chunk.  chunk_at != nullptr: its finished chunks come from there; else raw `data`
// (deflated here when gzip > 0 and H5Dwrite_chunk exists, plain H5Dwrite otherwise)
typedef std::function<bool(size_t, const unsigned char *&, size_t &)> ChunkAt;
int write_1d(H5 &H, hid_t file, const char *name, hid_t type, size_t elem_bytes, const void *data, hsize_t n, int gzip, hsize_t ce_want,
             const ChunkAt *chunk_at, Fault &fault, std::string &err)
{
    if (H.Lexists(file, name, 0) > 0) H.Ldelete(file, name, 0);
    hsize_t dims[1] = {n};
    hid_t space = H.Screate_simple(1, dims, nullptr);
    hid_t dcpl = 0;
    const hsize_t ce = n < ce_want ? (n ? n : 1) : ce_want;      // elements per chunk
    if (n > 0) {
        hsize_t chunk[1] = {ce};
        dcpl = make_dcpl(H, 1, chunk, gzip);
        if (dcpl < 0) { H.Sclose(space); err = std::string("dcpl for ") + name; return -1; }
    }
    hid_t ds = H.Dcreate2(file, name, type, space, 0, dcpl, 0);
    int rc = 0;
    if (ds < 0) { err = std::string("H5Dcreate2 ") + name; rc = -1; }
    else {
        if (n > 0 && gzip > 0 && H.Dwrite_chunk) {
            const size_t n_chunks = (size_t)((n + ce - 1) / ce), cb = (size_t)ce * elem_bytes;
            Streams own;
            if (!chunk_at) {
                const unsigned char *base = static_cast<const unsigned char *>(data);
                const bool ok = deflate_chunks(n_chunks, cb, gzip, [&](size_t i, unsigned char *buf) -> const unsigned char * {
                    const size_t e0 = i * (size_t)ce, ne = (size_t)std::min<hsize_t>(ce, n - e0);
                    if (ne == (size_t)ce) return base + e0 * elem_bytes;
                    memcpy(buf, base + e0 * elem_bytes, ne * elem_bytes);            // edge chunk: pad with the fill value 0
                    memset(buf + ne * elem_bytes, 0, cb - ne * elem_bytes);
                    return buf;
                }, own);
                if (!ok) { err = "deflate failed"; rc = -1; }
            }
            for (size_t i = 0; i < n_chunks && !rc; i++) {
                hsize_t off[1] = {i * ce};
                const unsigned char *p = nullptr;
                size_t len = 0;
                if (chunk_at) {
                    if (!(*chunk_at)(i, p, len)) { err = std::string(name) + ": the chunk producer stopped early"; rc = -1; break; }
                } else { p = own.at(i); len = own.len(i); }
                if (fault.hit() || H.Dwrite_chunk(ds, 0, 0, off, len, p) < 0) { err = std::string("H5Dwrite_chunk ") + name; rc = -1; }
            }
        } else if (n > 0 && (fault.hit() || H.Dwrite(ds, type, 0, 0, 0, data) < 0)) { err = std::string("H5Dwrite ") + name; rc = -1; }
        H.Dclose(ds);
    }
    if (dcpl > 0) H.Pclose(dcpl);
    H.Sclose(space);
    return rc;
}

}  // namespace

// ctx error plumbing lives in grm_api.cpp
extern "C" int grm_internal_fail(grm_matrix *m, int code, const char *msg);
extern "C" uint64_t *grm_internal_matrix_download_begin(grm_matrix *m, int *already);
extern "C" int grm_internal_matrix_download_rows(grm_matrix *m, size_t r0, size_t r1);
extern "C" void grm_internal_matrix_download_end(grm_matrix *m);
extern "C" int grm_internal_matrix_on_device(const grm_matrix *m);
typedef int (*grm_deflate_sink)(void *user, uint64_t first_chunk, uint64_t n, unsigned char *block, const uint64_t *starts, const uint32_t *lens, void *ring,
                                int slot);
extern "C" void grm_internal_deflate_release(void *ring, int slot);
extern "C" int grm_internal_deflate_stream(grm_matrix *m, int what, uint64_t chunk_elems, uint64_t max_slab, grm_deflate_sink sink, void *user, uint64_t *n_chunks);

// The append itself.  parts == nullptr: kmer_matrix holds m's own rows.  Otherwise kmer_matrix has n_rows_total word-rows whose
// chunks come finished from n_parts producers (ranks of a multi-GPU run, each having deflated the rows it filled): part p covers
// rows [row0[p], row0[p] + rows[p]) with its chunks row-major at streams[p] + starts[p][i], lens[p][i] bytes; m gives the
// dictionary (kmer_sequences, kmer_by_matrix_column) and must hold the same columns on every producer.
static int write_kover(grm_matrix *m, const char *existing_h5_path, int gzip_level, int chunk_cols, uint64_t n_rows_total, int n_parts,
                       const unsigned char *const *p_streams, const uint64_t *const *p_starts, const uint32_t *const *p_lens, const uint64_t *p_row0,
                       const uint64_t *p_rows)
{
    if (!m || !existing_h5_path) return GRM_ERR_ARG;
    if (gzip_level < 0 || gzip_level > 9) return grm_internal_fail(m, GRM_ERR_ARG, "gzip level must be 0..9");
    if (chunk_cols <= 0) chunk_cols = 100000;                        // BLOCK_SIZE, dataset/create.py:41
    H5 &H = g_h5;
    if (!H.load()) return grm_internal_fail(m, GRM_ERR_HDF5, H.err.c_str());
    const bool parts = n_parts > 0;
    if (parts && (gzip_level == 0 || !H.Dwrite_chunk)) return grm_internal_fail(m, GRM_ERR_ARG, "prepared chunks need gzip > 0 and H5Dwrite_chunk (HDF5 >= 1.10.3)");
    // GRM_TRACE=1: phase timings on stderr
    const bool trace = getenv("GRM_TRACE") && atoi(getenv("GRM_TRACE")) > 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_last = now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const double t = now();
        fprintf(stderr, "[grm_write_kover_h5] %-36s %8.1f ms\n", what, (t - t_last) * 1e3);
        t_last = t;
    };
    const size_t U = grm_matrix_n_kmers(m), R = parts ? (size_t)n_rows_total : grm_matrix_n_rows(m);
    const int k = grm_matrix_k(m);
    const hsize_t cw = U ? (U < (size_t)chunk_cols ? U : (hsize_t)chunk_cols) : 1;       // chunks (1, min(U, chunk_cols)) as from_tsv does (create.py:160,230)
    const size_t chunks_per_row = (U + cw - 1) / cw;
    if (parts) {
        uint64_t covered = 0;
        for (int p = 0; p < n_parts; p++) {
            if (!p_streams[p] && p_rows[p]) return grm_internal_fail(m, GRM_ERR_ARG, "prepared chunks: a part without streams");
            if (p_row0[p] != covered) return grm_internal_fail(m, GRM_ERR_ARG, "prepared chunks: the parts must cover the rows in order, without gaps");
            covered += p_rows[p];
        }
        if (covered != n_rows_total) return grm_internal_fail(m, GRM_ERR_ARG, "prepared chunks: the parts do not add up to the matrix's rows");
    }
    const char *mode = getenv("GRM_DEFLATE");
    const bool on_device = grm_internal_matrix_on_device(m) && gzip_level > 0 && H.Dwrite_chunk && !(mode && !strcmp(mode, "host"));

    // ---- the producers of the two large datasets start first: they run beside the HDF5 calls of this thread ----
    // device path: a thread drives the encoder slab by slab (k-mer strings, then the matrix rows) and this thread appends slab k
    // while slab k + 1 is encoded and copied
    constexpr hsize_t SEQ_CHUNK = 1 << 14;         // strings per chunk of kmer_sequences on the device path: 600 waves for 10 M k-mers (65536: 150)
    SlabQueue slabs;
    Streams mat_streams;
    std::atomic<int> dev_rc(GRM_OK);
    std::thread dev_worker;
    // host path, device-resident matrix: the dictionary comes down first (small), the matrix then follows row by row on a thread
    // of its own and the chunk deflaters below start on a row as soon as it has arrived
    const uint64_t *kmers = nullptr, *data = nullptr;
    std::atomic<size_t> rows_ready(0);
    std::atomic<int> copy_failed(0);
    std::thread copier;
    struct Joiner { std::thread &t; SlabQueue *q; ~Joiner() { if (q) q->abandon(); if (t.joinable()) t.join(); } } join_dev{dev_worker, &slabs}, join_copy{copier, nullptr};
    if (on_device) {
        dev_worker = std::thread([&]() {
            uint64_t n = 0;
            slabs.producing = 1;
            int rc = grm_internal_deflate_stream(m, 1, SEQ_CHUNK, 1024, SlabQueue::sink, &slabs, &n);
            if (rc == GRM_OK && !parts) {
                {
                    std::lock_guard<std::mutex> g(slabs.mu);
                    slabs.producing = 0;
                }
                rc = grm_internal_deflate_stream(m, 0, (uint64_t)chunk_cols, 512, SlabQueue::sink, &slabs, &n);
            }
            dev_rc = rc;
            slabs.finish();
        });
    } else {
        kmers = grm_matrix_kmers(m);
        if (!kmers) return GRM_ERR_HIP;
        if (!parts) {
            int already = 0;
            data = grm_internal_matrix_download_begin(m, &already);
            if (!data) return GRM_ERR_HIP;
            rows_ready = already ? R : 0;
            if (!already)
                copier = std::thread([&]() {
                    for (size_t r = 0; r < R; r++) {
                        if (grm_internal_matrix_download_rows(m, r, r + 1) != GRM_OK) { copy_failed = 1; rows_ready = R; return; }
                        rows_ready = r + 1;
                    }
                    grm_internal_matrix_download_end(m);
                });
        }
        lap("k-mers device -> host");
    }

    H.Eset_auto2(0, nullptr, nullptr);
    hid_t file = H.Fopen(existing_h5_path, 1u /* H5F_ACC_RDWR */, 0);
    if (file < 0) return grm_internal_fail(m, GRM_ERR_HDF5, (std::string("cannot open existing Kover HDF5 ") + existing_h5_path).c_str());
    std::string err;
    int rc = 0;
    Fault fault;

    // kmer_by_matrix_column: identity map in the minimum unsigned width (utils.py:117-130); small, deflated on the host.  On the device
    // path a thread of its own prepares its chunks while this one appends the two large datasets, and they are appended last.
    const size_t col_bytes = U <= 0xffu ? 1 : U <= 0xffffu ? 2 : U <= 0xffffffffull ? 4 : 8;
    const hid_t col_type = col_bytes == 1 ? H.U8 : col_bytes == 2 ? H.U16 : col_bytes == 4 ? H.U32 : H.U64;
    std::vector<unsigned char> col_raw;
    auto fill_col = [&]() {
        col_raw.resize((U + 1) * col_bytes);
        parallel_for(U, [&](size_t c0, size_t c1) {
            for (size_t i = c0; i < c1; i++) {
                if (col_bytes == 1) col_raw[i] = (uint8_t)i;
                else if (col_bytes == 2) reinterpret_cast<uint16_t *>(col_raw.data())[i] = (uint16_t)i;
                else if (col_bytes == 4) reinterpret_cast<uint32_t *>(col_raw.data())[i] = (uint32_t)i;
                else reinterpret_cast<uint64_t *>(col_raw.data())[i] = (uint64_t)i;
            }
        });
    };
    constexpr hsize_t COL_CHUNK = 1 << 16;
    Streams col_streams;
    std::atomic<int> col_ok(1);
    std::thread col_worker;
    struct ColJoin { std::thread &t; ~ColJoin() { if (t.joinable()) t.join(); } } join_col{col_worker};
    const bool col_aside = on_device && U > 0 && gzip_level > 0 && H.Dwrite_chunk;
    if (col_aside) {
        col_worker = std::thread([&]() {
            fill_col();
            const hsize_t ce = U < COL_CHUNK ? U : COL_CHUNK;
            const size_t n_chunks = (size_t)((U + ce - 1) / ce), cb = (size_t)ce * col_bytes;
            const bool ok = deflate_chunks(n_chunks, cb, gzip_level, [&](size_t i, unsigned char *buf) -> const unsigned char * {
                const size_t e0 = i * (size_t)ce, ne = std::min<size_t>((size_t)ce, U - e0);
                if (ne == (size_t)ce) return col_raw.data() + e0 * col_bytes;
                memcpy(buf, col_raw.data() + e0 * col_bytes, ne * col_bytes);
                memset(buf + ne * col_bytes, 0, cb - ne * col_bytes);
                return buf;
            }, col_streams);
            if (!ok) col_ok = 0;
        });
    } else {
        fill_col();
        rc = write_1d(H, file, "kmer_by_matrix_column", col_type, col_bytes, col_raw.data(), U, gzip_level, COL_CHUNK, nullptr, fault, err);
        lap("kmer_by_matrix_column");
    }

    // kmer_sequences: fixed-length S<k> (what numpy 'S31' becomes in h5py), null padded
    if (!rc) {
        hid_t st = H.Tcopy(H.C_S1);
        H.Tset_size(st, (size_t)k);
        H.Tset_strpad(st, 1);   // H5T_STR_NULLPAD
        if (on_device) {
            const ChunkAt from_device = [&](size_t i, const unsigned char *&p, size_t &len) { return slabs.get(1, i, p, len); };
            rc = write_1d(H, file, "kmer_sequences", st, (size_t)k, nullptr, U, gzip_level, SEQ_CHUNK, &from_device, fault, err);
        } else {
            std::vector<char> seq(U * (size_t)k + 1);
            const int words = grm_matrix_words(m);
            parallel_for(U, [&](size_t c0, size_t c1) {
                for (size_t c = c0; c < c1; c++) decode_kmer(kmers + c * (size_t)words, words, k, seq.data() + c * (size_t)k);
            });
            lap("decode k-mer strings");
            rc = write_1d(H, file, "kmer_sequences", st, (size_t)k, seq.data(), U, gzip_level, 1 << 16, nullptr, fault, err);
        }
        H.Tclose(st);
        lap("kmer_sequences");
    }

    // kmer_matrix
    if (!rc) {
        if (H.Lexists(file, "kmer_matrix", 0) > 0) H.Ldelete(file, "kmer_matrix", 0);
        hsize_t dims[2] = {R, U};
        hid_t space = H.Screate_simple(2, dims, nullptr);
        hid_t dcpl = 0;
        if (U && R) {
            hsize_t chunk[2] = {1, cw};
            dcpl = make_dcpl(H, 2, chunk, gzip_level);
        }
        hid_t ds = (dcpl < 0) ? -1 : H.Dcreate2(file, "kmer_matrix", H.U64, space, 0, dcpl, 0);
        if (ds < 0) { err = "H5Dcreate2 kmer_matrix"; rc = -1; }
        else if (U && R) {
            const size_t n_chunks = chunks_per_row * R;
            auto put = [&](size_t i, const unsigned char *p, size_t len) {
                hsize_t off[2] = {i / chunks_per_row, (i % chunks_per_row) * cw};
                if (fault.hit() || H.Dwrite_chunk(ds, 0, 0, off, len, p) < 0) { err = "H5Dwrite_chunk kmer_matrix"; rc = -1; }
            };
            if (parts) {
                for (int p = 0; p < n_parts && !rc; p++)
                    for (size_t j = 0; j < (size_t)p_rows[p] * chunks_per_row && !rc; j++)
                        put((size_t)p_row0[p] * chunks_per_row + j, p_streams[p] + p_starts[p][j], p_lens[p][j]);
            } else if (on_device) {
                for (size_t i = 0; i < n_chunks && !rc; i++) {
                    const unsigned char *p = nullptr;
                    size_t len = 0;
                    if (!slabs.get(0, i, p, len)) { err = "kmer_matrix: the device stopped before chunk " + std::to_string(i); rc = -1; break; }
                    put(i, p, len);
                }
            } else if (gzip_level > 0 && H.Dwrite_chunk) {
                // deflate every chunk on the host threads, then hand the streams to HDF5 in order
                const bool ok = deflate_chunks(n_chunks, (size_t)cw * 8, gzip_level, [&](size_t i, unsigned char *buf) -> const unsigned char * {
                    const size_t r = i / chunks_per_row, c0 = (i % chunks_per_row) * cw;
                    while (rows_ready.load() <= r) std::this_thread::sleep_for(std::chrono::microseconds(200));     // this row is still on its way
                    const size_t nc = (c0 + cw <= U) ? (size_t)cw : U - c0;
                    const uint64_t *src = data + r * U + c0;
                    if (nc == (size_t)cw) return reinterpret_cast<const unsigned char *>(src);
                    memcpy(buf, src, nc * 8);                         // edge chunk: HDF5 stores full chunks, pad with the fill value 0
                    memset(buf + nc * 8, 0, ((size_t)cw - nc) * 8);
                    return buf;
                }, mat_streams);
                if (!ok) { err = "deflate failed"; rc = -1; }
                if (copy_failed) { err = "device -> host copy of the matrix failed"; rc = -1; }
                lap("kmer_matrix download + deflate");
                for (size_t i = 0; i < n_chunks && !rc; i++) put(i, mat_streams.at(i), mat_streams.len(i));
            } else {
                if (copier.joinable()) copier.join();
                if (copy_failed) { err = "device -> host copy of the matrix failed"; rc = -1; }      // never write rows that did not arrive
                else if (fault.hit() || H.Dwrite(ds, H.U64, 0, 0, 0, data) < 0) { err = "H5Dwrite kmer_matrix"; rc = -1; }
            }
        }
        if (ds >= 0) H.Dclose(ds);
        if (dcpl > 0) H.Pclose(dcpl);
        H.Sclose(space);
    }
    lap("kmer_matrix write");
    if (on_device) {
        slabs.abandon();             // (nothing more is wanted; a producer that is still at work stops at its next hand-over)
        dev_worker.join();
        if (dev_rc != GRM_OK && (!rc || err.find("stopped") != std::string::npos)) { err = std::string("deflate on the device: ") + grm_matrix_last_error(m); rc = -1; }
    }
    if (col_aside) {
        col_worker.join();
        if (!rc) {
            if (!col_ok) { err = "deflate failed (kmer_by_matrix_column)"; rc = -1; }
            else {
                const ChunkAt from_worker = [&](size_t i, const unsigned char *&p, size_t &len) { p = col_streams.at(i); len = col_streams.len(i); return i < col_streams.n; };
                rc = write_1d(H, file, "kmer_by_matrix_column", col_type, col_bytes, nullptr, U, gzip_level, COL_CHUNK, &from_worker, fault, err);
            }
        }
        lap("kmer_by_matrix_column");
    }
    if (rc) {
        // whatever exists of the three datasets is incomplete: a reader must not find it (unwritten chunks read back as zeros)
        for (const char *name : {"kmer_sequences", "kmer_by_matrix_column", "kmer_matrix"})
            if (H.Lexists(file, name, 0) > 0) H.Ldelete(file, name, 0);
    }
    if (H.Fclose(file) < 0 && !rc) { err = "H5Fclose"; rc = -1; }
    lap("H5Fclose");
    if (rc) return grm_internal_fail(m, GRM_ERR_HDF5, (err + " (" + existing_h5_path + ": the partly written datasets were removed)").c_str());
    return GRM_OK;
}

extern "C" int grm_write_kover_h5(grm_matrix *m, const char *existing_h5_path, int gzip_level, int chunk_cols)
{
    return write_kover(m, existing_h5_path, gzip_level, chunk_cols, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int grm_write_kover_h5_parts(grm_matrix *dict, const char *existing_h5_path, int gzip_level, int chunk_cols, uint64_t n_rows_total, int n_parts,
                                        const unsigned char *const *streams, const uint64_t *const *starts, const uint32_t *const *lens,
                                        const uint64_t *row0, const uint64_t *rows)
{
    if (n_parts <= 0 || !streams || !starts || !lens || !row0 || !rows) return GRM_ERR_ARG;
    return write_kover(dict, existing_h5_path, gzip_level, chunk_cols, n_rows_total, n_parts, streams, starts, lens, row0, rows);
}
