"""Host-side mirror of the reference's k-mer tool boundary, on top of the C ABI.

The reference has no in-process API for this path: Kover calls
`multidsk` (dataset/tools/kmer_count.py:23-53) then `dsk2kover` (tools/kmer_pack.py:23-36).
`count_genome` / `build_matrix` below are those two steps; `Batch` is the fused
device-resident form.  numpy arrays out, plain pointers in: PyTorch is not needed here.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib

STATUS = {0: "GRM_OK", -1: "GRM_ERR_ARG", -2: "GRM_ERR_NO_DEVICE", -3: "GRM_ERR_HIP", -4: "GRM_ERR_IO",
          -5: "GRM_ERR_OOM", -6: "GRM_ERR_UNSUPPORTED", -7: "GRM_ERR_STATE", -8: "GRM_ERR_OVERFLOW",
          -9: "GRM_ERR_HDF5"}


class GrmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class Context:
    """one HIP device + stream (grm_ctx).  Fails loudly when no device is present."""

    def __init__(self, device=0):
        self.L = _lib.load()
        self.h = self.L.grm_create(device, 1)
        if not self.h:
            raise GrmError(-2, "grm_create(%d) failed: no usable HIP device (there is no CPU fallback)" % device)
        self.device = device
        self._children = weakref.WeakSet()      # live Batch / Matrix / KmerSet / DictAccum objects of this context

    def _adopt(self, child):
        self._children.add(child)
        return child

    def close(self):
        """Frees every handle made from this context that is still alive, then the context.  (The library also counts
        references -- a handle freed after grm_destroy is safe -- but freeing in order returns the HBM right away.)"""
        if self.h:
            for child in list(self._children):
                try:
                    child.free()
                except GrmError:        # (a free never fails in the library; anything else is a bug and must surface)
                    pass
            self.L.grm_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise GrmError(rc, (self.L.grm_last_error(self.h) or b"").decode(errors="replace"))

    def set_option(self, name, value):
        self._chk(self.L.grm_set_option(self.h, name.encode(), int(value)))

    # ---- timings ----
    def timing(self, on=True):
        self._chk(self.L.grm_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        self._chk(self.L.grm_timing_reset(self.h))

    def timings(self):
        """-> list of (kernel name, ms, units)"""
        out = []
        buf = C.create_string_buffer(64)
        for i in range(self.L.grm_timing_count(self.h)):
            ms, units = C.c_double(), C.c_uint64()
            self._chk(self.L.grm_timing_get(self.h, i, buf, 64, C.byref(ms), C.byref(units)))
            out.append((buf.value.decode(), ms.value, int(units.value)))
        return out

    # ---- multidsk: one genome -> solid k-mer set ----
    def count_genome(self, buffers, k, abundance_min=1):
        """buffers: list[bytes] = the file images of ONE genome -> KmerSet"""
        n = len(buffers)
        arr = (C.c_char_p * max(1, n))(*buffers)
        lens = (C.c_size_t * max(1, n))(*[len(b) for b in buffers])
        h = C.c_void_p()
        self._chk(self.L.grm_count_genome_buffers(self.h, arr, lens, n, k, abundance_min, C.byref(h)))
        return KmerSet(self, h)

    def count_genome_files(self, paths, k, abundance_min=1):
        n = len(paths)
        arr = (C.c_char_p * max(1, n))(*[p.encode() for p in paths])
        h = C.c_void_p()
        self._chk(self.L.grm_count_genome(self.h, arr, n, k, abundance_min, C.byref(h)))
        return KmerSet(self, h)

    def merge_counted_sets(self, sets, abundance_min=1):
        """pooled count: sums the counts of equal k-mers over the sets, keeps totals >= abundance_min"""
        arr = (C.c_void_p * max(1, len(sets)))(*[s.h for s in sets])
        h = C.c_void_p()
        self._chk(self.L.grm_merge_counted_sets(self.h, arr, len(sets), abundance_min, C.byref(h)))
        return KmerSet(self, h)

    def kmer_set_from_arrays(self, kmers, counts, k):
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
        counts = np.ascontiguousarray(counts, dtype=np.uint32).reshape(-1)
        h = C.c_void_p()
        self._chk(self.L.grm_kmer_set_from_host(self.h, kmers.ctypes.data, counts.ctypes.data, len(counts), k, C.byref(h)))
        return KmerSet(self, h)

    # ---- inputs beyond one device batch: dictionary accumulator + row stacking ----
    def dict_accum(self):
        h = C.c_void_p()
        self._chk(self.L.grm_dict_accum_create(self.h, C.byref(h)))
        return DictAccum(self, h)

    def stack_rows(self, parts):
        """parts: Matrix objects of consecutive genome blocks against the same dictionary"""
        arr = (C.c_void_p * max(1, len(parts)))(*[m.h for m in parts])
        h = C.c_void_p()
        self._chk(self.L.grm_matrix_stack_rows(arr, len(parts), C.byref(h)))
        return Matrix(self, h)

    # ---- dsk2kover: N sets -> dictionary + presence matrix ----
    def build_matrix(self, sets, filter_singleton=False):
        n = len(sets)
        arr = (C.c_void_p * max(1, n))(*[s.h for s in sets])
        h = C.c_void_p()
        self._chk(self.L.grm_build_matrix(self.h, arr, n, 1 if filter_singleton else 0, C.byref(h)))
        return Matrix(self, h)

    def batch(self, n_genomes):
        return Batch(self, n_genomes)


class _Handle:
    """A library handle.  Reading `.h` of a freed one raises instead of handing NULL to the C ABI (which would answer 0 / an empty
    array): Context.close() frees the handles that are still alive, so a Matrix carried out of its `with Context` block is dead."""
    _h = None

    @property
    def h(self):
        if self._h is None:
            raise GrmError(-7, "%s: the handle was freed (explicitly, or because its Context was closed)" % type(self).__name__)
        return self._h

    @h.setter
    def h(self, value):
        self._h = value


class KmerSet(_Handle):
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        ctx._adopt(self)

    def __len__(self):
        return self.ctx.L.grm_kmer_set_size(self.h)

    @property
    def k(self):
        return self.ctx.L.grm_kmer_set_k(self.h)

    @property
    def occurrences(self):
        return int(self.ctx.L.grm_kmer_set_occurrences(self.h))

    def kmers(self):
        n, w = len(self), self.ctx.L.grm_kmer_set_words(self.h)
        if n == 0:
            return np.zeros((0, w), np.uint64)
        p = self.ctx.L.grm_kmer_set_kmers(self.h)          # downloads a device-resident set on first use
        if not p:
            self.ctx._chk(-3)
        return np.ctypeslib.as_array(p, shape=(n * w,)).copy().reshape(n, w)

    def counts(self):
        n = len(self)
        if n == 0:
            return np.zeros(0, np.uint32)
        p = self.ctx.L.grm_kmer_set_counts(self.h)
        if not p:
            self.ctx._chk(-3)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def free(self):
        if self._h:
            self.ctx.L.grm_kmer_set_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Matrix(_Handle):
    """dictionary (ascending, A<C<T<G) + uint64 presence matrix [ceil(N/64)][U], MSB-first"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        ctx._adopt(self)

    @property
    def n_kmers(self):
        return self.ctx.L.grm_matrix_n_kmers(self.h)

    @property
    def n_rows(self):
        return self.ctx.L.grm_matrix_n_rows(self.h)

    @property
    def n_genomes(self):
        return self.ctx.L.grm_matrix_n_genomes(self.h)

    @property
    def k(self):
        return self.ctx.L.grm_matrix_k(self.h)

    def kmers(self):
        U, w = self.n_kmers, self.ctx.L.grm_matrix_words(self.h)
        if U == 0:
            return np.zeros((0, w), np.uint64)
        p = self.ctx.L.grm_matrix_kmers(self.h)
        if not p:
            self.ctx._chk(-3)
        return np.ctypeslib.as_array(p, shape=(U * w,)).copy().reshape(U, w)

    def data(self):
        U, r = self.n_kmers, self.n_rows
        if U == 0 or r == 0:
            return np.zeros((r, U), np.uint64)
        p = self.ctx.L.grm_matrix_data(self.h)
        if not p:
            self.ctx._chk(-3)
        return np.ctypeslib.as_array(p, shape=(r * U,)).copy().reshape(r, U)

    def column_counts(self):
        out = np.zeros(self.n_kmers, dtype=np.uint32)
        self.ctx._chk(self.ctx.L.grm_matrix_column_counts(self.h, out.ctypes.data))
        return out

    def sum_rows(self, genomes):
        """rules.py:201-267: per k-mer, how many of the given genome indices carry it (on the GPU)"""
        mask = np.zeros(max(1, self.n_rows), dtype=np.uint64)
        for g in genomes:
            mask[g // 64] |= np.uint64(1) << np.uint64(63 - (g % 64))
        out = np.zeros(self.n_kmers, dtype=np.uint32)
        self.ctx._chk(self.ctx.L.grm_matrix_sum_rows(self.h, mask.ctypes.data, out.ctypes.data))
        return out

    def _row_mask(self, genomes):
        mask = np.zeros(max(1, self.n_rows), dtype=np.uint64)
        for g in genomes:
            mask[int(g) // 64] |= np.uint64(1) << np.uint64(63 - (int(g) % 64))       # rules.py:210-222
        return mask

    def risk_tables(self, labels, train_idx):
        """dataset/split.py:171-188 on the device-resident matrix: -> (unique_risks float64, unique_risk_by_kmer,
        unique_risk_by_anti_kmer) exactly as the reference stores them.  The sweep of the matrix and the per-k-mer
        indexing run on the GPU; the rounding to 5 decimals and the unique() act on the <= n_train + 1 distinct
        error counts, with numpy, as the reference's own expressions."""
        from .kover_dataset import minimum_uint
        labels = np.asarray(labels)
        train_idx = np.asarray(train_idx, dtype=np.int64)
        pos, neg = train_idx[labels[train_idx] == 1], train_idx[labels[train_idx] == 0]
        n_train, U = len(train_idx), self.n_kmers
        pm, nm = self._row_mask(pos), self._row_mask(neg)
        hist = np.zeros(n_train + 1, dtype=np.uint64)
        self.ctx._chk(self.ctx.L.grm_matrix_risk_errors(self.h, pm.ctypes.data, nm.ctypes.data, len(pos), n_train, hist.ctypes.data))
        present = np.nonzero(hist)[0]
        risks = present.astype(np.float64)
        risks /= n_train
        np.round(risks, 5, out=risks)
        anti = 1.0 - risks
        np.round(anti, 5, out=anti)
        unique = np.unique(np.hstack((risks, anti)))
        lut_p = np.zeros(n_train + 1, dtype=np.uint32)
        lut_a = np.zeros(n_train + 1, dtype=np.uint32)
        lut_p[present] = np.searchsorted(unique, risks)
        lut_a[present] = np.searchsorted(unique, anti)
        by_kmer = np.zeros(U, dtype=np.uint32)
        by_anti = np.zeros(U, dtype=np.uint32)
        if U:
            self.ctx._chk(self.ctx.L.grm_matrix_risk_index(self.h, lut_p.ctypes.data, lut_a.ctypes.data, n_train + 1,
                                                           by_kmer.ctypes.data, by_anti.ctypes.data))
        dt = minimum_uint(len(unique))
        return unique, by_kmer.astype(dt), by_anti.astype(dt)

    def dev_ptrs(self):
        return self.ctx.L.grm_matrix_dev_kmers(self.h), self.ctx.L.grm_matrix_dev_data(self.h)

    def write_tsv(self, genome_ids, path):
        arr = (C.c_char_p * max(1, len(genome_ids)))(*[g.encode() for g in genome_ids])
        self.ctx._chk(self.ctx.L.grm_write_tsv(self.h, arr, path.encode()))

    def write_kover_h5(self, path, gzip_level=4, chunk_cols=100000):
        self.ctx._chk(self.ctx.L.grm_write_kover_h5(self.h, path.encode(), gzip_level, chunk_cols))

    def _streams(self, fn, chunk):
        buf, starts, lens, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self.ctx._chk(fn(self.h, int(chunk), C.byref(buf), C.byref(starts), C.byref(lens), C.byref(n)))
        try:
            n = int(n.value)
            if n == 0:
                return ChunkStreams(np.zeros(0, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
            st = np.ctypeslib.as_array(C.cast(starts, C.POINTER(C.c_uint64)), shape=(n,)).copy()
            ln = np.ctypeslib.as_array(C.cast(lens, C.POINTER(C.c_uint32)), shape=(n,)).copy()
            total = int(st[-1]) + int(ln[-1])
            return ChunkStreams(np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(total,)).copy(), st, ln)
        finally:
            for ptr in (buf, starts, lens):
                self.ctx.L.grm_host_free(ptr)

    def deflate_rows(self, chunk_cols=100000):
        """the kmer_matrix HDF5 chunks of this matrix's word-rows as zlib streams made on the device -> ChunkStreams"""
        return self._streams(self.ctx.L.grm_matrix_deflate_rows, chunk_cols)

    def deflate_kmer_strings(self, chunk_elems=65536):
        return self._streams(self.ctx.L.grm_matrix_deflate_kmer_strings, chunk_elems)

    def write_kover_h5_parts(self, path, parts, n_rows_total, gzip_level=4, chunk_cols=100000):
        """append the three datasets with kmer_matrix chunks made elsewhere: parts = [(first word-row, word-rows, ChunkStreams)]
        covering rows 0 .. n_rows_total in order; this matrix gives the dictionary"""
        n = len(parts)
        bufs = (C.c_void_p * n)(*[p[2].buf.ctypes.data for p in parts])
        starts = (C.c_void_p * n)(*[p[2].starts.ctypes.data for p in parts])
        lens = (C.c_void_p * n)(*[p[2].lens.ctypes.data for p in parts])
        row0 = (C.c_uint64 * n)(*[int(p[0]) for p in parts])
        rows = (C.c_uint64 * n)(*[int(p[1]) for p in parts])
        self.ctx._chk(self.ctx.L.grm_write_kover_h5_parts(self.h, path.encode(), gzip_level, chunk_cols, int(n_rows_total), n, bufs, starts, lens, row0, rows))

    def free(self):
        if self._h:
            self.ctx.L.grm_matrix_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ChunkStreams:
    """finished zlib streams of a dataset's HDF5 chunks: chunk i = buf[starts[i] : starts[i] + lens[i]]"""

    def __init__(self, buf, starts, lens):
        self.buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self.starts = np.ascontiguousarray(starts, dtype=np.uint64)
        self.lens = np.ascontiguousarray(lens, dtype=np.uint32)

    def __len__(self):
        return int(self.lens.shape[0])

    def chunk(self, i):
        a = int(self.starts[i])
        return self.buf[a: a + int(self.lens[i])].tobytes()

    def tofile(self, path):
        """one spool file: n, starts, lens, bytes (a rank of a multi-GPU run hands its rows' chunks to the writing rank)"""
        with open(path, "wb") as f:
            np.array([len(self), self.buf.shape[0]], dtype=np.uint64).tofile(f)
            self.starts.tofile(f)
            self.lens.tofile(f)
            self.buf.tofile(f)

    @classmethod
    def fromfile(cls, path):
        with open(path, "rb") as f:
            n, nb = (int(v) for v in np.fromfile(f, dtype=np.uint64, count=2))
            starts = np.fromfile(f, dtype=np.uint64, count=n)
            lens = np.fromfile(f, dtype=np.uint32, count=n)
            buf = np.fromfile(f, dtype=np.uint8, count=nb)
        return cls(buf, starts, lens)


class DictAccum(_Handle):
    """(k-mer, flag) entries of the local dictionaries of a sequence of batches (pass 1 of the
    two-pass build for inputs that do not fit HBM at once)"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        ctx._adopt(self)

    def add(self, batch):
        self.ctx._chk(self.ctx.L.grm_dict_accum_add(self.h, batch.h))

    def __len__(self):
        return int(self.ctx.L.grm_dict_accum_size(self.h))

    def free(self):
        if self._h:
            self.ctx.L.grm_dict_accum_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Batch(_Handle):
    """device-resident batch of genomes: parse -> partition -> dictionary -> presence bits"""

    def __init__(self, ctx, n_genomes):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._chk(ctx.L.grm_batch_create(ctx.h, n_genomes, C.byref(h)))
        self.h = h
        self.n_genomes = n_genomes
        ctx._adopt(self)

    def add(self, genome_index, data):
        buf = (C.c_char * len(data)).from_buffer_copy(data) if len(data) else None
        self.ctx._chk(self.ctx.L.grm_batch_add(self.h, genome_index, C.cast(buf, C.c_void_p) if buf is not None else None, len(data)))

    def add_array(self, genome_index, arr):
        """arr: contiguous uint8 numpy array (no extra Python-side copy)"""
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        self.ctx._chk(self.ctx.L.grm_batch_add(self.h, genome_index, arr.ctypes.data, arr.size))

    def add_file(self, genome_index, path):
        self.ctx._chk(self.ctx.L.grm_batch_add_file(self.h, genome_index, path.encode()))

    def upload(self):
        self.ctx._chk(self.ctx.L.grm_batch_upload(self.h))

    def run(self, k, abundance_min=1, filter_singleton=False):
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.grm_batch_run(self.h, k, abundance_min, 1 if filter_singleton else 0, C.byref(h)))
        return Matrix(self.ctx, h)

    # staged form (multi-GPU: one collective between local_dict and set_global_dict)
    def partition(self, k, abundance_min=1):
        self.ctx._chk(self.ctx.L.grm_batch_partition(self.h, k, abundance_min))

    def partition_counts(self, k, abundance_min=1):
        self.ctx._chk(self.ctx.L.grm_batch_partition_counts(self.h, k, abundance_min))

    def genome_set(self, genome_index):
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.grm_batch_genome_set(self.h, genome_index, C.byref(h)))
        return KmerSet(self.ctx, h)

    def local_dict(self):
        n = C.c_uint64()
        self.ctx._chk(self.ctx.L.grm_batch_local_dict(self.h, C.byref(n)))
        return int(n.value)

    def export_dict(self, dev_keys_ptr, dev_flags_ptr):
        self.ctx._chk(self.ctx.L.grm_batch_export_dict(self.h, dev_keys_ptr, dev_flags_ptr))

    def set_global_dict(self, dev_keys_ptr, dev_flags_ptr, n, filter_singleton):
        u = C.c_uint64()
        self.ctx._chk(self.ctx.L.grm_batch_set_global_dict(self.h, dev_keys_ptr, dev_flags_ptr, n, 1 if filter_singleton else 0, C.byref(u)))
        return int(u.value)

    def set_global_dict_accum(self, accum, filter_singleton):
        u = C.c_uint64()
        self.ctx._chk(self.ctx.L.grm_batch_set_global_dict_accum(self.h, accum.h, 1 if filter_singleton else 0, C.byref(u)))
        return int(u.value)

    def fill(self):
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.grm_batch_fill(self.h, C.byref(h)))
        return Matrix(self.ctx, h)

    # the exchange as one all-gather of fixed-stride records (see include/grm_kmer.h)
    @property
    def bucket_bits(self):
        """bucket geometry as ranks compare it: the bucket bits in the low byte, + 0x100 for minimizer buckets"""
        return int(self.ctx.L.grm_batch_bucket_bits(self.h))

    def exchange_layout(self, n_max, words, bucket_bits):
        """-> (flags_off, boff_off, stride) of a rank's record"""
        f, o, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.ctx.L.grm_exchange_layout(n_max, words, bucket_bits, C.byref(f), C.byref(o), C.byref(s))
        return int(f.value), int(o.value), int(s.value)

    def export_dict_ordered(self, dev_record_ptr, flags_off, boff_off):
        self.ctx._chk(self.ctx.L.grm_batch_export_dict_ordered(self.h, dev_record_ptr, flags_off, boff_off))

    def export_dict_record(self, dev_record_ptr, n_cap, bucket_bits):
        """the record of a layout fixed in advance (exchange_layout(n_cap, words, bucket_bits)): header always, lists if they fit
        -> True when they did"""
        fits = C.c_int()
        self.ctx._chk(self.ctx.L.grm_batch_export_dict_record(self.h, dev_record_ptr, int(n_cap), int(bucket_bits), C.byref(fits)))
        return bool(fits.value)

    def set_global_dict_gathered(self, dev_payload_ptr, n_max, counts, bucket_bits, filter_singleton, my_rank=-1):
        """my_rank: which record of the payload this batch wrote (export_dict_ordered) -- its entries then take their columns
        from the union's sort; -1: unknown (they are searched in the finished dictionary)"""
        n = len(counts)
        cnt = (C.c_uint64 * n)(*[int(v) for v in counts])
        bbs = (C.c_int * n)(*[int(v) for v in bucket_bits])
        u = C.c_uint64()
        self.ctx._chk(self.ctx.L.grm_batch_set_global_dict_gathered_from(self.h, dev_payload_ptr, n, int(my_rank), n_max, cnt, bbs,
                                                                         1 if filter_singleton else 0, C.byref(u)))
        return int(u.value)

    @property
    def n_symbols(self):
        return int(self.ctx.L.grm_batch_n_symbols(self.h))

    @property
    def n_occurrences(self):
        return int(self.ctx.L.grm_batch_n_occurrences(self.h))

    @property
    def input_bytes(self):
        return int(self.ctx.L.grm_batch_input_bytes(self.h))

    @property
    def n_local(self):
        """entries of the last local dictionary (distinct k-mers of this batch's genomes)"""
        return int(self.ctx.L.grm_batch_n_local(self.h))

    def memo_stats(self):
        """record memo of the last dictionary launch (ctx.set_option("memo_stats", 1)) -> dict, or None without a memo"""
        v = (C.c_uint64 * 4)()
        self.ctx._chk(self.ctx.L.grm_batch_memo_stats(self.h, v))
        held, asked, found, ends = (int(x) for x in v)
        if not ends:
            return None
        return {"records_held_mean": held / ends, "occurrences": asked, "found": found, "hit_rate": found / asked if asked else 0.0}

    def free(self):
        if self._h:
            self.ctx.L.grm_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostMatrix(Matrix):
    """Matrix built from host arrays (rows gathered from several ranks, or tests): only the
    accessors and the two writers work, no device behind it."""

    class _NoCtx:
        def __init__(self, L):
            self.L = L

        def _chk(self, rc):
            if rc != 0:
                raise GrmError(rc, "host-only matrix")

        def _adopt(self, child):
            return child

    def __init__(self, kmers, data, n_genomes, k):
        L = _lib.load()
        kmers = np.ascontiguousarray(kmers, dtype=np.uint64).reshape(-1)
        data = np.ascontiguousarray(data, dtype=np.uint64)
        h = C.c_void_p()
        words = (k + 31) // 32                     # ceil(k / 32) words per k-mer, most significant first
        rc = L.grm_matrix_from_host(kmers.ctypes.data, data.ctypes.data, kmers.size // words, n_genomes, k, C.byref(h))
        if rc:
            raise GrmError(rc, "grm_matrix_from_host")
        self.ctx, self.h = HostMatrix._NoCtx(L), h

    def _err(self):
        return (self.ctx.L.grm_matrix_last_error(self.h) or b"").decode(errors="replace")

    def to_device(self, ctx):
        """-> the same matrix, resident in HBM of `ctx` (a full Matrix: sum_rows, risk_tables, ... work)"""
        ctx._chk(ctx.L.grm_matrix_to_device(ctx.h, self.h))
        m = Matrix(ctx, self.h)
        self.h = None
        return m

    def write_tsv(self, genome_ids, path):
        arr = (C.c_char_p * max(1, len(genome_ids)))(*[g.encode() for g in genome_ids])
        rc = self.ctx.L.grm_write_tsv(self.h, arr, path.encode())
        if rc:
            raise GrmError(rc, self._err())

    def write_tsv_slice(self, genome_ids, path, first_kmer, n_kmers):
        """rows [first_kmer, first_kmer + n_kmers) in place (one of several writers of the file)"""
        arr = (C.c_char_p * max(1, len(genome_ids)))(*[g.encode() for g in genome_ids])
        rc = self.ctx.L.grm_write_tsv_slice(self.h, arr, path.encode(), int(first_kmer), int(n_kmers))
        if rc:
            raise GrmError(rc, self._err())

    def write_kover_h5(self, path, gzip_level=4, chunk_cols=100000):
        rc = self.ctx.L.grm_write_kover_h5(self.h, path.encode(), gzip_level, chunk_cols)
        if rc:
            raise GrmError(rc, self._err())


def decode_kmers(values, k):
    """uint64 k-mer values -> list of ACGT strings (GATB code A=0 C=1 T=2 G=3)"""
    letters = np.frombuffer(b"ACTG", dtype="S1")
    v = np.asarray(values, dtype=np.uint64).reshape(-1)
    shifts = (2 * (k - 1 - np.arange(k))).astype(np.uint64)
    codes = ((v[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.int64)
    return [b"".join(row).decode() for row in letters[codes]]
