"""Host-side restatement of Kover's dataset-creation glue around the k-mer tools
(bin/kover/core/kover/dataset/create.py): `from_contigs` (:278-396), `from_tsv` layout
(:119-275), `_parse_metadata` (:65-116), plus a reader that mirrors what
dataset/ds.py:26-148 and learning/common/rules.py:201-267 (`sum_rows`) need from the file.

The reference runs these as Python 2 + h5py and shells out to multidsk / dsk2kover; here the
orchestration is Python 3 + libhdf5 (ctypes) and the two tools are the gfx950 engine.
"""
import os
import time
import uuid

import numpy as np

from . import h5lite

KMER_MATRIX_PACKING_SIZE = 64          # create.py:38
BLOCK_SIZE = 100000                    # create.py:41


class KoverError(Exception):
    pass


def parse_metadata(metadata_path, matrix_genome_ids, warn=None):
    """create.py:65-116.  -> (ids kept [metadata order], labels uint8, tags, classification_type)"""
    warn = warn or (lambda m: None)
    rows = [l.split() for l in open(metadata_path) if l.strip()]
    md_ids = [r[0] for r in rows]
    md_lab = [r[1] for r in rows]
    uniq = sorted(set(md_lab))                       # np.unique sorts; the non-{0,1} branch re-sorts: same order
    index = {l: i for i, l in enumerate(uniq)}
    if len(uniq) < 2:
        raise KoverError("The dataset must contain at least 2 different phenotypes")
    if len(uniq) > 255:
        raise KoverError("The dataset can contain at most 255 different phenotypes")
    ctype = "binary" if len(uniq) == 2 else "multiclass"
    if len(md_ids) > len(set(md_ids)):
        raise KoverError("The metadata contains multiple values for the same genome.")
    matrix = set(matrix_genome_ids)
    only_matrix = matrix - set(md_ids)
    if only_matrix:
        warn("Missing metadata for %d genomes (%s). These genomes will be discarded." % (len(only_matrix), ", ".join(sorted(only_matrix))))
    only_md = set(md_ids) - matrix
    if only_md:
        warn("The metadata contains values for %d genomes that are not in the genomic data (%s)." % (len(only_md), ", ".join(sorted(only_md))))
    keep = [(i, index[l]) for i, l in zip(md_ids, md_lab) if i in matrix]
    ids = [k[0] for k in keep]
    labels = np.array([k[1] for k in keep], dtype=np.uint8)
    return ids, labels, uniq, ctype


def contigs_path_table(genome_contigs_dir):
    """rows of GRM's `<genome>_paths.tsv` (src/kover.py:40-49): `<file stem>`, `<path>` for every
    `*.fna` of `contigs/<genome name>/` (the tree src/app.py:576-583 downloads into).  Sorted by
    file name so the table does not depend on directory order."""
    names = sorted(f for f in os.listdir(genome_contigs_dir) if f.endswith(".fna"))
    return [(os.path.splitext(f)[0], os.path.abspath(os.path.join(genome_contigs_dir, f))) for f in names]


def create_contigs_path_tsv(contigs_path, genome_name):
    """src/kover.py:40-49: writes `<contigs_path>/<genome_name>_paths.tsv`, returns its path"""
    out = os.path.join(contigs_path, genome_name) + "_paths.tsv"
    with open(out, "w", newline="", encoding="utf-8") as f:
        for stem, p in contigs_path_table(os.path.join(contigs_path, genome_name)):
            f.write("%s\t%s\n" % (stem, p))
    return out


def parse_genome_list(path):
    """`GENOME_ID<ws>PATH` per line (create.py:302); duplicates are an error (:308-309).  A
    directory is read as GRM's contig tree directly (contigs_path_table), without the TSV.
    GRM quotes paths that contain blanks (src/util.py:111-112): the quotes are dropped."""
    out = {}
    order = []
    if os.path.isdir(path):
        for gid, p in contigs_path_table(path):
            out[gid] = p
            order.append(gid)
        if not order:
            raise KoverError("No *.fna file under %s" % path)
        return out, order
    for l in open(path):
        if not l.strip():
            continue
        parts = l.split(None, 1)
        if len(parts) != 2:
            raise KoverError("The genomic data file must hold `GENOME_ID<tab>PATH` lines (got %r)" % l.strip())
        gid, rest = parts[0], parts[1].strip()
        if len(rest) >= 2 and rest[0] == rest[-1] == '"':
            p = rest[1:-1]
        else:
            p = rest.split()[0]
        if gid in out:
            raise KoverError("The genomic data contains genomes with the same identifier.")
        out[gid] = p
        order.append(gid)
    return out, order


def write_header(output_path, source_type, genomic_data, phenotype_description, phenotype_metadata_path, gzip,
                 genome_ids, labels, tags, classification_type, filter_singleton=None):
    """attrs + phenotype / genome_identifiers / phenotype_tags (create.py:311-354); the file is
    closed afterwards exactly as Kover closes it before calling the tools (:356)."""
    with h5lite.File(output_path, "w") as f:
        f.set_attr("created", float(time.time()))
        f.set_attr("uuid", str(uuid.uuid1()))
        f.set_attr("genome_source_type", source_type)
        f.set_attr("genomic_data", genomic_data)
        f.set_attr("phenotype_description", phenotype_description if phenotype_description is not None else "NA")
        f.set_attr("phenotype_metadata_source", phenotype_metadata_path if phenotype_metadata_path is not None else "NA")
        if filter_singleton is not None:
            f.set_attr("filter", filter_singleton)
        f.set_attr("compression", "gzip (level %d)" % gzip)
        if labels is not None:
            f.set_attr("classification_type", classification_type)
            f.create_dataset("phenotype", np.asarray(labels, dtype=np.uint8), attrs={"description": phenotype_description})
        f.create_dataset("genome_identifiers", np.array([g.encode() for g in genome_ids], dtype="S"), gzip=gzip)
        if tags is not None:
            f.create_dataset("phenotype_tags", np.array([t.encode() for t in tags], dtype="S"), gzip=gzip)


def label_sorted(ids, labels):
    """genomes argsorted by numeric label (create.py:334-336); stable for reproducibility"""
    order = np.argsort(labels, kind="stable")
    return [ids[i] for i in order], labels[order]


def _input_bytes(path):
    n = os.path.getsize(path)
    return n * 4 if path.endswith(".gz") else n        # rough inflate factor for FASTA/FASTQ text


def plan_chunks(files_per_genome, budget_bytes, multiple=1):
    """greedy consecutive chunks of genomes whose (estimated, inflated) input stays under the
    budget; a genome larger than the budget gets a chunk of its own.  multiple = 64: every chunk but
    the last holds a multiple of 64 genomes (whole word-rows, for grm_matrix_stack_rows); None when
    the budget cannot hold 64 genomes."""
    chunks, cur, cur_bytes = [], [], 0
    for g, files in enumerate(files_per_genome):
        b = sum(_input_bytes(f) for f in files)
        if cur and cur_bytes + b > budget_bytes:
            chunks.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += b
    if cur:
        chunks.append(cur)
    if multiple > 1 and len(chunks) > 1:
        flat = [g for c in chunks for g in c]
        per = (min(len(c) for c in chunks[:-1]) // multiple) * multiple
        if per == 0:
            return None
        chunks = [flat[a:a + per] for a in range(0, len(flat), per)]
    return chunks


def counted_sets(ctx, files_per_genome, kmer_size, abundance_min, budget_bytes, progress=None):
    """multidsk's job: one solid k-mer set per genome, genomes processed in device batches that fit
    the byte budget (deep read sets do not fit one batch: 100x coverage = 100x the k-mers)"""
    progress = progress or (lambda m: None)
    sets = [None] * len(files_per_genome)
    chunks = plan_chunks(files_per_genome, budget_bytes)
    if kmer_size > 32:                                      # two-word k-mers: per-genome counting path
        for chunk in chunks:
            for g in chunk:
                sets[g] = ctx.count_genome_files(files_per_genome[g], kmer_size, abundance_min)
        return sets

    def load(chunk):
        """read + assemble + upload one chunk (host threads and the copy stream: it runs beside the
        device pass of the previous chunk)"""
        b = ctx.batch(len(chunk))
        try:
            for j, g in enumerate(chunk):
                for f in files_per_genome[g]:
                    b.add_file(j, f)
            b.upload()
        except BaseException:
            b.free()
            raise
        return b

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(load, chunks[0]) if chunks else None
        for i, chunk in enumerate(chunks):
            b = nxt.result()
            nxt = pool.submit(load, chunks[i + 1]) if i + 1 < len(chunks) else None
            try:
                b.partition_counts(kmer_size, abundance_min)
                for j, g in enumerate(chunk):
                    sets[g] = b.genome_set(j)
                progress("counted genomes %d..%d (%d k-mer occurrences)" % (chunk[0], chunk[-1], b.n_occurrences))
            except BaseException:
                if nxt is not None:
                    try:
                        nxt.result().free()
                    except Exception:       # noqa: BLE001 - the first error is the one to report
                        pass
                raise
            finally:
                b.free()
    return sets


def two_pass_matrix(ctx, files_per_genome, chunks, kmer_size, abundance_min, filter_singleton, progress=None):
    """presence matrix of more genomes than HBM holds at once: pass 1 collects every chunk's local
    dictionary in a device accumulator, pass 2 fills every chunk's word-rows against the merged
    dictionary, the row blocks are stacked.  The input is read and uploaded twice (the device pass
    is a small part of a chunk's time and the next chunk loads beside it)."""
    progress = progress or (lambda m: None)
    from concurrent.futures import ThreadPoolExecutor

    def load(chunk):
        b = ctx.batch(len(chunk))
        try:
            for j, g in enumerate(chunk):
                for f in files_per_genome[g]:
                    b.add_file(j, f)
            b.upload()
        except BaseException:
            b.free()
            raise
        return b

    def chunks_loaded(pool):
        nxt = pool.submit(load, chunks[0])
        for i in range(len(chunks)):
            b = nxt.result()
            nxt = pool.submit(load, chunks[i + 1]) if i + 1 < len(chunks) else None
            try:
                yield i, b
            except GeneratorExit:
                if nxt is not None:
                    nxt.result().free()
                raise
            finally:
                b.free()

    acc = ctx.dict_accum()
    parts = []
    try:
        with ThreadPoolExecutor(max_workers=1) as pool:
            occ = 0
            for i, b in chunks_loaded(pool):
                b.partition(kmer_size, abundance_min)
                b.local_dict()
                acc.add(b)
                occ += b.n_occurrences
                progress("pass 1: genomes %d..%d, %d dictionary candidates so far" % (chunks[i][0], chunks[i][-1], len(acc)))
            for i, b in chunks_loaded(pool):
                b.partition(kmer_size, abundance_min)
                b.local_dict()
                n = b.set_global_dict_accum(acc, filter_singleton)
                parts.append(b.fill())
                progress("pass 2: genomes %d..%d filled against %d k-mers" % (chunks[i][0], chunks[i][-1], n))
        m = ctx.stack_rows(parts)
        return m, occ
    finally:
        for p in parts:
            p.free()
        acc.free()


DEFAULT_BATCH_BYTES = int(os.environ.get("GRM_BATCH_BYTES", str(6 * 10**9)))


def matrix_of_files(ctx, files_per_genome, kmer_size, abundance_min, filter_singleton, progress=None):
    """genome x k-mer presence matrix of file sets, by the cheapest route that fits the device:
    one fused pass, two passes over chunks of whole word-rows, or counted sets then merge"""
    progress = progress or (lambda m: None)
    total = sum(_input_bytes(f) for fl in files_per_genome for f in fl)
    if total <= DEFAULT_BATCH_BYTES:
        batch = ctx.batch(len(files_per_genome))    # everything resident at once: fused pass
        for g, fl in enumerate(files_per_genome):
            for f in fl:
                batch.add_file(g, f)
        progress("read %d files (%.2f GB)" % (sum(len(fl) for fl in files_per_genome), total / 1e9))
        batch.upload()
        progress("uploaded")
        m = batch.run(kmer_size, abundance_min, filter_singleton)
        progress("device pass done: %d k-mer occurrences" % batch.n_occurrences)
        batch.free()
        return m
    rows_chunks = plan_chunks(files_per_genome, DEFAULT_BATCH_BYTES, multiple=KMER_MATRIX_PACKING_SIZE)
    if rows_chunks is not None:
        # contig sets beyond one device batch: two passes over chunks of whole word-rows
        progress("%d genomes in %d chunks, two passes" % (len(files_per_genome), len(rows_chunks)))
        return two_pass_matrix(ctx, files_per_genome, rows_chunks, kmer_size, abundance_min, filter_singleton, progress)[0]
    # deep read sets: the reference's own two steps, multidsk then dsk2kover
    progress("%d genomes (%.1f GB) in device batches of at most %.1f GB: counted sets, then their merge" % (len(files_per_genome), total / 1e9, DEFAULT_BATCH_BYTES / 1e9))
    sets = counted_sets(ctx, files_per_genome, kmer_size, abundance_min, DEFAULT_BATCH_BYTES, progress)
    progress("all genomes counted; merging %d solid sets" % len(sets))
    m = ctx.build_matrix(sets, filter_singleton)
    for s_ in sets:
        s_.free()
    return m


def plan_dataset(contig_list_path, phenotype_description, phenotype_metadata_path, warn=None):
    """what create.py:278-336 settles before the tools run: genome ids in ROW ORDER (argsorted by label when a phenotype is
    given, create.py:334-336), labels, tags, classification type, and every genome's files (from-reads: the directory's
    .fastq / .fastq.gz, create.py:402,484-487).  Deterministic: every rank of a multi-GPU run computes the same plan."""
    if (phenotype_description is None) != (phenotype_metadata_path is None):
        raise KoverError("If a phenotype is specified, it must have a description and a metadata file.")
    paths, order = parse_genome_list(contig_list_path)
    for gid, p in paths.items():
        if not os.path.exists(p):
            raise KoverError("The contig file for genome %s cannot be found: %s" % (gid, p))
    labels = tags = ctype = None
    ids = order
    if phenotype_description is not None:
        ids, labels, tags, ctype = parse_metadata(phenotype_metadata_path, list(paths.keys()), warn=warn)
        ids, labels = label_sorted(ids, labels)
    files_per_genome = []
    for gid in ids:
        p = paths[gid]
        if os.path.isdir(p):        # from-reads: every .fastq / .fastq.gz of the directory (create.py:402,484-487)
            fl = sorted(os.path.join(p, f) for f in os.listdir(p) if f.endswith((".fastq", ".fastq.gz", ".fq", ".fq.gz", ".fna", ".fa", ".fasta")))
        else:
            fl = [p]
        files_per_genome.append(fl)
    return ids, labels, tags, ctype, files_per_genome


def from_contigs(ctx, contig_list_path, output_path, kmer_size, filter_singleton, phenotype_description,
                 phenotype_metadata_path, gzip, progress=None, abundance_min=1, source_type="contigs"):
    """create.py:278-396 with the two tool calls replaced by one fused engine pass."""
    _say = progress or (lambda m: None)
    _t0 = time.time()
    progress = lambda m: _say("[%6.2fs] %s" % (time.time() - _t0, m))
    ids, labels, tags, ctype, files_per_genome = plan_dataset(contig_list_path, phenotype_description, phenotype_metadata_path, warn=progress)
    tmp = output_path + ".tmp"
    write_header(tmp, source_type, contig_list_path, phenotype_description, phenotype_metadata_path, gzip,
                 ids, labels, tags, ctype, "singleton" if filter_singleton else "nothing")
    progress("multidsk+dsk2kover (gfx950): %d genomes, k=%d" % (len(ids), kmer_size))
    m = matrix_of_files(ctx, files_per_genome, kmer_size, abundance_min, bool(filter_singleton), progress)
    progress("dictionary: %d k-mers; writing HDF5 (gzip %d)" % (m.n_kmers, gzip))
    m.write_kover_h5(tmp, gzip, BLOCK_SIZE)
    progress("HDF5 written")
    n = m.n_kmers
    m.free()
    os.replace(tmp, output_path)        # never leave a plausible partial output
    return n


def _read_tsv_uniform(tsv_path):
    """fast path for the layout the reference itself assumes (create.py:127-137: every row
    `kmer\tV\tV...V\n` of the same byte length, V one binary digit): the body is viewed as a
    [U][row_len] byte array and sliced -- no per-line Python.  -> (genome_ids, kmers S<k>[U],
    cells uint8 [U][n] of 0/1) or None when the file does not have that layout."""
    raw = np.fromfile(tsv_path, dtype=np.uint8)
    nl = np.flatnonzero(raw[:1 << 22] == 10)
    if nl.size == 0:
        return None
    header = raw[:nl[0]].tobytes().decode().rstrip("\r").split("\t")
    genome_ids = header[1:]
    n = len(genome_ids)
    body = raw[nl[0] + 1:]
    while body.size and body[-1] == 10 and body.size >= 2 and body[-2] == 10:
        body = body[:-1]                                   # trailing blank lines
    if body.size == 0:
        return genome_ids, np.zeros(0, dtype="S1"), np.zeros((0, n), np.uint8)
    if body[-1] != 10:
        body = np.concatenate([body, np.array([10], np.uint8)])
    first = np.flatnonzero(body[:1 << 22] == 10)
    if first.size == 0:
        return None
    row_len = int(first[0]) + 1
    klen = row_len - 1 - 2 * n
    if klen < 1 or body.size % row_len:
        return None
    rows = body.reshape(-1, row_len)
    if not ((rows[:, -1] == 10).all() and (rows[:, klen::2][:, :n] == 9).all()):
        return None
    cells = rows[:, klen + 1::2][:, :n] - np.uint8(48)
    if cells.size and cells.max() > 1:
        return None
    kmers = np.ascontiguousarray(rows[:, :klen]).view("S%d" % klen).reshape(-1)
    return genome_ids, kmers, cells


def _read_tsv_lines(tsv_path):
    """general parser (any row lengths, blank lines, CRLF)"""
    with open(tsv_path) as f:
        header = f.readline().rstrip("\r\n").split("\t")
        genome_ids = header[1:]
        kmers, cols = [], []
        for line in f:
            if not line.strip():
                continue
            cells = line.rstrip("\r\n").split("\t")
            kmers.append(cells[0].encode())
            cols.append(np.array(cells[1:], dtype=np.uint8))
    klen = len(kmers[0]) if kmers else 1
    cells = np.array(cols, dtype=np.uint8) if cols else np.zeros((0, len(genome_ids)), np.uint8)
    return genome_ids, np.array(kmers, dtype="S%d" % klen), cells


def pack_cells(cells, order):
    """cells uint8 [U][n] (k-mer major, as a TSV stores them), order = genome row permutation ->
    uint64 [ceil(len(order)/64)][U], genome i of `order` at word-row i//64, bit 63 - i%64
    (utils.py:133-156), built block-wise without a [n][U] dense intermediate"""
    U, rows = cells.shape[0], (len(order) + 63) // 64
    out = np.zeros((rows, U), dtype=np.uint64)
    idx = np.asarray(order, dtype=np.int64)
    for a in range(0, U, 1 << 20):
        blk = cells[a:a + (1 << 20)][:, idx]                                     # [u][n'] in output row order
        pad = rows * 64 - blk.shape[1]
        if pad:
            blk = np.concatenate([blk, np.zeros((blk.shape[0], pad), np.uint8)], axis=1)
        by = np.packbits(blk, axis=1, bitorder="big")                            # first genome = most significant bit
        out[:, a:a + blk.shape[0]] = by.view(">u8").astype(np.uint64).T
    return out


def from_tsv(tsv_path, output_path, phenotype_description, phenotype_metadata_path, gzip):
    """create.py:119-275: TSV matrix -> Kover HDF5 (pure host code; packs with the MSB-first
    layout of utils.py:133-156)."""
    if (phenotype_description is None) != (phenotype_metadata_path is None):
        raise KoverError("If a phenotype is specified, it must have a description and a metadata file.")
    parsed = _read_tsv_uniform(tsv_path) or _read_tsv_lines(tsv_path)
    genome_ids, kmers, cells = parsed
    if len(set(genome_ids)) < len(genome_ids):
        raise KoverError("The genomic data contains genomes with the same identifier.")
    labels = tags = ctype = None
    ids = genome_ids
    if phenotype_description is not None:
        ids, labels, tags, ctype = parse_metadata(phenotype_metadata_path, genome_ids)
        ids, labels = label_sorted(ids, labels)
    row_of = {g: i for i, g in enumerate(genome_ids)}
    packed = pack_cells(cells, [row_of[g] for g in ids])
    tmp = output_path + ".tmp"
    write_header(tmp, "tsv", tsv_path, phenotype_description, phenotype_metadata_path, gzip, ids, labels, tags, ctype)
    U = len(kmers)
    with h5lite.File(tmp, "r+") as f:
        f.create_dataset("kmer_sequences", kmers if U else np.zeros(0, dtype="S1"), gzip=gzip)
        f.create_dataset("kmer_matrix", packed, gzip=gzip, chunks=(1, max(1, min(U, BLOCK_SIZE))))
        f.create_dataset("kmer_by_matrix_column", np.arange(U, dtype=minimum_uint(U)), gzip=gzip)
    os.replace(tmp, output_path)
    return U


def minimum_uint(max_value):
    """utils.py:117-130"""
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        if max_value <= np.iinfo(dt).max:
            return dt
    raise ValueError(max_value)


def pack_rows(dense):
    """utils.py:133-156 (_pack_binary_bytes_to_ints, pack_size 64): genome i -> word i//64, bit 63-(i%64)"""
    n, U = dense.shape
    rows = (n + 63) // 64
    out = np.zeros((rows, U), dtype=np.uint64)
    for i in range(n):
        out[i // 64] |= dense[i].astype(np.uint64) << np.uint64(63 - (i % 64))
    return out


class KoverDatasetReader:
    """what dataset/ds.py:26-148 exposes, for verification of files we wrote"""

    def __init__(self, path):
        self.path = path

    def _open(self):
        return h5lite.File(self.path, "r")

    def attr(self, name):
        with self._open() as f:
            return f.get_attr(name)

    @property
    def genome_identifiers(self):
        with self._open() as f:
            return [x.decode() for x in f.read("genome_identifiers")]

    @property
    def phenotype(self):
        with self._open() as f:
            return f.read("phenotype"), [x.decode() for x in f.read("phenotype_tags")], f.dataset_attr("phenotype", "description")

    @property
    def kmer_sequences(self):
        with self._open() as f:
            return [x.decode() for x in f.read("kmer_sequences")]

    @property
    def kmer_matrix(self):
        with self._open() as f:
            return f.read("kmer_matrix")

    @property
    def kmer_by_matrix_column(self):
        with self._open() as f:
            return f.read("kmer_by_matrix_column")

    def layout(self, name):
        with self._open() as f:
            return f.layout(name)

    def sum_rows(self, rows):
        """rules.py:201-267 (KmerRuleClassifications.sum_rows): for every k-mer, the number of
        the given genomes that carry it = popcount(word & row_mask) summed over word-rows
        (popcount.pyx:76-95)."""
        m = self.kmer_matrix
        n_rows = m.shape[0]
        mask = np.zeros(n_rows, dtype=np.uint64)
        for r in rows:
            mask[r // 64] |= np.uint64(1) << np.uint64(63 - (r % 64))       # rules.py:210-222
        out = np.zeros(m.shape[1], dtype=np.uint32)
        for w in range(n_rows):
            if mask[w]:
                v = m[w] & mask[w]
                out += np.array([bin(int(x)).count("1") for x in v], dtype=np.uint32) if v.size < 4096 else _popcount64(v)
        return out


def _popcount64(v):
    v = v.copy()
    v = v - ((v >> np.uint64(1)) & np.uint64(0x5555555555555555))
    v = (v & np.uint64(0x3333333333333333)) + ((v >> np.uint64(2)) & np.uint64(0x3333333333333333))
    v = (v + (v >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return ((v * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.uint32)


def device_matrix(ctx, dataset_path):
    """the packed matrix of a .kover file placed in HBM (engine.Matrix): the learner-side kernels -- sum_rows,
    the risk tables of `dataset split` -- then run on the device"""
    from .engine import HostMatrix
    r = KoverDatasetReader(dataset_path)
    with r._open() as f:
        data = f.read("kmer_matrix")
        n_genomes = f.read("genome_identifiers").shape[0]
    # the k-mer values play no part in the learner-side kernels: a zero dictionary of the right length
    hm = HostMatrix(np.zeros(data.shape[1], dtype=np.uint64), data, n_genomes, 1)
    return hm.to_device(ctx)


# ---- kover dataset split (bin/kover/core/kover/dataset/split.py) -----------------------------------
def kmer_risk_tables(sum_rows, n_kmers, labels, train_idx):
    """split.py:171-188: individual risk of the presence rule of every k-mer on the training set,
    rounded to 5 decimals, stored as (unique values, index per k-mer, index per absence rule).
    sum_rows(genome_indices) -> per-k-mer carrier counts (the GPU's grm_matrix_sum_rows or the
    reader's restatement of learning/common/rules.py:201-267)."""
    train_idx = np.asarray(train_idx, dtype=np.int64)
    pos = train_idx[labels[train_idx] == 1]
    neg = train_idx[labels[train_idx] == 0]
    risks = (len(pos) - sum_rows(list(pos))[:n_kmers].astype(np.int64)).astype(np.float64)     # positive errors
    risks += sum_rows(list(neg))[:n_kmers]                                                    # negative errors
    risks /= len(train_idx)
    np.round(risks, 5, out=risks)
    anti = 1.0 - risks
    np.round(anti, 5, out=anti)
    unique, inverse = np.unique(np.hstack((risks, anti)), return_inverse=True)
    dt = minimum_uint(len(unique))
    return unique, inverse[:n_kmers].astype(dt), inverse[n_kmers:].astype(dt)


def split(dataset_path, split_name, train_idx, test_idx, random_seed, n_folds=0, random_generator=None, sum_rows=None,
          risk_tables=None):
    """split.py:110-230 (_split): writes splits/<name>/{train,test}_genome_idx, the risk tables and the
    cross-validation folds into the dataset file."""
    r = KoverDatasetReader(dataset_path)
    rng = random_generator if random_generator is not None else np.random.RandomState(random_seed)
    with h5lite.File(dataset_path, "r") as f:
        if f.get_attr("phenotype_description") == "NA":
            raise KoverError("A dataset must contain phenotypic metadata to be split.")
        if f.exists("splits/" + split_name):
            raise KoverError('A split with the identifier "%s" already exists in the dataset.' % split_name)
        labels = f.read("phenotype")
        n_genomes = f.read("genome_identifiers").shape[0]
        n_kmers = f.read("kmer_by_matrix_column").shape[0]
    train_idx = np.array(train_idx, dtype=np.int64)
    test_idx = np.array(test_idx, dtype=np.int64)
    if n_folds > len(train_idx):
        raise KoverError("There cannot be more cross-validation folds (%d) than genomes in the training set (%d)." % (n_folds, len(train_idx)))
    if n_folds == 1:
        raise KoverError("The number of cross-validation folds must be greater than 1.")
    if len(set(train_idx.tolist())) < len(train_idx):
        raise KoverError("The training set contains duplicate genomes.")
    if len(set(test_idx.tolist())) < len(test_idx):
        raise KoverError("The testing set contains duplicate genomes.")
    if len(set(train_idx.tolist()) | set(test_idx.tolist())) < len(train_idx) + len(test_idx):
        raise KoverError("The training and testing sets overlap.")
    sum_rows = sum_rows or r.sum_rows
    idx_dt = minimum_uint(n_genomes)

    def write_part(f, group, tr, te):
        f.create_dataset(group + "/train_genome_idx", np.sort(tr).astype(idx_dt))
        f.create_dataset(group + "/test_genome_idx", np.sort(te).astype(idx_dt))
        # risk_tables(labels, train_idx): the device form (engine.Matrix.risk_tables); else the numpy restatement
        unique, by_kmer, by_anti = risk_tables(labels, tr) if risk_tables else kmer_risk_tables(sum_rows, n_kmers, labels, tr)
        f.create_dataset(group + "/unique_risks", unique)
        f.create_dataset(group + "/unique_risk_by_kmer", by_kmer)
        f.create_dataset(group + "/unique_risk_by_anti_kmer", by_anti)

    with h5lite.File(dataset_path, "r+") as f:
        if not f.exists("splits"):
            f.create_group("splits")
        g = "splits/" + split_name
        f.create_group(g)
        f.set_group_attr(g, "random_seed", int(random_seed))
        f.set_group_attr(g, "n_folds", int(n_folds))
        f.set_group_attr(g, "train_proportion", 1.0 * len(train_idx) / n_genomes)
        f.set_group_attr(g, "test_proportion", 1.0 * len(test_idx) / n_genomes)
        write_part(f, g, train_idx, test_idx)
        if n_folds > 0:
            f.create_group(g + "/folds")
            fold_of = np.arange(len(train_idx)) % n_folds          # split.py:197-198
            rng.shuffle(fold_of)
            for fold in range(n_folds):
                fg = "%s/folds/fold_%d" % (g, fold + 1)
                f.create_group(fg)
                write_part(f, fg, train_idx[fold_of != fold], train_idx[fold_of == fold])


def split_with_proportion(dataset_path, split_name, train_prop, random_seed, n_folds=0, sum_rows=None, risk_tables=None):
    """split.py:86-107: the same RandomState stream as the reference (shuffle of arange, then the folds)"""
    rng = np.random.RandomState(random_seed)
    with h5lite.File(dataset_path, "r") as f:
        n = f.read("genome_identifiers").shape[0]
    n_train = int(np.ceil(train_prop * n))
    idx = np.arange(n)
    rng.shuffle(idx)
    split(dataset_path, split_name, idx[:n_train], idx[n_train:], random_seed, n_folds, rng, sum_rows, risk_tables)


def split_with_ids(dataset_path, split_name, train_ids_file, test_ids_file, random_seed, n_folds=0, sum_rows=None, risk_tables=None):
    """split.py:31-83"""
    rng = np.random.RandomState(random_seed)
    ids = KoverDatasetReader(dataset_path).genome_identifiers
    index = {g: i for i, g in enumerate(ids)}

    def parse(path, step):
        got = [l.strip() for l in open(path).read().split("\n") if l.strip()]
        missing = [g for g in got if g not in index]
        if missing:
            raise KoverError("The %s genome identifiers contain IDs that are not in the dataset: %s" % (step, ", ".join(missing)))
        return [index[g] for g in got]

    split(dataset_path, split_name, parse(train_ids_file, "training"), parse(test_ids_file, "testing"), random_seed, n_folds, rng, sum_rows,
          risk_tables)
