"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle, bit-exact.

Run on the MI355X box: python -m pytest tests -m gpu
"""
import os
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle_ctypes as orc
from tests import cases

grm = None
synth = None


@pytest.fixture(scope="module")
def ctx():
    global grm, synth
    import grm_amd
    grm = grm_amd
    synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
    c = grm_amd.Context(0)
    yield c
    c.close()


def _as_bytes(genomes):
    return [[t.encode() if isinstance(t, str) else bytes(t) for t in g] for g in genomes]


def _run_batch(ctx, genomes, k, amin, filt):
    b = ctx.batch(len(genomes))
    for g, files in enumerate(genomes):
        for f in files:
            b.add(g, f)
    b.upload()
    m = b.run(k, amin, filt)
    out = (m.kmers(), m.data(), b.n_occurrences, m.column_counts())
    m.free()
    b.free()
    return out


def _check(ctx, genomes, k, amin, filt):
    genomes = _as_bytes(genomes)
    want = orc.build_matrix(genomes, k, amin, filt)
    kmers, data, n_occ, colcnt = _run_batch(ctx, genomes, k, amin, filt)
    assert n_occ == want["n_occurrences"]
    assert kmers.shape == want["kmers"].shape
    assert (kmers == want["kmers"]).all()
    assert data.shape == want["matrix"].shape
    assert (data == want["matrix"]).all()
    assert (colcnt == want["n_genomes_with"]).all()


MICRO = [c for c in cases.micro_cases() if c[1] <= 32]


@pytest.mark.parametrize("name,k,genomes", MICRO, ids=lambda x: x if isinstance(x, str) else None)
@pytest.mark.parametrize("amin,filt", [(1, False), (1, True), (2, False), (2, True)])
def test_micro_matrix(ctx, name, k, genomes, amin, filt):
    _check(ctx, genomes, k, amin, filt)


@pytest.mark.parametrize("name,k,genomes", MICRO[:3], ids=lambda x: x if isinstance(x, str) else None)
def test_count_genome_sets(ctx, name, k, genomes):
    for texts in genomes:
        files = [t.encode() for t in texts]
        for amin in (1, 2):
            km, ct, nocc = orc.count_genome(files, k, amin)
            s = ctx.count_genome(files, k, amin)
            assert s.occurrences == nocc
            assert (s.kmers() == km).all() and s.kmers().shape == km.shape
            assert (s.counts() == ct).all()
            s.free()


def test_build_matrix_from_sets(ctx):
    name, k, genomes = MICRO[3]
    bg = _as_bytes(genomes)
    for filt in (False, True):
        want = orc.build_matrix(bg, k, 1, filt)
        sets = [ctx.count_genome(files, k, 1) for files in bg]
        m = ctx.build_matrix(sets, filt)
        assert (m.kmers() == want["kmers"]).all() and m.kmers().shape == want["kmers"].shape
        assert (m.data() == want["matrix"]).all()
        # sets rebuilt from host arrays (what dsk2kover does after re-loading multidsk output)
        sets2 = [ctx.kmer_set_from_arrays(s.kmers(), s.counts(), k) for s in sets]
        m2 = ctx.build_matrix(sets2, filt)
        assert (m2.data() == want["matrix"]).all()
        for o in sets + sets2 + [m, m2]:
            o.free()


def _medium_genomes(n=7, length=300_000, seed=5):
    pg = synth.PanGenome(genome_len=length, n_snps=3000, n_accessory=12, accessory_len=2000, seed=seed, n_contigs=3)
    return [[pg.genome(i).tobytes()] for i in range(n)]


def test_medium_pangenome_all_paths(ctx):
    """300 kbp genomes: the LDS (span-uniform) paths of hist / scatter and real bucket counts"""
    genomes = _medium_genomes()
    for (amin, filt) in [(1, True), (1, False), (2, True)]:
        _check(ctx, genomes, 31, amin, filt)


@pytest.mark.parametrize("opts", [
    {"groups_per_thread": 1}, {"groups_per_thread": 3, "bucket_bits": 5},
    {"bucket_bits": 0}, {"bucket_bits": 13}, {"sub_bits": 3},
    {"bucket_bits": 14}, {"bucket_bits": 16},          # deep mode: fine histogram from the level-1 output
    {"cap_log2": 8, "bucket_bits": 4},          # forces overflow -> sub-bucket retries
    {"no_slots": 1}, {"no_slots": 1, "sub_bits": 2},   # probing form of the fill
    {"dense_layout": 1}, {"dense_layout": 1, "bucket_bits": 11},      # histogram-sized layout instead of the slack layout
    {"direct_permute": 1},                             # scattered single-step fill
    {"records": 0},                                    # key form of the partition (hashed buckets, levels 1 and 2 on keys)
    {"dict_sort_prim": 1},                             # dictionary sorted by the general radix sort instead of the key-range sort
    {"rec_keys": 1}, {"rec_keys": 1, "bucket_bits": 10},       # record form, level 2 expanding to key segments
    {"rec_part_bits": 2}, {"rec_part_bits": 3, "rec_keys": 1},  # genomes cut into parts (one workgroup each)
    {"rec_bucket_shift": 1}, {"bucket_bits": 6}, {"bucket_bits": 9, "sub_bits": 1},
    {"rec_part_bits": 1, "no_slots": 1},               # probing fill asked of a partition in parts: falls back to the key form
    {"cap_log2": 7}, {"cap_log2": 8, "rec_bucket_shift": 0},   # record form: tables overflow -> more buckets (level 2 again), then sub-buckets
    {"rec_memo": 0}, {"rec_memo": 8}, {"rec_memo": 11},        # dict_build's record memo: none, smallest, largest
    {"rec_memo": 8, "cap_log2": 9}, {"rec_memo": 9, "sub_bits": 2}, {"rec_memo": 10, "rec_part_bits": 2, "bucket_bits": 7},
    {"cap_log2": 13}, {"cap_log2": 12, "rec_memo": 11},        # forced big tables: the memo only where it fits beside them
])
def test_medium_with_forced_geometry(ctx, opts):
    genomes = _medium_genomes(n=5, length=120_000, seed=9)
    try:
        for name, v in opts.items():
            ctx.set_option(name, v)
        _check(ctx, genomes, 31, 1, True)
        _check(ctx, genomes, 21, 1, False)
    finally:
        for name in opts:
            ctx.set_option(name, -1)


@pytest.mark.parametrize("k", [11, 12, 13, 16, 21, 27, 28, 29, 30, 32])
def test_record_form_over_k(ctx, k):
    """the record form of the partition (minimizer buckets, runs of k-mers as records) at both ends of its range of k: one
    m-mer per k-mer (k = 11, every start opens a run), the longest windows, and the widest keys (k = 32)"""
    genomes = _medium_genomes(n=4, length=90_000, seed=21 + k)
    rng = np.random.RandomState(k)
    genomes.append([cases.fasta([("low", "ACGT" * 500 + "A" * 300 + cases.rand_seq(rng, 4000) + "N" * 40 + "CAG" * 400)], width=70)])
    for opts in ({"records": 1}, {"records": 1, "rec_keys": 1}):        # (records = 1: also below k = 19, where the key form is the default)
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            _check(ctx, genomes, k, 1, False)
        finally:
            for name in opts:
                ctx.set_option(name, -1)


@pytest.mark.parametrize("opts", [{}, {"rec_memo": 8, "bucket_bits": 5}, {"rec_memo": 0}, {"rec_memo": 9, "bucket_bits": 9, "sub_bits": 1}])
def test_record_memo_over_word_rows(ctx, opts):
    """dict_build's record memo over several word-rows (150 related genomes = 3 rows: records met in row 0 are found again in
    rows 1 and 2, their k-mers' table slots resolved once), with a memo far too small for its bucket (most occurrences go
    the direct way next to the held ones), and with sub-buckets (a record's k-mers belong to different workgroups)"""
    pg = synth.PanGenome(genome_len=24_000, n_snps=900, n_accessory=6, accessory_len=900, seed=77, n_contigs=2)
    genomes = [[pg.genome(i).tobytes()] for i in range(150)]
    try:
        for name, v in opts.items():
            ctx.set_option(name, v)
        _check(ctx, genomes, 31, 1, True)
        _check(ctx, genomes, 23, 1, False)
    finally:
        for name in opts:
            ctx.set_option(name, -1)


def test_record_memo_switches_itself_off_on_unrelated_genomes(ctx):
    """unrelated genomes share no records: a small memo fills up in the first genomes, is hardly ever hit and is given up
    by its workgroup after a word-row (the pooled two-class form takes over) -- same matrix"""
    genomes = [[synth.random_genome(i, genome_len=16_000, seed=4321).tobytes()] for i in range(140)]
    try:
        ctx.set_option("rec_memo", 8)
        ctx.set_option("bucket_bits", 4)
        _check(ctx, genomes, 31, 1, False)
    finally:
        ctx.set_option("rec_memo", -1)
        ctx.set_option("bucket_bits", -1)


def _realistic_genomes(n, length, seed, **kw):
    pg = synth.realistic(genome_len=length, seed=seed, n_snps=length // 100, n_accessory=6, accessory_len=1500, **kw)
    return [[pg.genome(i).tobytes()] for i in range(n)]


@pytest.mark.parametrize("k", [31, 21])
def test_realistic_assemblies_bit_exact(ctx, k):
    """GRM's real inputs (src/app.py:576-583): every genome its own contigs, cut at its own places, in its own order, each on a
    random strand, with indels between the strains -- no two genomes share a coordinate frame.  Bit-exact against the oracle, on
    the record form with and without its memo, in parts, and on the key form"""
    genomes = _realistic_genomes(70, 60_000, 31, indel_sites=40, contigs=(3, 9))
    for opts in ({}, {"rec_memo": 0}, {"rec_part_bits": 2}, {"records": 0}):
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            _check(ctx, genomes, k, 1, True)
        finally:
            for name in opts:
                ctx.set_option(name, -1)


def test_record_memo_survives_frame_and_strand(ctx):
    """the same strains once as in round 2's generator (one coordinate frame, forward strand) and once as real assemblies
    (contigs cut anywhere, shuffled, on random strands, indels): run records are cut where the SEQUENCE says and stored
    strand-canonically, so dict_build's memo finds nearly every record occurrence in both"""
    rates = {}
    for name, kw in (("one frame", dict(indel_sites=0, contigs=None, shuffle_contigs=False, random_strand=False, n_contigs=1)),
                     ("assemblies", dict(indel_sites=30, contigs=(10, 30)))):
        genomes = _realistic_genomes(128, 300_000, 77, **kw)
        b = ctx.batch(len(genomes))
        for g, files in enumerate(genomes):
            b.add(g, files[0])
        b.upload()
        try:
            ctx.set_option("memo_stats", 1)
            m = b.run(31, 1, True)
            st = b.memo_stats()
        finally:
            ctx.set_option("memo_stats", -1)
        assert st is not None and st["occurrences"] > 1_000_000
        rates[name] = st["hit_rate"]
        m.free()
        b.free()
    assert rates["one frame"] > 0.95, rates
    assert rates["assemblies"] > 0.93, rates             # (a contig end or an indel costs the two runs around it)
    assert rates["assemblies"] > rates["one frame"] - 0.04, rates


def test_hundred_unrelated_genomes_through_the_matrix_path(ctx):
    """the shape of round 2's memory fault (bench.py --mode R --genomes 100: unrelated genomes through the record form's
    overflow ladder -- more buckets via level 2 again, then the key form), reduced to what the oracle checks in seconds"""
    genomes = [[synth.random_genome(i, genome_len=400_000, seed=99).tobytes()] for i in range(100)]
    _check(ctx, genomes, 31, 1, False)
    try:
        ctx.set_option("cap_log2", 9)                   # small tables: the ladder has to climb (more buckets, then sub-buckets / key form)
        _check(ctx, genomes, 31, 1, True)
    finally:
        ctx.set_option("cap_log2", -1)


def test_record_form_gives_way_to_the_key_form_on_repeats(ctx):
    """repeat-rich genomes: nearly all k-mers of a genome share a handful of minimizers, so a few record regions would hold
    most of its records -- the partition raises its overflow flag and is redone in the key form (hashed k-mers spread
    evenly whatever the sequence); the batch stays there for later runs"""
    rng = np.random.RandomState(4)
    genomes = []
    for g in range(3):
        unit = cases.rand_seq(rng, 37)
        genomes.append([cases.fasta([("rep%d" % g, unit * 6000 + cases.rand_seq(rng, 20_000) + "AC" * 40_000)], width=80)])
    ctx.timing(True)
    try:
        ctx.timing_reset()
        _check(ctx, genomes, 31, 1, False)
        names = [n for n, _, _ in ctx.timings()]
        assert "superkmer_l1" in names and "kmer_scatter_l1" in names, names        # tried, then redone
    finally:
        ctx.timing(False)


def test_dictionary_sort_with_crowded_key_ranges(ctx):
    """the dictionary's key-range sort (grm_dictsort.hip) splits the k-mers by their top bits and sorts every range inside LDS: 12 000
    k-mers that all begin with AAAAAA crowd into one range that does not fit -- the flag goes up and the general sort takes over; the
    same genomes with ordinary k-mers beside them; and the two sorts agree on a pan-genome"""
    rng = np.random.RandomState(3)
    crowd = "".join(">r%d\nAAAAAA%s\n" % (i, cases.rand_seq(rng, 25)) for i in range(12_000)).encode()
    crowd2 = "".join(">r%d\nAAAAAAA%s\n" % (i, cases.rand_seq(rng, 24)) for i in range(9_000)).encode()
    _check(ctx, [[crowd], [crowd2], [crowd[: len(crowd) // 2]]], 31, 1, False)
    _check(ctx, [[crowd, (">x\n" + cases.rand_seq(rng, 60_000) + "\n").encode()], [crowd2]], 31, 1, True)
    genomes = _medium_genomes(n=6, length=150_000, seed=5)
    for prim in (0, 1):
        try:
            ctx.set_option("dict_sort_prim", prim)
            _check(ctx, genomes, 31, 1, True)
            _check(ctx, genomes, 19, 1, False)
            _check(ctx, genomes, 32, 1, False)
        finally:
            ctx.set_option("dict_sort_prim", -1)


def test_single_pass_parse_gives_the_same_stream(ctx):
    """option parse_fused: one kernel reads every tile once and learns the parser state / symbol offset running into it by a
    decoupled look-back over the tiles before it (FASTA and FASTQ tiles, many files, tiles without symbols); the groups a tile
    boundary falls into are put together by parse_stitch.  Same matrices and counted sets as the two-pass parse."""
    rng = np.random.RandomState(5)
    genomes = _medium_genomes(n=9, length=130_000, seed=12)
    genomes.append([(">" + "h" * 40_000 + "\n" + cases.rand_seq(rng, 20_000) + "\n>" + "x" * 16_380 + "\n" + cases.rand_seq(rng, 30_000)).encode()])
    genomes.append([b">empty\n", cases.fasta([("a", cases.rand_seq(rng, 70_001))], width=61, crlf=True).encode()])
    try:
        ctx.set_option("parse_fused", 1)
        _check(ctx, genomes, 31, 1, False)
        _check(ctx, genomes, 21, 2, True)
        for name, k, gs in cases.fastq_cases():
            _check(ctx, gs, k, 1, False)
        reads = [cases.rand_seq(rng, 150) for _ in range(3000)]
        _check(ctx, [[cases.fastq(reads).encode()], [cases.fastq(reads[:1700]).encode()]], 21, 2, False)
    finally:
        ctx.set_option("parse_fused", -1)


def test_parser_stress_layouts(ctx):
    """single-line sequences spanning many 16 KiB tiles, headers longer than a tile, headers that
    straddle tile boundaries, CRLF, blank lines, no trailing newline, lowercase / N runs"""
    rng = np.random.RandomState(77)
    a = cases.rand_seq(rng, 120_000)
    b = cases.rand_seq(rng, 50_000)
    g0 = (">single line\n" + a + "\n").encode()                                  # no newline for 7 tiles
    g1 = (">" + "h" * 40_000 + "\n" + b[:20_000] + "\n>" + "x" * 16_380 + "\n" + b[20_000:]).encode()   # huge headers, no final newline
    g2 = cases.fasta([("crlf", a[:60_000].lower()), ("n", a[60_000:61_000] + "N" * 5000 + a[61_000:70_000])], width=61, crlf=True).encode()
    g3 = (">e\n\n\n" + b[:100] + "\n\n>f\n" + b[100:300] + "\n\n").encode()
    pad = 16384 - 2 - len(">p\n") - 1          # put the next header's '>' on the last byte of a tile
    g4 = (">p\n" + a[:pad - 1] + "\n>q straddle\n" + a[pad:pad + 5000] + "\n").encode()
    genomes = [[g0], [g1], [g2], [g3], [g4], [g0, g4], [g1, g3, g2]]
    for k in (31, 12):
        _check(ctx, genomes, k, 1, False)
    _check(ctx, genomes, 31, 2, True)


def test_more_than_64_genomes_and_ragged(ctx):
    rng = np.random.RandomState(17)
    core = cases.rand_seq(rng, 3000)
    genomes = []
    for g in range(131):
        s = list(core)
        for p in rng.randint(0, len(core), size=5):
            s[p] = "ACGT"[rng.randint(4)]
        ln = int(rng.randint(0, len(core)))
        genomes.append([cases.fasta([("g%d" % g, "".join(s[:ln]))], width=70)])
    genomes[5] = [""]           # empty file
    genomes[64] = [">only header\n"]
    _check(ctx, genomes, 31, 1, True)
    _check(ctx, genomes, 15, 1, False)


@pytest.mark.parametrize("name,k,genomes", cases.fastq_cases(), ids=lambda x: x if isinstance(x, str) else None)
@pytest.mark.parametrize("amin,filt", [(1, False), (2, False), (2, True)])
def test_fastq_reads(ctx, name, k, genomes, amin, filt):
    _check(ctx, genomes, k, amin, filt)


def test_fastq_gzip_and_mixed_formats(ctx):
    import gzip
    rng = np.random.RandomState(31)
    ref = cases.rand_seq(rng, 30000)
    reads = [ref[s:s + 150] for s in rng.randint(0, len(ref) - 150, size=3000)]
    for i in range(0, len(reads), 7):
        reads[i] = cases.revcomp(reads[i])
    fq1 = cases.fastq(reads[:1500]).encode()
    fq2 = cases.fastq(reads[1500:]).encode()
    fa = cases.fasta([("ref", ref)], width=80).encode()
    plain = [[fq1, fq2], [fa], [fq2]]
    gz = [[gzip.compress(fq1), fq2], [gzip.compress(fa)], [gzip.compress(fq2[:len(fq2) // 2]) + gzip.compress(fq2[len(fq2) // 2:])]]
    for amin in (1, 2, 3):
        want = orc.build_matrix(plain, 21, amin, False)
        kmers, data, n_occ, _ = _run_batch(ctx, gz, 21, amin, False)
        assert n_occ == want["n_occurrences"]
        assert (kmers == want["kmers"]).all() and kmers.shape == want["kmers"].shape
        assert (data == want["matrix"]).all()
    # per-genome counted set of a read set (multidsk on reads: abundance-min 2)
    km, ct, nocc = orc.count_genome([fq1, fq2], 21, 2)
    s = ctx.count_genome([gzip.compress(fq1), fq2], 21, 2)
    assert s.occurrences == nocc and (s.kmers() == km).all() and (s.counts() == ct).all()
    s.free()


def test_deep_read_set_counts(ctx):
    """one genome sequenced at ~40x with errors, k=21, abundance-min 2 (multidsk on reads): enough
    occurrences per bucket that the engine switches to more than 2^13 buckets on its own"""
    rng = np.random.RandomState(41)
    ref = cases.rand_seq(rng, 150_000)
    starts = rng.randint(0, len(ref) - 150, size=40_000)
    reads = []
    for s0 in starts:
        r = list(ref[s0:s0 + 150])
        for p in np.nonzero(rng.rand(150) < 0.005)[0]:
            r[p] = "ACGT"[rng.randint(4)]
        r = "".join(r)
        reads.append(cases.revcomp(r) if rng.rand() < 0.5 else r)
    fq = cases.fastq(reads).encode()
    # (default: the read set is cut into 64 parts for level 1 and counted over its parts' record segments, record_merge;
    # "rec_count" 0: the key form; a table of 2^13 slots from the start; fewer parts: the key form again)
    for opts in ({}, {"bucket_bits": 15}, {"rec_count": 0}, {"rec_count_cap": 13}, {"rec_part_bits": 5}, {"rec_part_bits": 2}):
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            for amin in (1, 2):
                km, ct, nocc = orc.count_genome([fq], 21, amin)
                ctx.timing(True)
                ctx.timing_reset()
                s = ctx.count_genome([fq], 21, amin)
                names = {t[0] for t in ctx.timings()}
                assert ("record_merge" in names) == (opts.get("rec_count", -1) != 0 and opts.get("rec_part_bits", 6) >= 5), names
                assert s.occurrences == nocc
                assert s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
                s.free()
        finally:
            ctx.timing(False)
            for name in opts:
                ctx.set_option(name, -1)
    # three read sets in one batch (the parts of a genome are merged, not those of its neighbours), one of them tiny
    fqs = [fq, cases.fastq(reads[:9000]).encode(), cases.fastq(reads[100:130]).encode()]
    for amin in (1, 3):
        b = ctx.batch(3)
        for g, f in enumerate(fqs):
            b.add(g, f)
        b.upload()
        b.partition_counts(21, amin)
        for g, f in enumerate(fqs):
            km, ct, nocc = orc.count_genome([f], 21, amin)
            s = b.genome_set(g)
            assert s.occurrences == nocc
            assert s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
            s.free()
        b.free()
    # ... and the matrix of the solid k-mers of the three (abundance-min 2: the dictionary is built from the merged, filtered segments)
    ctx.timing(True)
    ctx.timing_reset()
    try:
        _check(ctx, [[f] for f in fqs], 21, 2, False)
        assert "record_merge" in {t[0] for t in ctx.timings()}
    finally:
        ctx.timing(False)


@pytest.mark.parametrize("k,amin", [(31, 1), (63, 1), (40, 2), (70, 1), (100, 2), (128, 1)])
def test_staged_equals_fused(ctx, k, amin):
    """the staged (multi-GPU) API on one device; k = 63: keys travel as (hi, lo) pairs; k > 64, or an abundance filter at k > 32: the
    sort path in stages -- rows of three / four words, most significant first (grm_batch_run covers those in one call)"""
    import torch
    w = (k + 31) // 32
    genomes = _medium_genomes(n=4, length=100_000, seed=3)
    if amin > 1:          # every genome twice over, the second copy with its own SNPs: most k-mers are seen twice, some once
        other = _medium_genomes(n=4, length=100_000, seed=4)
        genomes = [[a[0] + b[0]] for a, b in zip(genomes, [other[0]] + genomes[:3])]
    want = orc.build_matrix(genomes, k, amin, True)
    b = ctx.batch(len(genomes))
    for g, files in enumerate(genomes):
        b.add(g, files[0])
    b.upload()
    b.partition(k, amin)
    n_local = b.local_dict()
    keys = torch.empty((max(1, n_local), w), dtype=torch.int64, device="cuda:0")
    flags = torch.empty(max(1, n_local), dtype=torch.uint8, device="cuda:0")
    b.export_dict(keys.data_ptr(), flags.data_ptr())
    torch.cuda.synchronize()
    # local dictionary = every distinct k-mer of the union, flag 2 where >1 genome carries it
    allm = orc.build_matrix(genomes, k, amin, False)
    k_host = keys.cpu().numpy().view(np.uint64)[:n_local]
    order = np.lexsort(tuple(k_host[:, j] for j in reversed(range(w))))
    assert n_local == allm["kmers"].shape[0] and (k_host[order].reshape(allm["kmers"].shape) == allm["kmers"]).all()
    assert ((flags.cpu().numpy()[:n_local][order] == 2) == (allm["n_genomes_with"] > 1)).all()
    u = b.set_global_dict(keys.data_ptr(), flags.data_ptr(), n_local, True)
    assert u == want["kmers"].shape[0]
    m = b.fill()
    assert (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
    m.free()
    # and as two "ranks" on the one device: the genomes in two batches, their dictionaries handed to both
    halves = [genomes[:1], genomes[1:]]
    bs, ks, fs, ns = [], [], [], []
    for part in halves:
        bb = ctx.batch(len(part))
        for g, files in enumerate(part):
            bb.add(g, files[0])
        bb.upload()
        bb.partition(k, amin)
        n = bb.local_dict()
        kk = torch.empty((max(1, n), w), dtype=torch.int64, device="cuda:0")
        ff = torch.empty(max(1, n), dtype=torch.uint8, device="cuda:0")
        bb.export_dict(kk.data_ptr(), ff.data_ptr())
        bs.append(bb); ks.append(kk[:n]); fs.append(ff[:n]); ns.append(n)
    torch.cuda.synchronize()
    allk, allf = torch.cat(ks).contiguous(), torch.cat(fs).contiguous()
    for filt in (True, False):
        wantf = orc.build_matrix(genomes, k, amin, filt)
        got = np.zeros_like(wantf["matrix"])
        for i, bb in enumerate(bs):
            assert bb.set_global_dict(allk.data_ptr(), allf.data_ptr(), sum(ns), filt) == wantf["kmers"].shape[0]
            mm = bb.fill()
            assert (mm.kmers() == wantf["kmers"]).all()
            d = mm.data()                      # one word-row per batch here: OR the rows together, each batch's genomes at its own bits
            shift = sum(len(h) for h in halves[:i])
            got[0] |= d[0] >> np.uint64(shift)
            mm.free()
        assert (got == wantf["matrix"]).all()
    for bb in bs:
        bb.free()
    b.free()


def _rank_worker(rank, world, port, n_genomes, k, genome_len, q, amin=1):
    """one rank of the sharded path with the REAL engine; both ranks share cuda:0 (gloo, host-staged)"""
    import torch
    import torch.distributed as dist
    import grm_amd
    D = import_module("genomic-resistance-mapping-grm-_amd.distributed")
    S = import_module("genomic-resistance-mapping-grm-_amd.synth")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        pg = S.PanGenome(genome_len=genome_len, n_snps=genome_len // 100, n_accessory=6, accessory_len=1500, seed=21, n_contigs=2)
        a, b_ = D.shard_genomes(n_genomes, world)[rank]
        with grm_amd.Context(0) as c:
            batch = c.batch(b_ - a)
            for g in range(a, b_):
                batch.add_array(g - a, pg.genome(g))
            batch.upload()
            dev = torch.device("cuda", 0)
            m = D.sharded_step(batch, k, amin, True, dev)
            rows = D.gather_rows(m.data(), dev)
            if rank == 0:
                q.put((m.kmers().copy(), rows))
            m.free()
            batch.free()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,genome_len,n_genomes", [(31, 60_000, 100), (47, 60_000, 100), (31, 1_500_000, 100), (63, 1_500_000, 100),
                                                     (47, 30_000, 400), (63, 30_000, 300), (95, 60_000, 100)])
def test_two_ranks_real_engine_one_gpu(ctx, k, genome_len, n_genomes):
    """N>1 path end to end with the HIP engine: 2 processes, genomes sharded 64 + 36, dictionary
    all-gather, identical global dictionary, rows stacked == single-process oracle matrix
    (1.5 Mbp genomes: millions of dictionary entries per rank; 400 / 300 genomes at k > 32: shards of 256 + 144 genomes, both through
    the 24-byte record form, and of 192 + 108, the second through the key form)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_worker, args=(r, 2, port, n_genomes, k, genome_len, q)) for r in range(2)]
    for p in procs:
        p.start()
    kmers, rows = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    pg = synth.PanGenome(genome_len=genome_len, n_snps=genome_len // 100, n_accessory=6, accessory_len=1500, seed=21, n_contigs=2)
    want = orc.pipeline([pg.genome(g).tobytes() for g in range(n_genomes)], k, 1, True, min(os.cpu_count() or 1, 32))[0]
    assert kmers.shape == want["kmers"].shape and (kmers == want["kmers"]).all()
    assert rows.shape == want["matrix"].shape and (rows == want["matrix"]).all()


def _growing_worker(rank, world, port, q):
    """two steps of one process group with the REAL engine: the second step's dictionaries are ten times the first's, so the layout the
    first step left behind does not hold them -- every rank sends its header alone, reads the others', and the step is repeated"""
    import torch
    import torch.distributed as dist
    import grm_amd
    D = import_module("genomic-resistance-mapping-grm-_amd.distributed")
    S = import_module("genomic-resistance-mapping-grm-_amd.synth")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        out = []
        with grm_amd.Context(0) as c:
            dev = torch.device("cuda", 0)
            for genome_len in (20_000, 200_000, 200_000):
                pg = S.PanGenome(genome_len=genome_len, n_snps=genome_len // 100, n_accessory=6, accessory_len=1500, seed=21, n_contigs=2)
                a, b_ = D.shard_genomes(100, world)[rank]
                batch = c.batch(b_ - a)
                for g in range(a, b_):
                    batch.add_array(g - a, pg.genome(g))
                batch.upload()
                stats = {"bytes": 0, "ms": 0.0, "calls": 0}
                m = D.sharded_step(batch, 31, 1, True, dev, stats=stats)
                rows = D.gather_rows(m.data(), dev)
                out.append((m.kmers().copy(), rows, stats["calls"]))
                m.free()
                batch.free()
        if rank == 0:
            q.put(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_a_dictionary_that_outgrows_the_exchange_layout(ctx):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_growing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [o[2] for o in out] == [1, 2, 1]          # collectives per step: the second step twice, the third fits what the second learnt
    for genome_len, (kmers, rows, _) in zip((20_000, 200_000, 200_000), out):
        pg = synth.PanGenome(genome_len=genome_len, n_snps=genome_len // 100, n_accessory=6, accessory_len=1500, seed=21, n_contigs=2)
        want = orc.pipeline([pg.genome(g).tobytes() for g in range(100)], 31, 1, True, min(os.cpu_count() or 1, 32))[0]
        assert kmers.shape == want["kmers"].shape and (kmers == want["kmers"]).all()
        assert rows.shape == want["matrix"].shape and (rows == want["matrix"]).all()


def _nccl_worker(port, q):
    import torch
    import torch.distributed as dist
    import grm_amd
    D = import_module("genomic-resistance-mapping-grm-_amd.distributed")
    S = import_module("genomic-resistance-mapping-grm-_amd.synth")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        pg = S.PanGenome(genome_len=60_000, n_snps=600, n_accessory=6, accessory_len=1500, seed=21, n_contigs=2)
        with grm_amd.Context(0) as c:
            batch = c.batch(70)
            for g in range(70):
                batch.add_array(g, pg.genome(g))
            batch.upload()
            dev = torch.device("cuda", 0)
            # force the collective code path although there is one rank: sizes over the gloo twin group, the
            # record all-gather over RCCL, the rank union on the gathered payload
            n_local = (batch.partition(31, 1), batch.local_dict())[1]
            payload, n_max, counts, bbs = D.exchange_dict(batch, n_local, dev, None, 1)
            assert counts == [n_local] and n_max >= max(1, n_local) and payload.is_cuda
            batch.set_global_dict_gathered(payload.data_ptr(), n_max, counts, bbs, True)
            m = batch.fill()
            rows = D.gather_rows(m.data(), dev)
            q.put((m.kmers().copy(), rows))
            m.free()
            batch.free()
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank(ctx):
    """the engine and torch's RCCL ("nccl") backend in ONE process on the GPU: same HIP runtime, device
    tensors through all_gather_into_tensor, raw pointers handed to the engine afterwards"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_nccl_worker, args=(port, q))
    p.start()
    kmers, rows = q.get(timeout=240)
    p.join(timeout=120)
    assert p.exitcode == 0
    pg = synth.PanGenome(genome_len=60_000, n_snps=600, n_accessory=6, accessory_len=1500, seed=21, n_contigs=2)
    want = orc.build_matrix([[pg.genome(g).tobytes()] for g in range(70)], 31, 1, True)
    assert (kmers == want["kmers"]).all() and (rows == want["matrix"]).all()


def test_tsv_roundtrip(ctx, tmp_path):
    name, k, genomes = MICRO[2]
    bg = _as_bytes(genomes)
    want = orc.build_matrix(bg, k, 1, False)
    b = ctx.batch(len(bg))
    for g, files in enumerate(bg):
        for f in files:
            b.add(g, f)
    b.upload()
    m = b.run(k, 1, False)
    ids = ["genome_%d" % g for g in range(len(bg))]
    path = str(tmp_path / "KmerMatrix.tsv")
    m.write_tsv(ids, path)
    lines = open(path).read().split("\n")
    assert lines[0] == "kmers\t" + "\t".join(ids)
    body = [l for l in lines[1:] if l]
    assert len(body) == want["kmers"].shape[0]
    assert len({len(l) for l in body}) == 1           # create.py:130-137 needs equal-length rows
    strs = orc.decode_kmers(want["kmers"], k)
    for c, line in enumerate(body):
        cells = line.split("\t")
        assert cells[0] == strs[c]
        for g in range(len(bg)):
            bit = (int(want["matrix"][g // 64, c]) >> (63 - g % 64)) & 1
            assert cells[1 + g] == str(bit)
    m.free()
    b.free()


@pytest.mark.parametrize("k", [33, 47, 63, 64])
def test_two_word_kmers(ctx, k, tmp_path):
    """k > 32 (BASELINE config C5: k = 63): sort-based path vs the oracle"""
    rng = np.random.RandomState(k)
    core = cases.rand_seq(rng, 4000)
    genomes = []
    for g in range(70):
        s = list(core)
        for p in rng.randint(0, len(core), size=6):
            s[p] = "ACGT"[rng.randint(4)]
        recs = [("c", "".join(s[: int(rng.randint(k - 1, len(core)))])), ("n", "".join(s[:200]) + "N" + "".join(s[200:300]))]
        genomes.append([cases.fasta(recs, width=70).encode()])
    genomes[3] = [b""]
    genomes[7].append(cases.fasta([("dup", core[:500] * 2)]).encode())       # repeated k-mers inside one genome
    # default: hash-partition pipeline (abundance-min 1), sort-based path otherwise / when forced
    for opts in ({}, {"wide_sort": 1}, {"bucket_bits": 3, "cap_log2": 7}):
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            for amin, filt in [(1, False), (1, True), (2, False)]:
                want = orc.build_matrix(genomes, k, amin, filt)
                kmers, data, n_occ, colcnt = _run_batch(ctx, genomes, k, amin, filt)
                assert n_occ == want["n_occurrences"]
                assert kmers.shape == want["kmers"].shape and (kmers == want["kmers"]).all()
                assert (data == want["matrix"]).all()
                assert (colcnt == want["n_genomes_with"]).all()
        finally:
            for name in opts:
                ctx.set_option(name, -1)
    for g in (0, 7):
        km, ct, nocc = orc.count_genome(genomes[g], k, 1)
        s = ctx.count_genome(genomes[g], k, 1)
        assert s.occurrences == nocc and s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
        s.free()
    # writers decode two-word k-mers
    if k == 63:
        kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
        want = orc.build_matrix(genomes, k, 1, False)
        b = ctx.batch(len(genomes))
        for g, files in enumerate(genomes):
            for f in files:
                b.add(g, f)
        b.upload()
        m = b.run(k, 1, False)                     # singletons kept, as C5 asks
        ids = ["g%d" % i for i in range(len(genomes))]
        path = str(tmp_path / "c5.kover")
        kd.write_header(path, "contigs", "l", None, None, 5, ids, None, None, None, "nothing")
        m.write_kover_h5(path, 5, 100000)          # gzip 5
        r = kd.KoverDatasetReader(path)
        assert r.kmer_sequences == orc.decode_kmers(want["kmers"], k)
        assert (r.kmer_matrix == want["matrix"]).all()
        assert (r.sum_rows(range(len(genomes))) == want["n_genomes_with"]).all()
        tsv = str(tmp_path / "m.tsv")
        m.write_tsv(ids, tsv)
        assert [l.split("\t")[0] for l in open(tsv).read().split("\n")[1:] if l] == orc.decode_kmers(want["kmers"], k)
        b.partition(k, 2)                          # the staged calls at k > 32 with an abundance filter: the sort path (test_staged_equals_fused)
        assert b.local_dict() == orc.build_matrix(genomes, k, 2, False)["kmers"].shape[0]
        m.free(); b.free()


def test_two_word_kmers_of_few_large_genomes_in_parts(ctx):
    """few genomes: level 1 cuts each into parts (one workgroup per part), the dictionary workgroups go through a genome's parts; runs that
    cross a part boundary, forced part counts, a genome shorter than the others"""
    gs = [synth.random_genome(i, genome_len=300_000 if i else 90_000, seed=5, n_contigs=2).tobytes() for i in range(3)]
    genomes = [[gs[0]], [gs[1]], [gs[2][:200_000] + gs[1][1000:150_000]]]
    for k in (47, 63, 64):
        for opts in ({}, {"rec_part_bits": 3}, {"rec_part_bits": 6}):
            try:
                for name, v in opts.items():
                    ctx.set_option(name, v)
                ctx.timing(True)
                ctx.timing_reset()
                _check(ctx, genomes, k, 1, False)
                assert "superkmer_l1" in {t[0] for t in ctx.timings()}
            finally:
                ctx.timing(False)
                for name in opts:
                    ctx.set_option(name, -1)


@pytest.mark.parametrize("k,opts", [(33, {}), (34, {}), (47, {}), (63, {}), (64, {}), (63, {"bucket_bits": 3, "cap_log2": 9}), (63, {"bucket_bits": 9}),
                                    (47, {"dict_sort_prim": 1})])
def test_two_word_kmers_through_the_record_form(ctx, k, opts):
    """two-word k-mers travel as 24-byte run records (the minimizer among the 21 / 22 m-mers in the middle of the
    k-mer): assemblies with their own contigs on either strand, indels, a repeated stretch and runs of N, against the oracle, with and
    without singletons; small tables (sub-buckets: every workgroup cuts the k-mers out of all its bucket's records) and many buckets;
    "records" 0 is the key form on the same input"""
    n = 131
    pg = synth.realistic(genome_len=20_000, seed=23 + k, contigs=(1, 5), indel_sites=8, n_snps=200, n_accessory=2, accessory_len=400)
    genomes = []
    for i in range(n):
        g = pg.genome(i).tobytes()
        if i % 7 == 0:
            g = g[:1000] + b"N" * (1 + i % 5) + g[1000:]        # (inside a line: bases that are none)
        genomes.append([g + g[: 1500 + 11 * i]])
    genomes[5] = [b""]
    genomes[9] = [b">tiny\n" + b"ACGT" * 10 + b"\n"]              # shorter than k
    want_names = {"superkmer_l1", "superkmer_l2", "wh_dict_build"}
    try:
        for name, v in opts.items():
            ctx.set_option(name, v)
        ctx.timing(True)
        ctx.timing_reset()
        _check(ctx, genomes, k, 1, False)
        names = {t[0] for t in ctx.timings()}
        assert want_names <= names and "wh_scatter_l1" not in names, names
        _check(ctx, genomes, k, 1, True)
        ctx.set_option("records", 0)
        ctx.timing_reset()
        _check(ctx, genomes, k, 1, True)
        assert "wh_scatter_l1" in {t[0] for t in ctx.timings()}
    finally:
        ctx.timing(False)
        ctx.set_option("records", -1)
        for name in opts:
            ctx.set_option(name, -1)


@pytest.mark.parametrize("k", [33, 63])
def test_two_word_sets_merge(ctx, k):
    """dsk2kover's job at k > 32: per-genome counted sets (abundance-min 2 applied per genome) ->
    grm_build_matrix, incl. sets that travelled through host arrays"""
    rng = np.random.RandomState(100 + k)
    core = cases.rand_seq(rng, 3000)
    genomes = []
    for g in range(67):
        s = list(core)
        for p in rng.randint(0, len(core), size=5):
            s[p] = "ACGT"[rng.randint(4)]
        t = "".join(s)
        genomes.append([cases.fasta([("a", t), ("b", t[: int(rng.randint(0, 2500))])], width=60).encode()])
    genomes[5] = [b">e\n"]
    for amin, filt in [(1, True), (2, False), (2, True)]:
        want = orc.build_matrix(genomes, k, amin, filt)
        sets = [ctx.count_genome(g, k, amin) for g in genomes]
        # half of them re-created from host arrays, as dsk2kover does after reading multidsk's files
        sets = [ctx.kmer_set_from_arrays(s.kmers(), s.counts(), k) if i % 2 else s for i, s in enumerate(sets)]
        m = ctx.build_matrix(sets, filt)
        assert m.kmers().shape == want["kmers"].shape and (m.kmers() == want["kmers"]).all()
        assert (m.data() == want["matrix"]).all()
        assert (m.column_counts() == want["n_genomes_with"]).all()
        m.free()
    with pytest.raises(grm.GrmError):
        ctx.build_matrix([ctx.count_genome(genomes[0], 31, 1), ctx.count_genome(genomes[1], k, 1)], False)


def test_two_word_large_dictionaries(ctx):
    """more than 2^19 two-word dictionary entries / set entries: the helper kernels that split and
    join (hi, lo) pairs run on a capped grid and must stride over the rest"""
    import torch
    k = 41
    genomes = [[synth.random_genome(i, genome_len=n, seed=77).tobytes()] for i, n in enumerate((600_000, 200_000, 200_000))]
    genomes.append(genomes[1])                                   # one shared genome: flags 2
    want = orc.build_matrix(genomes, k, 1, True)
    allm = orc.build_matrix(genomes, k, 1, False)
    assert allm["kmers"].shape[0] > (1 << 19)
    # staged API: export (join) -> set_global_dict (split)
    b = ctx.batch(len(genomes))
    for g, files in enumerate(genomes):
        b.add(g, files[0])
    b.upload()
    b.partition(k, 1)
    n_local = b.local_dict()
    assert n_local == allm["kmers"].shape[0]
    keys = torch.empty((n_local, 2), dtype=torch.int64, device="cuda:0")
    flags = torch.empty(n_local, dtype=torch.uint8, device="cuda:0")
    b.export_dict(keys.data_ptr(), flags.data_ptr())
    torch.cuda.synchronize()
    k_host = keys.cpu().numpy().view(np.uint64)
    order = np.lexsort((k_host[:, 1], k_host[:, 0]))
    assert (k_host[order] == allm["kmers"]).all()
    assert b.set_global_dict(keys.data_ptr(), flags.data_ptr(), n_local, True) == want["kmers"].shape[0]
    m = b.fill()
    assert (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
    m.free(); b.free()
    # merge of device-resident counted sets (split on the device)
    sets = [ctx.count_genome(g, k, 1) for g in genomes]
    assert max(len(s_) for s_ in sets) > (1 << 19)
    m = ctx.build_matrix(sets, True)
    assert (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
    m.free()


def test_two_word_medium_both_paths(ctx):
    """300 kbp genomes at k = 63: LDS (span-uniform) paths of the two-word hash pipeline vs the sort path vs oracle"""
    genomes = _medium_genomes(n=6, length=200_000, seed=13)
    want = orc.build_matrix(genomes, 63, 1, False)
    for opts in ({}, {"wide_sort": 1}, {"bucket_bits": 13}, {"sub_bits": 2}):
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            kmers, data, n_occ, colcnt = _run_batch(ctx, genomes, 63, 1, False)
            assert n_occ == want["n_occurrences"]
            assert kmers.shape == want["kmers"].shape and (kmers == want["kmers"]).all()
            assert (data == want["matrix"]).all()
        finally:
            for name in opts:
                ctx.set_option(name, -1)


def test_upload_sources_and_slabs(ctx, tmp_path):
    """the host image: files read at upload time straight into the pinned slabs (plain), inflated
    (.gz) or copied (memory buffers), mixed inside one genome, with slabs far smaller than a file so
    that files straddle slab and thread boundaries"""
    import gzip
    genomes = _medium_genomes(n=5, length=150_000, seed=31)
    fq = cases.fastq([cases.rand_seq(np.random.RandomState(i), 120) for i in range(300)]).encode()
    genomes[1] = [genomes[1][0], fq]
    genomes[3] = [b"", genomes[3][0]]
    want = orc.build_matrix(genomes, 25, 1, False)
    for slab_kb in (16, 100, -1):
        ctx.set_option("upload_slab_kb", slab_kb)
        try:
            b = ctx.batch(len(genomes))
            for g, files in enumerate(genomes):
                for n, f in enumerate(files):
                    path = str(tmp_path / ("g%d_%d_%d" % (g, n, slab_kb)))
                    if (g + n) % 3 == 0:
                        open(path, "wb").write(f)
                        b.add_file(g, path)
                    elif (g + n) % 3 == 1:
                        open(path + ".gz", "wb").write(gzip.compress(f))
                        b.add_file(g, path + ".gz")
                    else:
                        b.add(g, f)
            b.upload()
            m = b.run(25, 1, False)
            assert b.n_occurrences == want["n_occurrences"]
            assert (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
            m.free(); b.free()
        finally:
            ctx.set_option("upload_slab_kb", -1)
    b = ctx.batch(1)
    path = str(tmp_path / "vanishing.fna")
    open(path, "wb").write(genomes[0][0])
    b.add_file(0, path)
    open(path, "wb").write(b">x\nACGT\n")           # shrinks between add and upload
    with pytest.raises(grm.GrmError):
        b.upload()
    b.free()
    with pytest.raises(grm.GrmError):
        ctx.batch(1).add_file(0, str(tmp_path / "missing.fna"))


def test_errors_are_loud(ctx):
    b = ctx.batch(1)
    b.add(0, b">x\nACGT\n")
    with pytest.raises(grm.GrmError):
        b.partition(31, 1)          # before upload
    b.upload()
    with pytest.raises(grm.GrmError):
        b.run(0, 1, False)
    with pytest.raises(grm.GrmError) as e:
        b.run(129, 1, False)        # the reference's range ends at 128 (bin/kover/kover:114)
    assert e.value.code == -1
    with pytest.raises(grm.GrmError) as e:
        b.partition(129, 1)         # the staged calls: the same range
    assert e.value.code == -1
    b.partition(65, 1)              # (k > 64 in stages: the sort path; a 4-base input holds no 65-mer)
    assert b.local_dict() == 0
    b.free()


def test_scale_properties(ctx):
    """size-independent properties on a larger batch (no oracle at this size):
    sorted dictionary, carrier counts, singleton filter = columns with >= 2 carriers,
    a shared core => all-genome columns, determinism."""
    n, length = 96, 400_000
    pg = synth.PanGenome(genome_len=length, n_snps=4000, n_accessory=20, accessory_len=3000, seed=77)
    b = ctx.batch(n)
    for g in range(n):
        b.add_array(g, pg.genome(g))
    b.upload()
    m_all = b.run(31, 1, False)
    k_all, d_all, cc_all = m_all.kmers()[:, 0], m_all.data(), m_all.column_counts()
    assert (np.diff(k_all.astype(object)) > 0).all()
    m_f = b.run(31, 1, True)
    k_f, d_f, cc_f = m_f.kmers()[:, 0], m_f.data(), m_f.column_counts()
    keep = cc_all >= 2
    assert (k_f == k_all[keep]).all()
    assert (d_f == d_all[:, keep]).all()
    assert (cc_f == cc_all[keep]).all()
    assert cc_all.max() == n and (cc_all >= 1).all()
    # sum_rows (the learner's masked popcount) on the GPU == numpy on the host copy
    some = [0, 5, 63, 64, 70, 95]
    dense = np.array([((d_all[g // 64] >> np.uint64(63 - g % 64)) & np.uint64(1)) for g in some]).sum(axis=0)
    assert (m_all.sum_rows(some) == dense).all()
    assert (m_all.sum_rows(range(n)) == cc_all).all()
    # padding bits of the last word-row are zero
    pad = (1 << (64 - (n - 64))) - 1 if n % 64 else 0
    assert not (d_all[-1] & np.uint64(pad)).any()
    # per-genome distinct count from the matrix == an independent per-genome count on the GPU
    for g in (0, 37, 95):
        s = ctx.count_genome([pg.genome(g).tobytes()], 31, 1)
        col = (d_all[g // 64] >> np.uint64(63 - g % 64)) & np.uint64(1)
        assert int(col.sum()) == len(s)
        assert (k_all[col.astype(bool)] == s.kmers()[:, 0]).all()
        s.free()
    # two genomes against the CPU oracle
    sub = orc.build_matrix([[pg.genome(g).tobytes()] for g in (3, 4)], 31, 1, False)
    both = ((d_all[0] >> np.uint64(63 - 3)) & np.uint64(1)).astype(bool) | ((d_all[0] >> np.uint64(63 - 4)) & np.uint64(1)).astype(bool)
    assert (k_all[both] == sub["kmers"][:, 0]).all()
    m_all.free(); m_f.free(); b.free()


@pytest.mark.parametrize("k", [31, 63])
def test_all_paths_agree_at_scale(ctx, k, tmp_path):
    """120 x 1.5 Mbp pan-genome: large enough for millions of dictionary entries and multi-tile regions,
    small enough for the threaded oracle.  The fused pass must equal the oracle; every other route to
    the matrix (staged API, counted sets -> merge, forced sub-buckets, probing fill / sort-based
    two-word path) must equal the fused pass."""
    import torch
    n = 120
    pg = synth.PanGenome(genome_len=1_500_000, n_snps=15000, n_accessory=60, accessory_len=5000, seed=5)
    arrays = [pg.genome(g) for g in range(n)]
    want, _, _, occ = orc.pipeline([a.tobytes() for a in arrays], k, 1, True, min(os.cpu_count() or 1, 32))
    w = 2 if k > 32 else 1

    def batch():
        b = ctx.batch(n)
        for g in range(n):
            b.add_array(g, arrays[g])
        b.upload()
        return b

    def same(m, what):
        assert m.kmers().shape == want["kmers"].shape, what
        assert (m.kmers() == want["kmers"]).all(), what
        assert (m.data() == want["matrix"]).all(), what
        m.free()

    b = batch()
    m = b.run(k, 1, True)
    assert b.n_occurrences == occ
    assert want["kmers"].shape[0] > 1_000_000
    same(m, "fused")
    # staged API on one rank
    b.partition(k, 1)
    n_local = b.local_dict()
    keys = torch.empty((max(1, n_local), w), dtype=torch.int64, device="cuda:0")
    flags = torch.empty(max(1, n_local), dtype=torch.uint8, device="cuda:0")
    b.export_dict(keys.data_ptr(), flags.data_ptr())
    torch.cuda.synchronize()
    b.set_global_dict(keys.data_ptr(), flags.data_ptr(), n_local, True)
    same(b.fill(), "staged")
    # forced geometry / alternative kernels
    alts = [{"sub_bits": 1}, {"bucket_bits": 12}] + ([{"no_slots": 1}] if k <= 32 else [{"wide_sort": 1}])
    for opts in alts:
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            same(b.run(k, 1, True), str(opts))
        finally:
            for name in opts:
                ctx.set_option(name, -1)
    b.free()
    # multidsk -> dsk2kover: counted sets (device-resident), merged
    if k <= 32:
        b = batch()
        b.partition_counts(k, 1)
        sets = [b.genome_set(g) for g in range(n)]
        b.free()
    else:
        sets = [ctx.count_genome([arrays[g].tobytes()], k, 1) for g in range(0, n)]
    same(ctx.build_matrix(sets, True), "sets -> merge")
    for s_ in sets:
        s_.free()


def test_full_size_properties(ctx):
    """BASELINE.json configs[1] at full size (1000 x 5 Mbp, k=31, singleton filter): no oracle can
    run this, so the result is checked through size-independent properties and a 2-genome oracle slice"""
    n = 1000
    pg = synth.PanGenome(genome_len=5_000_000, seed=1234)
    b = ctx.batch(n)
    for g in range(n):
        b.add_array(g, pg.genome(g))
    b.upload()
    m = b.run(31, 1, True)
    kmers = m.kmers()[:, 0]
    cc = m.column_counts()
    assert b.n_occurrences > 4.9e9
    assert (np.diff(kmers.astype(np.int64)) > 0).all()          # 62-bit values: strictly ascending dictionary
    assert cc.min() >= 2 and cc.max() == n                        # singleton filter; the conserved core is in every genome
    # a second pass over the same resident inputs gives the identical result (the partition order inside
    # buckets is not deterministic, the output must be)
    m2 = b.run(31, 1, True)
    assert (m2.kmers()[:, 0] == kmers).all() and (m2.column_counts() == cc).all()
    # masked popcount of two genomes == their oracle sets restricted to the kept columns
    for g in (17, 700):
        km, ct, nocc = orc.count_genome([pg.genome(g).tobytes()], 31, 1)
        sel = m.sum_rows([g]).astype(bool)
        assert np.isin(kmers[sel], km[:, 0]).all()                # every set bit is a real k-mer of the genome
        kept = np.isin(km[:, 0], kmers)                           # its k-mers that survived the filter ...
        assert int(sel.sum()) == int(kept.sum())                  # ... are exactly the set bits
    m.free(); m2.free(); b.free()


# ---- low-sharing inputs (independent random genomes: every k-mer its own column) ----------------
def _dict_build_launches(ctx):
    return sum(1 for name, _, _ in ctx.timings() if name == "dict_build")


def test_low_sharing_sizes_the_dictionary_in_two_launches(ctx):
    """mode R: the union of a bucket over the genomes is ~n_genomes times one genome's share, far beyond one
    LDS table.  The failed first launch reports what it would have needed and the retry jumps there (no walk up
    one sub-bucket bit at a time); the batch remembers it, so a later pass needs ONE launch.  Bit-exact either way."""
    n, L, k = 40, 250_000, 31
    genomes = [[synth.random_genome(i, genome_len=L, seed=4242).tobytes()] for i in range(n)]
    want = orc.build_matrix(genomes, k, 1, False)
    b = ctx.batch(n)
    for g, files in enumerate(genomes):
        b.add(g, files[0])
    b.upload()
    ctx.timing(True)
    try:
        for expect_max in (2, 1):
            ctx.timing_reset()
            m = b.run(k, 1, False)
            launches = _dict_build_launches(ctx)
            assert 1 <= launches <= expect_max, "dict_build launched %d times" % launches
            assert m.n_kmers == want["kmers"].shape[0]
            assert (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
            m.free()
        # every k-mer of independent genomes is a singleton: the filter leaves nothing (up to chance collisions)
        m = b.run(k, 1, True)
        want_f = orc.build_matrix(genomes, k, 1, True)
        assert m.n_kmers == want_f["kmers"].shape[0] and (m.data() == want_f["matrix"]).all()
        m.free()
    finally:
        ctx.timing(False)
        b.free()


def test_pan_genome_needs_one_dictionary_launch(ctx):
    """mode P stays on the single-launch path (the sizing ladder must not cost the common case anything)"""
    genomes = _medium_genomes(n=70, length=200_000, seed=3)
    b = ctx.batch(len(genomes))
    for g, files in enumerate(genomes):
        b.add(g, files[0])
    b.upload()
    ctx.timing(True)
    try:
        ctx.timing_reset()
        m = b.run(31, 1, True)
        assert _dict_build_launches(ctx) == 1
        want = orc.build_matrix(genomes, 31, 1, True)
        assert (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
        m.free()
    finally:
        ctx.timing(False)
        b.free()


def test_counting_stage_on_random_genomes(ctx):
    """the counting stage bench.py times on mode R (parse -> partition -> per-genome dedup + count): every
    genome's counted set equals the oracle's, k-mer for k-mer and count for count"""
    n, L, k = 6, 400_000, 31
    genomes = [synth.random_genome(i, genome_len=L, seed=99, n_contigs=1 + i % 3).tobytes() for i in range(n)]
    b = ctx.batch(n)
    for g, f in enumerate(genomes):
        b.add(g, f)
    b.upload()
    b.partition_counts(k, 1)
    total = 0
    for g in range(n):
        km, ct, nocc = orc.count_genome([genomes[g]], k, 1)
        s = b.genome_set(g)
        assert s.occurrences == nocc
        assert s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
        total += nocc
        s.free()
    assert b.n_occurrences == total
    b.free()


@pytest.mark.parametrize("k,opts,by_records", [(31, {}, True), (21, {}, True), (32, {}, True), (31, {"rec_count": 0}, False), (25, {"bucket_bits": 7}, True),
                                               (31, {"rec_count_cap": 8}, True), (31, {"bucket_bits": 5, "rec_count_cap": 8}, False),
                                               (31, {"bucket_bits": 4}, False)])
def test_counting_stage_through_the_record_form(ctx, k, opts, by_records):
    """from 128 genomes on a counting partition moves records (level 1), expands them to key segments by minimizer bucket (level 2)
    and counts there: counted sets with a repeated stretch and contigs on either strand equal the oracle's, with and without the
    abundance filter; "rec_count" 0 is the key form on the same input; the matrix of the filtered sets too.  "rec_count_cap" 8: wave
    tables of 256 slots, so that segments of more than 224 k-mers take the workgroup's shared table (1024 slots) -- and with 2^5 buckets
    (~1000 distinct k-mers per segment) overflow it, which sends the batch to the key form; 2^4 buckets: segments too large to try"""
    n = 130
    pg = synth.realistic(genome_len=30_000, seed=17, contigs=(1, 4), indel_sites=10, n_snps=300, n_accessory=3, accessory_len=500)
    genomes = []
    for i in range(n):
        g = pg.genome(i).tobytes()
        genomes.append(g + g[: 4000 + 37 * i])                  # a repeated stretch (cut inside a line): counts above 1
    check = (0, 1, 63, 64, 127, n - 1)
    try:
        for name, v in opts.items():
            ctx.set_option(name, v)
        for amin in (1, 2):
            b = ctx.batch(n)
            for g, f in enumerate(genomes):
                b.add(g, f)
            b.upload()
            b.partition_counts(k, amin)
            assert (b.bucket_bits & 0x100 != 0) == by_records
            for g in check:
                km, ct, nocc = orc.count_genome([genomes[g]], k, amin)
                s = b.genome_set(g)
                assert s.occurrences == nocc
                assert s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
                s.free()
            b.free()
        _check(ctx, [[g] for g in genomes], k, 2, True)
    finally:
        for name in opts:
            ctx.set_option(name, -1)


def test_counting_stage_with_heavily_repeated_kmers(ctx):
    """a k-mer that one genome holds 6000 times (a homopolymer, a short tandem repeat): its segment is far beyond what a wave's one-word
    table slots take (448 k-mers, counts of 12 bits), so it is counted in the workgroup's 64-bit table -- or its region overflows and the
    batch goes to the key form; either way the counted sets equal the oracle's, count for count"""
    n, k = 130, 31
    rng = np.random.RandomState(3)
    genomes = []
    for i in range(n):
        body = cases.rand_seq(rng, 3000 if i % 2 else 400_000)
        if i == 0:
            body = "A" * 6030 + body + "ACGT" * 1600                 # A^31 6000 times; the four rotations of (ACGT)^n ~1590 times each
        if i == 77:
            body = body + "N" + "TG" * 5000                          # (TG)^n / (GT)^n: ~4985 times each, canonical forms of two k-mers
        genomes.append((">g%d\n%s\n" % (i, body)).encode())
    for amin in (1, 3):
        b = ctx.batch(n)
        for g, f in enumerate(genomes):
            b.add(g, f)
        b.upload()
        b.partition_counts(k, amin)
        for g in (0, 1, 77, n - 1):
            km, ct, nocc = orc.count_genome([genomes[g]], k, amin)
            s_ = b.genome_set(g)
            assert s_.occurrences == nocc
            assert s_.kmers().shape == km.shape and (s_.kmers() == km).all() and (s_.counts() == ct).all()
            if g == 0 and amin == 1:
                assert ct.max() >= 6000
            s_.free()
        b.free()


# ---- the reference-held fixture through the HIP path -------------------------------------------
def test_reference_31mers_through_the_engine(ctx, golden_dir):
    """the 196 canonical 31-mers the reference ships (page/results/**, output of the real DSK pipeline) as
    FASTA records, forward in one genome and reverse-complemented in another: the engine must produce exactly
    those 196 columns, in ascending GATB order, with both strands landing in the same column."""
    kmers = [l.strip() for l in open(os.path.join(golden_dir, "canonical_31mers.txt")) if l.strip() and not l.startswith("#")]
    assert len(kmers) == 196
    fwd = "".join(">f%d\n%s\n" % (i, km) for i, km in enumerate(kmers)).encode()
    rev = "".join(">r%d\n%s\n" % (i, cases.revcomp(km)) for i, km in enumerate(kmers)).encode()
    half = "".join(">h%d\n%s\n" % (i, km) for i, km in enumerate(kmers[::2])).encode()
    b = ctx.batch(3)
    b.add(0, fwd); b.add(1, rev); b.add(2, half)
    b.upload()
    m = b.run(31, 1, False)
    got = grm.decode_kmers(m.kmers()[:, 0], 31)
    order = {c: i for i, c in enumerate("ACTG")}
    assert got == sorted(kmers, key=lambda s: [order[c] for c in s])       # exactly those, ascending under A<C<T<G
    data = m.data()
    col = {km: i for i, km in enumerate(got)}
    for j, km in enumerate(kmers):
        w = int(data[0, col[km]])
        assert (w >> 63) & 1 and (w >> 62) & 1                              # forward and reverse strand: same column
        assert ((w >> 61) & 1) == (1 if j % 2 == 0 else 0)
    assert b.n_occurrences == 196 * 2 + 98
    m.free()
    # and as counted sets (multidsk's output): every k-mer once per genome
    s = ctx.count_genome([rev], 31, 1)
    assert grm.decode_kmers(s.kmers()[:, 0], 31) == got and (s.counts() == 1).all()
    s.free()
    b.free()


@pytest.mark.parametrize("opts", [{}, {"bucket_bits": 7}, {"bucket_bits": 6}, {"bucket_bits": 5, "cap_log2": 13}, {"dedup_wg": 1}])
def test_counted_sets_through_both_dedup_forms(ctx, opts):
    """per-segment dedup + count: the wave form (table sized to the segment), segments too dense for it handed to
    the workgroup form (bucket_bits 6: ~1900 distinct per segment > 7/8 of 2048; bucket_bits 5: ~3700 > 7/8 of 4096
    needs the 8192-slot table), and the workgroup form alone -- all equal to the oracle's counted sets"""
    genomes = [g[0] for g in _medium_genomes(n=4, length=120_000, seed=21)]
    try:
        for name, v in opts.items():
            ctx.set_option(name, v)
        for amin in (1, 2):
            b = ctx.batch(len(genomes))
            for g, f in enumerate(genomes):
                b.add(g, f + b"\n" + f[:30_000])          # a repeated stretch: counts above 1
            b.upload()
            b.partition_counts(31, amin)
            for g in range(len(genomes)):
                km, ct, nocc = orc.count_genome([genomes[g] + b"\n" + genomes[g][:30_000]], 31, amin)
                s = b.genome_set(g)
                assert s.occurrences == nocc
                assert s.kmers().shape == km.shape and (s.kmers() == km).all() and (s.counts() == ct).all()
                s.free()
            b.free()
    finally:
        for name in opts:
            ctx.set_option(name, -1)


@pytest.mark.parametrize("tracked", [False, True])
@pytest.mark.parametrize("opts", [{}, {"no_union": 1}, {"cap_log2": 8}])
@pytest.mark.parametrize("k", [31, 47])
def test_gathered_exchange_in_one_process(ctx, k, opts, tracked):
    """the exchange records of three "ranks" (three batches of one context) laid out as an all-gather would leave
    them; every batch builds the global dictionary from the payload (rank union in LDS tables, or the sort of
    everything with no_union / k > 32; cap_log2 8 forces the union through its sizing ladder) and fills its rows:
    stacked, they are the oracle's matrix.  One rank uses another bucket count in a second round: its finer or coarser
    buckets nest in the others', the union runs over the coarsest.  tracked: every batch says which record of the payload is its
    own, and its entries take their columns from the union's sort (the union notes where each of them fell) instead of a search."""
    import torch
    dev = torch.device("cuda", 0)
    pg = synth.PanGenome(genome_len=150_000, n_snps=1500, n_accessory=8, accessory_len=1500, seed=77, n_contigs=2)
    shards = [(0, 64), (64, 128), (128, 150)]
    genomes = [pg.genome(g).tobytes() for g in range(150)]
    words = 2 if k > 32 else 1
    for filt in (True, False):
        want = orc.pipeline(genomes, k, 1, filt, min(os.cpu_count() or 1, 16))[0]
        for odd_geometry in (False, True):
            if odd_geometry and (k > 32 or opts):
                continue
            batches, n_locals, bbs = [], [], []
            try:
                for name, v in opts.items():
                    ctx.set_option(name, v)
                for r, (a, b_) in enumerate(shards):
                    if odd_geometry and r == 1:
                        ctx.set_option("bucket_bits", 6)
                    b = ctx.batch(b_ - a)
                    for g in range(a, b_):
                        b.add(g - a, genomes[g])
                    b.upload()
                    b.partition(k, 1)
                    ctx.set_option("bucket_bits", -1)
                    n_locals.append(b.local_dict())
                    bbs.append(b.bucket_bits)
                    batches.append(b)
                if not opts:
                    assert (len(set(bbs)) > 1) == odd_geometry      # (small tables: a rank may have taken more buckets)
                n_max = max(1, max(n_locals))
                flags_off, boff_off, stride = batches[0].exchange_layout(n_max, words, max(v & 0xff for v in bbs))
                payload = torch.empty(len(shards) * stride, dtype=torch.uint8, device=dev)
                for r, b in enumerate(batches):
                    b.export_dict_ordered(payload.data_ptr() + r * stride, flags_off, boff_off)
                # the bucket offsets of a record partition its entries
                boff = payload[boff_off: boff_off + 4 * ((1 << (bbs[0] & 0xff)) + 1)].cpu().numpy().view(np.uint32)
                assert boff[0] == 0 and boff[-1] == n_locals[0] and (np.diff(boff.astype(np.int64)) >= 0).all()
                rows = []
                for r, b in enumerate(batches):
                    u = b.set_global_dict_gathered(payload.data_ptr(), n_max, n_locals, bbs, filt, my_rank=r if tracked else -1)
                    assert u == want["kmers"].shape[0]
                    m = b.fill()
                    assert (m.kmers().reshape(-1) == want["kmers"].reshape(-1)).all()
                    rows.append(m.data())
                    m.free()
                got = np.concatenate(rows)
                assert got.shape == want["matrix"].shape and (got == want["matrix"]).all()
            finally:
                for name in opts:
                    ctx.set_option(name, -1)
                ctx.set_option("bucket_bits", -1)
                for b in batches:
                    b.free()


def test_sum_rows_against_reference_popcount_vectors(ctx, golden_dir):
    """grm_matrix_sum_rows on the device vs vectors produced by the reference's compiled popcount.pyx and its
    build_row_mask (tests/golden/make_golden.py): learning/common/rules.py:201-267"""
    import json
    d = json.load(open(os.path.join(golden_dir, "popcount_vectors.json")))
    for c in d["cases"]:
        block = np.array([[int(x) for x in row] for row in c["block"]], dtype=np.uint64)
        hm = grm.HostMatrix(np.zeros(block.shape[1], dtype=np.uint64), block, c["n_genomes"], 31)
        m = hm.to_device(ctx)
        assert (m.sum_rows(c["selected"]) == np.array(c["column_sums"], dtype=np.uint32)).all()
        # all-ones mask = carrier counts
        assert (m.column_counts() == grm.engine.np.array([sum(bin(int(x)).count("1") for x in block[:, j]) for j in range(block.shape[1])])).all()
        m.free()


@pytest.mark.parametrize("k", [65, 95, 96, 97, 127, 128])
def test_three_and_four_word_kmers(ctx, k, tmp_path):
    """65 <= k <= 128 (three / four 64-bit words per k-mer; the reference's --kmer-size goes up to 128): fused run,
    counted sets, dsk2kover's merge of those sets, TSV and Kover HDF5 writers -- all against the oracle"""
    kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
    rng = np.random.RandomState(k)
    core = cases.rand_seq(rng, 3000)
    genomes = []
    for g in range(5):
        s_ = list(core)
        for p in rng.randint(0, len(core), size=12):
            s_[p] = "ACGT"[rng.randint(4)]
        s_ = "".join(s_)
        recs = [("c1", s_[:1400]), ("c2", s_[1400:] + "N" + cases.rand_seq(rng, 200)), ("short", s_[:k - 1]), ("rc", cases.revcomp(s_[200:900]))]
        genomes.append([cases.fasta(recs, width=73).encode(), cases.fasta([("extra", s_[500:800].lower())]).encode()])
    words = (k + 31) // 32
    for amin, filt in [(1, False), (1, True), (2, False)]:
        want = orc.build_matrix(genomes, k, amin, filt)
        kmers, data, n_occ, colcnt = _run_batch(ctx, genomes, k, amin, filt)
        assert n_occ == want["n_occurrences"]
        assert kmers.shape == want["kmers"].shape == (want["kmers"].shape[0], words) and (kmers == want["kmers"]).all()
        assert (data == want["matrix"]).all() and (colcnt == want["n_genomes_with"]).all()
    # multidsk's sets, then dsk2kover's merge (device-resident sets and sets rebuilt from host arrays)
    sets = []
    for files in genomes:
        km, ct, nocc = orc.count_genome(files, k, 1)
        s_ = ctx.count_genome(files, k, 1)
        assert s_.occurrences == nocc and s_.kmers().shape == km.shape and (s_.kmers() == km).all() and (s_.counts() == ct).all()
        sets.append(s_)
    want = orc.build_matrix(genomes, k, 1, True)
    m = ctx.build_matrix(sets, True)
    assert (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
    sets2 = [ctx.kmer_set_from_arrays(s_.kmers(), s_.counts(), k) for s_ in sets]
    m2 = ctx.build_matrix(sets2, True)
    assert (m2.kmers() == want["kmers"]).all() and (m2.data() == want["matrix"]).all()
    # writers
    ids = ["g%d" % i for i in range(len(genomes))]
    tsv = str(tmp_path / "m.tsv")
    m.write_tsv(ids, tsv)
    lines = open(tsv).read().split("\n")
    strs = orc.decode_kmers(want["kmers"], k)
    assert lines[0].split("\t") == ["kmers"] + ids and [l.split("\t")[0] for l in lines[1:] if l] == strs
    h5p = str(tmp_path / "d.kover")
    kd.write_header(h5p, "contigs", "l", None, None, 5, ids, None, None, None, "singleton")
    m.write_kover_h5(h5p, 5, 100000)
    r = kd.KoverDatasetReader(h5p)
    assert r.kmer_sequences == strs and (r.kmer_matrix == want["matrix"]).all()
    for o in sets + sets2 + [m, m2]:
        o.free()


def test_many_64mers_sharing_their_first_32_bases(ctx):
    """two-word dictionaries are sorted by the top 64 bits first and ties are put in order by one thread per group: a group longer
    than the kernel takes (repeats that share their first 32 bases at k = 64) must fall back to the radix sort, not crawl or misorder"""
    rng = np.random.RandomState(64)
    prefix = "A" * 20 + cases.rand_seq(rng, 12)
    recs = [("c%d" % i, prefix + cases.rand_seq(rng, 32)) for i in range(150)]
    recs += [("r%d" % i, cases.rand_seq(rng, 200)) for i in range(20)]
    genomes = [[cases.fasta(recs[: 100 + 10 * g], width=70).encode()] for g in range(5)]
    want = orc.build_matrix(genomes, 64, 1, False)
    b = ctx.batch(len(genomes))
    for g, files in enumerate(genomes):
        b.add(g, files[0])
    b.upload()
    m = b.run(64, 1, False)
    hi = m.kmers()[:, 0]
    assert np.max(np.unique(hi, return_counts=True)[1]) > 32          # the case is what it claims to be
    assert m.kmers().shape == want["kmers"].shape and (m.kmers() == want["kmers"]).all() and (m.data() == want["matrix"]).all()
    m.free()
    b.free()
