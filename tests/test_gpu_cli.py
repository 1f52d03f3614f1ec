"""GPU: the drop-in executables end to end (argv surface of the reference) vs the oracle."""
import os
import subprocess
import sys
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle_ctypes as orc
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "genomic-resistance-mapping-grm-_amd"
CLI = os.path.join(ROOT, PKG, "cli")


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    return r


@pytest.fixture(scope="module")
def genomes(tmp_path_factory):
    d = tmp_path_factory.mktemp("fna")
    rng = np.random.RandomState(8)
    core = cases.rand_seq(rng, 5000)
    out = []
    for g in range(9):
        s = list(core)
        for p in rng.randint(0, len(core), size=20):
            s[p] = "ACGT"[rng.randint(4)]
        recs = [("c1", "".join(s[:3000])), ("c2", "".join(s[3000:]) + "NN" + cases.rand_seq(rng, 300))]
        path = str(d / ("562.%d.fna" % (100 + g)))
        open(path, "w").write(cases.fasta(recs, width=70))
        out.append(path)
    return out


@pytest.mark.parametrize("k,per_genome_sets", [(31, False), (47, False), (31, True), (47, True)])
def test_multidsk_then_dsk2kover(genomes, tmp_path, k, per_genome_sets):
    """the pair as Kover drives it.  Default: multidsk leaves ONE combined artefact + a reference file
    under every name Kover expects and dsk2kover only selects / filters / writes; GRM_MULTIDSK_SETS=1:
    one counted set per genome (k = 47: two-word k-mers on the sort-based path) merged by dsk2kover."""
    import grm_amd  # noqa: F401
    kd = import_module(PKG + ".kover_dataset")
    env = {"GRM_MULTIDSK_SETS": "1"} if per_genome_sets else {}
    tmp = str(tmp_path)
    lst = os.path.join(tmp, "list_contigs_files")
    open(lst, "w").writelines(p + "\n" for p in genomes)
    # exactly kmer_count.py:28-37
    _run([os.path.join(CLI, "multidsk"), "-file", lst, "-out-dir", tmp, "-kmer-size", str(k), "-abundance-min", "1",
          "-out-compress", "4", "-nb-cores", "0", "-out-tmp", tmp, "-verbose", "0", "-progress", "True"], env=env)
    h5s = [os.path.join(tmp, os.path.basename(os.path.splitext(p + "\n")[0]) + ".h5") for p in genomes]   # create.py:375
    assert all(os.path.exists(p) for p in h5s)
    if not per_genome_sets:
        assert max(os.path.getsize(p) for p in h5s) < 4096           # references, not data
    list_h5 = os.path.join(tmp, "list_h5")
    out = os.path.join(tmp, "DATASET.kover")
    ids = [os.path.basename(p)[:-4] for p in genomes]
    orders = [list(range(len(genomes))), [4, 0, 7, 2]]               # as listed by Kover; a permuted subset
    for filt in ("singleton", "nothing"):
        for order in orders:
            open(list_h5, "w").writelines(h5s[i] + "\n" for i in order)
            kd.write_header(out, "contigs", lst, None, None, 4, [ids[i] for i in order], None, None, None, filt)
            # exactly kmer_pack.py:28-36
            _run([os.path.join(CLI, "dsk2kover"), "-file", list_h5, "-out", out, "-filter", filt, "-kmer-length", str(k),
                  "-compression", "4", "-chunk-size", "100000", "-nb-genomes", str(len(order)), "-verbose", "True"])
            want = orc.build_matrix([[open(genomes[i], "rb").read()] for i in order], k, 1, filt == "singleton")
            r = kd.KoverDatasetReader(out)
            assert r.kmer_sequences == orc.decode_kmers(want["kmers"], k)
            assert (r.kmer_matrix == want["matrix"]).all()
            assert r.genome_identifiers == [ids[i] for i in order]


def test_ray_under_fake_mpiexec(genomes, tmp_path):
    conf = str(tmp_path / "survey.conf")
    outdir = str(tmp_path / "survey.res")
    with open(conf, "w") as f:      # src/app.py:3820-3833
        f.write("-k 21\n-run-surveyor\n-output %s\n-write-kmer-matrix\n" % outdir)
        for p in genomes[:5]:
            f.write("-read-sample-assembly %s %s\n" % (os.path.basename(p)[:-4], p))
    for rank in (3, 1, 0, 2):       # mpiexec -n 4 starts four copies; only rank 0 works
        _run([os.path.join(CLI, "Ray"), conf], env={"PMI_RANK": str(rank)})
    tsv = os.path.join(outdir, "Surveyor", "KmerMatrix.tsv")
    lines = open(tsv).read().split("\n")
    want = orc.build_matrix([[open(p, "rb").read()] for p in genomes[:5]], 21, 1, False)
    assert lines[0].split("\t") == ["kmers"] + [os.path.basename(p)[:-4] for p in genomes[:5]]
    strs = orc.decode_kmers(want["kmers"], 21)
    body = [l for l in lines[1:] if l]
    assert [l.split("\t")[0] for l in body] == strs
    for c in (0, len(body) // 2, len(body) - 1):
        cells = body[c].split("\t")[1:]
        assert cells == [str((int(want["matrix"][g // 64, c]) >> (63 - g % 64)) & 1) for g in range(5)]


@pytest.mark.parametrize("k,budget", [(31, None), (31, 13000), (47, None)])
def test_dsk_pooled_count(genomes, tmp_path, k, budget):
    """src/app.py:1372: one pooled count over every listed file, DSK's default abundance-min 2.
    budget: a tiny byte budget forces the pool through several chunks and the device-side merge of
    their counted sets (counts summed, filter on the totals); k = 47: two-word k-mers, two columns."""
    h5 = import_module(PKG + ".h5lite")
    lst = str(tmp_path / "dsk_output")
    open(lst, "w").writelines(p + "\n" for p in genomes)
    _run([os.path.join(CLI, "dsk"), "-file", lst, "-out-dir", str(tmp_path), "-kmer-size", str(k)],
         env={"GRM_BATCH_BYTES": str(budget)} if budget else None)
    km, ct, nocc = orc.count_genome([open(p, "rb").read() for p in genomes], k, 2)
    with h5.File(str(tmp_path / "dsk_output.h5")) as f:
        got = f.read("kmers")
        assert got.shape == (km[:, 0].shape if k <= 32 else km.shape)
        assert (got == (km[:, 0] if k <= 32 else km)).all() and (f.read("abundances") == ct).all()
        assert f.get_attr("nb_kmers_total") == float(nocc)


@pytest.mark.parametrize("k", [31, 101])
def test_kover_create_from_contigs(genomes, tmp_path, k):
    """k = 101: three-word k-mers through the same command (the reference accepts --kmer-size up to 128, bin/kover/kover:114)"""
    import grm_amd  # noqa: F401
    kd = import_module(PKG + ".kover_dataset")
    ids = [os.path.basename(p)[:-4] for p in genomes]
    data = str(tmp_path / "paths.tsv")
    open(data, "w").writelines("%s\t%s\n" % (i, p) for i, p in zip(ids, genomes))                       # src/kover.py:40-49
    md = str(tmp_path / "md.tsv")
    open(md, "w").writelines("%s\t%s\n" % (i, "R" if n % 3 else "S") for n, i in enumerate(ids))
    out = str(tmp_path / "DATASET.kover")
    # the command GRM builds (src/kover.py:52-108)
    _run([os.path.join(CLI, "kover"), "dataset", "create", "from-contigs", "--genomic-data", data,
          "--phenotype-description", "desc", "--phenotype-metadata", md, "--output", out, "--kmer-size", str(k),
          "--n-cpu", "4", "--compression", "4", "-x"])
    r = kd.KoverDatasetReader(out)
    order = r.genome_identifiers
    labels = {i: (0 if n % 3 else 1) for n, i in enumerate(ids)}        # tags sorted: R=0, S=1
    assert [labels[i] for i in order] == sorted(labels.values())        # rows label-sorted (create.py:334-336)
    want = orc.build_matrix([[open(genomes[ids.index(i)], "rb").read()] for i in order], k, 1, True)
    assert r.kmer_sequences == orc.decode_kmers(want["kmers"], k)
    assert (r.kmer_matrix == want["matrix"]).all()
    assert r.attr("filter") == "singleton" and r.attr("genome_source_type") == "contigs"
    assert r.phenotype[1] == ["R", "S"]


def test_kover_create_from_contig_tree(genomes, tmp_path):
    """`--genomic-data` may name GRM's contigs/<genome name>/ directory itself (src/app.py:576-583):
    same dataset as through the `<genome>_paths.tsv` of src/kover.py:40-49"""
    import grm_amd  # noqa: F401
    kd = import_module(PKG + ".kover_dataset")
    tree = os.path.dirname(genomes[0])
    tsv = kd.create_contigs_path_tsv(os.path.dirname(tree), os.path.basename(tree))
    assert [l.split("\t")[0] for l in open(tsv)] == sorted(os.path.basename(p)[:-4] for p in genomes)
    outs = []
    for n, data in enumerate((tree, tsv)):
        out = str(tmp_path / ("d%d.kover" % n))
        _run([os.path.join(CLI, "kover"), "dataset", "create", "from-contigs", "--genomic-data", data, "--output", out,
              "--kmer-size", "31", "--singleton-kmers", "--compression", "1"])
        outs.append(kd.KoverDatasetReader(out))
    a, b = outs
    assert a.genome_identifiers == b.genome_identifiers == sorted(os.path.basename(p)[:-4] for p in genomes)
    assert a.kmer_sequences == b.kmer_sequences and (a.kmer_matrix == b.kmer_matrix).all()
    want = orc.build_matrix([[open(p, "rb").read()] for p in sorted(genomes)], 31, 1, False)
    assert (a.kmer_matrix == want["matrix"]).all()
    os.remove(tsv)


@pytest.mark.parametrize("k,filt", [(31, True), (47, False), (90, True)])
def test_kover_create_two_pass_chunks(tmp_path, k, filt):
    """contig sets beyond the device budget: two passes over chunks of 64 genomes (dictionary
    accumulator, row stacking); 150 genomes with a budget of ~70 genome files -> chunks 64, 64, 22.  k = 90: three-word k-mers, the
    staged calls over the sort path"""
    import grm_amd  # noqa: F401
    kd = import_module(PKG + ".kover_dataset")
    synth = import_module(PKG + ".synth")
    pg = synth.PanGenome(genome_len=20_000, n_snps=300, n_accessory=10, accessory_len=700, seed=11, n_contigs=2)
    n = 150
    ids, paths = [], []
    for g in range(n):
        p = tmp_path / ("%04d.fna" % g)
        pg.genome(g).tofile(str(p))
        ids.append("%04d" % g)
        paths.append(str(p))
    size = os.path.getsize(paths[0])
    data = str(tmp_path / "paths.tsv")
    open(data, "w").writelines("%s\t%s\n" % (i, p) for i, p in zip(ids, paths))
    out = str(tmp_path / "BIG.kover")
    r = _run([os.path.join(CLI, "kover"), "dataset", "create", "from-contigs", "--genomic-data", data, "--output", out,
              "--kmer-size", str(k), "--compression", "1", "-x"] + ([] if filt else ["--singleton-kmers"]),
             env={"GRM_BATCH_BYTES": str(size * 70)})
    assert "two passes" in r.stdout and "3 chunks" in r.stdout, r.stdout
    rd = kd.KoverDatasetReader(out)
    assert rd.genome_identifiers == ids
    want = orc.build_matrix([[open(p, "rb").read()] for p in paths], k, 1, filt)
    assert rd.kmer_sequences == orc.decode_kmers(want["kmers"], k)
    assert (rd.kmer_matrix == want["matrix"]).all()


def test_kover_create_from_reads_chunked(tmp_path):
    """from-reads (create.py:399-523): a directory of FASTQ files per genome, k=21, abundance-min 2;
    a tiny byte budget forces the multidsk-style chunked counting + dsk2kover-style merge"""
    import gzip
    import grm_amd  # noqa: F401
    kd = import_module(PKG + ".kover_dataset")
    rng = np.random.RandomState(3)
    ref = cases.rand_seq(rng, 20000)
    ids, dirs, imgs = [], [], []
    for g in range(5):
        var = list(ref)
        for p in rng.randint(0, len(ref), size=15):
            var[p] = "ACGT"[rng.randint(4)]
        var = "".join(var)
        reads = [var[s:s + 120] for s in rng.randint(0, len(var) - 120, size=1500)]
        reads = [cases.revcomp(r) if i % 3 == 0 else r for i, r in enumerate(reads)]
        d = tmp_path / ("reads_%d" % g)
        d.mkdir()
        f1, f2 = cases.fastq(reads[:750]).encode(), cases.fastq(reads[750:]).encode()
        (d / "a_1.fastq").write_bytes(f1)
        (d / "a_2.fastq.gz").write_bytes(gzip.compress(f2))
        ids.append("G%d" % g)
        dirs.append(str(d))
        imgs.append([f1, f2])
    data = str(tmp_path / "reads.tsv")
    open(data, "w").writelines("%s\t%s\n" % (i, d) for i, d in zip(ids, dirs))
    md = str(tmp_path / "md.tsv")
    open(md, "w").writelines("%s\t%d\n" % (i, n % 2) for n, i in enumerate(ids))
    out = str(tmp_path / "READS.kover")
    _run([os.path.join(CLI, "kover"), "dataset", "create", "from-reads", "--genomic-data", data,
          "--phenotype-description", "d", "--phenotype-metadata", md, "--output", out, "--kmer-size", "21",
          "--kmer-min-abundance", "2", "--singleton-kmers", "--compression", "4", "-x"], env={"GRM_BATCH_BYTES": "300000"})
    r = kd.KoverDatasetReader(out)
    order = r.genome_identifiers
    want = orc.build_matrix([imgs[ids.index(i)] for i in order], 21, 2, False)
    assert r.kmer_sequences == orc.decode_kmers(want["kmers"], 21)
    assert (r.kmer_matrix == want["matrix"]).all()
    assert r.attr("genome_source_type") == "reads" and r.attr("filter") == "nothing"


def test_kover_split_on_the_device(tmp_path):
    """`kover dataset split` with the per-k-mer risk tables from the device sweep (dataset/split.py:171-188): the same
    groups, indices and tables as the numpy restatement of the reference run on a copy of the file with the same seed;
    Matrix.risk_tables vs a dense recomputation; sum_rows vs numpy on 130 genomes (3 word-rows, last one ragged)."""
    import shutil
    import grm_amd
    kd = import_module(PKG + ".kover_dataset")
    h5 = import_module(PKG + ".h5lite")
    synth = import_module(PKG + ".synth")
    n, k = 130, 31
    pg = synth.PanGenome(genome_len=40_000, n_snps=800, n_accessory=10, accessory_len=1500, seed=5, n_contigs=2)
    d = str(tmp_path)
    paths = []
    for g in range(n):
        p = os.path.join(d, "g%03d.fna" % g)
        pg.genome(g).tofile(p)
        paths.append(p)
    data = os.path.join(d, "paths.tsv")
    open(data, "w").writelines("g%03d\t%s\n" % (g, p) for g, p in enumerate(paths))
    md = os.path.join(d, "md.tsv")
    rng = np.random.RandomState(3)
    lab = rng.randint(0, 2, size=n)
    open(md, "w").writelines("g%03d\t%d\n" % (g, lab[g]) for g in range(n))
    out = os.path.join(d, "D.kover")
    _run([os.path.join(CLI, "kover"), "dataset", "create", "from-contigs", "--genomic-data", data, "--phenotype-description", "p",
          "--phenotype-metadata", md, "--output", out, "--kmer-size", str(k), "--compression", "4", "-x"])
    ref = os.path.join(d, "REF.kover")
    shutil.copy(out, ref)
    _run([os.path.join(CLI, "kover"), "dataset", "split", "--dataset", out, "--id", "s", "--train-size", "0.7", "--folds", "3",
          "--random-seed", "11"])
    kd.split_with_proportion(ref, "s", 0.7, 11, n_folds=3)            # numpy restatement of rules.py / split.py
    with h5.File(out) as f, h5.File(ref) as r:
        groups = ["splits/s"] + ["splits/s/folds/fold_%d" % i for i in (1, 2, 3)]
        assert sorted(f.list_group("splits/s/folds")) == ["fold_1", "fold_2", "fold_3"]
        for g in groups:
            for name in ("train_genome_idx", "test_genome_idx", "unique_risks", "unique_risk_by_kmer", "unique_risk_by_anti_kmer"):
                a, b = f.read(g + "/" + name), r.read(g + "/" + name)
                assert a.dtype == b.dtype and a.shape == b.shape and (a == b).all(), (g, name)
        labels = f.read("phenotype")
        train = f.read("splits/s/train_genome_idx").astype(np.int64)
    # the library calls directly, against a dense recomputation
    with grm_amd.Context(0) as ctx:
        m = kd.device_matrix(ctx, out)
        packed = kd.KoverDatasetReader(out).kmer_matrix
        U = packed.shape[1]
        dense = np.zeros((n, U), dtype=np.int64)
        for g in range(n):
            dense[g] = (packed[g // 64] >> np.uint64(63 - g % 64)) & np.uint64(1)
        pos, neg = train[labels[train] == 1], train[labels[train] == 0]
        risk = np.round(((len(pos) - dense[pos].sum(axis=0)) + dense[neg].sum(axis=0)) / len(train), 5)
        uniq, by_kmer, by_anti = m.risk_tables(labels, train)
        assert (np.diff(uniq) > 0).all() and (uniq[by_kmer] == risk).all() and (uniq[by_anti] == np.round(1.0 - risk, 5)).all()
        some = sorted(rng.choice(n, size=57, replace=False).tolist())
        assert (m.sum_rows(some) == dense[some].sum(axis=0)).all()
        assert (m.column_counts() == dense.sum(axis=0)).all()
        m.free()


def test_bench_starts_its_ranks_itself():
    """`bench.py --gpus 2` as the driver runs it for N > 1 (here: both ranks on cuda:0 over gloo, GRM_BENCH_REHEARSAL=1): the parent
    starts the ranks, rank 0 prints ONE line with n_gpus = 2, and the strong-scaling headline (the same genome set split over the
    ranks in word-row blocks) ends with the same columns as the one-rank run"""
    import json
    common = ["--genomes", "160", "--genome-len", "150000", "--steps", "1", "--warmup", "1", "--no-e2e", "--no-random", "--cpu-genomes", "0"]
    one = _run([os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common)
    two = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env={"GRM_BENCH_REHEARSAL": "1"})
    l1 = [l for l in one.stdout.splitlines() if l.startswith("{")]
    l2 = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(l1) == 1 and len(l2) == 1, (one.stdout, two.stdout)
    a, b = json.loads(l1[0]), json.loads(l2[0])
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2
    assert b["scaling"] == "strong" and b["config"]["genomes_total"] == 160
    assert a["config"]["columns"] == b["config"]["columns"] > 0
    assert b["collective"]["world_size_seen"] == 2 and b["collective"]["allgather_calls"] >= 1
    for line in (a, b):
        assert line["roofline"] and line["unit"] == "k-mers/s" and line["value"] > 0


def test_bench_two_ranks_end_to_end_leg():
    """`bench.py --gpus 2` with its e2e leg: files -> .kover over the two ranks (multi_gpu.from_contigs_sharded: every rank reads its own
    block, one dictionary exchange, per-rank device deflate, rank 0 appends), a sample of the file checked against the CPU restatement"""
    import json
    two = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genomes", "160", "--genome-len", "150000", "--steps", "1", "--warmup", "1",
                "--no-random", "--no-weak", "--cpu-genomes", "0"], env={"GRM_BENCH_REHEARSAL": "1"})
    line = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][0])
    e = line["e2e"]
    assert e["ranks"] == 2 and e["genomes"] == 160 and e["columns"] == line["config"]["columns"] > 0
    assert e["bit_exact_sample"] is True and line["bit_exact"]["e2e"] is True
