"""N>1 path on CPU: world_size-2 gloo run of distributed.sharded_step / gather_rows.

The per-rank compute engine is replaced by a TEST-ONLY stand-in backed by the oracle (the
real engine needs a GPU); what is under test is the sharding plan, the single dictionary
all-gather, the deterministic merge contract (flags 1/2 -> singleton filter) and the row
assembly -- i.e. everything in genomic-resistance-mapping-grm-_amd/distributed.py."""
import ctypes as C
import os
import socket
from importlib import import_module

import numpy as np
import pytest

from oracle import oracle_ctypes as orc
from tests import cases

PKG = "genomic-resistance-mapping-grm-_amd"


def test_shard_plan():
    import grm_amd  # noqa: F401
    D = import_module(PKG + ".distributed")
    s = D.shard_genomes(1000, 8)
    assert [b - a for a, b in s] == [128] * 7 + [104]
    assert all(a % 64 == 0 for a, _ in s) and s[0][0] == 0 and s[-1][1] == 1000
    assert D.shard_genomes(10, 4) == [(0, 10), (10, 10), (10, 10), (10, 10)]
    assert D.shard_genomes(130, 2) == [(0, 128), (128, 130)]
    assert D.shard_genomes(0, 2) == [(0, 0), (0, 0)]


class FakeMatrix:
    def __init__(self, kmers, data):
        self._k, self._d = kmers, data

    def kmers(self):
        return self._k

    def data(self):
        return self._d

    def free(self):
        pass


class OracleBatch:
    """same staged interface as engine.Batch, host memory instead of HBM; keys are rows of
    `words` uint64 (most significant first), 2 words for 33 <= k <= 64"""

    def __init__(self, genomes):
        self.genomes = genomes

    def partition(self, k, abundance_min):
        self.w = 2 if k > 32 else 1
        self.sets = [orc.count_genome(g, k, abundance_min)[0].reshape(-1, self.w) for g in self.genomes]

    def local_dict(self):
        allk = np.concatenate(self.sets) if self.sets else np.zeros((0, self.w), np.uint64)
        vals, counts = np.unique(allk, axis=0, return_counts=True)
        perm = np.random.RandomState(len(vals)).permutation(len(vals))      # order must not matter
        self.lk = np.ascontiguousarray(vals[perm].astype(np.uint64))
        self.lf = np.where(counts[perm] > 1, 2, 1).astype(np.uint8)
        return len(self.lk)

    def export_dict(self, keys_ptr, flags_ptr):
        C.memmove(keys_ptr, self.lk.ctypes.data, self.lk.nbytes)
        C.memmove(flags_ptr, self.lf.ctypes.data, self.lf.nbytes)

    # the exchange record (include/grm_kmer.h: grm_exchange_layout and friends); the layout arithmetic is the
    # library's own (pure host code), the record is written into host memory
    bucket_bits = 3

    def exchange_layout(self, n_max, words, bucket_bits):
        import grm_amd
        f, o, s = C.c_uint64(), C.c_uint64(), C.c_uint64()
        grm_amd._lib.load().grm_exchange_layout(n_max, words, bucket_bits, C.byref(f), C.byref(o), C.byref(s))
        return int(f.value), int(o.value), int(s.value)

    def export_dict_ordered(self, rec_ptr, flags_off, boff_off):
        C.memmove(rec_ptr, self.lk.ctypes.data, self.lk.nbytes)
        C.memmove(rec_ptr + flags_off, self.lf.ctypes.data, self.lf.nbytes)
        boff = np.full((1 << self.bucket_bits) + 1, len(self.lk), dtype=np.uint32)      # everything in bucket 0
        boff[0] = 0
        C.memmove(rec_ptr + boff_off, boff.ctypes.data, boff.nbytes)

    def export_dict_record(self, rec_ptr, n_cap, bucket_bits):
        flags_off, boff_off, stride = self.exchange_layout(n_cap, self.w, bucket_bits)
        head = np.zeros(1, dtype=[("n", "<u8"), ("code", "<u4"), ("magic", "<u4")])
        head["n"], head["code"], head["magic"] = len(self.lk), self.bucket_bits, 0x584d5247
        C.memmove(rec_ptr + stride - 16, head.ctypes.data, 16)
        fits = len(self.lk) <= n_cap and self.bucket_bits <= (bucket_bits & 0xff)
        if fits:
            self.export_dict_ordered(rec_ptr, flags_off, boff_off)
        return fits

    def set_global_dict_gathered(self, payload_ptr, n_max, counts, bucket_bits, filter_singleton, my_rank=-1):
        flags_off, _, stride = self.exchange_layout(n_max, self.w, max(bucket_bits))
        raw = np.ctypeslib.as_array(C.cast(payload_ptr, C.POINTER(C.c_uint8)), shape=(stride * len(counts),))
        keys = [raw[r * stride: r * stride + n * 8 * self.w].view(np.uint64).reshape(n, self.w) for r, n in enumerate(counts)]
        flags = [raw[r * stride + flags_off: r * stride + flags_off + n] for r, n in enumerate(counts)]
        k = np.ascontiguousarray(np.concatenate(keys))
        f = np.ascontiguousarray(np.concatenate(flags))
        return self.set_global_dict(k.ctypes.data, f.ctypes.data, len(f), filter_singleton)

    def set_global_dict(self, keys_ptr, flags_ptr, n, filter_singleton):
        keys = np.ctypeslib.as_array(C.cast(keys_ptr, C.POINTER(C.c_uint64)), shape=(max(n, 1) * self.w,))[:n * self.w].copy().reshape(n, self.w)
        flags = np.ctypeslib.as_array(C.cast(flags_ptr, C.POINTER(C.c_uint8)), shape=(max(n, 1),))[:n].copy()
        vals, inv, counts = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
        top = np.zeros(len(vals), dtype=np.uint8)
        np.maximum.at(top, inv.reshape(-1), flags)
        multi = (counts > 1) | (top >= 2)
        self.dict = vals[multi] if filter_singleton else vals          # rows sorted (hi, lo): the column order
        return len(self.dict)

    def fill(self):
        rows = (len(self.genomes) + 63) // 64
        out = np.zeros((rows, len(self.dict)), dtype=np.uint64)
        col = {r.tobytes(): i for i, r in enumerate(self.dict)}
        for g, s in enumerate(self.sets):
            idx = np.array([col.get(r.tobytes(), -1) for r in s], dtype=np.int64)
            idx = idx[idx >= 0]
            out[g // 64, idx] |= np.uint64(1) << np.uint64(63 - g % 64)
        return FakeMatrix(self.dict, out)


def _genomes(n):
    rng = np.random.RandomState(2)
    core = cases.rand_seq(rng, 400)
    out = []
    for g in range(n):
        s = list(core)
        for p in rng.randint(0, len(core), size=3):
            s[p] = "ACGT"[rng.randint(4)]
        private = cases.rand_seq(rng, 40)
        out.append([cases.fasta([("g%d" % g, "".join(s)), ("p", private)]).encode()])
    return out


def _worker(rank, world, port, n_genomes, k, filt, q):
    import torch
    import torch.distributed as dist
    import grm_amd  # noqa: F401
    D = import_module(PKG + ".distributed")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        genomes = _genomes(n_genomes)
        a, b = D.shard_genomes(n_genomes, world)[rank]
        batch = OracleBatch(genomes[a:b])
        m = D.sharded_step(batch, k, 1, filt, torch.device("cpu"))
        rows = D.gather_rows(m.data(), torch.device("cpu"))
        if rank == 0:
            q.put((m.kmers().copy(), rows))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_genomes,filt,k", [(100, True, 15), (70, False, 15), (130, True, 15), (130, True, 40)])
def test_two_rank_gloo_matches_single_process_oracle(n_genomes, filt, k):
    """k = 40: two-word k-mers, the dictionary travels as (hi, lo) pairs"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_genomes, k, filt, q)) for r in range(2)]
    for p in procs:
        p.start()
    kmers, rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = orc.build_matrix(_genomes(n_genomes), k, 1, filt)
    assert kmers.shape == want["kmers"].shape and (kmers == want["kmers"]).all()
    assert rows.shape == want["matrix"].shape and (rows == want["matrix"]).all()


# ---- the sharded spans of multi_gpu.py (N GPUs behind the command surface) on CPU: ranks over gloo, oracle-backed engine ----
def _steps_worker(rank, world, port, sizes, q):
    import torch
    import torch.distributed as dist
    import grm_amd  # noqa: F401
    D = import_module(PKG + ".distributed")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        out = []
        for step, n_genomes in enumerate(sizes):
            genomes = _genomes(n_genomes)
            a, b = D.shard_genomes(n_genomes, world)[rank]
            batch = OracleBatch(genomes[a:b])
            if step == len(sizes) - 1:
                batch.bucket_bits = 5                        # and a rank geometry the layout did not plan for
            stats = {"bytes": 0, "ms": 0.0, "calls": 0}
            m = D.sharded_step(batch, 15, 1, True, torch.device("cpu"), stats=stats)
            rows = D.gather_rows(m.data(), torch.device("cpu"))
            out.append((m.kmers().copy(), rows, stats["calls"], stats["bytes"]))
        if rank == 0:
            q.put(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_a_step_is_one_collective_and_a_list_that_outgrows_the_layout_is_sent_again():
    """the record's stride comes from the previous step (distributed.exchange_dict): same sizes -> one all-gather and nothing else;
    a dictionary more than 1/16 larger, or more buckets, than the layout allows -> every rank reads that in the headers and all repeat
    the step once with the layout that fits; a smaller one -> one all-gather, and the layout shrinks for the step after"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sizes = [70, 70, 200, 200, 70, 70, 70]
    procs = [ctx.Process(target=_steps_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    calls = [o[2] for o in out]
    assert calls == [1, 1, 2, 1, 1, 1, 2], calls                 # the last: the bucket geometry of that step did not fit
    assert out[5][3] < out[4][3] == out[3][3]                    # the stride follows the sizes down again, one step behind
    for n_genomes, (kmers, rows, _, _) in zip(sizes, out):
        want = orc.build_matrix(_genomes(n_genomes), 15, 1, True)
        assert (kmers == want["kmers"]).all() and (rows == want["matrix"]).all()


class StandInMatrix(FakeMatrix):
    """what multi_gpu needs of engine.Matrix, on the host: the chunk streams come from zlib here (on the device: grm_deflate.hip),
    the append goes through the library's own grm_write_kover_h5_parts on a host-only matrix"""

    def __init__(self, kmers, data, n_genomes, k):
        super().__init__(kmers, data)
        self.n_genomes, self.k = n_genomes, k

    @property
    def n_kmers(self):
        return self._k.shape[0]

    def deflate_rows(self, chunk_cols):
        import zlib
        E = import_module(PKG + ".engine")
        U = self.n_kmers
        cw = min(U, chunk_cols) if U else 1
        blobs = []
        for r in range(self._d.shape[0]):
            for c0 in range(0, U, cw):
                raw = np.zeros(cw, np.uint64)
                part = self._d[r, c0:c0 + cw]
                raw[: part.shape[0]] = part
                blobs.append(zlib.compress(raw.tobytes(), 4))
        lens = np.array([len(b) for b in blobs], dtype=np.uint32)
        starts = np.concatenate(([0], np.cumsum((lens.astype(np.uint64) + 15) // 16 * 16)))[:-1].astype(np.uint64)
        buf = np.zeros(int(starts[-1] + lens[-1]) if len(blobs) else 0, np.uint8)
        for s_, b in zip(starts, blobs):
            buf[int(s_): int(s_) + len(b)] = np.frombuffer(b, np.uint8)
        return E.ChunkStreams(buf, starts, lens)

    def write_kover_h5_parts(self, path, parts, n_rows_total, gzip_level=4, chunk_cols=100000):
        import grm_amd
        hm = grm_amd.HostMatrix(self._k, self._d, self.n_genomes, self.k)
        try:
            grm_amd.Matrix.write_kover_h5_parts(hm, path, parts, n_rows_total, gzip_level, chunk_cols)
        finally:
            hm.free()


class FileOracleBatch(OracleBatch):
    def __init__(self, n):
        super().__init__([[] for _ in range(n)])

    def add_file(self, j, path):
        self.genomes[j].append(open(path, "rb").read())

    def upload(self):
        pass

    def free(self):
        pass

    def partition(self, k, abundance_min):
        self.k = k
        super().partition(k, abundance_min)

    def fill(self):
        m = super().fill()
        return StandInMatrix(m.kmers(), m.data(), len(self.genomes), self.k)


class StandInContext:
    def batch(self, n):
        return FileOracleBatch(n)


def _sharded_worker(rank, world, port, d, k, filt, what):
    import torch.distributed as dist
    import grm_amd  # noqa: F401
    mg = import_module(PKG + ".multi_gpu")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    R = mg.Ranks(rank, world, None, owns_group=True)
    try:
        if what == "kover":
            mg.from_contigs_sharded(StandInContext(), R, os.path.join(d, "paths.tsv"), os.path.join(d, "sharded.kover"), k, filt, "pheno",
                                    os.path.join(d, "md.tsv"), 4)
        else:
            samples = [l.split() for l in open(os.path.join(d, "paths.tsv"))]
            mg.tsv_sharded(StandInContext(), R, [(a, b) for a, b in samples], k, os.path.join(d, "sharded.tsv"))
    finally:
        R.close()


def _run_ranks(world, target, args):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=(r, world, port) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0


@pytest.fixture(scope="module")
def genome_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("sharded")
    gs = _genomes(130)
    with open(d / "paths.tsv", "w") as f:
        for g, bufs in enumerate(gs):
            p = d / ("g%03d.fna" % g)
            p.write_bytes(bufs[0])
            f.write("g%03d\t%s\n" % (g, p))
    with open(d / "md.tsv", "w") as f:
        for g in range(130):
            f.write("g%03d\t%d\n" % (g, (g * 7) % 3 == 0))
    return str(d), gs


@pytest.mark.parametrize("world,filt", [(2, True), (4, False)])
def test_kover_file_from_sharded_ranks(genome_files, world, filt):
    """world 4: 130 genomes fill three blocks of 64 -- the fourth rank holds nothing and only keeps the collective calendar"""
    import grm_amd  # noqa: F401
    kd = import_module(PKG + ".kover_dataset")
    d, gs = genome_files
    out = os.path.join(d, "sharded.kover")
    if os.path.exists(out):
        os.remove(out)
    _run_ranks(world, _sharded_worker, (d, 15, filt, "kover"))
    r = kd.KoverDatasetReader(out)
    ids = r.genome_identifiers
    labels = [(int(g[1:]) * 7) % 3 == 0 for g in ids]
    assert labels == sorted(labels) and sorted(ids) == ["g%03d" % g for g in range(130)]          # rows label-sorted (create.py:334-336)
    want = orc.build_matrix([gs[int(g[1:])] for g in ids], 15, 1, filt)
    assert r.kmer_sequences == orc.decode_kmers(want["kmers"], 15)
    assert (r.kmer_matrix == want["matrix"]).all()
    assert not [f for f in os.listdir(d) if f.endswith(".chunks") or f.endswith(".tmp")]           # spool files and temp output are gone
    if os.path.isdir("/dev/shm"):
        assert not [f for f in os.listdir("/dev/shm") if f.startswith(".grm_sharded.kover.")]


def test_tsv_from_sharded_ranks(genome_files):
    import grm_amd
    d, gs = genome_files
    _run_ranks(3, _sharded_worker, (d, 15, False, "tsv"))
    want = orc.build_matrix(gs, 15, 1, False)
    hm = grm_amd.HostMatrix(want["kmers"][:, 0], want["matrix"], 130, 15)
    ref = os.path.join(d, "single.tsv")
    hm.write_tsv(["g%03d" % g for g in range(130)], ref)
    hm.free()
    assert open(os.path.join(d, "sharded.tsv"), "rb").read() == open(ref, "rb").read()


def test_rank_planning():
    import grm_amd  # noqa: F401
    mg = import_module(PKG + ".multi_gpu")
    assert mg.ranks_worth_starting(10, 4) == 1 and mg.ranks_worth_starting(130, 4) == 3 and mg.ranks_worth_starting(1000, 8) == 8
    assert mg.ranks_worth_starting(64, 2) == 1 and mg.ranks_worth_starting(65, 2) == 2
    assert mg.shardable([["/dev/null"]] * 200, 2, 31, 1) is None
    assert mg.shardable([["/dev/null"]] * 200, 2, 65, 1) is None and mg.shardable([["/dev/null"]] * 200, 2, 47, 2) is None     # the sort path in stages
