"""N GPUs behind the command surface (SURVEY 8(e) reached from the drop-in executables).

The reference's one parallel entry point is `mpiexec -n 4 Ray survey.conf` (src/app.py:1310); its dataset span is
`kover dataset create` -> multidsk -> dsk2kover (bin/kover/core/kover/dataset/create.py:365-390).  Here:

  GRM_DEVICES=0,1,2,3   a drop-in that has touched no GPU yet starts ONE CHILD PER DEVICE (itself, same argv), the children
                        form a torch.distributed group (RCCL = backend "nccl" over xGMI; gloo with host staging when two
                        ranks share a device: the one-GPU rehearsal of the path) and split the genomes in blocks of whole
                        word-rows (distributed.shard_genomes);
  mpiexec -n 4 Ray      the ranks mpiexec starts ARE the ranks (PMI_RANK / PMI_SIZE; rendezvous through a loopback TCP store whose port
                        follows from the launch): rank r works on device GRM_DEVICES[r mod len] (default: all on GRM_DEVICE).

Per rank: read + upload ITS files (N file readers, N PCIe links), distributed.sharded_step (ONE dictionary all-gather), then
  .kover   the rank deflates the chunks of ITS word-rows on its device (chunks are (1, 100000): rows are independent), leaves the
           streams in a spool file, rank 0 appends everything with H5Dwrite_chunk (grm_write_kover_h5_parts);
  TSV      the word-rows are all-gathered (rows x U words: small), every rank formats a slice of the k-mers and writes it at
           its final offset of the one file (all TSV rows have the same byte length, create.py:130-137).
torch is used for the process group and the collective's buffers only.
"""
import os
import socket
import subprocess
import sys
import time

import numpy as np

from . import distributed as Dm
from . import kover_dataset as kd


# ---- launching ---------------------------------------------------------------------------------
def device_list():
    """GRM_DEVICES=0,1,... (or "all") -> list of device ordinals, None when unset"""
    v = os.environ.get("GRM_DEVICES", "").strip()
    if not v:
        return None
    if v == "all":
        import torch
        return list(range(torch.cuda.device_count()))            # (device_count alone does not initialise the GPU)
    return [int(x) for x in v.replace(" ", "").split(",") if x != ""]


def is_rank():
    return "GRM_RANK" in os.environ


def mpi_world():
    """(rank, size) when an MPI launcher started this process, else None"""
    for r, s in (("PMI_RANK", "PMI_SIZE"), ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"), ("PMIX_RANK", "PMIX_SIZE"),
                 ("MV2_COMM_WORLD_RANK", "MV2_COMM_WORLD_SIZE"), ("SLURM_PROCID", "SLURM_NTASKS")):
        if r in os.environ and s in os.environ:
            try:
                return int(os.environ[r]), int(os.environ[s])
            except ValueError:
                pass
    return None


def spawn_ranks(devices, argv=None):
    """parent of an N-rank run of this very executable: must not have touched the GPU.  -> exit code"""
    argv = argv or sys.argv
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    shared = len(set(devices)) < len(devices)
    procs = []
    for r, dev in enumerate(devices):
        env = dict(os.environ, GRM_RANK=str(r), GRM_WORLD=str(len(devices)), GRM_RANK_DEVICE=str(dev), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if shared:
            env.setdefault("GRM_DIST_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # one rank failed: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.02)
    return rc if rc >= 0 else 1


class Ranks:
    """this process's place in the run: rank, world, torch device, the group of the ranks that hold genomes"""

    def __init__(self, rank, world, device_ordinal, owns_group=True):
        import torch
        self.torch = torch
        self.rank, self.world, self.device_ordinal = rank, world, device_ordinal
        self.device = torch.device("cuda", device_ordinal) if device_ordinal is not None else torch.device("cpu")     # (None: CPU tests over gloo)
        self.owns_group = owns_group
        self.group = None
        self._stores = []

    @classmethod
    def from_env(cls):
        """a child of spawn_ranks"""
        import torch
        import torch.distributed as dist
        rank, world, dev = int(os.environ["GRM_RANK"]), int(os.environ["GRM_WORLD"]), int(os.environ["GRM_RANK_DEVICE"])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev)
        backend = os.environ.get("GRM_DIST_BACKEND", "nccl")
        init = "tcp://%s:%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ["MASTER_PORT"])
        if backend == "nccl":
            dist.init_process_group("nccl", init_method=init, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, init_method=init, rank=rank, world_size=world)
        return cls(rank, world, dev)

    @classmethod
    def from_mpi(cls, rendezvous_dir):
        """ranks started by mpiexec (src/app.py:1310); rendezvous_dir: the output directory all of them were given"""
        import torch
        import torch.distributed as dist
        rank, world = mpi_world()
        world = min(world, int(os.environ.get("GRM_MPI_RANKS_USED", world)))      # (ranks beyond the useful ones have left already)
        devices = device_list() or [int(os.environ.get("GRM_DEVICE", "0"))]
        dev = devices[rank % len(devices)]
        shared = world > len(set(devices))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev)
        # Rendezvous without a launcher-provided address: a TCP store on the loopback interface whose port every rank derives from what
        # the ranks of ONE launch share and other launches do not -- their parent (mpiexec's proxy) and the output directory.  (A
        # file store next to the output left its file behind whenever a slow rank closed after rank 0 had removed it.)
        import zlib
        port = int(os.environ.get("GRM_RENDEZVOUS_PORT", 20000 + zlib.crc32(("%d:%s" % (os.getppid(), os.path.abspath(rendezvous_dir))).encode()) % 30000))
        from datetime import timedelta
        store = dist.TCPStore("127.0.0.1", port, world, is_master=(rank == 0), timeout=timedelta(seconds=120))
        backend = os.environ.get("GRM_DIST_BACKEND", "gloo" if shared else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", store=store, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, store=store, rank=rank, world_size=world)
        return cls(rank, world, dev)

    @classmethod
    def from_initialized(cls, device_ordinal):
        """inside a process group somebody else made (bench.py under torchrun)"""
        import torch.distributed as dist
        return cls(dist.get_rank(), dist.get_world_size(), device_ordinal, owns_group=False)

    def active_group(self, n_active):
        """group of ranks 0 .. n_active-1 (those whose shard holds genomes) + its gloo twin for host integers.  Collective over
        ALL ranks: idle ranks call it too."""
        import torch.distributed as dist
        if n_active == self.world:
            self.group = None
            Dm._host_group(None)
            return None
        ranks = list(range(n_active))
        g = dist.new_group(ranks=ranks)
        host = g if dist.get_backend() == "gloo" else dist.new_group(ranks=ranks, backend="gloo")
        Dm._host_groups[id(g)] = host
        self.group = g
        return g

    def barrier(self):
        import torch.distributed as dist
        if self.device.type == "cuda" and dist.get_backend() == "nccl":
            self.torch.cuda.synchronize(self.device)
            dist.barrier(device_ids=[self.device_ordinal])
        else:
            dist.barrier()

    def close(self):
        import torch.distributed as dist
        if self.owns_group and dist.is_initialized():
            try:
                self.barrier()
            finally:
                dist.destroy_process_group()


def ranks_worth_starting(n_genomes, n_ranks):
    """ranks whose shard would hold genomes (blocks of 64 genomes = whole word-rows): ten genomes keep one rank busy"""
    return sum(1 for a, b in Dm.shard_genomes(n_genomes, max(1, n_ranks)) if b > a)


# ---- the sharded spans -------------------------------------------------------------------------
def shardable(files_per_genome, world, kmer_size, abundance_min):
    """None when the genomes can be split over `world` ranks, else why not (the caller then runs on one device)"""
    shards = Dm.shard_genomes(len(files_per_genome), world)
    for a, b in shards:
        if sum(kd._input_bytes(f) for fl in files_per_genome[a:b] for f in fl) > kd.DEFAULT_BATCH_BYTES:
            return "a rank's shard exceeds the device batch budget (GRM_BATCH_BYTES)"
    return None


def _rank_matrix(ctx, R, files_per_genome, kmer_size, abundance_min, filter_singleton, progress, stats=None):
    """this rank's word-rows against the GLOBAL dictionary (None for a rank without genomes); -> (matrix, shards, n_active)"""
    shards = Dm.shard_genomes(len(files_per_genome), R.world)
    n_active = sum(1 for a, b in shards if b > a)
    group = R.active_group(n_active)
    a, b = shards[R.rank]
    if b <= a:
        return None, shards, n_active
    batch = ctx.batch(b - a)
    try:
        nbytes = 0
        for j, fl in enumerate(files_per_genome[a:b]):
            for f in fl:
                batch.add_file(j, f)
                nbytes += os.path.getsize(f)
        batch.upload()
        progress("rank %d: genomes %d..%d uploaded (%.2f GB)" % (R.rank, a, b - 1, nbytes / 1e9))
        m = Dm.sharded_step(batch, kmer_size, abundance_min, bool(filter_singleton), R.device, group=group, stats=stats)
        progress("rank %d: device pass done, %d k-mers" % (R.rank, m.n_kmers))
    finally:
        batch.free()
    return m, shards, n_active


def from_contigs_sharded(ctx, R, contig_list_path, output_path, kmer_size, filter_singleton, phenotype_description, phenotype_metadata_path,
                         gzip, progress=None, abundance_min=1, source_type="contigs", spool_dir=None):
    """kover_dataset.from_contigs over the ranks of R (dataset/create.py:278-396 seen from N GPUs).  Every rank calls it with
    the same arguments; -> number of k-mers.  The file is complete when rank 0 returns; all ranks leave together."""
    from .engine import ChunkStreams
    _say = progress or (lambda m: None)
    t0 = time.time()
    progress = lambda m: _say("[%6.2fs] %s" % (time.time() - t0, m))
    ids, labels, tags, ctype, files_per_genome = kd.plan_dataset(contig_list_path, phenotype_description, phenotype_metadata_path,
                                                                 warn=progress if R.rank == 0 else None)
    why = shardable(files_per_genome, R.world, kmer_size, abundance_min)
    if why is not None:
        # one device does it all; the others only keep the collective calendar
        if R.rank == 0:
            progress("not sharded over %d ranks: %s" % (R.world, why))
            n = kd.from_contigs(ctx, contig_list_path, output_path, kmer_size, filter_singleton, phenotype_description, phenotype_metadata_path,
                                gzip, progress=_say, abundance_min=abundance_min, source_type=source_type)
        else:
            n = 0
        R.barrier()
        return n
    tmp = output_path + ".tmp"
    # the ranks' streams wait for rank 0 in memory-backed files when the node has them (the ranks share a node), else next to the output
    if not spool_dir:
        spool_dir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else (os.path.dirname(os.path.abspath(output_path)) or ".")
    import zlib
    tag = "%s.%08x" % (os.path.basename(output_path), zlib.crc32(os.path.abspath(output_path).encode()))
    spool = lambda r: os.path.join(spool_dir, ".grm_%s.rank%d.chunks" % (tag, r))
    if R.rank == 0:
        kd.write_header(tmp, source_type, contig_list_path, phenotype_description, phenotype_metadata_path, gzip, ids, labels, tags, ctype,
                        "singleton" if filter_singleton else "nothing")
        progress("multidsk+dsk2kover (gfx950 x %d): %d genomes, k=%d" % (R.world, len(ids), kmer_size))
    m, shards, n_active = _rank_matrix(ctx, R, files_per_genome, kmer_size, abundance_min, filter_singleton, progress)
    n = 0
    try:
        mine = None
        if m is not None:
            mine = m.deflate_rows(kd.BLOCK_SIZE)                    # this rank's chunks, made on its device
            if R.rank != 0:
                mine.tofile(spool(R.rank))
        R.barrier()                                                 # every spool file is complete
        if R.rank == 0:
            parts = []
            for r in range(n_active):
                a, b = shards[r]
                parts.append((a // 64, (b - a + 63) // 64, mine if r == 0 else ChunkStreams.fromfile(spool(r))))
            n = m.n_kmers
            progress("dictionary: %d k-mers; appending the chunks of %d ranks (gzip %d)" % (n, n_active, gzip))
            m.write_kover_h5_parts(tmp, parts, (len(ids) + 63) // 64, gzip, kd.BLOCK_SIZE)
            os.replace(tmp, output_path)
            progress("HDF5 written")
    finally:
        if m is not None:
            m.free()
        if R.rank == 0:
            for r in range(1, R.world):
                try:
                    os.remove(spool(r))
                except OSError:
                    pass
    R.barrier()
    return n


def gathered_matrix(ctx, R, files_per_genome, kmer_size, abundance_min, filter_singleton, progress=None):
    """the whole matrix on every rank that holds genomes (host-only HostMatrix; None on idle ranks): rows all-gathered"""
    from .engine import HostMatrix
    progress = progress or (lambda m: None)
    m, shards, n_active = _rank_matrix(ctx, R, files_per_genome, kmer_size, abundance_min, filter_singleton, progress)
    if m is None:
        return None, n_active
    try:
        rows = Dm.gather_rows(m.data(), R.device, group=R.group)
        return HostMatrix(m.kmers(), rows, len(files_per_genome), kmer_size), n_active
    finally:
        m.free()


def tsv_sharded(ctx, R, samples, kmer_size, tsv_path, progress=None):
    """Ray Surveyor's KmerMatrix.tsv from the ranks of R: samples = [(name, path)]; -> number of k-mers (on every rank)"""
    progress = progress or (lambda m: None)
    files = [[p] for _, p in samples]
    names = [n for n, _ in samples]
    why = shardable(files, R.world, kmer_size, 1)
    tmp = tsv_path + ".tmp"
    n = 0
    if why is not None:
        if R.rank == 0:
            m = kd.matrix_of_files(ctx, files, kmer_size, 1, False, progress)
            n = m.n_kmers
            m.write_tsv(names, tsv_path)
            m.free()
        R.barrier()
        return n
    if R.rank == 0 and os.path.exists(tmp):
        os.remove(tmp)
    R.barrier()
    hm, n_active = gathered_matrix(ctx, R, files, kmer_size, 1, False, progress)
    if hm is not None:
        n = hm.n_kmers
        per = (n + n_active - 1) // n_active if n_active else 0
        a = min(n, R.rank * per)
        hm.write_tsv_slice(names, tmp, a, min(n, a + per) - a)      # the header travels with slice 0
        hm.free()
    R.barrier()
    if R.rank == 0:
        os.replace(tmp, tsv_path)
    R.barrier()
    return n
