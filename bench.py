#!/usr/bin/env python3
"""bench.py -- benchmark of the k-mer-matrix hot path (BASELINE.json).

A "step" = one pass of the hot path over one batch of synthetic genomes whose FASTA bytes are
already resident in HBM: packed symbols -> hash-partitioned canonical k-mers -> dictionary +
presence bits -> genome x k-mer matrix in column order.

Legs of one invocation (rank 0 prints ONE JSON line):
  headline      N=1: BASELINE.json configs[1] -- 1000 x 5 Mbp pan-genome (mode P), k=31, singleton filter.
                N>1: configs[2] -- the SAME 1000-genome set split over the ranks in blocks of whole
                word-rows (128,...,104 at N=8), ONE dictionary all-gather over RCCL; "scaling": "strong".
                The weak form (1000 genomes per GPU) is measured too and reported under "weak".
  random_acgt   counting stage (parse -> partition -> per-genome dedup + count, matrix not materialised;
                SURVEY 8(d) C2(R)) on 1000 independent uniform-ACGT genomes (split over the ranks for N>1,
                no collective): bases/s and its own roofline.
  e2e           N=1 only: the 1000 genomes as FASTA files on disk -> `kover dataset create from-contigs`
                (in process) -> gzip-4 .kover, wall clock with phases, next to the CPU restatement over the
                same span (read + count + merge + HDF5 write).  The same CPU run is the cpu_baseline.

`--gpus N` with N > 1 starts N ranks itself (one child process per GPU, before anything touches
the GPU) unless a launcher (torchrun) already did: then WORLD_SIZE / RANK / LOCAL_RANK are taken
from the environment.  GRM_BENCH_REHEARSAL=1 puts every rank on cuda:0 with the gloo backend, to
walk the N > 1 path on a one-GPU box (never for reported numbers).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

# algorithmic bytes per unit for each kernel (DESIGN.md "Kernels"); unit = what TimeScope counts
ALGO_BYTES = {
    "parse_summarize": 1.0 + 0.25,   # per input byte: read 1, write the 4-byte scan prefix of every 16-byte chunk
    "parse_pack": 1.0 + 0.25 + 0.375,    # per input byte: read 1 + that prefix, write 2+1 bits per symbol (<= byte count)
    "kmer_hist": 0.375,              # per symbol: read packed stream
    "kmer_scatter_l1": 0.375 + 8.0,  # key form, per k-mer occurrence: read packed stream, write one u64 key
    "kmer_scatter_l2": 16.0,         # key form, per key: read 8, write 8 (second radix level)
    "superkmer_l2": 32.0,            # record form, per 16-byte run record: read 16, write 16
    "superkmer_l2_keys": None,       # record form with key segments: 16 per record read + 8 per k-mer written (below)
    "bucket_dedup": 16.0 + 4.0,      # per key: read 8, write <= 8 + a 4-byte count
    "dict_build": 8.0,               # per key: read 8 (what it writes is dictionary-sized, added per launch below)
    "matrix_fill": 16.0,             # per (entry, word-row): read one presence word, write it to its column
}
# algorithmic bytes per k-mer occurrence of the straightforward extract / sort / fill pipeline
SURVEY_BYTES_PER_KMER = {1: 57.0, 2: 113.0}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1000, help="genomes of the set (N=1 / strong) or per GPU (weak)")
    ap.add_argument("--weak-only", action="store_true", help="N>1: only the weak form (--genomes per GPU) as the headline")
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--mode", default="P", choices=["P", "R"], help="generator of the headline leg")
    ap.add_argument("--stage", default="matrix", choices=["matrix", "count"],
                    help="headline leg: whole path to the matrix, or the counting stage only")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--abundance-min", type=int, default=1)
    ap.add_argument("--keep-singletons", action="store_true")
    ap.add_argument("--no-random", action="store_true", help="skip the random-ACGT counting leg")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the weak-scaling leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> .kover end-to-end leg")
    ap.add_argument("--cpu-genomes", type=int, default=-1,
                    help="genomes the CPU restatement is timed on (default: all of the e2e set; 0 = skip)")
    ap.add_argument("--gzip", type=int, default=4)
    ap.add_argument("--tmp", default=None, help="directory for the e2e files (default: a fresh temp dir)")
    ap.add_argument("--opt", action="append", default=[], help="engine tuning knob name=value (grm_set_option)")
    return ap.parse_args()


def spawn_ranks(n):
    """parent of an N-rank run: has imported neither torch nor the engine, makes no GPU call"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
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
        time.sleep(0.05)
    sys.exit(rc if rc >= 0 else 1)


class Dist:
    """world description + the few host-side reductions the bench needs"""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.rehearsal = os.environ.get("GRM_BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            self.local_rank = 0
        self.backend = None
        import torch
        self.torch = torch
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(self.local_rank)
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.backend = dist.get_backend()
        self.device = torch.device("cuda", self.local_rank)

    def sync(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def _reduce(self, v, dtype, op):
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=dtype, device="cpu" if self.rehearsal else self.device)
        self.dist.all_reduce(t, op=op)
        return t.item()

    def max_f(self, v):
        return float(self._reduce(float(v), self.torch.float64, self.dist.ReduceOp.MAX if self.world > 1 else None))

    def sum_i(self, v):
        return int(self._reduce(int(v), self.torch.int64, self.dist.ReduceOp.SUM if self.world > 1 else None))


def kernel_table(ctx, n_cols, n_rows, n_local):
    """per-kernel device times (HIP events on the engine's stream, timed region only)"""
    per = {}
    for name, ms, units in ctx.timings():
        d = per.setdefault(name, {"ms": 0.0, "launches": 0, "units": 0})
        d["ms"] += ms
        d["launches"] += 1
        d["units"] += units
    kernels = {}
    # record form of the partition: the run records (16 B) level 1 wrote = the units of "superkmer_l2"
    n_records = per["superkmer_l2"]["units"] / per["superkmer_l2"]["launches"] if "superkmer_l2" in per else None
    for name, d in per.items():
        avg_ms = d["ms"] / d["launches"]
        e = {"avg_ms": round(avg_ms, 4), "launches": d["launches"]}
        byts = None
        if name == "superkmer_l1" and n_records is not None:
            byts = 0.375 * d["units"] / d["launches"] + 16.0 * n_records           # read the packed stream, write the records
        elif name == "dict_build" and n_records is not None:
            byts = 16.0 * n_records + 8.0 * n_local * n_rows + 9.0 * n_local        # read the records; presence words + (key, flag) per entry
        elif ALGO_BYTES.get(name) is not None:
            byts = ALGO_BYTES[name] * d["units"] / d["launches"]
            if name == "dict_build":
                byts += 8.0 * n_local * n_rows + 9.0 * n_local      # presence words + (key, flag) of every entry
        if byts is not None and avg_ms > 0:
            e["algo_GBps"] = round(byts / (avg_ms * 1e-3) / 1e9, 1)
            e["algo_bytes"] = byts
        kernels[name] = e
    return kernels


def roofline_of(kernels, traffic_of=None, occurrences=None, words=1):
    dom = max((n for n in kernels if "algo_GBps" in kernels[n]), key=lambda n: kernels[n]["avg_ms"] * kernels[n]["launches"], default=None)
    if not dom:
        return None
    a = kernels[dom]["algo_GBps"]
    traffic = traffic_of(dom) if traffic_of else None
    r = {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
         "traffic": traffic, "traffic_source": "committed PMC passes of this workload (profiles/hbm_traffic.json)" if traffic else None,
         "avg_launch_ms": kernels[dom]["avg_ms"]}
    # SURVEY.md 8(d) prices the stages of a straightforward extract / sort / fill pipeline per k-mer occurrence (W = 8 B per word):
    # 1 + W extract, 4W partition + sort, 2W + 1/8 fill.  The record form reaches the k-mers through ~3 bytes per occurrence of run
    # records, so a kernel here moves far fewer bytes than the stage it stands for: `achieved` / `frac` above are the bytes THIS
    # design's kernel must move over its duration; `survey_stage` is the stage's model bytes over the same
    # duration (above 1.0: the kernel is done sooner than HBM at its peak could move the model's bytes).
    stage = {"superkmer_l1": ("extract (write W) + first partition pass (read W + write W)", 3 * 8.0 * words),
             "kmer_scatter_l1": ("extract (write W) + first partition pass (read W + write W)", 3 * 8.0 * words),
             "dict_build": ("fill: key + dictionary probe + output bit", 16.0 * words + 0.125)}.get(dom)
    if stage and occurrences:
        g = stage[1] * occurrences / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        r["survey_stage"] = {"stage": stage[0], "bytes_per_occurrence": stage[1], "achieved": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4)}
    limits = {"superkmer_l1": "VALU issue: 3.5e9 wave instructions = 5.7 of its 7.7 ms (minimizers of 52 m-mers per 32 positions in registers); "
                              "writes 1.66x its records as partial lines (profiles/r02/final_sq_counters.csv, final_pmc_hbm.csv)",
              "dict_build": "latency of 128-byte reads of segments megabytes apart + LDS lookups, 4 waves per SIMD (profiles/r02/final_sq_counters.csv)"}
    if dom in limits and traffic:           # (the committed profile is of this very workload)
        r["limited_by"] = limits[dom]
    return r


def committed_traffic(args, genomes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes, only when this leg IS the workload they were taken on"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        w = tj["workload"]
        if (w["genomes"], w["genome_len"], w["mode"], w["k"]) == (genomes, args.genome_len, "P", args.k) and not args.opt:
            return lambda name: tj["bytes_per_launch"].get(name)
    except (OSError, ValueError, KeyError):
        pass
    return None


def make_batch(ctx, synth, mode, base, n, genome_len, keep=None):
    """generate genomes base .. base+n-1, hand them to a new batch, upload (inputs then resident in HBM)"""
    from concurrent.futures import ThreadPoolExecutor
    batch = ctx.batch(n)
    if mode == "P":
        pg = synth.PanGenome(genome_len=genome_len, seed=1234)
        gen = lambda i: pg.genome(base + i)
    else:
        gen = lambda i: synth.random_genome(base + i, genome_len=genome_len, seed=1234)
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as pool:     # numpy releases the GIL in the big copies
        for i, g in enumerate(pool.map(gen, range(n))):
            if keep is not None:
                keep(i, g)
            batch.add_array(i, g)
    batch.upload()
    return batch


def timed_leg(D, ctx, step, steps, warmup):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks"""
    info = None
    for _ in range(warmup):
        info = step()
    ctx.timing(True)
    ctx.timing_reset()
    D.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        info = step()
    D.sync()
    elapsed = D.max_f(time.perf_counter() - t0)
    ctx.timing(False)
    return elapsed, info


def matrix_leg(D, Dm, ctx, args, batch, steps, warmup):
    filt = not args.keep_singletons
    xfer = {"bytes": 0, "ms": 0.0, "calls": 0}

    def step():
        if D.world == 1:
            m = batch.run(args.k, args.abundance_min, filt)
        else:
            m = Dm.sharded_step(batch, args.k, args.abundance_min, filt, D.device, stats=xfer)
        info = (m.n_kmers, m.n_rows)
        m.free()
        return info

    elapsed, (n_cols, n_rows) = timed_leg(D, ctx, step, steps, warmup)
    n_local = batch.n_local
    kernels = kernel_table(ctx, n_cols, n_rows, n_local)
    return elapsed, n_cols, n_rows, kernels, xfer


def count_leg(D, ctx, args, batch, steps, warmup):
    def step():
        batch.partition_counts(args.k, args.abundance_min)
        return None

    elapsed, _ = timed_leg(D, ctx, step, steps, warmup)
    return elapsed, kernel_table(ctx, 0, 0, 0)


def e2e_leg(ctx, synth, args, orc, n_genomes, cpu_genomes):
    """files on disk -> .kover through kover_dataset.from_contigs (the span of dataset/create.py:365-390 plus
    the header), then the CPU restatement over the same span"""
    import shutil
    import tempfile
    import numpy as np
    from importlib import import_module
    kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
    eng = import_module("genomic-resistance-mapping-grm-_amd.engine")
    d = args.tmp or tempfile.mkdtemp(prefix="grm_e2e_")
    os.makedirs(d, exist_ok=True)
    out = {}
    try:
        pg = synth.PanGenome(genome_len=args.genome_len, seed=1234)
        t0 = time.perf_counter()
        paths = []
        for g in range(n_genomes):
            p = os.path.join(d, "g%05d.fna" % g)
            pg.genome(g).tofile(p)
            paths.append(p)
        out["files_written_s"] = round(time.perf_counter() - t0, 2)
        data = os.path.join(d, "paths.tsv")
        open(data, "w").writelines("g%05d\t%s\n" % (g, p) for g, p in enumerate(paths))
        md = os.path.join(d, "md.tsv")
        open(md, "w").writelines("g%05d\t%d\n" % (g, g % 2) for g in range(n_genomes))
        kover = os.path.join(d, "DATASET.kover")
        marks = []
        t0 = time.perf_counter()
        n_cols = kd.from_contigs(ctx, data, kover, args.k, not args.keep_singletons, "synthetic phenotype", md, args.gzip,
                                 progress=lambda m: marks.append((time.perf_counter() - t0, m)))
        wall = time.perf_counter() - t0

        def at(prefix):
            return next((t for t, m in marks if prefix in m), None)
        t_up, t_dev, t_h5 = at("uploaded"), at("device pass done"), at("HDF5 written")
        out.update({
            "seconds": round(wall, 3), "genomes": n_genomes, "columns": int(n_cols),
            "genomes_per_min": round(n_genomes / wall * 60, 1),
            "kover_bytes": os.path.getsize(kover), "gzip": args.gzip,
            "phases_s": {"read_files_and_upload": round(t_up, 3) if t_up else None,
                         "device_pass_incl_allocation": round(t_dev - t_up, 3) if t_dev and t_up else None,
                         "download_deflate_write_hdf5": round(t_h5 - t_dev, 3) if t_h5 and t_dev else None},
            "span": "FASTA files (page cache) -> label-sorted header -> engine pass -> kmer_sequences / kmer_matrix / "
                    "kmer_by_matrix_column in the .kover (dataset/create.py:311-390), in process",
        })
        cpu_baseline = None
        if cpu_genomes > 0:
            n = min(cpu_genomes, n_genomes)
            cores = min(os.cpu_count() or 1, 64)
            t0 = time.perf_counter()
            bufs = [open(p, "rb").read() for p in paths[:n]]
            t_read = time.perf_counter() - t0
            res, count_s, merge_s, occ_cpu = orc.pipeline(bufs, args.k, args.abundance_min, not args.keep_singletons, cores)
            del bufs
            t1 = time.perf_counter()
            cpu_kover = os.path.join(d, "CPU.kover")
            ids = ["g%05d" % g for g in range(n)]
            labels = np.array([g % 2 for g in range(n)], dtype=np.uint8)
            kd.write_header(cpu_kover, "contigs", data, "synthetic phenotype", md, args.gzip, ids, labels, ["0", "1"], "binary", "singleton")
            hm = eng.HostMatrix(res["kmers"], res["matrix"], n, args.k)
            hm.write_kover_h5(cpu_kover, args.gzip, 100000)
            hm.free()
            t_h5cpu = time.perf_counter() - t1
            cpu_total = t_read + count_s + merge_s + t_h5cpu
            out["cpu_seconds"] = round(cpu_total, 2)
            out["cpu"] = {"genomes": n, "cores": cores, "read_s": round(t_read, 2), "count_s": round(count_s, 2), "merge_s": round(merge_s, 2),
                          "hdf5_s": round(t_h5cpu, 2), "columns": int(res["kmers"].shape[0]),
                          "what": "CPU restatement of DSK + dsk2kover semantics (oracle/, threaded) + this library's host-side HDF5 writer; "
                                  "the reference binaries are absent"}
            if n == n_genomes:
                out["speedup_vs_cpu"] = round(cpu_total / wall, 2)
            cpu_baseline = {"value": round(occ_cpu / (count_s + merge_s), 1), "unit": "k-mers/s", "cores": cores, "kind": "port",
                            "sample": "%d of the %d genomes of the headline set (same generator): count %.2fs + merge %.2fs on %d threads; CPU "
                                      "restatement of DSK+dsk2kover semantics (the reference binaries are absent)" % (n, n_genomes, count_s, merge_s, cores)}
        return out, cpu_baseline
    finally:
        if not args.tmp:
            shutil.rmtree(d, ignore_errors=True)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)          # does not return

    import numpy as np      # noqa: F401
    import grm_amd
    from importlib import import_module
    synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
    Dm = import_module("genomic-resistance-mapping-grm-_amd.distributed")
    D = Dist(args)
    world, rank = D.world, D.rank

    t_setup = time.time()
    ctx = grm_amd.Context(D.local_rank)
    for kv in args.opt:
        name, val = kv.split("=")
        ctx.set_option(name, int(val))

    # ---- headline ----
    strong = world > 1 and not args.weak_only
    if strong:
        base, stop = Dm.shard_genomes(args.genomes, world)[rank]
        n_mine, genomes_total = stop - base, args.genomes
    else:
        base, n_mine, genomes_total = rank * args.genomes, args.genomes, args.genomes * world
    batch = make_batch(ctx, synth, args.mode, base, n_mine, args.genome_len)
    setup_s = time.time() - t_setup
    if args.stage == "matrix":
        elapsed, n_cols, n_rows, kernels, xfer = matrix_leg(D, Dm, ctx, args, batch, args.steps, args.warmup)
    else:
        elapsed, kernels = count_leg(D, ctx, args, batch, args.steps, args.warmup)
        n_cols = n_rows = 0
        xfer = None
    batch_occ = batch.n_occurrences
    occ_total = D.sum_i(batch_occ)
    syms_total = D.sum_i(batch.n_symbols)
    input_bytes = batch.input_bytes
    batch.free()
    roofline = roofline_of(kernels, committed_traffic(args, n_mine) if (world == 1 and args.mode == "P" and args.stage == "matrix") else None,
                           occurrences=batch_occ, words=1 if args.k <= 32 else 2)
    filt_txt = "singleton filter" if not args.keep_singletons else "singletons kept"
    out = None
    if rank == 0:
        what = "canonical k-mer occurrences -> genome x k-mer presence matrix" if args.stage == "matrix" else \
               "canonical k-mer occurrences -> per-genome distinct k-mers + counts (counting stage)"
        out = {
            "metric": "k-mers/sec (%s; device-resident pass: inputs in HBM, result in HBM)" % what,
            "value": round(occ_total * args.steps / elapsed, 1), "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if (strong or world == 1) and not args.weak_only else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s x %d bp synthetic contig genomes (mode %s, seed 1234), k=%d, abundance-min %d, %s, stage %s"
                                   % ("%d in total%s" % (genomes_total, ", split over the ranks in word-row blocks" if strong else "")
                                      if (strong or world == 1) else "%d per GPU" % args.genomes,
                                      args.genome_len, args.mode, args.k, args.abundance_min, filt_txt, args.stage),
                       "genomes_total": genomes_total, "columns": int(n_cols), "word_rows_rank0": int(n_rows),
                       "input_bytes_rank0": input_bytes, "parallelism": "genome-sharded x%d" % world},
            "genomes_per_min_device_pass": round(genomes_total * args.steps / elapsed * 60, 1),
            "bases_per_s": round(syms_total * args.steps / elapsed, 1),
            "roofline": roofline, "kernels": kernels, "setup_s": round(setup_s, 1),
        }
        if args.stage == "matrix" and args.k <= 64:
            # the whole pass against SURVEY.md 8(d)'s byte model of the straightforward extract / sort / fill pipeline
            # (1 + 7W bytes per occurrence): the record form moves fewer bytes than that model, so this can pass 1
            bpk = SURVEY_BYTES_PER_KMER[1 if args.k <= 32 else 2]
            gbps = occ_total * bpk * args.steps / elapsed / 1e9 / world
            out["survey_model"] = {"bytes_per_kmer": bpk, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                                   "frac": round(gbps / HBM_PEAK_GBS, 4)}
        if world > 1:
            out["collective"] = {"backend": D.backend, "world_size_seen": D.dist.get_world_size(),
                                 "allgather_calls": xfer["calls"] if xfer else 0,
                                 "allgather_bytes_received_per_step": int(xfer["bytes"] / max(1, xfer["calls"])) if xfer else 0,
                                 "exchange_ms_per_step": round(xfer["ms"] / max(1, xfer["calls"]), 3) if xfer else None}

    # ---- weak form (N > 1) ----
    if world > 1 and strong and not args.no_weak and args.stage == "matrix":
        wb = make_batch(ctx, synth, args.mode, rank * args.genomes, args.genomes, args.genome_len)
        w_el, w_cols, _, _, _ = matrix_leg(D, Dm, ctx, args, wb, args.steps, 1)
        w_occ = D.sum_i(wb.n_occurrences)
        wb.free()
        if rank == 0:
            out["weak"] = {"value": round(w_occ * args.steps / w_el, 1), "unit": "k-mers/s", "ms_per_step": round(1000 * w_el / args.steps, 3),
                           "genomes_per_gpu": args.genomes, "genomes_total": args.genomes * world, "columns": int(w_cols)}

    # ---- random-ACGT counting stage ----
    if not args.no_random and not (args.mode == "R" and args.stage == "count"):
        if world > 1:
            rbase, rstop = Dm.shard_genomes(args.genomes, world)[rank]
        else:
            rbase, rstop = 0, args.genomes
        rb = make_batch(ctx, synth, "R", rbase, rstop - rbase, args.genome_len)
        r_el, r_kernels = count_leg(D, ctx, args, rb, args.steps, 1)
        r_occ, r_syms = D.sum_i(rb.n_occurrences), D.sum_i(rb.n_symbols)
        rb.free()
        if rank == 0:
            out["random_acgt"] = {
                "workload": "%d independent uniform-ACGT genomes x %d bp (seed 1234+i)%s, k=%d: parse -> partition -> per-genome dedup + count, "
                            "matrix not materialised" % (args.genomes, args.genome_len, ", split over %d ranks" % world if world > 1 else "", args.k),
                "bases_per_s": round(r_syms * args.steps / r_el, 1), "kmers_per_s": round(r_occ * args.steps / r_el, 1),
                "ms_per_step": round(1000 * r_el / args.steps, 3), "roofline": roofline_of(r_kernels), "kernels": r_kernels}

    # ---- end to end + CPU restatement (N = 1) ----
    if world == 1 and rank == 0 and args.stage == "matrix":
        cpu_genomes = args.genomes if args.cpu_genomes < 0 else args.cpu_genomes
        if not args.no_e2e:
            from oracle import oracle_ctypes as orc       # checker / reported baseline only: after every timed GPU region
            e2e, cpu_baseline = e2e_leg(ctx, synth, args, orc, args.genomes, cpu_genomes)
            out["e2e"] = e2e
            out["cpu_baseline"] = cpu_baseline
        elif cpu_genomes > 0:
            from oracle import oracle_ctypes as orc
            pg = synth.PanGenome(genome_len=args.genome_len, seed=1234)
            bufs = [pg.genome(i).tobytes() for i in range(min(cpu_genomes, args.genomes))]
            cores = min(os.cpu_count() or 1, 64)
            _, count_s, merge_s, occ_cpu = orc.pipeline(bufs, args.k, args.abundance_min, not args.keep_singletons, cores)
            out["cpu_baseline"] = {"value": round(occ_cpu / (count_s + merge_s), 1), "unit": "k-mers/s", "cores": cores, "kind": "port",
                                   "sample": "%d of the %d genomes (same generator): count %.2fs + merge %.2fs; CPU restatement of DSK+dsk2kover "
                                             "semantics (the reference binaries are absent)" % (len(bufs), args.genomes, count_s, merge_s)}
    if rank == 0:
        out.setdefault("cpu_baseline", None)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        D.dist.destroy_process_group()


if __name__ == "__main__":
    main()
