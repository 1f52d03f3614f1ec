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
  realistic     N=1: the same 1000 strains as real assemblies -- every genome cut into its own 20-100 contigs at its own places,
                contigs in random order and on random strands, ~1 indel site per 10 kbp (GRM's inputs: src/app.py:576-583) --
                through the same pass: ms per pass, kernel table, hit rate of dict_build's record memo.
  c5            N=1: BASELINE configs[4] -- 500 genomes, k=63 (two-word k-mers), singletons kept; pass + gzip-5 .kover writer.
  c4            N=1: BASELINE configs[3] at full depth for 8 genomes -- 150 bp reads, 100x, 0.5 % errors, FASTQ, k=21,
                abundance-min 2: the counting stage, one genome's solid set compared with the CPU restatement.
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
    ap.add_argument("--mode", default="P", choices=["P", "R", "X"], help="generator of the headline leg (X = realistic assemblies)")
    ap.add_argument("--stage", default="matrix", choices=["matrix", "count"],
                    help="headline leg: whole path to the matrix, or the counting stage only")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--abundance-min", type=int, default=1)
    ap.add_argument("--keep-singletons", action="store_true")
    ap.add_argument("--no-random", action="store_true", help="skip the random-ACGT counting leg")
    ap.add_argument("--no-weak", action="store_true", help="N>1: skip the weak-scaling leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> .kover end-to-end leg")
    ap.add_argument("--no-realistic", action="store_true", help="skip the realistic-assemblies leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the reads leg (BASELINE configs[3])")
    ap.add_argument("--no-c5", action="store_true", help="skip the k=63 leg (BASELINE configs[4])")
    ap.add_argument("--rank-budget", type=int, default=8, metavar="N",
                    help="N=1 only: the headline set cut into the N shards of an N-GPU run (BASELINE configs[2]), every rank's pass timed in this "
                         "one process on this one GPU: partition + local dictionary per shard, the exchange records, the union of the N real "
                         "payloads, the fill -- what a rank of the N-GPU run spends, without the wire (default 8; 0: skip)")
    ap.add_argument("--weak-budget", type=int, default=0, metavar="N",
                    help="N=1 only: the WEAK form of the rank budget -- N shards of --genomes genomes each (N x the headline set), rank 0's pass timed against "
                         "the payload of all N real records (the shards are made and counted one after the other: N x the set-up time)")
    ap.add_argument("--rank-budget-curve", action="store_true", help="also run the rank budget for 2 and 4 ranks (an upper bound of the 1 / 2 / 4 / 8-GPU curve)")
    ap.add_argument("--only", default=None, choices=["headline", "realistic", "c4", "c5", "random"],
                    help="run ONE leg (profiling); for any leg but the headline, the headline shrinks to 16 genomes")
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
    # record form of the partition: the run records level 1 wrote = the units of "superkmer_l2" (16 bytes each; 24 for two-word k-mers,
    # whose dictionary kernel is wh_dict_build)
    n_records = per["superkmer_l2"]["units"] / per["superkmer_l2"]["launches"] if "superkmer_l2" in per else None
    rec_bytes = 24.0 if "wh_dict_build" in per else 16.0
    for name, d in per.items():
        avg_ms = d["ms"] / d["launches"]
        e = {"avg_ms": round(avg_ms, 4), "launches": d["launches"]}
        byts = None
        if name == "superkmer_l1" and n_records is not None:
            byts = 0.375 * d["units"] / d["launches"] + rec_bytes * n_records      # read the packed stream, write the records
        elif name == "dict_build" and n_records is not None:
            byts = 16.0 * n_records + 8.0 * n_local * n_rows + 9.0 * n_local        # read the records; presence words + (key, flag) per entry
        elif name == "wh_dict_build" and n_records is not None:
            byts = 24.0 * n_records + 8.0 * n_local * n_rows + 17.0 * n_local       # the same for two-word k-mers
        elif name == "superkmer_l2" and n_records is not None:
            byts = 2.0 * rec_bytes * n_records                                      # read and write every record
        elif name == "record_count" and n_records is not None:
            byts = 16.0 * n_records + 12.0 * d["units"] / d["launches"]             # read the records; write <= a key + a 4-byte count per k-mer
        elif name == "record_merge" and n_records is not None:
            byts = 16.0 * n_records                                                 # read the records (what it writes: the distinct k-mers, few at 100x)
        elif ALGO_BYTES.get(name) is not None:
            byts = ALGO_BYTES[name] * d["units"] / d["launches"]
            if name == "dict_build":
                byts += 8.0 * n_local * n_rows + 9.0 * n_local      # presence words + (key, flag) of every entry
        if byts is not None and avg_ms > 0:
            e["algo_GBps"] = round(byts / (avg_ms * 1e-3) / 1e9, 1)
            e["algo_bytes"] = byts
        kernels[name] = e
    return kernels


def roofline_of(kernels, traffic_of=None, occurrences=None, words=1, leg=None):
    dom = max((n for n in kernels if "algo_GBps" in kernels[n]), key=lambda n: kernels[n]["avg_ms"] * kernels[n]["launches"], default=None)
    if not dom:
        return None
    a = kernels[dom]["algo_GBps"]
    traffic = traffic_of(dom) if traffic_of else None
    r = {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
         "traffic": traffic, "traffic_source": "committed PMC passes of this workload (profiles/hbm_traffic.json; rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE)" if traffic else None,
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
        r["survey_stage"] = {"stage": stage[0], "bytes_per_occurrence": stage[1], "model_GBps_equivalent": round(g, 1),
                             "times_peak": round(g / HBM_PEAK_GBS, 3),
                             "what": "SURVEY 8(d)'s bytes for the stage this kernel stands for over the kernel's duration: NOT a roofline fraction (the "
                                     "record form moves fewer bytes than the model; `frac` above is the fraction)"}
    if traffic:           # (the committed profile is of this very workload): what bounds the kernel, as read from that profile
        try:
            lj = json.load(open(os.path.join(ROOT, "profiles", "limits.json")))
            if leg is not None:
                lj = lj.get("legs", {}).get(leg, {})
            if dom in lj.get("limited_by", {}):
                r["limited_by"] = {"text": lj["limited_by"][dom], "profile_commit": lj.get("commit"), "source": lj.get("source")}
        except (OSError, ValueError):
            pass
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


def leg_traffic(leg, args, genomes):
    """the same for the other legs (profiles/hbm_traffic.json -> legs, scripts/pmc_hbm_leg.sh): only at the leg's committed size"""
    full = {"c5": 500, "c4": 8, "random_acgt": 1000, "realistic": 1000}.get(leg)
    if genomes != full or args.genome_len != 5_000_000 or args.opt:
        return None
    try:
        lj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["legs"][leg]
        return lambda name: lj["bytes_per_launch"].get(name)
    except (OSError, ValueError, KeyError):
        return None


def make_batch(ctx, synth, mode, base, n, genome_len, keep=None):
    """generate genomes base .. base+n-1, hand them to a new batch, upload (inputs then resident in HBM)"""
    from concurrent.futures import ThreadPoolExecutor
    batch = ctx.batch(n)
    if mode == "P":
        pg = synth.PanGenome(genome_len=genome_len, seed=1234)
        gen = lambda i: pg.genome(base + i)
    elif mode == "X":
        pg = synth.realistic(genome_len=genome_len, seed=1234)
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


def rank_budget_leg(ctx, synth, Dm, args, n_ranks, device):
    """BASELINE configs[2] on one GPU, rank by rank: the 1000 genomes in the word-row blocks of an n_ranks-GPU run; every shard is
    partitioned and gets its local dictionary, writes its exchange record into ONE payload laid out as the all-gather would leave it,
    and every rank then builds the global dictionary from that payload and fills its rows.  Times are wall clock around the engine
    calls (each ends synchronised), best of the timed repetitions; `rank_ms` = what the slowest rank spends between the barrier that
    starts a pass and its matrix, without the all-gather itself (priced from the payload size at xGMI link speed)."""
    import torch
    filt = not args.keep_singletons
    shards = Dm.shard_genomes(args.genomes, n_ranks)
    batches = [make_batch(ctx, synth, "P", a, b - a, args.genome_len) for a, b in shards]
    words = 1 if args.k <= 32 else 2
    reps = max(2, args.steps)
    best = {}

    def note(key, r, ms):
        best.setdefault(key, {})
        best[key][r] = min(best[key].get(r, 1e30), ms)

    n_cols = 0
    per_kernel = {}
    for rep in range(reps + 1):                 # (the first repetition sizes tables and allocates: not kept)
        keep = rep > 0
        n_locals, bbs = [], []
        for r, b in enumerate(batches):
            if keep and r == 0:
                ctx.timing(True)
                ctx.timing_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.partition(args.k, args.abundance_min)
            t1 = time.perf_counter()
            n_locals.append(b.local_dict())
            t2 = time.perf_counter()
            bbs.append(b.bucket_bits)
            if keep:
                note("partition", r, (t1 - t0) * 1e3)
                note("local_dict", r, (t2 - t1) * 1e3)
                if r == 0:
                    for name, ms, _ in ctx.timings():
                        per_kernel[name] = min(per_kernel.get(name, 1e30), ms)
                    ctx.timing(False)
        n_max = max(1, max(n_locals))
        flags_off, boff_off, stride = batches[0].exchange_layout(n_max, words, max(v & 0xff for v in bbs))
        payload = torch.empty(n_ranks * stride, dtype=torch.uint8, device=device)
        for r, b in enumerate(batches):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.export_dict_ordered(payload.data_ptr() + r * stride, flags_off, boff_off)
            torch.cuda.synchronize()
            if keep:
                note("export_record", r, (time.perf_counter() - t0) * 1e3)
        for r, b in enumerate(batches):
            if keep and r == 0:
                ctx.timing(True)
                ctx.timing_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_cols = b.set_global_dict_gathered(payload.data_ptr(), n_max, n_locals, bbs, filt, my_rank=r)
            t1 = time.perf_counter()
            m = b.fill()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            m.free()
            if keep:
                note("global_dict_from_payload", r, (t1 - t0) * 1e3)
                note("fill", r, (t2 - t1) * 1e3)
                if r == 0:
                    for name, ms, _ in ctx.timings():
                        per_kernel[name] = min(per_kernel.get(name, 1e30), ms)
                    ctx.timing(False)
        del payload
    for b in batches:
        b.free()
    stages = ["partition", "local_dict", "export_record", "global_dict_from_payload", "fill"]
    per_rank = [round(sum(best[s][r] for s in stages), 3) for r in range(n_ranks)]
    gather_bytes = n_ranks * stride
    # every rank receives (n_ranks - 1) records, each over its own xGMI link (all-pairs): one record at the link's NOMINAL 153 GB/s is a
    # LOWER bound of the exchange (all seven links delivering at once into one receiver, no latency, no size exchange) -- no measured
    # single-node RCCL all-gather figure exists yet (no multi-GPU node was available in any round)
    wire_ms = stride / 153e9 * 1e3
    return {"ranks": n_ranks, "shards": [b - a for a, b in shards], "columns": int(n_cols),
            "stage_ms_slowest_rank": {s: round(max(best[s].values()), 3) for s in stages},
            "stage_ms_mean": {s: round(sum(best[s].values()) / n_ranks, 3) for s in stages},
            "rank_ms": max(per_rank), "rank_ms_all": per_rank,
            "exchange": {"record_bytes": int(stride), "payload_bytes": int(gather_bytes), "wire_ms_lower_bound": round(wire_ms, 3),
                         "wire_priced_as": "one record per link at the nominal 153 GB/s of an xGMI link, all seven links at once: a lower bound, not a measurement"},
            "kernels_of_rank0_ms": {k: round(v, 3) for k, v in per_kernel.items()},
            "what": "host wall clock around the engine's staged calls, one GPU, best of %d repetitions; the all-gather itself is not run "
                    "(one process): priced from the record size" % reps}


def weak_budget_leg(ctx, synth, Dm, args, n_ranks, device):
    """weak scaling on one GPU: n_ranks shards of args.genomes genomes each (genomes r * G .. of the same pan-genome); every shard is counted
    and leaves its exchange record in ONE payload; rank 0's pass -- partition, local dictionary, export, global dictionary from the
    payload of all ranks, fill -- is timed with its batch kept resident.  Best of the repetitions; the all-gather itself is priced."""
    import torch
    filt = not args.keep_singletons
    words = 1 if args.k <= 32 else 2
    G = args.genomes
    n_locals, bbs = [], []
    b0, payload = None, None
    # The records need ONE layout (the largest dictionary of any rank): it is fixed after the first shard with 10 % of head room (the
    # shards are equally large samples of one pan-genome) -- every later shard is counted, writes its record and is freed at once:
    # eight resident 1000-genome batches would not fit HBM.
    for r in range(n_ranks):
        b = make_batch(ctx, synth, "P", r * G, G, args.genome_len)
        b.partition(args.k, args.abundance_min)
        n_locals.append(b.local_dict())
        bbs.append(b.bucket_bits)
        if r == 0:
            b0 = b
            n_max = int(n_locals[0] * 1.1) + 1024
            flags_off, boff_off, stride = b0.exchange_layout(n_max, words, bbs[0] & 0xff)
            payload = torch.empty(n_ranks * stride, dtype=torch.uint8, device=device)
        if n_locals[r] > n_max or bbs[r] != bbs[0]:
            raise RuntimeError("weak budget: shard %d needs another record layout (%d entries, bucket code %#x)" % (r, n_locals[r], bbs[r]))
        b.export_dict_ordered(payload.data_ptr() + r * stride, flags_off, boff_off)
        if r:
            b.free()
    best = {}
    n_cols = 0
    for rep in range(max(2, args.steps) + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b0.partition(args.k, args.abundance_min)
        t1 = time.perf_counter()
        b0.local_dict()
        t2 = time.perf_counter()
        b0.export_dict_ordered(payload.data_ptr(), flags_off, boff_off)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        n_cols = b0.set_global_dict_gathered(payload.data_ptr(), n_max, n_locals, bbs, filt, my_rank=0)
        t4 = time.perf_counter()
        m = b0.fill()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        m.free()
        if rep:
            for key, v in (("partition", t1 - t0), ("local_dict", t2 - t1), ("export_record", t3 - t2), ("global_dict_from_payload", t4 - t3), ("fill", t5 - t4)):
                best[key] = min(best.get(key, 1e30), v * 1e3)
    b0.free()
    del payload
    rank_ms = sum(best.values())
    wire_ms = stride / 153e9 * 1e3
    return {"ranks": n_ranks, "genomes_per_rank": G, "genomes_total": G * n_ranks, "columns": int(n_cols), "stage_ms_rank0": {k: round(v, 3) for k, v in best.items()},
            "rank_ms": round(rank_ms, 3), "exchange": {"record_bytes": int(stride), "wire_ms_lower_bound": round(wire_ms, 3)},
            "genomes_per_s_bound": round(G * n_ranks / ((rank_ms + wire_ms) * 1e-3), 1),
            "what": "rank 0's pass of a weak-scaling run (every rank its own %d genomes of one pan-genome), one GPU, best of the repetitions; the all-gather is priced, "
                    "not run" % G}


def realistic_leg(D, Dm, ctx, synth, args):
    """the headline pass over the same strains given as real assemblies (own contigs per genome, shuffled, random strands, indels)"""
    b = make_batch(ctx, synth, "X", 0, args.genomes, args.genome_len)
    el, n_cols, n_rows, kernels, _ = matrix_leg(D, Dm, ctx, args, b, args.steps, 1)
    occ, syms = b.n_occurrences, b.n_symbols
    ctx.set_option("memo_stats", 1)            # one more pass, untimed, with dict_build's memo counters on
    try:
        m = b.run(args.k, args.abundance_min, not args.keep_singletons)
        memo = b.memo_stats()
        m.free()
    finally:
        ctx.set_option("memo_stats", -1)
    b.free()
    return {"workload": "%d x %d bp realistic assemblies (synth.realistic, seed 1234: the headline's strains + 1 indel site per 10 kbp, 20-100 contigs "
                        "per genome cut at its own places, shuffled, random strands), k=%d, %s" % (
                            args.genomes, args.genome_len, args.k, "singleton filter" if not args.keep_singletons else "singletons kept"),
            "ms_per_step": round(1000 * el / args.steps, 3), "kmers_per_s": round(occ * args.steps / el, 1), "bases_per_s": round(syms * args.steps / el, 1),
            "columns": int(n_cols), "record_memo": memo, "roofline": roofline_of(kernels, leg_traffic("realistic", args, args.genomes), occurrences=occ, leg="realistic"),
            "kernels": kernels}


def c5_leg(D, Dm, ctx, synth, args, orc):
    """BASELINE configs[4]: 500 genomes, k = 63 (two-word k-mers), singletons kept, gzip-5 .kover"""
    import copy
    import tempfile
    import numpy as np
    from importlib import import_module
    kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
    a5 = copy.copy(args)
    a5.k, a5.keep_singletons, a5.abundance_min = 63, True, 1
    n = min(500, args.genomes)
    kept = {}
    check = sorted(set(g for g in (3, n // 5, n // 2 + 7, n - 1) if 0 <= g < n))
    b = make_batch(ctx, synth, "P", 0, n, args.genome_len, keep=lambda i, g: kept.__setitem__(i, g) if i in check else None)
    el, n_cols, n_rows, kernels, _ = matrix_leg(D, Dm, ctx, a5, b, args.steps, 1)
    occ = b.n_occurrences
    out = {"workload": "%d x %d bp pan-genome (mode P), k=63, singletons kept" % (n, args.genome_len),
           "ms_per_step": round(1000 * el / args.steps, 3), "kmers_per_s": round(occ * args.steps / el, 1), "columns": int(n_cols),
           "roofline": roofline_of(kernels, leg_traffic("c5", args, n), occurrences=occ, words=2, leg="c5"), "kernels": kernels}
    m = b.run(63, 1, False)
    # some genomes' columns against the CPU restatement: with the singletons kept, exactly a genome's own k-mers carry its bit
    ok, n_checked = True, 0
    mk = m.kmers()
    md = m.data()
    for g in check:
        km, _, _ = orc.count_genome([kept[g].tobytes()], 63, 1)
        mine = (md[g // 64] >> np.uint64(63 - g % 64)) & np.uint64(1) == 1
        got = mk[mine]
        ok = ok and bool(got.shape == km.shape and np.array_equal(got, km))
        n_checked += km.shape[0]
    out["bit_exact_sample"] = ok
    out["bit_exact_what"] = "for genomes %s: the columns that carry the genome's bit = the CPU restatement's k-mer set of that genome (%d 63-mers in all)" % (check, n_checked)
    del mk
    d = tempfile.mkdtemp(prefix="grm_c5_")
    try:
        path = os.path.join(d, "C5.kover")
        ids = ["g%05d" % i for i in range(n)]
        t0 = time.perf_counter()
        kd.write_header(path, "contigs", "synthetic", None, None, 5, ids, None, None, None, "nothing")
        m.write_kover_h5(path, 5, 100000)
        out["kover_gzip5"] = {"seconds": round(time.perf_counter() - t0, 3), "bytes": os.path.getsize(path)}
        # the file back through libhdf5's own inflate: the matrix the device deflated chunk by chunk
        back = kd.KoverDatasetReader(path).kmer_matrix
        out["kover_gzip5"]["read_back_equal"] = bool(back.shape == md.shape and np.array_equal(back, md))
        out["bit_exact_sample"] = bool(out["bit_exact_sample"] and out["kover_gzip5"]["read_back_equal"])
        del back, md
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    m.free()
    b.free()
    return out


def _reads_fastq(torch, device, seq_u8, n_reads, read_len, seed):
    """4-line FASTQ image (numpy uint8) of n_reads reads of read_len from seq_u8 with 0.5 % substitution errors.  Built on the GPU
    with torch (generating 1 GB per genome with numpy would take longer than every timed leg together); not timed."""
    import numpy as np
    seq = torch.from_numpy(np.ascontiguousarray(seq_u8)).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    width = 3 + read_len + 3 + read_len + 1
    rec = torch.empty((n_reads, width), dtype=torch.uint8, device=device)
    rec[:, 0:3] = torch.tensor(list(b"@r\n"), dtype=torch.uint8, device=device)
    rec[:, 3 + read_len:6 + read_len] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=device)
    rec[:, 6 + read_len:6 + 2 * read_len] = ord("I")
    rec[:, -1] = 10
    ar = torch.arange(read_len, device=device)
    for a in range(0, n_reads, 1 << 20):
        b = min(n_reads, a + (1 << 20))
        st = torch.randint(0, seq.numel() - read_len, (b - a,), generator=gen, device=device)
        rec[a:b, 3:3 + read_len] = seq[st[:, None] + ar[None, :]]
    n_err = int(n_reads * read_len * 0.005)
    r = torch.randint(0, n_reads, (n_err,), generator=gen, device=device)
    c = torch.randint(0, read_len, (n_err,), generator=gen, device=device)
    rec[r, 3 + c] = acgt[torch.randint(0, 4, (n_err,), generator=gen, device=device)]
    out = rec.cpu().numpy().reshape(-1)
    del rec, seq
    return out


def c4_leg(D, ctx, synth, args, orc):
    """BASELINE configs[3] at its real depth for a few genomes: paired-end-like 150 bp reads at 100x with errors, FASTQ, k = 21,
    abundance-min 2 -- the counting stage (parse -> partition -> dedup + count + filter), then one genome's solid set against
    the CPU restatement (counted chunk-wise on the host threads; also the leg's CPU baseline)"""
    import copy
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    n_g, cov, rl, k, amin = 8, 100, 150, 21, 2
    torch = D.torch
    pg = synth.PanGenome(genome_len=args.genome_len, seed=1234)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    b = ctx.batch(n_g)
    first = None
    for g in range(n_g):
        fa = pg.genome(g)
        seq = fa[np.isin(fa, acgt)][:args.genome_len]
        img = _reads_fastq(torch, D.device, seq, args.genome_len * cov // rl, rl, 7700 + g)
        if g == 0:
            first = img
        b.add_array(g, img)
        del img
    torch.cuda.empty_cache()
    b.upload()
    a4 = copy.copy(args)
    a4.k, a4.abundance_min = k, amin
    el, kernels = count_leg(D, ctx, a4, b, max(1, min(args.steps, 2)), 1)
    occ, syms, in_bytes = b.n_occurrences, b.n_symbols, b.input_bytes
    steps = max(1, min(args.steps, 2))
    out = {"workload": "%d genomes x %d bp, %d bp reads at %dx with 0.5 %% substitution errors (%.1f GB of FASTQ resident), k=%d, abundance-min %d: "
                       "counting stage" % (n_g, args.genome_len, rl, cov, in_bytes / 1e9, k, amin),
           "ms_per_step": round(1000 * el / steps, 3), "kmers_per_s": round(occ * steps / el, 1), "bases_per_s": round(syms * steps / el, 1),
           "roofline": roofline_of(kernels, leg_traffic("c4", args, n_g), leg="c4"), "kernels": kernels}
    # genome 0: solid k-mers and their counts
    s0 = b.genome_set(0)
    gk, gc = s0.kmers()[:, 0], s0.counts().astype(np.int64)
    s0.free()
    b.free()
    # CPU: the FASTQ image cut at record boundaries into one chunk per thread, each counted with abundance-min 1
    cores = min(os.cpu_count() or 1, 32)
    rec_bytes = 3 + rl + 3 + rl + 1
    n_rec = first.size // rec_bytes
    cuts = [(n_rec * i // cores) * rec_bytes for i in range(cores + 1)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        parts = list(pool.map(lambda i: orc.count_genome([first[cuts[i]:cuts[i + 1]].tobytes()], k, 1), range(cores)))
    cpu_s = time.perf_counter() - t0
    total = np.zeros(gk.shape[0], dtype=np.int64)
    rest, ok, cpu_occ = [], True, 0
    for km, ct, nocc in parts:
        cpu_occ += nocc
        km = km[:, 0]
        at = np.searchsorted(gk, km)
        at[at >= gk.shape[0]] = gk.shape[0] - 1 if gk.shape[0] else 0
        hit = gk[at] == km if gk.shape[0] else np.zeros(km.shape, bool)
        total += np.bincount(at[hit], weights=ct[hit].astype(np.float64), minlength=gk.shape[0]).astype(np.int64)
        ok = ok and bool((ct[~hit] == 1).all())              # a k-mer outside the solid set occurs once ...
        rest.append(km[~hit])
    rest = np.sort(np.concatenate(rest)) if rest else np.zeros(0, np.uint64)
    ok = ok and bool((rest[1:] != rest[:-1]).all())          # ... in ONE chunk only
    ok = ok and bool(np.array_equal(total, gc)) and bool((gc >= amin).all())
    out["bit_exact_sample"] = ok
    out["bit_exact_what"] = ("genome 0: every solid k-mer's count = the sum of its counts over %d CPU-counted chunks of the same FASTQ image, "
                             "and every other k-mer of the image occurs exactly once (%d solid 21-mers)" % (cores, gk.shape[0]))
    out["cpu_baseline"] = {"value": round(cpu_occ / cpu_s, 1), "unit": "k-mers/s", "cores": cores, "kind": "port",
                           "sample": "genome 0 of the leg (%.2f GB FASTQ) counted in %d chunks on %d threads by the CPU restatement: %.1f s"
                                     % (first.size / 1e9, cores, cores, cpu_s)}
    return out


def kover_equals(reader, res, k):
    """the three datasets the GPU path appended to the .kover, read back, against the CPU restatement's result (dict from
    oracle_ctypes: "kmers" (U, words) ascending, "matrix" (rows, U)); -> True / False"""
    import numpy as np
    want_k, want_m = res["kmers"], res["matrix"]
    with reader._open() as f:
        seqs = f.read("kmer_sequences")
        if len(seqs) != want_k.shape[0]:
            return False
        # S<k> letters -> 2-bit words (A=0 C=1 T=2 G=3, first base most significant), most significant word first
        letters = np.frombuffer(seqs.tobytes(), dtype=np.uint8).reshape(len(seqs), k)
        words = want_k.shape[1]
        got = np.zeros((len(seqs), words), dtype=np.uint64)
        for j in range(k):
            w = words - 1 - (k - 1 - j) // 32
            got[:, w] = (got[:, w] << np.uint64(2)) | ((letters[:, j] >> np.uint8(1)) & np.uint8(3)).astype(np.uint64)
        if not np.array_equal(got, want_k):
            return False
        del got, letters, seqs
        col = f.read("kmer_by_matrix_column")
        if col.shape[0] != want_k.shape[0] or not np.array_equal(col.astype(np.uint64), np.arange(want_k.shape[0], dtype=np.uint64)):
            return False
        m = f.read("kmer_matrix")
        return m.shape == want_m.shape and m.dtype == np.uint64 and bool(np.array_equal(m, want_m))


def host_floors(paths, kover, gzip):
    """SURVEY 8(d) host-side floors of the end-to-end span, measured on this box: FASTA bytes / disk read bandwidth (page cache
    dropped per file with POSIX_FADV_DONTNEED), H2D bytes / PCIe bandwidth (pinned buffer), matrix bytes / zlib throughput x cores"""
    import zlib
    import numpy as np
    out = {}
    try:
        total = 0
        for p in paths:
            fd = os.open(p, os.O_RDONLY)
            try:
                os.fsync(fd)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            finally:
                os.close(fd)
        t0 = time.perf_counter()
        from concurrent.futures import ThreadPoolExecutor

        def rd(p):
            n = 0
            with open(p, "rb", buffering=0) as f:
                while True:
                    b = f.read(1 << 22)
                    if not b:
                        return n
                    n += len(b)
        with ThreadPoolExecutor(max_workers=16) as pool:
            total = sum(pool.map(rd, paths))
        dt = time.perf_counter() - t0
        out["disk_read"] = {"bytes": total, "GBps": round(total / dt / 1e9, 2), "floor_s": round(dt, 3),
                            "how": "all FASTA files read with 16 threads after POSIX_FADV_DONTNEED on each (a tmpfs / overlay may ignore the advice)"}
    except (OSError, AttributeError) as e:
        out["disk_read"] = {"error": str(e)}
    try:
        import torch
        nb = 1 << 30
        hbuf = torch.empty(nb, dtype=torch.uint8).pin_memory()
        dbuf = torch.empty(nb, dtype=torch.uint8, device="cuda")
        dbuf.copy_(hbuf, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dbuf.copy_(hbuf, non_blocking=True)
        torch.cuda.synchronize()
        gbps = 3 * nb / (time.perf_counter() - t0) / 1e9
        total = sum(os.path.getsize(p) for p in paths)
        out["h2d"] = {"GBps": round(gbps, 1), "floor_s": round(total / gbps / 1e9, 3), "how": "1 GiB pinned -> device, 3 copies"}
        del hbuf, dbuf
    except Exception as e:          # noqa: BLE001  (a missing torch build detail must not cost the bench line)
        out["h2d"] = {"error": str(e)}
    try:
        raw = os.path.getsize(kover)
        sample = np.random.default_rng(1).integers(0, 2, size=1 << 22, dtype=np.uint8)      # bits as sparse as a presence row is not: worst case
        sample = np.packbits(sample).tobytes() * 8
        t0 = time.perf_counter()
        z = zlib.compress(sample, gzip)
        mbps = len(sample) / (time.perf_counter() - t0) / 1e6
        cores = os.cpu_count() or 1
        quota = None
        try:                    # what the job may actually use: the cgroup's CPU quota (a 256-thread box gives a one-GPU job 16 CPUs)
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else round(int(q) / int(per), 1)
        except (OSError, ValueError):
            pass
        out["deflate"] = {"MBps_per_core_incompressible": round(mbps, 1), "cores": cores, "cpu_quota": quota, "kover_bytes": raw,
                          "how": "zlib level %d on 4 MiB of random bits (an upper bound on the work per byte; real rows compress faster)" % gzip}
        del z
    except Exception as e:          # noqa: BLE001
        out["deflate"] = {"error": str(e)}
    return out


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
        out["floors"] = host_floors(paths, kover, args.gzip)
        cpu_baseline = None
        if cpu_genomes > 0:
            n = min(cpu_genomes, n_genomes)
            cores = min(os.cpu_count() or 1, 64)
            # the rows of the .kover stand in label order (create.py:334-336): the CPU side takes the genomes in the order the
            # GPU-written file names them, so that the two matrices can be compared word for word
            rd = kd.KoverDatasetReader(kover)
            row_ids = rd.genome_identifiers
            # ... and that order must be the label order Kover prescribes (argsort of the labels, create.py:334-336), worked out HERE
            # from the metadata, not taken from the file: label g % 2 -> the even genomes, then the odd ones, each in list order
            expected_ids = ["g%05d" % g for g in range(n_genomes) if g % 2 == 0] + ["g%05d" % g for g in range(n_genomes) if g % 2 == 1]
            out["row_order_ok"] = bool(row_ids == expected_ids)
            row_paths = [paths[int(g[1:])] for g in row_ids]
            t0 = time.perf_counter()
            bufs = [open(p, "rb").read() for p in row_paths[:n]]
            t_read = time.perf_counter() - t0
            res, count_s, merge_s, occ_cpu = orc.pipeline(bufs, args.k, args.abundance_min, not args.keep_singletons, cores)
            del bufs
            if n == n_genomes:
                out["bit_exact"] = bool(kover_equals(rd, res, args.k) and out["row_order_ok"])
                out["bit_exact_what"] = ("kmer_sequences, kmer_matrix and kmer_by_matrix_column READ BACK from the GPU-written .kover against the "
                                         "CPU restatement's dictionary and packed rows for all %d genomes; the rows' order = the label order worked out from the metadata" % n)
            t1 = time.perf_counter()
            cpu_kover = os.path.join(d, "CPU.kover")
            ids = row_ids[:n]
            labels = np.array([int(g[1:]) % 2 for g in ids], dtype=np.uint8)
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
                            "cpu_quota": (out.get("floors", {}).get("deflate") or {}).get("cpu_quota"),
                            "sample": "%d of the %d genomes of the headline set (same generator): count %.2fs + merge %.2fs on %d threads; CPU "
                                      "restatement of DSK+dsk2kover semantics (the reference binaries are absent)" % (n, n_genomes, count_s, merge_s, cores)}
        return out, cpu_baseline
    finally:
        if not args.tmp:
            shutil.rmtree(d, ignore_errors=True)


def e2e_sharded_leg(D, ctx, synth, args, n_genomes):
    """N > 1: the e2e span over the ranks (multi_gpu.from_contigs_sharded, what `GRM_DEVICES=... kover dataset create` runs): every rank
    writes, reads and uploads the files of ITS word-row block, one dictionary all-gather, every rank deflates the chunks of its rows
    on its device, rank 0 appends.  Wall clock between two barriers; a sample of the file against the CPU restatement."""
    import shutil
    import tempfile
    import numpy as np
    from importlib import import_module
    mg = import_module("genomic-resistance-mapping-grm-_amd.multi_gpu")
    kd = import_module("genomic-resistance-mapping-grm-_amd.kover_dataset")
    Dm = import_module("genomic-resistance-mapping-grm-_amd.distributed")
    d = args.tmp or os.path.join(tempfile.gettempdir(), "grm_e2e_ranks_%s" % os.environ.get("MASTER_PORT", "0"))
    os.makedirs(d, exist_ok=True)
    pg = synth.PanGenome(genome_len=args.genome_len, seed=1234)
    # rows stand in label order (create.py:334-336): labels g % 2 -> the even genomes first; every rank writes the files of its block
    order = [g for g in range(n_genomes) if g % 2 == 0] + [g for g in range(n_genomes) if g % 2 == 1]
    a, b = Dm.shard_genomes(n_genomes, D.world)[D.rank]
    for g in order[a:b]:
        pg.genome(g).tofile(os.path.join(d, "g%05d.fna" % g))
    data, md = os.path.join(d, "paths.tsv"), os.path.join(d, "md.tsv")
    if D.rank == 0:
        open(data, "w").writelines("g%05d\t%s\n" % (g, os.path.join(d, "g%05d.fna" % g)) for g in range(n_genomes))
        open(md, "w").writelines("g%05d\t%d\n" % (g, g % 2) for g in range(n_genomes))
    R = mg.Ranks.from_initialized(D.local_rank)
    kover = os.path.join(d, "DATASET.kover")
    out = None
    try:
        D.sync()
        t0 = time.perf_counter()
        n_cols = mg.from_contigs_sharded(ctx, R, data, kover, args.k, not args.keep_singletons, "synthetic phenotype", md, args.gzip, spool_dir=d)
        D.sync()
        wall = D.max_f(time.perf_counter() - t0)
        if D.rank == 0:
            out = {"seconds": round(wall, 3), "genomes": n_genomes, "ranks": D.world, "columns": int(n_cols), "genomes_per_min": round(n_genomes / wall * 60, 1),
                   "kover_bytes": os.path.getsize(kover), "gzip": args.gzip,
                   "span": "FASTA files (page cache, every rank its own block) -> label-sorted header -> sharded engine pass with ONE dictionary "
                           "all-gather -> per-rank device deflate of the rank's word-rows -> rank 0 appends kmer_sequences / kmer_matrix / "
                           "kmer_by_matrix_column (dataset/create.py:311-390 over %d ranks)" % D.world}
            # sample: two genomes' columns of the written file against the CPU restatement's k-mer sets of those genomes
            from oracle import oracle_ctypes as orc
            rd = kd.KoverDatasetReader(kover)
            ids = rd.genome_identifiers
            ok = ids == ["g%05d" % g for g in order]
            with rd._open() as f:
                seqs = f.read("kmer_sequences")                     # fixed-width byte strings, one per column
                mat = f.read("kmer_matrix")
            for row in (0, n_genomes - 1):
                g = order[row]
                km, _, _ = orc.count_genome([pg.genome(g).tobytes()], args.k, 1)
                mine = (mat[row // 64] >> np.uint64(63 - row % 64)) & np.uint64(1) == 1
                # the letters of the marked columns -> 2-bit words (A=0 C=1 T=2 G=3, first base most significant, most significant word first)
                letters = np.frombuffer(seqs[mine].tobytes(), dtype=np.uint8).reshape(-1, args.k)
                words = km.shape[1]
                got = np.zeros((letters.shape[0], words), dtype=np.uint64)
                for j in range(args.k):
                    w = words - 1 - (args.k - 1 - j) // 32
                    got[:, w] = (got[:, w] << np.uint64(2)) | ((letters[:, j] >> np.uint8(1)) & np.uint8(3)).astype(np.uint64)
                void = np.dtype((np.void, 8 * words))
                gv, wv = np.ascontiguousarray(got).view(void).reshape(-1), np.ascontiguousarray(km).view(void).reshape(-1)
                # with the singleton filter the file holds the genome's k-mers that another genome carries too: a subset, and every
                # column the genome is marked in must be one of its k-mers
                sub = bool(np.isin(gv, wv).all())
                ok = ok and sub and (gv.shape[0] == wv.shape[0] if args.keep_singletons else gv.shape[0] > 0.9 * wv.shape[0])
            out["bit_exact_sample"] = bool(ok)
            out["bit_exact_what"] = "row order = label order; the columns carrying the first and the last row's bit are k-mers of those genomes (CPU restatement)"
    finally:
        D.sync()
        if not args.tmp and D.rank == 0:
            shutil.rmtree(d, ignore_errors=True)
    return out


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
    if args.only not in (None, "headline"):
        args.head_genomes = min(args.genomes, 16)      # profiling one of the other legs: the headline shrinks to a token set
    else:
        args.head_genomes = args.genomes
    strong = world > 1 and not args.weak_only
    if strong:
        base, stop = Dm.shard_genomes(args.head_genomes, world)[rank]
        n_mine, genomes_total = stop - base, args.head_genomes
    else:
        base, n_mine, genomes_total = rank * args.head_genomes, args.head_genomes, args.head_genomes * world
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
            out["survey_model"] = {"bytes_per_kmer": bpk, "model_GBps_equivalent": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                                   "times_peak": round(gbps / HBM_PEAK_GBS, 3),
                                   "what": "SURVEY 8(d)'s 1 + 7W bytes per occurrence over the pass's duration: above 1 because this design moves ~11 B per "
                                           "occurrence, not 57 -- NOT a roofline fraction (roofline.frac and pass_traffic are)"}
            tj = committed_traffic(args, n_mine) if world == 1 and args.mode == "P" else None
            if tj is not None:
                try:
                    total = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))["sum_over_all_kernels_per_pass"]
                    out["pass_traffic"] = {"bytes_per_pass": total, "achieved": round(total / (elapsed / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(total / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                           "what": "counter traffic of ALL kernels of a pass (committed PMC passes of this workload) over this run's ms per pass"}
                except (OSError, ValueError, KeyError):
                    pass
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

    solo = world == 1 and rank == 0 and args.stage == "matrix" and args.mode == "P"
    if solo and args.rank_budget > 1 and args.only in (None, "headline") and args.genomes >= 64 * args.rank_budget and args.k <= 64:
        rb = rank_budget_leg(ctx, synth, Dm, args, args.rank_budget, D.device)
        wire = rb["exchange"]["wire_ms_lower_bound"]
        rb["speedup_bound_vs_1gpu"] = round(out["ms_per_step"] / (rb["rank_ms"] + wire), 2)
        # Amdahl: what of a rank's pass does not shrink with its shard.  Every rank needs a column for each of its local entries and
        # ranks of a pan-genome hold nearly all k-mers, so union + sort + fill (and the insertions of the local dictionary) are
        # dictionary-sized on every rank; only parse / partition / the record look-ups are genome-sized.
        u_sized = sum(rb["kernels_of_rank0_ms"].get(k, 0.0) for k in ("dict_union", "dict_sort", "dict_entry_cols", "matrix_fill"))
        rb["dictionary_sized_kernels_ms"] = round(u_sized, 3)
        rb["note"] = ("an upper bound of the N-GPU speed-up over this run's one-GPU pass: slowest rank's pass + the wire's lower bound; %.2f ms of the "
                      "rank pass are dictionary-sized kernels every rank repeats (union, sort, columns, fill) -- with ONE all-gather every rank "
                      "builds the whole dictionary, so that part does not shrink with the shard" % u_sized)
        out["rank_budget"] = rb
        if args.rank_budget_curve:
            curve = {"1": 1.0}
            for nr in (2, 4, args.rank_budget):
                r2 = rb if nr == args.rank_budget else rank_budget_leg(ctx, synth, Dm, args, nr, D.device)
                curve[str(nr)] = {"rank_ms": r2["rank_ms"], "wire_ms_lower_bound": r2["exchange"]["wire_ms_lower_bound"],
                                  "speedup_bound": round(out["ms_per_step"] / (r2["rank_ms"] + r2["exchange"]["wire_ms_lower_bound"]), 2)}
            out["rank_budget_curve"] = curve
    if solo and args.weak_budget > 1 and args.only in (None, "headline") and args.k <= 64:
        wb = weak_budget_leg(ctx, synth, Dm, args, args.weak_budget, D.device)
        wb["throughput_bound_vs_1gpu"] = round(wb["genomes_per_s_bound"] / (args.genomes / (out["ms_per_step"] * 1e-3)), 2)
        out["weak_budget"] = wb
    # ---- the same strains as real assemblies ----
    if solo and not args.no_realistic and args.only in (None, "realistic"):
        out["realistic"] = realistic_leg(D, Dm, ctx, synth, args)
        out["realistic"]["vs_headline_ms"] = round(out["realistic"]["ms_per_step"] / out["ms_per_step"], 3)
    # ---- BASELINE configs[4] and configs[3] ----
    if solo and not args.no_c5 and args.only in (None, "c5") and args.k == 31:
        from oracle import oracle_ctypes as orc5      # checker only: after the leg's timed region
        out["c5"] = c5_leg(D, Dm, ctx, synth, args, orc5)
    if solo and not args.no_c4 and args.only in (None, "c4") and args.k == 31:
        from oracle import oracle_ctypes as orc4
        out["c4"] = c4_leg(D, ctx, synth, args, orc4)

    # ---- random-ACGT counting stage ----
    if not args.no_random and args.only in (None, "random") and not (args.mode == "R" and args.stage == "count"):
        if world > 1:
            rbase, rstop = Dm.shard_genomes(args.genomes, world)[rank]
        else:
            rbase, rstop = 0, args.genomes
        rb = make_batch(ctx, synth, "R", rbase, rstop - rbase, args.genome_len)
        r_el, r_kernels = count_leg(D, ctx, args, rb, args.steps, 1)
        r_occ, r_syms = D.sum_i(rb.n_occurrences), D.sum_i(rb.n_symbols)
        r_ok = None
        if rank == 0 and args.k <= 32:
            # two genomes' counted sets against the CPU restatement (after the timed region)
            from oracle import oracle_ctypes as orcr
            r_ok = True
            for g in (0, rstop - rbase - 1):
                km, ct, nocc = orcr.count_genome([synth.random_genome(rbase + g, genome_len=args.genome_len, seed=1234).tobytes()], args.k, args.abundance_min)
                sg = rb.genome_set(g)
                r_ok = r_ok and sg.occurrences == nocc and sg.kmers().shape == km.shape and bool((sg.kmers() == km).all()) and bool((sg.counts() == ct).all())
                sg.free()
        rb.free()
        if rank == 0:
            out["random_acgt"] = {
                "workload": "%d independent uniform-ACGT genomes x %d bp (seed 1234+i)%s, k=%d: parse -> partition -> per-genome dedup + count, "
                            "matrix not materialised" % (args.genomes, args.genome_len, ", split over %d ranks" % world if world > 1 else "", args.k),
                "bases_per_s": round(r_syms * args.steps / r_el, 1), "kmers_per_s": round(r_occ * args.steps / r_el, 1),
                "ms_per_step": round(1000 * r_el / args.steps, 3), "bit_exact_sample": r_ok,
                "bit_exact_what": "counted sets (k-mers and counts) of the first and the last genome of rank 0 = the CPU restatement's",
                "roofline": roofline_of(r_kernels, leg_traffic("random_acgt", args, args.genomes) if world == 1 else None, leg="random_acgt"), "kernels": r_kernels}

    # ---- end to end over the ranks (N > 1) ----
    if world > 1 and args.stage == "matrix" and args.mode == "P" and not args.no_e2e and args.only in (None, "headline") and args.k <= 64:
        e2e_n = e2e_sharded_leg(D, ctx, synth, args, args.genomes)
        if rank == 0:
            out["e2e"] = e2e_n

    # ---- end to end + CPU restatement (N = 1) ----
    if world == 1 and rank == 0 and args.stage == "matrix" and args.only in (None, "headline"):
        cpu_genomes = args.genomes if args.cpu_genomes < 0 else args.cpu_genomes
        if not args.no_e2e and args.mode == "P":
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
    failed = False
    if rank == 0:
        out.setdefault("cpu_baseline", None)
        # "bit-exact" is part of the metric: any leg that compared its result with the CPU restatement and found a difference fails the run
        checks = {"e2e": (out.get("e2e") or {}).get("bit_exact", (out.get("e2e") or {}).get("bit_exact_sample")), "c5": (out.get("c5") or {}).get("bit_exact_sample"),
                  "c4": (out.get("c4") or {}).get("bit_exact_sample"), "random_acgt": (out.get("random_acgt") or {}).get("bit_exact_sample")}
        out["bit_exact"] = {k: v for k, v in checks.items() if v is not None}
        failed = any(v is False for v in checks.values())
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        D.dist.destroy_process_group()
    if failed:
        sys.stderr.write("bench.py: a leg's result differs from the CPU restatement (see bit_exact)\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
