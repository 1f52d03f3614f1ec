#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer-matrix hot path (BASELINE.json).

A "step" = one pass of the hot path (FASTA bytes resident in HBM -> packed symbols ->
hash-partitioned canonical k-mers -> dictionary -> genome x k-mer presence matrix) over one
batch of synthetic genomes.  Workload at N=1: BASELINE.json configs[1]
(1000 x 5 Mbp synthetic contig genomes, k=31, pan-genome mode, singleton filter on).
N>1: one process per GPU, each with its own 1000-genome shard (weak scaling: per-GPU work fixed; a
rank's block is padded to whole word-rows) and the one dictionary all-gather over RCCL between the
local-dictionary and fill stages.  `--total-genomes T` is the strong-scaling form of BASELINE C3:
one T-genome set split over the ranks in blocks of whole word-rows.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

# algorithmic bytes per unit for each kernel (DESIGN.md "Kernels"); unit = what TimeScope counts
ALGO_BYTES = {
    "parse_summarize": 1.0,          # per input byte: read 1
    "parse_pack": 1.0 + 0.375,       # per input byte: read 1, write 2+1 bits per symbol (<= byte count)
    "kmer_hist": 0.375,              # per symbol: read packed stream
    "kmer_scatter_l1": 0.375 + 8.0,  # per k-mer occurrence: read packed stream, write one u64 key
    "kmer_scatter_l2": 16.0,         # per key: read 8, write 8 (second radix level)
    "bucket_dedup": 16.0,            # per key: read 8, write <= 8
    "dict_build": 8.0 + 2.0,         # per key: read 8, write the 2-byte slot id (dictionary output is U-sized)
    "matrix_fill": 2.0,              # per key: read the 2-byte slot id (+ rows x U x 8 written, added per launch below)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1000, help="genomes per GPU (weak scaling, the default)")
    ap.add_argument("--total-genomes", type=int, default=0,
                    help="strong-scaling form (BASELINE C3): ONE set of this many genomes split over the ranks in "
                         "blocks of whole word-rows (1000 over 8 GPUs = 128,...,104); overrides --genomes")
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--mode", default="P", choices=["P", "R"])
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--abundance-min", type=int, default=1)
    ap.add_argument("--keep-singletons", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=384, help="genomes in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--opt", action="append", default=[], help="engine tuning knob name=value (grm_set_option)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import grm_amd
    from importlib import import_module
    synth = import_module("genomic-resistance-mapping-grm-_amd.synth")
    D = import_module("genomic-resistance-mapping-grm-_amd.distributed")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # GRM_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend (collectives staged through
    # host memory) -- lets the N>1 code path run on a one-GPU box; never used for reported numbers
    rehearsal = os.environ.get("GRM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    filt = not args.keep_singletons

    # ---- synthetic inputs: generate, hand to the engine, upload once (resident in HBM) ----
    t0 = time.time()
    ctx = grm_amd.Context(local_rank)
    for kv in args.opt:
        name, val = kv.split("=")
        ctx.set_option(name, int(val))
    strong = args.total_genomes > 0
    if strong:
        base, stop = D.shard_genomes(args.total_genomes, world)[rank]
        args.genomes = stop - base
    else:
        base = rank * args.genomes
    batch = ctx.batch(args.genomes)
    if args.mode == "P":
        pg = synth.PanGenome(genome_len=args.genome_len, seed=1234)
        gen = lambda i: pg.genome(base + i)
    else:
        gen = lambda i: synth.random_genome(base + i, genome_len=args.genome_len, seed=1234)
    cpu_sample = []
    for i in range(args.genomes):
        g = gen(i)
        if rank == 0 and world == 1 and i < args.cpu_sample:
            cpu_sample.append(g.tobytes())
        batch.add_array(i, g)
    batch.upload()
    setup_s = time.time() - t0

    def step():
        if world == 1:
            return batch.run(args.k, args.abundance_min, filt)
        return D.sharded_step(batch, args.k, args.abundance_min, filt, device)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()


    for _ in range(args.warmup):
        m = step()
        m.free()
    ctx.timing(True)
    ctx.timing_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = step()
        n_cols, n_rows = m.n_kmers, m.n_rows
        m.free()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    occ = batch.n_occurrences
    if world > 1:
        t = torch.tensor([occ], dtype=torch.int64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        occ_total = int(t.item())
        t = torch.tensor([batch.input_bytes], dtype=torch.int64, device="cpu" if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        bytes_total = int(t.item())
    else:
        occ_total = occ
        bytes_total = batch.input_bytes
    genomes_total = args.total_genomes if strong else args.genomes * world

    # ---- per-kernel device times (HIP events on the engine's stream, timed region only) ----
    per = {}
    for name, ms, units in ctx.timings():
        d = per.setdefault(name, {"ms": 0.0, "launches": 0, "units": 0})
        d["ms"] += ms
        d["launches"] += 1
        d["units"] += units
    kernels = {}
    for name, d in per.items():
        avg_ms = d["ms"] / d["launches"]
        e = {"avg_ms": round(avg_ms, 4), "launches": d["launches"]}
        if name in ALGO_BYTES and avg_ms > 0:
            byts = ALGO_BYTES[name] * d["units"] / d["launches"]
            if name == "matrix_fill":
                byts += 8.0 * n_cols * n_rows
            e["algo_GBps"] = round(byts / (avg_ms * 1e-3) / 1e9, 1)
            e["algo_bytes"] = byts
        kernels[name] = e
    dom = max((n for n in kernels if "algo_GBps" in kernels[n]), key=lambda n: kernels[n]["avg_ms"] * kernels[n]["launches"], default=None)
    roofline = None
    # HBM traffic of the dominant kernel: from the committed PMC passes (profiles/hbm_traffic.json),
    # only when this run is the workload those passes were taken on
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        w = tj["workload"]
        if (w["genomes"], w["genome_len"], w["mode"], w["k"]) == (args.genomes, args.genome_len, args.mode, args.k) and not args.opt:
            traffic = tj["bytes_per_launch"].get(dom)
    except (OSError, ValueError, KeyError):
        pass
    if dom:
        a = kernels[dom]["algo_GBps"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_ms": kernels[dom]["avg_ms"]}

    # ---- CPU baseline: the oracle (a port), bounded sample of the same workload, rank 0, N=1 ----
    cpu_baseline = None
    if rank == 0 and world == 1 and cpu_sample:
        from oracle import oracle_ctypes as orc
        cores = min(os.cpu_count() or 1, 64)
        t0 = time.perf_counter()
        _, count_s, merge_s, occ_cpu = orc.pipeline(cpu_sample, args.k, args.abundance_min, filt, cores)
        cpu_s = time.perf_counter() - t0
        cpu_baseline = {"value": round(occ_cpu / cpu_s, 1), "unit": "k-mers/s", "cores": cores, "kind": "port",
                        "sample": "%d of the %d genomes (same generator), count %.2fs + merge %.2fs; CPU restatement of "
                                  "DSK+dsk2kover semantics (the reference binaries are absent)" % (len(cpu_sample), args.genomes, count_s, merge_s)}

    if rank == 0:
        value = occ_total * args.steps / elapsed
        out = {
            "metric": "k-mers/sec (canonical k-mer occurrences -> presence matrix, whole job)",
            "value": round(value, 1), "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s x %d bp synthetic contig genomes (mode %s, seed 1234), k=%d, abundance-min %d, %s"
                                   % ("%d in total, split in word-row blocks" % args.total_genomes if strong else "%d per GPU" % args.genomes,
                                      args.genome_len, args.mode, args.k, args.abundance_min,
                                      "singleton filter" if filt else "singletons kept"),
                       "genomes_total": genomes_total, "columns": int(n_cols), "word_rows_per_gpu": int(n_rows),
                       "input_bytes_per_gpu": batch.input_bytes, "parallelism": "genome-sharded x%d" % world},
            "genomes_per_min": round(genomes_total * args.steps / elapsed * 60, 1),
            "bases_per_s": round(bytes_total * args.steps / elapsed, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernels,
            "setup_s": round(setup_s, 1),
        }
        print(json.dumps(out))
    batch.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
