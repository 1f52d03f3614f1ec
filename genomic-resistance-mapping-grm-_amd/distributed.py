"""Multi-GPU sharding of the k-mer-matrix path: one process per GPU,
torch.distributed over RCCL ("nccl" backend on ROCm) / xGMI.

Genomes are independent through parse / partition / local dictionary, and again through
the presence-bit fill; the ONE exchange step is the union of the per-rank dictionaries:

    rank r:  local distinct k-mers (uint64; (hi, lo) pairs for 33 <= k <= 64) + flag
             (1 = carried by one local genome, 2 = by several)  n_r entries
    all-gather(n_r)  ->  all-gather of max-padded (keys, flags) buffers
    every rank: same deterministic sort / merge / singleton filter  ->  identical global
             dictionary and column order, no second collective
    rank r fills only its own word-rows; rank blocks are multiples of 64 genomes so a
    rank owns whole uint64 word-rows and the host (or rank 0) just stacks the rows.

The reference's only comparable step is Ray Surveyor's MPI k-mer delivery to owner ranks
under `mpiexec -n 4` (src/app.py:1310); this is a re-design for point-to-point xGMI, not a
translation: one bulk all-gather of sorted-able arrays instead of many ~4 KiB messages.

torch is used for the collective buffers only; the engine sees raw pointers.
"""
import numpy as np


def shard_genomes(n_genomes, world_size, block=64):
    """contiguous blocks of whole word-rows per rank: -> list of (start, stop) per rank.

    1000 genomes / 8 ranks -> 128,128,...,104."""
    n_blocks = (n_genomes + block - 1) // block
    per = (n_blocks + world_size - 1) // world_size
    out = []
    for r in range(world_size):
        a = min(n_genomes, r * per * block)
        b = min(n_genomes, (r + 1) * per * block)
        out.append((a, b))
    return out


def _all_gather(out, inp, group):
    """all_gather_into_tensor; with the gloo backend and device tensors (single-GPU rehearsal of
    the multi-rank path) the exchange is staged through host memory."""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        o = out.cpu()
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def allgather_dict(batch, n_local, device, group=None, words=1):
    """the single data-path collective.  batch: object with export_dict(keys_ptr, flags_ptr).
    -> (keys int64 tensor [n_total, words], flags uint8 tensor [n_total]) on `device`;
    words = 2 for 33 <= k <= 64: (hi, lo) pairs, 16 bytes per k-mer."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    keys = torch.empty((max(1, n_local), words), dtype=torch.int64, device=device)
    flags = torch.empty(max(1, n_local), dtype=torch.uint8, device=device)
    batch.export_dict(keys.data_ptr(), flags.data_ptr())
    if world == 1:
        return keys[:n_local], flags[:n_local]
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([n_local], dtype=torch.int64, device=device)
    _all_gather(counts, mine, group)
    counts_h = counts.cpu().tolist()
    n_max = max(1, max(counts_h))
    kpad = torch.zeros((n_max, words), dtype=torch.int64, device=device)
    fpad = torch.zeros(n_max, dtype=torch.uint8, device=device)
    kpad[:n_local] = keys[:n_local]
    fpad[:n_local] = flags[:n_local]
    kall = torch.empty((world * n_max, words), dtype=torch.int64, device=device)
    fall = torch.empty(world * n_max, dtype=torch.uint8, device=device)
    _all_gather(kall, kpad, group)
    _all_gather(fall, fpad, group)
    ks = [kall[r * n_max: r * n_max + counts_h[r]] for r in range(world)]
    fs = [fall[r * n_max: r * n_max + counts_h[r]] for r in range(world)]
    keys_all, flags_all = torch.cat(ks).contiguous(), torch.cat(fs).contiguous()
    if keys_all.is_cuda:
        # the engine launches on ITS OWN stream: the collective and the concatenation above were
        # only enqueued on torch's stream, so they must have finished before the raw pointers are
        # handed over (RCCL collectives return to the host before the GPU work is done)
        torch.cuda.current_stream(keys_all.device).synchronize()
    return keys_all, flags_all


def sharded_step(batch, k, abundance_min, filter_singleton, device, group=None, stats=None):
    """one pass of the hot path on this rank's genomes; returns the rank's Matrix
    (its word-rows against the GLOBAL dictionary).  stats (optional dict with "bytes", "ms", "calls")
    accumulates what the exchange step received and how long it took on the host clock."""
    import time
    batch.partition(k, abundance_min)
    n_local = batch.local_dict()
    t0 = time.perf_counter()
    keys, flags = allgather_dict(batch, n_local, device, group, words=2 if k > 32 else 1)
    if stats is not None:
        stats["bytes"] += int(keys.numel()) * 8 + int(flags.numel())
        stats["ms"] += (time.perf_counter() - t0) * 1e3
        stats["calls"] += 1
    batch.set_global_dict(keys.data_ptr(), flags.data_ptr(), int(keys.shape[0]), filter_singleton)
    return batch.fill()


def gather_rows(local_rows, device, group=None):
    """stack every rank's word-rows on all ranks (small: rows x U uint64).  local_rows: numpy
    uint64 [r_local, U] -> numpy uint64 [sum r_local, U]"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_rows
    U = local_rows.shape[1]
    nrows = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=device)
    allr = torch.zeros(world, dtype=torch.int64, device=device)
    _all_gather(allr, nrows, group)
    allr_h = allr.cpu().tolist()
    rmax = max(1, max(allr_h))
    pad = torch.zeros((rmax, U), dtype=torch.int64, device=device)
    if local_rows.size:
        pad[: local_rows.shape[0]] = torch.from_numpy(local_rows.view(np.int64)).to(device)
    out = torch.empty((world * rmax, U), dtype=torch.int64, device=device)
    _all_gather(out, pad, group)
    parts = [out[r * rmax: r * rmax + allr_h[r]] for r in range(world)]
    return torch.cat(parts).cpu().numpy().view(np.uint64)
