"""Multi-GPU sharding of the k-mer-matrix path: one process per GPU,
torch.distributed over RCCL ("nccl" backend on ROCm) / xGMI.

Genomes are independent through parse / partition / local dictionary, and again through
the presence-bit fill; the ONE exchange step is the union of the per-rank dictionaries:

    rank r:  local distinct k-mers (uint64; (hi, lo) pairs for 33 <= k <= 64), grouped by hash
             bucket, + flag (1 = carried by one local genome, 2 = by several)  n_r entries
    ONE all-gather over RCCL of one fixed-stride record per rank (keys | flags | bucket offsets | header with n_r and the
             bucket geometry); the stride comes from the previous step's sizes, a rank that outgrew it says so in its header
             and all ranks repeat (exchange_dict)
    every rank: the gathered lists are united bucket by bucket in LDS tables (ranks of a pan-genome
             hold nearly the same k-mers: what is left to sort is the union, not the sum), then the
             same deterministic sort / singleton filter  ->  identical global dictionary and
             column order, no second collective
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


_host_groups = {}


def _host_group(group):
    """process group for host-side integers (gloo): the default group when it is gloo already, else a
    gloo twin of it, created on first use (collectively: every rank reaches this point in the same step)"""
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        return group
    key = id(group)
    if key not in _host_groups:
        _host_groups[key] = dist.new_group(ranks=None if group is None else dist.get_process_group_ranks(group), backend="gloo")
    return _host_groups[key]


HEADER_BYTES = 16          # include/grm_kmer.h: GRM_EXCHANGE_HEADER_BYTES -- n_local u64 | bucket-bits code u32 | magic u32
MAGIC = 0x584d5247
_plans = {}                # per group: (n_cap, bucket bits) the next step lays its record out with


def _grow(n):
    """capacity for a list of n entries: 1/16 of slack, whole KiB of flags"""
    return max(1024, (n + n // 16 + 1023) // 1024 * 1024)


def forget_plans():
    """drop what earlier steps learnt about list sizes (tests; a group that is destroyed)"""
    _plans.clear()


def exchange_dict(batch, n_local, device, group=None, words=1, stats=None):
    """the exchange step: every rank learns every rank's local dictionary in ONE collective -- an all-gather of one fixed-stride
    byte record per rank (keys grouped by hash bucket | flags | bucket offsets | header; layout: grm_exchange_layout).

    The stride must be the same on every rank before any of them knows the others' sizes: it comes from what the PREVIOUS step of
    this group saw (largest list + 1/16, largest bucket count).  Every record ends in a header with its rank's n_local and bucket
    geometry, so after the all-gather every rank knows the same world sizes: if a list did not fit its rank sent the header alone,
    all ranks see that, and all repeat the step with the layout that fits -- no vote, no extra collective otherwise.  Only the very
    first step of a group has nothing to go by and asks for the sizes over the host (gloo) group first.
    -> (payload uint8 tensor [world * stride], n_cap, counts, bucket-bits codes)"""
    import time
    import torch
    import torch.distributed as dist
    t0 = time.perf_counter()
    world = dist.get_world_size(group)
    key = (id(group), words, world)
    plan = _plans.get(key)
    if plan is None:
        mine = torch.tensor([n_local, batch.bucket_bits], dtype=torch.int64)
        sizes = torch.empty(2 * world, dtype=torch.int64)
        dist.all_gather_into_tensor(sizes, mine, group=_host_group(group))
        plan = (_grow(max(sizes[0::2].tolist())), max(v & 0xff for v in sizes[1::2].tolist()))
    while True:
        n_cap, bits = plan
        _, _, stride = batch.exchange_layout(n_cap, words, bits)
        rec = torch.empty(stride, dtype=torch.uint8, device=device)       # padding is never read
        batch.export_dict_record(rec.data_ptr(), n_cap, bits)
        payload = torch.empty(world * stride, dtype=torch.uint8, device=device)
        _all_gather(payload, rec, group)
        # the headers, on the host: the copy is ordered behind the collective on torch's stream and the host waits for it -- the
        # engine (which launches on ITS OWN stream) may take the raw pointer afterwards without another synchronize
        heads = payload.view(world, stride)[:, stride - HEADER_BYTES:].contiguous().cpu().numpy()
        counts = [int(v) for v in heads[:, :8].copy().view(np.uint64).reshape(-1)]
        codes = heads[:, 8:].copy().view(np.uint32).reshape(world, 2)
        if not (codes[:, 1] == MAGIC).all():
            raise RuntimeError("exchange_dict: a record without a header (ranks of different versions?)")
        bbs = [int(v) for v in codes[:, 0]]
        if stats is not None:
            stats["bytes"] += world * stride
            stats["calls"] += 1
        need = (max(counts), max(v & 0xff for v in bbs))
        if need[0] <= n_cap and need[1] <= bits:
            break
        plan = (_grow(need[0]), max(bits, need[1]))
    _plans[key] = (_grow(need[0]), need[1])
    if stats is not None:
        stats["ms"] += (time.perf_counter() - t0) * 1e3
    return payload, n_cap, counts, bbs


def sharded_step(batch, k, abundance_min, filter_singleton, device, group=None, stats=None):
    """one pass of the hot path on this rank's genomes; returns the rank's Matrix
    (its word-rows against the GLOBAL dictionary).  stats (optional dict with "bytes", "ms", "calls")
    accumulates what the exchange step received and how long it took on the host clock."""
    import torch.distributed as dist
    batch.partition(k, abundance_min)
    n_local = batch.local_dict()
    words = (k + 31) // 32
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        import torch
        keys = torch.empty((max(1, n_local), words), dtype=torch.int64, device=device)
        flags = torch.empty(max(1, n_local), dtype=torch.uint8, device=device)
        batch.export_dict(keys.data_ptr(), flags.data_ptr())
        batch.set_global_dict(keys.data_ptr(), flags.data_ptr(), n_local, filter_singleton)
        return batch.fill()
    payload, n_max, counts, bbs = exchange_dict(batch, n_local, device, group, words, stats)
    batch.set_global_dict_gathered(payload.data_ptr(), n_max, counts, bbs, filter_singleton, my_rank=dist.get_rank(group))
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
