"""Multi-GPU plumbing: one process per GPU, queries sharded, index replicated, hits all-gathered.

The query loop of the reference carries no state between queries (motif_both_points.cpp:224), so
queries shard embarrassingly; the only exchange step is the variable-length all-gather of hit
tuples at the end (RCCL has no all-gatherv: counts first, then max-padded payloads).  The same code
runs over gloo on CPU tensors, which is how tests/test_dist_cpu.py covers it without GPUs.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_total, rank, world):
    """Contiguous block partition [lo, hi) of n_total queries for `rank` (sizes differ by <= 1)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allgather_hits(q, ids, table, distance, n_hits, q_offset=0, group=None, force=False):
    """All-gather hit tuples of every rank.

    q, ids, table: int32/uint32-like 1-D tensors (same device), distance: float64, only the first
    n_hits entries are meaningful.  q is local to the rank's shard; q_offset (the shard's first
    global query index) is added so the result is in global query numbering.  Returns
    (q, ids, table, distance) holding all ranks' hits concatenated in rank order, i.e. in the
    reference's global output order when shards are contiguous blocks.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1 and q.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal mode (ranks sharing GPUs): gloo moves host memory
        q, ids, table, distance = q.cpu(), ids.cpu(), table.cpu(), distance.cpu()
    dev = q.device
    q = q[:n_hits].to(torch.int64) + int(q_offset)
    ids = ids[:n_hits].to(torch.int64)
    table = table[:n_hits].to(torch.int64)
    distance = distance[:n_hits]
    if world == 1 and not (force and dist.is_initialized()):
        return q, ids, table, distance
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n_hits], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts_h = counts.cpu().tolist()
    m = max(max(counts_h), 1)
    # one payload: 3 int64 columns + the distance bits
    pack = torch.zeros((4, m), dtype=torch.int64, device=dev)
    pack[0, :n_hits] = q
    pack[1, :n_hits] = ids
    pack[2, :n_hits] = table
    pack[3, :n_hits] = distance.view(torch.int64)
    gathered = torch.empty((world, 4, m), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered.view(world * 4, m), pack, group=group)
    parts = [gathered[r, :, :counts_h[r]] for r in range(world)]
    allp = torch.cat(parts, dim=1)
    return allp[0], allp[1], allp[2], allp[3].view(torch.float64)


def hits_to_numpy(q, ids, table, distance):
    return dict(q=q.cpu().numpy().astype(np.uint32), id=ids.cpu().numpy().astype(np.uint32),
                table=table.cpu().numpy().astype(np.uint32), dist=distance.cpu().numpy())
