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


# Per process group: the padded per-rank capacity m of the payload (grown on demand, kept across
# calls) and the send / receive buffers of that size.  m is derived from EXCHANGED counts only, so it
# is the same on every rank of a group as long as every rank of the group makes the same sequence of
# calls on it (which a collective requires anyway); it is keyed by group so that ranks taking part in
# several groups cannot carry one group's capacity into another's collective.
_STATE = {}


def _group_state(group, dev):
    key = (id(group) if group is not None else 0, str(dev))
    st = _STATE.get(key)
    if st is None:
        st = _STATE[key] = {"m": 4096, "buf": None, "gathered": None}
    return st


def reset_state():
    """Forget capacities and buffers (call after destroying a process group)."""
    _STATE.clear()


def allgather_hits(q, ids, table, distance, n_hits, q_offset=0, group=None, force=False):
    """All-gather hit tuples of every rank.

    q, ids, table: int32/uint32-like 1-D tensors (same device), distance: float64, only the first
    n_hits entries are meaningful.  q is local to the rank's shard; q_offset (the shard's first
    global query index) is added so the result is in global query numbering.  Returns
    (q, ids, table, distance) holding all ranks' hits concatenated in rank order, i.e. in the
    reference's global output order when shards are contiguous blocks.

    One collective per call: every rank contributes one byte buffer [count | q | id | table | dist]
    padded to a common capacity m (RCCL has no all-gatherv); the counts travel in the same buffer,
    so the only host round trip is reading them back afterwards.  m follows the largest count any
    rank has reported (x1.25); a call whose counts exceed it repeats once with a larger m.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world > 1 and q.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal mode (ranks sharing GPUs): gloo moves host memory
        q, ids, table, distance = q.cpu(), ids.cpu(), table.cpu(), distance.cpu()
    dev = q.device
    n_hits = int(n_hits)
    if world == 1 and not (force and dist.is_initialized()):
        # nothing to exchange: the rank's own arrays, as views (no conversion kernels in the step)
        qv = q[:n_hits] if not q_offset else q[:n_hits] + int(q_offset)
        return qv, ids[:n_hits], table[:n_hits], distance[:n_hits]
    q32 = q[:n_hits].view(torch.int32) + int(q_offset)          # global numbering before the exchange
    cols = (q32, ids[:n_hits].view(torch.int32), table[:n_hits].view(torch.int32), distance[:n_hits])
    state = _group_state(group, dev)
    while True:
        # m must be the same on every rank: it is derived from exchanged counts only, never from
        # the local n_hits (a rank with more hits than m sends its count and a truncated payload;
        # every rank then sees that count and repeats the exchange with the same larger m)
        m = state["m"]
        nbytes = 8 + 20 * m
        if state["buf"] is None or state["buf"].numel() != nbytes:
            # (re)allocated only when m grows: nothing is allocated in a steady-state step
            state["buf"] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            state["gathered"] = torch.empty((world, nbytes), dtype=torch.uint8, device=dev)
        buf, gathered = state["buf"], state["gathered"]
        buf[:8].view(torch.int64).fill_(n_hits)
        n_send = min(n_hits, m)
        off = 8
        for c, w in zip(cols, (4, 4, 4, 8)):
            buf[off:off + w * n_send] = c[:n_send].contiguous().view(torch.uint8)
            off += w * m
        dist.all_gather_into_tensor(gathered.view(-1), buf, group=group)
        # the one host round trip of the exchange: the counts decide how much of each record is live
        counts = gathered[:, :8].contiguous().view(torch.int64).view(-1).cpu().tolist()
        if max(counts) <= m:
            break
        state["m"] = (int(max(counts) * 1.25) + 1024 + 1) // 2 * 2  # even: keeps float64 8-aligned
    out = []
    off = 8
    for w, dt in zip((4, 4, 4, 8), (torch.int32, torch.int32, torch.int32, torch.float64)):
        parts = [gathered[r, off:off + w * counts[r]].view(dt) for r in range(world)]
        col = torch.cat(parts) if world > 1 else parts[0].clone()   # copies: the buffer is reused
        out.append(col if dt == torch.float64 else col.to(torch.int64))
        off += w * m
    return out[0], out[1], out[2], out[3]


def table_shard_bounds(L, rank, world):
    """Tables [lo, hi) held by `rank` when the L tables are partitioned over the ranks in contiguous blocks."""
    return shard_bounds(L, rank, world)


def merge_table_partitioned(q, ids, table, distance):
    """The merge of the table-partitioned layout (every rank holds a block of the L tables over ALL k-mers and
    answers ALL queries; `table` already in GLOBAL numbering).  The reference reports an id in the FIRST table
    whose probed bucket holds it and never looks at it again (label[], motif_both_points.cpp:232-238), and
    whether it is a hit does not depend on the table (the distance does not): so of the gathered tuples of one
    (query, id) the one with the smallest table is the reference's line, the others are dropped, and the
    result is ordered by (query, table, id) = the reference's file order.  int64 tensors (one device) in and
    out; two sorts of 64-bit keys, no host round trip but the final size."""
    if q.numel() == 0:
        return q, ids, table, distance
    q, ids, table = q.to(torch.int64), ids.to(torch.int64), table.to(torch.int64)
    # (q, id, table) ascending: the first tuple of every (q, id) run carries the smallest table
    key = (q << 37) | (ids << 5) | table          # q < 2^26, id < 2^32, table < 32: 63 bits
    key, order = torch.sort(key)
    pair = key >> 5
    first = torch.ones_like(pair, dtype=torch.bool)
    first[1:] = pair[1:] != pair[:-1]
    keep = order[first]
    q, ids, table, distance = q[keep], ids[keep], table[keep], distance[keep]
    key2 = (q << 37) | (table << 32) | ids
    order2 = torch.argsort(key2)
    return q[order2], ids[order2], table[order2], distance[order2]


def table_costs(k, K, L, W, a, b, codes_sample, device=0, coords=None):
    """Estimated join work of every table, from a sample of the DB's k-mers: the sum over the table's buckets
    of (sample k-mers in the bucket)^2 -- for queries distributed like the DB, proportional to the (member,
    query) pairs the table contributes.  One short-lived handle with all L planes (no index) hashes the sample;
    the host counts.  Deterministic: every rank computes the same numbers from the same sample."""
    from . import capi
    eng = capi.Engine(k, K, L, W, a, b, device=device, coords=coords)
    try:
        ints = eng.hash_codes(np.ascontiguousarray(codes_sample, dtype=np.uint8))    # [n][L][K]
    finally:
        eng.close()
    cost = np.zeros(L)
    for l in range(L):
        rows = np.ascontiguousarray(ints[:, l, :]).view([("", np.int32)] * K).ravel()
        _, cnt = np.unique(rows, return_counts=True)
        cost[l] = float((cnt.astype(np.float64) ** 2).sum())
    return cost


def assign_tables(cost, L, world):
    """Tables of every rank (ascending global numbers) from hs_assign_tables (libhsearch_dist.so): longest
    processing time first by `cost` (None: round robin)."""
    from . import cdist
    owner = cdist.assign_tables(cost, L, world)
    return [np.nonzero(owner == r)[0].astype(np.int64) for r in range(world)]


def query_table_partitioned(eng, tables, d_centers_ptr, nq, R, out, cap, group=None, force=False, codes=False):
    """One pass of the table-partitioned layout on this rank: `eng` holds the tables `tables` (global numbers,
    ascending) over all k-mers, the nq queries are ALL queries (the same on every rank); its hits with the
    tables made global, all-gathered (allgather_hits, q_offset 0), merged (merge_table_partitioned).  `out` =
    dict of the rank's output tensors q / id / table / dist (cap entries).  Returns the merged (q, id, table,
    dist) -- every rank the same list, the reference's order -- and the rank's own hit count."""
    nh = eng.query_dev(d_centers_ptr, nq, R, out["q"].data_ptr(), out["id"].data_ptr(), out["table"].data_ptr(),
                       out["dist"].data_ptr(), cap, codes=codes)
    tmap = torch.as_tensor(np.asarray(tables, dtype=np.int64), device=out["table"].device)
    gtab = tmap[out["table"][:nh].to(torch.int64)].to(torch.int32)
    gq, gi, gt, gd = allgather_hits(out["q"], out["id"], gtab, out["dist"], nh, q_offset=0, group=group, force=force)
    return merge_table_partitioned(gq, gi, gt, gd), nh


def query_bucket_partitioned(eng, rank, world, d_centers_ptr, nq, R, out, cap, group=None, force=False, codes=False):
    """One pass of the bucket-partitioned layout on this rank: `eng` holds the WHOLE index (replicated), the nq
    queries are ALL queries (the same on every rank), and the rank searches only the buckets of part `rank` of
    `world` (hs_set_bucket_partition: a function of the bucket's key fingerprint); its hits (tables already
    global) are all-gathered (q_offset 0) and merged by the first-seen rule (merge_table_partitioned: the same
    rule -- per (query, id) the smallest table).  Returns the merged (q, id, table, dist) -- every rank the same
    list, the reference's order -- and the rank's own hit count."""
    eng.set_bucket_partition(rank, world)
    try:
        nh = eng.query_dev(d_centers_ptr, nq, R, out["q"].data_ptr(), out["id"].data_ptr(), out["table"].data_ptr(),
                           out["dist"].data_ptr(), cap, codes=codes)
    finally:
        eng.set_bucket_partition(0, 1)
    gq, gi, gt, gd = allgather_hits(out["q"], out["id"], out["table"], out["dist"], nh, q_offset=0, group=group,
                                    force=force)
    return merge_table_partitioned(gq, gi, gt, gd), nh


def hits_to_numpy(q, ids, table, distance):
    return dict(q=q.cpu().numpy().astype(np.uint32), id=ids.cpu().numpy().astype(np.uint32),
                table=table.cpu().numpy().astype(np.uint32), dist=distance.cpu().numpy())


def allgather_edges(edge_i, edge_j, group=None, device=None):
    """All ranks' (i, j) edge lists of one clustering table, concatenated in rank order (numpy in,
    numpy out).  The exchange is allgather_hits' single collective with i in the q column; over
    RCCL the host arrays pass through `device`."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return np.asarray(edge_i, dtype=np.uint32), np.asarray(edge_j, dtype=np.uint32)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n = len(edge_i)
    ti = torch.from_numpy(np.ascontiguousarray(edge_i, dtype=np.uint32).view(np.int32)).to(dev)
    tj = torch.from_numpy(np.ascontiguousarray(edge_j, dtype=np.uint32).view(np.int32)).to(dev)
    zt = torch.zeros(n, dtype=torch.int32, device=dev)
    zd = torch.zeros(n, dtype=torch.float64, device=dev)
    gi, gj, _, _ = allgather_hits(ti, tj, zt, zd, n, group=group)
    return gi.cpu().numpy().astype(np.uint32), gj.cpu().numpy().astype(np.uint32)


def clustering_sharded(k, K, L, W, a, b, codes, R, device=0, coords=None, group=None,
                       rank=None, world=None, gather=None):
    """Clustering() (hclust2.cpp:86-151) over the ranks of `group`: per table every rank joins its
    block of the not-yet-absorbed k-mers against the table (hs_clustering_table_edges), the edge
    lists are all-gathered (the path's one exchange step), and every rank runs the same greedy
    pass (hs_clustering_table_apply), so all ranks end with identical (merged, owner, table) --
    identical to the single-GPU hs_clustering.  `gather(edge_i, edge_j) -> (all_i, all_j)` may
    replace the collective (tests shard on one GPU this way)."""
    from . import capi
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if gather is None:
        gdev = None
        if dist.is_initialized() and dist.get_backend(group) == "nccl":
            gdev = "cuda:%d" % int(device)
        gather = lambda ei, ej: allgather_edges(ei, ej, group=group, device=gdev)
    st = capi.ClusterState(k, K, L, W, a, b, codes, R, device=device, coords=coords)
    for l in range(int(L)):
        ei, ej, _ = st.table_edges(l, rank, world)
        ai, aj = gather(ei, ej)
        st.table_apply(l, ai, aj)
    return st.end()


class TorchShardOps:
    """The collectives of index_build_sharded over torch.distributed (RCCL = backend "nccl": tensors stay on
    the rank's GPU; gloo: they pass through host memory).  Tensors in, tensors out, on `dev`."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        self.host = dist.get_backend(group) != "nccl"

    def allgather_blocks(self, block, counts):
        """block: int64 tensor [counts[rank]] -> int64 [sum(counts)], blocks in rank order (padded to the
        largest block for the collective: blocks differ by at most one element)."""
        m = max(counts)
        pad = torch.zeros(m, dtype=block.dtype, device="cpu" if self.host else block.device)
        pad[:block.numel()] = block.cpu() if self.host else block
        out = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(out, pad, group=self.group)
        return torch.cat([o[:c] for o, c in zip(out, counts)]).to(self.dev)

    def allreduce_sum(self, t):
        x = t.cpu() if self.host else t
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x.to(self.dev)

    def allreduce_max_int(self, v):
        x = torch.tensor([int(v)], dtype=torch.int64, device="cpu" if self.host else self.dev)
        dist.all_reduce(x, op=dist.ReduceOp.MAX, group=self.group)
        return int(x.item())


def index_build_sharded(eng, codes, ops, max_seeds=4):
    """SURVEY 8(e), "Index build": hs_index_build with the evaluation of the hash functions spread over the
    ranks of `ops` (include/hsearch.h, hs_index_shard_*): per table every rank hashes its block of the
    k-mers, the 8-byte fingerprints are all-gathered, every rank groups all of them, the buckets' tuples
    are summed from the ranks that hashed each bucket's first member, and every rank proves the exact
    HashKey-string membership of its own k-mers.  A fingerprint collision anywhere restarts every rank
    with the next seed, as hs_index_build does locally.  `eng`: an Engine (or anything with its shard_*
    methods); `ops`: TorchShardOps or an object with the same three methods.  Returns the index info."""
    n = len(codes)
    lo, cnt = eng.shard_begin(codes, ops.rank, ops.world)
    counts = []
    for r in range(ops.world):
        l0, h0 = shard_bounds(n, r, ops.world)
        counts.append(h0 - l0)
    assert (lo, cnt) == (shard_bounds(n, ops.rank, ops.world)[0], counts[ops.rank])
    def ready():
        # The library works on its own stream and returns with it drained; what torch has queued on ITS
        # stream (zero fills, concatenations, collectives) must be complete before a pointer is handed over.
        if ops.dev.type == "cuda":
            torch.cuda.current_stream(ops.dev).synchronize()

    for seed in range(max_seeds):
        collided = 0
        for l in range(eng.L):
            fp_block = torch.empty(max(cnt, 1), dtype=torch.int64, device=ops.dev)
            ready()
            eng.shard_hash(l, seed, fp_block.data_ptr())
            fp_all = ops.allgather_blocks(fp_block[:cnt], counts).contiguous()
            ready()
            nb = eng.shard_group(l, fp_all.data_ptr())
            tup = torch.zeros(max(nb, 1) * eng.K, dtype=torch.int32, device=ops.dev)
            ready()
            eng.shard_tuples(l, tup.data_ptr())
            tup_all = ops.allreduce_sum(tup).contiguous()
            ready()
            collided |= eng.shard_finish(l, tup_all.data_ptr())
        if not ops.allreduce_max_int(collided):
            return eng.shard_end(seed)
    raise RuntimeError("key fingerprints collided for %d seeds" % max_seeds)
