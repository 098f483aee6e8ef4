"""Multi-rank path on CPU: world_size-2 gloo processes run the same shard + all-gather plumbing
bench.py uses over RCCL.  The per-rank 'search' is the CPU oracle here (this is a test of the
plumbing, not of the kernels): sharded queries + all-gather must reproduce the single-process hit
list in the reference's global order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hsearch_amd import dist as hdist
from hsearch_amd import synth


def test_shard_bounds_partition():
    for n in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            cuts = [hdist.shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            for (lo, hi), (lo2, _) in zip(cuts, cuts[1:]):
                assert hi == lo2 and hi >= lo
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    k, K, L, W, R, n, nq = 25, 4, 4, 100.0, 40.0, 3000, 101
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(n, k)
    centers, _ = synth.make_queries(codes, nq, jitter=0.2)
    lo, hi = hdist.shard_bounds(nq, rank, world)
    res = O.search(a, b, W, R, O.embed_codes(codes), centers[lo:hi])
    nh = len(res["q"])
    pad = 7  # buffers larger than n_hits, as the device buffers are
    def t(x, dt):
        return torch.cat([torch.from_numpy(x.astype(np.int64)).to(dt), torch.zeros(pad, dtype=dt)])
    q, ids, tab, dd = hdist.allgather_hits(
        t(res["q"], torch.int32), t(res["id"], torch.int32), t(res["table"], torch.int32),
        torch.cat([torch.from_numpy(res["dist"]), torch.zeros(pad, dtype=torch.float64)]), nh,
        q_offset=lo)
    if rank == 0:
        full = O.search(a, b, W, R, O.embed_codes(codes), centers)
        ok = (np.array_equal(q.numpy(), full["q"].astype(np.int64)) and
              np.array_equal(ids.numpy(), full["id"].astype(np.int64)) and
              np.array_equal(tab.numpy(), full["table"].astype(np.int64)) and
              np.array_equal(dd.numpy(), full["dist"]) and len(full["q"]) > 0)
        ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_allgather_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    ok = ret.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def _table_edges_oracle(O, a_l, b_l, W, R, pts, active, lo, hi):
    """Edges (i, j) of one clustering table for the active k-mers active[lo:hi] as the i side:
    same bucket of the table built over `active`, i != j, sqrt(d2) <= R (hclust2.cpp:46-60,107-120)."""
    keys = O.hash_table(a_l, b_l, W, pts[active])
    _, inv = np.unique(keys, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    ei, ej = [], []
    for t in range(lo, hi):
        mates = np.nonzero(inv == inv[t])[0]
        mates = mates[mates != t]
        if len(mates) == 0:
            continue
        d2 = O.pairwise_square(pts[active[mates]], pts[active[t]][None, :])[0]
        for m in mates[np.sqrt(d2) <= R]:
            ei.append(active[t])
            ej.append(active[m])
    return np.array(ei, dtype=np.uint32), np.array(ej, dtype=np.uint32)


def _cluster_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    import hsearch_amd
    k, K, L, W, R = 25, 4, 4, 100.0, 60.0
    rng = np.random.default_rng(31)
    fam = rng.integers(0, 20, size=(12, k))
    rows = []
    for f in fam:
        for _ in range(25):
            r = f.copy()
            for _ in range(int(rng.integers(0, 4))):
                r[rng.integers(0, k)] = rng.integers(0, 20)
            rows.append(r)
    codes = np.concatenate([np.array(rows, dtype=np.uint8), synth.make_db(300, k, seed=4)])
    rng.shuffle(codes)
    a, b = synth.make_planes(k, K, L, W, seed=47)
    pts = O.embed_codes(codes)
    # the host half of the sharded Clustering(): state, all-gather of edges, greedy apply.  The
    # GPU half (hs_clustering_table_edges) is replaced by the oracle's edges of the same block.
    st = hsearch_amd.ClusterState(k, K, L, W, a, b, codes, R)
    merged = np.zeros(len(codes), dtype=np.uint8)
    total = 0
    for l in range(L):
        active = np.nonzero(merged != 2)[0].astype(np.uint32)
        lo, hi = hdist.shard_bounds(len(active), rank, world)
        ei, ej = _table_edges_oracle(O, a[l], b[l], W, R, pts, active, lo, hi)
        ai, aj = hdist.allgather_edges(ei, ej)
        total += len(ai)
        st.table_apply(l, ai, aj)
        # merged[] after the table, from an independent single-process run of the oracle
        merged, _ = O.clustering(a[:l + 1], b[:l + 1], W, R, pts)
    got_merged, got_owner, got_table = st.end()
    want_merged, want_owner = O.clustering(a, b, W, R, pts)
    ok = (np.array_equal(got_merged, want_merged) and np.array_equal(got_owner, want_owner)
          and (want_merged == 2).sum() > 50 and total > 100
          and ((got_table != 0xffffffff) == (got_merged == 2)).all())
    gathered = [None] * world
    dist.all_gather_object(gathered, bool(ok))
    if rank == 0:
        ret.put(all(gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_clustering_host_path_world2():
    """SURVEY 8(e), config 4 on CPU: hs_clustering_begin / table_apply / end and the edge
    all-gather over gloo, world 2, against the oracle's Clustering()."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cluster_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    ok = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
