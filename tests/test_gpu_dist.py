"""Two real ranks on the one GPU of the box (gloo between them): the multi-GPU drivers end to end --
query-sharded search with the hit all-gather, and Clustering() with per-table edge shards -- against
the single-process results.  (RCCL itself needs one GPU per rank; the collectives' logic is the
same code over gloo, see hsearch_amd/dist.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
    import hsearch_amd
    from hsearch_amd import Engine, synth
    from hsearch_amd import dist as hdist
    ok = True
    # ---- search: index replicated, queries sharded, hits all-gathered in global order
    k, K, L, W, R, n, nq = 25, 8, 4, 150.0, 45.0, 20000, 1001
    a, b = synth.make_planes(k, K, L, W, seed=3)
    codes = synth.make_db(n, k, seed=4)
    centers, _ = synth.make_queries(codes, nq, seed=5, jitter=0.2)
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    lo, hi = hdist.shard_bounds(nq, rank, world)
    mine = eng.query(centers[lo:hi], R, want_cand=False)
    t = lambda x, dt: torch.from_numpy(x.astype(np.int64)).to(dt)
    q, ids, tab, dd = hdist.allgather_hits(t(mine["q"], torch.int32), t(mine["id"], torch.int32),
                                           t(mine["table"], torch.int32), torch.from_numpy(mine["dist"]),
                                           len(mine["q"]), q_offset=lo)
    full = eng.query(centers, R, want_cand=False)
    ok = ok and len(full["q"]) > 100
    ok = ok and np.array_equal(q.numpy(), full["q"].astype(np.int64))
    ok = ok and np.array_equal(ids.numpy(), full["id"].astype(np.int64))
    ok = ok and np.array_equal(tab.numpy(), full["table"].astype(np.int64))
    ok = ok and np.array_equal(dd.numpy(), full["dist"])
    eng.close()
    # ---- clustering: per-table edge shards + all-gather + identical greedy pass on every rank
    rng = np.random.default_rng(8)
    fam = rng.integers(0, 20, size=(60, k), dtype=np.uint8)
    ccodes = fam[rng.integers(0, 60, size=4000)].copy()
    for row in ccodes:
        for _ in range(int(rng.integers(0, 4))):
            row[rng.integers(0, k)] = rng.integers(0, 20)
    ca, cb = synth.make_planes(k, 4, 6, 100.0, seed=9)
    got = hdist.clustering_sharded(k, 4, 6, 100.0, ca, cb, ccodes, 60.0)
    want = hsearch_amd.clustering(k, 4, 6, 100.0, ca, cb, ccodes, 60.0)
    for x, y in zip(got, want):
        ok = ok and np.array_equal(x, y)
    ok = ok and int((want[0] == 2).sum()) > 500
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        ret.put(all(flags))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_the_gpu():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    ok = ret.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
