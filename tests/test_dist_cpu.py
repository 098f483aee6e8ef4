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
