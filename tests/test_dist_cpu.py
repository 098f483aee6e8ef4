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


def _table_worker(rank, world, port, ret):
    """The table-partitioned layout over gloo: rank r searches ALL queries in ITS tables only (the oracle over
    the planes of those tables), local table numbers are made global, the tuples are all-gathered and merged
    (per (query, id) the smallest table, order (query, table, id)): the reference's full-L output."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    k, K, L, W, R, n, nq = 25, 4, 6, 120.0, 45.0, 4000, 150
    a, b = synth.make_planes(k, K, L, W)
    rng = np.random.default_rng(5)
    codes = synth.make_db(n, k)
    codes[rng.choice(n, 600, replace=False)] = codes[rng.choice(n, 600)]      # duplicates: pairs found by many tables
    centers, _ = synth.make_queries(codes, nq, jitter=0.2)
    # an uneven, interleaved assignment: rank 0 tables {0, 3, 4, 5}, rank 1 tables {1, 2}
    mine = np.array([[0, 3, 4, 5], [1, 2]][rank])
    res = O.search(a[mine], b[mine], W, R, O.embed_codes(codes), centers)
    nh = len(res["q"])
    def t(x, dt):
        return torch.from_numpy(x.astype(np.int64)).to(dt)
    gq, gi, gt, gd = hdist.allgather_hits(t(res["q"], torch.int32), t(res["id"], torch.int32),
                                          t(mine[res["table"].astype(np.int64)], torch.int32),
                                          torch.from_numpy(res["dist"]), nh, q_offset=0)
    mq, mi, mt, md = hdist.merge_table_partitioned(gq, gi, gt, gd)
    full = O.search(a, b, W, R, O.embed_codes(codes), centers)
    ok = (np.array_equal(mq.numpy(), full["q"].astype(np.int64)) and
          np.array_equal(mi.numpy(), full["id"].astype(np.int64)) and
          np.array_equal(mt.numpy(), full["table"].astype(np.int64)) and
          np.array_equal(md.numpy(), full["dist"]) and len(full["q"]) > 50 and len(gq) > len(mq) and
          len(np.unique(full["table"])) > 3)
    ret.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_table_partitioned_search_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_table_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: True, 1: True}       # every rank ends with the reference's list


def _bucket_worker(rank, world, port, ret):
    """The bucket-partitioned layout over gloo: the ranks share the BUCKETS.  Rank r's list = for every table the
    hits of the (query, bucket) probes that fall to it -- here by a hash of the bucket's int tuple, and for one
    big bucket per table by a hash of tuple AND query (the library's rule for giant buckets) --, reduced inside
    the rank by the first-seen rule over ITS probes, as a handle with hs_set_bucket_partition reports them; the
    tuples are all-gathered and merged (merge_table_partitioned): the reference's full output."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    k, K, L, W, R, n, nq = 25, 3, 6, 150.0, 45.0, 4000, 150
    a, b = synth.make_planes(k, K, L, W)
    rng = np.random.default_rng(6)
    codes = synth.make_db(n, k)
    codes[rng.choice(n, 600, replace=False)] = codes[rng.choice(n, 600)]      # duplicates: pairs found by many tables
    centers, _ = synth.make_queries(codes, nq, jitter=0.2)
    pts = O.embed_codes(codes)
    qints = O.hash_all(a, b, W, centers)                        # [nq][L][K] bucket ints of every probe
    dints = O.hash_all(a, b, W, pts)
    mine_q, mine_i, mine_t, mine_d = [], [], [], []
    for l in range(L):
        one = O.search(a[l:l + 1], b[l:l + 1], W, R, pts, centers)       # table l alone: all its hits
        tuples, counts = np.unique(dints[:, l, :], axis=0, return_counts=True)
        giant = tuple(tuples[np.argmax(counts)])
        def part(q):
            t = tuple(int(v) for v in qints[q, l])
            return hash((t, int(q)) if t == giant else t) % world
        keep = np.array([part(int(q)) == rank for q in one["q"]], dtype=bool)
        mine_q.append(one["q"][keep]); mine_i.append(one["id"][keep]); mine_d.append(one["dist"][keep])
        mine_t.append(np.full(int(keep.sum()), l))
    mq, mi, mt, md = (np.concatenate(x) for x in (mine_q, mine_i, mine_t, mine_d))
    def t64(x):
        return torch.from_numpy(np.asarray(x).astype(np.int64))
    # inside the rank: per (query, id) the first table among its own probes
    lq, li, lt, ld = hdist.merge_table_partitioned(t64(mq), t64(mi), t64(mt), torch.from_numpy(md))
    nh = len(lq)
    gq, gi, gt, gd = hdist.allgather_hits(lq.to(torch.int32), li.to(torch.int32), lt.to(torch.int32), ld, nh, q_offset=0)
    fq, fi, ft, fd = hdist.merge_table_partitioned(gq, gi, gt, gd)
    full = O.search(a, b, W, R, pts, centers)
    ok = (np.array_equal(fq.numpy(), full["q"].astype(np.int64)) and
          np.array_equal(fi.numpy(), full["id"].astype(np.int64)) and
          np.array_equal(ft.numpy(), full["table"].astype(np.int64)) and
          np.array_equal(fd.numpy(), full["dist"]) and len(full["q"]) > 50 and len(gq) > len(fq) and 0 < nh < len(full["q"]))
    ret.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_partitioned_search_world2():
    os.environ["PYTHONHASHSEED"] = "0"      # (the workers' stand-in for the fingerprint is Python's hash of a tuple of ints)
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(ret.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: True, 1: True}       # every rank ends with the reference's list


def test_merge_table_partitioned_small_cases():
    q = torch.tensor([3, 0, 3, 3, 0, 1], dtype=torch.int64)
    i = torch.tensor([7, 5, 7, 2, 5, 9], dtype=torch.int64)
    t = torch.tensor([4, 2, 1, 6, 0, 3], dtype=torch.int64)
    d = torch.tensor([1.5, 2.5, 1.5, 0.5, 2.5, 9.0], dtype=torch.float64)
    mq, mi, mt, md = hdist.merge_table_partitioned(q, i, t, d)
    assert mq.tolist() == [0, 1, 3, 3] and mi.tolist() == [5, 9, 7, 2] and mt.tolist() == [0, 3, 1, 6]
    assert md.tolist() == [2.5, 9.0, 1.5, 0.5]
    e = torch.empty(0, dtype=torch.int64)
    assert hdist.merge_table_partitioned(e, e, e, torch.empty(0, dtype=torch.float64))[0].numel() == 0


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


class _FakeShardEngine:
    """The per-rank steps of hs_index_shard_* restated in numpy on the oracle's bucket ints (this is a test
    of hsearch_amd.dist.index_build_sharded's plumbing over a real process group, not of the kernels;
    tests/test_gpu_parity.py::test_index_build_with_the_hashing_spread_over_ranks runs the real steps)."""

    def __init__(self, O, k, K, L, W, a, b):
        self.O, self.k, self.K, self.L, self.W, self.a, self.b = O, k, K, L, W, a, b
        self.tables = []

    @staticmethod
    def _view(ptr, count, ctype, dtype):
        import ctypes
        return np.frombuffer((ctype * count).from_address(ptr), dtype=dtype)

    def shard_begin(self, codes, rank, world):
        self.codes, self.n = codes, len(codes)
        self.lo, hi = hdist.shard_bounds(self.n, rank, world)
        self.cnt = hi - self.lo
        return self.lo, self.cnt

    def shard_hash(self, l, seed, ptr):
        from hsearch_amd import capi
        import ctypes
        pts = self.O.embed_codes(self.codes[self.lo:self.lo + self.cnt])
        self.ints = self.O.hash_all(self.a[l:l + 1], self.b[l:l + 1], self.W, pts)[:, 0, :]
        fp = np.array([capi.key_fingerprint(t, seed) for t in self.ints], dtype=np.uint64)
        self._view(ptr, max(self.cnt, 1), ctypes.c_uint64, np.uint64)[:self.cnt] = fp

    def shard_group(self, l, ptr):
        import ctypes
        fp = self._view(ptr, self.n, ctypes.c_uint64, np.uint64).copy()
        self.ids = np.argsort(fp, kind="stable").astype(np.uint32)
        self.dir_key, self.dir_start = np.unique(fp[self.ids], return_index=True)
        return len(self.dir_key)

    def shard_tuples(self, l, ptr):
        import ctypes
        nb = len(self.dir_key)
        out = self._view(ptr, max(nb, 1) * self.K, ctypes.c_int32, np.int32)
        first = self.ids[self.dir_start].astype(np.int64)
        mine = (first >= self.lo) & (first < self.lo + self.cnt)
        tup = np.zeros((nb, self.K), dtype=np.int32)
        tup[mine] = self.ints[first[mine] - self.lo]
        out[:nb * self.K] = tup.ravel()

    def shard_finish(self, l, ptr):
        from hsearch_amd import capi
        import ctypes
        nb = len(self.dir_key)
        tup = self._view(ptr, max(nb, 1) * self.K, ctypes.c_int32, np.int32)[:nb * self.K].reshape(nb, self.K).copy()
        pos_of = np.empty(self.n, dtype=np.int64)
        pos_of[self.ids] = np.arange(self.n)
        collided = 0
        for i in range(self.cnt):
            r = np.searchsorted(self.dir_start, pos_of[self.lo + i], side="right") - 1
            if not capi.key_strings_equal(self.ints[i], tup[r]):
                collided = 1
        self.tables.append((self.ids.copy(), self.dir_key.copy(), self.dir_start.copy(), tup))
        return collided

    def shard_end(self, seed):
        return dict(n=self.n, n_buckets=[len(t[1]) for t in self.tables], key_seed=seed)


def _shard_build_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    k, K, L, W, n = 25, 4, 3, 100.0, 1501          # 1501: uneven blocks
    a, b = synth.make_planes(k, K, L, W, seed=51)
    codes = synth.make_db(n, k, seed=52)
    eng = _FakeShardEngine(O, k, K, L, W, a, b)
    info = hdist.index_build_sharded(eng, codes, hdist.TorchShardOps())
    # the unsharded result of the same restatement
    one = _FakeShardEngine(O, k, K, L, W, a, b)

    class Solo:
        rank, world, dev = 0, 1, torch.device("cpu")
        def allgather_blocks(self, block, counts): return block
        def allreduce_sum(self, t): return t
        def allreduce_max_int(self, v): return int(v)
    info1 = hdist.index_build_sharded(one, codes, Solo())
    ok = info["n_buckets"] == info1["n_buckets"] == O.Index(a, b, W, O.embed_codes(codes)).table_sizes()
    for t, t1 in zip(eng.tables, one.tables):
        ok = ok and all(np.array_equal(x, y) for x, y in zip(t, t1))
    ret.put((rank, bool(ok), info["n_buckets"]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_index_build_world2():
    """SURVEY 8(e), "Index build": the exchange steps of the sharded build (all-gather of uneven fingerprint
    blocks, sum of the buckets' tuples, max of the collision flags) over a real gloo process group of two:
    both ranks end with the tables of the unsharded build."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_build_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = [ret.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in got) and got[0][2] == got[1][2]
