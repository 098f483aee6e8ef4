"""The C++/RCCL multi-GPU layer on the GPU box: one rank (the box has one GPU; RCCL refuses two
ranks on one device), so this covers the RCCL transport itself -- communicator creation,
pack kernel -> ncclAllGather -> unpack kernel with device pointers, hs_comm_query against the plain
query, and the one-process-per-GPU creation path (unique id) -- while the world > 1 layout logic is
covered on the CPU by tests/test_cdist_cpu.py over the loopback transport."""
import ctypes as C
import threading

import numpy as np
import pytest

from hsearch_amd import Engine, capi, cdist, synth

pytestmark = pytest.mark.gpu


def test_rccl_allgather_of_device_hits_world1():
    import torch
    comm = cdist.Comm(cdist.RCCL_LOCAL, 1, devices=[0])
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 4096, 100_001):
        q = torch.from_numpy(np.sort(rng.integers(0, 5000, size=n)).astype(np.int32)).to(dev)
        i = torch.from_numpy(rng.integers(0, 2**31 - 1, size=n).astype(np.int32)).to(dev)
        t = torch.from_numpy(rng.integers(0, 32, size=n).astype(np.int32)).to(dev)
        d = torch.from_numpy(rng.random(n) * 40.0).to(dev)
        cap = n + 3
        oq, oi, ot = (torch.full((cap,), -1, dtype=torch.int32, device=dev) for _ in range(3))
        od = torch.full((cap,), -1.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        st, tot = comm.allgather_hits(0, q.data_ptr(), i.data_ptr(), t.data_ptr(), d.data_ptr(), n, 1000,
                                      oq.data_ptr(), oi.data_ptr(), ot.data_ptr(), od.data_ptr(), cap)
        assert st == capi.HS_OK and tot == n
        assert torch.equal(oq[:n], q + 1000) and torch.equal(oi[:n], i) and torch.equal(ot[:n], t)
        assert torch.equal(od[:n], d)
        assert bool((oq[n:] == -1).all()) and bool((od[n:] == -1.0).all())
        if n:
            st, tot = comm.allgather_hits(0, q.data_ptr(), i.data_ptr(), t.data_ptr(), d.data_ptr(), n, 0,
                                          oq.data_ptr(), oi.data_ptr(), ot.data_ptr(), od.data_ptr(), n - 1)
            assert st == capi.HS_ERR_CAPACITY and tot == n
    comm.close()


def test_comm_query_equals_plain_query(oracle):
    k, K, L, W, R, n, nq = 25, 8, 6, 150.0, 40.0, 30000, 900
    a, b = synth.make_planes(k, K, L, W, seed=61)
    codes = synth.make_db(n, k, seed=62)
    centers, _ = synth.make_queries(codes, nq, seed=63, jitter=0.2)
    eng = Engine(k, K, L, W, a, b, device=0)
    eng.index_build(codes)
    want = eng.query(centers, R)
    assert len(want["q"]) > 100
    comm = cdist.Comm(cdist.RCCL_LOCAL, 1, devices=[0])
    got = comm.query(0, eng, centers, 0, R)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(got[f], want[f]), f
    # the table-partitioned form with one rank holding every table: pack -> ncclAllGather -> unpack -> merge
    got = comm.query_tables(0, eng, np.arange(L), centers, R, cap=8)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(got[f], want[f]), f
    # ... and the bucket-partitioned form with one rank = every bucket: the same pipeline over RCCL
    got = comm.query_buckets(0, eng, centers, R, cap=8)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(got[f], want[f]), f
    # a block in the middle of a larger query set: q comes back global
    got = comm.query(0, eng, centers[300:500], 300, R, cap=8)       # and the capacity retry
    sel = (want["q"] >= 300) & (want["q"] < 500)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(got[f], want[f][sel]), f
    comm.close()
    # one process per GPU: unique id -> ncclCommInitRank (world 1 here)
    lib = cdist.load()
    uid = C.create_string_buffer(128)
    assert lib.hs_comm_unique_id(uid) == capi.HS_OK
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    st = lib.hs_comm_create_rank(uid, C.c_uint32(0), C.c_uint32(1), C.c_int(0), C.byref(h), err, C.c_uint32(256))
    assert st == capi.HS_OK, err.value
    c2 = cdist.Comm.__new__(cdist.Comm)
    c2._lib, c2._h, c2.world = lib, h, 1
    got = c2.query(0, eng, centers, 0, R)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(got[f], want[f]), f
    c2.close()
    eng.close()
    _ = oracle


def test_rank_threads_with_two_live_handles_over_the_loopback(oracle):
    """Two (and three) rank threads, each with its OWN handle and index on GPU 0, through hs_comm_query
    over the host-memory transport created with devices: every rank gets every rank's hits in the
    reference's order = the one-handle query; blocks given as residue codes give the same
    (hs_comm_query_codes); the capacity retry is one decision; and a rank whose search fails (its
    handle has no index: HS_ERR_STATE) takes part in the exchange -- it returns its own status, the
    others HS_ERR_PEER, no thread hangs, and the communicator serves the next call."""
    import threading
    k, K, L, W, R, n, nq = 25, 8, 6, 150.0, 40.0, 30000, 901
    a, b = synth.make_planes(k, K, L, W, seed=61)
    codes = synth.make_db(n, k, seed=62)
    qcodes, _ = synth.make_query_codes(codes, nq, seed=64)
    centers = synth.embed(qcodes)
    one = Engine(k, K, L, W, a, b, device=0)
    one.index_build(codes)
    want = one.query(centers, R)
    one.close()
    assert len(want["q"]) > 100
    _ = oracle

    def run(world, fn):
        out, err = [None] * world, []

        def body(r):
            try:
                out[r] = fn(r)
            except BaseException as e:   # noqa
                err.append((r, repr(e)))
        ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ts), "ranks hung"
        assert not err, err
        return out
    for world in (2, 3):
        comm = cdist.Comm(cdist.LOOPBACK, world, devices=[0] * world)
        engs = [Engine(k, K, L, W, a, b, device=0) for _ in range(world)]
        for e in engs:
            e.index_build(codes)
        blocks = [cdist.shard_bounds(nq, world, r) for r in range(world)]
        for as_codes in (False, True):
            got = run(world, lambda r: comm.query(r, engs[r], (qcodes if as_codes else centers)[blocks[r][0]:blocks[r][1]],
                                                  blocks[r][0], R, cap=8, codes=as_codes))   # cap=8: the retry
            for r in range(world):
                for f in ("q", "id", "table", "dist"):
                    assert np.array_equal(got[r][f], want[f]), (world, r, f, as_codes)
        # rank 1's handle loses its index (a new plane family drops it): its query fails
        engs[1].set_planes(a, b)
        sts = run(world, lambda r: comm.query_status(r, engs[r], centers[blocks[r][0]:blocks[r][1]], blocks[r][0], R,
                                                     cap=1 << 16))
        assert sts[1] == capi.HS_ERR_STATE and all(s == capi.HS_ERR_PEER for i, s in enumerate(sts) if i != 1), sts
        assert "rank 1 failed" in comm.last_error(0)
        engs[1].index_build(codes)
        got = run(world, lambda r: comm.query(r, engs[r], centers[blocks[r][0]:blocks[r][1]], blocks[r][0], R))
        assert np.array_equal(got[0]["id"], want["id"])
        for e in engs:
            e.close()
        comm.close()


def test_merge_first_table_on_the_device():
    """hs_merge_first_table_dev against a numpy statement of the rule: per (query, id) the smallest table,
    then the order (query, table, id)."""
    import torch
    k, K, L, W = 25, 4, 2, 100.0
    a, b = synth.make_planes(k, K, L, W)
    eng = Engine(k, K, L, W, a, b)
    rng = np.random.default_rng(9)
    n = 200_000
    q = rng.integers(0, 3000, n).astype(np.uint32)
    i = rng.integers(0, 500, n).astype(np.uint32)
    t = rng.integers(0, 32, n).astype(np.uint32)
    d = (q * 1000.0 + i).astype(np.float64)           # a function of (q, id), as a distance is
    dev = torch.device("cuda", 0)
    tq, ti, tt = (torch.from_numpy(x.view(np.int32).copy()).to(dev) for x in (q, i, t))
    td = torch.from_numpy(d).to(dev)
    torch.cuda.synchronize()
    kept = eng.merge_first_table_dev(tq.data_ptr(), ti.data_ptr(), tt.data_ptr(), td.data_ptr(), n)
    key = (q.astype(np.int64) << 32) | i
    order = np.lexsort((t, key))
    first = np.ones(n, bool)
    first[1:] = key[order][1:] != key[order][:-1]
    sel = order[first]
    o2 = np.lexsort((i[sel], t[sel], q[sel]))
    sel = sel[o2]
    assert kept == len(sel) and kept < n
    assert np.array_equal(tq[:kept].cpu().numpy().view(np.uint32), q[sel])
    assert np.array_equal(ti[:kept].cpu().numpy().view(np.uint32), i[sel])
    assert np.array_equal(tt[:kept].cpu().numpy().view(np.uint32), t[sel])
    assert np.array_equal(td[:kept].cpu().numpy(), d[sel])
    assert eng.merge_first_table_dev(tq.data_ptr(), ti.data_ptr(), tt.data_ptr(), td.data_ptr(), 0) == 0
    eng.close()


@pytest.mark.parametrize("world", [2, 4])
def test_table_partitioned_search_equals_one_handle_with_all_tables(world):
    """The table-partitioned layout (hs_comm_query_tables) with `world` live handles on ONE GPU over the
    host-memory transport, each holding its subset of the tables over all k-mers: every rank's merged list ==
    the list of one handle with all tables (hits, table of first sight, order, bit-identical distances), for
    contiguous and for cost-balanced (interleaved, uneven) table assignments, queries as points and as codes."""
    k, K, L, W, R, n, nq = 25, 6, 8, 130.0, 45.0, 60_000, 3000
    a, b = synth.make_planes(k, K, L, W)
    rng = np.random.default_rng(11)
    codes = synth.make_db(n, k)
    codes[rng.choice(n, 5000, replace=False)] = codes[rng.choice(n, 5000)]   # pairs that share buckets in many tables
    qcodes, _ = synth.make_query_codes(codes, nq)
    centers = synth.embed(qcodes)
    one = Engine(k, K, L, W, a, b)
    one.index_build(codes)
    want = one.query(centers, R, want_cand=False)
    one.close()
    assert len(want["q"]) > 1000 and len(np.unique(want["table"])) > 4
    cost = np.array([5.0, 1.0, 1.0, 3.0, 1.0, 2.0, 1.0, 1.0])
    for owner in (np.arange(L) * world // L, cdist.assign_tables(cost, L, world)):
        tabs = [np.nonzero(owner == r)[0].astype(np.uint32) for r in range(world)]
        assert sorted(np.concatenate(tabs).tolist()) == list(range(L)) and all(len(t) for t in tabs)
        engs = []
        for r in range(world):
            e = Engine(k, K, len(tabs[r]), W, a[tabs[r]], b[tabs[r]])
            e.index_build(codes)
            engs.append(e)
        comm = cdist.Comm(cdist.LOOPBACK, world, devices=[0] * world)
        for as_codes in (False, True):
            got = [None] * world
            def run(r):
                got[r] = comm.query_tables(r, engs[r], tabs[r], qcodes if as_codes else centers, R, codes=as_codes)
            th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            for r in range(world):
                for f in ("q", "id", "table", "dist"):
                    assert np.array_equal(got[r][f], want[f]), (world, r, f, as_codes)
        # the capacity protocol counts MERGED hits, and a rank without its table list fails everybody
        st = [None] * world
        def bad(r):
            try:
                comm.query_tables(r, engs[r], tabs[r] if r else tabs[r][::-1].copy(), centers, R)
                st[r] = capi.HS_OK
            except capi.HsError as e:
                st[r] = e.status
        th = [threading.Thread(target=bad, args=(r,)) for r in range(world)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        if len(tabs[0]) > 1:
            assert st[0] == capi.HS_ERR_INVALID and all(x == capi.HS_ERR_PEER for x in st[1:])
        comm.close()
        for e in engs:
            e.close()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_bucket_partitioned_search_equals_the_unpartitioned_search(world):
    """The bucket-partitioned layout: the index replicated, every rank ALL queries in ITS part of the buckets
    (hs_set_bucket_partition: a function of the bucket's key fingerprint).  (1) On one handle: the parts' lists
    are disjoint in (query, table) probes, every part is a sub-list of what the tables hold, no part is empty,
    and the parts' union merged by hs_merge_first_table_dev is the unpartitioned list -- hits, table of first
    sight, order, bit-identical distances; the partition is switched off again by (0, 1).  (2) hs_comm_query_buckets
    with `world` live handles on ONE GPU over the host-memory transport: every rank ends with that list, queries
    as points and as residue codes."""
    k, K, L, W, R, n, nq = 25, 6, 8, 130.0, 45.0, 60_000, 3000
    a, b = synth.make_planes(k, K, L, W)
    rng = np.random.default_rng(12)
    codes = synth.make_db(n, k)
    codes[rng.choice(n, 5000, replace=False)] = codes[rng.choice(n, 5000)]   # pairs that share buckets in many tables
    qcodes, _ = synth.make_query_codes(codes, nq)
    centers = synth.embed(qcodes)
    one = Engine(k, K, L, W, a, b)
    one.index_build(codes)
    want = one.query(centers, R)
    assert len(want["q"]) > 1000 and len(np.unique(want["table"])) > 4
    parts = []
    cand = np.zeros_like(want["cand"])
    for r in range(world):
        one.set_bucket_partition(r, world)
        got = one.query(centers, R)
        assert len(got["q"]) > 0
        assert np.all((got["cand"] == 0) | (cand == 0))        # a (query, table) probe belongs to ONE part
        cand += got["cand"]
        parts.append(got)
    assert np.array_equal(cand, want["cand"])                    # ... and to some part
    with pytest.raises(capi.HsError):
        one.set_bucket_partition(world, world)
    one.set_bucket_partition(0, 1)
    again = one.query(centers, R)
    for f in ("q", "id", "table", "dist", "cand"):
        assert np.array_equal(again[f], want[f]), f
    # a part does not depend on how the call is cut into batches (a giant bucket's queries are dealt by their
    # number in the CALL) nor on the verify path (streaming filter: no grouping of the probes at all)
    small = Engine(k, K, L, W, a, b, options=dict(query_batch=301))
    small.index_build(codes)
    for r, mode in ((0, "auto"), (world - 1, "stream")):
        small.set_bucket_partition(r, world)
        small.set_verify_mode(mode)
        got = small.query(centers, R)
        for f in ("q", "id", "table", "dist", "cand"):
            assert np.array_equal(got[f], parts[r][f]), (r, mode, f)
    small.close()
    gq, gi, gt, gd = (np.concatenate([p_[f] for p_ in parts]) for f in ("q", "id", "table", "dist"))
    assert len(gq) >= len(want["q"])
    import torch
    dev = torch.device("cuda", 0)
    tq, ti, tt, td = (torch.from_numpy(x.astype(np.int32) if x.dtype != np.float64 else x).to(dev) for x in (gq, gi, gt, gd))
    kept = one.merge_first_table_dev(tq.data_ptr(), ti.data_ptr(), tt.data_ptr(), td.data_ptr(), len(gq))
    assert kept == len(want["q"])
    assert np.array_equal(tq[:kept].cpu().numpy().astype(np.uint32), want["q"])
    assert np.array_equal(ti[:kept].cpu().numpy().astype(np.uint32), want["id"])
    assert np.array_equal(tt[:kept].cpu().numpy().astype(np.uint32), want["table"])
    assert np.array_equal(td[:kept].cpu().numpy(), want["dist"])
    one.close()
    engs = []
    for r in range(world):
        e = Engine(k, K, L, W, a, b)
        e.index_build(codes)
        engs.append(e)
    comm = cdist.Comm(cdist.LOOPBACK, world, devices=[0] * world)
    for as_codes in (False, True):
        res = [None] * world
        def run(r):
            res[r] = comm.query_buckets(r, engs[r], qcodes if as_codes else centers, R, codes=as_codes)
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for r in range(world):
            for f in ("q", "id", "table", "dist"):
                assert np.array_equal(res[r][f], want[f]), (world, r, f, as_codes)
    # the handles answer for all buckets again after the call
    back = engs[0].query(centers, R, want_cand=False)
    assert np.array_equal(back["id"], want["id"])
    comm.close()
    for e in engs:
        e.close()


def test_assign_tables_is_balanced_and_deterministic():
    cost = np.array([9.0, 1, 1, 1, 8, 1, 1, 1, 7, 1, 1, 1, 1, 1, 1, 1])
    own = cdist.assign_tables(cost, 16, 4)
    loads = [cost[own == r].sum() for r in range(4)]
    assert max(loads) <= 10.0 and sorted(np.bincount(own, minlength=4).tolist())[0] >= 1
    assert np.array_equal(own, cdist.assign_tables(cost, 16, 4))
    assert np.array_equal(cdist.assign_tables(None, 8, 4), np.array([0, 1, 2, 3, 0, 1, 2, 3], dtype=np.uint32))
