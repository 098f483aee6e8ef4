"""The C++ multi-GPU layer (libhsearch_dist.so, include/hsearch_dist.h) without a GPU: exported
symbols, the shard rule, and the variable-length all-gather of hit tuples at world 2..5 over the
host-memory loopback transport -- one host thread per rank, as hs_motif_both_points --gpus n runs
it.  The RCCL transport shares everything but the copy itself; it runs on the GPU box
(tests/test_gpu_cdist.py)."""
import threading

import numpy as np
import pytest

from hsearch_amd import capi, cdist
from hsearch_amd import dist as pydist


def test_library_exports_every_declared_symbol():
    import os
    import re
    lib = cdist.load()
    hdr = open(os.path.join(os.path.dirname(cdist.lib_path()), "..", "include", "hsearch_dist.h")).read()
    declared = set(re.findall(r"HS_API\s+[\w\s\*]+?\b(hs_\w+)\s*\(", hdr))
    assert declared == set(cdist.EXPORTS)
    for s in declared:
        assert hasattr(lib, s), s


def test_shard_bounds_match_the_python_rule():
    for n in (0, 1, 7, 100, 1_000_003):
        for world in (1, 2, 3, 8):
            covered = 0
            for r in range(world):
                lo, hi = cdist.shard_bounds(n, world, r)
                assert (lo, hi) == tuple(pydist.shard_bounds(n, r, world))
                assert lo == covered
                covered = hi
            assert covered == n


def _fake_hits(rng, n, nq_block):
    q = np.sort(rng.integers(0, max(nq_block, 1), size=n)).astype(np.uint32)
    return dict(q=q, id=rng.integers(0, 10**6, size=n).astype(np.uint32),
                table=rng.integers(0, 32, size=n).astype(np.uint32), dist=rng.random(n) * 40.0)


def _run_ranks(world, fn):
    out, err = [None] * world, []

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as e:   # noqa: a failing rank must not leave the others waiting silently
            err.append((r, e))
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ts), "ranks hung"
    assert not err, err
    return out


@pytest.mark.parametrize("world,counts", [(2, [5, 3]), (2, [0, 4]), (2, [0, 0]), (3, [7, 0, 1]),
                                          (5, [1, 2, 3, 4, 1000]), (4, [33, 33, 33, 33])])
def test_loopback_allgather_orders_ranks_and_globalises_queries(world, counts):
    rng = np.random.default_rng(world * 100 + sum(counts))
    nq_total = 1000
    blocks = [cdist.shard_bounds(nq_total, world, r) for r in range(world)]
    local = [_fake_hits(rng, counts[r], blocks[r][1] - blocks[r][0]) for r in range(world)]
    comm = cdist.Comm(cdist.LOOPBACK, world)
    for rounds in range(3):   # repeated calls on one communicator (the barrier generations cycle)
        got = _run_ranks(world, lambda r: comm.gather_host(r, local[r], blocks[r][0]))
        want = {k: np.concatenate([local[r][k] + (blocks[r][0] if k == "q" else 0) for r in range(world)])
                for k in ("q", "id", "table", "dist")}
        for r in range(world):
            for k in want:
                assert np.array_equal(got[r][k], want[k]), (r, k)
            assert np.all(np.diff(got[r]["q"].astype(np.int64)) >= 0)   # global query order
    # without the table column
    got = _run_ranks(world, lambda r: comm.gather_host(r, local[r], blocks[r][0], with_table=False))
    assert all(g["table"] is None for g in got)
    assert np.array_equal(got[0]["id"], np.concatenate([l["id"] for l in local]))
    comm.close()


def test_capacity_protocol_is_one_decision_for_all_ranks():
    world = 3
    comm = cdist.Comm(cdist.LOOPBACK, world)
    rng = np.random.default_rng(9)
    local = [_fake_hits(rng, n, 10) for n in (4, 5, 6)]
    caps = [100, 9, 100]   # rank 1 offers too little: EVERY rank must report the need and write nothing

    def call(r):
        out = dict(q=np.full(caps[r], 77, np.uint32), id=np.empty(caps[r], np.uint32),
                   table=np.empty(caps[r], np.uint32), dist=np.empty(caps[r], np.float64))
        st, tot = comm.allgather_hits(r, local[r]["q"], local[r]["id"], local[r]["table"], local[r]["dist"],
                                      len(local[r]["q"]), 0, out["q"], out["id"], out["table"], out["dist"], caps[r])
        return st, tot, out["q"]
    for st, tot, q in _run_ranks(world, call):
        assert st == capi.HS_ERR_CAPACITY and tot == 15
        assert np.all(q == 77)
    comm.close()


def test_python_and_cxx_gathers_agree_over_two_real_processes():
    """dist.allgather_hits (torch.distributed, what bench.py uses) and hs_allgather_hits give the
    same list: checked through the loopback comm against the gloo world-2 result of the same
    per-rank inputs, computed in-process (both reduce to rank-order concatenation)."""
    rng = np.random.default_rng(4)
    world = 2
    blocks = [cdist.shard_bounds(501, world, r) for r in range(world)]
    local = [_fake_hits(rng, n, blocks[r][1] - blocks[r][0]) for r, n in enumerate((40, 17))]
    comm = cdist.Comm(cdist.LOOPBACK, world)
    got = _run_ranks(world, lambda r: comm.gather_host(r, local[r], blocks[r][0]))
    comm.close()
    cat = {k: np.concatenate([local[r][k] + (blocks[r][0] if k == "q" else 0) for r in range(world)])
           for k in ("q", "id", "table", "dist")}
    for k in cat:
        assert np.array_equal(got[0][k], cat[k]) and np.array_equal(got[1][k], cat[k])


@pytest.mark.parametrize("world,bad", [(2, 1), (3, 0), (4, 2)])
def test_a_failed_rank_stops_every_rank_and_nobody_waits(world, bad):
    """a rank that fails before the exchange must still take part in it.  Rank `bad` offers
    hits without arrays (HS_ERR_INVALID on that rank): it returns its own status, every other rank
    HS_ERR_PEER, nothing is written, no thread is left at a rendezvous -- and the communicator keeps
    working for the next (healthy) call."""
    rng = np.random.default_rng(50 + world)
    local = [_fake_hits(rng, 5 + r, 10) for r in range(world)]
    comm = cdist.Comm(cdist.LOOPBACK, world)

    def broken(r):
        out = dict(q=np.full(64, 77, np.uint32), id=np.empty(64, np.uint32), table=np.empty(64, np.uint32),
                   dist=np.empty(64, np.float64))
        h = local[r]
        args = (None, None, None, None) if r == bad else (h["q"], h["id"], h["table"], h["dist"])
        st, tot = comm.allgather_hits(r, *args, len(h["q"]), 0, out["q"], out["id"], out["table"], out["dist"], 64)
        return st, out["q"]
    for r, (st, q) in enumerate(_run_ranks(world, broken)):
        assert st == (capi.HS_ERR_INVALID if r == bad else capi.HS_ERR_PEER), (r, st)
        assert np.all(q == 77)
    assert "failed" in comm.last_error((bad + 1) % world)
    got = _run_ranks(world, lambda r: comm.gather_host(r, local[r], 0))
    assert np.array_equal(got[0]["id"], np.concatenate([l["id"] for l in local]))
    comm.close()


def test_comm_query_is_refused_without_devices_on_every_rank():
    """hs_comm_query over a loopback communicator created WITHOUT devices: every rank gets
    HS_ERR_INVALID through the exchange (no early return: both threads come back)."""
    world = 2
    comm = cdist.Comm(cdist.LOOPBACK, world)
    sts = _run_ranks(world, lambda r: comm.query_status(r, None, np.zeros((1, 200)), 0, 40.0))
    assert sts == [capi.HS_ERR_INVALID] * world
    comm.close()
