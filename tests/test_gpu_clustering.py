"""Row a12: Clustering() (hclust2.cpp:86-151) on the GPU path against the golden clusters files
dumped from the compiled reference, and against the CPU oracle on fresh seeded inputs."""
import json
import os

import numpy as np
import pytest

import hsearch_amd
from hsearch_amd import Engine, synth

pytestmark = pytest.mark.gpu


def test_clustering_matches_reference_golden(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "clustering.json")))
    for case in g["cases"]:
        z = np.load(os.path.join(golden_dir, case["file"]))
        merged, owner, table = hsearch_amd.clustering(case["k"], case["K"], case["L"], case["W"],
                                                      z["a"], z["b"], z["codes"], case["R"])
        assert hsearch_amd.clusters_file_text(merged, owner, table) == case["clusters_file"]


def _families(rng, k, fams, per, max_sub=4):
    rows = []
    for f in rng.integers(0, 20, size=(fams, k)):
        for _ in range(per):
            row = f.copy()
            for _ in range(int(rng.integers(0, max_sub + 1))):
                row[rng.integers(0, k)] = rng.integers(0, 20)
            rows.append(row)
    rows = np.array(rows, dtype=np.uint8)
    rng.shuffle(rows)
    return rows


@pytest.mark.parametrize("k,K,L,W,R,fams,per", [(25, 4, 8, 100.0, 60.0, 40, 50),
                                                (25, 16, 8, 200.0, 40.0, 60, 40),
                                                (25, 2, 4, 300.0, 45.0, 30, 80)])
def test_clustering_matches_oracle(oracle, k, K, L, W, R, fams, per):
    rng = np.random.default_rng(17)
    codes = np.concatenate([_families(rng, k, fams, per), synth.make_db(3000, k, seed=2)])
    rng.shuffle(codes)
    a, b = synth.make_planes(k, K, L, W, seed=41)
    want_merged, want_owner = oracle.clustering(a, b, W, R, oracle.embed_codes(codes))
    merged, owner, table = hsearch_amd.clustering(k, K, L, W, a, b, codes, R)
    assert np.array_equal(merged, want_merged)
    assert np.array_equal(owner, want_owner)
    assert (merged == 2).sum() > 100            # families really were merged
    assert ((table != 0xffffffff) == (merged == 2)).all()


@pytest.mark.parametrize("world", [2, 3, 7])
def test_clustering_sharded_equals_single(oracle, world):
    """the join of every table cut into `world` blocks of the active k-mers
    (what each rank computes), edges pooled, greedy pass applied -- identical to hs_clustering
    and to the oracle.  The pooling stands in for the all-gather (tests/test_dist_cpu.py covers
    the collective itself)."""
    from hsearch_amd import dist as hdist
    k, K, L, W, R = 25, 4, 8, 100.0, 60.0
    rng = np.random.default_rng(23)
    codes = np.concatenate([_families(rng, k, 40, 50), synth.make_db(2000, k, seed=9)])
    rng.shuffle(codes)
    a, b = synth.make_planes(k, K, L, W, seed=43)
    single = hsearch_amd.clustering(k, K, L, W, a, b, codes, R)
    st = hsearch_amd.capi.ClusterState(k, K, L, W, a, b, codes, R)
    n_edges = []
    for l in range(L):
        parts = [st.table_edges(l, r, world) for r in range(world)]
        n_edges.append([len(p[0]) for p in parts])
        # ranks' lists arrive in rank order from the all-gather; apply must not depend on it
        order = rng.permutation(world)
        st.table_apply(l, np.concatenate([parts[r][0] for r in order]),
                       np.concatenate([parts[r][1] for r in order]))
    got = st.end()
    for x, y in zip(got, single):
        assert np.array_equal(x, y)
    want_merged, want_owner = oracle.clustering(a, b, W, R, oracle.embed_codes(codes))
    assert np.array_equal(got[0], want_merged) and np.array_equal(got[1], want_owner)
    assert sum(map(sum, n_edges)) > 1000 and min(map(max, n_edges)) > 0
    # the driver with world = 1 (no process group) is the same computation
    drv = hdist.clustering_sharded(k, K, L, W, a, b, codes, R)
    for x, y in zip(drv, single):
        assert np.array_equal(x, y)


def test_self_join_range_partitions_the_edges():
    k, K, L, W, R = 25, 4, 3, 120.0, 50.0
    rng = np.random.default_rng(6)
    codes = _families(rng, k, 25, 40)
    a, b = synth.make_planes(k, K, L, W, seed=3)
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    full = eng.self_join(R)
    cuts = [0, 1, 333, 334, 1000]
    parts = [eng.self_join(R, first=lo, count=hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
    for key in ("i", "j", "table", "dist"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), full[key])
    for p, lo, hi in zip(parts, cuts[:-1], cuts[1:]):
        assert ((p["i"] >= lo) & (p["i"] < hi)).all()
    with pytest.raises(hsearch_amd.HsError):
        eng.self_join(R, first=900, count=200)


def test_self_join_edges_match_bruteforce_within_buckets(oracle):
    k, K, L, W, R = 25, 4, 3, 120.0, 50.0
    rng = np.random.default_rng(5)
    codes = _families(rng, k, 25, 40)
    a, b = synth.make_planes(k, K, L, W, seed=3)
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    e = eng.self_join(R, sqrt_test=True)
    pts = oracle.embed_codes(codes)
    ints = oracle.hash_all(a, b, W, pts)
    d2 = oracle.pairwise_square(pts, pts)
    want = []
    for i in range(len(codes)):
        for j in range(len(codes)):
            if i == j or not (np.sqrt(d2[i, j]) <= R):
                continue
            shared = [l for l in range(L) if hsearch_amd.key_string(ints[i, l]) == hsearch_amd.key_string(ints[j, l])]
            if shared:
                want.append((i, shared[0], j, np.sqrt(d2[i, j])))
    want.sort()
    got = list(zip(e["i"].tolist(), e["table"].tolist(), e["j"].tolist(), e["dist"].tolist()))
    assert got == want and len(want) > 1000
    eng.close()


@pytest.mark.parametrize("k,K,L,W,R", [(25, 4, 3, 120.0, 50.0), (39, 6, 4, 200.0, 60.0), (12, 3, 2, 90.0, 30.0),
                                      (25, 4, 3, 120.0, 171.0)])
def test_self_join_from_codes_equals_embedded_queries(k, K, L, W, R):
    """The self-join's two routes -- per-query rows from the residue codes (no centres, no hashing, no
    directory search) and the ordinary query path over embedded centres -- give the same edges, in
    the same order, with the same distances.  R = 171 is past what the int8 filter's digits carry
    (R^2 < 30000 still): that call must take the embedded route by itself."""
    rng = np.random.default_rng(k + int(R))
    codes = np.concatenate([_families(rng, k, 40, 30), synth.make_db(3000, k, seed=4)])
    a, b = synth.make_planes(k, K, L, W, seed=3)
    out = []
    for route in ("codes", "centres", "codes-min"):
        opts = {}
        if route == "centres":
            opts = dict(self_codes=0)
        if route == "codes-min":  # thin segments leave the join: the int8 thin filter serves them
            opts = dict(join_min_q=3, join_min_m=16)
        eng = Engine(k, K, L, W, a, b, options=opts)
        eng.index_build(codes)
        for sq in (False, True):
            out.append((route, sq, eng.self_join(R, sqrt_test=sq), eng.self_join(R, first=100, count=2345, sqrt_test=sq)))
        eng.close()
    ref = {(sq): (full, part) for route, sq, full, part in out if route == "centres"}
    assert len(ref[False][0]["i"]) > 1000
    for route, sq, full, part in out:
        for got, want in zip((full, part), ref[sq]):
            for key in ("i", "j", "table", "dist"):
                assert np.array_equal(got[key], want[key]), (route, sq, key)
