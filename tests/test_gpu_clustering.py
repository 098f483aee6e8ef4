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
