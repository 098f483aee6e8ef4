"""Randomised configurations of the whole search path against the oracle: k, K, L, W, R, database
shape (uniform / families / heavy duplicates), query kind (exact copies, mutated, jittered, far),
custom coordinate tables (the points-file route), all verify modes.  Seeds are fixed: a failure
names the case that reproduces it."""
import numpy as np
import pytest

from hsearch_amd import Engine, synth

pytestmark = pytest.mark.gpu


def _case(seed):
    r = np.random.default_rng(1000 + seed)
    k = int(r.choice([1, 2, 3, 7, 12, 15, 24, 25, 25, 25, 26, 33, 39, 50]))
    K = int(r.integers(1, 21))
    L = int(r.integers(1, 9))
    if seed >= 60:      # the second block of cases draws from the whole table range the ABI admits
        L = int(r.integers(9, 33))
    W = float(r.choice([0.7, 5.0, 30.0, 80.0, 150.0, 250.0, 400.0, 1e4]))
    R = float(r.choice([0.0, 1e-9, 10.0, 25.0, 40.0, 55.0, 80.0, 1e6]))
    n = int(r.choice([1, 2, 17, 300, 2000, 6000]))
    nq = int(r.choice([1, 3, 33, 700, 1500]))
    shape = r.choice(["uniform", "families", "duplicates"])
    if shape == "uniform":
        codes = r.integers(0, 20, size=(n, k), dtype=np.uint8)
    else:
        nf = max(1, n // (40 if shape == "families" else 200))
        fam = r.integers(0, 20, size=(nf, k), dtype=np.uint8)
        codes = fam[r.integers(0, nf, size=n)].copy()
        if shape == "families":
            for row in codes:
                for _ in range(int(r.integers(0, 4))):
                    row[r.integers(0, k)] = r.integers(0, 20)
    table = None
    if r.random() < 0.3:    # another alphabet: what a points-file database turns into
        rows = int(r.integers(20, 33))
        table = r.normal(0.0, float(r.choice([0.5, 8.0, 300.0])), size=(rows, 8))
        if r.random() < 0.5:
            table = np.array([[float("%g" % v) for v in row] for row in table])
        codes = r.integers(0, rows, size=(n, k), dtype=np.uint8)
    a = r.standard_normal((L, K, 8 * k))
    b = r.uniform(0.0, W, size=(L, K))
    if r.random() < 0.2:
        b = b - r.uniform(0.0, 3.0) * W          # offsets outside [0, W): negative bucket ints
    emb = (lambda c: synth.embed(c)) if table is None else (lambda c: table[c].reshape(len(c), -1))
    pts = emb(codes)
    src = r.integers(0, n, size=nq)
    q = codes[src].copy()
    kind = r.choice(["copies", "mutated", "jitter", "mixed"])
    alpha = 20 if table is None else len(table)
    if kind != "copies":
        for row in q:
            for _ in range(int(r.integers(0, 5))):
                row[r.integers(0, k)] = r.integers(0, alpha)
    centers = emb(q)
    if kind in ("jitter", "mixed"):
        centers = centers + r.normal(0.0, float(r.choice([1e-3, 0.3, 2.0])), size=centers.shape)
    if kind == "mixed" and nq > 8:
        centers[:3] *= 50.0            # far outside the table: the join filters must step aside
        centers[3:6] += 4.0e4
    return dict(k=k, K=K, L=L, W=W, R=R, a=a, b=b, codes=codes, table=table, pts=pts,
                centers=np.ascontiguousarray(centers), what="%s/%s" % (shape, kind))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("HS_FUZZ_CASES", "84"))))
def test_random_configuration_matches_oracle(oracle, seed):
    import os
    c = _case(seed)
    # every third case with the thin-segment routing on (segments with < 3 probing queries or < 16
    # members to the streaming filter hs_verify_kernel instead of the join); every other remaining case forces
    # the query-resident join kernel for its class; every fourth case takes its centres as POINTS even where
    # they are k-mers (recognise_kmers = 0: hs_qprep8_kernel, hs_finalize_kernel, hs_qtables -- half of the
    # cases have centres that would otherwise run from their residue codes)
    opts = dict(join_min_q=3, join_min_m=16) if seed % 3 == 0 else dict(join_resident=2 * (seed % 2))
    if seed % 4 == 1:
        opts["recognise_kmers"] = 0
    if seed % 5 == 2:      # hs_join8x_kernel's items in XCD-local runs of one chunk of 2..4 items (on by itself only for big batches)
        opts.update(join_xcd_run=1, join_chunk=2 + seed % 3)
    # (HS_TEST_SPLIT_ABOVE in the environment: the library's test build, which has that hook)
    eng = Engine(c["k"], c["K"], c["L"], c["W"], c["a"], c["b"], coords=c["table"],
                 hooks=bool(os.environ.get("HS_TEST_SPLIT_ABOVE")), options=opts)
    info = eng.index_build(c["codes"])
    ix = oracle.Index(c["a"], c["b"], c["W"], c["pts"])
    try:
        assert info["n_buckets"] == ix.table_sizes(), c["what"]
        want = ix.query(c["centers"], c["R"])
        for mode in ("auto", "stream", "join", "join16"):
            eng.set_verify_mode(mode)
            got = eng.query(c["centers"], c["R"])
            assert seed % 4 != 1 or eng.profile()["queries_recognised"] == 0
            assert np.array_equal(got["cand"], want["cand"]), (mode, c["what"])
            for key in ("q", "id", "table", "dist"):
                assert np.array_equal(got[key], want[key]), (mode, key, c["what"])
    finally:
        ix.close()
        eng.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("HS_FUZZ_CLUSTER_CASES", "24"))))
def test_random_clustering_matches_oracle(oracle, seed):
    """Clustering() on random family-structured inputs, single call and in 1-5 edge shards."""
    import hsearch_amd
    r = np.random.default_rng(5000 + seed)
    k = int(r.choice([3, 9, 15, 25, 25, 31]))
    K = int(r.integers(1, 17))
    L = int(r.integers(1, 9))
    W = float(r.choice([40.0, 100.0, 200.0, 500.0]))
    R = float(r.choice([0.0, 20.0, 40.0, 60.0, 90.0]))
    n = int(r.choice([1, 40, 900, 3000]))
    nf = max(1, n // int(r.choice([5, 30, 150])))
    fam = r.integers(0, 20, size=(nf, k), dtype=np.uint8)
    codes = fam[r.integers(0, nf, size=n)].copy()
    for row in codes:
        for _ in range(int(r.integers(0, 4))):
            row[r.integers(0, k)] = r.integers(0, 20)
    a = r.standard_normal((L, K, 8 * k))
    b = r.uniform(0.0, W, size=(L, K))
    want_merged, want_owner = oracle.clustering(a, b, W, R, oracle.embed_codes(codes))
    merged, owner, table = hsearch_amd.clustering(k, K, L, W, a, b, codes, R)
    assert np.array_equal(merged, want_merged) and np.array_equal(owner, want_owner)
    world = int(r.integers(1, 6))
    st = hsearch_amd.ClusterState(k, K, L, W, a, b, codes, R)
    for l in range(L):
        parts = [st.table_edges(l, rk, world) for rk in range(world)]
        st.table_apply(l, np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
    m2, o2, t2 = st.end()
    assert np.array_equal(m2, merged) and np.array_equal(o2, owner) and np.array_equal(t2, table)
