"""BASELINE.json's full sizes, where the CPU oracle cannot run: size-independent properties.

  configs[1]  10 M 25-mers, L=8, K=16, 100 k queries      (search)
  configs[2]  100 M 25-mers, L=32, K=20, one GPU's share  (search; 1 M queries / 8 GPUs = 125 k)
  configs[3]  1 M 25-mers all-vs-all + hclust             (clustering)
  configs[4]  mixed k in {15, 25, 39}                      (three indexes, one launch sequence)
"""
import numpy as np
import pytest

import hsearch_amd
from hsearch_amd import Engine, synth

pytestmark = pytest.mark.gpu


def _exact_d2(codes_rows, centers_rows):
    """Left-to-right fp64 sum of squared differences (motif_both_points.cpp:176-183), vectorised
    over pairs: the accumulation order over i is the reference's."""
    x = synth.embed(codes_rows)
    acc = np.zeros(len(x))
    for i in range(x.shape[1]):
        r = x[:, i] - centers_rows[:, i]
        acc = acc + r * r
    return acc


def test_config2_search_properties():
    # W = 212: the bench default (bench.py; the smallest W of the committed sweep with radius recall
    # >= 0.9), i.e. exactly the workload the headline number is quoted on
    k, K, L, W, R, n, nq = 25, 16, 8, 212.0, 40.0, 10_000_000, 100_000
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(n, k)
    qcodes, src = synth.make_query_codes(codes, nq)
    centers = synth.embed(qcodes)                       # = synth.make_queries(codes, nq): bench.py's queries
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    assert info["n"] == n and sum(info["n_buckets"]) > 0
    res = {}
    for mode in ("join", "join16", "stream"):
        eng.set_verify_mode(mode)
        res[mode] = eng.query(centers, R)
        assert (eng.profile()["join_i8_batches"] > 0) == (mode == "join")
    # the same queries as residue codes (hs_query_codes: no centre is ever embedded with the int8 join)
    eng.set_verify_mode("auto")
    as_codes = eng.query_codes(qcodes, R)
    assert eng.profile()["join_i8_batches"] > 0
    for f in ("q", "id", "table", "dist", "cand"):
        assert np.array_equal(as_codes[f], res["join"][f]), f
    # ... and as POINTS that are not looked at for being k-mers (option recognise_kmers = 0): the path of centres
    # that are no rows of the coordinate table (bench.py's general_centres) on the headline workload
    eng.set_verify_mode("join")
    eng.query(centers, R)
    assert eng.profile()["queries_recognised"] == nq      # (what res["join"] above ran as)
    eng.set_option("recognise_kmers", 0)
    as_points = eng.query(centers, R)
    pp = eng.profile()
    assert pp["join_i8_batches"] > 0 and pp["queries_recognised"] == 0
    eng.set_option("recognise_kmers", 1)
    for f in ("q", "id", "table", "dist", "cand"):
        assert np.array_equal(as_points[f], res["join"][f]), f
    j, s = res["join"], res["stream"]
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(res["join16"][f], s[f]), f
    # the two filter kernels are interchangeable: identical candidates, hits, order, distances
    assert np.array_equal(j["cand"], s["cand"])
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(j[f], s[f]), f
    q, ids, tab, dist = j["q"].astype(np.int64), j["id"].astype(np.int64), j["table"].astype(np.int64), j["dist"]
    # reference output order: query, then table of first sight, then ascending id; no duplicates
    key = (q << 40) | (tab << 32) | ids
    assert np.all(np.diff(key) > 0)
    assert len(np.unique((q << 32) | ids)) == len(q)
    # every reported distance is the exact fp64 value and within R
    sel = np.random.default_rng(0).choice(len(q), size=min(20000, len(q)), replace=False)
    d2 = _exact_d2(codes[ids[sel]], centers[q[sel]])
    assert np.array_equal(np.sqrt(d2), dist[sel])
    assert np.all(d2 <= R * R)
    # a query that is an exact copy of a DB k-mer shares every bucket with it: found in table 0
    exact = np.nonzero((synth.embed(codes[src[:5000]]) == centers[:5000]).all(axis=1))[0]
    assert len(exact) > 500
    hit = set(zip(q.tolist(), ids.tolist()))
    first_table = {(qq, ii): tt for qq, ii, tt in zip(q.tolist(), ids.tolist(), tab.tolist())}
    for qi in exact:
        assert (int(qi), int(src[qi])) in hit
        assert first_table[(int(qi), int(src[qi]))] == 0
    # candidate counts are bucket populations: bounded by the largest bucket of each table
    assert np.all(j["cand"].max(axis=0) <= np.array(info["max_bucket"]))
    # radius recall against the exhaustive scan on a query subsample, and exact top-10 sanity
    sub = centers[:128]
    bf = eng.bruteforce(sub, R)
    truth = set(zip(bf["q"].tolist(), bf["id"].tolist()))
    found = {(a_, b_) for a_, b_ in hit if a_ < 128}
    assert found <= truth                      # LSH never reports a pair the exhaustive scan lacks
    assert len(found) / len(truth) > 0.75
    nn, nd2 = eng.bruteforce_topk(sub[:32], 10)
    assert np.all(np.diff(nd2, axis=1) >= 0)
    for qi in range(32):
        mine = sorted((d_, i_) for (q_, i_), d_ in zip(zip(bf["q"].tolist(), bf["id"].tolist()), bf["dist"].tolist()) if q_ == qi)
        kk = min(len(mine), 10)
        assert [i_ for _, i_ in mine[:kk]] == nn[qi, :kk].tolist()
    eng.close()


def test_config3_shape_properties():
    """BASELINE.json configs[2] as bench.py's secondary block runs it (W = 160, the W its recall sweep picks):
    the 100 M-k-mer index replicated (L = 32, K = 20, ~157 GB of the 288 GB); one GPU's 125 000 of the 10^6
    queries (the 8-GPU job's per-rank share) AND all 10^6 in one batch (the one-GPU point of the scaling curve).
    The oracle cannot run at this size, so: the interchangeable filter kernels agree hit for hit, queries as
    codes and as points agree (recognised or not), a query's hits do not depend on the batch it arrives in,
    the output is in the reference's order (motif_both_points.cpp:224-245), distances are the exact fp64
    values, exact copies are found in table 0, and LSH hits are a subset of the exhaustive scan's."""
    k, K, L, W, R, n, nq, nq_all = 25, 20, 32, 160.0, 40.0, 100_000_000, 125_000, 1_000_000
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(n, k)
    qcodes_all, src_all = synth.make_query_codes(codes, nq_all)
    qcodes, src = qcodes_all[:nq], src_all[:nq]
    centers = synth.embed(qcodes)
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    assert info["n"] == n and len(info["n_buckets"]) == L and min(info["n_buckets"]) > 1000
    assert info["device_bytes"] < 200e9
    eng.set_verify_mode("join")
    j = eng.query(centers, R)
    prof = eng.profile()
    assert prof["join_i8_batches"] > 0 and prof["join_pairs"] > 0.9 * prof["candidates"]
    assert prof["join_items_resident"] > 0.5 * prof["join_items"]       # the query-resident kernel's regime
    assert prof["queries_recognised"] == nq
    # the same queries as residue codes, and as points that are NOT looked at for being k-mers
    c = eng.query_codes(qcodes, R)
    eng.set_option("recognise_kmers", 0)
    p_ = eng.query(centers, R)
    assert eng.profile()["queries_recognised"] == 0
    eng.set_option("recognise_kmers", 1)
    for f in ("q", "id", "table", "dist", "cand"):
        assert np.array_equal(c[f], j[f]), f
        assert np.array_equal(p_[f], j[f]), f
    # the streaming filter on a query subsample (it moves 58 MB per query at this size)
    ns = 20_000
    eng.set_verify_mode("stream")
    s = eng.query(centers[:ns], R)
    assert eng.profile()["join_batches"] == 0
    cut = int(np.searchsorted(j["q"], ns))
    assert np.array_equal(j["cand"][:ns], s["cand"])
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(j[f][:cut], s[f]), f
    eng.set_verify_mode("join16")
    h = eng.query(centers[:ns], R)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(h[f], s[f]), f
    q, ids, tab, dist = j["q"].astype(np.int64), j["id"].astype(np.int64), j["table"].astype(np.int64), j["dist"]
    key = (q << 40) | (tab << 32) | ids
    assert np.all(np.diff(key) > 0)                       # query, table of first sight, ascending id
    assert len(np.unique((q << 32) | ids)) == len(q)      # first-seen dedupe over 32 tables
    assert tab.max() < L and len(np.unique(tab)) > 8      # later tables do contribute first sights
    sel = np.random.default_rng(0).choice(len(q), size=min(20000, len(q)), replace=False)
    d2 = _exact_d2(codes[ids[sel]], centers[q[sel]])
    assert np.array_equal(np.sqrt(d2), dist[sel])
    assert np.all(d2 <= R * R)
    exact = np.nonzero((synth.embed(codes[src[:5000]]) == centers[:5000]).all(axis=1))[0]
    assert len(exact) > 500
    first_table = {(qq, ii): tt for qq, ii, tt in zip(q[:cut].tolist(), ids[:cut].tolist(), tab[:cut].tolist())}
    for qi in exact:
        assert first_table.get((int(qi), int(src[qi]))) == 0
    assert np.all(j["cand"].max(axis=0) <= np.array(info["max_bucket"]))
    # LSH never reports a pair the exhaustive scan lacks; with 32 tables it finds nearly all of them
    sub = centers[:64]
    bf = eng.bruteforce(sub, R)
    truth = set(zip(bf["q"].tolist(), bf["id"].tolist()))
    found = {(a_, b_) for a_, b_ in zip(q[:cut].tolist(), ids[:cut].tolist()) if a_ < 64}
    assert found <= truth
    assert len(found) / len(truth) > 0.85
    # all 10^6 queries in ONE batch (hs_capi.hip run_query sizes batches by the free HBM): the first 125 000
    # must come out exactly as they did in their own batch -- and those equal the streaming filter's above
    eng.set_verify_mode("join")
    big = eng.query_codes(qcodes_all, R, cap=4 * nq_all, want_cand=False)
    prof = eng.profile()
    assert prof["join_batches"] == 1 and prof["join_i8_batches"] == 1     # one batch, through the int8 join
    cut_b = int(np.searchsorted(big["q"], nq))
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(big[f][:cut_b], j[f]), f
    # the same batch again: the first pass at a new size runs with the hints of the batch before it (chunk sizes
    # of the item counters, capacity of the item list -- too small here: it runs twice), the second with its own;
    # nothing of that may show in the hits.  (r04: with chunks of 48 items and XCD-local runs the first pass
    # left 96 of 7.1 million work items unprocessed and lost 1-3 hits of 911 306.)
    again = eng.query_codes(qcodes_all, R, cap=4 * nq_all, want_cand=False)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(again[f], big[f]), f
    # ... nor the length of the XCD-local runs the join's work items are dealt in, nor the chunk size -- chosen so
    # that the chunk straddling the end of hs_join8x_kernel's share of the items has an empty tail of ONE or TWO
    # items, the case that lost chunks
    pr = eng.profile()
    share = pr["join_items"] - pr["join_items_resident"]
    chunks = [g for g in range(8, 65) if (-share) % g in (1, 2)][:3] + [0]
    assert len(chunks) >= 2
    for xr, g in [(1, chunks[0]), (128, chunks[0]), (1024, chunks[1]), (0, chunks[0]), (-1, chunks[-2]), (-1, 0)]:
        eng.set_option("join_xcd_run", xr)
        eng.set_option("join_chunk", g)
        other = eng.query_codes(qcodes_all, R, cap=4 * nq_all, want_cand=False)
        for f in ("q", "id", "table", "dist"):
            assert np.array_equal(other[f], big[f]), (xr, g, f)
    bq, bi, bt = big["q"].astype(np.int64), big["id"].astype(np.int64), big["table"].astype(np.int64)
    assert np.all(np.diff((bq << 40) | (bt << 32) | bi) > 0)
    sel = np.random.default_rng(1).choice(len(bq), size=20000, replace=False)
    d2 = _exact_d2(codes[bi[sel]], synth.embed(qcodes_all[bq[sel]]))
    assert np.array_equal(np.sqrt(d2), big["dist"][sel]) and np.all(d2 <= R * R)
    eng.close()


def test_config4_clustering_properties():
    k, K, L, W, R, n = 25, 16, 8, 200.0, 40.0, 1_000_000
    rng = np.random.default_rng(3)
    fam = rng.integers(0, 20, size=(2000, k), dtype=np.uint8)          # 2000 planted families x 50
    rows = np.repeat(fam, 50, axis=0)
    m = rng.integers(0, 5, size=len(rows))
    for s in range(4):
        sel = np.nonzero(m > s)[0]
        rows[sel, rng.integers(0, k, size=len(sel))] = rng.integers(0, 20, size=len(sel), dtype=np.uint8)
    codes = np.concatenate([rows, synth.make_db(n - len(rows), k, seed=9)])
    rng.shuffle(codes)
    a, b = synth.make_planes(k, K, L, W, seed=77)
    merged, owner, table = hsearch_amd.clustering(k, K, L, W, a, b, codes, R)
    m2, o2, t2 = hsearch_amd.clustering(k, K, L, W, a, b, codes, R)       # deterministic
    assert np.array_equal(merged, m2) and np.array_equal(owner, o2) and np.array_equal(table, t2)
    # the 4-rank form (each table's join cut into 4 blocks, edges pooled) is identical
    st = hsearch_amd.ClusterState(k, K, L, W, a, b, codes, R)
    for l in range(L):
        parts = [st.table_edges(l, r, 4) for r in range(4)]
        st.table_apply(l, np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
    m4, o4, t4 = st.end()
    assert np.array_equal(merged, m4) and np.array_equal(owner, o4) and np.array_equal(table, t4)
    absorbed = np.nonzero(merged == 2)[0]
    assert len(absorbed) > 50_000
    assert np.all(merged[owner[absorbed]] == 1)                 # absorbed only by real centers
    assert np.all(owner[merged != 2] == np.nonzero(merged != 2)[0])
    assert np.all((table != 0xffffffff) == (merged == 2))
    assert set(np.unique(merged)) <= {0, 1, 2}
    assert np.all(np.isin(np.nonzero(merged == 1)[0], owner[absorbed]))   # every center owns someone
    # every absorbed k-mer lies within R of its center (hclust2.cpp:119-120: sqrt form, exact fp64)
    sel = rng.choice(absorbed, size=20000, replace=False)
    d2 = _exact_d2(codes[sel], synth.embed(codes[owner[sel]]))
    assert np.all(np.sqrt(d2) <= R)
    # the clusters file accounts for every k-mer exactly once (num_of_kmers, hclust2.cpp:139,149)
    sizes = np.bincount(owner, minlength=n)[merged != 2]
    assert sizes.sum() == n


def test_config5_mixed_lengths(oracle):
    R, W, K, L = 40.0, 150.0, 8, 4
    for k in (15, 25, 39):
        a, b = synth.make_planes(k, K, L, W, seed=100 + k)
        codes = synth.make_db(200_000, k, seed=k)
        centers, _ = synth.make_queries(codes, 2000, seed=k + 1, jitter=0.1)
        eng = Engine(k, K, L, W, a, b)
        eng.index_build(codes)
        got = eng.query(centers, R)
        ix = oracle.Index(a, b, W, oracle.embed_codes(codes))
        want = ix.query(centers, R)
        for f in ("q", "id", "table", "dist"):
            assert np.array_equal(got[f], want[f]), (k, f)
        assert np.array_equal(got["cand"], want["cand"])
        assert len(want["q"]) > 100
        eng.close()
