"""The CPU oracle against the golden vectors dumped from the compiled reference
(tools/gen_golden.py).  This is what pins the oracle on machines where /root/reference is absent."""
import json
import os
import tempfile

import numpy as np
import pytest


def _load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_hash_dots_buckets_keys(oracle, golden_dir):
    for case in _load(golden_dir, "hash.json")["cases"]:
        z = np.load(os.path.join(golden_dir, case["file"]))
        pts = np.concatenate([oracle.embed_codes(z["codes"]), z["arb"]])
        for l in range(case["L"]):
            buckets, dots = oracle.hash_table(z["a"][l], z["b"][l], case["W"], pts, want_dots=True)
            assert np.array_equal(dots, z["dots"][:, l])          # fp64, same rounding sequence
            assert np.array_equal(buckets, z["buckets"][:, l])
            for i in range(len(pts)):
                assert oracle.key_string(buckets[i]) == case["keys"][i][l]


def test_search_hits_order_and_text(oracle, golden_dir):
    for case in _load(golden_dir, "search.json")["cases"]:
        z = np.load(os.path.join(golden_dir, case["file"]))
        res = oracle.search(z["a"], z["b"], case["W"], case["R"], oracle.embed_codes(z["codes"]),
                            z["centers"])
        assert np.array_equal(res["q"], z["hit_q"])
        assert np.array_equal(res["id"], z["hit_id"])
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "hits.txt")
            oracle.write_hits(p, res["q"], res["id"], res["dist"])
            text = [line.split()[2] for line in open(p)]
        assert text == case["hit_dist_text"]
        # first-seen table order: non-decreasing table inside a query
        for q in np.unique(res["q"]):
            t = res["table"][res["q"] == q]
            assert np.all(np.diff(t.astype(np.int64)) >= 0)


def test_pairwise_square(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "pairwise.npz"))
    got = oracle.pairwise_square(oracle.embed_codes(z["codes"]), z["centers"])
    assert np.array_equal(got, z["dist2"])


def test_bruteforce_consistent_with_pairwise(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "pairwise.npz"))
    db = oracle.embed_codes(z["codes"])
    R = 60.0
    bf = oracle.bruteforce(db, z["centers"], R)
    want = [(q, j) for q in range(len(z["centers"])) for j in range(len(db))
            if not (np.sqrt(z["dist2"][q, j]) > R)]
    assert list(zip(bf["q"].tolist(), bf["id"].tolist())) == want
    nn, d2 = oracle.bruteforce_topk(db, z["centers"], 10)
    for q in range(len(z["centers"])):
        order = np.lexsort((np.arange(len(db)), z["dist2"][q]))[:10]
        assert np.array_equal(nn[q], order)
        assert np.array_equal(d2[q], z["dist2"][q][order])


def test_clustering_file(oracle, golden_dir):
    for case in _load(golden_dir, "clustering.json")["cases"]:
        z = np.load(os.path.join(golden_dir, case["file"]))
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "clusters.txt")
            oracle.clustering_to_file(z["a"], z["b"], case["W"], case["R"],
                                      oracle.embed_codes(z["codes"]), p)
            assert open(p).read() == case["clusters_file"]


def test_evaluate(oracle, golden_dir):
    g = _load(golden_dir, "evaluate.json")
    with tempfile.TemporaryDirectory() as d:
        gt, hits = os.path.join(d, "gt"), os.path.join(d, "hits")
        open(gt, "w").write("\n".join(g["ground_truth"]) + "\n")
        open(hits, "w").write("\n".join(g["hits"]) + "\n")
        assert oracle.evaluate(gt, hits, g["R"]) == pytest.approx(g["weighted_recall"], abs=1e-15)


def test_klsh_oracle_and_planes_match_reference_golden(oracle, golden_dir):
    """the KLSH restatement (planes from the default-seeded engine, 3-mer
    features, serial dot, cos + threshold) against codes produced by the reference's own KLSH object
    (tests/golden/klsh.npz, tools/gen_golden.py::gen_klsh); the product's host-side plane
    generator must give the same planes."""
    import hsearch_amd
    z = np.load(os.path.join(golden_dir, "klsh.npz"))
    w, b, t = oracle.klsh_draw_planes(512, 16, 0.2)
    assert np.array_equal(w, z["w"]) and np.array_equal(b, z["b"]) and np.array_equal(t, z["t"])
    pw, pb, pt = hsearch_amd.klsh_draw_planes(512, 16, 0.2)
    assert np.array_equal(pw, z["w"]) and np.array_equal(pb, z["b"]) and np.array_equal(pt, z["t"])
    st = z["seq_start"].astype(np.int64)
    for i in range(len(st) - 1):
        f = oracle.klsh_features(z["classes"][st[i]:st[i + 1]])
        assert f.sum() == st[i + 1] - st[i] - 2
        assert oracle.klsh_hash(w, b, t, f) == int(z["codes"][i])
    assert str(z["first_sequences"][0]) and list(oracle.klsh_classes(str(z["first_sequences"][0]))) == \
        list(z["classes"][st[0]:st[1]])


# ---- row a11 as a program and SURVEY 8(f) row 4 (tools.json: the real reference's files) --------
def _sha(text):
    import hashlib
    return hashlib.sha256(text.encode()).hexdigest()


def test_tools_nolsh_files(oracle, golden_dir):
    t = _load(golden_dir, "tools.json")["nolsh"]
    z = np.load(os.path.join(golden_dir, t["file"]))
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "bf.txt")
        oracle.bruteforce_to_files(oracle.embed_codes(z["codes"]), z["centers"], t["R"], p)
        assert open(p).read() == t["hits"]
        rest = open(p + "notlessthan.txt").read()
        assert rest.split("\n")[:5] == t["notlessthan_head"]
        assert rest.count("\n") == t["notlessthan_lines"] and _sha(rest) == t["notlessthan_sha256"]


def test_tools_evaluate2_sort_and_weight(oracle, golden_dir):
    t = _load(golden_dir, "tools.json")["evaluate2"]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "hits.txt")
        open(p, "w").write(t["hits"])
        assert oracle.sort_hits_file(p) == t["hits"].count("\n")
        assert open(p + "sort.txt").read() == t["sorted"]
    assert [oracle.evaluate2_weight(x) for x in t["weight_in"]] == t["weight_out"]


def test_tools_cluster2datapoint(oracle, golden_dir):
    t = _load(golden_dir, "tools.json")["cluster2datapoint"]
    fams = [np.array([oracle.letters_to_codes(s_)[0] for s_ in f], dtype=np.uint8) for f in t["families"]]
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "hclust.format.txt")
        oracle.write_points_file(p, t["names"], oracle.family_centers(fams))
        assert open(p).read() == t["points_file"]


def test_tools_center_sampling(oracle, golden_dir):
    t = _load(golden_dir, "tools.json")["center_sampling"]
    codes = np.load(os.path.join(golden_dir, t["file"]))["codes"]
    # the reference reads the database back from the 6-significant-digit points file
    pts = np.array([[float("%g" % x) for x in row] for row in oracle.embed_codes(codes)])
    fams, cur = [], None
    for line in t["families_file"].split("\n"):      # main() :443-457, MIN_SIZE_CLUSTER 50
        if not line:
            continue
        if line[0] == "#":
            cur = []
            fams.append(cur)
        else:
            cur.append(oracle.letters_to_codes(line)[0])
    fams = [np.array(f, dtype=np.uint8) for f in fams if len(f) >= 50]
    assert len(fams) == 3
    with tempfile.TemporaryDirectory() as d:
        pi, pr = os.path.join(d, "inner.txt"), os.path.join(d, "rand.txt")
        oracle.center_sampling(pts, oracle.family_centers(fams), pi, pr)
        assert open(pi).read() == t["innercenter"]
        rand = open(pr).read()
        assert rand.split("\n")[:8] == t["random_head"] and rand.split("\n")[-9:-1] == t["random_tail"]
        assert rand.count("\n") == t["random_lines"] and _sha(rand) == t["random_sha256"]


def test_multithreaded_query_extension_equals_the_single_threaded_loop():
    """hso_index_query_mt (bench.py's all-cores CPU figure; not in the reference) returns the hits of
    hso_index_query in the same order, whatever the thread count."""
    import numpy as np
    from oracle import pyoracle as O
    from hsearch_amd import synth
    k, K, L, W, R = 25, 4, 4, 120.0, 45.0
    a, b = synth.make_planes(k, K, L, W, seed=3)
    codes = synth.make_db(4000, k, seed=4)
    centers, _ = synth.make_queries(codes, 257, seed=5, jitter=0.2)
    ix = O.Index(a, b, W, O.embed_codes(codes))
    one = ix.query(centers, R)
    assert len(one["q"]) > 50
    for threads in (1, 3, 8, 300):
        mt = ix.query_mt(centers, R, threads)
        for key in ("q", "id", "table", "dist"):
            assert np.array_equal(mt[key], one[key])
    ix.close()
