"""The CPU oracle against the real reference compiled into oracle/_ref, on fresh seeded inputs.
Runs only where oracle/_ref exists (the build container, or a box the .so travelled to)."""
import os
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")

LETTERS = "ARNDCQEGHILKMFPSTWYV"


def _data(seed, n, nq, k):
    r = np.random.default_rng(seed)
    codes = r.integers(0, 20, (n, k)).astype(np.uint8)
    qc = codes[r.integers(0, n, nq)].copy()
    for i in range(nq):
        for _ in range(r.integers(0, 5)):
            qc[i, r.integers(0, k)] = r.integers(0, 20)
    centers = O.embed_codes(qc) + r.normal(0, 0.3, (nq, 8 * k))
    return codes, centers


def test_constants():
    from hsearch_amd import synth
    coords, dist2, base = O.ref_constants()
    assert np.array_equal(coords, synth.coords())


@pytest.mark.parametrize("k,K,L,W,R", [(25, 4, 4, 100.0, 40.0), (25, 16, 8, 200.0, 40.0),
                                       (15, 6, 5, 7.0, 30.0), (39, 4, 3, 0.9, 50.0)])
def test_hash_and_search(k, K, L, W, R):
    codes, centers = _data(3, 3000, 150, k)
    db = O.embed_codes(codes)
    seqs = ["".join(LETTERS[c] for c in row) for row in codes]
    assert np.array_equal(db, O.ref_kmer_to_coordinates(seqs, k))
    a, b = O.ref_planes(11, 8 * k, K, L, W)
    for l in range(L):
        bk, dots, keys = O.ref_hash_table(a[l], b[l], W, db[:300], want_keys=True)
        bo, do = O.hash_table(a[l], b[l], W, db[:300], want_dots=True)
        assert np.array_equal(bk, bo) and np.array_equal(dots, do)
        assert all(O.key_string(bo[i]) == keys[i] for i in range(300))
    rq, rid, rdist = O.ref_search(11, db, centers, K, L, W, R)
    res = O.search(a, b, W, R, db, centers)
    assert np.array_equal(rq, res["q"]) and np.array_equal(rid, res["id"])
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "h")
        O.write_hits(p, res["q"], res["id"], res["dist"])
        assert [line.split()[2] for line in open(p)] == rdist


def test_clustering():
    r = np.random.default_rng(9)
    k = 25
    rows = []
    for f in r.integers(0, 20, (30, k)):
        for _ in range(40):
            row = f.copy()
            for _ in range(r.integers(0, 5)):
                row[r.integers(0, k)] = r.integers(0, 20)
            rows.append(row)
    rows = np.array(rows, dtype=np.uint8)
    r.shuffle(rows)
    seqs = ["".join(LETTERS[c] for c in row) for row in rows]
    with tempfile.TemporaryDirectory() as d:
        p1, p2 = os.path.join(d, "ref"), os.path.join(d, "port")
        O.ref_clustering_file(5, seqs, k, 4, 8, 100.0, 60.0, p1)
        a, b = O.ref_planes(5, 200, 4, 8, 100.0)
        O.clustering_to_file(a, b, 100.0, 60.0, O.embed_codes(rows), p2)
        assert open(p1).read() == open(p2).read()


def _have(name):
    return os.path.exists(os.path.join(os.path.dirname(O.__file__), "_ref", name))


@pytest.mark.skipif(not _have("libref_nolsh.so"), reason="oracle/_ref/libref_nolsh.so not built")
@pytest.mark.parametrize("k,R", [(25, 60.0), (15, 35.0), (39, 80.0)])
def test_nolsh_program_files(k, R):
    codes, centers = _data(50 + k, 250, 9, k)
    db = O.embed_codes(codes)
    with tempfile.TemporaryDirectory() as d:
        O.ref_nolsh_search(db, centers, R, d + "/r.txt")
        O.bruteforce_to_files(db, centers, R, d + "/o.txt")
        for suf in ("", "notlessthan.txt"):
            assert open(d + "/r.txt" + suf).read() == open(d + "/o.txt" + suf).read()
        assert 0 < os.path.getsize(d + "/r.txt")
        # evaluate2 on the hits file the reference just wrote
        if _have("libref_evaluate2.so"):
            open(d + "/o2.txt", "w").write(open(d + "/r.txt").read())
            O.ref_evaluate2_sort(d + "/r.txt")
            O.sort_hits_file(d + "/o2.txt")
            assert open(d + "/r.txtsort.txt").read() == open(d + "/o2.txtsort.txt").read()


@pytest.mark.skipif(not _have("libref_evaluate2.so"), reason="oracle/_ref/libref_evaluate2.so not built")
def test_evaluate2_weight():
    for x in np.concatenate([np.linspace(0, 120, 481), [49.38, 98.76]]):
        assert O.ref_evaluate2_weight(float(x)) == O.evaluate2_weight(float(x))


@pytest.mark.skipif(not _have("libref_centers.so"), reason="oracle/_ref/libref_centers.so not built")
def test_cluster2datapoint_points_file():
    r = np.random.default_rng(8)
    for k in (25, 9):
        fams = [r.integers(0, 20, (m, k)).astype(np.uint8) for m in (50, 3, 200, 1)]
        names = ["#f%d with spaces" % i for i in range(len(fams))]
        seqs = [["".join(LETTERS[c] for c in row) for row in f] for f in fams]
        with tempfile.TemporaryDirectory() as d:
            O.ref_cluster2datapoint(k, names, seqs, d + "/r_")
            O.write_points_file(d + "/o.txt", names, O.family_centers(fams))
            assert open(d + "/r_hclust.format.txt").read() == open(d + "/o.txt").read()


@pytest.mark.skipif(not _have("libref_hclust3.so"), reason="oracle/_ref/libref_hclust3.so not built")
def test_hclust3_is_hclust2():
    """hclust3.cpp embeds a k-mer each time it is looked at instead of once; same arithmetic, same
    clusters file -- so the hclust2 operator (oracle, hs_clustering, hs_hclust2) covers it."""
    r = np.random.default_rng(19)
    k = 25
    rows = []
    for f in r.integers(0, 20, (25, k)):
        for _ in range(30):
            row = f.copy()
            for _ in range(r.integers(0, 5)):
                row[r.integers(0, k)] = r.integers(0, 20)
            rows.append(row)
    rows = np.array(rows, dtype=np.uint8)
    r.shuffle(rows)
    seqs = ["".join(LETTERS[c] for c in row) for row in rows]
    with tempfile.TemporaryDirectory() as d:
        p2, p3, po = os.path.join(d, "h2"), os.path.join(d, "h3"), os.path.join(d, "port")
        O.ref_clustering_file(7, seqs, k, 6, 5, 120.0, 55.0, p2)
        O.ref_hclust3_clustering_file(7, seqs, k, 6, 5, 120.0, 55.0, p3)
        a, b = O.ref_planes(7, 200, 6, 5, 120.0)
        O.clustering_to_file(a, b, 120.0, 55.0, O.embed_codes(rows), po)
        assert open(p2).read() == open(p3).read() == open(po).read()
        assert open(p2).read().count("#clusterid") < len(seqs)
