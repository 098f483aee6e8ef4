#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference compiled in this container (oracle/_ref).

Runs only where /root/reference exists (the build container).  It writes data only -- inputs and
the reference's outputs on them -- never reference source text.  The fixtures pin the CPU oracle
(tests/test_oracle_golden.py) and, through it, the HIP path on the GPU box where the reference
cannot travel.

    python tools/gen_golden.py        # rewrites tests/golden/
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LETTERS = "ARNDCQEGHILKMFPSTWYV"  # row order of the table (base[], util.hpp:92)


def rng(seed):
    return np.random.Generator(np.random.MT19937(seed))


def seqs_of(codes):
    return ["".join(LETTERS[c] for c in row) for row in codes]


def gen_constants():
    coords, dist2, base = O.ref_constants()
    json.dump({"source": "util.hpp:21-64,92 as compiled into oracle/_ref",
               "coordinates": coords.tolist(), "DISTANCE_SQUARE": dist2.tolist(),
               "base": base.tolist()},
              open(os.path.join(OUT, "constants.json"), "w"))


def gen_hash():
    """LSH::DotProduct / HashBucketIndex / HashKey (lsh.hpp:33-59) with planes drawn by the
    reference's own constructor."""
    cases = []
    r = rng(101)
    for ci, (k, K, L, W, seed) in enumerate([(25, 4, 2, 100.0, 7), (25, 16, 1, 200.0, 8),
                                             (15, 3, 2, 2.5, 9), (39, 5, 1, 0.75, 10),
                                             (25, 2, 2, 3.0, 11)]):
        d = 8 * k
        a, b = O.ref_planes(seed, d, K, L, W)
        codes = r.integers(0, 20, size=(40, k), dtype=np.uint8)
        pts = O.ref_kmer_to_coordinates(seqs_of(codes), k)
        arb = pts[:10] + r.normal(0, 0.7, size=(10, d))  # arbitrary centres (not table points)
        allpts = np.concatenate([pts, arb])
        dots, buckets, keys = [], [], []
        for l in range(L):
            bk, dt, ks = O.ref_hash_table(a[l], b[l], W, allpts, want_keys=True)
            dots.append(dt)
            buckets.append(bk)
            keys.append(ks)
        name = "hash_case%d" % ci
        np.savez_compressed(os.path.join(OUT, name + ".npz"), a=a, b=b, codes=codes, arb=arb,
                            dots=np.stack(dots, 1), buckets=np.stack(buckets, 1))
        cases.append({"file": name + ".npz", "k": k, "K": K, "L": L, "W": W, "plane_seed": seed,
                      "keys": [[keys[l][i] for l in range(L)] for i in range(len(allpts))]})
    json.dump({"source": "lsh.hpp:33-59 via oracle/_ref ref_hash; points 0..39 are "
                         "KmerToCoordinates(codes) (hclust2.cpp:49-62), 40..49 are `arb`",
               "cases": cases}, open(os.path.join(OUT, "hash.json"), "w"))


def gen_search():
    """Search() (motif_both_points.cpp:195-250): hits in file order with the printed distance."""
    cases = []
    for ci, (k, K, L, W, R, n, nq, seed) in enumerate([
            (25, 4, 4, 100.0, 40.0, 1500, 120, 21),   # the binary's hard-coded K=L=4
            (25, 16, 8, 200.0, 40.0, 1500, 80, 22),   # BASELINE configs[1] shape
            (15, 3, 3, 4.0, 30.0, 800, 60, 23),       # small W: key strings alias
            (39, 6, 3, 150.0, 55.0, 600, 50, 24)]):
        r = rng(1000 + ci)
        d = 8 * k
        codes = r.integers(0, 20, size=(n, k), dtype=np.uint8)
        src = r.integers(0, n, size=nq)
        qc = codes[src].copy()
        for i in range(nq):
            for _ in range(int(r.integers(0, 5))):
                qc[i, r.integers(0, k)] = r.integers(0, 20)
        db = O.ref_kmer_to_coordinates(seqs_of(codes), k)
        centers = O.ref_kmer_to_coordinates(seqs_of(qc), k)
        centers[nq // 2:] += r.normal(0, 0.4, size=(nq - nq // 2, d))  # half arbitrary points
        a, b = O.ref_planes(seed, d, K, L, W)
        hq, hid, hdist = O.ref_search(seed, db, centers, K, L, W, R)
        bf_lines = None
        name = "search_case%d" % ci
        np.savez_compressed(os.path.join(OUT, name + ".npz"), a=a, b=b, codes=codes,
                            centers=centers, hit_q=hq, hit_id=hid)
        cases.append({"file": name + ".npz", "k": k, "K": K, "L": L, "W": W, "R": R,
                      "plane_seed": seed, "hit_dist_text": hdist})
    json.dump({"source": "motif_both_points.cpp:195-250 via oracle/_ref ref_search; names are "
                         "decimal indices; hit_dist_text is the third column of the hits file",
               "cases": cases}, open(os.path.join(OUT, "search.json"), "w"))


def gen_pairwise():
    r = rng(77)
    codes = r.integers(0, 20, size=(64, 25), dtype=np.uint8)
    db = O.ref_kmer_to_coordinates(seqs_of(codes), 25)
    centers = db[:8] + r.normal(0, 0.5, size=(8, 200))
    d2 = O.ref_pairwise_square(db, centers)
    np.savez_compressed(os.path.join(OUT, "pairwise.npz"), codes=codes, centers=centers, dist2=d2)


def gen_clustering():
    """Clustering() (hclust2.cpp:86-151): the clusters file, byte for byte."""
    cases = []
    for ci, (k, K, L, W, R, fams, per, seed) in enumerate([(25, 4, 8, 100.0, 60.0, 12, 25, 31),
                                                           (25, 16, 8, 200.0, 40.0, 10, 30, 32),
                                                           (15, 4, 4, 60.0, 35.0, 8, 20, 33)]):
        r = rng(2000 + ci)
        rows = []
        for f in r.integers(0, 20, size=(fams, k)):
            for _ in range(per):
                row = f.copy()
                for _ in range(int(r.integers(0, 5))):
                    row[r.integers(0, k)] = r.integers(0, 20)
                rows.append(row)
        rows = np.array(rows, dtype=np.uint8)
        r.shuffle(rows)
        fd, path = tempfile.mkstemp()
        os.close(fd)
        O.ref_clustering_file(seed, seqs_of(rows), k, K, L, W, R, path)
        text = open(path).read()
        os.unlink(path)
        a, b = O.ref_planes(seed, 8 * k, K, L, W)
        name = "clustering_case%d" % ci
        np.savez_compressed(os.path.join(OUT, name + ".npz"), a=a, b=b, codes=rows)
        cases.append({"file": name + ".npz", "k": k, "K": K, "L": L, "W": W, "R": R,
                      "plane_seed": seed, "clusters_file": text})
    json.dump({"source": "hclust2.cpp:86-151 via oracle/_ref ref2_clustering; member names are "
                         "decimal indices", "cases": cases},
              open(os.path.join(OUT, "clustering.json"), "w"))


def gen_evaluate():
    """evaulate() (motif_both_points.cpp:100-165) on a small ground-truth / hits pair."""
    gt = ["m0 p1 3.5", "m0 p2 25.5", "m0 p7 39.9", "m1 p0 0", "m1 p3 30", "m2 p2 24.5", "m2 p9 38"]
    hits = ["m1 p3 30", "m0 p1 3.5", "m2 p9 38", "m0 p7 39.9", "m3 p3 12"]
    d = tempfile.mkdtemp()
    g, h = os.path.join(d, "gt.txt"), os.path.join(d, "hits.txt")
    open(g, "w").write("\n".join(gt) + "\n")
    open(h, "w").write("\n".join(hits) + "\n")
    val = O.ref_evaluate(g, h, 40.0)
    for f in (g, h, h + ".accuracy.txt"):
        if os.path.exists(f):
            os.unlink(f)
    json.dump({"source": "motif_both_points.cpp:100-165 via oracle/_ref ref_evaluate",
               "ground_truth": gt, "hits": hits, "R": 40.0, "weighted_recall": val},
              open(os.path.join(OUT, "evaluate.json"), "w"))


def gen_klsh():
    """the reference's own KLSH object (pcluster lsh.cpp, compiled into
    oracle/_ref/libref_klsh.so) over PreClustering's features (pcluster.cpp:23-33) of random
    proteins.  The planes are the default-seeded engine's (lsh.hpp:49): data, not a choice."""
    w, b, t, refhash = O.ref_klsh(512, 16, 0.2)
    g = rng(4242)
    lens = g.integers(3, 700, size=400)
    lens[:6] = [3, 4, 5, 64, 65, 699]
    seqs = ["".join(LETTERS[c] for c in g.integers(0, 20, size=int(n))) for n in lens]
    classes = np.concatenate([O.klsh_classes(s_) for s_ in seqs])
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    codes = np.array([refhash(O.klsh_features(O.klsh_classes(s_))) for s_ in seqs], dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "klsh.npz"), w=w, b=b, t=t, classes=classes, seq_start=starts,
                        codes=codes, first_sequences=np.array(seqs[:6]))


def points_text(names, pts):
    """A points file as the reference's programs write them (operator<< at default precision)."""
    return "".join("%s\n%s\n" % (nm, " ".join("%g" % x for x in row)) for nm, row in zip(names, pts))


def gen_tools():
    """Row a11 as a program and SURVEY 8(f) row 4: motif_both_points_noLSH.cpp Search(), evaluate2's
    main(), centerDistanceSmapling's cluster2datapoint() and its main() -- all the REAL compiled
    reference (oracle/_ref/libref_nolsh.so, libref_evaluate2.so, libref_centers.so)."""
    import hashlib
    sha = lambda t: hashlib.sha256(t.encode()).hexdigest()
    d = tempfile.mkdtemp()
    out = {"source": "motif_both_points_noLSH.cpp:36-56, evaluate2.cpp:62-95, "
                     "centerDistanceSmapling.cpp:41-190,436-477 via oracle/_ref"}
    # -- exhaustive search: DB k-mers, centres = mutated DB k-mers + small real-valued offsets
    g = rng(901)
    k = 25
    codes = np.repeat(g.integers(0, 20, size=(40, k), dtype=np.uint8), 10, axis=0)   # 40 families
    for row in codes:
        for _ in range(int(g.integers(0, 4))):
            row[g.integers(0, k)] = g.integers(0, 20)
    g.shuffle(codes)
    db = O.ref_kmer_to_coordinates(seqs_of(codes), k)
    qc = codes[g.integers(0, 400, size=12)].copy()
    for row in qc:
        for _ in range(int(g.integers(0, 5))):
            row[g.integers(0, k)] = g.integers(0, 20)
    centers = O.ref_kmer_to_coordinates(seqs_of(qc), k) + g.normal(0, 0.05, size=(12, 8 * k))
    hits_path = os.path.join(d, "bf.txt")
    O.ref_nolsh_search(db, centers, 55.0, hits_path)
    hits = open(hits_path).read()
    rest = open(hits_path + "notlessthan.txt").read()
    assert hits.count("\n") > 12
    out["nolsh"] = {"file": "tools_nolsh.npz", "k": k, "R": 55.0, "hits": hits,
                    "notlessthan_sha256": sha(rest), "notlessthan_lines": rest.count("\n"),
                    "notlessthan_head": rest.split("\n")[:5]}
    np.savez_compressed(os.path.join(OUT, "tools_nolsh.npz"), codes=codes, centers=centers)
    # -- evaluate2 as it runs: <file>sort.txt; and its weight()
    O.ref_evaluate2_sort(hits_path)
    ws = [0.0, 1.0, 24.5, 49.38, 49.380001, 60.0, 98.76, 98.77, 200.0]
    out["evaluate2"] = {"hits": hits, "sorted": open(hits_path + "sort.txt").read(),
                        "weight_in": ws, "weight_out": [O.ref_evaluate2_weight(x) for x in ws]}
    # -- cluster2datapoint: centroids of 25-mer families as a points file
    fams = []
    for m in (50, 64, 131):
        seed_row = g.integers(0, 20, size=k)
        rows = np.repeat(seed_row[None, :], m, axis=0)
        mut = g.random(size=rows.shape) < 0.2
        rows[mut] = g.integers(0, 20, size=int(mut.sum()))
        fams.append(rows.astype(np.uint8))
    names = ["#PF%05d family_%d" % (100 + i, i) for i in range(len(fams))]
    O.ref_cluster2datapoint(k, names, [seqs_of(f) for f in fams], os.path.join(d, "c2d_"))
    out["cluster2datapoint"] = {"k": k, "names": names, "families": [seqs_of(f) for f in fams],
                                "points_file": open(os.path.join(d, "c2d_hclust.format.txt")).read()}
    # -- the program: families file (one family below MIN_SIZE_CLUSTER, blank lines) + a points
    #    file of 100000 3-mers (the reference reads that many whatever the file holds)
    k3, n3 = 3, 100000
    codes3 = g.integers(0, 20, size=(n3, k3), dtype=np.uint8)
    pts_text = points_text(["p%d" % i for i in range(n3)], O.ref_kmer_to_coordinates(seqs_of(codes3), k3))
    open(os.path.join(d, "db.points"), "w").write(pts_text)
    fam3 = [g.integers(0, 20, size=(m, k3), dtype=np.uint8) for m in (50, 49, 77, 120)]
    ftxt = "\n"
    for i, f in enumerate(fam3):
        ftxt += "#family %d\n" % i + "".join(s_ + "\n" for s_ in seqs_of(f)) + ("\n" if i == 1 else "")
    open(os.path.join(d, "fams.txt"), "w").write(ftxt)
    os.mkdir(os.path.join(d, "pro2centerdis"))
    assert O.ref_centers_main(d, os.path.join(d, "fams.txt"), os.path.join(d, "db.points"), k3, "g_") == 0
    inner = open(os.path.join(d, "pro2centerdis", "g_innercenter_protein_centers_0.txt")).read()
    rand = open(os.path.join(d, "pro2centerdis", "g_ramdom_protein_centers_0.txt")).read()
    out["center_sampling"] = {"file": "tools_centers.npz", "k": k3, "families_file": ftxt,
                              "points_file_sha256": sha(pts_text), "innercenter": inner,
                              "random_sha256": sha(rand), "random_lines": rand.count("\n"),
                              "random_head": rand.split("\n")[:8], "random_tail": rand.split("\n")[-9:-1]}
    np.savez_compressed(os.path.join(OUT, "tools_centers.npz"), codes=codes3)
    # -- protein2datapoints: its own main(), rand() seeded (the one seam of ref_p2d_harness.cpp)
    lens = [int(x) for x in g.integers(15, 400, size=9)]
    prots = ["".join(LETTERS[c] for c in g.integers(0, 20, size=n)) for n in lens]
    prots[4] = prots[1]                      # a repeated protein: windows already seen are skipped
    prots[6] = prots[6][:15]                 # exactly one window
    fasta = "\n" + "".join(">sp|P%05d|N%d_X text %d\n%s\n" % (i, i, i, p) for i, p in enumerate(prots))
    open(os.path.join(d, "p.fa"), "w").write(fasta)
    cases = []
    for seed, n_out in ((2026, 9), (5, 4)):
        assert O.ref_protein2datapoints(os.path.join(d, "p.fa"), 15, n_out, os.path.join(d, "p.points"), seed) == 0
        text = open(os.path.join(d, "p.points")).read()
        cases.append({"seed": seed, "num_out": n_out, "sha256": sha(text), "names": text.split("\n")[0::2][:-1],
                      "first_point": text.split("\n")[1]})
    out["protein2datapoints"] = {"k": 15, "fasta": fasta, "cases": cases,
                                 "note": "window strides are 30 + rand() % 20 of glibc's rand() after srand(seed)"}
    json.dump(out, open(os.path.join(OUT, "tools.json"), "w"))


def main():
    if not O.have_ref():
        O.build()
    if not O.have_ref():
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    gen_constants()
    gen_hash()
    gen_search()
    gen_pairwise()
    gen_clustering()
    gen_evaluate()
    gen_klsh()
    gen_tools()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote %d files, %.1f KiB" % (len(os.listdir(OUT)), total / 1024))


if __name__ == "__main__":
    main()
