"""C++ host layer (hsearch_amd/host): command-line contract of the reference's motif_both_points
(motif_both_points.cpp:302-335, 387-393) and the plane generator.  CPU part needs no GPU; the GPU
part runs the whole FASTA-less pipeline points-file in -> hits-file out."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# HS_HOST_BIN_DIR: another build of the programs (the sanitizer build, tests/test_sanitizers_cpu.py)
BIN_DIR = os.environ.get("HS_HOST_BIN_DIR") or os.path.join(ROOT, "hsearch_amd", "bin")
BIN = os.path.join(BIN_DIR, "hs_motif_both_points")


def _bin():
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", os.path.join(ROOT, "hsearch_amd", "host")], check=True,
                       stdout=subprocess.DEVNULL)
    return BIN


def _write_points(path, pts, fmt="%.17g"):
    with open(path, "w") as f:
        for i, row in enumerate(pts):
            f.write("p%d\n" % i)
            f.write(" ".join(fmt % v for v in row) + "\n")


def test_help_and_missing_option_exit_zero():
    r = subprocess.run([_bin()], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage" in r.stderr
    r = subprocess.run([_bin(), "-d", "x", "-c", "y"], capture_output=True, text=True)
    assert r.returncode == 0 and "missing required option" in r.stderr   # :332-335
    r = subprocess.run([_bin(), "-help"], capture_output=True, text=True)
    assert r.returncode == 0


def test_planes_match_reference_constructor(tmp_path, golden_dir):
    """--seed s must reproduce what L reference LSH objects seeded s, s+1, ... draw (lsh.hpp:10-31);
    the golden planes were dumped from the compiled reference."""
    import torch
    g = json.load(open(os.path.join(golden_dir, "search.json")))
    for case in g["cases"][:3]:
        z = np.load(os.path.join(golden_dir, case["file"]))
        d = 8 * case["k"]
        db, cen, out, planes = [str(tmp_path / n) for n in ("db", "cen", "out", "planes")]
        _write_points(db, np.zeros((2, d)))
        _write_points(cen, np.zeros((1, d)))
        r = subprocess.run([_bin(), "-d", db, "-c", cen, "-l", str(case["k"]), "-K", str(case["K"]),
                            "-L", str(case["L"]), "-W", repr(case["W"]), "-T", repr(case["R"]),
                            "-o", out, "--seed", str(case["plane_seed"]), "--planes-out", planes],
                           capture_output=True, text=True)
        if not torch.cuda.is_available():
            assert r.returncode == 1 and "gfx950" in r.stderr      # fails loudly, no CPU fallback
        raw = np.fromfile(planes, dtype=np.float64)
        na = case["L"] * case["K"] * d
        assert np.array_equal(raw[:na].reshape(z["a"].shape), z["a"])
        assert np.array_equal(raw[na:].reshape(z["b"].shape), z["b"])


@pytest.mark.gpu
def test_cli_hits_file_matches_reference_golden(tmp_path, golden_dir, oracle):
    g = json.load(open(os.path.join(golden_dir, "search.json")))
    for case in g["cases"]:
        z = np.load(os.path.join(golden_dir, case["file"]))
        db, cen, out = [str(tmp_path / n) for n in ("db", "cen", "out")]
        _write_points(db, oracle.embed_codes(z["codes"]))
        _write_points(cen, z["centers"])
        r = subprocess.run([_bin(), "-d", db, "-c", cen, "-l", str(case["k"]), "-K", str(case["K"]),
                            "-L", str(case["L"]), "-W", repr(case["W"]), "-T", repr(case["R"]),
                            "-o", out, "--seed", str(case["plane_seed"])],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = [ln.split() for ln in open(out)]
        assert [ln[0] for ln in lines] == ["p%d" % q for q in z["hit_q"]]
        assert [ln[1] for ln in lines] == ["p%d" % i for i in z["hit_id"]]
        assert [ln[2] for ln in lines] == case["hit_dist_text"]


@pytest.mark.gpu
def test_cli_lossy_points_file_and_evaluate(tmp_path, oracle):
    """DB written like protein2datapoints does (ostream default precision, 6 significant digits,
    protein2datapoints.cpp:23-29): the host derives codes + the rounded table from the file, so both
    sides hash the same doubles.  Checked against the oracle run on the parsed doubles."""
    from hsearch_amd import synth
    k, K, L, W, R, n, nq, seed = 25, 4, 4, 100.0, 40.0, 4000, 200, 77
    codes = synth.make_db(n, k, seed=5)
    centers, _ = synth.make_queries(codes, nq, seed=6, jitter=0.3)
    db, cen, out, planes, gt = [str(tmp_path / x) for x in ("db", "cen", "out", "planes", "gt")]
    _write_points(db, synth.embed(codes), fmt="%g")
    _write_points(cen, centers, fmt="%g")
    r = subprocess.run([_bin(), "-d", db, "-c", cen, "-l", str(k), "-K", str(K), "-L", str(L),
                        "-W", repr(W), "-T", repr(R), "-o", out, "--seed", str(seed),
                        "--planes-out", planes], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(planes, dtype=np.float64)
    a = raw[:L * K * 8 * k].reshape(L, K, 8 * k)
    b = raw[L * K * 8 * k:].reshape(L, K)
    parse = lambda p: np.array([[float(v) for v in ln.split()] for ln in open(p).read().split("\n")[1::2] if ln])
    dbp, cp = parse(db), parse(cen)
    assert np.abs(dbp - synth.embed(codes)).max() > 0     # really lossy
    want = oracle.search(a, b, W, R, dbp, cp)
    want_path = str(tmp_path / "want")
    oracle.write_hits(want_path, want["q"], want["id"], want["dist"])
    got = [ln.split() for ln in open(out)]
    exp = [ln.split() for ln in open(want_path)]
    assert len(got) == len(exp) > 0
    assert [(g_[0], g_[1], g_[2]) for g_ in got] == [("p" + e[0], "p" + e[1], e[2]) for e in exp]
    # -g: evaluation against a sorted brute-force file (the reference's evaluate2 + evaulate flow)
    bf = oracle.bruteforce(dbp, cp, R)
    rows = sorted(("p%d" % q, "p%d" % i, d_) for q, i, d_ in zip(bf["q"], bf["id"], bf["dist"]))
    with open(gt, "w") as f:
        for q, i, d_ in rows:
            f.write("%s %s %.6g\n" % (q, i, d_))
    r = subprocess.run([_bin(), "-d", db, "-c", cen, "-l", str(k), "-K", str(K), "-L", str(L),
                        "-W", repr(W), "-T", repr(R), "-o", out, "--seed", str(seed), "-g", gt],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    acc = float([ln for ln in r.stdout.splitlines() if ln.startswith("ACCURACY:")][0].split()[1])
    assert acc == pytest.approx(oracle.evaluate(gt, out, R), abs=1e-6)
    assert 0.5 < acc <= 1.0


@pytest.mark.gpu
def test_hclust2_cli_matches_reference_golden(tmp_path, golden_dir):
    """hs_hclust2 -k kmers.fa -l k -K -L -W -T -o out --seed s  ==  the compiled reference's hclust2
    Clustering() with its LSH objects seeded s, s+1, ... (byte-identical clusters file)."""
    binary = os.path.join(BIN_DIR, "hs_hclust2")
    if not os.path.exists(binary):
        subprocess.run(["make", "-C", os.path.join(ROOT, "hsearch_amd", "host")], check=True,
                       stdout=subprocess.DEVNULL)
    letters = "ARNDCQEGHILKMFPSTWYV"
    g = json.load(open(os.path.join(golden_dir, "clustering.json")))
    for case in g["cases"]:
        z = np.load(os.path.join(golden_dir, case["file"]))
        fa, out = str(tmp_path / "kmers.fa"), str(tmp_path / "clusters.txt")
        with open(fa, "w") as f:
            for i, row in enumerate(z["codes"]):
                f.write(">%d\n%s\n" % (i, "".join(letters[c] for c in row)))
        r = subprocess.run([binary, "-k", fa, "-l", str(case["k"]), "-K", str(case["K"]), "-L",
                            str(case["L"]), "-W", repr(case["W"]), "-T", repr(case["R"]), "-o", out,
                            "--seed", str(case["plane_seed"])], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out).read() == case["clusters_file"]
    r = subprocess.run([binary, "-k", "x"], capture_output=True, text=True)
    assert r.returncode == 0 and "missing required option" in r.stderr    # hclust2.cpp:223-226


@pytest.mark.gpu
def test_cli_fasta_database_matches_points_database(tmp_path, oracle):
    """-d <proteins.fa> (k-mers enumerated on the device, kmer_search.cpp:64-83
    order) must give the hits that -d <points of the same k-mers> gives, named like
    protein2datapoints names its samples (:66); a letter outside the alphabet breaks windows;
    --ref-compat-eq-swap embeds E as Gln and Q as Glu like the reference's ProteinDB."""
    import hsearch_amd
    k, K, L, W, R, seed = 25, 4, 4, 100.0, 40.0, 31
    rng = np.random.default_rng(5)
    letters = "ARNDCQEGHILKMFPSTWYV"                 # row order of the coordinate table
    seqs = ["".join(letters[i] for i in rng.integers(0, 20, size=n)) for n in (120, 24, 25, 300, 80)]
    seqs[3] = seqs[3][:100] + "X" + seqs[3][101:]    # an unknown letter in the middle
    seqs[0] = "EQ" + seqs[0][2:]
    fa = str(tmp_path / "db.fa")
    with open(fa, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">prot%d some description\n%s\n" % (i, s))
    for swap in (0, 1):
        names, rows = [], []
        for i, s in enumerate(seqs):
            for j in range(len(s) - k + 1):
                w = s[j:j + k]
                if "X" in w:
                    continue
                emb = w.translate(str.maketrans("EQ", "QE")) if swap else w
                shown = emb                                  # the reference prints the stored letters
                names.append("prot%d#%d$%d@%s*%d" % (i, i, j, shown, len(names)))
                rows.append([letters.index(c) for c in emb])
        codes = np.array(rows, dtype=np.uint8)
        pts = oracle.embed_codes(codes)
        centers = pts[rng.choice(len(pts), 40, replace=False)] + rng.normal(0, 0.3, size=(40, 8 * k))
        cen, out_fa, out_pt, dbp = [str(tmp_path / n) for n in ("cen", "out_fa", "out_pt", "db.points")]
        _write_points(cen, centers)
        with open(dbp, "w") as f:
            for nm, row in zip(names, pts):
                f.write(nm + "\n" + " ".join("%.17g" % v for v in row) + "\n")
        common = ["-c", cen, "-l", str(k), "-K", str(K), "-L", str(L), "-W", repr(W), "-T", repr(R),
                  "--seed", str(seed)]
        r1 = subprocess.run([_bin(), "-d", fa, "-o", out_fa, "--ref-compat-eq-swap", str(swap)] + common,
                            capture_output=True, text=True)
        assert r1.returncode == 0, r1.stderr
        assert "number of kmers %d" % len(names) in r1.stdout
        r2 = subprocess.run([_bin(), "-d", dbp, "-o", out_pt] + common, capture_output=True, text=True)
        assert r2.returncode == 0, r2.stderr
        got, want = open(out_fa).read(), open(out_pt).read()
        assert len(want.splitlines()) >= 40
        assert got == want


@pytest.mark.gpu
def test_cli_gpus_planes_file_and_kmer_centres(tmp_path, oracle):
    """SURVEY 8(b) CLI additions: --gpus 1 (the sharded path: host thread per GPU, hits all-gathered
    over RCCL -- here one rank) writes byte for byte what the plain path writes, for a points and
    for a FASTA database; --planes <file> reads what --planes-out wrote; -c <k-mers.fa> embeds the
    centres from the table (KmerToCoordinates, hclust2.cpp:49-62) = a points file of the same."""
    k, K, L, W, R, seed = 25, 6, 5, 140.0, 40.0, 77
    rng = np.random.default_rng(11)
    letters = "ARNDCQEGHILKMFPSTWYV"
    codes = rng.integers(0, 20, size=(3000, k), dtype=np.uint8)
    pts = oracle.embed_codes(codes)
    qcodes = codes[rng.choice(len(codes), 120, replace=False)].copy()
    for row in qcodes:
        for _ in range(int(rng.integers(0, 4))):
            row[rng.integers(0, k)] = rng.integers(0, 20)
    db, cen, cfa, planes = [str(tmp_path / n) for n in ("db.points", "cen.points", "cen.fa", "planes.bin")]
    _write_points(db, pts)
    with open(cen, "w") as f, open(cfa, "w") as g:
        for i, row in enumerate(qcodes):
            f.write("c%d\n" % i + " ".join("%.17g" % v for v in oracle.embed_codes(row[None])[0]) + "\n")
            g.write(">c%d\n%s\n" % (i, "".join(letters[c] for c in row)))
    common = ["-l", str(k), "-K", str(K), "-L", str(L), "-W", repr(W), "-T", repr(R)]

    def run(out, *extra):
        r = subprocess.run([_bin(), "-o", str(tmp_path / out)] + common + list(extra), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return open(tmp_path / out).read(), r.stdout
    plain, _ = run("plain", "-d", db, "-c", cen, "--seed", str(seed), "--planes-out", planes)
    assert len(plain.splitlines()) >= 80
    # the oracle agrees with the plain path (planes read back from the dump)
    raw = np.fromfile(planes, dtype=np.float64)
    a = raw[:L * K * 8 * k].reshape(L, K, 8 * k)
    b = raw[L * K * 8 * k:].reshape(L, K)
    want = oracle.search(a, b, W, R, pts, oracle.embed_codes(qcodes))
    assert [ln.split()[:2] for ln in plain.splitlines()] == [["c%d" % q, "p%d" % i] for q, i in zip(want["q"], want["id"])]
    sharded, so = run("gpus1", "-d", db, "-c", cen, "--seed", str(seed), "--gpus", "1")
    assert "gpus = 1" in so and sharded == plain
    from_file, so = run("planes", "-d", db, "-c", cen, "--planes", planes)
    assert "planes = " in so and from_file == plain
    kmer_centres, _ = run("cfa", "-d", db, "-c", cfa, "--planes", planes, "--gpus", "1")
    assert kmer_centres == plain
    # a wrong-sized planes file, a centre of another length, and more GPUs than the box has
    r = subprocess.run([_bin(), "-o", str(tmp_path / "x"), "-d", db, "-c", cen, "--planes", planes, "-l", str(k),
                        "-K", str(K + 1), "-L", str(L), "-W", repr(W), "-T", repr(R)], capture_output=True, text=True)
    assert r.returncode == 1 and "does not hold" in r.stderr
    cfa_ok = str(tmp_path / "cen_ok.fa")
    with open(cfa_ok, "w") as g:
        g.write(open(cfa).read())
    with open(cfa, "a") as g:
        g.write(">short\nARND\n")
    r = subprocess.run([_bin(), "-o", str(tmp_path / "x"), "-d", db, "-c", cfa, "--planes", planes] + common,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "does not have" in r.stderr
    import torch
    r = subprocess.run([_bin(), "-o", str(tmp_path / "x"), "-d", db, "-c", cen, "--planes", planes, "--gpus",
                        str(torch.cuda.device_count() + 1)] + common, capture_output=True, text=True)
    assert r.returncode == 1 and "GPU" in r.stderr
    # FASTA database through the sharded path
    fa = str(tmp_path / "db.fa")
    with open(fa, "w") as f:   # proteins made of the centres' own k-mers (+ other DB k-mers): every centre has hits
        for i in range(6):
            rows = np.concatenate([qcodes[20 * i:20 * i + 20], codes[40 * i:40 * i + 8]])
            f.write(">prot%d x\n%s\n" % (i, "".join(letters[c] for c in rows.ravel())))
    p1, _ = run("fa_plain", "-d", fa, "-c", cen, "--planes", planes)
    p2, _ = run("fa_gpus1", "-d", fa, "-c", cen, "--planes", planes, "--gpus", "1")
    assert p1 == p2
    # k-mer centres over a FASTA database travel as residue codes (hs_query_codes, 25 bytes each): the
    # same file as the embedded centres give (points file; --centers-as-points), plain and sharded
    p3, _ = run("fa_codes", "-d", fa, "-c", cfa_ok, "--planes", planes)
    p4, _ = run("fa_codes_emb", "-d", fa, "-c", cfa_ok, "--planes", planes, "--centers-as-points", "1")
    p5, _ = run("fa_codes_g1", "-d", fa, "-c", cfa_ok, "--planes", planes, "--gpus", "1")
    assert p3 == p1 and p4 == p1 and p5 == p1 and len(p1.splitlines()) > 0


@pytest.mark.gpu
def test_cli_two_rank_threads_with_live_handles_over_loopback(tmp_path, oracle):
    """hsearch::SearchSharded's rank threads with TWO live handles.  RCCL refuses
    two ranks on one device, so --transport loopback puts both ranks on --device and exchanges the hits
    through host memory: the barrier / capacity / failed-rank protocol with real hs_query_dev calls.
    --gpus 2 and 3 write byte for byte what --gpus 1 and the plain path write (points and FASTA
    database, centres as points and as codes); a rank made to fail (test build of the program:
    HS_TEST_FAIL_RANK) ends the run with an error on every rank instead of hanging rank 0."""
    k, K, L, W, R, seed = 25, 6, 5, 140.0, 40.0, 78
    rng = np.random.default_rng(12)
    letters = "ARNDCQEGHILKMFPSTWYV"
    codes = rng.integers(0, 20, size=(3000, k), dtype=np.uint8)
    pts = oracle.embed_codes(codes)
    qcodes = codes[rng.choice(len(codes), 121, replace=False)].copy()   # 121: uneven blocks
    for row in qcodes:
        for _ in range(int(rng.integers(0, 4))):
            row[rng.integers(0, k)] = rng.integers(0, 20)
    db, cen, cfa, fa = [str(tmp_path / n) for n in ("db.points", "cen.points", "cen.fa", "db.fa")]
    _write_points(db, pts)
    with open(cen, "w") as f, open(cfa, "w") as g:
        for i, row in enumerate(qcodes):
            f.write("c%d\n" % i + " ".join("%.17g" % v for v in oracle.embed_codes(row[None])[0]) + "\n")
            g.write(">c%d\n%s\n" % (i, "".join(letters[c] for c in row)))
    with open(fa, "w") as f:   # proteins made of the centres' own k-mers (+ other DB k-mers): every centre has hits
        for i in range(11):
            rows = np.concatenate([qcodes[11 * i:11 * i + 11], codes[50 * i:50 * i + 8]])
            f.write(">prot%d x\n%s\n" % (i, "".join(letters[c] for c in rows.ravel())))
    common = ["-l", str(k), "-K", str(K), "-L", str(L), "-W", repr(W), "-T", repr(R), "--seed", str(seed)]

    def run(out, *extra, binary=None, env=None, ok=True):
        r = subprocess.run([binary or _bin(), "-o", str(tmp_path / out)] + common + list(extra),
                           capture_output=True, text=True, timeout=300, env=env)
        if ok:
            assert r.returncode == 0, r.stderr
            return open(tmp_path / out).read(), r.stdout
        return r
    plain, _ = run("plain", "-d", db, "-c", cen)
    assert len(plain.splitlines()) >= 80
    for n in (1, 2, 3):
        got, so = run("lb%d" % n, "-d", db, "-c", cen, "--gpus", str(n), "--transport", "loopback")
        assert "%d ranks on device 0" % n in so and got == plain
    # the table-partitioned layout: every rank a subset of the L = 5 tables (dealt by estimated join work) over
    # all k-mers and ALL centres, merged behind the exchange by the first-seen rule: the same file, and the
    # same table sizes in the reference's "table size" lines
    _, so_plain = run("plain2", "-d", db, "-c", cen)
    for n in (1, 2, 3, 5):
        got, so = run("tp%d" % n, "-d", db, "-c", cen, "--gpus", str(n), "--transport", "loopback", "--partition", "tables")
        assert got == plain, n
        assert [l for l in so.splitlines() if l.startswith("table size")] == \
            [l for l in so_plain.splitlines() if l.startswith("table size")]
    r = run("tp6", "-d", db, "-c", cen, "--gpus", "6", "--transport", "loopback", "--partition", "tables", ok=False)
    assert r.returncode == 1 and "more GPUs than tables" in r.stderr
    # the bucket-partitioned layout: every rank the whole index, ALL centres and its share of the buckets
    # (hs_comm_query_buckets), merged by the same rule: the same file, with more ranks than tables too
    for n in (1, 2, 3, 6):
        got, so = run("bp%d" % n, "-d", db, "-c", cen, "--gpus", str(n), "--transport", "loopback", "--partition", "buckets")
        assert got == plain, n
        assert [l for l in so.splitlines() if l.startswith("table size")] == \
            [l for l in so_plain.splitlines() if l.startswith("table size")]
    r = run("pbad", "-d", db, "-c", cen, "--gpus", "2", "--transport", "loopback", "--partition", "members", ok=False)
    assert r.returncode == 1 and "--partition" in r.stderr
    fa_plain, _ = run("fa_plain", "-d", fa, "-c", cfa, "--centers-as-points", "1")
    assert len(fa_plain.splitlines()) > 0
    for extra in ((), ("--centers-as-points", "1"), ("--partition", "tables"), ("--partition", "buckets")):
        got, _ = run("fa_lb2", "-d", fa, "-c", cfa, "--gpus", "2", "--transport", "loopback", *extra)
        assert got == fa_plain
    # a failed rank: everybody stops, nobody hangs (timeout above), the message names the rank
    hooks = _bin() + "_hooks"
    env = dict(os.environ, HS_TEST_FAIL_RANK="1")
    r = run("fail", "-d", db, "-c", cen, "--gpus", "2", "--transport", "loopback", binary=hooks, env=env, ok=False)
    assert r.returncode == 1 and "hs_comm_query" in r.stderr
    got, _ = run("nofail", "-d", db, "-c", cen, "--gpus", "2", "--transport", "loopback", binary=hooks)
    assert got == plain
    r = run("bad", "-d", db, "-c", cen, "--gpus", "2", "--transport", "smoke", ok=False)
    assert r.returncode == 1 and "--transport" in r.stderr


@pytest.mark.gpu
def test_cli_best_centre_per_position(tmp_path, oracle):
    """--best-per-position: kmer_search.cpp's `matches` (:90,113-121) -- per database window the
    nearest centre, visited tables-outer / centres-inner, replaced only when strictly nearer --
    replayed here over the oracle's hit list of the same windows."""
    from hsearch_amd import synth
    k, K, L, W, R, seed = 25, 2, 4, 150.0, 45.0, 33
    rng = np.random.default_rng(15)
    letters = "ARNDCQEGHILKMFPSTWYV"
    base = "".join(letters[i] for i in rng.integers(0, 20, size=60))
    seqs = []
    for _ in range(12):                                    # near-copies: windows with many suitors
        s_ = list(base)
        for _ in range(3):
            s_[rng.integers(0, len(s_))] = letters[rng.integers(0, 20)]
        seqs.append("".join(s_))
    fa = str(tmp_path / "db.fa")
    with open(fa, "w") as f:
        for i, s_ in enumerate(seqs):
            f.write(">p%d\n%s\n" % (i, s_))
    names, rows = [], []
    for i, s_ in enumerate(seqs):
        for j in range(len(s_) - k + 1):
            names.append("p%d#%d$%d@%s*%d" % (i, i, j, s_[j:j + k], len(names)))
            rows.append([letters.index(c) for c in s_[j:j + k]])
    codes = np.array(rows, dtype=np.uint8)
    pts = oracle.embed_codes(codes)
    pick = rng.choice(len(pts), 30, replace=False)
    centers = np.concatenate([pts[pick[:15]], pts[pick[:15]],                 # exact ties between centres
                              pts[pick[15:]] + rng.normal(0, 0.3, size=(15, 8 * k))])
    cen, out = str(tmp_path / "cen"), str(tmp_path / "out")
    _write_points(cen, centers)
    r = subprocess.run([_bin(), "-d", fa, "-c", cen, "-o", out, "-l", str(k), "-K", str(K), "-L", str(L),
                        "-W", repr(W), "-T", repr(R), "--seed", str(seed), "--planes-out", out + ".planes",
                        "--best-per-position", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out + ".planes", dtype=np.float64)
    a = raw[:L * K * 8 * k].reshape(L, K, 8 * k)
    b = raw[L * K * 8 * k:].reshape(L, K)
    res = oracle.search(a, b, W, R, pts, centers)
    best = {}
    for t in np.lexsort((res["q"], res["table"])):        # tables outer, centres inner
        i, q, dist = int(res["id"][t]), int(res["q"][t]), float(res["dist"][t])
        if i not in best or best[i][1] > dist:
            best[i] = (q, dist)
    want = "".join("%s p%d %g\n" % (names[i], best[i][0], best[i][1]) for i in sorted(best))
    assert open(out).read() == want
    assert len(best) > 100 and len(set(q for q, _ in best.values())) > 10


@pytest.mark.gpu
def test_pcluster_pregroup_cli_matches_oracle(tmp_path, oracle):
    """hs_pcluster_pregroup (pcluster.cpp:11-81 on the GPU) groups proteins
    exactly as the oracle's KLSH restatement (itself pinned to the reference's KLSH object)."""
    exe = os.path.join(BIN_DIR, "hs_pcluster_pregroup")
    _bin()
    rng = np.random.default_rng(12)
    letters = "ARNDCQEGHILKMFPSTWYV"
    seqs = ["".join(letters[i] for i in rng.integers(0, 20, size=int(n)))
            for n in rng.integers(1, 900, size=200)]
    seqs[5] = "AR"                                   # shorter than 3: in no group
    fa = str(tmp_path / "p.fa")
    with open(fa, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">prot%d description text\n" % i)
            for j in range(0, len(s), 60):           # multi-line FASTA
                f.write(s[j:j + 60] + "\n")
    out = str(tmp_path / "groups.txt")
    r = subprocess.run([exe, "-d", fa, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    w, b, t = oracle.klsh_draw_planes()
    want = {}
    for i, s in enumerate(seqs):
        if len(s) >= 3:
            want.setdefault(oracle.klsh_hash(w, b, t, oracle.klsh_features(oracle.klsh_classes(s))), []).append("prot%d" % i)
    assert "[NUMBER OF PRE-GROUPS %d]" % len(want) in r.stderr
    got = {}
    for ln in open(out):
        code, name = ln.rstrip("\n").split("\t")
        got.setdefault(int(code), []).append(name)
    assert got == want and list(got) == sorted(got)
    assert subprocess.run([exe], capture_output=True).returncode == 0     # help, exit 0


# ---- row a11 as a program, SURVEY 8(f) row 4: noLSH search, evaluate2, centroid builder ----------
def _tool(name):
    _bin()
    return os.path.join(BIN_DIR, name)


def _points_text(names, pts):
    return "".join("%s\n%s\n" % (nm, " ".join("%g" % x for x in row)) for nm, row in zip(names, pts))


def test_evaluate2_cli_matches_reference_golden(tmp_path, golden_dir):
    """`evaluate2 <hits>` as the reference runs (evaluate2.cpp:73-95): <hits>sort.txt; host only."""
    t = json.load(open(os.path.join(golden_dir, "tools.json")))["evaluate2"]
    hits = str(tmp_path / "bf.txt")
    open(hits, "w").write(t["hits"])
    r = subprocess.run([_tool("hs_evaluate2"), hits], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == hits + "\n"
    assert open(hits + "sort.txt").read() == t["sorted"]
    # the comparison behind the reference's early return: a hits file that is a subset
    lines = t["hits"].strip().split("\n")
    part = str(tmp_path / "part.txt")
    open(part, "w").write("\n".join(lines[::2]) + "\nzz_extra k0 1\n")
    r = subprocess.run([_tool("hs_evaluate2"), hits, part], capture_output=True, text=True)
    assert r.returncode == 0
    acc = [ln for ln in r.stdout.split("\n") if ln.startswith("ACCURACY: ")]
    assert len(acc) == 1 and acc[0].endswith("\t" + part)
    from oracle import pyoracle as O
    want, tp, fn = O.evaluate2(hits, part)
    got = [float(x) for x in acc[0].split("\t")[0].split()[1:]]
    assert got == [float("%g" % tp), float("%g" % fn), float("%g" % want)] and 0.3 < want < 0.7
    r = subprocess.run([_tool("hs_evaluate2")], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage" in r.stderr


def test_centroid_points_file_cli_matches_reference_golden(tmp_path, golden_dir):
    """cluster2datapoint() (centerDistanceSmapling.cpp:110-136): family centroids as the `-c` points
    file of motif_both_points; host only (no database, no GPU)."""
    t = json.load(open(os.path.join(golden_dir, "tools.json")))["cluster2datapoint"]
    fam = str(tmp_path / "fams.txt")
    with open(fam, "w") as f:
        for nm, seqs in zip(t["names"], t["families"]):
            f.write(nm + "\n" + "".join(s_ + "\n" for s_ in seqs))
        f.write("#too small\n" + t["families"][0][0] + "\n")
    out = str(tmp_path / "o_")
    r = subprocess.run([_tool("hs_center_distance_sampling"), "-k", fam, "-l", str(t["k"]), "-o", out,
                        "-format", "points"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Number of Clusters: 3" in r.stdout
    assert open(out + "hclust.format.txt").read() == t["points_file"]
    r = subprocess.run([_tool("hs_center_distance_sampling"), "-k", fam, "-l", "24", "-o", out,
                        "-format", "points"], capture_output=True, text=True)
    assert r.returncode == 1 and "24-mer" in r.stderr
    r = subprocess.run([_tool("hs_center_distance_sampling"), "-k", fam], capture_output=True, text=True)
    assert r.returncode == 0 and "missing required option" in r.stderr


@pytest.mark.gpu
def test_nolsh_cli_matches_reference_golden(tmp_path, golden_dir):
    """motif_both_points_noLSH: hits file and (with -notlessthan) the excluded-pairs file, byte for
    byte what the compiled reference wrote for the same points."""
    from hsearch_amd import synth
    t = json.load(open(os.path.join(golden_dir, "tools.json")))["nolsh"]
    z = np.load(os.path.join(golden_dir, t["file"]))
    db, cen, out = [str(tmp_path / n) for n in ("db", "cen", "out")]
    with open(db, "w") as f:
        for i, row in enumerate(synth.embed(z["codes"])):
            f.write("k%d\n%s\n" % (i, " ".join("%.17g" % v for v in row)))
    with open(cen, "w") as f:
        for i, row in enumerate(z["centers"]):
            f.write("c%d\n%s\n" % (i, " ".join("%.17g" % v for v in row)))
    import hashlib
    for extra in ([], ["-notlessthan"]):
        if os.path.exists(out + "notlessthan.txt"):
            os.unlink(out + "notlessthan.txt")
        r = subprocess.run([_tool("hs_motif_both_points_noLSH"), "-d", db, "-c", cen, "-l", str(t["k"]),
                            "-T", repr(t["R"]), "-o", out] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "number of kmers 400" in r.stdout and "Searching takes" in r.stdout
        assert open(out).read() == t["hits"]
        if extra:
            rest = open(out + "notlessthan.txt").read()
            assert rest.count("\n") == t["notlessthan_lines"]
            assert hashlib.sha256(rest.encode()).hexdigest() == t["notlessthan_sha256"]
        else:
            assert not os.path.exists(out + "notlessthan.txt")


@pytest.mark.gpu
def test_center_distance_sampling_cli_matches_reference_golden(tmp_path, golden_dir):
    """centerDistanceSmapling as the reference runs it (sequencedatabase2centers): both distance
    files identical to the compiled reference's (300000 centre-to-k-mer distances on the GPU)."""
    import hashlib
    from hsearch_amd import synth
    t = json.load(open(os.path.join(golden_dir, "tools.json")))["center_sampling"]
    codes = np.load(os.path.join(golden_dir, t["file"]))["codes"]
    text = _points_text(["p%d" % i for i in range(len(codes))], synth.embed(codes))
    assert hashlib.sha256(text.encode()).hexdigest() == t["points_file_sha256"]
    db, fam = str(tmp_path / "db.points"), str(tmp_path / "fams.txt")
    open(db, "w").write(text)
    open(fam, "w").write(t["families_file"])
    r = subprocess.run([_tool("hs_center_distance_sampling"), "-k", fam, "-d", db, "-l", str(t["k"]),
                        "-o", "g_"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    assert "Number of Clusters: 3" in r.stdout
    sub = tmp_path / "pro2centerdis"
    assert open(sub / "g_innercenter_protein_centers_0.txt").read() == t["innercenter"]
    rand = open(sub / "g_ramdom_protein_centers_0.txt").read()
    assert rand.split("\n")[:8] == t["random_head"] and rand.split("\n")[-9:-1] == t["random_tail"]
    assert rand.count("\n") == t["random_lines"]
    assert hashlib.sha256(rand.encode()).hexdigest() == t["random_sha256"]


def test_protein2datapoints_cli_matches_reference_golden(tmp_path, golden_dir):
    """protein2datapoints (FASTA -> sampled k-mers as a points file): names, E <-> Q exchange,
    skipping of windows already seen and the embedding text, identical to what the compiled
    reference's own main() wrote under the same rand() seed; host only."""
    import hashlib
    t = json.load(open(os.path.join(golden_dir, "tools.json")))["protein2datapoints"]
    fa, out = str(tmp_path / "p.fa"), str(tmp_path / "p.points")
    open(fa, "w").write(t["fasta"])
    for case in t["cases"]:
        r = subprocess.run([_tool("hs_protein2datapoints"), "-d", fa, "-l", str(t["k"]), "-n",
                            str(case["num_out"]), "-o", out, "-s", str(case["seed"])],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "number of proteins 9" in r.stdout
        text = open(out).read()
        assert text.split("\n")[0::2][:-1] == case["names"]
        assert text.split("\n")[1] == case["first_point"]
        assert hashlib.sha256(text.encode()).hexdigest() == case["sha256"]
    # the points file feeds the search programs: every record parses back to its k-mer
    lines = open(out).read().split("\n")
    assert all(len(lines[2 * i + 1].split()) == 8 * t["k"] for i in range(len(t["cases"][-1]["names"])))
    # -Q 0: the correct embedding; an input E keeps its letter
    r = subprocess.run([_tool("hs_protein2datapoints"), "-d", fa, "-l", str(t["k"]), "-n", "1", "-o", out,
                        "-s", "1", "-Q", "0"], capture_output=True, text=True)
    assert r.returncode == 0
    first = t["fasta"].strip().split("\n")[1][:t["k"]]
    assert open(out).read().split("\n")[0] == "sp|P00000|N0_X#0$0@%s*0" % first
    r = subprocess.run([_tool("hs_protein2datapoints"), "-d", fa], capture_output=True, text=True)
    assert r.returncode == 0 and "missing required option" in r.stderr


def test_protein2datapoints_cli_matches_live_reference(tmp_path):
    from oracle import pyoracle as O
    if not os.path.exists(os.path.join(os.path.dirname(O.__file__), "_ref", "libref_p2d.so")):
        pytest.skip("oracle/_ref/libref_p2d.so not built")
    rng = np.random.default_rng(12)
    letters = "ARNDCQEGHILKMFPSTWYV"
    seqs = ["".join(letters[c] for c in rng.integers(0, 20, size=int(n))) for n in rng.integers(25, 700, size=25)]
    seqs[7] = seqs[2]
    fa = str(tmp_path / "db.fa")
    with open(fa, "w") as f:
        for i, s_ in enumerate(seqs):
            f.write(">prot%d\n%s\n\n" % (i, s_))
    for seed, k, n in [(1, 25, 25), (2, 25, 3), (3, 9, 100)]:
        ref, mine = str(tmp_path / "ref.points"), str(tmp_path / "mine.points")
        assert O.ref_protein2datapoints(fa, k, n, ref, seed) == 0
        r = subprocess.run([_tool("hs_protein2datapoints"), "-d", fa, "-l", str(k), "-n", str(n), "-o", mine,
                            "-s", str(seed)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(ref).read() == open(mine).read() and os.path.getsize(ref) > 0
