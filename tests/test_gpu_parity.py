"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Integer outputs (bucket ints, hit q/id/table, candidate counts) bit-exact; distances: the product
evaluates d2 in the reference's own fp64 order, so they are compared exactly too (the north-star
tolerance of 1e-5 relative is the documented bar, asserted as well)."""
import numpy as np
import pytest

from hsearch_amd import Engine, synth

pytestmark = pytest.mark.gpu


def _assert_hits_equal(got, want):
    assert len(got["q"]) == len(want["q"])
    assert np.array_equal(got["q"], want["q"])
    assert np.array_equal(got["id"], want["id"])
    if "table" in want:
        assert np.array_equal(got["table"], want["table"])
    assert np.allclose(got["dist"], want["dist"], rtol=1e-5, atol=0)
    assert np.array_equal(got["dist"], want["dist"])  # same fp64 evaluation order => identical


@pytest.mark.parametrize("k,K,L,W", [(25, 4, 4, 100.0), (25, 16, 8, 200.0), (25, 20, 3, 37.5),
                                     (15, 6, 5, 7.0), (39, 5, 2, 0.9), (25, 7, 3, 50.0)])
def test_bucket_ints_bit_exact(oracle, k, K, L, W):
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(3000, k, seed=11)
    eng = Engine(k, K, L, W, a, b)
    pts = oracle.embed_codes(codes)
    want = oracle.hash_all(a, b, W, pts)
    got = eng.hash_codes(codes)
    assert got.dtype == np.int32 and got.shape == (3000, L, K)
    assert np.array_equal(got, want)
    # arbitrary (non-table) points through the points entry
    rng = np.random.default_rng(5)
    cpts = pts[:500] + rng.normal(0, 0.5, size=(500, 8 * k))
    assert np.array_equal(eng.hash_points(cpts), oracle.hash_all(a, b, W, cpts))
    assert np.array_equal(eng.embed_codes(codes[:100]), pts[:100])
    eng.close()


@pytest.mark.parametrize("k,K,L,W,R,n,nq", [(25, 4, 4, 100.0, 40.0, 10000, 1000),
                                            (25, 16, 8, 200.0, 40.0, 20000, 500),
                                            (25, 4, 4, 50.0, 40.0, 5000, 300),
                                            (15, 4, 6, 60.0, 30.0, 8000, 400),
                                            (39, 8, 4, 150.0, 50.0, 6000, 300),
                                            (25, 2, 3, 3.0, 40.0, 4000, 200),
                                            # configs[2]'s table shape (L = 32, K = 20) at oracle size
                                            (25, 20, 32, 200.0, 40.0, 20000, 400),
                                            (25, 20, 32, 320.0, 45.0, 12000, 300),
                                            (39, 20, 32, 250.0, 50.0, 6000, 200)])
def test_search_hits_match_oracle(oracle, k, K, L, W, R, n, nq):
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(n, k)
    centers, _ = synth.make_queries(codes, nq, jitter=0.25)
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    ix = oracle.Index(a, b, W, oracle.embed_codes(codes))
    assert info["n_buckets"] == ix.table_sizes()
    want = ix.query(centers, R)
    assert len(want["q"]) > 0
    # both filter kernels sit in front of the same exact decision: identical results
    for mode in ("stream", "join", "join16", "auto"):
        eng.set_verify_mode(mode)
        got = eng.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"]), mode
        _assert_hits_equal(got, want)
        prof = eng.profile()
        assert prof["candidates"] == int(want["cand"].sum())
        if mode == "stream" or k > 50 or (k > 25 and mode == "join16"):
            assert prof["join_batches"] == 0          # no fp16 form for two packed words
        elif k > 25:
            assert prof["join_i8_batches"] == prof["join_batches"]
        elif W >= 50.0:
            assert prof["join_batches"] > 0      # the MFMA bucket join really ran
            assert 0 < prof["join_pairs"] <= prof["candidates"]
            # "join"/"auto" = the int8 kernel, "join16" = the fp16 kernel
            assert (prof["join_i8_batches"] > 0) == (mode != "join16")
    eng.close()


@pytest.mark.parametrize("k", [26, 33, 39, 41, 42, 49, 50])
def test_long_kmers_through_the_int8_join(oracle, k):
    """k in 26..50 (two packed words; configs[4]'s k = 39): the int8 bucket join with 6 (k <= 41) or 8
    k-steps and 64-member work items, its thin-segment and refinement kernels.  Coarse keys so that
    buckets are big and shared by many queries; DIMENSION = 8k (motif_both_points.cpp:337-338)."""
    K, L, W, R, n, nq = 3, 3, 500.0, 52.0, 20011, 1203
    a, b = synth.make_planes(k, K, L, W, seed=15)
    codes = synth.make_db(n, k, seed=16)
    centers, _ = synth.make_queries(codes, nq, seed=17, jitter=0.2)
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    assert max(info["max_bucket"]) > 1000
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    assert len(want["q"]) > 200
    for mode in ("join", "auto", "stream"):
        eng.set_verify_mode(mode)
        got = eng.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
        prof = eng.profile()
        if mode == "stream":
            assert prof["join_batches"] == 0
        else:
            assert prof["join_i8_batches"] > 0 and prof["join_pairs"] > 0.5 * prof["candidates"]
    # a query int8 cannot carry: no fp16 form for long k-mers, the batch streams
    wide = centers[:64].copy()
    wide[:8, ::8] += 9.0
    eng.set_verify_mode("join")
    _assert_hits_equal(eng.query(wide, R), oracle.search(a, b, W, R, oracle.embed_codes(codes), wide))
    assert eng.profile()["join_batches"] == 0
    eng.close()


@pytest.mark.parametrize("k,R", [(5, 20.0), (8, 30.0), (12, 36.0), (15, 40.0), (20, 40.0)])
def test_short_kmers_through_the_wide_int8_join(oracle, k, R):
    """k <= 20 (configs[4]'s k = 15): R^2 is no longer far below the 4-column distance of bucket mates,
    so the int8 rows carry all 8 coordinate columns (6 k-steps, 64-member work items, no refinement
    pass).  Hits, order and distances equal the oracle's in every verify mode, with the thin-segment
    filter in play (options join_min_q / join_min_m) and with the 4-column rows forced (option wide_rows = 3).
    (How many
    fewer survivors the wide rows leave at the bench's sizes: tools/regime_sweep.py.)"""
    K, L, W, n, nq = 3, 3, 260.0, 20011, 1203
    a, b = synth.make_planes(k, K, L, W, seed=25)
    codes = synth.make_db(n, k, seed=26)
    centers, _ = synth.make_queries(codes, nq, seed=27, jitter=0.2)
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    assert len(want["q"]) > 200
    survivors = {}
    for rows in ("wide", "wide-thin", "narrow"):
        opts = {}
        if rows == "narrow":  # 4-column rows for every k, whatever the radius
            opts = dict(wide_rows=3)
        if rows == "wide-thin":
            opts = dict(join_min_q=3, join_min_m=16)
        eng = Engine(k, K, L, W, a, b, options=opts)
        info = eng.index_build(codes)
        assert max(info["max_bucket"]) > 1000
        for mode in ("join", "auto", "join16", "stream"):
            eng.set_verify_mode(mode)
            got = eng.query(centers, R)
            assert np.array_equal(got["cand"], want["cand"])
            _assert_hits_equal(got, want)
            prof = eng.profile()
            if mode == "join":
                assert prof["join_i8_batches"] > 0 and prof["join_pairs"] > 0.5 * prof["candidates"]
                survivors[rows] = prof["provisional"]
        # a query far outside the table's range: int8 cannot carry it, the fp16 form takes the batch
        far = centers[:64].copy()
        far[:8, ::8] += 9.0
        eng.set_verify_mode("join")
        _assert_hits_equal(eng.query(far, R), oracle.search(a, b, W, R, oracle.embed_codes(codes), far))
        eng.close()
    assert survivors["wide"] > 0 and survivors["narrow"] > 0


@pytest.mark.parametrize("k,R", [(21, 40.0), (23, 52.0), (25, 58.0)])
def test_wide_rows_by_radius(oracle, tmp_path, k, R):
    """k = 21..25: calls whose radius is large for the k-mer length (R^2 within 3 standard deviations of
    the mean 4-column distance of random k-mers; forced for the others) run on 8-k-step rows over all 8
    columns, whose member records are built on first use -- also on an index that came from a file.
    Same hits as the oracle, as the 4-column rows (option wide_rows = 2) and as the streaming filter;
    calls at a small radius on the same handle keep using the 4-column rows."""
    K, L, W, n, nq = 3, 3, 300.0, 20011, 1203
    a, b = synth.make_planes(k, K, L, W, seed=45)
    codes = synth.make_db(n, k, seed=46)
    centers, _ = synth.make_queries(codes, nq, seed=47, jitter=0.2)
    pts = oracle.embed_codes(codes)
    want = oracle.search(a, b, W, R, pts, centers)
    want_small = oracle.search(a, b, W, 30.0, pts, centers)
    assert len(want["q"]) > 200
    path = str(tmp_path / "idx.bin")
    survivors = {}
    for rows in ("by-radius", "forced", "forced-thin", "narrow", "loaded"):
        opts = {}
        if rows.startswith("forced") or rows == "loaded":
            opts["wide_rows"] = 1
        if rows == "narrow":
            opts["wide_rows"] = 2
        if rows == "forced-thin":
            opts.update(join_min_q=3, join_min_m=16)
        eng = Engine(k, K, L, W, a, b, options=opts)
        if rows == "loaded":
            eng.index_load(path)
        else:
            eng.index_build(codes)
        if rows == "by-radius":
            eng.index_save(path)
        for mode in ("join", "auto", "stream"):
            eng.set_verify_mode(mode)
            got = eng.query(centers, R)
            assert np.array_equal(got["cand"], want["cand"])
            _assert_hits_equal(got, want)
            if mode == "join":
                survivors[rows] = eng.profile()["join_items"]   # 64 members per work item on wide rows, 128 else
        if rows == "by-radius":  # a small radius between two large ones: 4-column rows, then wide again
            eng.set_verify_mode("join")
            _assert_hits_equal(eng.query(centers, 30.0), want_small)
            _assert_hits_equal(eng.query(centers, R), want)
            sj = eng.self_join(R, sqrt_test=True)
        if rows == "narrow":
            sj_narrow = eng.self_join(R, sqrt_test=True)
        eng.close()
    # which rows a call ran on shows in its work items
    assert survivors["forced"] > survivors["narrow"] and survivors["loaded"] == survivors["forced"]
    # (23, 52) and (25, 58) are inside the radius rule, (21, 40) is not
    assert survivors["by-radius"] == (survivors["forced"] if k > 21 else survivors["narrow"])
    for key in ("i", "j", "table", "dist"):
        assert np.array_equal(sj[key], sj_narrow[key])


def test_batches_split_when_the_survivor_counter_would_overflow(oracle, monkeypatch):
    """A batch whose filters pass more pairs than the 32-bit survivor counter holds is repeated in
    halves (hs_capi.hip run_query).  HS_TEST_SPLIT_ABOVE -- a hook of the library's TEST build only
    (libhsearch_amd_hooks.so: the same kernel objects, C-ABI layer compiled with -DHS_TEST_HOOKS) --
    makes every batch above 150 queries report that overflow: hits, order, candidates and the
    self-join's edges are those of the unsplit run of the product library."""
    k, K, L, W, R, n, nq = 25, 4, 5, 150.0, 45.0, 20011, 1203
    a, b = synth.make_planes(k, K, L, W, seed=35)
    codes = synth.make_db(n, k, seed=36)
    centers, _ = synth.make_queries(codes, nq, seed=37, jitter=0.2)
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    plain = eng.self_join(R, sqrt_test=True)
    eng.close()
    monkeypatch.setenv("HS_TEST_SPLIT_ABOVE", "150")
    eng = Engine(k, K, L, W, a, b, hooks=True)
    eng.index_build(codes)
    for mode in ("auto", "stream"):
        eng.set_verify_mode(mode)
        got = eng.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
    assert eng.profile()["verify_launches"] >= nq // 150
    bf = eng.bruteforce(centers[:400], R)
    split = eng.self_join(R, sqrt_test=True)
    eng.close()
    monkeypatch.delenv("HS_TEST_SPLIT_ABOVE")
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    bf0 = eng.bruteforce(centers[:400], R)
    eng.close()
    for key in ("q", "id", "dist"):
        assert np.array_equal(bf[key], bf0[key])
    for key in ("i", "j", "table", "dist"):
        assert np.array_equal(split[key], plain[key])


def test_join_with_many_queries_per_bucket(oracle):
    """Coarse keys (K=2, large W): a handful of huge buckets, each probed by hundreds of queries --
    multi-chunk, multi-tile work items of the bucket join, plus ragged tile/chunk remainders."""
    k, K, L, W, R, n, nq = 25, 2, 3, 400.0, 42.0, 30011, 1531
    a, b = synth.make_planes(k, K, L, W, seed=5)
    codes = synth.make_db(n, k, seed=6)
    centers, _ = synth.make_queries(codes, nq, seed=7, jitter=0.2)
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    assert max(info["max_bucket"]) > 4000
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    for mode in ("join", "join16", "stream"):
        eng.set_verify_mode(mode)
        got = eng.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
    # queries slightly outside the table's range: int8 cannot carry them, fp16 can
    wide = centers[:64].copy()
    wide[:8, ::8] += 9.0
    eng.set_verify_mode("join")
    got = eng.query(wide, R)
    _assert_hits_equal(got, oracle.search(a, b, W, R, oracle.embed_codes(codes), wide))
    prof = eng.profile()
    assert prof["join_batches"] > 0 and prof["join_i8_batches"] == 0
    # far-away and huge-magnitude queries: fp16 cannot carry them either, the batch must stream
    far = centers[:64].copy()
    far[:8] *= 40.0
    far[8:16] += 3.0e4
    got = eng.query(far, R)
    _assert_hits_equal(got, oracle.search(a, b, W, R, oracle.embed_codes(codes), far))
    assert eng.profile()["join_batches"] == 0
    eng.close()
    # thin segments (< 3 probing queries or < 16 members) routed to the per-pair filter instead
    eng = Engine(k, K, L, W, a, b, options=dict(join_min_q=3, join_min_m=16))
    eng.index_build(codes)
    eng.set_verify_mode("join")
    got = eng.query(centers, R)
    _assert_hits_equal(got, want)
    prof = eng.profile()
    assert 0 < prof["join_pairs"] < prof["candidates"] == int(want["cand"].sum())   # some pairs took the thin path
    assert prof["join_pairs_issued"] >= prof["join_pairs"]
    eng.close()
    # the default routing: every segment through the join
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    eng.set_verify_mode("join")
    _assert_hits_equal(eng.query(centers, R), want)
    prof = eng.profile()
    assert prof["join_pairs"] == prof["candidates"] == int(want["cand"].sum())
    eng.close()


def test_bruteforce_matches_oracle(oracle):
    k, n, nq, R = 25, 6000, 200, 40.0
    a, b = synth.make_planes(k, 4, 2, 100.0)
    codes = synth.make_db(n, k)
    centers, _ = synth.make_queries(codes, nq, jitter=0.25)
    eng = Engine(k, 4, 2, 100.0, a, b)
    eng.index_build(codes)
    want = oracle.bruteforce(oracle.embed_codes(codes), centers, R)
    got = eng.bruteforce(centers, R)
    _assert_hits_equal(got, want)
    eng.close()


@pytest.mark.parametrize("n,nq,topk", [(6000, 50, 10), (60000, 40, 10), (50000, 20, 3)])
def test_bruteforce_topk_matches_oracle(oracle, n, nq, topk):
    k = 25
    a, b = synth.make_planes(k, 4, 2, 100.0)
    codes = synth.make_db(n, k, seed=3)
    centers, _ = synth.make_queries(codes, nq, jitter=0.25, seed=4)
    eng = Engine(k, 4, 2, 100.0, a, b)
    eng.index_build(codes)
    want_id, want_d2 = oracle.bruteforce_topk(oracle.embed_codes(codes), centers, topk)
    got_id, got_d2 = eng.bruteforce_topk(centers, topk)
    assert np.array_equal(got_id, want_id)
    assert np.array_equal(got_d2, want_d2)
    eng.close()


def test_aliased_key_strings_share_a_bucket(oracle):
    """Small W, K=2: distinct int tuples whose decimal concatenations coincide -- e.g. (1,23) and
    (12,3) -- are ONE key in the reference (lsh.hpp:51-59).  The index must reproduce that."""
    import hsearch_amd
    k, K, L, W, R, n, nq = 25, 2, 2, 1.0, 45.0, 30000, 300
    a, b = synth.make_planes(k, K, L, W, seed=99)
    codes = synth.make_db(n, k, seed=8)
    centers, _ = synth.make_queries(codes, nq, seed=9)
    pts = oracle.embed_codes(codes)
    ints = oracle.hash_all(a, b, W, pts)
    aliased = 0
    for l in range(L):
        groups = {}
        for t in map(tuple, ints[:, l]):
            groups.setdefault(hsearch_amd.key_string(t), set()).add(t)
        aliased += sum(1 for v in groups.values() if len(v) > 1)
    assert aliased > 0, "workload does not exercise key aliasing"
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    ix = oracle.Index(a, b, W, pts)
    assert info["n_buckets"] == ix.table_sizes()
    want = ix.query(centers, R)
    got = eng.query(centers, R)
    assert np.array_equal(got["cand"], want["cand"])
    _assert_hits_equal(got, want)
    eng.close()


@pytest.mark.parametrize("k,K,L,W,n", [(25, 20, 4, 160.0, 60011), (25, 24, 2, 90.0, 9001), (25, 2, 2, 1.0, 30000),
                                       (15, 3, 3, 0.004, 5003), (25, 28, 2, 300.0, 4001), (39, 16, 3, 212.0, 20011)])
def test_probe_reads_directory_records_or_the_arrays(oracle, k, K, L, W, n):
    """Probe, SURVEY 8(a) a8: by default the probe reads ONE 64-byte directory record per candidate bucket
    (fingerprint, boundaries, the bucket ints as int16: hs_dir_records_kernel) and falls back to the directory
    arrays where a table has none -- K > 24 (here K = 28), a bucket int outside 16 bits (W = 0.004: ints of
    +-10^5) -- or the option probe_records = 0 says so.  Every form gives the oracle's candidates and hits;
    aliased key strings (K = 2, W = 1: a found fingerprint whose tuple differs goes to the slow queue) included."""
    R = 45.0
    a, b = synth.make_planes(k, K, L, W, seed=131)
    codes = synth.make_db(n, k, seed=132)
    centers, _ = synth.make_queries(codes, 700, seed=133, jitter=0.01)
    pts = oracle.embed_codes(codes)
    ix = oracle.Index(a, b, W, pts)
    want = ix.query(centers, R)
    if W < 0.01:
        assert np.abs(oracle.hash_all(a, b, W, pts)).max() > 40000
    for records in (1, 0):
        eng = Engine(k, K, L, W, a, b, options=dict(probe_records=records))
        info = eng.index_build(codes)
        assert info["n_buckets"] == ix.table_sizes()
        for rep in range(2):
            got = eng.query(centers, R)
            assert np.array_equal(got["cand"], want["cand"]), (records, rep)
            _assert_hits_equal(got, want)
        eng.close()
    ix.close()


def test_edge_cases(oracle):
    k, K, L, W, R = 25, 4, 3, 100.0, 40.0
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(500, k)
    centers, _ = synth.make_queries(codes, 20)
    eng = Engine(k, K, L, W, a, b)
    import hsearch_amd
    with pytest.raises(hsearch_amd.HsError):           # query before build
        eng.query(centers, R)
    eng.index_build(codes[:0])                          # empty DB
    assert len(eng.query(centers, R)["q"]) == 0
    eng.index_build(codes)
    assert len(eng.query(centers[:0], R)["q"]) == 0     # empty query batch
    got = eng.query(centers, 0.0)                       # R = 0: exact duplicates only
    want = oracle.search(a, b, W, 0.0, oracle.embed_codes(codes), centers)
    _assert_hits_equal(got, want)
    bad = codes.copy()
    bad[3, 7] = 20
    with pytest.raises(hsearch_amd.HsError):            # residue code outside the alphabet
        eng.index_build(bad)
    # capacity protocol: too small a buffer reports the required size, then succeeds
    eng.index_build(codes)
    full = eng.query(centers, R)
    small = eng.query(centers, R, cap=1)
    _assert_hits_equal(small, full)
    # duplicates in the DB (same k-mer many times) and a single-point DB
    dup = np.repeat(codes[:5], 40, axis=0)
    eng.index_build(dup)
    want = oracle.search(a, b, W, R, oracle.embed_codes(dup), oracle.embed_codes(codes[:5]))
    _assert_hits_equal(eng.query(oracle.embed_codes(codes[:5]), R), want)
    assert len(want["q"]) == 200
    eng.close()


def test_multi_batch_queries_and_rebuild(oracle):
    """More queries than one internal batch (131072): batches must concatenate in the reference's
    global order; and a handle can be re-indexed with another DB."""
    k, K, L, W, R = 25, 4, 3, 120.0, 40.0
    a, b = synth.make_planes(k, K, L, W, seed=12)
    codes = synth.make_db(3000, k, seed=13)
    nq = 140_000
    centers, _ = synth.make_queries(codes, nq, seed=14)
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    got = eng.query(centers, R)
    ix = oracle.Index(a, b, W, oracle.embed_codes(codes))
    want = ix.query(centers, R)
    assert np.array_equal(got["cand"], want["cand"])
    _assert_hits_equal(got, want)
    assert len(want["q"]) > nq // 2 and want["q"].max() > 131072
    # rebuild with a different, larger DB on the same handle
    codes2 = synth.make_db(9000, k, seed=15)
    info = eng.index_build(codes2)
    assert info["n"] == 9000
    c2, _ = synth.make_queries(codes2, 500, seed=16, jitter=0.2)
    want2 = oracle.search(a, b, W, R, oracle.embed_codes(codes2), c2)
    _assert_hits_equal(eng.query(c2, R), want2)
    eng.close()


def test_windows_build_matches_explicit_kmers(oracle):
    """the DB given as a concatenated residue buffer + sequence starts; the
    windows are enumerated on the device (kmer_search.cpp:64-83 order).  Must equal the index over
    the explicitly materialised windows, and the oracle's answer over them."""
    k, K, L, W, R = 25, 4, 4, 100.0, 40.0
    rng = np.random.default_rng(77)
    lens = rng.integers(0, 400, size=300)
    lens[[3, 17, 18, 250]] = [0, k - 1, k, 5]          # empty / too short / exactly one window
    residues = rng.integers(0, 20, size=int(lens.sum()), dtype=np.uint8)
    seq_start = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    want_pos = np.concatenate([np.arange(int(s), int(s) + max(int(n) - k + 1, 0))
                               for s, n in zip(seq_start[:-1], lens)]).astype(np.uint32)
    codes = np.stack([residues[p:p + k] for p in want_pos])
    a, b = synth.make_planes(k, K, L, W, seed=21)
    centers, _ = synth.make_queries(codes, 400, seed=22, jitter=0.25)
    e1 = Engine(k, K, L, W, a, b)
    info1 = e1.index_build(codes)
    e2 = Engine(k, K, L, W, a, b)
    info2, pos = e2.index_build_windows(residues, seq_start)
    assert np.array_equal(pos, want_pos)
    assert info2["n"] == len(codes) and info2["n_buckets"] == info1["n_buckets"]
    assert info2["max_bucket"] == info1["max_bucket"]
    got1, got2 = e1.query(centers, R), e2.query(centers, R)
    assert np.array_equal(got1["cand"], got2["cand"])
    _assert_hits_equal(got2, got1)
    _assert_hits_equal(got2, oracle.search(a, b, W, R, oracle.embed_codes(codes), centers))
    # no sequence long enough: an empty index, queries return nothing
    info3, pos3 = e2.index_build_windows(residues[:10], np.array([0, 4, 10], dtype=np.uint64))
    assert info3["n"] == 0 and len(pos3) == 0
    assert len(e2.query(centers, R)["q"]) == 0
    e1.close()
    e2.close()


def test_index_save_load_round_trip(oracle, tmp_path):
    """a saved index restored into a fresh handle answers exactly like the
    handle that built it (all verify modes); a file written for other planes is refused."""
    import hsearch_amd
    k, K, L, W, R = 25, 8, 5, 150.0, 40.0
    a, b = synth.make_planes(k, K, L, W, seed=41)
    codes = synth.make_db(20000, k, seed=42)
    centers, _ = synth.make_queries(codes, 600, seed=43, jitter=0.25)
    e1 = Engine(k, K, L, W, a, b)
    info1 = e1.index_build(codes)
    want = e1.query(centers, R)
    path = tmp_path / "index.hsidx"
    e1.index_save(path)
    e1.close()
    e2 = Engine(k, K, L, W, a, b)
    with pytest.raises(hsearch_amd.HsError):
        e2.query(centers, R)                           # nothing loaded yet
    info2 = e2.index_load(path)
    for f in ("n", "n_buckets", "max_bucket", "key_seed"):
        assert info2[f] == info1[f], f
    for mode in ("auto", "stream", "join16"):
        e2.set_verify_mode(mode)
        got = e2.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
    _assert_hits_equal(want, oracle.search(a, b, W, R, oracle.embed_codes(codes), centers))
    e2.close()
    a2, b2 = synth.make_planes(k, K, L, W, seed=44)
    e3 = Engine(k, K, L, W, a2, b2)
    with pytest.raises(hsearch_amd.HsError) as ei:
        e3.index_load(path)
    assert "HS_ERR_IO" in str(ei.value)
    with pytest.raises(hsearch_amd.HsError):
        e3.index_load(tmp_path / "missing.hsidx")
    e3.close()


def test_index_load_checks_file_content(oracle, tmp_path):
    """hs_index_load must not trust a file's tables.  (1) The file a handle saves passes
    the host-side check, and is byte-identical to the one the test helper writes from the ORACLE's
    bucket ints (so the format is pinned from both sides).  (2) A handle loads the oracle-written
    file and answers like a handle that built the index.  (3) Bit flips, truncation and
    self-consistent files that break a content rule (out-of-range / repeated ids, broken bucket
    boundaries, mis-ordered or mismatched fingerprints) give HS_ERR_IO -- no device fault -- and
    the handle still works afterwards."""
    import hsearch_amd
    import indexfile
    k, K, L, W, R, n = 25, 6, 4, 120.0, 40.0, 5000
    a, b = synth.make_planes(k, K, L, W, seed=51)
    codes = synth.make_db(n, k, seed=52)
    centers, _ = synth.make_queries(codes, 300, seed=53, jitter=0.2)
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    want = eng.query(centers, R)
    saved = tmp_path / "saved.hsidx"
    eng.index_save(saved)
    hsearch_amd.index_file_check(saved)
    tables = indexfile.build_tables(oracle.hash_all(a, b, W, oracle.embed_codes(codes)), seed=info["key_seed"])
    mine = tmp_path / "oracle.hsidx"
    indexfile.write(mine, k, K, L, W, a, b, codes, tables, seed=info["key_seed"])
    assert mine.read_bytes() == saved.read_bytes()
    e2 = Engine(k, K, L, W, a, b)
    info2 = e2.index_load(mine)
    assert info2["n_buckets"] == info["n_buckets"] and info2["max_bucket"] == info["max_bucket"]
    _assert_hits_equal(e2.query(centers, R), want)

    def ids(s, l=0):
        return s[4 + 4 * l]
    tampers = [
        lambda s: ids(s).__setitem__(5, 10**9),
        lambda s: ids(s, 2).__setitem__(5, ids(s, 2)[6]),
        lambda s: s[4 + 2].__setitem__(0, 1),
        lambda s: s[4 + 2].__setitem__(3, s[4 + 2][2]),
        lambda s: s[4 + 2].__setitem__(len(s[4 + 2]) - 1, 7),
        lambda s: s[4 + 2].__setitem__(2, 4_000_000_000),
        lambda s: s[4 + 1].__setitem__(slice(0, 2), s[4 + 1][[1, 0]]),
        lambda s: s[4 + 3].__setitem__((0, 0), s[4 + 3][0, 0] + 1),
        lambda s: s[3].__setitem__((0, 0), 21),
    ]
    bad = tmp_path / "bad.hsidx"
    for t in tampers:
        indexfile.write(bad, k, K, L, W, a, b, codes, tables, seed=info["key_seed"], tamper=t)
        with pytest.raises(hsearch_amd.HsError) as ei:
            e2.index_load(bad)
        assert "HS_ERR_IO" in str(ei.value)
        with pytest.raises(hsearch_amd.HsError):
            e2.query(centers, R)                       # a refused file leaves no half-loaded index
    data = saved.read_bytes()
    rng = np.random.default_rng(2)
    for pos in rng.integers(300, len(data), size=12).tolist():
        flipped = bytearray(data)
        flipped[pos] ^= 0x40
        bad.write_bytes(bytes(flipped))
        with pytest.raises(hsearch_amd.HsError) as ei:
            e2.index_load(bad)
        assert "HS_ERR_IO" in str(ei.value)
    bad.write_bytes(data[:len(data) // 2])
    with pytest.raises(hsearch_amd.HsError):
        e2.index_load(bad)
    e2.index_load(saved)                               # and the handle is still usable
    _assert_hits_equal(e2.query(centers, R), want)
    e2.close()
    eng.close()


def test_klsh_codes_match_reference_golden(oracle, golden_dir):
    """hs_klsh_codes against the codes of the reference's own KLSH object, and
    against the oracle on fresh sequences (short ones are skipped like pcluster.cpp:22-24)."""
    import os
    import hsearch_amd
    z = np.load(os.path.join(golden_dir, "klsh.npz"))
    codes, unc = hsearch_amd.klsh_codes(z["classes"], z["seq_start"], z["w"], z["b"], z["t"])
    assert not unc.any()                      # no bit sits on the cos(.) + t = 0 boundary
    assert np.array_equal(codes, z["codes"])
    rng = np.random.default_rng(9)
    lens = rng.integers(0, 3000, size=500)
    lens[:4] = [0, 1, 2, 3]
    classes = rng.integers(0, 8, size=int(lens.sum()), dtype=np.uint8)
    start = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    w, b, t = hsearch_amd.klsh_draw_planes()
    codes, unc = hsearch_amd.klsh_codes(classes, start, w, b, t)
    assert not unc.any()
    for i, n in enumerate(lens):
        if n < 3:
            assert codes[i] == hsearch_amd.KLSH_NONE
        else:
            f = oracle.klsh_features(classes[int(start[i]):int(start[i + 1])])
            assert int(codes[i]) == oracle.klsh_hash(w, b, t, f)
    with pytest.raises(hsearch_amd.HsError):
        hsearch_amd.klsh_codes(np.array([9], dtype=np.uint8), np.array([0, 1], dtype=np.uint64), w, b, t)


def test_set_planes_rebuild_equals_fresh_handle(oracle):
    """hs_set_planes: a handle re-seeded with another family and rebuilt answers like a handle
    created with that family; queries before the rebuild are refused."""
    import hsearch_amd
    k, K, L, W, R, n, nq = 25, 6, 3, 120.0, 45.0, 5000, 300
    codes = synth.make_db(n, k, seed=21)
    centers, _ = synth.make_queries(codes, nq, seed=22, jitter=0.2)
    a1, b1 = synth.make_planes(k, K, L, W, seed=31)
    a2, b2 = synth.make_planes(k, K, L, W, seed=32)
    eng = Engine(k, K, L, W, a1, b1)
    eng.index_build(codes)
    first = eng.query(centers, R)
    _assert_hits_equal(first, oracle.search(a1, b1, W, R, oracle.embed_codes(codes), centers))
    eng.set_planes(a2, b2)
    with pytest.raises(hsearch_amd.HsError):
        eng.query(centers, R)                      # the tables were keyed by the old family
    assert np.array_equal(eng.hash_codes(codes[:50]), oracle.hash_all(a2, b2, W, oracle.embed_codes(codes[:50])))
    eng.index_build(codes)
    got = eng.query(centers, R)
    want = oracle.search(a2, b2, W, R, oracle.embed_codes(codes), centers)
    _assert_hits_equal(got, want)
    assert np.array_equal(got["cand"], want["cand"])
    assert not np.array_equal(first["cand"], got["cand"])
    eng.close()


def test_build_sorts_again_when_partial_fingerprints_interleave():
    """From 2^20 k-mers on, the build's radix sort looks at the fingerprints' top 48 bits only; distinct
    fingerprints that agree there would interleave, which hs_check_runs_kernel reports and a second
    sort on all bits repairs.  Option sort_from_bit = 56 (8 bits) makes that certain: the index and the hits
    must be those of the one-pass form (= 0) and of the default."""
    k, K, L, W, R, n, nq = 25, 8, 3, 150.0, 45.0, (1 << 20) + 77, 2003
    a, b = synth.make_planes(k, K, L, W, seed=55)
    codes = synth.make_db(n, k, seed=56)
    centers, _ = synth.make_queries(codes, nq, seed=57, jitter=0.2)
    res = {}
    for from_bit in ("0", "56", None):
        opts = dict(build_grouping=1)      # the sorting form of the grouping (default: hs_group.hip)
        if from_bit is not None:
            opts["sort_from_bit"] = int(from_bit)
        eng = Engine(k, K, L, W, a, b, options=opts)
        info = eng.index_build(codes)
        assert min(info["n_buckets"]) > 300          # far more buckets than 8 bits tell apart
        res[from_bit] = (info["n_buckets"], info["max_bucket"], eng.query(centers, R))
        eng.close()
    assert len(res["0"][2]["q"]) > 1000
    for other in ("56", None):
        assert res[other][0] == res["0"][0] and res[other][1] == res["0"][1]
        for key in ("q", "id", "table", "dist", "cand"):
            assert np.array_equal(res[other][2][key], res["0"][2][key])


@pytest.mark.parametrize("n", [(1 << 20) - 1, 1 << 20, (1 << 20) + 1])
def test_build_at_the_sort_path_boundary(n):
    """rocPRIM sorts up to merge_sort_limit = 2^20 items with a merge sort whose comparator for a bit
    range ending at bit 64 is built from 1 << 64 (hs_prims.hip): the build may hand it a range only
    above that size.  Round 2's cut was n >= 2^20, one too early.  Builds at the boundary and either
    side of it -- hs_index_build and hs_index_build_subset -- equal the all-bits build, and the
    exact-copy queries find their k-mers."""
    k, K, L, W, R, nq = 25, 8, 2, 150.0, 30.0, 1500
    a, b = synth.make_planes(k, K, L, W, seed=75)
    codes = synth.make_db(n, k, seed=76)
    src = np.random.default_rng(77).integers(0, n, size=nq)
    centers = synth.embed(codes[src])
    res = {}
    for mode in ("0", None, "subset", "group"):
        opts = dict(build_grouping=1)      # the sorting form of the grouping, whose boundary this is ...
        if mode == "group":
            opts = {}                      # ... and the default form (hs_group.hip) beside it
        if mode == "0":
            opts["sort_from_bit"] = 0
        eng = Engine(k, K, L, W, a, b, options=opts)
        info = eng.index_build_subset(codes, None) if mode == "subset" else eng.index_build(codes)
        res[mode] = (info["n_buckets"], info["max_bucket"], eng.query(centers, R))
        eng.close()
    got = res["0"][2]
    first = {}
    for q, i, t in zip(got["q"].tolist(), got["id"].tolist(), got["table"].tolist()):
        if i == src[q]:
            first[q] = t
    assert all(first.get(q) == 0 for q in range(nq))      # an exact copy shares its k-mer's bucket in table 0
    for other in (None, "subset", "group"):
        assert res[other][0] == res["0"][0] and res[other][1] == res["0"][1]
        for key in ("q", "id", "table", "dist", "cand"):
            assert np.array_equal(res[other][2][key], got[key])


@pytest.mark.parametrize("k,K,L,W,R", [(25, 6, 5, 120.0, 45.0), (25, 2, 3, 400.0, 42.0), (15, 5, 4, 90.0, 32.0),
                                      (39, 6, 3, 260.0, 50.0)])
def test_probe_grouping_by_counting_sort_and_by_probe_sort(oracle, k, K, L, W, R):
    """The probes are grouped by bucket in front of the join either by a counting sort over the bucket
    slots or, when buckets far outnumber probes (C3 shape: 1.3e8 slots for 4e6 probes), by a radix sort
    of the probes on their bucket number.  Both forms, forced in turn, give the oracle's hits -- for
    searches, with thin segments routed away from the join, and for the self-join from codes."""
    n, nq = 20011, 1203
    a, b = synth.make_planes(k, K, L, W, seed=65)
    base = synth.make_db(n // 2, k, seed=66)
    near = base.copy()                      # every k-mer once more with one substitution: self-join edges
    rng = np.random.default_rng(68)
    near[np.arange(len(near)), rng.integers(0, k, size=len(near))] = rng.integers(0, 20, size=len(near), dtype=np.uint8)
    codes = np.concatenate([base, near])
    centers, _ = synth.make_queries(codes, nq, seed=67, jitter=0.2)
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    assert len(want["q"]) > 100
    edges = {}
    for mode in ("dense", "sparse"):
        for thin in (False, True):
            opts = dict(seg_mode={"sparse": 1, "dense": 2}[mode])
            if thin:
                opts.update(join_min_q=3, join_min_m=16)
            eng = Engine(k, K, L, W, a, b, options=opts)
            eng.index_build(codes)
            for vm in ("join", "join16") if k <= 25 else ("join",):
                eng.set_verify_mode(vm)
                got = eng.query(centers, R)
                assert np.array_equal(got["cand"], want["cand"])
                _assert_hits_equal(got, want)
                assert eng.profile()["join_batches"] > 0
            eng.set_verify_mode("auto")
            edges[(mode, thin)] = eng.self_join(R, sqrt_test=True)
            eng.close()
    ref = edges[("dense", False)]
    assert len(ref["i"]) > 100
    for e in edges.values():
        for key in ("i", "j", "table", "dist"):
            assert np.array_equal(e[key], ref[key])


@pytest.mark.parametrize("families", [(700,), (700, 1500), (300, 9000), (17000,)])
def test_hit_ordering_paths_with_many_hits_per_query(oracle, families):
    """Output order = (query, table of first sight, id) (motif_both_points.cpp:224-245).  The hits are
    bucketed by query; a query with up to 48 is ordered by one thread, one with up to 1024 / 8192 by a
    block (bitonic sort in LDS, hs_hit_order_block_kernel), one with more by a block that sorts chunks in
    LDS and merges them through global memory (hs_hit_order_huge_kernel); any batch under option sort_hits = 1
    is radix-sorted on the full key.  A DB with families of near-identical k-mers (queries inside a family
    of m get ~m hits, most others a few: 700 -> the small blocks, 1500 -> the large ones, 9000 -> two
    chunks and one merge, 17000 -> three chunks, two merge levels) through all of them against the oracle."""
    k, K, L, W, R, n, nq = 25, 4, 5, 150.0, 40.0, 20011, 903
    a, b = synth.make_planes(k, K, L, W, seed=75)
    codes = synth.make_db(n, k, seed=76)
    rng = np.random.default_rng(77)
    at, nfq = 1000, 0
    for fi, m in enumerate(families):
        fam = np.repeat(codes[fi:fi + 1], m, axis=0)
        fam[np.arange(m), rng.integers(0, k, size=m)] = rng.integers(0, 20, size=m, dtype=np.uint8)
        codes[at:at + m] = fam
        at += m
    centers, _ = synth.make_queries(codes, nq, seed=78, jitter=0.2)
    at = 1000
    for m in families:
        centers[nfq:nfq + 20] = synth.embed(codes[at:at + 20])   # 20 queries inside each family
        nfq += 20
        at += m
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    per_q = np.bincount(want["q"], minlength=nq)
    assert per_q.max() > 0.7 * max(families) and (np.median(per_q) < 20 or sum(families) > n // 4)
    want_few = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers[nfq:])
    for sort_all in (False, True):
        eng = Engine(k, K, L, W, a, b, options=dict(sort_hits=1) if sort_all else None)
        eng.index_build(codes)
        for mode in ("auto", "stream"):
            eng.set_verify_mode(mode)
            _assert_hits_equal(eng.query(centers, R), want)
        # without the family queries every query has few hits: the one-thread ordering
        few = eng.query(centers[nfq:], R)
        _assert_hits_equal(few, want_few)
        eng.close()


@pytest.mark.parametrize("k,K,L,W,R", [(25, 6, 5, 140.0, 42.0), (15, 5, 4, 90.0, 32.0), (39, 6, 3, 260.0, 50.0),
                                      (60, 4, 3, 400.0, 70.0)])
def test_centres_that_are_kmers_run_from_their_codes(oracle, k, K, L, W, R):
    """hs_query looks at its centres first: when every group of 8 doubles is a row of the coordinate
    table bit for bit -- the reference's centres are k-mers (KmerToCoordinates, hclust2.cpp:49-62) -- the
    call runs from the residue codes like hs_query_codes.  Same hits, order, distances and candidate
    counts as the oracle's and as the same call with the recognition off; one centre moved by one ulp in
    one coordinate and the whole call stays on the points path; -0.0 for a 0.0 likewise."""
    n, nq = 15013, 703
    a, b = synth.make_planes(k, K, L, W, seed=95)
    codes = synth.make_db(n, k, seed=96)
    qcodes, _ = synth.make_query_codes(codes, nq, seed=97)
    centers = synth.embed(qcodes)
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    assert len(want["q"]) > 50
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    for mode in ("auto", "stream"):
        eng.set_verify_mode(mode)
        got = eng.query(centers, R)
        assert eng.profile()["queries_recognised"] == nq
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
    eng.set_verify_mode("auto")
    moved = centers.copy()
    moved[nq // 2, 8 * (k // 2) + 3] = np.nextafter(moved[nq // 2, 8 * (k // 2) + 3], np.inf)
    want_moved = oracle.search(a, b, W, R, oracle.embed_codes(codes), moved)
    got = eng.query(moved, R)
    assert eng.profile()["queries_recognised"] == 0
    _assert_hits_equal(got, want_moved)
    zeros = np.argwhere(centers == 0.0)
    if len(zeros):
        neg = centers.copy()
        neg[zeros[0][0], zeros[0][1]] = -0.0
        got = eng.query(neg, R)
        assert eng.profile()["queries_recognised"] == 0
        _assert_hits_equal(got, want)          # (-0.0 and 0.0 give the same distances and dot products)
    eng.close()
    eng = Engine(k, K, L, W, a, b, options=dict(recognise_kmers=0))
    eng.index_build(codes)
    got = eng.query(centers, R)
    assert eng.profile()["queries_recognised"] == 0
    assert np.array_equal(got["cand"], want["cand"])
    _assert_hits_equal(got, want)
    eng.close()


@pytest.mark.parametrize("k,K,L,W,R", [(25, 6, 5, 140.0, 42.0), (15, 5, 4, 90.0, 32.0), (39, 6, 3, 260.0, 50.0),
                                      (60, 4, 3, 400.0, 70.0), (25, 20, 32, 320.0, 45.0)])
def test_queries_given_as_residue_codes(oracle, k, K, L, W, R):
    """hs_query_codes -- the queries are k-mers (the reference's usual centres:
    KmerToCoordinates, hclust2.cpp:49-62) given as k residue codes instead of 8k doubles.  Hits, order,
    candidates and fp64 distances equal hs_query's on the embedded codes and the oracle's, in every
    verify mode: with the int8 join the whole batch runs from the codes (no centre is ever embedded),
    otherwise (streaming filter, fp16 join, k > 50) they are embedded on the device.  Several batches; a
    code outside the alphabet is HS_ERR_INVALID; custom coordinate tables."""
    from hsearch_amd import capi
    n, nq = 20011, 1203
    a, b = synth.make_planes(k, K, L, W, seed=85)
    codes = synth.make_db(n, k, seed=86)
    qcodes, _ = synth.make_query_codes(codes, nq, seed=87)
    centers = synth.embed(qcodes)
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    assert len(want["q"]) > 100
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    for mode in ("auto", "join", "stream", "join16"):
        eng.set_verify_mode(mode)
        got = eng.query_codes(qcodes, R)
        assert np.array_equal(got["cand"], want["cand"]), mode
        _assert_hits_equal(got, want)
        prof = eng.profile()
        if mode in ("auto", "join") and k <= 50:
            assert prof["join_i8_batches"] > 0
        ref = eng.query(centers, R)
        for key in ("q", "id", "table", "dist", "cand"):
            assert np.array_equal(got[key], ref[key]), (mode, key)
    eng.set_verify_mode("auto")
    bad = qcodes.copy()
    bad[nq // 2, k // 2] = 20
    with pytest.raises(capi.HsError) as e:
        eng.query_codes(bad, R)
    assert e.value.status == capi.HS_ERR_INVALID
    _assert_hits_equal(eng.query_codes(qcodes, R), want)          # the handle is fine afterwards
    assert len(eng.query_codes(qcodes[:0], R)["q"]) == 0
    eng.close()


@pytest.mark.parametrize("letters", [11, 21, 23, 29])
def test_queries_as_codes_in_batches_and_custom_table(oracle, letters):
    """An alphabet of its own (11 and 21 letters: the exact pass takes its terms from the table of rounded
    squares, hs_finalize_codes_kernel -- 21 is the largest it holds; 23 and 29 letters: above
    HS_FIN_TABLE_ALPHABET, the two-row form of hs_finalize_kernel), several query batches per call, a code outside the alphabet, and the same
    queries given as points (recognised as k-mers of that table)."""
    k, K, L, W, R, n, nq = 25, 5, 4, 120.0, 44.0, 9001, 777
    rng = np.random.default_rng(90)
    table = rng.normal(0.0, 6.0, size=(letters, 8))                # an alphabet of its own
    a, b = synth.make_planes(k, K, L, W, seed=91)
    codes = rng.integers(0, letters, size=(n, k), dtype=np.uint8)
    qcodes = codes[rng.integers(0, n, size=nq)].copy()
    qcodes[np.arange(nq), rng.integers(0, k, size=nq)] = rng.integers(0, letters, size=nq, dtype=np.uint8)
    pts = table[codes].reshape(n, -1)
    centers = table[qcodes].reshape(nq, -1)
    want = oracle.search(a, b, W, R, pts, centers)
    assert len(want["q"]) > 100
    eng = Engine(k, K, L, W, a, b, coords=table, options=dict(query_batch=100))
    eng.index_build(codes)
    for mode in ("auto", "stream"):
        eng.set_verify_mode(mode)
        got = eng.query_codes(qcodes, R)
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
    eng.set_verify_mode("auto")
    got = eng.query(centers, R)                                     # the same k-mers as points
    assert eng.profile()["queries_recognised"] == nq
    assert np.array_equal(got["cand"], want["cand"])
    _assert_hits_equal(got, want)
    bad = qcodes.copy()
    bad[5, 3] = letters                                             # a row the table does not have
    from hsearch_amd import capi
    with pytest.raises(capi.HsError):
        eng.query_codes(bad, R)
    eng.close()


@pytest.mark.parametrize("K,W,nq", [(2, 400.0, 3000), (3, 300.0, 1203), (5, 160.0, 6000), (8, 150.0, 900)])
def test_query_resident_and_query_streaming_join_kernels(oracle, K, W, nq):
    """k <= 25 with 4-column rows: segments (bucket x the batch's queries probing it) with at most 64
    probing queries go through hs_join8r_kernel (query rows resident in registers, member tiles
    streamed, 16-query column tiles), the others through hs_join8x_kernel (member operands resident,
    query tiles streamed).  Coarse to fine keys give segments of every class -- 1..16, 17..32, 33..48,
    49..64 and more queries; buckets of one member up to thousands, ragged last tiles -- and both
    routings (the resident class forced; option join_resident = 1: everything through the streaming kernel)
    must give the oracle's candidates, hits, order and distances, also with several query batches per call."""
    k, L, R, n = 25, 4, 44.0, 30011
    a, b = synth.make_planes(k, K, L, W, seed=95)
    codes = synth.make_db(n, k, seed=96)
    centers, _ = synth.make_queries(codes, nq, seed=97, jitter=0.2)
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    assert len(want["q"]) > 300
    seen = {}
    # (join_resident=2: by default the resident kernel runs only while its class is the bulk of a batch's items)
    # join_xcd_run: hs_join8x_kernel's items dealt in XCD-local runs of 1 / 4 chunks (on by itself only for big
    # batches) -- with this few items most XCDs find their runs dry at once and take from their neighbours'
    routings = {"default": dict(join_resident=2), "no_r": dict(join_resident=1),
                "batches": dict(join_resident=2, query_batch=257),
                "no_r_xcd_runs": dict(join_resident=1, join_xcd_run=1), "xcd_runs": dict(join_resident=2, join_xcd_run=4),
                # chunks of two / three work items per counter access: item lists that end one or two short of a chunk
                "no_r_xcd_runs_chunk3": dict(join_resident=1, join_xcd_run=2, join_chunk=3),
                "xcd_runs_chunk2": dict(join_resident=2, join_xcd_run=1, join_chunk=2)}
    for routing, opts in routings.items():
        eng = Engine(k, K, L, W, a, b, options=opts)
        eng.index_build(codes)
        eng.set_verify_mode("join")
        for rep in range(2):          # the second call runs on the first one's capacity hint (no host round trip)
            got = eng.query(centers, R)
            assert np.array_equal(got["cand"], want["cand"]), (routing, rep)
            _assert_hits_equal(got, want)
        p = eng.profile()
        assert p["join_i8_batches"] > 0 and p["join_pairs"] > 0
        seen[routing] = (p["join_pairs"], p["join_pairs_issued"], p["join_items_resident"])
        eng.close()
    # the same pairs either way; the resident kernel issues 16-query column tiles: no more padding, mostly less
    assert seen["default"][0] == seen["no_r"][0] and seen["default"][1] <= seen["no_r"][1]
    assert seen["no_r"][2] == 0 and (seen["default"][2] > 0 or K < 5)
    assert seen["no_r_xcd_runs"] == seen["no_r"] and seen["xcd_runs"] == seen["default"]
    assert seen["no_r_xcd_runs_chunk3"] == seen["no_r"] and seen["xcd_runs_chunk2"] == seen["default"]


def test_join_work_items_dealt_in_any_chunks_and_runs():
    """hs_join8x_kernel hands its work items to persistent waves in chunks (option join_chunk) from one counter
    or from one counter per XCD in runs of join_xcd_run chunks; a wave stops when it meets an item number past the
    end of the list.  Whatever the chunk size, the run length and the length of the list's last, partial chunk --
    here every tail length comes up: the item count is fixed, the chunk sizes run from 2 to 64 -- the same hits.
    A few hundred thousand items: every wave of the grid takes many chunks, the XCDs run dry at different times
    and take from each other.  (r04: a wave that walked an empty tail shorter than its look-ahead left with two
    valid chunks of another XCD unprocessed.)"""
    k, K, L, W, R, n, nq = 25, 6, 4, 170.0, 42.0, 8_000_003, 80_000
    a, b = synth.make_planes(k, K, L, W, seed=141)
    codes = synth.make_db(n, k, seed=142)
    qcodes, _ = synth.make_query_codes(codes, nq, seed=143)
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    ref, items = None, None
    for resident in (1, 2):          # everything through hs_join8x_kernel | its share beside hs_join8r_kernel's
        eng.set_option("join_resident", resident)
        for xr in (0, 1, 8, 128):
            for g in (0, 2, 3, 5, 7, 12, 17, 31, 48, 64):
                eng.set_option("join_xcd_run", xr)
                eng.set_option("join_chunk", g)
                got = eng.query_codes(qcodes, R, want_cand=False)
                p = eng.profile()
                if ref is None:
                    ref = got
                    assert len(ref["q"]) > 10_000 and p["join_items"] > 100_000
                assert p["join_i8_batches"] == 1
                for f in ("q", "id", "table", "dist"):
                    assert np.array_equal(got[f], ref[f]), (resident, xr, g, f)
    eng.close()


@pytest.mark.parametrize("k,K,L,W,n", [(25, 16, 4, 200.0, 300007), (25, 4, 3, 0.5, 50021), (15, 3, 5, 60.0, 4099),
                                       (25, 20, 3, 160.0, 1), (39, 6, 2, 260.0, 70001), (25, 1, 2, 1.0e6, 9001)])
def test_grouping_by_rank_equals_grouping_by_sort(oracle, tmp_path, k, K, L, W, n):
    """Index build, SURVEY 8(a) a7: the default grouping (hs_group.hip: table of distinct fingerprints,
    ranks, a radix sort of (rank, id) of its own) against the full-width sort of (fingerprint, id) pairs
    (option build_grouping = 1, rounds 1-2): the same index FILE byte for byte -- ids per bucket, directory keys,
    boundaries, tuples -- the same bucket statistics, the oracle's table sizes, and the oracle's hits.
    Shapes: many k-mers per bucket, nearly every k-mer its own bucket (W = 0.5: the table fills up and the
    build falls back to sorting), one k-mer, one bucket per table (W = 10^6), two packed words."""
    a, b = synth.make_planes(k, K, L, W, seed=105)
    codes = synth.make_db(n, k, seed=106)
    centers, _ = synth.make_queries(codes, min(500, 5 * n), seed=107, jitter=0.2)
    R = 30.0 + k
    ix = oracle.Index(a, b, W, oracle.embed_codes(codes))
    want = ix.query(centers, R)
    files = {}
    for form in ("rank", "sort"):
        eng = Engine(k, K, L, W, a, b, options=dict(build_grouping=1) if form == "sort" else None)
        info = eng.index_build(codes)
        assert info["n_buckets"] == ix.table_sizes(), form
        got = eng.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"]), form
        _assert_hits_equal(got, want)
        path = str(tmp_path / (form + ".idx"))
        eng.index_save(path)
        files[form] = (open(path, "rb").read(), info["max_bucket"])
        eng.index_build(codes[: max(1, n // 2)])          # a rebuild on the warm handle (scratch reused)
        assert eng.index_info()["n"] == max(1, n // 2)
        eng.close()
    assert files["rank"][1] == files["sort"][1]
    assert files["rank"][0] == files["sort"][0]


def test_grouping_falls_back_to_the_sort_per_table(oracle, monkeypatch):
    """A table whose fingerprint table fills up (nearly every key distinct) or meets the one fingerprint it
    cannot hold is grouped by the full-width sort instead, table by table.  The library's TEST build
    (HS_TEST_GROUP_FALLBACK) reports that condition for every other table: same index, same hits."""
    k, K, L, W, R, n, nq = 25, 6, 5, 140.0, 45.0, 40009, 800
    a, b = synth.make_planes(k, K, L, W, seed=115)
    codes = synth.make_db(n, k, seed=116)
    centers, _ = synth.make_queries(codes, nq, seed=117, jitter=0.2)
    ix = oracle.Index(a, b, W, oracle.embed_codes(codes))
    want = ix.query(centers, R)
    monkeypatch.setenv("HS_TEST_GROUP_FALLBACK", "1")
    eng = Engine(k, K, L, W, a, b, hooks=True)
    for _ in range(2):
        info = eng.index_build(codes)
        assert info["n_buckets"] == ix.table_sizes()
        got = eng.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
    eng.close()


@pytest.mark.parametrize("world,n,k,K,L,W", [(4, 40009, 25, 6, 5, 140.0), (3, 1001, 15, 3, 4, 60.0), (2, 5, 25, 4, 2, 100.0),
                                             (4, 30011, 25, 2, 3, 3.0)])
def test_index_build_with_the_hashing_spread_over_ranks(oracle, tmp_path, world, n, k, K, L, W):
    """SURVEY 8(e) "Index build" row: every rank hashes only its block of the k-mers,
    the ranks exchange fingerprints and the buckets' tuples, every rank proves the membership of its own
    k-mers (hs_index_shard_*).  Emulated on one GPU: `world` handles, the collectives done here on torch
    tensors.  Every rank's index FILE equals hs_index_build's byte for byte, and its hits the oracle's."""
    import torch
    from hsearch_amd import dist as hdist
    a, b = synth.make_planes(k, K, L, W, seed=125)
    codes = synth.make_db(n, k, seed=126)
    centers, _ = synth.make_queries(codes, min(300, 20 * n), seed=127, jitter=0.2)
    R = 30.0 + k
    plain = Engine(k, K, L, W, a, b)
    plain.index_build(codes)
    ref = str(tmp_path / "plain.idx")
    plain.index_save(ref)
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    _assert_hits_equal(plain.query(centers, R), want)
    plain.close()
    dev = torch.device("cuda", 0)
    engs = [Engine(k, K, L, W, a, b) for _ in range(world)]
    blocks = [e.shard_begin(codes, r, world) for r, e in enumerate(engs)]
    assert [b0 for b0, _ in blocks] == [hdist.shard_bounds(n, r, world)[0] for r in range(world)]
    assert sum(c for _, c in blocks) == n
    for l in range(L):
        fps = []
        for r, e in enumerate(engs):
            t = torch.empty(max(blocks[r][1], 1), dtype=torch.int64, device=dev)
            e.shard_hash(l, 0, t.data_ptr())
            fps.append(t[:blocks[r][1]])
        fp_all = torch.cat(fps).contiguous()                      # the all-gather
        torch.cuda.synchronize()     # (torch's stream is not the library's: its work first, then the pointer)
        nbs = [e.shard_group(l, fp_all.data_ptr()) for e in engs]
        assert len(set(nbs)) == 1
        tups = []
        for e in engs:
            t = torch.zeros(max(nbs[0], 1) * K, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            e.shard_tuples(l, t.data_ptr())
            tups.append(t)
        tup_all = torch.stack(tups).sum(0).to(torch.int32).contiguous()   # the all-reduce
        torch.cuda.synchronize()
        assert all(e.shard_finish(l, tup_all.data_ptr()) == 0 for e in engs)
    for r, e in enumerate(engs):
        info = e.shard_end(0)
        assert info["n"] == n
        path = str(tmp_path / ("rank%d.idx" % r))
        e.index_save(path)
        assert open(path, "rb").read() == open(ref, "rb").read(), r
        got = e.query(centers, R)
        assert np.array_equal(got["cand"], want["cand"])
        _assert_hits_equal(got, want)
        e.close()
