"""The LSH projection on the matrix cores (hs_proj.hip; SURVEY 7 hard part 1, north_star "MFMA ...
for the batched points x hyperplanes projection"): int8 MFMA pass with a proven error bound, values
within the bound of a bucket boundary recomputed in the reference's fp64 order.  The bucket
integers must equal the oracle's (lsh.hpp:33-49) bit for bit in every mode, for k-mers given as
codes and for arbitrary points, including values placed ON bucket boundaries."""
import numpy as np
import pytest

from hsearch_amd import Engine, synth

pytestmark = pytest.mark.gpu


def _check(eng, oracle, a, b, W, codes, pts, extra_pts=None):
    want = oracle.hash_all(a, b, W, pts)
    assert np.array_equal(eng.hash_codes(codes), want)
    flagged = eng.profile()["hash_flagged"], eng.profile()["hash_values"]
    assert np.array_equal(eng.hash_points(pts), want)
    if extra_pts is not None:
        assert np.array_equal(eng.hash_points(extra_pts), oracle.hash_all(a, b, W, extra_pts))
    return flagged


@pytest.mark.parametrize("k,K,L,W", [(25, 16, 8, 200.0), (25, 20, 32, 200.0), (15, 6, 5, 60.0), (39, 8, 4, 150.0),
                                     (50, 5, 3, 300.0), (7, 3, 2, 40.0), (25, 7, 3, 9.0), (60, 4, 2, 400.0)])
def test_bucket_ints_identical_in_every_hash_mode(oracle, k, K, L, W):
    a, b = synth.make_planes(k, K, L, W, seed=7)
    codes = synth.make_db(6000, k, seed=8)
    pts = oracle.embed_codes(codes)
    rng = np.random.default_rng(9)
    jit = pts[:1500] + rng.normal(0.0, 0.7, size=(1500, 8 * k))
    eng = Engine(k, K, L, W, a, b)
    seen = {}
    for mode in ("exact", "mfma", "auto"):
        eng.set_hash_mode(mode)
        seen[mode] = _check(eng, oracle, a, b, W, codes, pts, jit)
    assert seen["exact"] == (0, 0)                      # the vector-ALU kernel alone
    if k <= 52:
        fl, vals = seen["mfma"]
        assert vals == 6000 * K * L                     # the MFMA pass produced every value ...
        if W >= 40.0:
            assert 0 < fl < 0.02 * vals                 # ... and only a sliver needed the exact pass
    else:
        assert seen["mfma"] == (0, 0)                   # k > 52: no MFMA variant, exact kernel
    # an inflated bound sends (many) more values through the exact pass: same integers
    eng.set_hash_mode("mfma", eps_scale=200.0)
    fl2, vals2 = _check(eng, oracle, a, b, W, codes, pts, jit)
    if k <= 52:
        assert fl2 > 10 * max(seen["mfma"][0], 1) or fl2 > 0.2 * vals2
    eng.set_hash_mode("mfma", eps_scale=1e9)            # everything flagged: the list overflows,
    _check(eng, oracle, a, b, W, codes[:700], pts[:700])   # the fix kernel recomputes all
    eng.close()


def test_values_on_bucket_boundaries(oracle):
    """b chosen so that (dot + b) / W of many (point, function) pairs sits within a few ulps of an
    integer, on either side: the fast pass cannot decide those, the bound must catch every one."""
    k, K, L, W = 25, 8, 4, 200.0
    rng = np.random.default_rng(21)
    a, b = synth.make_planes(k, K, L, W, seed=22)
    codes = synth.make_db(4000, k, seed=23)
    pts = oracle.embed_codes(codes)
    dots = np.einsum("nd,lkd->nlk", pts[:L * K], a)          # one boundary point per function
    for f in range(L * K):
        l, kk = divmod(f, K)
        m = float(rng.integers(-3, 4))
        b[l, kk] = np.nextafter(m * W - dots[f, l, kk], rng.choice([-np.inf, np.inf]))
    eng = Engine(k, K, L, W, a, b)
    for mode in ("mfma", "auto", "exact"):
        eng.set_hash_mode(mode)
        _check(eng, oracle, a, b, W, codes, pts)
    # ... and the index built from such planes equals the oracle's
    info = eng.index_build(codes)
    ix = oracle.Index(a, b, W, pts)
    assert info["n_buckets"] == ix.table_sizes()
    ix.close()
    eng.close()


def test_unrepresentable_inputs_take_the_exact_path(oracle):
    k, K, L, W = 25, 6, 3, 120.0
    a, b = synth.make_planes(k, K, L, W, seed=31)
    codes = synth.make_db(500, k, seed=32)
    pts = oracle.embed_codes(codes)
    rng = np.random.default_rng(33)
    wild = pts.copy()
    wild[:50] *= 1e6                       # fine: per-point scale
    wild[50:100] *= 1e-9
    wild[100:120] = 0.0
    wild[120:140, ::7] = 1e250             # beyond the fixed-point range: flagged, recomputed
    wild[140:160] += rng.normal(0, 1e-12, size=(20, 8 * k))
    eng = Engine(k, K, L, W, a, b)
    eng.set_hash_mode("mfma")
    want = oracle.hash_all(a, b, W, wild)
    got = eng.hash_points(wild)
    ok = np.abs(want.astype(np.float64)) < 2.0**31 - 1     # int(floor(.)) of such values is UB in the reference
    assert np.array_equal(got[ok], want[ok])
    # a plane with a huge coefficient / offset: that function is always recomputed
    a2, b2 = a.copy(), b.copy()
    a2[1, 2, 5] = 1e240
    b2[0, 1] = 1e220
    e2 = Engine(k, K, L, W, a2, b2)
    e2.set_hash_mode("mfma")
    want = oracle.hash_all(a2, b2, W, pts)
    got = e2.hash_codes(codes)
    ok = np.abs(want.astype(np.float64)) < 2.0**31 - 1
    assert np.array_equal(got[ok], want[ok])
    # a custom coordinate table (the points-file route) with large entries
    table = rng.normal(0.0, 300.0, size=(27, 8))
    c3 = rng.integers(0, 27, size=(800, k), dtype=np.uint8)
    e3 = Engine(k, K, L, W, a, b, coords=table)
    e3.set_hash_mode("mfma")
    p3 = table[c3].reshape(800, -1)
    assert np.array_equal(e3.hash_codes(c3), oracle.hash_all(a, b, W, p3))
    for e in (eng, e2, e3):
        e.close()


def test_projection_runs_in_build_and_query(oracle):
    """The index build and the query hash go through the MFMA pass (profile counters), and the
    results equal the exact-mode results and the oracle."""
    k, K, L, W, R = 25, 16, 8, 200.0, 40.0
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(40000, k)
    centers, _ = synth.make_queries(codes, 1500, jitter=0.2)
    res = {}
    for mode in ("exact", "auto"):
        eng = Engine(k, K, L, W, a, b)
        eng.set_hash_mode(mode)
        info = eng.index_build(codes)
        pb = eng.profile()
        got = eng.query(centers, R)
        pq = eng.profile()
        res[mode] = (info["n_buckets"], got)
        if mode == "auto":
            assert pb["hash_values"] == 40000 * K * L and 0 < pb["hash_flagged"] < 0.01 * pb["hash_values"]
            assert pq["hash_values"] == 1500 * K * L
        else:
            assert pb["hash_values"] == 0 and pq["hash_values"] == 0
        eng.close()
    assert res["exact"][0] == res["auto"][0]
    for f in ("q", "id", "table", "dist", "cand"):
        assert np.array_equal(res["exact"][1][f], res["auto"][1][f])
    want = oracle.search(a, b, W, R, oracle.embed_codes(codes), centers)
    for f in ("q", "id", "table", "dist"):
        assert np.array_equal(res["auto"][1][f], want[f])
