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
                                            (25, 2, 3, 3.0, 40.0, 4000, 200)])
def test_search_hits_match_oracle(oracle, k, K, L, W, R, n, nq):
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(n, k)
    centers, _ = synth.make_queries(codes, nq, jitter=0.25)
    eng = Engine(k, K, L, W, a, b)
    info = eng.index_build(codes)
    ix = oracle.Index(a, b, W, oracle.embed_codes(codes))
    assert info["n_buckets"] == ix.table_sizes()
    want = ix.query(centers, R)
    got = eng.query(centers, R)
    assert np.array_equal(got["cand"], want["cand"])
    _assert_hits_equal(got, want)
    assert len(want["q"]) > 0
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
