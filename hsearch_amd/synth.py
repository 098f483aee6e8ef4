"""Synthetic workloads of SURVEY.md 8(d): i.i.d. uniform k-mers over the 20-letter alphabet,
queries = DB k-mers with 0..4 random substitutions embedded exactly from the table, Gaussian planes
a ~ N(0,1) and offsets b ~ U[0,W).  Seeds: DB 2026, queries 2027, planes 2028.  Host-side input
generation only (numpy); nothing here is on the measured path."""
import numpy as np

SEED_DB, SEED_QUERIES, SEED_PLANES = 2026, 2027, 2028

# include/hs_tables.h HS_AA_COORDS (reference util.hpp:21-42), loaded lazily from the header so the
# digits live in exactly one place.
_COORDS = None


def coords():
    global _COORDS
    if _COORDS is None:
        import os
        import re
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "hs_tables.h")
        text = open(path).read()
        body = text[text.index("HS_AA_COORDS[HS_ALPHABET][HS_AA_DIM] = {"):]
        body = body[:body.index("};")]
        rows = re.findall(r"\{([^{}]*)\}", body)
        _COORDS = np.array([[float(v) for v in r.split(",")] for r in rows], dtype=np.float64)
        assert _COORDS.shape == (20, 8)
    return _COORDS


def _rng(seed):
    return np.random.Generator(np.random.MT19937(seed))


def make_db(n, k, seed=SEED_DB):
    return _rng(seed).integers(0, 20, size=(n, k), dtype=np.uint8)


def make_planes(k, K, L, W, seed=SEED_PLANES):
    rng = _rng(seed)
    a = rng.standard_normal((L, K, 8 * k))
    b = rng.uniform(0.0, W, size=(L, K))
    return a, b


def make_query_codes(db_codes, nq, max_subst=4, seed=SEED_QUERIES):
    rng = _rng(seed)
    n, k = db_codes.shape
    src = rng.integers(0, n, size=nq)
    q = db_codes[src].copy()
    m = rng.integers(0, max_subst + 1, size=nq)
    for s in range(max_subst):
        rows = np.nonzero(m > s)[0]
        pos = rng.integers(0, k, size=len(rows))
        q[rows, pos] = rng.integers(0, 20, size=len(rows), dtype=np.uint8)
    return q, src


def embed(codes):
    """numpy restatement of the table lookup (input generation for query points only)."""
    codes = np.asarray(codes)
    return coords()[codes].reshape(codes.shape[0], -1)


def make_queries(db_codes, nq, max_subst=4, seed=SEED_QUERIES, jitter=0.0):
    """Query points [nq][8k] (float64) and the DB id each was derived from."""
    qc, src = make_query_codes(db_codes, nq, max_subst, seed)
    pts = embed(qc)
    if jitter:
        pts = pts + _rng(seed + 1).normal(0.0, jitter, size=pts.shape)
    return np.ascontiguousarray(pts), src
