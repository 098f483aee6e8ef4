"""Index files on the host (no GPU): hs_index_file_check accepts a well-formed file and names the
fault of a truncated, bit-flipped, or self-consistent-but-rule-breaking one -- the same rules
hs_index_load enforces on the device before any kernel indexes with a table."""
import numpy as np
import pytest

import hsearch_amd
from hsearch_amd import HsError, synth

import indexfile


@pytest.fixture(scope="module")
def small(oracle):
    k, K, L, W, n = 9, 3, 3, 12.0, 400
    a, b = synth.make_planes(k, K, L, W, seed=3)
    codes = synth.make_db(n, k, seed=4)
    buckets = oracle.hash_all(a, b, W, oracle.embed_codes(codes))
    return dict(k=k, K=K, L=L, W=W, a=a, b=b, codes=codes, tables=indexfile.build_tables(buckets))


def _write(path, c, **kw):
    indexfile.write(path, c["k"], c["K"], c["L"], c["W"], c["a"], c["b"], c["codes"], c["tables"], **kw)


def test_well_formed_file_passes(small, tmp_path):
    p = tmp_path / "ok.hsidx"
    _write(p, small)
    hsearch_amd.index_file_check(p)
    # tiny-W keys alias as strings ((1,23) vs (12,3)): aliased tuples share a bucket and a fingerprint
    assert sum(len(t[1]) for t in small["tables"]) > 50


def test_truncated_and_flipped_files_are_refused(small, tmp_path):
    p = tmp_path / "ok.hsidx"
    _write(p, small)
    data = p.read_bytes()
    for cut in (0, 7, 100, len(data) // 2, len(data) - 1):
        q = tmp_path / "cut.hsidx"
        q.write_bytes(data[:cut])
        with pytest.raises(HsError) as e:
            hsearch_amd.index_file_check(q)
        assert e.value.status == hsearch_amd.capi.HS_ERR_IO
    q = tmp_path / "long.hsidx"
    q.write_bytes(data + b"\0")
    with pytest.raises(HsError):
        hsearch_amd.index_file_check(q)
    rng = np.random.default_rng(1)
    for pos in rng.integers(0, len(data), size=40).tolist() + [8, len(data) - 1]:
        flipped = bytearray(data)
        flipped[pos] ^= 0x10
        q = tmp_path / "flip.hsidx"
        q.write_bytes(bytes(flipped))
        with pytest.raises(HsError) as e:
            hsearch_amd.index_file_check(q)
        assert e.value.status == hsearch_amd.capi.HS_ERR_IO, pos


def _ids(sections, l=0):
    return sections[4 + 4 * l]


@pytest.mark.parametrize("what,tamper", [
    ("id out of range", lambda s: _ids(s).__setitem__(5, 10**9)),
    ("id listed twice", lambda s: _ids(s, 1).__setitem__(5, _ids(s, 1)[6])),
    ("boundaries", lambda s: s[4 + 2].__setitem__(0, 1)),
    ("boundaries", lambda s: s[4 + 2].__setitem__(3, s[4 + 2][2])),
    ("boundaries", lambda s: s[4 + 2].__setitem__(len(s[4 + 2]) - 1, 7)),
    ("fingerprints not ascending", lambda s: s[4 + 1].__setitem__(slice(0, 2), s[4 + 1][[1, 0]])),
    ("tuple does not have its fingerprint", lambda s: s[4 + 3].__setitem__((0, 0), s[4 + 3][0, 0] + 1)),
    ("residue code", lambda s: s[3].__setitem__((0, 0), 21)),
])
def test_self_consistent_but_rule_breaking_files_are_refused(small, tmp_path, what, tamper):
    """The payload hash is recomputed after the edit, so only the content rules can catch these."""
    p = tmp_path / "bad.hsidx"
    _write(p, small, tamper=tamper)
    with pytest.raises(HsError) as e:
        hsearch_amd.index_file_check(p)
    assert e.value.status == hsearch_amd.capi.HS_ERR_IO
    assert what in str(e.value), str(e.value)


def test_ids_must_ascend_inside_a_bucket(small, tmp_path):
    def swap(sections):
        start = sections[4 + 2]
        b = int(np.argmax(np.diff(start.astype(np.int64))))       # a bucket with >= 2 members
        lo = int(start[b])
        assert start[b + 1] - lo >= 2
        ids = _ids(sections)
        ids[lo], ids[lo + 1] = ids[lo + 1], ids[lo]
    p = tmp_path / "bad.hsidx"
    _write(p, small, tamper=swap)
    with pytest.raises(HsError) as e:
        hsearch_amd.index_file_check(p)
    assert "ascending inside a bucket" in str(e.value)
