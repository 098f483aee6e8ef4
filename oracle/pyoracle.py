"""TEST INFRASTRUCTURE ONLY -- ctypes bindings of oracle/libhs_oracle.so (this repo's CPU
restatement of the reference hot path) and, when present, oracle/_ref/*.so (the real reference
compiled in the build container by oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing
under hsearch_amd/ does.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_P = C.POINTER
_dp = _P(C.c_double)
_u8p = _P(C.c_uint8)
_u32p = _P(C.c_uint32)
_i32p = _P(C.c_int32)
_u64p = _P(C.c_uint64)


def build(quiet=True):
    """(Re)build libhs_oracle.so and, if /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _ptr(arr, typ):
    return arr.ctypes.data_as(typ)


_lib = None


def lib():
    global _lib
    if _lib is None:
        # HS_ORACLE_LIB: another build of the restatement (the sanitizer build, tests/test_sanitizers_cpu.py)
        path = os.environ.get("HS_ORACLE_LIB") or os.path.join(_HERE, "libhs_oracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.hso_index_build.restype = C.c_void_p
        _lib.hso_index_table_size.restype = C.c_uint64
        _lib.hso_index_query.restype = C.c_uint64
        _lib.hso_bruteforce.restype = C.c_uint64
        _lib.hso_letters_to_codes.restype = C.c_uint64
        _lib.hso_key_string.restype = C.c_uint32
        _lib.hso_evaluate.restype = C.c_double
    return _lib


# ----------------------------------------------------------------------------- restatement ("port")
def embed_codes(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, k = codes.shape
    out = np.empty((n, 8 * k), dtype=np.float64)
    lib().hso_embed_codes(_ptr(codes, _u8p), C.c_uint64(n), C.c_uint32(k), _ptr(out, _dp))
    return out


def letters_to_codes(letters):
    raw = letters.encode() if isinstance(letters, str) else bytes(letters)
    codes = np.empty(len(raw), dtype=np.uint8)
    unknown = lib().hso_letters_to_codes(raw, C.c_uint64(len(raw)), _ptr(codes, _u8p))
    return codes, int(unknown)


def hash_table(a, b, W, pts, want_dots=False):
    """One table: a[K][d], b[K], pts[n][d] -> buckets[n][K] (and dots[n][K])."""
    a, b, pts = _f64(a), _f64(b), _f64(pts)
    K, d = a.shape
    n = pts.shape[0]
    buckets = np.empty((n, K), dtype=np.int32)
    dots = np.empty((n, K), dtype=np.float64) if want_dots else None
    lib().hso_hash(_ptr(a, _dp), _ptr(b, _dp), C.c_uint32(d), C.c_uint32(K), C.c_double(W),
                   _ptr(pts, _dp), C.c_uint64(n), _ptr(dots, _dp) if want_dots else None,
                   _ptr(buckets, _i32p))
    return (buckets, dots) if want_dots else buckets


def hash_all(a, b, W, pts):
    """All tables: a[L][K][d], b[L][K] -> buckets[n][L][K]."""
    a, b = _f64(a), _f64(b)
    L = a.shape[0]
    return np.stack([hash_table(a[l], b[l], W, pts) for l in range(L)], axis=1)


def key_string(buckets):
    buckets = np.ascontiguousarray(buckets, dtype=np.int32)
    buf = C.create_string_buffer(12 * len(buckets) + 1)
    lib().hso_key_string(_ptr(buckets, _i32p), C.c_uint32(len(buckets)), buf, C.c_uint32(len(buf)))
    return buf.value.decode()


class Index:
    """hso_index_build / hso_index_query: the reference Search() split into its two loops."""

    def __init__(self, a, b, W, db):
        a, b, db = _f64(a), _f64(b), _f64(db)
        self.L, self.K, self.d = a.shape
        self.n = db.shape[0]
        self._h = C.c_void_p(lib().hso_index_build(
            _ptr(a, _dp), _ptr(b, _dp), C.c_uint32(self.d), C.c_uint32(self.K), C.c_uint32(self.L),
            C.c_double(W), _ptr(db, _dp), C.c_uint64(self.n)))

    def table_sizes(self):
        return [int(lib().hso_index_table_size(self._h, C.c_uint32(l))) for l in range(self.L)]

    def query(self, centers, R, cap=None, want_cand=True):
        centers = _f64(centers)
        nq = centers.shape[0]
        cap = int(cap) if cap is not None else max(1024, 64 * nq)
        while True:
            hq = np.empty(cap, dtype=np.uint32)
            hid = np.empty(cap, dtype=np.uint32)
            ht = np.empty(cap, dtype=np.uint32)
            hd = np.empty(cap, dtype=np.float64)
            cand = np.zeros((nq, self.L), dtype=np.uint64) if want_cand else None
            n = int(lib().hso_index_query(
                self._h, _ptr(centers, _dp), C.c_uint64(nq), C.c_double(R), _ptr(hq, _u32p),
                _ptr(hid, _u32p), _ptr(ht, _u32p), _ptr(hd, _dp), C.c_uint64(cap),
                _ptr(cand, _u64p) if want_cand else None))
            if n <= cap:
                break
            cap = n
        return dict(q=hq[:n], id=hid[:n], table=ht[:n], dist=hd[:n], cand=cand)

    def query_mt(self, centers, R, threads, cap=None):
        """EXTENSION: Search()'s query loop over `threads` host threads (the reference is single-threaded);
        same hits in the same order.  For bench.py's all-cores CPU figure."""
        centers = _f64(centers)
        nq = centers.shape[0]
        cap = int(cap) if cap is not None else max(1024, 64 * nq)
        f = lib().hso_index_query_mt
        f.restype = C.c_uint64
        while True:
            hq = np.empty(cap, dtype=np.uint32)
            hid = np.empty(cap, dtype=np.uint32)
            ht = np.empty(cap, dtype=np.uint32)
            hd = np.empty(cap, dtype=np.float64)
            n = int(f(self._h, _ptr(centers, _dp), C.c_uint64(nq), C.c_double(R), C.c_uint32(int(threads)),
                      _ptr(hq, _u32p), _ptr(hid, _u32p), _ptr(ht, _u32p), _ptr(hd, _dp), C.c_uint64(cap)))
            if n <= cap:
                break
            cap = n
        return dict(q=hq[:n], id=hid[:n], table=ht[:n], dist=hd[:n])

    def close(self):
        if self._h:
            lib().hso_index_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def search(a, b, W, R, db, centers):
    ix = Index(a, b, W, db)
    try:
        return ix.query(centers, R)
    finally:
        ix.close()


def write_hits(path, hq, hid, hdist):
    hq = np.ascontiguousarray(hq, dtype=np.uint32)
    hid = np.ascontiguousarray(hid, dtype=np.uint32)
    hdist = _f64(hdist)
    rc = lib().hso_write_hits(path.encode(), _ptr(hq, _u32p), _ptr(hid, _u32p), _ptr(hdist, _dp),
                              C.c_uint64(len(hq)), None, None)
    assert rc == 0


def pairwise_square(db, centers):
    db, centers = _f64(db), _f64(centers)
    out = np.empty((centers.shape[0], db.shape[0]), dtype=np.float64)
    lib().hso_pairwise_square(_ptr(db, _dp), C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                              C.c_uint64(centers.shape[0]), C.c_uint32(db.shape[1]), _ptr(out, _dp))
    return out


def bruteforce(db, centers, R, cap=None):
    db, centers = _f64(db), _f64(centers)
    cap = int(cap) if cap is not None else max(1024, 64 * centers.shape[0])
    while True:
        hq = np.empty(cap, dtype=np.uint32)
        hid = np.empty(cap, dtype=np.uint32)
        hd = np.empty(cap, dtype=np.float64)
        n = int(lib().hso_bruteforce(_ptr(db, _dp), C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                                     C.c_uint64(centers.shape[0]), C.c_uint32(db.shape[1]),
                                     C.c_double(R), _ptr(hq, _u32p), _ptr(hid, _u32p),
                                     _ptr(hd, _dp), C.c_uint64(cap)))
        if n <= cap:
            return dict(q=hq[:n], id=hid[:n], dist=hd[:n])
        cap = n


def bruteforce_topk(db, centers, topk):
    db, centers = _f64(db), _f64(centers)
    nq = centers.shape[0]
    nn = np.empty((nq, topk), dtype=np.uint32)
    d2 = np.empty((nq, topk), dtype=np.float64)
    lib().hso_bruteforce_topk(_ptr(db, _dp), C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                              C.c_uint64(nq), C.c_uint32(db.shape[1]), C.c_uint32(topk),
                              _ptr(nn, _u32p), _ptr(d2, _dp))
    return nn, d2


def clustering(a, b, W, R, pts):
    a, b, pts = _f64(a), _f64(b), _f64(pts)
    L, K, d = a.shape
    n = pts.shape[0]
    merged = np.empty(n, dtype=np.uint8)
    owner = np.empty(n, dtype=np.uint32)
    lib().hso_clustering(_ptr(a, _dp), _ptr(b, _dp), C.c_uint32(d), C.c_uint32(K), C.c_uint32(L),
                         C.c_double(W), C.c_double(R), _ptr(pts, _dp), C.c_uint64(n),
                         _ptr(merged, _u8p), _ptr(owner, _u32p))
    return merged, owner


def clustering_to_file(a, b, W, R, pts, path):
    a, b, pts = _f64(a), _f64(b), _f64(pts)
    L, K, d = a.shape
    rc = lib().hso_clustering_to_file(_ptr(a, _dp), _ptr(b, _dp), C.c_uint32(d), C.c_uint32(K),
                                      C.c_uint32(L), C.c_double(W), C.c_double(R), _ptr(pts, _dp),
                                      C.c_uint64(pts.shape[0]), path.encode())
    assert rc == 0


def evaluate(ground_truth_path, hits_path, R):
    return float(lib().hso_evaluate(ground_truth_path.encode(), hits_path.encode(), C.c_double(R)))


# ---- KLSH pre-grouping (pcluster) -- restatement
_REDUCED = {c: k for k, grp in enumerate(["AST", "RKEDQ", "NH", "C", "G", "IVLM", "FYW", "P"]) for c in grp}


def klsh_classes(seq):
    """Reduced-alphabet classes (util.hpp:100-104) of a sequence of the 20 standard letters."""
    return np.array([_REDUCED[c] for c in seq], dtype=np.uint8)


def klsh_draw_planes(feat=512, bits=16, sigma=0.2):
    w = np.empty((bits, feat)); b = np.empty(bits); t = np.empty(bits)
    lib().hso_klsh_draw_planes(C.c_uint32(feat), C.c_uint32(bits), C.c_double(sigma), _ptr(w, _dp),
                               _ptr(b, _dp), _ptr(t, _dp))
    return w, b, t


def klsh_features(classes):
    classes = np.ascontiguousarray(classes, dtype=np.uint8)
    feat = np.empty(512)
    lib().hso_klsh_features(_ptr(classes, _u8p), C.c_uint64(len(classes)), _ptr(feat, _dp))
    return feat


def klsh_hash(w, b, t, feat):
    w, b, t, feat = _f64(w), _f64(b), _f64(t), _f64(feat)
    lib().hso_klsh_hash.restype = C.c_uint64
    return int(lib().hso_klsh_hash(_ptr(w, _dp), _ptr(b, _dp), _ptr(t, _dp), C.c_uint32(w.shape[1]),
                                   C.c_uint32(w.shape[0]), _ptr(feat, _dp)))


# --------------------------------------------------------------- the real reference (oracle/_ref)
_ref_search = None
_ref_hclust2 = None


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_search.so"))


def ref_search_lib():
    global _ref_search
    if _ref_search is None:
        _ref_search = C.CDLL(os.path.join(_HERE, "_ref", "libref_search.so"))
        _ref_search.ref_evaluate.restype = C.c_double
    return _ref_search


def ref_klsh_lib():
    lib_ = C.CDLL(os.path.join(_HERE, "_ref", "libref_klsh.so"))
    lib_.ref_klsh_hash.restype = C.c_uint64
    return lib_


def ref_klsh(feat=512, bits=16, sigma=0.2):
    """The reference's own KLSH object: (planes w, b, t, hash function over feature vectors)."""
    lib_ = ref_klsh_lib()
    lib_.ref_klsh_create(C.c_uint32(feat), C.c_uint32(bits), C.c_double(sigma))
    w = np.empty((bits, feat)); b = np.empty(bits); t = np.empty(bits)
    lib_.ref_klsh_planes(_ptr(w, _dp), _ptr(b, _dp), _ptr(t, _dp))
    return w, b, t, (lambda f: int(lib_.ref_klsh_hash(_ptr(_f64(f), _dp))))


def ref_hclust2_lib():
    global _ref_hclust2
    if _ref_hclust2 is None:
        _ref_hclust2 = C.CDLL(os.path.join(_HERE, "_ref", "libref_hclust2.so"))
    return _ref_hclust2


def ref_constants():
    coords = np.zeros((20, 8))
    dist2 = np.zeros((20, 20))
    base = np.zeros(26, dtype=np.int32)
    ref_search_lib().ref_constants(_ptr(coords, _dp), _ptr(dist2, _dp), _ptr(base, _i32p))
    return coords, dist2, base


def ref_planes(seed, d, K, L, W):
    """Planes as the reference's own LSH constructor (lsh.hpp:10-31) draws them when the l-th
    constructed object is seeded with seed+l: a[L][K][d], b[L][K]."""
    a = np.zeros((L, K, d))
    b = np.zeros((L, K))
    ref_search_lib().ref_lsh_planes(C.c_uint32(seed), C.c_uint32(d), C.c_uint32(K), C.c_double(W),
                                    C.c_uint32(L), _ptr(a, _dp), _ptr(b, _dp))
    return a, b


def ref_hash_table(a, b, W, pts, want_keys=False):
    a, b, pts = _f64(a), _f64(b), _f64(pts)
    K, d = a.shape
    n = pts.shape[0]
    dots = np.empty((n, K))
    buckets = np.empty((n, K), dtype=np.int32)
    stride = 12 * K + 1
    keys = C.create_string_buffer(int(n * stride)) if want_keys else None
    ref_search_lib().ref_hash(_ptr(a, _dp), _ptr(b, _dp), C.c_uint32(d), C.c_uint32(K),
                              C.c_double(W), _ptr(pts, _dp), C.c_uint64(n), _ptr(dots, _dp),
                              _ptr(buckets, _i32p), keys, C.c_uint32(stride))
    if want_keys:
        raw = keys.raw
        ks = [raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() for i in range(n)]
        return buckets, dots, ks
    return buckets, dots


def parse_hits(path):
    """'q id dist' lines -> arrays (names are decimal indices in the harnesses)."""
    q, i, dist = [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) == 3:
                q.append(int(p[0]))
                i.append(int(p[1]))
                dist.append(p[2])
    return (np.array(q, dtype=np.uint32), np.array(i, dtype=np.uint32), dist)


def ref_search(seed, db, centers, K, L, W, R, out_path=None):
    """The reference Search() (motif_both_points.cpp:195-250); returns (q, id, dist_text) in file
    order and leaves the hits file at out_path."""
    db, centers = _f64(db), _f64(centers)
    own = out_path is None
    if own:
        fd, out_path = tempfile.mkstemp(suffix=".hits")
        os.close(fd)
    ref_search_lib().ref_search(C.c_uint32(seed), C.c_uint32(db.shape[1]), _ptr(db, _dp),
                                C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                                C.c_uint64(centers.shape[0]), C.c_uint32(K), C.c_uint32(L),
                                C.c_double(W), C.c_double(R), out_path.encode())
    res = parse_hits(out_path)
    if own:
        os.unlink(out_path)
    return res


def ref_search_timed(seed, db, centers, K, L, W, R):
    """The reference Search() with its two phases timed from outside (ref_search_harness.cpp
    ref_search_timed): (seconds of the build loop, seconds from the end of the build loop to the return of
    Search(), number of hits)."""
    db, centers = _f64(db), _f64(centers)
    fd, out_path = tempfile.mkstemp(suffix=".hits")
    os.close(fd)
    tb, tr = C.c_double(0.0), C.c_double(0.0)
    rc = ref_search_lib().ref_search_timed(C.c_uint32(seed), C.c_uint32(db.shape[1]), _ptr(db, _dp),
                                           C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                                           C.c_uint64(centers.shape[0]), C.c_uint32(K), C.c_uint32(L),
                                           C.c_double(W), C.c_double(R), out_path.encode(), C.byref(tb),
                                           C.byref(tr))
    with open(out_path) as f:
        n_hits = sum(1 for _ in f)
    os.unlink(out_path)
    if rc != 0:
        raise RuntimeError("ref_search_timed: Search() did not flush once per table")
    return tb.value, tr.value, n_hits


def ref_pairwise_square(db, centers):
    db, centers = _f64(db), _f64(centers)
    out = np.empty((centers.shape[0], db.shape[0]))
    ref_search_lib().ref_pairwise_square(C.c_uint32(db.shape[1]), _ptr(db, _dp),
                                         C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                                         C.c_uint64(centers.shape[0]), _ptr(out, _dp))
    return out


def ref_evaluate(ground_truth_path, hits_path, R):
    return float(ref_search_lib().ref_evaluate(ground_truth_path.encode(), hits_path.encode(),
                                               C.c_double(R)))


def ref_kmer_to_coordinates(seqs, k):
    raw = "".join(seqs).encode()
    n = len(seqs)
    out = np.empty((n, 8 * k))
    ref_hclust2_lib().ref2_kmer_to_coordinates(raw, C.c_uint64(n), C.c_uint32(k), _ptr(out, _dp))
    return out


def ref_clustering_file(seed, seqs, k, K, L, W, R, out_path):
    raw = "".join(seqs).encode()
    ref_hclust2_lib().ref2_clustering(C.c_uint32(seed), raw, C.c_uint64(len(seqs)), C.c_uint32(k),
                                      C.c_uint32(K), C.c_uint32(L), C.c_double(W), C.c_double(R),
                                      out_path.encode())


def parse_clusters(path):
    """'#clusterid:<i>:size<m>' + member lines -> list of member-id lists, file order."""
    clusters = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith("#clusterid:"):
                clusters.append([])
            elif line:
                clusters[-1].append(int(line))
    return clusters


# ---- a11 as a program, SURVEY 8(f) row 4 (evaluate2, centroid builder) -- restatement -----------
def _cstrs(strs):
    arr = (C.c_char_p * len(strs))(*[s_.encode() for s_ in strs])
    return arr


def bruteforce_to_files(db, centers, R, out_path, q_names=None, db_names=None):
    """Search() of motif_both_points_noLSH.cpp:36-56: out_path and out_path + 'notlessthan.txt'."""
    db, centers = _f64(db), _f64(centers)
    qn = q_names if q_names is not None else ["c%d" % i for i in range(len(centers))]
    dn = db_names if db_names is not None else ["k%d" % i for i in range(len(db))]
    rc = lib().hso_bruteforce_to_files(_ptr(db, _dp), C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                                       C.c_uint64(centers.shape[0]), C.c_uint32(db.shape[1]),
                                       C.c_double(R), out_path.encode(), _cstrs(qn), _cstrs(dn))
    assert rc == 0


def sort_hits_file(path):
    lib().hso_sort_hits_file.restype = C.c_int64
    return int(lib().hso_sort_hits_file(path.encode()))


def evaluate2_weight(dis):
    lib().hso_evaluate2_weight.restype = C.c_double
    return float(lib().hso_evaluate2_weight(C.c_double(dis)))


def evaluate2(ground_truth_path, hits_path):
    lib().hso_evaluate2.restype = C.c_double
    tp, fn = C.c_double(0), C.c_double(0)
    acc = float(lib().hso_evaluate2(ground_truth_path.encode(), hits_path.encode(), C.byref(tp), C.byref(fn)))
    return acc, tp.value, fn.value


def family_centers(family_codes):
    """family_codes: list of uint8 arrays [members][k] -> centres [n_families][8k]."""
    k = family_codes[0].shape[1]
    first = np.concatenate([[0], np.cumsum([len(f) for f in family_codes])]).astype(np.uint32)
    codes = np.ascontiguousarray(np.concatenate(family_codes), dtype=np.uint8)
    out = np.empty((len(family_codes), 8 * k))
    lib().hso_family_centers(_ptr(codes, _u8p), _ptr(first, _u32p), C.c_uint32(len(family_codes)),
                             C.c_uint32(k), _ptr(out, _dp))
    return out


def write_points_file(path, names, pts):
    pts = _f64(pts)
    rc = lib().hso_write_points_file(path.encode(), _cstrs(names), _ptr(pts, _dp), C.c_uint64(pts.shape[0]),
                                     C.c_uint32(pts.shape[1]))
    assert rc == 0


def center_sampling(db, centers, inner_path, random_path):
    db, centers = _f64(db), _f64(centers)
    rc = lib().hso_center_sampling(_ptr(db, _dp), C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                                   C.c_uint64(centers.shape[0]), C.c_uint32(db.shape[1]),
                                   inner_path.encode(), random_path.encode())
    assert rc == 0


# ---- the real reference programs (oracle/_ref) ---------------------------------------------------
def ref_nolsh_search(db, centers, R, out_path):
    """names k<i> / c<i>; writes out_path and out_path + 'notlessthan.txt'."""
    db, centers = _f64(db), _f64(centers)
    lib_ = C.CDLL(os.path.join(_HERE, "_ref", "libref_nolsh.so"))
    lib_.refn_search(C.c_uint32(db.shape[1]), _ptr(db, _dp), C.c_uint64(db.shape[0]), _ptr(centers, _dp),
                     C.c_uint64(centers.shape[0]), C.c_double(R), out_path.encode())


def ref_evaluate2_lib():
    lib_ = C.CDLL(os.path.join(_HERE, "_ref", "libref_evaluate2.so"))
    lib_.refe_weight.restype = C.c_double
    return lib_


def ref_evaluate2_sort(path):
    return int(ref_evaluate2_lib().refe_sort(path.encode()))


def ref_evaluate2_weight(dis):
    return float(ref_evaluate2_lib().refe_weight(C.c_double(dis)))


def ref_centers_main(workdir, families_path, points_path, k, out_name):
    lib_ = C.CDLL(os.path.join(_HERE, "_ref", "libref_centers.so"))
    return int(lib_.refc_main(workdir.encode(), families_path.encode(), points_path.encode(),
                              C.c_uint32(k), out_name.encode()))


def ref_cluster2datapoint(k, names, family_seqs, out_prefix):
    lib_ = C.CDLL(os.path.join(_HERE, "_ref", "libref_centers.so"))
    first = np.concatenate([[0], np.cumsum([len(f) for f in family_seqs])]).astype(np.uint32)
    flat = [s_ for f in family_seqs for s_ in f]
    return int(lib_.refc_cluster2datapoint(C.c_uint32(k), C.c_uint32(len(names)), _cstrs(names),
                                           _ptr(first, _u32p), _cstrs(flat), out_prefix.encode()))


def ref_protein2datapoints(fasta_path, k, num_out, out_path, seed):
    """The real `protein2datapoints` main() with rand() seeded by `seed` (oracle/_ref/libref_p2d.so)."""
    lib_ = C.CDLL(os.path.join(_HERE, "_ref", "libref_p2d.so"))
    return int(lib_.refp_main(fasta_path.encode(), C.c_uint32(k), C.c_uint32(num_out), out_path.encode(),
                              C.c_uint32(seed)))


def ref_hclust3_clustering_file(seed, seqs, k, K, L, W, R, out_path):
    """hclust3.cpp's Clustering() (the lazy-embedding twin of hclust2's), oracle/_ref/libref_hclust3.so."""
    lib_ = C.CDLL(os.path.join(_HERE, "_ref", "libref_hclust3.so"))
    raw = "".join(seqs).encode()
    lib_.ref2_clustering(C.c_uint32(seed), raw, C.c_uint64(len(seqs)), C.c_uint32(k), C.c_uint32(K),
                         C.c_uint32(L), C.c_double(W), C.c_double(R), out_path.encode())
