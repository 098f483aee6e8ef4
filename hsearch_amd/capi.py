"""ctypes binding of include/hsearch.h.  No compute here; everything runs in libhsearch_amd.so."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

HS_OK, HS_ERR_INVALID, HS_ERR_NO_DEVICE, HS_ERR_HIP, HS_ERR_CAPACITY, HS_ERR_STATE, \
    HS_ERR_KEY_COLLISION, HS_ERR_NOMEM, HS_ERR_IO, HS_ERR_PEER = range(10)
_STATUS = ["HS_OK", "HS_ERR_INVALID", "HS_ERR_NO_DEVICE", "HS_ERR_HIP", "HS_ERR_CAPACITY",
           "HS_ERR_STATE", "HS_ERR_KEY_COLLISION", "HS_ERR_NOMEM", "HS_ERR_IO", "HS_ERR_PEER"]

# Row order of the embedding table (include/hs_tables.h HS_CODE_TO_LETTER): BLOSUM order.
_ALPHABET = "ARNDCQEGHILKMFPSTWYV"

EXPORTS = ["hs_create", "hs_destroy", "hs_last_error", "hs_get_profile", "hs_get_params", "hs_version",
           "hs_set_verify_mode", "hs_set_hash_mode", "hs_set_option", "hs_set_bucket_partition", "hs_wait_event", "hs_set_planes", "hs_self_join", "hs_self_join_range", "hs_clustering",
           "hs_clustering_begin", "hs_clustering_table_edges", "hs_clustering_table_apply",
           "hs_clustering_end",
           "hs_embed_codes", "hs_hash_codes", "hs_hash_points", "hs_key_string", "hs_key_fingerprint",
           "hs_key_strings_equal", "hs_index_build", "hs_index_build_subset", "hs_index_build_windows", "hs_index_shard_begin", "hs_index_shard_hash_dev", "hs_index_shard_group_dev",
           "hs_index_shard_tuples_dev", "hs_index_shard_finish_dev", "hs_index_shard_end", "hs_index_save", "hs_index_load", "hs_index_file_check", "hs_klsh_draw_planes", "hs_klsh_codes",
           "hs_index_info_get", "hs_query", "hs_query_dev", "hs_query_codes", "hs_query_codes_dev", "hs_bruteforce",
           "hs_bruteforce_topk", "hs_merge_first_table_dev"]


class HsError(RuntimeError):
    def __init__(self, status, message):
        self.status = status
        name = _STATUS[status] if 0 <= status < len(_STATUS) else str(status)
        super().__init__("%s: %s" % (name, message))


class _Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("K", C.c_uint32), ("L", C.c_uint32), ("W", C.c_double),
                ("device", C.c_int32), ("alphabet", C.c_uint32)]


class _Profile(C.Structure):
    _fields_ = [("ms_hash", C.c_double), ("ms_sort", C.c_double), ("ms_gather", C.c_double),
                ("ms_probe", C.c_double), ("ms_verify", C.c_double), ("ms_finalize", C.c_double),
                ("ms_total", C.c_double), ("candidates", C.c_uint64), ("provisional", C.c_uint64),
                ("hits", C.c_uint64), ("verify_launches", C.c_uint64), ("join_batches", C.c_uint64),
                ("ms_join", C.c_double), ("join_items", C.c_uint64), ("join_pairs", C.c_uint64),
                ("join_pairs_issued", C.c_uint64), ("join_i8_batches", C.c_uint64),
                ("hash_values", C.c_uint64), ("hash_flagged", C.c_uint64),
                ("join_row_bytes", C.c_uint32), ("join_wide", C.c_uint32), ("join_items_resident", C.c_uint64),
                ("join_async_retries", C.c_uint64), ("queries_recognised", C.c_uint64)]


class _IndexInfo(C.Structure):
    _fields_ = [("n", C.c_uint64), ("device_bytes", C.c_uint64), ("n_buckets", C.c_uint64 * 32),
                ("max_bucket", C.c_uint64 * 32), ("key_seed", C.c_uint32)]


profile_fields = [f[0] for f in _Profile._fields_]

_libs = {}


def lib_path(hooks=False):
    # HSEARCH_AMD_LIB: another build of the same library (A/B runs of two kernel versions on one box)
    if hooks:
        return os.path.join(_HERE, "libhsearch_amd_hooks.so")
    return os.environ.get("HSEARCH_AMD_LIB") or os.path.join(_HERE, "libhsearch_amd.so")


def load(hooks=False):
    """Load libhsearch_amd.so.  Raises (never falls back) when it has not been built.
    hooks=True: the TEST build of the same library (libhsearch_amd_hooks.so: the same kernel objects
    under a C-ABI layer compiled with -DHS_TEST_HOOKS = fault injection), for the tests that need it."""
    if hooks not in _libs:
        # One ROCm runtime per process: PyTorch ships its own libamdhip64 / libhsa-runtime64 / librccl.
        # If this library (linked against /opt/rocm's) is loaded BEFORE torch, the process ends up with
        # /opt/rocm's HIP + HSA and, once torch is imported, torch's RCCL, whose dlopen of
        # "libhsa-runtime64.so" then maps a second, uninitialised HSA copy (ncclCommInitAll: "no
        # ROCm-capable device").  Importing torch first makes every later load resolve to its copies.
        # (The C++ programs under hsearch_amd/host have no torch and use /opt/rocm's throughout.)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        path = lib_path(hooks)
        if not os.path.exists(path):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (or make -C hsearch_amd/csrc)" % path)
        lib = C.CDLL(path)
        lib.hs_version.restype = C.c_char_p
        lib.hs_last_error.restype = C.c_char_p
        lib.hs_last_error.argtypes = [C.c_void_p]
        lib.hs_destroy.restype = None
        lib.hs_destroy.argtypes = [C.c_void_p]
        lib.hs_key_string.restype = C.c_uint32
        lib.hs_key_fingerprint.restype = C.c_uint64
        lib.hs_key_strings_equal.restype = C.c_int
        _libs[hooks] = lib
    return _libs[hooks]


def alphabet():
    return _ALPHABET


def codes_from_letters(seqs):
    """['ARND...', ...] (equal lengths, letters of the 20-letter alphabet) -> uint8 [n][k]."""
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(_ALPHABET):
        lut[ord(ch)] = i
    arr = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    codes = lut[arr].reshape(len(seqs), -1)
    if (codes == 255).any():
        raise ValueError("letter outside ARNDCQEGHILKMFPSTWYV")
    return codes


def key_string(buckets):
    b = np.ascontiguousarray(buckets, dtype=np.int32)
    buf = C.create_string_buffer(12 * len(b) + 1)
    load().hs_key_string(b.ctypes.data_as(C.c_void_p), C.c_uint32(len(b)), buf, C.c_uint32(len(buf)))
    return buf.value.decode()


def key_fingerprint(buckets, seed=0):
    b = np.ascontiguousarray(buckets, dtype=np.int32)
    return int(load().hs_key_fingerprint(b.ctypes.data_as(C.c_void_p), C.c_uint32(len(b)),
                                         C.c_uint32(seed)))


def key_strings_equal(x, y):
    x = np.ascontiguousarray(x, dtype=np.int32)
    y = np.ascontiguousarray(y, dtype=np.int32)
    assert len(x) == len(y)
    return bool(load().hs_key_strings_equal(x.ctypes.data_as(C.c_void_p),
                                            y.ctypes.data_as(C.c_void_p), C.c_uint32(len(x))))


def _vp(arr):
    return arr.ctypes.data_as(C.c_void_p)


def index_file_check(path):
    """Host-only check of an index file (hs_index_file_check): header, payload length + hash and the
    content rules hs_index_load enforces on the device.  Raises HsError(HS_ERR_IO) naming the fault."""
    err = C.create_string_buffer(512)
    st = load().hs_index_file_check(str(path).encode(), err, C.c_uint32(len(err)))
    if st != HS_OK:
        raise HsError(st, err.value.decode())


KLSH_NONE = 0xffffffffffffffff
_REDUCED_CLASS = {c: k for k, grp in enumerate(["AST", "RKEDQ", "NH", "C", "G", "IVLM", "FYW", "P"])
                  for c in grp}   # pcluster util.hpp:100-104 (include/hs_tables.h HS_REDUCED_CLASS)


def klsh_draw_planes(feat=512, bits=16, sigma=0.2):
    """The planes KLSH::KLSH draws from its default-seeded engine (lsh.cpp:17-38): (w, b, t)."""
    w = np.empty((bits, feat)); b = np.empty(bits); t = np.empty(bits)
    st = load().hs_klsh_draw_planes(C.c_uint32(feat), C.c_uint32(bits), C.c_double(sigma), _vp(w),
                                    _vp(b), _vp(t))
    if st != HS_OK:
        raise HsError(st, "hs_klsh_draw_planes")
    return w, b, t


def klsh_codes(classes, seq_start, w, b, t, device=0):
    """KLSH code of every sequence of a concatenated buffer of reduced-alphabet
    classes (pcluster.cpp:11-33 + lsh.cpp:40-49) on the GPU.  Returns (codes uint64 [n_seq] with
    KLSH_NONE for sequences shorter than 3, uncertain-bit masks uint64 [n_seq])."""
    classes = np.ascontiguousarray(classes, dtype=np.uint8)
    seq_start = np.ascontiguousarray(seq_start, dtype=np.uint64)
    w = np.ascontiguousarray(w, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    n_seq = len(seq_start) - 1
    codes = np.empty(n_seq, dtype=np.uint64); unc = np.empty(n_seq, dtype=np.uint64)
    err = C.create_string_buffer(256)
    st = load().hs_klsh_codes(C.c_int(device), _vp(classes), C.c_uint64(len(classes)), _vp(seq_start),
                              C.c_uint64(n_seq), _vp(w), _vp(b), _vp(t), C.c_uint32(w.shape[0]),
                              _vp(codes), _vp(unc), err, C.c_uint32(256))
    if st != HS_OK:
        raise HsError(st, err.value.decode())
    return codes, unc


class Engine:
    """One handle = one GPU + one index.  Mirrors the reference's operator surface:
    LSH(dim, K, W) x L  ->  Engine(k, K, L, W, a, b);  HashBucketIndex -> hash_points/hash_codes;
    Search() build loop -> index_build; Search() query loop -> query; noLSH Search() -> bruteforce.
    """

    def __init__(self, k, K, L, W, a, b, device=0, coords=None, hooks=False, options=None):
        self._lib = load(hooks)
        self.k, self.K, self.L, self.W = int(k), int(K), int(L), float(W)
        self.d = 8 * self.k
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert a.shape == (self.L, self.K, self.d), a.shape
        assert b.shape == (self.L, self.K), b.shape
        cptr = None
        if coords is not None:
            coords = np.ascontiguousarray(coords, dtype=np.float64)
            assert coords.ndim == 2 and coords.shape[1] == 8 and 1 <= coords.shape[0] <= 32
            cptr = _vp(coords)
        params = _Params(self.k, self.K, self.L, self.W, int(device),
                         0 if coords is None else coords.shape[0])
        self._h = C.c_void_p()
        st = self._lib.hs_create(C.byref(params), _vp(a), _vp(b), cptr, C.byref(self._h))
        if st != HS_OK:
            msg = self._lib.hs_last_error(self._h).decode() if self._h else ""
            if self._h:
                self._lib.hs_destroy(self._h)
                self._h = C.c_void_p()
            raise HsError(st, msg or "hs_create failed (is a gfx950 GPU visible?)")
        for name, value in (options or {}).items():      # hs_set_option, before any index exists
            self.set_option(name, value)

    # -- helpers
    def _check(self, st):
        if st != HS_OK:
            raise HsError(st, self._lib.hs_last_error(self._h).decode())

    def set_verify_mode(self, mode):
        """'auto' | 'stream' | 'join' -- which filter kernel runs in front of the exact decision."""
        self._check(self._lib.hs_set_verify_mode(self._h, {"auto": 0, "stream": 1, "join": 2, "join16": 3}[mode]))

    def set_hash_mode(self, mode, eps_scale=1.0):
        """'auto' | 'exact' | 'mfma' -- how the bucket ints are evaluated (identical results)."""
        self._check(self._lib.hs_set_hash_mode(self._h, {"auto": 0, "exact": 1, "mfma": 2}[mode],
                                               C.c_double(eps_scale)))

    OPTIONS = {"query_batch": 1, "seg_mode": 2, "join_resident": 3, "recognise_kmers": 4, "build_grouping": 5,
               "wide_rows": 6, "refine8": 7, "self_codes": 8, "sort_hits": 10, "sync_items": 11,
               "join_min_q": 12, "join_min_m": 13, "sort_from_bit": 14, "build_serial": 15,
               "join_xcd_run": 16, "probe_records": 17, "join_chunk": 18}

    def set_option(self, name, value):
        """hs_set_option (include/hsearch.h hs_option): path selection / batch sizing; never changes a result."""
        self._check(self._lib.hs_set_option(self._h, C.c_int(self.OPTIONS[name]), C.c_int64(int(value))))

    def set_bucket_partition(self, part, n_parts):
        """hs_set_bucket_partition: the handle's searches probe only the buckets of `part` of `n_parts` (1: all)."""
        self._check(self._lib.hs_set_bucket_partition(self._h, C.c_uint32(part), C.c_uint32(n_parts)))

    def wait_event(self, event_handle):
        """hs_wait_event: the library's stream waits for a hipEvent_t (int handle, e.g. torch.cuda.Event.cuda_event)."""
        self._check(self._lib.hs_wait_event(self._h, C.c_void_p(event_handle)))

    def set_planes(self, a, b):
        """A new hash family of the same shape for this handle (hs_set_planes); drops the index."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert a.shape == (self.L, self.K, self.d) and b.shape == (self.L, self.K)
        self._check(self._lib.hs_set_planes(self._h, _vp(a), _vp(b)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def profile(self):
        p = _Profile()
        self._check(self._lib.hs_get_profile(self._h, C.byref(p)))
        return {f: getattr(p, f) for f in profile_fields}

    def index_info(self):
        info = _IndexInfo()
        self._check(self._lib.hs_index_info_get(self._h, C.byref(info)))
        return dict(n=info.n, device_bytes=info.device_bytes,
                    n_buckets=list(info.n_buckets)[:self.L], max_bucket=list(info.max_bucket)[:self.L],
                    key_seed=info.key_seed)

    # -- a2 / a4 / a5
    def embed_codes(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n = codes.shape[0]
        out = np.empty((n, self.d), dtype=np.float64)
        self._check(self._lib.hs_embed_codes(self._h, _vp(codes), C.c_uint64(n), _vp(out)))
        return out

    def hash_codes(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n = codes.shape[0]
        assert codes.shape == (n, self.k)
        out = np.empty((n, self.L, self.K), dtype=np.int32)
        self._check(self._lib.hs_hash_codes(self._h, _vp(codes), C.c_uint64(n), _vp(out)))
        return out

    def hash_points(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        n = pts.shape[0]
        assert pts.shape == (n, self.d)
        out = np.empty((n, self.L, self.K), dtype=np.int32)
        self._check(self._lib.hs_hash_points(self._h, _vp(pts), C.c_uint64(n), _vp(out)))
        return out

    # -- a7
    def index_build(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n = codes.shape[0]
        assert codes.ndim == 2 and codes.shape[1] == self.k
        self._check(self._lib.hs_index_build(self._h, _vp(codes), C.c_uint64(n)))
        return self.index_info()

    def index_build_subset(self, codes_all, subset):
        """Index over rows `subset` (uint32, or None for all) of a code array kept on the device across
        calls (hs_index_build_subset); the array must stay alive and unchanged between calls."""
        assert codes_all.dtype == np.uint8 and codes_all.flags["C_CONTIGUOUS"] and codes_all.shape[1] == self.k
        n_all = codes_all.shape[0]
        sub = None if subset is None else np.ascontiguousarray(subset, dtype=np.uint32)
        n_sub = n_all if sub is None else len(sub)
        self._check(self._lib.hs_index_build_subset(self._h, _vp(codes_all), C.c_uint64(n_all),
                                                    _vp(sub) if sub is not None else C.c_void_p(0),
                                                    C.c_uint64(n_sub)))
        return self.index_info()

    # -- a7 with the hashing spread over ranks (hs_index_shard_*: pointers are device pointers, ints)
    def shard_begin(self, codes, rank, world):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        assert codes.ndim == 2 and codes.shape[1] == self.k
        lo, cnt = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.hs_index_shard_begin(self._h, _vp(codes), C.c_uint64(codes.shape[0]), C.c_uint32(rank),
                                                   C.c_uint32(world), C.byref(lo), C.byref(cnt)))
        return lo.value, cnt.value

    def shard_hash(self, l, seed, d_fp_block):
        self._check(self._lib.hs_index_shard_hash_dev(self._h, C.c_uint32(l), C.c_uint32(seed), C.c_void_p(d_fp_block)))

    def shard_group(self, l, d_fp_all):
        nb = C.c_uint32(0)
        self._check(self._lib.hs_index_shard_group_dev(self._h, C.c_uint32(l), C.c_void_p(d_fp_all), C.byref(nb)))
        return nb.value

    def shard_tuples(self, l, d_tuples):
        self._check(self._lib.hs_index_shard_tuples_dev(self._h, C.c_uint32(l), C.c_void_p(d_tuples)))

    def shard_finish(self, l, d_tuples_all):
        col = C.c_uint32(0)
        self._check(self._lib.hs_index_shard_finish_dev(self._h, C.c_uint32(l), C.c_void_p(d_tuples_all), C.byref(col)))
        return col.value

    def shard_end(self, seed):
        self._check(self._lib.hs_index_shard_end(self._h, C.c_uint32(seed)))
        return self.index_info()

    def index_build_windows(self, residues, seq_start):
        """the DB = every length-k window of every sequence of a concatenated
        residue-code buffer (kmer_search.cpp:64-83 order).  residues: uint8 codes; seq_start:
        n_seq + 1 ascending offsets ending at len(residues).  Returns (index info, window start
        positions [n_windows] uint32); DB id i = window i."""
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        seq_start = np.ascontiguousarray(seq_start, dtype=np.uint64)
        n_seq = len(seq_start) - 1
        assert n_seq >= 0 and int(seq_start[-1]) == len(residues)
        lens = np.diff(seq_start.astype(np.int64))
        n_win = int(np.maximum(lens - self.k + 1, 0).sum())
        pos = np.empty(n_win, dtype=np.uint32)
        n = C.c_uint64(0)
        self._check(self._lib.hs_index_build_windows(self._h, _vp(residues), C.c_uint64(len(residues)),
                                                     _vp(seq_start), C.c_uint64(n_seq), C.byref(n),
                                                     _vp(pos)))
        assert int(n.value) == n_win
        return self.index_info(), pos

    def index_save(self, path):
        """write the built index (parameters, planes, table, codes, L tables)."""
        self._check(self._lib.hs_index_save(self._h, str(path).encode()))

    def index_load(self, path):
        """Restore an index written by index_save into a handle created with the same parameters,
        planes and coordinate table (HS_ERR_IO otherwise)."""
        self._check(self._lib.hs_index_load(self._h, str(path).encode()))
        return self.index_info()

    # -- a8..a10
    def query(self, centers, R, cap=None, want_cand=True):
        centers = np.ascontiguousarray(centers, dtype=np.float64)
        nq = centers.shape[0]
        assert centers.shape == (nq, self.d)
        cap = int(cap) if cap is not None else max(1024, 64 * nq)
        while True:
            hq = np.empty(cap, dtype=np.uint32)
            hid = np.empty(cap, dtype=np.uint32)
            ht = np.empty(cap, dtype=np.uint32)
            hd = np.empty(cap, dtype=np.float64)
            cand = np.zeros((nq, self.L), dtype=np.uint64) if want_cand else None
            n = C.c_uint64(0)
            st = self._lib.hs_query(self._h, _vp(centers), C.c_uint64(nq), C.c_double(R), _vp(hq),
                                    _vp(hid), _vp(ht), _vp(hd), C.c_uint64(cap), C.byref(n),
                                    _vp(cand) if want_cand else None)
            if st == HS_ERR_CAPACITY:
                cap = int(n.value)
                continue
            self._check(st)
            n = int(n.value)
            return dict(q=hq[:n], id=hid[:n], table=ht[:n], dist=hd[:n], cand=cand)

    def query_codes(self, qcodes, R, cap=None, want_cand=True):
        """hs_query_codes: the queries are k-mers given as residue codes [nq][k] (uint8)."""
        qcodes = np.ascontiguousarray(qcodes, dtype=np.uint8)
        nq = qcodes.shape[0]
        assert qcodes.shape == (nq, self.k)
        cap = int(cap) if cap is not None else max(1024, 64 * nq)
        while True:
            hq = np.empty(cap, dtype=np.uint32)
            hid = np.empty(cap, dtype=np.uint32)
            ht = np.empty(cap, dtype=np.uint32)
            hd = np.empty(cap, dtype=np.float64)
            cand = np.zeros((nq, self.L), dtype=np.uint64) if want_cand else None
            n = C.c_uint64(0)
            st = self._lib.hs_query_codes(self._h, _vp(qcodes), C.c_uint64(nq), C.c_double(R), _vp(hq),
                                          _vp(hid), _vp(ht), _vp(hd), C.c_uint64(cap), C.byref(n),
                                          _vp(cand) if want_cand else None)
            if st == HS_ERR_CAPACITY:
                cap = int(n.value)
                continue
            self._check(st)
            n = int(n.value)
            return dict(q=hq[:n], id=hid[:n], table=ht[:n], dist=hd[:n], cand=cand)

    def query_into(self, queries, R, hq, hid, ht, hd, codes=False):
        """hs_query / hs_query_codes (HOST pointers, the PCIe-inclusive entry points) into caller-owned
        numpy arrays -- no allocation per call, for timing.  Returns the number of hits."""
        queries = np.ascontiguousarray(queries, dtype=np.uint8 if codes else np.float64)
        n = C.c_uint64(0)
        fn = self._lib.hs_query_codes if codes else self._lib.hs_query
        st = fn(self._h, _vp(queries), C.c_uint64(queries.shape[0]), C.c_double(R), _vp(hq), _vp(hid), _vp(ht),
                _vp(hd), C.c_uint64(len(hq)), C.byref(n), None)
        if st == HS_ERR_CAPACITY:
            e = HsError(st, self._lib.hs_last_error(self._h).decode())
            e.needed = int(n.value)
            raise e
        self._check(st)
        return int(n.value)

    def query_dev(self, d_centers_ptr, nq, R, d_q, d_id, d_table, d_dist, cap, d_cand=None, codes=False):
        """Raw device-pointer call (ints from torch .data_ptr()).  Returns the number of hits; raises
        HsError(HS_ERR_CAPACITY) with the required size in .needed when cap is too small.
        codes=True: d_centers_ptr points at residue codes uint8 [nq][k] (hs_query_codes_dev)."""
        n = C.c_uint64(0)
        fn = self._lib.hs_query_codes_dev if codes else self._lib.hs_query_dev
        st = fn(self._h, C.c_void_p(d_centers_ptr), C.c_uint64(nq), C.c_double(R),
                                    C.c_void_p(d_q), C.c_void_p(d_id), C.c_void_p(d_table),
                                    C.c_void_p(d_dist), C.c_uint64(cap), C.byref(n),
                                    C.c_void_p(d_cand) if d_cand else None)
        if st == HS_ERR_CAPACITY:
            e = HsError(st, self._lib.hs_last_error(self._h).decode())
            e.needed = int(n.value)
            raise e
        self._check(st)
        return int(n.value)

    def merge_first_table_dev(self, d_q, d_id, d_table, d_dist, n):
        """hs_merge_first_table_dev (device pointers as ints; in place): the number of tuples kept."""
        n_out = C.c_uint64(0)
        self._check(self._lib.hs_merge_first_table_dev(self._h, C.c_void_p(d_q), C.c_void_p(d_id), C.c_void_p(d_table),
                                                       C.c_void_p(d_dist), C.c_uint64(n), C.byref(n_out)))
        return int(n_out.value)

    # -- a11
    def bruteforce(self, centers, R, cap=None):
        centers = np.ascontiguousarray(centers, dtype=np.float64)
        nq = centers.shape[0]
        cap = int(cap) if cap is not None else max(1024, 64 * nq)
        while True:
            hq = np.empty(cap, dtype=np.uint32)
            hid = np.empty(cap, dtype=np.uint32)
            hd = np.empty(cap, dtype=np.float64)
            n = C.c_uint64(0)
            st = self._lib.hs_bruteforce(self._h, _vp(centers), C.c_uint64(nq), C.c_double(R),
                                         _vp(hq), _vp(hid), _vp(hd), C.c_uint64(cap), C.byref(n))
            if st == HS_ERR_CAPACITY:
                cap = int(n.value)
                continue
            self._check(st)
            n = int(n.value)
            return dict(q=hq[:n], id=hid[:n], dist=hd[:n])

    def bruteforce_topk(self, centers, topk):
        """Exact k nearest DB k-mers per query: (ids [nq][topk], squared distances [nq][topk]),
        ordered by (d2, id); ground truth of recall@k."""
        centers = np.ascontiguousarray(centers, dtype=np.float64)
        nq = centers.shape[0]
        nn = np.empty((nq, topk), dtype=np.uint32)
        d2 = np.empty((nq, topk), dtype=np.float64)
        self._check(self._lib.hs_bruteforce_topk(self._h, _vp(centers), C.c_uint64(nq),
                                                 C.c_uint32(topk), _vp(nn), _vp(d2)))
        return nn, d2

    # -- a12
    def self_join(self, R, sqrt_test=True, cap=None, first=0, count=None):
        """All ordered within-bucket pairs (i, j), i != j, within R: dict(i, j, table, dist);
        first/count restrict the i side to a block of the indexed k-mers (one rank's shard)."""
        cap = int(cap) if cap is not None else 1 << 16
        count = self.index_info()["n"] - first if count is None else count
        while True:
            ei = np.empty(cap, dtype=np.uint32)
            ej = np.empty(cap, dtype=np.uint32)
            et = np.empty(cap, dtype=np.uint32)
            ed = np.empty(cap, dtype=np.float64)
            n = C.c_uint64(0)
            st = self._lib.hs_self_join_range(self._h, C.c_uint64(first), C.c_uint64(count),
                                              C.c_double(R), C.c_int(1 if sqrt_test else 0),
                                              _vp(ei), _vp(ej), _vp(et), _vp(ed), C.c_uint64(cap),
                                              C.byref(n))
            if st == HS_ERR_CAPACITY:
                cap = int(n.value)
                continue
            self._check(st)
            n = int(n.value)
            return dict(i=ei[:n], j=ej[:n], table=et[:n], dist=ed[:n])


def clustering(k, K, L, W, a, b, codes, R, device=0, coords=None):
    """Clustering() of hclust2.cpp:86-151 on the GPU path: (merged, owner, absorbed_table)."""
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n = codes.shape[0]
    assert a.shape == (L, K, 8 * k) and b.shape == (L, K) and codes.shape == (n, k)
    cptr, alpha = None, 0
    if coords is not None:
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cptr, alpha = _vp(coords), coords.shape[0]
    params = _Params(int(k), int(K), int(L), float(W), int(device), alpha)
    merged = np.empty(n, dtype=np.uint8)
    owner = np.empty(n, dtype=np.uint32)
    table = np.empty(n, dtype=np.uint32)
    err = C.create_string_buffer(512)
    st = lib.hs_clustering(C.byref(params), _vp(a), _vp(b), cptr, _vp(codes), C.c_uint64(n),
                           C.c_double(R), _vp(merged), _vp(owner), _vp(table), err, C.c_uint32(512))
    if st != HS_OK:
        raise HsError(st, err.value.decode())
    return merged, owner, table


class ClusterState:
    """Clustering() table by table (hs_clustering_begin/.../end): the form the multi-GPU driver
    (hsearch_amd.dist.clustering_sharded) uses; one GPU with world=1 equals clustering()."""

    def __init__(self, k, K, L, W, a, b, codes, R, device=0, coords=None):
        self._lib = load()
        self._a = np.ascontiguousarray(a, dtype=np.float64)
        self._b = np.ascontiguousarray(b, dtype=np.float64)
        self._codes = np.ascontiguousarray(codes, dtype=np.uint8)   # kept alive: the state borrows it
        self.n = self._codes.shape[0]
        self.L = int(L)
        assert self._a.shape == (L, K, 8 * k) and self._b.shape == (L, K) and self._codes.shape == (self.n, k)
        cptr, alpha = None, 0
        if coords is not None:
            coords = np.ascontiguousarray(coords, dtype=np.float64)
            cptr, alpha = _vp(coords), coords.shape[0]
        params = _Params(int(k), int(K), int(L), float(W), int(device), alpha)
        self._st = C.c_void_p()
        self._err = C.create_string_buffer(512)
        st = self._lib.hs_clustering_begin(C.byref(params), _vp(self._a), _vp(self._b), cptr,
                                           _vp(self._codes), C.c_uint64(self.n), C.c_double(R),
                                           C.byref(self._st), self._err, C.c_uint32(512))
        if st != HS_OK:
            raise HsError(st, self._err.value.decode())
        self._cap = 4 * self.n + 1024

    def table_edges(self, l, rank=0, world=1):
        """This rank's share of table l's within-bucket pairs within R: (i, j, dist), original ids."""
        while True:
            ei = np.empty(self._cap, dtype=np.uint32)
            ej = np.empty(self._cap, dtype=np.uint32)
            ed = np.empty(self._cap, dtype=np.float64)
            n = C.c_uint64(0)
            st = self._lib.hs_clustering_table_edges(self._st, C.c_uint32(l), C.c_uint32(rank),
                                                     C.c_uint32(world), _vp(ei), _vp(ej), _vp(ed),
                                                     C.c_uint64(self._cap), C.byref(n), self._err,
                                                     C.c_uint32(512))
            if st == HS_ERR_CAPACITY:
                self._cap = int(n.value)
                continue
            if st != HS_OK:
                raise HsError(st, self._err.value.decode())
            n = int(n.value)
            return ei[:n], ej[:n], ed[:n]

    def table_apply(self, l, edge_i, edge_j):
        """The greedy pass of table l over ALL ranks' edges (any order); host only."""
        edge_i = np.ascontiguousarray(edge_i, dtype=np.uint32)
        edge_j = np.ascontiguousarray(edge_j, dtype=np.uint32)
        assert edge_i.shape == edge_j.shape
        st = self._lib.hs_clustering_table_apply(self._st, C.c_uint32(l), _vp(edge_i), _vp(edge_j),
                                                 C.c_uint64(edge_i.shape[0]))
        if st != HS_OK:
            raise HsError(st, "hs_clustering_table_apply")

    def end(self):
        """(merged, owner, absorbed_table); frees the state."""
        merged = np.empty(self.n, dtype=np.uint8)
        owner = np.empty(self.n, dtype=np.uint32)
        table = np.empty(self.n, dtype=np.uint32)
        st, self._st = self._st, None
        rc = self._lib.hs_clustering_end(st, _vp(merged), _vp(owner), _vp(table))
        if rc != HS_OK:
            raise HsError(rc, "hs_clustering_end")
        return merged, owner, table

    def __del__(self):
        if getattr(self, "_st", None):
            self._lib.hs_clustering_end(self._st, None, None, None)
            self._st = None


def clusters_file_text(merged, owner, table, names=None):
    """The reference's clusters file (hclust2.cpp:137-150) from hs_clustering's outputs."""
    n = len(merged)
    members = {}
    order = np.lexsort((np.arange(n), table.astype(np.int64)))
    for i in order:
        if merged[i] == 2:
            members.setdefault(int(owner[i]), []).append(int(i))
    out, cid = [], 0
    for i in range(n):
        if merged[i] in (0, 1):
            ids = [i] + members.get(i, [])
            out.append("#clusterid:%d:size%d" % (cid, len(ids)))
            out.extend(str(j) if names is None else names[j] for j in ids)
            cid += 1
    return "\n".join(out) + "\n"
