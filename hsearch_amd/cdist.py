"""ctypes binding of include/hsearch_dist.h (libhsearch_dist.so): the C++/RCCL multi-GPU layer the
host programs use (hs_motif_both_points --gpus n).  Tests only; no compute here."""
import ctypes as C
import os

import numpy as np

from . import capi

RCCL_LOCAL, LOOPBACK = 0, 1
EXPORTS = ["hs_comm_create", "hs_comm_unique_id", "hs_comm_create_rank", "hs_comm_destroy", "hs_comm_world",
           "hs_comm_last_error", "hs_shard_bounds", "hs_allgather_hits", "hs_comm_barrier", "hs_comm_query",
           "hs_comm_query_codes", "hs_comm_query_tables", "hs_comm_query_buckets", "hs_assign_tables"]

_lib = None


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhsearch_dist.so")


def load():
    global _lib
    if _lib is None:
        capi.load()   # libhsearch_dist.so needs libhsearch_amd.so (found through its rpath as well)
        if not os.path.exists(lib_path()):
            raise ImportError("%s is missing: run __graft_entry__.build()" % lib_path())
        lib = C.CDLL(lib_path())
        lib.hs_comm_destroy.restype = None
        lib.hs_comm_destroy.argtypes = [C.c_void_p]
        lib.hs_comm_world.restype = C.c_uint32
        lib.hs_comm_world.argtypes = [C.c_void_p]
        lib.hs_comm_last_error.restype = C.c_char_p
        lib.hs_comm_last_error.argtypes = [C.c_void_p, C.c_uint32]
        lib.hs_shard_bounds.restype = None
        lib.hs_assign_tables.restype = None
        _lib = lib
    return _lib


def shard_bounds(n, world, rank):
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    load().hs_shard_bounds(C.c_uint64(n), C.c_uint32(world), C.c_uint32(rank), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def assign_tables(cost, L, world):
    """hs_assign_tables: owner[l] = rank of table l (longest processing time first by cost; None: equal)."""
    owner = np.zeros(L, dtype=np.uint32)
    c = None if cost is None else np.ascontiguousarray(cost, dtype=np.float64)
    assert c is None or len(c) == L
    load().hs_assign_tables(_p(c), C.c_uint32(L), C.c_uint32(world), _p(owner))
    return owner


def _p(x):
    """numpy array -> host pointer, int -> device pointer, None -> NULL."""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, int):
        return C.c_void_p(x)
    return x.ctypes.data_as(C.c_void_p)


class Comm:
    def __init__(self, kind, world, devices=None):
        self._lib = load()
        self._h = C.c_void_p()
        self.world = world
        err = C.create_string_buffer(512)
        dev = None
        if devices is not None:
            dev = (C.c_int * world)(*devices)
        st = self._lib.hs_comm_create(C.c_int(kind), dev, C.c_uint32(world), C.byref(self._h), err,
                                      C.c_uint32(len(err)))
        if st != capi.HS_OK:
            raise capi.HsError(st, err.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hs_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self, rank):
        return self._lib.hs_comm_last_error(self._h, C.c_uint32(rank)).decode()

    def query_status(self, rank, engine, centers_block, q_offset, R, cap=1024):
        """hs_comm_query's raw status (tests of the failure protocol); engine may be None."""
        centers_block = np.ascontiguousarray(centers_block, dtype=np.float64)
        hq = np.empty(cap, np.uint32); hid = np.empty(cap, np.uint32); ht = np.empty(cap, np.uint32)
        hd = np.empty(cap, np.float64)
        n_total = C.c_uint64(0)
        return self._lib.hs_comm_query(self._h, C.c_uint32(rank), engine._h if engine is not None else None,
                                       _p(centers_block), C.c_uint64(centers_block.shape[0]), C.c_uint32(q_offset),
                                       C.c_double(R), _p(hq), _p(hid), _p(ht), _p(hd), C.c_uint64(cap),
                                       C.byref(n_total))

    def allgather_hits(self, rank, q, id_, table, dist, n_local, q_offset, out_q, out_id, out_table, out_dist,
                       cap):
        """Raw call (pointers: numpy arrays = host, ints = device).  Returns (status, n_total)."""
        n_total = C.c_uint64(0)
        st = self._lib.hs_allgather_hits(self._h, C.c_uint32(rank), _p(q), _p(id_), _p(table), _p(dist),
                                         C.c_uint64(n_local), C.c_uint32(q_offset), _p(out_q), _p(out_id),
                                         _p(out_table), _p(out_dist), C.c_uint64(cap), C.byref(n_total))
        return st, n_total.value

    def gather_host(self, rank, hits, q_offset, with_table=True):
        """Loopback convenience: hits = dict(q, id, table, dist) of numpy arrays -> dict of all ranks'."""
        n = len(hits["q"])
        cap = 16
        while True:
            out = dict(q=np.empty(cap, np.uint32), id=np.empty(cap, np.uint32),
                       table=np.empty(cap, np.uint32) if with_table else None, dist=np.empty(cap, np.float64))
            st, tot = self.allgather_hits(rank, np.ascontiguousarray(hits["q"], np.uint32),
                                          np.ascontiguousarray(hits["id"], np.uint32),
                                          np.ascontiguousarray(hits["table"], np.uint32) if with_table else None,
                                          np.ascontiguousarray(hits["dist"], np.float64), n, q_offset,
                                          out["q"], out["id"], out["table"], out["dist"], cap)
            if st == capi.HS_ERR_CAPACITY:
                cap = tot
                continue
            if st != capi.HS_OK:
                raise capi.HsError(st, self._lib.hs_comm_last_error(self._h, C.c_uint32(rank)).decode())
            return {k: (v[:tot] if v is not None else None) for k, v in out.items()}

    def query(self, rank, engine, centers_block, q_offset, R, cap=None, codes=False):
        """hs_comm_query: this rank's block through `engine` (bound to the rank's GPU), all ranks'
        hits back on the host.  codes=True: the block is residue codes uint8 [nq][k]
        (hs_comm_query_codes)."""
        centers_block = np.ascontiguousarray(centers_block, dtype=np.uint8 if codes else np.float64)
        nq = centers_block.shape[0]
        cap = int(cap) if cap else max(1024, 64 * nq)
        while True:
            hq = np.empty(cap, np.uint32); hid = np.empty(cap, np.uint32); ht = np.empty(cap, np.uint32)
            hd = np.empty(cap, np.float64)
            n_total = C.c_uint64(0)
            fn = self._lib.hs_comm_query_codes if codes else self._lib.hs_comm_query
            st = fn(self._h, C.c_uint32(rank), engine._h, _p(centers_block), C.c_uint64(nq),
                    C.c_uint32(q_offset), C.c_double(R), _p(hq), _p(hid), _p(ht), _p(hd),
                    C.c_uint64(cap), C.byref(n_total))
            if st == capi.HS_ERR_CAPACITY:
                cap = n_total.value
                continue
            if st != capi.HS_OK:
                raise capi.HsError(st, self._lib.hs_comm_last_error(self._h, C.c_uint32(rank)).decode())
            n = n_total.value
            return dict(q=hq[:n], id=hid[:n], table=ht[:n], dist=hd[:n])

    def query_tables(self, rank, engine, tables, queries, R, cap=None, codes=False):
        """hs_comm_query_tables: the table-partitioned layout -- `engine` holds the tables `tables` (global
        numbers, ascending) over all k-mers, `queries` are ALL queries (points, or residue codes with
        codes=True); every rank gets the merged hits, the reference's order."""
        queries = np.ascontiguousarray(queries, dtype=np.uint8 if codes else np.float64)
        tables = np.ascontiguousarray(tables, dtype=np.uint32)
        nq = queries.shape[0]
        cap = int(cap) if cap else max(1024, 64 * nq)
        while True:
            hq = np.empty(cap, np.uint32); hid = np.empty(cap, np.uint32); ht = np.empty(cap, np.uint32)
            hd = np.empty(cap, np.float64)
            n_total = C.c_uint64(0)
            st = self._lib.hs_comm_query_tables(self._h, C.c_uint32(rank), engine._h if engine is not None else None,
                                                _p(tables), C.c_uint32(len(tables)),
                                                None if codes else _p(queries), _p(queries) if codes else None,
                                                C.c_uint64(nq), C.c_double(R), _p(hq), _p(hid), _p(ht), _p(hd),
                                                C.c_uint64(cap), C.byref(n_total))
            if st == capi.HS_ERR_CAPACITY:
                cap = n_total.value
                continue
            if st != capi.HS_OK:
                raise capi.HsError(st, self._lib.hs_comm_last_error(self._h, C.c_uint32(rank)).decode())
            n = n_total.value
            return dict(q=hq[:n], id=hid[:n], table=ht[:n], dist=hd[:n])

    def query_buckets(self, rank, engine, queries, R, cap=None, codes=False):
        """hs_comm_query_buckets: the bucket-partitioned layout -- every rank's `engine` holds the whole index,
        `queries` are ALL queries (points, or residue codes with codes=True), rank r searches the buckets of
        part r of world; every rank gets the merged hits, the reference's order."""
        queries = np.ascontiguousarray(queries, dtype=np.uint8 if codes else np.float64)
        nq = queries.shape[0]
        cap = int(cap) if cap else max(1024, 64 * nq)
        while True:
            hq = np.empty(cap, np.uint32); hid = np.empty(cap, np.uint32); ht = np.empty(cap, np.uint32)
            hd = np.empty(cap, np.float64)
            n_total = C.c_uint64(0)
            st = self._lib.hs_comm_query_buckets(self._h, C.c_uint32(rank), engine._h if engine is not None else None,
                                                 None if codes else _p(queries), _p(queries) if codes else None,
                                                 C.c_uint64(nq), C.c_double(R), _p(hq), _p(hid), _p(ht), _p(hd),
                                                 C.c_uint64(cap), C.byref(n_total))
            if st == capi.HS_ERR_CAPACITY:
                cap = n_total.value
                continue
            if st != capi.HS_OK:
                raise capi.HsError(st, self._lib.hs_comm_last_error(self._h, C.c_uint32(rank)).decode())
            n = n_total.value
            return dict(q=hq[:n], id=hid[:n], table=ht[:n], dist=hd[:n])
