"""Test helper: an index file (the format of hs_index_save, hsearch_amd/csrc/hs_capi.hip
IndexFileHeader + payload) written on the HOST from the oracle's bucket ints -- so the file checks
can be tested without a GPU, and a GPU handle can be asked to load an index it did not build."""
import struct

import numpy as np

from hsearch_amd import capi

MASK = (1 << 64) - 1
CHUNK = 64 << 20


class PayloadHash:
    """hs_capi.hip PayloadHash: order-dependent 64-bit hash over sections of at most 64 MiB."""

    def __init__(self):
        self.h = 0x9e3779b97f4a7c15
        self.bytes = 0

    @staticmethod
    def _mix(h, w):
        h = ((h ^ w) * 0xff51afd7ed558ccd) & MASK
        return h ^ (h >> 29)

    def add(self, data):
        data = bytes(data)
        for off in range(0, max(len(data), 1), CHUNK):
            part = data[off:off + CHUNK]
            self.h = self._mix(self.h, len(part))
            pad = part + b"\0" * (-len(part) % 8)
            for w in np.frombuffer(pad, dtype="<u8").tolist():
                self.h = self._mix(self.h, w)
            self.bytes += len(part)


def build_tables(buckets, seed=0):
    """buckets int32 [n][L][K] (oracle.hash_all) -> per table (ids, dir_key, dir_start, dir_tuple)
    as hs_index_build lays them out: entries sorted by (fingerprint, id)."""
    n, L, K = buckets.shape
    out = []
    for l in range(L):
        fp = np.array([capi.key_fingerprint(buckets[i, l], seed) for i in range(n)], dtype=np.uint64)
        order = np.lexsort((np.arange(n), fp))
        ids = order.astype(np.uint32)
        fps = fp[order]
        first = np.nonzero(np.concatenate([[True], fps[1:] != fps[:-1]]))[0] if n else np.zeros(0, dtype=np.int64)
        dir_key = fps[first]
        dir_start = np.concatenate([first, [n]]).astype(np.uint32)
        dir_tuple = buckets[ids[first], l].astype(np.int32).reshape(len(first), K)
        out.append((ids, dir_key, dir_start, dir_tuple))
    return out


def header(k, K, L, alphabet, W, n, seed, n_buckets, max_bucket, payload_bytes, payload_hash):
    nb = list(n_buckets) + [0] * (32 - len(n_buckets))
    mb = list(max_bucket) + [0] * (32 - len(max_bucket))
    return struct.pack("<8s4IdQ2I32Q32Q2Q", b"HSIDX002", k, K, L, alphabet, W, n, seed, 0, *nb, *mb,
                       payload_bytes, payload_hash)


def write(path, k, K, L, W, a, b, codes, tables, coords=None, seed=0, alphabet=20, tamper=None):
    """tamper(sections) may edit the list of payload sections (numpy arrays) before they are hashed
    and written: a file that is self-consistent (valid hash) but breaks a content rule."""
    from hsearch_amd import synth
    table = np.zeros((32, 8))
    table[:alphabet] = synth.coords() if coords is None else coords
    sections = [np.array(a, dtype=np.float64), np.array(b, dtype=np.float64), table,
                np.array(codes, dtype=np.uint8)]          # copies: tamper() edits them in place
    for ids, dir_key, dir_start, dir_tuple in tables:
        sections += [np.array(ids, dtype=np.uint32), np.array(dir_key, dtype=np.uint64),
                     np.array(dir_start, dtype=np.uint32), np.array(dir_tuple, dtype=np.int32)]
    if tamper:
        tamper(sections)
    ph = PayloadHash()
    for sec in sections:
        ph.add(sec.tobytes())
    n = len(codes)
    n_buckets = [len(t[1]) for t in tables]
    max_bucket = [int(np.diff(t[2].astype(np.int64)).max()) if len(t[1]) else 0 for t in tables]
    with open(path, "wb") as f:
        f.write(header(k, K, L, alphabet, W, n, seed, n_buckets, max_bucket, ph.bytes, ph.h))
        for sec in sections:
            f.write(sec.tobytes())
