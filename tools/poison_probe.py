"""Uninitialised-memory hunt: fill (nearly) all free HBM with a byte pattern, free it, then run a scenario -- device
buffers the library allocates afterwards come back holding the pattern instead of the zeros of a fresh process.
Scenario: configs[2] shape, 125 k queries (their own batch) against the first 125 k of a 10^6-query batch.
argv: [pattern byte, default 255] [n db k-mers] """
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hsearch_amd import Engine, synth
pat = int(sys.argv[1]) if len(sys.argv) > 1 else 255
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
dev = torch.device("cuda", 0)
if pat >= 0:
    free, total = torch.cuda.mem_get_info()
    chunks = []
    left = int(free * 0.97)
    while left > (1 << 30):
        sz = min(left, 16 << 30)
        try:
            t = torch.empty(sz, dtype=torch.uint8, device=dev)
        except Exception:
            break
        t.fill_(pat)
        chunks.append(t)
        left -= sz
    torch.cuda.synchronize()
    print("poisoned", sum(c.numel() for c in chunks) / 2**30, "GiB with", pat, file=sys.stderr)
    del chunks
    torch.cuda.empty_cache()
k, K, L, W, R = 25, 20, 32, 160.0, 40.0
nq, nq_all = 125_000, 1_000_000
a, b = synth.make_planes(k, K, L, W)
codes = synth.make_db(n, k)
qcodes_all, _ = synth.make_query_codes(codes, nq_all)
eng = Engine(k, K, L, W, a, b)
eng.index_build(codes)
j = eng.query_codes(qcodes_all[:nq], R, want_cand=False)
big = eng.query_codes(qcodes_all, R, cap=4 * nq_all, want_cand=False)
cut = int(np.searchsorted(big["q"], nq))
print("own batch", len(j["q"]), "prefix of the big batch", cut, file=sys.stderr)
kj = (j["q"].astype(np.int64) << 32) | j["id"]
kb = (big["q"][:cut].astype(np.int64) << 32) | big["id"][:cut]
miss = np.setdiff1d(kj, kb); extra = np.setdiff1d(kb, kj)
print("missing from the big batch", [(int(x >> 32), int(x & 0xffffffff)) for x in miss[:10]], "extra", [(int(x >> 32), int(x & 0xffffffff)) for x in extra[:10]], file=sys.stderr)
for x in miss[:5]:
    i = int(np.nonzero(kj == x)[0][0])
    print("  the missing hit in its own batch: q", j["q"][i], "id", j["id"][i], "table", j["table"][i], "dist", j["dist"][i], file=sys.stderr)
big2 = eng.query_codes(qcodes_all, R, cap=4 * nq_all, want_cand=False)
print("second big pass equal to the first:", all(np.array_equal(big[f], big2[f]) for f in ("q", "id", "table", "dist")), len(big["q"]), len(big2["q"]), file=sys.stderr)
k1 = (big["q"].astype(np.int64) << 32) | big["id"]
k2 = (big2["q"].astype(np.int64) << 32) | big2["id"]
m1 = np.setdiff1d(k2, k1); e1 = np.setdiff1d(k1, k2)
print("first pass lacks", len(m1), [(int(x >> 32), int(x & 0xffffffff)) for x in m1[:8]], "has extra", len(e1), [(int(x >> 32), int(x & 0xffffffff)) for x in e1[:8]], file=sys.stderr)
for x in m1[:8]:
    i = int(np.nonzero(k2 == x)[0][0])
    print("  lacking: q", big2["q"][i], "id", big2["id"][i], "table", big2["table"][i], "dist", big2["dist"][i], file=sys.stderr)
big3 = eng.query_codes(qcodes_all, R, cap=4 * nq_all, want_cand=False)
print("third == second:", all(np.array_equal(big3[f], big2[f]) for f in ("q", "id", "table", "dist")), file=sys.stderr)
cut2 = int(np.searchsorted(big2["q"], nq))
print("second big pass prefix", cut2, "equal to own batch:", cut2 == len(j["q"]) and np.array_equal(big2["id"][:cut2], j["id"]), file=sys.stderr)
eng.close()
