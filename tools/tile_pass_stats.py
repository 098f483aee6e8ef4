"""How often does a 32x32 (members x queries) tile of the bucket join contain a pair that passes a
PARTIAL lower bound (first P positions, 4 coordinates each)?  Decides whether a staged filter
(few k-steps first, the rest only for tiles with a candidate) would pay.  Runs on the GPU box
(hashing through the C ABI), distances in numpy."""
import sys
import numpy as np
sys.path.insert(0, '.')
from hsearch_amd import Engine, synth
import hsearch_amd
k, K, L, W, R, n, nq = 25, 16, 8, 200.0, 40.0, 10_000_000, 100_000
a, b = synth.make_planes(k, K, L, W); codes = synth.make_db(n, k); centers, _ = synth.make_queries(codes, nq)
eng = Engine(k, K, L, W, a, b)
keys = []
for lo in range(0, n, 1_000_000):
    ints = eng.hash_codes(codes[lo:lo + 1_000_000])[:, 0, :]
    keys.append(np.ascontiguousarray(ints).view([('', np.int32)] * K).ravel())
keys = np.concatenate(keys)
qints = np.ascontiguousarray(eng.hash_points(centers)[:, 0, :]).view([('', np.int32)] * K).ravel()
uq, inv, cnt = np.unique(qints, return_inverse=True, return_counts=True)
coords = np.array(hsearch_amd.AA_COORDS, dtype=np.float64) if hasattr(hsearch_amd, 'AA_COORDS') else None
if coords is None:
    pts = eng.embed_codes(np.arange(20, dtype=np.uint8).reshape(20, 1).repeat(k, 1))
    coords = pts[:, :8]
rng = np.random.default_rng(1)
for rank in (0, 3, 10, 40):
    j = np.argsort(-cnt)[rank]
    members = np.nonzero(keys == uq[j])[0]
    queries = np.nonzero(inv == j)[0]
    if len(members) < 64 or len(queries) < 64:
        continue
    ms = rng.choice(members, min(4096, len(members)), replace=False)
    qs = rng.choice(queries, min(1024, len(queries)), replace=False)
    X = coords[codes[ms]][:, :, :4]                    # [m][k][4]
    C = centers[qs].reshape(len(qs), k, 8)[:, :, :4]   # [q][k][4]
    print("bucket rank %d: M=%d nQ=%d (sample %dx%d)" % (rank, len(members), len(queries), len(ms), len(qs)))
    for P in (8, 12, 16, 25):
        x = X[:, :P].reshape(len(ms), -1); c = C[:, :P].reshape(len(qs), -1)
        d2 = (x * x).sum(1)[:, None] + (c * c).sum(1)[None, :] - 2.0 * x @ c.T
        for slack in (0.0, 80.0 * P / 25):
            ok = d2 <= R * R + slack
            mt, qt = len(ms) // 32, len(qs) // 32
            t = ok[:mt * 32, :qt * 32].reshape(mt, 32, qt, 32).any(axis=(1, 3))
            t64 = ok[:mt * 32 // 64 * 64, :qt * 32].reshape(mt // 2, 64, qt, 32).any(axis=(1, 3))
            print("  P=%2d slack %5.1f: pair pass %.2e  32x32 tiles with a pass %.3f  64x32 %.3f  median d2 %.0f" % (
                P, slack, ok.mean(), t.mean(), t64.mean(), np.median(d2)))
