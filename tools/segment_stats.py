"""Distribution of the C2 join work over segments (bucket x probing queries): where do pairs and
A-operand builds go?  Host-side analysis on top of the C ABI (hash_points + cand)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from hsearch_amd import Engine, synth
k, K, L, W, R, n, nq = 25, 16, 8, 200.0, 40.0, 10_000_000, 100_000
a, b = synth.make_planes(k, K, L, W); codes = synth.make_db(n, k); centers, _ = synth.make_queries(codes, nq)
eng = Engine(k, K, L, W, a, b); eng.index_build(codes)
ints = eng.hash_points(centers)                  # [nq][L][K]
res = eng.query(centers, R)
cand = res["cand"]                               # [nq][L] bucket sizes
segs = []
for l in range(L):
    keys = np.ascontiguousarray(ints[:, l, :]).view([('', np.int32)] * K).ravel()
    uq, inv, cnt = np.unique(keys, return_inverse=True, return_counts=True)
    M = np.zeros(len(uq), dtype=np.int64); M[inv] = cand[:, l]
    segs.append(np.stack([M, cnt], 1))
segs = np.concatenate(segs); segs = segs[segs[:, 0] > 0]
M, Q = segs[:, 0], segs[:, 1]
pairs = M * Q
builds = np.ceil(M / 128) * np.ceil(Q / 2048)     # wave-level A builds (128 members each)
print("segments", len(segs), "pairs %.3e" % pairs.sum(), "wave A-builds %.3e" % builds.sum())
for lo, hi in [(1, 2), (3, 7), (8, 31), (32, 127), (128, 511), (512, 2047), (2048, 10**9)]:
    m = (Q >= lo) & (Q <= hi)
    print("nQ %5d..%-9d segs %7d  pairs %5.1f%%  builds %5.1f%%  mean M %8.0f" % (lo, hi, m.sum(), 100 * pairs[m].sum() / pairs.sum(), 100 * builds[m].sum() / builds.sum(), M[m].mean() if m.any() else 0))
# wave-item view (128 members x <= 2048 queries per item): items and 32-query tiles per class
j = (Q >= 3) & (M >= 16)
items = np.ceil(M / 128) * np.ceil(Q / 2048)
tiles = np.ceil(M / 128) * np.ceil(Q / 32)
print("joined: items %.3e tiles %.3e" % (items[j].sum(), tiles[j].sum()))
for lo, hi in [(3, 32), (33, 96), (97, 256), (257, 512), (513, 2048), (2049, 10**9)]:
    m = j & (Q >= lo) & (Q <= hi)
    print("nQ %5d..%-9d items %5.1f%%  tiles %5.1f%%  tiles/item %6.1f" % (lo, hi, 100 * items[m].sum() / items[j].sum(), 100 * tiles[m].sum() / tiles[j].sum(), tiles[m].sum() / max(1, items[m].sum())))
