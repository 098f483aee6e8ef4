"""Distribution of the join work over segments (bucket x the queries of the batch that probe it): where
do pairs, work items (128 members) and issued MFMA columns go?  Host-side analysis on top of the C ABI
(hash_points + cand).  argv: n L K W nq  (default: configs[2]'s per-GPU shape at W = 160)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from hsearch_amd import Engine, synth
a_ = sys.argv[1:]
n, L, K, W, nq = (int(a_[0]), int(a_[1]), int(a_[2]), float(a_[3]), int(a_[4])) if len(a_) >= 5 else (100_000_000, 32, 20, 160.0, 125_000)
k, R = 25, 40.0
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
items = np.ceil(M / 128) * np.ceil(Q / 8192)
tiles32 = np.ceil(M / 128) * np.ceil(Q / 32)      # 32-column query tiles issued today
tiles16 = np.ceil(M / 128) * np.ceil(Q / 16)      # 16-column granularity
print("segments %d  probes %d  pairs %.3e  items(128) %.3e  issued/useful: 32-col %.2f, 16-col %.2f" % (
    len(segs), Q.sum(), pairs.sum(), items.sum(), tiles32.sum() * 128 * 32 / pairs.sum(), tiles16.sum() * 128 * 16 / pairs.sum()))
print("by queries per segment:")
for lo, hi in [(1, 1), (2, 4), (5, 8), (9, 16), (17, 32), (33, 48), (49, 64), (65, 96), (97, 256), (257, 10**9)]:
    m = (Q >= lo) & (Q <= hi)
    print("  nQ %4d..%-10d segs %8d  pairs %5.1f%%  items %5.1f%%  tiles32 %5.1f%%  mean M %9.0f" % (
        lo, hi, m.sum(), 100 * pairs[m].sum() / pairs.sum(), 100 * items[m].sum() / items.sum(),
        100 * tiles32[m].sum() / tiles32.sum(), M[m].mean() if m.any() else 0))
print("by members per segment:")
for lo, hi in [(1, 16), (17, 128), (129, 512), (513, 2048), (2049, 8192), (8193, 32768), (32769, 10**9)]:
    m = (M >= lo) & (M <= hi)
    print("  M %6d..%-10d segs %8d  pairs %5.1f%%  items %5.1f%%  mean nQ %7.1f  row fill %4.2f" % (
        lo, hi, m.sum(), 100 * pairs[m].sum() / pairs.sum(), 100 * items[m].sum() / items.sum(), Q[m].mean() if m.any() else 0,
        M[m].sum() / (np.ceil(M[m] / 128) * 128).sum() if m.any() else 0))
