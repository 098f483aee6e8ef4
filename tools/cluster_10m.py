import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import hsearch_amd
from hsearch_amd import synth
k, K, L, W, R, n = 25, 16, 8, 200.0, 40.0, 10_000_000
rng = np.random.default_rng(3)
fam = rng.integers(0, 20, size=(20000, k), dtype=np.uint8)
rows = np.repeat(fam, 50, axis=0)
m = rng.integers(0, 5, size=len(rows))
for s in range(4):
    sel = np.nonzero(m > s)[0]
    rows[sel, rng.integers(0, k, size=len(sel))] = rng.integers(0, 20, size=len(sel), dtype=np.uint8)
codes = np.concatenate([rows, synth.make_db(n - len(rows), k, seed=9)])
rng.shuffle(codes)
a, b = synth.make_planes(k, K, L, W, seed=77)
for it in range(2):
    t0 = time.time(); merged, owner, table = hsearch_amd.clustering(k, K, L, W, a, b, codes, R)
    dt = time.time() - t0
    print("hs_clustering n=%d: %.3f s, absorbed %d, clusters with members %d" % (n, dt, int((merged == 2).sum()) if hasattr(merged, 'sum') else -1, len(np.unique(owner[owner != np.arange(n)])) if hasattr(owner,'__len__') else -1), flush=True)
