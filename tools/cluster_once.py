"""hs_clustering at config 4, twice (the second run is the warm one); for rocprofv3."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsearch_amd
from hsearch_amd import synth
k, K, L, W, R, n = 25, 16, 8, 200.0, 40.0, 1_000_000
rng = np.random.default_rng(3)
fam = rng.integers(0, 20, size=(2000, k), dtype=np.uint8)
rows = np.repeat(fam, 50, axis=0)
m = rng.integers(0, 5, size=len(rows))
for s in range(4):
    sel = np.nonzero(m > s)[0]
    rows[sel, rng.integers(0, k, size=len(sel))] = rng.integers(0, 20, size=len(sel), dtype=np.uint8)
codes = np.concatenate([rows, synth.make_db(n - len(rows), k, seed=9)])
rng.shuffle(codes)
a, b = synth.make_planes(k, K, L, W, seed=77)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t0 = time.time(); hsearch_amd.clustering(k, K, L, W, a, b, codes, R)
    print("hs_clustering %.3f s" % (time.time() - t0), flush=True)
