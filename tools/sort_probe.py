import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from hsearch_amd import Engine, synth
k, K, L, n = 25, 20, 4, 20_000_000
for W in (200.0, 160.0, 120.0):
    a, b = synth.make_planes(k, K, L, W, seed=5)
    codes = synth.make_db(n, k, seed=6)
    eng = Engine(k, K, L, W, a, b)
    t0 = time.time(); info = eng.index_build(codes); dt = time.time() - t0
    print(W, info["n_buckets"], "build %.3f s" % dt, eng.profile()["ms_sort"], flush=True)
    eng.close()
