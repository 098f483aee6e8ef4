"""Phases of one clustering table at config 4 (n = 1 M, K = 16, L = 1): create, build, self-join."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from hsearch_amd import Engine, synth
k, K, W, R, n = 25, 16, 200.0, 40.0, 1_000_000
codes = synth.make_db(n, k, seed=9)
a, b = synth.make_planes(k, K, 8, W, seed=77)
for l in range(4):
    t0 = time.time(); eng = Engine(k, K, 1, W, a[l:l+1], b[l:l+1]); t1 = time.time()
    eng.index_build(codes); t2 = time.time()
    e = eng.self_join(R, cap=4 * n + 1024); t3 = time.time()
    e = eng.self_join(R, cap=4 * n + 1024); t4 = time.time()
    eng.close(); t5 = time.time()
    print("table %d: create %.1f ms, build %.1f ms, self_join %.1f ms (again %.1f ms), destroy %.1f ms, %d edges"
          % (l, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4), len(e["i"])), flush=True)
    print(eng.index_info() if False else "", end="")
