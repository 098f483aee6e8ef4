"""Two index builds of one shape (the second on the warm handle).  argv: n L K W"""
import sys, time
sys.path.insert(0, '.')
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from hsearch_amd import Engine, synth
n, L, K, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
k = 25
codes = synth.make_db(n, k); a, b = synth.make_planes(k, K, L, W)
eng = Engine(k, K, L, W, a, b)
for it in range(2):
    t0 = time.perf_counter(); eng.index_build(codes); dt = time.perf_counter() - t0
    p = eng.profile()
    print("build %d: %.1f ms wall; device total %.1f hash %.1f sort %.1f gather %.1f" % (it, 1e3 * dt, p["ms_total"], p["ms_hash"], p["ms_sort"], p["ms_gather"]), flush=True)
eng.close()
