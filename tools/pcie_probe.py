"""PCIe-inclusive rate of the host-pointer boundary (hs_query: centres from host memory, hits back
to host memory) at the bench workload, beside the HBM-resident hs_query_dev rate bench.py reports."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from hsearch_amd import Engine, synth

k, K, L, W, R, n, nq = 25, 16, 8, 200.0, 40.0, 10_000_000, 100_000
codes = synth.make_db(n, k)
a, b = synth.make_planes(k, K, L, W)
centers, _ = synth.make_queries(codes, nq)
eng = Engine(k, K, L, W, a, b)
eng.index_build(codes)
cap = 1 << 20
for it in range(8):
    t0 = time.perf_counter()
    res = eng.query(centers, R, cap=cap, want_cand=False)
    dt = time.perf_counter() - t0
    print("hs_query (host pointers): %.2f ms, %.2f M queries/s, %d hits" % (1e3 * dt, nq / dt / 1e6, len(res["q"])), flush=True)
