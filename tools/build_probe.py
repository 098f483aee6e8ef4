"""Index build time at the bench workload (10 M 25-mers, L = 8, K = 16): first build of a handle
(allocations included) and best of the following three; argv: "overlap" (default behaviour: the hash of
table l + 1 runs on the side stream beside the sort of table l) and/or "serial" (HS_BUILD_SERIAL)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from hsearch_amd import Engine, synth

k, K, L, W, n = 25, 16, 8, 200.0, int(os.environ.get('HS_PROBE_N', '10000000'))
codes = synth.make_db(n, k)
a, b = synth.make_planes(k, K, L, W)
for cus in sys.argv[1:]:
    os.environ.pop("HS_BUILD_SERIAL", None)
    if cus == "serial":
        os.environ["HS_BUILD_SERIAL"] = "1"
    eng = Engine(k, K, L, W, a, b)
    best = None
    first = None
    for it in range(4):
        print('  build', it, flush=True)
        t0 = time.perf_counter(); eng.index_build(codes); dt = time.perf_counter() - t0
        p = eng.profile()
        if first is None:
            first = dt
            continue
        if best is None or dt < best[0]:
            best = (dt, p)
    print("hash stream %4s: first build %.1f ms; then %.1f ms (%.0f M k-mers/s); device total %.1f, hash %.1f, sort %.1f, gather %.1f"
          % (cus, 1e3 * first, 1e3 * best[0], n / best[0] / 1e6, best[1]["ms_total"], best[1]["ms_hash"], best[1]["ms_sort"],
             best[1]["ms_gather"]), flush=True)
    eng.close()
