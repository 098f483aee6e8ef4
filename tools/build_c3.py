"""Index build at the configs[2] shape (10^8 25-mers, L = 32, K = 20, W = 160): first build of a process
(allocation of 157 GB included) and two rebuilds on the warm handle; also C2 (10^7, L = 8, K = 16, W = 212).
argv: [tables] (default 32; fewer = the same shape with fewer tables, for quick A/B runs)"""
import sys, time, json
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hsearch_amd import Engine, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = {}
for name, (n, K, Lx, W) in {"c2": (10_000_000, 16, 8, 212.0), "c3": (100_000_000, 20, L, 160.0)}.items():
    k = 25
    codes = synth.make_db(n, k)
    a, b = synth.make_planes(k, K, Lx, W)
    eng = Engine(k, K, Lx, W, a, b)
    runs = []
    for it in range(3):
        t0 = time.perf_counter(); info = eng.index_build(codes); dt = time.perf_counter() - t0
        p = eng.profile()
        runs.append({"wall_s": dt, "device_ms": p["ms_total"], "hash_ms": p["ms_hash"], "group_ms": p["ms_sort"], "copies_ms": p["ms_gather"]})
    out[name] = {"n": n, "L": Lx, "K": K, "runs": runs, "index_bytes": info["device_bytes"], "n_buckets_table0": info["n_buckets"][0]}
    eng.close()
print(json.dumps(out))
