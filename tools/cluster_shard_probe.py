"""Where the time of the sharded Clustering() goes at config 4 (1 M 25-mers, K=16, L=8): per table
the join of one rank's block (GPU), the pooled-edge greedy pass (host), for world = 1, 2, 4, 8
emulated on one GPU (each rank's block timed separately; a real run does them concurrently)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
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
t0 = time.time(); hsearch_amd.clustering(k, K, L, W, a, b, codes, R); t1 = time.time()
hsearch_amd.clustering(k, K, L, W, a, b, codes, R); t2 = time.time()
print("hs_clustering: first %.3f s, second %.3f s" % (t1 - t0, t2 - t1), flush=True)
for world in (1, 2, 4, 8):
    st = hsearch_amd.ClusterState(k, K, L, W, a, b, codes, R)
    t_edges_max = t_edges_sum = t_apply = 0.0
    n_edges = 0
    for l in range(L):
        parts, worst = [], 0.0
        for r in range(world):
            t = time.time(); parts.append(st.table_edges(l, r, world)); dt = time.time() - t
            worst = max(worst, dt); t_edges_sum += dt
        t_edges_max += worst
        ei = np.concatenate([p[0] for p in parts]); ej = np.concatenate([p[1] for p in parts])
        n_edges += len(ei)
        t = time.time(); st.table_apply(l, ei, ej); t_apply += time.time() - t
    st.end()
    print("world %d: edges (slowest rank, summed over tables) %.3f s, all ranks %.3f s, apply %.3f s, "
          "%d edges = %.1f MB exchanged" % (world, t_edges_max, t_edges_sum, t_apply, n_edges, n_edges * 8 / 1e6),
          flush=True)
