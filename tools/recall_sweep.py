"""W per config "chosen from a sweep so that radius recall >= 0.9".  For each W of a
grid: index build at the bench's shape, radius recall of the LSH hits against the exhaustive scan
(hs_bruteforce) on a query subsample, candidates per query and queries/s of a short timed loop.
Usage (GPU box): python tools/recall_sweep.py [--n 10000000 --K 16 --L 8 --queries 100000] > out.json"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hsearch_amd import Engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=100_000)
ap.add_argument("--k", type=int, default=25)
ap.add_argument("--K", type=int, default=16)
ap.add_argument("--L", type=int, default=8)
ap.add_argument("--R", type=float, default=40.0)
ap.add_argument("--grid", type=str, default="160,180,200,220,240,260,280,300,350,400")
ap.add_argument("--recall-queries", type=int, default=2000)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
codes = synth.make_db(args.n, args.k)
centers, _ = synth.make_queries(codes, args.queries)
dev = torch.device("cuda", 0)
d_centers = torch.from_numpy(centers).to(dev)
nr = min(args.recall_queries, args.queries)
truth = None
rows = []
for W in [float(w) for w in args.grid.split(",")]:
    a, b = synth.make_planes(args.k, args.K, args.L, W)
    eng = Engine(args.k, args.K, args.L, W, a, b)
    eng.index_build(codes)
    if truth is None:
        bf = eng.bruteforce(centers[:nr], args.R)
        truth = set(zip(bf["q"].tolist(), bf["id"].tolist()))
    lsh = eng.query(centers[:nr], args.R, want_cand=False)
    found = set(zip(lsh["q"].tolist(), lsh["id"].tolist()))
    cap = 64 * args.queries
    out = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(3)] + [torch.empty(cap, dtype=torch.float64, device=dev)]
    def step():
        return eng.query_dev(d_centers.data_ptr(), args.queries, args.R, out[0].data_ptr(), out[1].data_ptr(),
                             out[2].data_ptr(), out[3].data_ptr(), cap)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    p = eng.profile()
    rows.append({"W": W, "radius_recall": len(truth & found) / max(len(truth), 1), "true_pairs": len(truth),
                 "candidates_per_query": p["candidates"] / args.queries, "queries_per_s": args.queries / dt,
                 "ms_per_step": dt * 1e3, "ms_join": p["ms_join"], "hits": p["hits"]})
    print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    eng.close()
ok = [r["W"] for r in rows if r["radius_recall"] >= 0.9]
print(json.dumps({"shape": vars(args), "sweep": rows, "smallest_W_with_radius_recall_0.9": min(ok) if ok else None}, indent=1))
