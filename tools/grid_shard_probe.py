"""configs[2] on 8 GPUs as G query groups x P bucket parts (G * P = 8), index replicated: per-rank pass time of
rank (group 0, part 0) and of the slowest of the group's parts, emulated on one GPU / one handle.
P = 8: the bucket partition; P = 1: query blocks."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hsearch_amd import Engine, synth
k, K, L, W, R = 25, 20, 32, 160.0, 40.0
n, nq_all, N = 100_000_000, 1_000_000, 8
a, b = synth.make_planes(k, K, L, W)
codes = synth.make_db(n, k)
qcodes, _ = synth.make_query_codes(codes, nq_all, seed=synth.SEED_QUERIES)
dev = torch.device("cuda", 0)
d_centers = torch.from_numpy(synth.embed(qcodes)).to(dev)
cap = 4 * nq_all
out = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(3)] + [torch.empty(cap, dtype=torch.float64, device=dev)]
eng = Engine(k, K, L, W, a, b, device=0)
eng.index_build(codes)
res = {}
for P in (8, 4, 2, 1):
    G = N // P
    nq = nq_all // G
    times = []
    for part in range(P):
        eng.set_bucket_partition(part, P)
        def step():
            return eng.query_dev(d_centers.data_ptr(), nq, R, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), cap)
        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 3)
    eng.set_bucket_partition(0, 1)
    res["groups_%d_x_parts_%d" % (G, P)] = {"queries_per_rank": nq, "rank_seconds": times, "job_queries_per_s_before_exchange": nq_all / max(times)}
    print("G", G, "x P", P, "queries per rank", nq, "ms", [round(t * 1e3, 2) for t in times], "job M q/s", round(nq_all / max(times) / 1e6, 1), file=sys.stderr, flush=True)
eng.close()
print(json.dumps(res))
