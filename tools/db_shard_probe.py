"""SURVEY 8(e) decision data: configs[2] with the DATABASE sharded over 8 GPUs instead
of replicated, emulated on one GPU.  Replicated (the north star's layout, what bench.py's secondary block
measures): every GPU holds all 10^8 k-mers and searches 1/8 of the 10^6 queries.  Sharded: every GPU holds
1/8 of the k-mers (same planes, all 32 tables of its shard) and searches ALL 10^6 queries; a query's hits
are the union over shards (first-seen table per (query, id) is a per-id property, so it is shard-local;
the final order is a merge by (query, table, id)).  Both layouts scan the same (member, query) pairs per
GPU; what differs is the shape of the segments -- a bucket of a shard has 1/8 of the members and is probed by
8 x the queries of a batch -- and the query-side work (hash, probe, segment grouping), which every GPU now
does for all queries.  The job's rate is 10^6 queries / time of one GPU's pass in either layout.
argv: [shard_n] [queries] [batch]   (defaults 12_500_000 1_000_000 131072)"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, '.')
a_ = sys.argv[1:]
n = int(a_[0]) if len(a_) > 0 else 12_500_000
nq = int(a_[1]) if len(a_) > 1 else 1_000_000
batch = int(a_[2]) if len(a_) > 2 else 131072
os.environ["HS_OPTIONS"] = "query_batch=%d" % batch
import torch
from hsearch_amd import Engine, synth
k, K, L, W, R = 25, 20, 32, 160.0, 40.0
a, b = synth.make_planes(k, K, L, W)
codes = synth.make_db(n, k)                       # one rank's shard (i.i.d. k-mers: any 1/8 looks like this)
# queries: mutated k-mers; 1/8 of them from this shard (the others' sources live on the other shards)
own = nq // 8
qc_own, _ = synth.make_query_codes(codes, own, seed=synth.SEED_QUERIES)
qc_other, _ = synth.make_query_codes(synth.make_db(nq - own, k, seed=synth.SEED_DB + 5), nq - own, seed=synth.SEED_QUERIES + 5)
qcodes = np.concatenate([qc_own, qc_other])
np.random.default_rng(1).shuffle(qcodes)
centers = synth.embed(qcodes)
dev = torch.device("cuda", 0)
eng = Engine(k, K, L, W, a, b, device=0)
t0 = time.perf_counter(); eng.index_build(codes); t_build = time.perf_counter() - t0
d_centers = torch.from_numpy(centers).to(dev)
cap = 4 * nq + 4096
out = dict(q=torch.empty(cap, dtype=torch.int32, device=dev), id=torch.empty(cap, dtype=torch.int32, device=dev),
           table=torch.empty(cap, dtype=torch.int32, device=dev), dist=torch.empty(cap, dtype=torch.float64, device=dev))
def step():
    return eng.query_dev(d_centers.data_ptr(), nq, R, out["q"].data_ptr(), out["id"].data_ptr(),
                         out["table"].data_ptr(), out["dist"].data_ptr(), cap)
step()
torch.cuda.synchronize()
acc = {}
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    nh = step()
    p = eng.profile()
    for f in ("ms_hash", "ms_probe", "ms_verify", "ms_join", "ms_finalize", "ms_total"):
        acc[f] = acc.get(f, 0.0) + p[f] / steps
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"layout": "database sharded x8, emulated on one GPU", "shard_kmers": n, "queries": nq, "batch": batch,
                  "seconds_per_pass_over_all_queries": dt, "job_queries_per_s_8_gpus": nq / dt,
                  "device_ms": acc, "candidates_per_query_on_this_shard": p["candidates"] / nq,
                  "join_pairs": p["join_pairs"], "join_pairs_issued": p["join_pairs_issued"], "join_items": p["join_items"],
                  "hits": nh, "build_seconds": t_build}))
