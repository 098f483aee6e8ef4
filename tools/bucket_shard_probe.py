"""SURVEY 8(e) decision data: configs[2] on G GPUs with the BUCKETS shared among the ranks -- index replicated
(the north star's layout), every rank ALL 10^6 queries in one batch, rank r only the buckets whose key
fingerprint falls to part r of G (hs_set_bucket_partition) -- emulated on one GPU: the G passes one after the
other on ONE handle, each timed, the parts' lists merged by the first-seen rule (per (query, id) the smallest
table: motif_both_points.cpp:232-238) and compared hit for hit with the unpartitioned pass.  Beside it the
query-block layout's per-rank pass (10^6 / G queries) on the same handle.

argv: [--worlds 2,4,8] [--queries N] [--n N] [--steps S] [--out FILE]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--queries", type=int, default=1_000_000)
ap.add_argument("--n", type=int, default=100_000_000)
ap.add_argument("--L", type=int, default=32)
ap.add_argument("--K", type=int, default=20)
ap.add_argument("--W", type=float, default=160.0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--out", default=None)
args = ap.parse_args()

import torch
from hsearch_amd import Engine, synth
from hsearch_amd import dist as hdist

k, K, L, W, R = 25, args.K, args.L, args.W, 40.0
n, nq = args.n, args.queries
a, b = synth.make_planes(k, K, L, W)
codes = synth.make_db(n, k)
qcodes, _ = synth.make_query_codes(codes, nq, seed=synth.SEED_QUERIES)
centers = synth.embed(qcodes)
dev = torch.device("cuda", 0)
d_centers = torch.from_numpy(centers).to(dev)
cap = 4 * nq + 4096
out = dict(q=torch.empty(cap, dtype=torch.int32, device=dev), id=torch.empty(cap, dtype=torch.int32, device=dev),
           table=torch.empty(cap, dtype=torch.int32, device=dev), dist=torch.empty(cap, dtype=torch.float64, device=dev))
eng = Engine(k, K, L, W, a, b, device=0)
info = eng.index_build(codes)


def run(nq_, steps, offset=0):
    ptr = d_centers.data_ptr() + offset * centers.shape[1] * 8

    def step():
        return eng.query_dev(ptr, nq_, R, out["q"].data_ptr(), out["id"].data_ptr(), out["table"].data_ptr(),
                             out["dist"].data_ptr(), cap)
    step()
    step()
    torch.cuda.synchronize()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        nh = step()
        p = eng.profile()
        for f in ("ms_hash", "ms_probe", "ms_join", "ms_finalize", "ms_total"):
            acc[f] = acc.get(f, 0.0) + p[f] / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return nh, dict(seconds_per_pass=dt, device_ms=acc, join_pairs=p["join_pairs"], join_items=p["join_items"],
                    join_items_resident=p["join_items_resident"], hits=nh)


result = {"layout": "buckets shared among the ranks (hs_set_bucket_partition), index replicated, all queries on every "
                    "rank; emulated on one GPU, one handle",
          "db_kmers": n, "queries": nq, "L": L, "K": K, "W": W, "index_bytes": info["device_bytes"]}
nh_full, full = run(nq, args.steps)
result["one_gpu_all_queries"] = full
want = tuple(out[f][:nh_full].clone() for f in ("q", "id", "table", "dist"))
for G in [int(x) for x in args.worlds.split(",")]:
    per_rank, parts = [], []
    for r in range(G):
        eng.set_bucket_partition(r, G)
        nh, res = run(nq, args.steps)
        res["rank"] = r
        per_rank.append(res)
        parts.append(tuple(out[f][:nh].clone() for f in ("q", "id", "table", "dist")))
    eng.set_bucket_partition(0, 1)
    gq, gi, gt, gd = (torch.cat([p[j] for p in parts]) for j in range(4))
    t_merge = None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mq, mid, mt, md = hdist.merge_table_partitioned(gq, gi, gt, gd)
        torch.cuda.synchronize()
        t_merge = time.perf_counter() - t0
    same = (len(mq) == nh_full and bool((mq == want[0].to(torch.int64)).all()) and bool((mid == want[1].to(torch.int64)).all())
            and bool((mt == want[2].to(torch.int64)).all()) and bool((md == want[3]).all()))
    # the query-block layout's per-rank pass on the same handle (rank 0's block)
    nhb, blocks = run(nq // G, args.steps)
    slowest = max(r_["seconds_per_pass"] for r_ in per_rank)
    result["world_%d" % G] = {
        "per_rank": per_rank, "slowest_rank_seconds_per_pass": slowest,
        "fastest_rank_seconds_per_pass": min(r_["seconds_per_pass"] for r_ in per_rank),
        "hits_gathered": int(len(gq)), "hits_after_merge": int(len(mq)), "merge_seconds_on_one_gpu": t_merge,
        "merged_equals_unpartitioned_hit_for_hit": same,
        "job_queries_per_s_before_exchange": nq / slowest, "job_queries_per_s_with_merge": nq / (slowest + t_merge),
        "query_blocks_same_handle": blocks, "job_queries_per_s_query_blocks": nq / blocks["seconds_per_pass"]}
    print(G, "ranks: buckets", round(nq / slowest / 1e6, 2), "M q/s (slowest", round(slowest * 1e3, 2), "ms, fastest",
          round(result["world_%d" % G]["fastest_rank_seconds_per_pass"] * 1e3, 2), "), query blocks",
          round(nq / blocks["seconds_per_pass"] / 1e6, 2), "M q/s; merged == unpartitioned:", same, file=sys.stderr, flush=True)
eng.close()
text = json.dumps(result)
print(text)
if args.out:
    open(args.out, "w").write(text + "\n")
