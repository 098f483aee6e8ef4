"""SURVEY 8(e) decision data: configs[2] with the TABLES sharded over 8 GPUs instead of
the index replicated, emulated on one GPU.  Replicated (the north star's layout; bench.py's secondary block):
every GPU holds all 32 tables of all 10^8 k-mers and searches 1/8 of the 10^6 queries.  Table-sharded: GPU r
holds tables 4r .. 4r+3 of ALL 10^8 k-mers and searches ALL 10^6 queries in one batch; per GPU the same number
of probes (10^6 x 4 = 125 k x 32) and of (member, query) pairs, 8 x the queries per segment with full-size
buckets, 1/8 of the table bytes and of the build.  A hit (q, id) found by several ranks is reported at the
SMALLEST global table: the reference reports an id in the first table whose bucket holds it
(motif_both_points.cpp:232-238), whatever the other tables say -- so the merge is: all-gather, per (q, id) the
minimum table, order by (q, table, id).  This script times every rank's pass, merges on the device and -- with
--check -- compares the merged list with the replicated 32-table handle's output hit for hit.

argv: [--ranks 0,7 | all] [--queries N] [--n N] [--check] [--steps S] [--world G]"""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, '.')
ap = argparse.ArgumentParser()
ap.add_argument("--ranks", default="0")
ap.add_argument("--queries", type=int, default=1_000_000)
ap.add_argument("--n", type=int, default=100_000_000)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--L", type=int, default=32)
ap.add_argument("--K", type=int, default=20)
ap.add_argument("--W", type=float, default=160.0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--check", action="store_true")
ap.add_argument("--balance", action="store_true",
                help="deal the tables by estimated join work (dist.table_costs + hs_assign_tables) instead of in blocks")
ap.add_argument("--out", default=None)
args = ap.parse_args()

import torch
from hsearch_amd import Engine, synth
from hsearch_amd import dist as hdist

k, K, L, W, R, G = 25, args.K, args.L, args.W, 40.0, args.world
n, nq = args.n, args.queries
assert L % G == 0
Lr = L // G
a, b = synth.make_planes(k, K, L, W)
codes = synth.make_db(n, k)
qcodes, _ = synth.make_query_codes(codes, nq, seed=synth.SEED_QUERIES)
centers = synth.embed(qcodes)
dev = torch.device("cuda", 0)
d_centers = torch.from_numpy(centers).to(dev)
cap = 4 * nq + 4096


def alloc(c):
    return dict(q=torch.empty(c, dtype=torch.int32, device=dev), id=torch.empty(c, dtype=torch.int32, device=dev),
                table=torch.empty(c, dtype=torch.int32, device=dev), dist=torch.empty(c, dtype=torch.float64, device=dev))


def run(eng, out, nq_, steps):
    def step():
        return eng.query_dev(d_centers.data_ptr(), nq_, R, out["q"].data_ptr(), out["id"].data_ptr(),
                             out["table"].data_ptr(), out["dist"].data_ptr(), cap)
    step()
    step()
    torch.cuda.synchronize()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        nh = step()
        p = eng.profile()
        for f in ("ms_hash", "ms_probe", "ms_verify", "ms_join", "ms_finalize", "ms_total"):
            acc[f] = acc.get(f, 0.0) + p[f] / steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    frac = p["join_pairs"] * 256.0 / (acc["ms_join"] * 1e-3) / 5e15 if acc["ms_join"] > 0 else 0.0
    return nh, dict(seconds_per_pass=dt, device_ms=acc, join_frac_of_int8_peak=frac,
                    issued_over_useful=p["join_pairs_issued"] / max(p["join_pairs"], 1),
                    join_pairs=p["join_pairs"], join_items=p["join_items"], join_items_resident=p["join_items_resident"],
                    candidates_per_query=p["candidates"] / nq_, hits=nh)


ranks = list(range(G)) if args.ranks == "all" else [int(x) for x in args.ranks.split(",")]
if args.balance:
    cost = hdist.table_costs(k, K, L, W, a, b, codes[:32768], device=0)
    tabs = hdist.assign_tables(cost, L, G)
else:
    cost = None
    tabs = [np.arange(r * Lr, (r + 1) * Lr) for r in range(G)]
per_rank, parts = [], []
for r in ranks:
    eng = Engine(k, K, len(tabs[r]), W, a[tabs[r]], b[tabs[r]], device=0)
    t0 = time.perf_counter()
    info = eng.index_build(codes)
    t_build = time.perf_counter() - t0
    bp = eng.profile()
    out = alloc(cap)
    nh, res = run(eng, out, nq, args.steps)
    res.update(rank=r, tables=[int(x) for x in tabs[r]], build_seconds=t_build, build_device_ms=bp["ms_total"],
               index_bytes=info["device_bytes"],
               estimated_cost_share=float(cost[tabs[r]].sum() / cost.sum()) if cost is not None else None)
    per_rank.append(res)
    tmap = torch.as_tensor(np.asarray(tabs[r], dtype=np.int64), device=dev)
    parts.append((out["q"][:nh].clone(), out["id"][:nh].clone(), tmap[out["table"][:nh].to(torch.int64)], out["dist"][:nh].clone()))
    eng.close()
    del eng, out
    torch.cuda.empty_cache()
    print(json.dumps(res), file=sys.stderr, flush=True)

# the merge, as every rank would run it on the gathered lists: torch (bench.py's path) and the library's own
# (hs_merge_first_table_dev, the C++ host's path); the second call of each is the warm figure
gq, gi, gt, gd = (torch.cat([p[j] for p in parts]) for j in range(4))
t_merge = None
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mq, mid, mt, md = hdist.merge_table_partitioned(gq.to(torch.int64), gi.to(torch.int64), gt.to(torch.int64), gd)
    torch.cuda.synchronize()
    t_merge = time.perf_counter() - t0
eng_m = Engine(k, K, 1, W, a[:1], b[:1], device=0)
t_merge_lib = None
for _ in range(2):
    cq, ci, ct, cd = gq.to(torch.int32).clone(), gi.to(torch.int32).clone(), gt.to(torch.int32).clone(), gd.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kept = eng_m.merge_first_table_dev(cq.data_ptr(), ci.data_ptr(), ct.data_ptr(), cd.data_ptr(), len(cq))
    t_merge_lib = time.perf_counter() - t0
lib_equal = (kept == len(mq) and bool((cq[:kept].to(torch.int64) == mq).all()) and bool((ci[:kept].to(torch.int64) == mid).all())
             and bool((ct[:kept].to(torch.int64) == mt).all()) and bool((cd[:kept] == md).all()))
eng_m.close()
gathered = int(sum(len(p[0]) for p in parts))
slowest = max(r["seconds_per_pass"] for r in per_rank)
result = {"layout": "tables sharded x%d (rank r: tables %dr .. %dr+%d of all k-mers, all queries), emulated on one GPU"
                    % (G, Lr, Lr, Lr - 1),
          "db_kmers": n, "queries": nq, "L": L, "K": K, "W": W, "ranks_measured": ranks, "per_rank": per_rank,
          "slowest_rank_seconds_per_pass": slowest, "hits_gathered_from_measured_ranks": gathered,
          "hits_after_merge": int(len(mq)), "merge_seconds_on_one_gpu": t_merge,
          "merge_seconds_hs_merge_first_table_dev": t_merge_lib, "library_merge_equals_torch_merge": lib_equal,
          "tables_dealt_by_estimated_cost": bool(args.balance),
          "job_queries_per_s_%d_gpus_before_exchange" % G: nq / slowest,
          "job_queries_per_s_%d_gpus_with_merge" % G: nq / (slowest + t_merge)}
if args.check:
    assert len(ranks) == G, "--check needs every rank"
    eng = Engine(k, K, L, W, a, b, device=0)
    t0 = time.perf_counter()
    eng.index_build(codes)
    out = alloc(cap)
    nh, res = run(eng, out, nq, args.steps)
    result["replicated_one_gpu_all_queries"] = res
    same = (nh == len(mq) and bool((out["q"][:nh].to(torch.int64) == mq).all()) and
            bool((out["id"][:nh].to(torch.int64) == mid).all()) and
            bool((out["table"][:nh].to(torch.int64) == mt).all()) and bool((out["dist"][:nh] == md).all()))
    result["merged_equals_replicated_hit_for_hit"] = same
    # ... and the replicated layout's per-GPU share of the 8-GPU job, same handle
    nh8, res8 = run(eng, out, nq // G, args.steps)
    result["replicated_share_of_%d" % G] = res8
    result["job_queries_per_s_%d_gpus_replicated" % G] = nq / G / res8["seconds_per_pass"] * G
    eng.close()
text = json.dumps(result)
print(text)
if args.out:
    open(args.out, "w").write(text + "\n")
