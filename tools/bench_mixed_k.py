#!/usr/bin/env python3
"""BASELINE.json configs[4] as ONE measurement: three databases of the C2 size with k = 15 / 25 / 39
(same seeds, same planes generator, W and R of the bench default), three resident indexes on one
GPU, a step = one pass of the query hot path over each length's batch of queries back to back
(one launch sequence on the library's stream per handle, no host work in between but the hit
counts).  Prints one JSON line: aggregate queries/s over the three lengths and the per-length
split.  Not the driver's contract (bench.py is); run on the GPU box."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="15,25,39")
    ap.add_argument("--db-size", dest="n", type=int, default=10_000_000)
    ap.add_argument("--queries", dest="nq", type=int, default=100_000)
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--L", type=int, default=8)
    ap.add_argument("--W", type=float, default=212.0)
    ap.add_argument("--R", type=float, default=40.0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--general-steps", type=int, default=3,
                    help="passes per length with centres that are NOT k-mers (jittered: the reference's family "
                         "centroids' path, hs_finalize_kernel instead of hs_finalize_codes_kernel); 0: skip")
    ap.add_argument("--general-jitter", type=float, default=0.05)
    args = ap.parse_args()
    import torch
    from hsearch_amd import Engine, HsError, synth
    dev = torch.device("cuda", 0)
    ks = [int(x) for x in args.ks.split(",")]
    engines = []
    for k in ks:
        a, b = synth.make_planes(k, args.K, args.L, args.W)
        codes = synth.make_db(args.n, k)
        centers, _ = synth.make_queries(codes, args.nq, seed=synth.SEED_QUERIES)
        eng = Engine(k, args.K, args.L, args.W, a, b, device=0)
        t0 = time.perf_counter()
        eng.index_build(codes)
        tb = time.perf_counter() - t0
        dc = torch.from_numpy(centers).to(dev)
        dg = None
        if args.general_steps:
            general, _ = synth.make_queries(codes, args.nq, seed=synth.SEED_QUERIES, jitter=args.general_jitter)
            dg = torch.from_numpy(general).to(dev)
        engines.append({"k": k, "eng": eng, "centers": dc, "general": dg, "cap": 0, "out": None, "build_s": tb})
        del codes

    def alloc(c):
        return dict(q=torch.empty(c, dtype=torch.int32, device=dev), id=torch.empty(c, dtype=torch.int32, device=dev),
                    table=torch.empty(c, dtype=torch.int32, device=dev), dist=torch.empty(c, dtype=torch.float64, device=dev))

    def one(e, which="centers"):
        if e["out"] is None:
            e["cap"] = 16 * args.nq + 4096
            e["out"] = alloc(e["cap"])
        while True:
            o = e["out"]
            try:
                return e["eng"].query_dev(e[which].data_ptr(), args.nq, args.R, o["q"].data_ptr(),
                                          o["id"].data_ptr(), o["table"].data_ptr(), o["dist"].data_ptr(), e["cap"])
            except HsError as err:
                if getattr(err, "needed", 0) <= e["cap"]:
                    raise
                e["cap"] = int(err.needed * 1.25) + 1024
                e["out"] = alloc(e["cap"])

    for _ in range(args.warmup):
        for e in engines:
            one(e)
    per = {e["k"]: {"ms": 0.0, "join_ms": 0.0, "hits": 0, "cand": 0} for e in engines}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for e in engines:
            nh = one(e)
            p = e["eng"].profile()
            d = per[e["k"]]
            d["ms"] += p["ms_total"]
            d["join_ms"] += p["ms_join"]
            d["hits"] = int(nh)
            d["cand"] = int(p["candidates"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_step = dt / args.steps * 1e3
    general = None
    if args.general_steps:
        gper = {}
        for e in engines:
            one(e, "general")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.general_steps):
            for e in engines:
                nh = one(e, "general")
                p = e["eng"].profile()
                d = gper.setdefault(e["k"], {"ms": 0.0, "join_ms": 0.0, "fin_ms": 0.0, "hits": 0, "rec": 0})
                d["ms"] += p["ms_total"]
                d["join_ms"] += p["ms_join"]
                d["fin_ms"] += p["ms_finalize"]
                d["hits"] = int(nh)
                d["rec"] = int(p["queries_recognised"])
        torch.cuda.synchronize()
        gdt = time.perf_counter() - t0
        general = {"what": "the same step with centres that are no k-mers (every coordinate jittered by N(0, %g)): "
                           "embedded rows, hs_finalize_kernel" % args.general_jitter,
                   "value": len(ks) * args.nq * args.general_steps / gdt, "unit": "queries/s",
                   "ms_per_step": gdt / args.general_steps * 1e3, "steps": args.general_steps,
                   "per_length": {str(k): {"device_ms_per_step": v["ms"] / args.general_steps,
                                           "join_ms_per_step": v["join_ms"] / args.general_steps,
                                           "finalize_ms_per_step": v["fin_ms"] / args.general_steps,
                                           "hits_per_step": v["hits"], "queries_recognised_as_kmers": v["rec"]}
                                  for k, v in gper.items()}}
    line = {
        "metric": "motif queries/sec (LSH probe + verify, index resident in HBM), mixed k-mer lengths",
        "value": len(ks) * args.nq * args.steps / dt, "unit": "queries/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "data": "synthetic",
        "config": {"workload": "configs[4]: %d x k-mers each of k in %s, L=%d K=%d W=%g R=%g, %d queries per length, "
                               "three indexes resident on one GPU, one pass over each per step"
                               % (args.n, ks, args.L, args.K, args.W, args.R, args.nq)},
        "per_length": {str(k): {"device_ms_per_step": v["ms"] / args.steps, "join_ms_per_step": v["join_ms"] / args.steps,
                                "queries_per_s_alone": args.nq / (v["ms"] / args.steps * 1e-3),
                                "hits_per_step": v["hits"], "candidates_per_query": v["cand"] / args.nq}
                       for k, v in per.items()},
        "index_build_seconds": {str(e["k"]): e["build_s"] for e in engines},
        "general_centres": general,
    }
    print(json.dumps(line))
    for e in engines:
        e["eng"].close()


if __name__ == "__main__":
    main()
