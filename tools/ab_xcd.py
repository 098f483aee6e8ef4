"""A/B of HS_OPT_JOIN_XCD_RUN (hs_join8x_kernel's work items dealt in XCD-local runs) on ONE index per shape:
configs[2] (10^8 x 25-mers, L = 32, K = 20, W = 160) with 10^6 and 125 k queries per batch, and the bench
default (configs[1]: 10^7, L = 8, K = 16, W = 212, 10^5 queries).  Prints the join time per pass for each
run length; the hits of every setting are compared with the first one's.

argv: [--shape c3|c2|both] [--runs 0,8,32,128,512] [--steps S] [--out FILE]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="both")
ap.add_argument("--runs", default="0,8,32,128,512")
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--out", default=None)
args = ap.parse_args()

import torch
from hsearch_amd import Engine, synth

dev = torch.device("cuda", 0)
runs = [int(x) for x in args.runs.split(",")]
R, k = 40.0, 25
result = {}


def measure(eng, d_centers, nq, out, cap):
    def step():
        return eng.query_dev(d_centers.data_ptr(), nq, R, out["q"].data_ptr(), out["id"].data_ptr(),
                             out["table"].data_ptr(), out["dist"].data_ptr(), cap)
    step()
    step()
    acc = {}
    for _ in range(args.steps):
        nh = step()
        p = eng.profile()
        for f in ("ms_probe", "ms_join", "ms_verify", "ms_total"):
            acc[f] = acc.get(f, 0.0) + p[f] / args.steps
    acc["hits"] = nh
    acc["join_items"] = p["join_items"]
    acc["join_items_resident"] = p["join_items_resident"]
    return acc


def shape(name, n, L, K, W, batches):
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(n, k)
    eng = Engine(k, K, L, W, a, b, device=0)
    eng.index_build(codes)
    for nq in batches:
        qcodes, _ = synth.make_query_codes(codes, nq, seed=synth.SEED_QUERIES)
        d_centers = torch.from_numpy(synth.embed(qcodes)).to(dev)
        cap = 4 * nq + 4096
        out = dict(q=torch.empty(cap, dtype=torch.int32, device=dev), id=torch.empty(cap, dtype=torch.int32, device=dev),
                   table=torch.empty(cap, dtype=torch.int32, device=dev), dist=torch.empty(cap, dtype=torch.float64, device=dev))
        first = None
        for r in runs:
            eng.set_option("join_xcd_run", r)
            m = measure(eng, d_centers, nq, out, cap)
            nh = m["hits"]
            got = (out["q"][:nh].clone(), out["id"][:nh].clone(), out["table"][:nh].clone(), out["dist"][:nh].clone())
            if first is None:
                first = got
            m["equal_to_first_setting"] = all(len(x) == len(y) and bool((x == y).all()) for x, y in zip(got, first))
            result["%s_%dq_run%d" % (name, nq, r)] = m
            print(name, nq, "run", r, {f: round(v, 3) if isinstance(v, float) else v for f, v in m.items()},
                  file=sys.stderr, flush=True)
        del out, d_centers
    eng.close()
    del eng, codes
    torch.cuda.empty_cache()


if args.shape in ("c2", "both"):
    shape("c2", 10_000_000, 8, 16, 212.0, [100_000])
if args.shape in ("c3", "both"):
    shape("c3", 100_000_000, 32, 20, 160.0, [1_000_000, 125_000])
text = json.dumps(result)
print(text)
if args.out:
    open(args.out, "w").write(text + "\n")
