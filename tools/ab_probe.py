"""A/B of HS_OPT_PROBE_RECORDS (the probe reads one 64-byte directory record per bucket | the directory arrays) at
the configs[2] shape (10^8 x 25-mers, L = 32, K = 20, W = 160), 10^6 and 125 k queries, and at C2; prints the
probe + grouping phase per pass and checks that the hits do not change.  argv: [--shape c3|c2|both] [--steps S]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="both")
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--out", default=None)
args = ap.parse_args()
import torch
from hsearch_amd import Engine, synth

dev = torch.device("cuda", 0)
R, k = 40.0, 25
result = {}


def shape(name, n, L, K, W, batches):
    a, b = synth.make_planes(k, K, L, W)
    codes = synth.make_db(n, k)
    eng = Engine(k, K, L, W, a, b, device=0)
    info = eng.index_build(codes)
    result[name + "_index_bytes"] = info["device_bytes"]
    for nq in batches:
        qcodes, _ = synth.make_query_codes(codes, nq, seed=synth.SEED_QUERIES)
        d_centers = torch.from_numpy(synth.embed(qcodes)).to(dev)
        cap = 4 * nq + 4096
        out = dict(q=torch.empty(cap, dtype=torch.int32, device=dev), id=torch.empty(cap, dtype=torch.int32, device=dev),
                   table=torch.empty(cap, dtype=torch.int32, device=dev), dist=torch.empty(cap, dtype=torch.float64, device=dev))
        first = None
        for rec in (1, 0, 1, 0):
            eng.set_option("probe_records", rec)
            acc = {}
            for i in range(args.steps + 2):
                nh = eng.query_dev(d_centers.data_ptr(), nq, R, out["q"].data_ptr(), out["id"].data_ptr(),
                                   out["table"].data_ptr(), out["dist"].data_ptr(), cap)
                if i >= 2:
                    p = eng.profile()
                    for f in ("ms_hash", "ms_probe", "ms_join", "ms_total"):
                        acc[f] = acc.get(f, 0.0) + p[f] / args.steps
            got = tuple(out[f][:nh].clone() for f in ("q", "id", "table", "dist"))
            first = first or got
            acc["hits"] = nh
            acc["equal"] = all(len(x) == len(y) and bool((x == y).all()) for x, y in zip(got, first))
            result.setdefault("%s_%dq_records%d" % (name, nq, rec), []).append(acc)
            print(name, nq, "records", rec, {f: round(v, 3) if isinstance(v, float) else v for f, v in acc.items()},
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
