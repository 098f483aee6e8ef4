"""Which first-pass path of a 10^6-query batch at the configs[2] shape loses hits (the full count is 911 306)?
Each trial: a 1000-query batch (leaves a small item-capacity hint: the big batch's asynchronous attempt then fails
and the batch runs again synchronously), then the 10^6-query batch; per option set the hit counts of N trials."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hsearch_amd import Engine, synth
k, K, L, W, R = 25, 20, 32, 160.0, 40.0
n, nq_all = 100_000_000, 1_000_000
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a, b = synth.make_planes(k, K, L, W)
codes = synth.make_db(n, k)
qcodes_all, _ = synth.make_query_codes(codes, nq_all)
eng = Engine(k, K, L, W, a, b)
eng.index_build(codes)
base = dict(sync_items=0, join_xcd_run=-1, probe_records=1, join_resident=0, refine8=1)
variants = [("default", {}), ("sync_items", dict(sync_items=1)), ("no xcd runs", dict(join_xcd_run=0)), ("xcd runs of 1", dict(join_xcd_run=1)),
            ("xcd runs of 1024", dict(join_xcd_run=1024)), ("no resident kernel", dict(join_resident=1)), ("default again", {})]
for name, opts in variants:
    for o, v in {**base, **opts}.items():
        eng.set_option(o, v)
    counts, retries = [], []
    for t in range(trials):
        if not os.environ.get("NO_SMALL"): eng.query_codes(qcodes_all[:1000], R, want_cand=False)
        big = eng.query_codes(qcodes_all, R, cap=4 * nq_all, want_cand=False)
        counts.append(len(big["q"]))
        retries.append(eng.profile()["join_async_retries"])
    print("%-20s hits %s  async retries %s" % (name, counts, retries), file=sys.stderr, flush=True)
eng.close()
