"""Feasibility probe: does one query batch's pre-join phase (hash, probe, segment sort) overlap with
another batch's join kernel on the same GPU?  Two handles (two indexes, two streams), two host
threads calling hs_query_dev in a loop, against one handle doing the same number of batches."""
import sys, threading, time
import numpy as np, torch
sys.path.insert(0, '.')
from hsearch_amd import Engine, synth
k, K, L, W, R, n, nq = 25, 16, 8, 200.0, 40.0, 10_000_000, 100_000
a, b = synth.make_planes(k, K, L, W); codes = synth.make_db(n, k); centers, _ = synth.make_queries(codes, nq)
dev = torch.device("cuda:0")
engs = [Engine(k, K, L, W, a, b) for _ in range(2)]
for e in engs: e.index_build(codes)
d_c = torch.from_numpy(centers).to(dev)
cap = 64 * nq
outs = [tuple(torch.empty(cap, dtype=dt, device=dev) for dt in (torch.int32, torch.int32, torch.int32, torch.float64)) for _ in range(2)]
def run(i, steps):
    q, idd, t, d = outs[i]
    for _ in range(steps):
        engs[i].query_dev(d_c.data_ptr(), nq, R, q.data_ptr(), idd.data_ptr(), t.data_ptr(), d.data_ptr(), cap)
run(0, 2); run(1, 2); torch.cuda.synchronize()
steps = 10
t0 = time.perf_counter(); run(0, 2 * steps); torch.cuda.synchronize(); t1 = time.perf_counter()
print("one handle : %.2f ms/batch, %.2f M q/s" % ((t1 - t0) / (2 * steps) * 1e3, 2 * steps * nq / (t1 - t0) / 1e6))
th = [threading.Thread(target=run, args=(i, steps)) for i in range(2)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("two handles: %.2f ms/batch, %.2f M q/s" % ((t1 - t0) / (2 * steps) * 1e3, 2 * steps * nq / (t1 - t0) / 1e6))
