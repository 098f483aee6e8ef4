"""One part (0 of 8) of the bucket-partitioned configs[2] job, for rocprofv3: index, then N passes over all 10^6 queries."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hsearch_amd import Engine, synth
k, K, L, W, R = 25, 20, 32, 160.0, 40.0
n, nq = 100_000_000, 1_000_000
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a, b = synth.make_planes(k, K, L, W)
codes = synth.make_db(n, k)
qcodes, _ = synth.make_query_codes(codes, nq, seed=synth.SEED_QUERIES)
dev = torch.device("cuda", 0)
d_centers = torch.from_numpy(synth.embed(qcodes)).to(dev)
cap = 4 * nq
out = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(3)] + [torch.empty(cap, dtype=torch.float64, device=dev)]
eng = Engine(k, K, L, W, a, b, device=0)
eng.index_build(codes)
eng.set_bucket_partition(0, parts)
for _ in range(10):
    eng.query_dev(d_centers.data_ptr(), nq, R, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), cap)
torch.cuda.synchronize()
print(eng.profile()["ms_total"], file=sys.stderr)
eng.close()
