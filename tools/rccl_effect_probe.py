"""Does an initialised RCCL communicator slow the join kernel down?  Times query steps (1) before
any torch.distributed use, (2) after init_process_group('nccl') + one all_reduce, (3) after
destroy_process_group.  Single process, world size 1."""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, '.')
from hsearch_amd import Engine, synth
k, K, L, W, R, n, nq = 25, 16, 8, 200.0, 40.0, 10_000_000, 100_000
a, b = synth.make_planes(k, K, L, W); codes = synth.make_db(n, k); centers, _ = synth.make_queries(codes, nq)
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
EARLY = os.environ.get("EARLY")
if EARLY:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if EARLY == "devid":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
    if EARLY != "noop":
        dist.barrier()
eng = Engine(k, K, L, W, a, b); eng.index_build(codes)
d_c = torch.from_numpy(centers).to(dev); cap = 64 * nq
q, idd, t, d = (torch.empty(cap, dtype=dt, device=dev) for dt in (torch.int32, torch.int32, torch.int32, torch.float64))
def run(steps, tag):
    for _ in range(5): eng.query_dev(d_c.data_ptr(), nq, R, q.data_ptr(), idd.data_ptr(), t.data_ptr(), d.data_ptr(), cap)
    torch.cuda.synchronize(); t0 = time.perf_counter(); jm = 0.0
    for _ in range(steps):
        eng.query_dev(d_c.data_ptr(), nq, R, q.data_ptr(), idd.data_ptr(), t.data_ptr(), d.data_ptr(), cap)
        jm += eng.profile()["ms_join"]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-28s %.2f ms/step, join %.2f ms" % (tag, dt / steps * 1e3, jm / steps), flush=True)
run(20, "first timing (EARLY=%s)" % EARLY)
if EARLY:
    dist.destroy_process_group(); run(20, "after destroy"); sys.exit(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
run(20, "process group created")
x = torch.ones(4, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
run(20, "after one all_reduce")
dist.destroy_process_group()
run(20, "after destroy_process_group")
