#!/bin/bash
# A/B of the probe + segment chain at the configs[2] shape: the 10^6-query batch with the probes grouped by a
# counting sort over the bucket slots (dense) and by a sort of the probes (sparse); 125 k queries for reference.
cd $GRAFT_REPO_ROOT
B="python bench.py --db-size 100000000 --L 32 --K 20 --W 160 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --pcie-steps 0 --recall-queries 0 --planted-members 0 --general-steps 0"
HS_OPTIONS=seg_mode=2 $B --queries 1000000 > gpurun_out/ab_chain_1M_dense.json 2>/dev/null
HS_OPTIONS=seg_mode=1 $B --queries 1000000 > gpurun_out/ab_chain_1M_sparse.json 2>/dev/null
$B --queries 125000 > gpurun_out/ab_chain_125k.json 2>/dev/null
python - <<'PY'
import json
for f in ("ab_chain_1M_dense","ab_chain_1M_sparse","ab_chain_125k"):
    p=json.load(open("gpurun_out/%s.json"%f))
    print(f, round(p["value"]/1e6,2), "Mq/s", round(p["ms_per_step"],2), "ms", {k:round(v,2) for k,v in p["phases_ms_per_step"].items()}, round(p["roofline"]["frac"],3))
PY
