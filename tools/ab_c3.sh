#!/bin/bash
# same-box A/B of library builds at the configs[2] shape: tools/ab_c3.sh <libA.so> <libB.so> ... (paths in the repo)
ARGS="--db-size 100000000 --L 32 --K 20 --W 160 --queries 125000 --pcie-steps 0 --no-secondary --no-cpu-baseline --recall-queries 0 --planted-members 0 --steps 8 --warmup 2"
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so)_$round
    HSEARCH_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python bench.py $ARGS > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
    python - "$tag" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_%s.json"%sys.argv[1])); r=d["roofline"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), "join", round(r["kernel_ms_per_step"],2), "hits", d["hits_per_step_rank0"])
PY
  done
done
