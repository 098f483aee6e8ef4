#!/bin/bash
# A/B at the configs[2] shape: where do segments with ONE probing query go?
ARGS="--db-size 100000000 --L 32 --K 20 --W 160 --queries 125000 --pcie-steps 0 --no-secondary --no-cpu-baseline --recall-queries 0 --planted-members 0 --steps 8 --warmup 2"
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py $ARGS > gpurun_out/c3ab_$tag.json 2> gpurun_out/c3ab_$tag.err; python - "$tag" <<'PY'
import json,sys
d=json.load(open("gpurun_out/c3ab_%s.json"%sys.argv[1])); r=d["roofline"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), "join", round(r["kernel_ms_per_step"],2), "stream", round(r["streaming_kernel_ms_per_step"],2), d["phases_ms_per_step"], "hits", d["hits_per_step_rank0"], "streamed pairs", r["pairs_streamed_per_step"])
PY
}
run base HS_X=0
run norec HS_NO_RECOGNISE=1
run minq2_stream HS_JOIN_MIN_Q=2 HS_NO_THIN8=1 HS_NO_RECOGNISE=1
run minq2_thin HS_JOIN_MIN_Q=2
