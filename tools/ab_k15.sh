#!/bin/bash
# same-box A/B of library builds on the k = 15 leg of configs[4]: tools/ab_k15.sh <libA.so> <libB.so> ...
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so)_$round
    HSEARCH_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 python tools/bench_mixed_k.py --ks 15 > gpurun_out/abk15_$tag.json 2> gpurun_out/abk15_$tag.err
    python - "$tag" <<'PY'
import json,sys
d=json.load(open("gpurun_out/abk15_%s.json"%sys.argv[1])); v=d["per_length"]["15"]
print(sys.argv[1], round(d["value"]), "device ms", round(v["device_ms_per_step"],2), "join", round(v["join_ms_per_step"],2), "hits", v["hits_per_step"])
PY
  done
done
