#!/bin/bash
# PMC passes over the k = 15 leg of configs[4] (run on the GPU box): usage tools/pmc_k15.sh <tag> "<group 1>" ...
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_mixed_k.py --ks 15 --steps 1 --warmup 0 > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
