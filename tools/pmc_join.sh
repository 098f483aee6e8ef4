#!/bin/bash
# PMC passes over bench.py (run on the GPU box): separate rocprofv3 runs per counter group.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --recall-queries 0 --planted-members 0 --no-secondary --pcie-steps 0 --general-steps 0 $HS_BENCH_ARGS > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
ls -R $OUT | head -40
