#!/bin/bash
# PMC passes over an index build at the configs[2] shape with 2 tables (run on the GPU box): separate rocprofv3
# runs per counter group.  usage: tools/pmc_build.sh <tag> "<group 1>" "<group 2>" ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/build_c3.py 2 > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_kernels.py $OUT group_insert group_check4 gather_rec8 invert_perm
