#!/bin/bash
# rocprofv3 kernel statistics of the index build (run on the GPU box): usage tools/build_profile.sh <tag> <n> <L> <K> <W>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/buildprof_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/build_once.py $2 $3 $4 $5 > $OUT/run.log 2>&1 || echo "profile run failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + '/*/*_kernel_trace.csv'):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][-60:]
        agg[k][0] += 1
        agg[k][1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in agg.values())
with open(out + '/summary.txt', 'w') as fo:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        line = "%9.3f ms %6d x  %5.1f%%  %s" % (v[1], v[0], 100 * v[1] / tot, k)
        print(line); fo.write(line + "\n")
PY
