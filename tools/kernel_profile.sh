#!/bin/bash
# rocprofv3 kernel statistics of any of the repo's python programs (run on the GPU box):
# usage tools/kernel_profile.sh <tag> <script and its arguments ...>; summary in gpurun_out/kprof_<tag>/summary.txt
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/kprof_$1
shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1 || echo "profile run failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + '/*/*_kernel_trace.csv'):
    for row in csv.DictReader(open(f)):
        name = row['Kernel_Name']
        m = re.search(r'(hs_\w+(<[^>(]*>)?)', name)
        k = m.group(1) if m else name[:70]
        agg[k][0] += 1
        agg[k][1] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in agg.values())
with open(out + '/summary.txt', 'w') as fo:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        line = "%9.3f ms %6d x  %8.4f ms each %5.1f%%  %s" % (v[1], v[0], v[1] / v[0], 100 * v[1] / tot, k)
        print(line); fo.write(line + "\n")
PY
