"""Per-kernel sums of the rocprofv3 --pmc passes under gpurun_out/pmc_<tag> (tools/pmc_join.sh), and the
kernels' durations from the kernel trace of the first pass.  Usage: python tools/pmc_kernels.py <dir> [substr ...]"""
import collections, csv, glob, sys
src = sys.argv[1]
subs = sys.argv[2:] or ["join8r", "join8x"]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(src + '/p*/*/*_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        for s in subs:
            if s in row['Kernel_Name']:
                tot[s][row['Counter_Name']] += float(row['Counter_Value'])
                nd[s][row['Counter_Name']].add(row['Dispatch_Id'])
for s in subs:
    print("==", s)
    for c, v in sorted(tot[s].items()):
        print("  %-28s %.4g per launch (%d launches)" % (c, v / max(1, len(nd[s][c])), len(nd[s][c])))
for f in sorted(glob.glob(src + '/p1/*/*_kernel_trace.csv')):
    for row in csv.DictReader(open(f)):
        if any(s in row['Kernel_Name'] for s in subs):
            print("%-44s %.3f ms" % (row['Kernel_Name'][:44], (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6))
