"""HBM traffic of the secondary block's join kernels (configs[2] shape, 10^6 queries in one batch) from the
FETCH_SIZE / WRITE_SIZE passes of tools/pmc_join.sh run with HS_BENCH_ARGS of that shape: bytes per step summed
over hs_join8x_kernel and hs_join8r_kernel (one launch of each per step), written to
profiles/traffic_secondary.json, which bench.py reports as secondary.roofline.traffic while its kernel source
hash, queries per GPU and W are the running build's.  FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md section
HBM (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B); WRITE_SIZE (KB) as is.
Usage: python tools/pmc_secondary.py gpurun_out/pmc_<tag> <queries_per_gpu> <W>"""
import collections, csv, datetime, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("hs_join8.hip", "hs_join.hip", "hs_kernels.hip", "hs_internal.h"):
        h.update(open(os.path.join(ROOT, "hsearch_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


src, nq, W = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
kernels = ("hs_join8x_kernel", "hs_join8r_kernel")
tot = {k: collections.defaultdict(float) for k in kernels}
disp = {k: collections.defaultdict(set) for k in kernels}
for f in glob.glob(src + '/p*/*/*_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        for k in kernels:
            if k in row['Kernel_Name']:
                tot[k][row['Counter_Name']] += float(row['Counter_Value'])
                disp[k][row['Counter_Name']].add(row['Dispatch_Id'])
per = {}
for k in kernels:
    n = max(1, len(disp[k]["FETCH_SIZE"]))
    per[k] = {"launches": len(disp[k]["FETCH_SIZE"]), "fetch_size_kb_per_launch": tot[k]["FETCH_SIZE"] / n,
              "write_size_kb_per_launch": tot[k].get("WRITE_SIZE", 0.0) / max(1, len(disp[k]["WRITE_SIZE"])),
              "bytes_per_launch": (2.0 * tot[k]["FETCH_SIZE"] / n + tot[k].get("WRITE_SIZE", 0.0) / max(1, len(disp[k]["WRITE_SIZE"]))) * 1024.0}
res = {"kernels": list(kernels), "kernel_source_hash": kernel_source_hash(), "taken": datetime.date.today().isoformat(),
       "queries_per_gpu": nq, "W": W, "per_kernel": per,
       "verify_bytes_per_launch": sum(p["bytes_per_launch"] for p in per.values()),
       "method": __doc__.split("Usage")[0].strip()}
json.dump(res, open(os.path.join(ROOT, "profiles", "traffic_secondary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "method"}, indent=1))
