"""Regime sweep: the query path away from the headline configuration.

For a grid of (k, R, W, K, L) at a moderate DB size: hits of the default path (verify mode auto) must
equal the streaming filter's (two independent filters in front of the same exact decision), and the
timings show where a filter stops being selective (survivors per candidate) or a path is slower than
its alternative.  Prints one line per point and a JSON summary; run on the GPU box."""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hsearch_amd import Engine, synth

n = int(os.environ.get("SWEEP_N", 2_000_000))
nq = int(os.environ.get("SWEEP_NQ", 20_000))
grid = []
for k in (8, 12, 15, 18, 20, 21, 23, 25, 30, 39, 50):
    grid.append((k, 40.0, 212.0, 16, 8))
for R in (20.0, 30.0, 45.0, 50.0, 60.0):
    grid.append((25, R, 212.0, 16, 8))
for W in (100.0, 150.0, 300.0, 400.0):
    grid.append((25, 40.0, W, 16, 8))
for K, L in ((4, 4), (8, 8), (20, 32), (24, 16)):
    grid.append((25, 40.0, 212.0, K, L))
if os.environ.get("SWEEP_ONLY"):  # "k,R;k,R;..."
    want = [tuple(float(x) for x in e.split(",")) for e in os.environ["SWEEP_ONLY"].split(";")]
    grid = [g for g in grid if (float(g[0]), g[1]) in want and g[2:] == (212.0, 16, 8)]
out = []
for (k, R, W, K, L) in grid:
    a, b = synth.make_planes(k, K, L, W, seed=5)
    codes = synth.make_db(n, k, seed=6)
    centers, _ = synth.make_queries(codes, nq, seed=7)
    eng = Engine(k, K, L, W, a, b)
    eng.index_build(codes)
    row = {"k": k, "R": R, "W": W, "K": K, "L": L}
    res = {}
    for mode in ("auto", "stream"):
        eng.set_verify_mode(mode)
        eng.query(centers, R, want_cand=False)            # warm
        t0 = time.perf_counter()
        try:
            got = eng.query(centers, R, want_cand=False)
        except Exception as e:                            # capacity etc.
            row[mode + "_error"] = str(e)[:120]
            continue
        dt = time.perf_counter() - t0
        p = eng.profile()
        res[mode] = got
        row[mode + "_ms"] = round(dt * 1e3, 2)
        row[mode + "_dev_ms"] = round(p["ms_total"], 2)
        row[mode + "_join_ms"] = round(p["ms_join"], 2)
        row[mode + "_verify_ms"] = round(p["ms_verify"], 2)
        row[mode + "_final_ms"] = round(p["ms_finalize"], 2)
        row["cand_per_q"] = round(p["candidates"] / nq, 1)
        row[mode + "_survivors"] = int(p["provisional"])
        row["hits"] = int(len(got["q"]))
    if len(res) == 2:
        row["equal"] = all(np.array_equal(res["auto"][key], res["stream"][key]) for key in ("q", "id", "table", "dist"))
    eng.close()
    out.append(row)
    print(json.dumps(row), flush=True)
bad = [r for r in out if r.get("equal") is False]
print(json.dumps({"points": len(out), "unequal": len(bad)}))
sys.exit(1 if bad else 0)
