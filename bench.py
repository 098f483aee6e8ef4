#!/usr/bin/env python3
"""bench.py -- motif queries/s of the LSH search hot path on MI355X.

Workload (BASELINE.json configs[1]): 10 M synthetic 25-mers, L = 8 tables, K = 16, R = 40,
100 k queries per GPU (DB k-mers with 0..4 substitutions), index resident in HBM.  W = 212: the
smallest W of the committed sweep (tools/recall_sweep.py -> profiles/r02_recall_sweep_c2*.json:
W = 200 .. 220 in steps of 4, 6000 queries against the exhaustive scan) whose radius recall reaches
0.9, which is how SURVEY.md 8(d) fixes W (`--W 200` is the round-1 workload: recall 0.88).
A "step" = one pass of the query hot path (hash queries -> probe -> verify -> dedupe/exact
distance -> ordered hits [-> RCCL all-gather of hits when N > 1]) over the rank's query batch, with
the queries already resident in HBM.  N > 1: one process per GPU, index replicated, queries
sharded (weak scaling: 100 k queries per GPU), hits all-gathered.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel over its HIP-event time
(events on the library's stream): with the bucket join on (the default) that is hs_join8x_kernel
(k = 21..25, with hs_join8r_kernel beside it for segments of few probing queries; hs_join8xw_kernel for
k <= 20; hs_join8w_kernel for k = 26..50), an int8 MFMA GEMM of depth 128 (192 for
k <= 20 and 26..41, 256 for k <= 50) per (bucket member, probing query) pair, against the dense int8 MFMA peak; with --verify-mode stream (and wherever no join runs) it is hs_verify_kernel, priced by the
ALGORITHMIC bytes of SURVEY.md 8(d) against 8 TB/s.  `roofline.traffic` = measured HBM bytes per
launch from profiles/traffic_latest.json, reported only while that file's recorded kernel source
hash equals the hash of the kernel source this run was built from.  `cpu_baseline` (rank 0, N = 1 only)
times the reference ITSELF -- oracle/_ref, the reference's own translation units compiled by oracle/Makefile
(kind "reference"; the restatement oracle/ as kind "port" where _ref was not built) -- on a bounded sample of
the same workload: `value` is the MEASURED rate at the sample's N; the figure scaled to the bench's N is
reported beside it and labelled as an extrapolation.  `general_centres`: the same step with centres that are
no k-mers (jittered), the path of the reference's family centroids.

`secondary` (default on where the HBM holds it): BASELINE.json configs[2], the north-star target -- 10^8
25-mers, L = 32, K = 20, W = 160 from its own recall sweep, 10^6 queries -- as a STRONG-scaling job: one
query set sharded over the ranks, every rank one batch, the index replicated; `value` = 10^6 / the slowest
rank's pass.  At N = 1 it also times the 125 000-query pass that is one GPU's share of the 8-GPU job
(`one_gpu_share_of_eight`); at N > 1 the same job with the TABLES dealt over the ranks instead
(`table_partition`: every rank all queries on L / N tables, merged by the reference's first-seen rule) and with
the BUCKETS shared among the ranks of the replicated index (`bucket_partition`: every rank all queries and 1 / N
of the probes, the same merge; at N = 1 `one_gpu_share_of_eight_bucket_partition` times part 0 of 8).
`python bench.py --gpus N` typed as such starts its own N ranks (torch.distributed.run, 127.0.0.1).

Other BASELINE.json configs through the same script (the label in config.workload follows the
arguments): configs[2]'s per-GPU share `--db-size 100000000 --L 32 --K 20 --W 160 --queries 125000`,
configs[4] `--k 39` / `--k 15` (tools/bench_mixed_k.py runs the three lengths together).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
JOIN_K = 112.0     # GEMM depth of hs_join_kernel (fp16)


def join_i8_kernel(row_bytes, wide):
    """Which int8 join kernel hs_launch_join8w started, from what the library reports about the last
    batch (hs_profile.join_row_bytes = GEMM depth, join_wide): hs_join8.hip."""
    ks = row_bytes // 32
    if wide:
        if ks == 6:
            return "hs_join8xw_kernel"          # 16x16x64, 128-member work items
        return "hs_join8w_kernel<2,%d,wide>" % ks
    if ks > 4:
        return "hs_join8w_kernel<2,%d>" % ks
    return "hs_join8x_kernel"


MFMA_I8_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate per clock
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--db-size", dest="n", type=int, default=10_000_000, help="DB k-mers")
    ap.add_argument("--queries", dest="nq", type=int, default=100_000, help="queries per GPU")
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--L", type=int, default=8)
    ap.add_argument("--W", type=float, default=212.0,
                    help="bucket width; 212 = smallest W with radius recall >= 0.9 at the default shape "
                         "(profiles/r02_recall_sweep_c2_fine.json)")
    ap.add_argument("--R", type=float, default=40.0)
    ap.add_argument("--recall-queries", type=int, default=256)
    ap.add_argument("--planted-members", type=int, default=12,
                    help="members of each planted family of the recall@10 DB variant (0: skip)")
    ap.add_argument("--cpu-n", type=int, default=1_000_000, help="DB sample of the CPU baseline")
    ap.add_argument("--cpu-nq", type=int, default=200, help="query sample of the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-mode", choices=["auto", "stream", "join", "join16"], default="auto")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second block (configs[2]'s per-GPU shape: 10^8 25-mers, L=32, K=20)")
    ap.add_argument("--secondary-W", type=float, default=160.0,
                    help="W of the secondary block: profiles/r02_recall_sweep_c3_shape.json picks 152..160")
    ap.add_argument("--secondary-steps", type=int, default=8)
    ap.add_argument("--force-secondary", action="store_true", help="the secondary block whatever the primary workload")
    ap.add_argument("--no-secondary-tables", action="store_true",
                    help="N > 1: skip the table-partitioned form of the secondary block")
    ap.add_argument("--secondary-db-size", type=int, default=100_000_000,
                    help="DB k-mers of the secondary block (smaller: rehearsals of the multi-rank path on one GPU)")
    ap.add_argument("--secondary-queries", type=int, default=1_000_000,
                    help="queries of the secondary block in all (sharded over the ranks)")
    ap.add_argument("--general-steps", type=int, default=5,
                    help="steps of the `general_centres` measurement (centres that are not k-mers; 0: skip)")
    ap.add_argument("--general-jitter", type=float, default=0.05)
    ap.add_argument("--pcie-steps", type=int, default=5,
                    help="steps of the PCIe-inclusive measurement of hs_query / hs_query_codes (0: skip)")
    return ap.parse_args()


def workload_label(args):
    """Which BASELINE.json config the arguments are."""
    shape = "%d x %d-mers, L=%d K=%d W=%g R=%g, %d queries per GPU, index replicated per GPU" % (
        args.n, args.k, args.L, args.K, args.W, args.R, args.nq)
    if (args.n, args.k, args.K, args.L) == (10_000_000, 25, 16, 8):
        return "configs[1]: " + shape
    if (args.n, args.k, args.K, args.L) == (100_000_000, 25, 20, 32):
        return "configs[2], one GPU's share (the 10^6 queries are sharded over 8 GPUs): " + shape
    if args.k != 25:
        return "configs[4] (mixed lengths), k=%d: %s" % (args.k, shape)
    return "custom (no BASELINE.json config): " + shape


def kernel_source_hash():
    """sha256 over the sources of the verify kernels: ties profiles/traffic_latest.json to a build."""
    import hashlib
    h = hashlib.sha256()
    for f in ("hs_join8.hip", "hs_join.hip", "hs_kernels.hip", "hs_internal.h"):
        h.update(open(os.path.join(ROOT, "hsearch_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def synth_seed_planes():
    from hsearch_amd import synth
    return synth.SEED_PLANES


def cpu_baseline(args, a, b, codes, centers):
    """The reference's CPU path on a bounded sample of the bench workload: the first cpu_n DB k-mers (10^6),
    the first cpu_nq queries (200), one thread (the reference has no threads).
    kind "reference": oracle/_ref = the reference's own Search() compiled from its sources where they lie
    (oracle/Makefile), phases timed from outside (oracle/ref_search_harness.cpp ref_search_timed); its LSH
    constructor draws its own planes (same distributions).  The restatement (oracle/hs_oracle.cpp, "port",
    pinned bit-exact to it) is timed beside it on the same sample with the bench's planes, and is `value`
    only where oracle/_ref has not been built.  `value` is the rate MEASURED at the sample's N; bucket
    populations are linear in N, so value x cpu_n / N is an UPPER estimate at the bench's N (the cost per
    candidate grows once the vectors fall out of cache): value_scaled_to_bench_n, never `value`."""
    from oracle import pyoracle as O
    n_s = min(args.cpu_n, codes.shape[0])
    q_s = min(args.cpu_nq, centers.shape[0])
    db = O.embed_codes(codes[:n_s])
    cq = centers[:q_s]
    t0 = time.perf_counter()
    ix = O.Index(a, b, args.W, db)           # Search() build loop, motif_both_points.cpp:206-218
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = ix.query(cq, args.R)               # Search() query loop, :224-245
    t_query = time.perf_counter() - t0
    # EXTENSION: the same query loop over every host core of the box (the reference is
    # single-threaded; `value` stays the single-thread rate).
    n_thr = max(1, len(os.sched_getaffinity(0)))
    q_mt = min(centers.shape[0], max(q_s, 6 * n_thr))
    t0 = time.perf_counter()
    res_mt = ix.query_mt(centers[:q_mt], args.R, n_thr)
    t_mt = time.perf_counter() - t0
    ix.close()
    port = {"what": "oracle/hs_oracle.cpp: restatement with the reference's cost structure, pinned bit-exact to the "
                    "compiled reference; the bench's planes",
            "queries_per_s": q_s / t_query, "build_kmers_per_s": n_s / t_build, "hits": int(len(res["q"])),
            "seconds": t_build + t_query}
    out = {"value": port["queries_per_s"], "unit": "queries/s", "cores": 1, "kind": "port",
           "sample_n": n_s, "sample_nq": q_s, "build_kmers_per_s": port["build_kmers_per_s"],
           "sample_seconds": port["seconds"], "port": port}
    if O.have_ref():
        seed = synth_seed_planes()
        tb0, tr0, _ = O.ref_search_timed(seed, db, cq[:0], args.K, args.L, args.W, args.R)
        tb1, tr1, nh = O.ref_search_timed(seed, db, cq, args.K, args.L, args.W, args.R)
        tq = max(tr1 - tr0, 1e-9)     # query loop = (query loop + table destruction) - (table destruction alone)
        out.update({"value": q_s / tq, "kind": "reference", "build_kmers_per_s": n_s / tb1,
                    "sample_seconds": tb0 + tr0 + tb1 + tr1,
                    "reference": {"what": "oracle/_ref/libref_search.so: Search() of motif_both_points.cpp:195-250 "
                                          "compiled from the reference's sources; planes drawn by its own LSH "
                                          "constructor (seeded); build loop and query loop timed from outside",
                                  "queries_per_s": q_s / tq, "build_kmers_per_s": n_s / tb1, "hits": nh,
                                  "seconds_build": [tb0, tb1], "seconds_after_build": [tr0, tr1]}})
    out["sample"] = ("%s on the first %d DB k-mers and first %d queries of the bench workload, one thread: %.1f "
                     "queries/s MEASURED at N=%d (value); index build %.0f k-mers/s; %.1f s of CPU work"
                     % ("oracle/_ref (the reference's own Search(), compiled)" if out["kind"] == "reference"
                        else "oracle/hs_oracle.cpp (pinned restatement)", n_s, q_s, out["value"], n_s,
                        out["build_kmers_per_s"], out["sample_seconds"]))
    out["measured_qps_at_sample"] = out["value"]
    out["value_scaled_to_bench_n"] = out["value"] * n_s / codes.shape[0]
    out["scaling_note"] = ("value x sample_n / bench N: upper estimate (candidates per query are linear in N, the "
                           "cost per candidate grows with N)")
    out["all_cores_extension"] = {
        "what": "the restatement's query loop (oracle hso_index_query_mt) over all host threads of the box; not "
                "something the single-threaded reference does; the index build stays single-threaded",
        "threads": n_thr, "sample_nq": int(q_mt), "measured_qps_at_sample": q_mt / t_mt,
        "value_scaled_to_bench_n": q_mt / t_mt * n_s / codes.shape[0],
        "hits": int(len(res_mt["q"])), "seconds": t_mt}
    return out


def planted_family_recall(args, eng, codes, a, b, device):
    """recall@10 where it carries information (SURVEY 8(d) defines it; on the i.i.d. DB only the
    k-mer a query was mutated from lies within R, so at most 1 of its 10 nearest can be a hit): a
    DB variant in which every recall query has a planted family -- `planted_members` DB k-mers that
    are its source k-mer with 1..2 substitutions, all within R of the query -- so the 10 nearest
    neighbours are family members an LSH search is supposed to find.  Same planes, same W; the
    variant is a copy of the bench DB with nr * planted_members rows overwritten."""
    from hsearch_amd import Engine, synth
    nr, m, k = min(args.recall_queries, args.nq), args.planted_members, args.k
    rng = np.random.Generator(np.random.MT19937(synth.SEED_DB + 77))
    db = codes.copy()
    centre = rng.integers(0, 20, size=(nr, k), dtype=np.uint8)
    rows = rng.choice(len(db), size=nr * m, replace=False)
    fam = np.repeat(centre, m, axis=0)
    for s in range(2):
        sel = np.nonzero(rng.integers(0, 2, size=len(fam)) + (s == 0) > 0)[0]   # 1 or 2 substitutions
        fam[sel, rng.integers(0, k, size=len(sel))] = rng.integers(0, 20, size=len(sel), dtype=np.uint8)
    db[rows] = fam
    q = centre.copy()
    sel = np.nonzero(rng.integers(0, 2, size=nr))[0]
    q[sel, rng.integers(0, k, size=len(sel))] = rng.integers(0, 20, size=len(sel), dtype=np.uint8)
    centers = synth.embed(q)
    e2 = Engine(k, args.K, args.L, args.W, a, b, device=device)
    e2.index_build(db)
    lsh = e2.query(centers, args.R, want_cand=False)
    nn, nd2 = e2.bruteforce_topk(centers, 10)
    bf = e2.bruteforce(centers, args.R)
    e2.close()
    by_q = {}
    for qq, ii in zip(lsh["q"].tolist(), lsh["id"].tolist()):
        by_q.setdefault(qq, set()).add(ii)
    truth = set(zip(bf["q"].tolist(), bf["id"].tolist()))
    found = set(zip(lsh["q"].tolist(), lsh["id"].tolist()))
    return {"recall_at_10": float(np.mean([len(by_q.get(i, set()) & set(nn[i].tolist())) / 10.0 for i in range(nr)])),
            "radius_recall": len(truth & found) / max(len(truth), 1),
            "queries": nr, "members_per_family": m,
            "tenth_neighbour_within_R": float(np.mean(nd2[:, 9] <= args.R * args.R)),
            "true_neighbours_within_R": len(truth)}


class Workload:
    """One resident index + one resident query batch on this rank, and the timed loop over it."""

    def __init__(self, args, dev_index, dev, rank, synth, Engine, torch, label, queries=True):
        self.args, self.dev, self.rank, self.torch, self.label = args, dev, rank, torch, label
        k, K, L, W = args.k, args.K, args.L, args.W
        self.a, self.b = synth.make_planes(k, K, L, W)
        self.codes = synth.make_db(args.n, k)
        self.synth = synth
        self.eng = Engine(k, K, L, W, self.a, self.b, device=dev_index)
        self.eng.set_verify_mode(args.verify_mode)
        t0 = time.perf_counter()
        self.info = self.eng.index_build(self.codes)
        self.t_build = time.perf_counter() - t0
        self.build_prof = self.eng.profile()
        self.q_offset = rank * args.nq
        if queries:
            self.set_queries(args.nq, synth.SEED_QUERIES + 1000 * rank)

    def set_queries(self, nq, seed, jitter=0.0, q_offset=None, block_of=None):
        """The rank's resident query batch: nq DB k-mers with 0..4 substitutions, embedded exactly from the
        table (jitter > 0: Gaussian noise on every coordinate, so that no centre is a k-mer any more).
        block_of = (total, lo): the batch is the block [lo, lo + nq) of a set of `total` queries drawn with
        `seed` -- the same set on every rank (strong scaling: one query set, sharded)."""
        synth, torch = self.synth, self.torch
        self.args = argparse.Namespace(**vars(self.args))
        self.args.nq = nq
        if q_offset is not None:
            self.q_offset = q_offset
        if block_of is not None:
            qc, src = synth.make_query_codes(self.codes, block_of[0], seed=seed)
            self.qcodes, self.src = qc[block_of[1]:block_of[1] + nq].copy(), src[block_of[1]:block_of[1] + nq].copy()
        else:
            self.qcodes, self.src = synth.make_query_codes(self.codes, nq, seed=seed)
        self.centers = synth.embed(self.qcodes)
        if jitter:
            self.centers = self.centers + np.random.Generator(np.random.MT19937(seed + 1)).normal(
                0.0, jitter, size=self.centers.shape)
        self.d_centers = None
        self.out = None
        self.d_centers = torch.from_numpy(np.ascontiguousarray(self.centers)).to(self.dev)
        self.cap = 16 * nq + 4096
        self.out = self.alloc(self.cap)

    def alloc(self, c):
        t, dev = self.torch, self.dev
        return dict(q=t.empty(c, dtype=t.int32, device=dev), id=t.empty(c, dtype=t.int32, device=dev),
                    table=t.empty(c, dtype=t.int32, device=dev), dist=t.empty(c, dtype=t.float64, device=dev))

    def step(self, hdist, HsError, use_dist):
        """One pass of the query hot path over the rank's resident batch (+ the hit all-gather)."""
        args = self.args
        while True:
            o = self.out
            try:
                nh = self.eng.query_dev(self.d_centers.data_ptr(), args.nq, args.R, o["q"].data_ptr(),
                                        o["id"].data_ptr(), o["table"].data_ptr(), o["dist"].data_ptr(), self.cap)
                break
            except HsError as e:
                if getattr(e, "needed", 0) <= self.cap:
                    raise
                self.cap = int(e.needed * 1.25) + 1024
                self.out = self.alloc(self.cap)
        gathered = hdist.allgather_hits(o["q"], o["id"], o["table"], o["dist"], nh, q_offset=self.q_offset,
                                        force=use_dist)
        return nh, gathered

    def timed(self, steps, warmup, hdist, HsError, use_dist, fence, dist, backend):
        """W untimed steps, then exactly `steps` steps between two fences; max over ranks."""
        for _ in range(warmup):
            self.step(hdist, HsError, use_dist)
        acc = dict(verify_ms=0.0, hash_ms=0.0, probe_ms=0.0, fin_ms=0.0, join_ms=0.0, launches=0, join_batches=0,
                   join_i8=0, retries=0, recognised=0)
        jstat, qproj, join_rows, cand, hits_local = (0, 0, 0, 0), (0, 0), (128, 0), 0, 0
        gathered = None
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            nh, gathered = self.step(hdist, HsError, use_dist)
            p = self.eng.profile()
            acc["verify_ms"] += p["ms_verify"]
            acc["hash_ms"] += p["ms_hash"]
            acc["probe_ms"] += p["ms_probe"]
            acc["fin_ms"] += p["ms_finalize"]
            acc["launches"] += p["verify_launches"]
            acc["join_batches"] += p["join_batches"]
            acc["join_ms"] += p["ms_join"]
            acc["join_i8"] += p["join_i8_batches"]
            acc["retries"] += p["join_async_retries"]
            acc["recognised"] += p["queries_recognised"]
            jstat = (p["join_items"], p["join_pairs"], p["join_pairs_issued"], p["join_items_resident"])
            if p["join_i8_batches"]:
                join_rows = (int(p["join_row_bytes"]), int(p["join_wide"]))
            qproj = (p["hash_values"], p["hash_flagged"])
            cand = p["candidates"]
            hits_local = nh
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        acc.update(dt=dt, jstat=jstat, qproj=qproj, join_rows=join_rows, cand=cand, hits_local=hits_local,
                   total_hits=int(gathered[0].numel()) if gathered is not None else 0)
        return acc

    def roofline(self, m, steps, traffic=None, traffic_src=None):
        """The dominant kernel of the timed loop priced against its roofline (module docstring)."""
        args = self.args
        k, L, d = args.k, args.L, 8 * args.k
        steps = max(steps, 1)
        cand, hits_local, jstat, join_rows = m["cand"], m["hits_local"], m["jstat"], m["join_rows"]
        # ALGORITHMIC bytes of one step on this rank: per scanned bucket entry one
        # k-byte k-mer + one u32 id, + per query its vector (8d) + L bucket lookups (16 B) + 16 B per hit
        algo_bytes = cand * (k + 4) + args.nq * (8 * d + 16 * L) + 16 * hits_local
        v_ms = m["verify_ms"] / steps
        achieved = algo_bytes / (v_ms * 1e-3) / 1e9 if v_ms > 0 else 0.0
        if m["join_batches"]:
            # dominant kernel = the bucket join: an int8 MFMA GEMM of depth 128 (25 positions x 4
            # coordinates + 28 threshold-digit slots; 192 / 256 for longer rows: k = 26..50, and rows
            # over all 8 columns for k <= 20 or large radii -- the library reports which:
            # hs_profile.join_row_bytes / join_wide; hs_join8.hip) -- or, when a batch had to fall
            # back, the fp16 form of depth 112 (hs_join.hip) -- per (bucket member, probing query)
            # pair: 2 * depth operations per pair.
            j_ms = m["join_ms"] / steps
            i8 = m["join_i8"] > 0
            jk = float(join_rows[0]) if i8 else JOIN_K
            peak = MFMA_I8_PEAK_TOPS if i8 else MFMA_F16_PEAK_TFLOPS
            flop = jstat[1] * 2.0 * jk          # real (member, query) pairs routed to the join
            tf = flop / (j_ms * 1e-3) / 1e12 if j_ms > 0 else 0.0
            return {"bound": "mfma", "kernel": join_i8_kernel(*join_rows) if i8 else "hs_join_kernel",
                    "mfma_dtype": "i8" if i8 else "f16", "gemm_depth": jk, "achieved": tf,
                    "peak": peak, "unit": "TOP/s" if i8 else "TFLOP/s",
                    "frac": tf / peak, "traffic": traffic, "traffic_source": traffic_src,
                    "flop_per_step": flop, "pairs_per_step": jstat[1],
                    "pairs_issued_per_step": jstat[2], "work_items_per_step": jstat[0],
                    # ... of which in segments with few probing queries, run by hs_join8r_kernel (query rows
                    # resident, member tiles streamed); the rest by the kernel named above
                    "work_items_query_resident": jstat[3],
                    "issued_over_useful": (jstat[2] / jstat[1]) if jstat[1] else None,
                    "issued_tops": (jstat[2] * 2.0 * jk / (j_ms * 1e-3) / 1e12) if j_ms > 0 else 0.0,
                    "kernel_ms_per_step": j_ms, "streaming_kernel_ms_per_step": v_ms - j_ms,
                    "pairs_streamed_per_step": cand - jstat[1],
                    "launches_per_step": m["launches"] / steps,
                    "steps_with_a_repeated_join": m["retries"],
                    # SURVEY 8(d)'s byte count for the same step, for reference only: the join
                    # re-uses a bucket's rows across the queries probing it, so this is NOT a
                    # physical rate (it exceeds the HBM peak) and no fraction of 8 TB/s is given
                    "algorithmic_bytes_per_step": algo_bytes,
                    "algorithmic_bytes_per_s_nonphysical": achieved * 1e9}
        return {"bound": "hbm", "kernel": "hs_verify_kernel", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_step": algo_bytes,
                "kernel_ms_per_step": v_ms, "launches_per_step": m["launches"] / steps,
                "packed_stream_bytes_per_step": cand * 16 * ((k + 24) // 25)}

    def index_block(self):
        """Index build of this workload, with its SURVEY 8(d) roofline: bytes per k-mer = k (codes in)
        + 36 L (12 B (key, id) written, then one read + write pass to group them) + k L + 4 L (the
        bucket-ordered copies), over the device time of the build (hs_profile.ms_total)."""
        args, bp = self.args, self.build_prof
        bpk = args.k + 36 * args.L + args.k * args.L + 4 * args.L
        ms = bp["ms_total"]
        gbs = args.n * bpk / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"build_seconds": self.t_build, "build_kmers_per_s": args.n / self.t_build,
                "device_ms": {f: bp[f] for f in ("ms_hash", "ms_sort", "ms_gather", "ms_total")},
                "roofline": {"bound": "hbm", "bytes_per_kmer": bpk, "algorithmic_bytes": args.n * bpk,
                             "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "device_ms": ms},
                # LSH projection on the matrix cores (hs_proj.hip): values produced by the int8
                # MFMA pass / of which recomputed in the reference's fp64 order (within the
                # proven error bound of a bucket boundary)
                "projection": {"build_values": bp["hash_values"], "build_recomputed": bp["hash_flagged"]},
                "device_bytes": self.info["device_bytes"], "n_buckets": self.info["n_buckets"],
                "max_bucket": self.info["max_bucket"]}

    def recall(self, nr):
        """Radius recall and recall@10 on a query subsample; ground truth = the exhaustive scans on the
        GPU (hs_bruteforce / hs_bruteforce_topk, themselves parity-tested against the oracle)."""
        args, eng = self.args, self.eng
        sub = self.centers[:nr]
        bf = eng.bruteforce(sub, args.R)
        lsh = eng.query(sub, args.R, want_cand=False)
        truth = set(zip(bf["q"].tolist(), bf["id"].tolist()))
        found = set(zip(lsh["q"].tolist(), lsh["id"].tolist()))
        out = {"radius_recall": len(truth & found) / max(len(truth), 1), "recall_queries": nr,
               "true_neighbours_in_sample": len(truth)}
        # recall@10: |LSH hits of q  ∩  the 10 nearest DB k-mers of q| / 10, the
        # nearest ones by (d2, id) from the exact exhaustive top-k scan (hs_bruteforce_topk)
        nn, _ = eng.bruteforce_topk(sub, 10)
        by_q = {}
        for qq, ii in zip(lsh["q"].tolist(), lsh["id"].tolist()):
            by_q.setdefault(qq, set()).add(ii)
        out["recall_at_10"] = float(np.mean([len(by_q.get(qq, set()) & set(nn[qq].tolist())) / 10.0
                                             for qq in range(nr)]))
        return out

    def pcie_inclusive(self, steps):
        """The same step through the HOST-pointer entry points (queries cross PCIe inside the timed region,
        hits come back to host arrays): hs_query (8d bytes per query) and hs_query_codes (k bytes).
        Never part of `value`."""
        args = self.args
        cap = self.cap
        hq, hid, ht = (np.empty(cap, np.uint32) for _ in range(3))
        hd = np.empty(cap, np.float64)
        out = {}
        for name, arr, codes in (("hs_query", self.centers, False), ("hs_query_codes", self.qcodes, True)):
            self.eng.query_into(arr, args.R, hq, hid, ht, hd, codes=codes)        # warm-up (staging buffers)
            t0 = time.perf_counter()
            for _ in range(steps):
                nh = self.eng.query_into(arr, args.R, hq, hid, ht, hd, codes=codes)
            dt = (time.perf_counter() - t0) / steps
            out[name] = {"queries_per_s": args.nq / dt, "ms_per_step": dt * 1e3,
                         "bytes_in_per_query": int(arr.shape[1] * arr.itemsize), "hits": nh}
        return out

    def close(self):
        self.eng.close()
        self.d_centers = None
        self.out = None


def launch_ranks(args):
    """`python bench.py --gpus N` typed without a launcher: start one rank per GPU through
    torch.distributed.run as a CHILD process (this parent has touched neither torch nor HIP), relay rank 0's
    JSON line and the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def secondary_block(args, world, rank, dev_index, dev, synth, Engine, torch, hdist, HsError, use_dist, fence, dist,
                    backend):
    """BASELINE.json configs[2] -- the north star's own target: 10^8 25-mers, L = 32, K = 20, 10^6 queries in
    all -- as a STRONG-scaling job: the 10^6 queries are sharded over the ranks (10^6 / world per rank, one
    batch), the index is replicated; value = 10^6 / the slowest rank's time per pass.  At one rank the block
    also times 125 000 queries per pass: the share of one GPU of eight, the figure earlier rounds reported."""
    total_q = args.secondary_queries
    lo, hi = hdist.shard_bounds(total_q, rank, world)
    a2 = argparse.Namespace(**vars(args))
    a2.n, a2.L, a2.K, a2.W, a2.nq = args.secondary_db_size, 32, 20, args.secondary_W, hi - lo
    t0 = time.perf_counter()
    w2 = Workload(a2, dev_index, dev, rank, synth, Engine, torch, workload_label(a2), queries=False)
    w2.set_queries(hi - lo, synth.SEED_QUERIES, q_offset=lo, block_of=(total_q, lo))   # ONE query set, sharded
    steps = max(args.secondary_steps, 1)
    m2 = w2.timed(steps, 2, hdist, HsError, use_dist, fence, dist, backend)
    sec = None
    if rank == 0:
        sec = {"what": "BASELINE.json configs[2] (the north-star target: 100M x 25-mers, L=32, K=20, 10^6 queries), "
                       "strong scaling: the queries sharded over the ranks in contiguous blocks, one batch per rank "
                       "and pass, index replicated per GPU; W from profiles/r02_recall_sweep_c3_shape.json",
               "config": {"workload": "%s: %d x 25-mers, L=32 K=20 W=%g R=%g, %d queries in all"
                                      % ("configs[2]" if (a2.n, total_q) == (100_000_000, 1_000_000) else "custom",
                                         a2.n, a2.W, a2.R, total_q),
                          "db_kmers": a2.n, "k": a2.k, "L": a2.L, "K": a2.K, "W": a2.W, "R": a2.R,
                          "queries_total": total_q, "queries_per_gpu": a2.nq,
                          "parallelism": "query-sharded x%d (10^6 queries / %d ranks), index replicated" % (world, world)},
               "value": total_q * steps / m2["dt"], "unit": "queries/s", "n_gpus": world, "scaling": "strong",
               "steps": steps, "warmup": 2, "ms_per_step": m2["dt"] / steps * 1e3,
               "roofline": w2.roofline(m2, steps, *secondary_traffic(a2, world)),
               "phases_ms_per_step": {"hash_queries": m2["hash_ms"] / steps, "probe_segments": m2["probe_ms"] / steps,
                                      "verify": m2["verify_ms"] / steps, "finalize_sort": m2["fin_ms"] / steps},
               "candidates_per_query": m2["cand"] / a2.nq, "hits_per_step_rank0": m2["hits_local"],
               "hits_gathered": m2["total_hits"],
               "index": w2.index_block()}
        nr2 = min(args.recall_queries, a2.nq)
        if nr2 > 0:
            sec.update(w2.recall(nr2))
    if world == 1 and total_q > 125_000:
        # ... and one GPU's share when eight hold the replicated index and share the BUCKETS (hs_set_bucket_partition:
        # every rank all 10^6 queries, 1/8 of the probes): part 0 of 8 on this handle
        w2.eng.set_bucket_partition(0, 8)
        try:
            m4 = w2.timed(steps, 2, hdist, HsError, use_dist, fence, dist, backend)
        finally:
            w2.eng.set_bucket_partition(0, 1)
        sec["one_gpu_share_of_eight_bucket_partition"] = {
            "what": "the same index and ALL queries, the buckets of part 0 of 8 (hs_set_bucket_partition): the per-GPU "
                    "pass of the 8-GPU job in the bucket-partitioned layout, before the exchange (all-gather + "
                    "first-seen merge of ~ 1.1 hits per query).  The eight parts take the same time within 5 % "
                    "(profiles/r04_bucket_sharding_emulation.json)",
            "queries": total_q, "ms_per_step": m4["dt"] / steps * 1e3,
            "job_queries_per_s_if_all_eight_ranks_take_this_long": total_q * steps / m4["dt"],
            "roofline": w2.roofline(m4, steps),
            "phases_ms_per_step": {"hash_queries": m4["hash_ms"] / steps, "probe_segments": m4["probe_ms"] / steps,
                                   "verify": m4["verify_ms"] / steps, "finalize_sort": m4["fin_ms"] / steps},
            "hits_of_this_part": m4["hits_local"]}
        # one GPU's share when eight hold the replicated index (what BENCH_r02 / r03 carried as `secondary`)
        w2.set_queries(125_000, synth.SEED_QUERIES, q_offset=0)
        m3 = w2.timed(steps, 2, hdist, HsError, use_dist, fence, dist, backend)
        sec["one_gpu_share_of_eight"] = {
            "what": "the same index, 125 000 queries per pass: the per-GPU work of the 8-GPU job",
            "queries_per_gpu": 125_000, "value": 125_000 * steps / m3["dt"], "unit": "queries/s",
            "ms_per_step": m3["dt"] / steps * 1e3, "roofline": w2.roofline(m3, steps),
            "phases_ms_per_step": {"hash_queries": m3["hash_ms"] / steps, "probe_segments": m3["probe_ms"] / steps,
                                   "verify": m3["verify_ms"] / steps, "finalize_sort": m3["fin_ms"] / steps},
            "candidates_per_query": m3["cand"] / 125_000}
    if world > 1 and not args.no_secondary_tables:
        # the same job with the BUCKETS shared among the ranks (hsearch_dist.h): index replicated (this handle),
        # every rank ALL 10^6 queries in one batch and 1 / world of the probes; all-gather + first-seen merge
        bp = bucket_partition_block(args, a2, total_q, w2, world, rank, dev, synth, torch, hdist, use_dist, fence, dist,
                                    backend)
        if rank == 0:
            sec["bucket_partition"] = bp
    w2.close()
    codes2, qc_all = w2.codes, None
    del w2
    if world > 1 and not args.no_secondary_tables:
        # the same job in the TABLE-partitioned layout (hsearch_dist.h): every rank a subset of the 32 tables
        # (dealt by estimated join work) over all 10^8 k-mers and ALL 10^6 queries in one batch; hits with
        # global table numbers all-gathered, merged by the first-seen rule (smallest table per (query, id))
        tp = table_partition_block(args, a2, total_q, codes2, world, rank, dev_index, dev, synth, Engine, torch,
                                   hdist, HsError, use_dist, fence, dist, backend)
        if rank == 0:
            sec["table_partition"] = tp
    if rank == 0:
        sec["wall_seconds_of_this_block"] = time.perf_counter() - t0
    return sec


def bucket_partition_block(args, a2, total_q, w2, world, rank, dev, synth, torch, hdist, use_dist, fence, dist, backend):
    eng, R = w2.eng, a2.R
    qcodes, _ = synth.make_query_codes(w2.codes, total_q, seed=synth.SEED_QUERIES)     # the same on every rank
    d_centers = torch.from_numpy(synth.embed(qcodes)).to(dev)
    cap = 8 * total_q + 4096
    out = dict(q=torch.empty(cap, dtype=torch.int32, device=dev), id=torch.empty(cap, dtype=torch.int32, device=dev),
               table=torch.empty(cap, dtype=torch.int32, device=dev), dist=torch.empty(cap, dtype=torch.float64, device=dev))

    def step():
        return hdist.query_bucket_partitioned(eng, rank, world, d_centers.data_ptr(), total_q, R, out, cap, force=use_dist)
    steps = max(args.secondary_steps, 1)
    for _ in range(2):
        merged, nh = step()
    acc = {}
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        merged, nh = step()
        p = eng.profile()
        for f in ("ms_hash", "ms_probe", "ms_verify", "ms_join", "ms_finalize", "ms_total"):
            acc[f] = acc.get(f, 0.0) + p[f] / steps
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"what": "configs[2] with the BUCKETS shared among the ranks (index replicated, every rank all queries in "
                    "one batch and the probes of its part: hs_set_bucket_partition; all-gather + first-seen merge); "
                    "same output as the query-sharded layout",
            "value": total_q * steps / dt, "unit": "queries/s", "n_gpus": world, "ms_per_step": dt / steps * 1e3,
            "rank0_device_ms_per_step": acc, "rank0_hits_before_merge": int(nh), "hits_after_merge": int(merged[0].numel()),
            "rank0_join_frac_of_int8_peak": (p["join_pairs"] * 256.0 / (acc["ms_join"] * 1e-3) / 5e15) if acc.get("ms_join") else None}


def table_partition_block(args, a2, total_q, codes, world, rank, dev_index, dev, synth, Engine, torch, hdist, HsError,
                          use_dist, fence, dist, backend):
    k, K, L, W, R = a2.k, a2.K, a2.L, a2.W, a2.R
    a, b = synth.make_planes(k, K, L, W)
    cost = hdist.table_costs(k, K, L, W, a, b, codes[:32768], device=dev_index)
    tabs = hdist.assign_tables(cost, L, world)
    mine = tabs[rank]
    eng = Engine(k, K, len(mine), W, a[mine], b[mine], device=dev_index)
    t0 = time.perf_counter()
    info = eng.index_build(codes)
    t_build = time.perf_counter() - t0
    qcodes, _ = synth.make_query_codes(codes, total_q, seed=synth.SEED_QUERIES)     # the same on every rank
    d_centers = torch.from_numpy(synth.embed(qcodes)).to(dev)
    cap = [8 * total_q + 4096]
    out = [dict(q=torch.empty(cap[0], dtype=torch.int32, device=dev), id=torch.empty(cap[0], dtype=torch.int32, device=dev),
                table=torch.empty(cap[0], dtype=torch.int32, device=dev),
                dist=torch.empty(cap[0], dtype=torch.float64, device=dev))]

    def step():
        return hdist.query_table_partitioned(eng, mine, d_centers.data_ptr(), total_q, R, out[0], cap[0], force=use_dist)
    steps = max(args.secondary_steps, 1)
    for _ in range(2):
        merged, nh = step()
    acc = {}
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        merged, nh = step()
        p = eng.profile()
        for f in ("ms_hash", "ms_probe", "ms_verify", "ms_join", "ms_finalize", "ms_total"):
            acc[f] = acc.get(f, 0.0) + p[f] / steps
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res = {"what": "configs[2] with the TABLES partitioned over the ranks (every rank: its tables of all k-mers, all "
                   "queries in one batch; all-gather + first-seen merge); same output as the replicated layout",
           "value": total_q * steps / dt, "unit": "queries/s", "n_gpus": world, "ms_per_step": dt / steps * 1e3,
           "tables_of_rank0": [int(x) for x in mine], "tables_per_rank": [len(t) for t in tabs],
           "estimated_cost_share_per_rank": [float(cost[t].sum() / cost.sum()) for t in tabs],
           "rank0_device_ms_per_step": acc, "rank0_hits_before_merge": int(nh), "hits_after_merge": int(merged[0].numel()),
           "rank0_index_bytes": info["device_bytes"], "rank0_build_seconds": t_build,
           "rank0_join_frac_of_int8_peak": (p["join_pairs"] * 256.0 / (acc["ms_join"] * 1e-3) / 5e15) if acc.get("ms_join") else None}
    eng.close()
    return res


def secondary_traffic(a2, world):
    """HBM bytes per launch of the secondary block's dominant kernel from the PMC passes of
    tools/pmc_secondary.sh (profiles/traffic_secondary.json), while its recorded kernel source hash and
    shape are this run's."""
    tpath = os.path.join(ROOT, "profiles", "traffic_secondary.json")
    if not os.path.exists(tpath):
        return None, None
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None, None
    same = (tj.get("kernel_source_hash") == kernel_source_hash() and tj.get("queries_per_gpu") == a2.nq and
            tj.get("W") == a2.W)
    if not same:
        return None, {"file": "profiles/traffic_secondary.json", "stale": True,
                      "recorded_hash": tj.get("kernel_source_hash"), "current_hash": kernel_source_hash(),
                      "recorded_queries_per_gpu": tj.get("queries_per_gpu")}
    return tj.get("verify_bytes_per_launch"), {"file": "profiles/traffic_secondary.json",
                                               "kernel_source_hash": tj.get("kernel_source_hash"),
                                               "taken": tj.get("taken"), "kernels": tj.get("kernels")}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))          # (before anything touches torch or the GPU)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # stdout carries ONE line, the JSON result: libraries that print there (RCCL announces its
    # version on stdout when the first communicator is created) are sent to stderr for the whole run
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from hsearch_amd import Engine, HsError, synth
    from hsearch_amd import dist as hdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # HS_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share
    # devices, collectives go through host memory); the driver's runs use RCCL ("nccl").
    backend = os.environ.get("HS_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # HS_BENCH_FORCE_DIST=1: run the collective path even with one rank (RCCL rehearsal on one GPU)
    use_dist = world > 1 or bool(os.environ.get("HS_BENCH_FORCE_DIST"))
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_ranks_seen = dist.get_world_size() if use_dist else 1

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    k, K, L, W, R = args.k, args.K, args.L, args.W, args.R
    wl = Workload(args, dev_index, dev, rank, synth, Engine, torch, workload_label(args))
    m = wl.timed(args.steps, args.warmup, hdist, HsError, use_dist, fence, dist, backend)
    line = None
    if rank == 0:
        steps = max(args.steps, 1)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        # the PMC passes behind that file were taken on the default workload and kernel choice: the
        # figure is reported for that workload only (null otherwise)
        default_workload = (args.n, args.nq, args.k, args.K, args.L, args.W, args.R, args.verify_mode) == \
            (10_000_000, 100_000, 25, 16, 8, 212.0, 40.0, "auto")
        traffic_src = None
        if default_workload and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # only while the PMC passes were taken on the kernels this run was built from
                if tj.get("kernel_source_hash") == kernel_source_hash():
                    traffic = tj.get("verify_bytes_per_launch")
                    traffic_src = {"file": "profiles/traffic_latest.json",
                                   "kernel_source_hash": tj.get("kernel_source_hash"),
                                   "taken": tj.get("taken")}
                else:
                    traffic_src = {"file": "profiles/traffic_latest.json", "stale": True,
                                   "recorded_hash": tj.get("kernel_source_hash"),
                                   "current_hash": kernel_source_hash()}
            except Exception:
                traffic = None
        index = wl.index_block()
        index["projection"].update({"query_values_per_step": m["qproj"][0], "query_recomputed_per_step": m["qproj"][1]})
        line = {
            "metric": "motif queries/sec (LSH probe + verify, index resident in HBM)",
            "value": world * args.nq * steps / m["dt"], "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": m["dt"] / steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "n_ranks_seen": n_ranks_seen,
            "config": {"workload": wl.label,
                       "db_kmers": args.n, "k": k, "L": L, "K": K, "W": W, "R": R,
                       "queries_per_gpu": args.nq, "parallelism": "query-sharded x%d" % world},
            "roofline": wl.roofline(m, args.steps, traffic, traffic_src),
            "verify_mode": args.verify_mode,
            "phases_ms_per_step": {"hash_queries": m["hash_ms"] / steps, "probe_segments": m["probe_ms"] / steps,
                                   "verify": m["verify_ms"] / steps, "finalize_sort": m["fin_ms"] / steps},
            "candidates_per_query": m["cand"] / args.nq, "hits_per_step_rank0": m["hits_local"],
            "hits_gathered": m["total_hits"],
            # the synthetic centres are k-mers' points, as the reference's centres files hold: hs_query_dev finds
            # that out per call (one pass over the [nq][8k] doubles, inside the timed step) and then works from
            # the residue codes it read off them; `general_centres` below is the same step for centres that are
            # NOT k-mers (the points path)
            "queries_recognised_as_kmers_per_step": m["recognised"] / steps,
            "index": index,
        }
        nr = min(args.recall_queries, args.nq)
        if nr > 0:
            line.update(wl.recall(nr))
        if nr > 0 and args.planted_members > 0:
            line["planted_family_recall"] = planted_family_recall(args, wl.eng, wl.codes, wl.a, wl.b, dev_index)
        sweep = os.path.join(ROOT, "profiles", "r02_recall_sweep_c2_fine.json")
        if os.path.exists(sweep) and (args.n, args.k, args.K, args.L, args.R) == (10_000_000, 25, 16, 8, 40.0):
            sj = json.load(open(sweep))
            line["recall_sweep"] = {"file": "profiles/r02_recall_sweep_c2_fine.json (coarse grid: r02_recall_sweep_c2.json)",
                                    "smallest_W_with_radius_recall_0.9": sj["smallest_W_with_radius_recall_0.9"],
                                    "points": [{"W": r["W"], "radius_recall": round(r["radius_recall"], 4),
                                                "candidates_per_query": round(r["candidates_per_query"])}
                                               for r in sj["sweep"]]}
        if world == 1 and args.pcie_steps > 0:
            # the boundary's host-pointer entry points, PCIe inside the timed region; never `value`
            line["value_pcie_inclusive"] = wl.pcie_inclusive(args.pcie_steps)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, wl.a, wl.b, wl.codes, wl.centers)
    if world == 1 and args.general_steps > 0:
        # centres that are NOT k-mers (the reference's real centres are family centroids,
        # centerDistanceSmapling.cpp:67-78): the same queries with Gaussian noise on every coordinate, so that
        # hs_query_dev recognises nothing and every per-query quantity comes from the point rows
        wl.set_queries(args.nq, synth.SEED_QUERIES + 1000 * rank, jitter=args.general_jitter)
        mg = wl.timed(args.general_steps, 2, hdist, HsError, use_dist, fence, dist, backend)
        gs = max(args.general_steps, 1)
        line["general_centres"] = {
            "what": "the primary workload's queries + N(0, %g^2) noise per coordinate: no centre is a k-mer, the "
                    "points path (hs_quant_points / hs_qprep8 / hs_finalize_kernel) runs" % args.general_jitter,
            "value": args.nq * gs / mg["dt"], "unit": "queries/s", "steps": args.general_steps,
            "ms_per_step": mg["dt"] / gs * 1e3, "queries_recognised_as_kmers_per_step": mg["recognised"] / gs,
            "hits_per_step": mg["hits_local"],
            "phases_ms_per_step": {"hash_queries": mg["hash_ms"] / gs, "probe_segments": mg["probe_ms"] / gs,
                                   "verify": mg["verify_ms"] / gs, "finalize_sort": mg["fin_ms"] / gs}}
    wl.close()
    del wl

    # ---- the north star's own target as a second block of the same line (secondary_block).  Run when the
    # primary line is the default workload and every rank's GPU has the room (index: 157 GB).
    want_secondary = (not args.no_secondary and args.verify_mode == "auto" and
                      ((args.n, args.k, args.K, args.L) == (10_000_000, 25, 16, 8) or args.force_secondary))
    if want_secondary:
        torch.cuda.empty_cache()
        free_b, _total_b = torch.cuda.mem_get_info(dev_index)
        need_b = 190 * (1 << 30) * args.secondary_db_size // 100_000_000
        ok = torch.tensor([1 if free_b >= need_b else 0], dtype=torch.int32,
                          device=dev if (use_dist and backend == "nccl") else "cpu")
        if use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)          # every rank or none
        if int(ok.item()):
            sec = secondary_block(args, world, rank, dev_index, dev, synth, Engine, torch, hdist, HsError, use_dist,
                                  fence, dist, backend)
            if rank == 0:
                line["secondary"] = sec
        elif rank == 0:
            line["secondary"] = {"skipped": "less than 190 GB of HBM free on a rank (%.0f GB here)" % (free_b / 2**30)}
    if rank == 0:
        print(json.dumps(line), file=result_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
