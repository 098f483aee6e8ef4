/* hsearch_dist.h -- multi-GPU layer of the motif-search hot path (libhsearch_dist.so): queries shard
 * across the GPUs of one node, the index is replicated per GPU, and the one exchange step is a
 * variable-length all-gather of hit tuples over RCCL (xGMI) -- SURVEY.md 8(e).
 *
 * The reference is single-threaded and single-device: its query loop
 * (hclust/src/hclust/motif_both_points.cpp:224-245) carries no state from one query to the next
 * except the per-query label[] reset (:225), which is what makes contiguous query blocks per GPU a
 * correct partition, and its hits file (:240-241) is written in query order, which is what the
 * rank-ordered gather restores.
 *
 * Usage: one hs_comm for `world` ranks; every rank makes the same sequence of calls, each from its
 * own host thread (single process, HS_COMM_RCCL_LOCAL / HS_COMM_LOOPBACK) or its own process
 * (hs_comm_create_rank).  A rank's device pointers live on that rank's GPU.  Status codes and the
 * two-call capacity protocol are those of hsearch.h.
 *
 * Failures: a rank whose own part fails (bad argument, allocation, hs_query_dev) still takes part in
 * the exchange -- its status travels with the counts -- and then EVERY rank returns without exchanging
 * data: the failed rank with its own status, the others with HS_ERR_PEER.  No rank is left waiting at a
 * rendezvous or inside a collective because a neighbour returned early.
 */
#ifndef HSEARCH_DIST_H
#define HSEARCH_DIST_H

#include "hsearch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hs_comm hs_comm;

enum {
  HS_COMM_RCCL_LOCAL = 0, /* `world` GPUs of THIS process (ncclCommInitAll), one host thread per rank */
  HS_COMM_LOOPBACK = 1    /* the exchange goes through HOST memory between the rank threads; the pointers of
                             hs_allgather_hits are HOST pointers.  Created without devices it needs no
                             GPU (the CPU tests of the layout / merge logic); created WITH devices --
                             several ranks may name the same one -- hs_comm_query runs every rank's search
                             on its device and exchanges the hits through host memory: the whole rank
                             protocol on a box with fewer GPUs than ranks (hs_motif_both_points
                             --transport loopback; RCCL refuses two ranks on one device). */
};

/* devices[world]: HIP ordinals of the ranks.  HS_COMM_RCCL_LOCAL: distinct, NULL = 0..world-1.
 * HS_COMM_LOOPBACK: NULL = no devices (hs_comm_query is refused), else any valid ordinals. */
HS_API hs_status hs_comm_create(int kind, const int* devices, uint32_t world, hs_comm** out, char* err,
                                uint32_t err_cap);
/* One process per GPU: rank 0 obtains an id (hs_comm_unique_id), the launcher hands it to the
 * other ranks, every rank calls hs_comm_create_rank (ncclCommInitRank).  The returned object
 * serves this rank only (the `rank` argument of the calls below must be the one given here). */
#define HS_COMM_ID_BYTES 128
HS_API hs_status hs_comm_unique_id(char id[HS_COMM_ID_BYTES]);
HS_API hs_status hs_comm_create_rank(const char id[HS_COMM_ID_BYTES], uint32_t rank, uint32_t world,
                                     int device, hs_comm** out, char* err, uint32_t err_cap);
HS_API void hs_comm_destroy(hs_comm* c);
HS_API uint32_t hs_comm_world(const hs_comm* c);
HS_API const char* hs_comm_last_error(const hs_comm* c, uint32_t rank);

/* Contiguous query blocks: rank r of `world` owns queries [*lo, *hi) of n; the first n % world
 * ranks get one more.  (dist.py::shard_bounds is the same rule.) */
HS_API void hs_shard_bounds(uint64_t n, uint32_t world, uint32_t rank, uint64_t* lo, uint64_t* hi);

/* The exchange step.  Every rank contributes n_local hit tuples (q local to its block, id, table,
 * dist -- the outputs of hs_query_dev) and the global number q_offset of its first query; every
 * rank receives all tuples in rank order -- with contiguous blocks that is the reference's file
 * order -- with q made global.  One RCCL all-gather of per-rank records [q | id | table | dist]
 * padded to the largest count (RCCL has no all-gatherv); the counts travel first (host memory
 * between the threads of one process, a 16-byte all-gather between processes).
 * table / out_table may be NULL (both or neither on all ranks).  When the total exceeds cap,
 * nothing is written, *n_total holds the need and HS_ERR_CAPACITY is returned on every rank. */
HS_API hs_status hs_allgather_hits(hs_comm* c, uint32_t rank, const uint32_t* q, const uint32_t* id,
                                   const uint32_t* table, const double* dist, uint64_t n_local,
                                   uint32_t q_offset, uint32_t* out_q, uint32_t* out_id,
                                   uint32_t* out_table, double* out_dist, uint64_t cap,
                                   uint64_t* n_total);

/* Host-thread rendezvous of the ranks of one process (no-op for a hs_comm_create_rank object). */
HS_API hs_status hs_comm_barrier(hs_comm* c, uint32_t rank);

/* Query-sharded search of one rank: copies this rank's block of centres (HOST, [nq_local][d]) to
 * the handle's GPU, runs hs_query_dev, all-gathers the hits over the communicator and returns ALL
 * ranks' hits in HOST buffers, global order, q global.  `h` must be bound to the rank's device and
 * hold the (replicated) index.  Same capacity protocol (cap counts the hits of all ranks). */
HS_API hs_status hs_comm_query(hs_comm* c, uint32_t rank, hs_handle* h, const double* centers,
                               uint64_t nq_local, uint32_t q_offset, double R, uint32_t* hit_q,
                               uint32_t* hit_id, uint32_t* hit_table, double* hit_dist, uint64_t cap,
                               uint64_t* n_total);

/* The same for a block of queries given as residue codes [nq_local][k] (hs_query_codes_dev: k bytes
 * per query to the GPU instead of 8d). */
HS_API hs_status hs_comm_query_codes(hs_comm* c, uint32_t rank, hs_handle* h, const uint8_t* qcodes,
                                     uint64_t nq_local, uint32_t q_offset, double R, uint32_t* hit_q,
                                     uint32_t* hit_id, uint32_t* hit_table, double* hit_dist, uint64_t cap,
                                     uint64_t* n_total);

/* ---- the TABLE-partitioned layout -----------------------------------------------------------------
 * The alternative to query blocks over a replicated index: rank r holds a SUBSET of the L tables over ALL
 * k-mers (a handle created with the planes of those tables only: L_r = its number of tables) and answers ALL
 * queries; per rank the same probes and (member, query) pairs as in the replicated layout, but every bucket
 * meets all queries of the batch at once (the join's operand reuse), and a rank holds -- and builds -- 1/world
 * of the table bytes.  The exchange is the same all-gather of hit tuples; behind it every rank keeps, per
 * (query, id), the tuple with the smallest GLOBAL table number: the reference reports an id in the first table
 * whose probed bucket holds it (label[], motif_both_points.cpp:232-238), and whether it is a hit does not
 * depend on the table -- so the merged list IS the reference's output (hs_merge_first_table_dev, hsearch.h).
 *
 * hs_comm_query_tables: tables[n_tables] = the global numbers of the handle's tables, ascending (n_tables =
 * the handle's L); centers [nq][d] or qcodes [nq][k] (exactly one non-null) = ALL queries, the same on every
 * rank.  Returns all hits in HOST buffers in the reference's order, on every rank; *n_total = their number;
 * cap counts merged hits.  Failure protocol as hs_comm_query.
 * hs_assign_tables: owner[l] = rank of table l, balanced by cost[l] (longest processing time first; cost NULL:
 * equal costs = round robin); deterministic.  A useful cost: the sum over a table's buckets of
 * (k-mers of a DB sample in the bucket)^2 -- proportional to the (member, query) pairs the table's share of
 * the join meets for queries distributed like the DB. */
HS_API hs_status hs_comm_query_tables(hs_comm* c, uint32_t rank, hs_handle* h, const uint32_t* tables,
                                      uint32_t n_tables, const double* centers, const uint8_t* qcodes, uint64_t nq,
                                      double R, uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table,
                                      double* hit_dist, uint64_t cap, uint64_t* n_total);
HS_API void hs_assign_tables(const double* cost, uint32_t L, uint32_t world, uint32_t* owner);

/* ---- the BUCKET-partitioned layout ----------------------------------------------------------------
 * Index replicated as in hs_comm_query, but the ranks share the BUCKETS instead of the queries: every rank
 * answers ALL queries in the buckets that fall to it (hs_set_bucket_partition(h, rank, world), hsearch.h: a
 * function of the probe's bucket ints -- and of the query for the few giant buckets --, so the parts are even
 * whatever the tables look like), the tuples
 * are all-gathered and merged by the same first-seen rule as above.  Per rank 1/world of the probes, members
 * and (member, query) pairs, with every bucket meeting all the queries of the job at once -- the operand reuse
 * query blocks lose as the ranks multiply (motif_both_points.cpp:224-238's loops cut by bucket, not by query).
 * Arguments and results as hs_comm_query_tables (centers or qcodes = ALL queries, the same on every rank; the
 * handle holds all L tables); the handle's partition is set for the call and back to "everything" after it. */
HS_API hs_status hs_comm_query_buckets(hs_comm* c, uint32_t rank, hs_handle* h, const double* centers,
                                       const uint8_t* qcodes, uint64_t nq, double R, uint32_t* hit_q,
                                       uint32_t* hit_id, uint32_t* hit_table, double* hit_dist, uint64_t cap,
                                       uint64_t* n_total);

#ifdef __cplusplus
}
#endif
#endif /* HSEARCH_DIST_H */
