/* hsearch.h -- C ABI of the MI355X-native motif-search hot path (libhsearch_amd.so).
 *
 * The reference (acgtun/hsearch) has no FFI or plugin interface: its operator surface is the set of
 * C++ free functions and the LSH class in hclust/src/hclust/ (SURVEY.md 8b).  This header is the
 * boundary a maintainer of the reference would bind instead of those functions; each entry point
 * names the reference interface it replaces.  INTEGRATION.md shows the reference-side call sites.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every function returns an hs_status and never
 *     throws; hs_last_error() gives the message of the last failure on a handle;
 *   - the caller owns every buffer it passes in; the library owns device memory behind the handle;
 *   - functions WITHOUT a _dev suffix take HOST pointers and copy over PCIe; the _dev variants
 *     take DEVICE pointers (HBM-resident inputs/outputs) and are what bench.py times.  The library
 *     works on streams of its own and returns with them drained: device buffers a caller hands in must
 *     be COMPLETE (whatever the caller has queued on its own streams to fill them: finished) at the
 *     call, and are ready for any stream when it returns;
 *   - output capacity is explicit: when results exceed `cap` the call returns HS_ERR_CAPACITY and
 *     *n_out holds the required capacity (two-call pattern);
 *   - a handle is bound to one GPU and is not thread-safe; use one handle per GPU / process;
 *   - requires a gfx950 device: there is no CPU fallback, calls fail with HS_ERR_NO_DEVICE.
 *
 * Layouts (all row-major, densely packed)
 *   codes   [n][k]      uint8   rows of the coordinate table (0..alphabet-1), see hs_tables.h
 *   points  [n][d]      double  d = 8*k
 *   planes  a[L][K][d], b[L][K] double  (LSH::a, LSH::b of table l -- lsh.hpp:65-66)
 *   buckets [n][L][K]   int32
 */
#ifndef HSEARCH_H
#define HSEARCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_API __attribute__((visibility("default")))

typedef enum hs_status {
  HS_OK = 0,
  HS_ERR_INVALID = 1,       /* bad argument */
  HS_ERR_NO_DEVICE = 2,     /* no usable gfx950 device */
  HS_ERR_HIP = 3,           /* a HIP runtime call failed (see hs_last_error) */
  HS_ERR_CAPACITY = 4,      /* output buffer too small; required size reported */
  HS_ERR_STATE = 5,         /* e.g. query before hs_index_build */
  HS_ERR_KEY_COLLISION = 6, /* 64-bit key fingerprints collided for every retry seed */
  HS_ERR_NOMEM = 7,
  HS_ERR_IO = 8,            /* index file missing, truncated, or written for other parameters */
  HS_ERR_PEER = 9           /* hsearch_dist.h: ANOTHER rank of the communicator failed before the exchange;
                               nothing was exchanged (the failed rank returns its own status) */
} hs_status;

/* Replaces the (dimension, hash_K, hash_W) arguments of LSH::LSH (lsh.hpp:10-17) and the
 * (hash_K, hash_L, hash_W) arguments of Search()/Clustering() (motif_both_points.cpp:195-203,
 * hclust2.cpp:86-91).  DIMENSION = AACoordinateSize * KMERLENGTH (motif_both_points.cpp:337-338). */
typedef struct hs_params {
  uint32_t k;      /* residues per k-mer (KMERLENGTH); d = 8*k */
  uint32_t K;      /* hash functions per table (hash_K), 1..32 */
  uint32_t L;      /* hash tables (hash_L), 1..32 */
  double W;        /* bucket width (hash_W) */
  int32_t device;  /* HIP device ordinal */
  uint32_t alphabet; /* rows of the coordinate table, 1..32; 0 means 20 (the amino acids) */
} hs_params;

typedef struct hs_handle hs_handle;

/* Phase timings (HIP events on the library's stream) and counters of the LAST call on a handle. */
typedef struct hs_profile {
  double ms_hash;        /* projection + bucket ints + key fingerprints */
  double ms_sort;        /* build: radix sort + directory */
  double ms_gather;      /* build: bucket-ordered packed-code copies */
  double ms_probe;       /* query: directory lookup */
  double ms_verify;      /* query: candidate scan kernel (the dominant kernel) */
  double ms_finalize;    /* query: dedupe + exact fp64 distance + ordering of hits */
  double ms_total;       /* whole call, device side */
  uint64_t candidates;   /* sum over (q, l) of |B_l(q)| scanned by the last query call */
  uint64_t provisional;  /* candidates that passed the fp32 filter */
  uint64_t hits;
  uint64_t verify_launches;
  uint64_t join_batches;   /* query batches in which the MFMA bucket join ran */
  double ms_join;          /* of ms_verify: the hs_join_kernel part (rest = streaming kernel) */
  uint64_t join_items;     /* (member tile, query group) work items of the join */
  uint64_t join_pairs;     /* (member, query) pairs routed to the join */
  uint64_t join_pairs_issued; /* MFMA rows x columns actually issued for them (padding included) */
  uint64_t join_i8_batches;   /* of join_batches: those run by the int8 form (hs_join8_kernel) */
  uint64_t hash_values;       /* bucket ints produced by the MFMA projection pass (hs_proj_kernel) ... */
  uint64_t hash_flagged;      /* ... of which this many lay within the error bound of a bucket boundary
                                 and were recomputed in the reference's fp64 order (hs_proj_fix_kernel) */
  uint32_t join_row_bytes;    /* int8 join of the last batch: bytes of a row = GEMM depth (128 / 192 / 256; 0:
                                 it did not run) ... */
  uint32_t join_wide;         /* ... and 1 if the rows carried all 8 coordinate columns (short k-mers, large
                                 radii), 0 for the 4 filter columns */
  uint64_t join_items_resident; /* of join_items: those of segments with few probing queries, run by the
                                   query-resident kernel (hs_join8r_kernel) */
  uint64_t join_async_retries; /* batches whose join was launched on a capacity hint that turned out too small
                                  (or illegal) and ran a second time: exclude such a call from kernel timings */
  uint64_t queries_recognised; /* hs_query / hs_query_dev: the call's centres were all k-mers (every 8 doubles a
                                  row of the coordinate table) and ran from their residue codes: this many */
} hs_profile;

typedef struct hs_index_info {
  uint64_t n;               /* DB k-mers */
  uint64_t device_bytes;    /* HBM held by the index */
  uint64_t n_buckets[32];   /* distinct keys of table l (the reference prints this, :217) */
  uint64_t max_bucket[32];  /* population of the largest bucket of table l */
  uint32_t key_seed;        /* fingerprint seed that produced a collision-free directory */
} hs_index_info;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Replaces L constructions of LSH (lsh.hpp:10-31), except that the planes are an INPUT: the
 * reference draws them from std::random_device (lsh.hpp:19-20), which no caller can reproduce.
 * coords: [alphabet][8] embedding table, or NULL for HS_AA_COORDS (util.hpp:21-42).  A caller whose
 * DB arrives as points (the reference's points files carry the table rounded to 6 significant
 * digits, protein2datapoints.cpp:23-29) passes the distinct 8-tuples it found as the table and the
 * row indices as codes, so both sides hash exactly the same doubles. */
HS_API hs_status hs_create(const hs_params* params, const double* a, const double* b,
                           const double* coords, hs_handle** out);
HS_API void hs_destroy(hs_handle* h);
/* A new hash family (same k, K, L, W) for an existing handle: what constructing the next LSHTable
 * does in Clustering() (hclust2.cpp:104, one fresh family per table).  Drops the index (queries
 * return HS_ERR_STATE until the next build); device buffers are kept. */
HS_API hs_status hs_set_planes(hs_handle* h, const double* a, const double* b);
HS_API const char* hs_last_error(const hs_handle* h);
HS_API hs_status hs_get_profile(const hs_handle* h, hs_profile* out);
/* The parameters the handle was created with (alphabet resolved to the row count in use). */
HS_API hs_status hs_get_params(const hs_handle* h, hs_params* out);
/* Candidate-verification kernel: 0 = auto (bucket join when legal, else streaming), 1 = streaming
 * scan (hs_verify_kernel), 2 = MFMA bucket join wherever it is legal (int8 hs_join8_kernel, else
 * fp16 hs_join_kernel), 3 = the fp16 join only.  All are filters in front of the same exact fp64
 * decision, so results are identical; the environment variable HS_VERIFY_MODE=stream|join|join16
 * sets the default of new handles. */
HS_API hs_status hs_set_verify_mode(hs_handle* h, int mode);
/* How LSH::HashBucketIndex (lsh.hpp:33-49) is evaluated: 0 = auto (the int8 MFMA projection with a
 * proven error bound, values within the bound of a bucket boundary recomputed in the reference's
 * fp64 order -- hs_proj.hip -- unless the bound is so wide that most values would be recomputed,
 * or k > 52), 1 = the exact fp64 vector-ALU kernel only, 2 = the MFMA pass wherever it is compiled
 * (k <= 52).  Bucket integers are bit-identical in every mode.  eps_scale >= 1 inflates the error
 * bound (tests: more values take the recompute path); 1 is the proven bound.  The environment
 * variable HS_HASH_MODE=exact|mfma sets the default of new handles. */
HS_API hs_status hs_set_hash_mode(hs_handle* h, int mode, double eps_scale);
/* Path selection and batch sizing of a handle.  No option changes a result: each forces one of several
 * equivalent paths (the tests run both and compare) or sizes a batch.  Unknown option / value out of range:
 * HS_ERR_INVALID.  Options that shape the index (HS_OPT_BUILD_GROUPING) take effect at the next build. */
typedef enum hs_option {
  HS_OPT_QUERY_BATCH = 1,    /* queries per internal batch of a query call; 0 (default) = by L and free HBM */
  HS_OPT_SEG_MODE = 2,       /* grouping of the probes by bucket: 0 by the bucket : probe ratio, 1 sort the
                                probes, 2 counting sort over the bucket slots */
  HS_OPT_JOIN_RESIDENT = 3,  /* segments with few probing queries through hs_join8r_kernel: 0 by their share of
                                the previous batches' work items, 1 never, 2 always */
  HS_OPT_RECOGNISE_KMERS = 4,/* 1 (default): centres that are rows of the coordinate table run from their residue
                                codes; 0: always as points */
  HS_OPT_BUILD_GROUPING = 5, /* 0 (default): exact-membership table + radix sort on bucket ranks; 1: radix sort of
                                (fingerprint, id) pairs */
  HS_OPT_WIDE_ROWS = 6,      /* int8 rows over all 8 coordinate columns for k = 21..25: 0 by radius, 1 always,
                                2 never by radius; 3: 4-column rows for k <= 20 as well (drops the index: the
                                member records are built for one form) */
  HS_OPT_REFINE8 = 7,        /* 1 (default): survivors of the 4-column bound pass the 8-column bound first */
  HS_OPT_SELF_CODES = 8,     /* 1 (default): the self-join runs from residue codes; 0: from embedded centres */
  HS_OPT_SORT_HITS = 10,     /* 1: order a batch's hits by a radix sort of the whole list, not per query */
  HS_OPT_SYNC_ITEMS = 11,    /* 1: read the join's work-item count back before launching it */
  HS_OPT_JOIN_MIN_Q = 12,    /* segments with fewer probing queries ... */
  HS_OPT_JOIN_MIN_M = 13,    /* ... or fewer members (and fewer than 512) go to the streaming filter instead of the
                                join (default 1 / 1: none do) */
  HS_OPT_SORT_FROM_BIT = 14, /* HS_OPT_BUILD_GROUPING = 1: lowest fingerprint bit the first sort looks at (0..60) */
  HS_OPT_BUILD_SERIAL = 15,  /* 1: no overlap of a table's hashing with the previous table's grouping */
  HS_OPT_PROBE_RECORDS = 17, /* 1 (default): a probe reads one 64-byte directory record per bucket (fingerprint,
                                boundaries, the bucket ints as int16) where the index has them -- K <= 24 and
                                every bucket int of the table within 16 bits; 0: the directory arrays */
  HS_OPT_JOIN_CHUNK = 18,    /* work items a wave of hs_join8x_kernel takes per access to the item counters (2..64);
                                0 (default): from the previous batch's pairs per item */
  HS_OPT_JOIN_XCD_RUN = 16   /* hs_join8x_kernel's work items dealt in runs of this many chunks per XCD, each XCD's
                                waves on their own runs (a run's items stream the same query tiles: one L2 fetches
                                them instead of eight).  0: one counter for the chip; -1 (default): by the size
                                of the batch's query-tile array */
} hs_option;
HS_API hs_status hs_set_option(hs_handle* h, int option, int64_t value);
/* Bucket partition -- unlike the options above this CHANGES what a query call returns.  With n_parts > 1 the
 * searches of this handle (hs_query*, not the self-joins) probe only the buckets that fall to `part` of
 * `n_parts` (a fixed function of the probe's K bucket ints, the same on every handle; for the few buckets of
 * more than max(4096, n / 1024) members a function of the bucket ints AND of the query's number in the call, so
 * that a giant bucket's queries are shared among the parts instead of the bucket landing on one of them): the
 * loop over tables and buckets of motif_both_points.cpp:224-238 cut by BUCKET.
 * Every (query, table) probe belongs to exactly one part -- provided every part is given the same queries in
 * the same order --, so the union over the parts of the hits is the full call's hits plus
 * later-table sightings of ids an earlier table of another part already had: hs_merge_first_table_dev (per
 * (query, id) the smallest table, order (query, table, id)) of the parts' lists IS the full call's output.  n
 * GPUs with the index replicated and ALL queries on every GPU then share the buckets instead of the queries:
 * each meets 1/n of the probes with all the queries there are per bucket (hs_comm_query_buckets in
 * include/hsearch_dist.h).  n_parts = 1 (part 0): everything, the default. */
HS_API hs_status hs_set_bucket_partition(hs_handle* h, uint32_t part, uint32_t n_parts);
/* The library's work after this call starts only once `hip_event` (a hipEvent_t the caller has recorded on a
 * stream of its own) has completed: the device-side alternative to draining that stream before a _dev call. */
HS_API hs_status hs_wait_event(hs_handle* h, void* hip_event);
HS_API const char* hs_version(void);

/* ---- embedding + hashing (rows a2, a4, a5, a6) ----------------------------------------------- */

/* KmerToCoordinates (hclust2.cpp:49-62) for n k-mers given as codes: out[n][d]. */
HS_API hs_status hs_embed_codes(hs_handle* h, const uint8_t* codes, uint64_t n, double* out);

/* LSH::HashBucketIndex (lsh.hpp:44-49) for every (point, table, function): bit-exact ints,
 * strict left-to-right fp64 with separate multiply and add as lsh.hpp:33-42 evaluates it. */
HS_API hs_status hs_hash_codes(hs_handle* h, const uint8_t* codes, uint64_t n, int32_t* buckets);
HS_API hs_status hs_hash_points(hs_handle* h, const double* points, uint64_t n, int32_t* buckets);

/* LSH::HashKey (lsh.hpp:51-59): decimal strings of K ints concatenated without separator.
 * Host-side helper; returns the length, writes a NUL-terminated string of at most cap-1 chars. */
HS_API uint32_t hs_key_string(const int32_t* buckets, uint32_t K, char* out, uint32_t cap);
/* Diagnostics (host-side, no GPU): the 64-bit fingerprint of that character stream under which the
 * index groups keys, and the exact HashKey string equality of two K-tuples (the index never trusts
 * the fingerprint alone: it re-checks equality at build and at probe time). */
HS_API uint64_t hs_key_fingerprint(const int32_t* buckets, uint32_t K, uint32_t seed);
HS_API int hs_key_strings_equal(const int32_t* x, const int32_t* y, uint32_t K);

/* ---- index build (row a7) --------------------------------------------------------------------- */

/* Replaces the build loop of Search() (motif_both_points.cpp:206-218) / BuildLSHTalbe
 * (hclust2.cpp:74-84): L tables keyed by HashKey string equality, ids ascending inside a bucket.
 * The DB is kept as residue codes (k bytes per k-mer), never as 8k doubles. */
HS_API hs_status hs_index_build(hs_handle* h, const uint8_t* codes, uint64_t n);

/* Index over a SUBSET of a code array that stays the same across calls -- what BuildLSHTalbe does
 * per table inside Clustering() (hclust2.cpp:74-84: the k-mers with merged != 2).  codes_all
 * [n_all][k] crosses PCIe on the first call with this (pointer, n_all) and is kept on the device
 * (the caller must not change it while it keeps calling; a call with another pointer or n_all
 * replaces the copy); every call gathers the rows subset[0 .. n_subset) on the device -- DB id i of
 * the new index = row subset[i]; subset == NULL means all rows in order -- and builds the index. */
HS_API hs_status hs_index_build_subset(hs_handle* h, const uint8_t* codes_all, uint64_t n_all,
                                       const uint32_t* subset, uint64_t n_subset);

/* SURVEY 8(f) row 1 -- k-mer enumeration on the device.  The DB is every length-k window of every
 * sequence of one concatenated residue-code buffer: sequence s occupies residues[seq_start[s] ..
 * seq_start[s+1]), seq_start has n_seq + 1 ascending entries ending at n_residues.  Windows are
 * numbered the way kmer_search.cpp:64-83 walks them (sequence-major, ascending offset; they do not
 * cross sequence boundaries; sequences shorter than k contribute none), and that number is the DB
 * id every other call reports.  *n_windows receives their count; window_pos (optional, one uint32
 * per window -- call once with NULL to learn the count) receives each window's start position in
 * the buffer.  Replaces the host loop of BuildLSHTalbe(prodb, ...) kmer_search.cpp:64-83 together
 * with the build itself; the index is identical to hs_index_build over the materialised windows. */
HS_API hs_status hs_index_build_windows(hs_handle* h, const uint8_t* residues, uint64_t n_residues,
                                        const uint64_t* seq_start, uint64_t n_seq,
                                        uint64_t* n_windows, uint32_t* window_pos);
/* SURVEY 8(e), "Index build" row -- the build loop of Search() (motif_both_points.cpp:212-218) with the
 * evaluation of the hash functions SPREAD OVER RANKS (one handle per rank = GPU).  The index is replicated,
 * so every rank is given all n k-mers; rank r evaluates the L x K functions for its contiguous block of
 * them only (hs_shard_bounds' rule: *block_lo, *block_count), the ranks all-gather 8-byte fingerprints,
 * every rank groups all of them, and the exact HashKey-string membership proof of a k-mer is made by the
 * rank that hashed it, against the bucket's tuple -- the bucket ints of the bucket's first member,
 * contributed by the rank that hashed THAT k-mer.  The caller does the collectives (RCCL, or anything
 * else); every pointer with a d_ prefix is device memory of the handle's GPU.  Per table l = 0 .. L-1:
 *   hs_index_shard_hash_dev(h, l, seed, d_fp_block[block_count])     fingerprints of the rank's block
 *   <all-gather: d_fp_all[n], blocks in rank order>
 *   hs_index_shard_group_dev(h, l, d_fp_all, &nb)                    the table's buckets: nb of them
 *   hs_index_shard_tuples_dev(h, l, d_tuples[nb][K])                 zeros but for the buckets whose first
 *                                                                    member lies in the rank's block
 *   <sum over ranks (all-reduce): d_tuples_all[nb][K]>
 *   hs_index_shard_finish_dev(h, l, d_tuples_all, &collided)         proof of the own block, bucket-ordered copies
 * and after the last table <max over ranks of collided over all tables>: if set (two HashKey strings
 * under one fingerprint), start over from table 0 with seed + 1 (hs_index_build tries seeds 0..3); else
 * hs_index_shard_end(h, seed).  The index equals hs_index_build's bit for bit (same file from
 * hs_index_save).
 * STREAMS: every one of these calls works on the handle's own stream and returns with it drained.  A d_
 * buffer the caller fills between two calls (the gathered fingerprints, the summed tuples: outputs of the
 * caller's collectives on the caller's stream) must be COMPLETE when it is handed over -- drain that stream,
 * or record an event behind the collective and pass it to hs_wait_event(h, event) before the call; buffers
 * the library writes (d_fp_block, d_tuples) are complete when the call returns. */
HS_API hs_status hs_index_shard_begin(hs_handle* h, const uint8_t* codes, uint64_t n, uint32_t rank,
                                      uint32_t world, uint64_t* block_lo, uint64_t* block_count);
HS_API hs_status hs_index_shard_hash_dev(hs_handle* h, uint32_t l, uint32_t seed, uint64_t* d_fp_block);
HS_API hs_status hs_index_shard_group_dev(hs_handle* h, uint32_t l, const uint64_t* d_fp_all, uint32_t* n_buckets);
HS_API hs_status hs_index_shard_tuples_dev(hs_handle* h, uint32_t l, int32_t* d_tuples);
HS_API hs_status hs_index_shard_finish_dev(hs_handle* h, uint32_t l, const int32_t* d_tuples_all,
                                           uint32_t* collided);
HS_API hs_status hs_index_shard_end(hs_handle* h, uint32_t key_seed);

/* SURVEY 8(f) row 2 -- persistent index (no reference analogue: the reference rebuilds its tables
 * on every run, motif_both_points.cpp:206-218).  hs_index_save writes parameters, planes,
 * coordinate table, residue codes and the L tables (ids + bucket directory) of a built handle;
 * hs_index_load restores them into a handle created with the SAME parameters, planes and table
 * (checked bit for bit, HS_ERR_IO otherwise) and re-derives the bucket-ordered copies on the
 * device: queries then give exactly what they give after hs_index_build.
 * The file carries its payload's length and a 64-bit hash; hs_index_load checks them and, before any
 * kernel indexes with a table, the table's content on the device (ids a permutation of 0..n-1,
 * ascending inside a bucket; boundaries strictly ascending from 0 to n; fingerprints strictly
 * ascending and equal to the fingerprint of the bucket's tuple; bucket sizes recomputed): a corrupt,
 * truncated, stale or hand-edited file yields HS_ERR_IO, never an out-of-bounds access.
 * hs_index_file_check runs the same checks on the HOST (no GPU, no handle): HS_OK or HS_ERR_IO with
 * a message in err. */
HS_API hs_status hs_index_save(hs_handle* h, const char* path);
HS_API hs_status hs_index_load(hs_handle* h, const char* path);
HS_API hs_status hs_index_file_check(const char* path, char* err, uint32_t err_cap);

/* SURVEY 8(f) row 3 -- Kernel-LSH pre-grouping of whole proteins (pcluster.cpp:11-81).
 * hs_klsh_draw_planes: the planes KLSH::KLSH draws (lsh.cpp:17-38) from its default-seeded
 * std::default_random_engine (lsh.hpp:49) -- per bit t ~ U(-1,1), b ~ U(0, 2 pi), then `feat`
 * normals with standard deviation sigma*sigma (sic, lsh.cpp:22).  Host only; w[bits][feat].
 * hs_klsh_codes: one hash code per sequence of a concatenated buffer of reduced-alphabet classes
 * (0..7 per residue, include/hs_tables.h HS_REDUCED_CLASS): feature vector = counts of the
 * sequence's 3-mers (pcluster.cpp:27-33), bit i = (cos(Dot(p, w_i) + b_i) + t_i >= 0) with Dot
 * strictly left to right in fp64 (KLSH::GetHashValue lsh.cpp:40-49).  Sequences shorter than 3 get
 * HS_KLSH_NONE (the reference skips them, pcluster.cpp:22-24).  uncertain (optional, [n_seq])
 * receives a mask of the bits whose |cos(.) + t| is below 1e-9, i.e. where the device's cos and
 * libm's could disagree on the sign.  Runs on `device`; status only (no handle): err, if given,
 * receives a message. */
#define HS_KLSH_NONE 0xffffffffffffffffull
HS_API hs_status hs_klsh_draw_planes(uint32_t feat, uint32_t bits, double sigma, double* w, double* b,
                                     double* t);
HS_API hs_status hs_klsh_codes(int device, const uint8_t* classes, uint64_t n_residues,
                               const uint64_t* seq_start, uint64_t n_seq, const double* w,
                               const double* b, const double* t, uint32_t bits, uint64_t* codes,
                               uint64_t* uncertain, char* err, uint32_t err_cap);

HS_API hs_status hs_index_info_get(const hs_handle* h, hs_index_info* out);

/* ---- query = probe + dedupe + verify (rows a8, a9, a10) ---------------------------------------- */

/* Replaces the query loop of Search() (motif_both_points.cpp:224-245).  centers[nq][d] are
 * arbitrary points of R^d.  A hit is (query, DB id, table of first sight, sqrt(d2)) with
 * d2 = sum (x_i - c_i)^2 evaluated left to right in fp64 (motif_both_points.cpp:176-183) and
 * d2 <= R*R (:239).  Hits are returned in the reference's output order: query, then table of
 * first sight, then ascending DB id.  cand[nq][L] (may be NULL) receives |B_l(q)|. */
HS_API hs_status hs_query(hs_handle* h, const double* centers, uint64_t nq, double R,
                          uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table, double* hit_dist,
                          uint64_t cap, uint64_t* n_hits, uint64_t* cand);
/* Same with every pointer except n_hits in device memory (HBM-resident queries and hits).
 * STREAMS (this and every other _dev entry point): the library works on the handle's own stream.  d_centers
 * must be complete when the call is made (the caller's producer stream drained, or an event recorded behind
 * the producer handed to hs_wait_event(h, event) first); the call returns with the library's stream drained,
 * so the outputs are complete and the inputs may be reused at once, from any stream. */
HS_API hs_status hs_query_dev(hs_handle* h, const double* d_centers, uint64_t nq, double R,
                              uint32_t* d_hit_q, uint32_t* d_hit_id, uint32_t* d_hit_table,
                              double* d_hit_dist, uint64_t cap, uint64_t* n_hits,
                              uint64_t* d_cand);

/* The same search for queries that ARE k-mers -- the usual centres of the reference's own pipeline:
 * hclust2 embeds k-mer strings (KmerToCoordinates, hclust2.cpp:49-62) and `motif_both_points -c`
 * is fed k-mers embedded exactly from the table -- given as residue codes qcodes[nq][k] (rows of the
 * coordinate table, like the DB's): k bytes per query across PCIe instead of 8d = 64 k.  Results are
 * those of hs_query on the embedded codes, bit for bit (hash, filter rows and the exact fp64 distance
 * all read the table rows an embedded centre would hold).  A code outside the alphabet is
 * HS_ERR_INVALID.  hs_query_codes_dev: every pointer except n_hits in device memory. */
HS_API hs_status hs_query_codes(hs_handle* h, const uint8_t* qcodes, uint64_t nq, double R,
                                uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table, double* hit_dist,
                                uint64_t cap, uint64_t* n_hits, uint64_t* cand);
HS_API hs_status hs_query_codes_dev(hs_handle* h, const uint8_t* d_qcodes, uint64_t nq, double R,
                                    uint32_t* d_hit_q, uint32_t* d_hit_id, uint32_t* d_hit_table,
                                    double* d_hit_dist, uint64_t cap, uint64_t* n_hits,
                                    uint64_t* d_cand);

/* The merge step of the TABLE-partitioned multi-GPU layout (hsearch_dist.h hs_comm_query_tables): every rank
 * holds some of the L tables over ALL k-mers and answers ALL queries, so a (query, id) pair is reported by
 * every rank whose tables hold the id in the query's bucket, each time with the smallest of that rank's
 * tables (in GLOBAL table numbers).  The reference reports an id in the FIRST table whose probed bucket
 * holds it and never looks at it again (label[], motif_both_points.cpp:232-238); whether it is a hit does not
 * depend on the table.  So of the n gathered tuples the one with the smallest table per (query, id) is the
 * reference's line: this call keeps exactly those, ordered by (query, table, id) -- the reference's file
 * order -- in place in the first *n_out entries of the four device arrays.  q < 2^27, table < 32. */
HS_API hs_status hs_merge_first_table_dev(hs_handle* h, uint32_t* d_q, uint32_t* d_id, uint32_t* d_table,
                                          double* d_dist, uint64_t n, uint64_t* n_out);

/* ---- brute force (row a11) ---------------------------------------------------------------------- */

/* Replaces Search() of motif_both_points_noLSH.cpp:36-56: every (q, j) with !(sqrt(d2) > R),
 * query-major, ascending j (the order of the reference's hits file). */
HS_API hs_status hs_bruteforce(hs_handle* h, const double* centers, uint64_t nq, double R,
                               uint32_t* hit_q, uint32_t* hit_id, double* hit_dist, uint64_t cap,
                               uint64_t* n_hits);
/* Exact k nearest DB k-mers per query (ground truth of recall@k; ties by lower id).
 * nn_id[nq][topk], nn_dist2[nq][topk] (squared, fp64 left-to-right). */
HS_API hs_status hs_bruteforce_topk(hs_handle* h, const double* centers, uint64_t nq,
                                    uint32_t topk, uint32_t* nn_id, double* nn_dist2);

/* ---- all-vs-all near-neighbour graph + greedy clustering (row a12) ------------------------------- */

/* Every ordered pair (i, j), i != j, of indexed k-mers that share a bucket in some table and lie
 * within R of each other; edge_table = the first table in which they share a bucket.  Sorted by
 * (i, table, j).  sqrt_test != 0 selects hclust2's test sqrt(d2) <= R (hclust2.cpp:64-71,119-120)
 * instead of Search()'s d2 <= R*R.  This is the bucket-local member x center distance work of
 * Clustering() (hclust2.cpp:107-132) done as one join per table. */
HS_API hs_status hs_self_join(hs_handle* h, double R, int sqrt_test, uint32_t* edge_i,
                              uint32_t* edge_j, uint32_t* edge_table, double* edge_dist,
                              uint64_t cap, uint64_t* n_edges);

/* The same for the indexed k-mers [first, first + count) only (as the `i` side): the shard of one
 * rank when the join of a table is spread over GPUs. */
HS_API hs_status hs_self_join_range(hs_handle* h, uint64_t first, uint64_t count, double R,
                                    int sqrt_test, uint32_t* edge_i, uint32_t* edge_j,
                                    uint32_t* edge_table, double* edge_dist, uint64_t cap,
                                    uint64_t* n_edges);

/* Replaces Clustering() (hclust2.cpp:86-151) with explicit planes a[L][K][d], b[L][K]: table by
 * table, an LSH table over the not-yet-absorbed k-mers, then greedy leader clustering inside every
 * bucket in ascending id order.  The distance work runs on the GPU (hs_self_join per table), the
 * order-dependent greedy pass on the host.  Outputs: merged[n] in {0 unprocessed, 1 center,
 * 2 absorbed} (hclust2.cpp:93-96), owner[n] = absorbing center (itself if not absorbed),
 * absorbed_table[n] = table in which it was absorbed (0xffffffff if not): members of a cluster in
 * the reference's file order are its center followed by its members sorted by (absorbed_table, id). */
HS_API hs_status hs_clustering(const hs_params* params, const double* a, const double* b,
                               const double* coords, const uint8_t* codes, uint64_t n, double R,
                               uint8_t* merged, uint32_t* owner, uint32_t* absorbed_table,
                               char* err, uint32_t err_cap);

/* Clustering() spread over `world` GPUs: the distance work of a table --
 * the within-bucket join over the not-yet-absorbed k-mers -- is sharded by the `i` side, one
 * contiguous block of the active k-mers per rank; the one exchange step is an all-gather of the
 * edge lists (done by the caller, e.g. over RCCL); the order-dependent greedy pass then runs
 * identically on every rank.  Per table l = 0..L-1 every rank calls
 *   hs_clustering_table_edges(st, l, rank, world, ...)  -> its edges (i, j) in ORIGINAL k-mer
 *                                   numbers, sqrt(d2) <= R, two-call capacity protocol
 *   <all-gather of the edges>
 *   hs_clustering_table_apply(st, l, all edges in any order)   (host only, no GPU)
 * and after the last table hs_clustering_end, which writes the outputs of hs_clustering and frees
 * the state (outputs may be NULL to just free).  hs_clustering is the world = 1 composition.
 * `codes` must stay valid until hs_clustering_end. */
typedef struct hs_cluster_state hs_cluster_state;
HS_API hs_status hs_clustering_begin(const hs_params* params, const double* a, const double* b,
                                     const double* coords, const uint8_t* codes, uint64_t n, double R,
                                     hs_cluster_state** out, char* err, uint32_t err_cap);
HS_API hs_status hs_clustering_table_edges(hs_cluster_state* st, uint32_t l, uint32_t rank,
                                           uint32_t world, uint32_t* edge_i, uint32_t* edge_j,
                                           double* edge_dist, uint64_t cap, uint64_t* n_edges,
                                           char* err, uint32_t err_cap);
HS_API hs_status hs_clustering_table_apply(hs_cluster_state* st, uint32_t l, const uint32_t* edge_i,
                                           const uint32_t* edge_j, uint64_t n_edges);
HS_API hs_status hs_clustering_end(hs_cluster_state* st, uint8_t* merged, uint32_t* owner,
                                   uint32_t* absorbed_table);

#ifdef __cplusplus
}
#endif
#endif /* HSEARCH_H */
