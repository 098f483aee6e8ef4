/* hs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C ABI of the CPU restatement of the reference hot path (oracle/hs_oracle.cpp).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (hsearch_amd/, include/hsearch.h) never does.
 *
 * Parity status: PINNED.  The reference ships no tests or golden files, so
 * the restatement is pinned against outputs of the reference itself: oracle/_ref (the real
 * reference compiled in this container) and the committed fixtures under tests/golden/ that
 * tools/gen_golden.py produced from it.
 */
#ifndef HS_ORACLE_H
#define HS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* a2 KmerToCoordinates hclust2.cpp:49-62.  codes[n][k] rows of HS_AA_COORDS -> out[n][8k]. */
void hso_embed_codes(const uint8_t* codes, uint64_t n, uint32_t k, double* out);
/* letters -> codes through base[] (util.hpp:92).  Returns the number of letters that have no code
 * (B J O U X Z or non A-Z); those get code 255 (the reference substitutes rand()%20 there,
 * hclust2.cpp:54-56 -- nondeterministic, so the oracle reports instead of guessing). */
uint64_t hso_letters_to_codes(const char* letters, uint64_t n_letters, uint8_t* codes);

/* a4+a5 LSH::DotProduct / HashBucketIndex lsh.hpp:33-49 for one table: a[K][d], b[K], pts[n][d].
 * dots_out[n][K] may be NULL.  buckets_out[n][K]. */
void hso_hash(const double* a, const double* b, uint32_t d, uint32_t K, double W, const double* pts,
              uint64_t n, double* dots_out, int32_t* buckets_out);
/* a6 LSH::HashKey lsh.hpp:51-59: decimal strings of the K ints concatenated, no separator.
 * Writes a NUL-terminated string into out (cap bytes); returns its length (without NUL). */
uint32_t hso_key_string(const int32_t* buckets, uint32_t K, char* out, uint32_t cap);

/* a7..a10 Search() motif_both_points.cpp:195-250 with explicit planes a[L][K][d], b[L][K].
 * The index keeps the reference's cost structure (heap vector<double> per point,
 * unordered_map<string, vector<uint32_t>> per table). */
typedef struct hso_index hso_index;
hso_index* hso_index_build(const double* a, const double* b, uint32_t d, uint32_t K, uint32_t L,
                           double W, const double* db, uint64_t n);
void hso_index_free(hso_index* ix);
/* number of distinct keys in table l (the reference prints it, motif_both_points.cpp:217) */
uint64_t hso_index_table_size(const hso_index* ix, uint32_t l);
/* Query loop motif_both_points.cpp:224-245.  Hits come out in the reference's order (query, then
 * table of first sight, then ascending DB id).  hit_* arrays have room for cap entries; the return
 * value is the number of hits that exist (may exceed cap; only cap are written).
 * cand_out[nq][L] (may be NULL) receives |B_l(q)|, the bucket population before dedupe. */
uint64_t hso_index_query(hso_index* ix, const double* centers, uint64_t nq, double R,
                         uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table, double* hit_dist,
                         uint64_t cap, uint64_t* cand_out);
/* EXTENSION (the reference is single-threaded): the same loop over `threads` host threads, identical
 * results in identical order; for the CPU baseline's all-cores figure only. */
uint64_t hso_index_query_mt(hso_index* ix, const double* centers, uint64_t nq, double R, uint32_t threads,
                            uint32_t* hit_q, uint32_t* hit_id, uint32_t* hit_table, double* hit_dist,
                            uint64_t cap);

/* a10 hits writer motif_both_points.cpp:240-241: "<qname> <dbname> <dist>\n", dist with ostream
 * default formatting (6 significant digits).  names may be NULL -> decimal indices. */
int hso_write_hits(const char* path, const uint32_t* hit_q, const uint32_t* hit_id,
                   const double* hit_dist, uint64_t n_hits, const char* const* q_names,
                   const char* const* db_names);

/* a9 PairwiseDistance_square motif_both_points.cpp:176-183 : out[nq][n] squared distances. */
void hso_pairwise_square(const double* db, uint64_t n, const double* centers, uint64_t nq,
                         uint32_t d, double* out);
/* a11 brute force Search() motif_both_points_noLSH.cpp:36-56: every (q, j) with
 * !(sqrt(d2) > R), query-major, ascending j. Returns the count; writes up to cap. */
uint64_t hso_bruteforce(const double* db, uint64_t n, const double* centers, uint64_t nq,
                        uint32_t d, double R, uint32_t* hit_q, uint32_t* hit_id, double* hit_dist,
                        uint64_t cap);
/* Brute-force k nearest per query (ground truth of recall@k, SURVEY 8d: ties by lower id).
 * nn_id[nq][topk], nn_dist2[nq][topk]. */
void hso_bruteforce_topk(const double* db, uint64_t n, const double* centers, uint64_t nq,
                         uint32_t d, uint32_t topk, uint32_t* nn_id, double* nn_dist2);

/* a12 Clustering() hclust2.cpp:86-151 with explicit planes a[L][K][d], b[L][K]; pts[n][d].
 * merged_out[n] in {0,1,2}; owner_out[n] = the center a point was absorbed by (itself if not
 * absorbed).  Bucket visiting order is std::unordered_map<std::string,...> iteration order, as in
 * the reference (toolchain-pinned: same libstdc++ => same order). */
void hso_clustering(const double* a, const double* b, uint32_t d, uint32_t K, uint32_t L, double W,
                    double R, const double* pts, uint64_t n, uint8_t* merged_out,
                    uint32_t* owner_out);
/* clusters writer hclust2.cpp:137-150.  Member order inside a cluster = absorption order, which
 * hso_clustering records internally; this re-runs it and writes the file. */
int hso_clustering_to_file(const double* a, const double* b, uint32_t d, uint32_t K, uint32_t L,
                           double W, double R, const double* pts, uint64_t n, const char* path);

/* a13 weight()/evaulate() motif_both_points.cpp:67-165 on two hits files (ground truth must be
 * sorted by (motif, protein) as evaluate2.cpp:88-96 leaves it).  Returns tp/(tp+fn). */
double hso_evaluate(const char* ground_truth, const char* hits, double R);

/* a11 as a program: Search() of motif_both_points_noLSH.cpp:36-56, hits to out_path and the
 * excluded pairs to out_path + "notlessthan.txt". */
int hso_bruteforce_to_files(const double* db, uint64_t n, const double* centers, uint64_t nq, uint32_t d,
                            double R, const char* out_path, const char* const* q_names,
                            const char* const* db_names);

/* ---- evaluation tooling and centroid queries ---------------------------------
 * evaluate2.cpp:73-95 (sorted copy <path>sort.txt; returns the record count, -1 on error),
 * its weight() :62-71 and the comparison :98-153. */
int64_t hso_sort_hits_file(const char* path);
double hso_evaluate2_weight(double dis);
double hso_evaluate2(const char* ground_truth, const char* hits, double* tp_out, double* fn_out);
/* centerDistanceSmapling.cpp: Center() of the members' embeddings (:41-78), the points file of
 * cluster2datapoint() (:126-134), the two distance files of sequencedatabase2centers() (:138-190). */
void hso_family_centers(const uint8_t* codes, const uint32_t* first, uint32_t n_families, uint32_t k,
                        double* centers);
int hso_write_points_file(const char* path, const char* const* names, const double* pts, uint64_t n,
                          uint32_t d);
int hso_center_sampling(const double* db, uint64_t n, const double* centers, uint64_t nc, uint32_t d,
                        const char* inner_path, const char* random_path);

/* ---- Kernel-LSH pre-grouping of whole proteins (pcluster) ------------------
 * Planes of KLSH::KLSH (lsh.cpp:17-38): per bit t ~ U(-1,1), b ~ U(0, 2 pi), then feat normals
 * N(0, sigma*sigma) -- sigma*sigma is passed as the STANDARD DEVIATION, lsh.cpp:22 -- from one
 * default-seeded std::default_random_engine (lsh.hpp:49).  w[bits][feat], b[bits], t[bits]. */
void hso_klsh_draw_planes(uint32_t feat, uint32_t bits, double sigma, double* w, double* b, double* t);
/* Feature vector of one protein (pcluster.cpp:27-33): counts of its 3-mers over the reduced
 * classes (0..7 per residue), index = c0 + 8 c1 + 64 c2.  feat[512]; len >= 3. */
void hso_klsh_features(const uint8_t* classes, uint64_t len, double* feat);
/* KLSH::GetHashValue (lsh.cpp:40-49): bit i = (cos(Dot(p, w_i) + b_i) + t_i >= 0), Dot strictly
 * left to right (lsh.cpp:8-15). */
uint64_t hso_klsh_hash(const double* w, const double* b, const double* t, uint32_t feat, uint32_t bits,
                       const double* p);

#ifdef __cplusplus
}
#endif
#endif
