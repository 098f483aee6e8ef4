// hs_host.hpp -- C++ host side above the C ABI (include/hsearch.h): the reference's operator
// surface for the search path, with the same names, argument meaning and error behaviour, so a
// caller of acgtun/hsearch's Search() can switch by changing an include and one extra argument
// (the planes, which the reference draws from std::random_device and therefore cannot share).
//
//   reference                                         here
//   struct Point {vector<double> data;}  (:18-23)     hsearch::Point
//   LSH(dim, K, W) x L, random_device    (lsh.hpp)    hsearch::DrawPlanes(dim, K, L, W, seed)
//   Search(kmers, centers, names..., K, L, W, R, out) hsearch::Search(... , planes, device)
//   evaulate(ground_truth, out, R)       (:100-165)   hsearch::Evaluate
//   points-file reader in main()         (:343-370)   hsearch::ReadPointsFile
// (file:line = hclust/src/hclust/motif_both_points.cpp unless noted.)
#ifndef HS_HOST_HPP
#define HS_HOST_HPP

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

namespace hsearch {

struct Point {
  std::vector<double> data;
};
struct Kmer;

// L x K Gaussian normals a ~ N(0,1) (K*dim per table) and offsets b ~ U[0,W), drawn exactly as the
// reference's LSH constructor does (lsh.hpp:10-31: default_random_engine, normal_distribution,
// uniform_real_distribution, per function dim normals then one uniform, fresh distributions per
// table) with the table-l engine seeded by seed + l instead of random_device.
struct Planes {
  uint32_t dim = 0, K = 0, L = 0;
  double W = 0;
  std::vector<double> a;  // [L][K][dim]
  std::vector<double> b;  // [L][K]
};
Planes DrawPlanes(uint32_t dim, uint32_t K, uint32_t L, double W, uint32_t seed);

// The reference's points file: per record one name line and one line of `dim` doubles (:343-354).
bool ReadPointsFile(const std::string& path, uint32_t dim, std::vector<std::string>* names,
                    std::vector<Point>* points);

// DB points -> residue codes + the coordinate table they are embeddings of.  Every 8-tuple of every
// DB point must be one of at most 32 distinct rows (true for anything KmerToCoordinates or
// protein2datapoints produced, including the 6-significant-digit points files).  Returns false
// with *err set when the points are not embeddings of such an alphabet.
bool PointsToCodes(const std::vector<Point>& pts, uint32_t dim, std::vector<double>* table,
                   std::vector<uint8_t>* codes, std::string* err);

// Search() (:195-250).  Builds the L tables over `kmers`, probes them with every centre, verifies
// candidates by squared Euclidean distance <= R*R and writes "<center> <kmer> <dist>" lines in the
// reference's order (centre, table of first sight, ascending k-mer index).  Runs on `device`
// through libhsearch_amd.so; returns 0, or an hs_status with *err set.  table_sizes (optional)
// receives the number of distinct keys per table, which the reference prints (:217).
int Search(const std::vector<Point>& kmers, const std::vector<Point>& centers,
           const std::vector<std::string>& kmer_names, const std::vector<std::string>& center_names,
           const uint32_t& hash_K, const uint32_t& hash_L, const double& hash_W,
           const double& hash_R, const std::string& output_file, const Planes& planes, int device,
           std::string* err, std::vector<uint64_t>* table_sizes = nullptr);

// How the rank threads of the *Sharded() functions exchange their hits: RCCL over xGMI (one rank per
// GPU; the default), or host memory between rank threads whose `devices` may repeat -- the whole rank
// protocol (barrier, capacity decision, failed-rank handling) on a box with fewer GPUs than ranks,
// which RCCL refuses (`hs_motif_both_points --transport loopback`).  Process-wide; set before the call.
enum ShardTransport { kTransportRccl = 0, kTransportLoopback = 1 };
void SetShardTransport(ShardTransport t);
// What a rank of the *Sharded() functions holds.  kPartitionQueries (the default): the whole index,
// replicated, and a contiguous block of the centres.  kPartitionTables: a subset of the L tables over ALL
// k-mers (tables dealt to the GPUs by an estimate of their join work, hs_assign_tables) and ALL centres; behind
// the all-gather every rank keeps, per (centre, k-mer), the tuple of the smallest table -- the reference's
// first-seen rule (motif_both_points.cpp:232-238) -- so the file is the same (include/hsearch_dist.h).  A GPU
// then holds and builds 1 / n of the table bytes (configs[2]: 23 GB instead of 157 GB) and every bucket meets
// all centres of the batch at once.  kPartitionBuckets: the whole index, replicated, and ALL centres; the GPUs
// share the BUCKETS (hs_set_bucket_partition: a function of the bucket's key fingerprint, giant buckets shared
// by centre) and the lists are merged by the same rule -- 1 / n of the probes and pairs per GPU, every bucket
// with all the centres there are, even parts whatever the tables look like.  Process-wide; set before the call.
enum ShardPartition { kPartitionQueries = 0, kPartitionTables = 1, kPartitionBuckets = 2 };
void SetShardPartition(ShardPartition p);

// Search() spread over the GPUs `devices` of this node: one host thread and one handle
// per GPU, the index replicated (built on every GPU), centre i searched by the rank owning its
// contiguous block (hs_shard_bounds), the hits all-gathered over RCCL (include/hsearch_dist.h) and
// written by rank 0 -- the same file as Search().  use_comm forces the communicator path with a
// single GPU too (`--gpus 1`: RCCL with one rank); Search() is SearchSharded({device}, false).
int SearchSharded(const std::vector<Point>& kmers, const std::vector<Point>& centers,
                  const std::vector<std::string>& kmer_names, const std::vector<std::string>& center_names,
                  const uint32_t& hash_K, const uint32_t& hash_L, const double& hash_W,
                  const double& hash_R, const std::string& output_file, const Planes& planes,
                  const std::vector<int>& devices, bool use_comm, std::string* err,
                  std::vector<uint64_t>* table_sizes = nullptr);

// The planes file `--planes-out` writes and `--planes` reads: binary doubles a[L][K][dim], b[L][K].
bool ReadPlanesFile(const std::string& path, uint32_t dim, uint32_t K, uint32_t L, double W, Planes* planes,
                    std::string* err);
// Centres given as k-mers (">name" line, then the k letters -- the k-mer FASTA hclust2.cpp:231-241
// reads) instead of a points file: embedded exactly from the table (KmerToCoordinates,
// hclust2.cpp:49-62).  A letter outside the 20-letter alphabet or another length is an error.
// codes (optional) receives the same centres as rows of the coordinate table, [n][k].
bool CentersFromKmers(const std::vector<Kmer>& kmers, uint32_t kmer_length, std::vector<std::string>* names,
                      std::vector<Point>* centers, std::string* err, std::vector<uint8_t>* codes = nullptr);

// ---- FASTA database: k-mers enumerated on the device ------------------------
// ProteinDB of protein.hpp:41-71: lines starting with '>' are names, every other non-empty line is
// one whole sequence.  Residues are stored as rows of the coordinate table (include/hs_tables.h);
// letters outside the 20-letter alphabet -- the reference substitutes rand() % 20 for them,
// protein.hpp:58-61 -- are stored as kUnknown and no window may contain one.
// ref_compat_eq_swap reproduces the reference's E <-> Q exchange (it stores AA20[base[c]] and later
// embeds base[] of THAT letter, protein.hpp:62 + kmer_search.cpp:56); the default
// is the correct embedding (E -> Glu, Q -> Gln).
struct ProteinDB {
  static const uint8_t kUnknown = 255;
  std::vector<std::string> name;   // '>' lines, without the '>'
  std::vector<uint64_t> start;     // [n_sequences + 1] into residues
  std::vector<uint8_t> residues;   // all sequences, concatenated
  bool eq_swapped = false;
};
bool ReadProteinFasta(const std::string& path, bool ref_compat_eq_swap, ProteinDB* db);

// The search of kmer_search.cpp over a FASTA database, with motif_both_points' verification and
// output: DB = every length-k window (free of unknown letters) of every sequence, enumerated on the
// device by hs_index_build_windows in kmer_search.cpp:64-83 order; hits are written as
// "<center> <kmer name> <dist>" in Search()'s order, a k-mer being named like protein2datapoints
// names its samples (protein2datapoints.cpp:66): <first token of the protein name>#<sequence
// index>$<offset>@<k letters>*<window number>.  n_windows (optional) receives the DB size.
int SearchProteins(const ProteinDB& db, uint32_t kmer_length, const std::vector<Point>& centers,
                   const std::vector<std::string>& center_names, const uint32_t& hash_K,
                   const uint32_t& hash_L, const double& hash_W, const double& hash_R,
                   const std::string& output_file, const Planes& planes, int device, std::string* err,
                   std::vector<uint64_t>* table_sizes = nullptr, uint64_t* n_windows = nullptr,
                   bool best_per_position = false);
int SearchProteinsSharded(const ProteinDB& db, uint32_t kmer_length, const std::vector<Point>& centers,
                          const std::vector<std::string>& center_names, const uint32_t& hash_K,
                          const uint32_t& hash_L, const double& hash_W, const double& hash_R,
                          const std::string& output_file, const Planes& planes,
                          const std::vector<int>& devices, bool use_comm, std::string* err,
                          std::vector<uint64_t>* table_sizes = nullptr, uint64_t* n_windows = nullptr,
                          bool best_per_position = false, const std::vector<uint8_t>* center_codes = nullptr);
// center_codes (CentersFromKmers): the centres are k-mers of the exact table -- the one a FASTA
// database is embedded from -- and travel to the GPU as residue codes (hs_query_codes: k bytes per
// centre instead of 64 k); the hits are those of the embedded centres, bit for bit.
// best_per_position: what kmer_search.cpp's Search() accumulates in `matches` (:90,113-121) and
// never writes -- for every database window with a hit, its nearest centre: tables ascending,
// centres ascending within a table, replaced only by a strictly smaller distance.  Written as
// "<kmer name> <center> <dist>", windows ascending.  (The reference's own result is not
// observable and is computed from mis-hashed windows, kmer_search.cpp:73-80 reads one residue k
// times; this is the intended reduction over the pinned hit list -- parity unpinned.)

// Protein2Datapoints() (protein2datapoints.cpp:33-78): the sampler that makes the `-d` points file
// of motif_both_points from a FASTA database -- for each of the first num_of_protein_out
// sequences, windows at offsets 0, then +30..49 (30 + rand() % 20) further each time; a window seen
// before is skipped (same stride draw); each new one is written as the name line
// "<first token of the protein name>#<sequence>$<offset>@<letters>*<count>" and the embedding at
// the stream's default precision.  rand() is seeded with `seed` (the reference: time(NULL)).
// Sequences shorter than k are skipped (the reference's unsigned `length - k` wraps, :44) and so
// are windows with a letter outside the alphabet (the reference draws a random residue instead,
// protein.hpp:58-61).  Returns the number of points written, -1 if the file cannot be written.
int64_t Protein2Datapoints(const ProteinDB& db, uint32_t kmer_length, uint32_t num_of_protein_out,
                           const std::string& output_file, uint32_t seed);

// ---- Kernel-LSH pre-grouping of whole proteins (pcluster.cpp:11-81) ---------
// pcluster's FASTA reader (read_proteins.cpp:6-41): '>' lines start a protein, its name is the
// text up to the first space, the following lines are concatenated; letters of the 20-letter
// alphabet are kept, any other alphabetic character is replaced by a residue drawn from a
// generator seeded with unknown_seed (the reference uses rand() % 20, read_proteins.cpp:30-32),
// everything else is dropped.
struct PclusterDB {
  std::vector<std::string> names, seqs;
};
bool ReadPclusterFasta(const std::string& path, uint32_t unknown_seed, PclusterDB* db);
// PreClustering(proteinDB, hash_buckets): KLSH(8^3, 16, 0.2) codes of all proteins with at least 3
// residues, computed on `device` (hs_klsh_codes), grouped: buckets[code] = ascending protein
// indices.  Returns 0 or an hs_status with *err set.
int PreClustering(const PclusterDB& db, int device,
                  std::map<uint64_t, std::vector<uint32_t> >* hash_buckets, std::string* err);

// Clustering() of hclust2.cpp:86-151.  kmers: name + sequence (letters of the 20-letter alphabet; a
// letter outside it is replaced by a residue drawn from a generator seeded with `unknown_seed`,
// where the reference uses rand()%20, hclust2.cpp:54-56).  Writes the reference's clusters file
// ("#clusterid:<i>:size<m>" + member names, :137-150).  Returns 0 or an hs_status with *err set.
struct Kmer {
  std::string name, seq;
};
// hclust2's reader (:231-241): whitespace tokens, ">name" then the sequence token.
bool ReadKmerFasta(const std::string& path, std::vector<Kmer>* kmers);
int Clustering(const std::vector<Kmer>& kmers, const uint32_t& hash_K, const uint32_t& hash_L,
               const double& hash_W, const double& hash_R, const std::string& output_file,
               const Planes& planes, int device, uint32_t unknown_seed, std::string* err,
               uint64_t* n_clusters = nullptr);

// evaulate() (:100-165) with weight() (:67-87): weighted recall of a hits file against a ground
// truth file sorted by (motif, protein); also writes <output_file>.accuracy.txt.  Returns NaN
// where the reference would exit(0) on an inconsistent ground truth (:68-71).
double Evaluate(const std::string& ground_truth, const std::string& output_file, const double& hash_R);

// ---- brute force (row a11) ----------------------------------------------------------------------
// Search() of motif_both_points_noLSH.cpp:36-56: every (centre, k-mer) pair, centre-major, k-mer
// ascending; "<center> <kmer> <dist>" to output_file unless sqrt(d2) > R.  The reference also
// writes every excluded pair to <output_file>notlessthan.txt (:41-49, Q x N lines); here only when
// write_not_less_than is set.  Distances come from hs_bruteforce (exact fp64, reference order).
int SearchBruteForce(const std::vector<Point>& kmers, const std::vector<Point>& centers,
                     const std::vector<std::string>& kmer_names,
                     const std::vector<std::string>& center_names, const double& hash_R,
                     const std::string& output_file, int device, std::string* err,
                     bool write_not_less_than = false);

// ---- evaluation tooling ------------------------------------------------------
// evaluate2.cpp as it runs (:73-95): the hits file sorted by (motif, protein) and written
// tab-separated to <hits_file>sort.txt -- the ground-truth form evaulate()/Evaluate() expects.
bool SortHitsFile(const std::string& hits_file, uint64_t* n_records = nullptr);
// evaluate2.cpp's comparison (:98-153, unreachable in the reference behind the early return :95):
// weighted recall of hits_file against the ground truth with weight() of :62-71 (49.38 form).
// ground_truth is sorted here as main() does (:88); returns tp / (tp + fn).
double Evaluate2(const std::string& ground_truth, const std::string& hits_file, double* tp = nullptr,
                 double* fn = nullptr);

// ---- motif families -> centroid queries (centerDistanceSmapling.cpp) ---------------------------
// The motif-family file read by main() (:436-456): a line starting with '#' opens a family (the
// whole line is its name), every other non-empty line is one member k-mer; families with fewer
// than min_size members are dropped (MIN_SIZE_CLUSTER = 50, :12).
struct MotifFamily {
  std::string name;
  std::vector<std::string> seqs;
};
bool ReadMotifFamilies(const std::string& path, uint32_t min_size, std::vector<MotifFamily>* families);
// Center() (:67-78) of the members' embeddings (KmerToCoordinates :41-56): coordinate sums in
// member order, then one division by the member count.  A letter outside the 20-letter alphabet --
// the reference substitutes rand() % 20 (:47-49) -- or a member of another length is an error.
bool FamilyCenters(const std::vector<MotifFamily>& families, uint32_t kmer_length,
                   std::vector<Point>* centers, std::string* err);
// cluster2datapoint() (:110-136): the centres as a points file <output_file>hclust.format.txt
// (family name line, then the coordinates), i.e. the `-c` input of motif_both_points.
bool Cluster2DataPoint(const std::vector<MotifFamily>& families, const std::vector<Point>& centers,
                       const std::string& output_file);
// sequencedatabase2centers() (:138-190), the function main() runs: all centre-to-centre distances
// (i < j) to <dir>/<output_file>innercenter_protein_centers_0.txt, and the distance of each of
// the first min(100000, N) database points to each centre (centre-major) to
// <dir>/<output_file>ramdom_protein_centers_0.txt.  The reference hard-codes dir =
// "./pro2centerdis" (which must exist) and reads 100000 points whatever N is; here dir is created
// if missing.  The N x C distances are computed on `device` (hs_bruteforce).
int SequenceDatabase2Centers(const std::vector<Point>& kmers_proteins,
                             const std::vector<Point>& centers, const std::string& output_file,
                             const std::string& dir, int device, std::string* err);

}  // namespace hsearch

#endif
