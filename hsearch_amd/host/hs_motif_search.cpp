// hs_motif_search.cpp -- the `motif_both_points` program of the reference on the GPU path.
//
// Keeps the reference's command line (hclust/src/hclust/motif_both_points.cpp:302-320):
//     -d <db.points> -c <centers.points> -l <k> -W <w> -T <R> -g <groundtruth> -o <out>
// each also as -db/-center/-len/-window/-threshold/-groundtruth/-output (one or two dashes), and its
// exit behaviour: a missing required option prints the help text and exits 0 (:332-335), a runtime
// error prints to stderr and exits 1 (:387-393).  Additions: -K/-L (the reference parses them in
// kmer_search.cpp:186-189 but hard-codes 4,4 here, :380-381 -- 4,4 stay the defaults), --seed
// (planes are drawn like the reference's LSH constructor but from an explicit seed; default: from
// std::random_device like the reference), --planes <file> (read them instead: the binary doubles
// --planes-out writes), --device, --gpus n (the centres are sharded over GPUs
// device .. device+n-1 of this node, one host thread and one replicated index per GPU, hits
// all-gathered over RCCL and written in the same order -- the output file is the same for every
// n; --transport loopback runs the same rank threads with all ranks on --device and the exchange
// through host memory), and -g becomes optional (without it the evaluation step is skipped).  -c may also name a
// k-mer FASTA file (">name" + k letters, the format hclust2.cpp:231-241 reads): the centres are
// then embedded exactly from the table.  -d may also name a protein FASTA file (the
// database kmer_search.cpp:180-181 takes): every length-k window of every sequence is then a DB
// k-mer, enumerated on the device; --ref-compat-eq-swap reproduces the reference's E <-> Q exchange
// on that path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "hs_host.hpp"

namespace {

struct Opt {
  const char* long_name;
  char short_name;
  const char* descr;
  bool required;
};

const Opt kOpts[] = {
    {"db", 'd', "protein database file (points)", true},
    {"center", 'c', "centers from Pfam database (points)", true},
    {"len", 'l', "kmer length", true},
    {"hash_K", 'K', "number of random lines [4]", false},
    {"hash_L", 'L', "number of hash tables [4]", false},
    {"window", 'W', "bucket width", true},
    {"threshold", 'T', "kmer threshold", true},
    {"groundtruth", 'g', "groundtruth (sorted brute-force hits); optional", false},
    {"output", 'o', "output file name", true},
    {"seed", 's', "seed of the LSH planes [random_device]", false},
    {"device", 'G', "GPU ordinal (the first one with --gpus) [0]", false},
    {"gpus", 'N', "shard the centres over this many GPUs, hits all-gathered over RCCL [off: one GPU, no communicator]", false},
    {"partition", 'Y', "with --gpus: queries (every GPU the whole index and a block of the centres, the default), tables (every GPU a subset of the L tables over all k-mers and all centres) or buckets (every GPU the whole index, all centres and its share of the buckets); same output", false},
    {"transport", 'X', "with --gpus: rccl (one rank per GPU, the default) or loopback (host memory between the rank threads, all ranks on --device: the rank protocol on a box with fewer GPUs)", false},
    {"centers-as-points", 'E', "k-mer centres over a FASTA database: send them embedded (8k doubles each) instead of as residue codes [0]", false},
    {"planes", 'p', "read the planes from this file (as written by --planes-out) instead of drawing them", false},
    {"planes-out", 'P', "write the planes (binary doubles a[L][K][d] then b[L][K])", false},
    {"ref-compat-eq-swap", 'Q', "FASTA database: exchange E and Q like the reference's ProteinDB [0]", false},
    {"best-per-position", 'B', "FASTA database: one line per matched window, its nearest centre (kmer_search) [0]", false},
};

// A points file has a line of numbers after its first name line; a FASTA file has residue letters.
bool LooksLikeFasta(const std::string& path) {
  std::ifstream fin(path.c_str());
  std::string l1, l2;
  if (!std::getline(fin, l1) || l1.empty() || l1[0] != '>') return false;
  while (std::getline(fin, l2))
    if (!l2.empty()) break;
  if (l2.empty()) return false;
  char* end = nullptr;
  (void)strtod(l2.c_str(), &end);
  return end == l2.c_str();  // no number at the start of the second line
}

void Help(const char* prog) {
  fprintf(stderr, "Usage: %s [OPTIONS]\n\nOptions:\n", prog);
  for (const Opt& o : kOpts)
    fprintf(stderr, "  -%c, -%-12s %s%s\n", o.short_name, o.long_name, o.descr,
            o.required ? " [REQUIRED]" : "");
  fprintf(stderr, "\nHelp options:\n  -?, -help   print this help message\n\ncluster kmers to motifs\n");
}

}  // namespace

int main(int argc, const char* argv[]) {
  bool help = false;
  std::map<std::string, std::string> val;
  for (int i = 1; i < argc; ++i) {
    std::string arg = argv[i];
    if (arg == "-help" || arg == "--help" || arg == "-?" || arg == "-about") {
      help = true;
      continue;
    }
    if (arg.size() < 2 || arg[0] != '-') continue;  // leftover argument, ignored like the reference
    std::string name = arg.substr(arg[1] == '-' ? 2 : 1);
    const Opt* hit = nullptr;
    for (const Opt& o : kOpts)
      if (name == o.long_name || (name.size() == 1 && name[0] == o.short_name)) hit = &o;
    if (!hit) {
      fprintf(stderr, "unknown option %s\n", arg.c_str());
      return EXIT_FAILURE;
    }
    if (i + 1 >= argc) {
      fprintf(stderr, "option %s needs a value\n", arg.c_str());
      return EXIT_FAILURE;
    }
    val[hit->long_name] = argv[++i];
  }
  if (argc > 1 && !help) {
    fprintf(stdout, "[WELCOME TO HSEARCH v1.0 -- MI355X]\n[%s", argv[0]);
    for (int i = 1; i < argc; ++i) fprintf(stdout, " %s", argv[i]);
    fprintf(stdout, "]\n");
  }
  if (argc == 1 || help) {
    Help(argv[0]);
    return EXIT_SUCCESS;
  }
  for (const Opt& o : kOpts)
    if (o.required && !val.count(o.long_name)) {
      fprintf(stderr, "missing required option -%c\n", o.short_name);
      Help(argv[0]);
      return EXIT_SUCCESS;  // as the reference: option_missing() -> message, EXIT_SUCCESS
    }
  const uint32_t kmer_length = (uint32_t)strtoul(val["len"].c_str(), nullptr, 10);
  const uint32_t hash_K = val.count("hash_K") ? (uint32_t)strtoul(val["hash_K"].c_str(), nullptr, 10) : 4;
  const uint32_t hash_L = val.count("hash_L") ? (uint32_t)strtoul(val["hash_L"].c_str(), nullptr, 10) : 4;
  const double hash_W = strtod(val["window"].c_str(), nullptr);
  const double hash_R = strtod(val["threshold"].c_str(), nullptr);
  const int device = val.count("device") ? atoi(val["device"].c_str()) : 0;
  const uint32_t dim = 8 * kmer_length;
  uint32_t seed;
  if (val.count("seed")) {
    seed = (uint32_t)strtoul(val["seed"].c_str(), nullptr, 10);
  } else {
    std::random_device rd;
    seed = rd();
  }
  try {
    std::vector<std::string> kmer_names, center_names;
    std::vector<hsearch::Point> kmers, centers;
    std::vector<uint8_t> center_codes;  // -c <k-mers.fa>: the centres as rows of the coordinate table
    const bool fasta_db = LooksLikeFasta(val["db"]);
    hsearch::ProteinDB prodb;
    if (fasta_db) {
      std::cout << "Read protein sequences from " << val["db"] << std::endl;
      const bool swap = val.count("ref-compat-eq-swap") && atoi(val["ref-compat-eq-swap"].c_str()) != 0;
      if (!hsearch::ReadProteinFasta(val["db"], swap, &prodb)) {
        fprintf(stderr, "cannot open %s\n", val["db"].c_str());
        return EXIT_FAILURE;
      }
      std::cout << "number of proteins " << prodb.start.size() - 1 << std::endl;
      std::cout << "total length " << prodb.residues.size() << std::endl;
    } else {
      std::cout << "Read Kmers..." << std::endl;
      if (!hsearch::ReadPointsFile(val["db"], dim, &kmer_names, &kmers)) {
        fprintf(stderr, "cannot open %s\n", val["db"].c_str());
        return EXIT_FAILURE;
      }
    }
    std::cout << "Read Centers..." << std::endl;
    if (LooksLikeFasta(val["center"])) {
      std::vector<hsearch::Kmer> ckmers;
      std::string cerr;
      if (!hsearch::ReadKmerFasta(val["center"], &ckmers)) {
        fprintf(stderr, "cannot open %s\n", val["center"].c_str());
        return EXIT_FAILURE;
      }
      if (!hsearch::CentersFromKmers(ckmers, kmer_length, &center_names, &centers, &cerr, &center_codes)) {
        fprintf(stderr, "ERROR: %s\n", cerr.c_str());
        return EXIT_FAILURE;
      }
    } else if (!hsearch::ReadPointsFile(val["center"], dim, &center_names, &centers)) {
      fprintf(stderr, "cannot open %s\n", val["center"].c_str());
      return EXIT_FAILURE;
    }
    if (!fasta_db) std::cout << "number of kmers " << kmers.size() << std::endl;
    std::cout << "number of centers " << centers.size() << std::endl;
    hsearch::Planes planes;
    if (val.count("planes")) {
      std::string perr;
      if (!hsearch::ReadPlanesFile(val["planes"], dim, hash_K, hash_L, hash_W, &planes, &perr)) {
        fprintf(stderr, "ERROR: %s\n", perr.c_str());
        return EXIT_FAILURE;
      }
    } else {
      planes = hsearch::DrawPlanes(dim, hash_K, hash_L, hash_W, seed);
    }
    const bool use_comm = val.count("gpus") != 0;
    const int n_gpus = use_comm ? atoi(val["gpus"].c_str()) : 1;
    if (n_gpus < 1 || n_gpus > 64) {
      fprintf(stderr, "ERROR: --gpus must be 1..64\n");
      return EXIT_FAILURE;
    }
    const bool by_tables = val.count("partition") && val["partition"] == "tables";
    const bool by_buckets = val.count("partition") && val["partition"] == "buckets";
    if (val.count("partition") && !by_tables && !by_buckets && val["partition"] != "queries") {
      fprintf(stderr, "ERROR: --partition must be queries, tables or buckets\n");
      return EXIT_FAILURE;
    }
    hsearch::SetShardPartition(by_tables ? hsearch::kPartitionTables
                                         : by_buckets ? hsearch::kPartitionBuckets : hsearch::kPartitionQueries);
    const bool loopback = val.count("transport") && val["transport"] == "loopback";
    if (val.count("transport") && !loopback && val["transport"] != "rccl") {
      fprintf(stderr, "ERROR: --transport must be rccl or loopback\n");
      return EXIT_FAILURE;
    }
    if (loopback) hsearch::SetShardTransport(hsearch::kTransportLoopback);
    std::vector<int> devices;
    for (int g = 0; g < n_gpus; ++g) devices.push_back(loopback ? device : device + g);
    if (val.count("planes-out")) {
      std::ofstream pf(val["planes-out"].c_str(), std::ios::binary);
      pf.write(reinterpret_cast<const char*>(planes.a.data()), planes.a.size() * sizeof(double));
      pf.write(reinterpret_cast<const char*>(planes.b.data()), planes.b.size() * sizeof(double));
    }
    if (val.count("planes"))
      printf("hash_K = %u hash_L = %u planes = %s\n", hash_K, hash_L, val["planes"].c_str());
    else
      printf("hash_K = %u hash_L = %u seed = %u\n", hash_K, hash_L, seed);
    if (use_comm && loopback)
      printf("gpus = %d (%d ranks on device %d, hits exchanged through host memory)\n", n_gpus, n_gpus, device);
    else if (use_comm)
      printf("gpus = %d (devices %d..%d, RCCL all-gather of hits)\n", n_gpus, device, device + n_gpus - 1);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    std::string err;
    std::vector<uint64_t> table_sizes;
    uint64_t n_windows = 0;
    const int st =
        fasta_db ? hsearch::SearchProteinsSharded(prodb, kmer_length, centers, center_names, hash_K, hash_L,
                                                  hash_W, hash_R, val["output"], planes, devices, use_comm,
                                                  &err, &table_sizes, &n_windows,
                                                  val.count("best-per-position") &&
                                                      atoi(val["best-per-position"].c_str()) != 0,
                                                  // k-mer centres over a FASTA database share its exact
                                                  // table: they go to the GPU as codes (hs_query_codes)
                                                  center_codes.empty() || val.count("centers-as-points")
                                                      ? nullptr : &center_codes)
                 : hsearch::SearchSharded(kmers, centers, kmer_names, center_names, hash_K, hash_L, hash_W,
                                          hash_R, val["output"], planes, devices, use_comm, &err,
                                          &table_sizes);
    if (fasta_db && st == 0) std::cout << "number of kmers " << n_windows << std::endl;
    if (st != 0) {
      fprintf(stderr, "ERROR: %s (status %d)\n", err.c_str(), st);
      return EXIT_FAILURE;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (uint64_t ts : table_sizes) std::cout << "table size " << ts << std::endl;
    const double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if (val.count("groundtruth")) {
      std::cout << "evaulate ..." << std::endl;
      printf("ACCURACY: %lf %lf\n", hsearch::Evaluate(val["groundtruth"], val["output"], hash_R), secs);
    } else {
      printf("SEARCH: %lf seconds\n", secs);
    }
  } catch (const std::bad_alloc&) {
    fprintf(stderr, "ERROR: could not allocate memory\n");
    return EXIT_FAILURE;
  } catch (const std::exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
