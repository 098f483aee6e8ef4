// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Compiles the REAL reference FASTA -> points-file sampler (acgtun/hsearch
// hclust/src/hclust/protein2datapoints.cpp, with its protein.hpp reader) from the sources under
// /root/reference and runs its own main().  One seam: the program seeds rand() from the clock
// (srand(time(NULL)), protein.hpp:40, protein2datapoints.cpp:37,80); the calls are re-pointed at a
// no-op so the caller's seed stays in force and the run is reproducible.  Contains no reference
// code, only the #include.  Built into oracle/_ref/libref_p2d.so by oracle/Makefile.
#include "ref_tools_common.h"

static void hs_ref_srand_gate(unsigned) {}
#define srand(x) hs_ref_srand_gate(x)
#define main hs_ref_p2d_main
#include "hclust/src/hclust/protein2datapoints.cpp"
#undef main
#undef srand

// `protein2datapoints -d <fasta> -l <k> -n <num proteins> -o <out>` with rand() seeded by `seed`.
HS_REF_API int refp_main(const char* fasta, uint32_t kmer_length, uint32_t num_out, const char* out_path,
                         uint32_t seed) {
  HsRefCoutMute mute;
  const std::string len = std::to_string(kmer_length), num = std::to_string(num_out);
  const char* argv[] = {"protein2datapoints", "-d", fasta, "-l", len.c_str(), "-n", num.c_str(),
                        "-o", out_path, nullptr};
  FILE* saved = stdout;
  stdout = fopen("/dev/null", "w");
  srand(seed);
  const int rc = hs_ref_p2d_main(9, argv);
  fclose(stdout);
  stdout = saved;
  return rc;
}
