// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Compiles the REAL reference evaluate2 program (acgtun/hsearch hclust/src/hclust/evaluate2.cpp)
// from the sources under /root/reference; its main() is renamed and run as is.  Contains no
// reference code, only the #include.  Built into oracle/_ref/libref_evaluate2.so by oracle/Makefile.
#include "ref_tools_common.h"

#define main hs_ref_evaluate2_main
#include "hclust/src/hclust/evaluate2.cpp"
#undef main

// `evaluate2 <hits>`: writes <hits>sort.txt (evaluate2.cpp:73-95).
HS_REF_API int refe_sort(const char* hits_path) {
  HsRefCoutMute mute;
  const char* argv[] = {"evaluate2", hits_path, nullptr};
  return hs_ref_evaluate2_main(2, argv);
}

// weight() of evaluate2.cpp:62-71 (the comparison that uses it is unreachable in main()).
HS_REF_API double refe_weight(double dis) { return weight(dis); }
