// hs_evaluate2.cpp -- the `evaluate2` program of the reference (hclust/src/hclust/evaluate2.cpp).
//
//     hs_evaluate2 <bruteforce hits>                 as the reference runs: prints the file name and
//                                                    writes <file>sort.txt, sorted by (motif,
//                                                    protein), tab-separated (:73-95)
//     hs_evaluate2 <bruteforce hits> <hits|dir>      additionally the comparison the reference
//                                                    keeps behind its early return (:98-153): one
//                                                    "ACCURACY: tp fn tp/(tp+fn)\t<file>" line per
//                                                    hits file (every regular file of a directory)
// Host only: no GPU work.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <iostream>
#include <string>
#include <vector>

#include "hs_host.hpp"

int main(int argc, const char* argv[]) {
  if (argc < 2) {
    fprintf(stderr, "Usage: %s <bruteforce hits file> [<hits file or directory>]\n", argv[0]);
    return EXIT_SUCCESS;
  }
  std::cout << argv[1] << std::endl;  // :79
  if (!hsearch::SortHitsFile(argv[1])) {
    fprintf(stderr, "cannot open %s\n", argv[1]);
    return EXIT_FAILURE;
  }
  if (argc < 3) return EXIT_SUCCESS;  // :95
  std::vector<std::string> files;
  struct stat sb;
  if (stat(argv[2], &sb) == 0 && S_ISDIR(sb.st_mode)) {  // isdir / read_dir (:100-104)
    DIR* d = opendir(argv[2]);
    if (!d) {
      fprintf(stderr, "cannot read %s\n", argv[2]);
      return EXIT_FAILURE;
    }
    while (struct dirent* e = readdir(d)) {
      const std::string path = std::string(argv[2]) + "/" + e->d_name;
      if (stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) files.push_back(path);
    }
    closedir(d);
    std::sort(files.begin(), files.end());
  } else {
    files.push_back(argv[2]);
  }
  for (size_t k = 0; k < files.size(); ++k) {
    double tp = 0, fn = 0;
    const double acc = hsearch::Evaluate2(argv[1], files[k], &tp, &fn);
    std::cout << "ACCURACY: " << tp << " " << fn << " " << acc << "\t" << files[k] << std::endl;  // :149
  }
  return EXIT_SUCCESS;
}
