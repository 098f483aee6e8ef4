// TEST INFRASTRUCTURE ONLY.  Shared by the ref_*_harness.cpp files: the standard headers the
// reference's programs use (included before `main` is renamed) and a cout mute.
#ifndef HS_REF_TOOLS_COMMON_H
#define HS_REF_TOOLS_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <dirent.h>
#include <errno.h>
#include <sys/stat.h>
#include <unistd.h>

#define HS_REF_API extern "C" __attribute__((visibility("default")))

struct HsRefCoutMute {
  std::streambuf* old;
  std::ostringstream sink;
  HsRefCoutMute() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~HsRefCoutMute() { std::cout.rdbuf(old); }
};
#endif
