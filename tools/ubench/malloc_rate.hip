// How long does hipMalloc take on this box, by size and by number of pieces?  (The first index build of a
// process at the configs[2] shape allocates 157 GB + 22 GB of scratch: is that time per byte or per call?)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(nullptr);
  const size_t GB = (size_t)1 << 30;
  for (int round = 0; round < 2; ++round) {
    for (size_t pieces : {1, 8, 64}) {
      const size_t total = 160 * GB, each = total / pieces;
      std::vector<void*> p(pieces, nullptr);
      double t0 = now();
      for (size_t i = 0; i < pieces; ++i)
        if (hipMalloc(&p[i], each) != hipSuccess) { printf("hipMalloc failed at piece %zu\n", i); return 1; }
      double t1 = now();
      hipMemset(p[0], 0, 1 << 20);
      hipDeviceSynchronize();
      double t2 = now();
      for (void* q : p) hipFree(q);
      double t3 = now();
      printf("round %d: %3zu pieces of %6.2f GB: malloc %.3f s, first touch %.3f s, free %.3f s\n", round, pieces,
             (double)each / GB, t1 - t0, t2 - t1, t3 - t2);
    }
  }
  return 0;
}
