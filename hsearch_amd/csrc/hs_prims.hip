// hs_prims.hip -- device-wide sort / scan / run-length primitives used by the index build and by
// the final ordering of hits.  These are rocPRIM (ROCm's native primitive library, header-only,
// compiled here for gfx950); they are kept in their own translation unit because they compile
// slowly and are not the query hot loop.  The hot kernels are hand-written in hs_kernels.hip.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "hs_internal.h"

size_t hs_sort_pairs_u64_u32_temp(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                  (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 64, 0);
  return bytes;
}
// A bit range that starts above bit 0 and ends at the key's last bit is only safe on rocPRIM's
// onesweep path.  Up to radix_sort_config<>::merge_sort_limit items (1024 * 1024 here) the library
// block-sorts on [begin_bit, end_bit) and then MERGES with radix_merge_compare<.., true, T>(begin_bit,
// end_bit - begin_bit), whose mask is (T(1) << (radix_bits + start_bit)) - 1
// (rocprim/device/detail/device_radix_sort.hpp:685): with end_bit == 64 that is 1 << 64, undefined;
// the host evaluates it to a mask of the bits BELOW begin_bit, the merge then runs under an order its
// sorted blocks do not have, and its merge-path partitions leave their ranges -- the GPU memory fault
// of this library's first 48-bit sort (round 2).  The limit is taken from the library, not copied.
bool hs_sort_partial_bits_ok(size_t n) { return n > rocprim::radix_sort_config<>::merge_sort_limit; }
hipError_t hs_sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout,
                                 const uint32_t* vin, uint32_t* vout, size_t n, int begin_bit, int end_bit,
                                 hipStream_t s) {
  if (begin_bit > 0 && !hs_sort_partial_bits_ok(n)) return hipErrorInvalidValue;
  return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

size_t hs_sort_pairs_u32_u32_temp(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 32, 0);
  return bytes;
}
hipError_t hs_sort_pairs_u32_u32(void* temp, size_t temp_bytes, const uint32_t* kin, uint32_t* kout,
                                 const uint32_t* vin, uint32_t* vout, size_t n, int end_bit, hipStream_t s) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, 0, end_bit, s);
}

size_t hs_sort_pairs_u64_u64_temp(size_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                  (const uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64, 0);
  return bytes;
}
hipError_t hs_sort_pairs_u64_u64(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout,
                                 const uint64_t* vin, uint64_t* vout, size_t n, int end_bit,
                                 hipStream_t s) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, 0, end_bit, s);
}

size_t hs_scan_u32_temp(size_t n) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n,
                                rocprim::plus<uint32_t>(), 0);
  return bytes;
}
hipError_t hs_exclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out,
                                 size_t n, hipStream_t s) {
  return rocprim::exclusive_scan(temp, temp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
}

size_t hs_rle_u64_temp(size_t n) {
  size_t bytes = 0;
  (void)rocprim::run_length_encode(nullptr, bytes, (const uint64_t*)nullptr, n, (uint64_t*)nullptr,
                                   (uint32_t*)nullptr, (uint32_t*)nullptr, 0);
  return bytes;
}
hipError_t hs_rle_u64(void* temp, size_t temp_bytes, const uint64_t* in, uint64_t* unique_out,
                      uint32_t* counts_out, uint32_t* runs_out, size_t n, hipStream_t s) {
  return rocprim::run_length_encode(temp, temp_bytes, in, n, unique_out, counts_out, runs_out, s);
}
