// sort_kernels.hip -- reports into result order on the device when there are too many for the counting
// ranker (aux_kernels.hip: more than kRankLimit) -- dense results: plants every few KB, microsatellite
// patterns on repeat-rich text, millions of matches.  The host's std::sort of 7.4e5 (index, position) pairs
// took 63 ms of a 115 ms search; rocPRIM's radix sort (the library GEMM-of-sorting: plain use, as the task
// allows for library primitives) does the same in a fraction of a millisecond.
#include <hip/hip_runtime.h>

#include <cstring>  // (rocprim's texture iterator uses memset without including it)

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace sassy_hip {

namespace {
__global__ __launch_bounds__(256) void sort_keys_kernel(const Candidate* __restrict__ cand, uint32_t count,
                                                        unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  // multi-text buffers never come here; end positions are unique within one search of one strand
  if (i < count) keys[i] = cand[i].pos;
}
}  // namespace

// Bytes of scratch launch_sort_candidates needs for `count` reports (keys in, keys out, rocPRIM's own).
size_t sort_scratch_bytes(uint32_t count) {
  size_t temp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, temp, static_cast<unsigned long long*>(nullptr),
                                  static_cast<unsigned long long*>(nullptr), static_cast<Candidate*>(nullptr),
                                  static_cast<Candidate*>(nullptr), (size_t)count, 0, 64, hipStream_t(nullptr));
  return 2 * ((size_t)count * 8 + 256) + temp + 256;
}

// sorted[0 .. count) = cand[0 .. count) by ascending end position.
hipError_t launch_sort_candidates(const Candidate* d_cand, Candidate* d_sorted, uint32_t count, void* d_scratch,
                                  size_t scratch_bytes, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const size_t key_bytes = ((size_t)count * 8 + 255) / 256 * 256;
  if (scratch_bytes < 2 * key_bytes) return hipErrorInvalidValue;
  unsigned long long* keys_in = static_cast<unsigned long long*>(d_scratch);
  unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(d_scratch) + key_bytes);
  void* temp = static_cast<unsigned char*>(d_scratch) + 2 * key_bytes;
  size_t temp_bytes = scratch_bytes - 2 * key_bytes;
  hipLaunchKernelGGL(sort_keys_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_cand, count, keys_in);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, d_cand, d_sorted, (size_t)count, 0, 64, stream);
}

}  // namespace sassy_hip
