// Achievable-bandwidth reference for the roofline records of bench.py: a device-to-device copy with one 16-byte access per
// lane and instruction, the access shape the beam kernels' row moves use (MI355X_MICROARCH.md: 6.29 TB/s with a float4 copy,
// where a framework `Tensor.copy_` measures 4.8-5.2).  Not on the product path; nothing else calls it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void copy16_kernel(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    // four independent 16-byte loads in flight per lane before the first store
    v4u a = src[i], b, c, d;
    const bool hb = i + 256 < n16, hc = i + 512 < n16, hd = i + 768 < n16;
    if (hb) b = src[i + 256];
    if (hc) c = src[i + 512];
    if (hd) d = src[i + 768];
    if (AUX == 2) {
      __builtin_nontemporal_store(a, &dst[i]);
      if (hb) __builtin_nontemporal_store(b, &dst[i + 256]);
      if (hc) __builtin_nontemporal_store(c, &dst[i + 512]);
      if (hd) __builtin_nontemporal_store(d, &dst[i + 768]);
    } else {
      dst[i] = a;
      if (hb) dst[i + 256] = b;
      if (hc) dst[i + 512] = c;
      if (hd) dst[i + 768] = d;
    }
  }
}

}  // namespace opsamd

extern "C" int ops_hbm_copy16(const void* src, void* dst, size_t bytes, int non_temporal, void* stream) {
  if (!src || !dst || bytes % 16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return OPS_AMD_ERR_INVALID_ARG;
  if (bytes == 0) return OPS_AMD_OK;
  const size_t n16 = bytes / 16;
  const size_t want = (n16 + 1023) / 1024;
  const unsigned grid = (unsigned)(want < 256 * 16 ? (want ? want : 1) : 256 * 16);     // 16 workgroups per CU, grid-stride
  if (non_temporal)
    hipLaunchKernelGGL(opsamd::copy16_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const opsamd::v4u*)src, (opsamd::v4u*)dst, n16);
  else
    hipLaunchKernelGGL(opsamd::copy16_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const opsamd::v4u*)src, (opsamd::v4u*)dst, n16);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
