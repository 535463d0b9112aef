// Batch assembly for the surrogate training loops: out[b, :] = X[idx[b], :] + sigma * N(0, 1), written as float32 or
// bfloat16 -- the DataLoader gather (PINN_MultiCase.py:748), the decaying input noise (:743, :756) and the autocast cast of
// the first GEMM's operand in ONE launch instead of six (index_select, normal, mul, add, cast + fill).
// Noise: Box-Muller on a counter-based hash of (seed, call counter, element index); the counter lives in device memory and
// is advanced by the kernel.  The reference's noise comes from an unseeded framework generator: only N(0, sigma^2) is reproduced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"
#include "call_counter.hpp"

namespace opsamd {

__device__ __forceinline__ uint64_t ip_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint16_t ip_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(256) void gather_noise_kernel(int B, long F, const float* __restrict__ X, const long long* __restrict__ idx,
                                                            const float* __restrict__ sigma, unsigned long long seed,
                                                            unsigned long long* __restrict__ counter, void* __restrict__ out, int out_bf16) {
  const long n = (long)B * F;
  const float sg = sigma ? *sigma : 0.0f;
  const unsigned long long call = counter ? *counter : 0ull;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long b = i / F, f = i - b * F;
    float v = X[idx[b] * F + f];
    if (sg != 0.0f) {
      const uint64_t h = ip_mix(seed + 0x9E3779B97F4A7C15ull * (call + 1) + (uint64_t)i * 0xD1B54A32D192ED03ull);
      const float u1 = ((float)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
      const float u2 = (float)((h >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);   // [0, 1)
      v += sg * sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
    }
    if (out_bf16) ((uint16_t*)out)[i] = ip_f2bf(v);
    else ((float*)out)[i] = v;
  }
  // one increment per launch, by the last workgroup to finish (call_counter.hpp)
  if (counter && threadIdx.x == 0) call_counter_done(counter, gridDim.x);
}

}  // namespace opsamd

extern "C" int ops_gather_rows_noise_f32(int B, long F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                         unsigned long long* counter, void* out, int out_is_bf16, void* stream) {
  if (B < 1 || F < 1 || !X || !idx || !out) return OPS_AMD_ERR_INVALID_ARG;
  const long n = (long)B * F;
  const unsigned grid = (unsigned)((n + 255) / 256 > 128 ? 128 : (n + 255) / 256);   // grid-stride; <= 128 reports to the call counter
  hipLaunchKernelGGL(opsamd::gather_noise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, F, X, idx, sigma, seed, counter, out, out_is_bf16);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
