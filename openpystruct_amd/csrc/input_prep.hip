// Batch assembly for the surrogate training loops: out[b, :] = X[idx[b], :] + sigma * N(0, 1), written as float32 or
// bfloat16 -- the DataLoader gather (PINN_MultiCase.py:748), the decaying input noise (:743, :756) and the autocast cast of
// the first GEMM's operand in ONE launch instead of six (index_select, normal, mul, add, cast + fill).
// Noise: Box-Muller on a counter-based hash of (seed, call counter, element index); the counter lives in device memory and
// is advanced by the kernel.  The reference's noise comes from an unseeded framework generator: only N(0, sigma^2) is reproduced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"
#include "call_counter.hpp"
#include "input_noise.hpp"

namespace opsamd {

__device__ __forceinline__ uint16_t ip_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// VEC4: F % 4 == 0 and 16-byte aligned rows -- four consecutive features per thread and trip (one index load, one 16-byte row load,
// one 8- or 16-byte store); the grid is at most 128 workgroups (call_counter.hpp), so the trips must be wide
template <bool VEC4>
__global__ __launch_bounds__(256) void gather_noise_kernel(int B, long F, const float* __restrict__ X, const long long* __restrict__ idx,
                                                            const float* __restrict__ sigma, unsigned long long seed,
                                                            unsigned long long* __restrict__ counter, void* __restrict__ out, int out_bf16,
                                                            const float* __restrict__ Y, int C, float* __restrict__ Yout) {
  const long n = (long)B * F;
  const float sg = sigma ? *sigma : 0.0f;
  const unsigned long long call = counter ? *counter : 0ull;
  if (VEC4) {
    const long F4 = F >> 2, n4 = n >> 2;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (long)gridDim.x * 256) {
      const long b = q / F4, f = (q - b * F4) << 2, i = q << 2;
      const float4 x = *(const float4*)(X + idx[b] * F + f);
      const float v0 = ip_noisy(x.x, sg, seed, call, i), v1 = ip_noisy(x.y, sg, seed, call, i + 1);
      const float v2 = ip_noisy(x.z, sg, seed, call, i + 2), v3 = ip_noisy(x.w, sg, seed, call, i + 3);
      if (out_bf16) {
        uint2 o;
        o.x = (uint32_t)ip_f2bf(v0) | ((uint32_t)ip_f2bf(v1) << 16);
        o.y = (uint32_t)ip_f2bf(v2) | ((uint32_t)ip_f2bf(v3) << 16);
        *(uint2*)((uint16_t*)out + i) = o;
      } else {
        *(float4*)((float*)out + i) = make_float4(v0, v1, v2, v3);
      }
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
      const long b = i / F, f = i - b * F;
      const float v = ip_noisy(X[idx[b] * F + f], sg, seed, call, i);
      if (out_bf16) ((uint16_t*)out)[i] = ip_f2bf(v);
      else ((float*)out)[i] = v;
    }
  }
  // the batch's targets Y[idx] (float32, no noise): the framework's index_select was a node of its own per step
  if (Y)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)B * C; i += (long)gridDim.x * 256) {
      const long b = i / C;
      Yout[i] = Y[idx[b] * C + (i - b * C)];
    }
  // one increment per launch, by the last workgroup to finish (call_counter.hpp)
  if (counter) call_counter_done(counter, gridDim.x);   // every workgroup of this grid draws
}

}  // namespace opsamd

extern "C" int ops_gather_rows_noise_targets_f32(int B, long F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                                 unsigned long long* counter, void* out, int out_is_bf16, const float* Y, int C, float* Yout,
                                                 void* stream);
extern "C" int ops_gather_rows_noise_f32(int B, long F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                         unsigned long long* counter, void* out, int out_is_bf16, void* stream) {
  return ops_gather_rows_noise_targets_f32(B, F, X, idx, sigma, seed, counter, out, out_is_bf16, nullptr, 0, nullptr, stream);
}
extern "C" int ops_gather_rows_noise_targets_f32(int B, long F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                                 unsigned long long* counter, void* out, int out_is_bf16, const float* Y, int C, float* Yout,
                                                 void* stream) {
  if (B < 1 || F < 1 || !X || !idx || !out || (Y && (C < 1 || !Yout))) return OPS_AMD_ERR_INVALID_ARG;
  const long n = (long)B * F;
  const bool vec4 = (F % 4 == 0) && ((((uintptr_t)X | (uintptr_t)out) & 15) == 0);
  const long items = vec4 ? n / 4 : n;
  const unsigned grid = (unsigned)((items + 255) / 256 > 128 ? 128 : (items + 255) / 256);   // grid-stride; <= 128 reports to the call counter
  if (vec4)
    hipLaunchKernelGGL(opsamd::gather_noise_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, F, X, idx, sigma, seed, counter, out, out_is_bf16, Y, C, Yout);
  else
    hipLaunchKernelGGL(opsamd::gather_noise_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, F, X, idx, sigma, seed, counter, out, out_is_bf16, Y, C, Yout);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
