// Fused [sum of up to three addends] -> BatchNorm1d (batch statistics over the rows) -> optional LeakyReLU -> optional
// dropout, forward and backward, ONE launch per direction -- the elementwise tail around the GEMMs of the PINN's
// residual MLP (/root/reference/OpenPyStruct_PINN_MultiCase.py:425-452, :519-541):
//
//   input layer   :  dropout(LeakyReLU(input_norm(input_fc(x))))                    one addend, activation + dropout
//   residual block:  norm(fc2(...) + bn1(conv1(x)) + x)                             three addends, no activation
//   (and, with the normalisation switched off, the block's inner  dropout(LeakyReLU(fc1(x))) )
//
// Why: a captured training step of that model was 107 kernel nodes of ~5 us (profiles/r02_train_pinn_trace.txt); through
// the framework each of these tails is 6-8 nodes forward (two adds, num_batches_tracked += 1, collect statistics,
// update + invert, transform, LeakyReLU, dropout) and 5-6 backward (masked scale, LeakyReLU backward, reduce, elementwise,
// two gradient accumulations).  Here: one node each way; the parameter gradients go straight into the caller's flat
// gradient buffer (no accumulate kernels), the three addends share one gradient tensor.
//
// The tensors are tiny (128 x 350): one workgroup owns 16 columns and all rows -- column statistics never leave the
// workgroup (no atomics, no workspace).  Up to 128 rows every thread keeps its 8 rows in registers (all loads issued
// before the first use: one pass over memory, one exposed latency); larger batches re-read the data from L2 per pass.
// Activations are float32 or bfloat16 (the autocast dtype), statistics and parameters float32.
//
// Dropout: keep-mask from a counter-based hash of (seed, call counter, element index); the call counter lives in device
// memory and is advanced by the kernel itself, so a replayed HIP graph draws fresh masks.  The mask (1 byte per element)
// is stored for the backward pass.  The reference's masks come from an unseeded framework generator: only the
// distribution (Bernoulli(1 - p), scaling 1 / (1 - p)) is reproduced.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"
#include "dropout_stream.hpp"
#include "call_counter.hpp"

namespace opsamd {

constexpr int FB_CW = 16;        // columns per workgroup
constexpr int FB_RG = 16;        // row groups per workgroup (256 threads)
constexpr int FB_RPT = 8;        // rows a thread keeps in registers: batches up to FB_RG * FB_RPT = 128 rows take ONE pass over memory

__device__ __forceinline__ float fb_ld(const void* p, long i, int bf16) {
  return bf16 ? __uint_as_float((uint32_t)((const uint16_t*)p)[i] << 16) : ((const float*)p)[i];
}
__device__ __forceinline__ uint16_t fb_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void fb_st(void* p, long i, int bf16, float v) {
  if (bf16) ((uint16_t*)p)[i] = fb_f2bf(v);
  else ((float*)p)[i] = v;
}
// splitmix64 finaliser: a counter-based uniform in [0, 1)
__device__ __forceinline__ float fb_uniform(uint64_t seed, uint64_t call, uint64_t idx) { return drop_uniform(seed, call, idx); }   // csrc/dropout_stream.hpp
// sum over the FB_RG row groups of a column (threads t, t + 32, ...): every thread of the column gets the total
__device__ __forceinline__ float fb_colsum(float v, float* s_red, int col, int rg) {
  __syncthreads();
  s_red[rg * FB_CW + col] = v;
  __syncthreads();
  float t = 0.0f;
#pragma unroll
  for (int g = 0; g < FB_RG; ++g) t += s_red[g * FB_CW + col];
  return t;
}

struct FbArgs {
  int B, F;
  const void* x1; const void* x2; const void* x3;   // addends (x2, x3 may be NULL), activation dtype
  int act_bf16;
  const float* gamma; const float* beta;            // NULL gamma: no normalisation (activation + dropout only)
  float eps, momentum;
  int training;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float slope; int use_act;                         // LeakyReLU(slope) when use_act
  float p_drop;                                     // 0: no dropout
  unsigned long long seed; unsigned long long* call_counter;
  void* y;                                          // [B,F] activation dtype
  void* z_save;                                     // [B,F] activation dtype: the summed input (NULL when there is one addend: x1 is it)
  float* mean_save; float* rstd_save;               // [F]
  uint8_t* mask;                                    // [B,F] (p_drop > 0)
};

__global__ __launch_bounds__(256) void fused_bn_fwd_kernel(const FbArgs a) {
  __shared__ float s_red[FB_RG * FB_CW], s_red2[FB_RG * FB_CW], s_red3[FB_RG * FB_CW];
  const int col = threadIdx.x & (FB_CW - 1), rg = threadIdx.x / FB_CW;
  const int c = blockIdx.x * FB_CW + col;
  const bool live = c < a.F;
  const int B = a.B, F = a.F, bf = a.act_bf16;
  auto zload = [&](int r) -> float {
    const long i = (long)r * F + c;
    float z = fb_ld(a.x1, i, bf);
    if (a.x2) z += fb_ld(a.x2, i, bf);
    if (a.x3) z += fb_ld(a.x3, i, bf);
    // bfloat16 activations: normalise the value that is SAVED for the backward pass (the sum rounded to bfloat16), so that the
    // LeakyReLU sees the same pre-activation sign in both directions (and the framework's bf16 sum is what it would see too)
    if (bf && (a.x2 || a.x3)) z = __uint_as_float((uint32_t)fb_f2bf(z) << 16);
    return z;
  };
  // rows of this thread: rg, rg + FB_RG, ...; the first FB_RPT of them live in registers
  const bool cached = B <= FB_RG * FB_RPT;             // workgroup-uniform
  float zr[FB_RPT];
#pragma unroll
  for (int k = 0; k < FB_RPT; ++k) {
    const int r = rg + k * FB_RG;
    zr[k] = (live && cached && r < B) ? zload(r) : 0.0f;
  }
  auto zin = [&](int k, int r) -> float { return cached ? zr[k] : zload(r); };
  float mean = 0.0f, rstd = 1.0f, g = 1.0f, be = 0.0f;
  if (a.gamma) {
    if (a.training) {
      // per-thread (count, mean, M2) of its rows, then ONE workgroup exchange and Chan's pairwise combination: a single
      // barrier pair instead of separate mean and centred-variance reductions, and no E[z^2] - mean^2 cancellation
      float cnt = 0.0f, mu = 0.0f, m2 = 0.0f;
      auto push = [&](float z) { cnt += 1.0f; const float d = z - mu; mu += d / cnt; m2 += d * (z - mu); };
      if (live) {
        if (cached) {
#pragma unroll
          for (int k = 0; k < FB_RPT; ++k) if (rg + k * FB_RG < B) push(zr[k]);
        } else {
          for (int r = rg; r < B; r += FB_RG) push(zload(r));
        }
      }
      __syncthreads();
      s_red[rg * FB_CW + col] = cnt; s_red2[rg * FB_CW + col] = mu; s_red3[rg * FB_CW + col] = m2;
      __syncthreads();
      float N = 0.0f, M = 0.0f, Q = 0.0f;
#pragma unroll
      for (int gq = 0; gq < FB_RG; ++gq) {
        const float nb = s_red[gq * FB_CW + col], mb = s_red2[gq * FB_CW + col], qb = s_red3[gq * FB_CW + col];
        if (nb > 0.0f) {
          const float nt = N + nb, d = mb - M;
          Q += qb + d * d * (N * nb / nt);
          M += d * (nb / nt);
          N = nt;
        }
      }
      mean = M;
      const float var = Q / (float)B;                                     // biased: what normalises
      rstd = rsqrtf(var + a.eps);
      if (live && rg == 0) {
        a.mean_save[c] = mean; a.rstd_save[c] = rstd;
        if (a.running_mean) {                                             // momentum update with the UNBIASED variance
          const float unb = var * ((float)B / (float)(B > 1 ? B - 1 : 1));
          a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mean;
          a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * unb;
        }
      }
      if (blockIdx.x == 0 && threadIdx.x == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
    } else if (live) {
      mean = a.running_mean[c];
      rstd = rsqrtf(a.running_var[c] + a.eps);
    }
    if (live) { g = a.gamma[c]; be = a.beta[c]; }
  }
  const bool drop = a.training && a.p_drop > 0.0f;
  const unsigned long long call = drop ? *a.call_counter : 0ull;
  const float keep_scale = drop ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  if (live) {
    auto emit = [&](int k, int r) {
      const long i = (long)r * F + c;
      const float z = zin(k, r);
      if (a.z_save) fb_st(a.z_save, i, bf, z);
      float y = a.gamma ? __builtin_fmaf((z - mean) * rstd, g, be) : z;
      if (a.use_act) y = y > 0.0f ? y : y * a.slope;
      if (drop) {
        const bool keep = fb_uniform(a.seed, call, (unsigned long long)i) >= a.p_drop;
        a.mask[i] = keep ? 1 : 0;
        y = keep ? y * keep_scale : 0.0f;
      }
      fb_st(a.y, i, bf, y);
    };
    if (cached) {
#pragma unroll
      for (int k = 0; k < FB_RPT; ++k) { const int r = rg + k * FB_RG; if (r < B) emit(k, r); }
    } else {
      for (int r = rg; r < B; r += FB_RG) emit(0, r);
    }
  }
  // one increment per launch, by the last workgroup to finish (call_counter.hpp); streams of different call sites differ by their seeds
  if (drop) call_counter_done(a.call_counter, gridDim.x);   // every workgroup of this grid draws
}

struct FbBwdArgs {
  int B, F;
  const void* dy; int act_bf16;                     // [B,F]
  const void* z;                                    // [B,F] the summed input saved by the forward pass
  const float* mean; const float* rstd;             // [F]
  const float* gamma; const float* beta;            // NULL gamma: no normalisation
  float slope; int use_act;
  float p_drop; const uint8_t* mask;
  void* dz;                                         // [B,F] gradient w.r.t. every addend
  float* dgamma; float* dbeta;                      // [F], ASSIGNED (the caller's flat gradient slices)
};

__global__ __launch_bounds__(256) void fused_bn_bwd_kernel(const FbBwdArgs a) {
  __shared__ float s_red[FB_RG * FB_CW], s_red2[FB_RG * FB_CW];
  const int col = threadIdx.x & (FB_CW - 1), rg = threadIdx.x / FB_CW;
  const int c = blockIdx.x * FB_CW + col;
  const bool live = c < a.F;
  const int B = a.B, F = a.F, bf = a.act_bf16;
  float mean = 0.0f, rstd = 1.0f, g = 1.0f, be = 0.0f;
  if (a.gamma && live) { mean = a.mean[c]; rstd = a.rstd[c]; g = a.gamma[c]; be = a.beta[c]; }
  const float keep_scale = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  // gradient at the normalisation's output
  auto gout = [&](int r, float& xhat) -> float {
    const long i = (long)r * F + c;
    float gy = fb_ld(a.dy, i, bf);
    if (a.p_drop > 0.0f) gy = a.mask[i] ? gy * keep_scale : 0.0f;
    const float z = fb_ld(a.z, i, bf);
    xhat = (z - mean) * rstd;
    if (a.use_act) {
      const float pre = a.gamma ? __builtin_fmaf(xhat, g, be) : z;
      gy = pre > 0.0f ? gy : gy * a.slope;
    }
    return gy;
  };
  const bool cached = B <= FB_RG * FB_RPT;             // workgroup-uniform: one pass over memory
  float gr[FB_RPT], xr[FB_RPT];
#pragma unroll
  for (int k = 0; k < FB_RPT; ++k) {
    const int r = rg + k * FB_RG;
    gr[k] = 0.0f; xr[k] = 0.0f;
    if (live && cached && r < B) gr[k] = gout(r, xr[k]);
  }
  if (!a.gamma) {                                   // activation + dropout only
    if (live) {
      if (cached) {
#pragma unroll
        for (int k = 0; k < FB_RPT; ++k) { const int r = rg + k * FB_RG; if (r < B) fb_st(a.dz, (long)r * F + c, bf, gr[k]); }
      } else {
        for (int r = rg; r < B; r += FB_RG) { float xh; fb_st(a.dz, (long)r * F + c, bf, gout(r, xh)); }
      }
    }
    return;
  }
  float sg = 0.0f, sgx = 0.0f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int k = 0; k < FB_RPT; ++k) { sg += gr[k]; sgx += gr[k] * xr[k]; }
    } else {
      for (int r = rg; r < B; r += FB_RG) { float xh; const float gy = gout(r, xh); sg += gy; sgx += gy * xh; }
    }
  }
  __syncthreads();
  s_red[rg * FB_CW + col] = sg; s_red2[rg * FB_CW + col] = sgx;
  __syncthreads();
  sg = 0.0f; sgx = 0.0f;
#pragma unroll
  for (int gq = 0; gq < FB_RG; ++gq) { sg += s_red[gq * FB_CW + col]; sgx += s_red2[gq * FB_CW + col]; }
  if (live) {
    if (rg == 0) { a.dgamma[c] = sgx; a.dbeta[c] = sg; }
    const float inv = 1.0f / (float)B, k2 = g * rstd;
    if (cached) {
#pragma unroll
      for (int k = 0; k < FB_RPT; ++k) {
        const int r = rg + k * FB_RG;
        if (r < B) fb_st(a.dz, (long)r * F + c, bf, k2 * (gr[k] - sg * inv - xr[k] * sgx * inv));
      }
    } else {
      for (int r = rg; r < B; r += FB_RG) {
        float xh;
        const float gy = gout(r, xh);
        fb_st(a.dz, (long)r * F + c, bf, k2 * (gy - sg * inv - xh * sgx * inv));
      }
    }
  }
}

}  // namespace opsamd

using namespace opsamd;

extern "C" int ops_fused_bn_act_fwd(int B, int F, const void* x1, const void* x2, const void* x3, int act_is_bf16,
                                    const float* gamma, const float* beta, float eps, float momentum, int training,
                                    float* running_mean, float* running_var, long long* num_batches_tracked,
                                    float slope, int use_act, float p_drop, unsigned long long seed,
                                    unsigned long long* call_counter, void* y, void* z_save, float* mean_save,
                                    float* rstd_save, uint8_t* mask, void* stream) {
  if (B < 1 || F < 1 || !x1 || !y) return OPS_AMD_ERR_INVALID_ARG;
  if (gamma && (!beta || (training && (!mean_save || !rstd_save)) || (!training && (!running_mean || !running_var)))) return OPS_AMD_ERR_INVALID_ARG;
  if (training && p_drop > 0.0f && (!mask || !call_counter || p_drop >= 1.0f)) return OPS_AMD_ERR_INVALID_ARG;
  if ((x2 || x3) && training && !z_save) return OPS_AMD_ERR_INVALID_ARG;
  const FbArgs a{B, F, x1, x2, x3, act_is_bf16, gamma, beta, eps, momentum, training, running_mean, running_var, num_batches_tracked,
                 slope, use_act, p_drop, seed, call_counter, y, z_save, mean_save, rstd_save, mask};
  hipLaunchKernelGGL(fused_bn_fwd_kernel, dim3((unsigned)((F + FB_CW - 1) / FB_CW)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}

extern "C" int ops_fused_bn_act_bwd(int B, int F, const void* dy, int act_is_bf16, const void* z, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, float slope, int use_act, float p_drop, const uint8_t* mask,
                                    void* dz, float* dgamma, float* dbeta, void* stream) {
  if (B < 1 || F < 1 || !dy || !z || !dz) return OPS_AMD_ERR_INVALID_ARG;
  if (gamma && (!beta || !mean || !rstd || !dgamma || !dbeta)) return OPS_AMD_ERR_INVALID_ARG;
  if (p_drop > 0.0f && !mask) return OPS_AMD_ERR_INVALID_ARG;
  const FbBwdArgs a{B, F, dy, act_is_bf16, z, mean, rstd, gamma, beta, slope, use_act, p_drop, mask, dz, dgamma, dbeta};
  hipLaunchKernelGGL(fused_bn_bwd_kernel, dim3((unsigned)((F + FB_CW - 1) / FB_CW)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
