// Fused 3-tap stencil + single-channel batch normalisation, forward and backward: the PINN ResidualBlock's
// `bn1(conv1(x.unsqueeze(1))).squeeze(1)` with Conv1d(1, 1, 3, padding=1) and BatchNorm1d(1)
// (/root/reference/OpenPyStruct_PINN_MultiCase.py:425-452) in ONE launch per direction.
//
// Why a kernel: through the framework the pair costs ~85 launches per training step on ROCm (MIOpen im2col + GEMM +
// col2im per sample chunk for a 3-tap filter, and a single-workgroup spatial batch norm that takes ~100 us for the
// one channel; profiles/r01_notes.md).  The tensor is small (B x F = 128 x 350 floats, 179 KB), so the work is spread
// over up to 64 workgroups that leave per-workgroup partial sums in a caller-provided workspace: statistics pass +
// apply pass forward, statistics + apply + a one-thread parameter-gradient pass backward; no atomics, nothing to zero.
// Partial sums: float per thread, double across threads and workgroups.
//
//   y[r][i] = w0 x[r][i-1] + w1 x[r][i] + w2 x[r][i+1] + b          (zero padding at the row ends)
//   z       = gamma (y - mean) invstd + beta,   mean / biased var over all B*F values of y
//   running_mean/var: momentum update with the UNBIASED variance, num_batches_tracked += 1   (training)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

constexpr int SB_THREADS = 256;     // 4 waves: lane -> column, wave -> row
constexpr int SB_MAXG = 64;         // workgroups per launch (partial sums per workgroup, no atomics, no zeroing)

__device__ __forceinline__ double wave_sum_d(double v) {
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
// sums NV values over the workgroup; every thread gets the totals
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* s_red /*[4][NV]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum_d(v[k]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) s_red[wave * NV + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t = 0.0;
    for (int w = 0; w < SB_THREADS / 64; ++w) t += s_red[w * NV + k];
    v[k] = t;
  }
}
// totals of the per-workgroup partial sums part[G][NV] (every thread reads them: G <= 64)
template <int NV>
__device__ __forceinline__ void sum_partials(const double* __restrict__ part, int G, double (&v)[NV]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = 0.0;
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] += part[g * NV + k];
}

// bfloat16 <-> float (round to nearest even), for the autocast dtype of z / grad_z
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float ld_grad(const void* g, long i, int bf16) {
  return bf16 ? __uint_as_float((uint32_t)((const uint16_t*)g)[i] << 16) : ((const float*)g)[i];
}

__device__ __forceinline__ float stencil_at(const float* __restrict__ row, int i, int F, float w0, float w1, float w2, float b) {
  const float xm = i > 0 ? row[i - 1] : 0.0f, xc = row[i], xp = i + 1 < F ? row[i + 1] : 0.0f;
  return __builtin_fmaf(w0, xm, __builtin_fmaf(w1, xc, __builtin_fmaf(w2, xp, b)));
}

// ---- forward, pass 1: per-workgroup (sum y, sum y^2) ----
__global__ __launch_bounds__(SB_THREADS) void stencil_bn_fwd_stats_kernel(int B, int F, const float* __restrict__ x,
                                                                           const float* __restrict__ cw, const float* __restrict__ cb,
                                                                           double* __restrict__ part) {
  __shared__ double s_red[4 * 2];
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2], b = cb[0];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float p0 = 0.0f, p1 = 0.0f;        // float per thread (a few dozen terms), double across threads
  for (int r = blockIdx.x * (SB_THREADS / 64) + ty; r < B; r += gridDim.x * (SB_THREADS / 64)) {
    const float* row = x + (long)r * F;
    for (int i = tx; i < F; i += 64) {
      const float y = stencil_at(row, i, F, w0, w1, w2, b);
      p0 += y;
      p1 = __builtin_fmaf(y, y, p1);
    }
  }
  double acc[2] = {(double)p0, (double)p1};
  block_sum<2>(acc, s_red);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = acc[0]; part[blockIdx.x * 2 + 1] = acc[1]; }
}

// ---- forward, pass 2: statistics from the partials (or the running ones), running-stat update, z ----
__global__ __launch_bounds__(SB_THREADS) void stencil_bn_fwd_apply_kernel(int B, int F, const float* __restrict__ x, const float* __restrict__ cw,
                                                                           const float* __restrict__ cb, const float* __restrict__ gamma,
                                                                           const float* __restrict__ beta, float eps, float momentum, int training,
                                                                           const double* __restrict__ part, int G, float* running_mean,
                                                                           float* running_var, long long* num_batches,
                                                                           void* __restrict__ z, int z_bf16, float* __restrict__ save) {
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2], b = cb[0];
  const long n = (long)B * F;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float mean, invstd;
  if (training) {
    double acc[2];
    sum_partials<2>(part, G, acc);
    const double m = acc[0] / n, var = fmax(acc[1] / n - m * m, 0.0);
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      running_mean[0] = (1.0f - momentum) * running_mean[0] + momentum * (float)m;
      running_var[0] = (1.0f - momentum) * running_var[0] + momentum * (float)(var * n / (n > 1 ? n - 1 : 1));
      if (num_batches) num_batches[0] += 1;
    }
  } else {
    mean = running_mean[0];
    invstd = 1.0f / sqrtf(running_var[0] + eps);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { save[0] = mean; save[1] = invstd; }
  const float scale = gamma[0] * invstd, shift = beta[0] - mean * scale;
  for (int r = blockIdx.x * (SB_THREADS / 64) + ty; r < B; r += gridDim.x * (SB_THREADS / 64)) {
    const float* row = x + (long)r * F;
    for (int i = tx; i < F; i += 64) {
      const float v = __builtin_fmaf(stencil_at(row, i, F, w0, w1, w2, b), scale, shift);
      if (z_bf16) ((uint16_t*)z)[(long)r * F + i] = f2bf(v);
      else ((float*)z)[(long)r * F + i] = v;
    }
  }
}

// ---- backward, pass 1: per-workgroup (sum g, sum g yhat) ----
__global__ __launch_bounds__(SB_THREADS) void stencil_bn_bwd_stats_kernel(int B, int F, const float* __restrict__ x, const void* __restrict__ g,
                                                                           int g_bf16, const float* __restrict__ cw, const float* __restrict__ cb,
                                                                           const float* __restrict__ save, double* __restrict__ part) {
  __shared__ double s_red[4 * 2];
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2], b = cb[0], mean = save[0], invstd = save[1];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float q0 = 0.0f, q1 = 0.0f;
  for (int r = blockIdx.x * (SB_THREADS / 64) + ty; r < B; r += gridDim.x * (SB_THREADS / 64)) {
    const float* row = x + (long)r * F;
    for (int i = tx; i < F; i += 64) {
      const float yh = (stencil_at(row, i, F, w0, w1, w2, b) - mean) * invstd, gi = ld_grad(g, (long)r * F + i, g_bf16);
      q0 += gi;
      q1 = __builtin_fmaf(gi, yh, q1);
    }
  }
  double acc[2] = {(double)q0, (double)q1};
  block_sum<2>(acc, s_red);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = acc[0]; part[blockIdx.x * 2 + 1] = acc[1]; }
}

// ---- backward, pass 2: dx, per-workgroup (dw0, dw1, dw2, db) ----
//   dy = gamma invstd (g - mean(g) - yhat mean(g yhat))   (training; gamma invstd g with frozen statistics)
//   dx[i] = w0 dy[i+1] + w1 dy[i] + w2 dy[i-1]
__global__ __launch_bounds__(SB_THREADS) void stencil_bn_bwd_apply_kernel(int B, int F, const float* __restrict__ x, const void* __restrict__ g,
                                                                           int g_bf16, const float* __restrict__ cw, const float* __restrict__ cb,
                                                                           const float* __restrict__ gamma, const float* __restrict__ save,
                                                                           int training, const double* __restrict__ part, int G,
                                                                           float* __restrict__ dx, double* __restrict__ part4) {
  __shared__ double s_red[4 * 4];
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2], b = cb[0], mean = save[0], invstd = save[1];
  const long n = (long)B * F;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  double acc2[2];
  sum_partials<2>(part, G, acc2);
  const float k = gamma[0] * invstd, mg = training ? (float)(acc2[0] / n) : 0.0f, mgy = training ? (float)(acc2[1] / n) : 0.0f;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  for (int r = blockIdx.x * (SB_THREADS / 64) + ty; r < B; r += gridDim.x * (SB_THREADS / 64)) {
    const float* row = x + (long)r * F;
    float* dxrow = dx + (long)r * F;
    for (int i = tx; i < F; i += 64) {
      float dy[3];                                    // at i-1, i, i+1
#pragma unroll
      for (int d = -1; d <= 1; ++d) {
        const int q = i + d;
        if (q >= 0 && q < F) {
          const float yh = (stencil_at(row, q, F, w0, w1, w2, b) - mean) * invstd;
          dy[d + 1] = k * (ld_grad(g, (long)r * F + q, g_bf16) - mg - yh * mgy);
        } else {
          dy[d + 1] = 0.0f;
        }
      }
      dxrow[i] = __builtin_fmaf(w0, dy[2], __builtin_fmaf(w1, dy[1], w2 * dy[0]));
      const float xm = i > 0 ? row[i - 1] : 0.0f, xp = i + 1 < F ? row[i + 1] : 0.0f;
      a0 = __builtin_fmaf(dy[1], xm, a0);
      a1 = __builtin_fmaf(dy[1], row[i], a1);
      a2 = __builtin_fmaf(dy[1], xp, a2);
      a3 += dy[1];
    }
  }
  double acc4[4] = {(double)a0, (double)a1, (double)a2, (double)a3};
  block_sum<4>(acc4, s_red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) part4[blockIdx.x * 4 + q] = acc4[q];
}

// ---- backward, pass 3: the six parameter gradients from the partials ----
__global__ void stencil_bn_bwd_params_kernel(const double* __restrict__ part, const double* __restrict__ part4, int G, float* __restrict__ dparams) {
  if (threadIdx.x == 0) {
    double a2[2], a4[4];
    sum_partials<2>(part, G, a2);
    sum_partials<4>(part4, G, a4);
    dparams[0] = (float)a4[0]; dparams[1] = (float)a4[1]; dparams[2] = (float)a4[2]; dparams[3] = (float)a4[3];
    dparams[4] = (float)a2[1]; dparams[5] = (float)a2[0];       // dgamma = sum g yhat, dbeta = sum g
  }
}

}  // namespace opsamd

using namespace opsamd;

static int sb_grid(int B) {
  const int g = (B + SB_THREADS / 64 - 1) / (SB_THREADS / 64);
  return g < 1 ? 1 : (g > SB_MAXG ? SB_MAXG : g);
}

extern "C" size_t ops_stencil3_bn1_workspace_bytes(void) { return (size_t)SB_MAXG * 6 * sizeof(double); }

extern "C" int ops_stencil3_bn1_fwd_f32(int B, int F, const float* x, const float* conv_w, const float* conv_b, const float* gamma,
                                        const float* beta, float eps, float momentum, int training, float* running_mean,
                                        float* running_var, long long* num_batches_tracked, void* z, int z_is_bf16, float* save,
                                        void* workspace, void* stream) {
  if (B < 1 || F < 1 || !x || !conv_w || !conv_b || !gamma || !beta || !running_mean || !running_var || !z || !save || !workspace)
    return OPS_AMD_ERR_INVALID_ARG;
  if ((long)B * F > (1L << 26)) return OPS_AMD_ERR_UNSUPPORTED;
  const int G = sb_grid(B);
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)workspace;
  if (training) hipLaunchKernelGGL(stencil_bn_fwd_stats_kernel, dim3(G), dim3(SB_THREADS), 0, s, B, F, x, conv_w, conv_b, part);
  hipLaunchKernelGGL(stencil_bn_fwd_apply_kernel, dim3(G), dim3(SB_THREADS), 0, s, B, F, x, conv_w, conv_b, gamma, beta, eps, momentum, training,
                     part, G, running_mean, running_var, num_batches_tracked, z, z_is_bf16, save);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}

extern "C" int ops_stencil3_bn1_bwd_f32(int B, int F, const float* x, const void* grad_z, int grad_is_bf16, const float* conv_w, const float* conv_b,
                                        const float* gamma, const float* save, int training, float* dx, float* dparams, void* workspace,
                                        void* stream) {
  if (B < 1 || F < 1 || !x || !grad_z || !conv_w || !conv_b || !gamma || !save || !dx || !dparams || !workspace) return OPS_AMD_ERR_INVALID_ARG;
  if ((long)B * F > (1L << 26)) return OPS_AMD_ERR_UNSUPPORTED;
  const int G = sb_grid(B);
  hipStream_t s = (hipStream_t)stream;
  double* part = (double*)workspace;
  double* part4 = part + SB_MAXG * 2;
  hipLaunchKernelGGL(stencil_bn_bwd_stats_kernel, dim3(G), dim3(SB_THREADS), 0, s, B, F, x, grad_z, grad_is_bf16, conv_w, conv_b, save, part);
  hipLaunchKernelGGL(stencil_bn_bwd_apply_kernel, dim3(G), dim3(SB_THREADS), 0, s, B, F, x, grad_z, grad_is_bf16, conv_w, conv_b, gamma, save, training, part, G,
                     dx, part4);
  hipLaunchKernelGGL(stencil_bn_bwd_params_kernel, dim3(1), dim3(64), 0, s, part, part4, G, dparams);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
