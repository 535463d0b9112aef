// Surrogate training loss, value AND gradient w.r.t. the predictions in one pass:
//   TrainableL1L2Loss  (/root/reference/OpenPyStruct_PINN_MultiCase.py:549-601, TFD:581-633, FNN:382-430) on the first nI columns
//       alpha * mean|p - t| + (1 - alpha) * mean (p - t)^2 + w * sum(relu(min - p) + relu(p - max)),  alpha = clamp(alpha, 1e-6, 1)
//   CompositeLoss      (PINN:603-653) adds  penalty * (mean_d |p - t| / (|t| + eps) + mean_r |p - t| / (|t| + eps))
//       over the next nD (deflections) and the remaining nR (rotations) columns, eps = 1e-8
//   + (alpha0 - alpha)^2   (TFD:743; 0 as long as nobody trains alpha)
// Through the framework the forward and backward of this expression are ~80 kernel nodes of 2-6 us in the captured
// training step (profiles/r01_notes.md) -- almost half of the PINN step.  Here: one pass over the [B, C] predictions
// writes the gradient (already divided by the means' counts) and per-workgroup partial sums, a one-thread pass adds them.
// Predictions / gradient in float32 or bfloat16 (the autocast dtype of the output layer), targets float32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

constexpr int FL_THREADS = 256;
constexpr int FL_MAXG = 128;

__device__ __forceinline__ uint16_t fl_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(FL_THREADS) void fused_loss_kernel(int B, int C, int nI, int nD, const void* __restrict__ preds, int bf16,
                                                                 const float* __restrict__ targets, const float* __restrict__ alpha_p,
                                                                 const float* __restrict__ minc, const float* __restrict__ maxc, float w,
                                                                 float penalty, float eps, void* __restrict__ grad, double* __restrict__ part) {
  __shared__ double s_red[FL_THREADS / 64][5];
  const float alpha = fminf(fmaxf(alpha_p[0], 1e-6f), 1.0f);
  const bool has_min = minc != nullptr, has_max = maxc != nullptr;
  const float lo = has_min ? minc[0] : 0.0f, hi = has_max ? maxc[0] : 0.0f;
  const int nR = C - nI - nD;
  const float inv_nI = 1.0f / ((float)B * (float)nI), inv_nD = nD > 0 ? 1.0f / ((float)B * (float)nD) : 0.0f,
              inv_nR = nR > 0 ? 1.0f / ((float)B * (float)nR) : 0.0f;
  float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // sum |d|_I, sum d^2_I, sum box penalty, sum rel_d, sum rel_r
  const long n = (long)B * C;
  for (long e = (long)blockIdx.x * FL_THREADS + threadIdx.x; e < n; e += (long)gridDim.x * FL_THREADS) {
    const int col = (int)(e % C);
    const float p = bf16 ? __uint_as_float((uint32_t)((const uint16_t*)preds)[e] << 16) : ((const float*)preds)[e];
    const float t = targets[e], d = p - t, sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    float g;
    if (col < nI) {
      acc[0] += fabsf(d);
      acc[1] = __builtin_fmaf(d, d, acc[1]);
      g = (alpha * sg + (1.0f - alpha) * 2.0f * d) * inv_nI;
      if (has_min && p < lo) { acc[2] += lo - p; g -= w; }
      if (has_max && p > hi) { acc[2] += p - hi; g += w; }
    } else {
      const float den = fabsf(t) + eps, rel = fabsf(d) / den;
      if (col < nI + nD) { acc[3] += rel; g = penalty * sg / den * inv_nD; }
      else { acc[4] += rel; g = penalty * sg / den * inv_nR; }
    }
    if (bf16) ((uint16_t*)grad)[e] = fl_f2bf(g);
    else ((float*)grad)[e] = g;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double v = acc[k];
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    if (lane == 0) s_red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    double t = 0.0;
    for (int wv = 0; wv < FL_THREADS / 64; ++wv) t += s_red[wv][threadIdx.x];
    part[blockIdx.x * 5 + threadIdx.x] = t;
  }
}

__global__ void fused_loss_finish_kernel(int B, int C, int nI, int nD, const double* __restrict__ part, int G, const float* __restrict__ alpha_p,
                                         float alpha0, float w, float penalty, float* __restrict__ loss, float* __restrict__ loss_sum) {
  double t[5] = {0, 0, 0, 0, 0};      // one wave: lane g sums partials g, g + 64, ...; butterfly over the lanes
  for (int g = threadIdx.x; g < G; g += 64)
    for (int k = 0; k < 5; ++k) t[k] += part[g * 5 + k];
  for (int k = 0; k < 5; ++k)
    for (int s = 32; s >= 1; s >>= 1) t[k] += __shfl_xor(t[k], s, 64);
  if (threadIdx.x != 0) return;
  const double alpha = fmin(fmax((double)alpha_p[0], 1e-6), 1.0);
  const int nR = C - nI - nD;
  const double nIe = (double)B * nI;
  double v = alpha * t[0] / nIe + (1.0 - alpha) * t[1] / nIe + (double)w * t[2];
  if (nD > 0) v += (double)penalty * t[3] / ((double)B * nD);
  if (nR > 0) v += (double)penalty * t[4] / ((double)B * nR);
  const double da = alpha0 == alpha0 ? (double)alpha0 - (double)alpha_p[0] : 0.0;      // NaN alpha0: no such term
  loss[0] = (float)(v + da * da);
  if (loss_sum) loss_sum[0] += loss[0];      // the epoch's running sum (one thread, one launch at a time: deterministic)
}

}  // namespace opsamd

using namespace opsamd;

extern "C" size_t ops_surrogate_loss_workspace_bytes(void) { return (size_t)FL_MAXG * 5 * sizeof(double); }

extern "C" int ops_surrogate_loss_grad_sum_f32(int B, int C, int nI, int nD, const void* preds, int preds_is_bf16, const float* targets,
                                               const float* alpha, float alpha0, const float* min_constraint, const float* max_constraint,
                                               float box_weight, float rel_penalty, float* loss, float* loss_sum, void* grad, void* workspace,
                                               void* stream);
extern "C" int ops_surrogate_loss_grad_f32(int B, int C, int nI, int nD, const void* preds, int preds_is_bf16, const float* targets,
                                           const float* alpha, float alpha0, const float* min_constraint, const float* max_constraint,
                                           float box_weight, float rel_penalty, float* loss, void* grad, void* workspace, void* stream) {
  return ops_surrogate_loss_grad_sum_f32(B, C, nI, nD, preds, preds_is_bf16, targets, alpha, alpha0, min_constraint, max_constraint, box_weight,
                                         rel_penalty, loss, nullptr, grad, workspace, stream);
}
extern "C" int ops_surrogate_loss_grad_sum_f32(int B, int C, int nI, int nD, const void* preds, int preds_is_bf16, const float* targets,
                                               const float* alpha, float alpha0, const float* min_constraint, const float* max_constraint,
                                               float box_weight, float rel_penalty, float* loss, float* loss_sum, void* grad, void* workspace,
                                               void* stream) {
  if (B < 1 || C < 1 || nI < 1 || nD < 0 || nI + nD > C || !preds || !targets || !alpha || !loss || !grad || !workspace)
    return OPS_AMD_ERR_INVALID_ARG;
  const long n = (long)B * C;
  long nb = (n + FL_THREADS - 1) / FL_THREADS;
  const int G = (int)(nb < FL_MAXG ? nb : FL_MAXG);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fused_loss_kernel, dim3(G), dim3(FL_THREADS), 0, s, B, C, nI, nD, preds, preds_is_bf16, targets, alpha, min_constraint,
                     max_constraint, box_weight, rel_penalty, 1e-8f, grad, (double*)workspace);
  hipLaunchKernelGGL(fused_loss_finish_kernel, dim3(1), dim3(64), 0, s, B, C, nI, nD, (const double*)workspace, G, alpha, alpha0, box_weight,
                     rel_penalty, loss, loss_sum);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
