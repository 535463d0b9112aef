// Gradient clipping + Adam on ONE flat parameter buffer: the optimiser step of the surrogate training loops
// (`clip_grad_norm_(model.parameters(), 1.0)` + `optim.Adam(lr, weight_decay)` of
// /root/reference/OpenPyStruct_PINN_MultiCase.py:696, :766-768; TFD:678, :748-750) in two launches.
//
// Why: through the framework the pair is four multi-tensor launches of ~120 us per step for 0.6 M parameters (the
// fused multi-tensor Adam alone 70 us: 22 tensors, few workgroups each) out of a ~0.9 ms step.  Parameters, gradients and
// both moments live in flat float32 buffers here (the gradients already do: train.py all-reduces one flat buffer), so
// the step is a plain streaming kernel: 5 arrays x 2.4 MB.
//
//   pass 1: per-workgroup partial sums of (g * grad_scale)^2 (no atomics, nothing to zero); thread 0 advances the step
//   pass 2: clip = min(1, max_norm / (||g|| + 1e-6));  g' = clip * grad_scale * g + weight_decay * p   (L2 form, as torch)
//           (decoupled: p *= 1 - lr wd first and g' = clip * grad_scale * g -- torch.optim.AdamW, the GNN script's optimiser)
//           m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// `lr` and the step counter are device scalars so that a captured HIP graph sees the scheduler's updates.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

constexpr int FA_THREADS = 256;
constexpr int FA_NORM_BLOCKS = 128;

__global__ __launch_bounds__(FA_THREADS) void flat_grad_norm_kernel(long n, const float* __restrict__ g, float grad_scale,
                                                                     double* __restrict__ part, int32_t* __restrict__ step, float beta1,
                                                                     float beta2) {
  __shared__ double s_red[FA_THREADS / 64];
  float acc = 0.0f;
  for (long i = (long)blockIdx.x * FA_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * FA_THREADS) {
    const float v = g[i] * grad_scale;
    acc = __builtin_fmaf(v, v, acc);
  }
  double d = acc;
  for (int s = 32; s >= 1; s >>= 1) d += __shfl_xor(d, s, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < FA_THREADS / 64; ++w) t += s_red[w];
    part[blockIdx.x] = t;
    if (blockIdx.x == 0) {               // advance the step and tabulate its bias corrections ONCE (two double pow() per thread of
      const int st = step[0] + 1;        // the update kernel had cost more than its memory traffic)
      step[0] = st;
      part[FA_NORM_BLOCKS] = 1.0 - pow((double)beta1, (double)st);
      part[FA_NORM_BLOCKS + 1] = sqrt(1.0 - pow((double)beta2, (double)st));
    }
  }
}

__global__ __launch_bounds__(FA_THREADS) void flat_adam_kernel(long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                                float* __restrict__ v, const float* __restrict__ lr, const int32_t* __restrict__ step,
                                                                const double* __restrict__ part, int nparts, float max_norm, float grad_scale,
                                                                float beta1, float beta2, float eps, float weight_decay, int decoupled,
                                                                uint16_t* __restrict__ shadow) {
  double tot = 0.0;
  for (int k = 0; k < nparts; ++k) tot += part[k];
  const float norm = (float)sqrt(tot);
  float clip = max_norm > 0.0f ? max_norm / (norm + 1e-6f) : 1.0f;      // torch.nn.utils.clip_grad_norm_
  clip = clip < 1.0f ? clip : 1.0f;
  const float gs = clip * grad_scale;
  const float bc1 = (float)part[FA_NORM_BLOCKS], bc2s = (float)part[FA_NORM_BLOCKS + 1];
  const float step_size = lr[0] / bc1;
  for (long i = (long)blockIdx.x * FA_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * FA_THREADS) {
    float pi = p[i];
    float gi = g[i] * gs;
    if (decoupled) pi *= 1.0f - lr[0] * weight_decay;      // AdamW: p *= 1 - lr wd, the gradient stays clean
    else gi = __builtin_fmaf(weight_decay, pi, gi);        // Adam: L2 term in the gradient
    const float mi = __builtin_fmaf(beta1, m[i], (1.0f - beta1) * gi);
    const float vi = __builtin_fmaf(beta2, v[i], (1.0f - beta2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float pn = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
    p[i] = pn;
    if (shadow) {                      // bfloat16 copy of the parameters (round to nearest even) for the GEMMs of the next step
      uint32_t u = __float_as_uint(pn);
      u += 0x7fffu + ((u >> 16) & 1u);
      shadow[i] = (uint16_t)(u >> 16);
    }
  }
}

}  // namespace opsamd

using namespace opsamd;

extern "C" size_t ops_flat_adam_workspace_bytes(void) { return (size_t)(FA_NORM_BLOCKS + 2) * sizeof(double); }

extern "C" int ops_flat_clip_adam_step_f32(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr,
                                           int32_t* step, float max_norm, float grad_scale, float beta1, float beta2, float eps,
                                           float weight_decay, int decoupled_weight_decay, void* params_bf16, void* workspace,
                                           void* stream) {
  if (n < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !lr || !step || !workspace) return OPS_AMD_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  long nb = (n + FA_THREADS - 1) / FA_THREADS;
  const int nparts = (int)(nb < FA_NORM_BLOCKS ? nb : FA_NORM_BLOCKS);
  hipLaunchKernelGGL(flat_grad_norm_kernel, dim3(nparts), dim3(FA_THREADS), 0, s, n, grads, grad_scale, (double*)workspace, step, beta1,
                     beta2);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(flat_adam_kernel, dim3((unsigned)nb), dim3(FA_THREADS), 0, s, n, params, grads, exp_avg, exp_avg_sq, lr, step,
                     (const double*)workspace, nparts, max_norm, grad_scale, beta1, beta2, eps, weight_decay, decoupled_weight_decay,
                     (uint16_t*)params_bf16);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
