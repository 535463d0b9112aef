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
#include "repack_tiles.hpp"

namespace opsamd {

constexpr int FA_THREADS = 256;
constexpr int FA_NORM_BLOCKS = 128;     // <= FA_THREADS: the update reads one partial sum per thread

__global__ __launch_bounds__(FA_THREADS) void flat_grad_norm_kernel(long n, const float* __restrict__ g, float grad_scale,
                                                                     double* __restrict__ part, int32_t* __restrict__ step, float beta1,
                                                                     float beta2) {
  __shared__ double s_red[FA_THREADS / 64];
  // 16-byte loads, four independent partial sums per thread (the flat buffer is 16-byte aligned: a framework allocation)
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  const long n4 = ((uintptr_t)g & 15) == 0 ? n >> 2 : 0;
  for (long i = (long)blockIdx.x * FA_THREADS + threadIdx.x; i < n4; i += (long)gridDim.x * FA_THREADS) {
    const float4 q = ((const float4*)g)[i];
    const float x = q.x * grad_scale, y = q.y * grad_scale, z = q.z * grad_scale, w = q.w * grad_scale;
    a0 = __builtin_fmaf(x, x, a0); a1 = __builtin_fmaf(y, y, a1); a2 = __builtin_fmaf(z, z, a2); a3 = __builtin_fmaf(w, w, a3);
  }
  for (long i = 4 * n4 + (long)blockIdx.x * FA_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * FA_THREADS) {
    const float v = g[i] * grad_scale;
    a0 = __builtin_fmaf(v, v, a0);
  }
  double d = (double)a0 + (double)a1 + (double)a2 + (double)a3;
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
      part[OPS_FLAT_ADAM_MAX_PARTS] = 1.0 - pow((double)beta1, (double)st);
      part[OPS_FLAT_ADAM_MAX_PARTS + 1] = sqrt(1.0 - pow((double)beta2, (double)st));
    }
  }
}

__global__ __launch_bounds__(FA_THREADS) void flat_adam_kernel(long n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                                float* __restrict__ v, const float* __restrict__ lr, const int32_t* __restrict__ step,
                                                                const double* __restrict__ part, int nparts, float max_norm, float grad_scale,
                                                                float beta1, float beta2, float eps, float weight_decay, int flags,
                                                                uint16_t* __restrict__ shadow) {
  // ||g||^2 from the partial sums (the norm pass's <= FA_NORM_BLOCKS, or the gradient producers' <= OPS_FLAT_ADAM_MAX_PARTS): up to four
  // independent loads per thread, not a serial chain
  __shared__ double s_red[FA_THREADS / 64];
  double d = 0.0;
#pragma unroll
  for (int k = 0; k < OPS_FLAT_ADAM_MAX_PARTS / FA_THREADS; ++k) {
    const int i = (int)threadIdx.x + FA_THREADS * k;
    d += i < nparts ? part[i] : 0.0;
  }
  for (int s = 32; s >= 1; s >>= 1) d += __shfl_xor(d, s, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = d;
  const float bc1 = (float)part[OPS_FLAT_ADAM_MAX_PARTS], bc2s = (float)part[OPS_FLAT_ADAM_MAX_PARTS + 1];
  const float lr0 = lr[0];
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < FA_THREADS / 64; ++w) tot += s_red[w];
  const float norm = (float)sqrt(tot);
  float clip = max_norm > 0.0f ? max_norm / (norm + 1e-6f) : 1.0f;      // torch.nn.utils.clip_grad_norm_
  clip = clip < 1.0f ? clip : 1.0f;
  const float gs = clip * grad_scale;
  const float step_size = lr0 / bc1;
  const float wdf = 1.0f - lr0 * weight_decay;
  const bool decoupled = (flags & OPS_ADAM_DECOUPLED) != 0, zero_g = (flags & OPS_ADAM_ZERO_GRADS) != 0;
  auto update = [&](float pi, float gr, float& mi, float& vi) -> float {
    float gi = gr * gs;
    if (decoupled) pi *= wdf;                              // AdamW: p *= 1 - lr wd, the gradient stays clean
    else gi = __builtin_fmaf(weight_decay, pi, gi);        // Adam: L2 term in the gradient
    mi = __builtin_fmaf(beta1, mi, (1.0f - beta1) * gi);
    vi = __builtin_fmaf(beta2, vi, (1.0f - beta2) * gi * gi);
    return pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
  };
  auto to_bf16 = [](float f) -> uint16_t {                 // round to nearest even (parameters are finite)
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  };
  // 16-byte groups (the four flat buffers are framework allocations: 16-byte aligned; otherwise everything takes the scalar loop)
  const bool al = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (!shadow || ((uintptr_t)shadow & 7) == 0);
  const long n4 = al ? n >> 2 : 0;
  for (long i4 = (long)blockIdx.x * FA_THREADS + threadIdx.x; i4 < n4; i4 += (long)gridDim.x * FA_THREADS) {
    const float4 P = ((const float4*)p)[i4], G = ((const float4*)g)[i4];
    float4 M = ((const float4*)m)[i4], V = ((const float4*)v)[i4];
    float4 Q;
    Q.x = update(P.x, G.x, M.x, V.x); Q.y = update(P.y, G.y, M.y, V.y); Q.z = update(P.z, G.z, M.z, V.z); Q.w = update(P.w, G.w, M.w, V.w);
    ((float4*)m)[i4] = M; ((float4*)v)[i4] = V; ((float4*)p)[i4] = Q;
    if (zero_g) ((float4*)g)[i4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);        // optimizer.zero_grad() of the NEXT step (one fill node less per step)
    if (shadow) {                      // bfloat16 copy of the parameters for the GEMMs of the next step
      const uint16_t h0 = to_bf16(Q.x), h1 = to_bf16(Q.y), h2 = to_bf16(Q.z), h3 = to_bf16(Q.w);
      ((uint2*)shadow)[i4] = uint2{(uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16)};
    }
  }
  for (long i = 4 * n4 + (long)blockIdx.x * FA_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * FA_THREADS) {
    float mi = m[i], vi = v[i];
    const float pn = update(p[i], g[i], mi, vi);
    m[i] = mi; v[i] = vi; p[i] = pn;
    if (zero_g) g[i] = 0.0f;
    if (shadow) shadow[i] = to_bf16(pn);
  }
}


__global__ __launch_bounds__(256) void repack_tiles_kernel(const float* __restrict__ p, const AdamRepack rp, const TileJobs tj) {
  repack_tile_job(p, rp, tj, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

}  // namespace opsamd

using namespace opsamd;

extern "C" size_t ops_flat_adam_workspace_bytes(void) { return (size_t)(OPS_FLAT_ADAM_MAX_PARTS + 2) * sizeof(double); }

static int adam_step(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr, int32_t* step,
                     float max_norm, float grad_scale, float beta1, float beta2, float eps, float weight_decay, int decoupled_weight_decay,
                     void* params_bf16, void* workspace, void* stream, const AdamRepack* rp) {
  if (n < 1 || !params || !grads || !exp_avg || !exp_avg_sq || !lr || !step || !workspace) return OPS_AMD_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  long nb = (n + FA_THREADS - 1) / FA_THREADS;
  int nparts = (int)(nb < FA_NORM_BLOCKS ? nb : FA_NORM_BLOCKS);
  if (decoupled_weight_decay & OPS_ADAM_NORM_READY) {      // the gradient producers left the partial sums, the step and its corrections
    nparts = decoupled_weight_decay >> 16;
    if (nparts < 1 || nparts > OPS_FLAT_ADAM_MAX_PARTS) return OPS_AMD_ERR_INVALID_ARG;
  } else {
    hipLaunchKernelGGL(flat_grad_norm_kernel, dim3(nparts), dim3(FA_THREADS), 0, s, n, grads, grad_scale, (double*)workspace, step, beta1,
                       beta2);
  }
  nb = (n / 4 + FA_THREADS - 1) / FA_THREADS + 1;     // one 16-byte group per thread
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(flat_adam_kernel, dim3((unsigned)nb), dim3(FA_THREADS), 0, s, n, params, (float*)grads, exp_avg, exp_avg_sq, lr, step,
                     (const double*)workspace, nparts, max_norm, grad_scale, beta1, beta2, eps, weight_decay, decoupled_weight_decay & 0xFFFF,
                     (uint16_t*)params_bf16);
  if (rp) {      // the tiled weight copies: one wave per 1 KB tile, behind the update
    const TileJobs tj = make_tile_jobs(*rp);
    const int tot = tj.first[2 * rp->nmat];
    hipLaunchKernelGGL(repack_tiles_kernel, dim3((unsigned)((tot + 3) / 4)), dim3(256), 0, s, params, *rp, tj);
  }
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}

extern "C" int ops_flat_clip_adam_step_f32(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr,
                                           int32_t* step, float max_norm, float grad_scale, float beta1, float beta2, float eps,
                                           float weight_decay, int decoupled_weight_decay, void* params_bf16, void* workspace,
                                           void* stream) {
  return adam_step(n, params, grads, exp_avg, exp_avg_sq, lr, step, max_norm, grad_scale, beta1, beta2, eps, weight_decay,
                   decoupled_weight_decay, params_bf16, workspace, stream, nullptr);
}

extern "C" int ops_flat_clip_adam_step_repack_f32(long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr,
                                                  int32_t* step, float max_norm, float grad_scale, float beta1, float beta2, float eps,
                                                  float weight_decay, int decoupled_weight_decay, void* params_bf16, void* workspace,
                                                  int nmat, const ops_mlp_repack_entry* entries, void* stream) {
  AdamRepack rp{};
  const int rc = make_adam_repack(n, params, nmat, entries, &rp);
  if (rc != OPS_AMD_OK) return rc;
  return adam_step(n, params, grads, exp_avg, exp_avg_sq, lr, step, max_norm, grad_scale, beta1, beta2, eps, weight_decay,
                   decoupled_weight_decay, params_bf16, workspace, stream, &rp);
}
