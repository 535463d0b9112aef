// Per-case sizing optimiser step for a whole batch: loss, its gradient w.r.t. the element
// inertias, Adam update with exponential learning-rate decay, clamp, early-stop bookkeeping.
//
// Replaces, for B cases per launch, one epoch of the reference's per-sample loop
// (/root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py):
//   :189-190  shear / moment rounded to float32          (V32, M32 outputs)
//   :195-199  total_loss = sum(I) + a_M * sum(M^2 / (2 E I + 1e-6)) + a_V * sum(V^2 / (G * 0.03 * sqrt(I)))
//   :202      total_loss.backward()  (M, V are constants of the graph: only the explicit I terms differentiate)
//   :203-204  Adam(lr 0.01) step, ExponentialLR(0.98) step
//   :208      clamp_(min=1e-8)
//   :211-219  early stopping (loss < best - tol, patience)
// All arithmetic on I is float32 like the reference's I_tensor (:163); the widened float64 copy the
// next FE solve reads (`I_tensor[i].item()`, :107) is refreshed only while the case is still active,
// so the solver's outputs of a finished case stay those of its LAST solve (the reference records
// shear/moment/displacements one Adam step behind I_values, :189-208 vs :239).
//
// Mapping: one 64-lane wavefront per case, lane e handles elements e, e+64, ... (Ne <= 512) (HBM-bound
// elementwise work, coalesced rows, wave-level DPP/shuffle reduction for the loss).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
  return x;
}

template <typename TVM>   // double: the solver's rows; float: rows the solver already rounded (ops_beam_solve_forces_f32)
__global__ __launch_bounds__(256) void sizing_step_kernel(int B, int Ne, float* __restrict__ I, double* __restrict__ I64,
                                                          const TVM* __restrict__ V, const TVM* __restrict__ M,
                                                          float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                          float* __restrict__ best_loss, int32_t* __restrict__ patience_cnt,
                                                          int32_t* __restrict__ epochs_run, uint8_t* __restrict__ active,
                                                          float* __restrict__ last_loss, float* __restrict__ V32,
                                                          float* __restrict__ M32, const ops_sizing_params hp,
                                                          const float* __restrict__ schedule) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + wave;
  if (b >= B) return;
  if (!active[b]) return;   // wave-uniform
  const int t = epochs_run[b];          // 0-based epoch of this case == optimiser step count so far
  const float twoE = (float)(2.0 * hp.E), Gf = (float)hp.G;
  // step_size = lr gamma^t / (1 - beta1^(t+1)) and sqrt(1 - beta2^(t+1)): three double-precision pow() per wavefront cost
  // more than the rest of the kernel (0.33 ms of a 0.4 ms epoch at 2e5 cases); callers may pass them tabulated per epoch
  float step_size, bc2s;
  if (schedule) {
    step_size = schedule[2 * t];
    bc2s = schedule[2 * t + 1];
  } else {
    const float lr_t = (float)(hp.lr * pow(hp.gamma, (double)t));
    const float bc1 = (float)(1.0 - pow(hp.beta1, (double)(t + 1)));
    bc2s = (float)sqrt(1.0 - pow(hp.beta2, (double)(t + 1)));
    step_size = lr_t / bc1;
  }
  float lsum_I = 0.f, lsum_b = 0.f, lsum_s = 0.f;
  constexpr int KMAX = 8;               // Ne <= 512
  float Inew[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int e = lane + 64 * k;
    Inew[k] = 0.f;
    if (e < Ne) {
      const long o = b * Ne + e;
      const float Ie = I[o];
      const float m = (float)M[o], v = (float)V[o];      // torch.tensor(..., dtype=float32)
      if (V32) { V32[o] = v; M32[o] = m; }                            // wave-uniform: the generator rounds them once, at the end
      const float den_b = twoE * Ie + (float)hp.bend_eps;              // 2*E*I + 1e-6
      const float sq = sqrtf(Ie);                                      // I ** 0.5
      const float den_s = Gf * ((float)hp.area_coef * sq);             // G * (0.03 * I**0.5)
      lsum_I += Ie;
      lsum_b += (m * m) / den_b;
      lsum_s += (v * v) / den_s;
      // d/dI: 1 - a_M * M^2 * 2E / den_b^2 - a_V * V^2 / den_s^2 * G * 0.03 * 0.5 / sqrt(I)
      const float g = 1.0f - (float)hp.alpha_moment * ((m * m) / (den_b * den_b)) * twoE -
                      (float)hp.alpha_shear * ((v * v) / (den_s * den_s)) * (Gf * (float)hp.area_coef * (0.5f / sq));
      const float ea = (float)hp.beta1 * exp_avg[o] + (1.0f - (float)hp.beta1) * g;
      const float es = (float)hp.beta2 * exp_avg_sq[o] + (1.0f - (float)hp.beta2) * g * g;
      exp_avg[o] = ea;
      exp_avg_sq[o] = es;
      const float denom = sqrtf(es) / bc2s + (float)hp.adam_eps;
      float In = Ie - step_size * (ea / denom);
      In = fmaxf(In, (float)hp.clamp_min);
      I[o] = In;
      Inew[k] = In;
    }
  }
  const float loss = wave_sum(lsum_I) + (float)hp.alpha_moment * wave_sum(lsum_b) + (float)hp.alpha_shear * wave_sum(lsum_s);
  // early stopping (SingleCore.py:211-219), decided identically by every lane
  float best = best_loss[b];
  int cnt = patience_cnt[b];
  if (loss < best - (float)hp.tolerance) { best = loss; cnt = 0; } else { cnt += 1; }
  const bool stop = (cnt >= hp.patience) || (t + 1 >= hp.max_epochs);
  if (!stop) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int e = lane + 64 * k;
      if (e < Ne) I64[b * Ne + e] = (double)Inew[k];   // what the next solve reads
    }
  }
  if (lane == 0) {
    best_loss[b] = best;
    patience_cnt[b] = cnt;
    epochs_run[b] = t + 1;
    last_loss[b] = loss;
    if (stop) active[b] = 0;
  }
}

}  // namespace opsamd

extern "C" int ops_beam_sizing_step_f32(int B, int Ne, float* I, double* I64, const double* V, const double* M,
                                        float* exp_avg, float* exp_avg_sq, float* best_loss, int32_t* patience_cnt,
                                        int32_t* epochs_run, uint8_t* active, float* last_loss, float* V32, float* M32,
                                        const ops_sizing_params* hp, void* stream) {
  if (B < 0 || Ne < 1 || Ne > 512) return Ne > 512 ? OPS_AMD_ERR_UNSUPPORTED : OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!I || !I64 || !V || !M || !exp_avg || !exp_avg_sq || !best_loss || !patience_cnt || !epochs_run || !active ||
      !last_loss || ((V32 == nullptr) != (M32 == nullptr)) || !hp)
    return OPS_AMD_ERR_INVALID_ARG;
  const unsigned grid = (unsigned)((B + 3) / 4);
  hipLaunchKernelGGL(opsamd::sizing_step_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, Ne, I, I64, V, M,
                     exp_avg, exp_avg_sq, best_loss, patience_cnt, epochs_run, active, last_loss, V32, M32, *hp, (const float*)nullptr);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}

// schedule[2 t] = (float)(lr gamma^t) / (float)(1 - beta1^(t+1)), schedule[2 t + 1] = (float)sqrt(1 - beta2^(t+1)), t < max_epochs
extern "C" void ops_sizing_schedule_f32(const ops_sizing_params* hp, float* schedule_host) {
  for (int t = 0; t < hp->max_epochs; ++t) {
    const float lr_t = (float)(hp->lr * pow(hp->gamma, (double)t));
    const float bc1 = (float)(1.0 - pow(hp->beta1, (double)(t + 1)));
    schedule_host[2 * t] = lr_t / bc1;
    schedule_host[2 * t + 1] = (float)sqrt(1.0 - pow(hp->beta2, (double)(t + 1)));
  }
}

extern "C" int ops_beam_sizing_step_vm32_f32(int B, int Ne, float* I, double* I64, const float* V32, const float* M32,
                                             float* exp_avg, float* exp_avg_sq, float* best_loss, int32_t* patience_cnt,
                                             int32_t* epochs_run, uint8_t* active, float* last_loss, const ops_sizing_params* hp,
                                             const float* schedule, void* stream) {
  if (B < 0 || Ne < 1 || Ne > 512) return Ne > 512 ? OPS_AMD_ERR_UNSUPPORTED : OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!I || !I64 || !V32 || !M32 || !exp_avg || !exp_avg_sq || !best_loss || !patience_cnt || !epochs_run || !active || !last_loss || !hp)
    return OPS_AMD_ERR_INVALID_ARG;
  const unsigned grid = (unsigned)((B + 3) / 4);
  hipLaunchKernelGGL(opsamd::sizing_step_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, Ne, I, I64, V32, M32,
                     exp_avg, exp_avg_sq, best_loss, patience_cnt, epochs_run, active, last_loss, (float*)nullptr, (float*)nullptr, *hp,
                     schedule);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
