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
#include "sizing_math.hpp"

namespace opsamd {


template <typename TVM>   // double: the solver's rows; float: rows the solver already rounded (ops_beam_solve_forces_f32)
__global__ __launch_bounds__(256) void sizing_step_kernel(int B, int Ne, const TVM* __restrict__ V, const TVM* __restrict__ M,
                                                          const SizingArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + wave;
  if (b >= B) return;
  if (!a.active[b]) return;   // wave-uniform
  const TVM* Vb = V + b * Ne;
  const TVM* Mb = M + b * Ne;
  sizing_case(lane, b, Ne, a, [&](int e) { return (float)Vb[e]; }, [&](int e) { return (float)Mb[e]; });
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
  const opsamd::SizingArgs a{I, I64, nullptr, exp_avg, exp_avg_sq, best_loss, patience_cnt, epochs_run, active, last_loss, V32, M32, *hp, nullptr};
  hipLaunchKernelGGL(opsamd::sizing_step_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, Ne, V, M, a);
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
  const opsamd::SizingArgs a{I, I64, nullptr, exp_avg, exp_avg_sq, best_loss, patience_cnt, epochs_run, active, last_loss, nullptr, nullptr, *hp, schedule};
  hipLaunchKernelGGL(opsamd::sizing_step_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, Ne, V32, M32, a);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
