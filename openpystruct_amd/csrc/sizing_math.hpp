// One optimiser epoch of ONE case, executed by a whole 64-lane wavefront: shared by the stand-alone step kernel
// (csrc/sizing_step.hip) and the fused solve + step kernel (csrc/beam_solve.hip).  See sizing_step.hip for the reference
// lines this follows.  getV(e) / getM(e): this case's shear / moment of element e, already rounded to float32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

struct SizingArgs {
  float* I; double* I64;            // I64 (optional): widened copy for a separate solve kernel, frozen when a case stops
  float* I_last;                    // optional: the inertias a case's LAST solve used (written once, when it stops)
  float* exp_avg; float* exp_avg_sq;
  float* best_loss; int32_t* patience_cnt; int32_t* epochs_run; uint8_t* active; float* last_loss;
  float* V32; float* M32;            // optional records of the rounded forces (NULL: the caller rounds them later)
  ops_sizing_params hp;
  const float* schedule;             // optional [max_epochs, 2]: step size, sqrt(1 - beta2^(t+1)) (ops_sizing_schedule_f32)
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
  return x;
}

// state of one case as one wavefront holds it between the loads and the arithmetic (K = elements per lane)
template <int K>
struct CaseRegs {
  float I[K], m[K], v[K];
  int t, cnt;
  float best;
};

// issue every load of one case (nothing waits here: several cases' loads can be in flight before the first is used)
template <int K>
__device__ __forceinline__ void load_case(int lane, long b, int Ne, const SizingArgs& a, CaseRegs<K>& r) {
  r.t = a.epochs_run[b];                // 0-based epoch of this case == optimiser step count so far
  r.best = a.best_loss[b];
  r.cnt = a.patience_cnt[b];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = lane + 64 * k;
    const long o = b * Ne + (e < Ne ? e : 0);
    r.I[k] = a.I[o];
    r.m[k] = a.exp_avg[o];
    r.v[k] = a.exp_avg_sq[o];
  }
}

template <int K, class FV, class FM>
__device__ __forceinline__ void step_case(int lane, long b, int Ne, const SizingArgs& a, const CaseRegs<K>& r, FV getV, FM getM) {
  const ops_sizing_params& hp = a.hp;
  const int t = r.t;
  const float twoE = (float)(2.0 * hp.E), Gf = (float)hp.G;
  // step_size = lr gamma^t / (1 - beta1^(t+1)) and sqrt(1 - beta2^(t+1)): three double-precision pow() per wavefront cost
  // more than the rest of the kernel (0.33 ms of a 0.4 ms epoch at 2e5 cases); callers may pass them tabulated per epoch
  float step_size, bc2s;
  if (a.schedule) {
    step_size = a.schedule[2 * t];
    bc2s = a.schedule[2 * t + 1];
  } else {
    const float lr_t = (float)(hp.lr * pow(hp.gamma, (double)t));
    const float bc1 = (float)(1.0 - pow(hp.beta1, (double)(t + 1)));
    bc2s = (float)sqrt(1.0 - pow(hp.beta2, (double)(t + 1)));
    step_size = lr_t / bc1;
  }
  float lsum_I = 0.f, lsum_b = 0.f, lsum_s = 0.f;
  float Inew[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = lane + 64 * k;
    Inew[k] = 0.f;
    if (e < Ne) {
      const long o = b * Ne + e;
      const float Ie = r.I[k];
      const float m = getM(e), v = getV(e);                            // torch.tensor(..., dtype=float32)
      if (a.V32) { a.V32[o] = v; a.M32[o] = m; }                      // wave-uniform: the generator rounds them once, at the end
      const float den_b = twoE * Ie + (float)hp.bend_eps;              // 2*E*I + 1e-6
      const float sq = sqrtf(Ie);                                      // I ** 0.5
      const float den_s = Gf * ((float)hp.area_coef * sq);             // G * (0.03 * I**0.5)
      lsum_I += Ie;
      lsum_b += (m * m) / den_b;
      lsum_s += (v * v) / den_s;
      // d/dI: 1 - a_M * M^2 * 2E / den_b^2 - a_V * V^2 / den_s^2 * G * 0.03 * 0.5 / sqrt(I)
      const float g = 1.0f - (float)hp.alpha_moment * ((m * m) / (den_b * den_b)) * twoE -
                      (float)hp.alpha_shear * ((v * v) / (den_s * den_s)) * (Gf * (float)hp.area_coef * (0.5f / sq));
      const float ea = (float)hp.beta1 * r.m[k] + (1.0f - (float)hp.beta1) * g;
      const float es = (float)hp.beta2 * r.v[k] + (1.0f - (float)hp.beta2) * g * g;
      a.exp_avg[o] = ea;
      a.exp_avg_sq[o] = es;
      const float denom = sqrtf(es) / bc2s + (float)hp.adam_eps;
      float In = Ie - step_size * (ea / denom);
      In = fmaxf(In, (float)hp.clamp_min);
      a.I[o] = In;
      Inew[k] = In;
    }
  }
  const float loss = wave_sum(lsum_I) + (float)hp.alpha_moment * wave_sum(lsum_b) + (float)hp.alpha_shear * wave_sum(lsum_s);
  // early stopping (SingleCore.py:211-219), decided identically by every lane
  float best = r.best;
  int cnt = r.cnt;
  if (loss < best - (float)hp.tolerance) { best = loss; cnt = 0; } else { cnt += 1; }
  const bool stop = (cnt >= hp.patience) || (t + 1 >= hp.max_epochs);
  if (a.I64 && !stop) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int e = lane + 64 * k;
      if (e < Ne) a.I64[b * Ne + e] = (double)Inew[k];   // what the next (separate) solve reads
    }
  }
  if (a.I_last && stop) {                                // the fused epoch keeps no widened copy: it records, once, the
#pragma unroll                                           // float32 inertias this last solve ran on (one-step lag, :239)
    for (int k = 0; k < K; ++k) {
      const int e = lane + 64 * k;
      if (e < Ne) a.I_last[b * Ne + e] = r.I[k];
    }
  }
  if (lane == 0) {
    a.best_loss[b] = best;
    a.patience_cnt[b] = cnt;
    a.epochs_run[b] = t + 1;
    a.last_loss[b] = loss;
    if (stop) a.active[b] = 0;
  }
}

// the stand-alone form: one case, Ne <= 512
template <class FV, class FM>
__device__ __forceinline__ void sizing_case(int lane, long b, int Ne, const SizingArgs& a, FV getV, FM getM) {
  CaseRegs<8> r;
  load_case<8>(lane, b, Ne, a, r);
  step_case<8>(lane, b, Ne, a, r, getV, getM);
}

}  // namespace opsamd
