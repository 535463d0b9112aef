// Fused 3-tap stencil + single-channel batch normalisation, forward and backward: the PINN ResidualBlock's
// `bn1(conv1(x.unsqueeze(1))).squeeze(1)` with Conv1d(1, 1, 3, padding=1) and BatchNorm1d(1)
// (/root/reference/OpenPyStruct_PINN_MultiCase.py:425-452) in ONE launch per direction.
//
// Why a kernel: through the framework the pair costs ~85 launches per training step on ROCm (MIOpen im2col + GEMM +
// col2im per sample chunk for a 3-tap filter, and a single-workgroup spatial batch norm that takes ~100 us for the
// one channel) -- 60 % of the PINN step (rocprofv3, profiles/r01_notes.md).  The whole tensor is B x F = 128 x 350
// floats (179 KB): one 1024-thread workgroup streams it twice (statistics, then normalise) out of L2, reduces in
// LDS, and needs neither atomics nor a zeroed workspace.  Statistics are accumulated in double.
//
//   y[r][i] = w0 x[r][i-1] + w1 x[r][i] + w2 x[r][i+1] + b          (zero padding at the row ends)
//   z       = gamma (y - mean) invstd + beta,   mean / biased var over all B*F values of y
//   running_mean/var: momentum update with the UNBIASED variance, num_batches_tracked += 1   (training)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

constexpr int SB_THREADS = 1024;

__device__ __forceinline__ double wave_sum_d(double v) {
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
// sums NV values over the workgroup; every thread gets the totals
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* s_red /*[16][NV]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum_d(v[k]);
  __syncthreads();                         // s_red may still be read from a previous call
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

__device__ __forceinline__ float stencil_at(const float* __restrict__ row, int i, int F, float w0, float w1, float w2, float b) {
  const float xm = i > 0 ? row[i - 1] : 0.0f, xc = row[i], xp = i + 1 < F ? row[i + 1] : 0.0f;
  return __builtin_fmaf(w0, xm, __builtin_fmaf(w1, xc, __builtin_fmaf(w2, xp, b)));
}

__global__ __launch_bounds__(SB_THREADS) void stencil_bn_fwd_kernel(int B, int F, const float* __restrict__ x, const float* __restrict__ cw,
                                                                     const float* __restrict__ cb, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float eps, float momentum, int training,
                                                                     float* running_mean, float* running_var, long long* num_batches,
                                                                     float* __restrict__ z, float* __restrict__ save) {
  __shared__ double s_red[16 * 2];
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2], b = cb[0];
  const long n = (long)B * F;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float mean, invstd;
  if (training) {
    // lane -> column, wave -> row: no index division, coalesced rows; float partial sums per thread (<= a few dozen
    // terms), double across the workgroup
    float p0 = 0.0f, p1 = 0.0f;
    for (int r = ty; r < B; r += SB_THREADS / 64) {
      const float* row = x + (long)r * F;
      for (int i = tx; i < F; i += 64) {
        const float y = stencil_at(row, i, F, w0, w1, w2, b);
        p0 += y;
        p1 = __builtin_fmaf(y, y, p1);
      }
    }
    double acc[2] = {(double)p0, (double)p1};
    block_sum<2>(acc, s_red);
    const double m = acc[0] / n, var = fmax(acc[1] / n - m * m, 0.0);
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
      running_mean[0] = (1.0f - momentum) * running_mean[0] + momentum * (float)m;
      running_var[0] = (1.0f - momentum) * running_var[0] + momentum * (float)(var * n / (n > 1 ? n - 1 : 1));
      if (num_batches) num_batches[0] += 1;
    }
  } else {
    mean = running_mean[0];
    invstd = 1.0f / sqrtf(running_var[0] + eps);
  }
  if (threadIdx.x == 0) { save[0] = mean; save[1] = invstd; }
  const float scale = gamma[0] * invstd, shift = beta[0] - mean * scale;
  for (int r = ty; r < B; r += SB_THREADS / 64) {
    const float* row = x + (long)r * F;
    float* zrow = z + (long)r * F;
    for (int i = tx; i < F; i += 64) zrow[i] = __builtin_fmaf(stencil_at(row, i, F, w0, w1, w2, b), scale, shift);
  }
}

// training-mode backward (batch statistics are functions of x): dx and d(w0, w1, w2, b, gamma, beta)
__global__ __launch_bounds__(SB_THREADS) void stencil_bn_bwd_kernel(int B, int F, const float* __restrict__ x, const float* __restrict__ g,
                                                                     const float* __restrict__ cw, const float* __restrict__ cb,
                                                                     const float* __restrict__ gamma, const float* __restrict__ save,
                                                                     int training, float* __restrict__ dx, float* __restrict__ dparams) {
  __shared__ double s_red[16 * 4];
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2], b = cb[0];
  const float mean = save[0], invstd = save[1];
  const long n = (long)B * F;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float q0 = 0.0f, q1 = 0.0f;                       // sum g, sum g * yhat
  for (int r = ty; r < B; r += SB_THREADS / 64) {
    const float* row = x + (long)r * F;
    const float* grow = g + (long)r * F;
    for (int i = tx; i < F; i += 64) {
      const float yh = (stencil_at(row, i, F, w0, w1, w2, b) - mean) * invstd;
      q0 += grow[i];
      q1 = __builtin_fmaf(grow[i], yh, q1);
    }
  }
  double acc2[2] = {(double)q0, (double)q1};
  block_sum<2>(acc2, s_red);
  const float dbeta = (float)acc2[0], dgamma = (float)acc2[1];
  // dy = gamma invstd (g - mean(g) - yhat mean(g yhat)) in training mode; gamma invstd g with frozen statistics
  const float k = gamma[0] * invstd, mg = training ? (float)(acc2[0] / n) : 0.0f, mgy = training ? (float)(acc2[1] / n) : 0.0f;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;  // dw0, dw1, dw2, db
  for (int r = ty; r < B; r += SB_THREADS / 64) {
    const float* row = x + (long)r * F;
    const float* grow = g + (long)r * F;
    float* dxrow = dx + (long)r * F;
    for (int i = tx; i < F; i += 64) {
    float dy[3];                                    // at i-1, i, i+1
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
      const int q = i + d;
      if (q >= 0 && q < F) {
        const float yh = (stencil_at(row, q, F, w0, w1, w2, b) - mean) * invstd;
        dy[d + 1] = k * (grow[q] - mg - yh * mgy);
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
  if (threadIdx.x == 0) {
    dparams[0] = (float)acc4[0]; dparams[1] = (float)acc4[1]; dparams[2] = (float)acc4[2]; dparams[3] = (float)acc4[3];
    dparams[4] = dgamma; dparams[5] = dbeta;
  }
}

}  // namespace opsamd

using namespace opsamd;

extern "C" int ops_stencil3_bn1_fwd_f32(int B, int F, const float* x, const float* conv_w, const float* conv_b, const float* gamma,
                                        const float* beta, float eps, float momentum, int training, float* running_mean,
                                        float* running_var, long long* num_batches_tracked, float* z, float* save, void* stream) {
  if (B < 1 || F < 1 || !x || !conv_w || !conv_b || !gamma || !beta || !running_mean || !running_var || !z || !save)
    return OPS_AMD_ERR_INVALID_ARG;
  if ((long)B * F > (1L << 22)) return OPS_AMD_ERR_UNSUPPORTED;     // one workgroup streams the tensor: keep it small
  hipLaunchKernelGGL(stencil_bn_fwd_kernel, dim3(1), dim3(SB_THREADS), 0, (hipStream_t)stream, B, F, x, conv_w, conv_b, gamma, beta, eps,
                     momentum, training, running_mean, running_var, num_batches_tracked, z, save);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}

extern "C" int ops_stencil3_bn1_bwd_f32(int B, int F, const float* x, const float* grad_z, const float* conv_w, const float* conv_b,
                                        const float* gamma, const float* save, int training, float* dx, float* dparams, void* stream) {
  if (B < 1 || F < 1 || !x || !grad_z || !conv_w || !conv_b || !gamma || !save || !dx || !dparams) return OPS_AMD_ERR_INVALID_ARG;
  if ((long)B * F > (1L << 22)) return OPS_AMD_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(stencil_bn_bwd_kernel, dim3(1), dim3(SB_THREADS), 0, (hipStream_t)stream, B, F, x, grad_z, conv_w, conv_b, gamma, save,
                     training, dx, dparams);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
