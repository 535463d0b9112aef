// Layer blocks of the PINN's residual MLP (/root/reference/OpenPyStruct_PINN_MultiCase.py:395-541) for the training step:
//
//   input layer    y0 = dropout(LeakyReLU(input_norm(input_fc(x))))
//   residual block h  = dropout(LeakyReLU(fc1(o)));  z = fc2(h) + bn1(conv1(o)) + o;  o' = norm(z)          (x 2)
//   output layer   p  = output_fc(o'')
//
// one launch per Linear WITH everything that follows it up to the next Linear, and one per backward counterpart -- the
// captured step of that model was 63 kernel nodes (17 library GEMMs of 5.5-19 us for 8-30 MFLOP each, the elementwise tails,
// the stencil's five kernels, casts, bias reductions; profiles/r02_train_pinn_trace.txt) and becomes 17.
//
// Shape of the work.  The batch is 128 rows and the layers are 684/350/175/302 wide: every matrix fits in L2, every product is
// a few MFLOP -- launch and latency bound, not MFMA bound.  What decides the step time is the NUMBER of dependent launches and the
// exposed memory latency inside each.  So:
//   * a workgroup owns ALL rows of 16 output columns (8 waves x one 16x16 `v_mfma_f32_16x16x32_bf16` tile each): the
//     per-column batch statistics of BatchNorm1d never leave the workgroup, forward or backward;
//   * every operand is read as the MFMA fragment itself -- 16 contiguous bytes per lane straight from global memory (L2),
//     no LDS staging -- which needs both operands contiguous along the reduction index.  Hence the layout contract of
//     include/openpystruct_amd.h: activations, gradients and weights each exist row-major AND transposed, zero-padded to
//     whole 32-column steps; producing the second copy costs the producer one extra store of a tile it already holds;
//   * all fragment loads of a reduction chunk (up to 24 steps of 32) are issued before the first MFMA: one exposed latency;
//   * the epilogue re-maps the 128 x 16 tile through LDS to "32 lanes per column": column sums are half-wave butterflies, the
//     transposed copies of the other operands (residual, saved pre-normalisation values, forward outputs for the
//     activation/dropout masks) are read and written coalesced;
//   * the ResidualBlock's single-channel BatchNorm1d(1) normalises over the WHOLE tensor: its sums are collected as a side job
//     by the launch before the one that needs them (per-workgroup partial sums in a workspace, no atomics, no extra launch);
//   * the six weight gradients of the step are one grouped launch at the end (all operands are still resident).
// Arithmetic: bf16 operands, fp32 accumulation, layer outputs rounded to bf16 -- what nn.Linear under bf16 autocast does;
// statistics, normalisation and parameter gradients in fp32 (stencil sums in fp64).
// Dropout: keep-mask from a counter-based hash of (seed, call counter, element); the backward pass reads the mask and the
// LeakyReLU branch off the saved forward OUTPUT (0 = dropped, sign = branch), nothing else is stored.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

void set_last_error(const char* msg);   // beam_solve.hip: what ops_amd_last_error() reports

typedef __bf16 mb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float mb_f32x4 __attribute__((ext_vector_type(4)));

constexpr int MB_ROWS = OPS_MLP_MAX_ROWS;   // rows a strip workgroup owns
constexpr int MB_COLS = 16;                 // output columns per workgroup
constexpr int MB_THREADS = 512;             // 8 waves, one 16 x 16 MFMA tile each
constexpr int MB_NSUM = 12;                 // backward stencil sums per workgroup

__device__ __forceinline__ float mb_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t mb_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float mb_round(float f) { return mb_bf2f(mb_f2bf(f)); }
// element (row r, column c) of a transposed bf16 matrix [cols, 128]
__device__ __forceinline__ float mb_ldt(const void* p, int c, int r) { return mb_bf2f(((const uint16_t*)p)[(long)c * MB_ROWS + r]); }
// sum over the 32 lanes that share a column (a half wave)
__device__ __forceinline__ float mb_hsum(float v) {
#pragma unroll
  for (int s = 16; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
__device__ __forceinline__ double mb_wsum_d(double v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
// splitmix64 finaliser: a counter-based uniform in [0, 1) (the stream of csrc/fused_bn.hip)
__device__ __forceinline__ float mb_uniform(uint64_t seed, uint64_t call, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (call + 1) + idx * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// ---- the product: one 16 x 16 tile per wave, reduction in steps of 32, fragments straight from global memory ----
// lane l holds A[row l&15][k = 8 (l>>4) + j] and B[k][col l&15] (j = 0..7): 16 contiguous bytes of a row of either operand.
// KCH steps are loaded before the first MFMA of the chunk; steps past KS re-read the last step (a hot line) and are skipped.
template <int KCH>
__device__ __forceinline__ mb_f32x4 mb_tile_product(const uint16_t* __restrict__ a_lane, const uint16_t* __restrict__ b_lane, int KS) {
  mb_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int k0 = 0; k0 < KS; k0 += KCH) {
    uint4 fa[KCH], fb[KCH];
#pragma unroll
    for (int j = 0; j < KCH; ++j) {
      const int ks = k0 + j < KS ? k0 + j : KS - 1;
      fa[j] = *(const uint4*)(a_lane + ks * 32);
      fb[j] = *(const uint4*)(b_lane + ks * 32);
    }
#pragma unroll
    for (int j = 0; j < KCH; ++j)
      if (k0 + j < KS)      // wave-uniform
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mb_bf16x8, fa[j]), __builtin_bit_cast(mb_bf16x8, fb[j]), acc, 0, 0, 0);
  }
  return acc;
}

// 3-tap stencil value at (row r, column q) of the transposed block input (zero padding at the ends, zero outside [0, No))
__device__ __forceinline__ float mb_conv_at(const void* Ot, int No, int q, int r, float w0, float w1, float w2, float b) {
  const float xm = q > 0 ? mb_ldt(Ot, q - 1, r) : 0.0f, xc = mb_ldt(Ot, q, r), xp = q + 1 < No ? mb_ldt(Ot, q + 1, r) : 0.0f;
  return __builtin_fmaf(w0, xm, __builtin_fmaf(w1, xc, __builtin_fmaf(w2, xp, b)));
}

// sums NV doubles over the workgroup; thread 0 gets the totals
template <int NV>
__device__ __forceinline__ void mb_block_sum(double (&v)[NV], double* s_red /*[8][NV]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = mb_wsum_d(v[k]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) s_red[wave * NV + k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double t = 0.0;
      for (int w = 0; w < MB_THREADS / 64; ++w) t += s_red[w * NV + k];
      v[k] = t;
    }
}

template <int KCH>
__global__ __launch_bounds__(MB_THREADS) void mlp_strip_kernel(const ops_mlp_strip_args a) {
  __shared__ float s_t[MB_ROWS][MB_COLS + 1];
  __shared__ __attribute__((aligned(16))) uint16_t s_y[MB_ROWS][MB_COLS];
  __shared__ double s_red[(MB_THREADS / 64) * MB_NSUM];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = blockIdx.x * MB_COLS;
  const int B = a.B, N = a.N;

  // ---- product ----
  {
    const int KS = (a.K + 31) >> 5;
    const uint16_t* ap = (const uint16_t*)a.A + (long)(wave * 16 + (lane & 15)) * a.lda + 8 * (lane >> 4);
    const uint16_t* bp = (const uint16_t*)a.W + (long)(n0 + (lane & 15)) * a.ldw + 8 * (lane >> 4);
    const mb_f32x4 acc = mb_tile_product<KCH>(ap, bp, KS);
    // C layout: column lane & 15, rows 4 (lane >> 4) + i
#pragma unroll
    for (int i = 0; i < 4; ++i) s_t[wave * 16 + (lane >> 4) * 4 + i][lane & 15] = acc[i];
  }
  __syncthreads();

  // ---- epilogue: 32 lanes per column, 4 rows per lane ----
  const int cl = tid >> 5, q = tid & 31, c = n0 + cl;
  const bool clive = c < N;
  float v[4];
  bool rl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = q + 32 * i;
    rl[i] = clive && r < B;
    v[i] = s_t[r][cl];
  }
  const float invB = 1.0f / (float)B;
  const bool fwd = a.tail <= OPS_MLP_TAIL_BN;
  const bool has_bn = a.tail == OPS_MLP_TAIL_BN || a.tail == OPS_MLP_TAIL_BN_ACT_DROP || a.tail == OPS_MLP_TAIL_BWD_BN ||
                      a.tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP;
  const float keep_scale = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  float g = 1.0f, be = 0.0f;
  if (has_bn && clive) { g = a.gamma[c]; be = a.beta[c]; }

  if (fwd) {
    const float bias = (clive && a.bias) ? mb_round(a.bias[c]) : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = rl[i] ? mb_round(v[i] + bias) : 0.0f;       // the Linear's bf16 output
    if (a.add_mode == OPS_MLP_ADD_FWD_BLOCK) {
      // whole-tensor statistics of conv1(O) from the previous launch's partial sums
      double s0 = 0.0, s1 = 0.0;
      for (int p = 0; p < a.spart_rows; ++p) { s0 += a.spart[p * 2]; s1 += a.spart[p * 2 + 1]; }
      const double n = (double)B * (double)a.No, m = s0 / n, var = fmax(s1 / n - m * m, 0.0);
      const float mean_s = (float)m, inv_s = (float)(1.0 / sqrt(var + (double)a.seps));
      if (blockIdx.x == 0 && tid == 0) {
        a.ssave[0] = mean_s; a.ssave[1] = inv_s;
        a.srunning_mean[0] = (1.0f - a.smomentum) * a.srunning_mean[0] + a.smomentum * (float)m;
        a.srunning_var[0] = (1.0f - a.smomentum) * a.srunning_var[0] + a.smomentum * (float)(var * n / (n > 1.0 ? n - 1.0 : 1.0));
        if (a.snum_batches_tracked) a.snum_batches_tracked[0] += 1;
      }
      const float w0 = a.conv_w[0], w1 = a.conv_w[1], w2 = a.conv_w[2], cb = a.conv_b[0];
      const float scale = a.sgamma[0] * inv_s, shift = a.sbeta[0] - mean_s * scale;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rl[i]) {
          const int r = q + 32 * i;
          const float o = mb_ldt(a.Ot, c, r);
          const float s = mb_round(__builtin_fmaf(mb_conv_at(a.Ot, a.No, c, r, w0, w1, w2, cb), scale, shift));
          v[i] = mb_round(v[i] + s + o);
        }
    }
    if (has_bn) {
      float sm = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sm += v[i];                 // dead rows hold 0
      const float mean = mb_hsum(sm) * invB;
      float sq = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; sq += rl[i] ? d * d : 0.0f; }
      const float var = mb_hsum(sq) * invB;                   // biased: what normalises
      const float rstd = rsqrtf(var + a.eps);
      if (clive && q == 0) {
        a.mean[c] = mean; a.rstd[c] = rstd;
        if (a.running_mean) {                                 // momentum update with the UNBIASED variance
          const float unb = var * ((float)B / (float)(B > 1 ? B - 1 : 1));
          a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mean;
          a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * unb;
        }
      }
      if (blockIdx.x == 0 && tid == 0 && a.num_batches_tracked) a.num_batches_tracked[0] += 1;
      uint16_t* zt = (uint16_t*)a.Zt;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        zt[(long)c * MB_ROWS + q + 32 * i] = mb_f2bf(v[i]);  // exact: v is a bf16 value
        v[i] = rl[i] ? __builtin_fmaf((v[i] - mean) * rstd, g, be) : 0.0f;
      }
    }
    if (a.tail == OPS_MLP_TAIL_ACT_DROP || a.tail == OPS_MLP_TAIL_BN_ACT_DROP) {
      const bool drop = a.p_drop > 0.0f;
      const unsigned long long call = drop ? *a.call_counter : 0ull;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float y = v[i] > 0.0f ? v[i] : v[i] * a.slope;
        if (drop) {
          const uint64_t e = (uint64_t)(q + 32 * i) * (uint64_t)N + (uint64_t)c;
          y = mb_uniform(a.seed, call, e) >= a.p_drop ? y * keep_scale : 0.0f;
        }
        v[i] = y;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = rl[i] ? mb_round(v[i]) : 0.0f;              // the input gradient's bf16 value
    if (a.add_mode == OPS_MLP_ADD_BWD_BLOCK) {
      // + dZ (identity path) + conv1^T( bn1 backward (dZ) ) (stencil path); the whole-tensor means from the partial sums
      double t0 = 0.0, t1 = 0.0;
      for (int p = 0; p < a.spart_rows; ++p) { t0 += a.spart[p * MB_NSUM]; t1 += a.spart[p * MB_NSUM + 1]; }
      const double n = (double)B * (double)a.No;
      const float mg = (float)(t0 / n), mgy = (float)(t1 / n);
      const float mean_s = a.ssave[0], inv_s = a.ssave[1], kk = a.sgamma[0] * inv_s;
      const float w0 = a.conv_w[0], w1 = a.conv_w[1], w2 = a.conv_w[2], cb = a.conv_b[0];
      if (blockIdx.x == 0 && tid == 0) {
        double t[MB_NSUM];
        for (int k = 0; k < MB_NSUM; ++k) t[k] = 0.0;
        for (int p = 0; p < a.spart_rows; ++p)
          for (int k = 0; k < MB_NSUM; ++k) t[k] += a.spart[p * MB_NSUM + k];
        // dy = kk (g - mg - yhat mgy):  sum dy x_s = kk (sum g x_s - mg sum x_s - mgy sum yhat x_s);  sum dy likewise with x_s = 1
        for (int s = 0; s < 3; ++s)
          a.sdparams[s] = (float)((double)kk * (t[3 + s] - (double)mg * t[6 + s] - (double)mgy * t[9 + s]));
        a.sdparams[3] = (float)((double)kk * (t[0] - (double)mg * n - (double)mgy * t[2]));
        a.sdparams[4] = (float)t[1];        // d gamma = sum g yhat
        a.sdparams[5] = (float)t[0];        // d beta  = sum g
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rl[i]) {
          const int r = q + 32 * i;
          float dy[3];
#pragma unroll
          for (int d = -1; d <= 1; ++d) {
            const int qq = c + d;
            if (qq >= 0 && qq < a.No) {
              const float yh = (mb_conv_at(a.Ot, a.No, qq, r, w0, w1, w2, cb) - mean_s) * inv_s;
              dy[d + 1] = kk * (mb_ldt(a.dZt, qq, r) - mg - yh * mgy);
            } else {
              dy[d + 1] = 0.0f;
            }
          }
          const float sdx = __builtin_fmaf(w0, dy[2], __builtin_fmaf(w1, dy[1], w2 * dy[0]));
          v[i] = mb_round(v[i] + mb_ldt(a.dZt, c, r) + sdx);
        }
    }
    if (a.tail == OPS_MLP_TAIL_BWD_ACT_DROP || a.tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP) {
      // mask and LeakyReLU branch from the saved forward output: 0 = dropped, sign = sign of the pre-activation
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rl[i]) {
          const float y = mb_ldt(a.Yref_t, c, q + 32 * i);
          v[i] *= y == 0.0f ? 0.0f : (y > 0.0f ? keep_scale : a.slope * keep_scale);
        }
    }
    if (has_bn) {
      const float mean = clive ? a.mean[c] : 0.0f, rstd = clive ? a.rstd[c] : 1.0f;
      float xh[4], sg = 0.0f, sgx = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[i] = rl[i] ? (mb_ldt(a.Zt, c, q + 32 * i) - mean) * rstd : 0.0f;
        sg += v[i];
        sgx = __builtin_fmaf(v[i], xh[i], sgx);
      }
      sg = mb_hsum(sg); sgx = mb_hsum(sgx);
      if (clive && q == 0) { a.dgamma[c] = sgx; a.dbeta[c] = sg; }
      const float k2 = g * rstd;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = rl[i] ? k2 * (v[i] - sg * invB - xh[i] * sgx * invB) : 0.0f;
    }
    if (a.dbias) {
      float sb = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sb += mb_round(v[i]);
      sb = mb_hsum(sb);
      if (clive && q == 0) a.dbias[c] = sb;
    }
  }

  // ---- results: transposed copy from the registers (a lane's rows are contiguous there), row-major copy through LDS ----
  uint16_t yb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    yb[i] = rl[i] ? mb_f2bf(v[i]) : (uint16_t)0;
    s_y[q + 32 * i][cl] = yb[i];
  }
  if (a.Yt) {
    uint16_t* yt = (uint16_t*)a.Yt;
#pragma unroll
    for (int i = 0; i < 4; ++i) yt[(long)c * MB_ROWS + q + 32 * i] = yb[i];
  }
  __syncthreads();
  if (tid < 2 * MB_ROWS) {
    const int r = tid >> 1, h = tid & 1;
    *(uint4*)((uint16_t*)a.Y + (long)r * a.ldy + n0 + 8 * h) = *(const uint4*)&s_y[r][8 * h];
  }
  if (fwd && a.p_drop > 0.0f && (a.tail == OPS_MLP_TAIL_ACT_DROP || a.tail == OPS_MLP_TAIL_BN_ACT_DROP) && blockIdx.x == 0 && tid == 0)
    atomicAdd(a.call_counter, 1ull);       // one increment per launch (a late reader draws from the next stream: as good a mask)

  // ---- side job: partial sums for the whole-tensor BatchNorm1d(1) of the stencil path, over this workgroup's column slice ----
  if (a.side != OPS_MLP_SIDE_NONE) {
    const int cs = (a.No + (int)gridDim.x - 1) / (int)gridDim.x, c0 = blockIdx.x * cs;
    const float w0 = a.conv_w[0], w1 = a.conv_w[1], w2 = a.conv_w[2], cb = a.conv_b[0];
    if (a.side == OPS_MLP_SIDE_FWD_STENCIL_STATS) {
      float p0 = 0.0f, p1 = 0.0f;
      for (int e = tid; e < cs * MB_ROWS; e += MB_THREADS) {
        const int cc = c0 + e / MB_ROWS, r = e % MB_ROWS;
        if (cc < a.No && r < B) {
          const float y = mb_conv_at(a.Ot, a.No, cc, r, w0, w1, w2, cb);
          p0 += y;
          p1 = __builtin_fmaf(y, y, p1);
        }
      }
      double acc[2] = {(double)p0, (double)p1};
      mb_block_sum<2>(acc, s_red);
      if (tid == 0) { a.spart[blockIdx.x * 2] = acc[0]; a.spart[blockIdx.x * 2 + 1] = acc[1]; }
    } else {
      // 0: sum g   1: sum g yhat   2: sum yhat   3..5: sum g x_s   6..8: sum x_s   9..11: sum yhat x_s   (x_s = O shifted by s - 1)
      const float mean_s = a.ssave[0], inv_s = a.ssave[1];
      float t[MB_NSUM];
#pragma unroll
      for (int k = 0; k < MB_NSUM; ++k) t[k] = 0.0f;
      for (int e = tid; e < cs * MB_ROWS; e += MB_THREADS) {
        const int cc = c0 + e / MB_ROWS, r = e % MB_ROWS;
        if (cc < a.No && r < B) {
          const float xs[3] = {cc > 0 ? mb_ldt(a.Ot, cc - 1, r) : 0.0f, mb_ldt(a.Ot, cc, r), cc + 1 < a.No ? mb_ldt(a.Ot, cc + 1, r) : 0.0f};
          const float conv = __builtin_fmaf(w0, xs[0], __builtin_fmaf(w1, xs[1], __builtin_fmaf(w2, xs[2], cb)));
          const float yh = (conv - mean_s) * inv_s, gi = mb_ldt(a.dZt, cc, r);
          t[0] += gi;
          t[1] = __builtin_fmaf(gi, yh, t[1]);
          t[2] += yh;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            t[3 + s] = __builtin_fmaf(gi, xs[s], t[3 + s]);
            t[6 + s] += xs[s];
            t[9 + s] = __builtin_fmaf(yh, xs[s], t[9 + s]);
          }
        }
      }
      double acc[MB_NSUM];
#pragma unroll
      for (int k = 0; k < MB_NSUM; ++k) acc[k] = (double)t[k];
      mb_block_sum<MB_NSUM>(acc, s_red);
      if (tid == 0)
#pragma unroll
        for (int k = 0; k < MB_NSUM; ++k) a.spart[blockIdx.x * MB_NSUM + k] = acc[k];
    }
  }
}

// ---- grouped weight gradients: out [N, K] fp32 = At [N, 128] x Bt [K, 128]^T, one wave per 32 x 32 tile ----
struct WgradTable {
  int nprob;
  int tile0[OPS_MLP_MAX_WGRAD + 1];        // first tile of each problem
  int tiles_k[OPS_MLP_MAX_WGRAD];
  ops_mlp_wgrad_problem p[OPS_MLP_MAX_WGRAD];
};

__global__ __launch_bounds__(64) void mlp_wgrad_kernel(const WgradTable tb) {
  int pi = 0;
  while (pi + 1 < tb.nprob && (int)blockIdx.x >= tb.tile0[pi + 1]) ++pi;
  const ops_mlp_wgrad_problem pr = tb.p[pi];
  const int t = (int)blockIdx.x - tb.tile0[pi], tn = t / tb.tiles_k[pi], tk = t % tb.tiles_k[pi];
  const int lane = threadIdx.x;
  const uint16_t* ap = (const uint16_t*)pr.At + (long)(tn * 32 + (lane & 15)) * MB_ROWS + 8 * (lane >> 4);
  const uint16_t* bp = (const uint16_t*)pr.Bt + (long)(tk * 32 + (lane & 15)) * MB_ROWS + 8 * (lane >> 4);
  uint4 fa[2][4], fb[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[h][ks] = *(const uint4*)(ap + h * 16 * MB_ROWS + ks * 32);
      fb[h][ks] = *(const uint4*)(bp + h * 16 * MB_ROWS + ks * 32);
    }
  mb_f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      acc[i][j] = mb_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mb_bf16x8, fa[i][ks]), __builtin_bit_cast(mb_bf16x8, fb[j][ks]),
                                                            acc[i][j], 0, 0, 0);
    }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = tk * 32 + j * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = tn * 32 + i * 16 + (lane >> 4) * 4 + e;
        if (n < pr.N && k < pr.K) pr.out[(long)n * pr.ldo + k] = acc[i][j][e];
      }
    }
}

// ---- padded bf16 copies of the weights, plain and transposed, from the float32 parameters ----
struct RepackTable {
  int nmat;
  long e0[OPS_MLP_MAX_WGRAD + 1];
  ops_mlp_repack_entry m[OPS_MLP_MAX_WGRAD];
};

__global__ __launch_bounds__(256) void mlp_repack_kernel(const RepackTable tb) {
  const long total = tb.e0[tb.nmat];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    int mi = 0;
    while (mi + 1 < tb.nmat && e >= tb.e0[mi + 1]) ++mi;
    const ops_mlp_repack_entry m = tb.m[mi];
    const long le = e - tb.e0[mi];
    const int n = (int)(le / m.K), k = (int)(le - (long)n * m.K);
    const uint16_t h = mb_f2bf(m.W[le]);
    ((uint16_t*)m.Wp)[(long)n * m.ldw + k] = h;
    ((uint16_t*)m.Wtp)[(long)k * m.ldwt + n] = h;
  }
}

// ---- batch assembly into the layout: gather + noise + bf16, row-major and transposed (tile of 32 features x 128 rows) ----
__device__ __forceinline__ uint64_t mb_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void mlp_gather_noise_kernel(int B, int F, const float* __restrict__ X, const long long* __restrict__ idx,
                                                                const float* __restrict__ sigma, unsigned long long seed,
                                                                unsigned long long* __restrict__ counter, uint16_t* __restrict__ out, int ld,
                                                                uint16_t* __restrict__ out_t) {
  __shared__ uint16_t s_tile[32][MB_ROWS + 2];
  const int f0 = blockIdx.x * 32, fl = threadIdx.x & 31, f = f0 + fl;
  const float sg = sigma ? *sigma : 0.0f;
  const unsigned long long call = counter ? *counter : 0ull;
  for (int b = threadIdx.x >> 5; b < MB_ROWS; b += 8) {
    uint16_t h = 0;
    if (b < B && f < F) {
      float v = X[idx[b] * (long)F + f];
      if (sg != 0.0f) {
        const uint64_t hh = mb_mix(seed + 0x9E3779B97F4A7C15ull * (call + 1) + (uint64_t)((long)b * F + f) * 0xD1B54A32D192ED03ull);
        const float u1 = ((float)(hh >> 40) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
        const float u2 = (float)((hh >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);   // [0, 1)
        v += sg * sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
      }
      h = mb_f2bf(v);
    }
    if (f < ld) out[(long)b * ld + f] = h;
    s_tile[fl][b] = h;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * MB_ROWS; e += 256) {
    const int ff = e / MB_ROWS, b = e % MB_ROWS;
    out_t[(long)(f0 + ff) * MB_ROWS + b] = s_tile[ff][b];
  }
  if (counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counter, 1ull);
}

// ---- training loss on the layout: value, gradient (both layouts), output-bias gradient; last workgroup adds the partials ----
constexpr int ML_MAXG = 64;

__global__ __launch_bounds__(MB_THREADS) void mlp_loss_kernel(int B, int C, int nI, int nD, const uint16_t* __restrict__ preds, int ldp,
                                                               const float* __restrict__ targets, const float* __restrict__ alpha_p, float alpha0,
                                                               const float* __restrict__ minc, const float* __restrict__ maxc, float w,
                                                               float penalty, float eps, float* __restrict__ loss, uint16_t* __restrict__ grad,
                                                               int ldg, uint16_t* __restrict__ grad_t, float* __restrict__ dbias,
                                                               double* __restrict__ part, unsigned int* __restrict__ done) {
  __shared__ __attribute__((aligned(16))) uint16_t s_y[MB_ROWS][MB_COLS];
  __shared__ double s_red[(MB_THREADS / 64) * 5];
  __shared__ bool s_last;
  const int tid = threadIdx.x, cl = tid >> 5, q = tid & 31, n0 = blockIdx.x * MB_COLS, c = n0 + cl;
  const float alpha = fminf(fmaxf(alpha_p[0], 1e-6f), 1.0f);
  const bool has_min = minc != nullptr, has_max = maxc != nullptr;
  const float lo = has_min ? minc[0] : 0.0f, hi = has_max ? maxc[0] : 0.0f;
  const int nR = C - nI - nD;
  const float inv_nI = 1.0f / ((float)B * (float)nI), inv_nD = nD > 0 ? 1.0f / ((float)B * (float)nD) : 0.0f,
              inv_nR = nR > 0 ? 1.0f / ((float)B * (float)nR) : 0.0f;
  float acc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // sum |d|_I, sum d^2_I, sum box penalty, sum rel_d, sum rel_r
  float sb = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = q + 32 * i;
    uint16_t gb = 0;
    if (c < C && r < B) {
      const float p = mb_bf2f(preds[(long)r * ldp + c]);
      const float t = targets[(long)r * C + c], d = p - t, sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
      float g;
      if (c < nI) {
        acc[0] += fabsf(d);
        acc[1] = __builtin_fmaf(d, d, acc[1]);
        g = (alpha * sg + (1.0f - alpha) * 2.0f * d) * inv_nI;
        if (has_min && p < lo) { acc[2] += lo - p; g -= w; }
        if (has_max && p > hi) { acc[2] += p - hi; g += w; }
      } else {
        const float den = fabsf(t) + eps, rel = fabsf(d) / den;
        if (c < nI + nD) { acc[3] += rel; g = penalty * sg / den * inv_nD; }
        else { acc[4] += rel; g = penalty * sg / den * inv_nR; }
      }
      gb = mb_f2bf(g);
      sb += mb_bf2f(gb);
    }
    s_y[r][cl] = gb;
    grad_t[(long)c * MB_ROWS + r] = gb;
  }
  sb = mb_hsum(sb);
  if (c < C && q == 0) dbias[c] = sb;
  double accd[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) accd[k] = (double)acc[k];
  mb_block_sum<5>(accd, s_red);       // (contains the barriers that publish s_y)
  if (tid < 2 * MB_ROWS) {
    const int r = tid >> 1, h = tid & 1;
    *(uint4*)(grad + (long)r * ldg + n0 + 8 * h) = *(const uint4*)&s_y[r][8 * h];
  }
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) part[blockIdx.x * 5 + k] = accd[k];
    __threadfence();
    s_last = atomicAdd(done, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (s_last && tid == 0) {
    __threadfence();
    double t[5] = {0, 0, 0, 0, 0};
    for (int gq = 0; gq < (int)gridDim.x; ++gq)
      for (int k = 0; k < 5; ++k) t[k] += ((volatile double*)part)[gq * 5 + k];
    const double al = fmin(fmax((double)alpha_p[0], 1e-6), 1.0);
    const double nIe = (double)B * nI;
    double v = al * t[0] / nIe + (1.0 - al) * t[1] / nIe + (double)w * t[2];
    if (nD > 0) v += (double)penalty * t[3] / ((double)B * nD);
    if (nR > 0) v += (double)penalty * t[4] / ((double)B * nR);
    const double da = alpha0 == alpha0 ? (double)alpha0 - (double)alpha_p[0] : 0.0;      // NaN alpha0: no such term
    loss[0] = (float)(v + da * da);
    *done = 0u;                        // ready for the next launch (graph replay)
  }
}

}  // namespace opsamd

using namespace opsamd;

static inline int ru(int v, int m) { return (v + m - 1) / m * m; }

extern "C" size_t ops_mlp_spart_doubles(int N) { return (size_t)((N + MB_COLS - 1) / MB_COLS) * MB_NSUM; }

extern "C" int ops_mlp_strip_launch(const ops_mlp_strip_args* args, void* stream) {
  if (!args) return OPS_AMD_ERR_INVALID_ARG;
  const ops_mlp_strip_args& a = *args;
  if (a.B < 1 || a.B > MB_ROWS || a.N < 1 || a.K < 1 || !a.A || !a.W || !a.Y) return OPS_AMD_ERR_INVALID_ARG;
  if (a.lda % 8 || a.ldw % 8 || a.ldy % 8 || a.lda < ru(a.K, 32) || a.ldw < ru(a.K, 32) || a.ldy < ru(a.N, MB_COLS)) return OPS_AMD_ERR_INVALID_ARG;
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.Y) & 15) return OPS_AMD_ERR_INVALID_ARG;
  if (a.tail < OPS_MLP_TAIL_NONE || a.tail > OPS_MLP_TAIL_BWD_BN_ACT_DROP) return OPS_AMD_ERR_INVALID_ARG;
  const bool bn = a.tail == OPS_MLP_TAIL_BN || a.tail == OPS_MLP_TAIL_BN_ACT_DROP || a.tail == OPS_MLP_TAIL_BWD_BN || a.tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP;
  const bool bwd = a.tail >= OPS_MLP_TAIL_BWD_ACT_DROP;
  if (bn && (!a.gamma || !a.beta || !a.mean || !a.rstd || !a.Zt)) return OPS_AMD_ERR_INVALID_ARG;
  if (bn && bwd && (!a.dgamma || !a.dbeta)) return OPS_AMD_ERR_INVALID_ARG;
  if ((a.tail == OPS_MLP_TAIL_BWD_ACT_DROP || a.tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP) && !a.Yref_t) return OPS_AMD_ERR_INVALID_ARG;
  if ((a.tail == OPS_MLP_TAIL_ACT_DROP || a.tail == OPS_MLP_TAIL_BN_ACT_DROP) && a.p_drop > 0.0f && (!a.call_counter || a.p_drop >= 1.0f))
    return OPS_AMD_ERR_INVALID_ARG;
  if (a.add_mode != OPS_MLP_ADD_NONE || a.side != OPS_MLP_SIDE_NONE) {
    if (!a.Ot || a.No < 1 || !a.conv_w || !a.conv_b || !a.sgamma || !a.sbeta || !a.ssave || !a.spart) return OPS_AMD_ERR_INVALID_ARG;
    if (a.add_mode == OPS_MLP_ADD_FWD_BLOCK && (bwd || a.No != a.N || !a.srunning_mean || !a.srunning_var || a.spart_rows < 1)) return OPS_AMD_ERR_INVALID_ARG;
    if (a.add_mode == OPS_MLP_ADD_BWD_BLOCK && (!bwd || a.No != a.N || !a.dZt || !a.sdparams || a.spart_rows < 1)) return OPS_AMD_ERR_INVALID_ARG;
    if (a.side == OPS_MLP_SIDE_BWD_STENCIL_SUMS && !a.dZt) return OPS_AMD_ERR_INVALID_ARG;
  }
  const int KS = (a.K + 31) / 32;
  const dim3 grid((unsigned)((a.N + MB_COLS - 1) / MB_COLS)), block(MB_THREADS);
  hipStream_t s = (hipStream_t)stream;
  if (KS <= 8) hipLaunchKernelGGL(mlp_strip_kernel<8>, grid, block, 0, s, a);
  else if (KS <= 12) hipLaunchKernelGGL(mlp_strip_kernel<12>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(mlp_strip_kernel<24>, grid, block, 0, s, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}

extern "C" int ops_mlp_wgrad_group(int nprob, const ops_mlp_wgrad_problem* problems, void* stream) {
  if (nprob < 1 || nprob > OPS_MLP_MAX_WGRAD || !problems) return OPS_AMD_ERR_INVALID_ARG;
  WgradTable tb;
  tb.nprob = nprob;
  int tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const ops_mlp_wgrad_problem& p = problems[i];
    if (!p.At || !p.Bt || !p.out || p.N < 1 || p.K < 1 || p.ldo < p.K) return OPS_AMD_ERR_INVALID_ARG;
    if (((uintptr_t)p.At | (uintptr_t)p.Bt) & 15) return OPS_AMD_ERR_INVALID_ARG;
    tb.p[i] = p;
    tb.tile0[i] = tiles;
    tb.tiles_k[i] = (p.K + 31) / 32;
    tiles += ((p.N + 31) / 32) * tb.tiles_k[i];
  }
  tb.tile0[nprob] = tiles;
  hipLaunchKernelGGL(mlp_wgrad_kernel, dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, tb);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}

extern "C" int ops_mlp_repack_weights(int nmat, const ops_mlp_repack_entry* entries, void* stream) {
  if (nmat < 1 || nmat > OPS_MLP_MAX_WGRAD || !entries) return OPS_AMD_ERR_INVALID_ARG;
  RepackTable tb;
  tb.nmat = nmat;
  long tot = 0;
  for (int i = 0; i < nmat; ++i) {
    const ops_mlp_repack_entry& m = entries[i];
    if (!m.W || !m.Wp || !m.Wtp || m.N < 1 || m.K < 1 || m.ldw < ru(m.K, 32) || m.ldwt < ru(m.N, 32)) return OPS_AMD_ERR_INVALID_ARG;
    tb.m[i] = m;
    tb.e0[i] = tot;
    tot += (long)m.N * m.K;
  }
  tb.e0[nmat] = tot;
  long nb = (tot + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(mlp_repack_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, tb);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}

extern "C" int ops_mlp_gather_noise(int B, int F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                    unsigned long long* counter, void* out, int ld, void* out_t, void* stream) {
  if (B < 1 || B > MB_ROWS || F < 1 || !X || !idx || !out || !out_t || ld % 8 || ld < ru(F, 32)) return OPS_AMD_ERR_INVALID_ARG;
  hipLaunchKernelGGL(mlp_gather_noise_kernel, dim3((unsigned)((F + 31) / 32)), dim3(256), 0, (hipStream_t)stream, B, F, X, idx, sigma, seed,
                     counter, (uint16_t*)out, ld, (uint16_t*)out_t);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}

extern "C" size_t ops_mlp_loss_workspace_bytes(void) { return (size_t)ML_MAXG * 5 * sizeof(double) + 16; }

extern "C" int ops_mlp_loss_grad(int B, int C, int nI, int nD, const void* preds, int ldp, const float* targets, const float* alpha, float alpha0,
                                 const float* min_constraint, const float* max_constraint, float box_weight, float rel_penalty, float* loss,
                                 void* grad, int ldg, void* grad_t, float* dbias, void* workspace, void* stream) {
  if (B < 1 || B > MB_ROWS || C < 1 || nI < 1 || nD < 0 || nI + nD > C || !preds || !targets || !alpha || !loss || !grad || !grad_t || !dbias ||
      !workspace)
    return OPS_AMD_ERR_INVALID_ARG;
  const int G = (C + MB_COLS - 1) / MB_COLS;
  if (G > ML_MAXG || ldp < C || ldg % 8 || ldg < ru(C, MB_COLS) || ((uintptr_t)grad & 15)) return OPS_AMD_ERR_INVALID_ARG;
  double* part = (double*)workspace;
  unsigned int* done = (unsigned int*)(part + ML_MAXG * 5);
  hipLaunchKernelGGL(mlp_loss_kernel, dim3((unsigned)G), dim3(MB_THREADS), 0, (hipStream_t)stream, B, C, nI, nD, (const uint16_t*)preds, ldp, targets,
                     alpha, alpha0, min_constraint, max_constraint, box_weight, rel_penalty, 1e-8f, loss, (uint16_t*)grad, ldg, (uint16_t*)grad_t,
                     dbias, part, done);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}
