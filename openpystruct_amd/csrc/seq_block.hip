// Row-wise blocks of the Transformer-Diffusion surrogate's encoder (/root/reference/OpenPyStruct_TransformerDiffusionModule_MultiCase.py:
// 539-575: nn.TransformerEncoderLayer, post-norm, ReLU, batch_first, dropout 0.1) for the training step, one launch per direction each:
//
//   ops_seq_attention_fwd / _bwd      softmax(q k^T / sqrt(dh)) -> dropout -> @ v   for sequences of S <= 8 tokens ([CLS] + 6 load cases)
//   ops_dropout_add_layernorm_fwd / _bwd      LayerNorm(residual + dropout(x)), float32 AND bfloat16 copies of the result
//   ops_act_dropout_fwd / _bwd        dropout(ReLU(x)) (LeakyReLU(slope) in general)
//
// Why: through the framework an encoder layer of this model is ~90 kernel nodes per training step (profiles/r02_notes.md section 8):
// the attention of 7 tokens goes through a flash-attention kernel pair built for long sequences (8 us forward, 41 us backward) wrapped
// in a dozen layout copies and fills, every "dropout -> add -> LayerNorm" is 3 nodes forward and 6 backward (plus a 43 us mixed-dtype
// add), every ReLU / dropout a node each way.  These three kernels + the shadow GEMMs make it ~26.
//
// Dropout everywhere: keep-mask from a counter-based hash of (seed, call counter, element).  The counter lives in device memory and is
// advanced by the CALLER between steps (one increment node per step; fresh masks under HIP-graph replay) -- never inside these
// launches: every workgroup of a launch must read the same value, because the backward launch REGENERATES the mask from the value the
// forward launch left in `used_call` instead of reading a stored one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/openpystruct_amd.h"
#include "dropout_stream.hpp"

namespace opsamd {

__device__ __forceinline__ float sq_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t sq_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float sq_uniform(uint64_t seed, uint64_t call, uint64_t idx) { return drop_uniform(seed, call, idx); }   // csrc/dropout_stream.hpp
__device__ __forceinline__ float sq_wsum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// ================================================================================================================================
// attention over short sequences: a workgroup stages the q|k|v rows of NB samples in LDS (coalesced 16-byte loads), one thread per
// (sample, head, query) computes its row of the S x S problem from there
// ================================================================================================================================
constexpr int SA_MAXS = 8;
constexpr int SA_THREADS = 512;

struct SaArgs {
  int Bn, S, H, dh;              // samples, tokens per sample, heads, head width (d = H * dh)
  int NB;                        // samples per workgroup
  const uint16_t* qkv;           // [Bn * S, 3 d] bf16
  const uint16_t* dctx;          // backward: [Bn * S, d] bf16
  uint16_t* ctx;                 // forward out [Bn * S, d]
  uint16_t* dqkv;                // backward out [Bn * S, 3 d]
  float p_drop; unsigned long long seed; unsigned long long* counter; unsigned long long* used_call;
};

// Head vectors live in LDS padded to DHP (16 or 32) elements, 16-byte aligned: a thread reads a whole q / k / v / dO head vector
// with one or two ds_read_b128 instead of dh two-byte reads (the 7-token problem is LDS-instruction bound otherwise: 225 reads per
// thread forward, 650 backward, vs 30 / 74).  LDS image of a row: [q | k | v][H][DHP] (and [H][DHP] for dO rows).

// stage `rows` rows of `nseg` head segments of `dh` elements each (global: contiguous rows of nseg * dh) into the padded image
template <int DHP>
__device__ __forceinline__ void sa_stage_padded(uint16_t* __restrict__ dst, const uint16_t* __restrict__ src, int rows, int nseg, int dh) {
  const int rowlen = nseg * dh, n16 = rows * rowlen / 8;              // rowlen % 8 == 0 (checked on the host)
  for (int e0 = threadIdx.x; e0 < n16; e0 += 4 * SA_THREADS) {
    uint4 t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int e = e0 + k * SA_THREADS; t[k] = e < n16 ? ((const uint4*)src)[e] : uint4{0u, 0u, 0u, 0u}; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = e0 + k * SA_THREADS;
      if (e < n16) {
        const uint16_t* v = (const uint16_t*)&t[k];
        const int r = (e * 8) / rowlen, c0 = e * 8 - r * rowlen;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c0 + j, seg = c / dh, tt = c - seg * dh;
          dst[(r * nseg + seg) * DHP + tt] = v[j];
        }
      }
    }
  }
}
// zero the padding lanes dh .. DHP - 1 of every head vector (once per launch; they take part in the vector dot products)
template <int DHP>
__device__ __forceinline__ void sa_zero_pad(uint16_t* __restrict__ dst, int nvec, int dh) {
  const int np = DHP - dh;
  for (int e = threadIdx.x; e < nvec * np; e += SA_THREADS) dst[(e / np) * DHP + dh + e % np] = 0;
}
template <int DHP>
__device__ __forceinline__ void sa_ldvec(const uint16_t* __restrict__ p, float (&v)[DHP]) {
#pragma unroll
  for (int k = 0; k < DHP / 8; ++k) {
    const uint4 u = ((const uint4*)p)[k];
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[8 * k + 2 * j] = __uint_as_float(w[j] << 16); v[8 * k + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
  }
}
template <int DHP>
__device__ __forceinline__ float sa_dot(const float (&a)[DHP], const float (&b)[DHP]) {
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < DHP; ++t) s = __builtin_fmaf(a[t], b[t], s);
  return s;
}

// this thread's attention row: probabilities p[j] (after softmax) and the keep-scaled ones pk[j]; kbase: the row block's k vectors
template <int DHP>
__device__ __forceinline__ void sa_row(const float (&q)[DHP], const uint16_t* __restrict__ krow0, int rowstride, int S, float scale, float p_drop,
                                       uint64_t seed, uint64_t call, uint64_t eidx0, float (&p)[SA_MAXS], float (&pk)[SA_MAXS]) {
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < SA_MAXS; ++j) {
    p[j] = -3.0e38f;
    if (j < S) {
      float kv[DHP];
      sa_ldvec<DHP>(krow0 + j * rowstride, kv);
      p[j] = sa_dot<DHP>(q, kv) * scale;
      mx = fmaxf(mx, p[j]);
    }
  }
  float den = 0.0f;
#pragma unroll
  for (int j = 0; j < SA_MAXS; ++j) { p[j] = j < S ? __expf(p[j] - mx) : 0.0f; den += p[j]; }
  const float inv = 1.0f / den, ks = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
#pragma unroll
  for (int j = 0; j < SA_MAXS; ++j) {
    p[j] *= inv;
    const bool keep = !(p_drop > 0.0f) || sq_uniform(seed, call, eidx0 + j) >= p_drop;
    pk[j] = (j < S && keep) ? p[j] * ks : 0.0f;
  }
}

template <int DHP>
__global__ __launch_bounds__(SA_THREADS) void seq_attention_fwd_kernel(const SaArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t s_mem[];
  const int H = a.H, dh = a.dh, d = H * dh, S = a.S;
  const int b0 = blockIdx.x * a.NB, nb = min(a.NB, a.Bn - b0), rows = nb * S;
  const int rs = 3 * H * DHP;                                         // LDS row: [q | k | v][H][DHP]
  uint16_t* s_qkv = s_mem;
  sa_zero_pad<DHP>(s_qkv, a.NB * S * 3 * H, dh);
  sa_stage_padded<DHP>(s_qkv, a.qkv + (long)b0 * S * 3 * d, rows, 3 * H, dh);
  const unsigned long long call = a.p_drop > 0.0f ? *a.counter : 0ull;
  __syncthreads();
  const int tid = threadIdx.x, i = tid % S, h = (tid / S) % H, bl = tid / (S * H);
  if (bl < nb) {
    const float scale = rsqrtf((float)dh);
    const uint64_t e0 = (((uint64_t)(b0 + bl) * H + h) * S + i) * S;
    float q[DHP], p[SA_MAXS], pk[SA_MAXS], o[DHP];
    sa_ldvec<DHP>(s_qkv + (bl * S + i) * rs + h * DHP, q);
    sa_row<DHP>(q, s_qkv + (bl * S) * rs + (H + h) * DHP, rs, S, scale, a.p_drop, a.seed, call, e0, p, pk);
#pragma unroll
    for (int t = 0; t < DHP; ++t) o[t] = 0.0f;
#pragma unroll
    for (int j = 0; j < SA_MAXS; ++j)
      if (j < S) {
        float vv[DHP];
        sa_ldvec<DHP>(s_qkv + (bl * S + j) * rs + (2 * H + h) * DHP, vv);
#pragma unroll
        for (int t = 0; t < DHP; ++t) o[t] = __builtin_fmaf(pk[j], vv[t], o[t]);
      }
    uint16_t* out = a.ctx + ((long)(b0 + bl) * S + i) * d + h * dh;
#pragma unroll
    for (int t = 0; t < DHP; ++t)
      if (t < dh) out[t] = sq_f2bf(o[t]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.used_call) *a.used_call = call;
}

template <int DHP>
__global__ __launch_bounds__(SA_THREADS) void seq_attention_bwd_kernel(const SaArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t s_mem[];
  const int H = a.H, dh = a.dh, d = H * dh, S = a.S;
  const int b0 = blockIdx.x * a.NB, nb = min(a.NB, a.Bn - b0), rows = nb * S;
  const int rs = 3 * H * DHP, rso = H * DHP;
  uint16_t* s_qkv = s_mem;                                       // [NB*S][3][H][DHP]
  uint16_t* s_do = s_qkv + a.NB * S * rs;                        // [NB*S][H][DHP]
  float* s_ds = (float*)(s_do + a.NB * S * rso);                 // [NB][H][S][S]  scale * dS
  float* s_pk = s_ds + a.NB * H * S * S;                         // [NB][H][S][S]  keep-scaled probabilities
  sa_zero_pad<DHP>(s_qkv, a.NB * S * 4 * H, dh);                 // (s_do follows s_qkv: one run of head vectors)
  sa_stage_padded<DHP>(s_qkv, a.qkv + (long)b0 * S * 3 * d, rows, 3 * H, dh);
  sa_stage_padded<DHP>(s_do, a.dctx + (long)b0 * S * d, rows, H, dh);
  const unsigned long long call = a.p_drop > 0.0f ? *a.used_call : 0ull;
  __syncthreads();
  const int tid = threadIdx.x, i = tid % S, h = (tid / S) % H, bl = tid / (S * H);
  const float scale = rsqrtf((float)dh), ks = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  if (bl < nb) {
    const uint64_t e0 = (((uint64_t)(b0 + bl) * H + h) * S + i) * S;
    float q[DHP], go[DHP], p[SA_MAXS], pk[SA_MAXS], dp[SA_MAXS], ds[SA_MAXS];
    sa_ldvec<DHP>(s_qkv + (bl * S + i) * rs + h * DHP, q);
    sa_ldvec<DHP>(s_do + (bl * S + i) * rso + h * DHP, go);
    sa_row<DHP>(q, s_qkv + (bl * S) * rs + (H + h) * DHP, rs, S, scale, a.p_drop, a.seed, call, e0, p, pk);
#pragma unroll
    for (int j = 0; j < SA_MAXS; ++j) {
      dp[j] = 0.0f;
      if (j < S) {
        float vv[DHP];
        sa_ldvec<DHP>(s_qkv + (bl * S + j) * rs + (2 * H + h) * DHP, vv);
        dp[j] = sa_dot<DHP>(go, vv);
      }
    }
    // through the dropout (dP = keep / (1 - p) dP~) and the softmax (dS = P (dP - sum_k dP_k P_k))
    float D = 0.0f;
#pragma unroll
    for (int j = 0; j < SA_MAXS; ++j) {
      dp[j] = pk[j] != 0.0f ? dp[j] * ks : (a.p_drop > 0.0f ? 0.0f : dp[j]);
      D = __builtin_fmaf(dp[j], p[j], D);
    }
    float dq[DHP];
#pragma unroll
    for (int t = 0; t < DHP; ++t) dq[t] = 0.0f;
#pragma unroll
    for (int j = 0; j < SA_MAXS; ++j) {
      ds[j] = j < S ? p[j] * (dp[j] - D) * scale : 0.0f;
      if (j < S) {
        s_ds[((bl * H + h) * S + i) * S + j] = ds[j];
        s_pk[((bl * H + h) * S + i) * S + j] = pk[j];
        float kv[DHP];
        sa_ldvec<DHP>(s_qkv + (bl * S + j) * rs + (H + h) * DHP, kv);
#pragma unroll
        for (int t = 0; t < DHP; ++t) dq[t] = __builtin_fmaf(ds[j], kv[t], dq[t]);
      }
    }
    uint16_t* dqo = a.dqkv + ((long)(b0 + bl) * S + i) * 3 * d + h * dh;
#pragma unroll
    for (int t = 0; t < DHP; ++t)
      if (t < dh) dqo[t] = sq_f2bf(dq[t]);
  }
  __syncthreads();
  if (bl < nb) {
    // this thread's token as KEY / VALUE j = i: dk_j = sum_i dS_ij q_i, dv_j = sum_i P~_ij dO_i
    const int j = i;
    float dk[DHP], dv[DHP];
#pragma unroll
    for (int t = 0; t < DHP; ++t) { dk[t] = 0.0f; dv[t] = 0.0f; }
#pragma unroll
    for (int ii = 0; ii < SA_MAXS; ++ii)
      if (ii < S) {
        const float wds = s_ds[((bl * H + h) * S + ii) * S + j], wpk = s_pk[((bl * H + h) * S + ii) * S + j];
        float qv[DHP], gv[DHP];
        sa_ldvec<DHP>(s_qkv + (bl * S + ii) * rs + h * DHP, qv);
        sa_ldvec<DHP>(s_do + (bl * S + ii) * rso + h * DHP, gv);
#pragma unroll
        for (int t = 0; t < DHP; ++t) { dk[t] = __builtin_fmaf(wds, qv[t], dk[t]); dv[t] = __builtin_fmaf(wpk, gv[t], dv[t]); }
      }
    uint16_t* dko = a.dqkv + ((long)(b0 + bl) * S + j) * 3 * d + d + h * dh;
    uint16_t* dvo = dko + d;
#pragma unroll
    for (int t = 0; t < DHP; ++t)
      if (t < dh) { dko[t] = sq_f2bf(dk[t]); dvo[t] = sq_f2bf(dv[t]); }
  }
}

// ================================================================================================================================
// LayerNorm(residual + dropout(x)): one wave per row, a lane's columns lane, lane + 64, ... (d <= 256)
// ================================================================================================================================
constexpr int LN_MAXC = 4;           // columns per lane
constexpr int LN_THREADS = 256;      // 4 rows at a time
constexpr int LN_BWD_THREADS = 1024; // backward: 16 waves ...
constexpr int LN_ROWS_PER_WG = 32;   // ... x 2 rows: few workgroups add into gamma / beta's gradients (~40 ns per same-address atomic)

struct LnArgs {
  int T, d;
  const uint16_t* x;                 // [T, d] bf16: the sublayer's output
  const void* res; int res_bf16;     // [T, d] residual, float32 or bf16
  const float* gamma; const float* beta; float eps;
  float p_drop; unsigned long long seed; unsigned long long* counter; unsigned long long* used_call;
  float* y32; uint16_t* y16;         // [T, d] both
  float* z;                          // [T, d] float32: the normalisation's input, saved
  float* mean; float* rstd;          // [T]
  // backward
  const float* dy32; const uint16_t* dy16;     // either may be NULL
  uint16_t* dx; float* dres;
  float* dgamma; float* dbeta;       // ADDED to (float atomics): the caller zeroes them
};

__global__ __launch_bounds__(LN_THREADS) void dropout_add_ln_fwd_kernel(const LnArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, d = a.d;
  const unsigned long long call = a.p_drop > 0.0f ? *a.counter : 0ull;
  const float ks = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  float g[LN_MAXC], be[LN_MAXC];
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    g[k] = c < d ? a.gamma[c] : 0.0f;
    be[k] = c < d ? a.beta[c] : 0.0f;
  }
  for (int r = blockIdx.x * 4 + wave; r < a.T; r += gridDim.x * 4) {
    float z[LN_MAXC], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) {
      const int c = lane + 64 * k;
      z[k] = 0.0f;
      if (c < d) {
        const long e = (long)r * d + c;
        float xv = sq_bf2f(a.x[e]);
        if (a.p_drop > 0.0f) xv = sq_uniform(a.seed, call, (uint64_t)e) >= a.p_drop ? xv * ks : 0.0f;
        const float rv = a.res_bf16 ? sq_bf2f(((const uint16_t*)a.res)[e]) : ((const float*)a.res)[e];
        z[k] = rv + xv;
        sum += z[k];
      }
    }
    const float mean = sq_wsum(sum) / (float)d;
    float sq = 0.0f;
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) { const float dv = z[k] - mean; sq += (lane + 64 * k < d) ? dv * dv : 0.0f; }
    const float rstd = rsqrtf(sq_wsum(sq) / (float)d + a.eps);
    if (lane == 0) { a.mean[r] = mean; a.rstd[r] = rstd; }
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < d) {
        const long e = (long)r * d + c;
        const float y = __builtin_fmaf((z[k] - mean) * rstd, g[k], be[k]);
        a.z[e] = z[k];
        a.y32[e] = y;
        a.y16[e] = sq_f2bf(y);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.used_call) *a.used_call = call;
}

__global__ __launch_bounds__(LN_BWD_THREADS) void dropout_add_ln_bwd_kernel(const LnArgs a) {
  constexpr int NW = LN_BWD_THREADS / 64;
  __shared__ float s_g[NW][LN_MAXC * 64], s_b[NW][LN_MAXC * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, d = a.d;
  const unsigned long long call = a.p_drop > 0.0f ? *a.used_call : 0ull;
  const float ks = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  float g[LN_MAXC], pg[LN_MAXC], pb[LN_MAXC];
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) { g[k] = (lane + 64 * k < d) ? a.gamma[lane + 64 * k] : 0.0f; pg[k] = 0.0f; pb[k] = 0.0f; }
  const int r0 = blockIdx.x * LN_ROWS_PER_WG;
  // this wave's rows r0 + wave, + NW, ...: the next row's operands are requested before the current row's reductions
  auto load_row = [&](int r, float (&gv)[LN_MAXC], float (&zv)[LN_MAXC], float& mean, float& rstd) {
    mean = 0.0f; rstd = 0.0f;
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) { gv[k] = 0.0f; zv[k] = 0.0f; }
    if (r < a.T) {
      mean = a.mean[r]; rstd = a.rstd[r];
#pragma unroll
      for (int k = 0; k < LN_MAXC; ++k) {
        const int c = lane + 64 * k;
        if (c < d) {
          const long e = (long)r * d + c;
          gv[k] = a.dy32 ? a.dy32[e] : 0.0f;
          if (a.dy16) gv[k] += sq_bf2f(a.dy16[e]);
          zv[k] = a.z[e];
        }
      }
    }
  };
  float gv[LN_MAXC], zv[LN_MAXC], mean, rstd;
  load_row(r0 + wave, gv, zv, mean, rstd);
  for (int rr = wave; rr < LN_ROWS_PER_WG; rr += NW) {
    const int r = r0 + rr;
    if (r >= a.T) break;                                   // wave-uniform
    float gn[LN_MAXC], zn[LN_MAXC], mean_n, rstd_n;
    load_row(r + NW < r0 + LN_ROWS_PER_WG ? r + NW : a.T, gn, zn, mean_n, rstd_n);
    float gy[LN_MAXC], xh[LN_MAXC], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) {
      const bool live = lane + 64 * k < d;
      xh[k] = live ? (zv[k] - mean) * rstd : 0.0f;
      pg[k] = __builtin_fmaf(gv[k], xh[k], pg[k]);
      pb[k] += gv[k];
      gy[k] = gv[k] * g[k];
      s1 += gy[k];
      s2 = __builtin_fmaf(gy[k], xh[k], s2);
    }
    s1 = sq_wsum(s1) / (float)d; s2 = sq_wsum(s2) / (float)d;
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < d) {
        const long e = (long)r * d + c;
        const float dz = rstd * (gy[k] - s1 - xh[k] * s2);
        a.dres[e] = dz;
        const bool keep = !(a.p_drop > 0.0f) || sq_uniform(a.seed, call, (uint64_t)e) >= a.p_drop;
        a.dx[e] = sq_f2bf(keep ? dz * ks : 0.0f);
      }
    }
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) { gv[k] = gn[k]; zv[k] = zn[k]; }
    mean = mean_n; rstd = rstd_n;
  }
  // column sums over this workgroup's rows, then ONE hardware float atomic (unsafeAtomicAdd: global_atomic_add_f32, not the
  // compare-and-swap loop plain atomicAdd compiles to) per column and workgroup into the (zeroed) gamma / beta gradients.  (A
  // two-stage reduction of per-workgroup partial rows by one workgroup cost 40 us per launch; the order of these additions is not
  // fixed: ~1e-7 relative run to run.)
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) { s_g[wave][lane + 64 * k] = pg[k]; s_b[wave][lane + 64 * k] = pb[k]; }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += LN_BWD_THREADS) {
    float tg = 0.0f, tb = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { tg += s_g[w][c]; tb += s_b[w][c]; }
    unsafeAtomicAdd(&a.dgamma[c], tg);
    unsafeAtomicAdd(&a.dbeta[c], tb);
  }
}

// ================================================================================================================================
// dropout(LeakyReLU_slope(x)) elementwise on bf16 (slope 0: ReLU)
// ================================================================================================================================
__global__ __launch_bounds__(256) void act_dropout_fwd_kernel(long n, const uint16_t* __restrict__ x, uint16_t* __restrict__ y, float slope,
                                                               float p_drop, unsigned long long seed, unsigned long long* counter,
                                                               unsigned long long* used_call) {
  const unsigned long long call = p_drop > 0.0f ? *counter : 0ull;
  const float ks = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    float v = sq_bf2f(x[e]);
    v = v > 0.0f ? v : v * slope;
    if (p_drop > 0.0f) v = sq_uniform(seed, call, (uint64_t)e) >= p_drop ? v * ks : 0.0f;
    y[e] = sq_f2bf(v);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && used_call) *used_call = call;
}
__global__ __launch_bounds__(256) void act_dropout_bwd_kernel(long n, const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                               uint16_t* __restrict__ dx, float slope, float p_drop, unsigned long long seed,
                                                               const unsigned long long* used_call) {
  const unsigned long long call = p_drop > 0.0f ? *used_call : 0ull;
  const float ks = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    float g = sq_bf2f(dy[e]);
    if (p_drop > 0.0f) g = sq_uniform(seed, call, (uint64_t)e) >= p_drop ? g * ks : 0.0f;
    g = sq_bf2f(x[e]) > 0.0f ? g : g * slope;
    dx[e] = sq_f2bf(g);
  }
}

void set_last_error(const char* msg);
int deterministic_mode();              // frame_solve.hip: library option "deterministic" (ops_amd_set_option)

}  // namespace opsamd

using namespace opsamd;

static int sq_check(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  (void)what;
  return OPS_AMD_OK;
}

// samples per workgroup: as many as 512 threads (one per sample x head x token) and 64 KB of LDS hold
static int sa_samples_per_wg(int S, int H, size_t lds_per_sample) {
  int nb = SA_THREADS / (S * H);
  const int cap = (int)((64 * 1024) / lds_per_sample);
  if (nb > cap) nb = cap;
  return nb < 1 ? 0 : (nb > 8 ? 8 : nb);
}

extern "C" int ops_seq_attention_fwd(int Bn, int S, int H, int dh, const void* qkv, void* ctx, float p_drop, unsigned long long seed,
                                     unsigned long long* counter, unsigned long long* used_call, void* stream) {
  if (Bn < 1 || S < 1 || H < 1 || dh < 1 || !qkv || !ctx || p_drop < 0.0f || p_drop >= 1.0f || (p_drop > 0.0f && (!counter || !used_call)))
    return OPS_AMD_ERR_INVALID_ARG;
  const int d = H * dh, DHP = dh <= 16 ? 16 : 32;
  if (S > SA_MAXS || S * H > SA_THREADS || dh > 32) return OPS_AMD_ERR_UNSUPPORTED;
  const int NB = sa_samples_per_wg(S, H, (size_t)S * 3 * H * DHP * 2);
  if (NB < 1 || (3 * d) % 8 || ((uintptr_t)qkv & 15)) return OPS_AMD_ERR_UNSUPPORTED;
  const size_t lds = (size_t)NB * S * 3 * H * DHP * 2;
  const SaArgs a{Bn, S, H, dh, NB, (const uint16_t*)qkv, nullptr, (uint16_t*)ctx, nullptr, p_drop, seed, counter, used_call};
  const dim3 grid((unsigned)((Bn + NB - 1) / NB));
  if (DHP == 16) hipLaunchKernelGGL(seq_attention_fwd_kernel<16>, grid, dim3(SA_THREADS), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(seq_attention_fwd_kernel<32>, grid, dim3(SA_THREADS), lds, (hipStream_t)stream, a);
  return sq_check("seq_attention_fwd_kernel");
}

extern "C" int ops_seq_attention_bwd(int Bn, int S, int H, int dh, const void* qkv, const void* dctx, void* dqkv, float p_drop,
                                     unsigned long long seed, unsigned long long* used_call, void* stream) {
  if (Bn < 1 || S < 1 || H < 1 || dh < 1 || !qkv || !dctx || !dqkv || p_drop < 0.0f || p_drop >= 1.0f || (p_drop > 0.0f && !used_call))
    return OPS_AMD_ERR_INVALID_ARG;
  const int d = H * dh, DHP = dh <= 16 ? 16 : 32;
  if (S > SA_MAXS || S * H > SA_THREADS || dh > 32) return OPS_AMD_ERR_UNSUPPORTED;
  const size_t per_sample = (size_t)S * 4 * H * DHP * 2 + (size_t)2 * H * S * S * 4;
  const int NB = sa_samples_per_wg(S, H, per_sample);
  if (NB < 1 || (3 * d) % 8 || d % 8 || (((uintptr_t)qkv | (uintptr_t)dctx) & 15)) return OPS_AMD_ERR_UNSUPPORTED;
  const size_t lds = (size_t)NB * per_sample;
  const SaArgs a{Bn, S, H, dh, NB, (const uint16_t*)qkv, (const uint16_t*)dctx, nullptr, (uint16_t*)dqkv, p_drop, seed, nullptr, used_call};
  const dim3 grid((unsigned)((Bn + NB - 1) / NB));
  if (DHP == 16) hipLaunchKernelGGL(seq_attention_bwd_kernel<16>, grid, dim3(SA_THREADS), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(seq_attention_bwd_kernel<32>, grid, dim3(SA_THREADS), lds, (hipStream_t)stream, a);
  return sq_check("seq_attention_bwd_kernel");
}

extern "C" int ops_dropout_add_layernorm_fwd(int T, int d, const void* x, const void* res, int res_is_bf16, const float* gamma, const float* beta,
                                             float eps, float p_drop, unsigned long long seed, unsigned long long* counter,
                                             unsigned long long* used_call, float* y32, void* y16, float* z, float* mean, float* rstd,
                                             void* stream) {
  if (T < 1 || d < 1 || !x || !res || !gamma || !beta || !y32 || !y16 || !z || !mean || !rstd || p_drop < 0.0f || p_drop >= 1.0f ||
      (p_drop > 0.0f && (!counter || !used_call)))
    return OPS_AMD_ERR_INVALID_ARG;
  if (d > 64 * LN_MAXC) return OPS_AMD_ERR_UNSUPPORTED;
  LnArgs a{};
  a.T = T; a.d = d; a.x = (const uint16_t*)x; a.res = res; a.res_bf16 = res_is_bf16; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.p_drop = p_drop; a.seed = seed; a.counter = counter; a.used_call = used_call; a.y32 = y32; a.y16 = (uint16_t*)y16; a.z = z;
  a.mean = mean; a.rstd = rstd;
  int grid = (T + 3) / 4;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(dropout_add_ln_fwd_kernel, dim3((unsigned)grid), dim3(LN_THREADS), 0, (hipStream_t)stream, a);
  return sq_check("dropout_add_ln_fwd_kernel");
}

extern "C" int ops_dropout_add_layernorm_bwd(int T, int d, const float* dy32, const void* dy16, const float* z, const float* mean, const float* rstd,
                                             const float* gamma, float p_drop, unsigned long long seed, unsigned long long* used_call, void* dx,
                                             float* dres, float* dgamma, float* dbeta, void* stream) {
  if (T < 1 || d < 1 || (!dy32 && !dy16) || !z || !mean || !rstd || !gamma || !dx || !dres || !dgamma || !dbeta ||
      p_drop < 0.0f || p_drop >= 1.0f || (p_drop > 0.0f && !used_call))
    return OPS_AMD_ERR_INVALID_ARG;
  if (d > 64 * LN_MAXC) return OPS_AMD_ERR_UNSUPPORTED;
  LnArgs a{};
  a.T = T; a.d = d; a.gamma = gamma; a.p_drop = p_drop; a.seed = seed; a.used_call = used_call; a.z = (float*)z; a.mean = (float*)mean;
  a.rstd = (float*)rstd; a.dy32 = dy32; a.dy16 = (const uint16_t*)dy16; a.dx = (uint16_t*)dx; a.dres = dres; a.dgamma = dgamma; a.dbeta = dbeta;
  hipLaunchKernelGGL(dropout_add_ln_bwd_kernel, dim3((unsigned)((T + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG)), dim3(LN_BWD_THREADS), 0,
                     (hipStream_t)stream, a);
  return sq_check("dropout_add_ln_bwd_kernel");
}

extern "C" int ops_act_dropout_fwd(long n, const void* x, void* y, float slope, float p_drop, unsigned long long seed, unsigned long long* counter,
                                   unsigned long long* used_call, void* stream) {
  if (n < 1 || !x || !y || p_drop < 0.0f || p_drop >= 1.0f || (p_drop > 0.0f && (!counter || !used_call))) return OPS_AMD_ERR_INVALID_ARG;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(act_dropout_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, n, (const uint16_t*)x, (uint16_t*)y, slope,
                     p_drop, seed, counter, used_call);
  return sq_check("act_dropout_fwd_kernel");
}

extern "C" int ops_act_dropout_bwd(long n, const void* x, const void* dy, void* dx, float slope, float p_drop, unsigned long long seed,
                                   unsigned long long* used_call, void* stream) {
  if (n < 1 || !x || !dy || !dx || p_drop < 0.0f || p_drop >= 1.0f || (p_drop > 0.0f && !used_call)) return OPS_AMD_ERR_INVALID_ARG;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(act_dropout_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, n, (const uint16_t*)x, (const uint16_t*)dy,
                     (uint16_t*)dx, slope, p_drop, seed, (const unsigned long long*)used_call);
  return sq_check("act_dropout_bwd_kernel");
}

// ================================================================================================================================
// weight gradient of a Linear over MANY rows: dW [N, K] += dY^T X with dY [T, N], X [T, K] bf16 row-major, T in the thousands
// (every token of the batch), N, K a few hundred.  The library runs this as a handful of 64 x 64 tiles that each walk all T rows
// (25 us per product, ten products per Transformer-Diffusion step); here the rows are split over the grid as well: a workgroup owns
// a 64 x 64 tile of dW and 256 rows, transposes 32-row slabs of both operands through LDS into MFMA fragments (both operands are
// contiguous along the OUTPUT index in memory, the MFMA wants them contiguous along the reduction index), and adds its partial
// tile to the float32 gradient with hardware float atomics (the caller's flat gradient buffer is zeroed every step).  The bias
// gradient -- the column sums of dY -- comes out of the same pass (first column of tiles).  Same-address float atomics cost
// ~40 ns each (measured: 448 adders per address 19 us, 224: 9 us): 14 row blocks per address here.
// ================================================================================================================================
namespace opsamd {

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
constexpr int WG_SLAB = 32;       // rows per LDS slab = one MFMA reduction step
#define OPS_WG_MAX_UNITS 128
#define OPS_WG_MAX_UNITS_PER_XCD 24

// one 32-row slab of an operand: row lr, 8 columns from c0 + lc -- one 16-byte load when the matrix allows it
__device__ __forceinline__ uint4 wg_load8(const uint16_t* __restrict__ M, int T, int C, int ld, int t, int c, bool vec) {
  uint4 r = uint4{0u, 0u, 0u, 0u};
#if defined(OPS_WG_ABLATE) && OPS_WG_ABLATE == 3      /* no global loads */
  r.x = (unsigned)t; return r;
#endif
  if (t >= T || c >= C) return r;
  if (vec && c + 8 <= C) return *(const uint4*)(M + (long)t * ld + c);
  uint16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = c + j < C ? M[(long)t * ld + c + j] : (uint16_t)0;
  return *(const uint4*)v;
}

// r04 form.  What the r03 launch (34-37 us for the twelve products of a Transformer-Diffusion step) spent, by phase ablation
// (profiles/r04_notes.md 5): ~12 us float atomics (5 M of them: they execute at the memory side at ~1.3 TB/s of added bytes whatever
// the XCD), ~12 us the operands' loads, ~12 us everything else (a barrier pair and sixteen 2-byte LDS scatter writes per thread and
// slab, 1 218 workgroups).  Now:
//   * a workgroup still owns a 64 x 64 tile of dW, but each of its four waves walks ITS OWN rows (rw rows, a multiple of 32) and keeps
//     the whole tile in registers (16 accumulators); the four partial tiles meet in LDS once, at the end, and the workgroup adds one
//     tile for 4 rw rows: a quarter of the atomics at the same number of waves in flight;
//   * no workgroup barrier inside the row loop: a wave's slab lives in its own LDS region (LDS operations of one wave execute in order);
//   * the slab sits ROW-major in LDS as it arrives (four 16-byte writes per lane and operand) and the fragments -- 8 consecutive t of
//     one column -- are taken with gfx950's transposing LDS read (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block,
//     delivered column-major), two per fragment.  Row pitch 96 elements with rows 8 .. 15 (mod 16) skewed by 16: the transposed reads of
//     a 32-lane half then touch 64 distinct banks;
//   * eight 16-byte loads per lane in flight (the next slab of both operands) while a slab is multiplied.
constexpr int WG_P = 96;
constexpr int WG_SKEW = 16;
__device__ __forceinline__ int wg_lds_off(int r, int c) { return r * WG_P + c + (((r >> 3) & 1) ? WG_SKEW : 0); }
typedef short wg_v4i16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 wg_tr_read(const uint16_t* p) {      // EXEC must be all ones (every lane supplies an address)
  const wg_v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_v4i16*)p);
  return __builtin_bit_cast(uint2, r);
}
// rows per WAVE for a product over T rows: workgroups of ~1 024 rows, the rows shared evenly by the splits (T = 3 584: 4 splits of 896)
#ifndef OPS_WG_SPLIT_ROWS
#define OPS_WG_SPLIT_ROWS 1024
#endif
// det (library option "deterministic"): ONE split -- a tile of dW has exactly one contributing workgroup, so the float atomics that add it
// into the (zeroed) gradient cannot reorder anything; slower (a workgroup walks all T rows), bit-reproducible
__host__ __device__ inline int wg_rows_per_wave(int T, int det = 0) {
  const int nsplit = det ? 1 : (T + OPS_WG_SPLIT_ROWS - 1) / OPS_WG_SPLIT_ROWS;
  const int per_wg = (T + nsplit - 1) / nsplit;
  return (((per_wg + 3) / 4) + WG_SLAB - 1) / WG_SLAB * WG_SLAB;
}
__host__ __device__ inline int wg_row_splits(int T, int det = 0) { const int rw = wg_rows_per_wave(T, det); return (T + 4 * rw - 1) / (4 * rw); }

__device__ __forceinline__ void wgrad_tn_body(int T, int N, int K, const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X,
                                              float* __restrict__ dW, float* __restrict__ dbias, int bx, int by, int bz, int ldy = 0, int ldx = 0, int det = 0) {
  ldy = ldy > 0 ? ldy : N; ldx = ldx > 0 ? ldx : K;                    // row strides (elements) of dY / X: a strided row view needs no copy
  // [wave][operand][32 rows x pitch]: 48 KB; reused by the cross-wave reduction: [owner wave][source rank 0..2][tile 0..3][lane][4 floats]
  __shared__ __attribute__((aligned(16))) uint16_t s_slab[4 * 2 * WG_SLAB * WG_P];
  __shared__ float s_cs[4][64];                                        // bias job: per-wave column sums
  static_assert(4 * 2 * WG_SLAB * WG_P * 2 == 4 * 3 * 4 * 64 * 16, "the reduction reuses the slab area exactly");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rw = wg_rows_per_wave(T, det);
  const int n0 = bx * 64, k0 = by * 64, t0 = (bz * 4 + wave) * rw, t1 = min(t0 + rw, T);      // this WAVE's rows (possibly none)
  uint16_t* const sa = s_slab + (size_t)wave * 2 * WG_SLAB * WG_P;
  uint16_t* const sb = sa + WG_SLAB * WG_P;
  const bool va = (ldy & 7) == 0 && ((uintptr_t)dY & 15) == 0, vb = (ldx & 7) == 0 && ((uintptr_t)X & 15) == 0;
  wg_f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = wg_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const int lr = lane >> 3, lc = (lane & 7) * 8;                       // slab loader: rows lr + 8 i (i < 4), 8 columns from lc
  int woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) woff[i] = wg_lds_off(lr + 8 * i, lc);
  // fragment reads: lane (g, q, p) = (lane >> 4, (lane >> 2) & 3, lane & 3) supplies row 8 g + q (+ 4), columns 4 p .. 4 p + 3 of the
  // 16-column block; lane i of the group receives column i, rows 8 g .. 8 g + 7 = the MFMA operand of lane (i, k group g)
  const int fr_off = wg_lds_off(8 * (lane >> 4) + ((lane >> 2) & 3), 4 * (lane & 3));
  const bool bias_job = dbias != nullptr && by == 0;                   // the first column of tiles also sums dY's columns
  float csum[8];                                                       // this loader's 8 columns, summed over its rows
#pragma unroll
  for (int j = 0; j < 8; ++j) csum[j] = 0.0f;
  uint4 ra[4], rb[4];
  auto load = [&](int ts) {                                            // (rows >= t1: zeros, no access)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = wg_load8(dY, t1, N, ldy, ts + lr + 8 * i, n0 + lc, va);
      rb[i] = wg_load8(X, t1, K, ldx, ts + lr + 8 * i, k0 + lc, vb);
    }
  };
  auto park = [&]() {                                                  // registers -> this wave's slab (+ the bias job's tally)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)&sa[woff[i]] = ra[i];
      *(uint4*)&sb[woff[i]] = rb[i];
      if (bias_job) {
        const uint16_t* pv = (const uint16_t*)&ra[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[j] += sq_bf2f(pv[j]);
      }
    }
  };
  if (t0 < t1) {                                                       // wave-uniform
    load(t0);
    for (int ts = t0; ts < t1; ts += WG_SLAB) {
      park();
      if (ts + WG_SLAB < t1) load(ts + WG_SLAB);                       // the next slab's loads fly while this one multiplies
      wg_bf16x8 fa[4], fb[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const uint16_t* qa = sa + fr_off + 16 * h;
        const uint16_t* qb = sb + fr_off + 16 * h;
        const uint2 a0 = wg_tr_read(qa), a1 = wg_tr_read(qa + 4 * WG_P), b0 = wg_tr_read(qb), b1 = wg_tr_read(qb + 4 * WG_P);
        fa[h] = __builtin_bit_cast(wg_bf16x8, uint4{a0.x, a0.y, a1.x, a1.y});
        fb[h] = __builtin_bit_cast(wg_bf16x8, uint4{b0.x, b0.y, b1.x, b1.y});
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
  // ---- the four partial tiles meet: wave w ends up with row block w (rows 16 w .. 16 w + 15 of the tile) summed over the waves ----
  __syncthreads();                                                     // every wave is done with its slab region
  float4* const red = (float4*)s_slab;                                 // [owner][source rank][tile j][lane]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i == wave) continue;                                           // (wave-uniform)
    const int rank = wave < i ? wave : wave - 1;                       // this wave's place among the three sources of owner i
#pragma unroll
    for (int j = 0; j < 4; ++j) red[((i * 3 + rank) * 4 + j) * 64 + lane] = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
  }
  __syncthreads();
  wg_f32x4 mine[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // (a dynamic first index into acc[][] would spill the accumulators: select by comparison)
    wg_f32x4 v = wave == 0 ? acc[0][j] : wave == 1 ? acc[1][j] : wave == 2 ? acc[2][j] : acc[3][j];
#pragma unroll
    for (int r = 0; r < 3; ++r) {                                      // fixed order: the sum does not depend on timing
      const float4 u = red[((wave * 3 + r) * 4 + j) * 64 + lane];
      v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w;
    }
    mine[j] = v;
  }
  // C layout: column (k) = lane & 15, rows (n) = 4 (lane >> 4) + e
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + 16 * j + (lane & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + 16 * wave + 4 * (lane >> 4) + e;
#if defined(OPS_WG_ABLATE) && OPS_WG_ABLATE == 1      /* phase ablation builds (scripts/wgrad_ab.sh): plain stores instead of atomics */
      if (n < N && k < K) dW[(long)n * K + k] = mine[j][e];
#elif defined(OPS_WG_ABLATE) && OPS_WG_ABLATE == 2    /* no epilogue at all (one lane keeps the accumulators alive) */
      if (mine[j][e] == 1.2345e30f) dW[0] = 1.0f;
#else
      if (n < N && k < K) unsafeAtomicAdd(&dW[(long)n * K + k], mine[j][e]);
#endif
    }
  }
  if (bias_job) {
    // the 8 loaders of a column group inside a wave (lanes with equal lane & 7), then the four waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = csum[j];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if ((lane >> 3) == 0) s_cs[wave][lc + j] = v;
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < N) unsafeAtomicAdd(&dbias[n0 + tid], s_cs[0][tid] + s_cs[1][tid] + s_cs[2][tid] + s_cs[3][tid]);
  }
}

__global__ __launch_bounds__(256) void wgrad_tn_kernel(int T, int N, int K, const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X,
                                                        float* __restrict__ dW, float* __restrict__ dbias, int det) {
  wgrad_tn_body(T, N, K, dY, X, dW, dbias, blockIdx.x, blockIdx.y, blockIdx.z, 0, 0, det);
}

// several products in one launch (all weight gradients of a backward pass, once every dY exists): workgroup -> (problem, tile, rows)
struct WgGroup {
  int nprob, det;                      // det: library option "deterministic" (one row split per product, one workgroup per column-sum strip)
  int wg0[OPS_WGRAD_MAX_GROUP + 1];
  ops_wgrad_problem p[OPS_WGRAD_MAX_GROUP];
  // XCD-aware placement (r04): a UNIT = the tiles of one product over one row split (they read the same rows of dY and X) or one
  // column-sum job.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the XCD's share, MI355X_MICROARCH.md:
  // observed, for speed only), so unit u's workgroups get the ids  x + 8 q  of ONE residue x: the re-reads of its rows by its other
  // tiles then come out of that XCD's L2 instead of the Infinity Cache.  nunit = 0: the flat order above (too many units).
  int nunit, qmax;
  unsigned char unit_prob[OPS_WG_MAX_UNITS], unit_split[OPS_WG_MAX_UNITS];
  unsigned char xcnt[8], xunit[8][OPS_WG_MAX_UNITS_PER_XCD];
  unsigned short xoff[8][OPS_WG_MAX_UNITS_PER_XCD + 1];
};
// K = 0 job: out[c] += sum over the T rows of the float32 matrix M [T, N]; one workgroup per 64 columns x CS_ROWS rows: every thread has
// its 8 loads in flight at once (one workgroup per 64 columns walking all rows: 56 dependent trips, the launch's long pole at 41 us)
constexpr int CS_ROWS = 32;
__device__ __forceinline__ void colsum_body(int T, int N, const float* __restrict__ M, int ld, float* __restrict__ out, int id, int det) {
  __shared__ float s_p[4][64];
  const int nb = (N + 63) / 64, bx = id % nb, by = id / nb;
  const int tid = threadIdx.x, c = bx * 64 + (tid & 63), stripe = tid >> 6, r0 = by * CS_ROWS;
  float acc = 0.0f;
  if (det) {                           // one workgroup per 64 columns walks ALL rows in a fixed order (one contribution per column: nothing to reorder)
    for (int rb = 0; rb < T; rb += CS_ROWS) {
      float v[CS_ROWS / 4];
#pragma unroll
      for (int k = 0; k < CS_ROWS / 4; ++k) { const int r = rb + stripe + 4 * k; v[k] = (c < N && r < T) ? M[(long)r * ld + c] : 0.0f; }
#pragma unroll
      for (int k = 0; k < CS_ROWS / 4; ++k) acc += v[k];
    }
  } else {
    float v[CS_ROWS / 4];
#pragma unroll
    for (int k = 0; k < CS_ROWS / 4; ++k) { const int r = r0 + stripe + 4 * k; v[k] = (c < N && r < T) ? M[(long)r * ld + c] : 0.0f; }
#pragma unroll
    for (int k = 0; k < CS_ROWS / 4; ++k) acc += v[k];
  }
  s_p[stripe][tid & 63] = acc;
  __syncthreads();
  if (tid < 64 && c < N) unsafeAtomicAdd(&out[c], (s_p[0][tid] + s_p[1][tid]) + (s_p[2][tid] + s_p[3][tid]));
}

__global__ __launch_bounds__(256) void wgrad_tn_group_kernel(const WgGroup g) {
  int pi = 0, id, bz = -1;
  if (g.nunit > 0) {                                                   // unit placement: blockIdx.x = x + 8 q
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3, cnt = g.xcnt[x];
    int j = 0;
    while (j < cnt && q >= (int)g.xoff[x][j + 1]) ++j;
    if (j >= cnt) return;                                              // padding workgroup of a lighter XCD share (workgroup-uniform)
    const int u = g.xunit[x][j];
    pi = g.unit_prob[u];
    bz = g.unit_split[u];
    id = q - (int)g.xoff[x][j];                                        // tile within the unit
  } else {
    while (pi + 1 < g.nprob && (int)blockIdx.x >= g.wg0[pi + 1]) ++pi;
    id = (int)blockIdx.x - g.wg0[pi];
  }
  const ops_wgrad_problem pr = g.p[pi];
  if (pr.K == 0) {                                                     // workgroup-uniform
    colsum_body(pr.T, pr.N, (const float*)pr.dY, pr.ldy > 0 ? pr.ldy : pr.N, pr.dW, id, g.det);
    return;
  }
  const int tn = (pr.N + 63) / 64, tk = (pr.K + 63) / 64;
  if (bz < 0) { bz = id / (tn * tk); id -= bz * tn * tk; }
  wgrad_tn_body(pr.T, pr.N, pr.K, (const uint16_t*)pr.dY, (const uint16_t*)pr.X, pr.dW, pr.dbias, id % tn, id / tn, bz, pr.ldy, pr.ldx, g.det);
}

}  // namespace opsamd

extern "C" int ops_linear_wgrad_accumulate(int T, int N, int K, const void* dY, const void* X, float* dW, float* dbias, void* stream) {
  if (T < 1 || N < 1 || K < 1 || !dY || !X || !dW) return OPS_AMD_ERR_INVALID_ARG;
  const int det = opsamd::deterministic_mode();
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64), (unsigned)opsamd::wg_row_splits(T, det));
  hipLaunchKernelGGL(opsamd::wgrad_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, T, N, K, (const uint16_t*)dY, (const uint16_t*)X, dW, dbias, det);
  return sq_check("wgrad_tn_kernel");
}

extern "C" int ops_linear_wgrad_accumulate_group(int nprob, const ops_wgrad_problem* problems, void* stream) {
  if (nprob < 1 || nprob > OPS_WGRAD_MAX_GROUP || !problems) return OPS_AMD_ERR_INVALID_ARG;
  opsamd::WgGroup g;
  g.nprob = nprob;
  g.det = opsamd::deterministic_mode();
  int tot = 0;
  for (int i = 0; i < nprob; ++i) {
    const ops_wgrad_problem& p = problems[i];
    if (p.K == 0) {                 // column-sum job
      if (p.T < 1 || p.N < 1 || !p.dY || !p.dW || (p.ldy && p.ldy < p.N)) return OPS_AMD_ERR_INVALID_ARG;
      g.p[i] = p;
      g.wg0[i] = tot;
      tot += ((p.N + 63) / 64) * (g.det ? 1 : (p.T + opsamd::CS_ROWS - 1) / opsamd::CS_ROWS);
      continue;
    }
    if (p.T < 1 || p.N < 1 || p.K < 1 || !p.dY || !p.X || !p.dW || (p.ldy && p.ldy < p.N) || (p.ldx && p.ldx < p.K)) return OPS_AMD_ERR_INVALID_ARG;
    g.p[i] = p;
    g.wg0[i] = tot;
    tot += ((p.N + 63) / 64) * ((p.K + 63) / 64) * opsamd::wg_row_splits(p.T, g.det);
  }
  g.wg0[nprob] = tot;
  // units, heaviest first onto the lightest of the eight shares
  g.nunit = 0; g.qmax = 0;
  int nu = 0, wgs[OPS_WG_MAX_UNITS];
  bool fits = true;
  for (int i = 0; i < nprob && fits; ++i) {
    const ops_wgrad_problem& p = problems[i];
    const int splits = p.K == 0 ? 1 : opsamd::wg_row_splits(p.T, g.det);
    const int per = p.K == 0 ? g.wg0[i + 1] - g.wg0[i] : ((p.N + 63) / 64) * ((p.K + 63) / 64);
    for (int z = 0; z < splits; ++z) {
      if (nu >= OPS_WG_MAX_UNITS || z > 255 || per > 60000) { fits = false; break; }
      g.unit_prob[nu] = (unsigned char)i; g.unit_split[nu] = (unsigned char)z; wgs[nu] = per; ++nu;
    }
  }
  if (fits) {
    int order[OPS_WG_MAX_UNITS], load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int u = 0; u < nu; ++u) order[u] = u;
    for (int a = 1; a < nu; ++a) {                                     // insertion sort, descending workgroup count (stable)
      const int u = order[a]; int b = a;
      while (b > 0 && wgs[order[b - 1]] < wgs[u]) { order[b] = order[b - 1]; --b; }
      order[b] = u;
    }
    for (int x = 0; x < 8; ++x) { g.xcnt[x] = 0; g.xoff[x][0] = 0; }
    for (int a = 0; a < nu && fits; ++a) {
      const int u = order[a];
      int x = 0;
      for (int y = 1; y < 8; ++y) if (load[y] < load[x]) x = y;
      if (g.xcnt[x] >= OPS_WG_MAX_UNITS_PER_XCD || load[x] + wgs[u] > 65535) { fits = false; break; }
      g.xunit[x][g.xcnt[x]] = (unsigned char)u;
      load[x] += wgs[u];
      g.xoff[x][++g.xcnt[x]] = (unsigned short)load[x];
    }
    if (fits) {
      for (int x = 0; x < 8; ++x) g.qmax = load[x] > g.qmax ? load[x] : g.qmax;
      g.nunit = nu;
    }
  }
  const unsigned grid = g.nunit > 0 ? 8u * (unsigned)g.qmax : (unsigned)tot;
  hipLaunchKernelGGL(opsamd::wgrad_tn_group_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g);
  return sq_check("wgrad_tn_group_kernel");
}

// ================================================================================================================================
// diffusion front end of the Transformer-Diffusion surrogate (TransformerDiffusionModule_MultiCase.py:443-478, :563-567): the
// arithmetic around its two-layer MLP, the [CLS] token and the positional encoding -- a dozen framework nodes forward -- as two launches
//   noise   : x_noisy = sqrt(acp[t]) x + sqrt(1 - acp[t]) eps          (t, eps drawn by the caller: the framework's generators)
//   combine : z[b, 0] = cls + pe[0];  z[b, 1 + n] = (x_noisy - sb mlp(x_noisy)) / sa + pe[1 + n]
// ================================================================================================================================
namespace opsamd {

__global__ __launch_bounds__(256) void diffusion_noise_kernel(long rows, int d, const float* __restrict__ x, const long long* __restrict__ t,
                                                               const float* __restrict__ eps, const float* __restrict__ acp,
                                                               float* __restrict__ xn32, uint16_t* __restrict__ xn16, float* __restrict__ sa,
                                                               float* __restrict__ sb) {
  const long n = rows * d;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const long r = e / d;
    const float a = acp[t[r]], s_a = sqrtf(a), s_b = sqrtf(1.0f - a);
    const float v = s_a * x[e] + s_b * eps[e];
    xn32[e] = v;
    xn16[e] = sq_f2bf(v);
    if (e - r * d == 0) { sa[r] = s_a; sb[r] = s_b; }
  }
}

// The same with the step indices and the noise DRAWN HERE (counter-based: csrc/dropout_stream.hpp keys of (seed, call), element index):
// t[r] = floor(u T), eps = Box-Muller of two uniforms.  The framework generators cost the captured step four kernel nodes (randint,
// randn, and the two fills of the graph-safe generator's seed / offset tensors before every replay), ~20 us; the reference draws from
// an unseeded generator (TFD:452-456), so only the distributions are reproduced.  t_out / eps_out: optional copies of the draws.
__global__ __launch_bounds__(256) void diffusion_noise_draw_kernel(long rows, int d, int T, const float* __restrict__ x, const float* __restrict__ acp,
                                                                    unsigned long long seed, const unsigned long long* __restrict__ counter,
                                                                    float* __restrict__ xn32, uint16_t* __restrict__ xn16, float* __restrict__ sa,
                                                                    float* __restrict__ sb, long long* __restrict__ t_out, float* __restrict__ eps_out) {
  const long n = rows * d;
  const unsigned long long call = *counter;
  const DropKey kt = drop_key(seed, call), ke = drop_key(seed ^ 0x5851F42D4C957F2Dull, call);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const long r = e / d;
    int t = (int)(drop_uniform(kt, (uint64_t)r) * (float)T);
    t = t < T ? t : T - 1;
    const float a = acp[t], s_a = sqrtf(a), s_b = sqrtf(1.0f - a);
    const float u1 = 1.0f - drop_uniform(ke, 2 * (uint64_t)e), u2 = drop_uniform(ke, 2 * (uint64_t)e + 1);      // (0, 1], [0, 1)
    const float ep = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
    const float v = s_a * x[e] + s_b * ep;
    xn32[e] = v;
    xn16[e] = sq_f2bf(v);
    if (eps_out) eps_out[e] = ep;
    if (e - r * d == 0) { sa[r] = s_a; sb[r] = s_b; if (t_out) t_out[r] = t; }
  }
}

__global__ __launch_bounds__(256) void diffusion_combine_fwd_kernel(int B, int Nc, int d, const uint16_t* __restrict__ m, const float* __restrict__ xn32,
                                                                     const float* __restrict__ sa, const float* __restrict__ sb,
                                                                     const float* __restrict__ cls, const float* __restrict__ pe,
                                                                     float* __restrict__ z, uint16_t* __restrict__ z16) {
  const int S = Nc + 1;
  const long n = (long)B * S * d;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int c = (int)(e % d);
    const long bs = e / d;
    const int s = (int)(bs % S);
    const long b = bs / S;
    float v;
    if (s == 0) {
      v = cls[c];
    } else {
      const long r = b * Nc + (s - 1), q = r * d + c;
      v = (xn32[q] - sb[r] * sq_bf2f(m[q])) / sa[r];
    }
    v += pe[(long)s * d + c];
    z[e] = v;
    if (z16) z16[e] = sq_f2bf(v);                           // the first in-projection's operand: no cast node
  }
}

// dm = -(sb / sa) g[:, 1:, :] (bf16);  dcls += sum_b g[b, 0, :]  (float atomics, one per column and workgroup)
// g = g32 + g16 where both are given (the float32 residual stream's gradient and the bf16 in-projection's)
__global__ __launch_bounds__(256) void diffusion_combine_bwd_kernel(int B, int Nc, int d, const float* __restrict__ g, const uint16_t* __restrict__ g16,
                                                                     const float* __restrict__ sa,
                                                                     const float* __restrict__ sb, uint16_t* __restrict__ dm,
                                                                     float* __restrict__ dcls) {
  const int S = Nc + 1;
  const long n = (long)B * Nc * d;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256) {
    const int c = (int)(q % d);
    const long r = q / d, b = r / Nc;
    const int s = (int)(r - b * Nc) + 1;
    const long gi = (b * S + s) * d + c;
    dm[q] = sq_f2bf(-(sb[r] / sa[r]) * ((g ? g[gi] : 0.0f) + (g16 ? sq_bf2f(g16[gi]) : 0.0f)));
  }
  if (dcls && (int)blockIdx.x < 64) {                      // 64 workgroups share the [CLS] rows, eight independent loads per trip
    for (int c = threadIdx.x; c < d; c += 256) {
      float acc = 0.0f;
      for (long b0 = blockIdx.x; b0 < B; b0 += 64 * 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const long b = b0 + 64 * k, gi = (b * S) * d + c;
          v[k] = b < B ? (g ? g[gi] : 0.0f) + (g16 ? sq_bf2f(g16[gi]) : 0.0f) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
      }
      unsafeAtomicAdd(&dcls[c], acc);
    }
  }
}

}  // namespace opsamd

extern "C" int ops_diffusion_noise(long rows, int d, const float* x, const long long* t, const float* eps, const float* alpha_cumprod,
                                   float* xn32, void* xn16, float* sa, float* sb, void* stream) {
  if (rows < 1 || d < 1 || !x || !t || !eps || !alpha_cumprod || !xn32 || !xn16 || !sa || !sb) return OPS_AMD_ERR_INVALID_ARG;
  long nb = (rows * d + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(opsamd::diffusion_noise_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, rows, d, x, t, eps, alpha_cumprod, xn32,
                     (uint16_t*)xn16, sa, sb);
  return sq_check("diffusion_noise_kernel");
}

extern "C" int ops_diffusion_noise_draw(long rows, int d, int T, const float* x, const float* alpha_cumprod, unsigned long long seed,
                                        const unsigned long long* counter, float* xn32, void* xn16, float* sa, float* sb, long long* t_out,
                                        float* eps_out, void* stream) {
  if (rows < 1 || d < 1 || T < 1 || !x || !alpha_cumprod || !counter || !xn32 || !xn16 || !sa || !sb) return OPS_AMD_ERR_INVALID_ARG;
  long nb = (rows * d + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(opsamd::diffusion_noise_draw_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, rows, d, T, x, alpha_cumprod, seed,
                     counter, xn32, (uint16_t*)xn16, sa, sb, t_out, eps_out);
  return sq_check("diffusion_noise_draw_kernel");
}

extern "C" int ops_diffusion_combine_fwd(int B, int Nc, int d, const void* m, const float* xn32, const float* sa, const float* sb, const float* cls,
                                         const float* pe, float* z, void* z16, void* stream) {
  if (B < 1 || Nc < 1 || d < 1 || !m || !xn32 || !sa || !sb || !cls || !pe || !z) return OPS_AMD_ERR_INVALID_ARG;
  long nb = ((long)B * (Nc + 1) * d + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(opsamd::diffusion_combine_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, B, Nc, d, (const uint16_t*)m, xn32,
                     sa, sb, cls, pe, z, (uint16_t*)z16);
  return sq_check("diffusion_combine_fwd_kernel");
}

extern "C" int ops_diffusion_combine_bwd(int B, int Nc, int d, const float* g, const void* g16, const float* sa, const float* sb, void* dm, float* dcls,
                                         void* stream) {
  if (B < 1 || Nc < 1 || d < 1 || (!g && !g16) || !sa || !sb || !dm) return OPS_AMD_ERR_INVALID_ARG;
  long nb = ((long)B * Nc * d + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 64) nb = 64;
  hipLaunchKernelGGL(opsamd::diffusion_combine_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, B, Nc, d, g, (const uint16_t*)g16, sa, sb,
                     (uint16_t*)dm, dcls);
  return sq_check("diffusion_combine_bwd_kernel");
}
