// Layer blocks of the PINN's residual MLP (/root/reference/OpenPyStruct_PINN_MultiCase.py:395-541) for the training step:
//
//   input layer    y0 = dropout(LeakyReLU(input_norm(input_fc(x))))
//   residual block h  = dropout(LeakyReLU(fc1(o)));  z = fc2(h) + bn1(conv1(o)) + o;  o' = norm(z)          (x 2)
//   output layer   p  = output_fc(o'')
//
// one launch per Linear WITH everything that follows it up to the next Linear, and one per backward counterpart -- the
// captured step of that model was 63 kernel nodes (17 library GEMMs of 5.5-19 us for 8-30 MFLOP each, the elementwise tails,
// the stencil's five kernels, casts, bias reductions; profiles/r02_train_pinn_trace.txt) and becomes 15.
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
//   * the fragment loads of up to 16 reduction steps (512 columns) are in flight before the first MFMA, and EVERYTHING the epilogue
//     reads (bias, normalisation parameters, saved values, residual / stencil neighbourhoods, targets, partial sums) is requested
//     before them: one exposed memory latency per launch (a kernel's inputs were written by the previous launch on other XCDs, so
//     they come from the Infinity Cache, not from this XCD's L2: ~2 us per dependent round trip);
//   * the epilogue re-maps the 128 x 16 tile through LDS to "32 lanes per column": column sums are half-wave butterflies, the
//     transposed copies of the other operands (residual, saved pre-normalisation values, forward outputs for the
//     activation/dropout masks) are read and written coalesced;
//   * the ResidualBlock's single-channel BatchNorm1d(1) normalises over the WHOLE tensor: its sums are collected by extra
//     workgroups of the launch before the one that needs them (partial sums in a workspace, no atomics, no extra launch);
//   * the output layer's launch evaluates the training loss and its gradient on the tile it holds (no loss launch);
//   * the six weight gradients of the step are one grouped launch at the end (all operands are still resident).
// Arithmetic: bf16 operands, fp32 accumulation, layer outputs rounded to bf16 -- what nn.Linear under bf16 autocast does;
// statistics, normalisation and parameter gradients in fp32 (stencil sums in fp64).
// Dropout: keep-mask from a counter-based hash of (seed, call counter, element); the backward pass reads the mask and the
// LeakyReLU branch off the saved forward OUTPUT (0 = dropped, sign = branch), nothing else is stored.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"
#include "dropout_stream.hpp"
#include "call_counter.hpp"
#include "repack_tiles.hpp"

namespace opsamd {

#ifdef MB_EXP          // stand-alone experiment builds (scripts/mlp_launch_bench.py: 1 = no product, 2 = one reduction step)
inline void set_last_error(const char*) {}
#else
void set_last_error(const char* msg);   // beam_solve.hip: what ops_amd_last_error() reports
#endif

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
// Fragment-tiled storage of a [rows, K] bf16 matrix (rows padded to 16, K to 32; KS = K / 32): tile (rows >> 4, k >> 5) is 1 KB,
// stored in MFMA lane order -- lane = ((k >> 3) & 3) << 4 | (row & 15) holds 8 consecutive k -- so one wave-wide 16-byte load IS
// the A / B fragment of `v_mfma_f32_16x16x32_bf16` and touches 8 whole cache lines (row-major storage: 16 half lines per load,
// measured 2-4x the time per reduction step; profiles/r02_notes.md).
__device__ __forceinline__ long mb_toff(int row, int k, int KS) {
  return ((long)(row >> 4) * KS + (k >> 5)) * 512 + ((k >> 3) & 3) * 128 + (row & 15) * 8 + (k & 7);
}
// element (row r, column c) of a matrix whose TRANSPOSED copy [cols, 128] is stored tiled (rows = c, k = r, KS = 4)
__device__ __forceinline__ float mb_ldt(const void* p, int c, int r) { return mb_bf2f(((const uint16_t*)p)[mb_toff(c, r, MB_ROWS / 32)]); }
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
__device__ __forceinline__ float mb_uniform(uint64_t seed, uint64_t call, uint64_t idx) { return drop_uniform(seed, call, idx); }   // csrc/dropout_stream.hpp

// ---- the product: one 16 x 16 tile per wave, reduction in steps of 32, fragments straight from global memory ----
// lane l holds A[row l&15][k = 8 (l>>4) + j] and B[k][col l&15] (j = 0..7): 16 contiguous bytes of a row of either operand, and
// in the tiled storage the 64 lanes' 16 bytes are one contiguous KB.
// Chunks of MB_KCH steps, double buffered: the loads of chunk i + 1 are in flight while chunk i multiplies, and both of the first
// two chunks (up to 16 steps = 512 columns) are requested before the first MFMA.  Steps past KS re-read the last step (a hot
// line) and are skipped.
constexpr int MB_KCH = 8;

__device__ __forceinline__ void mb_load_chunk(uint4 (&fa)[MB_KCH], uint4 (&fb)[MB_KCH], const uint16_t* __restrict__ a_lane,
                                              const uint16_t* __restrict__ b_lane, int k0, int KS) {
#pragma unroll
  for (int j = 0; j < MB_KCH; ++j) {
    const int ks = k0 + j < KS ? k0 + j : KS - 1;
    fa[j] = *(const uint4*)(a_lane + ks * 512);
    fb[j] = *(const uint4*)(b_lane + ks * 512);
  }
}
__device__ __forceinline__ mb_f32x4 mb_mfma_chunk(mb_f32x4 acc, const uint4 (&fa)[MB_KCH], const uint4 (&fb)[MB_KCH], int k0, int KS) {
#pragma unroll
  for (int j = 0; j < MB_KCH; ++j)
    if (k0 + j < KS)      // wave-uniform
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mb_bf16x8, fa[j]), __builtin_bit_cast(mb_bf16x8, fb[j]), acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ mb_f32x4 mb_tile_product(const uint16_t* __restrict__ a_lane, const uint16_t* __restrict__ b_lane, int KS) {
  mb_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  uint4 fa0[MB_KCH], fb0[MB_KCH], fa1[MB_KCH], fb1[MB_KCH];
  mb_load_chunk(fa0, fb0, a_lane, b_lane, 0, KS);
  for (int k0 = 0; k0 < KS; k0 += 2 * MB_KCH) {
    if (k0 + MB_KCH < KS) mb_load_chunk(fa1, fb1, a_lane, b_lane, k0 + MB_KCH, KS);
    acc = mb_mfma_chunk(acc, fa0, fb0, k0, KS);
    if (k0 + 2 * MB_KCH < KS) mb_load_chunk(fa0, fb0, a_lane, b_lane, k0 + 2 * MB_KCH, KS);
    if (k0 + MB_KCH < KS) acc = mb_mfma_chunk(acc, fa1, fb1, k0 + MB_KCH, KS);
  }
  return acc;
}

// 3-tap stencil value at (row r, column q) of the transposed block input (zero padding at the ends, zero outside [0, No))
__device__ __forceinline__ float mb_conv_at(const void* Ot, int No, int q, int r, float w0, float w1, float w2, float b) {
  const float xm = q > 0 ? mb_ldt(Ot, q - 1, r) : 0.0f, xc = mb_ldt(Ot, q, r), xp = q + 1 < No ? mb_ldt(Ot, q + 1, r) : 0.0f;
  return __builtin_fmaf(w0, xm, __builtin_fmaf(w1, xc, __builtin_fmaf(w2, xp, b)));
}

// sums NV floats over the workgroup -- float inside a wave (64 terms), double across the waves; thread 0 gets the totals in `out`
template <int NV>
__device__ __forceinline__ void mb_block_sum_f(const float (&v)[NV], double (&out)[NV], double* s_red /*[8][NV]*/) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float w[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    w[k] = v[k];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) w[k] += __shfl_xor(w[k], s, 64);
  }
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) s_red[wave * NV + k] = (double)w[k];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double t = 0.0;
      for (int wv = 0; wv < MB_THREADS / 64; ++wv) t += s_red[wv * NV + k];
      out[k] = t;
    }
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

constexpr int MB_SIDE_COLS = 8;             // columns of the block input per side-job workgroup
constexpr int MB_MAX_SIDE = 64;             // side-job workgroups (= rows of partial sums) a launch may have: No <= 512
constexpr int ML_MAXG = 64;                 // strips of a TAIL_LOSS launch: C <= 1024

// ---- side job: partial sums for the whole-tensor BatchNorm1d(1) of the stencil path over 8 columns of the block input ----
// 1024 elements per workgroup, two per thread, every load independent: one exposed latency.
__device__ __forceinline__ void mb_side_job(const ops_mlp_strip_args& a, int row, double* s_red) {
  const int tid = threadIdx.x, r = tid & (MB_ROWS - 1), cj = tid >> 7;
  const int B = a.B, No = a.No;
  const float w0 = a.conv_w[0], w1 = a.conv_w[1], w2 = a.conv_w[2], cb = a.conv_b[0];
  const bool bwd = a.side == OPS_MLP_SIDE_BWD_STENCIL_SUMS;
  float xs[2][3], gi[2];
  bool live[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int cc = row * MB_SIDE_COLS + cj + 4 * e;
    live[e] = cc < No && r < B;
    xs[e][0] = (live[e] && cc > 0) ? mb_ldt(a.Ot, cc - 1, r) : 0.0f;
    xs[e][1] = live[e] ? mb_ldt(a.Ot, cc, r) : 0.0f;
    xs[e][2] = (live[e] && cc + 1 < No) ? mb_ldt(a.Ot, cc + 1, r) : 0.0f;
    gi[e] = (live[e] && bwd) ? mb_ldt(a.dZt, cc, r) : 0.0f;
  }
  if (!bwd) {
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (live[e]) {
        const float y = __builtin_fmaf(w0, xs[e][0], __builtin_fmaf(w1, xs[e][1], __builtin_fmaf(w2, xs[e][2], cb)));
        p0 += y;
        p1 = __builtin_fmaf(y, y, p1);
      }
    const float pv[2] = {p0, p1};
    double acc[2];
    mb_block_sum_f<2>(pv, acc, s_red);
    if (tid == 0) { a.spart[row * 2] = acc[0]; a.spart[row * 2 + 1] = acc[1]; }
  } else {
    // 0: sum g   1: sum g yhat   2: sum yhat   3..5: sum g x_s   6..8: sum x_s   9..11: sum yhat x_s   (x_s = O shifted by s - 1)
    const float mean_s = a.ssave[0], inv_s = a.ssave[1];
    float t[MB_NSUM];
#pragma unroll
    for (int k = 0; k < MB_NSUM; ++k) t[k] = 0.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (live[e]) {
        const float conv = __builtin_fmaf(w0, xs[e][0], __builtin_fmaf(w1, xs[e][1], __builtin_fmaf(w2, xs[e][2], cb)));
        const float yh = (conv - mean_s) * inv_s;
        t[0] += gi[e];
        t[1] = __builtin_fmaf(gi[e], yh, t[1]);
        t[2] += yh;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          t[3 + s] = __builtin_fmaf(gi[e], xs[e][s], t[3 + s]);
          t[6 + s] += xs[e][s];
          t[9 + s] = __builtin_fmaf(yh, xs[e][s], t[9 + s]);
        }
      }
    double acc[MB_NSUM];
    mb_block_sum_f<MB_NSUM>(t, acc, s_red);
    if (tid == 0)
#pragma unroll
      for (int k = 0; k < MB_NSUM; ++k) a.spart[row * MB_NSUM + k] = acc[k];
  }
}

// r06: the launch's mode (tail, the block's addition, evaluation statistics) as TEMPLATE parameters for the combinations the PINN step uses --
// the generic kernel decoded them per wave at run time (450-580 scalar instructions and 27 KB of code per strip launch, profiles/r06_notes.md 9);
// TAIL_ = -1: the generic form (any valid combination of the C ABI)
template <int TAIL_, int ADD_, int EVAL_>
__global__ __launch_bounds__(MB_THREADS) void mlp_strip_kernel(const ops_mlp_strip_args a, const int nstrips) {
  const int a_tail = TAIL_ >= 0 ? TAIL_ : a.tail, a_add_mode = TAIL_ >= 0 ? ADD_ : a.add_mode;
  const bool a_eval_stats = TAIL_ >= 0 ? (EVAL_ != 0) : (a.eval_stats != 0);
  __shared__ float s_t[MB_ROWS][MB_COLS + 1];
  __shared__ __attribute__((aligned(16))) uint16_t s_y[MB_ROWS][MB_COLS];
  __shared__ __attribute__((aligned(16))) uint16_t s_z[MB_ROWS][MB_COLS];
  __shared__ __attribute__((aligned(16))) uint16_t s_stage[70][MB_ROWS];      // transposed epilogue operands: [column slot][row]
  __shared__ double s_red[(MB_THREADS / 64) * MB_NSUM];
  __shared__ double s_tot[16];
  if ((int)blockIdx.x >= nstrips) {      // workgroup-uniform
    mb_side_job(a, (int)blockIdx.x - nstrips, s_red);
    return;
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int n0 = blockIdx.x * MB_COLS;
  // evaluation slots (grid.y): this workgroup's batch lives soff bytes behind every per-batch pointer
  const long soff = (long)blockIdx.y * (long)a.slot_stride;
  const auto in_slot = [soff](auto* p) { return p ? (decltype(p))((uintptr_t)p + (uintptr_t)soff) : p; };
  const void* const sA = in_slot(a.A);
  void* const sY = in_slot(a.Y);
  void* const sYt = in_slot(a.Yt);
  void* const sZt = in_slot(a.Zt);
  const void* const sOt = in_slot(a.Ot);
  void* const sP = in_slot(a.P);
  const float* const s_targets_t = in_slot(a.targets_t);
  void* const s_loss_ws = in_slot(a.loss_ws);
  float* const s_loss = in_slot(a.loss);
  float* const s_loss_sum = in_slot(a.loss_sum);
  const int B = a.n_slots > 0 ? min(a.B, a.slot_total_rows - (int)blockIdx.y * a.B) : a.B, N = a.N, No = a.No;
  const bool fwd = a_tail <= OPS_MLP_TAIL_BN || a_tail == OPS_MLP_TAIL_LOSS;
  const bool has_bn = a_tail == OPS_MLP_TAIL_BN || a_tail == OPS_MLP_TAIL_BN_ACT_DROP || a_tail == OPS_MLP_TAIL_BWD_BN ||
                      a_tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP;
  const bool act_bwd = a_tail == OPS_MLP_TAIL_BWD_ACT_DROP || a_tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP;
  // epilogue mapping: 32 lanes per column, 4 rows per lane
  const int cl = tid >> 5, q = tid & 31, c = n0 + cl;
  const bool clive = c < N;
  bool rl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rl[i] = clive && q + 32 * i < B;

  // ---- phase 0: everything the epilogue reads is requested BEFORE the product's operands (one exposed latency per launch) ----
  float bias = 0.0f, g = 1.0f, be = 0.0f, mean_c = 0.0f, rstd_c = 1.0f;
  if (clive) {
    if (fwd && a.bias) bias = a.bias[c];
    if (has_bn) {
      g = a.gamma[c]; be = a.beta[c];
      if (!fwd) { mean_c = a.mean[c]; rstd_c = a.rstd[c]; }
    }
  }
  // the transposed operands of the epilogue arrive as 16-byte chunks (column, 8 rows) -- up to 70 columns x 16 chunks per
  // workgroup, at most three per thread -- and are parked in LDS after the product; slots (columns) of the staging area:
  //   0..19  block input O, columns n0 - 2 .. n0 + 17        20..37  dZ, columns n0 - 1 .. n0 + 16
  //   38..53 saved pre-normalisation values, n0 .. n0 + 15   54..69  forward output of the activation tail, n0 .. n0 + 15
  uint4 ch[3];
  bool chv[3];
#pragma unroll
  for (int rep = 0; rep < 3; ++rep) {
    const int j = tid + MB_THREADS * rep, slot = j >> 4, gq = j & 15;
    const void* src = nullptr;
    int col = 0, lim = 0;
    if (slot < 20) { if (a_add_mode != OPS_MLP_ADD_NONE) { src = sOt; col = n0 - 2 + slot; lim = No; } }
    else if (slot < 38) { if (a_add_mode == OPS_MLP_ADD_BWD_BLOCK) { src = a.dZt; col = n0 - 1 + (slot - 20); lim = No; } }
    else if (slot < 54) { if (has_bn && !fwd) { src = sZt; col = n0 + (slot - 38); lim = N; } }
    else if (slot < 70) { if (act_bwd) { src = a.Yref_t; col = n0 + (slot - 54); lim = N; } }
    chv[rep] = slot < 70;
    ch[rep] = uint4{0u, 0u, 0u, 0u};
    if (src && col >= 0 && col < lim) ch[rep] = *(const uint4*)((const uint16_t*)src + mb_toff(col, 8 * gq, MB_ROWS / 32));
  }
  float pt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pt[i] = (rl[i] && a_tail == OPS_MLP_TAIL_LOSS) ? s_targets_t[(long)c * MB_ROWS + q + 32 * i] : 0.0f;
  // loss of the previous launch (TAIL_LOSS leaves per-strip partial sums): workgroup 0 adds them, one partial row per lane
  double lp[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const bool fin = a.loss_finish_rows > 0 && blockIdx.x == 0 && wave == 0;
  if (fin && lane < a.loss_finish_rows)
#pragma unroll
    for (int k = 0; k < 5; ++k) lp[k] = ((const double*)s_loss_ws)[lane * 5 + k];
  // partial sums of the stencil normalisation: lane = row of partials, wave = which sum (workgroup 0 needs all twelve backward)
  double pp0 = 0.0, pp1 = 0.0;
  if (a_add_mode != OPS_MLP_ADD_NONE) {
    const int rows = (No + MB_SIDE_COLS - 1) / MB_SIDE_COLS, NS = fwd ? 2 : MB_NSUM;
    if (lane < rows) {
      if (wave < 2 || (blockIdx.x == 0 && wave < NS)) pp0 = a.spart[lane * NS + wave];
      if (blockIdx.x == 0 && wave + 8 < NS) pp1 = a.spart[lane * NS + wave + 8];
    }
  }
  float sc_w0 = 0.0f, sc_w1 = 0.0f, sc_w2 = 0.0f, sc_b = 0.0f, sc_g = 0.0f, sc_be = 0.0f, sv_mean = 0.0f, sv_inv = 1.0f;
  if (a_add_mode != OPS_MLP_ADD_NONE) {
    sc_w0 = a.conv_w[0]; sc_w1 = a.conv_w[1]; sc_w2 = a.conv_w[2]; sc_b = a.conv_b[0]; sc_g = a.sgamma[0]; sc_be = a.sbeta[0];
    if (a_add_mode == OPS_MLP_ADD_BWD_BLOCK) { sv_mean = a.ssave[0]; sv_inv = a.ssave[1]; }
  }
  const unsigned long long call = (fwd && a.p_drop > 0.0f && a.call_counter) ? *a.call_counter : 0ull;

  // ---- product ----
  {
#if defined(MB_EXP) && MB_EXP == 2
    const int KS = 1;
#else
    const int KS = (a.K + 31) >> 5;
#endif
    // tile (row block, step) = 1 KB in lane order: this wave's row block of A, this strip's row block of W
    const uint16_t* ap = (const uint16_t*)sA + (long)wave * (a.lda >> 5) * 512 + lane * 8;
    const uint16_t* bp = (const uint16_t*)a.W + (long)blockIdx.x * (a.ldw >> 5) * 512 + lane * 8;
#if defined(MB_EXP) && MB_EXP == 1
    const mb_f32x4 acc = {(float)(ap - bp), (float)KS, 0.0f, 0.0f};
#else
    const mb_f32x4 acc = mb_tile_product(ap, bp, KS);
#endif
    // C layout: column lane & 15, rows 4 (lane >> 4) + i
#pragma unroll
    for (int i = 0; i < 4; ++i) s_t[wave * 16 + (lane >> 4) * 4 + i][lane & 15] = acc[i];
  }
  if (a_add_mode != OPS_MLP_ADD_NONE) {
    pp0 = mb_wsum_d(pp0); pp1 = mb_wsum_d(pp1);
    if (lane == 0) { s_tot[wave] = pp0; s_tot[wave + 8] = pp1; }
  }
#pragma unroll
  for (int rep = 0; rep < 3; ++rep)
    if (chv[rep]) *(uint4*)&s_stage[(tid + MB_THREADS * rep) >> 4][8 * (tid & 15)] = ch[rep];
  if (fin) {
#pragma unroll
    for (int k = 0; k < 5; ++k) lp[k] = mb_wsum_d(lp[k]);
    if (lane == 0) {
      const double al = fmin(fmax((double)a.alpha[0], 1e-6), 1.0);
      const int nI = a.nI, nD = a.nD, nR = a.loss_C - nI - nD;
      const double nIe = (double)B * nI;
      double val = al * lp[0] / nIe + (1.0 - al) * lp[1] / nIe + (double)a.box_weight * lp[2];
      if (nD > 0) val += (double)a.rel_penalty * lp[3] / ((double)B * nD);
      if (nR > 0) val += (double)a.rel_penalty * lp[4] / ((double)B * nR);
      const double da = a.alpha0 == a.alpha0 ? (double)a.alpha0 - (double)a.alpha[0] : 0.0;      // NaN alpha0: no such term
      s_loss[0] = (float)(val + da * da);
      if (s_loss_sum) s_loss_sum[0] += (float)(val + da * da);      // the epoch's running total (the caller zeroes it)
    }
  }
  __syncthreads();

  // ---- epilogue ----
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = s_t[q + 32 * i][cl];
  const float invB = 1.0f / (float)B;
  const float keep_scale = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  float lacc[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // TAIL_LOSS: sum |d|_I, sum d^2_I, sum box penalty, sum rel_d, sum rel_r

  if (fwd) {
    const float bias_b = mb_round(bias);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = rl[i] ? mb_round(v[i] + bias_b) : 0.0f;       // the Linear's bf16 output
    if (a_add_mode == OPS_MLP_ADD_FWD_BLOCK) {
      // whole-tensor statistics of conv1(O) from the previous launch's partial sums
      const double n = (double)B * (double)No, m = s_tot[0] / n, var = fmax(s_tot[1] / n - m * m, 0.0);
      const float mean_s = a_eval_stats ? a.srunning_mean[0] : (float)m;
      const float inv_s = a_eval_stats ? (float)(1.0 / sqrt((double)a.srunning_var[0] + (double)a.seps)) : (float)(1.0 / sqrt(var + (double)a.seps));
      if (blockIdx.x == 0 && tid == 0 && !a_eval_stats) {
        a.ssave[0] = mean_s; a.ssave[1] = inv_s;
        a.srunning_mean[0] = (1.0f - a.smomentum) * a.srunning_mean[0] + a.smomentum * (float)m;
        a.srunning_var[0] = (1.0f - a.smomentum) * a.srunning_var[0] + a.smomentum * (float)(var * n / (n > 1.0 ? n - 1.0 : 1.0));
        if (a.snum_batches_tracked) a.snum_batches_tracked[0] += 1;
      }
      const float scale = sc_g * inv_s, shift = sc_be - mean_s * scale;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rl[i]) {
          const int r = q + 32 * i;
          const float o = mb_bf2f(s_stage[cl + 2][r]);
          const float conv = __builtin_fmaf(sc_w0, mb_bf2f(s_stage[cl + 1][r]), __builtin_fmaf(sc_w1, o, __builtin_fmaf(sc_w2, mb_bf2f(s_stage[cl + 3][r]), sc_b)));
          const float s = mb_round(__builtin_fmaf(conv, scale, shift));
          v[i] = mb_round(v[i] + s + o);
        }
    }
    if (has_bn) {
      float sm = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sm += v[i];                 // dead rows hold 0
      float mean = mb_hsum(sm) * invB;
      float sq = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float d = v[i] - mean; sq += rl[i] ? d * d : 0.0f; }
      float var = mb_hsum(sq) * invB;                         // biased: what normalises
      float rstd = rsqrtf(var + a.eps);
      if (a_eval_stats) {                                     // model.eval(): the running statistics, read only
        mean = clive ? a.running_mean[c] : 0.0f;
        rstd = clive ? rsqrtf(a.running_var[c] + a.eps) : 1.0f;
      }
      if (clive && q == 0 && !a_eval_stats) {
        a.mean[c] = mean; a.rstd[c] = rstd;
        if (a.running_mean) {                                 // momentum update with the UNBIASED variance
          const float unb = var * ((float)B / (float)(B > 1 ? B - 1 : 1));
          a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mean;
          a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * unb;
        }
      }
      if (blockIdx.x == 0 && tid == 0 && a.num_batches_tracked && !a_eval_stats) a.num_batches_tracked[0] += 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s_z[q + 32 * i][cl] = mb_f2bf(v[i]);                 // exact: v is a bf16 value; leaves with the results below
        v[i] = rl[i] ? __builtin_fmaf((v[i] - mean) * rstd, g, be) : 0.0f;
      }
    }
    if (a_tail == OPS_MLP_TAIL_ACT_DROP || a_tail == OPS_MLP_TAIL_BN_ACT_DROP) {
      const bool drop = a.p_drop > 0.0f;
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
    if (a_tail == OPS_MLP_TAIL_LOSS) {
      // the predictions leave through LDS first, then v becomes d loss / d predictions
      if (sP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_y[q + 32 * i][cl] = mb_f2bf(v[i]);
        __syncthreads();
        if (tid < 2 * MB_ROWS) {
          const int r = tid >> 1, h = tid & 1;
          *(uint4*)((uint16_t*)sP + (long)r * a.ldp + n0 + 8 * h) = *(const uint4*)&s_y[r][8 * h];
        }
        __syncthreads();
      }
      const float alpha = fminf(fmaxf(a.alpha[0], 1e-6f), 1.0f);
      const bool has_min = a.min_constraint != nullptr, has_max = a.max_constraint != nullptr;
      const float lo = has_min ? a.min_constraint[0] : 0.0f, hi = has_max ? a.max_constraint[0] : 0.0f;
      const int nI = a.nI, nD = a.nD, nR = N - nI - nD;
      const float inv_nI = 1.0f / ((float)B * (float)nI), inv_nD = nD > 0 ? 1.0f / ((float)B * (float)nD) : 0.0f,
                  inv_nR = nR > 0 ? 1.0f / ((float)B * (float)nR) : 0.0f;
      const float w = a.box_weight, penalty = a.rel_penalty, eps = 1e-8f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float gg = 0.0f;
        if (rl[i]) {
          const float p = v[i], t = pt[i], d = p - t, sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
          if (c < nI) {
            lacc[0] += fabsf(d);
            lacc[1] = __builtin_fmaf(d, d, lacc[1]);
            gg = (alpha * sg + (1.0f - alpha) * 2.0f * d) * inv_nI;
            if (has_min && p < lo) { lacc[2] += lo - p; gg -= w; }
            if (has_max && p > hi) { lacc[2] += p - hi; gg += w; }
          } else {
            const float den = fabsf(t) + eps, rel = fabsf(d) / den;
            if (c < nI + nD) { lacc[3] += rel; gg = penalty * sg / den * inv_nD; }
            else { lacc[4] += rel; gg = penalty * sg / den * inv_nR; }
          }
        }
        v[i] = gg;
      }
      float sb = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) sb += mb_round(v[i]);
      sb = mb_hsum(sb);
      if (clive && q == 0 && a.dbias) a.dbias[c] = sb;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = rl[i] ? mb_round(v[i]) : 0.0f;              // the input gradient's bf16 value
    if (a_add_mode == OPS_MLP_ADD_BWD_BLOCK) {
      // + dZ (identity path) + conv1^T( bn1 backward (dZ) ) (stencil path); the whole-tensor means from the partial sums
      const double n = (double)B * (double)No;
      const float mg = (float)(s_tot[0] / n), mgy = (float)(s_tot[1] / n);
      const float kk = sc_g * sv_inv;
      if (blockIdx.x == 0 && tid == 0) {
        // dy = kk (g - mg - yhat mgy):  sum dy x_s = kk (sum g x_s - mg sum x_s - mgy sum yhat x_s);  sum dy likewise with x_s = 1
        for (int s = 0; s < 3; ++s)
          a.sdparams[s] = (float)((double)kk * (s_tot[3 + s] - (double)mg * s_tot[6 + s] - (double)mgy * s_tot[9 + s]));
        a.sdparams[3] = (float)((double)kk * (s_tot[0] - (double)mg * n - (double)mgy * s_tot[2]));
        a.sdparams[4] = (float)s_tot[1];        // d gamma = sum g yhat
        a.sdparams[5] = (float)s_tot[0];        // d beta  = sum g
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rl[i]) {
          const int r = q + 32 * i;
          float dy[3], po[5], pg[3];
#pragma unroll
          for (int d = 0; d < 5; ++d) po[d] = mb_bf2f(s_stage[cl + d][r]);
#pragma unroll
          for (int d = 0; d < 3; ++d) pg[d] = mb_bf2f(s_stage[20 + cl + d][r]);
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const int qq = c + d - 1;
            if (qq >= 0 && qq < No) {
              // columns outside [0, No) were prefetched as 0: the stencil's zero padding
              const float conv = __builtin_fmaf(sc_w0, po[d], __builtin_fmaf(sc_w1, po[d + 1], __builtin_fmaf(sc_w2, po[d + 2], sc_b)));
              const float yh = (conv - sv_mean) * sv_inv;
              dy[d] = kk * (pg[d] - mg - yh * mgy);
            } else {
              dy[d] = 0.0f;
            }
          }
          const float sdx = __builtin_fmaf(sc_w0, dy[2], __builtin_fmaf(sc_w1, dy[1], sc_w2 * dy[0]));
          v[i] = mb_round(v[i] + pg[1] + sdx);
        }
    }
    if (act_bwd) {
      // mask and LeakyReLU branch from the saved forward output: 0 = dropped, sign = sign of the pre-activation
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (rl[i]) {
          const float y = mb_bf2f(s_stage[54 + cl][q + 32 * i]);
          v[i] *= y == 0.0f ? 0.0f : (y > 0.0f ? keep_scale : a.slope * keep_scale);
        }
    }
    if (has_bn) {
      float xh[4], sg = 0.0f, sgx = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xh[i] = rl[i] ? (mb_bf2f(s_stage[38 + cl][q + 32 * i]) - mean_c) * rstd_c : 0.0f;
        sg += v[i];
        sgx = __builtin_fmaf(v[i], xh[i], sgx);
      }
      sg = mb_hsum(sg); sgx = mb_hsum(sgx);
      if (clive && q == 0) { a.dgamma[c] = sgx; a.dbeta[c] = sg; }
      const float k2 = g * rstd_c;
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

  // ---- results through LDS, 16-byte chunks in the tiled layout: threads 0..255 the row-major copy (row, 8 columns), threads
  // 256..511 the transposed copy (column, 8 rows); the saved pre-normalisation values likewise (transposed only) ----
#pragma unroll
  for (int i = 0; i < 4; ++i) s_y[q + 32 * i][cl] = rl[i] ? mb_f2bf(v[i]) : (uint16_t)0;
  __syncthreads();
  if (tid < 2 * MB_ROWS) {
    const int r = tid >> 1, h = tid & 1;
    *(uint4*)((uint16_t*)sY + mb_toff(r, n0 + 8 * h, a.ldy >> 5)) = *(const uint4*)&s_y[r][8 * h];
  } else if (sYt) {
    const int u = tid - 2 * MB_ROWS, c2 = u & 15, gq = u >> 4;
    uint16_t t8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t8[j] = s_y[8 * gq + j][c2];
    *(uint4*)((uint16_t*)sYt + mb_toff(n0 + c2, 8 * gq, MB_ROWS / 32)) = *(const uint4*)t8;
  }
  if (fwd && has_bn && tid < 2 * MB_ROWS) {
    const int c2 = tid & 15, gq = tid >> 4;
    uint16_t t8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t8[j] = s_z[8 * gq + j][c2];
    *(uint4*)((uint16_t*)sZt + mb_toff(n0 + c2, 8 * gq, MB_ROWS / 32)) = *(const uint4*)t8;
  }
  // one increment per launch, by the last STRIP workgroup to finish: the side-job workgroups behind them left without reading the
  // counter and do not report (call_counter.hpp)
  if (fwd && a.p_drop > 0.0f && (a_tail == OPS_MLP_TAIL_ACT_DROP || a_tail == OPS_MLP_TAIL_BN_ACT_DROP))
    call_counter_done(a.call_counter, (unsigned)nstrips * gridDim.y);

  if (a_tail == OPS_MLP_TAIL_LOSS) {
    // loss value: per-strip partial sums; workgroup 0 of the NEXT launch (loss_finish_rows) adds them -- no atomics, no fence here
    double accd[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) accd[k] = (double)lacc[k];
    mb_block_sum<5>(accd, s_red);
    if (tid == 0)
#pragma unroll
      for (int k = 0; k < 5; ++k) ((double*)s_loss_ws)[blockIdx.x * 5 + k] = accd[k];
  }
}

// ---- grouped weight gradients: out [N, K] fp32 = At [N, 128] x Bt [K, 128]^T, one wave per 32 x 32 tile ----
struct WgradTable {
  int nprob;
  int tile0[OPS_MLP_MAX_WGRAD + 1];        // first tile of each problem
  int tiles_k[OPS_MLP_MAX_WGRAD];
  ops_mlp_wgrad_problem p[OPS_MLP_MAX_WGRAD];
};
// r05: what the optimiser needs of the gradient norm, left by the launch that produces the gradients (ops_mlp_wgrad_group_norm)
struct WgradNorm {
  double* part;                            // [tiles + nrange] partial sums of (g * scale)^2, then the two bias corrections at OPS_FLAT_ADAM_MAX_PARTS
  int32_t* step;
  float scale, beta1, beta2;
  int nrange;
  const float* rptr[OPS_MLP_MAX_NORM_RANGES];
  int rlen[OPS_MLP_MAX_NORM_RANGES];
};
__device__ __forceinline__ float mb_wsum_f(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

template <bool NORM>
__global__ __launch_bounds__(64) void mlp_wgrad_kernel(const WgradTable tb, const WgradNorm nm) {
  if constexpr (NORM) {
    const int ntile = tb.tile0[tb.nprob];
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // the optimiser step this launch belongs to: count and bias corrections, once
      const int st = nm.step[0] + 1;
      nm.step[0] = st;
      nm.part[OPS_FLAT_ADAM_MAX_PARTS] = 1.0 - pow((double)nm.beta1, (double)st);
      nm.part[OPS_FLAT_ADAM_MAX_PARTS + 1] = sqrt(1.0 - pow((double)nm.beta2, (double)st));
    }
    if ((int)blockIdx.x >= ntile) {                 // a range of gradients no matrix covers (written by earlier launches)
      const int r = (int)blockIdx.x - ntile;
      const float* g = nullptr; int len = 0;
#pragma unroll
      for (int k = 0; k < OPS_MLP_MAX_NORM_RANGES; ++k) if (k == r) { g = nm.rptr[k]; len = nm.rlen[k]; }
      float a0 = 0.0f, a1 = 0.0f;
      for (int i = threadIdx.x; i < len; i += 128) {
        const float x = g[i] * nm.scale, y = i + 64 < len ? g[i + 64] * nm.scale : 0.0f;
        a0 = __builtin_fmaf(x, x, a0); a1 = __builtin_fmaf(y, y, a1);
      }
      const double d = mb_wsum_d((double)a0 + (double)a1);
      if (threadIdx.x == 0) nm.part[blockIdx.x] = d;
      return;
    }
  }
  int pi = 0;
  while (pi + 1 < tb.nprob && (int)blockIdx.x >= tb.tile0[pi + 1]) ++pi;
  const ops_mlp_wgrad_problem pr = tb.p[pi];
  const int t = (int)blockIdx.x - tb.tile0[pi], tn = t / tb.tiles_k[pi], tk = t % tb.tiles_k[pi];
  const int lane = threadIdx.x;
  // tiled storage, KS = 4: row block rb is 4 consecutive KB
  const uint16_t* ap = (const uint16_t*)pr.At + (long)(tn * 2) * 4 * 512 + lane * 8;
  const uint16_t* bp = (const uint16_t*)pr.Bt + (long)(tk * 2) * 4 * 512 + lane * 8;
  uint4 fa[2][4], fb[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[h][ks] = *(const uint4*)(ap + (h * 4 + ks) * 512);
      fb[h][ks] = *(const uint4*)(bp + (h * 4 + ks) * 512);
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
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = tk * 32 + j * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = tn * 32 + i * 16 + (lane >> 4) * 4 + e;
        if (n < pr.N && k < pr.K) {
          pr.out[(long)n * pr.ldo + k] = acc[i][j][e];
          if constexpr (NORM) { const float v = acc[i][j][e] * nm.scale; sq = __builtin_fmaf(v, v, sq); }
        }
      }
    }
  if constexpr (NORM) {                     // 16 values per lane in float, the 64 lanes and everything after in double
    const double d = mb_wsum_d((double)sq);
    if (lane == 0) nm.part[blockIdx.x] = d;
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
    ((uint16_t*)m.Wp)[mb_toff(n, k, m.ldw >> 5)] = h;
    ((uint16_t*)m.Wtp)[mb_toff(k, n, m.ldwt >> 5)] = h;
  }
}

// ---- batch assembly into the layout: gather + noise + bf16, row-major and transposed (tile of 32 features x 128 rows) ----
__device__ __forceinline__ uint64_t mb_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// r05: REPACK -- the tiled bf16 weight copies of the PREVIOUS optimiser step are rebuilt by extra workgroups of this launch (four 1 KB
// tiles each, repack_tiles.hpp): the step's last launch (the copies) and the next step's first (the batch) depend on nothing of each other,
// so they are one launch -- one kernel boundary and the copies' 5 us (shorter than the assembly's 8) off every step.
constexpr int MB_RSPLIT = 4, MB_GROWS = MB_ROWS / MB_RSPLIT;      // row groups of the batch assembly (32 rows each)
template <bool REPACK>
__global__ __launch_bounds__(256) void mlp_gather_noise_kernel(int B, int F, const float* __restrict__ X, const long long* __restrict__ idx,
                                                                const float* __restrict__ sigma, unsigned long long seed,
                                                                unsigned long long* __restrict__ counter, uint16_t* __restrict__ out, int ld,
                                                                uint16_t* __restrict__ out_t, int nfb, const float* __restrict__ Ysrc, int C,
                                                                float* __restrict__ yout_t, int ngb, const float* __restrict__ params,
                                                                const AdamRepack rp, const TileJobs tj) {
  if constexpr (REPACK) {
    if ((int)blockIdx.x >= ngb) {          // workgroup-uniform: a weight-copy workgroup
      repack_tile_job(params, rp, tj, ((int)blockIdx.x - ngb) * 4 + (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
      return;
    }
  }
  // r06: a workgroup assembles 32 features (or 32 target columns) x 32 ROWS -- a quarter of the 128-row batch (MB_GROWS), four rows per thread:
  // the launch is one round of waves whose length is a thread's chain (indices -> rows -> noise -> stores), and sixteen rows per thread made
  // that chain the longest thing in the step's first launch (8.2 us for the assembly alone against 5.2 for the weight copies riding with it)
  __shared__ __attribute__((aligned(16))) uint16_t s_tile[32][MB_GROWS + 8];     // [feature][row of the group]
  __shared__ float s_tt[32][MB_GROWS + 1];
  const int fl = threadIdx.x & 31, wq = threadIdx.x >> 5;
  const int tile = (int)blockIdx.x / MB_RSPLIT, rg = (int)blockIdx.x % MB_RSPLIT, b0 = rg * MB_GROWS;
  long src[4];
  float xv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = b0 + wq + 8 * j;
    src[j] = b < B ? (long)idx[b] : -1;
  }
  if (tile >= nfb) {
    // targets: yout_t [C, 128] float32 (transposed: the loss tail reads a column's rows contiguously), rows >= B zero
    const int c0 = (tile - nfb) * 32, c = c0 + fl;
#pragma unroll
    for (int j = 0; j < 4; ++j) xv[j] = (src[j] >= 0 && c < C) ? Ysrc[src[j] * (long)C + c] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s_tt[fl][wq + 8 * j] = xv[j];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * MB_GROWS; e += 256) {
      const int cc = e / MB_GROWS, b = e % MB_GROWS;
      if (c0 + cc < C) yout_t[(long)(c0 + cc) * MB_ROWS + b0 + b] = s_tt[cc][b];
    }
    return;
  }
  const int f0 = tile * 32, f = f0 + fl;
  const float sg = sigma ? *sigma : 0.0f;
  const unsigned long long call = counter ? *counter : 0ull;
#pragma unroll
  for (int j = 0; j < 4; ++j) xv[j] = (src[j] >= 0 && f < F) ? X[src[j] * (long)F + f] : 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int bl = wq + 8 * j, b = b0 + bl;
    uint16_t h = 0;
    if (b < B && f < F) {
      float v = xv[j];
      if (sg != 0.0f) {
        const uint64_t hh = mb_mix(seed + 0x9E3779B97F4A7C15ull * (call + 1) + (uint64_t)((long)b * F + f) * 0xD1B54A32D192ED03ull);
        const float u1 = ((float)(hh >> 40) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
        const float u2 = (float)((hh >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);   // [0, 1)
        v += sg * sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
      }
      h = mb_f2bf(v);
    }
    s_tile[fl][bl] = h;
  }
  __syncthreads();
  // both tiled copies in 16-byte chunks: 128 chunks (row, 8 features) of x and 128 chunks (feature, 8 rows) of x^T, one per thread
  {
    const int u = threadIdx.x & 127;
    if (threadIdx.x < 128) {
      const int bl = u >> 2, fg = u & 3;                      // row b0 + bl, features f0 + 8 fg ..
      uint16_t t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t8[j] = s_tile[8 * fg + j][bl];
      *(uint4*)(out + mb_toff(b0 + bl, f0 + 8 * fg, ld >> 5)) = *(const uint4*)t8;
    } else {
      const int ff = u & 31, bg = u >> 5;                     // feature f0 + ff, rows b0 + 8 bg ..
      *(uint4*)(out_t + mb_toff(f0 + ff, b0 + 8 * bg, MB_ROWS / 32)) = *(const uint4*)&s_tile[ff][8 * bg];
    }
  }
  if (counter) call_counter_done(counter, (unsigned)(nfb * MB_RSPLIT));   // the target workgroups behind the feature ones left early, unreported
}

}  // namespace opsamd

using namespace opsamd;

static inline int ru(int v, int m) { return (v + m - 1) / m * m; }

extern "C" size_t ops_mlp_spart_doubles(int No) { return (size_t)((No + MB_SIDE_COLS - 1) / MB_SIDE_COLS) * MB_NSUM; }

extern "C" int ops_mlp_strip_launch(const ops_mlp_strip_args* args, void* stream) {
  if (!args) return OPS_AMD_ERR_INVALID_ARG;
  const ops_mlp_strip_args& a = *args;
  if (a.B < 1 || a.B > MB_ROWS || a.N < 1 || a.K < 1 || !a.A || !a.W || !a.Y) return OPS_AMD_ERR_INVALID_ARG;
  if (a.lda % 32 || a.ldw % 32 || a.ldy % 32 || a.lda < ru(a.K, 32) || a.ldw < ru(a.K, 32) || a.ldy < ru(a.N, 32)) return OPS_AMD_ERR_INVALID_ARG;
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.Y) & 15) return OPS_AMD_ERR_INVALID_ARG;
  if (a.tail < OPS_MLP_TAIL_NONE || a.tail > OPS_MLP_TAIL_LOSS) return OPS_AMD_ERR_INVALID_ARG;
  const bool bn = a.tail == OPS_MLP_TAIL_BN || a.tail == OPS_MLP_TAIL_BN_ACT_DROP || a.tail == OPS_MLP_TAIL_BWD_BN || a.tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP;
  const bool bwd = a.tail >= OPS_MLP_TAIL_BWD_ACT_DROP && a.tail <= OPS_MLP_TAIL_BWD_BN_ACT_DROP;
  if (bn && (!a.gamma || !a.beta || !a.mean || !a.rstd || !a.Zt)) return OPS_AMD_ERR_INVALID_ARG;
  if (a.eval_stats && (bwd || a.p_drop > 0.0f || a.side != OPS_MLP_SIDE_NONE || (bn && (!a.running_mean || !a.running_var)))) return OPS_AMD_ERR_INVALID_ARG;
  if (bn && bwd && (!a.dgamma || !a.dbeta)) return OPS_AMD_ERR_INVALID_ARG;
  if ((a.tail == OPS_MLP_TAIL_BWD_ACT_DROP || a.tail == OPS_MLP_TAIL_BWD_BN_ACT_DROP) && !a.Yref_t) return OPS_AMD_ERR_INVALID_ARG;
  if ((a.tail == OPS_MLP_TAIL_ACT_DROP || a.tail == OPS_MLP_TAIL_BN_ACT_DROP) && a.p_drop > 0.0f && (!a.call_counter || a.p_drop >= 1.0f))
    return OPS_AMD_ERR_INVALID_ARG;
  const int nstrips = (a.N + MB_COLS - 1) / MB_COLS;
  if (a.tail == OPS_MLP_TAIL_LOSS) {
    if (!a.targets_t || !a.alpha || !a.loss_ws || !a.Yt || a.nI < 1 || a.nD < 0 || a.nI + a.nD > a.N || nstrips > ML_MAXG) return OPS_AMD_ERR_INVALID_ARG;
    if (a.P && (a.ldp % 8 || a.ldp < ru(a.N, MB_COLS) || ((uintptr_t)a.P & 15))) return OPS_AMD_ERR_INVALID_ARG;
  }
  if (a.loss_finish_rows < 0 || a.loss_finish_rows > ML_MAXG || (a.loss_finish_rows > 0 && (!a.loss_ws || !a.loss || !a.alpha || a.loss_C < 1)))
    return OPS_AMD_ERR_INVALID_ARG;
  int nside = 0;
  if (a.add_mode != OPS_MLP_ADD_NONE || a.side != OPS_MLP_SIDE_NONE) {
    if (!a.Ot || a.No < 1 || !a.conv_w || !a.conv_b || !a.sgamma || !a.sbeta || !a.ssave || !a.spart) return OPS_AMD_ERR_INVALID_ARG;
    if (a.No > MB_MAX_SIDE * MB_SIDE_COLS) return OPS_AMD_ERR_UNSUPPORTED;
    if (a.add_mode == OPS_MLP_ADD_FWD_BLOCK && (bwd || a.No != a.N || !a.srunning_mean || !a.srunning_var)) return OPS_AMD_ERR_INVALID_ARG;
    if (a.add_mode == OPS_MLP_ADD_BWD_BLOCK && (!bwd || a.No != a.N || !a.dZt || !a.sdparams)) return OPS_AMD_ERR_INVALID_ARG;
    if (a.side == OPS_MLP_SIDE_BWD_STENCIL_SUMS && !a.dZt) return OPS_AMD_ERR_INVALID_ARG;
    if (a.side != OPS_MLP_SIDE_NONE) {
      if (a.add_mode != OPS_MLP_ADD_NONE) return OPS_AMD_ERR_INVALID_ARG;      // one workspace: a launch either fills or reads it
      nside = (a.No + MB_SIDE_COLS - 1) / MB_SIDE_COLS;
    }
  }
  if (a.n_slots < 0 || a.n_slots > 65535) return OPS_AMD_ERR_INVALID_ARG;
  if (a.n_slots > 0) {      // evaluation slots: no batch statistics are written, no side jobs, the last slot holds the remainder
    if (bwd || nside || a.p_drop > 0.0f || (bn && !a.eval_stats) || (a.add_mode != OPS_MLP_ADD_NONE && !a.eval_stats)) return OPS_AMD_ERR_INVALID_ARG;
    if (a.slot_stride < 0 || a.slot_stride % 16 || a.slot_total_rows <= a.B * (a.n_slots - 1) || a.slot_total_rows > a.B * a.n_slots)
      return OPS_AMD_ERR_INVALID_ARG;
  }
  const dim3 grid((unsigned)(nstrips + nside), (unsigned)(a.n_slots > 0 ? a.n_slots : 1)), block(MB_THREADS);
  hipStream_t s = (hipStream_t)stream;
  // the combinations of the PINN step (pinn_fused.py _build) are compiled with their mode folded in; anything else takes the generic kernel
  const int key = a.tail * 100 + a.add_mode * 10 + (a.eval_stats ? 1 : 0);
#ifdef OPS_MLP_GENERIC_ONLY
  const int use_key = -1;
#else
  const int use_key = key;
#endif
#define MB_LAUNCH(T_, A_, E_) hipLaunchKernelGGL((mlp_strip_kernel<T_, A_, E_>), grid, block, 0, s, a, nstrips)
  switch (use_key) {
    case OPS_MLP_TAIL_BN_ACT_DROP * 100 + 0: MB_LAUNCH(OPS_MLP_TAIL_BN_ACT_DROP, OPS_MLP_ADD_NONE, 0); break;
    case OPS_MLP_TAIL_ACT_DROP * 100 + 0: MB_LAUNCH(OPS_MLP_TAIL_ACT_DROP, OPS_MLP_ADD_NONE, 0); break;
    case OPS_MLP_TAIL_BN * 100 + OPS_MLP_ADD_FWD_BLOCK * 10 + 0: MB_LAUNCH(OPS_MLP_TAIL_BN, OPS_MLP_ADD_FWD_BLOCK, 0); break;
    case OPS_MLP_TAIL_LOSS * 100 + 0: MB_LAUNCH(OPS_MLP_TAIL_LOSS, OPS_MLP_ADD_NONE, 0); break;
    case OPS_MLP_TAIL_BWD_BN * 100 + 0: MB_LAUNCH(OPS_MLP_TAIL_BWD_BN, OPS_MLP_ADD_NONE, 0); break;
    case OPS_MLP_TAIL_BWD_ACT_DROP * 100 + 0: MB_LAUNCH(OPS_MLP_TAIL_BWD_ACT_DROP, OPS_MLP_ADD_NONE, 0); break;
    case OPS_MLP_TAIL_BWD_BN * 100 + OPS_MLP_ADD_BWD_BLOCK * 10 + 0: MB_LAUNCH(OPS_MLP_TAIL_BWD_BN, OPS_MLP_ADD_BWD_BLOCK, 0); break;
    case OPS_MLP_TAIL_BWD_BN_ACT_DROP * 100 + OPS_MLP_ADD_BWD_BLOCK * 10 + 0: MB_LAUNCH(OPS_MLP_TAIL_BWD_BN_ACT_DROP, OPS_MLP_ADD_BWD_BLOCK, 0); break;
    case OPS_MLP_TAIL_BN_ACT_DROP * 100 + 1: MB_LAUNCH(OPS_MLP_TAIL_BN_ACT_DROP, OPS_MLP_ADD_NONE, 1); break;
    case OPS_MLP_TAIL_ACT_DROP * 100 + 1: MB_LAUNCH(OPS_MLP_TAIL_ACT_DROP, OPS_MLP_ADD_NONE, 1); break;
    case OPS_MLP_TAIL_BN * 100 + OPS_MLP_ADD_FWD_BLOCK * 10 + 1: MB_LAUNCH(OPS_MLP_TAIL_BN, OPS_MLP_ADD_FWD_BLOCK, 1); break;
    case OPS_MLP_TAIL_LOSS * 100 + 1: MB_LAUNCH(OPS_MLP_TAIL_LOSS, OPS_MLP_ADD_NONE, 1); break;
    default: MB_LAUNCH(-1, 0, 0); break;
  }
#undef MB_LAUNCH
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}

static int wgrad_launch(int nprob, const ops_mlp_wgrad_problem* problems, const WgradNorm* nm, int32_t* nparts, void* stream) {
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
  if (nm) {
    if (tiles + nm->nrange > OPS_FLAT_ADAM_MAX_PARTS) return OPS_AMD_ERR_UNSUPPORTED;
    *nparts = tiles + nm->nrange;
    hipLaunchKernelGGL(mlp_wgrad_kernel<true>, dim3((unsigned)(tiles + nm->nrange)), dim3(64), 0, (hipStream_t)stream, tb, *nm);
  } else {
    hipLaunchKernelGGL(mlp_wgrad_kernel<false>, dim3((unsigned)tiles), dim3(64), 0, (hipStream_t)stream, tb, WgradNorm{});
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}

extern "C" int ops_mlp_wgrad_group(int nprob, const ops_mlp_wgrad_problem* problems, void* stream) {
  return wgrad_launch(nprob, problems, nullptr, nullptr, stream);
}

extern "C" int ops_mlp_wgrad_group_norm(int nprob, const ops_mlp_wgrad_problem* problems, int nranges, const float* const* range_ptr,
                                        const int32_t* range_len, float grad_scale, void* workspace, int32_t* step, float beta1, float beta2,
                                        int32_t* nparts, void* stream) {
  if (nranges < 0 || nranges > OPS_MLP_MAX_NORM_RANGES || (nranges > 0 && (!range_ptr || !range_len)) || !workspace || !step || !nparts)
    return OPS_AMD_ERR_INVALID_ARG;
  WgradNorm nm{};
  nm.part = (double*)workspace; nm.step = step; nm.scale = grad_scale; nm.beta1 = beta1; nm.beta2 = beta2; nm.nrange = nranges;
  for (int i = 0; i < nranges; ++i) {
    if (!range_ptr[i] || range_len[i] < 1) return OPS_AMD_ERR_INVALID_ARG;
    nm.rptr[i] = range_ptr[i]; nm.rlen[i] = range_len[i];
  }
  return wgrad_launch(nprob, problems, &nm, nparts, stream);
}

extern "C" int ops_mlp_repack_weights(int nmat, const ops_mlp_repack_entry* entries, void* stream) {
  if (nmat < 1 || nmat > OPS_MLP_MAX_WGRAD || !entries) return OPS_AMD_ERR_INVALID_ARG;
  RepackTable tb;
  tb.nmat = nmat;
  long tot = 0;
  for (int i = 0; i < nmat; ++i) {
    const ops_mlp_repack_entry& m = entries[i];
    if (!m.W || !m.Wp || !m.Wtp || m.N < 1 || m.K < 1 || m.ldw % 32 || m.ldwt % 32 || m.ldw < ru(m.K, 32) || m.ldwt < ru(m.N, 32)) return OPS_AMD_ERR_INVALID_ARG;
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

static int gather_launch(int B, int F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                         unsigned long long* counter, void* out, int ld, void* out_t, const float* Y, int C, float* targets_t,
                         const float* params, const AdamRepack* rp, void* stream) {
  if (B < 1 || B > MB_ROWS || F < 1 || !X || !idx || !out || !out_t || ld % 32 || ld < ru(F, 32)) return OPS_AMD_ERR_INVALID_ARG;
  if (Y && (C < 1 || !targets_t)) return OPS_AMD_ERR_INVALID_ARG;
  const int nfb = (F + 31) / 32, ntb = Y ? (C + 31) / 32 : 0, ngb = (nfb + ntb) * MB_RSPLIT;      // assembly workgroups: (tile, row group)
  if (rp) {
    const TileJobs tj = make_tile_jobs(*rp);
    const int nrb = (tj.first[2 * rp->nmat] + 3) / 4;
    hipLaunchKernelGGL(mlp_gather_noise_kernel<true>, dim3((unsigned)(ngb + nrb)), dim3(256), 0, (hipStream_t)stream, B, F, X, idx, sigma,
                       seed, counter, (uint16_t*)out, ld, (uint16_t*)out_t, nfb, Y, C, targets_t, ngb, params, *rp, tj);
  } else {
    hipLaunchKernelGGL(mlp_gather_noise_kernel<false>, dim3((unsigned)ngb), dim3(256), 0, (hipStream_t)stream, B, F, X, idx, sigma, seed,
                       counter, (uint16_t*)out, ld, (uint16_t*)out_t, nfb, Y, C, targets_t, ngb, nullptr, AdamRepack{}, TileJobs{});
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
  return OPS_AMD_OK;
}

extern "C" int ops_mlp_gather_noise(int B, int F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                    unsigned long long* counter, void* out, int ld, void* out_t, const float* Y, int C, float* targets_t,
                                    void* stream) {
  return gather_launch(B, F, X, idx, sigma, seed, counter, out, ld, out_t, Y, C, targets_t, nullptr, nullptr, stream);
}

extern "C" int ops_mlp_gather_noise_repack(int B, int F, const float* X, const long long* idx, const float* sigma, unsigned long long seed,
                                           unsigned long long* counter, void* out, int ld, void* out_t, const float* Y, int C,
                                           float* targets_t, long n_params, const float* params, int nmat, const ops_mlp_repack_entry* entries,
                                           void* stream) {
  AdamRepack rp{};
  const int rc = make_adam_repack(n_params, params, nmat, entries, &rp);
  if (rc != OPS_AMD_OK) return rc;
  return gather_launch(B, F, X, idx, sigma, seed, counter, out, ld, out_t, Y, C, targets_t, params, &rp, stream);
}

extern "C" size_t ops_mlp_loss_workspace_bytes(void) { return (size_t)ML_MAXG * 5 * sizeof(double); }
