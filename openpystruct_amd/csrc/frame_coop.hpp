// Batched 2-D frame solve, LATENCY kernel (r06, late): ONE WORKGROUP OF FOUR WAVES PER FRAME, the register window split by COLUMNS.
// (included by frame_solve.hip behind frame_wave.hpp: FwPlan, the parking-area layout, fw_readlane, fw_backward come from there)
//
// For the reference's own use of the frame solve -- ONE frame per epoch (/root/reference/OpenPyStruct_FrameOpt_Discrete_Beta.py:178-183) -- and
// every batch too small to fill the chip, what counts is how long ONE frame takes.  A lone wave of the wave-per-frame kernel needs ~0.6 us
// per column (its ~90 vector instructions of a step issue one after the other, nothing else runs on its SIMD); the r01 workgroup kernels
// ~0.31 us while their band (n x (kd + 4) doubles) is LDS-resident, ~0.6 us through their LDS ring when it is not (BASELINE config 5: 15 x 16).
// Here the four waves of a workgroup sit on the four SIMDs of one CU and share one frame, the window in REGISTERS whatever the frame's size:
//
//   * same arithmetic and same row mapping as frame_wave.hpp (column-by-column band LDL^T = dpbsv's, FR:134; row R in lane R mod 64 of EVERY
//     wave), but wave q holds only the window columns whose slot (C mod W) lies in [q W/4, (q+1) W/4): W/4 multiply-adds per step and wave;
//   * step j: the wave that owns column j + 1 updates it first and publishes it -- pivot d_(j+1) at index 0, A[j+1+rel][j+1] at index rel --
//     as the LDS line of the next step; everybody reads pivot, own-row entry (one per-lane read) and its W/4 column entries from the line of
//     step j; ONE workgroup barrier per step (two line buffers);
//   * wave 0 also carries the right-hand side (forward substitution, w_j = z_j / d_j into LDS), wave 3 stores column j of L to the workspace;
//   * rows enter in groups of eight through the same parking area and assembly plan as frame_wave.hpp, built by all 256 threads;
//   * backward substitution: wave 0 alone (fw_backward), results by all 256 threads.
// Measured (profiles/r06_notes.md 22): 0.33-0.39 us per equation -- each wave's ~35 instructions per column issue one behind the other; two
// columns per barrier changed nothing.  That equals the r01 kernels where their band is resident, so the dispatch (frame_solve.hip
// frame_family) takes this kernel where it is not (15 x 16 x 1: 455 -> 294 us) or where the r01 workgroup fits a CU only once and the batch is
// beyond one frame per CU (10 x 10 x 512: 192 -> 145 us).
#pragma once

namespace opsamd {

constexpr int FC_NW = 4;
constexpr int FC_LINE = 72;                                   // doubles per line buffer (index rel <= 63)
__host__ __device__ inline int fc_width(int kd) { return kd < 20 ? 20 : kd < 36 ? 36 : kd < 52 ? 52 : 56; }      // compiled window widths (multiples of four)
__host__ __device__ inline size_t fc_frame_doubles(int n, int kd) { return (size_t)(n + 4) * fc_width(kd); }      // column j of L at [j * W, j * W + kd)
__host__ __device__ inline size_t fc_lds_doubles(int n, int W) { return ((2 * FC_LINE + (size_t)FW_G * fw_pitch(W) + (size_t)(n + 64)) + 1) & ~(size_t)1; }

// workgroup barrier that waits for this wave's LDS operations only: __syncthreads() also drains vmcnt, i.e. it would wait every step for wave 3's
// store of a column of L (and for the plan words in flight) to come back from L2
__device__ __forceinline__ void fc_barrier() {
  __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int W, int Q, int S>
__device__ __forceinline__ void fc_step(double (&reg)[W / FC_NW], double& y, int j, int lane, int n, int kd, double* __restrict__ line,
                                        double* __restrict__ rows, double* __restrict__ xs, int& bad) {
  constexpr int WQ = W / FC_NW, S1 = (S + 1) % W;
  const int rel = (lane - j) & 63, R = j + rel;
  const bool inwin = rel >= 1 && rel <= kd && R < n;
  const double* cb = line + (j & 1) * FC_LINE;
  const double d = cb[0];
  const double rdj = frcp(d);
  bad |= (j < n) & !(d > 0.0);
  const double a = cb[rel];                                   // own row's entry of column j (zero outside the window: the publisher masks)
  const double l = inwin ? a * rdj : 0.0;
  if constexpr (S1 / WQ == Q) {                               // this wave owns column j + 1: final after one multiply-add, published for the next step
    constexpr int c1 = S1 % WQ;
    reg[c1] = __builtin_fma(-l, cb[1], reg[c1]);
    const int rel1 = (lane - (j + 1)) & 63;
    const bool in1 = rel1 <= kd && j + 1 + rel1 < n;          // rel1 = 0: the pivot
    (line + ((j + 1) & 1) * FC_LINE)[rel1] = in1 ? reg[c1] : 0.0;
  }
  if constexpr (Q == 0) {                                     // right-hand side: the pivot row's is final
    const double zj = fw_readlane(y, j & 63);
    if (rel == 0 && j < n) xs[j] = zj * rdj;                  // w_j = z_j / d_j
    y = __builtin_fma(-l, zj, y);
  }
  if constexpr (Q == FC_NW - 1) {
    if (inwin) rows[(size_t)j * W + (rel - 1)] = l;           // column j of L: one coalesced store
  }
#pragma unroll
  for (int c = 0; c < WQ; ++c) {
    const int t = (Q * WQ + c - S + W) % W;                   // column j + t sits in this register
    if (t >= 2) reg[c] = __builtin_fma(-l, cb[t], reg[c]);    // (t = 0: column j itself, finished; t = 1: done above by its owner)
  }
  fc_barrier();                                               // line j + 1 is complete; line j may be overwritten from the next step on
}

template <int W, int Q>
__device__ __forceinline__ void frame_coop_body(const FrameParams& p, double* __restrict__ rows, double* __restrict__ lds, int lane, long b,
                                                const FwPlan& pl) {
  constexpr int G = FW_G, WQ = W / FC_NW, PITCH = fw_pitch(W);
  static_assert(W % (4 * FC_NW) == 0 || W % 4 == 0, "frame_coop: window widths are multiples of four");
  const int n = p.n_eq, kd = p.kd, tid = Q * 64 + lane;
  const int KG = (kd / G + 1) * G;                            // registers hold the rows below j + KG + G at step j
  double* line = lds;                                         // [2][FC_LINE]
  double* stage = lds + 2 * FC_LINE;                          // [G][PITCH]
  double* xs = stage + (size_t)G * PITCH;                     // [n + 64]: w, then x
  double reg[WQ];
#pragma unroll
  for (int c = 0; c < WQ; ++c) reg[c] = 0.0;
  double y = 0.0;
  int bad = 0;

  // fused assembly (plan of frame_wave.hpp: 192 entry slots per group of eight rows): thread tid < 192 carries ONE entry of the next group
  const double* Ib = p.I + b * p.Ne;
  const double* lb = p.loads + b * p.loads_bs;
  const int ti = tid < FW_EPG ? tid : 0;
  // software pipeline over the row groups: the entry WORD of a group is requested two boundaries ahead, its inertia / coefficients / right-hand
  // side one boundary ahead (they need the word) -- a boundary never waits for a round trip to L2, and no barrier drains vmcnt
  unsigned eB = 0u, eN = 0u;
  int dofB = 0, dofN = 0, gB = 0;
  double bi = 0.0, ba = 0.0, bb = 0.0, by1 = 0.0, by2 = 0.0;
  const int n_extra = pl.hdr[0];
  auto ents_next = [&](int g0) {                              // words of group g0 -> (eN, dofN)
    const int gi = g0 / G < pl.ng ? g0 / G : pl.ng;
    eN = pl.ent[(size_t)gi * FW_EPG + ti];
    const int r = g0 + (tid < G ? tid : 0);
    dofN = pl.eq_dof[r < n ? r : n];
  };
  auto build_issue = [&]() {                                  // values of group gB (its words are in eB / dofB)
    const int gi = gB / G < pl.ng ? gB / G : pl.ng;
    bi = Ib[(eB >> FW_SLOT_BITS) & 0x1FFFFF];
    ba = pl.ka[(size_t)gi * FW_EPG + ti];
    bb = pl.kb[(size_t)gi * FW_EPG + ti];
    const int r = gB + (tid < G ? tid : 0);
    by1 = pl.rhs_base[r < n ? r : n];
    by2 = lb[dofB];
  };
  auto advance = [&]() {                                      // the group after gB becomes gB: request its values and the following group's words
    eB = eN; dofB = dofN; gB += G;
    build_issue();
    ents_next(gB + G);
  };
  auto zero_stage = [&]() {
    for (int i = tid; i < G * PITCH; i += 64 * FC_NW) stage[i] = 0.0;
  };
  auto build_finish = [&]() {                                 // (behind a barrier after zero_stage)
    if (tid < FW_EPG && (int)eB < 0) atomicAdd(&stage[eB & FW_SLOT_MASK], __builtin_fma(bi, bb, ba));
    if (n_extra != 0) {                                       // nodes with more than four elements: extra blocks, not prefetched
      const int gi = gB / G < pl.ng ? gB / G : pl.ng;
      for (int blk = pl.xstart[gi]; blk < pl.xstart[gi + 1]; ++blk)
        for (int i0 = tid; i0 < FW_EPG; i0 += 64 * FC_NW) {
          const size_t i = (size_t)(pl.ng + 1 + blk) * FW_EPG + i0;
          const unsigned w = pl.ent[i];
          if ((int)w < 0) atomicAdd(&stage[w & FW_SLOT_MASK], __builtin_fma(Ib[(w >> FW_SLOT_BITS) & 0x1FFFFF], pl.kb[i], pl.ka[i]));
        }
    }
    if (tid < G) stage[tid * PITCH + W] = (gB + tid < n) ? by1 + by2 : 0.0;
  };
  auto take_group = [&](int g0) {                             // rows g0 .. g0 + G - 1: this wave's columns (wave 0: + the right-hand side)
    const int slot = (lane - g0) & 63;
    if (slot < G) {
      const double* q = stage + (size_t)slot * PITCH + Q * WQ;
#pragma unroll
      for (int c = 0; c < WQ; ++c) reg[c] = q[c];
      if constexpr (Q == 0) y = stage[(size_t)slot * PITCH + W];
    }
  };
  // prologue: rows [0, KG + G) into registers, the next group parked, the one after on its way
  ents_next(0);
  eB = eN; dofB = dofN; gB = 0;
  build_issue();
  ents_next(G);
  for (int g0 = 0; g0 < KG + 2 * G; g0 += G) {
    zero_stage();
    fc_barrier();
    build_finish();                                           // group g0 = gB
    advance();
    fc_barrier();
    if (g0 < KG + G) { take_group(g0); fc_barrier(); }
  }

  // line of column 0: slot 0 belongs to wave 0
  if constexpr (Q == 0) {
    const int rel0 = lane;
    line[rel0] = (rel0 <= kd && rel0 < n) ? reg[0] : 0.0;
  }
  fc_barrier();

  for (int j0 = 0; j0 < n; j0 += W) {
    auto boundary = [&](int j) {                              // j % G == 0, j > 0: rows [j + KG, j + KG + G) enter
      take_group(j + KG);
      fc_barrier();
      zero_stage();
      fc_barrier();
      build_finish();                                         // group j + KG + G = gB (requested a boundary ago; read eight steps -- eight barriers -- later)
      advance();
    };
#define FC_STEP(S_)                                                                   \
    {                                                                                 \
      const int j = j0 + (S_);                                                        \
      if constexpr ((S_) % 4 == 0) if (j > 0 && (j % G) == 0 && j < n) boundary(j);   \
      fc_step<W, Q, (S_)>(reg, y, j, lane, n, kd, line, rows, xs, bad);               \
    }
#define FC_STEP4(S_)                                                                  \
    if constexpr ((S_) < W) {                                                         \
      if (j0 + (S_) < n) { FC_STEP(S_) FC_STEP(S_ + 1) FC_STEP(S_ + 2) FC_STEP(S_ + 3) } \
    }
#define FC_STEP8(S_) FC_STEP4(S_) FC_STEP4(S_ + 4)
    static_assert(W % 4 == 0 && W <= 56, "frame_coop: window widths are multiples of four");
    FC_STEP8(0) FC_STEP8(8) FC_STEP8(16) FC_STEP8(24) FC_STEP8(32) FC_STEP8(40) FC_STEP8(48)
#undef FC_STEP4
#undef FC_STEP8
#undef FC_STEP
  }
  __syncthreads();                                            // wave 3's stores of L (and wave 0's w) are visible to wave 0
  if constexpr (Q == 0) fw_backward<W>(rows, xs, n, kd, lane);
  __syncthreads();
  write_results(p, b, xs, bad != 0, tid, 64 * FC_NW);
}

template <int W>
__global__ __launch_bounds__(64 * FC_NW) void frame_coop_kernel(const FrameParams p, double* __restrict__ ws, const FwPlan pl) {
  extern __shared__ double lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long b = blockIdx.x;
  double* rows = ws + b * fc_frame_doubles(p.n_eq, p.kd);
  // every wave runs its own instantiation (the slots it owns are compile-time constants); all of them pass the same barriers
  switch (wave) {
    case 0: frame_coop_body<W, 0>(p, rows, lds, lane, b, pl); break;
    case 1: frame_coop_body<W, 1>(p, rows, lds, lane, b, pl); break;
    case 2: frame_coop_body<W, 2>(p, rows, lds, lane, b, pl); break;
    default: frame_coop_body<W, 3>(p, rows, lds, lane, b, pl); break;
  }
}

}  // namespace opsamd
