// Batched 2-D frame solve, second generation: ONE WAVEFRONT PER FRAME, the band window resident in REGISTERS.
// (included by frame_solve.hip; FrameParams, frcp, elem_global_k, write_results come from there)
//
// Same arithmetic as the band solver the reference selects for `setup_frame_model`
// (/root/reference/OpenPyStruct_FrameOpt_Discrete_Beta.py:134 `system('BandGeneral')` on an SPD matrix; restated as the
// column-by-column band LDL^T that LAPACK dpbsv performs): pivots in equation order, no pivoting.
//
// Why: the r01 kernels (one 320..1024-thread workgroup per frame, band in LDS, three columns per workgroup barrier) ran at
// ~2 % of the FP64 vector rate: every window entry cost four LDS operations per fused multiply-add and every block step a
// ~1 400-cycle serial chain behind a barrier.  Here a frame is a single wave, so there are no barriers at all, and the
// kd x kd window never touches LDS:
//
//   * lane (R mod 64) OWNS ROW R of the band while it is inside the window (kd <= 55 < 64 rows are in flight at a time);
//     entry A[R][C] sits in that lane's register  reg[C mod W]  (W >= kd + 1, a compile-time constant): the register
//     index is the same for every lane, so the kernel is unrolled W-fold over (column mod W) and needs no dynamic register
//     indexing;
//   * step j: every window lane scales its own entry (l = A[R][j] / d_j), the unscaled column goes through a 64-entry LDS
//     line as a BROADCAST (one ds_write + kd/2 conflict-free 16-byte broadcast reads), and each lane updates its row:
//     reg[C mod W] -= l * A[C][j]  -- one FMA per column of the window, operands in registers.  The line of step j + 1 and
//     the reciprocal of its pivot (one v_readlane) are produced DURING step j, right after column j + 1 has taken its update
//     (fw_prepare), so neither the reciprocal chain nor the LDS round trip sits between two steps;
//   * forward substitution rides along (one readlane + one FMA); column j of L leaves with ONE coalesced store into the
//     per-frame HBM workspace;
//   * rows enter the window in groups of 8 through an LDS parking area: the group is ASSEMBLED there (fused assembly: plan
//     blocks below, ds_add_f64) one boundary before its owners' lanes fall free and take it;
//   * backward substitution: blocked dot form, eight columns per pass (8 x 8 lanes: strided runs of eight columns of L
//     against x from LDS, 8-lane DPP sums, the block's own 8 x 8 triangle as a readlane + FMA chain).
//
// Bound (r03 phase ablation, 15 x 16: 4.0 ms per 12 288 frames): the elimination steps 2.5 ms = VALU issue (82 VALU + 25 LDS
// instructions per step and wave, 54 of them FP64 FMAs, three waves per SIMD), group entry 0.5, L stores 0.3, backward sweep
// 0.7 ms.  HBM: the factor workspace is written once and read once (2 n W 8 bytes per frame: 15 x 16 0.64 MB) -- 2 TB/s, not
// the bound.  No MFMA: the update is rank-1 per column on a window that slides by one column per step; a
// 16-column panel (what v_mfma_f64_16x16x4_f64 tiles would need to keep their tile <-> lane mapping fixed) is as wide as
// the whole band of the reference's frames (kd = 3 (bays + 1) + 2 <= 35) and its panel factorisation is the serial part
// (and the FP64 matrix rate of the MI355X equals its vector rate).
#pragma once

namespace opsamd {

constexpr int FW_G = 8;     // rows per load group
__host__ __device__ constexpr int fw_pitch(int W) { return W + 2; }   // doubles per parked row: W entries + right-hand side, even (16-byte reads)

__host__ __device__ inline int fw_width(int kd) {           // compiled register-window widths (half bandwidths below 28: frame_pack.hpp)
  return kd < 36 ? 36 : kd < 52 ? 52 : 56;
}
__host__ __device__ inline size_t fw_frame_doubles(int n, int kd) { return (size_t)(n + 4) * fw_width(kd); }      // column j of L at [j * W, j * W + kd)
constexpr int FW_CB = 72;   // broadcast line: entry `rel` at index rel (pairs (t, t + 1), t even, are 16-byte aligned), two buffers
__host__ __device__ inline size_t fw_lds_doubles(int n, int W) { return ((2 * FW_CB + (size_t)FW_G * fw_pitch(W) + (size_t)(n + 64)) + 1) & ~(size_t)1; }

__device__ __forceinline__ double fw_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// DPP move of a double (quad_perm, row_half_mirror, ...: no LDS)
template <int CTRL>
__device__ __forceinline__ double fw_dpp(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// LDS operations of one wave execute in order; this pins the compiler and lands earlier reads
__device__ __forceinline__ void fw_fence() {
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ---- assembly plan: the topology-only part of the assembly, built once per call ----
// Row R of the band is  sum over its incident element entries (e, r, q):  ka[e][r][q] + I_e * kb[e][r][q]  at column slot
// eq[q] mod W (ka: the axial part of the rotated ElasticBeam2d matrix, kb: the bending part per unit inertia -- both
// independent of the frame), and its right-hand side is rhs_base[R] (consistent beamUniform loads, FR:131) + the nodal load
// of its DOF.  With the plan the solve kernel builds every 8-row group directly in its LDS parking area one group ahead of
// need: the assembled band never exists in HBM (r02 first version: assembly kernel writes it, solve kernel reads it back =
// half of the solve's memory traffic and 20 % of its time).
//
// Layout (r03): the entries of row group gi (rows 8 gi .. 8 gi + 7) are a fixed block of FW_EPG = 192 slots -- entry word
// (valid | element << 9 | parking slot) and its two coefficients at the SAME index -- zero-padded, so the solve kernel's loads are
// coalesced, unconditional and depend on nothing but the group number: one entry-word load a group ahead, then ONE round
// trip (inertia gather + coefficients) per group.  (r02: entry -> element -> three gathers behind per-row pointers = seven
// serialised round trips per group, 17 % of the solve.)  A group with more than 192 entries (nodes with more than four
// elements) continues in extra blocks; a padded all-zero block serves the groups past the last equation.
constexpr int FW_SLOT_BITS = 10;          // entry word: bit 31 valid | element << FW_SLOT_BITS | parking slot
constexpr unsigned FW_SLOT_MASK = (1u << FW_SLOT_BITS) - 1u;
constexpr int FW_KE = 3;                  // entry words a lane carries per group
constexpr int FW_EPG = 64 * FW_KE;        // entries per block
struct FwPlan {
  const int* hdr;            // [4]   [0] = number of extra blocks (0 for the reference's grid frames)
  const int* eq_dof;         // [n + 1]   index into loads[Nn*3]; [n] = 0
  const int* xstart;         // [ng + 2]  first extra block of a group (prefix sums); groups past the end: no extra blocks
  const unsigned* ent;       // [nblk][FW_EPG]  bit 31 (valid) | element << FW_SLOT_BITS | parking slot: (row in group) * fw_pitch(W) + column slot
  const double* ka;          // [nblk][FW_EPG]
  const double* kb;          // [nblk][FW_EPG]
  const double* rhs_base;    // [n + 1]; [n] = 0
  int ng;                    // row groups; block ng is the all-zero block
};
// (G rows per group, EPG entry slots per block: 8 / 192 for the wave-per-frame kernel, 4 / 96 or 8 / 192 for the packed kernel, frame_pack.hpp)
__host__ __device__ inline int fw_groups(int n, int G = FW_G) { return (n + G - 1) / G; }
__host__ __device__ inline size_t fw_plan_blocks(int n, int Ne, int G = FW_G, int EPG = FW_EPG) { return (size_t)fw_groups(n, G) + 1 + ((size_t)Ne * 21 + EPG - 1) / EPG; }
__host__ __device__ inline size_t fw_plan_bytes(int n, int Ne, int G = FW_G, int EPG = FW_EPG) {
  const size_t ints = 4 + (size_t)(n + 1) + (size_t)(fw_groups(n, G) + 2);
  return ((ints + 1) / 2) * 8 + fw_plan_blocks(n, Ne, G, EPG) * EPG * (4 + 8 + 8) + (size_t)(n + 1) * 8;
}
__host__ __device__ inline FwPlan fw_plan_at(void* base, int n, int Ne, int G = FW_G, int EPG = FW_EPG) {
  char* q = (char*)base;
  const size_t nblk = fw_plan_blocks(n, Ne, G, EPG);
  FwPlan pl;
  pl.ng = fw_groups(n, G);
  pl.ka = (const double*)q;        q += nblk * EPG * 8;
  pl.kb = (const double*)q;        q += nblk * EPG * 8;
  pl.rhs_base = (const double*)q;  q += (size_t)(n + 1) * 8;
  pl.ent = (const unsigned*)q;     q += nblk * EPG * 4;
  pl.hdr = (const int*)q;          q += 4 * 4;
  pl.eq_dof = (const int*)q;       q += (size_t)(n + 1) * 4;
  pl.xstart = (const int*)q;
  return pl;
}

// one workgroup; LDS: 3 * (ng + 2) ints
__global__ __launch_bounds__(1024) void frame_plan_kernel(const FrameParams p, int W, void* plan_base, int G, int EPG) {
  extern __shared__ int s_plan[];
  const FwPlan pl = fw_plan_at(plan_base, p.n_eq, p.Ne, G, EPG);
  const int n = p.n_eq, ng = pl.ng, tid = threadIdx.x, T = blockDim.x;
  int* cnt = s_plan;                  // [ng + 1] entries per group
  int* cur = s_plan + (ng + 2);       // [ng + 1] fill cursors
  int* xs_ = s_plan + 2 * (ng + 2);   // [ng + 2] first extra block per group
  int* hdr = const_cast<int*>(pl.hdr);
  int* eq_dof = const_cast<int*>(pl.eq_dof);
  int* xstart = const_cast<int*>(pl.xstart);
  unsigned* ent = const_cast<unsigned*>(pl.ent);
  double* ka = const_cast<double*>(pl.ka);
  double* kb = const_cast<double*>(pl.kb);
  double* rhs_base = const_cast<double*>(pl.rhs_base);
  for (int i = tid; i <= ng; i += T) { cnt[i] = 0; cur[i] = 0; }
  for (int i = tid; i <= n; i += T) { rhs_base[i] = 0.0; eq_dof[i] = 0; }
  __syncthreads();
  for (int i = tid; i < p.Nn * 3; i += T) { const int q = p.node_eq[i]; if (q >= 0) eq_dof[q] = i; }
  // pass A: entries per group, consistent nodal loads of the element loads
  for (int e = tid; e < p.Ne; e += T) {
    const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
    const double wy = p.elem_w[2 * e], wx = p.elem_w[2 * e + 1];
    const double pl6[6] = {wx * L / 2, wy * L / 2, wy * L * L / 12, wx * L / 2, wy * L / 2, -wy * L * L / 12};
    const double pg[6] = {c * pl6[0] - s * pl6[1], s * pl6[0] + c * pl6[1], pl6[2], c * pl6[3] - s * pl6[4], s * pl6[3] + c * pl6[4], pl6[5]};
    for (int r = 0; r < 6; ++r) {
      const int er = p.elem_eq[6 * e + r];
      if (er < 0) continue;
      atomicAdd(&rhs_base[er], pg[r]);
      for (int q = 0; q < 6; ++q) { const int eq = p.elem_eq[6 * e + q]; if (eq >= 0 && eq <= er) atomicAdd(&cnt[er / G], 1); }
    }
  }
  // rows between the last equation and the end of its group: unit diagonal (a pivot there divides nothing by zero; the row-per-lane
  // kernel masks these rows anyway)
  if (tid < G && n + tid < G * ng) atomicAdd(&cnt[(n + tid) / G], 1);
  __syncthreads();
  if (tid == 0) {                            // extra blocks per group: exclusive scan (ng <= a few hundred)
    int acc = 0;
    for (int g = 0; g < ng; ++g) { xs_[g] = acc; acc += cnt[g] > EPG ? (cnt[g] - 1) / EPG : 0; }
    xs_[ng] = acc; xs_[ng + 1] = acc;
    hdr[0] = acc; hdr[1] = hdr[2] = hdr[3] = 0;
  }
  __syncthreads();
  const int nblk = ng + 1 + xs_[ng];
  for (int i = tid; i < ng + 2; i += T) xstart[i] = xs_[i];
  for (long i = tid; i < (long)nblk * EPG; i += T) { ent[i] = 0u; ka[i] = 0.0; kb[i] = 0.0; }
  __syncthreads();
  // pass B: fill (the order inside a group is the order of arrival: the parking area accumulates with LDS atomics anyway)
  for (int e = tid; e < p.Ne; e += T) {
    const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
    double k_a[6][6], k_b[6][6];
    elem_global_k(L, c, s, p.elem_EA[e], 0.0, k_a);
    elem_global_k(L, c, s, 0.0, p.elem_E[e], k_b);
    for (int r = 0; r < 6; ++r) {
      const int er = p.elem_eq[6 * e + r];
      if (er < 0) continue;
      const int g = er / G;
      for (int q = 0; q < 6; ++q) {
        const int eq = p.elem_eq[6 * e + q];
        if (eq >= 0 && eq <= er) {
          const int pos = atomicAdd(&cur[g], 1), blk = pos / EPG;
          const long idx = (long)(blk == 0 ? g : ng + 1 + xs_[g] + blk - 1) * EPG + pos % EPG;
          const int slot = (er % G) * fw_pitch(W) + eq % W;
          ent[idx] = 0x80000000u | ((unsigned)e << FW_SLOT_BITS) | (unsigned)slot;
          ka[idx] = k_a[r][q];
          kb[idx] = k_b[r][q];
        }
      }
    }
  }
  if (tid < G && n + tid < G * ng) {
    const int er = n + tid, g = er / G;
    const int pos = atomicAdd(&cur[g], 1), blk = pos / EPG;
    const long idx = (long)(blk == 0 ? g : ng + 1 + xs_[g] + blk - 1) * EPG + pos % EPG;
    const int slot = (er % G) * fw_pitch(W) + er % W;
    ent[idx] = 0x80000000u | (unsigned)slot;                 // (element 0's inertia times kb = 0)
    ka[idx] = 1.0;
    kb[idx] = 0.0;
  }
}

// ---- the solve: one wave per frame ----
#ifndef FW_NO_BOUNDARY                                     // (phase ablation builds: no rows enter after the prologue)
#define FW_NO_BOUNDARY 0
#endif
template <int W>
struct FwState {
  double reg[W];     // own row: A[R][C] at index C mod W
  double y;          // own right-hand side (forward), own w / x (backward)
};

// what step j needs from outside its own arithmetic, made one step AHEAD (the wave is latency-bound: reciprocal chain, LDS
// round trip of the broadcast line): column j is final in every lane's reg[S] once step j - 1 has updated it, so the line is
// written and the pivot reciprocal started while step j - 1 still has its kd - 1 other columns to update
template <int W, int S>
__device__ __forceinline__ void fw_prepare(const FwState<W>& st, int j, int lane, int n, int kd, double* __restrict__ colbuf, double& rd,
                                           int& bad) {
  const int rel = (lane - j) & 63, R = j + rel;
  const bool inwin = rel >= 1 && rel <= kd && R < n;
  const double a = st.reg[S];
  colbuf[(j & 1) * FW_CB + rel] = inwin ? a : 0.0;          // double-buffered broadcast line
  const double d = fw_readlane(a, j & 63);
  rd = frcp(d);
  bad |= (j < n) & !(d > 0.0);                              // (called unconditionally, also for j = n: no branch in the step)
}

// one factorisation step; S = j mod W at compile time; `rd` = 1 / d_j on entry, 1 / d_(j+1) on return
template <int W, int S>
__device__ __forceinline__ void fw_step(FwState<W>& st, int j, int lane, int n, int kd, double* __restrict__ colbuf,
                                        double* __restrict__ Lc, double* __restrict__ xs, double& rd, int& bad) {
  const int rel = (lane - j) & 63, R = j + rel;
  const bool inwin = rel >= 1 && rel <= kd && R < n;
  const double a = st.reg[S], rdj = rd;
  const double l = inwin ? a * rdj : 0.0;
  const double zj = fw_readlane(st.y, j & 63);              // the pivot row's right-hand side is final
  // column j + 1 first, its multiplier A[j + 1][j] through a readlane instead of the line: the next step's line and
  // reciprocal start here
  const double a1 = fw_readlane(a, (j + 1) & 63);
  st.reg[(S + 1) % W] = __builtin_fma(-l, a1, st.reg[(S + 1) % W]);
  fw_fence();                                               // line j (written one step ago) has landed
  fw_prepare<W, (S + 1) % W>(st, j + 1, lane, n, kd, colbuf, rd, bad);
  const double* cb = colbuf + (j & 1) * FW_CB;
#ifndef FW_SKIP_LSTORE                                     // (phase ablation builds, scripts/frame_phase_ab.sh)
  if (inwin) Lc[(size_t)j * W + (rel - 1)] = l;             // column j of L: one coalesced store
#endif
  if (rel == 0) xs[j] = zj * rdj;                           // w_j = z_j / d_j
  st.y = __builtin_fma(-l, zj, st.y);
  // reg[(S + t) mod W] -= l * A[j + t][j], t = 2 .. W - 1: two columns per 16-byte broadcast read.  (The compiler keeps three
  // reads in flight; an explicit read-ahead of 5 .. 25 reads, at two or three waves per SIMD, measured slower: r03_notes 4.)
#pragma unroll
  for (int t = 2; t < W; t += 2) {
    const double2 ac = *reinterpret_cast<const double2*>(cb + t);
    st.reg[(S + t) % W] = __builtin_fma(-l, ac.x, st.reg[(S + t) % W]);
    if (t + 1 < W) st.reg[(S + t + 1) % W] = __builtin_fma(-l, ac.y, st.reg[(S + t + 1) % W]);
  }
  // a step's multiply-adds stay in the step: left free (no branch between two steps), the compiler defers them until the
  // column is next read and keeps -- spills -- the line values of several steps
#pragma unroll
  for (int c = 0; c < W; ++c) __asm__ volatile("" : "+v"(st.reg[c]));
}

// move one staged group (rows g0 .. g0 + G - 1: parked in `stage`) into the registers of the lanes that own them
template <int W>
__device__ __forceinline__ void fw_take_group(FwState<W>& st, int g0, int lane, const double* __restrict__ stage) {
  const int slot = (lane - g0) & 63;                        // row g0 + slot belongs to this lane when slot < G
  if (slot < FW_G) {
    const double2* r = reinterpret_cast<const double2*>(stage + (size_t)slot * fw_pitch(W));      // W even: 16-byte reads
#pragma unroll
    for (int c = 0; c < W; c += 2) { const double2 v = r[c / 2]; st.reg[c] = v.x; st.reg[c + 1] = v.y; }
    st.y = stage[(size_t)slot * fw_pitch(W) + W];
  }
}

// ---- backward substitution, blocked dot form: eight columns per pass, x_j = w_j - sum_t L[j+t][j] x_(j+t) ----
// Lane (u, k) = (lane >> 3, lane & 7) works for column j_u = jb - u: it multiplies the rows  j_u + k + 1 + 8 m  of that
// column (m < (W + 7) / 8: strided 64-byte runs of the column, x from LDS) and an 8-lane DPP sum gives every column's
// contribution of the rows ABOVE the block (x known); the rows inside the block (L[j_v][j_u], v < u: loaded by all eight
// lanes of group u into register v) follow as a serial chain of seven readlane + FMA.  ~80 VALU instructions per EIGHT
// columns against 36 per column of the r02 form (one column per pass: coalesced column load, 64-lane DPP / readlane
// reduction) and ~12 of the axpy form the narrow windows used (every lane walking its own column of L: up to kd cache
// lines per load).  10 x 10: 1.93 -> 1.84 ms per 16 384 frames, 15 x 16: 3.94 -> 3.87 ms per 12 288, 5 x 5: 0.72 -> 0.66 ms
// per 32 768, 3 x 3: 0.64 -> 0.59 ms per 65 536: what remains of the sweep is the stream of L out of HBM (all waves of a
// round reach their sweep together), r03_notes section 4.
template <int W>
__device__ __forceinline__ void fw_backward(const double* __restrict__ rows, double* __restrict__ xs, int n, int kd, int lane) {
  constexpr int MF = (W + 7) / 8;
  const int u = lane >> 3, k = lane & 7;
  xs[n + lane] = 0.0;                                     // rows past the last equation: x = 0 (the idle steps left garbage)
  fw_fence();
  double fA[MF], bA[7], fB[MF], bB[7];                   // the next block's L is in flight while a block is worked on (two
                                                          // blocks ahead measured no faster: the sweep streams L from HBM)
  auto issue = [&](int jb, double (&f)[MF], double (&bv)[7]) {      // unconditional loads from clamped addresses
    const int ju = jb - u, jc = ju > 0 ? ju : 0;
    const double* col = rows + (size_t)jc * W;
#pragma unroll
    for (int m = 0; m < MF; ++m) f[m] = col[k + 8 * m < W ? k + 8 * m : W - 1];
#pragma unroll
    for (int v = 0; v < 7; ++v) bv[v] = col[u - v - 1 > 0 ? u - v - 1 : 0];
  };
  auto block = [&](int jb, const double (&lf)[MF], const double (&lbv)[7]) {   // (jb < 0: every lane masked, no write)
    const int ju = jb - u, jc = ju > 0 ? ju : 0;
    const int kdj = ju >= 0 ? (kd < n - 1 - ju ? kd : n - 1 - ju) : 0;      // rows of this column below the diagonal
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int rel = k + 1 + 8 * m;
      const bool far = rel <= kdj && (m > 0 || k >= u);   // rows above the block top (rel > u); inside the band
      const double l = far ? lf[m] : 0.0;
      const double x = xs[jc + rel];
      if (m & 1) acc1 = __builtin_fma(l, x, acc1); else acc0 = __builtin_fma(l, x, acc0);
    }
    double s_ = acc0 + acc1;
    s_ += fw_dpp<0xB1>(s_);                               // quad_perm [1,0,3,2]
    s_ += fw_dpp<0x4E>(s_);                               // quad_perm [2,3,0,1]
    s_ += fw_dpp<0x141>(s_);                              // row_half_mirror: the eight lanes of a group hold its sum
    double t = xs[jc] - s_;
#pragma unroll
    for (int v = 0; v < 7; ++v) {                         // x of column j_v is final when its turn comes
      const double xv = fw_readlane(t, 8 * v);
      const double l = (v < u && u - v <= kdj) ? lbv[v] : 0.0;
      t = __builtin_fma(-l, xv, t);
    }
    if (k == 0 && ju >= 0) xs[ju] = t;
    fw_fence();
  };
#ifdef FW_SKIP_BACKWARD
  int jb = -1;
#else
  int jb = n - 1;
#endif
  issue(jb, fA, bA);
  for (; jb >= 0; jb -= 16) {                             // two blocks per pass: the buffers alternate without copies
    issue(jb - 8, fB, bB);
    block(jb, fA, bA);
    issue(jb - 16, fA, bA);
    block(jb - 8, fB, bB);
  }
}

template <int W>
__device__ __forceinline__ void frame_wave_body(const FrameParams& p, double* __restrict__ wsf, double* __restrict__ lds, int lane, long b,
                                                const FwPlan& pl) {
  constexpr int G = FW_G;
  const int n = p.n_eq, kd = p.kd;
  const int KG = (kd / G + 1) * G;                          // > kd (column j + 1 is read during step j): registers hold the rows below j + KG + G at step j
  double* colbuf = lds;                                     // [2][FW_CB]
  double* stage = lds + 2 * FW_CB;                          // [G][fw_pitch(W)]: rows + right-hand sides of one group
  double* xs = stage + (size_t)G * fw_pitch(W);             // [n + 64]: w, then x
  double* rows = wsf;                                       // [n][W]: column j of L at [j * W, j * W + kd)
  FwState<W> st;
#pragma unroll
  for (int c = 0; c < W; ++c) st.reg[c] = 0.0;
  st.y = 0.0;
  int bad = 0;

  // rows enter in groups of G: the group's values are parked in `stage` one boundary before the owners take them
  const double* Ib = p.I + b * p.Ne;
  const double* lb = p.loads + b * p.loads_bs;
  // -- fused assembly (plan): entry words of the NEXT group to build travel in registers (loaded one boundary ahead);
  //    build = one round trip (inertia gather, coefficients, right-hand side), issued before the boundary's LDS work
  constexpr int KE = FW_KE;
  unsigned eB[KE] = {0u, 0u, 0u};
  int dofB = 0, gB = 0;
  double bi[KE], ba[KE], bb[KE], by1 = 0.0, by2 = 0.0;
  const int n_extra = pl.hdr[0];
  auto ents = [&](int g0) {                                 // group g0 (a multiple of G): entry words + load index, no wait
    const int gi = g0 / G < pl.ng ? g0 / G : pl.ng;         // past the last equation: the all-zero block
    const unsigned* e = pl.ent + (size_t)gi * FW_EPG + lane;
#pragma unroll
    for (int k = 0; k < KE; ++k) eB[k] = e[64 * k];
    const int r = g0 + (lane < G ? lane : 0);
    dofB = pl.eq_dof[r < n ? r : n];
    gB = g0;
  };
  auto build_issue = [&]() {                                // the loads of group gB
    const int gi = gB / G < pl.ng ? gB / G : pl.ng;
    const double* ka = pl.ka + (size_t)gi * FW_EPG + lane;
    const double* kb = pl.kb + (size_t)gi * FW_EPG + lane;
#pragma unroll
    for (int k = 0; k < KE; ++k) { bi[k] = Ib[(eB[k] >> FW_SLOT_BITS) & 0x1FFFFF]; ba[k] = ka[64 * k]; bb[k] = kb[64 * k]; }
    const int r = gB + (lane < G ? lane : 0);
    by1 = pl.rhs_base[r < n ? r : n];
    by2 = lb[dofB];
  };
  auto build_finish = [&]() {                               // ... accumulated into the (zeroed) stage
#pragma unroll
    for (int k = 0; k < KE; ++k)
      if ((int)eB[k] < 0) atomicAdd(&stage[eB[k] & FW_SLOT_MASK], __builtin_fma(bi[k], bb[k], ba[k]));
    if (n_extra != 0) {                                     // nodes with more than four elements: extra blocks, not prefetched
      const int gi = gB / G < pl.ng ? gB / G : pl.ng;
      for (int blk = pl.xstart[gi]; blk < pl.xstart[gi + 1]; ++blk)
        for (int k = 0; k < KE; ++k) {
          const size_t i = (size_t)(pl.ng + 1 + blk) * FW_EPG + lane + 64 * k;
          const unsigned w = pl.ent[i];
          if ((int)w < 0) atomicAdd(&stage[w & FW_SLOT_MASK], __builtin_fma(Ib[(w >> FW_SLOT_BITS) & 0x1FFFFF], pl.kb[i], pl.ka[i]));
        }
    }
    fw_fence();
    if (lane < G) stage[lane * fw_pitch(W) + W] = (gB + lane < n) ? by1 + by2 : 0.0;
  };
  auto zero_stage = [&]() {
    for (int i = lane; i < G * fw_pitch(W); i += 64) stage[i] = 0.0;
    fw_fence();
  };
  // prologue: rows [0, KG + G) into registers, the next group parked, the one after on its way
  for (int g0 = 0; g0 < KG + 2 * G; g0 += G) {
    ents(g0);
    build_issue();
    zero_stage();
    build_finish();
    fw_fence();
    if (g0 < KG + G) { fw_take_group<W>(st, g0, lane, stage); fw_fence(); }
  }
  ents(KG + 2 * G);

  // ---- factorisation + forward substitution ----
  double rd = 0.0;
  fw_prepare<W, 0>(st, 0, lane, n, kd, colbuf, rd, bad);
  for (int j0 = 0; j0 < n; j0 += W) {
    auto boundary = [&](int j) {                            // j % G == 0, j > 0: rows [j + KG, j + KG + G) enter
      build_issue();                                        // group j + KG + G: its round trip runs under the LDS work below
      fw_take_group<W>(st, j + KG, lane, stage);
      fw_fence();
      zero_stage();
      build_finish();
      fw_fence();
      ents(j + KG + 2 * G);
    };
    // W-fold unrolled: the register index of column j is j mod W.  Guarded per FOUR steps (W % 4 == 0): a step past the
    // last equation is a no-op (no lane is inside its window; xs has 64 spare entries), and a branch per step made the
    // compiler split the step's reads from its multiply-adds (all kd / 2 reads live at once: +50 VGPRs)
#define FW_STEP(S_)                                                                   \
    {                                                                                 \
      const int j = j0 + (S_);                                                        \
      if constexpr ((S_) % 4 == 0) if (j > 0 && (j % G) == 0 && j < n) boundary(j);   /* j0 % 4 == 0 */ \
      fw_step<W, (S_)>(st, j, lane, n, kd, colbuf, rows, xs, rd, bad);                \
    }
#define FW_STEP4(S_)                                                                  \
    if constexpr ((S_) < W) {                                                         \
      if (j0 + (S_) < n) { FW_STEP(S_) FW_STEP(S_ + 1) FW_STEP(S_ + 2) FW_STEP(S_ + 3) } \
    }
#define FW_STEP8(S_) FW_STEP4(S_) FW_STEP4(S_ + 4)
    static_assert(W % 4 == 0 && W <= 56, "frame_wave: window widths are multiples of four");
    FW_STEP8(0) FW_STEP8(8) FW_STEP8(16) FW_STEP8(24) FW_STEP8(32) FW_STEP8(40) FW_STEP8(48)
#undef FW_STEP4
#undef FW_STEP8
#undef FW_STEP
  }
  fw_fence();

  fw_backward<W>(rows, xs, n, kd, lane);
  write_results(p, b, xs, bad != 0, lane, 64);
}

// waves per SIMD the register allocator is asked to make room for (the kernel is latency-bound per wave: rcp chain, LDS
// round trip of the broadcast line): 2 W VGPRs of window + ~55
#ifndef FW_WAVES_36
#define FW_WAVES_36 4      /* r06, late (profiles/r06_frame_wave_scalar_ab.txt): with the wave index in an SGPR the 36-wide kernel needs 133 VGPRs; held to 128 (28 B of scratch) four waves: 10 x 10 +2.5 % */
#endif
#ifndef FW_WAVES_56
#define FW_WAVES_56 2
#endif
constexpr int fw_waves(int W) { return W == 36 ? FW_WAVES_36 : W <= 52 ? 3 : FW_WAVES_56; }   // (56-wide at three waves: 16 B of scratch, -4 %)   // measured: only the 52-wide window gains (170 -> 168 VGPRs: 3 waves)

template <int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(fw_waves(W))))
void frame_wave_kernel(const FrameParams p, double* __restrict__ ws, const FwPlan pl) {
  extern __shared__ double lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // (wave: an SGPR -- frame number and workspace address are scalar)
  // one frame per wave, its own workspace slot.  (r06, measured: persistent waves with one slot per resident wave -- what keeps the packed
  // kernel's factor in cache, frame_pack.hpp -- buy nothing here: at 28 <= kd the factors in flight are 0.3 .. 1 GB, far beyond L2 and the
  // Infinity Cache either way (15 x 16: 8.8 GB of fabric traffic per 12 288 frames = the factor written once and read once, before and after),
  // and the loop around the body cost 30 bytes of scratch and 4 % of the time.)
  const long b = (long)blockIdx.x * 4 + wave;
  if (b >= p.B) return;
  frame_wave_body<W>(p, ws + b * fw_frame_doubles(p.n_eq, p.kd), lds + (size_t)wave * fw_lds_doubles(p.n_eq, W), lane, b, pl);
}

}  // namespace opsamd
