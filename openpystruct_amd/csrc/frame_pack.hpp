// Batched 2-D frame solve, third generation (r06): SEVERAL FRAMES PER WAVEFRONT for the narrow bands the reference actually draws.
// (included by frame_solve.hip after frame_wave.hpp: FrameParams, frcp, write_results, FwPlan, fw_dpp, fw_fence come from there)
//
// /root/reference/OpenPyStruct_FrameOpt_Discrete_Beta.py:17-18, :50-52 draws bays, stories ~ U{1..10}; along the short side of the grid
// (FrameTopology(numbering="auto")) the half bandwidth is kd = 3 m + 2, m = min(stories, bays + 1): 97 of the 100 draws have kd < 32 and
// more than half kd < 16.  The wave-per-frame kernel (frame_wave.hpp: lane = row, the kd x kd window in registers) gives every frame 64
// lanes, so for those frames half to nine tenths of every vector instruction is idle lanes (5 x 5: 1.8 % of the FP64 vector rate in r05).
// Here a frame gets P = 16 or 32 lanes and a wave carries F = 64 / P frames through ONE instruction stream (all frames of a launch share
// topology, so n, kd, the plan and every branch are wave-uniform; only inertias and loads differ between the lane groups).
//
// Same arithmetic as frame_wave.hpp -- column-by-column band LDL^T in equation order, no pivoting: what LAPACK does for the reference's
// `system('BandGeneral')` on an SPD matrix (FR:134) -- and the same row-per-lane register window (row R in lane R mod P, entry A[R][C] in
// reg[C mod W]).  What changes is every place where the 64-lane kernel used a wave-wide scalar (v_readlane): a lane group needs ITS frame's
// pivot, next-column multiplier and pivot-row right-hand side, so all three travel through the group's own broadcast line in LDS:
//
//   line of column c (written during step c - 1, double-buffered):   [0] = z_(c-1)   [1 + rel] = A[c + rel][c], rel = 0 .. kd  (0 outside)
//
// Every lane writes exactly one entry per step (lane of row c - 1 -- finished, its right-hand side final -- writes z, the window lanes
// their column entry); every lane reads the line as 16-byte broadcast reads: (z, d), (a1, a2) one step ahead (so the reciprocal of the pivot
// is computed under the previous step's multiply-adds), the rest as the step's operands.  The forward substitution therefore runs ONE
// COLUMN BEHIND the factorisation (y -= l_(j-1) z_(j-1) at the top of step j), which is what lets z ride in the line at no extra LDS
// operation.  No value ever crosses from one lane group to another: a NaN or non-positive pivot in one frame cannot reach its neighbours.
//
// Lines of the F frames of a wave sit at LDS strides that are NOT multiples of 256 B, so the 16-lane passes of a ds_read_b128 that mix two
// frames (P = 16) touch distinct banks.
//
// Rows enter the window in groups of G (8, or 4 where P - kd leaves no room for 8) through the fused assembly plan of frame_wave.hpp,
// each lane group building its own frame's rows in its own parking area.  Backward substitution: P / 8 columns per pass, eight lanes per
// column (strided runs of the column against x from LDS, 8-lane DPP sum); the pass's own small triangle is solved REDUNDANTLY by every lane
// (all P / 8 partial results through LDS, the three / one / six coefficients loaded by every lane) instead of a readlane chain.
#pragma once

namespace opsamd {

// dispatch: lanes per frame, rows per entering group, register window width (even, > kd).  KG = (kd / G + 1) G rows are in
// flight below the entering group: KG + G <= P.
__host__ __device__ inline bool fp_config(int kd, int* P, int* G, int* W) {
  // window widths: kd + 1 rounded up to even for the half bandwidths of grid frames (kd = 3 m + 2: 5, 8, 11, 14, 17, 20, 23, 26); any other
  // band takes the next wider one.  r06 first version: multiples of four (8, 12, 16, 20, 24, 28): two to four dead multiply-adds per step
  if (kd <= 5) { *P = 16; *G = 4; *W = 6; return true; }
  if (kd <= 9) { *P = 16; *G = 4; *W = 10; return true; }
  if (kd <= 11) { *P = 16; *G = 4; *W = 12; return true; }
  if (kd <= 15) { *P = 32; *G = 8; *W = 16; return true; }
  if (kd <= 17) { *P = 32; *G = 8; *W = 18; return true; }
  if (kd <= 21) { *P = 32; *G = 8; *W = 22; return true; }
  if (kd <= 23) { *P = 32; *G = 8; *W = 24; return true; }
  if (kd <= 27) { *P = 32; *G = 4; *W = 28; return true; }
  if (kd <= 29) { *P = 32; *G = 2; *W = 30; return true; }      // 9 x 9 .. 9 x 10: rows enter in pairs (KG + G = 32)
#ifdef FP_WITH_P64     // (r06, measured: this kernel at 64 lanes per frame -- correct on every size, 10 x 10 0.91 x, 12 x 12 1.09 x, 15 x 16 0.98 x of
                       //  frame_wave.hpp's kernel, whose wave-wide readlanes and 8-column backward chain it replaces with LDS traffic: not the default)
  if (kd <= 35) { *P = 64; *G = 8; *W = 36; return true; }      // one frame per wave: the same kernel, F = 1
  if (kd <= 43) { *P = 64; *G = 8; *W = 44; return true; }
  if (kd <= 51) { *P = 64; *G = 8; *W = 52; return true; }
  if (kd <= 55) { *P = 64; *G = 8; *W = 56; return true; }
#endif
  return false;
}
// entry slots per plan block: the grid frames of the reference's range need at most 60 (G = 4) / 116 (G = 8) entries per row group (nodes with
// up to four elements); a group with more continues in extra blocks (frame_wave.hpp)
__host__ __device__ constexpr int fp_epg(int G) { return 16 * G; }
__host__ __device__ constexpr int fp_tb(int P);
// per frame in LDS (doubles): two lines [P], the backward pass's partial results [P / 8], the parking area [G][W + 2], right-hand side -> w -> x
// [n + P], the frame's inertias [Ne]
__host__ __device__ inline size_t fp_lds_doubles(int n, int Ne, int P, int G, int W) {
  size_t d = 2 * (size_t)P + fp_tb(P) + (size_t)G * (W + 2) + (size_t)(n + P) + (size_t)Ne;
  d = (d + 1) & ~(size_t)1;
  if (d % 32 < 2 || d % 32 > 30) d += 2;        // neighbouring frames' lines on distinct banks (P = 16: two frames per 16-byte read pass)
  return d;
}
// per frame in the HBM workspace: column j of L at [j * W, j * W + kd); slot W - 1 of every column takes the (unconditional) stores of the
// lanes outside the window; three spare columns for the idle steps past the last equation (steps are guarded four at a time)
__host__ __device__ inline size_t fp_frame_doubles(int n, int W) { return (size_t)(n + 3) * W; }

template <int W>
struct FpState {
  double reg[W];     // own row: A[R][C] at index C mod W
  double y;          // own right-hand side (forward)
  double lp;         // own multiplier of the previous step (the forward substitution runs one column behind)
  double w;          // w = z / d of the lane's last finished row (leaves for LDS at the next group boundary)
};

// line of column 0 (before the first step)
template <int W, int P>
__device__ __forceinline__ void fp_first_line(const FpState<W>& st, int r, int kd, int n, double* __restrict__ line) {
  const int idx = (r + 1) & (P - 1);                          // 1 + rel; lane P - 1: the z slot
  const int lim = kd + 1 < n ? kd + 1 : n;
  line[idx] = (idx >= 1 && idx - 1 < lim) ? st.reg[0] : 0.0;
}

// one factorisation step; S = j mod W at compile time (W even: S & 1 = j & 1).  On entry: rd = 1 / d_j, zp = z_(j-1), a1 = A[j+1][j],
// a2 = A[j+2][j] (line values); on return the same for step j + 1.  (A lane group past the end of the batch solves the batch's LAST frame once
// more, into its own workspace slot, and writes no results: n, kd and every window bound are wave-uniform.)  No branch: the column of L leaves with an unconditional store (lanes outside the window write slot W - 1 of the
// column, which nothing reads), w stays in a register until the next group boundary.
// Lw: this WAVE's first frame in the workspace (uniform), loff: the lane group's frame offset in doubles.
template <int W, int P, int S>
__device__ __forceinline__ void fp_step(FpState<W>& st, int j, int r, int kd, int n, double* __restrict__ line, double* __restrict__ Lw,
                                        unsigned loff, double& rd, double& zp, double& a1, double& a2, int& bad) {
  const int rel = (r - j) & (P - 1);
  const int below = n - 1 - j > 0 ? n - 1 - j : 0;            // rows below the diagonal that exist (uniform)
  const int lim = kd < below ? kd : below;                    // (uniform, >= 0)
  const bool inwin = (unsigned)(rel - 1) < (unsigned)lim;     // 1 <= rel <= min(kd, n - 1 - j)
  st.y = __builtin_fma(-st.lp, zp, st.y);                     // column j - 1's part of the forward substitution
  const double a = st.reg[S], rdj = rd;
  const double l = inwin ? a * rdj : 0.0;
  st.lp = l;
  st.reg[(S + 1) % W] = __builtin_fma(-l, a1, st.reg[(S + 1) % W]);      // column j + 1 is final: its line leaves now
  // Line of column j + 1: the lane of row j its right-hand side, every other lane its row's entry of column j + 1 -- UNMASKED.  Entries from
  // outside the band are finite leftovers (a row ahead of the window may hold a later column in that slot); they only ever multiply (i) a zero
  // multiplier (lanes outside the window: l = 0) or (ii) into slots (column j + t, t > kd) that are dead for every window row (above its diagonal
  // or already eliminated).  A row past the last equation contributes zeros (unit-diagonal / zero rows of the plan).
  line[((S + 1) & 1) * P + rel] = rel == 0 ? st.y : st.reg[(S + 1) % W];
#ifndef FP_SKIP_LSTORE                                       // (phase ablation builds: wrong answers, scripts/frame_pack_ablation.sh)
  // column j of L: one coalesced store per lane group (scalar column address + a 32-bit byte offset: a frame slot is far below 4 GB)
  *reinterpret_cast<double*>(reinterpret_cast<char*>(Lw + (size_t)j * W) + (size_t)((loff + (unsigned)(inwin ? rel - 1 : W - 1)) * 8u)) = l;
#endif
  st.w = (rel == 0 && j < n) ? st.y * rdj : st.w;             // w_j = z_j / d_j
  const double* cb = line + (S & 1) * P;
  if constexpr (W > 2) st.reg[(S + 2) % W] = __builtin_fma(-l, a2, st.reg[(S + 2) % W]);
  // reg[(S + t) mod W] -= l * A[j + t][j], t = 3 .. W - 1: line index t + 1, two columns per 16-byte broadcast read
#pragma unroll
  for (int t = 3; t < W; t += 2) {
    const double2 ac = *reinterpret_cast<const double2*>(cb + t + 1);
    st.reg[(S + t) % W] = __builtin_fma(-l, ac.x, st.reg[(S + t) % W]);
    if (t + 1 < W) st.reg[(S + t + 1) % W] = __builtin_fma(-l, ac.y, st.reg[(S + t + 1) % W]);
  }
  // the next step's early operands (this wave's LDS operations execute in order: the line written above is what these reads return)
  __asm__ volatile("" ::: "memory");
  const double* nb = line + ((S + 1) & 1) * P;
  const double2 p0 = *reinterpret_cast<const double2*>(nb), p1 = *reinterpret_cast<const double2*>(nb + 2);
  zp = p0.x;
  rd = frcp(p0.y);
  bad |= (j + 1 < n) & !(p0.y > 0.0);
  a1 = p1.x;
  a2 = p1.y;
  // a step's multiply-adds stay in the step (frame_wave.hpp fw_step: left free, the compiler defers them and spills line values)
#pragma unroll
  for (int c = 0; c < W; ++c) __asm__ volatile("" : "+v"(st.reg[c]));
}

// move one parked group (rows g0 .. g0 + G - 1) into the registers of the lanes that own them
template <int W, int P, int G>
__device__ __forceinline__ void fp_take_group(FpState<W>& st, int g0, int r, const double* __restrict__ stage) {
  const int slot = (r - g0) & (P - 1);
  if (slot < G) {
    const double2* q = reinterpret_cast<const double2*>(stage + (size_t)slot * (W + 2));
#pragma unroll
    for (int c = 0; c < W; c += 2) { const double2 v = q[c / 2]; st.reg[c] = v.x; st.reg[c + 1] = v.y; }
    st.y = stage[(size_t)slot * (W + 2) + W];
  }
}

// LDS operations of one wave execute in issue order, atomics included: between a write and the read that wants it (or a read and the write
// that replaces what it read) the compiler must not reorder, but the wave need not wait -- frame_wave.hpp's fw_fence also drains lgkmcnt
__device__ __forceinline__ void fp_order() {
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ---- backward substitution: U = P / 8 columns per pass, x_j = w_j - sum_t L[j+t][j] x_(j+t) ----
// Lane (u, k) = (r >> 3, r & 7) works for column j_u = jb - u: the rows j_u + k + 1 + 8 m of that column against x from LDS, an 8-lane DPP sum
// gives every column's contribution of the rows ABOVE the pass (x known).  The pass's own triangle -- L[jb - v][jb - w], v < w < U: the first
// entries of the pass's columns, held by the lanes (w, k < w) -- goes through LDS with the U partial results, and every lane solves the
// U x U triangle redundantly (no cross-lane chain).  L streams back from L2 / HBM (phase ablation, r06: the sweep was 24 % of the launch
// with one pass in flight ahead of the work): the loads of the next FP_BD passes are in flight while FP_BD passes are worked on.
#ifndef FP_BD
#define FP_BD 6
#endif
__host__ __device__ constexpr int fp_tb(int P) { return (P / 8 + (P / 8) * (P / 8 - 1) / 2 + 1) & ~1; }      // partial results + triangle
template <int W, int P>
__device__ __forceinline__ void fp_backward(const double* __restrict__ Lc, double* __restrict__ xs, double* __restrict__ tb, int n,
                                            int kd, int r) {
  // passes in flight: six at three waves per SIMD; four where four waves share the SIMD and the window is 12 .. 18 wide (measured late in r06: 4 x 4 +5 %, 5 x 5 / 3 x 3 +0.3 %, eight: -7 %)
  constexpr int U = P / 8, MF = (W + 7) / 8, NT = U * (U - 1) / 2, D = P == 64 ? 2 : (W >= 12 && W <= 18 && FP_BD > 4) ? 4 : FP_BD;      // (W = 10: four passes -6 %)
  static_assert(U >= 2, "frame_pack: at least 16 lanes per frame");
  const int u = r >> 3, k = r & 7;
  xs[n + r] = 0.0;                                        // rows past the last equation (the idle steps left garbage there)
  fp_order();
  double fa[D][MF], fb[D][MF];
  auto issue = [&](int jb, double (&f)[MF]) {             // unconditional loads from clamped addresses
    const int ju = jb - u, jc = ju > 0 ? ju : 0;
    const double* col = Lc + (unsigned)(jc * W);
#pragma unroll
    for (int m = 0; m < MF; ++m) f[m] = col[k + 8 * m < W ? k + 8 * m : W - 1];
  };
  auto block = [&](int jb, const double (&lf)[MF]) {
    const int ju = jb - u, jc = ju > 0 ? ju : 0;
    const int kdj = ju >= 0 ? (kd < n - 1 - ju ? kd : n - 1 - ju) : 0;       // rows of this column below the diagonal
    if (k < u) tb[U + u * (u - 1) / 2 + (u - 1 - k)] = (k + 1 <= kdj) ? lf[0] : 0.0;      // L[jb - v][jb - u], v = u - 1 - k: offset k
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int m = 0; m < MF; ++m) {
      const int rel = k + 1 + 8 * m;
      const bool far = rel <= kdj && rel > u;             // rows above the pass's top (x known)
      const double l = far ? lf[m] : 0.0;
      const double x = xs[jc + rel];
      if (m & 1) acc1 = __builtin_fma(l, x, acc1); else acc0 = __builtin_fma(l, x, acc0);
    }
    double s_ = acc0 + acc1;
    s_ += fw_dpp<0xB1>(s_);                               // quad_perm [1,0,3,2]
    s_ += fw_dpp<0x4E>(s_);                               // quad_perm [2,3,0,1]
    s_ += fw_dpp<0x141>(s_);                              // row_half_mirror: the eight lanes of a column hold its sum
    if (k == 0) tb[u] = xs[jc] - s_;
    fp_order();
    double x[U], tr[NT];
#pragma unroll
    for (int w = 0; w < U; w += 2) { const double2 q = *reinterpret_cast<const double2*>(tb + w); x[w] = q.x; x[w + 1] = q.y; }
#pragma unroll
    for (int i = 0; i < NT; i += 2) {
      const double2 q = *reinterpret_cast<const double2*>(tb + U + i);
      tr[i] = q.x;
      if (i + 1 < NT) tr[i + 1] = q.y;
    }
    {
      int i = 0;
#pragma unroll
      for (int w = 1; w < U; ++w)
#pragma unroll
        for (int v = 0; v < w; ++v) x[w] = __builtin_fma(-tr[i++], x[v], x[w]);
    }
    double mine = x[0];
#pragma unroll
    for (int w = 1; w < U; ++w) mine = (r == w) ? x[w] : mine;
    if (r < U && jb - r >= 0) xs[jb - r] = mine;
    fp_order();
  };
#ifdef FP_SKIP_BACKWARD
  int jb = -1;
#else
  int jb = n - 1;
#endif
#pragma unroll
  for (int d = 0; d < D; ++d) issue(jb - d * U, fa[d]);
  for (; jb >= 0; jb -= 2 * D * U) {                      // 2 D passes per trip: two register sets alternate without copies
#pragma unroll
    for (int d = 0; d < D; ++d) { issue(jb - (D + d) * U, fb[d]); block(jb - d * U, fa[d]); }
#pragma unroll
    for (int d = 0; d < D; ++d) { issue(jb - (2 * D + d) * U, fa[d]); block(jb - (D + d) * U, fb[d]); }
  }
}

template <int W, int P, int G>
__device__ __forceinline__ void frame_pack_body(const FrameParams& p, double* __restrict__ Lw, unsigned loff, double* __restrict__ lds, int r, long b,
                                                bool live, const FwPlan& pl, bool stage_I) {
  constexpr int EPG = fp_epg(G), KE = EPG / P, PITCH = W + 2;
  static_assert(EPG % P == 0 && W % 2 == 0 && (P & (P - 1)) == 0, "frame_pack: sizes");
  const int n = p.n_eq, kd = p.kd;
  const double* Lc = Lw + loff;                             // this frame's columns of L
  const int KG = (kd / G + 1) * G;                          // > kd: registers hold the rows below j + KG + G at step j
  double* line = lds;                                       // [2][P]
  double* tb = lds + 2 * P;                                 // [fp_tb(P)]
  double* stage = tb + fp_tb(P);                            // [G][PITCH]
  double* xs = stage + G * PITCH;                           // [n + P]
  FpState<W> st;
#pragma unroll
  for (int c = 0; c < W; ++c) st.reg[c] = 0.0;
  st.y = 0.0;
  st.lp = 0.0;
  st.w = 0.0;
  int bad = 0;

  // the frame's own data, ONE round trip to HBM for the whole solve: inertias and right-hand side (consistent element loads of the plan + the
  // nodal load of each equation's DOF) into LDS.  xs[q] holds the right-hand side of row q until the row's group is built (long before step q
  // overwrites it with w_q).  Everything the row groups need after this is the plan -- the same addresses for every frame of the launch: L2.
  double* Il = xs + (n + P);                                // [Ne] (stage_I; a frame with so many elements that they do not fit: gathered from HBM)
  const double* Ib = p.I + b * p.Ne;
  {
    const double* lb = p.loads + b * p.loads_bs;
    if (stage_I) for (int e = r; e < p.Ne; e += P) Il[e] = Ib[e];
    for (int q = r; q < n + P; q += P) xs[q] = q < n ? pl.rhs_base[q] + lb[pl.eq_dof[q]] : 0.0;
  }
  unsigned eB[KE];
  int gB = 0;
  double ba[KE], bb[KE];
#pragma unroll
  for (int k = 0; k < KE; ++k) eB[k] = 0u;
  const int n_extra = pl.hdr[0];
  auto ents = [&](int g0) {                                 // group g0 (a multiple of G): entry words and coefficients, no wait
    const int gi = g0 / G < pl.ng ? g0 / G : pl.ng;         // past the last equation: the all-zero block
    const unsigned* e = pl.ent + (size_t)gi * EPG + r;
    const double* ka = pl.ka + (size_t)gi * EPG + r;
    const double* kb = pl.kb + (size_t)gi * EPG + r;
#pragma unroll
    for (int k = 0; k < KE; ++k) { eB[k] = e[P * k]; ba[k] = ka[P * k]; bb[k] = kb[P * k]; }
    gB = g0;
  };
  auto build = [&]() {                                      // group gB accumulated into the (zeroed) parking area
#pragma unroll
    for (int k = 0; k < KE; ++k)
      if ((int)eB[k] < 0) {
        const unsigned e = (eB[k] >> FW_SLOT_BITS) & 0x1FFFFF;
        atomicAdd(&stage[eB[k] & FW_SLOT_MASK], __builtin_fma(stage_I ? Il[e] : Ib[e], bb[k], ba[k]));
      }
    if (n_extra != 0) {                                     // nodes with more than four elements: extra blocks, not prefetched
      const int gi = gB / G < pl.ng ? gB / G : pl.ng;
      for (int blk = pl.xstart[gi]; blk < pl.xstart[gi + 1]; ++blk)
        for (int k = 0; k < KE; ++k) {
          const size_t i = (size_t)(pl.ng + 1 + blk) * EPG + r + P * k;
          const unsigned w = pl.ent[i];
          if ((int)w < 0) atomicAdd(&stage[w & FW_SLOT_MASK], __builtin_fma(Ib[(w >> FW_SLOT_BITS) & 0x1FFFFF], pl.kb[i], pl.ka[i]));
        }
    }
    if (r < G) stage[r * PITCH + W] = xs[gB + r];           // (rows past the last equation: zero, staged above)
  };
  auto zero_stage = [&]() {
    for (int i = r; i < G * PITCH; i += P) stage[i] = 0.0;
    fp_order();
  };
  // prologue: rows [0, KG + G) into registers, the next group parked, the one after on its way
  ents(0);
  for (int g0 = 0; g0 < KG + 2 * G; g0 += G) {
    zero_stage();
    build();
    ents(g0 + G);
    fp_order();
    if (g0 < KG + G) { fp_take_group<W, P, G>(st, g0, r, stage); fp_order(); }
  }

  // ---- factorisation + forward substitution ----
  double rd, zp, a1, a2;
  fp_first_line<W, P>(st, r, kd, n, line);
  fp_order();
  {
    const double2 p0 = *reinterpret_cast<const double2*>(line), p1 = *reinterpret_cast<const double2*>(line + 2);
    zp = 0.0;
    (void)p0.x;
    rd = frcp(p0.y);
    bad |= !(p0.y > 0.0);
    a1 = p1.x;
    a2 = p1.y;
  }
  for (int j0 = 0; j0 < n; j0 += W) {
    auto boundary = [&](int j) {                            // j % G == 0, j > 0: rows [j + KG, j + KG + G) enter
      { const int slot = (r - (j - G)) & (P - 1); if (slot < G) xs[j - G + slot] = st.w; }      // rows [j - G, j) are finished: their w
      fp_take_group<W, P, G>(st, j + KG, r, stage);
      fp_order();
      zero_stage();
      build();                                              // group j + KG + G, from the plan words loaded one boundary ago
      fp_order();
      ents(j + KG + 2 * G);
    };
#ifndef FP_NO_BOUNDARY
#define FP_NO_BOUNDARY 0
#endif
    // W-fold unrolled (the register index of column j is j mod W), guarded per TWO steps (W even: j0 and every boundary -- a multiple of G --
    // are even, so boundaries fall on even S): a step past the last equation is a no-op
#define FP_STEP(S_)                                                                   \
    {                                                                                 \
      const int j = j0 + (S_);                                                        \
      if constexpr (!FP_NO_BOUNDARY && (S_) % 2 == 0) if (j > 0 && (j % G) == 0 && j < n) boundary(j);   \
      fp_step<W, P, (S_)>(st, j, r, kd, n, line, Lw, loff, rd, zp, a1, a2, bad);      \
    }
#define FP_STEP4(S_)                                                                  \
    if constexpr ((S_) < W) {                                                         \
      if (j0 + (S_) < n) { FP_STEP(S_) FP_STEP(S_ + 1) }                              \
    }                                                                                 \
    if constexpr ((S_) + 2 < W) {                                                     \
      if (j0 + (S_) + 2 < n) { FP_STEP(S_ + 2) FP_STEP(S_ + 3) }                      \
    }
    static_assert(W <= 56 && W % 2 == 0, "frame_pack: even window widths up to 56");
    FP_STEP4(0) FP_STEP4(4) FP_STEP4(8) FP_STEP4(12) FP_STEP4(16) FP_STEP4(20) FP_STEP4(24) FP_STEP4(28) FP_STEP4(32) FP_STEP4(36) FP_STEP4(40)
    FP_STEP4(44) FP_STEP4(48) FP_STEP4(52)
#undef FP_STEP4
#undef FP_STEP
  }
  {                                                         // the rows finished since the last boundary (and, again, up to P - G before)
    const int Rl = n - 1 - ((n - 1 - r) & (P - 1));
    if (Rl >= 0) xs[Rl] = st.w;
  }
  fp_order();

  fp_backward<W, P>(Lc, xs, tb, n, kd, r);
  if (live) write_results(p, b, xs, bad != 0, r, P);
}

// waves per SIMD the register allocator is asked to make room for: the wave is latency-bound (LDS round trips of the line, reciprocal chain)
#ifndef FP_WAVES
#define FP_WAVES(W) ((W) <= 18 ? 4 : (W) <= 30 ? 3 : 2)       /* measured (scripts/experiments/frame_waves_ab.sh): four waves +5-8 % at W = 16, +2 % at 18 (36 B of scratch), -5 / -7 % at 22 / 24 */
#endif
template <int W, int P, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FP_WAVES(W))))
void frame_pack_kernel(const FrameParams p, double* __restrict__ ws, const FwPlan pl, int ne_lds) {
  extern __shared__ double lds[];
  constexpr int F = 64 / P;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, sub = lane / P, r = lane & (P - 1);      // (wave: an SGPR -- the workspace slot's address is scalar)
  // PERSISTENT waves: the launch has at most as many workgroups as the chip holds at once, and a wave walks over its share of the batch with ONE
  // workspace slot (its own): the factor columns of all frames in flight are ~100 MB whatever the batch -- they stay in L2 / Infinity Cache
  // between the forward and the backward sweep instead of streaming through HBM (per-frame slots: 2 x 15 KB of HBM traffic per 5 x 5 frame
  // against 5 KB of inputs and results)
  const size_t fd = fp_frame_doubles(p.n_eq, W);            // (F frames of at most a few hundred KB: the lane group's offset fits 32 bits)
  const long wslot = (long)blockIdx.x * 4 + wave, stride = (long)gridDim.x * 4 * F;
  const unsigned ldso = (unsigned)((wave * F + sub) * fp_lds_doubles(p.n_eq, ne_lds, P, G, W));     // ne_lds = Ne (inertias staged in LDS) or 0
  for (long first = wslot * F; first < p.B; first += stride) {      // (wave-uniform)
    // the lane's coordinates are re-read "opaquely" per frame: left loop-invariant, every per-lane address of the body (plan, LDS areas, workspace)
    // is hoisted out of this loop and kept in registers across it (+40 VGPRs, 100-250 bytes of scratch)
    int r_ = r, sub_ = sub;
    unsigned ldso_ = ldso;
    __asm__ volatile("" : "+v"(r_), "+v"(sub_), "+v"(ldso_));
    const long slot = first + sub_;
    const bool live = slot < p.B;
    const long b = live ? slot : (long)p.B - 1;
    frame_pack_body<W, P, G>(p, ws + (size_t)wslot * F * fd, (unsigned)(sub_ * fd), lds + ldso_, r_, b, live, pl, ne_lds != 0);
    fp_order();
  }
}

}  // namespace opsamd
