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

// dispatch: lanes per frame, rows per entering group, register window width (multiples of four, > kd).  KG = (kd / G + 1) G rows are in
// flight below the entering group: KG + G <= P.
__host__ __device__ inline bool fp_config(int kd, int* P, int* G, int* W) {
  if (kd <= 7) { *P = 16; *G = 4; *W = 8; return true; }
  if (kd <= 11) { *P = 16; *G = 4; *W = 12; return true; }
  if (kd <= 15) { *P = 32; *G = 8; *W = 16; return true; }
  if (kd <= 23) { *P = 32; *G = 8; *W = 24; return true; }
  if (kd <= 27) { *P = 32; *G = 4; *W = 28; return true; }
  return false;
}
__host__ __device__ constexpr int fp_epg(int G) { return 24 * G; }            // entry slots per plan block (192 for G = 8, as frame_wave.hpp)
__host__ __device__ constexpr int fp_tb(int P) { return P / 8 < 2 ? 2 : P / 8; }
// per frame in LDS (doubles): two lines [P], the backward pass's partial results [P / 8], the parking area [G][W + 2], x [n + P]
__host__ __device__ inline size_t fp_lds_doubles(int n, int P, int G, int W) {
  size_t d = 2 * (size_t)P + fp_tb(P) + (size_t)G * (W + 2) + (size_t)(n + P);
  d = (d + 1) & ~(size_t)1;
  if (d % 32 < 2 || d % 32 > 30) d += 2;        // neighbouring frames' lines on distinct banks (P = 16: two frames per 16-byte read pass)
  return d;
}
// per frame in the HBM workspace: column j of L at [j * W, j * W + kd)
__host__ __device__ inline size_t fp_frame_doubles(int n, int W) { return (size_t)n * W; }

template <int W>
struct FpState {
  double reg[W];     // own row: A[R][C] at index C mod W
  double y;          // own right-hand side (forward)
  double lp;         // own multiplier of the previous step (the forward substitution runs one column behind)
};

// line of column 0 (before the first step)
template <int W, int P>
__device__ __forceinline__ void fp_first_line(const FpState<W>& st, int r, int nl, int kd, double* __restrict__ line) {
  const int idx = (r + 1) & (P - 1);                          // 1 + rel; lane P - 1: the z slot
  const bool in = idx >= 1 && idx - 1 <= kd && idx - 1 < nl;
  line[idx] = in ? st.reg[0] : 0.0;
}

// one factorisation step; S = j mod W at compile time.  On entry: rd = 1 / d_j, zp = z_(j-1), a1 = A[j+1][j], a2 = A[j+2][j] (line
// values); on return the same for step j + 1.
template <int W, int P, int S>
__device__ __forceinline__ void fp_step(FpState<W>& st, int j, int r, int nl, int n, int kd, double* __restrict__ line,
                                        double* __restrict__ Lc, double* __restrict__ xs, double& rd, double& zp, double& a1, double& a2,
                                        int& bad) {
  const int rel = (r - j) & (P - 1), R = j + rel;
  const bool inwin = rel >= 1 && rel <= kd && R < nl;
  st.y = __builtin_fma(-st.lp, zp, st.y);                     // column j - 1's part of the forward substitution
  const double a = st.reg[S], rdj = rd;
  const double l = inwin ? a * rdj : 0.0;
  st.lp = l;
  st.reg[(S + 1) % W] = __builtin_fma(-l, a1, st.reg[(S + 1) % W]);      // column j + 1 is final: its line leaves now
  {
    const bool in1 = rel >= 1 && rel <= kd + 1 && R < nl;     // row R in column j + 1's line (rel = 1: the pivot d_(j+1) itself)
    line[((j + 1) & 1) * P + rel] = rel == 0 ? st.y : (in1 ? st.reg[(S + 1) % W] : 0.0);
  }
  if (inwin) Lc[(size_t)j * W + (rel - 1)] = l;               // column j of L: one coalesced store per lane group
  if (rel == 0) xs[j] = st.y * rdj;                           // w_j = z_j / d_j
  const double* cb = line + (j & 1) * P;
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
  const double* nb = line + ((j + 1) & 1) * P;
  const double2 p0 = *reinterpret_cast<const double2*>(nb), p1 = *reinterpret_cast<const double2*>(nb + 2);
  zp = p0.x;
  rd = frcp(p0.y);
  bad |= (j + 1 < nl) & !(p0.y > 0.0);
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

// ---- backward substitution: U = P / 8 columns per pass, x_j = w_j - sum_t L[j+t][j] x_(j+t) ----
template <int W, int P>
__device__ __forceinline__ void fp_backward(const double* __restrict__ Lc, double* __restrict__ xs, double* __restrict__ tb, int n, int nl,
                                            int kd, int r) {
  constexpr int U = P / 8, MF = (W + 7) / 8, NT = U * (U - 1) / 2;
  const int u = r >> 3, k = r & 7;
  xs[n + r] = 0.0;                                        // rows past the last equation (the idle steps left garbage there)
  fw_fence();
  double fA[MF], fB[MF], tA[NT > 0 ? NT : 1], tB[NT > 0 ? NT : 1];
  auto issue = [&](int jb, double (&f)[MF], double (&tr)[NT > 0 ? NT : 1]) {       // unconditional loads from clamped addresses
    const int ju = jb - u, jc = ju > 0 ? ju : 0;
    const double* col = Lc + (size_t)jc * W;
#pragma unroll
    for (int m = 0; m < MF; ++m) f[m] = col[k + 8 * m < W ? k + 8 * m : W - 1];
    // the pass's own triangle: L[jb - v][jb - w], v < w < U, at column (jb - w), offset w - v - 1 -- every lane loads all of them
    int i = 0;
#pragma unroll
    for (int w = 1; w < U; ++w)
#pragma unroll
      for (int v = 0; v < w; ++v) { const int jw = jb - w > 0 ? jb - w : 0; tr[i++] = Lc[(size_t)jw * W + (w - v - 1)]; }
  };
  auto block = [&](int jb, const double (&lf)[MF], const double (&tr)[NT > 0 ? NT : 1]) {
    const int ju = jb - u, jc = ju > 0 ? ju : 0;
    const int kdj = ju >= 0 ? (kd < nl - 1 - ju ? kd : nl - 1 - ju) : 0;      // rows of this column below the diagonal
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
    const double t = xs[jc] - s_;
    double x[U];
    if constexpr (U == 1) {
      x[0] = t;
    } else {
      if (k == 0) tb[u] = t;
      fw_fence();
#pragma unroll
      for (int w = 0; w < U; w += 2) { const double2 q = *reinterpret_cast<const double2*>(tb + w); x[w] = q.x; x[w + 1] = q.y; }
      int i = 0;
#pragma unroll
      for (int w = 1; w < U; ++w) {
        const int jw = jb - w, kdw = jw >= 0 ? (kd < nl - 1 - jw ? kd : nl - 1 - jw) : 0;
#pragma unroll
        for (int v = 0; v < w; ++v) { const double l = (w - v <= kdw) ? tr[i] : 0.0; ++i; x[w] = __builtin_fma(-l, x[v], x[w]); }
      }
    }
    double mine = x[0];
#pragma unroll
    for (int w = 1; w < U; ++w) mine = (r == w) ? x[w] : mine;
    if (r < U && jb - r >= 0 && jb - r < nl) xs[jb - r] = mine;
    fw_fence();
  };
  int jb = n - 1;
  issue(jb, fA, tA);
  for (; jb >= 0; jb -= 2 * U) {                          // two passes per trip: the buffers alternate without copies
    issue(jb - U, fB, tB);
    block(jb, fA, tA);
    issue(jb - 2 * U, fA, tA);
    block(jb - U, fB, tB);
  }
}

template <int W, int P, int G>
__device__ __forceinline__ void frame_pack_body(const FrameParams& p, double* __restrict__ Lc, double* __restrict__ lds, int r, long b, bool live,
                                                const FwPlan& pl) {
  constexpr int EPG = fp_epg(G), KE = EPG / P, PITCH = W + 2;
  static_assert(EPG % P == 0 && W % 4 == 0 && (P & (P - 1)) == 0, "frame_pack: sizes");
  const int n = p.n_eq, kd = p.kd;
  const int nl = live ? n : 0;                              // a lane group past the end of the batch: nothing is in its window, nothing is stored
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
  int bad = 0;

  const double* Ib = p.I + b * p.Ne;
  const double* lb = p.loads + b * p.loads_bs;
  unsigned eB[KE];
  int dofB = 0, gB = 0;
  double bi[KE], ba[KE], bb[KE], by1 = 0.0, by2 = 0.0;
#pragma unroll
  for (int k = 0; k < KE; ++k) eB[k] = 0u;
  const int n_extra = pl.hdr[0];
  auto ents = [&](int g0) {                                 // group g0 (a multiple of G): entry words + load index, no wait
    const int gi = g0 / G < pl.ng ? g0 / G : pl.ng;         // past the last equation: the all-zero block
    const unsigned* e = pl.ent + (size_t)gi * EPG + r;
#pragma unroll
    for (int k = 0; k < KE; ++k) eB[k] = e[P * k];
    const int q = g0 + (r < G ? r : 0);
    dofB = pl.eq_dof[q < n ? q : n];
    gB = g0;
  };
  auto build_issue = [&]() {                                // the loads of group gB
    const int gi = gB / G < pl.ng ? gB / G : pl.ng;
    const double* ka = pl.ka + (size_t)gi * EPG + r;
    const double* kb = pl.kb + (size_t)gi * EPG + r;
#pragma unroll
    for (int k = 0; k < KE; ++k) { bi[k] = Ib[(eB[k] >> FW_SLOT_BITS) & 0x1FFFFF]; ba[k] = ka[P * k]; bb[k] = kb[P * k]; }
    const int q = gB + (r < G ? r : 0);
    by1 = pl.rhs_base[q < n ? q : n];
    by2 = lb[dofB];
  };
  auto build_finish = [&]() {                               // ... accumulated into the (zeroed) parking area
#pragma unroll
    for (int k = 0; k < KE; ++k)
      if ((int)eB[k] < 0) atomicAdd(&stage[eB[k] & FW_SLOT_MASK], __builtin_fma(bi[k], bb[k], ba[k]));
    if (n_extra != 0) {                                     // nodes with more than four elements: extra blocks, not prefetched
      const int gi = gB / G < pl.ng ? gB / G : pl.ng;
      for (int blk = pl.xstart[gi]; blk < pl.xstart[gi + 1]; ++blk)
        for (int k = 0; k < KE; ++k) {
          const size_t i = (size_t)(pl.ng + 1 + blk) * EPG + r + P * k;
          const unsigned w = pl.ent[i];
          if ((int)w < 0) atomicAdd(&stage[w & FW_SLOT_MASK], __builtin_fma(Ib[(w >> FW_SLOT_BITS) & 0x1FFFFF], pl.kb[i], pl.ka[i]));
        }
    }
    fw_fence();
    if (r < G) stage[r * PITCH + W] = (gB + r < n) ? by1 + by2 : 0.0;
  };
  auto zero_stage = [&]() {
    for (int i = r; i < G * PITCH; i += P) stage[i] = 0.0;
    fw_fence();
  };
  // prologue: rows [0, KG + G) into registers, the next group parked, the one after on its way
  for (int g0 = 0; g0 < KG + 2 * G; g0 += G) {
    ents(g0);
    build_issue();
    zero_stage();
    build_finish();
    fw_fence();
    if (g0 < KG + G) { fp_take_group<W, P, G>(st, g0, r, stage); fw_fence(); }
  }
  ents(KG + 2 * G);

  // ---- factorisation + forward substitution ----
  double rd, zp, a1, a2;
  fp_first_line<W, P>(st, r, nl, kd, line);
  fw_fence();
  {
    const double2 p0 = *reinterpret_cast<const double2*>(line), p1 = *reinterpret_cast<const double2*>(line + 2);
    zp = 0.0;
    (void)p0.x;
    rd = frcp(p0.y);
    bad |= (0 < nl) & !(p0.y > 0.0);
    a1 = p1.x;
    a2 = p1.y;
  }
  for (int j0 = 0; j0 < n; j0 += W) {
    auto boundary = [&](int j) {                            // j % G == 0, j > 0: rows [j + KG, j + KG + G) enter
      build_issue();                                        // group j + KG + G: its round trip runs under the LDS work below
      fp_take_group<W, P, G>(st, j + KG, r, stage);
      fw_fence();
      zero_stage();
      build_finish();
      fw_fence();
      ents(j + KG + 2 * G);
    };
#define FP_STEP(S_)                                                                   \
    {                                                                                 \
      const int j = j0 + (S_);                                                        \
      if constexpr ((S_) % 4 == 0) if (j > 0 && (j % G) == 0 && j < n) boundary(j);   \
      fp_step<W, P, (S_)>(st, j, r, nl, n, kd, line, Lc, xs, rd, zp, a1, a2, bad);    \
    }
#define FP_STEP4(S_)                                                                  \
    if constexpr ((S_) < W) {                                                         \
      if (j0 + (S_) < n) { FP_STEP(S_) FP_STEP(S_ + 1) FP_STEP(S_ + 2) FP_STEP(S_ + 3) } \
    }
    static_assert(W <= 28, "frame_pack: window widths up to 28");
    FP_STEP4(0) FP_STEP4(4) FP_STEP4(8) FP_STEP4(12) FP_STEP4(16) FP_STEP4(20) FP_STEP4(24)
#undef FP_STEP4
#undef FP_STEP
  }
  fw_fence();

  fp_backward<W, P>(Lc, xs, tb, n, nl, kd, r);
  if (live) write_results(p, b, xs, bad != 0, r, P);
}

template <int W, int P, int G>
__global__ __launch_bounds__(256)
void frame_pack_kernel(const FrameParams p, double* __restrict__ ws, const FwPlan pl) {
  extern __shared__ double lds[];
  constexpr int F = 64 / P;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane / P, r = lane & (P - 1);
  const long slot = ((long)blockIdx.x * 4 + wave) * F + sub;
  if (((long)blockIdx.x * 4 + wave) * F >= p.B) return;     // (wave-uniform)
  const bool live = slot < p.B;
  const long b = live ? slot : (long)p.B - 1;
  frame_pack_body<W, P, G>(p, ws + b * fp_frame_doubles(p.n_eq, W), lds + (size_t)(wave * F + sub) * fp_lds_doubles(p.n_eq, P, G, W), r, b, live, pl);
}

}  // namespace opsamd
