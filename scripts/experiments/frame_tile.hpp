// Batched 2-D frame solve, MEASURED ALTERNATIVE (r04, OPS_AMD_FRAME_TILE=1; not the default): one wavefront per frame, the band window as an
// 8 x 8 LANE GRID OF REGISTER TILES.  Outcome (profiles/r04_notes.md 11): correct on every size, 36 instead of 52 multiply-adds and 15-22
// instead of 26 LDS instructions per step at 15 x 16 -- and still 15-35 % slower than the row-per-lane window, because what a step costs is
// not its multiply-adds: ~60 other VALU instructions (publishing the pivot column, scaling the multipliers, right-hand side, store of L,
// reciprocal) on either mapping, and here eight exec-masked publishing stores + 68 spilled registers on top.  Kept for the next attempt.
// (included by frame_solve.hip behind frame_wave.hpp: plan, parking-slot map, backward sweep, fw_readlane / fw_fence come from there)
//
// Same arithmetic, entry for entry and in the same order, as frame_wave.hpp (the column-by-column band LDL^T of dpbsv,
// /root/reference/OpenPyStruct_FrameOpt_Discrete_Beta.py:134): every entry A[R][C] of the window takes  A[R][C] -= (A[R][j] / d_j) A[C][j]
// once per step j as one FMA with the multiplier rounded first -- the factor columns equal the row-per-lane kernel's bit for bit except where
// the LDS atomic additions of the assembly met in another order (last-bit differences of a few assembled entries; scripts/frame_tile_ab.py).
//
// Why another mapping: with a row per lane every lane needs the WHOLE pivot column (kd values) per step -- kd / 2 16-byte LDS broadcast
// reads per wave and step (15 x 16: 26 reads per step).  Here the window is a ring of N = 8 M equations; lane (p, q) =
// (lane >> 3, lane & 7) holds the M x M entries whose row is = p and whose column is = q (mod 8):
//
//     A[R][C]  ->  lane (R & 7, C & 7),  register  reg[(R >> 3) mod M][(C >> 3) mod M]
//
// so a step needs only the M multipliers of ITS rows and the M pivot-row entries of ITS columns: 2 M doubles = M 16-byte reads, from a
// line in which the eight lanes that hold column j (q = j & 7) have laid their M entries side by side (line[8 p + slot]).  Only the lower
// triangle is kept: with the slot numbers relative to the pivot's own block (dr, dc = distance of the row / column slot from slot (j >> 3)
// mod M) the pairs dr >= dc are live -- M (M + 1) / 2 FMAs per step (M = 8: 36, against 52 of the 52-wide row window), the other register
// pairs are dead and the allocator reuses them.  Nothing is masked but the pivot's own block (rows / columns of the block at or before j):
// entries beyond the band are zero because the band is (no fill outside it), rows past the last equation are zero rows.
//
//   * the ring slot of the block that has just been eliminated takes the next group of eight rows at that block's last step (the same LDS
//     parking area and fused assembly as frame_wave.hpp; parked rows are laid out so that lane (p, q) takes its M entries of row p with
//     M / 2 conflict-free 16-byte reads): kd <= N - 8;
//   * the line of step j + 1 is written, and 1 / d_(j+1) started, DURING step j right after the M entries of column j + 1 have taken their
//     update (the remaining FMAs cover the reciprocal chain and the LDS round trip);
//   * the line is published a second time by ring position: the lane at ring position x carries the right-hand side of the row there, takes
//     its multiplier from that copy and stores it as that row's entry of column j of L -- the workspace layout and the backward sweep
//     (fw_backward) are those of frame_wave.hpp.
//
// The code is unrolled M-fold over the pivot's slot (every register index is a compile-time constant), the eight steps of a block are a
// run-time loop (lane predicates and readlane indices from an SGPR).
#pragma once

#include <type_traits>

namespace opsamd {

__host__ __device__ constexpr int ft_M(int W) { return W == 16 ? 3 : W == 24 ? 4 : W == 36 ? 6 : 8; }     // ring slots for the kd range of window width W
__host__ __device__ inline size_t ft_lds_doubles(int n) { return ((size_t)(2 * (8 * 10 + 64) + FW_G * FT_P + n + 64) + 1) & ~(size_t)1; }

template <int M>
struct FtRegs {
  double r[M][M];
  double y;          // right-hand side of the row at ring position `lane` (lanes < 8 M)
};

constexpr int FT_LS = 10;          // line pitch: the M entries of residue p at 10 p (80-byte steps: the eight 16-byte reads of a group of lanes hit eight different banks)
constexpr int FT_LINE = 8 * FT_LS + 64;   // doubles per line buffer: [8][FT_LS] by (residue, slot) + [64] by ring position

// is the register pair (rs, cs) live while the pivot is in slot SJ?  (lower triangle in block units)
template <int M>
__host__ __device__ constexpr bool ft_live(int rs, int cs, int SJ) { return (rs - SJ + M) % M >= (cs - SJ + M) % M; }

// the eight lanes that hold column j1 (q = pj1) lay it out for everybody: by (residue, slot) for the multipliers / pivot-row entries of the
// tiles, by ring position for the right-hand sides and the store of L.  Rows of the pivot's own block at or before the pivot (residue <= pj1
// in slot CS1) are finished: they go out as zeros, so no reader masks anything.
template <int M, int CS1>
__device__ __forceinline__ void ft_publish(const FtRegs<M>& st, int p, int q, int pj1, double* __restrict__ nxt) {
  if (q == pj1) {
    double v[M];
#pragma unroll
    for (int rs = 0; rs < M; ++rs) v[rs] = st.r[rs][CS1];
    v[CS1] = p > pj1 ? v[CS1] : 0.0;
#pragma unroll
    for (int rs = 0; rs + 1 < M; rs += 2) *reinterpret_cast<double2*>(nxt + FT_LS * p + rs) = double2{v[rs], v[rs + 1]};
    if (M & 1) nxt[FT_LS * p + M - 1] = v[M - 1];
#pragma unroll
    for (int rs = 0; rs < M; ++rs) nxt[8 * FT_LS + 8 * rs + p] = v[rs];
  }
}

// One elimination step.  SJ: slot of the pivot's block; LAST: the block's eighth step (pj = 7; the slot SJ already holds the NEXT group: its
// rows take no update -- their multipliers are zeros of the line --, the next pivot column is in slot SJ + 1).  lv / av: this step's line values
// (rows / columns of this lane), replaced by the next step's on return; rd = 1 / d_j on entry, 1 / d_(j+1) on return; good: lane mask, all
// ones while every pivot was positive.
template <int M, int W, int SJ, bool LAST>
__device__ __forceinline__ void ft_step(FtRegs<M>& st, double (&lv)[M], double (&av)[M], double& rd, unsigned long long& good, int j, int pj, int lane, int n,
                                        int kd, double* __restrict__ line, double* __restrict__ Lc, double* __restrict__ xs, double y_last = 0.0) {
  constexpr int N = 8 * M, CS1 = LAST ? (SJ + 1) % M : SJ;
  const int p = lane >> 3, q = lane & 7;
  const double rdj = rd;
  double l[M];
#pragma unroll
  for (int s = 0; s < M; ++s) l[s] = lv[s] * rdj;
  // column j + 1 first: its line and the reciprocal of its pivot start here
#pragma unroll
  for (int rs = 0; rs < M; ++rs)
    if (ft_live<M>(rs, CS1, SJ) && !(LAST && rs == SJ)) st.r[rs][CS1] = __builtin_fma(-l[rs], av[CS1], st.r[rs][CS1]);
  double* cur = line + (j & 1) * FT_LINE;
  double* nxt = line + ((j + 1) & 1) * FT_LINE;
  const int pj1 = LAST ? 0 : pj + 1;
  ft_publish<M, CS1>(st, p, q, pj1, nxt);
  const double d1 = fw_readlane(st.r[CS1][CS1], 9 * pj1);
  rd = frcp(d1);                                             // (rows between the last equation and its group's end carry a unit diagonal: frame_plan_kernel)
  good &= __builtin_amdgcn_fcmp(d1, 0.0, 2 /* ogt */) | (j + 1 < n ? 0ull : ~0ull);
  // right-hand side and column j of L: the lane at ring position x works for the row there (this step's line, written one step ago)
  {
    const double ly = cur[8 * FT_LS + lane] * rdj;            // (positions >= N of the line stay zero)
    const double yj = LAST ? y_last : fw_readlane(st.y, 8 * SJ + pj);   // (LAST: the pivot row's lane already carries the next group's row)
    if (lane == 0) xs[j] = yj * rdj;                          // w_j = z_j / d_j
    st.y = __builtin_fma(-ly, yj, st.y);
#ifndef FW_SKIP_LSTORE
    int rel = lane - (8 * SJ + pj);
    rel = rel < 0 ? rel + N : rel;
    if ((N == 64 || lane < N) && (unsigned)(rel - 1) < (unsigned)kd) Lc[(size_t)j * W + rel - 1] = ly;
#endif
  }
  // the other columns
#pragma unroll
  for (int cs = 0; cs < M; ++cs) {
    if (cs == CS1 || (LAST && cs == SJ)) continue;          // (last step of a block: its own columns are finished)
#pragma unroll
    for (int rs = 0; rs < M; ++rs)
      if (ft_live<M>(rs, cs, SJ) && !(LAST && rs == SJ)) st.r[rs][cs] = __builtin_fma(-l[rs], av[cs], st.r[rs][cs]);
  }
  fw_fence();                                                 // the next line has landed
#pragma unroll
  for (int s = 0; s + 1 < M; s += 2) {
    const double2 u = *reinterpret_cast<const double2*>(nxt + FT_LS * p + s), v = *reinterpret_cast<const double2*>(nxt + FT_LS * q + s);
    lv[s] = u.x; lv[s + 1] = u.y; av[s] = v.x; av[s + 1] = v.y;
  }
  if (M & 1) { lv[M - 1] = nxt[FT_LS * p + M - 1]; av[M - 1] = nxt[FT_LS * q + M - 1]; }
}

// the parked group (rows 8 g + p) into ring slot S
template <int M, int S>
__device__ __forceinline__ void ft_take(FtRegs<M>& st, int lane, const double* __restrict__ stage) {
  const int p = lane >> 3, q = lane & 7;
#pragma unroll
  for (int i = 0; 2 * i < M; ++i) {
    const double2 v = *reinterpret_cast<const double2*>(stage + p * FT_P + 16 * i + 2 * q);
    st.r[S][2 * i] = v.x;
    if (2 * i + 1 < M) st.r[S][2 * i + 1] = v.y;
  }
  // the group's own 8 x 8 block: parked rows hold the lower triangle; the entry above the diagonal comes from the transposed position
  const double t = stage[q * FT_P + ft_col_slot(8 * S + p)];
  st.r[S][S] += p < q ? t : 0.0;
  if (p == S) st.y = stage[q * FT_P + 64];                    // lane 8 S + r: ring position of row r of the group
}

template <int M, int W>
__device__ __forceinline__ void frame_tile_body(const FrameParams& p, double* __restrict__ wsf, double* __restrict__ lds, int lane, long b,
                                                const FwPlan& pl) {
  constexpr int G = FW_G;
  static_assert(G == 8, "frame_tile: groups of eight rows");
  const int n = p.n_eq, kd = p.kd;
  double* line = lds;                                       // [2][FT_LINE]
  double* stage = lds + 2 * FT_LINE;                        // [8][FT_P]
  double* xs = stage + (size_t)G * FT_P;                    // [n + 64]: w, then x
  double* rows = wsf;                                       // [fw_rows(n)][W]: column j of L at row j
  FtRegs<M> st;
#pragma unroll
  for (int r = 0; r < M; ++r)
#pragma unroll
    for (int c = 0; c < M; ++c) st.r[r][c] = 0.0;
  st.y = 0.0;
  unsigned long long good = ~0ull;
  for (int i = lane; i < 2 * FT_LINE; i += 64) line[i] = 0.0;

  // ---- fused assembly of the next group into the parking area (as frame_wave.hpp; parking slots: frame_plan_kernel, ring = N) ----
  const double* Ib = p.I + b * p.Ne;
  const double* lb = p.loads + b * p.loads_bs;
  constexpr int KE = FW_KE;
  unsigned eB[KE] = {0u, 0u, 0u};
  int dofB = 0, gB = 0;
  double bi[KE], ba[KE], bb[KE], by1 = 0.0, by2 = 0.0;
  const int n_extra = pl.hdr[0];
  auto ents = [&](int g0) {
    const int gi = g0 / G < pl.ng ? g0 / G : pl.ng;
    const unsigned* e = pl.ent + (size_t)gi * FW_EPG + lane;
#pragma unroll
    for (int k = 0; k < KE; ++k) eB[k] = e[64 * k];
    const int r = g0 + (lane < G ? lane : 0);
    dofB = pl.eq_dof[r < n ? r : n];
    gB = g0;
  };
  auto build_issue = [&]() {
    const int gi = gB / G < pl.ng ? gB / G : pl.ng;
    const double* ka = pl.ka + (size_t)gi * FW_EPG + lane;
    const double* kb = pl.kb + (size_t)gi * FW_EPG + lane;
#pragma unroll
    for (int k = 0; k < KE; ++k) { bi[k] = Ib[(eB[k] >> FW_SLOT_BITS) & 0x1FFFFF]; ba[k] = ka[64 * k]; bb[k] = kb[64 * k]; }
    const int r = gB + (lane < G ? lane : 0);
    by1 = pl.rhs_base[r < n ? r : n];
    by2 = lb[dofB];
  };
  auto build_finish = [&]() {
#pragma unroll
    for (int k = 0; k < KE; ++k)
      if ((int)eB[k] < 0) atomicAdd(&stage[eB[k] & FW_SLOT_MASK], __builtin_fma(bi[k], bb[k], ba[k]));
    if (n_extra != 0) {
      const int gi = gB / G < pl.ng ? gB / G : pl.ng;
      for (int blk = pl.xstart[gi]; blk < pl.xstart[gi + 1]; ++blk)
        for (int k = 0; k < KE; ++k) {
          const size_t i = (size_t)(pl.ng + 1 + blk) * FW_EPG + lane + 64 * k;
          const unsigned w = pl.ent[i];
          if ((int)w < 0) atomicAdd(&stage[w & FW_SLOT_MASK], __builtin_fma(Ib[(w >> FW_SLOT_BITS) & 0x1FFFFF], pl.kb[i], pl.ka[i]));
        }
    }
    fw_fence();
    if (lane < G) stage[lane * FT_P + 64] = (gB + lane < n) ? by1 + by2 : 0.0;
  };
  auto zero_stage = [&]() {
#pragma unroll
    for (int i = 0; i < G * FT_P / 128; ++i) reinterpret_cast<double2*>(stage)[lane + 64 * i] = double2{0.0, 0.0};
    fw_fence();
  };
  static_assert((G * FT_P) % 128 == 0, "frame_tile: parking area zeroed in 16-byte pieces");

  // prologue: groups 0 .. M - 1 into the ring, group M parked, the entry words of group M + 1 on their way
  auto prologue = [&](auto Sc) {
    constexpr int S = decltype(Sc)::value;
    ents(G * S);
    build_issue();
    zero_stage();
    build_finish();
    fw_fence();
    if constexpr (S < M) { ft_take<M, S>(st, lane, stage); fw_fence(); }
  };
  prologue(std::integral_constant<int, 0>{});
  prologue(std::integral_constant<int, 1>{});
  prologue(std::integral_constant<int, 2>{});
  if constexpr (M >= 3) prologue(std::integral_constant<int, 3>{});
  if constexpr (M >= 4) prologue(std::integral_constant<int, 4>{});
  if constexpr (M >= 5) prologue(std::integral_constant<int, 5>{});
  if constexpr (M >= 6) prologue(std::integral_constant<int, 6>{});
  if constexpr (M >= 7) prologue(std::integral_constant<int, 7>{});
  if constexpr (M >= 8) prologue(std::integral_constant<int, 8>{});
  ents(G * (M + 1));

  // ---- factorisation + forward substitution ----
  double rd, lv[M], av[M];
  {
    const int pp = lane >> 3, q = lane & 7;
    fw_fence();
    ft_publish<M, 0>(st, pp, q, 0, line);
    const double d0 = fw_readlane(st.r[0][0], 0);
    rd = frcp(d0);
    good &= __builtin_amdgcn_fcmp(d0, 0.0, 2);
    fw_fence();
#pragma unroll
    for (int s = 0; s < M; ++s) { lv[s] = line[FT_LS * pp + s]; av[s] = line[FT_LS * q + s]; }
  }
  int jb = 0;                                               // block: steps 8 jb .. 8 jb + 7, pivot slot jb mod M
  auto block = [&](auto Sc) {
    constexpr int SJ = decltype(Sc)::value;
    const int j0 = 8 * jb;
#pragma unroll 1
    for (int pj = 0; pj < 7; ++pj) ft_step<M, W, SJ, false>(st, lv, av, rd, good, j0 + pj, pj, lane, n, kd, line, rows, xs);
    // the block's rows are finished: group jb + M enters their slot (parked one boundary ago), the group after it is assembled
    build_issue();
    const double y7 = fw_readlane(st.y, 8 * SJ + 7);             // right-hand side of the block's last row, before its lane is taken over
    ft_take<M, SJ>(st, lane, stage);
    fw_fence();
    zero_stage();
    build_finish();
    fw_fence();
    ents(G * (jb + M + 2));
    ft_step<M, W, SJ, true>(st, lv, av, rd, good, j0 + 7, 7, lane, n, kd, line, rows, xs, y7);
    ++jb;
  };
  while (true) {
    block(std::integral_constant<int, 0>{});
    if (8 * jb >= n) break;
    block(std::integral_constant<int, 1>{});
    if (8 * jb >= n) break;
    block(std::integral_constant<int, 2>{});
    if (8 * jb >= n) break;
    if constexpr (M >= 4) { block(std::integral_constant<int, 3>{}); if (8 * jb >= n) break; }
    if constexpr (M >= 5) { block(std::integral_constant<int, 4>{}); if (8 * jb >= n) break; }
    if constexpr (M >= 6) { block(std::integral_constant<int, 5>{}); if (8 * jb >= n) break; }
    if constexpr (M >= 7) { block(std::integral_constant<int, 6>{}); if (8 * jb >= n) break; }
    if constexpr (M >= 8) { block(std::integral_constant<int, 7>{}); if (8 * jb >= n) break; }
  }
  fw_fence();

  fw_backward<W>(rows, xs, n, kd, lane);
  write_results(p, b, xs, good != ~0ull, lane, 64);
}

// waves per SIMD the register allocator is asked to make room for
#ifndef FT_WAVES_8
#define FT_WAVES_8 3
#endif
#ifndef FT_WAVES_6
#define FT_WAVES_6 3
#endif
constexpr int ft_waves(int W) { return ft_M(W) >= 8 ? FT_WAVES_8 : ft_M(W) >= 6 ? FT_WAVES_6 : 3; }

template <int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ft_waves(W))))
void frame_tile_kernel(const FrameParams p, double* __restrict__ ws, const FwPlan pl) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + wave;
  if (b >= p.B) return;
  frame_tile_body<ft_M(W), W>(p, ws + b * fw_frame_doubles(p.n_eq, p.kd), lds + (size_t)wave * ft_lds_doubles(p.n_eq), lane, b, pl);
}

}  // namespace opsamd
