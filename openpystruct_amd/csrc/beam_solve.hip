// Batched Euler-Bernoulli beam FE solve for gfx950 (MI355X): assembly, Dirichlet handling,
// block-tridiagonal Cholesky solve and end-force recovery fused in one kernel.
//
// Replaces, for B beams per launch, the per-case command sequence of the reference's
// `setup_model` + `ops.analyze(1)` + `ops.eleResponse` + `ops.nodeDisp`
// (/root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py:89-124, :180-190, :224-232).
// C ABI: include/openpystruct_amd.h.  Arithmetic: beam_math.hpp.  Design: DESIGN.md.
//
// Mapping: one 64-lane wavefront per workgroup; P lanes co-operate on one beam
// (P = 64 is "one wavefront per beam"), each lane owning M consecutive elements whose
// condensation state lives in its VGPRs.  The wave's inputs (I, Fy rows of its 64/P
// consecutive beams: contiguous in HBM) are staged through LDS with coalesced loads;
// the unit-inertia element stiffness tile entries (2E/L, 6E/L^2, 12E/L^3), 1/L and the
// consistent UDL loads are computed once per workgroup into an LDS table; outputs go
// back through LDS to coalesced row stores.  FP64 throughout; no MFMA (no dense
// contraction on this path).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"
#include "beam_math.hpp"

namespace opsamd {

struct BeamParams {
  int B, Ne;
  const double* x;  long x_bs;
  const double* E;  long E_bs;
  const double* I;  long I_bs;
  const uint8_t* fix; long fix_bs;
  const double* Fy; long Fy_bs;
  const double* wy; long wy_bs;
  double* v; double* theta; double* V; double* M;
  int32_t* status;
  // host-derived: rows of I/Fy/outputs are dense and every wave's chunk is 16-byte aligned, so
  // the wave moves its beams as one flat run of 16-byte accesses; magic numbers for idx / Ne, idx / N
  int dense;
  unsigned magic_ne, magic_n;
};

// lane-local view of the LDS-staged inputs (see beam_math.hpp "Acc")
struct LdsAcc {
  const double* t2; const double* t6; const double* t12; const double* trl; const double* tpw; const double* tmw;
  const double* sI; const double* sF;
  unsigned long long bits;
  __device__ __forceinline__ double c2(int i) const { return t2[i]; }
  __device__ __forceinline__ double c6(int i) const { return t6[i]; }
  __device__ __forceinline__ double c12(int i) const { return t12[i]; }
  __device__ __forceinline__ double rL(int i) const { return trl[i]; }
  __device__ __forceinline__ double pw(int i) const { return tpw[i]; }
  __device__ __forceinline__ double mw(int i) const { return tmw[i]; }
  __device__ __forceinline__ double Ie(int i) const { return sI[i]; }
  __device__ __forceinline__ double Fy(int i) const { return sF[i]; }
  __device__ __forceinline__ unsigned long long fixbits() const { return bits; }
  __device__ __forceinline__ void fence() const { __asm__ volatile("" ::: "memory"); }
};

template <int M>
struct LaneOut {
  double* sV; double* sM;   // LDS slots of the lane's elements
  double v[M], th[M];       // nodal results stay in registers until the element rows are stored
  __device__ __forceinline__ void elem(int i, double Vv, double Mv) { sV[i] = Vv; sM[i] = Mv; }
  __device__ __forceinline__ void node(int i, double vv, double tt) { v[i] = vv; th[i] = tt; }
};

// ---- cross-lane exchange inside the P-lane group of a beam ------------------------------
// from_minus<S>(x): value of lane-S (0.0 when j < S); from_plus<S>(x): value of lane+S (0.0 when
// j + S >= P).  P <= 16: the group lies inside one 16-lane DPP row, so the fetch is a pair of
// v_mov_b32 with a row_shr / row_shl modifier (bound_ctrl writes 0 for lanes shifted in from outside
// the row); no LDS crossbar, no wait.  P = 8 shares its row with a second beam and masks the lanes
// that would read across the group edge.  P >= 32: ds_bpermute (__shfl).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int P>
struct Xch {
  template <int S>
  static __device__ __forceinline__ double from_minus(double x, int lane, int j) {
    if constexpr (P <= 16 && S < 16) {
      const double r = dpp_mov<0x110 + S>(x);  // row_shr:S
      if constexpr (P < 16) return j >= S ? r : 0.0;
      return r;
    } else {
      const double r = __shfl(x, lane - S, 64);
      return j >= S ? r : 0.0;
    }
  }
  template <int S>
  static __device__ __forceinline__ double from_plus(double x, int lane, int j) {
    if constexpr (P <= 16 && S < 16) {
      const double r = dpp_mov<0x100 + S>(x);  // row_shl:S
      if constexpr (P < 16) return j + S < P ? r : 0.0;
      return r;
    } else {
      const double r = __shfl(x, lane + S, 64);
      return j + S < P ? r : 0.0;
    }
  }
  template <int S> static __device__ __forceinline__ Sym2 from_minus(const Sym2& s, int l, int j) {
    return Sym2{from_minus<S>(s.a, l, j), from_minus<S>(s.b, l, j), from_minus<S>(s.c, l, j)};
  }
  template <int S> static __device__ __forceinline__ Mat2 from_minus(const Mat2& m, int l, int j) {
    return Mat2{from_minus<S>(m.a, l, j), from_minus<S>(m.b, l, j), from_minus<S>(m.c, l, j), from_minus<S>(m.d, l, j)};
  }
  template <int S> static __device__ __forceinline__ Vec2 from_minus(const Vec2& u, int l, int j) {
    return Vec2{from_minus<S>(u.x, l, j), from_minus<S>(u.y, l, j)};
  }
  template <int S> static __device__ __forceinline__ Sym2 from_plus(const Sym2& s, int l, int j) {
    return Sym2{from_plus<S>(s.a, l, j), from_plus<S>(s.b, l, j), from_plus<S>(s.c, l, j)};
  }
  template <int S> static __device__ __forceinline__ Mat2 from_plus(const Mat2& m, int l, int j) {
    return Mat2{from_plus<S>(m.a, l, j), from_plus<S>(m.b, l, j), from_plus<S>(m.c, l, j), from_plus<S>(m.d, l, j)};
  }
  template <int S> static __device__ __forceinline__ Vec2 from_plus(const Vec2& u, int l, int j) {
    return Vec2{from_plus<S>(u.x, l, j), from_plus<S>(u.y, l, j)};
  }
};

// parallel cyclic reduction over the P rows of a beam: steps S = 1, 2, 4, ..., P/2
template <int P, int S>
__device__ __forceinline__ void pcr_all(IfaceRow& row, int lane, int j, int& bad) {
  if constexpr (S < P) {
    using X = Xch<P>;
    constexpr bool LAST = (2 * S >= P);
    const Sym2 G = inv_spd(row.D, bad);
    const Sym2 Gm = X::template from_minus<S>(G, lane, j);
    const Vec2 fm = X::template from_minus<S>(row.f, lane, j);
    const Sym2 Gp = X::template from_plus<S>(G, lane, j);
    const Vec2 fp = X::template from_plus<S>(row.f, lane, j);
    Mat2 Am{0, 0, 0, 0}, Cp{0, 0, 0, 0};
    if constexpr (!LAST) {
      Am = X::template from_minus<S>(row.Alow, lane, j);
      Cp = X::template from_plus<S>(row.Cup, lane, j);
    }
    pcr_step<LAST>(row, Gm, Am, fm, Gp, Cp, fp);
    pcr_all<P, 2 * S>(row, lane, j, bad);
  }
}

// SHARED: x, E and wy are the same for every beam (strides 0): one LDS table per workgroup.
template <int P, int M, bool SHARED>
__global__ __launch_bounds__(64) void beam_solve_kernel(const BeamParams p) {
  constexpr int BPW = 64 / P;       // beams per wavefront
  constexpr int PM = P * M;         // padded nodes per beam (>= N)
  constexpr int TG = SHARED ? 1 : BPW;
  __shared__ double s_tab[6][TG][PM];
  __shared__ double s_a[BPW][PM];   // I      -> M  -> theta
  __shared__ double s_b[BPW][PM];   // Fy     -> V  -> v
  __shared__ uint8_t s_fix[BPW][PM + 8];
  __shared__ int s_bad[BPW];

  const int lane = threadIdx.x;
  const int Ne = p.Ne, N = p.Ne + 1;
  const long beam0 = (long)blockIdx.x * BPW;

  constexpr int NPAIR = (BPW * PM / 2 + 63) / 64;   // 16-byte pieces per lane for one staged array
  const int nb = (p.B - beam0 < BPW) ? (int)(p.B - beam0) : BPW;   // live beams of this wave
  // ---- stage 1a: issue the wave's global loads first (I, Fy rows of its beams: one contiguous run) ----
  double2 rI[NPAIR], rF[NPAIR];
  if (p.dense) {
    const double* gI = p.I + beam0 * Ne;
    const double* gF = p.Fy + beam0 * N;
    const int nI = nb * Ne, nF = nb * N;
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const int i0 = 2 * (lane + 64 * k);
      rI[k] = double2{0.0, 0.0};
      rF[k] = double2{0.0, 0.0};
      if (i0 + 1 < nI) rI[k] = *reinterpret_cast<const double2*>(gI + i0);
      else if (i0 < nI) rI[k].x = gI[i0];
      if (i0 + 1 < nF) rF[k] = *reinterpret_cast<const double2*>(gF + i0);
      else if (i0 < nF) rF[k].x = gF[i0];
    }
  }
  // zero fill (padding elements / nodes and dead beams); LDS ops of one wave execute in order
  for (int idx = lane; idx < BPW * PM; idx += 64) {
    (&s_a[0][0])[idx] = 0.0;
    (&s_b[0][0])[idx] = 0.0;
  }
  // ---- stage 0: element table (unit-inertia stiffness tile entries, 1/L, UDL loads) ----
  for (int idx = lane; idx < TG * PM; idx += 64) {
    const int tb = idx / PM, e = idx - tb * PM;
    long bb = beam0 + tb;
    if (bb >= p.B) bb = p.B - 1;
    double c2 = 0.0, c6 = 0.0, c12 = 0.0, rl = 0.0, pw = 0.0, mw = 0.0;
    if (e < Ne) {
      const double* xb = p.x + bb * p.x_bs;
      const double L = xb[e + 1] - xb[e];
      const double Ee = p.E_bs ? p.E[bb * p.E_bs + e] : p.E[0];
      const double w = p.wy_bs ? p.wy[bb * p.wy_bs + e] : p.wy[0];
      rl = fast_rcp(L);
      c2 = 2.0 * Ee * rl;
      c6 = 3.0 * c2 * rl;
      c12 = 2.0 * c6 * rl;
      pw = 0.5 * w * L;
      mw = pw * L * (1.0 / 6.0);
    }
    s_tab[0][tb][e] = c2;  s_tab[1][tb][e] = c6;  s_tab[2][tb][e] = c12;
    s_tab[3][tb][e] = rl;  s_tab[4][tb][e] = pw;  s_tab[5][tb][e] = mw;
  }
  // ---- stage 1b: constraint bytes, then the staged rows into their padded LDS rows ----
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    const bool live = b < nb;
    const uint8_t* fb = p.fix + (live ? beam0 + b : 0) * p.fix_bs;
    for (int e = lane; e < PM + 8; e += 64) s_fix[b][e] = (live && e < N) ? (uint8_t)(fb[e] & 3) : (uint8_t)3;
  }
  if (p.dense) {
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const int i0 = 2 * (lane + 64 * k);
      if (i0 < nb * Ne) {
        const int b0 = (int)(((unsigned)i0 * p.magic_ne) >> 20), e = i0 - b0 * Ne;
        s_a[b0][e] = rI[k].x;
        if (e + 1 < Ne) s_a[b0][e + 1] = rI[k].y;
        else if (b0 + 1 < nb) s_a[b0 + 1][0] = rI[k].y;
      }
      if (i0 < nb * N) {
        const int b0 = (int)(((unsigned)i0 * p.magic_n) >> 20), e = i0 - b0 * N;
        s_b[b0][e] = rF[k].x;
        if (e + 1 < N) s_b[b0][e + 1] = rF[k].y;
        else if (b0 + 1 < nb) s_b[b0 + 1][0] = rF[k].y;
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      if (b >= nb) break;
      const double* Ib = p.I + (beam0 + b) * p.I_bs;
      const double* Fb = p.Fy + (beam0 + b) * p.Fy_bs;
      for (int e = lane; e < N; e += 64) {
        if (e < Ne) s_a[b][e] = Ib[e];
        s_b[b][e] = Fb[e];
      }
    }
  }
  if (lane < BPW) s_bad[lane] = 0;
  __syncthreads();

  // ---- stage 2: per-lane condensation of the segment interior ----
  const int g = lane / P, j = lane - g * P, e0 = j * M;
  LdsAcc acc;
  {
    const int tb = SHARED ? 0 : g;
    acc.t2 = &s_tab[0][tb][e0];  acc.t6 = &s_tab[1][tb][e0];  acc.t12 = &s_tab[2][tb][e0];
    acc.trl = &s_tab[3][tb][e0]; acc.tpw = &s_tab[4][tb][e0]; acc.tmw = &s_tab[5][tb][e0];
    acc.sI = &s_a[g][e0];
    acc.sF = &s_b[g][e0];
    unsigned long long bits = 0;
#pragma unroll
    for (int i = 0; i <= M; ++i) bits |= (unsigned long long)s_fix[g][e0 + i] << (2 * i);
    acc.bits = bits;
  }
  int bad = 0;
  SegState<M> st;
  seg_condense<M>(st, acc, bad);
  // The element data is re-read from LDS in stage 4 instead of being carried in ~16*M VGPRs
  // across the reduction: the clobber stops the compiler from merging the two sets of loads.
  __asm__ volatile("" ::: "memory");

  // ---- stage 3: interface system over the P lanes of the beam, parallel cyclic reduction ----
  using X = Xch<P>;
  IfaceRow row;
  {
    const Mat2 cup = masked_cup<M>(st, acc.bits);
    const Sym2 pc = X::template from_minus<1>(st.Scc, lane, j);
    const Vec2 pg = X::template from_minus<1>(st.gc, lane, j);
    const Mat2 pb = X::template from_minus<1>(cup, lane, j);
    row = make_row<M>(st, cup, pc, pg, pb, acc.bits);
  }
  pcr_all<P, 1>(row, lane, j, bad);
  const Vec2 uL = mul(inv_spd(row.D, bad), row.f);
  const Vec2 uR = X::template from_plus<1>(uL, lane, j);

  // ---- stage 4: interior solve + end forces; element rows out first ----
  LaneOut<M> out;
  out.sV = &s_b[g][e0];
  out.sM = &s_a[g][e0];
  seg_solve<M>(st, acc, uL, uR, out);
  if (bad) s_bad[g] = 1;
  __syncthreads();

  const double qnan = __builtin_nan("");
  // element rows (V in s_b, M in s_a) -> global
  if (p.dense) {
    double* gV = p.V + beam0 * Ne;
    double* gM = p.M + beam0 * Ne;
    const int nE = nb * Ne;
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const int i0 = 2 * (lane + 64 * k);
      if (i0 < nE) {
        const int b0 = (int)(((unsigned)i0 * p.magic_ne) >> 20), e = i0 - b0 * Ne;
        const bool wrap = e + 1 >= Ne;
        const int b1 = wrap ? b0 + 1 : b0, e1 = wrap ? 0 : e + 1;
        double2 vV{s_bad[b0] ? qnan : s_b[b0][e], 0.0}, vM{s_bad[b0] ? qnan : s_a[b0][e], 0.0};
        if (i0 + 1 < nE) {
          vV.y = s_bad[b1] ? qnan : s_b[b1][e1];
          vM.y = s_bad[b1] ? qnan : s_a[b1][e1];
          *reinterpret_cast<double2*>(gV + i0) = vV;
          *reinterpret_cast<double2*>(gM + i0) = vM;
        } else {
          gV[i0] = vV.x;
          gM[i0] = vM.x;
        }
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      if (b >= nb) break;
      const bool nbad = s_bad[b] != 0;
      for (int e = lane; e < Ne; e += 64) {
        p.V[(beam0 + b) * Ne + e] = nbad ? qnan : s_b[b][e];
        p.M[(beam0 + b) * Ne + e] = nbad ? qnan : s_a[b][e];
      }
    }
  }
  __syncthreads();
  // nodal rows: v -> s_b, theta -> s_a -> global
#pragma unroll
  for (int i = 0; i < M; ++i) {
    s_b[g][e0 + i] = out.v[i];
    s_a[g][e0 + i] = out.th[i];
  }
  __syncthreads();
  if (p.dense) {
    double* gv = p.v + beam0 * N;
    double* gt = p.theta + beam0 * N;
    const int nN = nb * N;
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const int i0 = 2 * (lane + 64 * k);
      if (i0 < nN) {
        const int b0 = (int)(((unsigned)i0 * p.magic_n) >> 20), e = i0 - b0 * N;
        const bool wrap = e + 1 >= N;
        const int b1 = wrap ? b0 + 1 : b0, e1 = wrap ? 0 : e + 1;
        double2 vv{s_bad[b0] ? qnan : s_b[b0][e], 0.0}, vt{s_bad[b0] ? qnan : s_a[b0][e], 0.0};
        if (i0 + 1 < nN) {
          vv.y = s_bad[b1] ? qnan : s_b[b1][e1];
          vt.y = s_bad[b1] ? qnan : s_a[b1][e1];
          *reinterpret_cast<double2*>(gv + i0) = vv;
          *reinterpret_cast<double2*>(gt + i0) = vt;
        } else {
          gv[i0] = vv.x;
          gt[i0] = vt.x;
        }
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      if (b >= nb) break;
      const bool nbad = s_bad[b] != 0;
      for (int e = lane; e < N; e += 64) {
        p.v[(beam0 + b) * N + e] = nbad ? qnan : s_b[b][e];
        p.theta[(beam0 + b) * N + e] = nbad ? qnan : s_a[b][e];
      }
    }
  }
  if (lane < nb && p.status) p.status[beam0 + lane] = s_bad[lane] ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// host side: tiling choice + launch
// ------------------------------------------------------------------------------------------
struct Tiling { int P, M; const char* name_shared; const char* name_general; };

// every compiled (P, M); a tiling serves Ne with Ne + 1 <= P * M
static const Tiling kTilings[] = {
    {8, 13, "beam_solve_kernel<8, 13, true>", "beam_solve_kernel<8, 13, false>"},
    {16, 7, "beam_solve_kernel<16, 7, true>", "beam_solve_kernel<16, 7, false>"},
    {32, 4, "beam_solve_kernel<32, 4, true>", "beam_solve_kernel<32, 4, false>"},
    {64, 2, "beam_solve_kernel<64, 2, true>", "beam_solve_kernel<64, 2, false>"},
    {64, 4, "beam_solve_kernel<64, 4, true>", "beam_solve_kernel<64, 4, false>"},
    {64, 8, "beam_solve_kernel<64, 8, true>", "beam_solve_kernel<64, 8, false>"},
    {64, 16, "beam_solve_kernel<64, 16, true>", "beam_solve_kernel<64, 16, false>"},
};
static const int kNumTilings = sizeof(kTilings) / sizeof(kTilings[0]);

static const Tiling* choose_tiling(int B, int Ne, int tiling) {
  const int N = Ne + 1;
  if (tiling != 0) {
    for (int t = 0; t < kNumTilings; ++t)
      if (kTilings[t].P == tiling && kTilings[t].P * kTilings[t].M >= N) return &kTilings[t];
    return nullptr;
  }
  (void)B;
  // default: 16 lanes per beam where it fits (best at the 10^4-beam batch of BASELINE config 2),
  // otherwise the narrowest tiling that holds the beam
  if (N <= 16 * 7) return &kTilings[1];
  for (int t = 0; t < kNumTilings; ++t)
    if (kTilings[t].P * kTilings[t].M >= N) return &kTilings[t];
  return nullptr;
}

static thread_local char g_last_error[256] = {0};

template <int P, int M>
static hipError_t launch(const BeamParams& p, bool shared, hipStream_t stream) {
  constexpr int BPW = 64 / P;
  const unsigned grid = (unsigned)((p.B + BPW - 1) / BPW);
  if (shared)
    hipLaunchKernelGGL((beam_solve_kernel<P, M, true>), dim3(grid), dim3(64), 0, stream, p);
  else
    hipLaunchKernelGGL((beam_solve_kernel<P, M, false>), dim3(grid), dim3(64), 0, stream, p);
  return hipGetLastError();
}

}  // namespace opsamd

using namespace opsamd;

extern "C" {

int ops_amd_abi_version(void) { return OPS_AMD_ABI_VERSION; }
int ops_amd_max_elements(void) { return 64 * 16 - 1; }
const char* ops_amd_last_error(void) { return g_last_error; }

const char* ops_beam_solve_kernel_name(int B, int Ne, int tiling) {
  const Tiling* t = choose_tiling(B, Ne, tiling);
  return t ? t->name_shared : "";
}

int ops_beam_solve_batched_f64(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                               const double* I, long I_bstride, const uint8_t* fix, long fix_bstride,
                               const double* Fy, long Fy_bstride, const double* wy, long wy_bstride, double* v,
                               double* theta, double* V, double* M, int32_t* status, int tiling, void* stream) {
  if (B < 0 || Ne < 1) return OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!x || !E || !I || !fix || !Fy || !wy || !v || !theta || !V || !M) return OPS_AMD_ERR_INVALID_ARG;
  if (I_bstride < Ne || Fy_bstride < Ne + 1) return OPS_AMD_ERR_INVALID_ARG;
  if ((x_bstride != 0 && x_bstride < Ne + 1) || (fix_bstride != 0 && fix_bstride < Ne + 1) ||
      (E_bstride != 0 && E_bstride < Ne) || (wy_bstride != 0 && wy_bstride < Ne))
    return OPS_AMD_ERR_INVALID_ARG;
  if (Ne > ops_amd_max_elements()) return OPS_AMD_ERR_UNSUPPORTED;
  const Tiling* t = choose_tiling(B, Ne, tiling);
  if (!t) return tiling ? OPS_AMD_ERR_INVALID_ARG : OPS_AMD_ERR_UNSUPPORTED;

  BeamParams p{B, Ne, x, x_bstride, E, E_bstride, I, I_bstride, fix, fix_bstride, Fy, Fy_bstride,
               wy, wy_bstride, v, theta, V, M, status, 0, 0u, 0u};
  {
    const int bpw = 64 / t->P;
    const uintptr_t bits = (uintptr_t)I | (uintptr_t)Fy | (uintptr_t)v | (uintptr_t)theta | (uintptr_t)V | (uintptr_t)M;
    p.dense = (I_bstride == Ne) && (Fy_bstride == Ne + 1) && ((bits & 15u) == 0) &&
              ((bpw * Ne) % 2 == 0) && ((bpw * (Ne + 1)) % 2 == 0);
    p.magic_ne = ((1u << 20) + (unsigned)Ne - 1u) / (unsigned)Ne;
    p.magic_n = ((1u << 20) + (unsigned)Ne) / (unsigned)(Ne + 1);
  }
  const bool shared = (x_bstride == 0 && E_bstride == 0 && wy_bstride == 0);
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipSuccess;
  if (t->P == 8 && t->M == 13) err = launch<8, 13>(p, shared, s);
  else if (t->P == 16 && t->M == 7) err = launch<16, 7>(p, shared, s);
  else if (t->P == 32 && t->M == 4) err = launch<32, 4>(p, shared, s);
  else if (t->P == 64 && t->M == 2) err = launch<64, 2>(p, shared, s);
  else if (t->P == 64 && t->M == 4) err = launch<64, 4>(p, shared, s);
  else if (t->P == 64 && t->M == 8) err = launch<64, 8>(p, shared, s);
  else if (t->P == 64 && t->M == 16) err = launch<64, 16>(p, shared, s);
  if (err != hipSuccess) {
    const char* msg = hipGetErrorString(err);
    int k = 0;
    for (; msg && msg[k] && k < 255; ++k) g_last_error[k] = msg[k];
    g_last_error[k] = 0;
    return OPS_AMD_ERR_LAUNCH;
  }
  return OPS_AMD_OK;
}

}  // extern "C"
