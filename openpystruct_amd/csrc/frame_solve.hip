// Batched 2-D elastic frame solve (3 DOF per node): assembly of rotated ElasticBeam2d stiffness matrices,
// banded LDL^T factorisation (three columns per workgroup barrier), substitution and global end-force recovery, one
// workgroup per frame with the band matrix resident in LDS.
//
// SURVEY section 8(f1) / BASELINE config 5: generalises the beam path to the model built by
// `setup_frame_model` (/root/reference/OpenPyStruct_FrameOpt_Discrete_Beta.py:75-139): columns and beams of
// a rectangular grid, ground row fully fixed (:96-98), lateral nodal loads (:126-128), `beamUniform(w, w)` on
// the beams -- transverse AND axial (:131) --, `system('BandGeneral')` + `algorithm('Newton')`
// for a linear problem (:134-138): one linear banded solve.  The matrix is SPD, so the band LDL^T below is the
// same factorisation without pivoting.
//
// All frames of a launch share one topology (node coordinates, connectivity, constraints, equation numbers:
// prepared once on the host, `openpystruct_amd/frames.py`) and differ in the element inertias and loads.
// LDS: ab[n3][ld] (lower band, column j holds A[j..j+kd][j]; n3 = n_eq rounded up to 3, ld = kd + 1 rounded up to
// even so that the diagonal walks of the sweeps hit distinct banks) + rhs[n3]; n3 * (ld + 1) * 8 B <= 160 KB.
// The factorisation is LDS-latency / barrier bound (n_eq * kd^2 / 2 FMAs behind n_eq / 3 barriers), not HBM bound:
// ~15 KB of HBM traffic per frame.
//
// Block steps.  Columns j, j+1, j+2 are eliminated together: every thread factors the 3 x 3 pivot block P
// redundantly (scalar LDL^T of P: the same pivots, in the same order, as the column-by-column algorithm, so the
// result differs from it by rounding only), and applies the rank-3 update  A[R][C] -= B_R P^-1 B_C^T  to the one or
// two window entries (R, C) it owns for the whole factorisation.  The panel B (rows j+3 .. j+2+kd of the three
// columns) is left in place UNSCALED; the factored pivot (1/d1, l21, l31, 1/d2, l32, 1/d3) replaces P one step later.
// Forward substitution rides along in the last wave; backward substitution is done by wave 0 alone, in "axpy" form
// (each solved block is subtracted from the <= kd earlier right-hand sides), with wave-local LDS ordering instead of
// workgroup barriers.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <string_view>

#include "../../include/openpystruct_amd.h"

namespace opsamd { void set_last_error(const char* msg); }   // beam_solve.hip: what ops_amd_last_error() reports
static inline void set_frame_error(const char* msg) { opsamd::set_last_error(msg); }

namespace opsamd {

struct FrameParams {
  int B, Nn, Ne, n_eq, kd;
  const double* elem_geo;    // [Ne,3]  L, cos, sin
  const double* elem_EA;     // [Ne]    E*A
  const double* elem_E;      // [Ne]    E
  const double* elem_w;      // [Ne,2]  wy, wx (local transverse, local axial)
  const int32_t* elem_eq;    // [Ne,6]  equation number per element DOF (-1 = constrained)
  const int32_t* node_eq;    // [Nn,3]
  const double* I;           // [B,Ne]
  const double* loads; long loads_bs;   // [Nn,3] shared (stride 0) or [B,Nn,3]
  double* disp;              // [B,Nn,3]
  double* forces;            // [B,Ne,6]
  double* V; double* M;      // [B,Ne] = forces[:, :, 1], forces[:, :, 2] (what the sizing loss reads, FR:151-153)
  int32_t* status;
};

__device__ __forceinline__ double frcp(double d) {   // v_rcp_f64 + two Newton steps (full precision, normal range)
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}

__device__ __forceinline__ void elem_global_k(double L, double c, double s, double EA, double EI, double k[6][6]) {
  const double rL = frcp(L), a = EA * rL, b2 = 2.0 * EI * rL, b4 = 2.0 * b2, b6 = 3.0 * b2 * rL, b12 = 2.0 * b6 * rL;
  const double kxx = a * c * c + b12 * s * s, kxy = (a - b12) * c * s, kyy = a * s * s + b12 * c * c;
  const double kxt = -b6 * s, kyt = b6 * c;
  const double v[6][6] = {{kxx, kxy, kxt, -kxx, -kxy, kxt},  {kxy, kyy, kyt, -kxy, -kyy, kyt},   {kxt, kyt, b4, -kxt, -kyt, b2},
                          {-kxx, -kxy, -kxt, kxx, kxy, -kxt}, {-kxy, -kyy, -kyt, kxy, kyy, -kyt}, {kxt, kyt, b2, -kxt, -kyt, b4}};
  for (int r = 0; r < 6; ++r)
    for (int q = 0; q < 6; ++q) k[r][q] = v[r][q];
}

constexpr int FRAME_PP = 4;   // window entries per thread, at most (kd <= 63: 2016 pairs <= 2 * 1024)

// kd + 3 offsets per column (rounded to even): offsets kd+1, kd+2 are never written and read as the zeros that lie outside
// the band, so the panel reads of a block step need no guards
__host__ __device__ inline int frame_ld(int kd) { return (kd + 4) & ~1; }
__host__ __device__ inline int frame_n3(int n) { return (n + 2) / 3 * 3; }

struct Pivot3 {   // LDL^T of the 3 x 3 pivot block
  double rd1, l21, l31, rd2, l32, rd3;
  bool bad;
};
__device__ __forceinline__ Pivot3 pivot_factor(double p11, double p21, double p31, double p22, double p32, double p33) {
  Pivot3 f;
  f.rd1 = frcp(p11);
  f.l21 = p21 * f.rd1;
  f.l31 = p31 * f.rd1;
  const double d2 = __builtin_fma(-f.l21, p21, p22);
  f.rd2 = frcp(d2);
  const double u32 = __builtin_fma(-f.l31, p21, p32);
  f.l32 = u32 * f.rd2;
  const double d3 = __builtin_fma(-f.l32, u32, __builtin_fma(-f.l31, p31, p33));
  f.rd3 = frcp(d3);
  f.bad = !(p11 > 0.0) || !(d2 > 0.0) || !(d3 > 0.0);
  return f;
}
// w = P^-1 b
__device__ __forceinline__ void pivot_solve(const Pivot3& f, double b1, double b2, double b3, double& w1, double& w2, double& w3) {
  const double y2 = __builtin_fma(-f.l21, b1, b2);
  const double y3 = __builtin_fma(-f.l32, y2, __builtin_fma(-f.l31, b1, b3));
  w3 = y3 * f.rd3;
  w2 = __builtin_fma(-f.l32, w3, y2 * f.rd2);
  w1 = __builtin_fma(-f.l31, w3, __builtin_fma(-f.l21, w2, b1 * f.rd1));
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Look-ahead of the factorisation, run by ONE wave while the others update the window: the next pivot block
// P' = P_next - B_top P^-1 B_top^T (B_top: the first three window rows), one entry per lane 0..5, gathered with
// readlane, factored, and written over P_next as (1/d1, l21, l31, 1/d2, l32, 1/d3) -- lane i owns slot i.
// c0, c1, c2: the three columns of the current block, n0/n1/n2: the three columns of the next one.
__device__ __forceinline__ bool lookahead_pivot(const Pivot3& f, const double* c0, const double* c1, const double* c2,
                                                double* n0, double* n1, double* n2, int kd, int dl) {
  const int er = dl == 0 ? 0 : (dl == 1 || dl == 3) ? 1 : 2, ec = dl < 3 ? 0 : dl < 5 ? 1 : 2;
  const double a0 = c0[3 + er], a1 = c1[2 + er], a2 = c2[1 + er];
  const double b0 = c0[3 + ec], b1 = c1[2 + ec], b2 = c2[1 + ec];
  double w1, w2, w3;
  pivot_solve(f, b0, b1, b2, w1, w2, w3);
  double* slot = (ec == 0 ? n0 : ec == 1 ? n1 : n2) + (er - ec);
  const double pn = __builtin_fma(-a0, w1, __builtin_fma(-a1, w2, __builtin_fma(-a2, w3, *slot)));
  const Pivot3 fn = pivot_factor(readlane_f64(pn, 0), readlane_f64(pn, 1), readlane_f64(pn, 2), readlane_f64(pn, 3),
                                 readlane_f64(pn, 4), readlane_f64(pn, 5));
  if (dl < 6) *slot = dl == 0 ? fn.rd1 : dl == 1 ? fn.l21 : dl == 2 ? fn.l31 : dl == 3 ? fn.rd2 : dl == 4 ? fn.l32 : fn.rd3;
  return fn.bad;
}
// the first block: nothing to subtract
__device__ __forceinline__ bool first_pivot(double* n0, double* n1, double* n2, int dl) {
  const Pivot3 fn = pivot_factor(n0[0], n0[1], n0[2], n1[0], n1[1], n2[0]);
  __asm__ volatile("" ::: "memory");
  double* slot = (dl < 3 ? n0 : dl < 5 ? n1 : n2) + (dl < 3 ? dl : dl < 5 ? dl - 3 : 0);
  if (dl < 6) *slot = dl == 0 ? fn.rd1 : dl == 1 ? fn.l21 : dl == 2 ? fn.l31 : dl == 3 ? fn.rd2 : dl == 4 ? fn.l32 : fn.rd3;
  return fn.bad;
}
__device__ __forceinline__ Pivot3 load_pivot(const double* c0, const double* c1, const double* c2) {
  Pivot3 f;
  f.rd1 = c0[0]; f.l21 = c0[1]; f.l31 = c0[2]; f.rd2 = c1[0]; f.l32 = c1[1]; f.rd3 = c2[0]; f.bad = false;
  return f;
}

// wave-local LDS ordering (one wave's LDS operations execute in order; this pins the compiler and the counter)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// the window entries a thread owns: (r, c), 0 <= c <= r < kd, enumerated column-major so that a wave's targets and
// panel reads are consecutive LDS words
struct Pairs {
  int r[FRAME_PP], c[FRAME_PP];
  bool on[FRAME_PP];
};
__device__ __forceinline__ Pairs own_pairs(int tid, int T, int kd, int pp_use) {
  Pairs q;
#pragma unroll
  for (int k = 0; k < FRAME_PP; ++k) {
    int rem = tid + k * T, c = 0;
    while (c < kd && rem >= kd - c) { rem -= kd - c; ++c; }
    q.c[k] = c;
    q.r[k] = c + rem;
    q.on[k] = tid >= 0 && (k < pp_use) && c < kd && q.r[k] >= 3;   // r < 3: the next pivot block, the look-ahead wave's
  }
  return q;
}

// element stiffness + consistent loads of element e into a band `ab` / right-hand side `rhs` (LDS or HBM atomics)
__device__ __forceinline__ void assemble_element(const FrameParams& p, const double* Ib, int e, double* ab, double* rhs, int ld) {
  const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
  double k[6][6];
  elem_global_k(L, c, s, p.elem_EA[e], p.elem_E[e] * Ib[e], k);
  const double wy = p.elem_w[2 * e], wx = p.elem_w[2 * e + 1];
  // consistent loads (ElasticBeam2d::addLoad beamUniform), local -> global
  const double pl[6] = {wx * L / 2, wy * L / 2, wy * L * L / 12, wx * L / 2, wy * L / 2, -wy * L * L / 12};
  const double pg[6] = {c * pl[0] - s * pl[1], s * pl[0] + c * pl[1], pl[2], c * pl[3] - s * pl[4], s * pl[3] + c * pl[4], pl[5]};
  int eq[6];
  for (int r = 0; r < 6; ++r) eq[r] = p.elem_eq[6 * e + r];
  for (int r = 0; r < 6; ++r) {
    if (eq[r] < 0) continue;
    atomicAdd(&rhs[eq[r]], pg[r]);
    for (int q = 0; q < 6; ++q) {
      if (eq[q] < 0 || eq[q] > eq[r]) continue;          // lower triangle: row eq[r] >= column eq[q]
      atomicAdd(&ab[(long)eq[q] * ld + (eq[r] - eq[q])], k[r][q]);
    }
  }
}

// nodal displacements and global element end forces (ElasticBeam2d::getResistingForce through LinearCrdTransf2d)
__device__ __forceinline__ void write_results(const FrameParams& p, long b, const double* rhs, bool bad, int tid, int T) {
  const double qnan = __builtin_nan("");
  for (int i = tid; i < p.Nn * 3; i += T) {
    const int q = p.node_eq[i];
    p.disp[b * (long)p.Nn * 3 + i] = bad ? qnan : (q >= 0 ? rhs[q] : 0.0);
  }
  const double* Ib = p.I + b * p.Ne;
  for (int e = tid; e < p.Ne; e += T) {
    const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
    const double EA = p.elem_EA[e], EI = p.elem_E[e] * Ib[e], wy = p.elem_w[2 * e], wx = p.elem_w[2 * e + 1];
    double ug[6];
    for (int r = 0; r < 6; ++r) { const int q = p.elem_eq[6 * e + r]; ug[r] = q >= 0 ? rhs[q] : 0.0; }
    const double ul[6] = {c * ug[0] + s * ug[1], -s * ug[0] + c * ug[1], ug[2], c * ug[3] + s * ug[4], -s * ug[3] + c * ug[4], ug[5]};
    const double rL = frcp(L), chord = (ul[4] - ul[1]) * rL, b2 = 2 * EI * rL, fem = wy * L * L * (1.0 / 12.0);
    const double q0 = EA * rL * (ul[3] - ul[0]) - wx * L / 2;
    const double q1 = 2 * b2 * (ul[2] - chord) + b2 * (ul[5] - chord) - fem;
    const double q2 = b2 * (ul[2] - chord) + 2 * b2 * (ul[5] - chord) + fem;
    const double pl[6] = {-q0 - wx * L, (q1 + q2) * rL - wy * L / 2, q1, q0, -(q1 + q2) * rL - wy * L / 2, q2};
    const double f[6] = {c * pl[0] - s * pl[1], s * pl[0] + c * pl[1], pl[2], c * pl[3] - s * pl[4], s * pl[3] + c * pl[4], pl[5]};
    double* fo = p.forces + (b * (long)p.Ne + e) * 6;
    for (int r = 0; r < 6; ++r) fo[r] = bad ? qnan : f[r];
    p.V[b * (long)p.Ne + e] = bad ? qnan : f[1];
    p.M[b * (long)p.Ne + e] = bad ? qnan : f[2];
  }
  if (tid == 0 && p.status) p.status[b] = bad ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------
// LDS-resident band.  blockDim.x = T: enough threads for one (or, for kd > 43, two) window entries each, plus the
// forward-substitution wave.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void frame_solve_kernel(const FrameParams p, int pp_use) {
  extern __shared__ double lds[];
  const int n = p.n_eq, kd = p.kd, ld = frame_ld(kd), n3 = frame_n3(n);
  double* ab = lds;                     // [n3][ld]
  double* rhs = lds + (size_t)n3 * ld;  // [n3]
  __shared__ int s_bad;
  const int tid = threadIdx.x, T = blockDim.x;
  const long b = blockIdx.x;
  if (tid == 0) s_bad = 0;
  for (int i = tid; i < n3 * ld + n3; i += T) lds[i] = 0.0;
  __syncthreads();
  if (tid < n3 - n) ab[(size_t)(n + tid) * ld] = 1.0;       // padding equations: x = 0
  const double* Ib = p.I + b * p.Ne;
  for (int e = tid; e < p.Ne; e += T) assemble_element(p, Ib, e, ab, rhs, ld);
  const double* lb = p.loads + b * p.loads_bs;
  for (int i = tid; i < p.Nn * 3; i += T) {
    const int q = p.node_eq[i];
    if (q >= 0) atomicAdd(&rhs[q], lb[i]);
  }
  // wave 0 (the oldest, and raised: it is the serial chain of every step): look-ahead pivot; wave 1: forward
  // substitution; waves 2..: the window entries
  const Pairs own = own_pairs(tid - 128, T - 128, kd, pp_use);
  const bool look = tid < 64;
  const int fl = tid - 64;
  if (__builtin_amdgcn_readfirstlane(tid) < 64) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  if (look && first_pivot(ab, ab + ld, ab + 2 * ld, tid) && tid == 0) s_bad = 1;
  __syncthreads();

  // ---- factorisation, three columns per barrier ----
  for (int j = 0; j < n3; j += 3) {
    double* c0 = ab + (size_t)j * ld;
    double* c1 = c0 + ld;
    double* c2 = c1 + ld;
    const Pivot3 f = load_pivot(c0, c1, c2);
    if (look) {
      if (j + 3 < n3 && lookahead_pivot(f, c0, c1, c2, c2 + ld, c2 + 2 * ld, c2 + 3 * ld, kd, tid) && tid == 0) s_bad = 1;
    } else if (fl < 64 && fl < kd && j + 3 + fl < n3) {     // f_X -= B_X P^-1 f_P
      const double a0 = c0[3 + fl], a1 = c1[2 + fl], a2 = c2[1 + fl];
      double w1, w2, w3;
      pivot_solve(f, rhs[j], rhs[j + 1], rhs[j + 2], w1, w2, w3);
      rhs[j + 3 + fl] = __builtin_fma(-a0, w1, __builtin_fma(-a1, w2, __builtin_fma(-a2, w3, rhs[j + 3 + fl])));
    }
#pragma unroll
    for (int k = 0; k < FRAME_PP; ++k) {
      const int r = own.r[k], c = own.c[k];
      if (own.on[k] && j + 3 + r < n3) {
        // panel rows R = j+3+r and C = j+3+c; column j+q holds them at offsets 3+r-q (outside the band: zero)
        const double a0 = c0[3 + r], a1 = c1[2 + r], a2 = c2[1 + r];
        const double b0 = c0[3 + c], b1 = c1[2 + c], b2 = c2[1 + c];
        double w1, w2, w3;
        pivot_solve(f, b0, b1, b2, w1, w2, w3);
        double* t = ab + (size_t)(j + 3 + c) * ld + (r - c);
        *t = __builtin_fma(-a0, w1, __builtin_fma(-a1, w2, __builtin_fma(-a2, w3, *t)));
      }
    }
    __syncthreads();
  }
  if (__builtin_amdgcn_readfirstlane(tid) < 64) __builtin_amdgcn_s_setprio(0);
  // ---- backward substitution: wave 0, no workgroup barriers ----
  if (tid < 64) {
    const int lane = tid;
    for (int k = n3 - 3; k >= 0; k -= 3) {
      const double* c0 = ab + (size_t)k * ld;
      const Pivot3 f = load_pivot(c0, c0 + ld, c0 + 2 * ld);
      const double z1 = rhs[k], z2 = rhs[k + 1], z3 = rhs[k + 2];
      const int e = k - 1 - lane;        // earlier equation; rows k, k+1, k+2 sit at offsets 1+lane, 2+lane, 3+lane
      const bool on = lane < kd && e >= 0;
      const double* ce = ab + (size_t)(on ? e : 0) * ld;
      const double a0 = on ? ce[1 + lane] : 0.0, a1 = on ? ce[2 + lane] : 0.0,
                   a2 = on ? ce[3 + lane] : 0.0, re = rhs[on ? e : 0];
      double x1, x2, x3;
      pivot_solve(f, z1, z2, z3, x1, x2, x3);
      if (on) rhs[e] = __builtin_fma(-a0, x1, __builtin_fma(-a1, x2, __builtin_fma(-a2, x3, re)));
      if (lane == 0) { rhs[k] = x1; rhs[k + 1] = x2; rhs[k + 2] = x3; }
      wave_lds_fence();
    }
  }
  __syncthreads();
  write_results(p, b, rhs, s_bad != 0, tid, T);
}

// ------------------------------------------------------------------------------------------------------
// Frames whose band does not fit LDS (BASELINE config 5: ~500 elements): the assembled band lives in a
// caller-provided HBM workspace (ws[b] = band n3 x ld followed by the right-hand side n3) and the factorisation
// slides a (kd+6)-column ring through LDS: block step j touches columns j .. j+2+kd.  The three finished columns
// (panel unscaled, factored pivot in the place of P) go back to HBM during their own step, the next three come
// into the slots the previous block vacated, so there is still one barrier per block.  Forward substitution rides
// along; backward substitution streams the columns back in chunks (waves 1.. prefetch, wave 0 substitutes in
// dot-product form: a block's own three columns hold everything it needs).
// ------------------------------------------------------------------------------------------------------
// Assembly for the workspace path: one workgroup per frame builds the band slab by slab (FRAME_SLAB columns at a time)
// in LDS with ds_add_f64 and streams each finished slab to HBM with plain coalesced stores -- no global atomics and no
// memset of the workspace (first version: 0.52 ms of a 1.8 ms launch for 1024 frames of 15 x 16).
constexpr int FRAME_SLAB_MAX = 128;

__global__ __launch_bounds__(256) void frame_assemble_kernel(const FrameParams p, double* __restrict__ ws, int slab_cols) {
  extern __shared__ double lds[];
  const long b = blockIdx.x;
  const int ld = frame_ld(p.kd), n3 = frame_n3(p.n_eq), tid = threadIdx.x;
  const int FRAME_SLAB = slab_cols;            // (FRAME_SLAB_MAX for the ring kernel; fewer, wider columns for frame_wide_kernel)
  double* slab = lds;                          // [FRAME_SLAB][ld]
  double* rhs = lds + (size_t)FRAME_SLAB * ld;   // [n3]
  double* ab = ws + b * ((long)n3 * ld + n3);
  const double* Ib = p.I + b * p.Ne;
  for (int i = tid; i < n3; i += 256) rhs[i] = 0.0;
  for (int c0 = 0; c0 < n3; c0 += FRAME_SLAB) {
    const int nc = (n3 - c0 < FRAME_SLAB) ? n3 - c0 : FRAME_SLAB;
    for (int i = tid; i < nc * ld; i += 256) slab[i] = 0.0;
    __syncthreads();
    for (int i = tid; i < nc; i += 256)
      if (c0 + i >= p.n_eq) slab[(size_t)i * ld] = 1.0;                    // padding equations
    for (int e = tid; e < p.Ne; e += 256) {
      int eq[6], lo = 1 << 30, hi = -1;
      for (int r = 0; r < 6; ++r) {
        eq[r] = p.elem_eq[6 * e + r];
        if (eq[r] >= 0) { lo = eq[r] < lo ? eq[r] : lo; hi = eq[r] > hi ? eq[r] : hi; }
      }
      if (hi < c0 || lo >= c0 + nc) continue;                              // no column of this element in the slab
      const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
      double k[6][6];
      elem_global_k(L, c, s, p.elem_EA[e], p.elem_E[e] * Ib[e], k);
      const double wy = p.elem_w[2 * e], wx = p.elem_w[2 * e + 1];
      const double pl[6] = {wx * L / 2, wy * L / 2, wy * L * L / 12, wx * L / 2, wy * L / 2, -wy * L * L / 12};
      const double pg[6] = {c * pl[0] - s * pl[1], s * pl[0] + c * pl[1], pl[2], c * pl[3] - s * pl[4], s * pl[3] + c * pl[4], pl[5]};
      for (int q = 0; q < 6; ++q) {
        if (eq[q] < c0 || eq[q] >= c0 + nc) continue;                      // column eq[q] belongs to this slab
        atomicAdd(&rhs[eq[q]], pg[q]);                                     // each equation's load once: with its own column
        for (int r = 0; r < 6; ++r)
          if (eq[r] >= eq[q]) atomicAdd(&slab[(size_t)(eq[q] - c0) * ld + (eq[r] - eq[q])], k[r][q]);
      }
    }
    __syncthreads();
    for (int i = tid; i < nc * ld; i += 256) ab[(long)c0 * ld + i] = slab[i];
    __syncthreads();
  }
  const double* lb = p.loads + b * p.loads_bs;
  for (int i = tid; i < p.Nn * 3; i += 256) {
    const int q = p.node_eq[i];
    if (q >= 0) atomicAdd(&rhs[q], lb[i]);
  }
  __syncthreads();
  for (int i = tid; i < n3; i += 256) ab[(long)n3 * ld + i] = rhs[i];
}

constexpr int FRAME_CH = 24;   // columns per chunk of the backward sweep (8 blocks)
constexpr int FRAME_PD = 4;    // block steps between issuing a column load and needing it in LDS

// workgroup barrier that waits for this wave's LDS operations only: __syncthreads() would also drain vmcnt, i.e. put
// the HBM latency of the column write-back and of the prefetch on every block step
__device__ __forceinline__ void lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(1024) void frame_factor_big_kernel(const FrameParams p, double* __restrict__ ws, int pp_use) {
  extern __shared__ double lds[];
  const int n = p.n_eq, kd = p.kd, ld = frame_ld(kd), n3 = frame_n3(n), W = kd + 6;
  double* win = lds;                       // [W][ld]   ring of columns, slot = column % W
  double* rhs = win + (size_t)W * ld;      // [n3]
  double* blk = rhs + n3;                  // [2][FRAME_CH][ld] column chunks of the backward sweep
  __shared__ int s_bad;
  const int tid = threadIdx.x, T = blockDim.x;
  const long b = blockIdx.x;
  double* ab = ws + b * ((long)n3 * ld + n3);
  const double* rhs_g = ab + (long)n3 * ld;
  if (tid == 0) s_bad = 0;
  for (int i = tid; i < n3; i += T) rhs[i] = rhs_g[i];
  for (int i = tid; i < (kd + 3) * ld && i < n3 * ld; i += T) win[i] = ab[i];   // columns 0 .. kd+2: slot = column
  const Pairs own = own_pairs(tid - 128, T - 128, kd, pp_use);
  const bool look = tid < 64;
  const int fl = tid - 64;
  if (__builtin_amdgcn_readfirstlane(tid) < 64) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  if (look && first_pivot(win, win + ld, win + 2 * ld, tid) && tid == 0) s_bad = 1;
  __syncthreads();
  // column movers: thread (q, t) carries entry t of the q-th column of a block.  Loads run FRAME_PD block steps ahead
  // through a rotating register queue; the per-step barrier waits for LDS only (lds_barrier), never for HBM.
  const bool mover = tid >= 128 && tid < 128 + 3 * ld;
  const int mq = mover ? (tid - 128) / ld : 0, mt = mover ? (tid - 128) % ld : 0;
  double pre[FRAME_PD];
#pragma unroll
  for (int i = 0; i < FRAME_PD - 1; ++i) {
    const int cin = kd + 3 + 3 * i + mq;
    pre[i] = ab[(long)(cin < n3 ? cin : n3 - 1) * ld + mt];
  }
  pre[FRAME_PD - 1] = 0.0;
  for (int j0 = 0; j0 < n3; j0 += 3 * FRAME_PD) {
#pragma unroll
    for (int u = 0; u < FRAME_PD; ++u) {
      const int j = j0 + 3 * u;
      if (j < n3) {
        double* c0 = win + (size_t)(j % W) * ld;
        double* c1 = win + (size_t)((j + 1) % W) * ld;
        double* c2 = win + (size_t)((j + 2) % W) * ld;
        const Pivot3 f = load_pivot(c0, c1, c2);
        // finished columns j..j+2 (panel unscaled, pivot factored) back to HBM; columns j+kd+3.. (loaded FRAME_PD-1
        // steps ago) into the slots block j-3 left; the load for step j+3*(FRAME_PD-1) goes out
        const double landed = pre[u];
        if (mover) {
          ab[(long)(j + mq) * ld + mt] = win[(size_t)((j + mq) % W) * ld + mt];
          const int cnext = j + kd + 3 + 3 * (FRAME_PD - 1) + mq;
          pre[(u + FRAME_PD - 1) % FRAME_PD] = ab[(long)(cnext < n3 ? cnext : n3 - 1) * ld + mt];
        }
        if (look) {
          if (j + 3 < n3 && lookahead_pivot(f, c0, c1, c2, win + (size_t)((j + 3) % W) * ld, win + (size_t)((j + 4) % W) * ld,
                                            win + (size_t)((j + 5) % W) * ld, kd, tid) && tid == 0) s_bad = 1;
        } else if (fl < 64 && fl < kd && j + 3 + fl < n3) {
          const double a0 = c0[3 + fl], a1 = c1[2 + fl], a2 = c2[1 + fl];
          double w1, w2, w3;
          pivot_solve(f, rhs[j], rhs[j + 1], rhs[j + 2], w1, w2, w3);
          rhs[j + 3 + fl] = __builtin_fma(-a0, w1, __builtin_fma(-a1, w2, __builtin_fma(-a2, w3, rhs[j + 3 + fl])));
        }
#pragma unroll
        for (int k = 0; k < FRAME_PP; ++k) {
          const int r = own.r[k], c = own.c[k];
          if (own.on[k] && j + 3 + r < n3) {
            const double a0 = c0[3 + r], a1 = c1[2 + r], a2 = c2[1 + r];
            const double b0 = c0[3 + c], b1 = c1[2 + c], b2 = c2[1 + c];
            double w1, w2, w3;
            pivot_solve(f, b0, b1, b2, w1, w2, w3);
            double* t = win + (size_t)((j + 3 + c) % W) * ld + (r - c);
            *t = __builtin_fma(-a0, w1, __builtin_fma(-a1, w2, __builtin_fma(-a2, w3, *t)));
          }
        }
        const int cin = j + kd + 3 + mq;
        if (mover && cin < n3) win[(size_t)(cin % W) * ld + mt] = landed;
        lds_barrier();
      }
    }
  }
  __syncthreads();                       // the written-back columns are read again below
  if (__builtin_amdgcn_readfirstlane(tid) < 64) __builtin_amdgcn_s_setprio(0);
  // backward substitution, chunks of FRAME_CH columns, last chunk first
  const int nch = (n3 + FRAME_CH - 1) / FRAME_CH;
  for (int i = tid; i < FRAME_CH * ld; i += T) {
    const int col = (nch - 1) * FRAME_CH + i / ld;
    blk[i] = col < n3 ? ab[(long)col * ld + i % ld] : 0.0;
  }
  __syncthreads();
  for (int kb = nch - 1; kb >= 0; --kb) {
    const double* cur = blk + (size_t)((nch - 1 - kb) & 1) * FRAME_CH * ld;
    double* nxt = blk + (size_t)((nch - kb) & 1) * FRAME_CH * ld;
    if (tid >= 64 && kb > 0) {
      for (int i = tid - 64; i < FRAME_CH * ld; i += T - 64) nxt[i] = ab[(long)((kb - 1) * FRAME_CH + i / ld) * ld + i % ld];
    }
    if (tid < 64) {
      const int lane = tid;
      for (int jj = FRAME_CH - 3; jj >= 0; jj -= 3) {
        const int k = kb * FRAME_CH + jj;
        if (k >= n3) continue;
        const double* c0 = cur + (size_t)jj * ld;
        const double* c1 = c0 + ld;
        const double* c2 = c1 + ld;
        // z_q = y_q - sum_X A[X][k+q] x_X over the window rows X = k+3+lane
        const int X = k + 3 + lane;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (lane < kd && X < n3) {
          const double xX = rhs[X];
          s0 = c0[3 + lane] * xX;
          s1 = c1[2 + lane] * xX;
          s2 = c2[1 + lane] * xX;
        }
        for (int sft = 32; sft >= 1; sft >>= 1) {
          s0 += __shfl_xor(s0, sft, 64);
          s1 += __shfl_xor(s1, sft, 64);
          s2 += __shfl_xor(s2, sft, 64);
        }
        const Pivot3 f = load_pivot(c0, c1, c2);
        double x1, x2, x3;
        pivot_solve(f, rhs[k] - s0, rhs[k + 1] - s1, rhs[k + 2] - s2, x1, x2, x3);
        wave_lds_fence();
        if (lane == 0) { rhs[k] = x1; rhs[k + 1] = x2; rhs[k + 2] = x3; }
        wave_lds_fence();
      }
    }
    __syncthreads();
  }
  write_results(p, b, rhs, s_bad != 0, tid, T);
}

// ------------------------------------------------------------------------------------------------------
// Half bandwidths beyond 63 (r05; more than 20 bays AND more than 20 stories: outside the reference's random range, FR:17-18, and
// outside BASELINE's ~500 elements): the plain column-by-column band LDL^T -- dpbsv's order, no blocking -- on the band in the HBM
// workspace (the layout of frame_assemble_kernel), one workgroup per frame, two barriers per column; the trailing window is updated in
// place through L2.  A FALLBACK so that the C ABI answers for any band (milliseconds per frame), not a tuned path: the window rows of the
// block kernels are one 64-lane wave and the register window of the wave kernel ends at 55.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void frame_wide_kernel(const FrameParams p, double* __restrict__ ws) {
  extern __shared__ double lds[];
  const int n = p.n_eq, kd = p.kd, ld = frame_ld(kd), n3 = frame_n3(n);
  double* rhs = lds;                 // [n3]  right-hand side -> z -> x
  double* col = rhs + n3;            // [ld]  column j, unscaled
  __shared__ int s_bad;
  const int tid = threadIdx.x, T = blockDim.x;
  const long b = blockIdx.x;
  double* ab = ws + b * ((long)n3 * ld + n3);
  if (tid == 0) s_bad = 0;
  for (int i = tid; i < n3; i += T) rhs[i] = ab[(long)n3 * ld + i];
  __syncthreads();
  for (int j = 0; j < n3; ++j) {
    double* cj = ab + (long)j * ld;
    for (int r = tid; r <= kd; r += T) col[r] = cj[r];
    __syncthreads();
    const double d = col[0], rd = frcp(d), zj = rhs[j];
    if (tid == 0 && !(d > 0.0)) s_bad = 1;
    // column j of L (the diagonal keeps d_j) and the forward substitution
    for (int r = tid + 1; r <= kd; r += T) {
      if (j + r < n3) {
        const double l = col[r] * rd;
        cj[r] = l;
        rhs[j + r] = __builtin_fma(-l, zj, rhs[j + r]);
      }
    }
    // trailing window: A[j + r][j + c] -= (A[j + r][j] / d_j) A[j + c][j], 1 <= c <= r <= kd (consecutive threads: consecutive r)
    for (int idx = tid; idx < kd * kd; idx += T) {
      const int c = idx / kd + 1, r = idx - (c - 1) * kd + 1;
      if (r >= c && j + r < n3) {
        double* t = ab + (long)(j + c) * ld + (r - c);
        *t = __builtin_fma(-(col[r] * rd), col[c], *t);
      }
    }
    __syncthreads();
  }
  // w = D^-1 z, then L^T x = w from the last equation up: wave 0, the next column's loads under way while a column is summed
  if (tid < 64) {
    const int lane = tid;
    constexpr int MAXR = 16;         // kd <= 64 * MAXR
    double la[MAXR], lb[MAXR], da = 1.0, db = 1.0;
    auto issue = [&](int j, double (&l)[MAXR], double& dj) {
      const double* cj = ab + (long)(j > 0 ? j : 0) * ld;
      dj = cj[0];
#pragma unroll
      for (int m = 0; m < MAXR; ++m) { const int r = lane + 1 + 64 * m; l[m] = (m * 64 < kd && r <= kd) ? cj[r] : 0.0; }
    };
    auto solve = [&](int j, const double (&l)[MAXR], double dj) {
      double s_ = 0.0;
#pragma unroll
      for (int m = 0; m < MAXR; ++m) { const int r = lane + 1 + 64 * m; if (m * 64 < kd && r <= kd && j + r < n3) s_ = __builtin_fma(l[m], rhs[j + r], s_); }
      for (int sft = 32; sft >= 1; sft >>= 1) s_ += __shfl_xor(s_, sft, 64);
      wave_lds_fence();
      if (lane == 0) rhs[j] = rhs[j] * frcp(dj) - s_;
      wave_lds_fence();
    };
    issue(n3 - 1, la, da);
    for (int j = n3 - 1; j >= 0; j -= 2) {
      issue(j - 1, lb, db);
      solve(j, la, da);
      if (j - 1 >= 0) { issue(j - 2, la, da); solve(j - 1, lb, db); }
    }
  }
  __syncthreads();
  write_results(p, b, rhs, s_bad != 0, tid, T);
}
constexpr int FRAME_WIDE_MAX_KD = 1024;      // frame_wide_kernel's backward sweep keeps a column in 16 registers per lane

}  // namespace opsamd

#include "frame_wave.hpp"
#include "frame_pack.hpp"
#include "frame_coop.hpp"

using namespace opsamd;

static const size_t LDS_MAX = 160 * 1024 - 64;

// ---- library options (ops_amd_set_option: the one place a caller -- tests, A/B scripts -- steers the dispatch; no environment variable is read) ----
static std::atomic<long> g_frame_latency_batch{-1};      // "frame_latency_batch": -1 = the model below; 0 = tuned kernels for every batch
static std::atomic<long> g_frame_coop{1};                // "frame_coop": 0 = never four waves per frame; 1 = where measured faster (default); 2 = for every small batch (A/B, tests)
static std::atomic<long> g_frame_pack{1};                // "frame_pack": 0 = one wave per frame for every half bandwidth (A/B)
static std::atomic<int> g_deterministic{0};              // "deterministic": 1 = fixed-order reductions in the Transformer-Diffusion step's gradient launches
namespace opsamd {
int deterministic_mode() { return g_deterministic.load(std::memory_order_relaxed); }
void reset_head_ticket();              // seq_layer.hip
}

extern "C" int ops_amd_set_option(const char* name, long value) {
  if (!name) return OPS_AMD_ERR_INVALID_ARG;
  const std::string_view n(name);
  if (n == "frame_latency_batch") { g_frame_latency_batch.store(value < 0 ? -1 : value); return OPS_AMD_OK; }
  if (n == "frame_pack") { g_frame_pack.store(value != 0); return OPS_AMD_OK; }
  if (n == "frame_coop") { if (value < 0 || value > 2) return OPS_AMD_ERR_INVALID_ARG; g_frame_coop.store(value); return OPS_AMD_OK; }
  if (n == "deterministic") {
    g_deterministic.store(value != 0);
    int ndev = 0;
    if (value != 0 && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) opsamd::reset_head_ticket();      // (current device; no GPU: nothing to re-arm)
    return OPS_AMD_OK;
  }
  return OPS_AMD_ERR_INVALID_ARG;
}
extern "C" long ops_amd_get_option(const char* name) {
  if (!name) return -2;
  const std::string_view n(name);
  if (n == "frame_latency_batch") return g_frame_latency_batch.load();
  if (n == "frame_pack") return g_frame_pack.load();
  if (n == "frame_coop") return g_frame_coop.load();
  if (n == "deterministic") return g_deterministic.load();
  return -2;
}

static size_t frame_lds_resident_bytes(int n_eq, int kd) {
  return ((size_t)frame_n3(n_eq) * frame_ld(kd) + frame_n3(n_eq)) * sizeof(double);
}

// workgroup size and window entries per thread: two service waves (look-ahead, forward substitution) + the entry owners
static void frame_threads(int kd, bool resident, int* T, int* pp_use) {
  const int npairs = kd * (kd + 1) / 2;
  auto threads = [&](int pp) { return ((npairs + pp - 1) / pp + 63) / 64 * 64 + 128; };
  // entries per thread (fewer waves per barrier, more workgroups per CU), measured (profiles/r01_notes.md): four where
  // several workgroups share a CU -- narrow bands (5x5: 3.5e7 vs 3.15e7/s) and the workspace path (15x16: 9.4e5 vs
  // 8.55e5/s) --, two for the one-workgroup-per-CU frames in between (10x10: 2.44e6 vs 2.40e6/s)
  int pp = (!resident || kd <= 24) ? 4 : 2;
  while (threads(pp) > 1024 && pp < FRAME_PP) ++pp;
  *pp_use = pp;
  *T = threads(pp) < 192 ? 192 : threads(pp);
}

static int eff_kd(int half_bandwidth) { return half_bandwidth < 3 ? 3 : half_bandwidth; }   // a block step's band covers its own pivot

// Which kernel family serves a call.  Batches of at most about one frame per CU are LATENCY-bound, and there a workgroup per frame (r01 kernels:
// the whole workgroup works on the one frame its CU has) answers sooner than a wave per frame (one wave works, the CU's other 15 wave slots idle):
// 10 x 10 118 us against 207 - 221 us for 1 .. 256 frames, 5 x 5 47 / 84, 3 x 3 36 / 55, 15 x 16 (band in HBM) 472 - 492 / 535 - 576; at 1 024 frames
// the wave kernel is ahead (10 x 10: 231 / 426 us) (scripts/frame_small_batch_ab.py).  That is the reference's own use of the frame solve -- ONE
// frame per epoch (FR:178-183) -- and the command shim's.  The batch at which the two families meet is the threshold (latency_batch below), at
// least 256 (one frame per CU) and at most 4 000.  Option "frame_latency_batch" overrides it (0: tuned kernels for every batch).
// what one launch of the tuned kernels never takes less than (one wave's chain): fit of r06, see latency_batch
static double tuned_floor_seconds(int n_eq, int kd) {
  int P, G, W;
  const bool pack = g_frame_pack.load() && fp_config(kd, &P, &G, &W);
  return pack ? 10e-6 + 0.5e-6 * n_eq : (0.275e-6 + 0.0075e-6 * kd) * n_eq;
}
static int latency_batch(int n_eq, int kd) {
  const long o = g_frame_latency_batch.load();
  if (o >= 0) return (int)(o > 0x7fffffff ? 0x7fffffff : o);
  // r06 fit (scripts/frame_dispatch_sweep.py on the packed kernel, profiles/r06_frame_dispatch_sweep.txt): a launch of the tuned kernels never
  // takes less than one wave's chain -- packed (kd <= 29): ~10 us + 0.5 us per equation (2 x 2: 17 us, 5 x 5: 47, 8 x 8: 123); a wave per frame:
  // (0.275 + 0.0075 kd) us per equation (10 x 10: 170 us, 15 x 16: 500) -- and the workgroup-per-frame kernels, flat up to a frame per CU, then
  // cost max(n kd^2 / 8.5e11, 6 ns + 0.22 ns n) per frame (5 x 5: 30 ns, 3 x 3: 14 ns, 10 x 10: 0.40 us).  The two meet at the quotient: 2 x 2 ~1 700
  // frames, 5 x 5 ~1 500, 10 x 10 ~430 (measured crossovers: between 1 024 and 2 048, 1 024 and 2 048, 256 and 512).
  const double tuned_floor_s = tuned_floor_seconds(n_eq, kd);
  const double a = (double)n_eq * kd * kd / 8.5e11, c = 6e-9 + 0.22e-9 * n_eq;
  const double b = tuned_floor_s / (a > c ? a : c);
  return b < 256.0 ? 256 : b > 4000.0 ? 4000 : (int)b;
}
static bool legacy_kernels_serve(int n_eq, int kd) {
  if (kd > 63) return false;
  if (frame_lds_resident_bytes(n_eq, kd) <= LDS_MAX) return true;
  const size_t ring = ((size_t)(kd + 6) * frame_ld(kd) + (size_t)frame_n3(n_eq) + 2 * (size_t)FRAME_CH * frame_ld(kd)) * sizeof(double);
  return ring <= LDS_MAX;
}
// (ops_frame_workspace_bytes does not know the element count: both it and the solve decide with the bound the plan's workspace share is sized by)
static int ne_bound(int n_eq) { return 4 * n_eq + 64; }
// the assembly plan sits at the START of the workspace (r06: where it is does not depend on the batch, so a caller that keeps the workspace may
// keep the plan: OPS_FRAME_REUSE_PLAN), the per-wave factor storage behind it
static size_t plan_region_bytes(int n_eq, int G, int EPG) { return (fw_plan_bytes(n_eq, ne_bound(n_eq), G, EPG) + 255) & ~(size_t)255; }

enum FrameFamily { FAM_WIDE, FAM_LEGACY, FAM_WAVE, FAM_PACK, FAM_COOP };
// frame slots of the packed kernel's factor storage: whole waves (a lane group past the end of the batch solves the last frame again, into its own slot)
static size_t pack_slots(int B, int P) { const size_t F = 64 / P; return ((size_t)B + F - 1) / F * F; }
// kd <= 29 (98 of the 100 (bays, stories) draws of FR:17-18): frame_pack.hpp, 16 or 32 lanes per frame; 30..55: frame_wave.hpp, a wave per
// frame; small batches and 56..63: the workgroup-per-frame kernels; beyond: the column-by-column fallback
static FrameFamily frame_family(int B, int n_eq, int kd) {
  if (kd > 63) return FAM_WIDE;
  if (kd > 55) return FAM_LEGACY;
  // Small batches (scripts/frame_coop_sweep.py, profiles/r06_frame_coop_sweep*.txt).  The r01 workgroup-per-frame kernels win while their band is
  // LDS-resident and the batch is at most their share of the chip (10 x 10, 256 frames: 107 us against 123 for four waves per frame and 186 for a
  // wave per frame).  Four waves per frame (frame_coop.hpp: 8 KB of LDS and 256 threads, three to a CU) win (i) where that band is NOT resident
  // (15 x 16, up to 512 frames: 303-358 us against 471-646 and 503; two workgroups of the 52-wide kernel to a CU) and (ii) between one and three frames per CU where the r01 kernel fits only
  // once per CU and needs a second round (10 x 10, 384-768 frames: 140-162 us against 204-306 and 189; 12 x 12, 512: 226 / 290 / 312); its cost
  // ~0.37 us per equation, +15 % per further frame per CU, is held against the tuned kernels' floor.  An explicit "frame_latency_batch" option keeps
  // its meaning: that many frames or fewer never take the tuned kernels.
  const bool lat_forced = g_frame_latency_batch.load() >= 0;
  const int lat = latency_batch(n_eq, kd);
  const bool coop_ok = g_frame_coop.load() && fc_lds_doubles(n_eq, fc_width(kd)) * sizeof(double) <= LDS_MAX;
  const bool legacy_ok = legacy_kernels_serve(n_eq, kd);
  const size_t band = frame_lds_resident_bytes(n_eq, kd);
  if (coop_ok && !lat_forced) {
    const double coop_s = 0.37e-6 * n_eq * (1.0 + 0.15 * ((B - 1) / 256));
    const int coop_max = fc_width(kd) >= 52 ? 512 : 768;      // (the 52- / 56-wide kernels: two workgroups to a CU -- 15 x 16 x 768: 630 us against 518)
    if (band > LDS_MAX && B <= coop_max) return FAM_COOP;
    if (band > LDS_MAX / 2 && B > 256 && B <= coop_max && coop_s < tuned_floor_seconds(n_eq, kd)) return FAM_COOP;
  }
  if (B <= lat) {
    if (coop_ok && g_frame_coop.load() == 2) return FAM_COOP;
    if (legacy_ok && (band <= LDS_MAX || !coop_ok)) return FAM_LEGACY;
    if (coop_ok) return FAM_COOP;
  }
  int P, G, W;
  // the packed kernel's 8 or 16 frames per workgroup keep x (n_eq) and their parking areas in LDS (+ the inertias if those fit too: launch_pack)
  if (g_frame_pack.load() && fp_config(kd, &P, &G, &W) && 4 * (size_t)(64 / P) * fp_lds_doubles(n_eq, 0, P, G, W) * sizeof(double) <= LDS_MAX)
    return FAM_PACK;
  // the wave kernel (window width 36 serves every narrower band): its four waves' LDS must fit one CU; frames beyond that (tall and narrow:
  // thousands of equations) take the workgroup-per-frame kernels, whose band streams through an LDS ring
  if (4 * fw_lds_doubles(n_eq, fw_width(kd)) * sizeof(double) <= LDS_MAX) return FAM_WAVE;
  return FAM_LEGACY;
}

// as many workgroups as the chip holds at once (persistent waves): asked of the runtime once per device, kernel and LDS size
static hipError_t resident_workgroups(const void* fn, size_t lds, int devid, std::atomic<long long>* key_slot, std::atomic<int>* val_slot, int* cap) {
  const long long key = ((long long)lds << 1) | 1;
  if (key_slot[devid & 63].load(std::memory_order_acquire) == key) {
    *cap = val_slot[devid & 63].load(std::memory_order_relaxed);
    if (*cap > 0) return hipSuccess;
  }
  int per_cu = 0, cus = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid);
  if (e != hipSuccess) return e;
  *cap = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
  key_slot[devid & 63].store(0, std::memory_order_release);
  val_slot[devid & 63].store(*cap, std::memory_order_relaxed);
  key_slot[devid & 63].store(key, std::memory_order_release);
  return hipSuccess;
}

template <int W, int P, int G>
static hipError_t launch_pack(const FrameParams& p, double* ws, hipStream_t s, bool reuse_plan) {
  static std::atomic<unsigned long long> done{0};
  static std::atomic<long long> cap_key[64];
  static std::atomic<int> cap_val[64];
  int devid = 0;
  hipError_t e = hipGetDevice(&devid);
  if (e != hipSuccess) return e;
  constexpr int F = 64 / P;
  const bool stage_I = 4 * (size_t)F * fp_lds_doubles(p.n_eq, p.Ne, P, G, W) * sizeof(double) <= LDS_MAX / 2;     // (at least two workgroups per CU)
  const int ne_lds = stage_I ? p.Ne : 0;
  const size_t lds = 4 * (size_t)F * fp_lds_doubles(p.n_eq, ne_lds, P, G, W) * sizeof(double);
  const unsigned long long bit = 1ull << (devid & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute((const void*)frame_pack_kernel<W, P, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
  }
  int cap = 0;
  e = resident_workgroups((const void*)frame_pack_kernel<W, P, G>, lds, devid, cap_key, cap_val, &cap);
  if (e != hipSuccess) return e;
  const long need = ((long)p.B + 4 * F - 1) / (4 * F);
  const unsigned grid = (unsigned)(need < cap ? need : cap);
  void* plan_base = ws;
  double* factor = (double*)((char*)ws + plan_region_bytes(p.n_eq, G, fp_epg(G)));
  const FwPlan pl = fw_plan_at(plan_base, p.n_eq, p.Ne, G, fp_epg(G));
  if (!reuse_plan)
    hipLaunchKernelGGL(frame_plan_kernel, dim3(1), dim3(1024), (size_t)3 * (fw_groups(p.n_eq, G) + 2) * sizeof(int), s, p, W, plan_base, G, fp_epg(G));
  hipLaunchKernelGGL((frame_pack_kernel<W, P, G>), dim3(grid), dim3(256), lds, s, p, factor, pl, ne_lds);
  return hipGetLastError();
}

template <int W>
static hipError_t launch_wave(const FrameParams& p, double* ws_all, hipStream_t s, bool reuse_plan) {
  static std::atomic<unsigned long long> done{0};
  int devid = 0;
  hipError_t e = hipGetDevice(&devid);
  if (e != hipSuccess) return e;
  const size_t lds = 4 * fw_lds_doubles(p.n_eq, W) * sizeof(double);      // size limits: frame_family (checked by the caller)
  const unsigned long long bit = 1ull << (devid & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute((const void*)frame_wave_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
  }
  void* plan_base = ws_all;
  double* ws = (double*)((char*)ws_all + plan_region_bytes(p.n_eq, FW_G, FW_EPG));
  const FwPlan pl = fw_plan_at(plan_base, p.n_eq, p.Ne);
  if (!reuse_plan)
    hipLaunchKernelGGL(frame_plan_kernel, dim3(1), dim3(1024), (size_t)3 * (fw_groups(p.n_eq) + 2) * sizeof(int), s, p, W, plan_base, FW_G, FW_EPG);
  hipLaunchKernelGGL((frame_wave_kernel<W>), dim3((unsigned)((p.B + 3) / 4)), dim3(256), lds, s, p, ws, pl);
  return hipGetLastError();
}

template <int W>
static hipError_t launch_coop(const FrameParams& p, double* ws_all, hipStream_t s, bool reuse_plan) {
  static std::atomic<unsigned long long> done{0};
  int devid = 0;
  hipError_t e = hipGetDevice(&devid);
  if (e != hipSuccess) return e;
  const size_t lds = fc_lds_doubles(p.n_eq, W) * sizeof(double);          // size limit: frame_family (checked by the caller)
  const unsigned long long bit = 1ull << (devid & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute((const void*)frame_coop_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
  }
  void* plan_base = ws_all;
  double* ws = (double*)((char*)ws_all + plan_region_bytes(p.n_eq, FW_G, FW_EPG));
  const FwPlan pl = fw_plan_at(plan_base, p.n_eq, p.Ne);
  if (!reuse_plan)
    hipLaunchKernelGGL(frame_plan_kernel, dim3(1), dim3(1024), (size_t)3 * (fw_groups(p.n_eq) + 2) * sizeof(int), s, p, W, plan_base, FW_G, FW_EPG);
  hipLaunchKernelGGL((frame_coop_kernel<W>), dim3((unsigned)p.B), dim3(64 * FC_NW), lds, s, p, ws, pl);
  return hipGetLastError();
}

extern "C" size_t ops_frame_workspace_bytes(int B, int n_eq, int half_bandwidth) {
  if (B <= 0 || n_eq < 1 || half_bandwidth < 0) return 0;
  const int kd = eff_kd(half_bandwidth);
  switch (frame_family(B, n_eq, kd)) {
    case FAM_WIDE: return (size_t)B * frame_lds_resident_bytes(n_eq, kd);      // frame_wide_kernel: the band always lives in HBM
    case FAM_PACK: {                               // per frame: the columns of L (the kernel uses one slot per resident wave); once: the plan
      int P, G, W;
      fp_config(kd, &P, &G, &W);
      return pack_slots(B, P) * fp_frame_doubles(n_eq, W) * sizeof(double) + plan_region_bytes(n_eq, G, fp_epg(G));
    }
    case FAM_WAVE: return (size_t)B * fw_frame_doubles(n_eq, kd) * sizeof(double) + plan_region_bytes(n_eq, FW_G, FW_EPG);
    case FAM_COOP: return (size_t)B * fc_frame_doubles(n_eq, kd) * sizeof(double) + plan_region_bytes(n_eq, FW_G, FW_EPG);
    default: break;
  }
  if (frame_lds_resident_bytes(n_eq, kd) <= LDS_MAX) return 0;   // the band lives in LDS
  return (size_t)B * frame_lds_resident_bytes(n_eq, kd);
}

// which plan (if any) a call of this shape builds at the start of its workspace: 0 = none; equal values = interchangeable plans
extern "C" long ops_frame_plan_signature(int B, int n_eq, int half_bandwidth) {
  if (B <= 0 || n_eq < 1 || half_bandwidth < 0) return 0;
  const int kd = eff_kd(half_bandwidth);
  int P, G, W;
  switch (frame_family(B, n_eq, kd)) {
    case FAM_PACK: fp_config(kd, &P, &G, &W); return (2L << 24) | (W << 16) | (G << 8) | P;
    case FAM_WAVE: return (1L << 24) | (fw_width(kd) << 16) | (FW_G << 8) | 64;
    case FAM_COOP: return (3L << 24) | (fc_width(kd) << 16) | (FW_G << 8) | 64;
    default: return 0;
  }
}

extern "C" int ops_frame_solve_batched_f64(int B, int n_nodes, int n_elems, int n_eq, int half_bandwidth,
                                           const double* elem_geo, const double* elem_EA, const double* elem_E,
                                           const double* elem_w, const int32_t* elem_eq, const int32_t* node_eq,
                                           const double* I, const double* loads, long loads_bstride, double* disp,
                                           double* forces, double* V, double* M, int32_t* status, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  return ops_frame_solve_batched_f64_ex(B, n_nodes, n_elems, n_eq, half_bandwidth, elem_geo, elem_EA, elem_E, elem_w, elem_eq, node_eq, I, loads,
                                        loads_bstride, disp, forces, V, M, status, workspace, workspace_bytes, stream, 0u);
}

extern "C" int ops_frame_solve_batched_f64_ex(int B, int n_nodes, int n_elems, int n_eq, int half_bandwidth,
                                              const double* elem_geo, const double* elem_EA, const double* elem_E,
                                              const double* elem_w, const int32_t* elem_eq, const int32_t* node_eq,
                                              const double* I, const double* loads, long loads_bstride, double* disp,
                                              double* forces, double* V, double* M, int32_t* status, void* workspace,
                                              size_t workspace_bytes, void* stream, unsigned flags) {
  const bool reuse_plan = (flags & OPS_FRAME_REUSE_PLAN) != 0;
  if (B < 0 || n_nodes < 2 || n_elems < 1 || n_eq < 1 || half_bandwidth < 0) return OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!elem_geo || !elem_EA || !elem_E || !elem_w || !elem_eq || !node_eq || !I || !loads || !disp || !forces || !V || !M)
    return OPS_AMD_ERR_INVALID_ARG;
  if (half_bandwidth > FRAME_WIDE_MAX_KD) return OPS_AMD_ERR_UNSUPPORTED;
  const int kd = eff_kd(half_bandwidth);
  const size_t lds_bytes = frame_lds_resident_bytes(n_eq, kd);
  {
    // the dynamic-LDS limit is a per-DEVICE function attribute: set once per device this thread-safe way (a process may
    // drive several GPUs: ops.set_device / FrameTopology(device=...))
    static std::atomic<unsigned long long> attr_done{0};
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return OPS_AMD_ERR_LAUNCH;
    const unsigned long long bit = 1ull << devid;
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute((const void*)frame_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)frame_assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)frame_factor_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)frame_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
      if (e != hipSuccess) { set_frame_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
      attr_done.fetch_or(bit, std::memory_order_release);      // idempotent: two threads racing here both set the same value
    }
  }
  FrameParams p{B, n_nodes, n_elems, n_eq, kd, elem_geo, elem_EA, elem_E, elem_w, elem_eq, node_eq,
                I, loads, loads_bstride, disp, forces, V, M, status};
  hipStream_t s = (hipStream_t)stream;
  const FrameFamily fam = frame_family(B, n_eq, kd);
  if (fam == FAM_WIDE) {
    // the window rows of a block step are one 64-lane wave: beyond that, the plain column-by-column fallback on the band in HBM
    const int ld = frame_ld(kd), n3 = frame_n3(n_eq);
    const size_t need = (size_t)B * lds_bytes, lds_wide = ((size_t)n3 + ld) * sizeof(double);
    if (lds_wide > LDS_MAX) return OPS_AMD_ERR_UNSUPPORTED;
    long slab = ((long)(LDS_MAX / sizeof(double)) - n3) / ld;          // columns per LDS slab of the assembly
    if (slab > FRAME_SLAB_MAX) slab = FRAME_SLAB_MAX;
    if (slab < 4) return OPS_AMD_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < need) return OPS_AMD_ERR_INVALID_ARG;
    hipLaunchKernelGGL(frame_assemble_kernel, dim3((unsigned)B), dim3(256), ((size_t)slab * ld + (size_t)n3) * sizeof(double), s, p, (double*)workspace, (int)slab);
    hipLaunchKernelGGL(frame_wide_kernel, dim3((unsigned)B), dim3(1024), lds_wide, s, p, (double*)workspace);
    return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
  }
  if (fam == FAM_PACK) {
    int P, G, W;
    fp_config(kd, &P, &G, &W);
    const size_t need = pack_slots(B, P) * fp_frame_doubles(n_eq, W) * sizeof(double) + plan_region_bytes(n_eq, G, fp_epg(G));
    if (!workspace || workspace_bytes < need || n_elems > ne_bound(n_eq)) return OPS_AMD_ERR_INVALID_ARG;
    hipError_t e = hipSuccess;
    switch (W) {
      case 6: e = launch_pack<6, 16, 4>(p, (double*)workspace, s, reuse_plan); break;
      case 10: e = launch_pack<10, 16, 4>(p, (double*)workspace, s, reuse_plan); break;
      case 12: e = launch_pack<12, 16, 4>(p, (double*)workspace, s, reuse_plan); break;
      case 16: e = launch_pack<16, 32, 8>(p, (double*)workspace, s, reuse_plan); break;
      case 18: e = launch_pack<18, 32, 8>(p, (double*)workspace, s, reuse_plan); break;
      case 22: e = launch_pack<22, 32, 8>(p, (double*)workspace, s, reuse_plan); break;
      case 24: e = launch_pack<24, 32, 8>(p, (double*)workspace, s, reuse_plan); break;
      case 28: e = launch_pack<28, 32, 4>(p, (double*)workspace, s, reuse_plan); break;
      default: e = launch_pack<30, 32, 2>(p, (double*)workspace, s, reuse_plan); break;
    }
    if (e != hipSuccess) { set_frame_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
    return OPS_AMD_OK;
  }
  if (fam == FAM_COOP) {
    const int W = fc_width(kd);
    const size_t need = (size_t)B * fc_frame_doubles(n_eq, kd) * sizeof(double) + plan_region_bytes(n_eq, FW_G, FW_EPG);
    if (!workspace || workspace_bytes < need || n_elems > ne_bound(n_eq)) return OPS_AMD_ERR_INVALID_ARG;
    hipError_t e = hipSuccess;
    switch (W) {
      case 20: e = launch_coop<20>(p, (double*)workspace, s, reuse_plan); break;
      case 36: e = launch_coop<36>(p, (double*)workspace, s, reuse_plan); break;
      case 52: e = launch_coop<52>(p, (double*)workspace, s, reuse_plan); break;
      default: e = launch_coop<56>(p, (double*)workspace, s, reuse_plan); break;
    }
    if (e != hipSuccess) { set_frame_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
    return OPS_AMD_OK;
  }
  if (fam == FAM_WAVE) {
    const int W = fw_width(kd);
    const size_t need = (size_t)B * fw_frame_doubles(n_eq, kd) * sizeof(double) + plan_region_bytes(n_eq, FW_G, FW_EPG);
    if (!workspace || workspace_bytes < need || n_elems > ne_bound(n_eq)) return OPS_AMD_ERR_INVALID_ARG;
    hipError_t e = hipSuccess;
    switch (W) {
      case 36: e = launch_wave<36>(p, (double*)workspace, s, reuse_plan); break;
      case 52: e = launch_wave<52>(p, (double*)workspace, s, reuse_plan); break;
      default: e = launch_wave<56>(p, (double*)workspace, s, reuse_plan); break;
    }
    if (e != hipSuccess) { set_frame_error(hipGetErrorString(e)); return OPS_AMD_ERR_LAUNCH; }
    return OPS_AMD_OK;
  }
  const bool resident = lds_bytes <= LDS_MAX;
  int T, pp_use;
  frame_threads(kd, resident, &T, &pp_use);
  if (!resident) {
    // band in the HBM workspace, sliding LDS ring
    const int ld = frame_ld(kd), n3 = frame_n3(n_eq);
    const size_t need = (size_t)B * lds_bytes;
    const size_t lds2 = ((size_t)(kd + 6) * ld + (size_t)n3 + 2 * (size_t)FRAME_CH * ld) * sizeof(double);
    if (lds2 > LDS_MAX) return OPS_AMD_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < need) return OPS_AMD_ERR_INVALID_ARG;
    if (T < 128 + 3 * ld) T = (128 + 3 * ld + 63) / 64 * 64;   // the column movers sit behind the two service waves
    const size_t lds_asm = ((size_t)FRAME_SLAB_MAX * ld + (size_t)n3) * sizeof(double);
    hipLaunchKernelGGL(frame_assemble_kernel, dim3((unsigned)B), dim3(256), lds_asm, s, p, (double*)workspace, FRAME_SLAB_MAX);
    hipLaunchKernelGGL(frame_factor_big_kernel, dim3((unsigned)B), dim3((unsigned)T), lds2, s, p, (double*)workspace, pp_use);
    return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(frame_solve_kernel, dim3((unsigned)B), dim3((unsigned)T), lds_bytes, s, p, pp_use);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
