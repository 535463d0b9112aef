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
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "../../include/openpystruct_amd.h"
#include "beam_io.hpp"

namespace opsamd {


// lane-local view of the LDS-staged inputs (see beam_math.hpp "Acc")
struct LdsAcc {
  const double* t2; const double* t6; const double* t12; const double* trl; const double* tpw; const double* tmw;
  const double* sI; const double* sF;
  const double* sZero; int nReal;   // nodal loads: local nodes >= nReal are padding and read *sZero (0.0)
  unsigned long long bits;
  __device__ __forceinline__ double c2(int i) const { return t2[i]; }
  __device__ __forceinline__ double c6(int i) const { return t6[i]; }
  __device__ __forceinline__ double c12(int i) const { return t12[i]; }
  __device__ __forceinline__ double rL(int i) const { return trl[i]; }
  __device__ __forceinline__ double pw(int i) const { return tpw[i]; }
  __device__ __forceinline__ double mw(int i) const { return tmw[i]; }
  __device__ __forceinline__ double Ie(int i) const { return sI[i]; }
  __device__ __forceinline__ double Fy(int i) const { return *(i < nReal ? sF + i : sZero); }
  __device__ __forceinline__ unsigned long long fixbits() const { return bits; }
  __device__ __forceinline__ void fence() const { __asm__ volatile("" ::: "memory"); }
};

// Results of a lane.  Lanes at the end of a beam own fewer than M real elements / nodes.  Instead of
// predicating every LDS store, stores are issued for DESCENDING i to slot min(i, last): the surplus ones
// land on the lane's own last real slot and are overwritten, in order, by the real value (LDS operations
// of a wave execute in order); a lane with nothing real points at a dummy slot.
template <int M>
struct LaneOut {
  double* sV;               // LDS slots (flat, row stride Ne) of the lane's element shears (or the dummy slot)
  unsigned lastE;           // index of the lane's last real element (0 when it has none)
  double v[M], th[M], Mz[M];  // the rest stays in registers until the inputs in LDS are dead
  __device__ __forceinline__ void elem(int i, double Vv, double Mv) {   // called for i = M-1 ... 0
    sV[(unsigned)i < lastE ? (unsigned)i : lastE] = Vv;
    Mz[i] = Mv;
  }
  __device__ __forceinline__ void node(int i, double vv, double tt) { v[i] = vv; th[i] = tt; }
};
template <int M>
__device__ __forceinline__ void lds_store_desc(double* slot, unsigned last, const double (&val)[M]) {
#pragma unroll
  for (int i = M - 1; i >= 0; --i) slot[(unsigned)i < last ? (unsigned)i : last] = val[i];
}


// Stages 2-4 for one lane: condensation, interface reduction, interior solve.  RZ: see Flags<RZ>.
template <int P, int M, bool RZ>
__device__ __forceinline__ void solve_lanes(const LdsAcc& acc, int lane, int j, int& bad, LaneOut<M>& out, double* sVrows,
                                            double* sDummy, int Ne) {
  using X = Xch<P>;
  SegState<M> st;
  __asm__ volatile("" ::: "memory");   // no LDS read of either RZ variant is hoisted above the (wave-uniform) choice between them
  seg_condense<M, RZ>(st, acc, bad);
  // The element data is re-read from LDS in the last stage instead of being carried in ~16*M VGPRs
  // across the reduction: the clobber stops the compiler from merging the two sets of loads.
  __asm__ volatile("" ::: "memory");
  IfaceRow row;
  {
    const Mat2 cup = masked_cup<M, RZ>(st, acc.bits);
    const Sym2 pc = X::template from_minus<1>(st.Scc, lane, j);
    const Vec2 pg = X::template from_minus<1>(st.gc, lane, j);
    const Mat2 pb = X::template from_minus<1>(cup, lane, j);
    row = make_row<M, RZ>(st, cup, pc, pg, pb, acc.bits);
  }
  cr_forward<P, 1>(row, lane, j, bad);
  const Sym2 G = inv_spd(row.D, bad);              // every lane's own (frozen, or for row 0 fully reduced) pivot block
  Vec2 uL = mul(G, row.f);                         // meaningful for row 0; the others are overwritten level by level
  if (j != 0) uL = Vec2{0.0, 0.0};
  cr_backward<P, P / 2>(row, G, uL, lane, j);
  const Vec2 uR = X::template from_plus<1>(uL, lane, j);
  __asm__ volatile("" ::: "memory");               // the interior solve's LDS reads stay behind the interface solve
  {  // where the lane's shears go: derived from an OPAQUE copy of the lane id, so that nothing computed for it in the
     // prologue has to stay in a register (or in scratch) across condensation and interface solve
    unsigned lz = (unsigned)lane;
    __asm__ volatile("" : "+v"(lz));
    const int gz = lz / P, e0z = (int)(lz - gz * P) * M;
    const int cntE = (Ne - e0z < 0) ? 0 : (Ne - e0z < M ? Ne - e0z : M);
    out.sV = cntE ? sVrows + gz * Ne + e0z : sDummy;
    out.lastE = cntE ? (unsigned)(cntE - 1) : 0u;
  }
  seg_solve<M, RZ>(st, acc, uL, uR, out);
}

// SHARED: x, E and wy are the same for every beam (strides 0): one LDS table per workgroup.
//
// LDS plan (one 64-lane wavefront per workgroup, BPW = 64/P beams):
//   s_tab  unit-inertia element tile entries 2E/L, 6E/L^2, 12E/L^3, 1/L, wL/2, wL^2/12 per element
//   s_a    I in padded rows [BPW][PM]            -> M (flat, row stride Ne) -> theta (flat, stride N)
//   s_b    Fy flat (row stride N, as in HBM)     -> V (flat, row stride Ne) -> v     (flat, stride N)
// waves per SIMD the register allocator is asked to leave room for (0 = no request)
constexpr int waves_per_simd(int P, int M, bool shared) {
  if (!shared) return 1;   // per-beam geometry tables make those variants LDS-limited anyway
  return (P == 16 && M == 7) ? 3 : (P == 8 && M == 13) ? 2 : (P == 32 && M == 4) ? 3 : (P == 64 && M == 2) ? 3
       : (P == 64 && M == 4) ? 3 : 1;
}


// SHARED: x, E and wy are the same for every beam (strides 0): one element table per workgroup.
// DENSE : rows of I / Fy / outputs are contiguous and every wave's run is 16-byte aligned (host-checked):
//         the wave moves each of its six streams as one flat run of 16-byte buffer accesses.
// SIZING: after the solve the wave runs the optimiser epoch of its cases on the forces it still holds in LDS
//         (sizing_math.hpp) instead of writing them out: the fused solve + step of the dataset generator.
template <int P, int M, bool SHARED, bool DENSE, bool SIZING>
__device__ __forceinline__ void beam_body(const BeamParams& p, const SizingArgs* sz) {
  constexpr int BPW = 64 / P;       // beams per wavefront
  constexpr int PM = P * M;         // padded nodes per beam (>= N)
  constexpr int TG = SHARED ? 1 : BPW;
  constexpr int NPAIR = (BPW * PM / 2 + 63) / 64;   // 16-byte pieces per lane for one staged array
#ifdef OPS_AMD_ST                                   // A/B builds of the output store policy (scripts/store_policy_ab.sh)
  constexpr int ST = OPS_AMD_ST;
#else
  constexpr int ST = (P <= 8) ? 2 : 16;             // store policy: nt for the large-batch tiling, sc1 otherwise
#endif
  constexpr int NT = (TG * PM + 63) / 64;           // table entries per lane
  __shared__ double s_tab[6][TG][PM];
  __shared__ __attribute__((aligned(16))) double s_a[BPW * PM];
  __shared__ __attribute__((aligned(16))) double s_b[BPW * PM];
  __shared__ double s_dummy[2];

  const unsigned lane = threadIdx.x;
#ifdef OPS_AMD_TRACE
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int Ne = p.Ne, N = p.Ne + 1;
  const long beam0 = (long)blockIdx.x * BPW;
  const int nb = (p.B - beam0 < BPW) ? (int)(p.B - beam0) : BPW;   // live beams of this wave
  const int nE = nb * Ne, nN = nb * N;
  if (p.active) {                   // wave-uniform: finished cases of a sizing run cost one scalar load
    unsigned any = 0;
    for (int b = 0; b < nb; ++b) any |= p.active[beam0 + b];
    if (!any) return;
  }

  // ---- stage 1a: every global load is issued before anything waits; cache-resident ones FIRST ----
  // (vmcnt retires in order: the element-table inputs and the constraint bytes hit in L2 and go out before
  //  the wave's HBM rows, so the table is computed and written to LDS while the rows are in flight; a load
  //  inside a divergent `if` would make the compiler wait for it at the join and serialise the prologue.)
  double tx0[NT], tx1[NT], tE[NT], tw[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const unsigned idx = lane + 64u * k, tb = SHARED ? 0u : idx / PM, e = idx - tb * PM;
    long bb = beam0 + (tb < (unsigned)nb ? tb : 0u);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + bb * p.x_bs, (unsigned)N * 8u);
    tx0[k] = buf_load_d(rx, e * 8u);            // out of range (padding elements) -> 0
    tx1[k] = buf_load_d(rx, e * 8u + 8u);
    if (SHARED) {                               // compile-time: scalars through the scalar cache
      tE[k] = p.E[0];
      tw[k] = p.wy[0];
    } else {
      tE[k] = buf_load_d(make_rsrc(p.E_bs ? p.E + bb * p.E_bs : p.E, p.E_bs ? (unsigned)Ne * 8u : 8u), p.E_bs ? e * 8u : 0u);
      tw[k] = buf_load_d(make_rsrc(p.wy_bs ? p.wy + bb * p.wy_bs : p.wy, p.wy_bs ? (unsigned)Ne * 8u : 8u), p.wy_bs ? e * 8u : 0u);
    }
  }
  // constraint bytes of the lane's own M + 1 nodes (L1/L2 hits), straight into the lane's bit field;
  // nodes >= N (padding) read as 0 = free
  const int g = lane / P, j = lane - g * P, e0 = j * M;
  unsigned char rfix[M + 1];
  {
    const unsigned row = (g < nb ? (unsigned)g : 0u);
    const __amdgpu_buffer_rsrc_t rf = make_rsrc(p.fix + beam0 * p.fix_bs, (unsigned)(p.fix_bs ? (nb - 1) * p.fix_bs + N : N));
    const unsigned base = row * (unsigned)p.fix_bs + (unsigned)e0;
#pragma unroll
    for (int i = 0; i <= M; ++i)
      rfix[i] = __builtin_amdgcn_raw_buffer_load_b8(rf, (int)(base + i), 0, 0);
  }
  double2 rI[NPAIR], rF[NPAIR];
  double tailI = 0.0, tailF = 0.0;              // last element of an odd-length run
  if (DENSE) {
    const __amdgpu_buffer_rsrc_t rsF = make_rsrc(p.Fy + beam0 * N, (unsigned)nN * 8u);
    if constexpr (SIZING) {                     // float32 inertias (the reference's I_tensor): 8-byte pairs, widened in registers
      const __amdgpu_buffer_rsrc_t rsI = make_rsrc(p.I32 + beam0 * Ne, (unsigned)nE * 4u);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) {
        const unsigned off = (lane + 64u * k) * 8u;
        const float2 f = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsI, (int)off, 0, 0));
        rI[k] = make_double2((double)f.x, (double)f.y);
        rF[k] = buf_load_d2(rsF, off * 2u);
      }
      tailI = (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsI, (int)((unsigned)(nE - 1) * 4u), 0, 0));
    } else {
      const __amdgpu_buffer_rsrc_t rsI = make_rsrc(p.I + beam0 * Ne, (unsigned)nE * 8u);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) {
        const unsigned off = (lane + 64u * k) * 16u;
        rI[k] = buf_load_d2(rsI, off);          // a pair that is not entirely inside the run comes back as 0
        rF[k] = buf_load_d2(rsF, off);
      }
      tailI = buf_load_d(rsI, (unsigned)(nE - 1) * 8u);
    }
    tailF = buf_load_d(rsF, (unsigned)(nN - 1) * 8u);
  }
#ifdef OPS_AMD_TRACE
  const unsigned long long ta = __builtin_amdgcn_s_memrealtime();   // all loads issued
#endif
  // padding defaults: I = 1 for the unit elements of the padding chain (columns Ne .. PM-1 of every row, and
  // whole rows of beams beyond B); padded nodal loads are read through LdsAcc::Fy's zero slot.  LDS
  // operations of one wave execute in order, so the real rows written below win.
  if (lane < 2) s_dummy[lane] = 0.0;
  if (nb == BPW) {                              // wave-uniform; the common case touches 12 columns per row
#pragma unroll
    for (int b = 0; b < BPW; ++b)
      for (unsigned e = Ne + lane; e < PM; e += 64) s_a[b * PM + e] = 1.0;
  } else {
    for (unsigned idx = lane; idx < BPW * PM; idx += 64) s_a[idx] = 1.0;
  }
  // ---- stage 0: element table (unit-inertia stiffness tile entries, 1/L, UDL loads) ----
  // element Ne has no stiffness; the elements beyond it are unit elements (L = 1, EI = 1): together with the
  // implicit clamp behind the last lane (its right neighbour does not exist: u = 0) the padding is a free
  // cantilever hanging off nothing -- every pivot positive, no real DOF touched, no constraint flags needed.
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const unsigned idx = lane + 64u * k, tb = SHARED ? 0u : idx / PM, e = idx - tb * PM;
    if (idx < (unsigned)(TG * PM)) {
      const bool real = (int)e < Ne, pad = (int)e > Ne;
      const double L = tx1[k] - tx0[k];
      const double rl0 = fast_rcp(real ? L : 1.0);
      const double c2r = 2.0 * tE[k] * rl0, pwr = 0.5 * tw[k] * L;
      const double rl = real ? rl0 : (pad ? 1.0 : 0.0);
      const double c2 = real ? c2r : (pad ? 2.0 : 0.0);
      const double c6 = 3.0 * c2 * rl, c12 = 2.0 * c6 * rl;
      const double pw = real ? pwr : 0.0, mw = pw * L * (1.0 / 6.0);
      s_tab[0][tb][e] = c2;  s_tab[1][tb][e] = c6;  s_tab[2][tb][e] = c12;
      s_tab[3][tb][e] = rl;  s_tab[4][tb][e] = pw;  s_tab[5][tb][e] = mw;
    }
  }
#ifdef OPS_AMD_TRACE
  const unsigned long long tb_ = __builtin_amdgcn_s_memrealtime();  // table written
#endif
  // ---- stage 1b: the staged rows into LDS ----
  if (DENSE) {
    const bool oddI = nE & 1, oddF = nN & 1;    // wave-uniform
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const unsigned i0 = 2u * (lane + 64u * k);
      // I -> padded rows: (beam, element) of flat index i0 by multiply-shift division
      const unsigned b0 = (i0 * p.magic_ne) >> 20, e = i0 - b0 * Ne;
      if (!(Ne & 1)) {                          // wave-uniform: a pair never straddles two rows
        if ((int)i0 < nE) *reinterpret_cast<double2*>(&s_a[b0 * PM + e]) = rI[k];
      } else if ((int)(i0 + 1) < nE) {
        s_a[b0 * PM + e] = rI[k].x;
        if ((int)(e + 1) < Ne) s_a[b0 * PM + e + 1] = rI[k].y;
        else s_a[(b0 + 1) * PM] = rI[k].y;
      } else if (oddI && (int)i0 == nE - 1) {
        s_a[b0 * PM + e] = tailI;
      }
      // Fy -> flat, as in HBM
      if ((int)(i0 + 1) < nN) *reinterpret_cast<double2*>(&s_b[i0]) = rF[k];
      else if (oddF && (int)i0 == nN - 1) s_b[i0] = tailF;
    }
  } else {
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
      if (b >= nb) break;
      const double* Fb = p.Fy + (beam0 + b) * p.Fy_bs;
      for (int e = lane; e < N; e += 64) {
        if (e < Ne) {
          if constexpr (SIZING) s_a[b * PM + e] = (double)p.I32[(beam0 + b) * Ne + e];
          else s_a[b * PM + e] = p.I[(beam0 + b) * p.I_bs + e];
        }
        s_b[b * N + e] = Fb[e];
      }
    }
  }
  __syncthreads();
#ifdef OPS_AMD_TRACE
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- stages 2-4: per-lane condensation, interface reduction, interior solve ----
  LdsAcc acc;
  {
    const int tb = SHARED ? 0 : g;
    acc.t2 = &s_tab[0][tb][e0];  acc.t6 = &s_tab[1][tb][e0];  acc.t12 = &s_tab[2][tb][e0];
    acc.trl = &s_tab[3][tb][e0]; acc.tpw = &s_tab[4][tb][e0]; acc.tmw = &s_tab[5][tb][e0];
    acc.sI = &s_a[g * PM + e0];
    acc.sF = &s_b[g * N + e0];
    acc.sZero = &s_dummy[1];
    acc.nReal = N - e0;   // may be <= 0 (a lane that owns padding only) or >= M
    unsigned long long bits = 0;
#pragma unroll
    for (int i = 0; i <= M; ++i)   // a node at or beyond N is padding: free (the row's bytes end at N)
      bits |= (unsigned long long)((e0 + i < N) ? (rfix[i] & 3) : 0) << (2 * i);
    acc.bits = bits;
  }
  int bad = 0;
  LaneOut<M> out;                   // sV / lastE: set by solve_lanes right before the interior solve
  const bool any_rz = __ballot((acc.bits & 0xAAAAAAAAAAAAAAAAull) != 0ull) != 0ull;   // wave-uniform
  if (any_rz) solve_lanes<P, M, true>(acc, lane, j, bad, out, s_b, s_dummy, Ne);
  else        solve_lanes<P, M, false>(acc, lane, j, bad, out, s_b, s_dummy, Ne);
  // the epilogue's lane geometry from an opaque copy of the lane id (see solve_lanes): recomputed, not kept
  unsigned lane_e = lane;
  __asm__ volatile("" : "+v"(lane_e));
  const int g_e = lane_e / P, j_e = (int)(lane_e - g_e * P), e0_e = j_e * M;
  const int cntE = (Ne - e0_e < 0) ? 0 : (Ne - e0_e < M ? Ne - e0_e : M);   // real elements / nodes of this lane
  const int cntN = (N - e0_e < 0) ? 0 : (N - e0_e < M ? N - e0_e : M);

  // a beam is bad if any of its P lanes met a non-positive pivot; its outputs become NaN
  const unsigned long long bal = __ballot(bad != 0);
  const unsigned long long grp = (P == 64) ? ~0ull : (((1ull << (P % 64)) - 1ull) << (g_e * P));
  const bool gbad = (bal & grp) != 0ull;
  const double qnan = __builtin_nan("");
  if (bal != 0ull && gbad) {                    // first test is wave-uniform: nothing to do in the common case
#pragma unroll
    for (int i = 0; i < M; ++i) {
      out.v[i] = qnan; out.th[i] = qnan; out.Mz[i] = qnan;
      if (i < cntE) out.sV[i] = qnan;
    }
  }
  if (j_e == 0 && g_e < nb && p.status)   // write-through like the other outputs
    __hip_atomic_store(&p.status[beam0 + g_e], gbad ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef OPS_AMD_TRACE
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
#endif

  // element rows: V already in s_b (flat, stride Ne); M joins it in s_a now that I is dead
  lds_store_desc<M>(cntE ? &s_a[g_e * Ne + e0_e] : s_dummy, out.lastE, out.Mz);
  wave_lds_fence();
  if constexpr (SIZING) {           // V in s_b, M in s_a (flat, stride Ne): one optimiser epoch per live, active case
    // every case's state loads go out before the first one is used (a wave holds BPW cases: one HBM latency, not BPW)
    CaseRegs<2> cr[BPW];            // host-checked: Ne <= 128
    bool on[BPW];
#pragma unroll
    for (int gb = 0; gb < BPW; ++gb) {
      on[gb] = gb < nb && sz->active[beam0 + gb] != 0;     // wave-uniform
      if (on[gb]) load_case<2>((int)lane_e, beam0 + gb, Ne, *sz, cr[gb]);
    }
#pragma unroll
    for (int gb = 0; gb < BPW; ++gb) {
      if (!on[gb]) continue;
      const double* Vb = &s_b[gb * Ne];
      const double* Mb = &s_a[gb * Ne];
      step_case<2>((int)lane_e, beam0 + gb, Ne, *sz, cr[gb], [&](int e) { return (float)Vb[e]; }, [&](int e) { return (float)Mb[e]; });
    }
    return;
  }
  if (p.f32_forces) {               // wave-uniform; forces-only by construction (host-checked)
    float* V32 = reinterpret_cast<float*>(p.V) + beam0 * Ne;
    float* M32 = reinterpret_cast<float*>(p.M) + beam0 * Ne;
    if (DENSE) {
      const __amdgpu_buffer_rsrc_t rV = make_rsrc(V32, (unsigned)nE * 4u);
      const __amdgpu_buffer_rsrc_t rM = make_rsrc(M32, (unsigned)nE * 4u);
#pragma unroll
      for (int k = 0; k < NPAIR; ++k) {
        const unsigned i0 = 2u * (lane_e + 64u * k);
        if (k + 1 < NPAIR || i0 < BPW * PM) {
          const double2 dv = *reinterpret_cast<const double2*>(&s_b[i0]), dm = *reinterpret_cast<const double2*>(&s_a[i0]);
          const float2 fv = make_float2((float)dv.x, (float)dv.y), fm = make_float2((float)dm.x, (float)dm.y);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, fv), rV, (int)(i0 * 4u), 0, ST);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, fm), rM, (int)(i0 * 4u), 0, ST);
        }
      }
      if ((nE & 1) && lane_e == 0) { V32[nE - 1] = (float)s_b[nE - 1]; M32[nE - 1] = (float)s_a[nE - 1]; }
    } else {
      for (int idx = lane_e; idx < nE; idx += 64) { V32[idx] = (float)s_b[idx]; M32[idx] = (float)s_a[idx]; }
    }
    return;
  }
  // two rows of doubles (s_b -> ra, s_a -> rb) of n values each, 16-byte stores with cache policy AUX
  auto store_rows = [&](auto aux, const __amdgpu_buffer_rsrc_t ra, const __amdgpu_buffer_rsrc_t rb, int n_) {
    constexpr int AUX = decltype(aux)::value;
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
      const unsigned i0 = 2u * (lane_e + 64u * k);
      if (k + 1 < NPAIR || i0 < BPW * PM) {     // LDS bound; the buffer descriptor drops pairs beyond the run
        buf_store_d2<AUX>(ra, i0 * 8u, *reinterpret_cast<const double2*>(&s_b[i0]));
        buf_store_d2<AUX>(rb, i0 * 8u, *reinterpret_cast<const double2*>(&s_a[i0]));
      }
    }
    if ((n_ & 1) && lane_e == 0) {
      buf_store_d<AUX>(ra, (unsigned)(n_ - 1) * 8u, s_b[n_ - 1]);
      buf_store_d<AUX>(rb, (unsigned)(n_ - 1) * 8u, s_a[n_ - 1]);
    }
  };
  const bool nt_out = p.stream_out != 0 && ST != 2;          // wave-uniform
  if (DENSE) {
    const __amdgpu_buffer_rsrc_t rV = make_rsrc(p.V + beam0 * Ne, (unsigned)nE * 8u);
    const __amdgpu_buffer_rsrc_t rM = make_rsrc(p.M + beam0 * Ne, (unsigned)nE * 8u);
    if (nt_out) store_rows(std::integral_constant<int, 2>{}, rV, rM, nE);
    else        store_rows(std::integral_constant<int, ST>{}, rV, rM, nE);
  } else {
    for (int idx = lane_e; idx < nE; idx += 64) {
      p.V[beam0 * Ne + idx] = s_b[idx];
      p.M[beam0 * Ne + idx] = s_a[idx];
    }
  }
  if (!p.v) return;                 // forces-only call (wave-uniform): the sizing epochs never read displacements
  wave_lds_fence();
  // nodal rows (flat, stride N)
  lds_store_desc<M>(cntN ? &s_b[g_e * N + e0_e] : s_dummy, cntN ? (unsigned)(cntN - 1) : 0u, out.v);
  lds_store_desc<M>(cntN ? &s_a[g_e * N + e0_e] : s_dummy, cntN ? (unsigned)(cntN - 1) : 0u, out.th);
  wave_lds_fence();
  if (DENSE) {
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(p.v + beam0 * N, (unsigned)nN * 8u);
    const __amdgpu_buffer_rsrc_t rt = make_rsrc(p.theta + beam0 * N, (unsigned)nN * 8u);
    if (nt_out) store_rows(std::integral_constant<int, 2>{}, rv, rt, nN);
    else        store_rows(std::integral_constant<int, ST>{}, rv, rt, nN);
  } else {
    for (int idx = lane_e; idx < nN; idx += 64) {
      p.v[beam0 * N + idx] = s_b[idx];
      p.theta[beam0 * N + idx] = s_a[idx];
    }
  }
#ifdef OPS_AMD_TRACE
  if (p.trace && lane_e == 0) {   // per-wave phase stamps (100 MHz clock) + hardware id, for scripts/trace_run.py
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((16 - 1) << 11));
    unsigned long long* q = p.trace + 8 * (unsigned long long)blockIdx.x;
    q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; q[4] = hw; q[5] = ta; q[6] = tb_; q[7] = tb_;
  }
#endif
}

template <int P, int M, bool SHARED, bool DENSE>
__global__ __launch_bounds__(64, waves_per_simd(P, M, SHARED)) void beam_solve_kernel(const BeamParams p) {
  beam_body<P, M, SHARED, DENSE, false>(p, nullptr);
}

template <int P, int M, bool SHARED, bool DENSE>
__global__ __launch_bounds__(64, waves_per_simd(P, M, SHARED)) void beam_sizing_epoch_kernel(const BeamParams p, const SizingArgs sz) {
  beam_body<P, M, SHARED, DENSE, true>(p, &sz);
}

// ------------------------------------------------------------------------------------------
// host side: tiling choice + launch
// ------------------------------------------------------------------------------------------
struct Tiling { int P, M; const char* name_shared; const char* name_general; };

// every compiled (P, M); a tiling serves Ne with Ne + 1 <= P * M
static const Tiling kTilings[] = {
    {8, 13, "beam_solve_kernel<8, 13, true, true>", "beam_solve_kernel<8, 13, false, true>"},
    {16, 7, "beam_solve_kernel<16, 7, true, true>", "beam_solve_kernel<16, 7, false, true>"},
    {32, 4, "beam_solve_kernel<32, 4, true, true>", "beam_solve_kernel<32, 4, false, true>"},
    {64, 2, "beam_solve_kernel<64, 2, true, true>", "beam_solve_kernel<64, 2, false, true>"},
    {64, 4, "beam_solve_kernel<64, 4, true, true>", "beam_solve_kernel<64, 4, false, true>"},
    {64, 8, "beam_solve_kernel<64, 8, true, true>", "beam_solve_kernel<64, 8, false, true>"},
    {64, 16, "beam_solve_kernel<64, 16, true, true>", "beam_solve_kernel<64, 16, false, true>"},
};
static const int kNumTilings = sizeof(kTilings) / sizeof(kTilings[0]);

// Default use of a row-staged tiling (same-box A/B, profiles/r03_notes.md; us per launch, classic 16 / rows 16 / rows 8 /
// classic 8): 10^4 beams 13.1 / 12.7 / 13.1 / 14.8 eager (13.9 / 12.1 graph-replayed), 4e4 41.8 / 38.5 / 41.7 / 42.8,
// 1e5 100 / 91 / 88 / 92, 3e5 306 / 279 / 282 / 274, 2^20 1067 / 993 / 987 / 951.  The 16-lane rows kernel serves every
// batch below 2^17 beams; above, the flat 16-byte-aligned streams of beam_solve.hip's 8-lane kernel are worth 3-4 %.
// The fat-wave tiling (P = 6) runs a third fewer instructions per SIMD at 10^4 beams, but with ONE wave per SIMD every
// wave of the chip is in the same phase and the store phase (32 MB, ~3.8 us from cache) overlaps nothing: 15.1 us.
static bool fat_default(const FatTiling& f, int B, int Ne) {
  return f.P == 16 && Ne + 1 <= f.P * f.M && B < (1 << 17);
}

static const Tiling* choose_tiling(int B, int Ne, int tiling) {
  const int N = Ne + 1;
  if (tiling != 0) {
    for (int t = 0; t < kNumTilings; ++t)
      if (kTilings[t].P == tiling && kTilings[t].P * kTilings[t].M >= N) return &kTilings[t];
    return nullptr;
  }
  // default (measured, profiles/r01_notes.md): 16 lanes per beam while the batch is a single round of
  // waves (its short waves finish a 10^4-beam launch soonest); 8 lanes per beam (40 % less arithmetic per
  // beam, two waves per SIMD) once the batch is large enough to keep every SIMD busy for several rounds
  if (N <= 8 * 13 && B >= 32768) return &kTilings[0];   // (reached for B >= 2^17 when the rows kernel is eligible)
  if (N <= 16 * 7) return &kTilings[1];
  for (int t = 0; t < kNumTilings; ++t)
    if (kTilings[t].P * kTilings[t].M >= N) return &kTilings[t];
  return nullptr;
}

// Row-staged tilings (beam_fat.hip).  P = 6 (the fat-wave tiling) exists only there; P = 16 / 8 exist in both files and
// an explicit `tiling` picks the beam_fat.hip variant with OPS_AMD_TILING_ROWS.  The default dispatch (tiling 0) uses one
// where fat_default says so.
static const FatTiling* choose_fat(int B, int Ne, int tiling) {
  const int N = Ne + 1;
  const bool rows = (tiling & OPS_AMD_TILING_ROWS) != 0;
  tiling &= ~OPS_AMD_TILING_ROWS;
  for (int t = 0; t < kNumFatTilings; ++t) {
    const FatTiling& f = kFatTilings[t];
    const bool serves = f.P * f.M >= N;
    if (tiling == f.P && (rows || f.P == 6)) return serves ? &f : nullptr;
    if (tiling == 0 && !rows && serves && fat_default(f, B, Ne)) return &f;
  }
  return nullptr;
}
static bool is_fat_tiling(int tiling) { return tiling == 6 || (tiling & OPS_AMD_TILING_ROWS) != 0; }

// ---- automatic cache policy of the result stores ---------------------------------------------------------------
// Write-through (sc1) stores are the best when the result buffers sit in the 256 MiB Infinity Cache and the worst when
// they do not (10^4 beams: 13.8 / 18.4 us against 15.4 / 15.8 us with non-temporal stores, profiles/r02_notes.md).  The
// kernel cannot see residency; the library can estimate it: it remembers when (in bytes moved by its own launches) each
// result buffer was last written.  A buffer is taken as resident if the traffic since then plus this call's own bytes
// fit the cache: a caller that re-solves into the same buffers gets sc1, one that cycles through more output than the
// cache holds -- or a single launch bigger than the cache -- gets nt, no flag needed (OPS_AMD_TILING_STREAM_OUT still
// forces nt).  Under HIP-graph capture the choice made at capture time is what replays.
struct OutSeen { const void* p; unsigned long long stamp; };
static std::mutex g_seen_mu;
static OutSeen g_seen[64];
static unsigned g_seen_pos = 0;
static unsigned long long g_traffic = 0;
static bool results_cache_resident(const void* out, unsigned long long call_bytes) {
  constexpr unsigned long long kFits = 224ull << 20;
  std::lock_guard<std::mutex> lock(g_seen_mu);
  bool resident = false, found = false;
  for (OutSeen& e : g_seen) {
    if (e.p == out) {
      resident = (g_traffic - e.stamp) + call_bytes <= kFits;
      e.stamp = g_traffic + call_bytes;
      found = true;
      break;
    }
  }
  if (!found) g_seen[g_seen_pos++ % 64] = OutSeen{out, g_traffic + call_bytes};
  g_traffic += call_bytes;
  return resident;
}

static thread_local char g_last_error[256] = {0};

// shared by every translation unit of the library (ops_amd_last_error reports it)
void set_last_error(const char* msg) {
  int k = 0;
  for (; msg && msg[k] && k < 255; ++k) g_last_error[k] = msg[k];
  g_last_error[k] = 0;
}

template <int P, int M>
static hipError_t launch(const BeamParams& p, bool shared, hipStream_t stream) {
  constexpr int BPW = 64 / P;
  const unsigned grid = (unsigned)((p.B + BPW - 1) / BPW);
  if (shared && p.dense)
    hipLaunchKernelGGL((beam_solve_kernel<P, M, true, true>), dim3(grid), dim3(64), 0, stream, p);
  else if (shared)
    hipLaunchKernelGGL((beam_solve_kernel<P, M, true, false>), dim3(grid), dim3(64), 0, stream, p);
  else if (p.dense)
    hipLaunchKernelGGL((beam_solve_kernel<P, M, false, true>), dim3(grid), dim3(64), 0, stream, p);
  else
    hipLaunchKernelGGL((beam_solve_kernel<P, M, false, false>), dim3(grid), dim3(64), 0, stream, p);
  return hipGetLastError();
}

template <int P, int M>
static hipError_t launch_sizing(const BeamParams& p, const SizingArgs& sz, bool shared, hipStream_t stream) {
  constexpr int BPW = 64 / P;
  const unsigned grid = (unsigned)((p.B + BPW - 1) / BPW);
  if (shared && p.dense)
    hipLaunchKernelGGL((beam_sizing_epoch_kernel<P, M, true, true>), dim3(grid), dim3(64), 0, stream, p, sz);
  else if (shared)
    hipLaunchKernelGGL((beam_sizing_epoch_kernel<P, M, true, false>), dim3(grid), dim3(64), 0, stream, p, sz);
  else if (p.dense)
    hipLaunchKernelGGL((beam_sizing_epoch_kernel<P, M, false, true>), dim3(grid), dim3(64), 0, stream, p, sz);
  else
    hipLaunchKernelGGL((beam_sizing_epoch_kernel<P, M, false, false>), dim3(grid), dim3(64), 0, stream, p, sz);
  return hipGetLastError();
}

}  // namespace opsamd

using namespace opsamd;

extern "C" {

int ops_amd_abi_version(void) { return OPS_AMD_ABI_VERSION; }
int ops_amd_max_elements(void) { return 64 * 16 - 1; }
const char* ops_amd_last_error(void) { return g_last_error; }

const char* ops_beam_solve_kernel_name(int B, int Ne, int tiling) {
  tiling &= ~OPS_AMD_TILING_STREAM_OUT;
  if (const FatTiling* f = choose_fat(B, Ne, tiling)) return f->name;   // shared geometry, dense rows (what the name is asked for)
  if (is_fat_tiling(tiling)) return "";
  const Tiling* t = choose_tiling(B, Ne, tiling & ~OPS_AMD_TILING_ROWS);
  return t ? t->name_shared : "";
}

static int solve_impl(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                      const double* I, long I_bstride, const uint8_t* fix, long fix_bstride,
                      const double* Fy, long Fy_bstride, const double* wy, long wy_bstride, double* v,
                      double* theta, double* V, double* M, int32_t* status, const uint8_t* active, int f32_forces, int tiling,
                      void* stream, const SizingArgs* sz = nullptr) {
  if (B < 0 || Ne < 1) return OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!x || !E || (!I && !sz) || !fix || !Fy || !wy || ((!V || !M) && !sz) || ((v == nullptr) != (theta == nullptr))) return OPS_AMD_ERR_INVALID_ARG;
  if (I_bstride < Ne || Fy_bstride < Ne + 1) return OPS_AMD_ERR_INVALID_ARG;
  if ((x_bstride != 0 && x_bstride < Ne + 1) || (fix_bstride != 0 && fix_bstride < Ne + 1) ||
      (E_bstride != 0 && E_bstride < Ne) || (wy_bstride != 0 && wy_bstride < Ne))
    return OPS_AMD_ERR_INVALID_ARG;
  if (Ne > ops_amd_max_elements()) return OPS_AMD_ERR_UNSUPPORTED;
  int stream_out = (tiling & OPS_AMD_TILING_STREAM_OUT) ? 1 : 0;
  tiling &= ~OPS_AMD_TILING_STREAM_OUT;
  if (!sz && !f32_forces) {   // full-precision result rows: 4 925 B (or 3 309 B, forces only) per solve
    const unsigned long long bytes = (unsigned long long)B * (v ? 4925ull : 3309ull) * (unsigned long long)Ne / 100ull;
    if (!results_cache_resident(V, bytes)) stream_out = 1;
  }
  const bool shared = (x_bstride == 0 && E_bstride == 0 && wy_bstride == 0);
  const uintptr_t align_bits = (uintptr_t)I | (uintptr_t)Fy | (uintptr_t)v | (uintptr_t)theta | (uintptr_t)V | (uintptr_t)M | (sz ? 2 * (uintptr_t)sz->I : 0);
  const bool rows_dense = (I_bstride == Ne) && (Fy_bstride == Ne + 1) && ((align_bits & 15u) == 0);
  // fat-wave tilings: shared geometry and constraint mask, plain double-precision solve (rows move one by one: any
  // stride, 8-byte alignment)
  const FatTiling* fat = nullptr;
  {
    // (the fused sizing epoch: 16-lane rows kernel only; an `active` mask is the sizing epoch's, nobody else's)
    const bool fat_ok = shared && fix_bstride == 0 && !f32_forces && (sz ? true : !active);
    const FatTiling* f = choose_fat(B, Ne, tiling);
    if (f && fat_ok && (!sz || f->P == 16)) fat = f;
    else if (is_fat_tiling(tiling)) return f ? OPS_AMD_ERR_UNSUPPORTED : OPS_AMD_ERR_INVALID_ARG;
  }
  const Tiling* t = fat ? nullptr : choose_tiling(B, Ne, tiling & ~OPS_AMD_TILING_ROWS);
  if (!t && !fat) return tiling ? OPS_AMD_ERR_INVALID_ARG : OPS_AMD_ERR_UNSUPPORTED;

  BeamParams p{B, Ne, x, x_bstride, E, E_bstride, I, I_bstride, fix, fix_bstride, Fy, Fy_bstride,
               wy, wy_bstride, v, theta, V, M, status, sz ? sz->I : nullptr, active, stream_out, f32_forces, 0, nullptr, 0u, 0u};
#ifdef OPS_AMD_TRACE
  { const char* e = getenv("OPS_AMD_TRACE_PTR"); if (e) p.trace = (unsigned long long*)strtoull(e, nullptr, 0); }
#endif
  {
    const int bpw = 64 / (fat ? fat->P : t->P);
    // 16-byte aligned rows (8-byte for the float32 inertias of the sizing epoch, which are read in 8-byte pairs)
    p.dense = rows_dense && ((bpw * Ne) % 2 == 0) && ((bpw * (Ne + 1)) % 2 == 0);
    p.magic_ne = ((1u << 20) + (unsigned)Ne - 1u) / (unsigned)Ne;
    p.magic_n = ((1u << 20) + (unsigned)Ne) / (unsigned)(Ne + 1);
  }
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipSuccess;
  if (fat && sz) err = launch_fat_sizing(p, *sz, fat->P, fat->M, s);
  else if (fat) err = launch_fat(p, fat->P, fat->M, s);
  else if (sz) {
    if (t->P == 8 && t->M == 13) err = launch_sizing<8, 13>(p, *sz, shared, s);
    else if (t->P == 16 && t->M == 7) err = launch_sizing<16, 7>(p, *sz, shared, s);
    else if (t->P == 32 && t->M == 4) err = launch_sizing<32, 4>(p, *sz, shared, s);
    else if (t->P == 64 && t->M == 2) err = launch_sizing<64, 2>(p, *sz, shared, s);
    else if (t->P == 64 && t->M == 4) err = launch_sizing<64, 4>(p, *sz, shared, s);
    else if (t->P == 64 && t->M == 8) err = launch_sizing<64, 8>(p, *sz, shared, s);
    else return OPS_AMD_ERR_UNSUPPORTED;
  }
  else if (t->P == 8 && t->M == 13) err = launch<8, 13>(p, shared, s);
  else if (t->P == 16 && t->M == 7) err = launch<16, 7>(p, shared, s);
  else if (t->P == 32 && t->M == 4) err = launch<32, 4>(p, shared, s);
  else if (t->P == 64 && t->M == 2) err = launch<64, 2>(p, shared, s);
  else if (t->P == 64 && t->M == 4) err = launch<64, 4>(p, shared, s);
  else if (t->P == 64 && t->M == 8) err = launch<64, 8>(p, shared, s);
  else if (t->P == 64 && t->M == 16) err = launch<64, 16>(p, shared, s);
  if (err != hipSuccess) {
    set_last_error(hipGetErrorString(err));
    return OPS_AMD_ERR_LAUNCH;
  }
  return OPS_AMD_OK;
}

int ops_beam_solve_batched_f64(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                               const double* I, long I_bstride, const uint8_t* fix, long fix_bstride,
                               const double* Fy, long Fy_bstride, const double* wy, long wy_bstride, double* v,
                               double* theta, double* V, double* M, int32_t* status, int tiling, void* stream) {
  if (B > 0 && Ne >= 1 && (!v || !theta)) return OPS_AMD_ERR_INVALID_ARG;
  return solve_impl(B, Ne, x, x_bstride, E, E_bstride, I, I_bstride, fix, fix_bstride, Fy, Fy_bstride, wy, wy_bstride, v, theta, V, M,
                    status, nullptr, 0, tiling, stream);
}

int ops_beam_solve_forces_f64(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                              const double* I, long I_bstride, const uint8_t* fix, long fix_bstride,
                              const double* Fy, long Fy_bstride, const double* wy, long wy_bstride, double* V, double* M,
                              int32_t* status, const uint8_t* active, int tiling, void* stream) {
  return solve_impl(B, Ne, x, x_bstride, E, E_bstride, I, I_bstride, fix, fix_bstride, Fy, Fy_bstride, wy, wy_bstride, nullptr, nullptr,
                    V, M, status, active, 0, tiling, stream);
}

int ops_beam_sizing_epoch_f32(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                              const uint8_t* fix, long fix_bstride, const double* Fy, long Fy_bstride, const double* wy,
                              long wy_bstride, float* I, float* I_last, float* exp_avg, float* exp_avg_sq, float* best_loss,
                              int32_t* patience_cnt, int32_t* epochs_run, uint8_t* active, float* last_loss,
                              const ops_sizing_params* hp, const float* schedule, int32_t* status, int tiling, void* stream) {
  if (Ne > 128) return OPS_AMD_ERR_UNSUPPORTED;       // two elements per lane in the fused step
  if (!I || !I_last || !exp_avg || !exp_avg_sq || !best_loss || !patience_cnt || !epochs_run || !active || !last_loss || !hp)
    return OPS_AMD_ERR_INVALID_ARG;
  const SizingArgs sz{I, nullptr, I_last, exp_avg, exp_avg_sq, best_loss, patience_cnt, epochs_run, active, last_loss, nullptr, nullptr, *hp, schedule};
  // the fused kernel carries the cases' optimiser state in registers next to the solve: the 16-lane tiling (three waves
  // per SIMD) beats the 8-lane one at every batch size here (2e5 cases: 0.095 vs 0.104 s), unlike the plain solve
  if (tiling == 0 && Ne + 1 <= 16 * 7) {
    const bool rows_ok = x_bstride == 0 && E_bstride == 0 && wy_bstride == 0 && fix_bstride == 0;   // else: per-case geometry (random bridges)
    tiling = rows_ok ? (16 | OPS_AMD_TILING_ROWS) : 16;
  }
  return solve_impl(B, Ne, x, x_bstride, E, E_bstride, nullptr, Ne, fix, fix_bstride, Fy, Fy_bstride, wy, wy_bstride, nullptr, nullptr,
                    nullptr, nullptr, status, active, 0, tiling, stream, &sz);
}

int ops_beam_solve_forces_f32(int B, int Ne, const double* x, long x_bstride, const double* E, long E_bstride,
                              const double* I, long I_bstride, const uint8_t* fix, long fix_bstride,
                              const double* Fy, long Fy_bstride, const double* wy, long wy_bstride, float* V32, float* M32,
                              int32_t* status, const uint8_t* active, int tiling, void* stream) {
  return solve_impl(B, Ne, x, x_bstride, E, E_bstride, I, I_bstride, fix, fix_bstride, Fy, Fy_bstride, wy, wy_bstride, nullptr, nullptr,
                    reinterpret_cast<double*>(V32), reinterpret_cast<double*>(M32), status, active, 1, tiling, stream);
}

}  // extern "C"
