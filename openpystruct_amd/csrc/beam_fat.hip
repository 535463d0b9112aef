// "Fat wave" tilings of the batched Euler-Bernoulli beam solve: few lanes per beam, many beams per wave.
//
// Same path as beam_solve.hip (reference: `setup_model` + `ops.analyze(1)` + `ops.eleResponse` + `ops.nodeDisp`,
// /root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py:89-124, :180-190, :224-232), same arithmetic
// (beam_math.hpp), same C ABI.  What changes is the mapping:
//
//   * P lanes per beam with P NOT a divisor of 64: BPW = 64 / P beams per wave, 64 - BPW * P lanes idle.  The
//     contract workload (10 000 beams x 100 elements) on 1 024 SIMDs is 9.77 beams per SIMD: P = 6, M = 17 gives
//     10 beams per wave = 1 000 waves = ONE wave per SIMD, where the 16-lane tiling (4 beams per wave) leaves the
//     busiest SIMDs with three waves = 12 beams' worth of instructions, and the interface reduction -- a third of a
//     16-lane wave's instructions, paid per WAVE whatever the number of live rows -- is shared by ten beams
//     instead of four.
//   * the interface rows are exchanged through the LDS crossbar (ds_bpermute; groups of 6 lanes do not sit inside
//     DPP rows), at no VALU cost;
//   * a single resident wave has no partner to hide LDS latency behind: every element's inputs are requested one
//     element ahead (beam_math.hpp, seg_condense_pf / seg_solve_pf), and everything an interface level does not
//     need the pivot inverse for is fetched before the inverse is computed;
//   * every staged array is a set of PADDED rows [beam][P * M] in LDS, moved row by row: one 16-byte access per lane
//     and row (a row of <= 128 doubles is one wave instruction), no index arithmetic, any row stride and any
//     8-byte alignment in HBM.  A lane's results go straight into its own slots of such rows while the back
//     substitution produces them; the intermediate right-hand sides h_i of the interior solve wait in the very
//     slots that the moments and deflections overwrite (registers: pivot inverses and rotations only);
//   * constraint flags: the wave ballots the shared mask once (two byte loads per lane) instead of M + 1 byte
//     loads per lane;
//   * 37 KB of LDS per one-wave workgroup: at most four per CU, i.e. one per SIMD, whatever the register count.
//
// Shared geometry only (the generator's and the bench's case); everything else stays with beam_solve.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/openpystruct_amd.h"
#include "beam_io.hpp"

namespace opsamd {

// lane-local view of the staged inputs: one element's inputs per call (beam_math.hpp, "Acc" of the _pf phases)
template <int PM>
struct FatAcc {
  const double* tab;     // &s_tab[e0]; the six table rows are PM doubles apart
  const double* sI;      // &s_a[g * PM + e0]
  const double* sF;      // &s_b[g * PM + e0]   (padding nodes hold 0)
  FixPair bits;
  __device__ __forceinline__ ElemIn elem(int i) const {
    return ElemIn{tab[i], tab[PM + i], tab[2 * PM + i], tab[3 * PM + i], tab[4 * PM + i], tab[5 * PM + i], sI[i], sF[i]};
  }
  __device__ __forceinline__ FixPair fixbits() const { return bits; }
  __device__ __forceinline__ void fence() const { __asm__ volatile("" ::: "memory"); }
};

// Results of a lane.  V always goes into the lane's own slots of the (dead) load rows.  FAT: M and v go straight into
// rows of their own and only theta waits in registers; LEAN (two row sets only): M overwrites the inertia slot it has
// just consumed, v and theta wait in registers for the epilogue.
template <int M, bool FAT>
struct RowOut {
  double* sV; double* sM; double* sv;
  double th[M];
  double vr[FAT ? 1 : M];
  __device__ __forceinline__ void elem(int i, double Vv, double Mv) { sV[i] = Vv; sM[i] = Mv; }
  __device__ __forceinline__ void node(int i, double vv, double tt) {
    if constexpr (FAT) sv[i] = vv; else vr[i] = vv;
    th[i] = tt;
  }
};
// h_i of the interior solve.  FAT: parked in the slots the moment / deflection of element / node i will overwrite
// (the tiling has to stay inside 256 VGPRs); LEAN: registers.
template <int M, bool FAT>
struct ParkH {
  double* hx; double* hy;
  Vec2 r[FAT ? 1 : M];
  __device__ __forceinline__ void put(int i, const Vec2& h) {
    if constexpr (FAT) { hx[i] = h.x; hy[i] = h.y; } else r[i] = h;
  }
  __device__ __forceinline__ Vec2 get(int i) const {
    if constexpr (FAT) return Vec2{hx[i], hy[i]}; else return r[i];
  }
};

#ifdef OPS_AMD_TRACE
#define FAT_STAMP(k) stamps[k] = __builtin_amdgcn_s_memrealtime()
#else
#define FAT_STAMP(k) ((void)0)
#endif

// the interface pieces of the wave's lanes in LDS (beam_math.hpp, iface_thomas): lane l's piece at [l]
struct FatIface {
  const IfacePiece* grp;   // piece of lane 0 of the reader's beam
  __device__ __forceinline__ IfacePiece piece(int k) const { return grp[k]; }
  __device__ __forceinline__ Mat2 cup(int k) const { return grp[k].cup; }
  __device__ __forceinline__ void fence() const { __asm__ volatile("" ::: "memory"); }
};

// FAT : interface by publish / gather / block-Thomas in every lane (few rows; groups do not sit in DPP rows)
// LEAN: interface by cyclic reduction over DPP row shifts, as beam_solve.hip (P a divisor of 16)
template <int P, int M, bool RZ, bool FAT, int PF>
__device__ __forceinline__ void row_solve_lanes(const FatAcc<P * M>& acc, int g, int j, int& bad, RowOut<M, FAT>& out, ParkH<M, FAT>& hs,
                                                IfacePiece* pieces, unsigned lane, unsigned long long* stamps) {
  SegState<M> st;
  __asm__ volatile("" ::: "memory");
  seg_condense_pf<M, RZ>(st, acc, bad);
  __asm__ volatile("" ::: "memory");
  FAT_STAMP(6);
  Vec2 uL, uR{0.0, 0.0};
  if constexpr (FAT) {
    // publish, read the beam's P pieces back, eliminate in natural order (every lane of the beam the same)
    pieces[lane] = make_piece<M, RZ>(st, acc.bits);
    __asm__ volatile("" ::: "memory");
    Vec2 u[P];
    iface_thomas<P>(FatIface{pieces + g * P}, u, bad);
    uL = u[0];
#pragma unroll
    for (int k = 1; k < P; ++k) {
      if (j == k) uL = u[k];
      if (j + 1 == k) uR = u[k];
    }
  } else {
    using X = Xch<P>;
    IfaceRow row;
    {
      const Mat2 cup = masked_cup<M, RZ>(st, acc.bits);
      const Sym2 pc = X::template from_minus<1>(st.Scc, (int)lane, j);
      const Vec2 pg = X::template from_minus<1>(st.gc, (int)lane, j);
      const Mat2 pb = X::template from_minus<1>(cup, (int)lane, j);
      row = make_row<M, RZ>(st, cup, pc, pg, pb, acc.bits);
    }
    cr_forward<P, 1>(row, (int)lane, j, bad);
    const Sym2 G = inv_spd(row.D, bad);
    uL = mul(G, row.f);
    if (j != 0) uL = Vec2{0.0, 0.0};
    cr_backward<P, cr_top_level(P)>(row, G, uL, (int)lane, j);
    uR = X::template from_plus<1>(uL, (int)lane, j);
  }
  __asm__ volatile("" ::: "memory");
  FAT_STAMP(7);
  seg_solve_pf<M, RZ, PF>(st, acc, uL, uR, out, hs);
}

// bits [e0, e0 + 32) of a 128-bit node mask held in two wave-uniform 64-bit halves
__device__ __forceinline__ unsigned mask_window(unsigned long long lo, unsigned long long hi, int e0) {
  const unsigned w0 = (unsigned)lo, w1 = (unsigned)(lo >> 32), w2 = (unsigned)hi, w3 = (unsigned)(hi >> 32);
  const int q = e0 >> 5, r = e0 & 31;
  const unsigned a = q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : w3;
  const unsigned b = q == 0 ? w1 : q == 1 ? w2 : q == 2 ? w3 : 0u;
  return (unsigned)((((unsigned long long)b << 32) | a) >> r);
}

// FAT  (P = 6): four row sets, one wave per SIMD.      LEAN (P = 16 / 8): two row sets, WPS waves per SIMD.
// SIZING (LEAN only): the inertias are the optimiser's float32 rows; after the solve the wave runs the optimiser epoch of
// its cases on the shears / moments it holds in LDS (sizing_math.hpp) instead of storing them: the generator's fused
// solve + step (SingleCore.py:174-219), as beam_solve.hip's beam_sizing_epoch_kernel.
template <int P, int M, int WPS, bool FAT, bool SIZING>
__device__ __forceinline__ void beam_rows_body(const BeamParams& p, const SizingArgs* sz) {
  constexpr int BPW = 64 / P;       // beams per wavefront
  constexpr int LIVE = BPW * P;     // lanes that own a segment
  constexpr int PM = P * M;         // padded nodes per beam (>= N); a row is one wave instruction of 16-byte lanes
  constexpr int NT = (PM + 63) / 64;
  constexpr int ROWS = BPW * PM;
  static_assert(PM <= 128 && PM % 2 == 0 && M + 1 <= 32, "row tiling limits");
  static_assert(FAT || (16 % P == 0 && LIVE == 64), "the lean mapping exchanges over DPP rows");
  constexpr int PF = FAT ? 3 : 1;   // prefetch distance of the interior solve (a lone wave hides nothing behind a partner)
#ifdef OPS_AMD_ST
  constexpr int ST = OPS_AMD_ST;
#else
  constexpr int ST = 16;            // sc1 (write-through); p.stream_out selects nt
#endif
  __shared__ double s_tab[6 * PM];
  __shared__ __attribute__((aligned(16))) double s_a[ROWS];   // I                  -> theta
  __shared__ __attribute__((aligned(16))) double s_b[ROWS];   // Fy                 -> V
  __shared__ __attribute__((aligned(16))) double s_m[FAT ? ROWS : 2];   // FAT: pieces, h.x of the sweep -> M   (LEAN: M -> s_a)
  __shared__ __attribute__((aligned(16))) double s_v[FAT ? ROWS : 2];   // FAT: h.y of the sweep         -> v   (LEAN: v -> s_b)
  __shared__ double s_dummy[M];                               // what the idle lanes write to
  // FAT: more than 32 KB and at most 40 KB per one-wave workgroup = four, never five, workgroups per CU
  static_assert((6 * PM + (FAT ? 4 : 2) * ROWS + M + 4) * 8 <= 160 * 1024 / (4 * WPS), "LDS per one-wave workgroup");
  static_assert(!FAT || (WPS == 1 && (6 * PM + 4 * ROWS) * 8 > 32 * 1024), "the fat tiling relies on LDS to keep a fifth wave off the CU");

  const unsigned lane = threadIdx.x;
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // diagnostic builds (-DOPS_AMD_TRACE) only
  FAT_STAMP(0);
  const int Ne = p.Ne, N = p.Ne + 1;
  const long beam0 = (long)blockIdx.x * BPW;
  const int nb = (p.B - beam0 < BPW) ? (int)(p.B - beam0) : BPW;   // live beams of this wave
  if (p.active) {                   // wave-uniform: finished cases of a sizing run cost one scalar load each
    unsigned any = 0;
    for (int b = 0; b < nb; ++b) any |= p.active[beam0 + b];
    if (!any) return;
  }

  // ---- stage 1a: every global load is issued before anything waits; cache-resident ones first ----
  double tx0[NT], tx1[NT];
  const double tE = p.E[0], tw = p.wy[0];
  {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned)N * 8u);
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const unsigned e = lane + 64u * k;
      tx0[k] = buf_load_d(rx, e * 8u);            // out of range (padding elements) -> 0
      tx1[k] = buf_load_d(rx, e * 8u + 8u);
    }
  }
  const int g_raw = lane / P, j = lane - g_raw * P, e0 = j * M;
  const bool live = (int)lane < LIVE;
  const int g = live ? g_raw : BPW - 1;          // idle lanes shadow the last group's low lanes; they write to s_dummy
  // constraint bytes (one mask for all beams, host-checked): nodes `lane` and `lane + 64`; the wave ballots them
  unsigned char fb0, fb1;
  {
    const __amdgpu_buffer_rsrc_t rf = make_rsrc(p.fix, (unsigned)N);
    fb0 = __builtin_amdgcn_raw_buffer_load_b8(rf, (int)lane, 0, 0);            // nodes >= N: out of range -> 0 = free
    fb1 = __builtin_amdgcn_raw_buffer_load_b8(rf, (int)(lane + 64u), 0, 0);
  }
  // the wave's rows: lane l moves doubles 2l, 2l + 1 of every row (16 bytes), the last double of an odd row apart
  const int hE = Ne >> 1, hN = N >> 1;
  double2 rI[BPW], rF[BPW];
  double tI = 0.0, tF = 0.0;
  {
    const __amdgpu_buffer_rsrc_t rsF = make_rsrc(p.Fy + beam0 * p.Fy_bs, (unsigned)(((long)(nb - 1) * p.Fy_bs + N) * 8));
    const unsigned oobF = (int)lane < hN ? lane * 16u : 0x40000000u;
    const unsigned tb = lane < (unsigned)nb ? lane : 0u;               // lane b fetches the odd tail of row b
    if constexpr (SIZING) {          // float32 inertias (the reference's I_tensor, dense rows): 8-byte pairs, widened in registers
      const __amdgpu_buffer_rsrc_t rsI = make_rsrc(p.I32 + beam0 * Ne, (unsigned)(nb * Ne) * 4u);
      const unsigned oobI = (int)lane < hE ? lane * 8u : 0x40000000u;
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        const float2 f = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsI, (int)((unsigned)(b * Ne) * 4u + oobI), 0, 0));
        rI[b] = make_double2((double)f.x, (double)f.y);
        rF[b] = buf_load_d2(rsF, (unsigned)(b * p.Fy_bs * 8) + oobF);
      }
      if (Ne & 1) tI = (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsI, (int)((tb * Ne + Ne - 1) * 4u), 0, 0));
    } else {
      const __amdgpu_buffer_rsrc_t rsI = make_rsrc(p.I + beam0 * p.I_bs, (unsigned)(((long)(nb - 1) * p.I_bs + Ne) * 8));
      const unsigned oobI = (int)lane < hE ? lane * 16u : 0x40000000u;
#pragma unroll
      for (int b = 0; b < BPW; ++b) {
        rI[b] = buf_load_d2(rsI, (unsigned)(b * p.I_bs * 8) + oobI);     // rows of beams beyond B, lanes beyond the row: 0
        rF[b] = buf_load_d2(rsF, (unsigned)(b * p.Fy_bs * 8) + oobF);
      }
      if (Ne & 1) tI = buf_load_d(rsI, (unsigned)((tb * p.I_bs + Ne - 1) * 8));
    }
    if (N & 1) tF = buf_load_d(rsF, (unsigned)((tb * p.Fy_bs + N - 1) * 8));
  }
  FAT_STAMP(5);
  // ---- stage 0: element table (unit-inertia stiffness tile entries, 1/L, UDL loads); padding as in beam_solve.hip ----
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const unsigned e = lane + 64u * k;
    if (e < (unsigned)PM) {
      const bool real = (int)e < Ne, pad = (int)e > Ne;
      const double L = tx1[k] - tx0[k];
      const double rl0 = fast_rcp(real ? L : 1.0);
      const double c2r = 2.0 * tE * rl0, pwr = 0.5 * tw * L;
      const double rl = real ? rl0 : (pad ? 1.0 : 0.0);
      const double c2 = real ? c2r : (pad ? 2.0 : 0.0);
      const double c6 = 3.0 * c2 * rl, c12 = 2.0 * c6 * rl;
      const double pw = real ? pwr : 0.0, mw = pw * L * (1.0 / 6.0);
      s_tab[0 * PM + e] = c2;  s_tab[1 * PM + e] = c6;  s_tab[2 * PM + e] = c12;
      s_tab[3 * PM + e] = rl;  s_tab[4 * PM + e] = pw;  s_tab[5 * PM + e] = mw;
    }
  }
  // ---- constraint flags of the lane's M + 1 nodes ----
  FixPair bits;
  bool any_rz;
  {
    const unsigned long long v0 = __ballot(fb0 & 1), v1 = __ballot(fb1 & 1);
    const unsigned long long r0 = __ballot(fb0 & 2), r1 = __ballot(fb1 & 2);
    bits.v = mask_window(v0, v1, e0) & ((2u << M) - 1u);
    bits.t = mask_window(r0, r1, e0) & ((2u << M) - 1u);
    any_rz = (r0 | r1) != 0ull;
  }

  // ---- stage 1b: the rows into LDS.  Padding: I = 1 (unit elements of the padding chain, beams beyond B), Fy = 0
  //      (what the out-of-range lanes of the row loads returned).  LDS operations of a wave execute in order.
  if (lane < M) s_dummy[lane] = 0.0;
#pragma unroll
  for (int k = 0; k < (ROWS / 2 + 63) / 64; ++k) {
    const unsigned idx = lane + 64u * k;
    if (k + 1 < (ROWS / 2 + 63) / 64 || idx < ROWS / 2) *reinterpret_cast<double2*>(&s_a[2 * idx]) = make_double2(1.0, 1.0);
  }
#pragma unroll
  for (int b = 0; b < BPW; ++b) {
    if ((int)lane < (b < nb ? hE : 0)) *reinterpret_cast<double2*>(&s_a[b * PM + 2 * lane]) = rI[b];
    if (lane < PM / 2) *reinterpret_cast<double2*>(&s_b[b * PM + 2 * lane]) = rF[b];
  }
  if ((Ne & 1) && (int)lane < nb) s_a[lane * PM + Ne - 1] = tI;
  if ((N & 1) && (int)lane < nb) s_b[lane * PM + N - 1] = tF;
  __syncthreads();
  FAT_STAMP(1);

  // ---- stages 2-4 ----
  FatAcc<PM> acc;
  acc.tab = &s_tab[e0];
  acc.sI = &s_a[g * PM + e0];
  acc.sF = &s_b[g * PM + e0];
  acc.bits = bits;
  int bad = 0;
  RowOut<M, FAT> out;
  out.sV = live ? &s_b[g * PM + e0] : s_dummy;
  out.sM = live ? (FAT ? &s_m[g * PM + e0] : &s_a[g * PM + e0]) : s_dummy;
  out.sv = (FAT && live) ? &s_v[g * PM + e0] : s_dummy;
  ParkH<M, FAT> hs;
  hs.hx = out.sM; hs.hy = out.sv;
  static_assert(!FAT || 64 * sizeof(IfacePiece) <= ROWS * 8, "the interface pieces borrow the moment rows");
  IfacePiece* const pieces = reinterpret_cast<IfacePiece*>(s_m);   // FAT: free until the interior solve parks its h_i there
  if (any_rz) row_solve_lanes<P, M, true, FAT, PF>(acc, g, j, bad, out, hs, pieces, lane, stamps);
  else        row_solve_lanes<P, M, false, FAT, PF>(acc, g, j, bad, out, hs, pieces, lane, stamps);
  FAT_STAMP(2);

  // a beam is bad if any of its P lanes met a non-positive pivot; its outputs become NaN
  const unsigned long long bal = __ballot(bad != 0 && live);
  const unsigned long long grp = (P == 64 ? ~0ull : ((1ull << (P % 64)) - 1ull)) << (g * P);
  const bool gbad = (bal & grp) != 0ull;
  if (bal != 0ull && gbad && live) {              // first test wave-uniform: nothing to do in the common case
    const double qnan = __builtin_nan("");
#pragma unroll
    for (int i = 0; i < M; ++i) {
      out.sV[i] = qnan; out.sM[i] = qnan; out.th[i] = qnan;
      if constexpr (FAT) out.sv[i] = qnan; else out.vr[i] = qnan;
    }
  }
  if (j == 0 && live && g < nb && p.status)
    __hip_atomic_store(&p.status[beam0 + g], gbad ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double* const sth = live ? &s_a[g * PM + e0] : s_dummy;
  if constexpr (FAT) {                            // rotations: from registers into the dead inertia rows
#pragma unroll
    for (int i = 0; i < M; ++i) sth[i] = out.th[i];
  }
  wave_lds_fence();
  if constexpr (SIZING) {           // V in s_b, M in s_a (padded rows): one optimiser epoch per live, active case
    static_assert(!SIZING || !FAT, "the fused epoch uses the lean mapping");
    // every case's state loads go out before the first one is used (a wave holds BPW cases: one HBM latency, not BPW)
    CaseRegs<2> cr[BPW];            // host-checked: Ne <= 128
    bool on[BPW];
#pragma unroll
    for (int gb = 0; gb < BPW; ++gb) {
      on[gb] = gb < nb && sz->active[beam0 + gb] != 0;     // wave-uniform
      if (on[gb]) load_case<2>((int)lane, beam0 + gb, Ne, *sz, cr[gb]);
    }
#pragma unroll
    for (int gb = 0; gb < BPW; ++gb) {
      if (!on[gb]) continue;
      const double* Vb = &s_b[gb * PM];
      const double* Mb = &s_a[gb * PM];
      step_case<2>((int)lane, beam0 + gb, Ne, *sz, cr[gb], [&](int e) { return (float)Vb[e]; }, [&](int e) { return (float)Mb[e]; });
    }
    return;
  }

  // ---- stage 5: rows out, one 16-byte store per lane and row.  All rows of an array are read from LDS before the
  //      first of them is stored (a store behind every read would pay one LDS round trip per row), and the next
  //      array's reads are issued before this array's stores.
  const bool two = p.v != nullptr;                // wave-uniform: forces-only calls store two arrays
  auto read_rows = [&](double2 (&r)[BPW], double& tail, const double* s, int n_) {
    const unsigned lrow = (int)lane < (n_ >> 1) ? lane : 0u;
#pragma unroll
    for (int b = 0; b < BPW; ++b) r[b] = *reinterpret_cast<const double2*>(&s[b * PM + 2 * lrow]);
    tail = s[((int)lane < nb ? lane : 0u) * PM + n_ - 1];
  };
  auto store_rows = [&](auto aux, double* dst, const double2 (&r)[BPW], double tail, int n_) {
    constexpr int AUX = decltype(aux)::value;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(dst + beam0 * n_, (unsigned)(nb * n_) * 8u);
    const unsigned off = (int)lane < (n_ >> 1) ? lane * 16u : 0x40000000u;     // lanes beyond the row: out of range, dropped
#pragma unroll
    for (int b = 0; b < BPW; ++b)                 // rows of beams beyond B are out of range as a whole
      buf_store_d2<AUX>(rs, (unsigned)(b * n_) * 8u + off, r[b]);
    if ((n_ & 1) && (int)lane < nb) buf_store_d<AUX>(rs, (unsigned)(lane * n_ + n_ - 1) * 8u, tail);
  };
  auto all_rows = [&](auto aux) {
    double2 ra[BPW], rb[BPW];
    double ta, tb;
    read_rows(ra, ta, s_b, Ne);
    read_rows(rb, tb, FAT ? s_m : s_a, Ne);
    if constexpr (FAT) {
      store_rows(aux, p.V, ra, ta, Ne);
      if (two) read_rows(ra, ta, s_v, N);
      store_rows(aux, p.M, rb, tb, Ne);
      if (two) {
        read_rows(rb, tb, s_a, N);
        store_rows(aux, p.v, ra, ta, N);
        store_rows(aux, p.theta, rb, tb, N);
      }
    } else {
      if (two) {                                  // the row sets are in registers now: deflections / rotations take their place
        wave_lds_fence();
        double* const svv = live ? &s_b[g * PM + e0] : s_dummy;
#pragma unroll
        for (int i = 0; i < M; ++i) { svv[i] = out.vr[i]; sth[i] = out.th[i]; }
      }
      store_rows(aux, p.V, ra, ta, Ne);
      store_rows(aux, p.M, rb, tb, Ne);
      if (two) {
        wave_lds_fence();
        read_rows(ra, ta, s_b, N);
        read_rows(rb, tb, s_a, N);
        store_rows(aux, p.v, ra, ta, N);
        store_rows(aux, p.theta, rb, tb, N);
      }
    }
  };
  if (p.stream_out) all_rows(std::integral_constant<int, 2>{});   // wave-uniform
  else              all_rows(std::integral_constant<int, ST>{});
#ifdef OPS_AMD_TRACE
  if (p.trace && lane == 0) {   // per-wave phase stamps (100 MHz clock) + hardware id, for scripts/trace_run.py
    FAT_STAMP(3);
    stamps[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((16 - 1) << 11));
    unsigned long long* q = p.trace + 8 * (unsigned long long)blockIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = stamps[k];
  }
#endif
}

template <int P, int M, int WPS, bool FAT>
__global__ __launch_bounds__(64, WPS) void beam_rows_kernel(const BeamParams p) {
  beam_rows_body<P, M, WPS, FAT, false>(p, nullptr);
}
template <int P, int M, int WPS>
__global__ __launch_bounds__(64, WPS) void beam_rows_sizing_kernel(const BeamParams p, const SizingArgs sz) {
  beam_rows_body<P, M, WPS, false, true>(p, &sz);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
const FatTiling kFatTilings[] = {
    {6, 17, "beam_rows_kernel<6, 17, 1, true>"},
    {16, 7, "beam_rows_kernel<16, 7, 3, false>"},
    {8, 13, "beam_rows_kernel<8, 13, 2, false>"},
};
const int kNumFatTilings = sizeof(kFatTilings) / sizeof(kFatTilings[0]);

hipError_t launch_fat_sizing(const BeamParams& p, const SizingArgs& sz, int P, int M, hipStream_t stream) {
  const unsigned grid = (unsigned)((p.B + 64 / P - 1) / (64 / P));
  if (P == 16 && M == 7) hipLaunchKernelGGL((beam_rows_sizing_kernel<16, 7, 3>), dim3(grid), dim3(64), 0, stream, p, sz);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_fat(const BeamParams& p, int P, int M, hipStream_t stream) {
  const int bpw = 64 / P;
  const unsigned grid = (unsigned)((p.B + bpw - 1) / bpw);
  if (P == 6 && M == 17) hipLaunchKernelGGL((beam_rows_kernel<6, 17, 1, true>), dim3(grid), dim3(64), 0, stream, p);
  else if (P == 16 && M == 7) hipLaunchKernelGGL((beam_rows_kernel<16, 7, 3, false>), dim3(grid), dim3(64), 0, stream, p);
  else if (P == 8 && M == 13) hipLaunchKernelGGL((beam_rows_kernel<8, 13, 2, false>), dim3(grid), dim3(64), 0, stream, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace opsamd
