// Shared pieces of the beam-solve kernels (beam_solve.hip: the P | 64 tilings with DPP exchange;
// beam_fat.hip: the "fat wave" tilings): launch parameters, buffer-resource I/O, cross-lane exchange,
// the cyclic-reduction drivers over beam_math.hpp, the wave-local LDS fence.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "beam_math.hpp"
#include "sizing_math.hpp"

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
  const float* I32;            // sizing epochs: the inertias are float32 rows (dense, stride Ne), widened while staging; I unused
  const uint8_t* active;       // optional [B]: a wave whose beams are all inactive returns at once (sizing epochs)
  int stream_out;              // OPS_AMD_TILING_STREAM_OUT: non-temporal output stores (buffers that will not be re-read from cache)
  int f32_forces;              // V / M point to float rows (the sizing loop rounds them to float32 anyway, SingleCore.py:189-190)
  // host-derived: rows of I/Fy/outputs are dense and every wave's chunk is 16-byte aligned, so
  // the wave moves its beams as one flat run of 16-byte accesses; magic numbers for idx / Ne, idx / N
  int dense;
  unsigned long long* trace;   // diagnostic builds (-DOPS_AMD_TRACE) only
  unsigned magic_ne, magic_n;
};

// fat-wave tilings (beam_fat.hip): shared geometry and constraint mask only; a tiling serves Ne with Ne + 1 <= P * M
struct FatTiling { int P, M; const char* name; };
extern const FatTiling kFatTilings[];
extern const int kNumFatTilings;
hipError_t launch_fat(const BeamParams& p, int P, int M, hipStream_t stream);
hipError_t launch_fat_sizing(const BeamParams& p, const SizingArgs& sz, int P, int M, hipStream_t stream);   // P = 16 only

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

// The same fetches through the LDS crossbar (ds_bpermute_b32), for groups that do not sit inside one DPP row
// (P not a divisor of 16: the fat tilings of beam_fat.hip).  A lane whose neighbour at distance S does not exist
// fetches from ITSELF instead of receiving 0: every use of such a value is multiplied by a coupling that is
// exactly zero (beam_math.hpp, cr_eliminate), so any finite value of the lane's OWN beam will do -- and nothing of
// another beam can leak in.  No VALU instruction per exchanged value (a DPP fetch costs two v_mov_b32 per double).
template <int P>
struct XchPerm {
  static __device__ __forceinline__ double fetch(double x, int src_lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)u);
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  }
  template <int S> static __device__ __forceinline__ double from_minus(double x, int lane, int j) { return fetch(x, j >= S ? lane - S : lane); }
  template <int S> static __device__ __forceinline__ double from_plus(double x, int lane, int j) { return fetch(x, j + S < P ? lane + S : lane); }
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

// cyclic reduction over the P rows of a beam (beam_math.hpp): levels S = 1, 2, 4, ... < P.  Every lane runs
// the exchange (a DPP / bpermute fetch must not sit inside a divergent region: disabled source lanes read as 0);
// only the rows that are active at the level apply the update -- exec-masked, the others keep their frozen row.
// EARLY (crossbar exchange): everything that does not depend on the level's pivot inverse is fetched BEFORE the
// inverse is computed, so that the crossbar latency runs under its reciprocal chain.
template <int P, int S, class X = Xch<P>, bool EARLY = false>
__device__ __forceinline__ void cr_forward(IfaceRow& row, int lane, int j, int& bad) {
  if constexpr (S < P) {
    constexpr bool LAST = (2 * S >= P);
    const bool act = cr_active(j, S);
    if constexpr (EARLY) {
      const Vec2 fm = X::template from_minus<S>(row.f, lane, j);
      const Vec2 fp = X::template from_plus<S>(row.f, lane, j);
      Mat2 Am{0, 0, 0, 0}, Cp{0, 0, 0, 0};
      if constexpr (!LAST) {
        Am = X::template from_minus<S>(row.Alow, lane, j);
        Cp = X::template from_plus<S>(row.Cup, lane, j);
      }
      const Sym2 G = inv_spd(row.D, bad);
      const Sym2 Gm = X::template from_minus<S>(G, lane, j);
      const Sym2 Gp = X::template from_plus<S>(G, lane, j);
      if (act) {
        cr_absorb<LAST>(row.D, row.f, row.Alow, Gm, Am, fm);
        cr_absorb<LAST>(row.D, row.f, row.Cup, Gp, Cp, fp);
      }
    } else {
      const Sym2 G = inv_spd(row.D, bad);
      // the two sides one after the other: half the exchange registers live at a time
      const Sym2 Gm = X::template from_minus<S>(G, lane, j);
      const Vec2 fm = X::template from_minus<S>(row.f, lane, j);
      Mat2 Am{0, 0, 0, 0};
      if constexpr (!LAST) Am = X::template from_minus<S>(row.Alow, lane, j);
      const Vec2 fp = X::template from_plus<S>(row.f, lane, j);      // fetched before the minus side rewrites row.f
      if (act) cr_absorb<LAST>(row.D, row.f, row.Alow, Gm, Am, fm);
      const Sym2 Gp = X::template from_plus<S>(G, lane, j);
      Mat2 Cp{0, 0, 0, 0};
      if constexpr (!LAST) Cp = X::template from_plus<S>(row.Cup, lane, j);
      if (act) cr_absorb<LAST>(row.D, row.f, row.Cup, Gp, Cp, fp);
    }
    cr_forward<P, 2 * S, X, EARLY>(row, lane, j, bad);
  }
}
// back substitution from the top level down: the rows frozen at level S take their neighbours' displacements
template <int P, int S, class X = Xch<P>>
__device__ __forceinline__ void cr_backward(const IfaceRow& row, const Sym2& G, Vec2& u, int lane, int j) {
  if constexpr (S >= 1) {
    const Vec2 um = X::template from_minus<S>(u, lane, j);
    const Vec2 up = X::template from_plus<S>(u, lane, j);
    if (cr_frozen(j, S)) u = cr_back(row, G, um, up);
    cr_backward<P, S / 2, X>(row, G, u, lane, j);
  }
}
// the top reduction level of a P-row interface system: the largest power of two below P
constexpr int cr_top_level(int P) { int s = 1; while (2 * s < P) s *= 2; return s; }

// Orders LDS traffic inside ONE wavefront (the workgroup is a single wave): LDS instructions of a wave
// execute in order, so all that is needed is to stop the compiler from moving LDS accesses across this
// point and to have earlier LDS reads landed in registers.  Unlike __syncthreads() it does not wait for
// outstanding global stores (vmcnt), which would serialise the store phases behind HBM write latency.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) only
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ---- buffer-resource I/O: hardware bounds checking instead of tail branches --------------------
// A raw buffer descriptor (base, num_records in bytes) makes out-of-range lanes of a buffer_load return 0
// and out-of-range lanes of a buffer_store do nothing; offsets are 32-bit, the base sits in SGPRs.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double2 buf_load_d2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
#ifndef OPS_AMD_LD_AUX
#define OPS_AMD_LD_AUX 0
#endif
  const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, OPS_AMD_LD_AUX);   // row loads (A/B: -DOPS_AMD_LD_AUX=2 nt)
  return __builtin_bit_cast(double2, v);
}
__device__ __forceinline__ double buf_load_d(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(double, v);
}
// Output stores carry a cache policy (the `aux` immediate: 16 = sc1 write-through, 2 = nt).  Plain stores
// leave up to an L2's worth of dirty lines behind that the end-of-kernel release has to write back; measured
// per 10^4-beam launch (A/B, same device, profiles/r01_notes.md): plain 16.4 us, sc1 14.35, sc0+sc1 14.4,
// nt 15.1; at 2^20 beams nt is the best (969 vs 983 sc1 vs 990 us plain).
template <int AUX>
__device__ __forceinline__ void buf_store_d2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double2 x) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, x), r, (int)byte_off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void buf_store_d(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double x) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, x), r, (int)byte_off, 0, AUX);
}

}  // namespace opsamd
