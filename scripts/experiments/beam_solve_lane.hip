// MEASURED ALTERNATIVE, not the product path: one LANE per beam (hard part 1-iii; the layout of the
// vendor gtsvInterleavedBatch routines).  A wavefront holds 64 beams; every lane runs the sequential block-Thomas
// elimination of its own beam (2x2 blocks, forward sweep storing the pivot inverses and the reduced right-hand sides,
// backward sweep, end-force recovery).  The row-major [B, Ne] rows of the API are transposed through LDS in tiles of
// 16 columns so that global accesses stay coalesced (128-byte row pieces); the factor of a beam does not fit registers
// or LDS and goes through an HBM workspace laid out [node][5][B] (coalesced across lanes): +8 080 B per solve on top of
// the 4 925 algorithmic bytes.  Shared geometry / supports only (the bench configuration).  Same semantics as
// beam_solve_kernel (beam_math.hpp), checked against it in tests/test_gpu_parity.py; numbers in profiles/r01_notes.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"
#include "beam_math.hpp"

namespace opsamd {

constexpr int LT = 16;            // columns per transposition tile
constexpr int LTP = LT + 1;       // padded row of a tile (conflict-free column reads)

struct LaneParams {
  int B, Ne;
  const double* x; const double* E; const double* wy;   // shared: [N], scalar, scalar
  const uint8_t* fix;                                   // shared [N]
  const double* I; const double* Fy;
  double* v; double* theta; double* V; double* M;
  int32_t* status;
  double* ws;                                           // [N][5][Bp]
  long Bp;
};

// global rows [64 beams][ncol] starting at column c0 -> tile[64][LTP] (lane l moves beam 4k + l/16, column l%16)
__device__ __forceinline__ void tile_load(const double* __restrict__ g, long row0, int nrows, int stride, int c0, int ncols, double* tile, int lane) {
  const int cc = lane & 15, rr = lane >> 4;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = 4 * k + rr;
    double val = 0.0;
    if (r < nrows && c0 + cc < ncols) val = g[(row0 + r) * (long)stride + c0 + cc];
    tile[r * LTP + cc] = val;
  }
}
__device__ __forceinline__ void tile_store(double* __restrict__ g, long row0, int nrows, int stride, int c0, int ncols, const double* tile, int lane) {
  const int cc = lane & 15, rr = lane >> 4;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = 4 * k + rr;
    if (r < nrows && c0 + cc < ncols) g[(row0 + r) * (long)stride + c0 + cc] = tile[r * LTP + cc];
  }
}

__device__ __forceinline__ Sym2 masked_inverse(const Sym2& S, int fixbits, int& bad) {
  const bool fy = (fixbits & 1) == 0, fr = (fixbits & 2) == 0;
  if (fy && fr) return inv_spd(S, bad);
  if (fy) { bad |= !(S.a > 0.0); return Sym2{fast_rcp(S.a), 0.0, 0.0}; }
  if (fr) { bad |= !(S.c > 0.0); return Sym2{0.0, 0.0, fast_rcp(S.c)}; }
  return Sym2{0.0, 0.0, 0.0};
}

__global__ __launch_bounds__(64) void beam_solve_lane_kernel(const LaneParams p) {
  extern __shared__ double lds[];
  const int Ne = p.Ne, N = Ne + 1, lane = threadIdx.x;
  double* tab = lds;                       // [6][Ne]: c2, c6, c12, (unused), pw, mw
  double* tA = tab + 6 * Ne;               // tiles
  double* tB = tA + 64 * LTP;
  double* tC = tB + 64 * LTP;
  double* tD = tC + 64 * LTP;
  uint8_t* sfix = reinterpret_cast<uint8_t*>(tD + 64 * LTP);   // [N]
  const long beam0 = (long)blockIdx.x * 64;
  const int nb = (p.B - beam0 < 64) ? (int)(p.B - beam0) : 64;
  const long b = beam0 + lane;
  const double E = p.E[0], w = p.wy[0];
  for (int e = lane; e < Ne; e += 64) {
    const double L = p.x[e + 1] - p.x[e], rl = fast_rcp(L), c2 = 2.0 * E * rl, c6 = 3.0 * c2 * rl, c12 = 2.0 * c6 * rl;
    const double pw = 0.5 * w * L, mw = pw * L * (1.0 / 6.0);
    tab[e] = c2; tab[Ne + e] = c6; tab[2 * Ne + e] = c12; tab[4 * Ne + e] = pw; tab[5 * Ne + e] = mw;
  }
  for (int i = lane; i < N; i += 64) sfix[i] = p.fix[i];
  __syncthreads();

  // ---- forward sweep ----
  int bad = 0;
  Sym2 S{0.0, 0.0, 0.0};      // accumulated diagonal block of the current node
  Vec2 f{0.0, 0.0};           // accumulated right-hand side of the current node
  Sym2 Gp{0.0, 0.0, 0.0};     // pivot inverse of the previous node
  Vec2 fp{0.0, 0.0};          // reduced right-hand side of the previous node
  ElemK kp{0.0, 0.0, 0.0, 0.0};
  for (int c0 = 0; c0 < N; c0 += LT) {
    tile_load(p.I, beam0, nb, Ne, c0, Ne, tA, lane);
    tile_load(p.Fy, beam0, nb, N, c0, N, tB, lane);
    __syncthreads();
    for (int j = 0; j < LT && c0 + j < N; ++j) {
      const int i = c0 + j;
      // node i: left element i-1 (kp, already folded into S / f below), right element i
      S = Sym2{0.0, 0.0, 0.0};
      f = Vec2{tB[lane * LTP + j], 0.0};
      if (i > 0) {
        // k22 of element i-1 and its loads, then the Schur complement of node i-1: S -= k12^T Gp k12, f -= k12^T Gp fp
        const Mat2 k12{-kp.kA, kp.kB, -kp.kB, kp.kD};
        S = Sym2{kp.kA, -kp.kB, kp.kC};
        f.x += tab[4 * Ne + i - 1];
        f.y -= tab[5 * Ne + i - 1];
        const Mat2 Gk = Mat2{__builtin_fma(Gp.a, k12.a, Gp.b * k12.c), __builtin_fma(Gp.a, k12.b, Gp.b * k12.d),
                             __builtin_fma(Gp.b, k12.a, Gp.c * k12.c), __builtin_fma(Gp.b, k12.b, Gp.c * k12.d)};   // Gp k12
        S = Sym2{S.a - __builtin_fma(k12.a, Gk.a, k12.c * Gk.c), S.b - __builtin_fma(k12.a, Gk.b, k12.c * Gk.d),
                 S.c - __builtin_fma(k12.b, Gk.b, k12.d * Gk.d)};
        const Vec2 Gf = mul(Gp, fp);
        f = sub_mulT(f, k12, Gf);
      }
      ElemK kc{0.0, 0.0, 0.0, 0.0};
      if (i < Ne) {
        kc = elem_k(tab[i], tab[Ne + i], tab[2 * Ne + i], tA[lane * LTP + j]);
        S = Sym2{S.a + kc.kA, S.b + kc.kB, S.c + kc.kC};
        f.x += tab[4 * Ne + i];
        f.y += tab[5 * Ne + i];
      }
      const Sym2 G = masked_inverse(S, sfix[i], bad);
      if (lane < nb) {
        double* wsn = p.ws + ((long)i * 5) * p.Bp + b;
        wsn[0] = G.a; wsn[p.Bp] = G.b; wsn[2 * p.Bp] = G.c; wsn[3 * p.Bp] = f.x; wsn[4 * p.Bp] = f.y;
      }
      Gp = G; fp = f; kp = kc;
    }
    __syncthreads();
  }

  // ---- backward sweep + recovery: tiles of 16 nodes, descending ----
  Vec2 un{0.0, 0.0};          // displacement of node i+1
  const double qnan = __builtin_nan("");
  for (int c0 = ((N - 1) / LT) * LT; c0 >= 0; c0 -= LT) {
    tile_load(p.I, beam0, nb, Ne, c0, Ne, tA, lane);
    __syncthreads();
    for (int j = LT - 1; j >= 0; --j) {
      const int i = c0 + j;
      if (i >= N) continue;
      Sym2 G{0.0, 0.0, 0.0};
      Vec2 fi{0.0, 0.0};
      if (lane < nb) {
        const double* wsn = p.ws + ((long)i * 5) * p.Bp + b;
        G = Sym2{wsn[0], wsn[p.Bp], wsn[2 * p.Bp]};
        fi = Vec2{wsn[3 * p.Bp], wsn[4 * p.Bp]};
      }
      Vec2 ui;
      if (i < Ne) {
        const ElemK k = elem_k(tab[i], tab[Ne + i], tab[2 * Ne + i], tA[lane * LTP + j]);
        const Mat2 k12{-k.kA, k.kB, -k.kB, k.kD};
        ui = mul(G, sub_mul(fi, k12, un));
        // ElasticBeam2d resisting forces at end I: K u - consistent loads
        const double dv = ui.x - un.x;
        tC[lane * LTP + j] = __builtin_fma(k.kA, dv, k.kB * (ui.y + un.y)) - tab[4 * Ne + i];
        tD[lane * LTP + j] = __builtin_fma(k.kB, dv, __builtin_fma(k.kC, ui.y, k.kD * un.y)) - tab[5 * Ne + i];
      } else {
        ui = mul(G, fi);
      }
      tB[lane * LTP + j] = ui.x;      // v
      // theta goes to tA's slot of this column only after I was consumed above
      tA[lane * LTP + j] = ui.y;
      un = ui;
    }
    if (bad) {
#pragma unroll
      for (int j = 0; j < LT; ++j) { tA[lane * LTP + j] = qnan; tB[lane * LTP + j] = qnan; tC[lane * LTP + j] = qnan; tD[lane * LTP + j] = qnan; }
    }
    __syncthreads();
    tile_store(p.v, beam0, nb, N, c0, N, tB, lane);
    tile_store(p.theta, beam0, nb, N, c0, N, tA, lane);
    tile_store(p.V, beam0, nb, Ne, c0, Ne, tC, lane);
    tile_store(p.M, beam0, nb, Ne, c0, Ne, tD, lane);
    __syncthreads();
  }
  if (lane < nb && p.status) p.status[b] = bad ? 1 : 0;
}

}  // namespace opsamd

using namespace opsamd;

extern "C" size_t ops_beam_solve_lane_workspace_bytes(int B, int Ne) {
  const long Bp = ((long)B + 63) / 64 * 64;
  return (size_t)(Ne + 1) * 5 * (size_t)Bp * sizeof(double);
}

extern "C" int ops_beam_solve_lane_per_beam_f64(int B, int Ne, const double* x, const double* E, const double* I, const uint8_t* fix,
                                                const double* Fy, const double* wy, double* v, double* theta, double* V, double* M,
                                                int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || Ne < 1) return OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!x || !E || !I || !fix || !Fy || !wy || !v || !theta || !V || !M || !workspace) return OPS_AMD_ERR_INVALID_ARG;
  if (workspace_bytes < ops_beam_solve_lane_workspace_bytes(B, Ne)) return OPS_AMD_ERR_INVALID_ARG;
  const size_t lds = (size_t)(6 * Ne + 4 * 64 * LTP) * sizeof(double) + (size_t)(Ne + 1 + 15) / 16 * 16;
  if (lds > 64 * 1024) return OPS_AMD_ERR_UNSUPPORTED;
  const LaneParams p{B, Ne, x, E, wy, fix, I, Fy, v, theta, V, M, status, (double*)workspace, ((long)B + 63) / 64 * 64};
  hipLaunchKernelGGL(beam_solve_lane_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), lds, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
