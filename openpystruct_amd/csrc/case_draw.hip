// Case randomisation of the dataset generator as ONE launch.
//
// What it restates: /root/reference/OpenPyStruct_BeamOpt_training_SingleCore.py:133-160 (the draws `generate_sample` makes
// per sample: with random_bridge = 1 the length L = L_min + U(0, 1) L_max and 1 .. N_rollers_max distinct rollers among
// nodes 2 .. N - 1; then 1 .. M_forces_max distinct loaded nodes among the candidates that are not rollers, each load
// U(max_force, min_force)) and :100-113 (`ops.fix`, `ops.load`: the support mask and the nodal load vector of the case).
// The reference never seeds `random`; here case i is a pure function of (seed, i): a counter-based stream (the keyed 32-bit
// hash of csrc/dropout_stream.hpp, key = (seed, case number), counter = draw number), so a shard, a chunk or a rank draws
// its range of the global list without drawing the rest.
//
// Why a kernel: the vectorised torch form of the same draws (sizing._make_case_block: argsort-based distinct picks, ~25
// framework ops per block of 16 384 cases) costs 1.9 ms of host-bound launches per 50 000 cases -- 8 % of a 23.6 ms generator
// shard (scripts/generator_breakdown.py).  One wavefront per case: the draws are wave-uniform scalar work, the lanes write
// the case's rows (support mask, load vector) coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/openpystruct_amd.h"
#include "dropout_stream.hpp"

namespace opsamd {

struct CaseDrawArgs {
  long B;
  unsigned long long first_case, seed;
  int N, R, F, random_bridge, n_fixed;
  int fixed[OPS_CASE_MAX_PICKS];
  double L_min, L_max, max_force, min_force;
  double* Ls; long long* r_nodes; long long* nr; long long* f_nodes; long long* kf; double* f_vals; uint8_t* fix; double* Fy;
};

__device__ __forceinline__ uint32_t cd_hash(DropKey k, uint32_t idx) {      // the integer stage of drop_uniform
  uint32_t x = idx ^ k.k0;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= k.k1;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
struct CaseStream {
  DropKey key; uint32_t ctr;
  __device__ uint32_t bits() { return cd_hash(key, ctr++); }
  __device__ int below(int n) { return (int)(((uint64_t)bits() * (uint64_t)n) >> 32); }                  // 0 .. n - 1
  __device__ double uniform() {                                                                        // [0, 1), 53 bits
    const uint64_t hi = bits(), lo = bits();
    return (double)(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
  }
};

__global__ __launch_bounds__(64) void case_draw_kernel(const CaseDrawArgs a) {
  const long b = blockIdx.x;
  if (b >= a.B) return;
  const int lane = threadIdx.x, N = a.N, ncand = N - 2;                     // candidates: nodes 2 .. N - 1 (1-based), SC:63 / :138
  CaseStream rs{drop_key(a.seed, a.first_case + (unsigned long long)b), 0u};
  constexpr int MP = OPS_CASE_MAX_PICKS;
  int r[MP], f[MP];                                                         // (every loop over them is unrolled to MP: registers)
  double fv[MP];
#pragma unroll
  for (int q = 0; q < MP; ++q) { r[q] = 0; f[q] = 0; fv[q] = 0.0; }
  auto among = [&](const int (&set)[MP], int upto, int node) {              // node among set[0 .. upto - 1]?
    bool t = false;
#pragma unroll
    for (int q = 0; q < MP; ++q) t |= q < upto && set[q] == node;
    return t;
  };
  int nr, R;
  double L;
  if (a.random_bridge == 1) {
    L = a.L_min + rs.uniform() * a.L_max;                                   // SC:134
    R = a.R;
    nr = 1 + rs.below(R);                                                   // SC:139
#pragma unroll
    for (int s = 0; s < MP; ++s) {
      if (s < nr) {                                                         // distinct picks (random.sample, SC:142-151)
        int c = 2 + rs.below(ncand);
        for (int tries = 0; among(r, s, c); ++tries) {
          c = 2 + rs.below(ncand);
          if (tries > 64) { c = 2; while (among(r, s, c)) ++c; }            // (never in practice: at most 8 of the candidates are taken)
        }
        r[s] = c;
      }
    }
  } else {
    L = a.L_max;
    R = a.n_fixed;
    nr = R;
#pragma unroll
    for (int s = 0; s < MP; ++s) r[s] = s < R ? a.fixed[s] : 0;             // SC:153
  }
  int n_avail = ncand;
#pragma unroll
  for (int q = 0; q < MP; ++q) n_avail -= (q < R && r[q] >= 2 && r[q] < N) ? 1 : 0;
  int k = 1 + rs.below(a.F);                                                // SC:157-158
  k = k < n_avail ? k : n_avail;
#pragma unroll
  for (int s = 0; s < MP; ++s) {
    if (s < k) {
      int c = 2 + rs.below(ncand);
      for (int tries = 0; among(r, R, c) || among(f, s, c); ++tries) {      // not a roller, not drawn before (SC:157-159)
        c = 2 + rs.below(ncand);
        if (tries > 256) { c = 2; while (among(r, R, c) || among(f, s, c)) ++c; }
      }
      f[s] = c;
      fv[s] = a.max_force + rs.uniform() * (a.min_force - a.max_force);     // SC:160
    }
  }
  if (lane == 0) { a.Ls[b] = L; a.nr[b] = nr; a.kf[b] = k; }
#pragma unroll
  for (int s = 0; s < MP; ++s) {                                            // lane s writes slot s (no dynamic register index)
    if (lane == s && s < R) a.r_nodes[b * R + s] = r[s];
    if (lane == s && s < a.F) { a.f_nodes[b * a.F + s] = f[s]; a.f_vals[b * a.F + s] = fv[s]; }
  }
  for (int n = lane; n < N; n += 64) {                                      // rows: ops.fix (SC:100-102), ops.load (SC:113)
    const int node = n + 1;
    a.fix[b * N + n] = (uint8_t)((n == 0 || among(r, R, node)) ? 1 : 0);
    double load = 0.0;
#pragma unroll
    for (int q = 0; q < MP; ++q) load += (f[q] == node) ? fv[q] : 0.0;      // unused slots hold node 0
    a.Fy[b * N + n] = load;
  }
}

}  // namespace opsamd

using namespace opsamd;

extern "C" int ops_sizing_draw_cases_f64(long B, unsigned long long first_case, unsigned long long seed, int num_nodes, int n_rollers_max,
                                         int m_forces_max, int random_bridge, const int32_t* fixed_rollers, int n_fixed, double L_min,
                                         double L_max, double max_force, double min_force, double* L, long long* roller_nodes,
                                         long long* n_rollers, long long* force_nodes, long long* n_forces, double* force_values,
                                         uint8_t* fix, double* Fy, void* stream) {
  if (B < 0 || num_nodes < 4 || m_forces_max < 1 || m_forces_max > OPS_CASE_MAX_PICKS) return OPS_AMD_ERR_INVALID_ARG;
  if (random_bridge == 1 ? (n_rollers_max < 1 || n_rollers_max > OPS_CASE_MAX_PICKS || n_rollers_max > num_nodes - 2)
                         : (n_fixed < 0 || n_fixed > OPS_CASE_MAX_PICKS || (n_fixed > 0 && !fixed_rollers)))
    return OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!L || !roller_nodes || !n_rollers || !force_nodes || !n_forces || !force_values || !fix || !Fy) return OPS_AMD_ERR_INVALID_ARG;
  if (B > 0x7FFFFFFFl) return OPS_AMD_ERR_UNSUPPORTED;
  CaseDrawArgs a{};
  a.B = B; a.first_case = first_case; a.seed = seed;
  a.N = num_nodes; a.R = n_rollers_max; a.F = m_forces_max; a.random_bridge = random_bridge; a.n_fixed = random_bridge == 1 ? 0 : n_fixed;
  for (int i = 0; i < a.n_fixed; ++i) a.fixed[i] = fixed_rollers[i];
  a.L_min = L_min; a.L_max = L_max; a.max_force = max_force; a.min_force = min_force;
  a.Ls = L; a.r_nodes = roller_nodes; a.nr = n_rollers; a.f_nodes = force_nodes; a.kf = n_forces; a.f_vals = force_values; a.fix = fix; a.Fy = Fy;
  hipLaunchKernelGGL(case_draw_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
