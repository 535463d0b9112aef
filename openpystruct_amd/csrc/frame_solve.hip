// Batched 2-D elastic frame solve (3 DOF per node): assembly of rotated ElasticBeam2d stiffness matrices,
// banded LDL^T factorisation, substitution and global end-force recovery, one 1024-thread workgroup per frame with the
// band matrix resident in LDS.
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
// LDS: ab[n_eq][kd+1] (lower band, column j holds A[j..j+kd][j]) + rhs[n_eq]; n_eq * (kd + 2) * 8 B <= 160 KB.
// The factorisation is LDS-bandwidth / barrier bound (n_eq * kd^2 / 2 FMAs, one barrier per column), not HBM
// bound: ~15 KB of HBM traffic per frame.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

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

__device__ __forceinline__ void elem_global_k(double L, double c, double s, double EA, double EI, double k[6][6]) {
  const double a = EA / L, b12 = 12.0 * EI / (L * L * L), b6 = 6.0 * EI / (L * L), b4 = 4.0 * EI / L, b2 = 2.0 * EI / L;
  const double kxx = a * c * c + b12 * s * s, kxy = (a - b12) * c * s, kyy = a * s * s + b12 * c * c;
  const double kxt = -b6 * s, kyt = b6 * c;
  const double v[6][6] = {{kxx, kxy, kxt, -kxx, -kxy, kxt},  {kxy, kyy, kyt, -kxy, -kyy, kyt},   {kxt, kyt, b4, -kxt, -kyt, b2},
                          {-kxx, -kxy, -kxt, kxx, kxy, -kxt}, {-kxy, -kyy, -kyt, kxy, kyy, -kyt}, {kxt, kyt, b2, -kxt, -kyt, b4}};
  for (int r = 0; r < 6; ++r)
    for (int q = 0; q < 6; ++q) k[r][q] = v[r][q];
}

// FRAME_THREADS: 256 for narrow bands (more workgroups per CU, cheaper barriers: 1.5e7 5x5 frames/s vs 5.4e6 with
// 1024), 1024 for wide ones (the trailing update of a column, up to 63 * 64 / 2 pairs, in one pass: 7.9e5 10x10
// frames/s vs 6.6e5 with 256).
template <int FRAME_THREADS>
__global__ __launch_bounds__(FRAME_THREADS) void frame_solve_kernel(const FrameParams p) {
  extern __shared__ double lds[];
  const int n = p.n_eq, kd = p.kd, ld = kd + 1;
  double* ab = lds;                  // [n][ld]
  double* rhs = lds + (size_t)n * ld;  // [n]
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  if (tid == 0) s_bad = 0;
  for (int i = tid; i < n * ld + n; i += FRAME_THREADS) lds[i] = 0.0;
  __syncthreads();

  // ---- assembly: one thread per element, LDS atomics (elements sharing a node collide) ----
  const double* Ib = p.I + b * p.Ne;
  for (int e = tid; e < p.Ne; e += FRAME_THREADS) {
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
        atomicAdd(&ab[(size_t)eq[q] * ld + (eq[r] - eq[q])], k[r][q]);
      }
    }
  }
  const double* lb = p.loads + b * p.loads_bs;
  for (int i = tid; i < p.Nn * 3; i += FRAME_THREADS) {
    const int q = p.node_eq[i];
    if (q >= 0) atomicAdd(&rhs[q], lb[i]);
  }
  __syncthreads();

  // ---- band LDL^T, right-looking: column j holds d_j = ab[j][0] and the UNSCALED entries L_kj d_j ----
  for (int j = 0; j < n; ++j) {
    const double d = ab[(size_t)j * ld];
    if (!(d > 0.0)) { if (tid == 0) s_bad = 1; }
    const double rd = 1.0 / d;
    const int kmax = (kd < n - 1 - j) ? kd : n - 1 - j;
    // pairs (r, c), 1 <= r <= c <= kmax: A[j+c][j+r] -= A[j+r][j] A[j+c][j] / d
    const int c = 1 + (tid & 63);
    if (c <= kmax) {
      const double lc = ab[(size_t)j * ld + c] * rd;
      for (int r = 1 + (tid >> 6); r <= c; r += FRAME_THREADS / 64)
        ab[(size_t)(j + r) * ld + (c - r)] -= ab[(size_t)j * ld + r] * lc;
    }
    __syncthreads();
  }
  // ---- forward substitution (unit lower), diagonal scaling, backward substitution ----
  for (int j = 0; j < n; ++j) {
    const double yj = rhs[j], rd = 1.0 / ab[(size_t)j * ld];
    const int kmax = (kd < n - 1 - j) ? kd : n - 1 - j;
    const int k = 1 + tid;
    if (k <= kmax) rhs[j + k] -= ab[(size_t)j * ld + k] * rd * yj;
    __syncthreads();
  }
  for (int j = tid; j < n; j += FRAME_THREADS) rhs[j] /= ab[(size_t)j * ld];
  __syncthreads();
  for (int j = n - 1; j >= 0; --j) {   // x_j = z_j - sum_k (L_{j+k,j}) x_{j+k}: one wave reduces the <= kd terms
    if (tid < 64) {
      const int kmax = (kd < n - 1 - j) ? kd : n - 1 - j;
      double acc = 0.0;
      for (int k = 1 + tid; k <= kmax; k += 64) acc += ab[(size_t)j * ld + k] * rhs[j + k];
      for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
      if (tid == 0) rhs[j] -= acc / ab[(size_t)j * ld];
    }
    __syncthreads();
  }
  const bool bad = s_bad != 0;
  const double qnan = __builtin_nan("");
  // ---- nodal displacements ----
  for (int i = tid; i < p.Nn * 3; i += FRAME_THREADS) {
    const int q = p.node_eq[i];
    p.disp[b * (long)p.Nn * 3 + i] = bad ? qnan : (q >= 0 ? rhs[q] : 0.0);
  }
  // ---- element end forces (ElasticBeam2d::getResistingForce through LinearCrdTransf2d), global ----
  for (int e = tid; e < p.Ne; e += FRAME_THREADS) {
    const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
    const double EA = p.elem_EA[e], EI = p.elem_E[e] * Ib[e], wy = p.elem_w[2 * e], wx = p.elem_w[2 * e + 1];
    double ug[6];
    for (int r = 0; r < 6; ++r) { const int q = p.elem_eq[6 * e + r]; ug[r] = q >= 0 ? rhs[q] : 0.0; }
    const double ul[6] = {c * ug[0] + s * ug[1], -s * ug[0] + c * ug[1], ug[2], c * ug[3] + s * ug[4], -s * ug[3] + c * ug[4], ug[5]};
    const double chord = (ul[4] - ul[1]) / L;
    const double q0 = EA / L * (ul[3] - ul[0]) - wx * L / 2;
    const double q1 = 4 * EI / L * (ul[2] - chord) + 2 * EI / L * (ul[5] - chord) - wy * L * L / 12;
    const double q2 = 2 * EI / L * (ul[2] - chord) + 4 * EI / L * (ul[5] - chord) + wy * L * L / 12;
    const double pl[6] = {-q0 - wx * L, (q1 + q2) / L - wy * L / 2, q1, q0, -(q1 + q2) / L - wy * L / 2, q2};
    const double f[6] = {c * pl[0] - s * pl[1], s * pl[0] + c * pl[1], pl[2], c * pl[3] - s * pl[4], s * pl[3] + c * pl[4], pl[5]};
    double* fo = p.forces + (b * (long)p.Ne + e) * 6;
    for (int r = 0; r < 6; ++r) fo[r] = bad ? qnan : f[r];
    p.V[b * (long)p.Ne + e] = bad ? qnan : f[1];
    p.M[b * (long)p.Ne + e] = bad ? qnan : f[2];
  }
  if (tid == 0 && p.status) p.status[b] = bad ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------
// Frames whose band does not fit LDS (BASELINE config 5: ~500 elements): the assembled band lives in a
// caller-provided HBM workspace (ws[b] = band n_eq x (kd+1) followed by the right-hand side n_eq) and the
// factorisation slides a (kd+2)-column window through LDS: the right-looking update of column j only touches
// columns j+1 .. j+kd.  Column j leaves the window as a finished L column (written back over the assembled
// one), column j+kd+1 is prefetched into the slot column j-1 vacated one step earlier, so there is still ONE
// barrier per column.  Forward substitution rides along; backward substitution streams the L columns back in
// blocks.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_assemble_kernel(const FrameParams p, double* __restrict__ ws) {
  const long b = blockIdx.y;
  const int ld = p.kd + 1;
  double* ab = ws + b * ((long)p.n_eq * ld + p.n_eq);
  double* rhs = ab + (long)p.n_eq * ld;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const double* Ib = p.I + b * p.Ne;
  if (t < p.Ne) {
    const int e = t;
    const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
    double k[6][6];
    elem_global_k(L, c, s, p.elem_EA[e], p.elem_E[e] * Ib[e], k);
    const double wy = p.elem_w[2 * e], wx = p.elem_w[2 * e + 1];
    const double pl[6] = {wx * L / 2, wy * L / 2, wy * L * L / 12, wx * L / 2, wy * L / 2, -wy * L * L / 12};
    const double pg[6] = {c * pl[0] - s * pl[1], s * pl[0] + c * pl[1], pl[2], c * pl[3] - s * pl[4], s * pl[3] + c * pl[4], pl[5]};
    int eq[6];
    for (int r = 0; r < 6; ++r) eq[r] = p.elem_eq[6 * e + r];
    for (int r = 0; r < 6; ++r) {
      if (eq[r] < 0) continue;
      atomicAdd(&rhs[eq[r]], pg[r]);
      for (int q = 0; q < 6; ++q) {
        if (eq[q] < 0 || eq[q] > eq[r]) continue;
        atomicAdd(&ab[(long)eq[q] * ld + (eq[r] - eq[q])], k[r][q]);
      }
    }
  }
  if (t < p.Nn * 3) {
    const int q = p.node_eq[t];
    if (q >= 0) atomicAdd(&rhs[q], (p.loads + b * p.loads_bs)[t]);
  }
}

__global__ __launch_bounds__(1024) void frame_factor_big_kernel(const FrameParams p, double* __restrict__ ws) {
  extern __shared__ double lds[];
  const int n = p.n_eq, kd = p.kd, ld = kd + 1, W = kd + 2;
  double* win = lds;                       // [W][ld]   ring of columns, slot = column % W
  double* rhs = win + (size_t)W * ld;      // [n]
  double* dinv = rhs + n;                  // [n]       1 / d_j
  double* blk = dinv + n;                  // [2][BLK][ld] column blocks of the backward sweep
  constexpr int BLK = 16;
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  double* ab = ws + b * ((long)n * ld + n);
  const double* rhs_g = ab + (long)n * ld;
  if (tid == 0) s_bad = 0;
  for (int i = tid; i < n; i += 1024) rhs[i] = rhs_g[i];
  for (int i = tid; i < (kd + 1) * ld && i < n * ld; i += 1024) win[i] = ab[i];   // columns 0..kd: slot = column
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    const double* colj = win + (size_t)(j % W) * ld;
    const double d = colj[0];
    if (!(d > 0.0)) { if (tid == 0) s_bad = 1; }
    const double rd = 1.0 / d;
    const int kmax = (kd < n - 1 - j) ? kd : n - 1 - j;
    // finished column j back to HBM; column j+kd+1 into the slot column j-1 left (not touched in this step)
    double pre = 0.0;
    const int cin = j + kd + 1;
    if (tid < ld) {
      ab[(long)j * ld + tid] = colj[tid];
      if (cin < n) pre = ab[(long)cin * ld + tid];
    }
    if (tid == 0) dinv[j] = rd;
    // forward substitution rides along (unit lower factor): threads 64.. so that the update pairs keep wave 0..
    if (tid >= 1024 - 64) {
      const int k = 1 + (tid - (1024 - 64));
      if (k <= kmax) rhs[j + k] -= colj[k] * rd * rhs[j];
    }
    const int c = 1 + (tid & 63);
    if (c <= kmax) {
      const double lc = colj[c] * rd;
      for (int r = 1 + (tid >> 6); r <= c; r += 16)
        win[(size_t)((j + r) % W) * ld + (c - r)] -= colj[r] * lc;
    }
    if (tid < ld && cin < n) win[(size_t)(cin % W) * ld + tid] = pre;
    __syncthreads();
  }
  for (int i = tid; i < n; i += 1024) rhs[i] *= dinv[i];
  __syncthreads();
  // backward substitution: L columns stream back in blocks of BLK (waves 1.. prefetch, wave 0 substitutes)
  const int nblk = (n + BLK - 1) / BLK;
  for (int i = tid; i < BLK * ld; i += 1024) {      // last block first
    const int col = (nblk - 1) * BLK + i / ld;
    blk[i] = col < n ? ab[(long)col * ld + i % ld] : 0.0;
  }
  __syncthreads();
  for (int kb = nblk - 1; kb >= 0; --kb) {
    double* cur = blk + (size_t)((nblk - 1 - kb) & 1) * BLK * ld;
    double* nxt = blk + (size_t)((nblk - kb) & 1) * BLK * ld;
    if (tid >= 64 && kb > 0) {
      for (int i = tid - 64; i < BLK * ld; i += 1024 - 64) nxt[i] = ab[(long)((kb - 1) * BLK + i / ld) * ld + i % ld];
    }
    if (tid < 64) {
      for (int jj = BLK - 1; jj >= 0; --jj) {
        const int j = kb * BLK + jj;
        if (j >= n) continue;
        const int kmax = (kd < n - 1 - j) ? kd : n - 1 - j;
        double acc = 0.0;
        for (int k = 1 + tid; k <= kmax; k += 64) acc += cur[jj * ld + k] * rhs[j + k];
        for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
        if (tid == 0) rhs[j] -= acc * dinv[j];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __asm__ volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
  }
  const bool bad = s_bad != 0;
  const double qnan = __builtin_nan("");
  for (int i = tid; i < p.Nn * 3; i += 1024) {
    const int q = p.node_eq[i];
    p.disp[b * (long)p.Nn * 3 + i] = bad ? qnan : (q >= 0 ? rhs[q] : 0.0);
  }
  const double* Ib = p.I + b * p.Ne;
  for (int e = tid; e < p.Ne; e += 1024) {
    const double L = p.elem_geo[3 * e], c = p.elem_geo[3 * e + 1], s = p.elem_geo[3 * e + 2];
    const double EA = p.elem_EA[e], EI = p.elem_E[e] * Ib[e], wy = p.elem_w[2 * e], wx = p.elem_w[2 * e + 1];
    double ug[6];
    for (int r = 0; r < 6; ++r) { const int q = p.elem_eq[6 * e + r]; ug[r] = q >= 0 ? rhs[q] : 0.0; }
    const double ul[6] = {c * ug[0] + s * ug[1], -s * ug[0] + c * ug[1], ug[2], c * ug[3] + s * ug[4], -s * ug[3] + c * ug[4], ug[5]};
    const double chord = (ul[4] - ul[1]) / L;
    const double q0 = EA / L * (ul[3] - ul[0]) - wx * L / 2;
    const double q1 = 4 * EI / L * (ul[2] - chord) + 2 * EI / L * (ul[5] - chord) - wy * L * L / 12;
    const double q2 = 2 * EI / L * (ul[2] - chord) + 4 * EI / L * (ul[5] - chord) + wy * L * L / 12;
    const double pl[6] = {-q0 - wx * L, (q1 + q2) / L - wy * L / 2, q1, q0, -(q1 + q2) / L - wy * L / 2, q2};
    const double f[6] = {c * pl[0] - s * pl[1], s * pl[0] + c * pl[1], pl[2], c * pl[3] - s * pl[4], s * pl[3] + c * pl[4], pl[5]};
    double* fo = p.forces + (b * (long)p.Ne + e) * 6;
    for (int r = 0; r < 6; ++r) fo[r] = bad ? qnan : f[r];
    p.V[b * (long)p.Ne + e] = bad ? qnan : f[1];
    p.M[b * (long)p.Ne + e] = bad ? qnan : f[2];
  }
  if (tid == 0 && p.status) p.status[b] = bad ? 1 : 0;
}

}  // namespace opsamd

using namespace opsamd;

extern "C" size_t ops_frame_workspace_bytes(int B, int n_eq, int half_bandwidth) {
  const size_t lds_bytes = ((size_t)n_eq * (half_bandwidth + 1) + n_eq) * sizeof(double);
  if (lds_bytes <= 160 * 1024 - 64) return 0;   // the band lives in LDS
  return (size_t)B * ((size_t)n_eq * (half_bandwidth + 1) + n_eq) * sizeof(double);
}

extern "C" int ops_frame_solve_batched_f64(int B, int n_nodes, int n_elems, int n_eq, int half_bandwidth,
                                           const double* elem_geo, const double* elem_EA, const double* elem_E,
                                           const double* elem_w, const int32_t* elem_eq, const int32_t* node_eq,
                                           const double* I, const double* loads, long loads_bstride, double* disp,
                                           double* forces, double* V, double* M, int32_t* status, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (B < 0 || n_nodes < 2 || n_elems < 1 || n_eq < 1 || half_bandwidth < 0) return OPS_AMD_ERR_INVALID_ARG;
  if (B == 0) return OPS_AMD_OK;
  if (!elem_geo || !elem_EA || !elem_E || !elem_w || !elem_eq || !node_eq || !I || !loads || !disp || !forces || !V || !M)
    return OPS_AMD_ERR_INVALID_ARG;
  if (half_bandwidth > 63) return OPS_AMD_ERR_UNSUPPORTED;    // one 64-lane row of update columns
  const size_t lds_bytes = ((size_t)n_eq * (half_bandwidth + 1) + n_eq) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)frame_solve_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
        hipFuncSetAttribute((const void*)frame_solve_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess ||
        hipFuncSetAttribute((const void*)frame_factor_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64) != hipSuccess)
      return OPS_AMD_ERR_LAUNCH;
    attr_set = true;
  }
  const FrameParams p{B, n_nodes, n_elems, n_eq, half_bandwidth, elem_geo, elem_EA, elem_E, elem_w, elem_eq, node_eq,
                      I, loads, loads_bstride, disp, forces, V, M, status};
  if (lds_bytes > 160 * 1024 - 64) {
    // band in the HBM workspace, sliding LDS window
    const int ld = half_bandwidth + 1;
    const size_t need = ops_frame_workspace_bytes(B, n_eq, half_bandwidth);
    const size_t lds2 = ((size_t)(half_bandwidth + 2) * ld + 2 * (size_t)n_eq + 2 * 16 * (size_t)ld) * sizeof(double);
    if (lds2 > 160 * 1024 - 64) return OPS_AMD_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < need) return OPS_AMD_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, need, s) != hipSuccess) return OPS_AMD_ERR_LAUNCH;
    const int work = n_elems > n_nodes * 3 ? n_elems : n_nodes * 3;
    hipLaunchKernelGGL(frame_assemble_kernel, dim3((unsigned)((work + 255) / 256), (unsigned)B), dim3(256), 0, s, p, (double*)workspace);
    hipLaunchKernelGGL(frame_factor_big_kernel, dim3((unsigned)B), dim3(1024), lds2, s, p, (double*)workspace);
    return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
  }
  if (half_bandwidth > 24)
    hipLaunchKernelGGL(frame_solve_kernel<1024>, dim3((unsigned)B), dim3(1024), lds_bytes, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(frame_solve_kernel<256>, dim3((unsigned)B), dim3(256), lds_bytes, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? OPS_AMD_OK : OPS_AMD_ERR_LAUNCH;
}
