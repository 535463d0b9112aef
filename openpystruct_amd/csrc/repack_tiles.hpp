// Fragment-tiled bf16 copies of weight matrices (the layout of csrc/mlp_block.hip) from the float32 parameters: ONE WAVE per 1 KB tile.
// Shared by the optimiser's launch behind its update (flat_adam.hip repack_tiles_kernel) and, r05, by the PINN's batch-assembly launch
// (mlp_block.hip: the tile jobs ride along as extra workgroups of the next step's gather -- two adjacent launches without a mutual
// dependency become one).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/openpystruct_amd.h"

namespace opsamd {

// bf16 copies of up to OPS_MLP_MAX_REPACK weight matrices in the layout of csrc/mlp_block.hip (padded, plain and transposed), rebuilt
// by repack_tiles_kernel right behind the update: the PINN's and the TFD's layer-block launches read them
struct AdamRepack {
  int nmat;
  long off[OPS_MLP_MAX_REPACK];        // first element of matrix i in the flat parameter buffer
  int N[OPS_MLP_MAX_REPACK], K[OPS_MLP_MAX_REPACK], ldw[OPS_MLP_MAX_REPACK], ldwt[OPS_MLP_MAX_REPACK];
  uint16_t* Wp[OPS_MLP_MAX_REPACK];
  uint16_t* Wtp[OPS_MLP_MAX_REPACK];
};

// The tiled copies as a launch of their own BEHIND the update (r03).  Written from inside the update they were two scattered 2-byte
// stores per element -- 2-byte pieces of 128-byte lines that eight different waves (often on different XCDs) fill: the update took
// 12.2 us instead of 4.7 for the TFD model (0.36 M parameters), 17.4 us for the PINN's 0.6 M.  Here ONE WAVE builds one 1 KB tile:
// lane l gathers its eight values (the fragment of MFMA lane l) from the float32 parameters the update has just written (L2 / Infinity
// Cache hits) and stores 16 bytes at tile * 1024 + 16 l -- every store instruction of a wave is one contiguous KB.
struct TileJobs {
  int nmat;
  int first[2 * OPS_MLP_MAX_REPACK + 1];       // first tile of (matrix q, plain) = first[2 q], (matrix q, transposed) = first[2 q + 1]
};
__device__ __forceinline__ void repack_tile_job(const float* __restrict__ p, const AdamRepack& rp, const TileJobs& tj, int tile, int lane) {
  if (tile >= tj.first[2 * tj.nmat]) return;                       // wave-uniform
  int job = 0;
#pragma unroll
  for (int k = 1; k < 2 * OPS_MLP_MAX_REPACK; ++k)
    if (k < 2 * tj.nmat && tile >= tj.first[k]) job = k;
  const int q = job >> 1, tr = job & 1, t = tile - tj.first[job];
  long off = 0; int N = 0, K = 0, ldw = 0, ldwt = 0; uint16_t* Wp = nullptr; uint16_t* Wtp = nullptr;
#pragma unroll
  for (int k = 0; k < OPS_MLP_MAX_REPACK; ++k)                       // (constant indices: the argument block stays in scalar registers)
    if (k == q) { off = rp.off[k]; N = rp.N[k]; K = rp.K[k]; ldw = rp.ldw[k]; ldwt = rp.ldwt[k]; Wp = rp.Wp[k]; Wtp = rp.Wtp[k]; }
  const float* W = p + off;
  const int ks = (tr ? ldwt : ldw) >> 5, tb = t / ks, tk = t - tb * ks;      // tile (block of 16 along the tile's "row" axis, reduction step)
  const int a = 16 * tb + (lane & 15), b0 = 32 * tk + 8 * (lane >> 4);      // plain: a = row r, b = column c; transposed: a = column c, b = row r
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int b = b0 + j;
    const int r = tr ? b : a, c = tr ? a : b;
    v[j] = (r < N && c < K) ? W[(long)r * K + c] : 0.0f;
  }
  auto to_bf16 = [](float f) -> uint32_t {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
  };
  uint4 o;
  o.x = to_bf16(v[0]) | (to_bf16(v[1]) << 16); o.y = to_bf16(v[2]) | (to_bf16(v[3]) << 16);
  o.z = to_bf16(v[4]) | (to_bf16(v[5]) << 16); o.w = to_bf16(v[6]) | (to_bf16(v[7]) << 16);
  *(uint4*)((tr ? Wtp : Wp) + (long)t * 512 + 8 * lane) = o;
}


inline TileJobs make_tile_jobs(const AdamRepack& rp) {
  TileJobs tj;
  tj.nmat = rp.nmat;
  int tot = 0;
  for (int q = 0; q < rp.nmat; ++q) {
    tj.first[2 * q] = tot;
    tot += ((rp.N[q] + 15) / 16) * (rp.ldw[q] / 32);
    tj.first[2 * q + 1] = tot;
    tot += ((rp.K[q] + 15) / 16) * (rp.ldwt[q] / 32);
  }
  tj.first[2 * rp.nmat] = tot;
  return tj;
}

// entries -> the device table (every W must point into [params, params + n))
inline int make_adam_repack(long n, const float* params, int nmat, const ops_mlp_repack_entry* entries, AdamRepack* out) {
  if (nmat < 1 || nmat > OPS_MLP_MAX_REPACK || !entries || !params) return OPS_AMD_ERR_INVALID_ARG;
  AdamRepack rp{};
  rp.nmat = nmat;
  for (int i = 0; i < nmat; ++i) {
    const ops_mlp_repack_entry& e = entries[i];
    if (!e.W || !e.Wp || !e.Wtp || e.N < 1 || e.K < 1 || e.ldw % 32 || e.ldwt % 32 || e.ldw < (e.K + 31) / 32 * 32 || e.ldwt < (e.N + 31) / 32 * 32) return OPS_AMD_ERR_INVALID_ARG;
    const long off = e.W - params;
    if (off < 0 || off + (long)e.N * e.K > n) return OPS_AMD_ERR_INVALID_ARG;      // the matrix must live inside the flat buffer
    rp.off[i] = off; rp.N[i] = e.N; rp.K[i] = e.K; rp.ldw[i] = e.ldw; rp.ldwt[i] = e.ldwt;
    rp.Wp[i] = (uint16_t*)e.Wp; rp.Wtp[i] = (uint16_t*)e.Wtp;
  }
  *out = rp;
  return OPS_AMD_OK;
}

}  // namespace opsamd
