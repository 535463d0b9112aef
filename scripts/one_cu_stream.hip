// How fast does ONE workgroup pull a small network's weights through MFMA?  (feasibility number for a whole-step PINN kernel)
// 16 waves; weights in MFMA fragment order (1 KB per (16-column tile, 32-deep step)); A operand (128 rows) in LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int NW>
__global__ __launch_bounds__(64 * NW) void stream_kernel(const uint4* __restrict__ W, int n_tiles, int ksteps, float* out, int reps) {
  __shared__ __attribute__((aligned(16))) uint16_t sA[128 * 360];       // 90 KB: 128 rows x 352 (+8) columns
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 128 * 360; i += 64 * NW) sA[i] = (uint16_t)(0x3c00 + (i & 63));
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = f32x4{0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep)
    for (int t = wave; t < n_tiles; t += NW) {            // a wave owns column tiles t; all 8 row blocks of A against it
      const uint4* wt = W + ((size_t)t * ksteps) * 64 + lane;
      for (int k0 = 0; k0 < ksteps; k0 += 4) {
        uint4 fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = wt[(size_t)(k0 + j < ksteps ? k0 + j : ksteps - 1) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (k0 + j >= ksteps) break;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const uint4 fa = *(const uint4*)(sA + (16 * r + (lane & 15)) * 360 + 32 * (k0 + j) + 8 * (lane >> 4));
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb[j]), acc[r], 0, 0, 0);
          }
        }
      }
    }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
  out[blockIdx.x * 64 * NW + tid] = s;
}

// store rate of one workgroup: float32 "weight gradients" from registers
template <int NW>
__global__ __launch_bounds__(64 * NW) void store_kernel(float4* __restrict__ out, long n4, int reps) {
  const int tid = threadIdx.x;
  for (int rep = 0; rep < reps; ++rep)
    for (long i = tid; i < n4; i += 64 * NW) out[i] = float4{(float)i, 1.f, 2.f, (float)rep};
}

int main() {
  const int ksteps = 11, n_tiles_layer = 11;              // one 350 -> 175 layer: 11 column tiles x 11 steps of 32 = 121 KB
  const int layers = 9;                                   // ~1.1 MB of weights
  const int n_tiles = n_tiles_layer * layers;
  const size_t wbytes = (size_t)n_tiles * ksteps * 1024;
  uint4* W; float* out; float4* g;
  CK(hipMalloc(&W, wbytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&g, 4 << 20));
  CK(hipMemset(W, 0x3c, wbytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("weights %.2f MB (%d tiles x %d steps)\n", wbytes / 1e6, n_tiles, ksteps);
  for (int wg = 1; wg <= 4; wg *= 2)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      stream_kernel<16><<<wg, 1024>>>(W, n_tiles, ksteps, out, 20);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("stream + MFMA, %d workgroup(s) of 16 waves (each pulls ALL weights): %.2f us per pass of %.2f MB = %.1f GB/s per workgroup\n", wg, ms * 1e3 / 20, wbytes / 1e6, wbytes / (ms * 1e-3 / 20) / 1e9);
    }
  const long n4 = (2200000 / 16);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    store_kernel<16><<<1, 1024>>>(g, n4, 20);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 2) printf("store 2.2 MB of float32 from one workgroup: %.2f us per pass = %.1f GB/s\n", ms * 1e3 / 20, 2.2e6 / (ms * 1e-3 / 20) / 1e9);
  }
  return 0;
}
