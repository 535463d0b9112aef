// Probe (r06): can a wave broadcast a 52-double line to itself through HBM-backed memory and the SCALAR data path instead of LDS?
//   per step: every lane stores one double (512 B per wave, coalesced) -> s_waitcnt vmcnt(0) -> s_load_dwordx16 x 4 + x 3 (glc) -> W FMAs with
//   SGPR operands.  Measures the chain (1 wave per SIMD) and the throughput (2..4 waves per SIMD), and checks that the scalar loads see the stores.
// build: hipcc -O3 --offload-arch=gfx950 sline_probe.hip -o sline_probe ; run: ./sline_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ double dd(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

template <int NF>
__global__ __launch_bounds__(256) void probe(double* __restrict__ ws, double* __restrict__ out, int n, int pitch) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  double* base = ws + wave * (long)n * pitch;
  double acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.0;
  double v = (double)(lane + 1);
  double chk = 0.0;
  for (int j = 0; j < n; ++j) {
    double* col = base + (long)j * pitch;
    col[lane] = v;
    const unsigned long long a = (unsigned long long)col;
    const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)a), ahi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long sa = ((unsigned long long)ahi << 32) | alo;
    u32x16 c0, c1, c2, c3;
    __asm__ volatile("s_waitcnt vmcnt(0)\n\ts_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\t"
                     "s_load_dwordx16 %2, %4, 0x80 glc\n\ts_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3) : "s"(sa) : "memory");
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[t] = __builtin_fma(v, dd(c0[2 * t], c0[2 * t + 1]), acc[t]);
      acc[t] = __builtin_fma(v, dd(c1[2 * t], c1[2 * t + 1]), acc[t]);
      acc[t] = __builtin_fma(v, dd(c2[2 * t], c2[2 * t + 1]), acc[t]);
      acc[t] = __builtin_fma(v, dd(c3[2 * t], c3[2 * t + 1]), acc[t]);
    }
    if (NF > 32) {
      u32x16 c4, c5;
      u32x8 c6;
      __asm__ volatile("s_load_dwordx16 %0, %3, 0x100 glc\n\ts_load_dwordx16 %1, %3, 0x140 glc\n\ts_load_dwordx8 %2, %3, 0x180 glc\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(c4), "=&s"(c5), "=&s"(c6) : "s"(sa) : "memory");
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t] = __builtin_fma(v, dd(c4[2 * t], c4[2 * t + 1]), acc[t]);
        acc[t] = __builtin_fma(v, dd(c5[2 * t], c5[2 * t + 1]), acc[t]);
        if (t < 4) acc[t] = __builtin_fma(v, dd(c6[2 * t], c6[2 * t + 1]), acc[t]);
      }
      s = dd(c6[6], c6[7]);                      // entry 51 of the line
    } else {
      s = dd(c3[14], c3[15]);                    // entry 31
    }
    chk += s;
    // next line depends on what came back (the chain the factorisation has)
    v = __builtin_fma(s, 1e-9, (double)(lane + 1 + ((j + 1) & 7)));
  }
  double r = chk;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i] * 1e-30;
  if (lane == 0) out[wave] = r;
}

int main() {
  const int n = 768, pitch = 64;
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  for (int wps = 1; wps <= 4; ++wps) {
    const int blocks = cus * wps;                       // 4 waves per block: wps waves per SIMD if one block per CU round-robin
    const long waves = (long)blocks * 4;
    double *ws, *out;
    hipMalloc(&ws, waves * n * pitch * 8); hipMalloc(&out, waves * 8);
    hipMemset(ws, 0, waves * n * pitch * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      probe<52><<<blocks, 256>>>(ws, out, n, pitch);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<double> h(waves); hipMemcpy(h.data(), out, waves * 8, hipMemcpyDeviceToHost);
      // expected chk: entry 51 of line j = lane 51's v_j ; v_0 = 52, v_j = s_(j-1) * 1e-9 + 52 + (j & 7)
      double vv = 52.0, ex = 0.0;
      for (int j = 0; j < n; ++j) { ex += vv; vv = vv * 1e-9 + (double)(52 + ((j + 1) & 7)); }
      int badw = 0; for (long w = 0; w < waves; ++w) if (fabs(h[w] - ex) > 1e-6 * ex) ++badw;
      printf("waves/SIMD %d  blocks %d  %.3f ms  per step %.1f ns  (%.0f cycles at 2.4 GHz)  wrong waves %d of %ld (got %.6f want %.6f)\n", wps, blocks, ms,
             ms * 1e6 / n, ms * 1e6 / n * 2.4, badw, waves, h[0], ex);
    }
    hipFree(ws); hipFree(out);
  }
  return 0;
}
