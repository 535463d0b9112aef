// Second probe: issue interval of independent v_mfma_f64_16x16x4_f64 (distinct operands: no CSE), and what overlaps with an MFMA in
// flight -- FP64 VALU FMAs, FP32 FMAs, integer VALU, LDS reads -- inside one wave and across two waves of one SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define TICK(var) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); var = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(var)); __builtin_amdgcn_sched_barrier(0); }
#define MF(c, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define F64(x, y) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(y))
#define F32(x, y) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(y))
#define I32(x, y) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y))

__global__ void k(double* out, long long* t, double x0, int mode) {
  __shared__ double sh[1024];
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a = x0 + l * 1e-9, b = 1.0 + l * 1e-12;
  d4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (d4){i + a, i + b, i * a, i * b};
  double f[4] = {a, a + 1, a + 2, a + 3};
  float g[4] = {1.f + l, 2.f, 3.f, 4.f};
  unsigned u[4] = {(unsigned)l, 2u, 3u, 4u};
  sh[threadIdx.x & 1023] = a;
  __syncthreads();
  long long t0, t1;
  if (blockDim.x == 64) {
    long long r[8];
    TICK(t0)
#pragma unroll
    for (int i = 0; i < 32; ++i) { MF(c[0], a, b); MF(c[1], a, b); MF(c[2], a, b); MF(c[3], a, b); MF(c[4], a, b); MF(c[5], a, b); MF(c[6], a, b); MF(c[7], a, b); }
    TICK(t1) r[0] = t1 - t0;                                 // 256 independent MFMAs
    TICK(t0)
#pragma unroll
    for (int i = 0; i < 64; ++i) { MF(c[i & 7], a, b);
#pragma unroll
      for (int q = 0; q < 3; ++q) { F64(f[0], b); F64(f[1], b); F64(f[2], b); F64(f[3], b); } }
    TICK(t1) r[1] = t1 - t0;                                 // 64 x (1 MFMA + 12 FP64 FMA)
    TICK(t0)
#pragma unroll
    for (int i = 0; i < 64; ++i) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { F64(f[0], b); F64(f[1], b); F64(f[2], b); F64(f[3], b); } }
    TICK(t1) r[2] = t1 - t0;                                 // 64 x 12 FP64 FMA
    TICK(t0)
#pragma unroll
    for (int i = 0; i < 64; ++i) { MF(c[i & 7], a, b);
#pragma unroll
      for (int q = 0; q < 3; ++q) { F32(g[0], g[3]); F32(g[1], g[3]); F32(g[2], g[3]); I32(u[0], u[3]); } }
    TICK(t1) r[3] = t1 - t0;                                 // 64 x (1 MFMA + 9 FP32 FMA + 3 int add)
    TICK(t0)
#pragma unroll
    for (int i = 0; i < 64; ++i) { MF(c[i & 7], a, b);
#pragma unroll
      for (int q = 0; q < 6; ++q) { typedef double dd2 __attribute__((ext_vector_type(2))); dd2 v; unsigned ad = (unsigned)(((l * 2 + 16 * q) & 1022) * 8); asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(ad)); asm volatile("" :: "v"(v)); } }
    TICK(t1) r[4] = t1 - t0;                                 // 64 x (1 MFMA + 6 ds_read_b128)
    if (l == 0) for (int i = 0; i < 5; ++i) t[i] = r[i];
  } else {                                                    // 5 waves: 0 and 4 share a SIMD
    __syncthreads();
    TICK(t0)
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) { MF(c[0], a, b); MF(c[1], a, b); MF(c[2], a, b); MF(c[3], a, b); MF(c[4], a, b); MF(c[5], a, b); MF(c[6], a, b); MF(c[7], a, b); }
    } else if (wave == 4) {
      if (mode == 0) {
#pragma unroll
        for (int i = 0; i < 1024; ++i) { F64(f[0], b); F64(f[1], b); F64(f[2], b); F64(f[3], b); }
      } else if (mode == 1) {
#pragma unroll
        for (int i = 0; i < 1024; ++i) { F32(g[0], g[3]); F32(g[1], g[3]); F32(g[2], g[3]); I32(u[0], u[3]); }
      } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) { MF(c[0], a, b); MF(c[1], a, b); MF(c[2], a, b); MF(c[3], a, b); MF(c[4], a, b); MF(c[5], a, b); MF(c[6], a, b); MF(c[7], a, b); }
      }
    }
    TICK(t1)
    if (l == 0) t[8 + wave] = t1 - t0;
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[threadIdx.x] = s + f[0] + f[1] + f[2] + f[3] + g[0] + g[1] + g[2] + u[0];
}
int main() {
  double* o; long long* t; (void)hipMalloc(&o, 8 * 512); (void)hipMalloc(&t, 8 * 32);
  long long h[32];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, 1.0, 0); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("256 independent MFMA f64: %.1f cycles each\n", h[0] / 256.0);
  printf("per round: MFMA + 12 FMA64 %.1f | 12 FMA64 alone %.1f | MFMA + 9 FMA32 + 3 int %.1f | MFMA + 6 ds_read_b128 %.1f\n", h[1] / 64.0, h[2] / 64.0, h[3] / 64.0, h[4] / 64.0);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(320), 0, 0, o, t, 1.0, mode); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    printf("two waves on one SIMD, wave 0 = 256 MFMA, wave 4 = %s: %lld and %lld cycles\n", mode == 0 ? "4096 FMA64" : mode == 1 ? "3072 FMA32 + 1024 int" : "256 MFMA", h[8], h[12]);
  }
  return 0;
}
