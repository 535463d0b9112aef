// Micro-benchmark: dependent / independent v_fma_f64, v_rcp_f64 and v_fma_f32 issue cost per wave, and the clock64 rate
// (hipcc --offload-arch=gfx950 -O3 -o fp64_latency fp64_latency.hip).  MI355X: 4.3 cycles per dependent FP64 FMA, 27 per rcp+add, 2.37 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define TICK(var, dep) { asm volatile("s_nop 0" : "+v"(dep)); __builtin_amdgcn_sched_barrier(0); var = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(var)); __builtin_amdgcn_sched_barrier(0); }
__global__ void k(double* out, long long* t, double x0) {
  double x = x0 + threadIdx.x * 1e-9, y = 1.0000001;
  long long c0, c1, c2, c3, c4, c5, c6; TICK(c0, x)
#pragma unroll
  for (int i = 0; i < 256; ++i) x = __builtin_fma(x, y, 1e-30);
  TICK(c1, x)
  double r = x;
#pragma unroll
  for (int i = 0; i < 64; ++i) r = __builtin_amdgcn_rcp(r) + 1.5;   // rcp + add dependent
  TICK(c2, r)
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    a0 = __builtin_fma(a0, y, 1e-30); a1 = __builtin_fma(a1, y, 1e-30); a2 = __builtin_fma(a2, y, 1e-30); a3 = __builtin_fma(a3, y, 1e-30);
    a4 = __builtin_fma(a4, y, 1e-30); a5 = __builtin_fma(a5, y, 1e-30); a6 = __builtin_fma(a6, y, 1e-30); a7 = __builtin_fma(a7, y, 1e-30);
  }
  a0 += a1 + a2 + a3 + a4 + a5 + a6 + a7; TICK(c3, a0)
  float f = (float)x;
#pragma unroll
  for (int i = 0; i < 256; ++i) f = __builtin_fmaf(f, 1.0000001f, 1e-30f);
  TICK(c4, f)
  unsigned long long w0 = wall_clock64();
  TICK(c5, x)
  for (int i = 0; i < 2000; ++i) x = __builtin_fma(x, y, 1e-30);
  TICK(c6, x)
  unsigned long long w1 = wall_clock64();
  out[threadIdx.x] = x + r + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f;
  if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = c2 - c1; t[2] = c3 - c2; t[3] = c4 - c3; t[4] = c6 - c5; t[5] = (long long)(w1 - w0); }
}
int main() {
  double* o; long long* t; hipMalloc(&o, 8 * 64); hipMalloc(&t, 64);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, 1.0); hipDeviceSynchronize(); }
  long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
  printf("dep fma64: %.1f ticks/op; dep rcp64+add: %.1f ticks/pair; 8 indep fma64 chains: %.1f ticks/op; dep fma32: %.1f ticks/op\n", h[0] / 256.0, h[1] / 64.0, h[2] / 256.0, h[3] / 256.0);
  printf("clock64 ticks per 100MHz tick: %.2f (=> clock64 at %.0f MHz)\n", (double)h[4] / h[5], 100.0 * h[4] / h[5]);
  return 0;
}
