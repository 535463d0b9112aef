// Probe for the FP64 matrix instruction on gfx950 (MI355X), before building the frame kernel on it:
//   1. layout of v_mfma_f64_16x16x4_f64 with exact integer data (asymmetric operands),
//   2. cycles per instruction: independent accumulators back to back, one dependent accumulator chain,
//   3. do VALU FP64 FMAs of the SAME wave issue under an MFMA in flight (mixed stream vs the sum of its parts)?
//   4. two waves on one SIMD: one issuing MFMAs, one issuing VALU FMAs (do the pipes overlap across waves?)
//   5. v_permlane32_swap / v_permlane16_swap on 64-bit values.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe scripts/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));
#define TICK(var) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); var = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(var)); __builtin_amdgcn_sched_barrier(0); }

__global__ void layout_kernel(const double* A, const double* B, double* D) {   // A [16][4], B [4][16] row-major, D [16][16]
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[((l >> 4) + 4 * i) * 16 + (l & 15)] = c[i];
}

__global__ void swap_kernel(double* out) {
  const int l = threadIdx.x;
  double v = (double)l;
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  // permlane32_swap(old, src): swaps the upper 32 lanes of `old`.. semantics probed: we print what each lane sees
  auto r32lo = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto r32hi = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  auto r16lo = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto r16hi = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  out[l] = __hiloint2double((int)r32hi[0], (int)r32lo[0]);
  out[64 + l] = __hiloint2double((int)r32hi[1], (int)r32lo[1]);
  out[128 + l] = __hiloint2double((int)r16hi[0], (int)r16lo[0]);
  out[192 + l] = __hiloint2double((int)r16hi[1], (int)r16lo[1]);
}

__global__ void rate_kernel(double* out, long long* t, double x0) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a = x0 + l * 1e-9, b = 1.0 + l * 1e-12;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  long long t0, t1, t2, t3, t4, t5;
  double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
  if (blockDim.x == 64) {
    TICK(t0)
#pragma unroll
    for (int i = 0; i < 32; ++i) {    // 256 independent-accumulator MFMAs
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c7, 0, 0, 0);
    }
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    TICK(t1)
#pragma unroll
    for (int i = 0; i < 64; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);     // dependent chain
    asm volatile("" : "+v"(c0));
    TICK(t2)
#pragma unroll
    for (int i = 0; i < 64; ++i) {     // mixed: 1 MFMA (independent accumulators, 4-way) + 12 VALU FMAs per round
      if ((i & 3) == 0) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      else if ((i & 3) == 1) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      else if ((i & 3) == 2) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
      else c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 3; ++k) { f0 = __builtin_fma(f0, b, 1e-30); f1 = __builtin_fma(f1, b, 1e-30); f2 = __builtin_fma(f2, b, 1e-30); f3 = __builtin_fma(f3, b, 1e-30); }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
    TICK(t3)
#pragma unroll
    for (int i = 0; i < 64; ++i) {     // the VALU part alone
#pragma unroll
      for (int k = 0; k < 3; ++k) { f0 = __builtin_fma(f0, b, 1e-30); f1 = __builtin_fma(f1, b, 1e-30); f2 = __builtin_fma(f2, b, 1e-30); f3 = __builtin_fma(f3, b, 1e-30); }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
    TICK(t4)
    // MFMA result consumed by VALU right away (latency until a dependent VALU read)
#pragma unroll
    for (int i = 0; i < 32; ++i) { c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0); a = __builtin_fma(c5[0], 1e-300, a); }
    asm volatile("" : "+v"(c5), "+v"(a));
    TICK(t5)
    if (l == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; }
  } else {
    // 5 waves: waves 0 and 4 land on the same SIMD (round-robin over 4 SIMDs): wave 0 MFMAs, wave 4 VALU; waves 1..3 idle
    __syncthreads();
    TICK(t0)
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c7, 0, 0, 0);
      }
      asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    } else if (wave == 4) {
#pragma unroll
      for (int i = 0; i < 1024; ++i) { f0 = __builtin_fma(f0, b, 1e-30); f1 = __builtin_fma(f1, b, 1e-30); f2 = __builtin_fma(f2, b, 1e-30); f3 = __builtin_fma(f3, b, 1e-30); }
      asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
    }
    TICK(t1)
    if (l == 0) t[8 + wave] = t1 - t0;
  }
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[0] + c6[0] + c7[0] + f0 + f1 + f2 + f3 + a;
}

int main() {
  double hA[64], hB[64], hD[256], *dA, *dB, *dD;
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i + 100 * k;       // asymmetric
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 3 + 7 * j + 1000 * k;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double r = 0; for (int k = 0; k < 4; ++k) r += hA[i * 4 + k] * hB[k * 16 + j]; bad += (r != hD[i * 16 + j]); }
  printf("layout: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15], D reg i = D[(l>>4)+4i][l&15]: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
  double hs[256], *ds; hipMalloc(&ds, sizeof hs);
  hipLaunchKernelGGL(swap_kernel, dim3(1), dim3(64), 0, 0, ds); hipMemcpy(hs, ds, sizeof hs, hipMemcpyDeviceToHost);
  for (int q = 0; q < 4; ++q) { printf("%s:", q == 0 ? "permlane32_swap[0]" : q == 1 ? "permlane32_swap[1]" : q == 2 ? "permlane16_swap[0]" : "permlane16_swap[1]");
    for (int l = 0; l < 64; l += 1) printf(" %d", (int)hs[64 * q + l]); printf("\n"); }
  double* o; long long* t; hipMalloc(&o, 8 * 512); hipMalloc(&t, 8 * 32);
  long long h[32];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(64), 0, 0, o, t, 1.0); hipDeviceSynchronize(); }
  hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("mfma_f64_16x16x4: %.1f ticks/instr independent (8 accumulators), %.1f dependent chain\n", h[0] / 256.0, h[1] / 64.0);
  printf("mixed 1 MFMA + 12 FMA64 per round: %.1f ticks/round; the 12 FMAs alone: %.1f ticks/round; MFMA -> dependent VALU read: %.1f ticks/pair\n", h[2] / 64.0, h[3] / 64.0, h[4] / 32.0);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(rate_kernel, dim3(1), dim3(320), 0, 0, o, t, 1.0); hipDeviceSynchronize(); }
  hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
  printf("two waves on one SIMD: wave 0 (256 MFMAs) %lld ticks, wave 4 (4096 FMA64) %lld ticks  [alone: %.0f and ~%.0f]\n", h[8], h[12], 256 * 1.0 * 0, 4096 * 4.0);
  return 0;
}
