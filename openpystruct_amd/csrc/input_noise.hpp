// The input-noise stream of the batch assembly (PINN:743-756: x + sigma * N(0, 1)): element i of call `call` draws from a splitmix64
// hash of (seed, call, i) -- shared by csrc/input_prep.hip (the assembly launch) and csrc/seq_layer.hip (the TFD front end assembling its
// own batch, r04): the two must give the same numbers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace opsamd {

__device__ __forceinline__ uint64_t ip_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float ip_noisy(float v, float sg, unsigned long long seed, unsigned long long call, long i) {
  if (sg == 0.0f) return v;
  const uint64_t h = ip_mix(seed + 0x9E3779B97F4A7C15ull * (call + 1) + (uint64_t)i * 0xD1B54A32D192ED03ull);
  const float u1 = ((float)(h >> 40) + 1.0f) * (1.0f / 16777216.0f);          // (0, 1]
  const float u2 = (float)((h >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);   // [0, 1)
  return v + sg * sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
}

}  // namespace opsamd
