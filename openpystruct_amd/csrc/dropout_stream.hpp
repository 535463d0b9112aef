// The counter-based dropout stream of every training kernel (csrc/fused_bn.hip, mlp_block.hip, seq_block.hip, seq_layer.hip):
// keep(element) = uniform(seed, call, element index) >= p.  Forward and backward launches regenerate the same mask from the same
// (seed, call) -- `call` is the device-side launch counter of csrc/call_counter.hpp -- so no mask is ever stored.
//
// r03: the first stream was three rounds of 64-bit multiply-xorshift PER ELEMENT.  gfx950 has no 64-bit integer multiplier: a 64 x 64
// product is four quarter-rate 32-bit multiplies, ~280 cycles of a SIMD per wave and element -- 18 elements per lane of the fused
// encoder layer = 3.7 of its 25 us (profiles/r03_notes.md 7).  Now the 64-bit mixing happens ONCE per (seed, call) on the scalar unit
// (both are wave-uniform) and gives two 32-bit keys; an element costs two 32-bit multiplies: a keyed variant of the two-round
// multiply-xorshift integer hash ("lowbias32" constants), key 0 folded in before the first round and key 1 (and the high index word)
// between the rounds, so that two calls' masks are not index permutations of each other.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace opsamd {

struct DropKey { uint32_t k0, k1; };

__device__ __forceinline__ DropKey drop_key(uint64_t seed, uint64_t call) {
  const uint32_t cl = __builtin_amdgcn_readfirstlane((uint32_t)call), ch = __builtin_amdgcn_readfirstlane((uint32_t)(call >> 32));
  const uint32_t sl = __builtin_amdgcn_readfirstlane((uint32_t)seed), sh = __builtin_amdgcn_readfirstlane((uint32_t)(seed >> 32));
  uint64_t z = (((uint64_t)sh << 32) | sl) + 0x9E3779B97F4A7C15ull * ((((uint64_t)ch << 32) | cl) + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return DropKey{(uint32_t)z, (uint32_t)(z >> 32)};
}

__device__ __forceinline__ float drop_uniform(DropKey k, uint64_t idx) {   // [0, 1), 24 bits
  uint32_t x = (uint32_t)idx ^ k.k0;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= k.k1 ^ (uint32_t)(idx >> 32);
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float drop_uniform(uint64_t seed, uint64_t call, uint64_t idx) { return drop_uniform(drop_key(seed, call), idx); }

}  // namespace opsamd
