// The call counter of a counter-based dropout / noise stream lives in device memory (fresh draws under HIP-graph replay).
// Every workgroup of a launch reads it at its start; it must therefore not advance while any workgroup of that launch has
// still to start -- a workgroup that read the advanced value would repeat, element for element, what the NEXT launch draws.
//   counter[0]: calls so far      counter[1]: workgroups of the running launch that are done
// One thread of every workgroup reports in when its workgroup is done; the LAST one advances the call count and clears the
// tally (launches that share a counter are ordered on one stream).  Grids here are <= 128 workgroups: the same-address
// atomics (~40 ns each) arrive spread over the launch and cost nothing measurable.
#pragma once
#include <hip/hip_runtime.h>

namespace opsamd {

__device__ __forceinline__ void call_counter_done(unsigned long long* counter, unsigned nblocks) {
  const unsigned long long done = atomicAdd(counter + 1, 1ull);
  if (done + 1ull == (unsigned long long)nblocks) {
    atomicExch(counter + 1, 0ull);
    atomicAdd(counter, 1ull);
  }
}

}  // namespace opsamd
