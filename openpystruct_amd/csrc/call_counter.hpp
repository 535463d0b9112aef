// The call counter of a counter-based dropout / noise stream lives in device memory (fresh draws under HIP-graph replay).
// Every workgroup of a launch reads it at its start; it must therefore not advance while any workgroup of that launch has
// still to start -- a workgroup that read the advanced value would repeat, element for element, what the NEXT launch draws.
//   counter[0]: calls so far      counter[1]: workgroups of the running launch that are done
// One thread of every workgroup THAT READS THE COUNTER reports in when its workgroup is done; the LAST one advances the call
// count and clears the tally (launches that share a counter are ordered on one stream).  `reporters` is the number of
// workgroups of the launch that report -- NOT gridDim.x when the grid also carries workgroups that leave early without
// drawing (side jobs of mlp_strip_kernel, the target gather of mlp_gather_noise_kernel): a tally that is compared with a
// count it can never reach is carried into the next launch and the stream freezes (r03 bug: every later step repeated one
// dropout mask).  `call_counter_done` is a workgroup-collective call: it barriers first, so that every wave of the
// workgroup has read counter[0] before the report that may advance it.  Grids here are <= 128 workgroups: the
// same-address atomics (~40 ns each) arrive spread over the launch and cost nothing measurable.
#pragma once
#include <hip/hip_runtime.h>

namespace opsamd {

// every thread of the workgroup calls this (workgroup-uniform control flow)
__device__ __forceinline__ void call_counter_done(unsigned long long* counter, unsigned reporters) {
#ifndef OPS_COUNTER_NO_BARRIER      /* (A/B builds only) */
  __syncthreads();
#endif
  if (threadIdx.x == 0) {
    const unsigned long long done = atomicAdd(counter + 1, 1ull);
    if (done + 1ull == (unsigned long long)reporters) {
      atomicExch(counter + 1, 0ull);
      atomicAdd(counter, 1ull);
    }
  }
}

// the same, and thread 0 of the LAST reporting workgroup gets true (it may then advance other per-launch state, e.g. a batch cursor)
__device__ __forceinline__ bool call_counter_done_last(unsigned long long* counter, unsigned reporters) {
  __syncthreads();
  bool last = false;
  if (threadIdx.x == 0) {
    const unsigned long long done = atomicAdd(counter + 1, 1ull);
    if (done + 1ull == (unsigned long long)reporters) {
      atomicExch(counter + 1, 0ull);
      atomicAdd(counter, 1ull);
      last = true;
    }
  }
  return last;
}

}  // namespace opsamd
