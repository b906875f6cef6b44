// out[i] = sum_p partials[p * n + i] - the element body of lasr_reduce_many (norm.hip), shared with the weight-gradient launch that
// runs a stage's ALREADY-COMPLETE reductions on the CUs its tiles leave idle (gemm_bf16.hip, round 5).  f64 accumulation in a fixed
// order, 16 loads in flight: the same numbers whichever kernel does it.
#pragma once
#include "common.h"

namespace lasr {

__device__ __forceinline__ void reduce_many_elem(const lasr_reduce_desc& q, int64_t i) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int p = 0; p < q.n_partials; p += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float x = q.partials[(int64_t)min(p + u, q.n_partials - 1) * q.n + i];
      v[u] = p + u < q.n_partials ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; u += 4) { a0 += (double)v[u]; a1 += (double)v[u + 1]; a2 += (double)v[u + 2]; a3 += (double)v[u + 3]; }
  }
  q.out[i] = (float)((a0 + a1) + (a2 + a3));
}

}  // namespace lasr
