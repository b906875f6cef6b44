// nn.Dropout(p) of SeprationConv / last_cnn2 (models/QuartNet.py:26,38,149) as a counter-based mask: no mask tensor is ever
// stored - forward and both backward passes regenerate it from (seed, training-step counter, unit, element index).
//
//   keep(e) = u16(e) >= round(p * 65536),   u16(e) = 16 bits of Philox4x32-10(key = seed, counter = (e / 8, unit, step_lo, step_hi))
//
// one Philox call serves 8 consecutive elements of the row-major [N][C] tensor (the 16-byte bf16 vector a lane handles).
// Kept elements are scaled by 1/(1-p) (inverted dropout, as torch).  The step counter is a DEVICE scalar bumped by the first
// launch of every training forward, so a step replayed from a captured hipGraph still draws a fresh mask.
// torch's own mask stream cannot be reproduced (it is an implementation detail of its Philox offset bookkeeping): parity is
// checked with the mask read back through lasr_dropout_mask and applied inside the oracle.
#pragma once
#include "common.h"

namespace lasr {

struct DropArgs {
  const unsigned long long* step;   // device scalar: index of the current training forward; null = dropout off
  unsigned long long seed;
  uint32_t unit;                    // which unit of the plan (distinct masks per layer)
  uint32_t thresh;                  // round(p * 65536): drop when u16 < thresh
  float inv_keep;                   // 1 / (1 - p)
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep-mask bits of the 8 elements [8*vec, 8*vec + 8): bit i set = element kept
__device__ __forceinline__ uint32_t drop_keep8(const DropArgs& d, unsigned long long step, uint32_t vec) {
  uint32_t r[4];
  philox4x32_10(vec, d.unit, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)d.seed, (uint32_t)(d.seed >> 32), r);
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t u = (r[i >> 1] >> (16 * (i & 1))) & 0xffffu;
    m |= (u >= d.thresh ? 1u : 0u) << i;
  }
  return m;
}

// scale factors of the V (4 or 8) consecutive elements starting at element index e0 (a multiple of V): 0 or 1/(1-p)
template <int V>
__device__ __forceinline__ void drop_scale(const DropArgs& d, unsigned long long step, uint32_t e0, float (&s)[V]) {
  const uint32_t m = drop_keep8(d, step, e0 >> 3) >> (e0 & 7u);   // V = 4: the low or the high half of the call's 8 elements
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = ((m >> j) & 1u) ? d.inv_keep : 0.f;
}

}  // namespace lasr
