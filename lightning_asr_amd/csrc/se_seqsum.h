// The SE squeeze's per-utterance column sums (models/QuartNetContextSE.py:19-20: mean over T of the BN output is affine in sum_t y) as
// a device function, shared by its stand-alone kernel (se.hip) and by the BN finalize launch that runs it in its own grid (norm.hip,
// round 4: the finalize of a unit's GEMM statistics and the squeeze of the same unit's y are independent of each other).
#pragma once
#include "common.h"

namespace lasr {

// workgroup (bx, b): 64 channels x one utterance; 256 threads = CT column threads x RL row lanes, eight row loads in flight per thread,
// clamped rows (no branch around a load).  s_red: [256 / CT][65] floats.
template <typename T>
__device__ __forceinline__ void seqsum_vec_body(const T* __restrict__ x, int Tt, int C, float* __restrict__ sums, int bx, int b,
                                                float (*s_red)[65]) {
  constexpr int V = Vec<T>::kN, CT = 64 / V, RL = 256 / CT, RB = 8;
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  const int c = bx * 64 + cl * V;
  const T* xb = x + (size_t)b * Tt * C + min(c, C - V);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  for (int t0 = rl; t0 < Tt; t0 += RB * RL) {
    uint4 r[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) r[i] = Vec<T>::raw(xb + (size_t)min(t0 + i * RL, Tt - 1) * C);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      float v[V];
      Vec<T>::unpack(r[i], v);
      const bool live = t0 + i * RL < Tt;
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += live ? v[j] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) s_red[rl][cl * V + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = bx * 64 + threadIdx.x;
    if (cc < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < RL; ++r) s += s_red[r][threadIdx.x];
      sums[(size_t)b * C + cc] = s;
    }
  }
}

// norm.hip: lasr_bn_finalize_partials and lasr_seqsum(x) in ONE launch.  Returns 0 (launched), 1 (shape outside the vector form:
// nothing launched, make the two calls) or an error.
int bn_finalize_partials_seqsum(const lasr_bn_branch* branches, int n_branches, int64_t C, int64_t n_rows, float eps, float momentum,
                                const void* x, int dtype, int64_t B, int64_t T_, float* sums, void* stream);

}  // namespace lasr
