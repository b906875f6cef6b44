// Training-mode BatchNorm coefficients from the column sums of a pre-BN tensor: shared by the stand-alone finalize kernels
// (norm.hip) and the GEMM epilogue that finalises in its last-arriving workgroup (gemm_bf16.hip).
#pragma once
#include "common.h"

namespace lasr {

__device__ __forceinline__ void bn_finalize_channel(float s, float q, int64_t c, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ rmean,
                                                    float* __restrict__ rvar, float* __restrict__ coef, float* __restrict__ saved,
                                                    int64_t C, float n, float eps, float momentum, int training) {
  float mean, var;
  if (training) {
    // sums arrive in f32; the subtraction is done in double to keep E[x^2]-E[x]^2 benign
    const double m = (double)s / n;
    double v = (double)q / n - m * m;
    if (v < 0) v = 0;
    mean = (float)m;
    var = (float)v;
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      const float unbiased = n > 1.f ? (float)(v * (double)n / ((double)n - 1.0)) : var;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float a = gamma[c] * rstd;
  coef[c] = a;
  coef[C + c] = beta[c] - mean * a;
  if (saved) { saved[c] = mean; saved[C + c] = rstd; }
}

// What the last-arriving workgroup of a column tile needs to turn the per-tile partial sums into BN coefficients.
// ticket: one zero-initialised counter per column tile (the finishing workgroup resets it), or null = no fused finalize.
struct BnFinal {
  const float* gamma; const float* beta; float* rmean; float* rvar; float* coef; float* saved; float* stats;
  unsigned* ticket;
  float n, eps, momentum;
};

}  // namespace lasr
