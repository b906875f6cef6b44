// Layout, masking-length, BatchNorm (training statistics) and fused BN-apply/residual/activation
// kernels, forward and backward, over channels-last [B][T][C] activations.
// Replaces models/QuartNet.py:33-37 (MaskCNN lengths, BatchNorm1d(eps=1e-3), ReLU) and :74-77
// (residual add + ReLU), plus their autograd backward.  All statistics are f32.
#include "common.h"

namespace lasr {

// ------------------------------------------------------------------ layout ------------------
// (B, C, T) f32 <-> [B][T][C] T via a 32x32 LDS tile.
template <typename T, bool TO_BTC>
__global__ __launch_bounds__(256) void transpose_kernel(const void* __restrict__ in_, void* __restrict__ out_,
                                                        int64_t C, int64_t Tt) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.y * 32, t0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (TO_BTC) {
    const float* in = reinterpret_cast<const float*>(in_) + (int64_t)b * C * Tt;
    T* out = reinterpret_cast<T*>(out_) + (int64_t)b * C * Tt;
    for (int i = ty; i < 32; i += 8)
      if (c0 + i < C && t0 + tx < Tt) tile[i][tx] = in[(c0 + i) * Tt + t0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
      if (t0 + i < Tt && c0 + tx < C) Elem<T>::st(out + (t0 + i) * C + c0 + tx, tile[tx][i]);
  } else {
    const T* in = reinterpret_cast<const T*>(in_) + (int64_t)b * C * Tt;
    float* out = reinterpret_cast<float*>(out_) + (int64_t)b * C * Tt;
    for (int i = ty; i < 32; i += 8)
      if (t0 + i < Tt && c0 + tx < C) tile[i][tx] = Elem<T>::ld(in + (t0 + i) * C + c0 + tx);
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
      if (c0 + i < C && t0 + tx < Tt) out[(c0 + i) * Tt + t0 + tx] = tile[tx][i];
  }
}

__global__ void mask_lengths_kernel(const float* __restrict__ pct, int64_t B, float Tf, int32_t* __restrict__ lens) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < B) lens[i] = (int32_t)(Tf * pct[i]);  // f32 product, truncation toward zero (torch .int())
}

// ------------------------------------------------------------------ BN finalize --------------
__global__ void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float* __restrict__ coef, float* __restrict__ saved, int64_t C, float n, float eps,
                                   float momentum, int training) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    // sums arrive in f32; the subtraction is done in double to keep E[x^2]-E[x]^2 benign
    const double m = (double)stats[c] / n;
    double v = (double)stats[C + c] / n - m * m;
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

// ------------------------------------------------------------------ BN apply + add + act -----
// vectorised by 4 channels; C % 4 == 0.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ coef,
                                                         const T* __restrict__ y2, const float* __restrict__ coef2,
                                                         const float* __restrict__ se, T* __restrict__ out,
                                                         int64_t rows, int64_t Tt, int64_t C, int act) {
  const int64_t c4n = C >> 2;
  const int64_t total = rows * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c4n;
    const int64_t c = (i - row * c4n) << 2;
    float v[4], r[4], o[4];
    Elem<T>::ld4(y + row * C + c, v);
    const float4 a = *reinterpret_cast<const float4*>(coef + c);
    const float4 bb = *reinterpret_cast<const float4*>(coef + C + c);
    o[0] = fmaf(v[0], a.x, bb.x); o[1] = fmaf(v[1], a.y, bb.y); o[2] = fmaf(v[2], a.z, bb.z); o[3] = fmaf(v[3], a.w, bb.w);
    if (se) {
      const float4 s = *reinterpret_cast<const float4*>(se + (row / Tt) * C + c);
      o[0] *= s.x; o[1] *= s.y; o[2] *= s.z; o[3] *= s.w;
    }
    if (y2) {
      Elem<T>::ld4(y2 + row * C + c, r);
      const float4 a2 = *reinterpret_cast<const float4*>(coef2 + c);
      const float4 b2 = *reinterpret_cast<const float4*>(coef2 + C + c);
      o[0] += fmaf(r[0], a2.x, b2.x); o[1] += fmaf(r[1], a2.y, b2.y);
      o[2] += fmaf(r[2], a2.z, b2.z); o[3] += fmaf(r[3], a2.w, b2.w);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = act_fwd(o[j], act);
    Elem<T>::st4(out + row * C + c, o);
  }
}

// d(act)/d(pre-activation)
__device__ __forceinline__ float act_grad(float p, int act) {
  if (act == LASR_ACT_RELU) return p > 0.f ? 1.f : 0.f;
  if (act == LASR_ACT_SWISH) {
    const float s = 1.f / (1.f + __expf(-p));
    return s * (1.f + p * (1.f - s));
  }
  return 1.f;
}

// Shared per-element backward math: returns d1 (grad into branch-1 BN output), d2 (branch 2), yhat1, yhat2
struct BwdElem { float d1, d2, h1, h2; };
__device__ __forceinline__ BwdElem bwd_elem(float dout, float y1, float a1, float b1, float mu1, float rs1, float se,
                                            float seg, bool has2, float y2, float a2, float b2, float mu2, float rs2,
                                            int act) {
  const float z1 = fmaf(y1, a1, b1) * se;
  const float z2 = has2 ? fmaf(y2, a2, b2) : 0.f;
  const float d = dout * act_grad(z1 + z2, act);
  BwdElem e;
  e.d1 = fmaf(d, se, seg);
  e.d2 = d;
  e.h1 = (y1 - mu1) * rs1;
  e.h2 = has2 ? (y2 - mu2) * rs2 : 0.f;
  return e;
}

static constexpr int kRowsPerBlock = 64;

// pass 1: per-block partial sums -> partials[blk][4][C]  (s1, s2, s1', s2')
// block = 256 threads: threads split as (C/4 column-vectors) x (row lanes)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                           const float* __restrict__ coef, const float* __restrict__ saved,
                                                           const T* __restrict__ y2, const float* __restrict__ coef2,
                                                           const float* __restrict__ saved2, const float* __restrict__ se,
                                                           const float* __restrict__ seg, float* __restrict__ partials,
                                                           int64_t rows, int64_t Tt, int64_t C, int act) {
  extern __shared__ __attribute__((aligned(16))) float s_part[];  // [row_lanes][4][C]
  const int c4n = (int)(C >> 2);
  const int col_threads = c4n < 256 ? c4n : 256;
  const int row_lanes = 256 / col_threads;
  const int cl = threadIdx.x % col_threads, rl = threadIdx.x / col_threads;
  const int64_t r0 = (int64_t)blockIdx.x * kRowsPerBlock;
  const int64_t r1 = r0 + kRowsPerBlock < rows ? r0 + kRowsPerBlock : rows;
  const bool has2 = y2 != nullptr;
  for (int cv = cl; cv < c4n; cv += col_threads) {
    const int c = cv << 2;
    float a1[4], b1[4], m1[4], q1[4], a2[4] = {0, 0, 0, 0}, b2[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a1[j] = coef[c + j]; b1[j] = coef[C + c + j]; m1[j] = saved[c + j]; q1[j] = saved[C + c + j];
      if (has2) { a2[j] = coef2[c + j]; b2[j] = coef2[C + c + j]; m2[j] = saved2[c + j]; q2[j] = saved2[C + c + j]; }
    }
    float acc[4][4] = {};
    if (rl < row_lanes) {
      for (int64_t r = r0 + rl; r < r1; r += row_lanes) {
        float dv[4], yv[4], rv[4] = {0, 0, 0, 0}, sv[4] = {1, 1, 1, 1}, gv[4] = {0, 0, 0, 0};
        Elem<T>::ld4(dout + r * C + c, dv);
        Elem<T>::ld4(y + r * C + c, yv);
        if (has2) Elem<T>::ld4(y2 + r * C + c, rv);
        if (se) {
          const int64_t b = r / Tt;
#pragma unroll
          for (int j = 0; j < 4; ++j) { sv[j] = se[b * C + c + j]; gv[j] = seg ? seg[b * C + c + j] : 0.f; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          BwdElem e = bwd_elem(dv[j], yv[j], a1[j], b1[j], m1[j], q1[j], sv[j], gv[j], has2, rv[j], a2[j], b2[j], m2[j], q2[j], act);
          acc[0][j] += e.d1; acc[1][j] = fmaf(e.d1, e.h1, acc[1][j]);
          acc[2][j] += e.d2; acc[3][j] = fmaf(e.d2, e.h2, acc[3][j]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) s_part[((int64_t)rl * 4 + k) * C + c + j] = acc[k][j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C; i += 256) {
    float s = 0.f;
    for (int l = 0; l < row_lanes; ++l) s += s_part[(int64_t)l * 4 * C + i];
    partials[(int64_t)blockIdx.x * 4 * C + i] = s;
  }
}

// sums[k][c] = sum over blocks of partials[blk][k][c]; k<2 -> sums, k>=2 -> sums2
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ partials, int nblk, int64_t C,
                                                            float* __restrict__ sums, float* __restrict__ sums2) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= 4 * C) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)partials[(int64_t)b * 4 * C + i];
  if (i < 2 * C) sums[i] = (float)s;
  else if (sums2) sums2[i - 2 * C] = (float)s;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                           const float* __restrict__ coef, const float* __restrict__ saved,
                                                           const float* __restrict__ gamma, const T* __restrict__ y2,
                                                           const float* __restrict__ coef2, const float* __restrict__ saved2,
                                                           const float* __restrict__ gamma2, const float* __restrict__ se,
                                                           const float* __restrict__ seg, const float* __restrict__ sums,
                                                           const float* __restrict__ sums2, const int32_t* __restrict__ row_lens,
                                                           T* __restrict__ dy, T* __restrict__ dy2, int64_t rows, int64_t Tt,
                                                           int64_t C, int act) {
  const int64_t c4n = C >> 2;
  const int64_t total = rows * c4n;
  const float inv_n = 1.0f / (float)rows;
  const bool has2 = y2 != nullptr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c4n;
    const int64_t c = (i - row * c4n) << 2;
    const int64_t b = row / Tt;
    const bool masked = row_lens && (row - b * Tt) >= row_lens[b];
    float dv[4], yv[4], rv[4] = {0, 0, 0, 0}, o1[4], o2[4];
    Elem<T>::ld4(dout + row * C + c, dv);
    Elem<T>::ld4(y + row * C + c, yv);
    if (has2) Elem<T>::ld4(y2 + row * C + c, rv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sej = se ? se[b * C + c + j] : 1.f;
      const float sgj = (se && seg) ? seg[b * C + c + j] : 0.f;
      BwdElem e = bwd_elem(dv[j], yv[j], coef[c + j], coef[C + c + j], saved[c + j], saved[C + c + j], sej, sgj, has2,
                           rv[j], has2 ? coef2[c + j] : 0.f, has2 ? coef2[C + c + j] : 0.f, has2 ? saved2[c + j] : 0.f,
                           has2 ? saved2[C + c + j] : 0.f, act);
      const float g1 = gamma[c + j] * saved[C + c + j];
      o1[j] = masked ? 0.f : g1 * (e.d1 - sums[c + j] * inv_n - e.h1 * sums[C + c + j] * inv_n);
      if (has2) {
        const float g2 = gamma2[c + j] * saved2[C + c + j];
        o2[j] = g2 * (e.d2 - sums2[c + j] * inv_n - e.h2 * sums2[C + c + j] * inv_n);
      }
    }
    Elem<T>::st4(dy + row * C + c, o1);
    if (has2) Elem<T>::st4(dy2 + row * C + c, o2);
  }
}

__global__ void bn_param_grad_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t C) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = sums[c];
  if (dgamma) dgamma[c] = sums[C + c];
}

static inline int ew_grid(int64_t total) {
  int64_t g = cdiv(total, 256);
  return (int)(g > 256 * 8 ? 256 * 8 : (g < 1 ? 1 : g));
}

}  // namespace lasr

using namespace lasr;

#define DISPATCH_DTYPE(dtype, ...)                      \
  if ((dtype) == LASR_F32) { using T = float; __VA_ARGS__; } \
  else { using T = bf16_t; __VA_ARGS__; }

extern "C" int lasr_bct_to_btc(const float* in, void* out, int dtype, int64_t B, int64_t C, int64_t T_, void* stream) {
  LASR_CHECK_ARG(in && out && (dtype == LASR_F32 || dtype == LASR_BF16), "lasr_bct_to_btc: bad argument");
  LASR_CHECK_SHAPE(B > 0 && C > 0 && T_ > 0 && B < 65536, "lasr_bct_to_btc: shape");
  dim3 grid((unsigned)cdiv(T_, 32), (unsigned)cdiv(C, 32), (unsigned)B);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_kernel<T, true>), grid, dim3(256), 0, as_stream(stream), in, out, C, T_));
  LASR_LAUNCH_CHECK("transpose_kernel");
  return 0;
}
extern "C" int lasr_btc_to_bct(const void* in, int dtype, float* out, int64_t B, int64_t C, int64_t T_, void* stream) {
  LASR_CHECK_ARG(in && out && (dtype == LASR_F32 || dtype == LASR_BF16), "lasr_btc_to_bct: bad argument");
  LASR_CHECK_SHAPE(B > 0 && C > 0 && T_ > 0 && B < 65536, "lasr_btc_to_bct: shape");
  dim3 grid((unsigned)cdiv(T_, 32), (unsigned)cdiv(C, 32), (unsigned)B);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_kernel<T, false>), grid, dim3(256), 0, as_stream(stream), in, out, C, T_));
  LASR_LAUNCH_CHECK("transpose_kernel");
  return 0;
}

extern "C" int lasr_mask_lengths(const float* pct, int64_t B, int64_t T_, int32_t* lens, void* stream) {
  LASR_CHECK_ARG(pct && lens && B > 0 && T_ > 0, "lasr_mask_lengths: bad argument");
  hipLaunchKernelGGL(mask_lengths_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, as_stream(stream), pct, B, (float)T_, lens);
  LASR_LAUNCH_CHECK("mask_lengths_kernel");
  return 0;
}

extern "C" int lasr_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float* coef, float* saved, int64_t C, int64_t n_rows, float eps,
                                float momentum, int training, void* stream) {
  LASR_CHECK_ARG(gamma && beta && coef && C > 0 && n_rows > 0, "lasr_bn_finalize: bad argument");
  LASR_CHECK_ARG(training ? stats != nullptr : (running_mean && running_var), "lasr_bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream), stats, gamma, beta,
                     running_mean, running_var, coef, saved, C, (float)n_rows, eps, momentum, training);
  LASR_LAUNCH_CHECK("bn_finalize_kernel");
  return 0;
}

extern "C" int lasr_bn_act_fwd(const void* y, const float* coef, const void* y2, const float* coef2, const float* se_scale,
                               void* out, int dtype, int64_t B, int64_t T_, int64_t C, int act, void* stream) {
  LASR_CHECK_ARG(y && coef && out && (!y2 || coef2), "lasr_bn_act_fwd: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_bn_act_fwd: bad dtype");
  LASR_CHECK_SHAPE(C % 4 == 0 && B > 0 && T_ > 0, "lasr_bn_act_fwd: C=%lld must be a multiple of 4", (long long)C);
  const int64_t rows = B * T_;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, as_stream(stream),
                                           (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, rows, T_, C, act));
  LASR_LAUNCH_CHECK("bn_act_fwd_kernel");
  return 0;
}

extern "C" size_t lasr_bn_bwd_workspace_bytes(int64_t B, int64_t T_, int64_t C) {
  return (size_t)cdiv(B * T_, kRowsPerBlock) * 4 * C * sizeof(float);
}

extern "C" int lasr_bn_act_bwd_stats(const void* dout, const void* y, const float* coef, const float* saved, const void* y2,
                                     const float* coef2, const float* saved2, const float* se_scale, const float* se_grad,
                                     float* sums, float* sums2, int dtype, int64_t B, int64_t T_, int64_t C, int act,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(dout && y && coef && saved && sums && workspace, "lasr_bn_act_bwd_stats: null pointer");
  LASR_CHECK_ARG(!y2 || (coef2 && saved2 && sums2), "lasr_bn_act_bwd_stats: branch-2 pointers");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_bn_act_bwd_stats: bad dtype");
  LASR_CHECK_SHAPE(C % 4 == 0 && C <= 4096 && B > 0 && T_ > 0, "lasr_bn_act_bwd_stats: C=%lld", (long long)C);
  const int64_t rows = B * T_;
  const int nblk = (int)cdiv(rows, kRowsPerBlock);
  if (workspace_bytes < lasr_bn_bwd_workspace_bytes(B, T_, C)) return fail(LASR_E_WORKSPACE, "lasr_bn_act_bwd_stats: workspace");
  const int c4n = (int)(C / 4);
  const int col_threads = c4n < 256 ? c4n : 256;
  const int row_lanes = 256 / col_threads;
  const size_t shmem = (size_t)row_lanes * 4 * C * sizeof(float);
  LASR_CHECK_SHAPE(shmem <= 64 * 1024, "lasr_bn_act_bwd_stats: C too large for LDS staging");
  float* partials = reinterpret_cast<float*>(workspace);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_bwd_stats_kernel<T>, dim3(nblk), dim3(256), shmem, as_stream(stream), (const T*)dout,
                                           (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad, partials,
                                           rows, T_, C, act));
  LASR_LAUNCH_CHECK("bn_bwd_stats_kernel");
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)cdiv(4 * C, 256)), dim3(256), 0, as_stream(stream), partials, nblk, C,
                     sums, sums2);
  LASR_LAUNCH_CHECK("bn_bwd_reduce_kernel");
  return 0;
}

extern "C" int lasr_bn_act_bwd_apply(const void* dout, const void* y, const float* coef, const float* saved,
                                     const float* gamma, const void* y2, const float* coef2, const float* saved2,
                                     const float* gamma2, const float* se_scale, const float* se_grad, const float* sums,
                                     const float* sums2, const int32_t* row_lens, void* dy, void* dy2, float* dgamma,
                                     float* dbeta, float* dgamma2, float* dbeta2, int dtype, int64_t B, int64_t T_, int64_t C,
                                     int act, void* stream) {
  LASR_CHECK_ARG(dout && y && coef && saved && gamma && sums && dy, "lasr_bn_act_bwd_apply: null pointer");
  LASR_CHECK_ARG(!y2 || (coef2 && saved2 && gamma2 && sums2 && dy2), "lasr_bn_act_bwd_apply: branch-2 pointers");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_bn_act_bwd_apply: bad dtype");
  LASR_CHECK_SHAPE(C % 4 == 0 && B > 0 && T_ > 0, "lasr_bn_act_bwd_apply: C=%lld", (long long)C);
  const int64_t rows = B * T_;
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, as_stream(stream),
                                           (const T*)dout, (const T*)y, coef, saved, gamma, (const T*)y2, coef2, saved2, gamma2,
                                           se_scale, se_grad, sums, sums2, row_lens, (T*)dy, (T*)dy2, rows, T_, C, act));
  LASR_LAUNCH_CHECK("bn_bwd_apply_kernel");
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream), sums, dgamma, dbeta, C);
    LASR_LAUNCH_CHECK("bn_param_grad_kernel");
  }
  if (y2 && (dgamma2 || dbeta2)) {
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream), sums2, dgamma2, dbeta2, C);
    LASR_LAUNCH_CHECK("bn_param_grad_kernel");
  }
  return 0;
}

// ------------------------------------------------------------------ small reductions ----------
namespace lasr {
static constexpr int kColsumRows = 256;
// partials[blk][c] = sum over a 256-row slab; block = 256 threads walking columns
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t rows, int64_t C,
                                                             float* __restrict__ partials) {
  const int64_t r0 = (int64_t)blockIdx.x * kColsumRows;
  const int64_t r1 = r0 + kColsumRows < rows ? r0 + kColsumRows : rows;
  for (int64_t c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += x[r * C + c];
    partials[(int64_t)blockIdx.x * C + c] = s;
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partials, int nblk, int64_t C,
                                                           float* __restrict__ out) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)partials[(int64_t)b * C + c];
  out[c] = (float)s;
}
// out[0] = scale * sum(x[0..n)), one wave, fixed order (n is the batch size)
__global__ __launch_bounds__(64) void scale_sum_kernel(const float* __restrict__ x, int64_t n, float scale, float* __restrict__ out) {
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 64) s += x[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}
}  // namespace lasr

extern "C" size_t lasr_colsum_workspace_bytes(int64_t rows, int64_t C) {
  return (size_t)cdiv(rows, kColsumRows) * C * sizeof(float);
}
extern "C" int lasr_colsum_f32(const float* x, float* out, int64_t rows, int64_t C, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(x && out && workspace && rows > 0 && C > 0, "lasr_colsum_f32: bad argument");
  if (workspace_bytes < lasr_colsum_workspace_bytes(rows, C)) return fail(LASR_E_WORKSPACE, "lasr_colsum_f32: workspace");
  const int nblk = (int)cdiv(rows, kColsumRows);
  float* partials = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), x, rows, C, partials);
  LASR_LAUNCH_CHECK("colsum_partial_kernel");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream), partials, nblk, C, out);
  LASR_LAUNCH_CHECK("colsum_final_kernel");
  return 0;
}
extern "C" int lasr_scale_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream) {
  LASR_CHECK_ARG(x && out && n > 0, "lasr_scale_sum_f32: bad argument");
  hipLaunchKernelGGL(scale_sum_kernel, dim3(1), dim3(64), 0, as_stream(stream), x, n, scale, out);
  LASR_LAUNCH_CHECK("scale_sum_kernel");
  return 0;
}
