// Squeeze-excite (reduction 8, no bias) around the BatchNorm output of every separable conv of the
// ContextSE variant: s = sigmoid(W2 relu(W1 mean_T(BN(y)))), applied after BN and before the
// activation (models/QuartNetContextSE.py:8-23,55).  The mean runs over ALL T' frames, padding
// included.  Because BN is affine per channel, mean_T(BN(y)) = a_c * mean_T(y) + b_c, so the
// squeeze needs only per-(utterance, channel) sums of the pre-BN tensor: one extra read of y in
// forward; the excite scale is folded into lasr_bn_act_fwd / lasr_bn_act_bwd_*.
#include "common.h"

namespace lasr {

// sums[b][c] = sum_t x[b][t][c].  grid (ceil(C/64), B), block 256 = 16 channel lanes (x4) x 16 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void seqsum_kernel(const T* __restrict__ x, int64_t Tt, int64_t C, float* __restrict__ sums) {
  __shared__ float s_red[16][65];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const T* xb = x + (int64_t)b * Tt * C + c;
    for (int64_t t = rl; t < Tt; t += 16) {
      float v[4];
      Elem<T>::ld4(xb + t * C, v);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) s_red[rl][cl * 4 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int64_t cc = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += s_red[r][threadIdx.x];
      sums[(int64_t)b * C + cc] = s;
    }
  }
}

// ds[b][c] = sum_t dout * act'(z) * (a1*y + b1),  z = (a1*y+b1)*se + (a2*y2+b2): gradient w.r.t. the scale
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                            const float* __restrict__ coef, const T* __restrict__ y2,
                                                            const float* __restrict__ coef2, const float* __restrict__ se,
                                                            int64_t Tt, int64_t C, int act, float* __restrict__ ds) {
  __shared__ float s_red[16][65];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    float a1[4], b1[4], a2[4], b2[4], sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a1[j] = coef[c + j]; b1[j] = coef[C + c + j]; sc[j] = se[(int64_t)b * C + c + j];
      a2[j] = y2 ? coef2[c + j] : 0.f; b2[j] = y2 ? coef2[C + c + j] : 0.f;
    }
    const int64_t base = (int64_t)b * Tt * C + c;
    for (int64_t t = rl; t < Tt; t += 16) {
      float dv[4], yv[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
      Elem<T>::ld4(dout + base + t * C, dv);
      Elem<T>::ld4(y + base + t * C, yv);
      if (y2) Elem<T>::ld4(y2 + base + t * C, rv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z1 = fmaf(yv[j], a1[j], b1[j]);
        const float z = z1 * sc[j] + (y2 ? fmaf(rv[j], a2[j], b2[j]) : 0.f);
        float g = 1.f;
        if (act == LASR_ACT_RELU) g = z > 0.f ? 1.f : 0.f;
        else if (act == LASR_ACT_SWISH) { const float s = 1.f / (1.f + __expf(-z)); g = s * (1.f + z * (1.f - s)); }
        acc[j] = fmaf(dv[j] * g, z1, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) s_red[rl][cl * 4 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int64_t cc = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += s_red[r][threadIdx.x];
      ds[(int64_t)b * C + cc] = s;
    }
  }
}

// One workgroup per utterance.  pooled = a*sums/T + b ; hidden = relu(W1 pooled) ; scale = sigmoid(W2 hidden)
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ sums, const float* __restrict__ coef,
                                                         const float* __restrict__ W1, const float* __restrict__ W2, int C, int H,
                                                         float inv_T, float* __restrict__ pooled, float* __restrict__ hidden,
                                                         float* __restrict__ scale) {
  extern __shared__ float sm[];  // pooled[C] | hidden[H]
  float* s_p = sm;
  float* s_h = sm + C;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float p = fmaf(coef[c], sums[(int64_t)b * C + c] * inv_T, coef[C + c]);
    s_p[c] = p;
    pooled[(int64_t)b * C + c] = p;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int h = wid; h < H; h += 4) {  // one wave per hidden unit: dot over C
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(W1[(int64_t)h * C + c], s_p[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = fmaxf(acc, 0.f);
      s_h[h] = v;
      hidden[(int64_t)b * H + h] = v;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int h = 0; h < H; ++h) acc = fmaf(W2[(int64_t)c * H + h], s_h[h], acc);
    scale[(int64_t)b * C + c] = 1.f / (1.f + expf(-acc));
  }
}

// One workgroup per utterance: back through sigmoid, W2, relu, W1, the mean over T.
// Writes seg[b][c] = d(loss)/d(BN output z1[b,t,c]) through the pooled path (same for every t), and the
// per-utterance weight-gradient contributions pW1[b][H][C], pW2[b][C][H] (summed over b afterwards).
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ scale,
                                                         const float* __restrict__ hidden, const float* __restrict__ pooled,
                                                         const float* __restrict__ W1, const float* __restrict__ W2, int C, int H,
                                                         float inv_T, float* __restrict__ seg, float* __restrict__ pW1,
                                                         float* __restrict__ pW2) {
  extern __shared__ float sm[];  // dpre2[C] | dh[H] | hid[H]
  float* s_d2 = sm;
  float* s_dh = sm + C;
  float* s_hid = s_dh + H;
  const int b = blockIdx.x;
  for (int h = threadIdx.x; h < H; h += 256) s_hid[h] = hidden[(int64_t)b * H + h];
  for (int c = threadIdx.x; c < C; c += 256) {
    const float s = scale[(int64_t)b * C + c];
    s_d2[c] = ds[(int64_t)b * C + c] * s * (1.f - s);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * H; i += 256) {  // dW2[c][h] = dpre2[c] * hidden[h]
    const int c = i / H, h = i - c * H;
    pW2[(int64_t)b * C * H + i] = s_d2[c] * s_hid[h];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int h = wid; h < H; h += 4) {  // dhidden[h] = relu'(.) * sum_c W2[c][h] dpre2[c]
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(W2[(int64_t)c * H + h], s_d2[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) s_dh[h] = s_hid[h] > 0.f ? acc : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * C; i += 256) {  // dW1[h][c] = dhidden[h] * pooled[c]
    const int h = i / C, c = i - h * C;
    pW1[(int64_t)b * H * C + i] = s_dh[h] * pooled[(int64_t)b * C + c];
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int h = 0; h < H; ++h) acc = fmaf(W1[(int64_t)h * C + c], s_dh[h], acc);
    seg[(int64_t)b * C + c] = acc * inv_T;
  }
}

}  // namespace lasr

using namespace lasr;

extern "C" int lasr_seqsum(const void* x, int dtype, int64_t B, int64_t T_, int64_t C, float* sums, void* stream) {
  LASR_CHECK_ARG(x && sums && (dtype == LASR_F32 || dtype == LASR_BF16), "lasr_seqsum: bad argument");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && C > 0 && C % 4 == 0, "lasr_seqsum: shape");
  dim3 grid((unsigned)cdiv(C, 64), (unsigned)B);
  if (dtype == LASR_F32) hipLaunchKernelGGL(seqsum_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, T_, C, sums);
  else hipLaunchKernelGGL(seqsum_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x, T_, C, sums);
  LASR_LAUNCH_CHECK("seqsum_kernel");
  return 0;
}

extern "C" int lasr_se_fwd(const float* sums, const float* coef, const float* W1, const float* W2, int64_t B, int64_t T_, int64_t C,
                           float* pooled, float* hidden, float* scale, void* stream) {
  LASR_CHECK_ARG(sums && coef && W1 && W2 && pooled && hidden && scale, "lasr_se_fwd: null pointer");
  LASR_CHECK_SHAPE(B > 0 && T_ > 0 && C >= 8 && C % 8 == 0 && C <= 8192, "lasr_se_fwd: C=%lld", (long long)C);
  const int H = (int)(C / 8);
  hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3((unsigned)B), dim3(256), (size_t)(C + H) * sizeof(float), as_stream(stream), sums, coef, W1,
                     W2, (int)C, H, 1.0f / (float)T_, pooled, hidden, scale);
  LASR_LAUNCH_CHECK("se_mlp_fwd_kernel");
  return 0;
}

extern "C" size_t lasr_se_bwd_workspace_bytes(int64_t B, int64_t C) {
  return align_up((size_t)B * C * sizeof(float), 256) + 2 * align_up((size_t)B * C * (C / 8) * sizeof(float), 256);
}

extern "C" int lasr_se_bwd(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                           const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T_,
                           int64_t C, int act, float* seg, float* dW1, float* dW2, void* workspace, size_t workspace_bytes,
                           void* stream) {
  LASR_CHECK_ARG(dout && y && coef && scale && hidden && pooled && W1 && W2 && seg && dW1 && dW2 && workspace, "lasr_se_bwd: null pointer");
  LASR_CHECK_ARG(!y2 || coef2, "lasr_se_bwd: residual coefficients");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_se_bwd: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && C >= 8 && C % 8 == 0 && C <= 8192, "lasr_se_bwd: C=%lld", (long long)C);
  if (workspace_bytes < lasr_se_bwd_workspace_bytes(B, C)) return fail(LASR_E_WORKSPACE, "lasr_se_bwd: workspace");
  const int H = (int)(C / 8);
  char* w = reinterpret_cast<char*>(workspace);
  float* ds = reinterpret_cast<float*>(w);
  float* pW1 = reinterpret_cast<float*>(w + align_up((size_t)B * C * sizeof(float), 256));
  float* pW2 = reinterpret_cast<float*>(w + align_up((size_t)B * C * sizeof(float), 256) + align_up((size_t)B * C * H * sizeof(float), 256));
  dim3 grid((unsigned)cdiv(C, 64), (unsigned)B);
  hipStream_t st = as_stream(stream);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(se_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)y, coef, (const float*)y2,
                       coef2, scale, T_, C, act, ds);
  else
    hipLaunchKernelGGL(se_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)y, coef,
                       (const bf16_t*)y2, coef2, scale, T_, C, act, ds);
  LASR_LAUNCH_CHECK("se_bwd_reduce_kernel");
  hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3((unsigned)B), dim3(256), (size_t)(C + 2 * H) * sizeof(float), st, ds, scale, hidden, pooled, W1,
                     W2, (int)C, H, 1.0f / (float)T_, seg, pW1, pW2);
  LASR_LAUNCH_CHECK("se_mlp_bwd_kernel");
  LASR_TRY(launch_reduce_partials(pW1, (int)B, (int64_t)H * C, dW1, (int64_t)H * C, nullptr, st));
  return launch_reduce_partials(pW2, (int)B, (int64_t)C * H, dW2, (int64_t)C * H, nullptr, st);
}
