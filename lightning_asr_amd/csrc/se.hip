// Squeeze-excite (reduction 8, no bias) around the BatchNorm output of every separable conv of the
// ContextSE variant: s = sigmoid(W2 relu(W1 mean_T(BN(y)))), applied after BN and before the
// activation (models/QuartNetContextSE.py:8-23,55).  The mean runs over ALL T' frames, padding
// included.  Because BN is affine per channel, mean_T(BN(y)) = a_c * mean_T(y) + b_c, so the
// squeeze needs only per-(utterance, channel) sums of the pre-BN tensor: one extra read of y in
// forward; the excite scale is folded into lasr_bn_act_fwd / lasr_bn_act_bwd_*.
#include "common.h"
#include "dropout.h"

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
                                                            int64_t Tt, int64_t C, int act, float* __restrict__ ds, DropArgs drop) {
  __shared__ float s_red[16][65];
  const bool dropping = drop.step != nullptr;            // workgroup-uniform
  const unsigned long long drop_step = dropping ? *drop.step : 0ull;
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
      float dsc[4] = {1.f, 1.f, 1.f, 1.f};
      if (dropping) drop_scale<4>(drop, drop_step, (uint32_t)(base + t * C), dsc);   // same mask as bn_act_fwd (dropout.h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z1 = fmaf(yv[j], a1[j], b1[j]);
        const float z = z1 * sc[j] * (y2 ? dsc[j] : 1.f) + (y2 ? fmaf(rv[j], a2[j], b2[j]) : 0.f);
        float g = 1.f;
        if (act == LASR_ACT_RELU) g = z > 0.f ? 1.f : 0.f;
        else if (act == LASR_ACT_SWISH) { const float s = 1.f / (1.f + __expf(-z)); g = s * (1.f + z * (1.f - s)); }
        acc[j] = fmaf(dv[j] * g * dsc[j], z1, acc[j]);       // d(out)/d(scale) = [act' . dropout scale] * BN output
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

// One workgroup per utterance.  pooled = a*sums/T + b ; hidden = relu(W1 pooled) ; scale = sigmoid(W2 hidden).
// The two mat-vecs are latency-, not bandwidth-work (256 KB of weights per utterance out of L2): what matters is how many
// loads are in flight.  W1 [H][C]: one wave per group of 4 hidden units, lanes along C (coalesced), the 4 x C/64 loads of a
// group issued before the first reduction (one unit at a time = 16 dependent trips per wave: 36 us per launch);
// W2 [C][H]: one thread per output channel reads its own row as H/4 16-byte loads, all in flight.
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ sums, const float* __restrict__ coef,
                                                         const float* __restrict__ W1, const float* __restrict__ W2, int C, int H,
                                                         float inv_T, float* __restrict__ pooled, float* __restrict__ hidden,
                                                         float* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // pooled[C] | hidden[H]
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
  for (int h0 = wid * 8; h0 < H; h0 += 32) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int c = lane; c < C; c += 64) {
      const float pv = s_p[c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = fmaf(W1[(int64_t)min(h0 + u, H - 1) * C + c], pv, acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float v = fmaxf(wave_sum(acc[u]), 0.f);
      if (lane == 0 && h0 + u < H) {
        s_h[h0 + u] = v;
        hidden[(int64_t)b * H + h0 + u] = v;
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float4* w = reinterpret_cast<const float4*>(W2 + (int64_t)c * H);   // H % 4 == 0 (H = C/8, C % 32 == 0 checked on the host)
    float acc = 0.f;
    for (int h4 = 0; h4 < H / 4; h4 += 16) {
      float4 wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wv[u] = w[min(h4 + u, H / 4 - 1)];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (h4 + u < H / 4) {
          const float4 hv = *reinterpret_cast<const float4*>(s_h + 4 * (h4 + u));
          acc = fmaf(wv[u].x, hv.x, acc); acc = fmaf(wv[u].y, hv.y, acc); acc = fmaf(wv[u].z, hv.z, acc); acc = fmaf(wv[u].w, hv.w, acc);
        }
      }
    }
    scale[(int64_t)b * C + c] = 1.f / (1.f + expf(-acc));
  }
}

// One workgroup per utterance: back through sigmoid, W2, relu, W1, the mean over T.
// Writes seg[b][c] = d(loss)/d(BN output z1[b,t,c]) through the pooled path (same for every t) and the two small per-utterance
// vectors the batched weight-gradient kernel below needs: d2[b][c] = d(pre-sigmoid), dh[b][h] = d(pre-relu).
//   dh[h] = relu'(.) sum_c W2[c][h] d2[c]: lanes along h (a row of W2 is one coalesced 4*H-byte read), each wave walks C/4 rows
//   with 8 row loads in flight, the four waves' partial vectors meet in LDS;
//   seg[c] = sum_h W1[h][c] dh[h] / T: one thread per c, rows of W1 read coalesced across c, 8 in flight.
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ scale,
                                                         const float* __restrict__ hidden, const float* __restrict__ W1,
                                                         const float* __restrict__ W2, int C, int H, float inv_T,
                                                         float* __restrict__ seg, float* __restrict__ d2_out, float* __restrict__ dh_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // d2[C] | part[4][H] | dh[H]
  float* s_d2 = sm;
  float* s_part = sm + C;
  float* s_dh = s_part + 4 * H;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float s = scale[(int64_t)b * C + c];
    const float v = ds[(int64_t)b * C + c] * s * (1.f - s);
    s_d2[c] = v;
    d2_out[(int64_t)b * C + c] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  {
    const int rows = C / 4, c0 = wid * rows;       // this wave's rows of W2
    float acc = 0.f;
    const int hl = min(lane, H - 1);
    for (int r = 0; r < rows; r += 32) {
      float wv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) wv[u] = W2[(int64_t)(c0 + min(r + u, rows - 1)) * H + hl];
#pragma unroll
      for (int u = 0; u < 32; ++u)
        if (r + u < rows) acc = fmaf(wv[u], s_d2[c0 + r + u], acc);
    }
    if (lane < H) s_part[wid * H + lane] = acc;
  }
  __syncthreads();
  if (threadIdx.x < H) {
    const int h = threadIdx.x;
    const float a = (s_part[h] + s_part[H + h]) + (s_part[2 * H + h] + s_part[3 * H + h]);
    const float v = hidden[(int64_t)b * H + h] > 0.f ? a : 0.f;
    s_dh[h] = v;
    dh_out[(int64_t)b * H + h] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int h = 0; h < H; h += 32) {
      float wv[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) wv[u] = W1[(int64_t)min(h + u, H - 1) * C + c];
#pragma unroll
      for (int u = 0; u < 32; ++u)
        if (h + u < H) acc = fmaf(wv[u], s_dh[h + u], acc);
    }
    seg[(int64_t)b * C + c] = acc * inv_T;
  }
}

// dW2[c][h] = sum_b d2[b][c] * hidden[b][h],  dW1[h][c] = sum_b dh[b][h] * pooled[b][c]: two 32-term outer-product sums per
// weight, for all utterances in ONE launch (was: per-utterance [C][H] slabs, 2 x 4 MB written and re-read by two reductions).
// grid (ceil(C/64), 2): blockIdx.y = 0 -> dW2 tile [64 c][H], 1 -> dW1 tile [H][64 c].  The batch's four small matrices sit in LDS.
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ d2, const float* __restrict__ hidden,
                                                       const float* __restrict__ dh, const float* __restrict__ pooled, int B, int C, int H,
                                                       float* __restrict__ dW1, float* __restrict__ dW2) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // a[B][64] | v[B][H]
  float* s_a = sm;
  float* s_v = sm + (size_t)B * 64;
  const int c0 = blockIdx.x * 64;
  const bool w2 = blockIdx.y == 0;
  const float* A = w2 ? d2 : pooled;      // [B][C], the 64-channel slice
  const float* V = w2 ? hidden : dh;      // [B][H]
  for (int i = threadIdx.x; i < B * 64; i += 256) {
    const int bb = i >> 6, cc = i & 63;
    s_a[i] = c0 + cc < C ? A[(int64_t)bb * C + c0 + cc] : 0.f;
  }
  for (int i = threadIdx.x; i < B * H; i += 256) s_v[i] = V[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * H; i += 256) {
    // dW2 is [C][H]: consecutive threads along h;  dW1 is [H][C]: consecutive threads along c  (coalesced stores either way)
    const int cc = w2 ? i / H : i & 63;
    const int h = w2 ? i - cc * H : i >> 6;
    if (c0 + cc >= C) continue;
    float acc = 0.f;
    for (int bb = 0; bb < B; ++bb) acc = fmaf(s_a[bb * 64 + cc], s_v[bb * H + h], acc);
    if (w2) dW2[(int64_t)(c0 + cc) * H + h] = acc;
    else dW1[(int64_t)h * C + c0 + cc] = acc;
  }
}

}  // namespace lasr

namespace lasr {
// back through the excite MLP from ds [B][C] = d(loss)/d(scale): seg [B][C], dW1, dW2; d2 [B][C] and dh [B][C/8] are scratch
int launch_se_mlp_bwd(const float* ds, const float* scale, const float* hidden, const float* pooled, const float* W1, const float* W2,
                      int64_t B, int64_t T_, int64_t C, float* seg, float* dW1, float* dW2, float* d2, float* dh, hipStream_t st) {
  const int H = (int)(C / 8);
  if (C % 32 != 0 || H > 64 || (size_t)B * (64 + H) * sizeof(float) > 64 * 1024)
    return fail(LASR_E_SHAPE, "lasr_se_bwd: C=%lld B=%lld (the kernels are built for C <= 512 in multiples of 32)", (long long)C, (long long)B);
  hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3((unsigned)B), dim3(256), (size_t)(C + 5 * H) * sizeof(float), st, ds, scale, hidden, W1, W2,
                     (int)C, H, 1.0f / (float)T_, seg, d2, dh);
  LASR_LAUNCH_CHECK("se_mlp_bwd_kernel");
  hipLaunchKernelGGL(se_wgrad_kernel, dim3((unsigned)cdiv(C, 64), 2), dim3(256), (size_t)B * (64 + H) * sizeof(float), st, d2, hidden, dh,
                     pooled, (int)B, (int)C, H, dW1, dW2);
  LASR_LAUNCH_CHECK("se_wgrad_kernel");
  return 0;
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
  LASR_CHECK_SHAPE(B > 0 && T_ > 0 && C >= 32 && C % 32 == 0 && C <= 8192, "lasr_se_fwd: C=%lld", (long long)C);
  const int H = (int)(C / 8);
  hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3((unsigned)B), dim3(256), (size_t)(C + H) * sizeof(float), as_stream(stream), sums, coef, W1,
                     W2, (int)C, H, 1.0f / (float)T_, pooled, hidden, scale);
  LASR_LAUNCH_CHECK("se_mlp_fwd_kernel");
  return 0;
}

extern "C" size_t lasr_se_bwd_workspace_bytes(int64_t B, int64_t C) {
  // ds [B][C] | d2 [B][C] | dh [B][C/8]
  return 2 * align_up((size_t)B * C * sizeof(float), 256) + align_up((size_t)B * (C / 8) * sizeof(float), 256);
}

extern "C" int lasr_se_bwd_drop(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                                const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T_,
                                int64_t C, int act, const lasr_dropout* dropout, float* seg, float* dW1, float* dW2, void* workspace,
                                size_t workspace_bytes, void* stream);
extern "C" int lasr_se_bwd(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                           const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T_,
                           int64_t C, int act, float* seg, float* dW1, float* dW2, void* workspace, size_t workspace_bytes,
                           void* stream) {
  return lasr_se_bwd_drop(dout, y, coef, y2, coef2, scale, hidden, pooled, W1, W2, dtype, B, T_, C, act, nullptr, seg, dW1, dW2, workspace,
                          workspace_bytes, stream);
}

extern "C" int lasr_se_bwd_drop(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                                const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T_,
                                int64_t C, int act, const lasr_dropout* dropout, float* seg, float* dW1, float* dW2, void* workspace,
                                size_t workspace_bytes, void* stream) {
  DropArgs da;
  da.step = nullptr; da.seed = 0; da.unit = 0; da.thresh = 0; da.inv_keep = 1.f;
  if (dropout && dropout->step && dropout->p > 0.f) {    // (same conversion as norm.hip's make_drop)
    da.step = reinterpret_cast<const unsigned long long*>(dropout->step);
    da.seed = dropout->seed; da.unit = dropout->unit;
    const float p = dropout->p < 0.999f ? dropout->p : 0.999f;
    da.thresh = (uint32_t)(p * 65536.f + 0.5f);
    da.inv_keep = 1.f / (1.f - (float)da.thresh / 65536.f);
  }
  LASR_CHECK_ARG(dout && y && coef && scale && hidden && pooled && W1 && W2 && seg && dW1 && dW2 && workspace, "lasr_se_bwd: null pointer");
  LASR_CHECK_ARG(!y2 || coef2, "lasr_se_bwd: residual coefficients");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_se_bwd: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && C >= 8 && C % 8 == 0 && C <= 8192, "lasr_se_bwd: C=%lld", (long long)C);
  if (workspace_bytes < lasr_se_bwd_workspace_bytes(B, C)) return fail(LASR_E_WORKSPACE, "lasr_se_bwd: workspace");
  char* w = reinterpret_cast<char*>(workspace);
  float* ds = reinterpret_cast<float*>(w);
  float* d2 = reinterpret_cast<float*>(w + align_up((size_t)B * C * sizeof(float), 256));
  float* dh = reinterpret_cast<float*>(w + 2 * align_up((size_t)B * C * sizeof(float), 256));
  dim3 grid((unsigned)cdiv(C, 64), (unsigned)B);
  hipStream_t st = as_stream(stream);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(se_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)y, coef, (const float*)y2,
                       coef2, scale, T_, C, act, ds, da);
  else
    hipLaunchKernelGGL(se_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)y, coef,
                       (const bf16_t*)y2, coef2, scale, T_, C, act, ds, da);
  LASR_LAUNCH_CHECK("se_bwd_reduce_kernel");
  return launch_se_mlp_bwd(ds, scale, hidden, pooled, W1, W2, B, T_, C, seg, dW1, dW2, d2, dh, st);
}
