// Depthwise 1-D convolution over channels-last [B][T][C] activations: forward (also used, with the
// taps reversed, as the data gradient) and the weight gradient.
// Replaces nn.Conv1d(groups=C, padding=k//2, bias=False) at models/QuartNet.py:19-21,30 and its
// autograd backward.  VALU work (0.13 of 2.5 GMAC/utterance); time tiles are staged through LDS
// with their k-1 halo, channels ride the lanes so every global access is a full coalesced row.
#include "common.h"

namespace lasr {

static constexpr int kCB = 64;    // channels per workgroup (16 lanes x 4 channels)
static constexpr int kTT = 128;   // output frames per workgroup (16 lanes x 8 outputs)
static constexpr int kR = 8;      // outputs per thread
static constexpr int kMaxK = 128;

// grid: (ceil(Tout/kTT), ceil(C/kCB), B), block 256, dynamic LDS: x tile + taps
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const T* __restrict__ addend, T* __restrict__ y, int64_t Tin,
                                                         int64_t Tout, int64_t C, int k, int stride, int flip) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = k / 2;
  const int in_rows = (kTT - 1) * stride + k;
  float* s_x = smem;                         // [in_rows][kCB]
  float* s_w = smem + (size_t)in_rows * kCB;  // [k][kCB]
  const int b = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.y * kCB;
  const int64_t t0 = (int64_t)blockIdx.x * kTT;
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int64_t c = c0 + cl * 4;
  const bool c_ok = c < C;  // C % 4 == 0
  const T* xb = x + (int64_t)b * Tin * C;

  // taps -> LDS [j][channel]
  for (int i = threadIdx.x; i < k * kCB; i += 256) {
    const int ch = i / k, j = i - ch * k;
    float v = 0.f;
    if (c0 + ch < C) v = w[(c0 + ch) * k + (flip ? (k - 1 - j) : j)];
    s_w[j * kCB + ch] = v;
  }
  // input rows t0*stride - pad ... (+in_rows), zero outside [0, Tin)
  const int64_t in0 = t0 * stride - pad;
  for (int r = tl; r < in_rows; r += 16) {
    const int64_t ti = in0 + r;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c_ok && ti >= 0 && ti < Tin) Elem<T>::ld4(xb + ti * C + c, v);
    *reinterpret_cast<float4*>(s_x + (size_t)r * kCB + cl * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();

  float acc[kR][4];
#pragma unroll
  for (int r = 0; r < kR; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
  const float* xs = s_x + (size_t)(tl * kR * stride) * kCB + cl * 4;
  const float* ws = s_w + cl * 4;
  for (int j = 0; j < k; ++j) {
    const float4 wv = *reinterpret_cast<const float4*>(ws + j * kCB);
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + (size_t)(r * stride + j) * kCB);
      acc[r][0] = fmaf(wv.x, xv.x, acc[r][0]);
      acc[r][1] = fmaf(wv.y, xv.y, acc[r][1]);
      acc[r][2] = fmaf(wv.z, xv.z, acc[r][2]);
      acc[r][3] = fmaf(wv.w, xv.w, acc[r][3]);
    }
  }
  if (!c_ok) return;
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const int64_t t = t0 + tl * kR + r;
    if (t < Tout) {
      const int64_t off = ((int64_t)b * Tout + t) * C + c;
      if (addend) {
        float a[4];
        Elem<T>::ld4(addend + off, a);
        acc[r][0] += a[0]; acc[r][1] += a[1]; acc[r][2] += a[2]; acc[r][3] += a[3];
      }
      Elem<T>::st4(y + off, acc[r]);
    }
  }
}

// Weight gradient.  grid: (n_chunks, ceil(C/kCB), B); each workgroup walks `chunk` output frames in
// steps of kWT and writes a partial [kCB][k] to partials[(b*n_chunks+chunk)][C][k].
static constexpr int kWT = 64;       // frames per LDS step
static constexpr int kWChunk = 256;  // frames per workgroup
static constexpr int kQ = kMaxK / 16;

template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* __restrict__ partials, int64_t Tin, int64_t Tout,
                                                           int64_t C, int k, int stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = k / 2;
  const int in_rows = (kWT - 1) * stride + k;
  float* s_x = smem;                           // [in_rows][kCB]
  float* s_d = smem + (size_t)in_rows * kCB;    // [kWT][kCB]
  const int b = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.y * kCB;
  const int cl = threadIdx.x & 15, jl = threadIdx.x >> 4;
  const int64_t c = c0 + cl * 4;
  const bool c_ok = c < C;
  const T* xb = x + (int64_t)b * Tin * C;
  const T* db = dy + (int64_t)b * Tout * C;
  float acc[kQ][4];
#pragma unroll
  for (int q = 0; q < kQ; ++q) acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
  const int nq = (k + 15) / 16;
  const int64_t tbeg = (int64_t)blockIdx.x * kWChunk;
  const int64_t tend = tbeg + kWChunk < Tout ? tbeg + kWChunk : Tout;
  for (int64_t t0 = tbeg; t0 < tend; t0 += kWT) {
    __syncthreads();
    const int64_t in0 = t0 * stride - pad;
    for (int r = jl; r < in_rows; r += 16) {
      const int64_t ti = in0 + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (c_ok && ti >= 0 && ti < Tin) Elem<T>::ld4(xb + ti * C + c, v);
      *reinterpret_cast<float4*>(s_x + (size_t)r * kCB + cl * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    for (int r = jl; r < kWT; r += 16) {
      const int64_t t = t0 + r;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (c_ok && t < tend) Elem<T>::ld4(db + t * C + c, v);
      *reinterpret_cast<float4*>(s_d + (size_t)r * kCB + cl * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    for (int t = 0; t < kWT; ++t) {
      const float4 dv = *reinterpret_cast<const float4*>(s_d + (size_t)t * kCB + cl * 4);
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        if (q < nq) {
          const int j = jl + 16 * q;  // rows past k-1 read the next frames' data: harmless, discarded below
          const int row = t * stride + j;
          if (row < in_rows) {
            const float4 xv = *reinterpret_cast<const float4*>(s_x + (size_t)row * kCB + cl * 4);
            acc[q][0] = fmaf(dv.x, xv.x, acc[q][0]);
            acc[q][1] = fmaf(dv.y, xv.y, acc[q][1]);
            acc[q][2] = fmaf(dv.z, xv.z, acc[q][2]);
            acc[q][3] = fmaf(dv.w, xv.w, acc[q][3]);
          }
        }
      }
    }
  }
  if (!c_ok) return;
  float* out = partials + ((int64_t)b * gridDim.x + blockIdx.x) * C * k;
#pragma unroll
  for (int q = 0; q < kQ; ++q) {
    const int j = jl + 16 * q;
    if (q < nq && j < k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(c + e) * k + j] = acc[q][e];
    }
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partials, int n_part, int64_t n,
                                                           float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < n_part; ++p) s += partials[(int64_t)p * n + i];
  out[i] = s;
}

}  // namespace lasr

using namespace lasr;

static int64_t conv_out_len(int64_t Tin, int k, int stride) { return (Tin + 2 * (k / 2) - k) / stride + 1; }

extern "C" int lasr_dwconv_fwd(const void* x, const float* w, const void* addend, void* y, int dtype, int64_t B, int64_t Tin,
                               int64_t C, int k, int stride, int flip, void* stream) {
  LASR_CHECK_ARG(x && w && y, "lasr_dwconv_fwd: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_dwconv_fwd: bad dtype");
  LASR_CHECK_SHAPE(k >= 1 && k <= kMaxK && (k & 1) && (stride == 1 || stride == 2) && C % 4 == 0 && B > 0 && B < 65536 && Tin > 0,
                   "lasr_dwconv_fwd: k=%d stride=%d C=%lld", k, stride, (long long)C);
  LASR_CHECK_SHAPE(!(flip && stride != 1), "lasr_dwconv_fwd: flip needs stride 1");
  const int64_t Tout = conv_out_len(Tin, k, stride);
  const int in_rows = (kTT - 1) * stride + k;
  const size_t shmem = ((size_t)in_rows + k) * kCB * sizeof(float);
  dim3 grid((unsigned)cdiv(Tout, kTT), (unsigned)cdiv(C, kCB), (unsigned)B);
  const int tok = prof_begin(LASR_PROF_DWCONV, as_stream(stream), 2.0 * (double)B * Tout * C * k,
                             (double)B * (Tin + Tout * (addend ? 2 : 1)) * C * dtype_size(dtype));
  if (dtype == LASR_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dwconv_fwd_kernel<float>, grid, dim3(256), shmem, as_stream(stream), (const float*)x, w,
                       (const float*)addend, (float*)y, Tin, Tout, C, k, stride, flip);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dwconv_fwd_kernel<bf16_t>, grid, dim3(256), shmem, as_stream(stream), (const bf16_t*)x, w,
                       (const bf16_t*)addend, (bf16_t*)y, Tin, Tout, C, k, stride, flip);
  }
  prof_end(tok, as_stream(stream));
  LASR_LAUNCH_CHECK("dwconv_fwd_kernel");
  return 0;
}

extern "C" size_t lasr_dwconv_wgrad_workspace_bytes(int64_t B, int64_t Tout, int64_t C, int k) {
  return (size_t)B * cdiv(Tout, kWChunk) * C * k * sizeof(float);
}

extern "C" int lasr_dwconv_wgrad(const void* x, const void* dy, float* dw, int dtype, int64_t B, int64_t Tin, int64_t C, int k,
                                 int stride, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(x && dy && dw && workspace, "lasr_dwconv_wgrad: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_dwconv_wgrad: bad dtype");
  LASR_CHECK_SHAPE(k >= 1 && k <= kMaxK && (k & 1) && (stride == 1 || stride == 2) && C % 4 == 0 && B > 0 && B < 65536 && Tin > 0,
                   "lasr_dwconv_wgrad: k=%d stride=%d C=%lld", k, stride, (long long)C);
  const int64_t Tout = conv_out_len(Tin, k, stride);
  if (workspace_bytes < lasr_dwconv_wgrad_workspace_bytes(B, Tout, C, k)) return fail(LASR_E_WORKSPACE, "lasr_dwconv_wgrad: workspace");
  const int n_chunks = (int)cdiv(Tout, kWChunk);
  const int in_rows = (kWT - 1) * stride + k;
  const size_t shmem = ((size_t)in_rows + kWT) * kCB * sizeof(float);
  dim3 grid((unsigned)n_chunks, (unsigned)cdiv(C, kCB), (unsigned)B);
  float* partials = reinterpret_cast<float*>(workspace);
  if (dtype == LASR_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), shmem, as_stream(stream), (const float*)x, (const float*)dy,
                       partials, Tin, Tout, C, k, stride);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, grid, dim3(256), shmem, as_stream(stream), (const bf16_t*)x,
                       (const bf16_t*)dy, partials, Tin, Tout, C, k, stride);
  }
  LASR_LAUNCH_CHECK("dwconv_wgrad_kernel");
  const int64_t n = C * k;
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), partials,
                     (int)(B * n_chunks), n, dw);
  LASR_LAUNCH_CHECK("sum_partials_kernel");
  return 0;
}
