// Multi-tensor NovoGrad over flat f32 buffers: two launches per step for any number of tensors.
// Replaces scheduler/novograd.py:75-145 (betas=(0.8,0.5), eps=1e-8, weight_decay; no amsgrad,
// grad_averaging or luc: train.py:46), which issues ~10 tiny kernels per parameter tensor.
#include "common.h"

namespace lasr {

// Stage 1: every workgroup owns a 16K-element slice of the flat gradient, walks the tensor segments
// that intersect it and adds each segment's sum of squares to norm2[tensor] (f64 atomics: a handful
// per workgroup; f64 keeps the result independent of arrival order to well below f32 resolution).
static constexpr int kNormChunk = 16384;
__global__ __launch_bounds__(256) void novograd_norm_kernel(const float* __restrict__ grads, const int64_t* __restrict__ offsets,
                                                            int n_tensors, int64_t n, double* __restrict__ norm2, float grad_scale) {
  __shared__ double s_red[4];
  const int64_t beg = (int64_t)blockIdx.x * kNormChunk;
  const int64_t end = beg + kNormChunk < n ? beg + kNormChunk : n;
  int lo = 0, hi = n_tensors;  // offsets[lo] <= beg < offsets[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offsets[mid] <= beg) lo = mid; else hi = mid;
  }
  int64_t seg = beg;
  for (int t = lo; t < n_tensors && seg < end; ++t) {
    const int64_t tend = offsets[t + 1] < end ? offsets[t + 1] : end;
    double acc = 0.0;
    int64_t e0 = seg;
    if ((seg & 3) == 0) {   // 16-byte groups of the segment (every tensor of the model starts on one), scalar tail
      const int64_t nv = (tend - seg) >> 2;
      // four 16-byte loads in flight per thread (clamped address, masked value): a 64 KB slice is 4 rounds instead of 16 dependent trips
      // (round 3: 13.6 -> 10.1 us.  Round 5 tried sixteen in flight - one round per slice: 10.8 -> 18.2 us, the short tensors' slices
      // then issue sixteen clamped duplicate loads each)
      constexpr int kInFlight = 4;
      for (int64_t q0 = threadIdx.x; q0 < nv; q0 += kInFlight * 256) {
        float4 g4[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) g4[u] = *reinterpret_cast<const float4*>(grads + seg + 4 * min(q0 + 256 * u, nv - 1));
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
          const float mk = q0 + 256 * u < nv ? grad_scale : 0.f;
          const float a = g4[u].x * mk, b = g4[u].y * mk, c = g4[u].z * mk, d = g4[u].w * mk;
          acc += ((double)a * (double)a + (double)b * (double)b) + ((double)c * (double)c + (double)d * (double)d);
        }
      }
      e0 = seg + 4 * nv;
    }
    for (int64_t e = e0 + threadIdx.x; e < tend; e += 256) {
      const float g = grads[e] * grad_scale;
      acc += (double)g * (double)g;
    }
    acc = wave_sum_d(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && tend > seg) atomicAdd(&norm2[t], (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    seg = tend;
  }
}

// Stage 2: v_i <- first ? ||g||^2 : b2 v + (1-b2)||g||^2 ; denom_i = sqrt(v_i)+eps ; norm2 re-zeroed
__global__ __launch_bounds__(256) void novograd_moment_kernel(double* __restrict__ norm2, float* __restrict__ exp_avg_sq,
                                                              float* __restrict__ denom, int n_tensors, float beta2, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tensors) return;
  const float norm = (float)norm2[i];
  norm2[i] = 0.0;
  float v = exp_avg_sq[i];
  v = (v == 0.f) ? norm : beta2 * v + (1.f - beta2) * norm;  // "if exp_avg_sq == 0: copy" novograd.py:115
  exp_avg_sq[i] = v;
  denom[i] = sqrtf(v) + eps;
}

__global__ __launch_bounds__(256) void novograd_update_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                              float* __restrict__ exp_avg, const int64_t* __restrict__ offsets,
                                                              int n_tensors, const float* __restrict__ denom,
                                                              const float* __restrict__ lr_ptr, float beta1, float wd,
                                                              float grad_scale, int64_t n) {
  // The tensor of an element is found by bisection over the offset table: the table lives in LDS (a search from
  // global memory is seven dependent loads per element) and one search serves four consecutive elements, which
  // move as 16-byte vectors whenever the group lies inside one tensor (always, for this model's shapes).
  extern __shared__ int64_t s_off[];   // [n_tensors + 1]
  for (int i = threadIdx.x; i <= n_tensors; i += 256) s_off[i] = offsets[i];
  __syncthreads();
  const float lr = *lr_ptr;
  const int64_t ngrp = (n + 3) >> 2;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < ngrp; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = 4 * q;
    int lo = 0, hi = n_tensors;  // s_off[lo] <= e < s_off[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_off[mid] <= e) lo = mid; else hi = mid;
    }
    if (e + 4 <= n && s_off[lo + 1] >= e + 4) {
      const float dn = denom[lo];
      const float4 p = *reinterpret_cast<const float4*>(params + e);
      const float4 g = *reinterpret_cast<const float4*>(grads + e);
      const float4 m0 = *reinterpret_cast<const float4*>(exp_avg + e);
      float pv[4] = {p.x, p.y, p.z, p.w}, gv[4] = {g.x, g.y, g.z, g.w}, mv[4] = {m0.x, m0.y, m0.z, m0.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float gi = (gv[i] * grad_scale) / dn;
        if (wd != 0.f) gi = gi + wd * pv[i];
        mv[i] = mv[i] * beta1 + gi;
        pv[i] = pv[i] - lr * mv[i];
      }
      *reinterpret_cast<float4*>(exp_avg + e) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *reinterpret_cast<float4*>(params + e) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    } else {
      for (int64_t ee = e; ee < e + 4 && ee < n; ++ee) {
        while (lo + 1 < n_tensors && s_off[lo + 1] <= ee) ++lo;
        const float pe = params[ee];
        float gi = (grads[ee] * grad_scale) / denom[lo];
        if (wd != 0.f) gi = gi + wd * pe;
        const float m = exp_avg[ee] * beta1 + gi;
        exp_avg[ee] = m;
        params[ee] = pe - lr * m;
      }
    }
  }
}

// f32 -> bf16 copy of a flat buffer (weights for the bf16 MFMA kernels)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = f32_to_bf16(in[e]);
}

// out[r][c] = bf16(in[r][c]) for c < cols, 0 for cols <= c < ld_out: a bf16 copy whose rows start on 16-byte
// boundaries (ld_out % 8 == 0) so that GEMMs over a narrow matrix (the 28-class logit gradient) take the
// aligned operand path.
__global__ __launch_bounds__(256) void cast_pad_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t rows,
                                                            int64_t cols, int64_t ld_out) {
  const int64_t total = rows * ld_out;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ld_out, c = i - r * ld_out;
    out[i] = c < cols ? f32_to_bf16(in[r * cols + c]) : (bf16_t)0;
  }
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_novograd_workspace_bytes(int64_t n_tensors, int64_t n_elems) {
  (void)n_elems;
  return align_up((size_t)n_tensors * sizeof(float), 256) + align_up((size_t)n_tensors * sizeof(double), 256);
}

static int novograd_step_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* offsets,
                              int64_t n_tensors, int64_t n_elems, const float* lr, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, void* workspace, size_t workspace_bytes, void* stream, bool zeroed);

extern "C" int lasr_novograd_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* offsets,
                                  int64_t n_tensors, int64_t n_elems, const float* lr, float beta1, float beta2, float eps,
                                  float weight_decay, float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
  return novograd_step_impl(params, grads, exp_avg, exp_avg_sq, offsets, n_tensors, n_elems, lr, beta1, beta2, eps, weight_decay, grad_scale,
                            workspace, workspace_bytes, stream, false);
}
// the same step on a workspace the caller keeps: zero before its FIRST use, left zero by every call (the moment kernel re-zeroes the
// norm accumulators it has consumed) - the per-step memset launch (4.6 us in the cfg2 trace) is the caller's one-time torch.zeros
extern "C" int lasr_novograd_step_keep(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* offsets,
                                       int64_t n_tensors, int64_t n_elems, const float* lr, float beta1, float beta2, float eps,
                                       float weight_decay, float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
  return novograd_step_impl(params, grads, exp_avg, exp_avg_sq, offsets, n_tensors, n_elems, lr, beta1, beta2, eps, weight_decay, grad_scale,
                            workspace, workspace_bytes, stream, true);
}

static int novograd_step_impl(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* offsets,
                              int64_t n_tensors, int64_t n_elems, const float* lr, float beta1, float beta2, float eps,
                              float weight_decay, float grad_scale, void* workspace, size_t workspace_bytes, void* stream, bool zeroed) {
  LASR_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && offsets && lr && workspace, "lasr_novograd_step: null pointer");
  LASR_CHECK_SHAPE(n_tensors > 0 && n_tensors < (1 << 20) && n_elems > 0, "lasr_novograd_step: n_tensors=%lld", (long long)n_tensors);
  if (workspace_bytes < lasr_novograd_workspace_bytes(n_tensors, n_elems)) return fail(LASR_E_WORKSPACE, "lasr_novograd_step: workspace");
  float* denom = reinterpret_cast<float*>(workspace);
  double* norm2 = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + align_up((size_t)n_tensors * sizeof(float), 256));
  hipStream_t st = as_stream(stream);
  // (bench.py's class table) gradients read twice, parameters and momentum read and written
  const int tok = prof_begin(LASR_PROF_OTHER, st, 0.0, 6.0 * (double)n_elems * sizeof(float));
  struct End { int t; hipStream_t s; ~End() { prof_end(t, s); } } end_{tok, st};
  if (!zeroed) {
    hipError_t me = hipMemsetAsync(norm2, 0, (size_t)n_tensors * sizeof(double), st);
    if (me != hipSuccess) return hip_fail(me, "lasr_novograd_step memset");
  }
  hipLaunchKernelGGL(novograd_norm_kernel, dim3((unsigned)cdiv(n_elems, kNormChunk)), dim3(256), 0, st, grads, offsets, (int)n_tensors,
                     n_elems, norm2, grad_scale);
  LASR_LAUNCH_CHECK("novograd_norm_kernel");
  hipLaunchKernelGGL(novograd_moment_kernel, dim3((unsigned)cdiv(n_tensors, 256)), dim3(256), 0, st, norm2, exp_avg_sq, denom,
                     (int)n_tensors, beta2, eps);
  LASR_LAUNCH_CHECK("novograd_moment_kernel");
  int64_t blocks = cdiv(cdiv(n_elems, 4), 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  const size_t off_bytes = (size_t)(n_tensors + 1) * sizeof(int64_t);
  LASR_CHECK_SHAPE(off_bytes <= 48 * 1024, "lasr_novograd_step: offset table too large for LDS (%lld tensors)", (long long)n_tensors);
  hipLaunchKernelGGL(novograd_update_kernel, dim3((unsigned)blocks), dim3(256), off_bytes, st, params, grads, exp_avg, offsets,
                     (int)n_tensors, denom, lr, beta1, weight_decay, grad_scale, n_elems);
  LASR_LAUNCH_CHECK("novograd_update_kernel");
  return 0;
}

extern "C" int lasr_cast_pad_f32_to_bf16(const float* in, void* out, int64_t rows, int64_t cols, int64_t ld_out, void* stream) {
  LASR_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld_out >= cols, "lasr_cast_pad_f32_to_bf16: bad argument");
  int64_t blocks = cdiv(rows * ld_out, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in, reinterpret_cast<bf16_t*>(out), rows,
                     cols, ld_out);
  LASR_LAUNCH_CHECK("cast_pad_bf16_kernel");
  return 0;
}

extern "C" int lasr_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  LASR_CHECK_ARG(in && out && n > 0, "lasr_cast_f32_to_bf16: bad argument");
  int64_t blocks = cdiv(n, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), in, reinterpret_cast<bf16_t*>(out), n);
  LASR_LAUNCH_CHECK("cast_bf16_kernel");
  return 0;
}
