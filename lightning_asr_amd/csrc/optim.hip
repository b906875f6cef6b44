// Multi-tensor NovoGrad over flat f32 buffers: two launches per step for any number of tensors.
// Replaces scheduler/novograd.py:75-145 (betas=(0.8,0.5), eps=1e-8, weight_decay; no amsgrad,
// grad_averaging or luc: train.py:46), which issues ~10 tiny kernels per parameter tensor.
#include "common.h"

namespace lasr {

// grid = n_tensors, block 1024: v_i <- first ? ||g||^2 : b2 v + (1-b2)||g||^2 ; denom_i = sqrt(v_i)+eps
__global__ __launch_bounds__(1024) void novograd_norm_kernel(const float* __restrict__ grads, const int64_t* __restrict__ offsets,
                                                             float* __restrict__ exp_avg_sq, float* __restrict__ denom,
                                                             float beta2, float eps, float grad_scale) {
  __shared__ double s_red[16];
  const int i = blockIdx.x;
  const int64_t beg = offsets[i], end = offsets[i + 1];
  double acc = 0.0;
  for (int64_t e = beg + threadIdx.x; e < end; e += 1024) {
    const float g = grads[e] * grad_scale;
    acc += (double)g * (double)g;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += s_red[w];
    const float norm = (float)s;
    float v = exp_avg_sq[i];
    v = (v == 0.f) ? norm : beta2 * v + (1.f - beta2) * norm;  // "if exp_avg_sq == 0: copy" novograd.py:115
    exp_avg_sq[i] = v;
    denom[i] = sqrtf(v) + eps;
  }
}

__global__ __launch_bounds__(256) void novograd_update_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                              float* __restrict__ exp_avg, const int64_t* __restrict__ offsets,
                                                              int n_tensors, const float* __restrict__ denom,
                                                              const float* __restrict__ lr_ptr, float beta1, float wd,
                                                              float grad_scale, int64_t n) {
  const float lr = *lr_ptr;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_tensors;  // offsets[lo] <= e < offsets[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= e) lo = mid; else hi = mid;
    }
    const float p = params[e];
    float g = (grads[e] * grad_scale) / denom[lo];
    if (wd != 0.f) g = g + wd * p;
    const float m = exp_avg[e] * beta1 + g;
    exp_avg[e] = m;
    params[e] = p - lr * m;
  }
}

// f32 -> bf16 copy of a flat buffer (weights for the bf16 MFMA kernels)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = f32_to_bf16(in[e]);
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_novograd_workspace_bytes(int64_t n_tensors, int64_t n_elems) {
  (void)n_elems;
  return align_up((size_t)n_tensors * sizeof(float), 256);
}

extern "C" int lasr_novograd_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* offsets,
                                  int64_t n_tensors, int64_t n_elems, const float* lr, float beta1, float beta2, float eps,
                                  float weight_decay, float grad_scale, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && offsets && lr && workspace, "lasr_novograd_step: null pointer");
  LASR_CHECK_SHAPE(n_tensors > 0 && n_tensors < (1 << 20) && n_elems > 0, "lasr_novograd_step: n_tensors=%lld", (long long)n_tensors);
  if (workspace_bytes < lasr_novograd_workspace_bytes(n_tensors, n_elems)) return fail(LASR_E_WORKSPACE, "lasr_novograd_step: workspace");
  float* denom = reinterpret_cast<float*>(workspace);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(novograd_norm_kernel, dim3((unsigned)n_tensors), dim3(1024), 0, st, grads, offsets, exp_avg_sq, denom, beta2, eps, grad_scale);
  LASR_LAUNCH_CHECK("novograd_norm_kernel");
  int64_t blocks = cdiv(n_elems, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(novograd_update_kernel, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg, offsets, (int)n_tensors,
                     denom, lr, beta1, weight_decay, grad_scale, n_elems);
  LASR_LAUNCH_CHECK("novograd_update_kernel");
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
