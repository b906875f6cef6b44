// Bidirectional LSTM "context" branch (hidden 40, one layer) of the Context / ContextSE variants:
// pack_padded_sequence -> nn.LSTM(256, 40, bidirectional) -> pad_packed_sequence
// (models/QuartNetContext.py:171-173,186-199).  Gate order i,f,g,o; the reverse direction starts at
// each utterance's own last valid frame; outputs are zero for t >= len_b.
//
// The input projection x W_ih^T is one MFMA GEMM per direction (lasr_gemm, f32 out); what is left is
// a latency-bound recurrence of len_b dependent steps, run by one persistent workgroup per
// (utterance, direction): thread j owns gate row j with its 40 recurrent weights in registers, the
// hidden state lives in LDS, two barriers per step.  All LSTM arithmetic is f32.
#include "common.h"
#include <math.h>

namespace lasr {

static constexpr int H = 40, G = 4 * H;  // hidden size, gate rows

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gx [B][T][G] f32 per direction (x W_ih^T, no bias); saved [B][T][2][G + 2H]: gates(i,f,g,o) | c | h
// out: columns [col0 + dir*H, +H) of a [B][T][ldo] tensor in T (zeros for t >= len).
template <typename T>
__global__ __launch_bounds__(256) void bilstm_fwd_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_r,
                                                         const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                         const float* __restrict__ bih_f, const float* __restrict__ bhh_f,
                                                         const float* __restrict__ bih_r, const float* __restrict__ bhh_r,
                                                         const int32_t* __restrict__ lens, int64_t Tt, T* __restrict__ out,
                                                         int64_t ldo, int64_t col0, float* __restrict__ saved) {
  __shared__ __attribute__((aligned(16))) float s_h[H];
  __shared__ float s_g[G];
  const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
  const float* gx = (dir ? gx_r : gx_f) + (int64_t)b * Tt * G;
  const float* whh = dir ? whh_r : whh_f;
  int len = lens[b];
  if (len > Tt) len = (int)Tt;
  float w[H];
  float bias = 0.f;
  if (j < G) {
#pragma unroll
    for (int k = 0; k < H; ++k) w[k] = whh[j * H + k];
    bias = (dir ? bih_r : bih_f)[j] + (dir ? bhh_r : bhh_f)[j];
  }
  if (j < H) s_h[j] = 0.f;
  float c = 0.f;
  // zero the padded frames of this direction's output slice
  for (int64_t i = (int64_t)len * H + j; i < Tt * H; i += 256) {
    const int64_t t = i / H;
    const int k = (int)(i - t * H);
    Elem<T>::st(out + ((int64_t)b * Tt + t) * ldo + col0 + dir * H + k, 0.f);
  }
  __syncthreads();
  float gnext = (j < G && len > 0) ? gx[(int64_t)(dir ? len - 1 : 0) * G + j] : 0.f;
  for (int s = 0; s < len; ++s) {
    const int t = dir ? len - 1 - s : s;
    const float gcur = gnext;
    if (j < G && s + 1 < len) gnext = gx[(int64_t)(dir ? len - 2 - s : s + 1) * G + j];
    if (j < G) {
      float acc = gcur + bias;
#pragma unroll
      for (int k4 = 0; k4 < H; k4 += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(s_h + k4);
        acc = fmaf(w[k4], hv.x, acc); acc = fmaf(w[k4 + 1], hv.y, acc);
        acc = fmaf(w[k4 + 2], hv.z, acc); acc = fmaf(w[k4 + 3], hv.w, acc);
      }
      const float a = (j >= 2 * H && j < 3 * H) ? tanhf(acc) : sigmoidf_(acc);
      s_g[j] = a;
      saved[(((int64_t)b * Tt + t) * 2 + dir) * (G + 2 * H) + j] = a;
    }
    __syncthreads();
    if (j < H) {
      const float ig = s_g[j], fg = s_g[H + j], gg = s_g[2 * H + j], og = s_g[3 * H + j];
      c = fmaf(fg, c, ig * gg);
      const float h = og * tanhf(c);
      s_h[j] = h;
      float* sv = saved + (((int64_t)b * Tt + t) * 2 + dir) * (G + 2 * H);
      sv[G + j] = c;
      sv[G + H + j] = h;
      Elem<T>::st(out + ((int64_t)b * Tt + t) * ldo + col0 + dir * H + j, h);
    }
    __syncthreads();
  }
}

// dout: columns [col0 + dir*H, +H) of a [B][T][ldd] tensor in T (gradient w.r.t. the LSTM output).
// dg [B][T][G] f32 per direction = gradient w.r.t. the gate pre-activations (zero rows for t >= len);
// pwhh [B][2][G][H] = this utterance's contribution to dW_hh.
template <typename T>
__global__ __launch_bounds__(256) void bilstm_bwd_kernel(const T* __restrict__ dout, int64_t ldd, int64_t col0,
                                                         const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                         const int32_t* __restrict__ lens, int64_t Tt, const float* __restrict__ saved,
                                                         float* __restrict__ dg_f, float* __restrict__ dg_r, float* __restrict__ pwhh) {
  __shared__ __attribute__((aligned(16))) float s_dg[G];
  __shared__ __attribute__((aligned(16))) float s_hprev[H];
  __shared__ float s_part[4][H];
  const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
  const float* whh = dir ? whh_r : whh_f;
  float* dg = (dir ? dg_r : dg_f) + (int64_t)b * Tt * G;
  int len = lens[b];
  if (len > Tt) len = (int)Tt;
  // thread (k = j % H, p = j / H) holds W_hh[40p .. 40p+39][k] for the dh_prev = W_hh^T dgates product
  const int k = j % H, p = j / H;
  float wt[H];
  float dw[H];
  if (j < G) {
#pragma unroll
    for (int q = 0; q < H; ++q) { wt[q] = whh[(p * H + q) * H + k]; dw[q] = 0.f; }
  }
  for (int64_t i = (int64_t)len * G + j; i < Tt * G; i += 256) dg[i] = 0.f;
  float dh_next = 0.f, dc_next = 0.f;  // carried by threads j < H
  for (int s = len - 1; s >= 0; --s) {
    const int t = dir ? len - 1 - s : s;             // step s of the forward recurrence touched frame t
    const int tp = dir ? t + 1 : t - 1;              // frame of the previous step (s-1), if s > 0
    const float* sv = saved + (((int64_t)b * Tt + t) * 2 + dir) * (G + 2 * H);
    if (j < H) {
      const float ig = sv[j], fg = sv[H + j], gg = sv[2 * H + j], og = sv[3 * H + j], c = sv[G + j];
      float cprev = 0.f, hprev = 0.f;
      if (s > 0) {
        const float* sp = saved + (((int64_t)b * Tt + tp) * 2 + dir) * (G + 2 * H);
        cprev = sp[G + j];
        hprev = sp[G + H + j];
      }
      s_hprev[j] = hprev;
      const float dh = Elem<T>::ld(dout + ((int64_t)b * Tt + t) * ldd + col0 + dir * H + j) + dh_next;
      const float tc = tanhf(c);
      const float d_o = dh * tc * og * (1.f - og);
      const float dc = fmaf(dh * og, 1.f - tc * tc, dc_next);
      s_dg[j] = dc * gg * ig * (1.f - ig);
      s_dg[H + j] = dc * cprev * fg * (1.f - fg);
      s_dg[2 * H + j] = dc * ig * (1.f - gg * gg);
      s_dg[3 * H + j] = d_o;
      dc_next = dc * fg;
    }
    __syncthreads();
    if (j < G) {
      const float mine = s_dg[j];
      dg[(int64_t)t * G + j] = mine;
      float acc = 0.f;
#pragma unroll
      for (int q4 = 0; q4 < H; q4 += 4) {
        const float4 hv = *reinterpret_cast<const float4*>(s_hprev + q4);
        dw[q4] = fmaf(mine, hv.x, dw[q4]); dw[q4 + 1] = fmaf(mine, hv.y, dw[q4 + 1]);
        dw[q4 + 2] = fmaf(mine, hv.z, dw[q4 + 2]); dw[q4 + 3] = fmaf(mine, hv.w, dw[q4 + 3]);
        const float4 gv = *reinterpret_cast<const float4*>(s_dg + p * H + q4);
        acc = fmaf(wt[q4], gv.x, acc); acc = fmaf(wt[q4 + 1], gv.y, acc);
        acc = fmaf(wt[q4 + 2], gv.z, acc); acc = fmaf(wt[q4 + 3], gv.w, acc);
      }
      s_part[p][k] = acc;
    }
    __syncthreads();
    if (j < H) dh_next = (s_part[0][j] + s_part[1][j]) + (s_part[2][j] + s_part[3][j]);
  }
  if (j < G) {
    float* o = pwhh + (((int64_t)b * 2 + dir) * G + j) * H;
#pragma unroll
    for (int q = 0; q < H; ++q) o[q] = dw[q];
  }
}

// dst[n][dcol0 + c] = src[n][scol0 + c] for c < ncols (optionally += ), with dtype conversion
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy_cols_kernel(const TS* __restrict__ src, int64_t lds, int64_t scol0, TD* __restrict__ dst,
                                                        int64_t ldd, int64_t dcol0, int64_t rows, int64_t ncols, int accumulate) {
  const int64_t total = rows * ncols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ncols, c = i - r * ncols;
    float v = Elem<TS>::ld(src + r * lds + scol0 + c);
    if (accumulate) v += Elem<TD>::ld(dst + r * ldd + dcol0 + c);
    Elem<TD>::st(dst + r * ldd + dcol0 + c, v);
  }
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_bilstm_saved_bytes(int64_t B, int64_t T_) { return (size_t)B * T_ * 2 * (G + 2 * H) * sizeof(float); }

extern "C" int lasr_bilstm_fwd(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const float* bih_f,
                               const float* bhh_f, const float* bih_r, const float* bhh_r, const int32_t* lens, int64_t B, int64_t T_,
                               void* out, int dtype, int64_t ld_out, int64_t col0, float* saved, void* stream) {
  LASR_CHECK_ARG(gx_f && gx_r && whh_f && whh_r && bih_f && bhh_f && bih_r && bhh_r && lens && out && saved, "lasr_bilstm_fwd: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_bilstm_fwd: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && col0 >= 0 && ld_out >= col0 + 2 * H, "lasr_bilstm_fwd: shape");
  dim3 grid((unsigned)B, 2);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(bilstm_fwd_kernel<float>, grid, dim3(256), 0, as_stream(stream), gx_f, gx_r, whh_f, whh_r, bih_f, bhh_f, bih_r, bhh_r,
                       lens, T_, (float*)out, ld_out, col0, saved);
  else
    hipLaunchKernelGGL(bilstm_fwd_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), gx_f, gx_r, whh_f, whh_r, bih_f, bhh_f, bih_r, bhh_r,
                       lens, T_, (bf16_t*)out, ld_out, col0, saved);
  LASR_LAUNCH_CHECK("bilstm_fwd_kernel");
  return 0;
}

extern "C" size_t lasr_bilstm_bwd_workspace_bytes(int64_t B) { return (size_t)B * 2 * G * H * sizeof(float); }

extern "C" int lasr_bilstm_bwd(const void* dout, int dtype, int64_t ld_dout, int64_t col0, const float* whh_f, const float* whh_r,
                               const int32_t* lens, int64_t B, int64_t T_, const float* saved, float* dg_f, float* dg_r, float* dwhh_f,
                               float* dwhh_r, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(dout && whh_f && whh_r && lens && saved && dg_f && dg_r && dwhh_f && dwhh_r && workspace, "lasr_bilstm_bwd: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_bilstm_bwd: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && col0 >= 0 && ld_dout >= col0 + 2 * H, "lasr_bilstm_bwd: shape");
  if (workspace_bytes < lasr_bilstm_bwd_workspace_bytes(B)) return fail(LASR_E_WORKSPACE, "lasr_bilstm_bwd: workspace");
  float* pwhh = reinterpret_cast<float*>(workspace);
  dim3 grid((unsigned)B, 2);
  hipStream_t st = as_stream(stream);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(bilstm_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, ld_dout, col0, whh_f, whh_r, lens, T_, saved,
                       dg_f, dg_r, pwhh);
  else
    hipLaunchKernelGGL(bilstm_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, ld_dout, col0, whh_f, whh_r, lens, T_, saved,
                       dg_f, dg_r, pwhh);
  LASR_LAUNCH_CHECK("bilstm_bwd_kernel");
  // pwhh is [B][2][G*H]: sum over b with a stride of 2*G*H -> view as B partials of 2*G*H columns, split at G*H
  return launch_reduce_partials(pwhh, (int)B, (int64_t)2 * G * H, dwhh_f, (int64_t)G * H, dwhh_r, st);
}

extern "C" int lasr_copy_cols(const void* src, int src_dtype, int64_t ld_src, int64_t scol0, void* dst, int dst_dtype, int64_t ld_dst,
                              int64_t dcol0, int64_t rows, int64_t ncols, int accumulate, void* stream) {
  LASR_CHECK_ARG(src && dst, "lasr_copy_cols: null pointer");
  LASR_CHECK_ARG((src_dtype == LASR_F32 || src_dtype == LASR_BF16) && (dst_dtype == LASR_F32 || dst_dtype == LASR_BF16), "lasr_copy_cols: bad dtype");
  LASR_CHECK_SHAPE(rows > 0 && ncols > 0 && scol0 >= 0 && dcol0 >= 0 && ld_src >= scol0 + ncols && ld_dst >= dcol0 + ncols, "lasr_copy_cols: shape");
  int64_t blocks = cdiv(rows * ncols, 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = as_stream(stream);
#define LASR_CC(TS_, TD_)                                                                                                      \
  hipLaunchKernelGGL((copy_cols_kernel<TS_, TD_>), dim3((unsigned)blocks), dim3(256), 0, st, (const TS_*)src, ld_src, scol0, (TD_*)dst, \
                     ld_dst, dcol0, rows, ncols, accumulate)
  if (src_dtype == LASR_F32 && dst_dtype == LASR_F32) LASR_CC(float, float);
  else if (src_dtype == LASR_F32) LASR_CC(float, bf16_t);
  else if (dst_dtype == LASR_F32) LASR_CC(bf16_t, float);
  else LASR_CC(bf16_t, bf16_t);
#undef LASR_CC
  LASR_LAUNCH_CHECK("copy_cols_kernel");
  return 0;
}
