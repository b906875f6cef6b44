// Internal hand-overs between kernels of neighbouring units (not part of the C ABI).
#pragma once
#include "common.h"

namespace lasr {

// Depthwise forward of unit i+1 (stride 1, bf16) whose input is made on the fly from unit i's GEMM outputs:
//   out = act(y a + b [+ y2 a2 + b2])     (bn_act_fwd_kernel's arithmetic: SeprationConv's BatchNorm, the block's residual add and
//                                          ReLU, models/QuartNet.py:33-37,74-77 - bit-identical to the separate launch)
//   u   = depthwise_conv(out, w)          (models/QuartNet.py:15-19 of the next unit)
// coef / coef2: [2][C] (scale | shift) as lasr_bn_finalize_partials leaves them; y2 / coef2 null: no residual branch.
// Returns 0 (launched), 1 (this shape takes no fused kernel: nothing was launched, run the two launches) or a negative error.
int dwconv_fwd_bn(const void* y, const float* coef, const void* y2, const float* coef2, int act, const float* w, void* out, void* u,
                  int64_t B, int64_t T, int64_t C, int k, void* stream);

// Depthwise tap tables made ONCE per training step (round 4) instead of by every workgroup of every depthwise launch: per layer
// [2 (flip)][C][kDwTapRow] packed bf16 pairs - the layout dwconv_s1_mfma_body keeps in LDS: row = TE[0..79] | TO[0..79] of the
// zero-padded tap row W[i] = w'[i - 24], TE[i] = (W[2i], W[2i+1]), TO[i] = (W[2i+1], W[2i+2]); flip = 1: taps reversed (data gradient).
static constexpr int kDwTapRow = 160;
struct DwTapJobs {                       // up to 16 layers; blk0: first 256-entry block of each layer's table in the builder's grid
  int n; int blk0[17]; const float* w[16]; uint32_t* out[16]; int C[16]; int k[16];
};
__device__ __forceinline__ uint32_t dw_tap_entry(const float* __restrict__ w, int C, int k, int e) {   // e in [0, 2 * C * kDwTapRow)
  const int flip = e / (C * kDwTapRow), r = e - flip * (C * kDwTapRow);
  const int c = r / kDwTapRow, idx = r - c * kDwTapRow;
  const int i0 = idx < kDwTapRow / 2 ? 2 * idx : 2 * (idx - kDwTapRow / 2) + 1;
  const int j0 = i0 - 24, j1 = i0 - 23;
  const float* wc = w + (size_t)c * k;
  const int q0 = min(max(j0, 0), k - 1), q1 = min(max(j1, 0), k - 1);
  const float a0 = wc[flip ? k - 1 - q0 : q0], a1 = wc[flip ? k - 1 - q1 : q1];
  const uint32_t m0 = (j0 >= 0 && j0 < k) ? 0xffffffffu : 0u, m1 = (j1 >= 0 && j1 < k) ? 0xffffffffu : 0u;
  return (uint32_t)f32_to_bf16(__uint_as_float(__float_as_uint(a0) & m0)) | ((uint32_t)f32_to_bf16(__uint_as_float(__float_as_uint(a1) & m1)) << 16);
}
// The tables of the model call in progress on this thread (model.hip sets it around its forward / backward; null outside):
// lasr_dwconv_fwd / dwconv_fwd_bn / lasr_dwconv_bwd_fused look their weight pointer up and hand the table to the kernel, which then
// copies its 64 (32) rows instead of building them.  LASR_DW_TAPS=0 disables the whole path.
struct DwTapCtx { int n; const float* w[16]; const uint32_t* t[16]; int C[16]; };
void dw_taps_set_ctx(const DwTapCtx* ctx);
const uint32_t* dw_taps_for(const float* w, int flip, int64_t C);
bool dw_taps_enabled();

// norm.hip: lasr_mask_lengths_step + lasr_cast_f32_to_bf16 (+ the depthwise tap tables, jobs != null) in one launch - the start of
// every bf16 training forward; 1 = not taken
int mask_lengths_step_cast(const float* pct, int64_t B, int64_t T_, int32_t* lens, uint64_t* step_counter, const float* in, void* out,
                           int64_t n, const DwTapJobs* jobs, void* stream);
// norm.hip: lasr_cast_pad_f32_to_bf16 + lasr_colsum_f32's first stage + lasr_scale_sum_f32 in one launch, then the column sums' second
// stage (the tail of the dense small-vocabulary loss head); 1 = not taken
int head_tail(const float* gl, int64_t rows, int64_t C, void* gl_bf16, int64_t ld_out, float* bias_grad, void* workspace,
              size_t workspace_bytes, const float* nll, int64_t n_nll, float scale, float* loss, void* stream);

// se.hip (round 5): the excite MLP (lasr_se_fwd's two launches) inside the BN + SE + residual add + activation pass, dealt over
// (64-channel slab x utterance) workgroups that each recompute their utterance's hidden vector; bit-identical to the three launches.
// 0 = launched, 1 = not taken (f32, dropout, odd shapes: the caller runs lasr_se_fwd + lasr_bn_act_fwd), negative = error
int bn_se_act_fwd(const void* y, const float* coef, const void* y2, const float* coef2, const float* sums, const float* W1, const float* W2,
                  void* out, float* pooled, float* hidden, float* scale, int dtype, int64_t B, int64_t T_, int64_t C, int act, void* stream);

// ctc.hip: the decoder GEMM's split-K reduction (+ bias) and lasr_log_softmax in one launch, for C <= 64 (log_softmax_split_on)
bool log_softmax_split_on(int64_t C);
int log_softmax_split(const float* partials, int split, const float* bias, float* logits, float* logp, int32_t* argmax, int64_t N, int64_t C,
                      void* stream);

}  // namespace lasr
