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

// norm.hip: lasr_mask_lengths_step + lasr_cast_f32_to_bf16 in one launch (the start of every bf16 training forward); 1 = not taken
int mask_lengths_step_cast(const float* pct, int64_t B, int64_t T_, int32_t* lens, uint64_t* step_counter, const float* in, void* out,
                           int64_t n, void* stream);
// norm.hip: lasr_cast_pad_f32_to_bf16 + lasr_colsum_f32's first stage + lasr_scale_sum_f32 in one launch, then the column sums' second
// stage (the tail of the dense small-vocabulary loss head); 1 = not taken
int head_tail(const float* gl, int64_t rows, int64_t C, void* gl_bf16, int64_t ld_out, float* bias_grad, void* workspace,
              size_t workspace_bytes, const float* nll, int64_t n_nll, float scale, float* loss, void* stream);

// ctc.hip: the decoder GEMM's split-K reduction (+ bias) and lasr_log_softmax in one launch, for C <= 64 (log_softmax_split_on)
bool log_softmax_split_on(int64_t C);
int log_softmax_split(const float* partials, int split, const float* bias, float* logits, float* logp, int32_t* argmax, int64_t N, int64_t C,
                      void* stream);

}  // namespace lasr
