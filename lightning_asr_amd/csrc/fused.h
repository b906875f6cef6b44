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

}  // namespace lasr
