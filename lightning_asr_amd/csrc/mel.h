// Internal interface of the feature front-end (mel.hip) for the plan (model.hip): where the samples come from.
#pragma once
#include "common.h"

namespace lasr {

// lasr_wave_src (include/lasr.h) as the kernels see it
struct WaveSrc {
  const void* wave;                    // (B, L) f32 in [-1, 1) or int16 PCM
  int pcm16;
  const float* dither;                 // (B, L) N(0,1) noise, or null
  const unsigned long long* dstep;     // null: no generated noise.  Otherwise (and dither == null) noise is generated in the kernel
  unsigned long long dseed;
};

int wave_src_from_c(const lasr_wave_src* s, WaveSrc* out, const char* who);
int mel_fwd_src(const WaveSrc& src, const int32_t* sample_lens, const int32_t* aug, int64_t B, int64_t L, int normalize,
                float* out_bft, void* out_btf, int dtype, int32_t* frames_out, float* pct_out, void* workspace,
                size_t workspace_bytes, void* stream);
int ctc_loss_mel_src(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T,
                     int64_t C, int64_t S_max, int blank, float* nll, float* grad, const float* gscale, void* ctc_workspace,
                     size_t ctc_workspace_bytes, const WaveSrc& src, const int32_t* sample_lens, const int32_t* aug, int64_t Bm,
                     int64_t L, int normalize, float* out_bft, void* out_btf, int dtype, int32_t* frames_out, float* pct_out,
                     void* mel_workspace, size_t mel_workspace_bytes, void* stream);

}  // namespace lasr
