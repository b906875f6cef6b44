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
  int64_t pitch = 0;                   // elements between the rows of wave / dither; 0: L.  (L then only defines T = 1 + (L + 64) / 160: a batch
                                       // may sit in rows wider than its longest utterance - a row pitch that is the same for every
                                       // batch of a frame-count class - without an extra frame of padding entering T)
};

// a feature transform waiting to ride in the grid of a CTC lattice launch (the prefetch of the next batch's features)
struct MelJob {
  WaveSrc src; const int32_t* sample_lens; const int32_t* aug; int64_t B, L; int normalize;
  void* out_btf; int dtype; int32_t* frames_out; float* pct_out; void* ws; size_t ws_bytes;
};
// the compact (gathered-emission) lattice of the large-vocabulary head + `job` in one grid; falls back to nothing: the caller
// checks `fits` first (emissions + labels within the LDS of one workgroup)
bool compact_lattice_mel_fits(int64_t T, int64_t CE);
int launch_compact_lattice_mel(const float* E, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T,
                               int64_t CE, int64_t S_max, int blank_col, float* alpha, float* beta, int32_t* next_same, float* nll, int ns,
                               const MelJob& job, void* stream);
// lasr_ctc_loss_lean with an optional MelJob (ctc_lean.hip)
int ctc_loss_lean_job(const void* logits, int64_t ldc, const float* row_stat, const int32_t* row_arg, int n_col_tiles,
                      const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T, int64_t C,
                      int64_t S_max, int blank, float* nll, int32_t* argmax, void* grad, float* bias_grad, const float* gscale,
                      void* workspace, size_t workspace_bytes, const MelJob* job, void* stream);

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
