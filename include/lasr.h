/* lasr.h — C ABI of liblasr.so: the MI355X (gfx950) QuartzNet-CTC training hot path.
 *
 * The reference (kouyt5/lightning-asr) is pure Python and has NO FFI layer: its hot path runs
 * through torch / torchaudio operator calls.  Each entry point below therefore cites the
 * reference Python call site (path:line under the reference tree) whose arithmetic it replaces;
 * INTEGRATION.md shows the ctypes binding a maintainer adds at that call site.
 *
 * Conventions
 *   - plain C, POD arguments only: raw device pointers, int64 sizes, enums, hipStream_t as void*.
 *   - the CALLER owns every buffer (inputs, outputs, workspaces); the library allocates nothing
 *     on the device and never synchronises it.  All work is enqueued on `stream`.
 *   - return value: 0 = ok, <0 = LASR_E_* (bad argument/shape), >0 = hipError_t passthrough.
 *     lasr_last_error() returns a thread-local message for the last non-zero return.
 *   - activation layout is channels-last: a (B, C, T) reference tensor is stored [B][T][C]
 *     ("rows" n = b*T + t).  dtype of activations: LASR_F32 (parity mode) or LASR_BF16.
 *     Statistics, parameters, log-probs, CTC and the optimiser are always f32.
 */
#ifndef LASR_H
#define LASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LASR_VERSION 102   /* 101: lasr_mel_fwd_src / lasr_wav_read_batch / lasr_step_metrics / lasr_model_set_prefetch_src
                              102: LASR_LEN_LEAD (crop after pre-emphasis), lasr_wav_read_batch(lead_in), lasr_comm_timing* */

enum { LASR_F32 = 0, LASR_BF16 = 1 };
enum { LASR_ACT_NONE = 0, LASR_ACT_RELU = 1, LASR_ACT_SWISH = 2 };
enum { LASR_VARIANT_PLAIN = 0, LASR_VARIANT_CONTEXT = 1, LASR_VARIANT_CONTEXT_SE = 2 };
enum {
  LASR_E_ARG = -1,      /* null pointer / negative size / unsupported enum */
  LASR_E_SHAPE = -2,    /* shape outside what the kernels are built for */
  LASR_E_WORKSPACE = -3 /* workspace too small */
};

int lasr_version(void);
const char* lasr_last_error(void);

/* ---------------------------------------------------------------- features ----------------
 * data_module.py:150-174 AudioParser.parse_audio (torchaudio MelSpectrogram(16000, n_fft=512,
 * pad=32, win_length=320, hop_length=160, n_mels=64) + AmplitudeToDB('power'), :68-71):
 * dither (optional, noise passed in) -> pre-emphasis 0.97 -> |STFT|^2 -> HTK mel(64) ->
 * 10 log10(max(.,1e-10)) -> [SpecAugment rows/cols := 0, :97-122] -> (x-mean)/std (unbiased
 * over the utterance's own 64*T_b values, :171-172) -> zero padding past T_b (collate :222-248).
 */
int64_t lasr_mel_num_frames(int64_t n_samples); /* 1 + (n_samples + 64) / 160 */
size_t lasr_mel_workspace_bytes(int64_t B, int64_t T);
/* wave (B, L) f32, sample_lens (B) int32 valid samples per row (NULL = all L),
 * dither (B, L) f32 N(0,1) noise or NULL, aug (B,4) int32 {rect_x,w_x,rect_y,w_y} or NULL.
 * out_bft: (B, 64, T) f32 reference layout, may be NULL.
 * out_btf: (B, T, 64) channels-last in `dtype`, may be NULL.
 * frames_out (B) int32 = frames per utterance; pct_out (B) f32 = frames/T (collate :243).
 * normalize: 0 = stop at dB (no mean/std), 1 = full chain.                                   */
int lasr_mel_fwd(const float* wave, const int32_t* sample_lens, const float* dither, const int32_t* aug,
                 int64_t B, int64_t L, int normalize, float* out_bft, void* out_btf, int dtype,
                 int32_t* frames_out, float* pct_out, void* workspace, size_t workspace_bytes, void* stream);

/* lasr_mel_fwd with the samples as they sit in the wav file and the dither drawn on the device:
 *   wave_dtype LASR_WAVE_PCM16: wave is (B, L) int16 PCM; the kernel scales by 1/32768 (exact in f32 - the values
 *     torchaudio.load's normalisation hands data_module.py:153), so the H2D copy moves half the bytes;
 *   dither != NULL: explicit (B, L) N(0,1) noise, as lasr_mel_fwd;
 *   dither == NULL and dither_step != NULL: `y += 1e-5 * randn_like(y)` (data_module.py:155) is generated inside the kernel -
 *     Philox4x32-10 keyed by dither_seed with counter (sample / 4, utterance, *dither_step), Box-Muller - and *dither_step
 *     (device scalar, uint64) is incremented by the call, so a step replayed from a hipGraph draws fresh noise;
 *   both NULL: no dither.
 * lasr_dither_noise writes the noise a call with the same (seed, *step) uses: out (B, L) f32 (verification; the counter is
 * left untouched).                                                                                                   */
/* Crop AFTER dither + pre-emphasis (data_module.py:155-159: `y += noise; y = cat(y[0], y[1:] - 0.97 y[:-1]); y = sub_secquence(y)`):
 * the first sample of a crop that starts at file sample loc > 0 is y[loc] - 0.97 y[loc-1], not y[loc].  A caller that crops the RAW
 * waveform hands the sample before the crop over as a lead-in: row b = [x[loc-1], x[loc], ..., x[loc+n-1]] and
 * sample_lens[b] = n | LASR_LEN_LEAD.  The lead-in sample is dithered like every other sample (explicit noise: the row's first noise
 * value; generated noise: the row's first counter) and produces no output sample of its own.  n + 1 <= L.                     */
#define LASR_LEN_LEAD (1 << 30)
enum { LASR_WAVE_F32 = 0, LASR_WAVE_PCM16 = 1 };
/* pitch (round 5): elements between consecutive rows of `wave` (and of `dither`); 0 = L.  With pitch > L the batch sits in rows wider
 * than the L that defines the frame count T = 1 + (L + 64) / 160 (every utterance still ends at its own sample_lens[b] <= L, lead-in
 * sample included in the row: n + 1 <= pitch): a loader may keep ONE row pitch per frame-count class (static shapes for a captured
 * hipGraph) while T stays the reference's "pad to the longest utterance" value (data_module.py:222-248).                        */
typedef struct {
  const void* wave; int32_t wave_dtype; const float* dither; uint64_t dither_seed; uint64_t* dither_step; int64_t pitch;
} lasr_wave_src;
int lasr_mel_fwd_src(const lasr_wave_src* src, const int32_t* sample_lens, const int32_t* aug, int64_t B, int64_t L,
                     int normalize, float* out_bft, void* out_btf, int dtype, int32_t* frames_out, float* pct_out,
                     void* workspace, size_t workspace_bytes, void* stream);
int lasr_dither_noise(uint64_t seed, const uint64_t* step, int64_t B, int64_t L, float* out, void* stream);

/* AudioParser.spec_augment as a stand-alone call (data_module.py:97-122; inside the training chain the same zeros are written by
 * lasr_mel_fwd's `aug`): out = in with rows [rect_x, rect_x + w_x) and frames [rect_y, rect_y + w_y) of every (F, T) plane zeroed;
 * in / out (B, F, T) f32 (out may alias in), aug (B, 4) int32 = (rect_x, w_x, rect_y, w_y) drawn by the caller.        */
int lasr_spec_augment(const float* in, float* out, const int32_t* aug, int64_t B, int64_t F, int64_t T, void* stream);

/* (B, C, T) f32 reference layout -> [B][T][C] channels-last `dtype`   (models/QuartNet.py:154 squeeze) */
int lasr_bct_to_btc(const float* in, void* out, int dtype, int64_t B, int64_t C, int64_t T, void* stream);
int lasr_btc_to_bct(const void* in, int dtype, float* out, int64_t B, int64_t C, int64_t T, void* stream);

/* lens[b] = int32(trunc(f32(T) * pct[b]))   (models/QuartNet.py:311, train.py:76) */
int lasr_mask_lengths(const float* pct, int64_t B, int64_t T, int32_t* lens, void* stream);

/* ---------------------------------------------------------------- conv block pieces -------
 * models/QuartNet.py:29-39 SeprationConv.forward and :71-78 QuartNetBlock.forward.            */

/* depthwise conv1d, groups=C, zero padding k/2, no bias (models/QuartNet.py:19-21,30).
 * x [B][Tin][C] -> y [B][Tout][C], Tout = (Tin + 2*(k/2) - k)/stride + 1.  w (C, k) f32.
 * flip=1 correlates with the time-reversed taps (the data-gradient of a stride-1 conv).
 * addend (same shape as y, may be NULL) is added to the result.                               */
int lasr_dwconv_fwd(const void* x, const float* w, const void* addend, void* y, int dtype, int64_t B,
                    int64_t Tin, int64_t C, int k, int stride, int flip, void* stream);
/* dw (C, k) f32 = sum_{b,t} dy[b,t,c] * x[b, t*stride + j - k/2, c].  workspace: f32 partials. */
size_t lasr_dwconv_wgrad_workspace_bytes(int64_t B, int64_t Tout, int64_t C, int k);
int lasr_dwconv_wgrad(const void* x, const void* dy, float* dw, int dtype, int64_t B, int64_t Tin,
                      int64_t C, int k, int stride, void* workspace, size_t workspace_bytes, void* stream);

/* General GEMM on row-major device matrices, f32 accumulate on MFMA:
 *   C[M][N] = sum_k opA(A)[m][k] * opB(B)[n][k]  (+ bias[n]) (+ addend[m][n])
 * transA=0: A is [M][K] (K contiguous); transA=1: A is [K][M].  transB=0: B is [N][K]; 1: [K][N].
 * dtype_ab / dtype_c: LASR_F32 or LASR_BF16 (A and B share a dtype).
 * row_lens/rows_per_seq: if row_lens != NULL, output row m = b*rows_per_seq + t is written as
 *   zero when t >= row_lens[b]  (MaskCNN, models/QuartNet.py:309-321, applied BEFORE BN).
 * stats (2*N f32, may be NULL): stats[n] = sum_m C[m][n], stats[N+n] = sum_m C[m][n]^2 over the
 *   stored (masked, dtype-rounded) values, summed in a fixed order through `workspace`
 *   (feeds training-mode BatchNorm1d, :35).  Replaces pointwise_conv / reside.0 / last_cnn2.0 /
 *   decoder (models/QuartNet.py:31,63,146,275) and their autograd backward GEMMs.
 * split_k > 1 sums partial products through `workspace` (split_k*M*N f32), deterministic;
 *   it excludes row masking and statistics.                                                     */
size_t lasr_gemm_workspace_bytes(int64_t M, int64_t N, int split_k, int want_stats);
int lasr_gemm(const void* A, const void* B, void* C, int dtype_ab, int dtype_c, int64_t M, int64_t N,
              int64_t K, int transA, int transB, const float* bias, const void* addend,
              const int32_t* row_lens, int64_t rows_per_seq, float* stats, int split_k, void* workspace,
              size_t workspace_bytes, void* stream);

/* One or two independent GEMMs of the same kind (dtypes, transposition, split_k) in ONE launch: a unit's
 * main + residual 1x1 convolution (models/QuartNet.py:31 and :63), their two data gradients or their two
 * weight gradients.  Same semantics per problem as lasr_gemm (no addend).                              */
typedef struct {
  const void* A; const void* B; void* C;
  int64_t M, N, K;
  const float* bias; const int32_t* row_lens; int64_t rows_per_seq; float* stats;
} lasr_gemm_problem;
size_t lasr_gemm_batch_workspace_bytes(const lasr_gemm_problem* probs, int n_probs, int split_k);
int lasr_gemm_batch(const lasr_gemm_problem* probs, int n_probs, int dtype_ab, int dtype_c, int transA, int transB,
                    int split_k, void* workspace, size_t workspace_bytes, void* stream);

/* lasr_gemm with explicit leading dimensions (elements; 0 = the packed default) and no masking / statistics:
 * operands or results that are sub-blocks or row-padded copies of a larger matrix.  bf16 operands take the
 * aligned 16-byte path when the pitch is a multiple of 8 and the base 16-byte aligned, whatever M, N, K. */
int lasr_gemm_ld(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int dtype_ab, int dtype_c,
                 int64_t M, int64_t N, int64_t K, int transA, int transB, const float* bias, int split_k, void* workspace,
                 size_t workspace_bytes, void* stream);

/* Eval-mode BN folded into the 1x1 convs:  for a unit  out = act(BN(mask(W u)) + BN_res(Wr x))  with
 * eval coefficients (a, b), (a2, b2):   out = act([mask(u) | x] . [a W | a2 Wr]^T + (b + b2)).
 *   lasr_fold_bn_weights_many: w_out [co][ci (+ci)] bf16 and bias_out [co] f32 for up to 32 units in one launch
 *       (coef / coef2 = the [a | b] vectors of lasr_bn_eval_coef_many; w_res = coef2 = NULL without a residual branch);
 *   lasr_gemm_dual: C[M][N] bf16 = act([A1 (rows past row_lens zeroed) | A2] . W^T + bias), A1 [M][K1], A2 [M][K2] bf16,
 *       W [N][K1+K2] bf16; K1, K2 multiples of 64, everything 16-byte aligned.  One GEMM, no separate BN pass.   */
typedef struct {
  const float* w; const float* w_res; const float* coef; const float* coef2; void* w_out; float* bias_out; int64_t co, ci;
} lasr_fold_desc;
int lasr_fold_bn_weights_many(const lasr_fold_desc* descs, int n_descs, void* stream);
int lasr_gemm_dual(const void* A1, int64_t K1, const void* A2, int64_t K2, const void* W, const float* bias, void* C,
                   int64_t M, int64_t N, const int32_t* row_lens, int64_t rows_per_seq, int act, void* stream);

/* Deferred reductions.  The split-K weight-gradient GEMMs and the depthwise weight gradient end in a small
 * "sum the partial slabs" kernel each; a backward stage can instead leave the slabs where they are and sum all of
 * them with ONE launch at its end:
 *   lasr_gemm_batch_split_partials: lasr_gemm_batch(split_k > 1, f32 result) without the final sums;
 *       partials[i] -> [splits[i]][M_i*N_i] f32 inside `workspace` (which the caller must keep until the reduce);
 *   lasr_dwconv_wgrad_partials: lasr_dwconv_wgrad without the final sum; `workspace` holds [*n_partials][C*k];
 *   lasr_reduce_many: out[i] = sum_p partials[p*n + i] for up to 64 segments, f64 accumulation, fixed order. */
typedef struct { const float* partials; float* out; int64_t n; int32_t n_partials; } lasr_reduce_desc;
int lasr_reduce_many(const lasr_reduce_desc* descs, int n_descs, void* stream);
int lasr_gemm_batch_split_partials(const lasr_gemm_problem* probs, int n_probs, int dtype_ab, int transA, int transB,
                                   int split_k, void* workspace, size_t workspace_bytes, const float** partials,
                                   int* splits, void* stream);
/* Up to 32 bf16 weight-gradient problems C_i[M_i][N_i] = A_i^T B_i (A_i is [K][M_i], B_i is [K][N_i]: transA = transB = 1)
 * in ONE split-K launch; slabs[i] receives [splits[i]][M_i*N_i] f32 (splits[i] <= split_k), to be summed by
 * lasr_reduce_many.  Batching a whole backward stage makes the K slices longer and the slabs smaller. */
int lasr_gemm_multi_split_partials(const lasr_gemm_problem* probs, int n_probs, int split_k, float* const* slabs,
                                   int* splits, void* stream);
int lasr_dwconv_wgrad_partials(const void* x, const void* dy, int dtype, int64_t B, int64_t Tin, int64_t C, int k,
                               int stride, void* workspace, size_t workspace_bytes, int* n_partials, void* stream);
/* Both consumers of a unit's d(depthwise output) in one launch (stride 1): the weight-gradient partials of
 * lasr_dwconv_wgrad_partials(x, dy) and dx = lasr_dwconv_fwd(dy, w, addend, flip = 1).  Falls back to those two calls
 * for shapes / dtypes the fused MFMA kernel does not take.   (models/QuartNet.py:15-19 backward) */
int lasr_dwconv_bwd_fused(const void* x, const void* dy, const float* w, const void* addend, void* dx, int dtype,
                          int64_t B, int64_t T, int64_t C, int k, void* workspace, size_t workspace_bytes,
                          int* n_partials, void* stream);

/* lasr_gemm_batch that leaves each problem's BN partial sums UNREDUCED in the workspace (no split-K):
 * for every problem with stats != NULL, stat_partials[i] points at [stat_tiles[i]][2][N] f32 inside
 * `workspace` (valid until the workspace is reused) and problem.stats itself is not written.
 * lasr_bn_finalize_partials consumes them: 1 launch instead of 2 reductions + 2 finalizes per unit. */
int lasr_gemm_batch_partials(const lasr_gemm_problem* probs, int n_probs, int dtype_ab, int dtype_c, int transA,
                             int transB, void* workspace, size_t workspace_bytes, const float** stat_partials,
                             int* stat_tiles, void* stream);

/* Training-mode BatchNorm1d(eps) statistics -> affine coefficients (models/QuartNet.py:24,35):
 * mean = s/n, var = q/n - mean^2 (biased); coef[c] = gamma*rstd, coef[C+c] = beta - mean*gamma*rstd;
 * saved[c] = mean, saved[C+c] = rstd; running_mean/var updated with momentum (unbiased var) when
 * non-NULL.  training=0: coefficients from the running statistics, stats ignored.              */
int lasr_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float* coef, float* saved, int64_t C, int64_t n_rows, float eps,
                     float momentum, int training, void* stream);
/* Eval mode (training = 0) coefficients of up to 64 BN layers in one launch: coef[c] = gamma*rstd,
 * coef[C+c] = beta - running_mean*gamma*rstd with rstd = 1/sqrt(running_var + eps); the running statistics are read only. */
typedef struct {
  const float* gamma; const float* beta; const float* running_mean; const float* running_var; float* coef; int64_t C;
} lasr_bn_eval_desc;
int lasr_bn_eval_coef_many(const lasr_bn_eval_desc* descs, int n_descs, float eps, void* stream);

/* Training-mode lasr_bn_finalize for one or two BN layers of the same width (a unit's main and residual
 * branch) straight from GEMM partial sums: fixed-order f64 column reduction + the same arithmetic, one
 * launch.  stats (optional) receives the reduced [sum | sumsq] (2C f32).   (models/QuartNet.py:24,35) */
typedef struct {
  const float* partials; int n_partials;          /* [n_partials][2][C] */
  const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* coef; float* saved; float* stats;
} lasr_bn_branch;
int lasr_bn_finalize_partials(const lasr_bn_branch* branches, int n_branches, int64_t C, int64_t n_rows, float eps,
                              float momentum, void* stream);

/* nn.Dropout(p) of SeprationConv / last_cnn2 (models/QuartNet.py:26,38,149) folded into the BN-apply kernels as a counter-based
 * (Philox4x32-10) mask: forward and both backward passes regenerate it from (seed, *step, unit, element index); no mask tensor
 * exists.  step: device scalar holding the index of the current training forward (lasr_mask_lengths_step bumps it as the first
 * launch of a forward, so replayed graphs draw fresh masks); unit: distinct per layer.  A residual unit drops its MAIN branch
 * before the add (the Dropout at the end of its SeprationConv), first_cnn / last_cnn2 drop after the activation.  Kept
 * elements are scaled by 1/(1-p).  NULL descriptor or p = 0: off (the plain entry points).  lasr_dropout_mask writes the mask
 * itself (1 = kept) for verification against a CPU reference.                                                           */
typedef struct { const uint64_t* step; uint64_t seed; uint32_t unit; float p; } lasr_dropout;
int lasr_mask_lengths_step(const float* pct, int64_t B, int64_t T, int32_t* lens, uint64_t* step_counter, void* stream);
int lasr_dropout_mask(const lasr_dropout* dropout, int64_t n, uint8_t* keep, void* stream);
int lasr_bn_act_fwd_drop(const void* y, const float* coef, const void* y2, const float* coef2, const float* se_scale, void* out,
                         int dtype, int64_t B, int64_t T, int64_t C, int act, const lasr_dropout* dropout, void* stream);
int lasr_bn_act_bwd_stats_drop(const void* dout, const void* y, const float* coef, const float* saved, const void* y2,
                               const float* coef2, const float* saved2, const float* se_scale, const float* se_grad, float* sums,
                               float* sums2, int dtype, int64_t B, int64_t T, int64_t C, int act, const lasr_dropout* dropout,
                               void* workspace, size_t workspace_bytes, void* stream);
int lasr_se_bwd_drop(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                     const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T,
                     int64_t C, int act, const lasr_dropout* dropout, float* seg, float* dW1, float* dW2, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Backward of a ContextSE unit's  out = act(BN(y) * se + BN_res(y2))  (models/QuartNetContextSE.py:19-23,54-57) in TWO passes over
 * (dout, y, y2): per-utterance raw sums -> SE-scale gradient by algebra (gamma*sum(d*xhat) + beta*sum(d)) -> excite-MLP backward
 * (seg_out [B][C], dW1, dW2) -> BN-backward constants -> dy, dy2, dgamma, dbeta.  ysum [B][C] = sum_t y from lasr_seqsum.
 * Replaces lasr_se_bwd + lasr_bn_act_bwd_stats + lasr_bn_act_bwd_apply (three passes) for the SE units.                      */
size_t lasr_bn_se_bwd_workspace_bytes(int64_t B, int64_t T, int64_t C);
int lasr_bn_se_bwd(const void* dout, const void* y, const float* coef, const float* saved, const float* gamma, const float* beta,
                   const void* y2, const float* coef2, const float* saved2, const float* gamma2, const float* se_scale,
                   const float* se_hidden, const float* se_pooled, const float* ysum, const float* W1, const float* W2,
                   const int32_t* row_lens, void* dy, void* dy2, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2,
                   float* dW1, float* dW2, float* seg_out, int dtype, int64_t B, int64_t T, int64_t C, int act,
                   const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream);
int lasr_bn_act_bwd_apply_drop(const void* dout, const void* y, const float* coef, const float* saved, const float* gamma,
                               const void* y2, const float* coef2, const float* saved2, const float* gamma2,
                               const float* se_scale, const float* se_grad, const float* sums, const float* sums2,
                               const int32_t* row_lens, void* dy, void* dy2, float* dgamma, float* dbeta, float* dgamma2,
                               float* dbeta2, int dtype, int64_t B, int64_t T, int64_t C, int act, const lasr_dropout* dropout,
                               void* workspace, size_t workspace_bytes, void* stream);

/* out = act( (y*coef_a + coef_b) * se_scale[b][c] + (y2*coef2_a + coef2_b) )
 * y2/coef2 (residual branch) and se_scale ([B][C] f32) may be NULL.
 * (BN-apply + SE scale + residual add + ReLU: models/QuartNet.py:35-37,74-77; ContextSE :55) */
int lasr_bn_act_fwd(const void* y, const float* coef, const void* y2, const float* coef2,
                    const float* se_scale, void* out, int dtype, int64_t B, int64_t T, int64_t C, int act,
                    void* stream);

/* Backward of the above + BatchNorm backward, two passes over the activations:
 * pass 1 (stats): d = dout * act'(.) ; sums[0..C) = sum d*se, [C..2C) = sum d*se*yhat for branch 1
 *                 and the same for branch 2 in sums2 (se=1 there); yhat = (y-mean)*rstd.
 * pass 2 (apply): dy = gamma*rstd * (d*se - s1/n - yhat*s2/n), rows t >= row_lens[b] zeroed for
 *                 branch 1 only (the residual branch is never masked, models/QuartNet.py:75).
 * The pre-activation is rebuilt from y/y2 and the coefficients, so the forward output is not read.
 * dgamma = s2, dbeta = s1 are written by pass 2 (f32, may be NULL).
 * se_grad ([B][C] f32, may be NULL): extra per-(b,c) gradient added to d*se (SE pooled path).
 * Fused hand-over: call stats with sums = sums2 = NULL and apply with sums = sums2 = NULL and the SAME
 * workspace (lasr_bn_bwd_workspace_bytes): the per-block partial sums stay in the workspace and pass 2
 * reduces them while folding its per-channel constants (one launch less, no f32 round trip).   */
size_t lasr_bn_bwd_workspace_bytes(int64_t B, int64_t T, int64_t C);
int lasr_bn_act_bwd_stats(const void* dout, const void* y, const float* coef,
                          const float* saved, const void* y2, const float* coef2, const float* saved2,
                          const float* se_scale, const float* se_grad, float* sums, float* sums2, int dtype,
                          int64_t B, int64_t T, int64_t C, int act, void* workspace, size_t workspace_bytes,
                          void* stream);
int lasr_bn_act_bwd_apply(const void* dout, const void* y, const float* coef,
                          const float* saved, const float* gamma, const void* y2, const float* coef2,
                          const float* saved2, const float* gamma2, const float* se_scale,
                          const float* se_grad, const float* sums, const float* sums2,
                          const int32_t* row_lens, void* dy, void* dy2, float* dgamma, float* dbeta,
                          float* dgamma2, float* dbeta2, int dtype, int64_t B, int64_t T, int64_t C, int act,
                          void* workspace, size_t workspace_bytes, void* stream);
size_t lasr_bn_bwd_apply_workspace_bytes(int64_t C);   /* folded per-channel constants, 10*C f32 */

/* ---------------------------------------------------------------- SE + BiLSTM context --------
 * models/QuartNetContextSE.py:8-23,55  SELayer(reduction=8, no bias): s = sigmoid(W2 relu(W1 mean_T(BN(y)))),
 * mean over ALL T' frames.  BN is affine per channel, so the squeeze needs only sums[b][c] = sum_t y[b,t,c]
 * (lasr_seqsum) and the BN coefficients; the scale is applied inside lasr_bn_act_fwd (se_scale).          */
int lasr_seqsum(const void* x, int dtype, int64_t B, int64_t T, int64_t C, float* sums, void* stream);
/* pooled (B,C) = coef_a*sums/T + coef_b ; hidden (B,C/8) = relu(W1 pooled) ; scale (B,C) = sigmoid(W2 hidden) */
int lasr_se_fwd(const float* sums, const float* coef, const float* W1, const float* W2, int64_t B, int64_t T, int64_t C,
                float* pooled, float* hidden, float* scale, void* stream);
/* Backward of the excite path: seg (B,C) = gradient reaching every frame of the BN output through the pooled
 * mean (feed it to lasr_bn_act_bwd_* as se_grad), dW1 (C/8,C), dW2 (C,C/8).  C a multiple of 32; the whole batch is
 * handled by two launches whose per-workgroup tables (B x C/8 floats and change) must fit 64 KB of LDS (B <= ~200 at C = 512). */
size_t lasr_se_bwd_workspace_bytes(int64_t B, int64_t C);
int lasr_se_bwd(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T,
                int64_t C, int act, float* seg, float* dW1, float* dW2, void* workspace, size_t workspace_bytes, void* stream);

/* models/QuartNetContext.py:171-173,186-199: pack_padded_sequence -> nn.LSTM(256, 40, bidirectional) ->
 * pad_packed_sequence.  gx_f/gx_r (B,T,160) f32 = x W_ih^T per direction (lasr_gemm; biases are added here),
 * gate order i,f,g,o, reverse direction starts at each utterance's last valid frame, outputs zero for
 * t >= lens[b].  Writes h into columns [col0, col0+80) of a [B][T][ld_out] tensor (the cat() of :173).
 * saved: lasr_bilstm_saved_bytes(B,T) of f32 state for the backward pass.                                  */
size_t lasr_bilstm_saved_bytes(int64_t B, int64_t T);
int lasr_bilstm_fwd(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const float* bih_f,
                    const float* bhh_f, const float* bih_r, const float* bhh_r, const int32_t* lens, int64_t B, int64_t T,
                    void* out, int dtype, int64_t ld_out, int64_t col0, float* saved, void* stream);
/* dout: columns [col0, col0+80) of a [B][T][ld_dout] gradient.  dg_f/dg_r (B,T,160) f32 = gradient w.r.t. the gate
 * pre-activations (=> dW_ih = dg^T x, db = colsum(dg), dx = dg W_ih by lasr_gemm), dwhh_f/dwhh_r (160,40).  */
size_t lasr_bilstm_bwd_workspace_bytes(int64_t B);
int lasr_bilstm_bwd(const void* dout, int dtype, int64_t ld_dout, int64_t col0, const float* whh_f, const float* whh_r,
                    const int32_t* lens, int64_t B, int64_t T, const float* saved, float* dg_f, float* dg_r, float* dwhh_f,
                    float* dwhh_r, void* workspace, size_t workspace_bytes, void* stream);
/* dst[n][dcol0+c] (+)= src[n][scol0+c], c < ncols, with dtype conversion (torch.cat / its backward slice, :173) */
int lasr_copy_cols(const void* src, int src_dtype, int64_t ld_src, int64_t scol0, void* dst, int dst_dtype, int64_t ld_dst,
                   int64_t dcol0, int64_t rows, int64_t ncols, int accumulate, void* stream);

/* ---------------------------------------------------------------- head + loss --------------
 * models/QuartNet.py:287-290 log_softmax over classes; train.py:76-78,196 CTCLoss(blank=C-1,
 * reduction='none', zero_infinity=False) and the batch mean; utils/asr_metrics.py:138-171.     */

/* logits [N][C] f32 -> logp [N][C] f32 (may alias logits), argmax (N) int32 (may be NULL).
 * Ties resolve to the lowest class id, as torch.argmax does on CPU.                            */
int lasr_log_softmax(const float* logits, float* logp, int32_t* argmax, int64_t N, int64_t C, void* stream);
/* grad_logits = grad_logp - exp(logp) * sum_c grad_logp   (autograd of F.log_softmax) */
int lasr_log_softmax_bwd(const float* logp, const float* grad_logp, float* grad_logits, int64_t N, int64_t C,
                         void* stream);

size_t lasr_ctc_workspace_bytes(int64_t B, int64_t T, int64_t S_max);
/* logp (B, T, C) f32 log-probs; targets (B, S_max) int64 zero padded; in_lens/tgt_lens (B) int32.
 * nll (B) f32 per-sample negative log-likelihood (+inf when infeasible).
 * If grad != NULL: grad (B,T,C) f32 = gscale[b] * (exp(logp) - occupancy): exactly what torch's
 *   CTCLoss backward returns for grad_output = gscale.  Its class-sum is 0, so log_softmax backward
 *   maps it to itself: it is also d/d(logits).  Rows t >= in_lens[b] are zero; rows of an infeasible
 *   sample are NaN (zero_infinity=False).  gscale (B) f32 or NULL (= 1/B: batch mean, train.py:77).
 *   S_max <= 511 (lattice of 2S+1 states held 4/8/16 per lane of one wave).                       */
int lasr_ctc_loss(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens,
                  int64_t B, int64_t T, int64_t C, int64_t S_max, int blank, float* nll, float* grad,
                  const float* gscale, void* workspace, size_t workspace_bytes, void* stream);
/* lasr_ctc_loss for one batch and lasr_mel_fwd for ANOTHER batch's waveforms in one launch sequence: the lattice kernel
 * keeps 32 workgroups busy for ~0.1 ms of dependent steps, the feature transform (2 016 workgroups) fills the other CUs
 * meanwhile (one grid: lattice workgroups first).  Same results as the two calls; falls back to them for shapes the
 * fused grid does not take (more than 127 labels, C % 4 != 0, emissions over 78 KB).  This is the data-loader prefetch of
 * data_module.py's workers: the features of step i+1 are produced while step i computes its loss. */
int lasr_ctc_loss_mel(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens,
                      int64_t B, int64_t T, int64_t C, int64_t S_max, int blank, float* nll, float* grad,
                      const float* gscale, void* ctc_workspace, size_t ctc_workspace_bytes, const float* wave,
                      const int32_t* sample_lens, const float* dither, const int32_t* aug, int64_t Bm, int64_t L,
                      int normalize, float* out_bft, void* out_btf, int dtype, int32_t* frames_out, float* pct_out,
                      void* mel_workspace, size_t mel_workspace_bytes, void* stream);

/* Greedy CTC collapse of argmax ids (B, T) int32 truncated to lens (B) (NULL = T):
 * tokens (B, T) int32, n_tokens (B) int32.   utils/asr_metrics.py:159-166                      */
int lasr_greedy_decode(const int32_t* ids, const int32_t* lens, int64_t B, int64_t T, int blank,
                       int32_t* tokens, int32_t* n_tokens, void* stream);

/* ---------------------------------------------------------------- optimiser ----------------
 * scheduler/novograd.py:75-145 with betas=(0.8,0.5), eps=1e-8, no amsgrad/grad_averaging/luc
 * (train.py:46), applied to all tensors in one pass over flat f32 buffers.
 * offsets (n_tensors+1) int64 element offsets into params/grads/exp_avg; exp_avg_sq (n_tensors) f32
 * (0 = "not initialised", scheduler/novograd.py:115).  lr is read from the device (1 f32) so a
 * captured graph can replay it.  grad_scale multiplies grads first (1/world after all-reduce).   */
size_t lasr_novograd_workspace_bytes(int64_t n_tensors, int64_t n_elems);
int lasr_novograd_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                       const int64_t* offsets, int64_t n_tensors, int64_t n_elems, const float* lr, float beta1, float beta2,
                       float eps, float weight_decay, float grad_scale, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The same step for a caller that KEEPS its workspace: zero it once before the first call (the per-tensor norm accumulators live
 * there); every call leaves it zeroed, so the memset launch of lasr_novograd_step is not issued (round 5).                  */
int lasr_novograd_step_keep(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                            const int64_t* offsets, int64_t n_tensors, int64_t n_elems, const float* lr, float beta1, float beta2,
                            float eps, float weight_decay, float grad_scale, void* workspace,
                            size_t workspace_bytes, void* stream);

/* roctx ranges (round 5; SURVEY 5 aux "tracing"): with LASR_ROCTX=1 in the environment the plan brackets its stages and units with
 * named ranges - "lasr:forward", "lasr:fwd <unit>", "lasr:head", "lasr:backward", "lasr:bwd <unit>", "lasr:wgrad flush", "lasr:reduce" -
 * through librocprofiler-sdk-roctx (dlopen'd on first use; a missing library switches the ranges off), so that
 * `rocprofv3 --marker-trace --kernel-trace` attributes kernels to units.  The host framework adds its own with the two calls below
 * (lightning_asr_amd/step.py: "lasr:step", "lasr:optimizer").  Host-side only: nothing is enqueued on any stream.                */
int lasr_roctx_range_push(const char* name);
int lasr_roctx_range_pop(void);
int lasr_roctx_enabled(void);

/* ---------------------------------------------------------------- measurement --------------
 * Optional per-kernel-class timing with HIP events recorded on the launch stream (bench.py's
 * roofline leg; off by default, costs two event records per instrumented launch when on).
 * lasr_prof_collect synchronises the recorded events and returns, per class: summed milliseconds,
 * summed algorithmic FLOPs and bytes (operands + result, each counted once) and launch count.     */
#define LASR_PROF_KINDS 5
enum { LASR_PROF_GEMM = 0,    /* every 1x1-conv GEMM launch (forward, data gradient, weight gradient, decoder) */
       LASR_PROF_DWCONV = 1,  /* depthwise conv forward / fused backward */
       LASR_PROF_BN = 2,      /* BatchNorm finalize + apply (+SE) forward, statistics + apply backward: one bracket per unit and direction */
       LASR_PROF_HEAD = 3,    /* log_softmax / CTC lattice + gradient (+ the next batch's features in its grid) / bias sums / loss mean */
       LASR_PROF_OTHER = 4 }; /* weight casts, length masks, deferred reductions, BiLSTM, column copies */
int lasr_prof_enable(int on);
int lasr_prof_collect(double* ms, double* flops, double* bytes, int64_t* count);
/* mean elapsed ms of n empty event pairs on `stream`: the bracketing overhead included in every ms[] entry */
int lasr_prof_overhead_ms(void* stream, int n, double* ms_per_pair);

/* Levenshtein distance between two token-id sequences, on the HOST (plain C++, no device work):
 * replaces editdistance.eval at utils/asr_metrics.py:54,220.  Returns -1 on bad arguments.       */
int64_t lasr_edit_distance(const int32_t* a, int64_t na, const int32_t* b, int64_t nb);

/* small helpers used by the plan and by the host */
int lasr_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
/* out[r][0..cols) = bf16(in[r][:]), out[r][cols..ld_out) = 0: row-padded bf16 copy (ld_out % 8 == 0 keeps rows 16-byte aligned) */
int lasr_cast_pad_f32_to_bf16(const float* in, void* out, int64_t rows, int64_t cols, int64_t ld_out, void* stream);
size_t lasr_colsum_workspace_bytes(int64_t rows, int64_t C);
int lasr_colsum_f32(const float* x, float* out, int64_t rows, int64_t C, void* workspace, size_t workspace_bytes,
                    void* stream);                                   /* out[c] = sum_r x[r][c] (decoder bias grad) */
int lasr_scale_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream); /* torch.mean, train.py:77 */

/* ---------------------------------------------------------------- whole model ---------------
 * MyModel2 (models/QuartNet.py:264-291; QuartNetContext.py:202; QuartNetContextSE.py:220) driven
 * as one native plan: the library sequences every kernel of forward, loss and backward on
 * `stream` from a single call, over caller-owned flat buffers.                                  */
typedef struct lasr_model lasr_model_t;

typedef struct {
  int32_t variant;   /* LASR_VARIANT_* */
  int32_t n_class;   /* len(labels)+1; blank = n_class-1 */
  int32_t in_c;      /* 64 */
  int32_t mask;      /* model.mask (conf/conf.yaml:36) */
  int32_t act;       /* LASR_ACT_RELU (reference) or LASR_ACT_SWISH */
  int32_t dtype;     /* activation dtype */
} lasr_model_config;

int lasr_model_create(const lasr_model_config* cfg, lasr_model_t** out);
void lasr_model_destroy(lasr_model_t* m);
/* One-shot feature prefetch: the next lasr_model_loss_backward[_partial] call on this model computes its CTC loss through
 * lasr_ctc_loss_mel with these lasr_mel_fwd arguments (channels-last output only), then forgets the request.  All
 * pointers are the caller's device buffers and must stay valid until that call has been enqueued. */
int lasr_model_set_prefetch(lasr_model_t* m, const float* wave, const int32_t* sample_lens, const float* dither,
                            const int32_t* aug, int64_t B, int64_t L, int normalize, void* out_btf, int dtype,
                            int32_t* frames_out, float* pct_out, void* mel_workspace, size_t mel_workspace_bytes);
/* the same with a lasr_wave_src (int16 PCM, dither generated on the device) */
int lasr_model_set_prefetch_src(lasr_model_t* m, const lasr_wave_src* src, const int32_t* sample_lens, const int32_t* aug,
                                int64_t B, int64_t L, int normalize, void* out_btf, int dtype, int32_t* frames_out,
                                float* pct_out, void* mel_workspace, size_t mel_workspace_bytes);
int lasr_model_clear_prefetch(lasr_model_t* m);   /* forget an armed request */
/* Tensors in reference state_dict order.  kind: 0 = parameter (lives in the flat param buffer at
 * `offset` elements), 1 = f32 buffer (running_mean/var, flat buffer array), 2 = num_batches_tracked
 * (int64, kept by the host).  Returns the number of tensors; fills row i when i >= 0.            */
/* nn.Dropout(p = drop_rate) of models/QuartNet.py:26,38,149 in every training forward / backward of this model (see
 * lasr_dropout above): step_counter is a caller-owned device scalar (uint64, zero-initialised); p = 0 switches it off.   */
int lasr_model_set_dropout(lasr_model_t* m, float p, uint64_t seed, uint64_t* step_counter);
int64_t lasr_model_tensor_info(const lasr_model_t* m, int64_t i, char* name, size_t name_cap,
                               int64_t shape[4], int32_t* ndim, int32_t* kind, int64_t* offset);
int64_t lasr_model_param_elems(const lasr_model_t* m);
int64_t lasr_model_buffer_elems(const lasr_model_t* m);
int64_t lasr_model_out_frames(const lasr_model_t* m, int64_t T_in);
size_t lasr_model_workspace_bytes(lasr_model_t* m, int64_t B, int64_t T_in, int64_t S_max);
/* Named intermediate ("tap": unit name, "<unit>.y", "<unit>.y2", "<unit>.u", "<unit>.se_hidden" (f32 [B][1][C/8]), "ctx_in", "logits", "grad_logits", "lens";
 * after a staged backward call "bwd.g_cur" = d(input of the last unit processed), "bwd.g_prev" = d(its output), both
 * [N][c] at the head of an [N][cmax] allocation) inside the workspace: returns its byte offset, or -1.            */
int64_t lasr_model_tap(lasr_model_t* m, const char* name, int64_t B, int64_t T_in, int64_t S_max, int64_t shape[3]);

/* feats: [B][T_in][in_c] channels-last in cfg.dtype (from lasr_mel_fwd / lasr_bct_to_btc).
 * logp_out (B, T', n_class) f32; argmax_out (B, T') int32 or NULL.  training=1 uses batch
 * statistics, updates `buffers`, and keeps what backward needs in the workspace.                */
int lasr_model_forward(lasr_model_t* m, const float* params, float* buffers, const void* feats,
                       const float* pct, int64_t B, int64_t T_in, int training, float* logp_out,
                       int32_t* argmax_out, void* workspace, size_t workspace_bytes, void* stream);
/* After a training forward in the same workspace: logp (B,T',C) as returned by it, grad_logp = dL/d logp
 * -> grads (flat f32, same layout as params; every element overwritten).                          */
int lasr_model_backward(lasr_model_t* m, const float* params, const void* feats, const float* logp,
                        const float* grad_logp, int64_t B, int64_t T_in, float* grads, void* workspace,
                        size_t workspace_bytes, void* stream);
/* forward + mean CTC (train.py:75-78) + backward in one call.  loss_out (1) f32, nll_out (B) f32. */
int lasr_model_loss_backward(lasr_model_t* m, const float* params, float* buffers, const void* feats,
                             const float* pct, const int64_t* targets, const int32_t* tgt_lens, int64_t B,
                             int64_t T_in, int64_t S_max, float* logp_out, float* loss_out, float* nll_out,
                             int32_t* argmax_out, float* grads, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Backward in stages, so the host can start the RCCL all-reduce of a gradient bucket while the units
 * below it are still being differentiated (Lightning DDP's bucketed overlap, conf/conf.yaml:30).
 * Units are numbered in forward order (lasr_model_unit_info gives their names).  _partial runs forward,
 * loss, the decoder and units [unit_stop, n); each _continue call runs units [unit_stop, previous stop).
 * Gradients of a unit's parameters are final once the call that covers the unit has been enqueued.      */
int64_t lasr_model_num_units(const lasr_model_t* m);
int lasr_model_unit_info(const lasr_model_t* m, int64_t i, char* name, size_t name_cap);
int lasr_model_loss_backward_partial(lasr_model_t* m, const float* params, float* buffers, const void* feats,
                                     const float* pct, const int64_t* targets, const int32_t* tgt_lens, int64_t B,
                                     int64_t T_in, int64_t S_max, float* logp_out, float* loss_out, float* nll_out,
                                     int32_t* argmax_out, float* grads, void* workspace, size_t workspace_bytes,
                                     int64_t unit_stop, void* stream);
int lasr_model_backward_continue(lasr_model_t* m, const float* params, const void* feats, int64_t B, int64_t T_in,
                                 float* grads, void* workspace, size_t workspace_bytes, int64_t unit_stop, void* stream);

/* ---- learning-rate schedule on the device -------------------------------------------------------------------------------
 * CosineAnnealingWarmupRestarts (scheduler/cosine_annearing_with_warmup.py:53-89; stepped per batch, train.py:57-61) as one
 * device-side state + a one-thread kernel: lasr_lr_schedule_init fills a HOST image of the state (the caller uploads it),
 * lasr_lr_schedule_step advances it and writes the rate lasr_novograd_step reads.  No host scalar enters a training step,
 * so the step can be captured into a hipGraph and replayed.                                                               */
size_t lasr_lr_schedule_state_bytes(void);
int lasr_lr_schedule_init(void* state_host_out, size_t bytes, int64_t first_cycle_steps, double cycle_mult, double max_lr,
                          double min_lr, int64_t warmup_steps, double gamma, int64_t cycle, int64_t step_in_cycle,
                          int64_t cur_cycle_steps, int64_t last_epoch);
int lasr_lr_schedule_step(void* state_dev, float* lr_dev, void* stream);

/* ---- large-vocabulary loss head (BASELINE cfg5: C = 4334): decoder 1x1 conv + log_softmax + CTC + their backward without the
 * (B, T', C) f32 log-prob / gradient tensors of models/QuartNet.py:275-290 and train.py:76-78 (444 MB each at bs = 32).
 * lasr_gemm_rowstat: C [M][ldc] bf16 = A [M][K] . B [N][K]^T + bias plus, per (row, 256-column tile), the softmax statistics
 *   of the STORED values: row_stat [M][tiles][2] = (max, sum exp(x - max)), row_arg [M][tiles] = first argmax column.
 * lasr_ctc_loss_lean: from those, lse and argmax per row, the lattice over the gathered target / blank emissions only, and
 *   grad [B*T][ldc] bf16 = gscale_b * d nll_b / d logits (1/B when gscale is NULL; zero rows past in_lens, NaN for an
 *   infeasible utterance like torch), bias_grad (C) f32 = column sums of the unrounded gradient.  ldc = C rounded up to 8. */
int lasr_gemm_rowstat(const void* A, const void* B, const float* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                      float* row_stat, int32_t* row_arg, int* n_col_tiles, void* stream);
size_t lasr_gemm_rowstat_bytes(int64_t M, int64_t N);
size_t lasr_ctc_lean_workspace_bytes(int64_t B, int64_t T, int64_t C, int64_t S_max);
int lasr_ctc_loss_lean(const void* logits, int64_t ldc, const float* row_stat, const int32_t* row_arg, int n_col_tiles,
                       const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T, int64_t C,
                       int64_t S_max, int blank, float* nll, int32_t* argmax, void* grad, float* bias_grad, const float* gscale,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Levenshtein distances of a batch ON THE DEVICE (utils/asr_metrics.py:26-59,187-228: editdistance.eval per utterance on
 * the host): hyp_tokens [B][ld_hyp] i32 / hyp_lens as written by lasr_greedy_decode, ref_tokens [B][ld_ref] i64 / ref_lens
 * as the collate's targets / target_sizes.  space_id < 0: units are token ids (CER; the reference's file-path
 * vocabularies); space_id >= 0: units are words = runs of tokens between space tokens (str.split()).
 * dist[b], ref_units[b] (B) i32; totals (may be NULL): totals[0] += sum dist, totals[1] += sum ref_units (the metric's
 * `scores` / `words` states).  At most 2048 tokens per utterance and side.                                               */
int lasr_edit_distance_batch(const int32_t* hyp_tokens, const int32_t* hyp_lens, int64_t ld_hyp, const int64_t* ref_tokens,
                             const int32_t* ref_lens, int64_t ld_ref, int64_t B, int space_id, int32_t* dist, int32_t* ref_units,
                             int64_t* totals, void* stream);

/* One training step's logged scalars (train.py:79-81: self.log('train_loss', loss) / self.log('train_wer', wer), on_step +
 * on_epoch) folded into DEVICE accumulators, so logging costs no D2H per step: with wer = sum(dist) / sum(ref_units) of the batch
 * (utils/asr_metrics.py:225-228; inf when the batch has no reference units),
 *   acc[0] += loss, acc[1] += wer, acc[2] += 1, acc[3] = loss, acc[4] = wer, acc[5] += sum(dist), acc[6] += sum(ref_units)
 * (7 doubles).  The host reads acc when it logs (every log_every_n_steps steps and at the end of the epoch).               */
int lasr_step_metrics(const float* loss, const int32_t* dist, const int32_t* ref_units, int64_t B, double* acc, void* stream);

/* ---- host ingest: wav files -> one int16 batch buffer -------------------------------------------------------------------
 * Replaces torchaudio.load in the reference's DataLoader workers (data_module.py:153; conf/conf.yaml:14 num_worker) and the
 * training-time random sub-sequence (data_module.py:138-148,158-159): n 16-bit PCM RIFF/WAVE files are decoded by up to
 * n_threads host threads straight into `out` (typically a pinned buffer) as rows of channel-0 samples,
 *   out[i][0 .. lens_out[i]) = the file's samples (or its crop), zeros up to the row pitch *ld_out = max length rounded up to 8,
 * ready for ONE H2D copy and lasr_mel_fwd_src(LASR_WAVE_PCM16).  crop_u (n, 2) doubles in [0,1) or NULL: per file,
 *   target = int(length * (crop_weight + (1 - crop_weight) * u0)); first = int(u1 * (length - target)); slice [first, target)
 * - the reference's sub_secquence, slice end included as it is there.  expect_rate > 0: fail on another sample rate.
 * No device work; returns LASR_E_WORKSPACE when n * ld exceeds out_capacity (elements).                                 */
int lasr_wav_info(const char* path, int64_t* n_frames, int32_t* n_channels, int32_t* sample_rate, int32_t* bits);
/* lead_in != 0: a crop that does not begin at the file's first sample is preceded in its row by the sample before it and its
 *   lens_out entry carries LASR_LEN_LEAD (see above: the reference crops AFTER pre-emphasis); the pitch counts the lead-in. */
int lasr_wav_read_batch(const char* const* paths, int64_t n, const double* crop_u, double crop_weight, int16_t* out,
                        int64_t out_capacity, int64_t* ld_out, int32_t* lens_out, int32_t expect_rate, int n_threads, int lead_in);

/* ---- data-parallel gradient exchange: RCCL over xGMI, called by the library itself ------------------------------
 * Replaces the NCCL all-reduce the reference gets from Lightning's DDP plugin (conf/conf.yaml:30-31 `accelerator: ddp`,
 * train.py:239): SUM over ranks of the flat f32 gradient, bucket by bucket while backward is still
 * running, plus the wrap-time broadcast of parameters and buffers from rank 0.
 * One communicator per process (one process per GPU).  Collectives run on a library-owned side stream:
 *   lasr_comm_allreduce / _ranges / lasr_comm_broadcast   wait (event) for everything enqueued on `producer_stream` so far,
 *                                                         then run in place on the side stream;
 *   lasr_comm_wait                                        makes `consumer_stream` wait for every collective issued so far.
 * No call synchronises the host.  Positive return codes 1000+n carry ncclResult_t n.
 * Bootstrap: rank 0 calls lasr_comm_unique_id and the host carries the LASR_COMM_ID_BYTES bytes to the other ranks by
 * whatever rendez-vous it has (torch.distributed's store, MPI, a file); every rank then calls lasr_comm_init.        */
#define LASR_COMM_ID_BYTES 128
typedef struct lasr_comm lasr_comm_t;
int lasr_comm_unique_id(void* id_out, size_t id_bytes);
int lasr_comm_init(lasr_comm_t** out, const void* unique_id, size_t id_bytes, int world, int rank, int device);
int lasr_comm_destroy(lasr_comm_t* comm);
int lasr_comm_world(const lasr_comm_t* comm);
int lasr_comm_rank(const lasr_comm_t* comm);
int lasr_comm_allreduce(lasr_comm_t* comm, float* buf, int64_t count, void* producer_stream);
/* one bucket made of several pieces of the flat buffer (ncclGroupStart/End: one fused launch) */
int lasr_comm_allreduce_ranges(lasr_comm_t* comm, float* base, const int64_t* lo, const int64_t* hi, int n_ranges,
                               void* producer_stream);
int lasr_comm_broadcast(lasr_comm_t* comm, float* buf, int64_t count, int root, void* producer_stream);
int lasr_comm_wait(lasr_comm_t* comm, void* consumer_stream);
/* LASR_COMM_MAX_CHANNELS=n (> 0) makes lasr_comm_init set NCCL_MAX_NCHANNELS=n (the persistent workgroups a collective keeps resident
 * on the CUs it shares with the backward) unless that variable is already set; default 0 = RCCL's own choice (measured: the
 * interference grows with the LENGTH of the exchange, not with the CUs it holds - DESIGN 5).
 * Timing of the exchange (bench.py's `comm` record; eager launches only): with timing on, every collective is bracketed by events on
 * the side stream and every lasr_comm_wait by events on the consumer stream.  lasr_comm_timing_collect synchronises them and returns,
 * in call order, coll_us[i] / coll_bytes[i] (duration and payload of collective i, peers' arrival included) and wait_us[j] (how long
 * consumer stream j-th wait actually stalled: the part of the exchange backward did not hide); at most max_recs entries are written,
 * *n_coll / *n_waits are the numbers recorded since timing was switched on.                                                      */
int lasr_comm_timing(lasr_comm_t* comm, int on);
int lasr_comm_timing_collect(lasr_comm_t* comm, int max_recs, double* coll_us, double* coll_bytes, int* n_coll, double* wait_us,
                             int* n_waits);

#ifdef __cplusplus
}
#endif
#endif /* LASR_H */
