// log_softmax(+argmax), wave-synchronous CTC forward/backward and greedy CTC collapse.
// Replaces F.log_softmax (models/QuartNet.py:290), nn.CTCLoss(blank=C-1, reduction='none',
// zero_infinity=False) + its backward (train.py:77-78,196) and the Python greedy decoder
// (utils/asr_metrics.py:155-166).
//
// CTC lattice: one workgroup of two waves per utterance; wave 0 runs the alpha recursion forward
// in time while wave 1 runs beta backward.  Each lane owns NS consecutive lattice states, the
// s-1 / s-2 (s+1 / s+2) neighbours cross lanes with one shuffle each, so a time step needs no
// LDS and no barrier; the emission log-probs of step t+1 are fetched while step t is computed.
#include "ctc_lattice.h"
#include "fused.h"

namespace lasr {

// ------------------------------------------------------------------ log_softmax + argmax ------
// one wave per row
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ logits, float* __restrict__ logp,
                                                          int32_t* __restrict__ argmax, int64_t N, int64_t C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* x = logits + row * C;
  float m = kNegInf;
  int64_t mi = 0x7fffffff;
  for (int64_t c = lane; c < C; c += 64) {
    const float v = x[c];
    if (v > m || (v == m && c < mi)) { m = v; mi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int64_t oi = __shfl_xor((long long)mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float s = 0.f;
  for (int64_t c = lane; c < C; c += 64) s += expf(x[c] - m);
  s = wave_sum(s);
  const float lse = logf(s);
  float* y = logp + row * C;
  for (int64_t c = lane; c < C; c += 64) y[c] = (x[c] - m) - lse;
  if (argmax && lane == 0) argmax[row] = (int32_t)mi;
}

// The same for a narrow head (C <= 64: one class per lane) straight from the decoder GEMM's split-K slabs: logits = sum of the slabs
// (in slab order, as gemm_split_reduce_kernel sums them) + bias, written out for the taps, then the row's log_softmax and argmax in
// registers - the split reduction's own launch and one round trip of the logits are gone; the arithmetic is log_softmax_kernel's.
__global__ __launch_bounds__(256) void log_softmax_split_kernel(const float* __restrict__ ws, int split, const float* __restrict__ bias,
                                                                float* __restrict__ logits, float* __restrict__ logp,
                                                                int32_t* __restrict__ argmax, int64_t N, int64_t C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const bool live = lane < C;
  const int64_t i = row * C + (live ? lane : 0);
  float v = 0.f;
  for (int p = 0; p < split; ++p) v += ws[(int64_t)p * N * C + i];
  if (bias) v += bias[live ? lane : 0];
  if (live) logits[i] = v;
  float m = live ? v : kNegInf;
  int64_t mi = live ? (int64_t)lane : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int64_t oi = __shfl_xor((long long)mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float s = live ? expf(v - m) : 0.f;
  s = wave_sum(s);
  const float lse = logf(s);
  if (live) logp[i] = (v - m) - lse;
  if (argmax && lane == 0) argmax[row] = (int32_t)mi;
}

// g_logits = g - exp(logp) * sum_c g   (general log_softmax backward), one wave per row
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ logp, const float* __restrict__ g,
                                                              float* __restrict__ out, int64_t N, int64_t C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float s = 0.f;
  for (int64_t c = lane; c < C; c += 64) s += g[row * C + c];
  s = wave_sum(s);
  for (int64_t c = lane; c < C; c += 64) out[row * C + c] = g[row * C + c] - expf(logp[row * C + c]) * s;
}

template <int NS, bool EM_LDS>
__global__ __launch_bounds__(128) void ctc_alpha_beta_kernel(const float* __restrict__ logp, const int64_t* __restrict__ targets,
                                                             const int32_t* __restrict__ in_lens,
                                                             const int32_t* __restrict__ tgt_lens, int64_t T, int64_t C,
                                                             int64_t S_max, int blank, float* __restrict__ alpha,
                                                             float* __restrict__ beta, int32_t* __restrict__ next_same,
                                                             float* __restrict__ nll) {
  __shared__ int32_t s_tg[kCtcMaxS];
  extern __shared__ __attribute__((aligned(16))) float s_lp[];
  ctc_alpha_beta_body<NS, EM_LDS, 128>(logp, targets, in_lens, tgt_lens, T, C, S_max, blank, alpha, beta, next_same, nll, blockIdx.x, s_tg, s_lp);
}

// One wave per (b, t) row: grad[b][t][c] = gs * (exp(logp) - occupancy_c), zero for t >= in_len.
// grid: ceil(B*T/4) blocks of 256 threads; dynamic LDS: 4 * (C + S_max) floats.
template <int NS>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logp, const int64_t* __restrict__ targets,
                                                       const int32_t* __restrict__ in_lens, const int32_t* __restrict__ tgt_lens,
                                                       int64_t B, int64_t T, int64_t C, int64_t S_max, int blank,
                                                       const float* __restrict__ alpha, const float* __restrict__ beta,
                                                       const int32_t* __restrict__ next_same, const float* __restrict__ nll,
                                                       const float* __restrict__ gscale, float* __restrict__ grad) {
  constexpr int SP = 64 * NS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  float* s_row = smem + (size_t)wid * (C + S_max);
  float* s_v = s_row + C;
  const bool live = row < B * T;
  const int64_t b = live ? row / T : 0;
  const int64_t t = live ? row - b * T : 0;
  const int Tb = in_lens[b];
  const int S = tgt_lens[b];
  const int SS = 2 * S + 1;
  const float gs = gscale ? gscale[b] : 1.0f / (float)B;
  float* g = grad + row * C;
  const float* lp = logp + row * C;
  if (live && t >= Tb) {
    for (int64_t c = lane; c < C; c += 64) g[c] = 0.f;
  }
  const bool work = live && t < Tb;
  const float nl = nll[b];
  const bool infeasible = isinf(nl);
  if (work) {
    for (int64_t c = lane; c < C; c += 64) s_row[c] = expf(lp[c]);
    // occupancy of every lattice state
    const float* al = alpha + (b * T + t) * SP;
    const float* be = beta + (b * T + t) * SP;
    const int64_t* tg = targets + b * S_max;
    float blank_occ = 0.f;
    // the lane's NS lattice values of either direction as 16-byte loads, its labels and their log-probs requested with them (clamped
    // indices, no branch around a load): as eight scalar loads + a label load + a gather per state inside `if (s < SS)` the row was a
    // chain of dependent round trips (round 5)
    float av[NS], bv[NS], lpc[NS];
    int cls[NS];
#pragma unroll
    for (int i = 0; i < NS; i += 4) {
      const float4 a4 = *reinterpret_cast<const float4*>(al + lane * NS + i);
      const float4 b4 = *reinterpret_cast<const float4*>(be + lane * NS + i);
      av[i] = a4.x; av[i + 1] = a4.y; av[i + 2] = a4.z; av[i + 3] = a4.w;
      bv[i] = b4.x; bv[i + 1] = b4.y; bv[i + 2] = b4.z; bv[i + 3] = b4.w;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = lane * NS + i;
      const int64_t lab = S_max > 0 ? tg[min((int64_t)(s >> 1), S_max - 1)] : 0;   // (S_max = 0: an empty targets tensor)
      cls[i] = (s & 1) ? (int)min(max(lab, (int64_t)0), C - 1) : blank;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) lpc[i] = lp[cls[i]];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = lane * NS + i;
      if (s < SS) {
        const float v = expf(av[i] + bv[i] + nl - lpc[i]);
        if (s & 1) s_v[s >> 1] = v;
        else blank_occ += v;
      }
    }
    blank_occ = wave_sum(blank_occ);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have landed (single-wave hand-off)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) s_row[blank] -= blank_occ;
    const int32_t* nx = next_same + b * S_max * 2;
    for (int i = lane; i < S; i += 64) {
      if (nx[S_max + i]) {  // first occurrence of its label: sum the chain in target order
        float acc = 0.f;
        for (int j = i; j >= 0; j = nx[j]) acc += s_v[j];
        s_row[min(max(tg[i], (int64_t)0), C - 1)] -= acc;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int64_t c = lane; c < C; c += 64) g[c] = infeasible ? __builtin_nanf("") : gs * s_row[c];
  }
}

// ------------------------------------------------------------------ greedy collapse -----------
// one wave per utterance; ballot-compaction of the kept frames, 64 frames per step
__global__ __launch_bounds__(64) void greedy_decode_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                           int64_t T, int blank, int32_t* __restrict__ tokens,
                                                           int32_t* __restrict__ n_tokens) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int32_t* p = ids + (int64_t)b * T;
  int32_t* o = tokens + (int64_t)b * T;
  int64_t L = lens ? lens[b] : T;
  if (L > T) L = T;
  if (L < 0) L = 0;
  int count = 0;
  for (int64_t t0 = 0; t0 < L; t0 += 64) {
    const int64_t t = t0 + lane;
    const bool in = t < L;
    const int cur = in ? p[t] : blank;
    const int prev = (in && t > 0) ? p[t - 1] : blank;
    const bool keep = in && cur != blank && (cur != prev || prev == blank);
    const unsigned long long m = __ballot(keep);
    const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) o[pos] = cur;
    count += __popcll(m);
  }
  for (int64_t t = count + lane; t < T; t += 64) o[t] = -1;
  if (lane == 0) n_tokens[b] = count;
}

static inline int ctc_ns(int64_t S_max) {
  const int64_t ss = 2 * S_max + 1;
  if (ss <= 64 * 4) return 4;
  if (ss <= 64 * 8) return 8;
  if (ss <= 64 * 16) return 16;
  return 0;
}

}  // namespace lasr

using namespace lasr;

extern "C" int lasr_log_softmax(const float* logits, float* logp, int32_t* argmax, int64_t N, int64_t C, void* stream) {
  LASR_CHECK_ARG(logits && logp && N > 0 && C > 0, "lasr_log_softmax: bad argument");
  hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, as_stream(stream), logits, logp, argmax, N, C);
  LASR_LAUNCH_CHECK("log_softmax_kernel");
  return 0;
}

bool lasr::log_softmax_split_on(int64_t C) {
  static const bool off = getenv("LASR_LOGSOFTMAX_SPLIT") && atoi(getenv("LASR_LOGSOFTMAX_SPLIT")) == 0;   // dev switch
  return !off && C >= 1 && C <= 64;
}

int lasr::log_softmax_split(const float* partials, int split, const float* bias, float* logits, float* logp, int32_t* argmax, int64_t N,
                            int64_t C, void* stream) {
  LASR_CHECK_ARG(partials && logits && logp && N > 0 && split >= 1 && C >= 1 && C <= 64, "log_softmax_split: bad argument");
  hipLaunchKernelGGL(log_softmax_split_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, as_stream(stream), partials, split, bias, logits, logp,
                     argmax, N, C);
  LASR_LAUNCH_CHECK("log_softmax_split_kernel");
  return 0;
}

extern "C" int lasr_log_softmax_bwd(const float* logp, const float* grad_logp, float* grad_logits, int64_t N, int64_t C, void* stream) {
  LASR_CHECK_ARG(logp && grad_logp && grad_logits && N > 0 && C > 0, "lasr_log_softmax_bwd: bad argument");
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, as_stream(stream), logp, grad_logp, grad_logits, N, C);
  LASR_LAUNCH_CHECK("log_softmax_bwd_kernel");
  return 0;
}

extern "C" size_t lasr_ctc_workspace_bytes(int64_t B, int64_t T, int64_t S_max) {
  const int ns = ctc_ns(S_max);
  if (!ns) return 0;
  return align_up((size_t)2 * B * T * 64 * ns * sizeof(float), 256) + align_up((size_t)B * (S_max > 0 ? S_max : 1) * 2 * sizeof(int32_t), 256);
}

// gradient pass over a lattice already in `workspace` (layout of lasr_ctc_loss)
namespace lasr {
int launch_ctc_grad(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T,
                    int64_t C, int64_t S_max, int blank, const float* nll, float* grad, const float* gscale, void* workspace,
                    void* stream) {
  const int ns = ctc_ns(S_max);
  const size_t ab = (size_t)B * T * 64 * ns;
  float* alpha = reinterpret_cast<float*>(workspace);
  float* beta = alpha + ab;
  int32_t* next_same = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + align_up(2 * ab * sizeof(float), 256));
  hipStream_t st = as_stream(stream);
  const int64_t sm = S_max > 0 ? S_max : 1;
  const size_t shmem = 4 * (size_t)(C + sm) * sizeof(float);
  LASR_CHECK_SHAPE(shmem <= 160 * 1024, "lasr_ctc_loss: C=%lld too large for the LDS row buffer", (long long)C);
  dim3 grid((unsigned)cdiv(B * T, 4));
#define LASR_CTC_G(NS_)                                                                                                    \
  do {                                                                                                                     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_grad_kernel<NS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL(ctc_grad_kernel<NS_>, grid, dim3(256), shmem, st, logp, targets, in_lens, tgt_lens, B, T, C, sm, blank, alpha, \
                       beta, next_same, nll, gscale, grad);                                                                \
  } while (0)
  if (ns == 4) LASR_CTC_G(4); else if (ns == 8) LASR_CTC_G(8); else LASR_CTC_G(16);
#undef LASR_CTC_G
  LASR_LAUNCH_CHECK("ctc_grad_kernel");
  return 0;
}
}  // namespace lasr

extern "C" int lasr_ctc_loss(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B,
                             int64_t T, int64_t C, int64_t S_max, int blank, float* nll, float* grad, const float* gscale,
                             void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(logp && targets && in_lens && tgt_lens && nll && workspace, "lasr_ctc_loss: null pointer");
  LASR_CHECK_SHAPE(B > 0 && T > 0 && C > 1 && S_max >= 0 && blank >= 0 && blank < C, "lasr_ctc_loss: shape");
  const int ns = ctc_ns(S_max);
  LASR_CHECK_SHAPE(ns != 0, "lasr_ctc_loss: S_max=%lld exceeds the 511-label lattice the kernels are built for", (long long)S_max);
  if (workspace_bytes < lasr_ctc_workspace_bytes(B, T, S_max)) return fail(LASR_E_WORKSPACE, "lasr_ctc_loss: workspace");
  const size_t ab = (size_t)B * T * 64 * ns;
  float* alpha = reinterpret_cast<float*>(workspace);
  float* beta = alpha + ab;
  int32_t* next_same = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + align_up(2 * ab * sizeof(float), 256));
  hipStream_t st = as_stream(stream);
  const int64_t sm = S_max > 0 ? S_max : 1;
  // emissions in LDS when one utterance's T x C f32 block fits beside the kernel's other needs (cfg2: 56 KB)
  const size_t em_bytes = (size_t)(T + 2) * C * sizeof(float);   // one pad row on either side
  const bool em_lds = em_bytes <= 144 * 1024 && C % 4 == 0 && reinterpret_cast<uintptr_t>(logp) % 16 == 0 &&
                      !getenv("LASR_CTC_NO_LDS");
#define LASR_CTC_AB(NS_)                                                                                                   \
  do {                                                                                                                     \
    if (em_lds) {                                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_alpha_beta_kernel<NS_, true>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); /* + 2 KB static */     \
      hipLaunchKernelGGL((ctc_alpha_beta_kernel<NS_, true>), dim3((unsigned)B), dim3(128), em_bytes, st, logp, targets, in_lens, \
                         tgt_lens, T, C, sm, blank, alpha, beta, next_same, nll);                                          \
    } else {                                                                                                               \
      hipLaunchKernelGGL((ctc_alpha_beta_kernel<NS_, false>), dim3((unsigned)B), dim3(128), 0, st, logp, targets, in_lens, \
                         tgt_lens, T, C, sm, blank, alpha, beta, next_same, nll);                                          \
    }                                                                                                                      \
  } while (0)
  if (ns == 4) LASR_CTC_AB(4); else if (ns == 8) LASR_CTC_AB(8); else LASR_CTC_AB(16);
#undef LASR_CTC_AB
  LASR_LAUNCH_CHECK("ctc_alpha_beta_kernel");
  if (grad) return launch_ctc_grad(logp, targets, in_lens, tgt_lens, B, T, C, S_max, blank, nll, grad, gscale, workspace, stream);
  return 0;
}

extern "C" int lasr_greedy_decode(const int32_t* ids, const int32_t* lens, int64_t B, int64_t T, int blank, int32_t* tokens,
                                  int32_t* n_tokens, void* stream) {
  LASR_CHECK_ARG(ids && tokens && n_tokens && B > 0 && T > 0, "lasr_greedy_decode: bad argument");
  hipLaunchKernelGGL(greedy_decode_kernel, dim3((unsigned)B), dim3(64), 0, as_stream(stream), ids, lens, T, blank, tokens, n_tokens);
  LASR_LAUNCH_CHECK("greedy_decode_kernel");
  return 0;
}
