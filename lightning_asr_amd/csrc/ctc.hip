// log_softmax(+argmax), wave-synchronous CTC forward/backward and greedy CTC collapse.
// Replaces F.log_softmax (models/QuartNet.py:290), nn.CTCLoss(blank=C-1, reduction='none',
// zero_infinity=False) + its backward (train.py:77-78,196) and the Python greedy decoder
// (utils/asr_metrics.py:155-166).
//
// CTC lattice: one workgroup of two waves per utterance; wave 0 runs the alpha recursion forward
// in time while wave 1 runs beta backward.  Each lane owns NS consecutive lattice states, the
// s-1 / s-2 (s+1 / s+2) neighbours cross lanes with one shuffle each, so a time step needs no
// LDS and no barrier; the emission log-probs of step t+1 are fetched while step t is computed.
#include "common.h"
#include <math.h>

namespace lasr {

static constexpr float kNegInf = -INFINITY;

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

// ------------------------------------------------------------------ CTC ------------------------
// log-sum-exp on the hardware exp2/log2 units (__expf/__logf -> v_exp_f32/v_log_f32): the recursion
// is a ~500-step dependent chain per utterance, so transcendental LATENCY is the kernel's run time
// (libm expf/logf: ~0.5 ms per step of the bench; these: ~10x less).  Arguments are in [-90, 0] and
// [1, 3]; the relative error per step (~1e-6) stays far inside the 1e-4 loss tolerance.
// Branch-free: with every input -inf the shifted sum is exp(-inf)*3 = 0 and log(0) = -inf, so no
// per-lane early exit is needed (divergent exits cost an exec-mask branch per state per step).
// Neighbour exchange of the lattice recursion on the DPP path (gfx9 wave-wide shifts, one VALU op) instead of
// ds_bpermute (an LDS round trip on the critical path of every one of the T' dependent steps):
// wave_shr1: lane i receives lane i-1, lane 0 keeps `fill`; wave_shl1: lane i receives lane i+1, lane 63 `fill`.
__device__ __forceinline__ float wave_shr1(float v, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  const float mm = (m == kNegInf) ? 0.f : m;
  return mm + __logf(__expf(a - mm) + __expf(b - mm) + __expf(c - mm));
}
__device__ __forceinline__ float lse2(float a, float b) { return lse3(a, b, kNegInf); }

// Workspace layout per utterance: alpha [T][SP], beta [T][SP] (SP = 64*NS), then next_same [S_max] int32.
// grid: B blocks of 128 threads.
// lse3 for the alpha/beta recursions on the raw transcendental units: v_exp_f32 / v_log_f32 ARE 2^x / log2(x), so
// the differences are scaled by log2(e) and the logarithm by ln(2) (the sum is in [1,3], or exactly 0 when all
// three are -inf: no denormal fix-ups, no range checks) - 15 instructions against ~30 for __expf/__logf with
// their scalings and guards, with the same values in natural-log units (the rounding that matters, of
// m + log(sum) at |alpha| ~ 1e3, is unchanged; a pure base-2 lattice was 1.4x faster still but its
// unit conversions at that magnitude cost 30 % more gradient error against an f64 reference).
static constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  const float mm = (m == kNegInf) ? 0.f : m;
  const float s = __builtin_amdgcn_exp2f((a - mm) * kLog2e) + __builtin_amdgcn_exp2f((b - mm) * kLog2e) +
                  __builtin_amdgcn_exp2f((c - mm) * kLog2e);
  return fmaf(kLn2, __builtin_amdgcn_logf(s), mm);
}

__device__ __forceinline__ float lse2_fast(float a, float b) {
  const float m = fmaxf(a, b);
  const float mm = (m == kNegInf) ? 0.f : m;
  const float s = __builtin_amdgcn_exp2f((a - mm) * kLog2e) + __builtin_amdgcn_exp2f((b - mm) * kLog2e);
  return fmaf(kLn2, __builtin_amdgcn_logf(s), mm);
}

// EM_LDS: the utterance's whole emission matrix logp[b] (T x C f32; 56 KB at T'=501, C=28) is copied into LDS once
// with coalesced 16-byte loads and both waves gather their per-state emissions from there one step ahead, so the
// T' dependent steps contain no global load and never wait on vmcnt (which also counts the lattice stores).
// Large vocabularies (C=4334) keep the register ring of global prefetches.
template <int NS, bool EM_LDS>
__global__ __launch_bounds__(128) void ctc_alpha_beta_kernel(const float* __restrict__ logp, const int64_t* __restrict__ targets,
                                                             const int32_t* __restrict__ in_lens,
                                                             const int32_t* __restrict__ tgt_lens, int64_t T, int64_t C,
                                                             int64_t S_max, int blank, float* __restrict__ alpha,
                                                             float* __restrict__ beta, int32_t* __restrict__ next_same,
                                                             float* __restrict__ nll) {
  constexpr int SP = 64 * NS;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const bool is_beta = (threadIdx.x >> 6) != 0;  // wave-uniform
  const int Tb = in_lens[b];
  const int S = tgt_lens[b];
  const int SS = 2 * S + 1;
  const int64_t* tg = targets + (int64_t)b * S_max;
  const float* lp = logp + (int64_t)b * T * C;
  float* out = (is_beta ? beta : alpha) + (int64_t)b * T * SP;

  // chain of equal labels (for the deterministic per-class sum in the gradient kernel)
  for (int i = threadIdx.x; i < S; i += 128) {
    const int64_t me = tg[i];
    int nx = -1;
    for (int j = i + 1; j < S; ++j)
      if (tg[j] == me) { nx = j; break; }
    int first = 1;
    for (int j = 0; j < i; ++j)
      if (tg[j] == me) { first = 0; break; }
    next_same[(int64_t)b * S_max * 2 + i] = nx;
    next_same[(int64_t)b * S_max * 2 + S_max + i] = first;
  }
  if (Tb <= 0) {
    if (threadIdx.x == 0) nll[b] = (S == 0) ? 0.f : INFINITY;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float s_lp[];
  if (EM_LDS) {
    const int64_t n = (int64_t)Tb * C;            // the host checked (T*C) % 4 == 0 and 16-byte alignment of logp
    const int64_t n4 = n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += 128) {
      const float4 v = reinterpret_cast<const float4*>(lp)[i];
      reinterpret_cast<float4*>(s_lp)[i] = v;
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 128) s_lp[i] = lp[i];
    __syncthreads();
  }
  // per-lane state description
  int cls[NS];
  bool skip_ok[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = lane * NS + i;
    int c = blank;
    bool sk = false;
    if (s < SS && (s & 1)) {
      c = (int)tg[s >> 1];
      if (!is_beta) sk = s >= 3 ? (tg[(s >> 1) - 1] != tg[s >> 1]) : false;          // from s-2 into s
      else sk = (s + 2 < SS) ? (tg[(s >> 1) + 1] != tg[s >> 1]) : false;              // from s into s+2
    }
    cls[i] = c;
    skip_ok[i] = sk;
  }
  float a[NS], em[NS];
  const int t_first = is_beta ? Tb - 1 : 0;
  const int dt = is_beta ? -1 : 1;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = lane * NS + i;
    em[i] = (s < SS) ? lp[(int64_t)t_first * C + cls[i]] : kNegInf;
    bool start = is_beta ? (s == SS - 1 || s == SS - 2) : (s == 0 || s == 1);
    a[i] = (start && s >= 0 && s < SS) ? em[i] : kNegInf;
    out[(int64_t)t_first * SP + s] = a[i];
  }
  // one recursion step: a[] (t - dt) -> a[] (t) with emissions em[], lattice row stored
  auto advance = [&](int t) {
    float n[NS];
    if (!is_beta) {
      const float p1 = wave_shr1(a[NS - 1], kNegInf);
      const float p2 = NS >= 2 ? wave_shr1(a[NS - 2], kNegInf) : kNegInf;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        // i==0: s-1 is the previous lane's last state, s-2 its second to last; i==1: s-2 is the previous lane's last
        const float s1 = i >= 1 ? a[i - 1] : p1;
        const float s2v = (i == 0) ? p2 : (i == 1 ? p1 : a[i - 2]);
        // even states are blanks (NS is even, so the parity of s is the parity of i): no skip transition, two terms
        n[i] = ((i & 1) ? lse3_fast(a[i], s1, skip_ok[i] ? s2v : kNegInf) : lse2_fast(a[i], s1)) + em[i];
      }
    } else {
      const float q1 = wave_shl1(a[0], kNegInf);
      const float q2 = NS >= 2 ? wave_shl1(a[1], kNegInf) : kNegInf;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const float s1 = i + 1 < NS ? a[i + 1] : q1;
        const float s2v = (i + 2 < NS) ? a[i + 2] : (i + 2 == NS ? q1 : q2);
        n[i] = ((i & 1) ? lse3_fast(a[i], s1, skip_ok[i] ? s2v : kNegInf) : lse2_fast(a[i], s1)) + em[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = lane * NS + i;
      a[i] = (s < SS) ? n[i] : kNegInf;
      out[(int64_t)t * SP + s] = a[i];
    }
  };
  if (EM_LDS) {
    float nx[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) nx[i] = Tb > 1 ? s_lp[(t_first + dt) * (int)C + cls[i]] : 0.f;
    for (int step = 1; step < Tb; ++step) {
      const int t = t_first + dt * step;
#pragma unroll
      for (int i = 0; i < NS; ++i) em[i] = nx[i];
      const int tn = step + 1 < Tb ? t + dt : t;      // next step's emissions: issued before this step's arithmetic
#pragma unroll
      for (int i = 0; i < NS; ++i) nx[i] = s_lp[tn * (int)C + cls[i]];
      advance(t);
    }
  } else {
  // Emissions are fetched kPre steps ahead into a register ring.  On CDNA4 s_waitcnt vmcnt counts
  // stores as well as loads, in issue order: with a one-step prefetch every step would also wait for
  // the previous step's lattice stores to retire (~0.7 us).  Eight steps of slack hide both.
  constexpr int kPre = 8;
  float ring[kPre][NS];
#pragma unroll
  for (int u = 0; u < kPre; ++u)
#pragma unroll
    for (int i = 0; i < NS; ++i)
      ring[u][i] = (1 + u < Tb && lane * NS + i < SS) ? lp[(int64_t)(t_first + dt * (1 + u)) * C + cls[i]] : kNegInf;
  for (int step0 = 1; step0 < Tb; step0 += kPre) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int step = step0 + u;
      if (step < Tb) {  // wave-uniform
        const int t = t_first + dt * step;
#pragma unroll
        for (int i = 0; i < NS; ++i) em[i] = ring[u][i];
        if (step + kPre < Tb) {
#pragma unroll
          for (int i = 0; i < NS; ++i)
            ring[u][i] = (lane * NS + i < SS) ? lp[(int64_t)(t + dt * kPre) * C + cls[i]] : kNegInf;
        }
        advance(t);
      }
    }
  }
  }
  if (!is_beta) {
    // ll = lse(alpha_{T-1}(SS-1), alpha_{T-1}(SS-2))
    float v = kNegInf;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = lane * NS + i;
      if (s == SS - 1 || s == SS - 2) v = lse2(v, a[i]);
    }
    // two candidate lanes at most: combine across the wave
    float m = wave_max(v);
    float e = (v == kNegInf) ? 0.f : expf(v - m);
    e = wave_sum(e);
    if (lane == 0) nll[b] = (m == kNegInf) ? INFINITY : -(m + logf(e));
  }
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
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = lane * NS + i;
      if (s < SS) {
        const int c = (s & 1) ? (int)tg[s >> 1] : blank;
        const float v = expf(al[s] + be[s] + nl - lp[c]);
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
        s_row[tg[i]] -= acc;
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
  const size_t em_bytes = (size_t)T * C * sizeof(float);
  const bool em_lds = em_bytes <= 144 * 1024 && (T * C) % 4 == 0 && reinterpret_cast<uintptr_t>(logp) % 16 == 0 &&
                      !getenv("LASR_CTC_NO_LDS");
#define LASR_CTC_AB(NS_)                                                                                                   \
  do {                                                                                                                     \
    if (em_lds) {                                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_alpha_beta_kernel<NS_, true>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                   \
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
  if (grad) {
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
  }
  return 0;
}

extern "C" int lasr_greedy_decode(const int32_t* ids, const int32_t* lens, int64_t B, int64_t T, int blank, int32_t* tokens,
                                  int32_t* n_tokens, void* stream) {
  LASR_CHECK_ARG(ids && tokens && n_tokens && B > 0 && T > 0, "lasr_greedy_decode: bad argument");
  hipLaunchKernelGGL(greedy_decode_kernel, dim3((unsigned)B), dim3(64), 0, as_stream(stream), ids, lens, T, blank, tokens, n_tokens);
  LASR_LAUNCH_CHECK("greedy_decode_kernel");
  return 0;
}
