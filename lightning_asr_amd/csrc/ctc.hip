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
// the differences are scaled by log2(e) and the logarithm by ln(2): the same values in natural-log units (the
// rounding that matters, of m + log(sum) at |alpha| ~ 1e3, is unchanged; a pure base-2 lattice was 1.4x faster
// still but its unit conversions at that magnitude cost 30 % more gradient error against an f64 reference).
//
// The recursion is ISSUE-bound (one wave per SIMD, ~60 VALU instructions per time step), so the per-step
// instruction count is the kernel's run time.  What keeps it short:
//   * unreachable states hold the finite sentinel kDead = -1e30 instead of -inf: it absorbs every update
//     (-1e30 + log(3) + emission rounds back to -1e30), exp2 of differences against it is exactly 0, and no
//     "all three are -inf" special case (compare + two selects per state) is left in the chain;
//   * the largest term of the sum is exp(0) = 1: only the smaller ones go through the quarter-rate v_exp_f32;
//   * max3 / med3 / min3 as single instructions without the IEEE-mode canonicalisation of their inputs;
//   * alpha and beta are separate instantiations selected by a scalar branch (a per-lane `is_beta` compiled to
//     exec-mask divergence: both bodies' register shuffles ran every step);
//   * no `s < SS` masking: alpha's states >= SS never feed a lower state, beta's start dead and stay dead.
static constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
static constexpr float kDead = -1e30f;
__device__ __forceinline__ float v_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float v_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float v_med3(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float v_max2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_min2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
  const float m = v_max3(a, b, c);
  const float md = v_med3(a, b, c);
  const float lo = v_min3(a, b, c);
  const float s = 1.f + __builtin_amdgcn_exp2f((md - m) * kLog2e) + __builtin_amdgcn_exp2f((lo - m) * kLog2e);
  return fmaf(kLn2, __builtin_amdgcn_logf(s), m);
}

__device__ __forceinline__ float lse2_fast(float a, float b) {
  const float m = v_max2(a, b);
  const float lo = v_min2(a, b);
  const float s = 1.f + __builtin_amdgcn_exp2f((lo - m) * kLog2e);
  return fmaf(kLn2, __builtin_amdgcn_logf(s), m);
}

// One direction of the lattice for one utterance, run by one wave.  s_tg: the utterance's targets in LDS;
// s_rows (EM_LDS): the emission matrix in LDS with one pad row on either side (row t at s_rows + t*C), so the
// one-step-ahead gather needs no end-of-sequence clamp.  Stored rows hold kDead for unreachable states and
// unspecified values for s >= 2S+1 (the gradient kernel reads s < 2S+1 only).
template <int NS, bool EM_LDS, bool BETA>
__device__ __forceinline__ void ctc_lattice(const float* __restrict__ lp, const float* s_rows, const int32_t* s_tg, int lane, int Tb,
                                            int S, int C, int blank, float* __restrict__ out, float* __restrict__ nll_b) {
  constexpr int SP = 64 * NS;
  const int SS = 2 * S + 1;
  int cls4[NS];       // byte offset of the state's class inside an emission row
  bool skip_ok[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = lane * NS + i;
    int c = blank;
    bool sk = false;
    if (s < SS && (s & 1)) {
      c = s_tg[s >> 1];
      if (!BETA) sk = s >= 3 ? (s_tg[(s >> 1) - 1] != c) : false;          // from s-2 into s
      else sk = (s + 2 < SS) ? (s_tg[(s >> 1) + 1] != c) : false;          // from s into s+2
    }
    cls4[i] = c * 4;
    skip_ok[i] = sk;
  }
  float a[NS], em[NS];
  const int t_first = BETA ? Tb - 1 : 0;
  constexpr int dt = BETA ? -1 : 1;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int s = lane * NS + i;
    const bool start = BETA ? (s == SS - 1 || s == SS - 2) : (s == 0 || s == 1);
    a[i] = (start && s < SS) ? lp[(int64_t)t_first * C + (cls4[i] >> 2)] : kDead;
  }
  float* o = out + (int64_t)t_first * SP + lane * NS;
#pragma unroll
  for (int i = 0; i < NS; ++i) o[i] = a[i];
  // one recursion step: a[] (t - dt) -> a[] (t) with emissions em[], lattice row stored
  auto advance = [&]() {
    float n[NS];
    if (!BETA) {
      const float p1 = wave_shr1(a[NS - 1], kDead);
      const float p2 = wave_shr1(a[NS - 2], kDead);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        // i==0: s-1 is the previous lane's last state, s-2 its second to last; i==1: s-2 is the previous lane's last
        const float s1 = i >= 1 ? a[i - 1] : p1;
        const float s2v = (i == 0) ? p2 : (i == 1 ? p1 : a[i - 2]);
        // even states are blanks (NS is even, so the parity of s is the parity of i): no skip transition, two terms
        n[i] = ((i & 1) ? lse3_fast(a[i], s1, skip_ok[i] ? s2v : kDead) : lse2_fast(a[i], s1)) + em[i];
      }
    } else {
      const float q1 = wave_shl1(a[0], kDead);
      const float q2 = wave_shl1(a[1], kDead);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const float s1 = i + 1 < NS ? a[i + 1] : q1;
        const float s2v = (i + 2 < NS) ? a[i + 2] : (i + 2 == NS ? q1 : q2);
        n[i] = ((i & 1) ? lse3_fast(a[i], s1, skip_ok[i] ? s2v : kDead) : lse2_fast(a[i], s1)) + em[i];
      }
    }
    o += dt * SP;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      a[i] = n[i];
      o[i] = n[i];
    }
  };
  if (EM_LDS) {
    const char* row = reinterpret_cast<const char*>(s_rows) + (int64_t)(t_first + dt) * C * 4;
    const int drow = dt * C * 4;
    float nx[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) nx[i] = *reinterpret_cast<const float*>(row + cls4[i]);
    for (int step = 1; step < Tb; ++step) {
#pragma unroll
      for (int i = 0; i < NS; ++i) em[i] = nx[i];
      row += drow;                                   // next step's emissions: issued before this step's arithmetic
#pragma unroll                                       // (the last one reads the pad row)
      for (int i = 0; i < NS; ++i) nx[i] = *reinterpret_cast<const float*>(row + cls4[i]);
      advance();
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
        ring[u][i] = (1 + u < Tb) ? lp[(int64_t)(t_first + dt * (1 + u)) * C + (cls4[i] >> 2)] : 0.f;
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
            for (int i = 0; i < NS; ++i) ring[u][i] = lp[(int64_t)(t + dt * kPre) * C + (cls4[i] >> 2)];
          }
          advance();
        }
      }
    }
  }
  if (!BETA) {
    // ll = lse(alpha_{T-1}(SS-1), alpha_{T-1}(SS-2)); two candidate states, in at most two lanes
    float v = kNegInf;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int s = lane * NS + i;
      if ((s == SS - 1 || s == SS - 2) && a[i] > 0.5f * kDead) v = lse2(v, a[i]);
    }
    const float m = wave_max(v);
    float e = (v == kNegInf) ? 0.f : expf(v - m);
    e = wave_sum(e);
    if (lane == 0) *nll_b = (m == kNegInf) ? INFINITY : -(m + logf(e));
  }
}

// Workspace layout per utterance: alpha [T][SP], beta [T][SP] (SP = 64*NS), then next_same [2][S_max] int32.
// grid: B blocks of 128 threads (wave 0: alpha, wave 1: beta).
// EM_LDS: the utterance's whole emission matrix logp[b] (T x C f32; 56 KB at T'=501, C=28) is copied into LDS once
// with coalesced 16-byte loads and both waves gather their per-state emissions from there one step ahead, so the
// T' dependent steps contain no global load and never wait on vmcnt (which also counts the lattice stores).
// Large vocabularies (C=4334) keep the register ring of global prefetches.
static constexpr int kCtcMaxS = 512;
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
  const int Tb = in_lens[b];
  const int S = tgt_lens[b];
  const int64_t* tg = targets + (int64_t)b * S_max;
  const float* lp = logp + (int64_t)b * T * C;
  __shared__ int32_t s_tg[kCtcMaxS];
  extern __shared__ __attribute__((aligned(16))) float s_lp[];
  for (int i = threadIdx.x; i < S; i += 128) s_tg[i] = (int32_t)tg[i];
  if (EM_LDS && Tb > 0) {
    // emission rows 0..Tb-1 behind one pad row; 4 x 16-byte loads in flight per thread
    const int n4 = (int)(((int64_t)Tb * C) >> 2);   // the host checked C % 4 == 0 and the 16-byte alignment of logp
    const float4* src = reinterpret_cast<const float4*>(lp);
    float4* dst = reinterpret_cast<float4*>(s_lp + C);
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * 128) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + u * 128, n4 - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * 128 < n4) dst[i0 + u * 128] = v[u];
    }
    for (int i = threadIdx.x; i < (int)C; i += 128) {   // pad rows (read one step past either end, never used)
      s_lp[i] = 0.f;
      s_lp[(int64_t)(Tb + 1) * C + i] = 0.f;
    }
  }
  __syncthreads();
  // chain of equal labels (for the deterministic per-class sum in the gradient kernel)
  for (int i = threadIdx.x; i < S; i += 128) {
    const int me = s_tg[i];
    int nx = -1;
    for (int j = S - 1; j > i; --j) nx = (s_tg[j] == me) ? j : nx;
    int first = 1;
    for (int j = 0; j < i; ++j) first = (s_tg[j] == me) ? 0 : first;
    next_same[(int64_t)b * S_max * 2 + i] = nx;
    next_same[(int64_t)b * S_max * 2 + S_max + i] = first;
  }
  if (Tb <= 0) {
    if (threadIdx.x == 0) nll[b] = (S == 0) ? 0.f : INFINITY;
    return;
  }
  const bool is_beta = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;   // scalar: a real branch
  if (is_beta)
    ctc_lattice<NS, EM_LDS, true>(lp, s_lp + C, s_tg, lane, Tb, S, (int)C, blank, beta + (int64_t)b * T * SP, nullptr);
  else
    ctc_lattice<NS, EM_LDS, false>(lp, s_lp + C, s_tg, lane, Tb, S, (int)C, blank, alpha + (int64_t)b * T * SP, nll + b);
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
