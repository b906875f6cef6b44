// CTC lattice device code (alpha / beta recursions) shared by ctc.hip (stand-alone kernel) and mel.hip (the lattice
// workgroups inside the feature-prefetch grid).  Replaces nn.CTCLoss forward (train.py:77-78,196).
#pragma once
#include "common.h"
#include <math.h>

namespace lasr {

static constexpr float kNegInf = -INFINITY;

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
// COMPACT: `lp` / `s_rows` is the gathered emission matrix of ctc_lean.hip - row t holds the emission of label POSITION i in
// column i and the blank's in column `blank` (= S_max): the class of an odd state is its position, the skip rule still
// compares the labels themselves.
template <int NS, bool EM_LDS, bool BETA, bool COMPACT = false>
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
      const int lab = s_tg[s >> 1];
      c = COMPACT ? (s >> 1) : lab;
      if (!BETA) sk = s >= 3 ? (s_tg[(s >> 1) - 1] != lab) : false;        // from s-2 into s
      else sk = (s + 2 < SS) ? (s_tg[(s >> 1) + 1] != lab) : false;        // from s into s+2
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
          {   // unconditional refill from a clamped row: a load inside a branch is drained (s_waitcnt vmcnt(0)) at the branch's
              // join, which put the memory round trip back on every step's critical path
            const int tq = BETA ? max(t - kPre, 0) : min(t + kPre, Tb - 1);
#pragma unroll
            for (int i = 0; i < NS; ++i) ring[u][i] = lp[(int64_t)tq * C + (cls4[i] >> 2)];
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
// b: utterance; NT: threads of the workgroup that take part (128 in the stand-alone kernel, 256 inside the fused
// mel + CTC grid: the upper waves help with the LDS fill, then leave); s_tg: kCtcMaxS ints, s_lp: (T + 2) * C floats.
template <int NS, bool EM_LDS, int NT, bool COMPACT = false>
__device__ __forceinline__ void ctc_alpha_beta_body(const float* __restrict__ logp, const int64_t* __restrict__ targets,
                                                    const int32_t* __restrict__ in_lens, const int32_t* __restrict__ tgt_lens,
                                                    int64_t T, int64_t C, int64_t S_max, int blank, float* __restrict__ alpha,
                                                    float* __restrict__ beta, int32_t* __restrict__ next_same,
                                                    float* __restrict__ nll, int b, int32_t* s_tg, float* s_lp) {
  constexpr int SP = 64 * NS;
  const int lane = threadIdx.x & 63;
  const int Tb = in_lens[b];
  const int S = tgt_lens[b];
  const int64_t* tg = targets + (int64_t)b * S_max;
  const float* lp = logp + (int64_t)b * T * C;
  // (labels clamped into the emission row: invalid user data must not become an out-of-bounds device access)
  for (int i = threadIdx.x; i < S; i += NT) s_tg[i] = (int32_t)min(max(tg[i], (int64_t)0), (int64_t)(COMPACT ? 0x7fffffff : C - 1));
  if (EM_LDS && Tb > 0) {
    // emission rows 0..Tb-1 behind one pad row; 4 x 16-byte loads in flight per thread
    const int n4 = (int)(((int64_t)Tb * C) >> 2);   // the host checked C % 4 == 0 and the 16-byte alignment of logp
    const float4* src = reinterpret_cast<const float4*>(lp);
    float4* dst = reinterpret_cast<float4*>(s_lp + C);
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * NT) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[min(i0 + u * NT, n4 - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + u * NT < n4) dst[i0 + u * NT] = v[u];
    }
    for (int i = threadIdx.x; i < (int)C; i += NT) {   // pad rows (read one step past either end, never used)
      s_lp[i] = 0.f;
      s_lp[(int64_t)(Tb + 1) * C + i] = 0.f;
    }
  }
  __syncthreads();
  // chain of equal labels (for the deterministic per-class sum in the gradient kernel).  By the helper waves where the workgroup has
  // any (the fused feature + lattice grid): as two O(S) loops of dependent-latency LDS reads per label on the lattice waves they
  // stood 6.5 us (S = 100) in front of the recursion (phase stamps, round 5).  One pass over the labels finds both answers: the
  // smallest j > i with the same label, and whether any j < i has it.
  constexpr int kChainT0 = NT > 128 ? 128 : 0, kChainN = NT > 128 ? NT - 128 : NT;
  for (int i = (int)threadIdx.x - kChainT0; i >= 0 && i < S; i += kChainN) {
    const int me = s_tg[i];
    int nx = -1, first = 1;
#pragma unroll 8
    for (int j = S - 1; j >= 0; --j) {
      const bool same = s_tg[j] == me;
      nx = (same && j > i) ? j : nx;
      first = (same && j < i) ? 0 : first;
    }
    next_same[(int64_t)b * S_max * 2 + i] = nx;
    next_same[(int64_t)b * S_max * 2 + S_max + i] = first;
  }
  if (Tb <= 0) {
    if (threadIdx.x == 0) nll[b] = (S == 0) ? 0.f : INFINITY;
    return;
  }
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: real branches
  if (wv >= 2) return;                                                        // helper waves of a wider workgroup
  const bool is_beta = wv != 0;
  if (is_beta)
    ctc_lattice<NS, EM_LDS, true, COMPACT>(lp, s_lp + C, s_tg, lane, Tb, S, (int)C, blank, beta + (int64_t)b * T * SP, nullptr);
  else
    ctc_lattice<NS, EM_LDS, false, COMPACT>(lp, s_lp + C, s_tg, lane, Tb, S, (int)C, blank, alpha + (int64_t)b * T * SP, nll + b);
}


// ctc.hip
int launch_ctc_grad(const float* logp, const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T,
                    int64_t C, int64_t S_max, int blank, const float* nll, float* grad, const float* gscale, void* workspace,
                    void* stream);

}  // namespace lasr
