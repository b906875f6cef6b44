// The BiLSTM backward recurrence (models/QuartNetContext.py:186-199, autograd of nn.LSTM) as a device function, shared by its
// stand-alone kernel (lstm.hip) and by the grid that runs it BESIDE the stage-batched 1x1 weight gradients (gemm_bf16.hip, round 4:
// the recurrence is 437 us of dependent steps on 64 small workgroups; the weight-gradient launch of the units above block3 does not
// depend on it and fills the other CUs meanwhile).  One recurrence = 160 threads: thread (u, q) = (tid >> 2, tid & 3) owns gate row
// q*H + u.  See lstm.hip for what sits on the per-step dependency chain and why.
#pragma once
#include "common.h"

namespace lasr {
namespace lstm {

static constexpr int H = 40, G = 4 * H;  // hidden size, gate rows
static constexpr int kPre = 8;            // per-step operands are fetched this many steps ahead (register ring)

__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {   // lane Q of every quad -> the whole quad
  constexpr int ctrl = Q | (Q << 2) | (Q << 4) | (Q << 6);
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}

struct BwdArgs {
  const void* dout; int64_t ldd, col0;
  const float* whh_f; const float* whh_r; const int32_t* lens; int64_t Tt; const float* saved;
  float* dg_f; float* dg_r; float* pwhh;
  // optional (null: not written): bf16 copies of dg for the GEMMs behind the recurrence (dW_ih, dx), and per-workgroup column sums of
  // dg (the bias gradients) next to the dW_hh partials
  bf16_t* dgb_f; bf16_t* dgb_r; float* pbias;
};
struct BwdSmem {
  __attribute__((aligned(16))) float s_dg[2][G];          // the step's gate gradients, double-buffered: ONE barrier per step
};
static constexpr int kDwZ = 4;                             // time chunks of the dW_hh launch (partials per utterance)

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// dout: columns [col0 + dir*H, +H) of a [B][T][ldd] tensor in T (gradient w.r.t. the LSTM output).
// dg [B][T][G] f32 per direction = gradient w.r.t. the gate pre-activations (zero rows for t >= len).
// Round 4: what sits on the per-step chain is only what the NEXT step needs - dh_prev = W_hh^T dg.  Thread (u, q) sums gate q's 40
// rows of column u (its 40 weights in registers, the gate gradients from LDS as 10 broadcast 16-byte reads) and the unit's quad
// adds the four partial sums with two DPP steps: no second LDS exchange, so ONE barrier per step with the gate gradients double-
// buffered (the forward kernel's shape).  dW_hh = sum_t dg_t (x) h_{t-1} does not feed the chain: it left the recurrence for its own
// launch over the stored dg (dwhh_partials below; 40 of the step's 80 FMAs and its second barrier went with it).
// tid: thread index inside the recurrence's slot; nthr: threads of the slot that run this body (>= G).  PADDED: the slot has lanes
// past G (a 192-thread slot of a larger workgroup): they compute on clamped indices and store nothing.  BARRIER(): a barrier over
// exactly the waves that run recurrences in this workgroup; they must all walk the same number of steps (the two directions of
// one utterance do).
template <typename T, bool PADDED, int KP, typename Barrier>
__device__ __forceinline__ void bilstm_bwd_body(const BwdArgs& a, int b, int dir, int tid_in, int nthr, BwdSmem& sm, Barrier&& barrier) {
  const bool live = !PADDED || tid_in < G;
  const int tid = PADDED ? min(tid_in, G - 1) : tid_in;
  const T* dout = reinterpret_cast<const T*>(a.dout);
  const int64_t Tt = a.Tt, ldd = a.ldd, col0 = a.col0;
  const int u = tid >> 2, q = tid & 3, j = q * H + u;
  const float* whh = dir ? a.whh_r : a.whh_f;
  float* dg = (dir ? a.dg_r : a.dg_f) + (int64_t)b * Tt * G;
  bf16_t* dgb = dir ? a.dgb_r : a.dgb_f;                  // (slot-uniform)
  if (dgb) dgb += (int64_t)b * Tt * G;
  const float* saved = a.saved;
  int len = a.lens[b];
  if (len > Tt) len = (int)Tt;
  float wt[H];
#pragma unroll
  for (int qq = 0; qq < H; ++qq) wt[qq] = whh[(q * H + qq) * H + u];
  for (int64_t i = (int64_t)len * G + tid_in; i < Tt * G; i += nthr) dg[i] = 0.f;
  if (dgb)
    for (int64_t i = (int64_t)len * G + tid_in; i < Tt * G; i += nthr) dgb[i] = 0;
  // per-step operands, KP steps ahead, kept RAW in the ring (any arithmetic on a loaded value - a select, the bf16 widening - at
  // fetch time makes the step wait for the load it has just issued: s_waitcnt vmcnt(0) on the chain, found in the ISA in round 4):
  // every lane its own gate (ra), lane q of a quad c (q = 0) or c_prev (q = 1) of unit u (rs), and the unit's d(out) (rd)
  auto fetch = [&](int s, float& ga, float& gs, uint32_t& gd) {
    const int t = dir ? max(len - 1 - s, 0) : s;         // (len = 0: nothing runs, the ring's priming loads stay in bounds)
    const int tq = q == 1 ? (s > 0 ? (dir ? t + 1 : t - 1) : t) : t;    // q = 1: the previous step's cell state (none at s = 0: zeroed at use)
    ga = saved[(((int64_t)b * Tt + t) * 2 + dir) * (G + 2 * H) + j];
    gs = saved[(((int64_t)b * Tt + tq) * 2 + dir) * (G + 2 * H) + G + u];
    if constexpr (sizeof(T) == 2) gd = dout[((int64_t)b * Tt + t) * ldd + col0 + dir * H + u];      // (zero-extending load: no arithmetic)
    else gd = __float_as_uint(dout[((int64_t)b * Tt + t) * ldd + col0 + dir * H + u]);
  };
  const bool q0 = q == 0, q1 = q == 1, q2 = q == 2, q3 = q == 3;
  float dc_next = 0.f, dh_rec = 0.f;
  int buf = 0;
  auto step = [&](int s, float av, float xs_raw, uint32_t xd_raw) {
    const int t = dir ? len - 1 - s : s;
    const float xs = (s == 0 && q1) ? 0.f : xs_raw;      // no previous step: c_prev = 0
    const float xd = sizeof(T) == 2 ? __uint_as_float(xd_raw << 16) : __uint_as_float(xd_raw);
    const float ig = quad_bcast<0>(av), fg = quad_bcast<1>(av), gg = quad_bcast<2>(av), og = quad_bcast<3>(av);
    const float c = quad_bcast<0>(xs), cprev = quad_bcast<1>(xs);
    const float dh = xd + dh_rec;                        // + dh from step s+1
    const float tc = tanh_fast(c);
    const float dc = fmaf(dh * og, 1.f - tc * tc, dc_next);
    // this lane's gate: d(pre-activation) = factor * derivative, by selects (no divergent code on the chain)
    //   i: dc g i(1-i)   f: dc c_prev f(1-f)   g: dc i (1-g^2)   o: dh tanh(c) o(1-o)
    const float deriv = q2 ? fmaf(-av, av, 1.f) : av * (1.f - av);
    const float other = q0 ? gg : (q1 ? cprev : ig);
    const float mine = (q3 ? dh * tc : dc * other) * deriv;
    dc_next = dc * fg;
    if (live) {
      sm.s_dg[buf][j] = mine;
      dg[(int64_t)t * G + j] = mine;
      if (dgb) dgb[(int64_t)t * G + j] = f32_to_bf16(mine);
    }
    barrier();
    const float* sd = sm.s_dg[buf] + q * H;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < H; q4 += 4) {
      const float4 gv = *reinterpret_cast<const float4*>(sd + q4);
      a0 = fmaf(wt[q4], gv.x, a0); a1 = fmaf(wt[q4 + 1], gv.y, a1);
      a2 = fmaf(wt[q4 + 2], gv.z, a2); a3 = fmaf(wt[q4 + 3], gv.w, a3);
    }
    float part = (a0 + a1) + (a2 + a3);
    part += quad_perm<0xb1>(part);                       // quad_perm [1,0,3,2]
    dh_rec = part + quad_perm<0x4e>(part);               // quad_perm [2,3,0,1]: all four lanes hold the unit's dh_prev
    buf ^= 1;                                            // (step s+2 rewrites this buffer only after the barrier of step s+1)
  };
  // The steps run len-1 ... 0.  First the len % KP odd ones, each fetching its own operands (and waiting for them); what is left is
  // a whole number of KP-step rounds whose loop body is ONE basic block: with the `s >= 0` tests of a ragged last round inside it
  // the compiler's wait-count bookkeeping gave up at the loop header and drained every outstanding load and store once per round.
  const int odd = len % KP;
  for (int s = len - 1; s >= len - odd; --s) {
    float ga, gs;
    uint32_t gd;
    fetch(s, ga, gs, gd);
    step(s, ga, gs, gd);
  }
  const int top = len - odd - 1;                         // top + 1 is a multiple of KP
  float ra[KP], rs[KP];
  uint32_t rd[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) fetch(max(top - k, 0), ra[k], rs[k], rd[k]);
  // the priming loads land before the loop is entered: the waits inside it are then priced on the loop's own round-to-round distances
  // (24+ younger operations), not on the order the scheduler gave the priming loads
#pragma unroll
  for (int k = 0; k < KP; ++k) asm volatile("" : "+v"(ra[k]), "+v"(rs[k]), "+v"(rd[k]));   // (a use: the compiler waits for them here)
  for (int s0 = top; s0 >= 0; s0 -= KP) {
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int s = s0 - k;
      // the ring slot's old values move out BEFORE the refill is issued (explicit copies): otherwise the new load lands in a
      // temporary that is copied into the slot later in the step - behind a wait for it
      float av, xs;
      uint32_t xd;
      asm volatile("v_mov_b32 %0, %1" : "=v"(av) : "v"(ra[k]));
      asm volatile("v_mov_b32 %0, %1" : "=v"(xs) : "v"(rs[k]));
      asm volatile("v_mov_b32 %0, %1" : "=v"(xd) : "v"(rd[k]));
      fetch(max(s - KP, 0), ra[k], rs[k], rd[k]);        // unconditional (clamped) refill: no branch around a load
      step(s, av, xs, xd);
    }
  }
}

// lstm.hip: dW_hh partials from the stored gate gradients and the saved hidden states - pwhh [2][B * kDwZ][G*H], to be summed over
// the middle index (a.pbias, when set: [2][B * kDwZ][G] column sums of dg, the bias gradients' partials)
int launch_dwhh_partials(const BwdArgs& a, int64_t B, hipStream_t st);

}  // namespace lstm
}  // namespace lasr
