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
};
struct BwdSmem {
  __attribute__((aligned(16))) float s_dg[G];
  __attribute__((aligned(16))) float s_hprev[H];
  float s_part[4][H];
};

// dout: columns [col0 + dir*H, +H) of a [B][T][ldd] tensor in T (gradient w.r.t. the LSTM output).
// dg [B][T][G] f32 per direction = gradient w.r.t. the gate pre-activations (zero rows for t >= len);
// pwhh [B][2][G][H] = this utterance's contribution to dW_hh.
// tid: thread index inside the recurrence's slot; nthr: threads of the slot that run this body (>= G).  PADDED: the slot has lanes
// past G (a 192-thread slot of a larger workgroup): they compute on clamped indices and store nothing.  BARRIER(): a barrier over
// exactly the waves that run recurrences in this workgroup (they all run the same number of them per step).  SOLO: one recurrence per
// workgroup (n_steps_wg == its own length): no idle phase.
template <typename T, bool PADDED, int KP, bool SOLO, typename Barrier>
__device__ __forceinline__ void bilstm_bwd_body(const BwdArgs& a, int b, int dir, int tid_in, int nthr, BwdSmem& sm, Barrier&& barrier, int n_steps_wg) {
  const bool live = !PADDED || tid_in < G;
  const int tid = PADDED ? min(tid_in, G - 1) : tid_in;
  const T* dout = reinterpret_cast<const T*>(a.dout);
  const int64_t Tt = a.Tt, ldd = a.ldd, col0 = a.col0;
  const int u = tid >> 2, q = tid & 3, j = q * H + u;                    // gate-gradient role
  const int kk = tid % H, p = tid / H;                                  // dh_prev role: hidden index kk, rows 40p .. 40p+39
  const float* whh = dir ? a.whh_r : a.whh_f;
  float* dg = (dir ? a.dg_r : a.dg_f) + (int64_t)b * Tt * G;
  const float* saved = a.saved;
  int len = a.lens[b];
  if (len > Tt) len = (int)Tt;
  float wt[H], dw[H];
#pragma unroll
  for (int qq = 0; qq < H; ++qq) { wt[qq] = whh[(p * H + qq) * H + kk]; dw[qq] = 0.f; }
  for (int64_t i = (int64_t)len * G + tid_in; i < Tt * G; i += nthr) dg[i] = 0.f;
  if (tid_in < H) { sm.s_part[0][tid_in] = 0.f; sm.s_part[1][tid_in] = 0.f; sm.s_part[2][tid_in] = 0.f; sm.s_part[3][tid_in] = 0.f; }
  // per-step operands, KP steps ahead: every lane its own gate; lane q of a quad one of (c, c_prev, h_prev, d(out)) of unit u
  auto fetch = [&](int s, float& ga, float& gb) {
    const int t = dir ? len - 1 - s : s;
    const int tp = dir ? t + 1 : t - 1;
    const float* sv = saved + (((int64_t)b * Tt + t) * 2 + dir) * (G + 2 * H);
    const float* sp = saved + (((int64_t)b * Tt + (s > 0 ? tp : t)) * 2 + dir) * (G + 2 * H);
    ga = sv[j];
    // lane q of the quad: c, c_prev, h_prev (one f32 load from a selected address) or d(out) (a T load): both issued by every
    // lane, the right one selected - no divergent branch around a load
    const float* src = q == 0 ? sv + G + u : (q == 1 ? sp + G + u : sp + G + H + u);
    const float vs = *src;
    const float vd = Elem<T>::ld(dout + ((int64_t)b * Tt + t) * ldd + col0 + dir * H + u);
    const float v = q == 3 ? vd : vs;
    gb = (s == 0 && (q == 1 || q == 2)) ? 0.f : v;       // no previous step: c_prev = h_prev = 0
  };
  float ra[KP], rb_[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    ra[k] = 0.f; rb_[k] = 0.f;
    if (len - 1 - k >= 0) fetch(len - 1 - k, ra[k], rb_[k]);
  }
  float dc_next = 0.f;
  barrier();
  // every recurrence of the workgroup walks n_steps_wg steps (the longest utterance among them) so that the barriers pair up: a
  // shorter one is through after its own len steps and only keeps the barriers company for the rest (s < 0: slot-uniform)
  for (int s0 = n_steps_wg - 1; s0 >= 0; s0 -= KP) {
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int sw = s0 - k;                             // step index of the workgroup
      if (sw >= 0) {                                     // workgroup-uniform
        const int s = sw - (n_steps_wg - len);           // this recurrence's own step: len-1 ... 0, then negative (idle)
        const bool act = SOLO || s >= 0;                 // slot-uniform (whole waves); SOLO: one recurrence per workgroup, always active
        const int sc = SOLO ? s : max(s, 0);
        const int t = dir ? len - 1 - sc : sc;
        const float av = ra[k], x = rb_[k];
        // unconditional (clamped) refill, OUTSIDE any branch: a load inside a branch is followed by s_waitcnt vmcnt(0) at its join
        fetch(max(sc - KP, 0), ra[k], rb_[k]);
        const float ig = quad_bcast<0>(av), fg = quad_bcast<1>(av), gg = quad_bcast<2>(av), og = quad_bcast<3>(av);
        const float c = quad_bcast<0>(x), cprev = quad_bcast<1>(x), hprev = quad_bcast<2>(x), dy = quad_bcast<3>(x);
        const float dh = dy + (sm.s_part[0][u] + sm.s_part[1][u]) + (sm.s_part[2][u] + sm.s_part[3][u]);    // + dh from step s+1
        const float tc = tanh_fast(c);
        const float d_o = dh * tc * og * (1.f - og);
        const float dc = fmaf(dh * og, 1.f - tc * tc, dc_next);
        float mine;
        if (q == 0) mine = dc * gg * ig * (1.f - ig);
        else if (q == 1) mine = dc * cprev * fg * (1.f - fg);
        else if (q == 2) mine = dc * ig * (1.f - gg * gg);
        else mine = d_o;
        dc_next = dc * fg;
        if (live && act) {
          sm.s_dg[j] = mine;                             // (s_part is rewritten only after the second barrier below: no hazard with the reads above)
          dg[(int64_t)t * G + j] = mine;
          if (q == 0) sm.s_hprev[u] = hprev;
        }
        barrier();
        if (act) {
          const float my = sm.s_dg[tid];                 // row tid of dW_hh
          float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
          for (int q4 = 0; q4 < H; q4 += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(sm.s_hprev + q4);
            dw[q4] = fmaf(my, hv.x, dw[q4]); dw[q4 + 1] = fmaf(my, hv.y, dw[q4 + 1]);
            dw[q4 + 2] = fmaf(my, hv.z, dw[q4 + 2]); dw[q4 + 3] = fmaf(my, hv.w, dw[q4 + 3]);
            const float4 gv = *reinterpret_cast<const float4*>(sm.s_dg + p * H + q4);
            acc0 = fmaf(wt[q4], gv.x, acc0); acc1 = fmaf(wt[q4 + 1], gv.y, acc1);
            acc0 = fmaf(wt[q4 + 2], gv.z, acc0); acc1 = fmaf(wt[q4 + 3], gv.w, acc1);
          }
          if (live) sm.s_part[p][kk] = acc0 + acc1;
        }
        barrier();
      }
    }
  }
  if (live) {
    float* o = a.pwhh + (((int64_t)b * 2 + dir) * G + tid) * H;
#pragma unroll
    for (int qq = 0; qq < H; ++qq) o[qq] = dw[qq];
  }
}

}  // namespace lstm
}  // namespace lasr
