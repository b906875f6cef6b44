// Squeeze-excite (reduction 8, no bias) around the BatchNorm output of every separable conv of the
// ContextSE variant: s = sigmoid(W2 relu(W1 mean_T(BN(y)))), applied after BN and before the
// activation (models/QuartNetContextSE.py:8-23,55).  The mean runs over ALL T' frames, padding
// included.  Because BN is affine per channel, mean_T(BN(y)) = a_c * mean_T(y) + b_c, so the
// squeeze needs only per-(utterance, channel) sums of the pre-BN tensor: one extra read of y in
// forward; the excite scale is folded into lasr_bn_act_fwd / lasr_bn_act_bwd_*.
#include "common.h"
#include "se_seqsum.h"
#include "dropout.h"
#include "fused.h"
#include <algorithm>

namespace lasr {

// sums[b][c] = sum_t x[b][t][c].  grid (ceil(C/64), B), block 256 = 16 channel lanes (x4) x 16 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void seqsum_kernel(const T* __restrict__ x, int64_t Tt, int64_t C, float* __restrict__ sums) {
  __shared__ float s_red[16][65];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const T* xb = x + (int64_t)b * Tt * C + c;
    for (int64_t t = rl; t < Tt; t += 16) {
      float v[4];
      Elem<T>::ld4(xb + t * C, v);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) s_red[rl][cl * 4 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int64_t cc = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += s_red[r][threadIdx.x];
      sums[(int64_t)b * C + cc] = s;
    }
  }
}

// The same sums with 16-byte loads (C a multiple of the vector width, 16-byte rows): block 256 = 64/V column threads x row lanes,
// eight row loads in flight per thread, clamped rows (no branch around a load).  The first form above issues one 8-byte load per
// thread and trip: 11.5 us for the 16 MB of a 512-channel unit at cfg4.
template <typename T>
__global__ __launch_bounds__(256) void seqsum_vec_kernel(const T* __restrict__ x, int Tt, int C, float* __restrict__ sums) {
  constexpr int RL = 256 / (64 / Vec<T>::kN);
  __shared__ float s_red[RL][65];
  seqsum_vec_body<T>(x, Tt, C, sums, blockIdx.x, blockIdx.y, s_red);
}

// ds[b][c] = sum_t dout * act'(z) * (a1*y + b1),  z = (a1*y+b1)*se + (a2*y2+b2): gradient w.r.t. the scale
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                            const float* __restrict__ coef, const T* __restrict__ y2,
                                                            const float* __restrict__ coef2, const float* __restrict__ se,
                                                            int64_t Tt, int64_t C, int act, float* __restrict__ ds, DropArgs drop) {
  __shared__ float s_red[16][65];
  const bool dropping = drop.step != nullptr;            // workgroup-uniform
  const unsigned long long drop_step = dropping ? *drop.step : 0ull;
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    float a1[4], b1[4], a2[4], b2[4], sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a1[j] = coef[c + j]; b1[j] = coef[C + c + j]; sc[j] = se[(int64_t)b * C + c + j];
      a2[j] = y2 ? coef2[c + j] : 0.f; b2[j] = y2 ? coef2[C + c + j] : 0.f;
    }
    const int64_t base = (int64_t)b * Tt * C + c;
    for (int64_t t = rl; t < Tt; t += 16) {
      float dv[4], yv[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
      Elem<T>::ld4(dout + base + t * C, dv);
      Elem<T>::ld4(y + base + t * C, yv);
      if (y2) Elem<T>::ld4(y2 + base + t * C, rv);
      float dsc[4] = {1.f, 1.f, 1.f, 1.f};
      if (dropping) drop_scale<4>(drop, drop_step, (uint32_t)(base + t * C), dsc);   // same mask as bn_act_fwd (dropout.h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z1 = fmaf(yv[j], a1[j], b1[j]);
        const float z = z1 * sc[j] * (y2 ? dsc[j] : 1.f) + (y2 ? fmaf(rv[j], a2[j], b2[j]) : 0.f);
        float g = 1.f;
        if (act == LASR_ACT_RELU) g = z > 0.f ? 1.f : 0.f;
        else if (act == LASR_ACT_SWISH) { const float s = 1.f / (1.f + __expf(-z)); g = s * (1.f + z * (1.f - s)); }
        acc[j] = fmaf(dv[j] * g * dsc[j], z1, acc[j]);       // d(out)/d(scale) = [act' . dropout scale] * BN output
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) s_red[rl][cl * 4 + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int64_t cc = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (cc < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += s_red[r][threadIdx.x];
      ds[(int64_t)b * C + cc] = s;
    }
  }
}

// ---- excite MLP, forward, for the whole batch ------------------------------------------------------------------------------
// pooled = a*sums/T + b ; hidden = relu(W1 pooled) ; scale = sigmoid(W2 hidden).  The two mat-vecs are latency-, not
// bandwidth-work; one workgroup per utterance (the first form) made every workgroup stream both weight matrices - 256 KB out of
// L2 through one CU, 14.8 us per unit at cfg4.  Here each launch is dealt over (outputs x utterances), a workgroup reads only the
// weight rows of its outputs, and every wave has all of its loads in flight at once: two launches of one memory round trip each.
static constexpr int kSeHid = 4;      // hidden units per workgroup (se_hidden_kernel)
static constexpr int kSeUtt = 8;      // utterances per workgroup (both forward kernels)
static constexpr int kSeCh = 32;      // output channels per workgroup (se_scale_kernel)

// grid (ceil(H / 4), ceil(B / 8)), block 256: wave w takes utterances 8*by + w and + 4, lanes along c (coalesced rows of W1
// and of the sums); the 4 hidden units of the workgroup share every pooled value.  Workgroups of column 0 also store `pooled`.
__global__ __launch_bounds__(256) void se_hidden_kernel(const float* __restrict__ sums, const float* __restrict__ coef,
                                                        const float* __restrict__ W1, int B, int C, int H, float inv_T,
                                                        float* __restrict__ pooled, float* __restrict__ hidden) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int h0 = blockIdx.x * kSeHid;
  const int b0 = blockIdx.y * kSeUtt + wid, b1 = b0 + 4;
  const int bb0 = min(b0, B - 1), bb1 = min(b1, B - 1);          // clamped: no branch around the loads
  float acc0[kSeHid], acc1[kSeHid];
#pragma unroll
  for (int u = 0; u < kSeHid; ++u) { acc0[u] = 0.f; acc1[u] = 0.f; }
#pragma unroll 8
  for (int c = lane; c < C; c += 64) {
    const float a = coef[c], bc = coef[C + c];
    const float p0 = fmaf(a, sums[(size_t)bb0 * C + c] * inv_T, bc);
    const float p1 = fmaf(a, sums[(size_t)bb1 * C + c] * inv_T, bc);
    if (blockIdx.x == 0) {
      if (b0 < B) pooled[(size_t)b0 * C + c] = p0;
      if (b1 < B) pooled[(size_t)b1 * C + c] = p1;
    }
#pragma unroll
    for (int u = 0; u < kSeHid; ++u) {
      const float w = W1[(size_t)min(h0 + u, H - 1) * C + c];
      acc0[u] = fmaf(w, p0, acc0[u]);
      acc1[u] = fmaf(w, p1, acc1[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < kSeHid; ++u) {
    const float v0 = fmaxf(wave_sum(acc0[u]), 0.f), v1 = fmaxf(wave_sum(acc1[u]), 0.f);
    if (lane == 0 && h0 + u < H) {
      if (b0 < B) hidden[(size_t)b0 * H + h0 + u] = v0;
      if (b1 < B) hidden[(size_t)b1 * H + h0 + u] = v1;
    }
  }
}

// grid (ceil(C / 32), ceil(B / 8)), block 256 = 32 channels x 8 utterances; the workgroup's rows of W2 ([32][H], contiguous) and
// its utterances' hidden vectors sit in LDS (W2 rows at pitch H + 1: the 32 channel lanes hit 32 banks).
__global__ __launch_bounds__(256) void se_scale_kernel(const float* __restrict__ hidden, const float* __restrict__ W2, int B, int C, int H,
                                                       float* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // w2[32][H + 1] | hid[8][H]
  float* s_w2 = sm;
  float* s_hid = sm + kSeCh * (H + 1);
  const int c0 = blockIdx.x * kSeCh, b0 = blockIdx.y * kSeUtt;
  // (clamped addresses, values masked by bit operations, eight loads in flight: as `row < C ? W2[i] : 0.f` every element was a
  //  branch around its load with a wait behind it - eight memory round trips in a row in a 5.7 us kernel; round 4)
  const size_t w2_last = (size_t)C * H - 1, hid_last = (size_t)B * H - 1;
  for (int base = 0; base < kSeCh * H; base += 8 * 256) {          // all of a round's loads first, then its LDS writes
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = W2[min((size_t)c0 * H + base + u * 256 + threadIdx.x, w2_last)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + (int)threadIdx.x;
      if (i < kSeCh * H) {
        const int cl = i / H, h = i - cl * H;
        s_w2[cl * (H + 1) + h] = __uint_as_float(__float_as_uint(wv[u]) & (c0 + cl < C ? 0xffffffffu : 0u));
      }
    }
  }
  for (int base = 0; base < kSeUtt * H; base += 2 * 256) {
    float hv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) hv[u] = hidden[min((size_t)b0 * H + base + u * 256 + threadIdx.x, hid_last)];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = base + u * 256 + (int)threadIdx.x;
      if (i < kSeUtt * H) s_hid[i] = __uint_as_float(__float_as_uint(hv[u]) & (b0 + i / H < B ? 0xffffffffu : 0u));
    }
  }
  __syncthreads();
  const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
  float acc = 0.f;
  for (int h = 0; h < H; ++h) acc = fmaf(s_w2[cl * (H + 1) + h], s_hid[bl * H + h], acc);
  if (c0 + cl < C && b0 + bl < B) scale[(size_t)(b0 + bl) * C + c0 + cl] = 1.f / (1.f + expf(-acc));
}

// ---- the whole excite path INSIDE the BN + SE + add + activation pass (round 5) -----------------------------------------------
// The SE forward chain of a unit was four launches behind its GEMM: finalize + squeeze | hidden | scale | apply, the middle two at the
// launch floor (5.7 + 4.8 us, 15 units: 0.16 ms of a 3.05 ms cfg4 step) because hidden needs ALL channels of an utterance and scale
// ALL hidden units - grid-wide dependencies when the work is dealt over (outputs x utterances).  Dealt over (64-channel slab x
// utterance) instead - the grid the apply pass wants anyway - a workgroup can close both dependencies by itself at the price of
// redundancy: it recomputes its utterance's whole hidden vector (H x C MACs: 32 K at C = 512; W1 = 128 KB out of L2, 8 slabs x 32
// utterances = 32 MB per launch) and then needs only its own 64 rows of W2.  Same arithmetic in the same order as se_hidden_kernel /
// se_scale_kernel / bn_act_fwd_kernel (lane-strided fmaf chain + wave_sum; sequential fmaf over h; fmaf(y, a, b) * scale + ...), so the
// results are bit-identical to the three launches (tests/test_gpu_switches.py, LASR_SE_FWD_FOLD=0).
//   grid (C / 64, B, ts), block 256;  ts time lanes share an utterance when (C / 64) x B alone would leave CUs idle (C = 256).
// Slab 0 / time lane 0 of every utterance also stores pooled, hidden (the backward reads them); every slab its 64 scales.
template <bool HAS2>
__global__ __launch_bounds__(256) void bn_se_act_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ coef,
                                                            const bf16_t* __restrict__ y2, const float* __restrict__ coef2,
                                                            const float* __restrict__ sums, const float* __restrict__ W1,
                                                            const float* __restrict__ W2, bf16_t* __restrict__ out,
                                                            float* __restrict__ pooled, float* __restrict__ hidden,
                                                            float* __restrict__ scale, int Tt, int C, int H, float inv_T, int act_rt) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // pool[C] | hid[H] | w2[64][H + 1] | scale[64]
  float* s_pool = sm;
  float* s_hid = sm + C;
  float* s_w2 = s_hid + H;
  float* s_scale = s_w2 + 64 * (H + 1);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int c0 = blockIdx.x * 64, b = blockIdx.y, tz = blockIdx.z, nz = gridDim.z;
  const bool keeper = blockIdx.x == 0 && tz == 0;               // workgroup-uniform
  // the slab's first rows are requested before anything else: they travel under the excite arithmetic
  const int cl8 = (tid & 7) * 8, rl = tid >> 3;                  // lane's channel octet inside the slab, row lane (32 rows per pass)
  const int tlo = (int)(((int64_t)Tt * tz) / nz), thi = (int)(((int64_t)Tt * (tz + 1)) / nz);
  const bf16_t* yb = y + ((size_t)b * Tt) * C + c0 + cl8;
  const bf16_t* y2b = HAS2 ? y2 + ((size_t)b * Tt) * C + c0 + cl8 : nullptr;
  bf16_t* ob = out + ((size_t)b * Tt) * C + c0 + cl8;
  uint4 rv[2], rw[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = min(tlo + rl + 32 * u, thi - 1);
    rv[u] = Vec<bf16_t>::raw(yb + (size_t)t * C);
    if (HAS2) rw[u] = Vec<bf16_t>::raw(y2b + (size_t)t * C);
  }
  // this slab's rows of W2 -> LDS (pitch H + 1), all loads of a round first
  for (int base = 0; base < 64 * H; base += 8 * 256) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = W2[min((size_t)c0 * H + base + u * 256 + tid, (size_t)C * H - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * 256 + tid;
      if (i < 64 * H) { const int cl = i / H, h = i - cl * H; s_w2[cl * (H + 1) + h] = wv[u]; }
    }
  }
  // pooled = a * (sum_t y / T) + b for every channel of the utterance (se_hidden_kernel's expression)
  for (int c = tid; c < C; c += 256) {
    const float pv = fmaf(coef[c], sums[(size_t)b * C + c] * inv_T, coef[C + c]);
    s_pool[c] = pv;
    if (keeper) pooled[(size_t)b * C + c] = pv;
  }
  __syncthreads();
  // hidden[h] = relu(W1[h] . pooled): wave w takes rows 8 w .. 8 w + 7 (+ 32, ...); lane-strided fmaf chain per row, then the wave sum.
  //   ALL of a batch of 8 rows x 8 column steps = 64 loads are requested before the first is used (as a loop over c with four loads per
  //   trip the prologue was 32 dependent L2 round trips: the fold measured 0.066 ms per cfg4 step SLOWER than the three launches)
  for (int h0 = wid * 8; h0 < H; h0 += 32) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb < C; cb += 512) {
      float w[8][8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) w[u][i] = W1[(size_t)min(h0 + u, H - 1) * C + min(cb + lane + 64 * i, C - 1)];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = cb + lane + 64 * i;
        if (c < C) {                                            // wave-uniform (C is a multiple of 64)
          const float pv = s_pool[c];
#pragma unroll
          for (int u = 0; u < 8; ++u) acc[u] = fmaf(w[u][i], pv, acc[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float v = fmaxf(wave_sum(acc[u]), 0.f);
      if (lane == 0 && h0 + u < H) {
        s_hid[h0 + u] = v;
        if (keeper) hidden[(size_t)b * H + h0 + u] = v;
      }
    }
  }
  __syncthreads();
  if (tid < 64) {                                               // scale of the slab's channels (se_scale_kernel's chain)
    float acc = 0.f;
    for (int h = 0; h < H; ++h) acc = fmaf(s_w2[tid * (H + 1) + h], s_hid[h], acc);
    const float sc = 1.f / (1.f + expf(-acc));
    s_scale[tid] = sc;
    if (tz == 0) scale[(size_t)b * C + c0 + tid] = sc;
  }
  __syncthreads();
  // per-lane constants of its channel octet
  float a[8], bb[8], a2[8], b2[8], sc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = coef[c0 + cl8 + j]; bb[j] = coef[C + c0 + cl8 + j]; sc[j] = s_scale[cl8 + j];
    a2[j] = HAS2 ? coef2[c0 + cl8 + j] : 0.f; b2[j] = HAS2 ? coef2[C + c0 + cl8 + j] : 0.f;
  }
  with_act(act_rt, [&](auto act_c) {
    constexpr int act = decltype(act_c)::value;
    for (int t0 = tlo + rl; t0 < thi; t0 += 64) {
      // the rows after these two are requested before these are worked on
      uint4 nv[2], nw[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = min(t0 + 64 + 32 * u, thi - 1);
        nv[u] = Vec<bf16_t>::raw(yb + (size_t)t * C);
        if (HAS2) nw[u] = Vec<bf16_t>::raw(y2b + (size_t)t * C);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = t0 + 32 * u;
        float v[8], w[8], o[8];
        Vec<bf16_t>::unpack(rv[u], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = fmaf(v[j], a[j], bb[j]); o[j] *= sc[j]; }
        if (HAS2) {
          Vec<bf16_t>::unpack(rw[u], w);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += fmaf(w[j], a2[j], b2[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = act_fwd(o[j], act);
        if (t < thi) Vec<bf16_t>::store(ob + (size_t)t * C, o);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) { rv[u] = nv[u]; if (HAS2) rw[u] = nw[u]; }
    }
  });
}

// out = act((a y + b) * sigmoid(W2 relu(W1 (a mean_T(y) + b))) [+ a2 y2 + b2]) with pooled / hidden / scale left for the backward:
// the excite MLP of lasr_se_fwd and lasr_bn_act_fwd(se_scale) in one launch.  0 = launched, 1 = this shape / dtype keeps the three
// launches (nothing was launched), negative = error.
int bn_se_act_fwd(const void* y, const float* coef, const void* y2, const float* coef2, const float* sums, const float* W1, const float* W2,
                  void* out, float* pooled, float* hidden, float* scale, int dtype, int64_t B, int64_t T_, int64_t C, int act, void* stream) {
  static const bool off = getenv("LASR_SE_FWD_FOLD") && atoi(getenv("LASR_SE_FWD_FOLD")) == 0;
  const int64_t H = C / 8;
  const size_t smem = (size_t)(C + H + 64 * (H + 1) + 64) * sizeof(float);
  if (off || dtype != LASR_BF16 || C % 64 != 0 || C < 64 || C > 4096 || smem > 64 * 1024 || B >= 65536 || T_ < 1 || T_ >= (1 << 30) ||
      reinterpret_cast<uintptr_t>(y) % 16 || reinterpret_cast<uintptr_t>(out) % 16 || (y2 && reinterpret_cast<uintptr_t>(y2) % 16))
    return 1;
  LASR_CHECK_ARG(y && coef && sums && W1 && W2 && out && pooled && hidden && scale && (!y2 || coef2), "bn_se_act_fwd: null pointer");
  const int64_t wgs = (C / 64) * B;
  const unsigned ts = wgs >= 256 ? 1u : (unsigned)std::min<int64_t>(std::max<int64_t>(256 / std::max<int64_t>(wgs, 1), 1), 4);
  const dim3 grid((unsigned)(C / 64), (unsigned)B, ts);
  if (y2)
    hipLaunchKernelGGL(bn_se_act_fwd_kernel<true>, grid, dim3(256), smem, as_stream(stream), (const bf16_t*)y, coef, (const bf16_t*)y2, coef2, sums,
                       W1, W2, (bf16_t*)out, pooled, hidden, scale, (int)T_, (int)C, (int)H, 1.0f / (float)T_, act);
  else
    hipLaunchKernelGGL(bn_se_act_fwd_kernel<false>, grid, dim3(256), smem, as_stream(stream), (const bf16_t*)y, coef, (const bf16_t*)nullptr,
                       (const float*)nullptr, sums, W1, W2, (bf16_t*)out, pooled, hidden, scale, (int)T_, (int)C, (int)H, 1.0f / (float)T_, act);
  LASR_LAUNCH_CHECK("bn_se_act_fwd_kernel");
  return 0;
}

// ---- excite MLP, backward, for the whole batch -----------------------------------------------------------------------------
// Was four dependent launches per unit (fold of the raw sums, one-workgroup-per-utterance MLP backward, batched weight gradients,
// BN-backward constants: 6.8 + 15.3 + 12.2 + 16.7 us at cfg4).  Two launches, each dealt over channel chunks with the whole
// batch inside a workgroup, so the batch sums (weight gradients, BN constants) close inside the workgroup that owns the channels:
//   se_bwd_hidden_kernel (chunk of 32 channels): ds -> d2 = ds*s*(1-s) ; dW2[c][:] = sum_b d2[b][c] hidden[b][:] ;
//                                                 dh_part[chunk][b][h] = sum_{c in chunk} W2[c][h] d2[b][c]
//   se_bwd_pool_kernel   (chunk of 16 channels): dW2 = sum_groups dW2_part ; dh = relu'(hidden) * sum_chunks dh_part ; seg[b][c] = sum_h W1[h][c] dh[b][h] / T ;
//                                                 dW1[:][c] = sum_b dh[b][:] pooled[b][c] ; [BN constants of the chunk's channels]
static constexpr int kSeC1 = 32, kSeC2 = 16;

// Fused form (bn.partials != nullptr): ds comes from the per-(utterance, slab) raw sums P[b][k][c], k = 0..3: sum_t dm,
// sum_t dm*xhat1 (main branch, dm = gradient reaching the BN output without SE factors), sum_t d, sum_t d*xhat2 (residual branch):
//   ds[b][c] = sum_t dm * z1 = gamma_c * P1 + beta_c * P0      (z1 = BN output = gamma*xhat1 + beta)
// - no pass of its own over (dout, y, y2).  The folded sums P [B][4][C] are kept for the second launch.
// grid (C / 32, ceil(B / 8)), block 256 = 8 utterances x 32 channels (the fold of the raw sums is the bulk of the bytes - 4 MB per
// unit at cfg4 - so it is dealt over utterance groups as well: with the whole batch in one workgroup per chunk it took 12 us).
// dW2 therefore leaves as one partial per utterance group, dW2_part[group][C][H], summed by the second launch.
__global__ __launch_bounds__(256) void se_bwd_hidden_kernel(const float* __restrict__ ds, const float* __restrict__ partials, int nslab,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ scale, const float* __restrict__ hidden,
                                                            const float* __restrict__ W2, int B, int C, int H, float* __restrict__ P,
                                                            float* __restrict__ dW2_part, float* __restrict__ dh_part) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // d2[8][32] | hid[8][H] | w2[32][H]
  float* s_d2 = sm;
  float* s_hid = s_d2 + kSeUtt * kSeC1;
  float* s_w2 = s_hid + kSeUtt * H;
  const int c0 = blockIdx.x * kSeC1, b0 = blockIdx.y * kSeUtt;
  {
    const size_t hid_last = (size_t)B * H - 1;             // (clamped address + bit mask: no branch around the load)
#pragma unroll 2
    for (int i = threadIdx.x; i < kSeUtt * H; i += 256)
      s_hid[i] = __uint_as_float(__float_as_uint(hidden[min((size_t)b0 * H + i, hid_last)]) & (b0 + i / H < B ? 0xffffffffu : 0u));
  }
#pragma unroll 8
  for (int i = threadIdx.x; i < kSeC1 * H; i += 256) s_w2[i] = W2[(size_t)c0 * H + i];      // C % 32 == 0: whole chunks only
  {
    const int bl = threadIdx.x >> 5, b = min(b0 + bl, B - 1), c = c0 + (threadIdx.x & 31);
    float dsv;
    // (the three per-channel / per-utterance operands of the tail are requested with the slabs' loads, not one after the other behind them)
    const float gm = partials ? gamma[c] : 0.f, bt = partials ? beta[c] : 0.f;
    const float s = scale[(size_t)b * C + c];
    if (partials) {
      double a[4] = {0.0, 0.0, 0.0, 0.0};
      const float* p = partials + (size_t)b * nslab * 4 * C + c;
#pragma unroll 4
      for (int sl = 0; sl < nslab; ++sl) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += (double)p[((size_t)sl * 4 + k) * C];
      }
      if (b0 + bl < B) {
#pragma unroll
        for (int k = 0; k < 4; ++k) P[((size_t)b * 4 + k) * C + c] = (float)a[k];
      }
      dsv = (float)((double)gm * a[1] + (double)bt * a[0]);
    } else {
      dsv = ds[(size_t)b * C + c];
    }
    s_d2[threadIdx.x] = b0 + bl < B ? dsv * s * (1.f - s) : 0.f;      // utterances past the batch contribute nothing
  }
  __syncthreads();
  // this utterance group's share of the chunk's dW2 rows: [32][H], consecutive threads along h
  float* w2p = dW2_part + ((size_t)blockIdx.y * C + c0) * H;
  for (int i = threadIdx.x; i < kSeC1 * H; i += 256) {
    const int cl = i / H, h = i - cl * H;
    float acc = 0.f;
#pragma unroll
    for (int b = 0; b < kSeUtt; ++b) acc = fmaf(s_d2[b * kSeC1 + cl], s_hid[b * H + h], acc);
    w2p[i] = acc;
  }
  // this chunk's share of d(hidden pre-activation), before relu', for the group's utterances
  float* out = dh_part + ((size_t)blockIdx.x * B + b0) * H;
  for (int i = threadIdx.x; i < kSeUtt * H; i += 256) {
    const int bl = i / H, h = i - bl * H;
    float acc = 0.f;
#pragma unroll 8
    for (int cl = 0; cl < kSeC1; ++cl) acc = fmaf(s_w2[cl * H + h], s_d2[bl * kSeC1 + cl], acc);
    if (b0 + bl < B) out[i] = acc;
  }
}

// Fused form (bn.tab != nullptr): the BN-backward constants of the chunk's channels from the per-utterance sums, the SE scale and
// the pooled-path gradient seg:
//   d1 = dm*se + seg  =>  s1 = sum_b (se*P0 + T*seg),  s2 = sum_b (se*P1 + seg*X1),  X1[b][c] = sum_t xhat1 = (sum_t y - T*mean)*rstd
__global__ __launch_bounds__(512) void se_bwd_pool_kernel(const float* __restrict__ dh_part, int nchunk, const float* __restrict__ hidden,
                                                          const float* __restrict__ pooled, const float* __restrict__ W1,
                                                          const float* __restrict__ scale, const float* __restrict__ P,
                                                          const float* __restrict__ dW2_part, int ngroup, int B, int Tt, int C, int H,
                                                          float* __restrict__ seg, float* __restrict__ dW1, float* __restrict__ dW2,
                                                          SeBwdBn bn) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dh[B][H] | w1[H][16] | pool[B][16] | seg[B][16] | red[4][32][16] f64
  float* s_dh = sm;
  float* s_w1 = s_dh + (size_t)B * H;
  float* s_pool = s_w1 + (size_t)H * kSeC2;
  float* s_seg = s_pool + (size_t)B * kSeC2;
  const int c0 = blockIdx.x * kSeC2;
  const float inv_T = 1.f / (float)Tt;
  // The first round of every array this phase stages (W1's 16 columns, the pooled values, the groups' dW2 shares) is requested
  // before anything is waited for - with the d(hidden) shares below, ONE memory round trip for the usual shapes (C <= 512, B <= 32)
  // where each staging loop used to be its own.
  const int n_w1 = H * kSeC2, n_pl = B * kSeC2;
  float r_w1[2], r_pl, r_g[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = min(u * 512 + (int)threadIdx.x, n_w1 - 1);
    r_w1[u] = W1[(size_t)(i >> 4) * C + c0 + (i & 15)];
  }
  {
    const int i = min((int)threadIdx.x, n_pl - 1);
    r_pl = pooled[(size_t)(i >> 4) * C + c0 + (i & 15)];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const size_t row = ((size_t)min(u, ngroup - 1) * C + c0) * H;
    const uint32_t mk = u < ngroup ? 0xffffffffu : 0u;
#pragma unroll
    for (int e = 0; e < 2; ++e) r_g[e][u] = __uint_as_float(__float_as_uint(dW2_part[row + min(e * 512 + (int)threadIdx.x, n_w1 - 1)]) & mk);
  }
  // ... and so are the per-(utterance, channel) operands of the LAST phase (fused form): they depend on nothing this kernel computes
  const int pf_cl = threadIdx.x & 15, pf_b = min((int)(threadIdx.x >> 4), B - 1), pf_c = c0 + pf_cl;
  float pf_p[4] = {0.f, 0.f, 0.f, 0.f}, pf_sc = 0.f, pf_ys = 0.f, pf_mean = 0.f, pf_rstd = 0.f;
  if (bn.tab) {                                                 // workgroup-uniform
#pragma unroll
    for (int k = 0; k < 4; ++k) pf_p[k] = P[((size_t)pf_b * 4 + k) * C + pf_c];
    pf_sc = scale[(size_t)pf_b * C + pf_c];
    pf_ys = bn.ysum[(size_t)pf_b * C + pf_c];
    pf_mean = bn.saved[pf_c]; pf_rstd = bn.saved[C + pf_c];
  }
  __builtin_amdgcn_sched_barrier(0);
  // d(hidden) = relu'(hidden) * sum over the channel chunks' shares, chunks in index order.  Four elements per thread and eight chunks
  // per round are in flight together (clamped indices, values masked by bit operations): written as one element after the other, every
  // element waited for its own 16 loads - four to eight memory round trips in a row at the head of a 11 us kernel (round 4)
  {
    const int n_e = B * H;
    for (int i0 = 0; i0 < n_e; i0 += 4 * 512) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      int ie[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) ie[e] = min(i0 + e * 512 + (int)threadIdx.x, n_e - 1);
      float hv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hv[e] = hidden[ie[e]];     // (with the first round's loads, not one by one behind them)
      for (int k0 = 0; k0 < nchunk; k0 += 8) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const size_t row = (size_t)min(k0 + u, nchunk - 1) * n_e;
          const uint32_t mk = k0 + u < nchunk ? 0xffffffffu : 0u;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e][u] = __uint_as_float(__float_as_uint(dh_part[row + ie[e]]) & mk);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += v[e][u];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = i0 + e * 512 + (int)threadIdx.x;
        if (i < n_e) s_dh[i] = hv[e] > 0.f ? acc[e] : 0.f;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (u * 512 + (int)threadIdx.x < n_w1) s_w1[u * 512 + threadIdx.x] = r_w1[u];
  for (int i = 2 * 512 + threadIdx.x; i < n_w1; i += 512) s_w1[i] = W1[(size_t)(i >> 4) * C + c0 + (i & 15)];
  if ((int)threadIdx.x < n_pl) s_pool[threadIdx.x] = r_pl;
  for (int i = 512 + threadIdx.x; i < n_pl; i += 512) s_pool[i] = pooled[(size_t)(i >> 4) * C + c0 + (i & 15)];
  // dW2 rows of the chunk ([16][H], contiguous): the utterance groups' partials in a fixed order
  for (int i0 = 0; i0 < kSeC2 * H; i0 += 2 * 512) {         // (two elements x four groups in flight, groups in index order)
    const int n_w = kSeC2 * H;
    float acc[2] = {0.f, 0.f};
    int ie[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) ie[e] = min(i0 + e * 512 + (int)threadIdx.x, n_w - 1);
    for (int g0 = 0; g0 < ngroup; g0 += 4) {
      float v[2][4];
      if (i0 == 0 && g0 == 0) {                           // workgroup-uniform: requested at the top of the kernel
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 2; ++e) v[e][u] = r_g[e][u];
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t row = ((size_t)min(g0 + u, ngroup - 1) * C + c0) * H;
          const uint32_t mk = g0 + u < ngroup ? 0xffffffffu : 0u;
#pragma unroll
          for (int e = 0; e < 2; ++e) v[e][u] = __uint_as_float(__float_as_uint(dW2_part[row + ie[e]]) & mk);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[e] += v[e][u];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = i0 + e * 512 + (int)threadIdx.x;
      if (i < n_w) dW2[(size_t)c0 * H + i] = acc[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B * kSeC2; i += 512) {
    const int b = i >> 4, cl = i & 15;
    float acc = 0.f;
    for (int h = 0; h < H; ++h) acc = fmaf(s_w1[h * kSeC2 + cl], s_dh[b * H + h], acc);
    acc *= inv_T;
    s_seg[i] = acc;
    seg[(size_t)b * C + c0 + cl] = acc;
  }
  for (int i = threadIdx.x; i < H * kSeC2; i += 512) {
    const int h = i >> 4, cl = i & 15;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc = fmaf(s_dh[b * H + h], s_pool[b * kSeC2 + cl], acc);
    dW1[(size_t)h * C + c0 + cl] = acc;
  }
  if (!bn.tab) return;                                          // workgroup-uniform
  __syncthreads();
  // 32 utterance lanes x 16 channels, f64 sums, fixed-order fold
  size_t red_off = (size_t)(B * H + H * kSeC2 + 2 * B * kSeC2) * sizeof(float);
  red_off = (red_off + 7) & ~(size_t)7;
  double* s_red = reinterpret_cast<double*>(reinterpret_cast<char*>(sm) + red_off);   // [4][32][16] | [4][16]
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4, c = c0 + cl;
  const float mean = pf_mean, rstd = pf_rstd;
  double s1 = 0.0, s2 = 0.0, q1 = 0.0, q2 = 0.0;
  for (int b = bl; b < B; b += 32) {
    const bool first = b == bl;                                 // (requested at the top of the kernel)
    const float p0 = first ? pf_p[0] : P[((size_t)b * 4 + 0) * C + c], p1 = first ? pf_p[1] : P[((size_t)b * 4 + 1) * C + c];
    const float sc = first ? pf_sc : scale[(size_t)b * C + c], sg = s_seg[b * kSeC2 + cl];
    const float x1 = ((first ? pf_ys : bn.ysum[(size_t)b * C + c]) - (float)Tt * mean) * rstd;
    s1 += (double)sc * p0 + (double)Tt * sg;
    s2 += (double)sc * p1 + (double)sg * x1;
    q1 += (double)(first ? pf_p[2] : P[((size_t)b * 4 + 2) * C + c]);
    q2 += (double)(first ? pf_p[3] : P[((size_t)b * 4 + 3) * C + c]);
  }
  s_red[(0 * 32 + bl) * kSeC2 + cl] = s1; s_red[(1 * 32 + bl) * kSeC2 + cl] = s2;
  s_red[(2 * 32 + bl) * kSeC2 + cl] = q1; s_red[(3 * 32 + bl) * kSeC2 + cl] = q2;
  __syncthreads();
  double* s_tot = s_red + 4 * 32 * kSeC2;
  if (threadIdx.x < 4 * kSeC2) {                                // one thread per (quantity, channel): 32 terms in lane order
    const int q = threadIdx.x >> 4;
    double t = 0.0;
    for (int l = 0; l < 32; ++l) t += s_red[(q * 32 + l) * kSeC2 + cl];
    s_tot[q * kSeC2 + cl] = t;
  }
  __syncthreads();
  if (threadIdx.x < kSeC2)
    bn_bwd_table_channel(c, C, (float)s_tot[cl], (float)s_tot[kSeC2 + cl], (float)s_tot[2 * kSeC2 + cl], (float)s_tot[3 * kSeC2 + cl],
                         bn.coef2 != nullptr, bn.coef, bn.saved, bn.gamma, bn.coef2, bn.saved2, bn.gamma2, bn.inv_n, bn.tab, bn.dgamma, bn.dbeta,
                         bn.dgamma2, bn.dbeta2);
}

}  // namespace lasr

namespace lasr {
static size_t se_pool_smem(int64_t B, int64_t H) {
  size_t n = (size_t)(B * H + H * kSeC2 + 2 * B * kSeC2) * sizeof(float);
  n = (n + 7) & ~(size_t)7;
  return n + (size_t)(4 * 32 * kSeC2 + 4 * kSeC2) * sizeof(double);
}
static size_t se_hidden_smem(int64_t H) { return (size_t)(kSeUtt * kSeC1 + kSeUtt * H + kSeC1 * H) * sizeof(float); }

// work = P [B][4][C] | dh_part [C/32][B][C/8] | dW2_part [ceil(B/8)][C][C/8]
size_t se_bwd_work_bytes(int64_t B, int64_t C) {
  return align_up((size_t)B * 4 * C * sizeof(float), 256) + align_up((size_t)(C / kSeC1) * B * (C / 8) * sizeof(float), 256) +
         align_up((size_t)cdiv(B, kSeUtt) * C * (C / 8) * sizeof(float), 256);
}

int launch_se_bwd(const float* ds, const SeBwdBn* bn, const float* scale, const float* hidden, const float* pooled, const float* W1,
                  const float* W2, int64_t B, int64_t T_, int64_t C, float* seg, float* dW1, float* dW2, void* work, hipStream_t st) {
  const int H = (int)(C / 8);
  if (C % 32 != 0 || se_pool_smem(B, H) > 64 * 1024 || se_hidden_smem(H) > 64 * 1024 || B >= 65536 * kSeUtt)
    return fail(LASR_E_SHAPE, "lasr_se_bwd: C=%lld B=%lld (C a multiple of 32; the batch's excite vectors must fit 64 KB of LDS)",
                (long long)C, (long long)B);
  char* w = reinterpret_cast<char*>(work);
  float* P = reinterpret_cast<float*>(w);
  w += align_up((size_t)B * 4 * C * sizeof(float), 256);
  float* dh_part = reinterpret_cast<float*>(w);
  w += align_up((size_t)(C / kSeC1) * B * H * sizeof(float), 256);
  float* dW2_part = reinterpret_cast<float*>(w);
  const int nchunk = (int)(C / kSeC1), ngroup = (int)cdiv(B, kSeUtt);
  const bool fused = bn && bn->partials;
  hipLaunchKernelGGL(se_bwd_hidden_kernel, dim3((unsigned)nchunk, (unsigned)ngroup), dim3(256), se_hidden_smem(H), st, ds,
                     fused ? bn->partials : nullptr, fused ? bn->nslab : 0, fused ? bn->gamma : nullptr, fused ? bn->beta : nullptr, scale,
                     hidden, W2, (int)B, (int)C, H, P, dW2_part, dh_part);
  LASR_LAUNCH_CHECK("se_bwd_hidden_kernel");
  SeBwdBn b2;
  if (fused) b2 = *bn;
  else { memset(&b2, 0, sizeof(b2)); }
  hipLaunchKernelGGL(se_bwd_pool_kernel, dim3((unsigned)(C / kSeC2)), dim3(512), se_pool_smem(B, H), st, dh_part, nchunk, hidden, pooled, W1,
                     scale, P, dW2_part, ngroup, (int)B, (int)T_, (int)C, H, seg, dW1, dW2, b2);
  LASR_LAUNCH_CHECK("se_bwd_pool_kernel");
  return 0;
}
}  // namespace lasr

using namespace lasr;

extern "C" int lasr_seqsum(const void* x, int dtype, int64_t B, int64_t T_, int64_t C, float* sums, void* stream) {
  LASR_CHECK_ARG(x && sums && (dtype == LASR_F32 || dtype == LASR_BF16), "lasr_seqsum: bad argument");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && C > 0 && C % 4 == 0, "lasr_seqsum: shape");
  dim3 grid((unsigned)cdiv(C, 64), (unsigned)B);
  const int v = dtype == LASR_F32 ? 4 : 8;
  const bool vec = C % v == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && T_ < ((int64_t)1 << 30) && C < ((int64_t)1 << 30);
  if (vec && dtype == LASR_F32) hipLaunchKernelGGL(seqsum_vec_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (int)T_, (int)C, sums);
  else if (vec) hipLaunchKernelGGL(seqsum_vec_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x, (int)T_, (int)C, sums);
  else if (dtype == LASR_F32) hipLaunchKernelGGL(seqsum_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, T_, C, sums);
  else hipLaunchKernelGGL(seqsum_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x, T_, C, sums);
  LASR_LAUNCH_CHECK("seqsum_kernel");
  return 0;
}

extern "C" int lasr_se_fwd(const float* sums, const float* coef, const float* W1, const float* W2, int64_t B, int64_t T_, int64_t C,
                           float* pooled, float* hidden, float* scale, void* stream) {
  LASR_CHECK_ARG(sums && coef && W1 && W2 && pooled && hidden && scale, "lasr_se_fwd: null pointer");
  LASR_CHECK_SHAPE(B > 0 && T_ > 0 && C >= 32 && C % 32 == 0 && C <= 8192, "lasr_se_fwd: C=%lld", (long long)C);
  const int H = (int)(C / 8);
  LASR_CHECK_SHAPE(B < 65536 * kSeUtt, "lasr_se_fwd: B=%lld", (long long)B);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(se_hidden_kernel, dim3((unsigned)cdiv(H, kSeHid), (unsigned)cdiv(B, kSeUtt)), dim3(256), 0, st, sums, coef, W1, (int)B,
                     (int)C, H, 1.0f / (float)T_, pooled, hidden);
  LASR_LAUNCH_CHECK("se_hidden_kernel");
  hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)cdiv(C, kSeCh), (unsigned)cdiv(B, kSeUtt)), dim3(256),
                     (size_t)(kSeCh * (H + 1) + kSeUtt * H) * sizeof(float), st, hidden, W2, (int)B, (int)C, H, scale);
  LASR_LAUNCH_CHECK("se_scale_kernel");
  return 0;
}

extern "C" size_t lasr_se_bwd_workspace_bytes(int64_t B, int64_t C) {
  // ds [B][C] | the two launches' hand-over (se_bwd_work_bytes)
  return align_up((size_t)B * C * sizeof(float), 256) + se_bwd_work_bytes(B, C);
}

extern "C" int lasr_se_bwd_drop(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                                const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T_,
                                int64_t C, int act, const lasr_dropout* dropout, float* seg, float* dW1, float* dW2, void* workspace,
                                size_t workspace_bytes, void* stream);
extern "C" int lasr_se_bwd(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                           const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T_,
                           int64_t C, int act, float* seg, float* dW1, float* dW2, void* workspace, size_t workspace_bytes,
                           void* stream) {
  return lasr_se_bwd_drop(dout, y, coef, y2, coef2, scale, hidden, pooled, W1, W2, dtype, B, T_, C, act, nullptr, seg, dW1, dW2, workspace,
                          workspace_bytes, stream);
}

extern "C" int lasr_se_bwd_drop(const void* dout, const void* y, const float* coef, const void* y2, const float* coef2, const float* scale,
                                const float* hidden, const float* pooled, const float* W1, const float* W2, int dtype, int64_t B, int64_t T_,
                                int64_t C, int act, const lasr_dropout* dropout, float* seg, float* dW1, float* dW2, void* workspace,
                                size_t workspace_bytes, void* stream) {
  DropArgs da;
  da.step = nullptr; da.seed = 0; da.unit = 0; da.thresh = 0; da.inv_keep = 1.f;
  if (dropout && dropout->step && dropout->p > 0.f) {    // (same conversion as norm.hip's make_drop)
    da.step = reinterpret_cast<const unsigned long long*>(dropout->step);
    da.seed = dropout->seed; da.unit = dropout->unit;
    const float p = dropout->p < 0.999f ? dropout->p : 0.999f;
    da.thresh = (uint32_t)(p * 65536.f + 0.5f);
    da.inv_keep = 1.f / (1.f - (float)da.thresh / 65536.f);
  }
  LASR_CHECK_ARG(dout && y && coef && scale && hidden && pooled && W1 && W2 && seg && dW1 && dW2 && workspace, "lasr_se_bwd: null pointer");
  LASR_CHECK_ARG(!y2 || coef2, "lasr_se_bwd: residual coefficients");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_se_bwd: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && C >= 8 && C % 8 == 0 && C <= 8192, "lasr_se_bwd: C=%lld", (long long)C);
  if (workspace_bytes < lasr_se_bwd_workspace_bytes(B, C)) return fail(LASR_E_WORKSPACE, "lasr_se_bwd: workspace");
  char* w = reinterpret_cast<char*>(workspace);
  float* ds = reinterpret_cast<float*>(w);
  void* work = w + align_up((size_t)B * C * sizeof(float), 256);
  dim3 grid((unsigned)cdiv(C, 64), (unsigned)B);
  hipStream_t st = as_stream(stream);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(se_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)y, coef, (const float*)y2,
                       coef2, scale, T_, C, act, ds, da);
  else
    hipLaunchKernelGGL(se_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)y, coef,
                       (const bf16_t*)y2, coef2, scale, T_, C, act, ds, da);
  LASR_LAUNCH_CHECK("se_bwd_reduce_kernel");
  return launch_se_bwd(ds, nullptr, scale, hidden, pooled, W1, W2, B, T_, C, seg, dW1, dW2, work, st);
}
