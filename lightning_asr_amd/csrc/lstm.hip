// Bidirectional LSTM "context" branch (hidden 40, one layer) of the Context / ContextSE variants:
// pack_padded_sequence -> nn.LSTM(256, 40, bidirectional) -> pad_packed_sequence
// (models/QuartNetContext.py:171-173,186-199).  Gate order i,f,g,o; the reverse direction starts at
// each utterance's own last valid frame; outputs are zero for t >= len_b.
//
// The input projection x W_ih^T is one MFMA GEMM per direction (lasr_gemm, f32 out); what is left is
// a latency-bound recurrence of len_b dependent steps, run by one persistent workgroup per
// (utterance, direction): thread j owns gate row j with its 40 recurrent weights in registers, the
// hidden state lives in LDS.  All LSTM arithmetic is f32.
#include "common.h"
#include "lstm_body.h"
#include <math.h>

namespace lasr {

using lstm::H; using lstm::G; using lstm::kPre; using lstm::sigmoid_fast; using lstm::tanh_fast; using lstm::quad_bcast;

// Latency work: a time step is ~200 dependent instructions per wave, so the kernels are built around what sits ON that chain.
//   * thread (u, q) = (tid >> 2, tid & 3) owns gate row q*H + u: the four gates of a hidden unit live in one quad and meet
//     through DPP quad broadcasts (no LDS round trip, no barrier), every lane of the quad keeps the unit's cell state;
//   * ONE barrier per forward step (the new h vector, double-buffered in LDS), two per backward step - LDS-only barriers
//     (lds_barrier): __syncthreads() would drain the prefetched global loads and the step's stores at every step;
//   * v_exp_f32 / v_rcp_f32 sigmoid and tanh (libm expf / tanhf: ~3x the instructions);
//   * the per-step global operands (gx; the saved gates, cell states and d(out) in backward) come through a register ring
//     fetched kPre steps ahead: s_waitcnt vmcnt counts loads AND the step's stores in issue order, so a one-step prefetch
//     made every step wait for the previous step's stores to retire (forward 352 us, backward 525 us for T' = 501).
static constexpr int kLstmThreads = G;     // one thread per gate row (2.5 waves: the hardware masks the missing lanes, no `tid < G` branches -
                                           // a divergent branch around the step's loads made the compiler drain vmcnt at its join)

// gx [B][T][G] f32 per direction (x W_ih^T, no bias); saved [B][T][2][G + 2H]: gates(i,f,g,o) | c | h
// out: columns [col0 + dir*H, +H) of a [B][T][ldo] tensor in T (zeros for t >= len).
template <typename T>
__global__ __launch_bounds__(kLstmThreads) void bilstm_fwd_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_r,
                                                                  const float* __restrict__ whh_f, const float* __restrict__ whh_r,
                                                                  const float* __restrict__ bih_f, const float* __restrict__ bhh_f,
                                                                  const float* __restrict__ bih_r, const float* __restrict__ bhh_r,
                                                                  const int32_t* __restrict__ lens, int64_t Tt, T* __restrict__ out,
                                                                  int64_t ldo, int64_t col0, float* __restrict__ saved) {
  __shared__ __attribute__((aligned(16))) float s_h[2][H];
  const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
  const int u = tid >> 2, q = tid & 3, j = q * H + u;
  const float* gx = (dir ? gx_r : gx_f) + (int64_t)b * Tt * G;
  const float* whh = dir ? whh_r : whh_f;
  int len = lens[b];
  if (len > Tt) len = (int)Tt;
  float w[H];
#pragma unroll
  for (int k = 0; k < H; ++k) w[k] = whh[j * H + k];
  const float bias = (dir ? bih_r : bih_f)[j] + (dir ? bhh_r : bhh_f)[j];
  if (tid < H) { s_h[0][tid] = 0.f; s_h[1][tid] = 0.f; }
  // zero the padded frames of this direction's output slice
  for (int64_t i = (int64_t)len * H + tid; i < Tt * H; i += kLstmThreads) {
    const int64_t t = i / H;
    const int k = (int)(i - t * H);
    Elem<T>::st(out + ((int64_t)b * Tt + t) * ldo + col0 + dir * H + k, 0.f);
  }
  float c = 0.f;
  __syncthreads();
  auto step = [&](int s, float gxv) {
    const int t = dir ? len - 1 - s : s;
    float a0 = gxv + bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float* hp = s_h[s & 1];
#pragma unroll
    for (int k4 = 0; k4 < H; k4 += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(hp + k4);
      a0 = fmaf(w[k4], hv.x, a0); a1 = fmaf(w[k4 + 1], hv.y, a1); a2 = fmaf(w[k4 + 2], hv.z, a2); a3 = fmaf(w[k4 + 3], hv.w, a3);
    }
    const float pre = (a0 + a1) + (a2 + a3);
    // one exp + one rcp for either non-linearity, no divergent branch: tanh(x) = 2 sigmoid(2x) - 1
    const float sg = sigmoid_fast(q == 2 ? 2.f * pre : pre);
    const float a = q == 2 ? fmaf(2.f, sg, -1.f) : sg;
    float* sv = saved + (((int64_t)b * Tt + t) * 2 + dir) * (G + 2 * H);
    sv[j] = a;
    const float ig = quad_bcast<0>(a), fg = quad_bcast<1>(a), gg = quad_bcast<2>(a), og = quad_bcast<3>(a);
    c = fmaf(fg, c, ig * gg);
    const float h = og * tanh_fast(c);
    if (q == 0) {
      s_h[(s + 1) & 1][u] = h;
      sv[G + u] = c;
      sv[G + H + u] = h;
      Elem<T>::st(out + ((int64_t)b * Tt + t) * ldo + col0 + dir * H + u, h);
    }
    lds_barrier();
  };
  // The first len % kPre steps fetch their own operand (and wait for it); the rest is a whole number of kPre-step rounds whose loop
  // body is ONE basic block - with the `s < len` tests of a ragged last round inside it, the compiler's wait-count bookkeeping gave
  // up at the loop header and drained every outstanding load and store once per round (round 4, read off the ISA).
  const int odd = len % kPre;
  for (int s = 0; s < odd; ++s) step(s, gx[(int64_t)(dir ? len - 1 - s : s) * G + j]);
  float ring[kPre];
#pragma unroll
  for (int k = 0; k < kPre; ++k) {
    const int sn = min(odd + k, max(len - 1, 0));
    ring[k] = gx[(int64_t)(dir ? max(len - 1 - sn, 0) : sn) * G + j];
  }
#pragma unroll
  for (int k = 0; k < kPre; ++k) asm volatile("" : "+v"(ring[k]));   // the priming loads land here (see lstm_body.h: the loop's waits are priced on its own round-to-round distances)
  for (int s0 = odd; s0 < len; s0 += kPre) {
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      const int s = s0 + k;
      float gxv;                                 // the slot's old value moves out before the refill is issued (lstm_body.h)
      asm volatile("v_mov_b32 %0, %1" : "=v"(gxv) : "v"(ring[k]));
      // unconditional (clamped) refill: a load inside a branch is followed by s_waitcnt vmcnt(0) at the join - the whole HBM round
      // trip on the step's critical path (0.65 us per step measured), which is what the ring is there to hide
      const int sn = min(s + kPre, len - 1);
      ring[k] = gx[(int64_t)(dir ? len - 1 - sn : sn) * G + j];
      step(s, gxv);
    }
  }
}

// the backward recurrence: lstm_body.h (shared with the grid that runs it beside the stage's weight-gradient GEMMs)
template <typename T>
__global__ __launch_bounds__(kLstmThreads) void bilstm_bwd_kernel(lstm::BwdArgs a) {
  __shared__ lstm::BwdSmem sm;
  lstm::bilstm_bwd_body<T, false, kPre>(a, blockIdx.x, blockIdx.y, threadIdx.x, kLstmThreads, sm, [] { lds_barrier(); });
}

// dW_hh[j][k] = sum_{b,t} dg[b,t,j] h_prev[b,t,k] (h_prev = the direction's previous hidden state, nothing at its first step) from
// the stored gate gradients and the forward pass's saved states - off the recurrence's chain since round 4.  Workgroup = one
// (utterance, direction, quarter of the time axis), four 192-thread slots that each walk a sixteenth of the steps: thread j keeps row
// j of dW_hh in 40 registers, the step's hidden vector arrives through wave-uniform addresses.  The slots' sums meet in LDS (two
// rounds over a [2][G*H] image); pwhh [2][B * kDwZ][G*H] partials are summed over the middle index by the caller's reduction.
static constexpr int kDwSlices = 4, kDwSlot = 192;
__global__ __launch_bounds__(kDwSlices * kDwSlot) void bilstm_dwhh_kernel(lstm::BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float s_acc[2][G * H];
  __shared__ float s_db[kDwSlices][G];
  const int b = blockIdx.x, dir = blockIdx.y, z = blockIdx.z;
  const int slice = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kDwSlot);
  const int tid_in = (int)threadIdx.x - slice * kDwSlot;
  const bool live = tid_in < G;
  const int j = min(tid_in, G - 1);
  const int64_t Tt = a.Tt;
  int len = a.lens[b];
  if (len > Tt) len = (int)Tt;
  // steps that have a previous hidden state: forward t = 1 .. len-1 with h(t-1), reverse t = 0 .. len-2 with h(t+1)
  const int n = max(len - 1, 0), parts = lstm::kDwZ * kDwSlices;
  const int per = (n + parts - 1) / parts, part = z * kDwSlices + slice;
  const int i0 = min(part * per, n), i1 = min(i0 + per, n);
  const float* dg = (dir ? a.dg_r : a.dg_f) + (int64_t)b * Tt * G + j;
  const float* hbase = a.saved + ((int64_t)b * Tt * 2 + dir) * (G + 2 * H) + G + H;
  float dw[H];
#pragma unroll
  for (int k = 0; k < H; ++k) dw[k] = 0.f;
  // the column sum of dg (bias gradient) rides along: every step of the chunk, and the one step without a previous state once
  float db = (part == 0 && len > 0) ? dg[(int64_t)(dir ? len - 1 : 0) * G] : 0.f;
#pragma unroll 2
  for (int i = i0; i < i1; ++i) {
    const int t = dir ? i : i + 1, tp = dir ? t + 1 : t - 1;
    const float my = dg[(int64_t)t * G];
    const float* hp = hbase + (int64_t)tp * 2 * (G + 2 * H);
    db += my;
#pragma unroll
    for (int k = 0; k < H; ++k) dw[k] = fmaf(my, hp[k], dw[k]);
  }
  // slots 0, 1 lay their rows down, slots 2, 3 add theirs on top, then the two images are summed on the way out
  float* mine = s_acc[slice & 1] + j * H;
  if (live) s_db[slice][j] = db;
  if (live && slice < 2) {
#pragma unroll
    for (int k = 0; k < H; k += 4) *reinterpret_cast<float4*>(mine + k) = make_float4(dw[k], dw[k + 1], dw[k + 2], dw[k + 3]);
  }
  __syncthreads();
  if (live && slice >= 2) {
#pragma unroll
    for (int k = 0; k < H; k += 4) {
      float4 v = *reinterpret_cast<float4*>(mine + k);
      v.x += dw[k]; v.y += dw[k + 1]; v.z += dw[k + 2]; v.w += dw[k + 3];
      *reinterpret_cast<float4*>(mine + k) = v;
    }
  }
  __syncthreads();
  const int64_t row = ((int64_t)dir * gridDim.x + b) * lstm::kDwZ + z;
  float* out = a.pwhh + row * (G * H);
  for (int e = threadIdx.x; e < G * H; e += kDwSlices * kDwSlot) out[e] = s_acc[0][e] + s_acc[1][e];
  if (a.pbias && threadIdx.x < G) a.pbias[row * G + threadIdx.x] = (s_db[0][threadIdx.x] + s_db[1][threadIdx.x]) + (s_db[2][threadIdx.x] + s_db[3][threadIdx.x]);
}

int lstm::launch_dwhh_partials(const lstm::BwdArgs& a, int64_t B, hipStream_t st) {
  hipLaunchKernelGGL(bilstm_dwhh_kernel, dim3((unsigned)B, 2, lstm::kDwZ), dim3(kDwSlices * kDwSlot), 0, st, a);
  LASR_LAUNCH_CHECK("bilstm_dwhh_kernel");
  return 0;
}

// dst[n][dcol0 + c] = src[n][scol0 + c] for c < ncols (optionally += ), with dtype conversion
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void copy_cols_kernel(const TS* __restrict__ src, int64_t lds, int64_t scol0, TD* __restrict__ dst,
                                                        int64_t ldd, int64_t dcol0, int64_t rows, int64_t ncols, int accumulate) {
  const int64_t total = rows * ncols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ncols, c = i - r * ncols;
    float v = Elem<TS>::ld(src + r * lds + scol0 + c);
    if (accumulate) v += Elem<TD>::ld(dst + r * ldd + dcol0 + c);
    Elem<TD>::st(dst + r * ldd + dcol0 + c, v);
  }
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_bilstm_saved_bytes(int64_t B, int64_t T_) { return (size_t)B * T_ * 2 * (G + 2 * H) * sizeof(float); }

extern "C" int lasr_bilstm_fwd(const float* gx_f, const float* gx_r, const float* whh_f, const float* whh_r, const float* bih_f,
                               const float* bhh_f, const float* bih_r, const float* bhh_r, const int32_t* lens, int64_t B, int64_t T_,
                               void* out, int dtype, int64_t ld_out, int64_t col0, float* saved, void* stream) {
  LASR_CHECK_ARG(gx_f && gx_r && whh_f && whh_r && bih_f && bhh_f && bih_r && bhh_r && lens && out && saved, "lasr_bilstm_fwd: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_bilstm_fwd: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && col0 >= 0 && ld_out >= col0 + 2 * H, "lasr_bilstm_fwd: shape");
  dim3 grid((unsigned)B, 2);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(bilstm_fwd_kernel<float>, grid, dim3(kLstmThreads), 0, as_stream(stream), gx_f, gx_r, whh_f, whh_r, bih_f, bhh_f, bih_r, bhh_r,
                       lens, T_, (float*)out, ld_out, col0, saved);
  else
    hipLaunchKernelGGL(bilstm_fwd_kernel<bf16_t>, grid, dim3(kLstmThreads), 0, as_stream(stream), gx_f, gx_r, whh_f, whh_r, bih_f, bhh_f, bih_r, bhh_r,
                       lens, T_, (bf16_t*)out, ld_out, col0, saved);
  LASR_LAUNCH_CHECK("bilstm_fwd_kernel");
  return 0;
}

extern "C" size_t lasr_bilstm_bwd_workspace_bytes(int64_t B) { return (size_t)B * lstm::kDwZ * 2 * G * H * sizeof(float); }

extern "C" int lasr_bilstm_bwd(const void* dout, int dtype, int64_t ld_dout, int64_t col0, const float* whh_f, const float* whh_r,
                               const int32_t* lens, int64_t B, int64_t T_, const float* saved, float* dg_f, float* dg_r, float* dwhh_f,
                               float* dwhh_r, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(dout && whh_f && whh_r && lens && saved && dg_f && dg_r && dwhh_f && dwhh_r && workspace, "lasr_bilstm_bwd: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_bilstm_bwd: bad dtype");
  LASR_CHECK_SHAPE(B > 0 && B < 65536 && T_ > 0 && col0 >= 0 && ld_dout >= col0 + 2 * H, "lasr_bilstm_bwd: shape");
  if (workspace_bytes < lasr_bilstm_bwd_workspace_bytes(B)) return fail(LASR_E_WORKSPACE, "lasr_bilstm_bwd: workspace");
  float* pwhh = reinterpret_cast<float*>(workspace);
  dim3 grid((unsigned)B, 2);
  hipStream_t st = as_stream(stream);
  const lstm::BwdArgs a = {dout, ld_dout, col0, whh_f, whh_r, lens, T_, saved, dg_f, dg_r, pwhh, nullptr, nullptr, nullptr};
  if (dtype == LASR_F32) hipLaunchKernelGGL(bilstm_bwd_kernel<float>, grid, dim3(kLstmThreads), 0, st, a);
  else hipLaunchKernelGGL(bilstm_bwd_kernel<bf16_t>, grid, dim3(kLstmThreads), 0, st, a);
  LASR_LAUNCH_CHECK("bilstm_bwd_kernel");
  LASR_TRY(lstm::launch_dwhh_partials(a, B, st));
  // pwhh is [2][B * kDwZ][G*H]: per direction, B * kDwZ partials of G*H columns
  const int np = (int)B * lstm::kDwZ;
  LASR_TRY(launch_reduce_partials(pwhh, np, (int64_t)G * H, dwhh_f, (int64_t)G * H, nullptr, st));
  return launch_reduce_partials(pwhh + (int64_t)np * G * H, np, (int64_t)G * H, dwhh_r, (int64_t)G * H, nullptr, st);
}

extern "C" int lasr_copy_cols(const void* src, int src_dtype, int64_t ld_src, int64_t scol0, void* dst, int dst_dtype, int64_t ld_dst,
                              int64_t dcol0, int64_t rows, int64_t ncols, int accumulate, void* stream) {
  LASR_CHECK_ARG(src && dst, "lasr_copy_cols: null pointer");
  LASR_CHECK_ARG((src_dtype == LASR_F32 || src_dtype == LASR_BF16) && (dst_dtype == LASR_F32 || dst_dtype == LASR_BF16), "lasr_copy_cols: bad dtype");
  LASR_CHECK_SHAPE(rows > 0 && ncols > 0 && scol0 >= 0 && dcol0 >= 0 && ld_src >= scol0 + ncols && ld_dst >= dcol0 + ncols, "lasr_copy_cols: shape");
  int64_t blocks = cdiv(rows * ncols, 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = as_stream(stream);
#define LASR_CC(TS_, TD_)                                                                                                      \
  hipLaunchKernelGGL((copy_cols_kernel<TS_, TD_>), dim3((unsigned)blocks), dim3(256), 0, st, (const TS_*)src, ld_src, scol0, (TD_*)dst, \
                     ld_dst, dcol0, rows, ncols, accumulate)
  if (src_dtype == LASR_F32 && dst_dtype == LASR_F32) LASR_CC(float, float);
  else if (src_dtype == LASR_F32) LASR_CC(float, bf16_t);
  else if (dst_dtype == LASR_F32) LASR_CC(bf16_t, float);
  else LASR_CC(bf16_t, bf16_t);
#undef LASR_CC
  LASR_LAUNCH_CHECK("copy_cols_kernel");
  return 0;
}
