// Layout, masking-length, BatchNorm (training statistics) and fused BN-apply/residual/activation
// kernels, forward and backward, over channels-last [B][T][C] activations.
// Replaces models/QuartNet.py:33-37 (MaskCNN lengths, BatchNorm1d(eps=1e-3), ReLU) and :74-77
// (residual add + ReLU), plus their autograd backward.  All statistics are f32.
#include "common.h"
#include "reduce_body.h"
#include "se_seqsum.h"
#include "fused.h"
#include "dropout.h"
#include <algorithm>
#include <string.h>

namespace lasr {

// ------------------------------------------------------------------ layout ------------------
// (B, C, T) f32 <-> [B][T][C] T via a 32x32 LDS tile.
template <typename T, bool TO_BTC>
__global__ __launch_bounds__(256) void transpose_kernel(const void* __restrict__ in_, void* __restrict__ out_,
                                                        int64_t C, int64_t Tt) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.y * 32, t0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (TO_BTC) {
    const float* in = reinterpret_cast<const float*>(in_) + (int64_t)b * C * Tt;
    T* out = reinterpret_cast<T*>(out_) + (int64_t)b * C * Tt;
    for (int i = ty; i < 32; i += 8)
      if (c0 + i < C && t0 + tx < Tt) tile[i][tx] = in[(c0 + i) * Tt + t0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
      if (t0 + i < Tt && c0 + tx < C) Elem<T>::st(out + (t0 + i) * C + c0 + tx, tile[tx][i]);
  } else {
    const T* in = reinterpret_cast<const T*>(in_) + (int64_t)b * C * Tt;
    float* out = reinterpret_cast<float*>(out_) + (int64_t)b * C * Tt;
    for (int i = ty; i < 32; i += 8)
      if (t0 + i < Tt && c0 + tx < C) tile[i][tx] = Elem<T>::ld(in + (t0 + i) * C + c0 + tx);
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
      if (c0 + i < C && t0 + tx < Tt) out[(c0 + i) * Tt + t0 + tx] = tile[tx][i];
  }
}

__global__ void mask_lengths_kernel(const float* __restrict__ pct, int64_t B, float Tf, int32_t* __restrict__ lens,
                                    unsigned long long* __restrict__ step_counter) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < B) lens[i] = (int32_t)(Tf * pct[i]);  // f32 product, truncation toward zero (torch .int())
  if (i == 0 && step_counter) *step_counter += 1ull;   // index of this training forward: the dropout masks' counter (dropout.h)
}

// The two launches every training forward starts with - lens = int(T' * pct) (models/QuartNet.py:311) and the bf16 shadow of the
// weights - are independent of each other: one grid (round 4; the last workgroup does the lengths).
// grid: [cast_blocks of the weight shadow | the depthwise tap tables' blocks (fused.h), one 256-entry block each | the lengths]
__global__ __launch_bounds__(256) void mask_lengths_cast_kernel(const float* __restrict__ pct, int64_t B, float Tf, int32_t* __restrict__ lens,
                                                                unsigned long long* __restrict__ step_counter, const float* __restrict__ in,
                                                                bf16_t* __restrict__ out, int64_t n, int cast_blocks, DwTapJobs jobs) {
  if (blockIdx.x == gridDim.x - 1) {                     // workgroup-uniform
    for (int64_t i = threadIdx.x; i < B; i += 256) lens[i] = (int32_t)(Tf * pct[i]);
    if (threadIdx.x == 0 && step_counter) *step_counter += 1ull;
    return;
  }
  if ((int)blockIdx.x >= cast_blocks) {                  // workgroup-uniform: one block of one layer's tap tables
    const int tb = (int)blockIdx.x - cast_blocks;
    int l = 0;
    while (l + 1 < jobs.n && jobs.blk0[l + 1] <= tb) ++l;
    const int e = (tb - jobs.blk0[l]) * 256 + (int)threadIdx.x;
    if (e < 2 * jobs.C[l] * kDwTapRow) jobs.out[l][e] = dw_tap_entry(jobs.w[l], jobs.C[l], jobs.k[l], e);
    return;
  }
  const int64_t n4 = n >> 2;                             // 16-byte loads, 8-byte stores (the buffers are 256-byte aligned)
  const int64_t stride = (int64_t)cast_blocks * 256;
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < n4; e += stride) {
    const float4 v = reinterpret_cast<const float4*>(in)[e];
    uint2 o;
    o.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    o.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    reinterpret_cast<uint2*>(out)[e] = o;
  }
  if (blockIdx.x == 0)
    for (int64_t e = (n4 << 2) + threadIdx.x; e < n; e += 256) out[e] = f32_to_bf16(in[e]);
}

// ------------------------------------------------------------------ BN finalize --------------
__device__ __forceinline__ void bn_finalize_channel(float s, float q, int64_t c, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ rmean,
                                                    float* __restrict__ rvar, float* __restrict__ coef, float* __restrict__ saved,
                                                    int64_t C, float n, float eps, float momentum, int training) {
  float mean, var;
  if (training) {
    // sums arrive in f32; the subtraction is done in double to keep E[x^2]-E[x]^2 benign
    const double m = (double)s / n;
    double v = (double)q / n - m * m;
    if (v < 0) v = 0;
    mean = (float)m;
    var = (float)v;
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      const float unbiased = n > 1.f ? (float)(v * (double)n / ((double)n - 1.0)) : var;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
    }
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  const float a = gamma[c] * rstd;
  coef[c] = a;
  coef[C + c] = beta[c] - mean * a;
  if (saved) { saved[c] = mean; saved[C + c] = rstd; }
}

__global__ void bn_finalize_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float* __restrict__ coef, float* __restrict__ saved, int64_t C, float n, float eps,
                                   float momentum, int training) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_finalize_channel(training ? stats[c] : 0.f, training ? stats[C + c] : 0.f, c, gamma, beta, rmean, rvar, coef, saved, C, n, eps,
                      momentum, training);
}

// eval mode: the coefficients of up to 64 BN layers from their running statistics in ONE launch (an eval forward issued
// one tiny launch per layer: 28 x ~5 us of pure latency for the 15 units of the plain model)
struct BnEvalMany { lasr_bn_eval_desc d[64]; };
__global__ __launch_bounds__(256) void bn_eval_coef_many_kernel(BnEvalMany a, float eps) {
  const lasr_bn_eval_desc q = a.d[blockIdx.y];
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= q.C) return;
  bn_finalize_channel(0.f, 0.f, c, q.gamma, q.beta, const_cast<float*>(q.running_mean), const_cast<float*>(q.running_var), q.coef, nullptr,
                      q.C, 1.f, eps, 0.f, 0);
}

// ------------------------------------------------------------------ fixed-order partial sums ----
// block = 32 columns x 8 partial lanes; each lane strides the partial rows with 4 independent sums.
// Accumulation is f64: these sums are BatchNorm statistics and gradient reductions, and the
// network's backward map amplifies relative noise in them by ~1e2-1e3 (measured), so f32 sums
// of a few hundred partials cost visible gradient parity.
// one partial lane's share of column c: rows pl, pl+8, ... in a fixed order.  The partials were written by the
// previous kernel on other XCDs, so every load is a trip to the fabric: 16 rows are requested before the
// first is added (a loop of 4 loads per trip exposed that latency ~16 times for 501 rows: 8.6 us per call).
template <int LANES = 8>
__device__ __forceinline__ double partial_lane_sum(const float* __restrict__ partials, int n_part, int64_t ncols, int64_t c, int pl) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int p = pl; p < n_part; p += 16 * LANES) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = p + LANES * i;
      const float x = partials[(int64_t)min(r, n_part - 1) * ncols + c];   // clamped address, masked value: no branch around the load
      v[i] = r < n_part ? x : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      a0 += (double)v[i]; a1 += (double)v[i + 1]; a2 += (double)v[i + 2]; a3 += (double)v[i + 3];
    }
  }
  return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int n_part, int64_t ncols,
                                                              float* __restrict__ out0, int64_t split, float* __restrict__ out1) {
  __shared__ double s_acc[8][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int64_t c = (int64_t)blockIdx.x * 32 + cl;
  s_acc[pl][cl] = c < ncols ? partial_lane_sum(partials, n_part, ncols, c, pl) : 0.0;
  __syncthreads();
  if (pl == 0 && c < ncols) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += s_acc[i][cl];
    if (c < split) out0[c] = (float)s;
    else if (out1) out1[c - split] = (float)s;
  }
}

// GEMM-epilogue BN partial sums ([tile][sum | sumsq][C]) -> training-mode BN coefficients, for the main
// and the residual branch of a unit in ONE launch (blockIdx.y = branch): the fixed-order f64 column
// reduction of reduce_partials_kernel and the arithmetic of bn_finalize_kernel, without the f32 round
// trip of the statistics through HBM and without three of the four launches.
struct BnBranch {
  const float* partials; int n_part;
  const float* gamma; const float* beta; float* rmean; float* rvar; float* coef; float* saved; float* stats;
};
struct BnBranch2 { BnBranch b[2]; };

__global__ __launch_bounds__(256) void bn_finalize_partials_kernel(BnBranch2 br, int64_t C, float n, float eps, float momentum) {
  // block = 16 channels x {sum, sumsq} = 32 columns x 8 partial lanes (the parallelism of reduce_partials_kernel)
  __shared__ double s_acc[8][33];
  const BnBranch& b = blockIdx.y ? br.b[1] : br.b[0];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int64_t c = (int64_t)blockIdx.x * 16 + (cl & 15);
  s_acc[pl][cl] = c < C ? partial_lane_sum(b.partials, b.n_part, 2 * C, (int64_t)(cl >> 4) * C + c, pl) : 0.0;
  __syncthreads();
  if (pl == 0 && cl < 16 && c < C) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += s_acc[i][cl]; q += s_acc[i][16 + cl]; }
    if (b.stats) { b.stats[c] = (float)s; b.stats[C + c] = (float)q; }
    bn_finalize_channel((float)s, (float)q, c, b.gamma, b.beta, b.rmean, b.rvar, b.coef, b.saved, C, n, eps, momentum, 1);
  }
}

// The same finalize with the SE squeeze of the unit's y in its grid (ContextSE units, round 4): the first nf workgroups are the
// finalize's, the rest seqsum_vec_kernel's - two launch-floor kernels of the forward chain (4.9 us each) that do not depend on each
// other become one.
template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_seqsum_kernel(BnBranch2 br, int64_t C, float n, float eps, float momentum, int nfx, int nf,
                                                                 const T* __restrict__ x, int Tt, int gx, float* __restrict__ sums) {
  constexpr int RL = 256 / (64 / Vec<T>::kN);
  __shared__ double s_acc[8][33];
  __shared__ float s_red[RL][65];
  if ((int)blockIdx.x < nf) {                              // workgroup-uniform
    const int fx = blockIdx.x % nfx, fy = blockIdx.x / nfx;
    const BnBranch& b = fy ? br.b[1] : br.b[0];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int64_t c = (int64_t)fx * 16 + (cl & 15);
    s_acc[pl][cl] = c < C ? partial_lane_sum(b.partials, b.n_part, 2 * C, (int64_t)(cl >> 4) * C + c, pl) : 0.0;
    __syncthreads();
    if (pl == 0 && cl < 16 && c < C) {
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) { s += s_acc[i][cl]; q += s_acc[i][16 + cl]; }
      if (b.stats) { b.stats[c] = (float)s; b.stats[C + c] = (float)q; }
      bn_finalize_channel((float)s, (float)q, c, b.gamma, b.beta, b.rmean, b.rvar, b.coef, b.saved, C, n, eps, momentum, 1);
    }
    return;
  }
  const int id = blockIdx.x - nf;
  seqsum_vec_body<T>(x, Tt, (int)C, sums, id % gx, id / gx, s_red);
}

int bn_finalize_partials_seqsum(const lasr_bn_branch* branches, int n_branches, int64_t C, int64_t n_rows, float eps, float momentum,
                                const void* x, int dtype, int64_t B, int64_t T_, float* sums, void* stream) {
  static const bool off = getenv("LASR_SE_SEQSUM_IN_FINALIZE") && atoi(getenv("LASR_SE_SEQSUM_IN_FINALIZE")) == 0;
  const int v = dtype == LASR_F32 ? 4 : 8;
  if (off || C % v != 0 || reinterpret_cast<uintptr_t>(x) % 16 != 0 || T_ >= ((int64_t)1 << 30) || C >= ((int64_t)1 << 30) || B >= 65536) return 1;
  LASR_CHECK_ARG(branches && n_branches >= 1 && n_branches <= 2 && C > 0 && n_rows > 0 && x && sums, "bn_finalize_partials_seqsum: bad argument");
  BnBranch2 br;
  for (int i = 0; i < 2; ++i) {
    const lasr_bn_branch& q = branches[i < n_branches ? i : 0];
    LASR_CHECK_ARG(q.partials && q.n_partials > 0 && q.gamma && q.beta && q.coef, "bn_finalize_partials_seqsum: null pointer");
    br.b[i] = {q.partials, q.n_partials, q.gamma, q.beta, q.running_mean, q.running_var, q.coef, q.saved, q.stats};
  }
  const int nfx = (int)cdiv(C, 16), nf = nfx * n_branches, gx = (int)cdiv(C, 64);
  const unsigned grid = (unsigned)(nf + gx * B);
  if (dtype == LASR_F32)
    hipLaunchKernelGGL(bn_finalize_seqsum_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), br, C, (float)n_rows, eps, momentum, nfx, nf,
                       (const float*)x, (int)T_, gx, sums);
  else
    hipLaunchKernelGGL(bn_finalize_seqsum_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), br, C, (float)n_rows, eps, momentum, nfx, nf,
                       (const bf16_t*)x, (int)T_, gx, sums);
  LASR_LAUNCH_CHECK("bn_finalize_seqsum_kernel");
  return 0;
}

int launch_reduce_partials(const float* partials, int n_part, int64_t ncols, float* out0, int64_t split, float* out1,
                           hipStream_t st) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(ncols, 32)), dim3(256), 0, st, partials, n_part, ncols, out0, split, out1);
  LASR_LAUNCH_CHECK("reduce_partials_kernel");
  return 0;
}

// ------------------------------------------------------------------ BN apply + add + act -----
// Thread = one 16-byte column vector (4 f32 / 8 bf16 channels) walking a slab of rows, so the
// per-channel coefficients are fetched once and every access is a full 16 B per lane.
static constexpr int kRowsPerBlock = 32;  // rows per workgroup, backward statistics

// d(act)/d(pre-activation)
__device__ __forceinline__ float act_grad(float p, int act) {
  if (act == LASR_ACT_RELU) return p > 0.f ? 1.f : 0.f;
  if (act == LASR_ACT_SWISH) {
    const float s = 1.f / (1.f + __expf(-p));
    return s * (1.f + p * (1.f - s));
  }
  return 1.f;
}

struct ColGeom { int cv, col_threads, row_lanes, cl, rl; };
template <int V>
__device__ __forceinline__ ColGeom col_geom(int C) {
  ColGeom g;
  g.cv = (int)(C / V);
  g.col_threads = g.cv < 256 ? g.cv : 256;
  g.row_lanes = 256 / g.col_threads;
  g.cl = threadIdx.x % g.col_threads;
  g.rl = threadIdx.x / g.col_threads;
  return g;
}

// Flat streaming form of the two element-wise BN kernels: a workgroup owns a slab of kSlabRows rows, keeps
// the per-channel constants of the whole row in LDS (copied once with 16-byte loads) and its 256 threads
// walk the slab's 16-byte vectors row-major, two vectors in flight per thread.  Registers stay small
// (no per-thread coefficient arrays), so 6-8 waves per SIMD keep enough loads in flight.
static constexpr int kSlabRows = 16;     // backward apply (40*C-byte table per slab)
static constexpr int kFwdSlabRows = 8;   // forward (16*C-byte table per slab)

__device__ __forceinline__ void lds_vec8(const float* p, float (&o)[8]) {
  const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
  o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
}
__device__ __forceinline__ void lds_vec8(const float* p, float (&o)[4]) {
  const float4 lo = *reinterpret_cast<const float4*>(p);
  o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w;
}

// LDS tables of per-channel constants, [q][C] f32.  A bf16 lane owns 8 consecutive channels = two float4: read
// straight from a linear table the lanes' addresses are 32 B apart and every 16-byte read is a 2-way bank
// conflict (measured: 42-47 % of the LDS cycles of these kernels).  So each table row is stored as two planes -
// the low float4 of every channel octet, then the high float4 - and both reads are unit-stride.  (f32 lanes own
// 4 channels: the linear layout is already unit-stride.)
template <int V>
__device__ __forceinline__ int tab_pos(int i, int C) {        // i: linear index (multiple of 4) into [q][C]
  if (V == 4) return i;
  const int q = i / C, r = i - q * C, j = r >> 2;
  return q * C + ((j & 1) ? (C >> 1) : 0) + ((j >> 1) << 2);
}
__device__ __forceinline__ void tab_vec(const float* row, int c, int C, float (&o)[8]) {   // row = table + q*C
  const float4 lo = *reinterpret_cast<const float4*>(row + (c >> 1));
  const float4 hi = *reinterpret_cast<const float4*>(row + (C >> 1) + (c >> 1));
  o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
}
__device__ __forceinline__ void tab_vec(const float* row, int c, int C, float (&o)[4]) {
  (void)C;
  const float4 lo = *reinterpret_cast<const float4*>(row + c);
  o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w;
}

// V consecutive f32 of a per-(utterance, channel) table (the SE scale / SE gradient) as 16-byte loads, issued together with the
// activation loads (eight scalar loads per vector in the middle of the arithmetic made the SE variants of these kernels 2-3x
// slower: 18.5 / 31.7 / 42.5 us against 8.5 / 14.1 / 15.4 for forward / statistics / apply)
__device__ __forceinline__ void ld_tab(const float* p, float (&o)[8]) { lds_vec8(p, o); }
__device__ __forceinline__ void ld_tab(const float* p, float (&o)[4]) { lds_vec8(p, o); }

// out = act((a*y + b)*se + a2*y2 + b2);  s_tab = [a | b | a2 | b2][C].  Two items per thread in flight, no branch
// around a load (HAS2 is a template parameter, the second item's index is clamped and only its store predicated).
template <typename T, bool HAS2, bool SE, bool DROP>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, const float* __restrict__ coef,
                                                         const T* __restrict__ y2, const float* __restrict__ coef2,
                                                         const float* __restrict__ se, T* __restrict__ out,
                                                         int rows, int Tt, int C, int act_rt, DropArgs drop) {
  extern __shared__ __attribute__((aligned(16))) float s_tab[];
  constexpr int V = Vec<T>::kN;
  const unsigned long long drop_step = DROP ? *drop.step : 0ull;
  for (int i = threadIdx.x * 4; i < 2 * C; i += 1024) {
    const int d = tab_pos<V>(i, C);
    *reinterpret_cast<float4*>(s_tab + d) = *reinterpret_cast<const float4*>(coef + i);
    *reinterpret_cast<float4*>(s_tab + 2 * C + d) = HAS2 ? *reinterpret_cast<const float4*>(coef2 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int cv = C / V;
  const int r0 = blockIdx.x * kFwdSlabRows;
  const int n_items = (min(r0 + kFwdSlabRows, rows) - r0) * cv;
  with_act(act_rt, [&](auto act_c) {
  constexpr int act = decltype(act_c)::value;
  for (int it0 = threadIdx.x; it0 < n_items; it0 += 512) {
    uint4 rv[2], rw[2];
    uint32_t off[2];
    int cc[2], rr[2];
    float sev[2][V];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = min(it0 + 256 * u, n_items - 1);
      const int rl = it / cv;
      cc[u] = (it - rl * cv) * V; rr[u] = r0 + rl;
      off[u] = (uint32_t)rr[u] * (uint32_t)C + (uint32_t)cc[u];
      rv[u] = Vec<T>::raw(y + off[u]);
      if (HAS2) rw[u] = Vec<T>::raw(y2 + off[u]);
      if (SE) ld_tab(se + (uint32_t)(rr[u] / Tt) * (uint32_t)C + (uint32_t)cc[u], sev[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = cc[u];
      float v[V], w[V], a[V], b[V], o[V];
      Vec<T>::unpack(rv[u], v);
      tab_vec(s_tab, c, C, a); tab_vec(s_tab + C, c, C, b);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        o[j] = fmaf(v[j], a[j], b[j]);
        if (SE) o[j] *= sev[u][j];
      }
      // nn.Dropout sits at the end of SeprationConv (models/QuartNet.py:38): on the main branch BEFORE the residual add of a
      // block, AFTER the activation of first_cnn / last_cnn2
      float dsc[V];
      if (DROP) drop_scale<V>(drop, drop_step, off[u], dsc);
      if (DROP && HAS2) {
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] *= dsc[j];
      }
      if (HAS2) {
        Vec<T>::unpack(rw[u], w);
        tab_vec(s_tab + 2 * C, c, C, a); tab_vec(s_tab + 3 * C, c, C, b);
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] += fmaf(w[j], a[j], b[j]);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = act_fwd(o[j], act);
      if (DROP && !HAS2) {
#pragma unroll
        for (int j = 0; j < V; ++j) o[j] *= dsc[j];
      }
      if (it0 + 256 * u < n_items) Vec<T>::store(out + off[u], o);
    }
  }
  });
}

// pass 1: per-block partial sums -> partials[blk][4][C]  (s1, s2 of branch 1; s1', s2' of branch 2).
// HAS2 is a template parameter and every load is unconditional (clamped row, contribution masked), so there is
// no branch around a load; a thread's rows are taken two at a time with all six 16-byte loads issued before the
// first is unpacked (four at a time, or the constants streamed from LDS, cost the second wave per SIMD:
// measured 34 us against 11).  The kernel was a chain of eight exposed memory round trips per slab.
template <typename T, bool HAS2, bool SE, bool DROP>
__global__ __launch_bounds__(256, 2) void bn_bwd_stats_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                              const float* __restrict__ coef, const float* __restrict__ saved,
                                                              const T* __restrict__ y2, const float* __restrict__ coef2,
                                                              const float* __restrict__ saved2, const float* __restrict__ se,
                                                              const float* __restrict__ seg, float* __restrict__ partials,
                                                              int rows, int Tt, int C, int act, DropArgs drop, int per_utt) {
  extern __shared__ __attribute__((aligned(16))) float s_part[];  // [row_lanes][4][C]
  constexpr int V = Vec<T>::kN;
  const unsigned long long drop_step = DROP ? *drop.step : 0ull;
  constexpr int RB = 2;                                           // rows in flight per thread
  const ColGeom g = col_geom<V>(C);
  // per_utt (fused SE + BN backward): slabs do not cross utterances - workgroup (b, slab) - and the sums are those of the gradient
  // at the BN output WITHOUT the SE scale / pooled-path term (per-utterance factors, applied when the slabs are folded)
  int r0 = blockIdx.x * kRowsPerBlock;
  int r1 = min(r0 + kRowsPerBlock, rows);
  if (per_utt) {
    const int nslab = (Tt + kRowsPerBlock - 1) / kRowsPerBlock;
    const int ub = blockIdx.x / nslab, sl = blockIdx.x - ub * nslab;
    r0 = ub * Tt + sl * kRowsPerBlock;
    r1 = min(r0 + kRowsPerBlock, (ub + 1) * Tt);
  }
  if (g.rl < g.row_lanes) {
    for (int cvi = g.cl; cvi < g.cv; cvi += g.col_threads) {
      const int c = cvi * V;
      float a1[V], b1[V], m1[V], q1[V], a2[V], b2[V], m2[V], q2[V];
      lds_vec8(coef + c, a1); lds_vec8(coef + C + c, b1); lds_vec8(saved + c, m1); lds_vec8(saved + C + c, q1);
      if (HAS2) {
        lds_vec8(coef2 + c, a2); lds_vec8(coef2 + C + c, b2); lds_vec8(saved2 + c, m2); lds_vec8(saved2 + C + c, q2);
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) a2[j] = b2[j] = m2[j] = q2[j] = 0.f;
      }
      float acc[4][V];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[k][j] = 0.f;
      for (int rb = r0 + g.rl; rb < r1; rb += RB * g.row_lanes) {
        uint4 rd[RB], ry[RB], rr[RB];
        float sev[RB][V], sgv[RB][V];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          const uint32_t rowc = (uint32_t)min(rb + i * g.row_lanes, rows - 1);
          const uint32_t off = rowc * (uint32_t)C + (uint32_t)c;
          rd[i] = Vec<T>::raw(dout + off);
          ry[i] = Vec<T>::raw(y + off);
          if (HAS2) rr[i] = Vec<T>::raw(y2 + off);
          if (SE) {
            const uint32_t so = (rowc / (uint32_t)Tt) * (uint32_t)C + (uint32_t)c;
            ld_tab(se + so, sev[i]);
            if (seg) ld_tab(seg + so, sgv[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          float dvi[V], yvi[V], rvi[V];
          Vec<T>::unpack(rd[i], dvi);
          Vec<T>::unpack(ry[i], yvi);
          if (HAS2) Vec<T>::unpack(rr[i], rvi);
          const int r = rb + i * g.row_lanes;
          const float live = r < r1 ? 1.f : 0.f;                  // rows past the slab contribute nothing
          float dsc[V];
          if (DROP) drop_scale<V>(drop, drop_step, (uint32_t)min(r, rows - 1) * (uint32_t)C + (uint32_t)c, dsc);
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const float sej = SE ? sev[i][j] : 1.f;
            const float zm = fmaf(yvi[j], a1[j], b1[j]) * sej * ((DROP && HAS2) ? dsc[j] : 1.f);
            const float z = zm + (HAS2 ? fmaf(rvi[j], a2[j], b2[j]) : 0.f);
            const float d = dvi[j] * act_grad(z, act) * live * ((DROP && !HAS2) ? dsc[j] : 1.f);   // gradient at the pre-activation
            const float dmn = (DROP && HAS2) ? d * dsc[j] : d;                                       // ... reaching the main branch
            const float d1 = per_utt ? dmn : fmaf(dmn, sej, (SE && seg) ? sgv[i][j] * live : 0.f);
            acc[0][j] += d1;
            acc[1][j] = fmaf(d1, (yvi[j] - m1[j]) * q1[j], acc[1][j]);
            if (HAS2) {
              acc[2][j] += d;
              acc[3][j] = fmaf(d, (rvi[j] - m2[j]) * q2[j], acc[3][j]);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) s_part[(g.rl * 4 + k) * C + c + j] = acc[k][j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C; i += 256) {
    float s = 0.f;
    for (int l = 0; l < g.row_lanes; ++l) s += s_part[l * 4 * C + i];
    partials[(size_t)blockIdx.x * 4 * C + i] = s;
  }
}

// (pass 2a's per-channel fold, bn_bwd_table_channel, lives in common.h: se.hip's excite-backward kernel ends in it too)
__global__ __launch_bounds__(256) void bn_bwd_table_kernel(const float* __restrict__ coef, const float* __restrict__ saved,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           const float* __restrict__ coef2, const float* __restrict__ saved2,
                                                           const float* __restrict__ gamma2, const float* __restrict__ sums2,
                                                           float inv_n, int C, float* __restrict__ tab, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma2,
                                                           float* __restrict__ dbeta2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const bool has2 = coef2 != nullptr;
  bn_bwd_table_channel(c, C, sums[c], sums[C + c], has2 ? sums2[c] : 0.f, has2 ? sums2[C + c] : 0.f, has2, coef, saved, gamma, coef2,
                       saved2, gamma2, inv_n, tab, dgamma, dbeta, dgamma2, dbeta2);
}

// the same table straight from the unreduced pass-1 partials ([blk][s1 | s2 | s1' | s2'][C]): four f64
// column sums per channel, then the fold (one launch instead of two)
__global__ __launch_bounds__(1024) void bn_bwd_table_partials_kernel(const float* __restrict__ partials, int n_part,
                                                                     const float* __restrict__ coef, const float* __restrict__ saved,
                                                                     const float* __restrict__ gamma, const float* __restrict__ coef2,
                                                                     const float* __restrict__ saved2, const float* __restrict__ gamma2,
                                                                     float inv_n, int C, float* __restrict__ tab,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                     float* __restrict__ dgamma2, float* __restrict__ dbeta2) {
  // block = 8 channels x 4 sums = 32 columns x 32 partial lanes: the 501 partial rows of a cfg2 unit are ONE batch of 16 loads per
  // thread (with 8 lanes they were four dependent trips to the fabric: 7.2 us per launch)
  __shared__ double s_acc[32][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 8 + (cl & 7), k = cl >> 3;
  const bool has2 = coef2 != nullptr;
  s_acc[pl][cl] = (c < C && (k < 2 || has2)) ? partial_lane_sum<32>(partials, n_part, 4 * (int64_t)C, (int64_t)k * C + c, pl) : 0.0;
  __syncthreads();
  if (pl < 4) {                                    // lanes 8*pl .. 8*pl+7 of every column, then the four quarter sums
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += s_acc[pl * 8 + i][cl];
    s_acc[pl * 8][cl] = t;
  }
  __syncthreads();
  if (pl == 0 && cl < 8 && c < C) {
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = (s_acc[0][q * 8 + cl] + s_acc[8][q * 8 + cl]) + (s_acc[16][q * 8 + cl] + s_acc[24][q * 8 + cl]);
    bn_bwd_table_channel(c, C, (float)t[0], (float)t[1], (float)t[2], (float)t[3], has2, coef, saved, gamma, coef2, saved2, gamma2,
                         inv_n, tab, dgamma, dbeta, dgamma2, dbeta2);
  }
}

// pass 2b: dy = G1*d1 + B1*y + C1 (rows past the utterance length zeroed), dy2 = G2*d + B2*y2 + C2.
// Two items per thread in flight, HAS2 a template parameter: no branch around a load.
template <typename T, bool HAS2, bool SE, bool DROP>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y,
                                                           const T* __restrict__ y2, const float* __restrict__ tab,
                                                           const float* __restrict__ se, const float* __restrict__ seg,
                                                           const int32_t* __restrict__ row_lens, T* __restrict__ dy,
                                                           T* __restrict__ dy2, int rows, int Tt, int C, int act, DropArgs drop) {
  extern __shared__ __attribute__((aligned(16))) float s_tab[];  // [10][C]
  constexpr int V = Vec<T>::kN;
  const unsigned long long drop_step = DROP ? *drop.step : 0ull;
  for (int i = threadIdx.x * 4; i < 10 * C; i += 1024) *reinterpret_cast<float4*>(s_tab + tab_pos<V>(i, C)) = *reinterpret_cast<const float4*>(tab + i);
  __syncthreads();
  const int cv = C / V;
  const int r0 = blockIdx.x * kSlabRows;
  const int n_items = (min(r0 + kSlabRows, rows) - r0) * cv;
  for (int it0 = threadIdx.x; it0 < n_items; it0 += 512) {
    uint4 rd[2], ry[2], rr2[2];
    uint32_t off[2];
    int cc[2], rr[2];
    float sev[2][V], sgv[2][V];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = min(it0 + 256 * u, n_items - 1);
      const int rl = it / cv;
      cc[u] = (it - rl * cv) * V; rr[u] = r0 + rl;
      off[u] = (uint32_t)rr[u] * (uint32_t)C + (uint32_t)cc[u];
      rd[u] = Vec<T>::raw(dout + off[u]);
      ry[u] = Vec<T>::raw(y + off[u]);
      if (HAS2) rr2[u] = Vec<T>::raw(y2 + off[u]);
      if (SE) {
        const uint32_t so = (uint32_t)(rr[u] / Tt) * (uint32_t)C + (uint32_t)cc[u];
        ld_tab(se + so, sev[u]);
        if (seg) ld_tab(seg + so, sgv[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = cc[u], r = rr[u];
      const bool live = it0 + 256 * u < n_items;
      float dv[V], yv[V], rv[V], o1[V], o2[V];
      Vec<T>::unpack(rd[u], dv);
      Vec<T>::unpack(ry[u], yv);
      if (HAS2) Vec<T>::unpack(rr2[u], rv);
      const int ub = r / Tt;
      const bool masked = row_lens && (r - ub * Tt) >= row_lens[ub];
      float a1[V], b1[V], G[V], Bc[V], Cc[V];
      tab_vec(s_tab, c, C, a1); tab_vec(s_tab + C, c, C, b1);
      float z[V];
#pragma unroll
      for (int j = 0; j < V; ++j) z[j] = fmaf(yv[j], a1[j], b1[j]) * (SE ? sev[u][j] : 1.f);
      float dsc[V];
      if (DROP) drop_scale<V>(drop, drop_step, off[u], dsc);
      if (DROP && HAS2) {
#pragma unroll
        for (int j = 0; j < V; ++j) z[j] *= dsc[j];
      }
      if (HAS2) {
        tab_vec(s_tab + 5 * C, c, C, a1); tab_vec(s_tab + 6 * C, c, C, b1);
#pragma unroll
        for (int j = 0; j < V; ++j) z[j] += fmaf(rv[j], a1[j], b1[j]);
      }
      float d[V];
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = dv[j] * act_grad(z[j], act) * ((DROP && !HAS2) ? dsc[j] : 1.f);
      tab_vec(s_tab + 2 * C, c, C, G); tab_vec(s_tab + 3 * C, c, C, Bc); tab_vec(s_tab + 4 * C, c, C, Cc);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d1 = fmaf((DROP && HAS2) ? d[j] * dsc[j] : d[j], SE ? sev[u][j] : 1.f, (SE && seg) ? sgv[u][j] : 0.f);
        o1[j] = masked ? 0.f : fmaf(G[j], d1, fmaf(Bc[j], yv[j], Cc[j]));
      }
      if (live) Vec<T>::store(dy + off[u], o1);
      if (HAS2) {
        tab_vec(s_tab + 7 * C, c, C, G); tab_vec(s_tab + 8 * C, c, C, Bc); tab_vec(s_tab + 9 * C, c, C, Cc);
#pragma unroll
        for (int j = 0; j < V; ++j) o2[j] = fmaf(G[j], d[j], fmaf(Bc[j], rv[j], Cc[j]));
        if (live) Vec<T>::store(dy2 + off[u], o2);
      }
    }
  }
}



// ---- channel-sliced BN backward (bf16, residual or plain units without SE / dropout) -----------------------------------------
// The row-major pair above needs a launch between its passes that folds 501 partial rows into per-channel constants, because a
// workgroup that walks whole rows needs the constants of ALL channels.  Here a workgroup owns a SLICE of 64 channels (128 bytes of
// every row: whole cache lines) and a chunk of rows, so the statistics pass leaves only `nchunk` partial rows per channel and
// every workgroup of the apply pass folds the constants of its own 64 channels in its prologue (nchunk x 4 x 64 floats from L2):
// statistics -> apply, nothing in between.  block = 8 column threads (16 bytes each) x 64 row lanes.
static constexpr int kSlCh = 64, kSlThreads = 512, kSlLanes = 64;

// SE (ContextSE units): a chunk is one utterance (rpc = T'), so the excite scale of the thread's 8 channels is a register constant
// and the partial rows ARE the per-utterance sums P[b][4][C] the excite backward starts from (sums of the gradient at the BN
// output without the SE factors, as bn_bwd_stats_kernel(per_utt)).
template <bool HAS2, bool SE, bool NTL = false>
__global__ __launch_bounds__(512) void bn_bwd_stats_sliced_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                                  const float* __restrict__ coef, const float* __restrict__ saved,
                                                                  const bf16_t* __restrict__ y2, const float* __restrict__ coef2,
                                                                  const float* __restrict__ saved2, const float* __restrict__ se,
                                                                  float* __restrict__ partials, int rows, int C, int act_rt, int rpc) {
  __shared__ float s_red[8][4][kSlCh];
  constexpr int V = 8, RB = 4;
  const int tid = threadIdx.x, cl = tid & 7, rl = tid >> 3, lane = tid & 63, wid = tid >> 6;
  const int c = blockIdx.x * kSlCh + cl * V;
  const int r0 = blockIdx.y * rpc, r1 = min(r0 + rpc, rows);
  float a1[V], b1[V], m1[V], q1[V], a2[V], b2[V], m2[V], q2[V];
  lds_vec8(coef + c, a1); lds_vec8(coef + C + c, b1); lds_vec8(saved + c, m1); lds_vec8(saved + C + c, q1);
  if (HAS2) {
    lds_vec8(coef2 + c, a2); lds_vec8(coef2 + C + c, b2); lds_vec8(saved2 + c, m2); lds_vec8(saved2 + C + c, q2);
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) a2[j] = b2[j] = m2[j] = q2[j] = 0.f;
  }
  float sev[V];
  if (SE) lds_vec8(se + (size_t)blockIdx.y * C + c, sev);
  float acc[4][V];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[k][j] = 0.f;
  // (act stays a runtime value here: as a constant the loop body is one basic block, the scheduler overlaps the four rows further and
  //  the 256-register budget spills - measured equal, 12.0 us either way: the pass waits on memory)
  const int act = act_rt;
  for (int rb = r0 + rl; rb < r1; rb += RB * kSlLanes) {
    uint4 rd[RB], ry[RB], rr[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const uint32_t off = (uint32_t)min(rb + i * kSlLanes, rows - 1) * (uint32_t)C + (uint32_t)c;
      rd[i] = Vec<bf16_t>::raw(dout + off);                     // (written a moment ago by the unit above: default policy)
      ry[i] = Vec<bf16_t>::raw_if_nt<NTL>(y + off);
      if (HAS2) rr[i] = Vec<bf16_t>::raw_if_nt<NTL>(y2 + off);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      float dvi[V], yvi[V], rvi[V];
      Vec<bf16_t>::unpack(rd[i], dvi);
      Vec<bf16_t>::unpack(ry[i], yvi);
      if (HAS2) Vec<bf16_t>::unpack(rr[i], rvi);
      const float live = rb + i * kSlLanes < r1 ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float z = fmaf(yvi[j], a1[j], b1[j]) * (SE ? sev[j] : 1.f) + (HAS2 ? fmaf(rvi[j], a2[j], b2[j]) : 0.f);
        const float d = dvi[j] * act_grad(z, act) * live;
        acc[0][j] += d;
        acc[1][j] = fmaf(d, (yvi[j] - m1[j]) * q1[j], acc[1][j]);
        if (HAS2) {
          acc[2][j] += d;
          acc[3][j] = fmaf(d, (rvi[j] - m2[j]) * q2[j], acc[3][j]);
        }
      }
    }
  }
  // the wave's 8 row lanes (lane bits 3-5), then the 8 waves through LDS in a fixed order
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float v = acc[k][j];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      acc[k][j] = v;
    }
  if (lane < 8) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) s_red[wid][k][cl * V + j] = acc[k][j];
  }
  __syncthreads();
  if (tid < 4 * kSlCh) {
    const int k = tid >> 6, ch = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += s_red[w][k][ch];
    partials[((size_t)blockIdx.y * 4 + k) * C + blockIdx.x * kSlCh + ch] = t;
  }
}

// SE: the constants come folded from the excite backward (tab_in [10][C], se_bwd_pool_kernel); d1 = d*se + seg with the
// utterance's scale and pooled-path gradient as register constants (a chunk is one utterance).
// RB: rows in flight per thread; WPE: waves per SIMD the register allocation must leave room for (2 = one 512-thread workgroup per
// CU, 4 = two).
template <bool HAS2, bool SE, int RB = 2, int WPE = 2, bool NTL = false>
__global__ __launch_bounds__(512, WPE) void bn_bwd_apply_sliced_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ y,
                                                                  const bf16_t* __restrict__ y2, const float* __restrict__ partials,
                                                                  int nchunk, const float* __restrict__ tab_in, const float* __restrict__ se,
                                                                  const float* __restrict__ seg, const float* __restrict__ coef,
                                                                  const float* __restrict__ saved,
                                                                  const float* __restrict__ gamma, const float* __restrict__ coef2,
                                                                  const float* __restrict__ saved2, const float* __restrict__ gamma2,
                                                                  float inv_n, const int32_t* __restrict__ row_lens, bf16_t* __restrict__ dy,
                                                                  bf16_t* __restrict__ dy2, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ dgamma2, float* __restrict__ dbeta2, int rows, int Tt,
                                                                  int C, int act_rt, int rpc) {
  __shared__ double s_sum[4][kSlCh];
  __shared__ __attribute__((aligned(16))) float s_tab[10][kSlCh];
  constexpr int V = 8;
  const int tid = threadIdx.x, cl = tid & 7, rl = tid >> 3;
  const int c0 = blockIdx.x * kSlCh;
  if (SE) {
    for (int i = tid; i < 10 * kSlCh; i += kSlThreads) s_tab[i >> 6][i & 63] = tab_in[(size_t)(i >> 6) * C + c0 + (i & 63)];
  } else {
    if (tid < 4 * kSlCh) {                               // waves 0-3: the slice's four sums over the chunks, f64, fixed order
      const int k = tid >> 6, ch = tid & 63;
      double a0 = 0.0, a1 = 0.0;
      if (k < 2 || HAS2) {
        const float* p = partials + (size_t)k * C + c0 + ch;
        for (int q = 0; q < nchunk; q += 16) {
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float x = p[(size_t)min(q + i, nchunk - 1) * 4 * C];
            v[i] = q + i < nchunk ? x : 0.f;
          }
#pragma unroll
          for (int i = 0; i < 16; i += 2) { a0 += (double)v[i]; a1 += (double)v[i + 1]; }
        }
      }
      s_sum[k][ch] = a0 + a1;
    }
    __syncthreads();
    if (tid < kSlCh) {                                   // the constants of pass 2a (bn_bwd_table_channel), kept in LDS
      const int c = c0 + tid;
      const float s1 = (float)s_sum[0][tid], s2 = (float)s_sum[1][tid];
      {
        const float q = saved[C + c], w = s2 * inv_n, G = gamma[c] * q;
        s_tab[0][tid] = coef[c]; s_tab[1][tid] = coef[C + c]; s_tab[2][tid] = G; s_tab[3][tid] = -G * q * w;
        s_tab[4][tid] = G * (saved[c] * q * w - s1 * inv_n);
        if (blockIdx.y == 0) { dbeta[c] = s1; dgamma[c] = s2; }
      }
      if (HAS2) {
        const float s1b = (float)s_sum[2][tid], s2b = (float)s_sum[3][tid];
        const float q = saved2[C + c], w = s2b * inv_n, G = gamma2[c] * q;
        s_tab[5][tid] = coef2[c]; s_tab[6][tid] = coef2[C + c]; s_tab[7][tid] = G; s_tab[8][tid] = -G * q * w;
        s_tab[9][tid] = G * (saved2[c] * q * w - s1b * inv_n);
        if (blockIdx.y == 0) { dbeta2[c] = s1b; dgamma2[c] = s2b; }
      } else {
#pragma unroll
        for (int k = 5; k < 10; ++k) s_tab[k][tid] = 0.f;
      }
    }
  }
  __syncthreads();
  const int c = c0 + cl * V;
  const int r0 = blockIdx.y * rpc, r1 = min(r0 + rpc, rows);
  float sev[V], sgv[V];
  if (SE) { lds_vec8(se + (size_t)blockIdx.y * C + c, sev); lds_vec8(seg + (size_t)blockIdx.y * C + c, sgv); }
  with_act(act_rt, [&](auto act_c) {
  constexpr int act = decltype(act_c)::value;
  for (int rb = r0 + rl; rb < r1; rb += RB * kSlLanes) {
    uint4 rd[RB], ry[RB], rr2[RB];
    uint32_t off[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      off[u] = (uint32_t)min(rb + u * kSlLanes, rows - 1) * (uint32_t)C + (uint32_t)c;
      rd[u] = Vec<bf16_t>::raw(dout + off[u]);
      ry[u] = Vec<bf16_t>::raw_if_nt<NTL>(y + off[u]);
      if (HAS2) rr2[u] = Vec<bf16_t>::raw_if_nt<NTL>(y2 + off[u]);
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int r = rb + u * kSlLanes;
      const bool live = r < r1;
      const int rc = min(r, rows - 1);
      float dv[V], yv[V], rv[V], o1[V], o2[V];
      Vec<bf16_t>::unpack(rd[u], dv);
      Vec<bf16_t>::unpack(ry[u], yv);
      if (HAS2) Vec<bf16_t>::unpack(rr2[u], rv);
      const int ub = rc / Tt;
      const bool masked = row_lens && (rc - ub * Tt) >= row_lens[ub];
      float ca[V], cb[V], G[V], Bc[V], Cc[V], z[V], d[V];
      lds_vec8(&s_tab[0][cl * V], ca); lds_vec8(&s_tab[1][cl * V], cb);
#pragma unroll
      for (int j = 0; j < V; ++j) z[j] = fmaf(yv[j], ca[j], cb[j]) * (SE ? sev[j] : 1.f);
      if (HAS2) {
        lds_vec8(&s_tab[5][cl * V], ca); lds_vec8(&s_tab[6][cl * V], cb);
#pragma unroll
        for (int j = 0; j < V; ++j) z[j] += fmaf(rv[j], ca[j], cb[j]);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) d[j] = dv[j] * act_grad(z[j], act);
      lds_vec8(&s_tab[2][cl * V], G); lds_vec8(&s_tab[3][cl * V], Bc); lds_vec8(&s_tab[4][cl * V], Cc);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d1 = SE ? fmaf(d[j], sev[j], sgv[j]) : d[j];
        o1[j] = masked ? 0.f : fmaf(G[j], d1, fmaf(Bc[j], yv[j], Cc[j]));
      }
      if (live) Vec<bf16_t>::store(dy + off[u], o1);
      if (HAS2) {
        lds_vec8(&s_tab[7][cl * V], G); lds_vec8(&s_tab[8][cl * V], Bc); lds_vec8(&s_tab[9][cl * V], Cc);
#pragma unroll
        for (int j = 0; j < V; ++j) o2[j] = fmaf(G[j], d[j], fmaf(Bc[j], rv[j], Cc[j]));
        if (live) Vec<bf16_t>::store(dy2 + off[u], o2);
      }
    }
  }
  });
}

// rows per chunk of the sliced pair for this shape (0: take the row-major kernels).  One 512-thread workgroup per CU (the
// statistics kernel holds 222 registers per lane): 256 workgroups measured best at cfg2 (128: 2.339, 192: 2.298, 256: 2.262,
// 384: 2.447, 512: 2.419 ms per step; row-major pair with its table launch: 2.372).  LASR_BN_SLICED=<n> sets the count, 0 disables.
static int bn_sliced_rpc(int dtype, int64_t rows, int64_t C, bool se, bool drop, int per_utt, bool reduced_sums) {
  static const int target = getenv("LASR_BN_SLICED") ? atoi(getenv("LASR_BN_SLICED")) : 256;
  if (target <= 0 || dtype != LASR_BF16 || se || drop || per_utt || reduced_sums || C % kSlCh != 0 || rows < 4096) return 0;
  const int64_t slices = C / kSlCh;
  // at least one row per row lane in a chunk: the partial rows ([nchunk][4][C]) then also fit the workspace the row-major pair
  // is sized for ([rows / 32][4][C])
  const int64_t nchunk = std::max<int64_t>(1, std::min<int64_t>(target / slices, rows / kSlLanes));
  return (int)cdiv(rows, nchunk);
}

}  // namespace lasr

using namespace lasr;

extern "C" int lasr_bn_act_bwd_stats_drop(const void* dout, const void* y, const float* coef, const float* saved, const void* y2,
                                          const float* coef2, const float* saved2, const float* se_scale, const float* se_grad,
                                          float* sums, float* sums2, int dtype, int64_t B, int64_t T_, int64_t C, int act,
                                          const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int lasr_bn_act_bwd_apply_drop(const void* dout, const void* y, const float* coef, const float* saved,
                                          const float* gamma, const void* y2, const float* coef2, const float* saved2,
                                          const float* gamma2, const float* se_scale, const float* se_grad, const float* sums,
                                          const float* sums2, const int32_t* row_lens, void* dy, void* dy2, float* dgamma,
                                          float* dbeta, float* dgamma2, float* dbeta2, int dtype, int64_t B, int64_t T_, int64_t C,
                                          int act, const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream);

static int bn_bwd_stats_impl(const void* dout, const void* y, const float* coef, const float* saved, const void* y2,
                             const float* coef2, const float* saved2, const float* se_scale, const float* se_grad,
                             float* sums, float* sums2, int dtype, int64_t B, int64_t T_, int64_t C, int act,
                             const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream, int per_utt);

#define DISPATCH_DTYPE(dtype, ...)                      \
  if ((dtype) == LASR_F32) { using T = float; __VA_ARGS__; } \
  else { using T = bf16_t; __VA_ARGS__; }

// public descriptor -> kernel arguments (p = 0 or a null descriptor: dropout off)
static DropArgs make_drop(const lasr_dropout* d) {
  DropArgs a;
  a.step = nullptr; a.seed = 0; a.unit = 0; a.thresh = 0; a.inv_keep = 1.f;
  if (d && d->step && d->p > 0.f) {
    a.step = reinterpret_cast<const unsigned long long*>(d->step);
    a.seed = d->seed; a.unit = d->unit;
    const float p = d->p < 0.999f ? d->p : 0.999f;
    a.thresh = (uint32_t)(p * 65536.f + 0.5f);
    a.inv_keep = 1.f / (1.f - (float)a.thresh / 65536.f);     // scale by the keep probability actually realised
  }
  return a;
}

extern "C" int lasr_bct_to_btc(const float* in, void* out, int dtype, int64_t B, int64_t C, int64_t T_, void* stream) {
  LASR_CHECK_ARG(in && out && (dtype == LASR_F32 || dtype == LASR_BF16), "lasr_bct_to_btc: bad argument");
  LASR_CHECK_SHAPE(B > 0 && C > 0 && T_ > 0 && B < 65536, "lasr_bct_to_btc: shape");
  dim3 grid((unsigned)cdiv(T_, 32), (unsigned)cdiv(C, 32), (unsigned)B);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_kernel<T, true>), grid, dim3(256), 0, as_stream(stream), in, out, C, T_));
  LASR_LAUNCH_CHECK("transpose_kernel");
  return 0;
}
extern "C" int lasr_btc_to_bct(const void* in, int dtype, float* out, int64_t B, int64_t C, int64_t T_, void* stream) {
  LASR_CHECK_ARG(in && out && (dtype == LASR_F32 || dtype == LASR_BF16), "lasr_btc_to_bct: bad argument");
  LASR_CHECK_SHAPE(B > 0 && C > 0 && T_ > 0 && B < 65536, "lasr_btc_to_bct: shape");
  dim3 grid((unsigned)cdiv(T_, 32), (unsigned)cdiv(C, 32), (unsigned)B);
  DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((transpose_kernel<T, false>), grid, dim3(256), 0, as_stream(stream), in, out, C, T_));
  LASR_LAUNCH_CHECK("transpose_kernel");
  return 0;
}

extern "C" int lasr_mask_lengths_step(const float* pct, int64_t B, int64_t T_, int32_t* lens, uint64_t* step_counter, void* stream) {
  LASR_CHECK_ARG(pct && lens && B > 0 && T_ > 0, "lasr_mask_lengths: bad argument");
  hipLaunchKernelGGL(mask_lengths_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, as_stream(stream), pct, B, (float)T_, lens,
                     reinterpret_cast<unsigned long long*>(step_counter));
  LASR_LAUNCH_CHECK("mask_lengths_kernel");
  return 0;
}
int lasr::mask_lengths_step_cast(const float* pct, int64_t B, int64_t T_, int32_t* lens, uint64_t* step_counter, const float* in, void* out,
                                 int64_t n, const DwTapJobs* jobs, void* stream) {
  LASR_CHECK_ARG(pct && lens && B > 0 && T_ > 0 && in && out && n > 0, "mask_lengths_step_cast: bad argument");
  if (reinterpret_cast<uintptr_t>(in) % 16 || reinterpret_cast<uintptr_t>(out) % 8) return 1;
  int64_t blocks = cdiv(n >> 2, 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  DwTapJobs j;
  memset(&j, 0, sizeof(j));
  int tap_blocks = 0;
  if (jobs) {
    j = *jobs;
    tap_blocks = j.blk0[j.n];
  }
  hipLaunchKernelGGL(mask_lengths_cast_kernel, dim3((unsigned)(blocks + tap_blocks + 1)), dim3(256), 0, as_stream(stream), pct, B, (float)T_, lens,
                     reinterpret_cast<unsigned long long*>(step_counter), in, reinterpret_cast<bf16_t*>(out), n, (int)blocks, j);
  LASR_LAUNCH_CHECK("mask_lengths_cast_kernel");
  return 0;
}

extern "C" int lasr_mask_lengths(const float* pct, int64_t B, int64_t T_, int32_t* lens, void* stream) {
  return lasr_mask_lengths_step(pct, B, T_, lens, nullptr, stream);
}

// keep[e] = 1 / 0 for the first n elements of the mask a kernel draws for `dropout` at the CURRENT value of its step counter
// (verification: the oracle applies the very same mask)
namespace lasr {
__global__ __launch_bounds__(256) void dropout_mask_kernel(DropArgs d, int64_t n, uint8_t* __restrict__ keep) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v * 8 >= n) return;
  const uint32_t m = d.step ? drop_keep8(d, *d.step, (uint32_t)v) : 0xffu;
  for (int i = 0; i < 8 && v * 8 + i < n; ++i) keep[v * 8 + i] = (m >> i) & 1u;
}
}  // namespace lasr
extern "C" int lasr_dropout_mask(const lasr_dropout* dropout, int64_t n, uint8_t* keep, void* stream) {
  LASR_CHECK_ARG(dropout && keep && n > 0 && n < ((int64_t)1 << 34), "lasr_dropout_mask: bad argument");
  hipLaunchKernelGGL(lasr::dropout_mask_kernel, dim3((unsigned)cdiv(cdiv(n, 8), 256)), dim3(256), 0, as_stream(stream), make_drop(dropout), n, keep);
  LASR_LAUNCH_CHECK("dropout_mask_kernel");
  return 0;
}

extern "C" int lasr_bn_finalize(const float* stats, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float* coef, float* saved, int64_t C, int64_t n_rows, float eps,
                                float momentum, int training, void* stream) {
  LASR_CHECK_ARG(gamma && beta && coef && C > 0 && n_rows > 0, "lasr_bn_finalize: bad argument");
  LASR_CHECK_ARG(training ? stats != nullptr : (running_mean && running_var), "lasr_bn_finalize: missing statistics");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, as_stream(stream), stats, gamma, beta,
                     running_mean, running_var, coef, saved, C, (float)n_rows, eps, momentum, training);
  LASR_LAUNCH_CHECK("bn_finalize_kernel");
  return 0;
}

extern "C" int lasr_bn_finalize_partials(const lasr_bn_branch* branches, int n_branches, int64_t C, int64_t n_rows, float eps,
                                         float momentum, void* stream) {
  LASR_CHECK_ARG(branches && (n_branches == 1 || n_branches == 2) && C > 0 && n_rows > 0, "lasr_bn_finalize_partials: bad argument");
  BnBranch2 br;
  for (int i = 0; i < 2; ++i) {
    const lasr_bn_branch& q = branches[i < n_branches ? i : 0];
    LASR_CHECK_ARG(q.partials && q.n_partials > 0 && q.gamma && q.beta && q.coef, "lasr_bn_finalize_partials: null pointer");
    br.b[i] = {q.partials, q.n_partials, q.gamma, q.beta, q.running_mean, q.running_var, q.coef, q.saved, q.stats};
  }
  hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3((unsigned)cdiv(C, 16), (unsigned)n_branches), dim3(256), 0, as_stream(stream), br, C,
                     (float)n_rows, eps, momentum);
  LASR_LAUNCH_CHECK("bn_finalize_partials_kernel");
  return 0;
}

static int check_bn_shape(const char* who, int dtype, int64_t B, int64_t T_, int64_t C) {
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "%s: bad dtype", who);
  const int v = dtype == LASR_F32 ? 4 : 8;
  LASR_CHECK_SHAPE(C % v == 0 && C <= 4096 && B > 0 && T_ > 0, "%s: C=%lld must be a multiple of %d (16-byte channel vectors)", who,
                   (long long)C, v);
  LASR_CHECK_SHAPE(B * T_ * C < ((int64_t)1 << 31), "%s: tensor exceeds the kernels' 32-bit element offsets", who);
  return 0;
}

extern "C" int lasr_bn_act_fwd_drop(const void* y, const float* coef, const void* y2, const float* coef2, const float* se_scale,
                                    void* out, int dtype, int64_t B, int64_t T_, int64_t C, int act, const lasr_dropout* dropout,
                                    void* stream) {
  LASR_CHECK_ARG(y && coef && out && (!y2 || coef2), "lasr_bn_act_fwd: null pointer");
  LASR_TRY(check_bn_shape("lasr_bn_act_fwd", dtype, B, T_, C));
  const DropArgs da = make_drop(dropout);
  const int64_t rows = B * T_;
  const size_t shmem = (size_t)4 * C * sizeof(float);
  if (y2) {
    if (se_scale) { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_act_fwd_kernel<T, true, true, true>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_act_fwd_kernel<T, true, true, false>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } }); } else { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_act_fwd_kernel<T, true, false, true>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_act_fwd_kernel<T, true, false, false>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } }); }
  } else {
    if (se_scale) { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_act_fwd_kernel<T, false, true, true>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_act_fwd_kernel<T, false, true, false>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } }); } else { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_act_fwd_kernel<T, false, false, true>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_act_fwd_kernel<T, false, false, false>), dim3((unsigned)cdiv(rows, kFwdSlabRows)), dim3(256), shmem,
                                             as_stream(stream), (const T*)y, coef, (const T*)y2, coef2, se_scale, (T*)out, (int)rows,
                                             (int)T_, (int)C, act, da); } }); }
  }
  LASR_LAUNCH_CHECK("bn_act_fwd_kernel");
  return 0;
}

extern "C" int lasr_bn_act_fwd(const void* y, const float* coef, const void* y2, const float* coef2, const float* se_scale,
                               void* out, int dtype, int64_t B, int64_t T_, int64_t C, int act, void* stream) {
  return lasr_bn_act_fwd_drop(y, coef, y2, coef2, se_scale, out, dtype, B, T_, C, act, nullptr, stream);
}

extern "C" size_t lasr_bn_bwd_workspace_bytes(int64_t B, int64_t T_, int64_t C) {
  // pass-1 partials [blk][4][C], then the folded constants [10][C] of pass 2
  return (size_t)cdiv(B * T_, kRowsPerBlock) * 4 * C * sizeof(float) + (size_t)10 * C * sizeof(float);
}

extern "C" int lasr_bn_act_bwd_stats(const void* dout, const void* y, const float* coef, const float* saved, const void* y2,
                                     const float* coef2, const float* saved2, const float* se_scale, const float* se_grad,
                                     float* sums, float* sums2, int dtype, int64_t B, int64_t T_, int64_t C, int act,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  return lasr_bn_act_bwd_stats_drop(dout, y, coef, saved, y2, coef2, saved2, se_scale, se_grad, sums, sums2, dtype, B, T_, C, act, nullptr,
                                    workspace, workspace_bytes, stream);
}

extern "C" int lasr_bn_act_bwd_stats_drop(const void* dout, const void* y, const float* coef, const float* saved, const void* y2,
                                          const float* coef2, const float* saved2, const float* se_scale, const float* se_grad,
                                          float* sums, float* sums2, int dtype, int64_t B, int64_t T_, int64_t C, int act,
                                          const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream) {
  return bn_bwd_stats_impl(dout, y, coef, saved, y2, coef2, saved2, se_scale, se_grad, sums, sums2, dtype, B, T_, C, act, dropout, workspace,
                           workspace_bytes, stream, 0);
}

// per_utt = 1: partials [B * ceil(T/32)][4][C] of the raw sums, one workgroup per (utterance, 32-frame slab) (see the kernel)
static int bn_bwd_stats_impl(const void* dout, const void* y, const float* coef, const float* saved, const void* y2,
                             const float* coef2, const float* saved2, const float* se_scale, const float* se_grad,
                             float* sums, float* sums2, int dtype, int64_t B, int64_t T_, int64_t C, int act,
                             const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream, int per_utt) {
  const DropArgs da = make_drop(dropout);
  LASR_CHECK_ARG(dout && y && coef && saved && workspace, "lasr_bn_act_bwd_stats: null pointer");
  LASR_CHECK_ARG(!y2 || (coef2 && saved2 && (sums2 || !sums)), "lasr_bn_act_bwd_stats: branch-2 pointers");
  LASR_TRY(check_bn_shape("lasr_bn_act_bwd_stats", dtype, B, T_, C));
  const int64_t rows = B * T_;
  const int nblk = per_utt ? (int)(B * cdiv(T_, kRowsPerBlock)) : (int)cdiv(rows, kRowsPerBlock);
  if (workspace_bytes < (size_t)nblk * 4 * C * sizeof(float)) return fail(LASR_E_WORKSPACE, "lasr_bn_act_bwd_stats: workspace");
  if (const int rpc = bn_sliced_rpc(dtype, rows, C, se_scale != nullptr, da.step != nullptr, per_utt, sums != nullptr)) {
    const dim3 grid((unsigned)(C / kSlCh), (unsigned)cdiv(rows, rpc));
    float* partials = reinterpret_cast<float*>(workspace);
#define LASR_STATS_SL(H2_, NT_)                                                                                                         \
  hipLaunchKernelGGL((bn_bwd_stats_sliced_kernel<H2_, false, NT_>), grid, dim3(kSlThreads), 0, as_stream(stream), (const bf16_t*)dout, \
                     (const bf16_t*)y, coef, saved, (const bf16_t*)y2, coef2, saved2, nullptr, partials, (int)rows, (int)C, act, rpc)
    const bool ntl = (nt_loads_mask() & 1) != 0;
    if (y2) { if (ntl) LASR_STATS_SL(true, true); else LASR_STATS_SL(true, false); }
    else { if (ntl) LASR_STATS_SL(false, true); else LASR_STATS_SL(false, false); }
#undef LASR_STATS_SL
    LASR_LAUNCH_CHECK("bn_bwd_stats_sliced_kernel");
    return 0;
  }
  const int cv = (int)(C / (dtype == LASR_F32 ? 4 : 8));
  const int col_threads = cv < 256 ? cv : 256;
  const int row_lanes = 256 / col_threads;
  const size_t shmem = (size_t)row_lanes * 4 * C * sizeof(float);
  LASR_CHECK_SHAPE(shmem <= 64 * 1024, "lasr_bn_act_bwd_stats: C too large for LDS staging");
  float* partials = reinterpret_cast<float*>(workspace);
  if (y2) {
    if (se_scale) { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, true, true, true>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } else { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, true, true, false>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } }); } else { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, true, false, true>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } else { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, true, false, false>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } }); }
  } else {
    if (se_scale) { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, false, true, true>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } else { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, false, true, false>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } }); } else { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, false, false, true>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } else { hipLaunchKernelGGL((bn_bwd_stats_kernel<T, false, false, false>), dim3(nblk), dim3(256), shmem, as_stream(stream),
                                             (const T*)dout, (const T*)y, coef, saved, (const T*)y2, coef2, saved2, se_scale, se_grad,
                                             partials, (int)rows, (int)T_, (int)C, act, da, per_utt); } }); }
  }
  LASR_LAUNCH_CHECK("bn_bwd_stats_kernel");
  if (!sums) return 0;   // partials stay in the workspace for lasr_bn_act_bwd_apply(sums = NULL)
  return launch_reduce_partials(partials, nblk, 4 * C, sums, 2 * C, sums2, as_stream(stream));
}

extern "C" size_t lasr_bn_bwd_apply_workspace_bytes(int64_t C) { return (size_t)10 * C * sizeof(float); }

extern "C" int lasr_bn_act_bwd_apply(const void* dout, const void* y, const float* coef, const float* saved,
                                     const float* gamma, const void* y2, const float* coef2, const float* saved2,
                                     const float* gamma2, const float* se_scale, const float* se_grad, const float* sums,
                                     const float* sums2, const int32_t* row_lens, void* dy, void* dy2, float* dgamma,
                                     float* dbeta, float* dgamma2, float* dbeta2, int dtype, int64_t B, int64_t T_, int64_t C,
                                     int act, void* workspace, size_t workspace_bytes, void* stream) {
  return lasr_bn_act_bwd_apply_drop(dout, y, coef, saved, gamma, y2, coef2, saved2, gamma2, se_scale, se_grad, sums, sums2, row_lens, dy, dy2,
                                    dgamma, dbeta, dgamma2, dbeta2, dtype, B, T_, C, act, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int lasr_bn_act_bwd_apply_drop(const void* dout, const void* y, const float* coef, const float* saved,
                                          const float* gamma, const void* y2, const float* coef2, const float* saved2,
                                          const float* gamma2, const float* se_scale, const float* se_grad, const float* sums,
                                          const float* sums2, const int32_t* row_lens, void* dy, void* dy2, float* dgamma,
                                          float* dbeta, float* dgamma2, float* dbeta2, int dtype, int64_t B, int64_t T_, int64_t C,
                                          int act, const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream) {
  const DropArgs da = make_drop(dropout);
  LASR_CHECK_ARG(dout && y && coef && saved && gamma && dy && workspace, "lasr_bn_act_bwd_apply: null pointer");
  LASR_CHECK_ARG(!y2 || (coef2 && saved2 && gamma2 && (sums2 || !sums) && dy2), "lasr_bn_act_bwd_apply: branch-2 pointers");
  LASR_TRY(check_bn_shape("lasr_bn_act_bwd_apply", dtype, B, T_, C));
  const int64_t rows = B * T_;
  hipStream_t st = as_stream(stream);
  if (const int rpc = bn_sliced_rpc(dtype, rows, C, se_scale != nullptr, da.step != nullptr, 0, sums != nullptr)) {
    if (workspace_bytes < lasr_bn_bwd_workspace_bytes(B, T_, C)) return fail(LASR_E_WORKSPACE, "lasr_bn_act_bwd_apply: workspace");
    LASR_CHECK_ARG(dgamma && dbeta && (!y2 || (dgamma2 && dbeta2)), "lasr_bn_act_bwd_apply: parameter-gradient pointers");
    const int nchunk = (int)cdiv(rows, rpc);          // partial rows of the statistics pass
    const float* partials = reinterpret_cast<const float*>(workspace);
    // The apply pass takes row chunks HALF as long as the statistics pass (LASR_BN_APPLY_SPLIT = 2; 1: the same chunks): 512
    // workgroups with one row in flight per thread in 121 registers, so two are resident per CU and their load, arithmetic and
    // store phases interleave - against one round of 256 workgroups that are all in the same phase at the same time.  Measured in
    // the cfg2 step, one call: split 1 / 2 / 3 / 4 = 2.228 / 2.191 / 2.220 / 2.235 ms.  (The statistics pass does not gain from the
    // same treatment: 2.219 with both split against 2.198.)  The arithmetic per element is unchanged.
    static const int asplit = getenv("LASR_BN_APPLY_SPLIT") ? atoi(getenv("LASR_BN_APPLY_SPLIT")) : 2;
    const bool ntl = (nt_loads_mask() & 2) != 0;
#define LASR_APPLY_SL(H2_, RB_, W_, RPC_, G_) do { if (ntl) LASR_APPLY_SL2(H2_, RB_, W_, RPC_, G_, true); else LASR_APPLY_SL2(H2_, RB_, W_, RPC_, G_, false); } while (0)
#define LASR_APPLY_SL2(H2_, RB_, W_, RPC_, G_, NT_)                                                                                    \
  hipLaunchKernelGGL((bn_bwd_apply_sliced_kernel<H2_, false, RB_, W_, NT_>), G_, dim3(kSlThreads), 0, st, (const bf16_t*)dout, (const bf16_t*)y, \
                     (const bf16_t*)y2, partials, nchunk, nullptr, nullptr, nullptr, coef, saved, gamma, coef2, saved2, gamma2,          \
                     1.0f / (float)rows, row_lens, (bf16_t*)dy, (bf16_t*)dy2, dgamma, dbeta, dgamma2, dbeta2, (int)rows, (int)T_, (int)C, \
                     act, RPC_)
    const int rpc2 = asplit > 1 ? (int)cdiv(rows, (int64_t)nchunk * asplit) : rpc;
    if (asplit > 1 && rpc2 >= kSlLanes) {
      const dim3 grid2((unsigned)(C / kSlCh), (unsigned)cdiv(rows, rpc2));
      if (y2) LASR_APPLY_SL(true, 1, 4, rpc2, grid2); else LASR_APPLY_SL(false, 1, 4, rpc2, grid2);
    } else {
      const dim3 grid((unsigned)(C / kSlCh), (unsigned)nchunk);
      if (y2) LASR_APPLY_SL(true, 2, 2, rpc, grid); else LASR_APPLY_SL(false, 2, 2, rpc, grid);
    }
#undef LASR_APPLY_SL
#undef LASR_APPLY_SL2
    LASR_LAUNCH_CHECK("bn_bwd_apply_sliced_kernel");
    return 0;
  }
  float* tab;
  if (sums) {
    if (workspace_bytes < lasr_bn_bwd_apply_workspace_bytes(C)) return fail(LASR_E_WORKSPACE, "lasr_bn_act_bwd_apply: workspace");
    tab = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(bn_bwd_table_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, st, coef, saved, gamma, sums, y2 ? coef2 : nullptr,
                       saved2, gamma2, sums2, 1.0f / (float)rows, (int)C, tab, dgamma, dbeta, dgamma2, dbeta2);
    LASR_LAUNCH_CHECK("bn_bwd_table_kernel");
  } else {
    // the workspace of the preceding lasr_bn_act_bwd_stats(sums = NULL): unreduced partials, then room for the table
    if (workspace_bytes < lasr_bn_bwd_workspace_bytes(B, T_, C)) return fail(LASR_E_WORKSPACE, "lasr_bn_act_bwd_apply: workspace");
    const int nblk = (int)cdiv(rows, kRowsPerBlock);
    const float* partials = reinterpret_cast<const float*>(workspace);
    tab = reinterpret_cast<float*>(workspace) + (size_t)nblk * 4 * C;
    hipLaunchKernelGGL(bn_bwd_table_partials_kernel, dim3((unsigned)cdiv(C, 8)), dim3(1024), 0, st, partials, nblk, coef, saved, gamma,
                       y2 ? coef2 : nullptr, saved2, gamma2, 1.0f / (float)rows, (int)C, tab, dgamma, dbeta, dgamma2, dbeta2);
    LASR_LAUNCH_CHECK("bn_bwd_table_partials_kernel");
  }
  const size_t shmem = (size_t)10 * C * sizeof(float);
  LASR_CHECK_SHAPE(shmem <= 64 * 1024, "lasr_bn_act_bwd_apply: C too large for the LDS coefficient table");
  if (y2) {
    if (se_scale) { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, true, true>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, true, false>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } }); } else { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, false, true>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, false, false>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } }); }
  } else {
    if (se_scale) { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, true, true>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, true, false>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } }); } else { DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, false, true>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } else { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, false, false>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, se_grad, row_lens, (T*)dy,
                                             (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } }); }
  }
  LASR_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return 0;
}

extern "C" size_t lasr_bn_se_bwd_workspace_bytes(int64_t B, int64_t T_, int64_t C) {
  // raw partials [B*ceil(T/32)][4][C] | folded table [10][C] | the excite backward's hand-over (se_bwd_work_bytes)
  return align_up((size_t)B * cdiv(T_, kRowsPerBlock) * 4 * C * sizeof(float), 256) + align_up((size_t)10 * C * sizeof(float), 256) +
         se_bwd_work_bytes(B, C);
}

// Backward of a ContextSE unit's  out = act(BN(y) * se + BN_res(y2))  (models/QuartNetContextSE.py:19-23,54-57) in TWO passes over
// (dout, y, y2) instead of three: per-utterance raw sums -> ds (SE scale gradient) by algebra -> excite-MLP backward (seg, dW1,
// dW2) -> BN-backward constants -> apply.  ysum [B][C] = sum_t y of the forward (lasr_seqsum).  seg_out [B][C] is also returned.
extern "C" int lasr_bn_se_bwd(const void* dout, const void* y, const float* coef, const float* saved, const float* gamma, const float* beta,
                              const void* y2, const float* coef2, const float* saved2, const float* gamma2, const float* se_scale,
                              const float* se_hidden, const float* se_pooled, const float* ysum, const float* W1, const float* W2,
                              const int32_t* row_lens, void* dy, void* dy2, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2,
                              float* dW1, float* dW2, float* seg_out, int dtype, int64_t B, int64_t T_, int64_t C, int act,
                              const lasr_dropout* dropout, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(dout && y && coef && saved && gamma && beta && se_scale && se_hidden && se_pooled && ysum && W1 && W2 && dy && dgamma &&
                     dbeta && dW1 && dW2 && seg_out && workspace, "lasr_bn_se_bwd: null pointer");
  LASR_CHECK_ARG(!y2 || (coef2 && saved2 && gamma2 && dy2 && dgamma2 && dbeta2), "lasr_bn_se_bwd: branch-2 pointers");
  LASR_TRY(check_bn_shape("lasr_bn_se_bwd", dtype, B, T_, C));
  if (workspace_bytes < lasr_bn_se_bwd_workspace_bytes(B, T_, C)) return fail(LASR_E_WORKSPACE, "lasr_bn_se_bwd: workspace");
  hipStream_t st = as_stream(stream);
  const int nslab = (int)cdiv(T_, kRowsPerBlock);
  char* w = reinterpret_cast<char*>(workspace);
  float* partials = reinterpret_cast<float*>(w);
  w += align_up((size_t)B * nslab * 4 * C * sizeof(float), 256);
  float* tab = reinterpret_cast<float*>(w);
  w += align_up((size_t)10 * C * sizeof(float), 256);
  void* se_work = w;
  // channel-sliced form: one workgroup per (64 channels, utterance) in both passes - the statistics pass then leaves the
  // per-utterance sums themselves (nslab = 1: nothing for the excite backward to fold)
  static const bool se_sliced_off = getenv("LASR_BN_SLICED") && atoi(getenv("LASR_BN_SLICED")) <= 0;
  const bool sliced = !se_sliced_off && dtype == LASR_BF16 && !(dropout && dropout->step && dropout->p > 0.f) && C % kSlCh == 0 &&
                      B * (C / kSlCh) >= 128 && T_ >= 64;
  if (sliced) {
    const dim3 grid((unsigned)(C / kSlCh), (unsigned)B);
    if (y2) hipLaunchKernelGGL((bn_bwd_stats_sliced_kernel<true, true>), grid, dim3(kSlThreads), 0, st, (const bf16_t*)dout, (const bf16_t*)y, coef,
                               saved, (const bf16_t*)y2, coef2, saved2, se_scale, partials, (int)(B * T_), (int)C, act, (int)T_);
    else hipLaunchKernelGGL((bn_bwd_stats_sliced_kernel<false, true>), grid, dim3(kSlThreads), 0, st, (const bf16_t*)dout, (const bf16_t*)y, coef,
                            saved, (const bf16_t*)y2, coef2, saved2, se_scale, partials, (int)(B * T_), (int)C, act, (int)T_);
    LASR_LAUNCH_CHECK("bn_bwd_stats_sliced_kernel");
    SeBwdBn bn;
    bn.partials = partials; bn.nslab = 1; bn.gamma = gamma; bn.beta = beta; bn.ysum = ysum;
    bn.coef = coef; bn.saved = saved; bn.coef2 = y2 ? coef2 : nullptr; bn.saved2 = saved2; bn.gamma2 = gamma2;
    bn.inv_n = 1.0f / (float)(B * T_); bn.tab = tab; bn.dgamma = dgamma; bn.dbeta = dbeta; bn.dgamma2 = dgamma2; bn.dbeta2 = dbeta2;
    LASR_TRY(launch_se_bwd(nullptr, &bn, se_scale, se_hidden, se_pooled, W1, W2, B, T_, C, seg_out, dW1, dW2, se_work, st));
    // (the apply pass stays on whole-utterance chunks here: the half-utterance form of lasr_bn_act_bwd_apply_drop needs the SE
    //  constants in 128 registers, spills 9 of them and measured 3.80 against 3.78 ms per cfg4 step)
    if (y2) hipLaunchKernelGGL((bn_bwd_apply_sliced_kernel<true, true>), grid, dim3(kSlThreads), 0, st, (const bf16_t*)dout, (const bf16_t*)y,
                               (const bf16_t*)y2, nullptr, 0, tab, se_scale, seg_out, coef, saved, gamma, coef2, saved2, gamma2, 0.f, row_lens,
                               (bf16_t*)dy, (bf16_t*)dy2, dgamma, dbeta, dgamma2, dbeta2, (int)(B * T_), (int)T_, (int)C, act, (int)T_);
    else hipLaunchKernelGGL((bn_bwd_apply_sliced_kernel<false, true>), grid, dim3(kSlThreads), 0, st, (const bf16_t*)dout, (const bf16_t*)y,
                            (const bf16_t*)y2, nullptr, 0, tab, se_scale, seg_out, coef, saved, gamma, coef2, saved2, gamma2, 0.f, row_lens,
                            (bf16_t*)dy, (bf16_t*)dy2, dgamma, dbeta, dgamma2, dbeta2, (int)(B * T_), (int)T_, (int)C, act, (int)T_);
    LASR_LAUNCH_CHECK("bn_bwd_apply_sliced_kernel");
    return 0;
  }
  // pass 1: raw per-(utterance, slab) sums (the SE scale enters the pre-activation z only)
  LASR_TRY(bn_bwd_stats_impl(dout, y, coef, saved, y2, coef2, saved2, se_scale, nullptr, nullptr, nullptr, dtype, B, T_, C, act, dropout,
                             partials, (size_t)B * nslab * 4 * C * sizeof(float), stream, 1));
  // fold -> ds -> excite-MLP backward (seg, dW1, dW2) -> BN-backward constants: two launches (se.hip)
  SeBwdBn bn;
  bn.partials = partials; bn.nslab = nslab; bn.gamma = gamma; bn.beta = beta; bn.ysum = ysum;
  bn.coef = coef; bn.saved = saved; bn.coef2 = y2 ? coef2 : nullptr; bn.saved2 = saved2; bn.gamma2 = gamma2;
  bn.inv_n = 1.0f / (float)(B * T_); bn.tab = tab; bn.dgamma = dgamma; bn.dbeta = dbeta; bn.dgamma2 = dgamma2; bn.dbeta2 = dbeta2;
  LASR_TRY(launch_se_bwd(nullptr, &bn, se_scale, se_hidden, se_pooled, W1, W2, B, T_, C, seg_out, dW1, dW2, se_work, st));
  // pass 2: dy = G*(d*se + seg) + Bc*y + Cc, dy2 = G2*d + ...
  const DropArgs da = make_drop(dropout);
  const int64_t rows = B * T_;
  const size_t shmem = (size_t)10 * C * sizeof(float);
  LASR_CHECK_SHAPE(shmem <= 64 * 1024, "lasr_bn_se_bwd: C too large for the LDS coefficient table");
  if (y2) {
    DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, true, true>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, seg_out, row_lens, (T*)dy, (T*)dy2, (int)rows, (int)T_, (int)C, act, da); }
                            else { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, true, false>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, seg_out, row_lens, (T*)dy, (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } });
  } else {
    DISPATCH_DTYPE(dtype, { if (da.step) { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, true, true>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, seg_out, row_lens, (T*)dy, (T*)dy2, (int)rows, (int)T_, (int)C, act, da); }
                            else { hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, true, false>), dim3((unsigned)cdiv(rows, kSlabRows)), dim3(256), shmem, st,
                                             (const T*)dout, (const T*)y, (const T*)y2, tab, se_scale, seg_out, row_lens, (T*)dy, (T*)dy2, (int)rows, (int)T_, (int)C, act, da); } });
  }
  LASR_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return 0;
}

extern "C" int lasr_bn_eval_coef_many(const lasr_bn_eval_desc* descs, int n_descs, float eps, void* stream) {
  LASR_CHECK_ARG(descs && n_descs >= 1 && n_descs <= 64, "lasr_bn_eval_coef_many: 1..64 layers");
  lasr::BnEvalMany a;
  int64_t cmax = 0;
  for (int i = 0; i < n_descs; ++i) {
    LASR_CHECK_ARG(descs[i].gamma && descs[i].beta && descs[i].running_mean && descs[i].running_var && descs[i].coef && descs[i].C > 0,
                   "lasr_bn_eval_coef_many: bad layer");
    a.d[i] = descs[i];
    cmax = std::max<int64_t>(cmax, descs[i].C);
  }
  hipLaunchKernelGGL(lasr::bn_eval_coef_many_kernel, dim3((unsigned)cdiv(cmax, 256), (unsigned)n_descs), dim3(256), 0, as_stream(stream), a, eps);
  LASR_LAUNCH_CHECK("bn_eval_coef_many_kernel");
  return 0;
}

// ------------------------------------------------------------------ small reductions ----------
namespace lasr {
static constexpr int kColsumRows = 256;
// rows per workgroup: a narrow matrix (the 160-column LSTM gate gradients, the 28-class logit gradient) has only row slabs to make
// workgroups of - 32-row slabs give 501 workgroups at N = 16 032 instead of 63 (24 -> ~6 us for [16 032][160])
static inline int colsum_rows(int64_t C) { return C >= 1024 ? kColsumRows : 32; }
// partials[blk][c] = sum over a 256-row slab.  Narrow matrices (the 28-class logit gradient) would leave most
// of a column-per-thread block idle, so the 256 threads are dealt as col_threads x row_lanes: a lane sums
// every row_lanes-th row of the slab, the lanes are combined through LDS in a fixed order.
// grid (row slabs, column chunks of col_threads): a wide matrix (the 4334-class logit gradient: 444 MB) gets one workgroup per
// (slab, chunk) - 1717 of them - and every thread keeps 8 row loads in flight; with the column loop inside the workgroup
// (101 workgroups, one load in flight per thread) the same sum ran at 0.27 TB/s (1.67 ms per step at cfg5).
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t rows, int64_t C,
                                                             float* __restrict__ partials, int rows_per_wg) {
  __shared__ float s_p[256];
  const int col_threads = C < 256 ? (int)C : 256;
  const int row_lanes = 256 / col_threads;
  const int cl = threadIdx.x % col_threads, rl = threadIdx.x / col_threads;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  const int64_t c = (int64_t)blockIdx.y * col_threads + cl;
  float s = 0.f;
  if (rl < row_lanes && c < C) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0 + rl; r < r1; r += 8 * row_lanes) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t rr = r + (int64_t)u * row_lanes;
        const float xv = x[(rr < r1 ? rr : r1 - 1) * C + c];
        v[u] = rr < r1 ? xv : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += v[u];
    }
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  s_p[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
    for (int l = 0; l < row_lanes; ++l) t += s_p[l * col_threads + cl];
    partials[(int64_t)blockIdx.x * C + c] = t;
  }
}
// out[0] = scale * sum(x[0..n)), one wave, fixed order (n is the batch size)
__global__ __launch_bounds__(64) void scale_sum_kernel(const float* __restrict__ x, int64_t n, float scale, float* __restrict__ out) {
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 64) s += x[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * scale;
}
// The three small launches behind the dense head's CTC gradient - the row-padded bf16 copy of d(logits) (operand of the decoder's
// two gradient GEMMs), the slab partials of its column sums (decoder bias gradient) and the batch mean of the per-utterance losses
// (train.py:77) - read the same two tensors and do not depend on each other: one grid, three kinds of workgroup (round 4).
__global__ __launch_bounds__(256) void head_tail_kernel(const float* __restrict__ gl, int64_t rows, int64_t C, bf16_t* __restrict__ gl_bf16,
                                                        int64_t ld_out, int n_cast, float* __restrict__ partials, int rows_per_wg, int n_sum,
                                                        const float* __restrict__ nll, int64_t n_nll, float scale, float* __restrict__ loss) {
  const int id = blockIdx.x;
  if (id < n_cast) {                                     // cast_pad_bf16_kernel
    const int64_t total = rows * ld_out;
    for (int64_t i = id * (int64_t)256 + threadIdx.x; i < total; i += (int64_t)n_cast * 256) {
      const int64_t r = i / ld_out, c = i - r * ld_out;
      gl_bf16[i] = c < C ? f32_to_bf16(gl[r * C + c]) : (bf16_t)0;
    }
    return;
  }
  if (id < n_cast + n_sum) {                             // colsum_partial_kernel (C <= 256: one column chunk)
    __shared__ float s_p[256];
    const int blk = id - n_cast;
    const int col_threads = (int)C, row_lanes = 256 / col_threads;
    const int cl = threadIdx.x % col_threads, rl = threadIdx.x / col_threads;
    const int64_t r0 = (int64_t)blk * rows_per_wg;
    const int64_t r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
    float s = 0.f;
    if (rl < row_lanes) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int64_t r = r0 + rl; r < r1; r += 8 * row_lanes) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t rr = r + (int64_t)u * row_lanes;
          const float xv = gl[(rr < r1 ? rr : r1 - 1) * C + cl];
          v[u] = rr < r1 ? xv : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += v[u];
      }
      s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    s_p[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
      float t = 0.f;
      for (int l = 0; l < row_lanes; ++l) t += s_p[l * col_threads + cl];
      partials[(int64_t)blk * C + cl] = t;
    }
    return;
  }
  if (threadIdx.x < 64) {                                // scale_sum_kernel: one wave, fixed order
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n_nll; i += 64) s += nll[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s * scale;
  }
}
}  // namespace lasr

// the dense small-vocabulary head's tail in one launch (see head_tail_kernel): gl_bf16 = row-padded bf16 copy of gl [rows][C],
// bias_grad = column sums of gl (through `workspace`, second stage = launch_reduce_partials, fixed order), loss = scale * sum(nll).
// Returns 1 (nothing launched) for shapes the merged grid does not take (C > 256).
int lasr::head_tail(const float* gl, int64_t rows, int64_t C, void* gl_bf16, int64_t ld_out, float* bias_grad, void* workspace,
                    size_t workspace_bytes, const float* nll, int64_t n_nll, float scale, float* loss, void* stream) {
  static const bool off = getenv("LASR_HEAD_TAIL_MERGED") && atoi(getenv("LASR_HEAD_TAIL_MERGED")) == 0;
  if (off || C > 256 || C < 1) return 1;
  LASR_CHECK_ARG(gl && gl_bf16 && bias_grad && workspace && nll && loss && rows > 0 && ld_out >= C && n_nll > 0, "head_tail: bad argument");
  if (workspace_bytes < lasr_colsum_workspace_bytes(rows, C)) return fail(LASR_E_WORKSPACE, "head_tail: workspace");
  const int rpw = colsum_rows(C);
  const int n_sum = (int)cdiv(rows, rpw);
  int64_t n_cast = cdiv(rows * ld_out, 256);
  if (n_cast > 256 * 16) n_cast = 256 * 16;
  float* partials = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(lasr::head_tail_kernel, dim3((unsigned)(n_cast + n_sum + 1)), dim3(256), 0, as_stream(stream), gl, rows, C,
                     reinterpret_cast<bf16_t*>(gl_bf16), ld_out, (int)n_cast, partials, rpw, n_sum, nll, n_nll, scale, loss);
  LASR_LAUNCH_CHECK("head_tail_kernel");
  return launch_reduce_partials(partials, n_sum, C, bias_grad, C, nullptr, as_stream(stream));   // f64, fixed order
}

// out[i] = sum_p partials[p*n + i] for up to 64 independent (partials, out, n, n_partials) segments in ONE launch:
// the deferred reductions of a backward stage (split-K slabs of the 1x1 weight gradients, per-(utterance, chunk)
// partials of the depthwise weight gradients).  f64 accumulation in a fixed order, 16 loads in flight.
namespace lasr {
struct ReduceMany { lasr_reduce_desc d[64]; };
static constexpr int kReduceEl = 4;        // elements per thread (reduce_many_kernel)
__global__ __launch_bounds__(256) void reduce_many_kernel(ReduceMany a) {
  const lasr_reduce_desc& q = a.d[blockIdx.y];
  const int64_t i0 = (int64_t)blockIdx.x * (256 * kReduceEl) + threadIdx.x;
  if (q.n_partials <= 4 && i0 >= q.n) return;
  if (q.n_partials <= 4 && i0 + 256 * (kReduceEl - 1) < q.n) {
    // split-K slabs (2-4 partials): kReduceEl elements per thread with all of their loads in flight before the first add (one
    // element per thread left 3 loads in flight per lane: 59 MB at 2 TB/s).  Per element the same f64 sum in the same order as
    // reduce_many_elem (partials 0..3 into accumulators a0..a3, then (a0 + a1) + (a2 + a3)).
    float v[kReduceEl][4];
#pragma unroll
    for (int e = 0; e < kReduceEl; ++e)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float x = q.partials[(int64_t)min(u, q.n_partials - 1) * q.n + i0 + 256 * e];
        v[e][u] = u < q.n_partials ? x : 0.f;
      }
#pragma unroll
    for (int e = 0; e < kReduceEl; ++e) {
      const double a0 = 0.0 + (double)v[e][0], a1 = 0.0 + (double)v[e][1], a2 = 0.0 + (double)v[e][2], a3 = 0.0 + (double)v[e][3];
      q.out[i0 + 256 * e] = (float)((a0 + a1) + (a2 + a3));
    }
    return;
  }
  if (q.n_partials <= 4) {                                 // a slab segment's ragged last block
#pragma unroll 1
    for (int e = 0; e < kReduceEl; ++e) {
      const int64_t i = i0 + 256 * e;
      if (i < q.n) reduce_many_elem(q, i);                 // reduce_body.h
    }
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // many partials: one element per thread, 16 loads in flight each
  if (i < q.n) reduce_many_elem(q, i);
}
}  // namespace lasr

extern "C" int lasr_reduce_many(const lasr_reduce_desc* descs, int n_descs, void* stream) {
  LASR_CHECK_ARG(descs && n_descs >= 1 && n_descs <= 64, "lasr_reduce_many: 1..64 segments");
  lasr::ReduceMany a;
  int64_t nmax = 0;
  static const bool dump = getenv("LASR_REDUCE_DUMP") != nullptr;      // dev aid: the segments of every launch on stderr
  if (dump) {
    fprintf(stderr, "lasr_reduce_many: %d segments:", n_descs);
    for (int i = 0; i < n_descs; ++i) fprintf(stderr, " %lldx%d", (long long)descs[i].n, descs[i].n_partials);
    fprintf(stderr, "\n");
  }
  for (int i = 0; i < n_descs; ++i) {
    LASR_CHECK_ARG(descs[i].partials && descs[i].out && descs[i].n > 0 && descs[i].n_partials > 0, "lasr_reduce_many: bad segment");
    a.d[i] = descs[i];
    nmax = std::max<int64_t>(nmax, cdiv(descs[i].n, descs[i].n_partials <= 4 ? 256 * lasr::kReduceEl : 256));   // blocks this segment needs
  }
  hipLaunchKernelGGL(lasr::reduce_many_kernel, dim3((unsigned)nmax, (unsigned)n_descs), dim3(256), 0, as_stream(stream), a);
  LASR_LAUNCH_CHECK("reduce_many_kernel");
  return 0;
}

extern "C" size_t lasr_colsum_workspace_bytes(int64_t rows, int64_t C) {
  return (size_t)cdiv(rows, colsum_rows(C)) * C * sizeof(float);
}
extern "C" int lasr_colsum_f32(const float* x, float* out, int64_t rows, int64_t C, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(x && out && workspace && rows > 0 && C > 0, "lasr_colsum_f32: bad argument");
  if (workspace_bytes < lasr_colsum_workspace_bytes(rows, C)) return fail(LASR_E_WORKSPACE, "lasr_colsum_f32: workspace");
  const int rpw = colsum_rows(C);
  const int nblk = (int)cdiv(rows, rpw);
  float* partials = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, (unsigned)cdiv(C, C < 256 ? C : 256)), dim3(256), 0, as_stream(stream), x, rows, C, partials, rpw);
  LASR_LAUNCH_CHECK("colsum_partial_kernel");
  return launch_reduce_partials(partials, nblk, C, out, C, nullptr, as_stream(stream));   // f64, fixed order
}
extern "C" int lasr_scale_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream) {
  LASR_CHECK_ARG(x && out && n > 0, "lasr_scale_sum_f32: bad argument");
  hipLaunchKernelGGL(scale_sum_kernel, dim3(1), dim3(64), 0, as_stream(stream), x, n, scale, out);
  LASR_LAUNCH_CHECK("scale_sum_kernel");
  return 0;
}
