// Large-vocabulary loss head (BASELINE cfg5: AISHELL, C = 4334, T' up to 801): log-softmax and CTC WITHOUT the (B, T', C)
// f32 log-prob / gradient tensors (444 MB each at bs = 32; models/QuartNet.py:275-290, train.py:76-78).
//
//   decoder GEMM epilogue (gemm_bf16.hip)   bf16 logits [N][ldc] + per (row, column tile) partial (max, sum exp, argmax)
//   lse_gather_kernel                       lse[row], argmax[row]; E[row][i] = logit[row][target_i] - lse[row] (i < S),
//                                           E[row][S_max] = logit[row][blank] - lse[row]  - the only emissions the lattice reads
//   ctc_alpha_beta_kernel<.., COMPACT>      the same lattice recursion over E (class of an odd state = its label POSITION)
//   ctc_grad_lean_kernel                    bf16 d(loss)/d(logits) = gs * (exp(logit - lse) - occupancy) straight from the bf16
//                                           logits (no cast pass), and the decoder-bias gradient's column sums on the way
// HBM traffic per step at cfg5: 222 MB logits written once and read once, 222 MB gradient written once (+ what the two
// decoder gradient GEMMs read), against 444 + 888 + 888 + 666 + 444 MB for logits / log_softmax / CTC gradient / cast / colsum.
// The statistics are taken from the logits AS STORED (bf16), so exp(logit - lse) sums to one over the stored row.
#include "ctc_lattice.h"
#include "mel.h"

namespace lasr {

// one wave per row.  stat [N][gn][2] = (max, sum exp(x - max)) per column tile, arg [N][gn] = first argmax inside the tile
__global__ __launch_bounds__(256) void lse_gather_kernel(const bf16_t* __restrict__ logits, int64_t ldc, const float* __restrict__ stat,
                                                         const int32_t* __restrict__ arg, int gn, const int64_t* __restrict__ targets,
                                                         const int32_t* __restrict__ tgt_lens, int64_t N, int64_t T, int64_t S_max,
                                                         int CE, int blank, float* __restrict__ lse, int32_t* __restrict__ argmax,
                                                         float* __restrict__ E) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float m = kNegInf, s = 0.f;
  int mi = 0x7fffffff;
  for (int j = lane; j < gn; j += 64) {          // gn <= 64 for C <= 8192 with 128-wide tiles: one trip
    const float mj = stat[(row * gn + j) * 2], sj = stat[(row * gn + j) * 2 + 1];
    const int ij = arg[row * gn + j];
    if (mj > m) { s = s * __expf(m - mj) + sj; m = mj; mi = ij; }
    else { s += sj * __expf(mj - m); if (mj == m && ij < mi) mi = ij; }
  }
  const float M = wave_max(m);
  float part = (m == kNegInf) ? 0.f : s * __expf(m - M);
  part = wave_sum(part);
  int cand = (m == M) ? mi : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  const float l = M + __logf(part);
  if (lane == 0) {
    lse[row] = l;
    if (argmax) argmax[row] = cand;
  }
  const int64_t b = row / T;
  const int S = tgt_lens[b];
  const bf16_t* x = logits + row * ldc;
  float* e = E + row * CE;
  const int64_t* tg = targets + b * S_max;
  for (int i = lane; i < CE; i += 64) {
    float v = 0.f;
    if (i < S) v = bf16_to_f32(x[min(max(tg[i], (int64_t)0), (int64_t)blank)]) - l;   // (blank = C-1: the last class; bad labels stay in bounds)
    else if (i == (int)S_max) v = bf16_to_f32(x[blank]) - l;
    e[i] = v;
  }
}

// rows_per_wg rows per workgroup (a quarter per wave, one row at a time).  grad [N][ldc] bf16 (pad columns written as zeros);
// bias_partials [grid][C] f32.  KV = 16-byte vectors per lane and row (9: C <= 4608, 18: C <= 9216).
// A row is two HBM round trips (its logits; lse / alpha / beta / emissions) and three single-wave LDS phases.  Every global load
// of row k + 1 is issued - unconditionally, from clamped addresses - before row k is processed, the per-utterance scalars and
// label tables are (re)loaded only when the utterance changes, and the wave index is made scalar so that they are scalar loads:
// the row loop itself waits on nothing but the previous iteration's prefetch.  (First form: loads inside `if (v < nvec)`,
// vector loads of per-utterance scalars and label-chain hops from global memory in the middle of a row - 19 us per row and wave.)
template <int NS, int KV>
__global__ __launch_bounds__(256, 2) void ctc_grad_lean_kernel(const bf16_t* __restrict__ logits, int64_t ldc, const float* __restrict__ lse,
                                                            const float* __restrict__ E, int CE, const int64_t* __restrict__ targets,
                                                            const int32_t* __restrict__ in_lens, const int32_t* __restrict__ tgt_lens,
                                                            int64_t B, int64_t T, int64_t C, int64_t S_max, int blank,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta,
                                                            const int32_t* __restrict__ next_same, const float* __restrict__ nll,
                                                            const float* __restrict__ gscale, bf16_t* __restrict__ grad,
                                                            float* __restrict__ bias_partials, int rows_per_wg) {
  constexpr int SP = 64 * NS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cpad = (int)((C + 7) & ~(int64_t)7);      // LDS row: C rounded up to whole 8-column vectors
  const size_t pitch = (size_t)cpad + 4 * (size_t)S_max;
  float* s_row = smem + (size_t)wid * pitch;
  float* s_v = s_row + cpad;                                          // [S_max] occupancy of the label states
  int32_t* s_tg = reinterpret_cast<int32_t*>(s_v + S_max);            // [S_max] labels of the current utterance, clamped
  int32_t* s_nx = s_tg + S_max;                                       // [2 * S_max] same-label chains (ctc_lattice.h)
  const int nvec = (int)(ldc >> 3);                   // 16-byte vectors per row (ldc % 8 == 0)
  float acc[KV][8];
#pragma unroll
  for (int j = 0; j < KV; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
  const int rpw = rows_per_wg >> 2;
  const int64_t NT = B * T;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg + (int64_t)wid * rpw;

  // prefetch registers of one row
  uint4 xr[KV];
  float pa[NS], pb[NS], pe[NS], pl;
  auto issue = [&](int64_t row_) {
    const int64_t r = row_ < NT ? row_ : NT - 1;
    const bf16_t* x = logits + r * ldc;
#pragma unroll
    for (int j = 0; j < KV; ++j) xr[j] = Vec<bf16_t>::raw(x + (size_t)min(lane + 64 * j, nvec - 1) * 8);
    pl = lse[r];
    const float* al = alpha + r * SP;
    const float* be = beta + r * SP;
    const float* er = E + r * CE;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int st = lane * NS + i;
      pa[i] = al[st]; pb[i] = be[st];
      pe[i] = (st & 1) ? er[min(st >> 1, (int)S_max)] : er[S_max];
    }
  };
  int64_t cached_b = -1;
  int Tb = 0, S = 0;
  float gs = 0.f, nl = 0.f;
  if (r_begin < NT) issue(r_begin);
  for (int k = 0; k < rpw; ++k) {
    const int64_t row = r_begin + k;
    if (row >= NT) break;                             // wave-uniform
    const int64_t b = row / T, t = row - b * T;
    if (b != cached_b) {                              // wave-uniform, once per utterance
      cached_b = b;
      Tb = in_lens[b]; S = tgt_lens[b];
      gs = gscale ? gscale[b] : 1.0f / (float)B;
      nl = nll[b];
      const int64_t* tg = targets + b * S_max;
      const int32_t* nx = next_same + b * S_max * 2;
      for (int i = lane; i < (int)S_max; i += 64) s_tg[i] = (int)min(max(tg[i], (int64_t)0), C - 1);
      for (int i = lane; i < 2 * (int)S_max; i += 64) s_nx[i] = nx[i];
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    // this row's operands out of the prefetch registers, the next row's loads in flight behind them
    uint4 cx[KV];
    float ca[NS], cb[NS], ce[NS];
#pragma unroll
    for (int j = 0; j < KV; ++j) cx[j] = xr[j];
#pragma unroll
    for (int i = 0; i < NS; ++i) { ca[i] = pa[i]; cb[i] = pb[i]; ce[i] = pe[i]; }
    const float l = pl;
    issue(row + 1);
    bf16_t* g = grad + row * ldc;
    if (t >= Tb) {                                    // frames past the utterance: zero gradient
#pragma unroll
      for (int j = 0; j < KV; ++j) {
        const int v = lane + 64 * j;
        if (v < nvec) *reinterpret_cast<uint4*>(g + (size_t)v * 8) = make_uint4(0u, 0u, 0u, 0u);
      }
      continue;
    }
    const int SS = 2 * S + 1;
    const bool infeasible = isinf(nl);
    // softmax of the stored row -> LDS
#pragma unroll
    for (int j = 0; j < KV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        float xv[8];
        Vec<bf16_t>::unpack(cx[j], xv);
        float4 lo, hi;
        lo.x = __expf(xv[0] - l); lo.y = __expf(xv[1] - l); lo.z = __expf(xv[2] - l); lo.w = __expf(xv[3] - l);
        hi.x = __expf(xv[4] - l); hi.y = __expf(xv[5] - l); hi.z = __expf(xv[6] - l); hi.w = __expf(xv[7] - l);
        *reinterpret_cast<float4*>(s_row + (size_t)v * 8) = lo;
        *reinterpret_cast<float4*>(s_row + (size_t)v * 8 + 4) = hi;
      }
    }
    // occupancy of every lattice state (emissions of the states come from the compact matrix)
    float blank_occ = 0.f;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const int st = lane * NS + i;
      if (st < SS) {
        const float v = expf(ca[i] + cb[i] + nl - ce[i]);
        if (st & 1) s_v[st >> 1] = v;
        else blank_occ += v;
      }
    }
    blank_occ = wave_sum(blank_occ);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes have landed (single-wave hand-off)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) s_row[blank] -= blank_occ;
    for (int i = lane; i < S; i += 64) {
      if (s_nx[S_max + i]) {   // first occurrence of its label: sum the chain in target order (deterministic)
        float a = 0.f;
        for (int j = i; j >= 0; j = s_nx[j]) a += s_v[j];
        s_row[s_tg[i]] -= a;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < KV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        const float4 lo = *reinterpret_cast<const float4*>(s_row + (size_t)v * 8);
        const float4 hi = *reinterpret_cast<const float4*>(s_row + (size_t)v * 8 + 4);
        float o[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool real = v * 8 + e < C;               // pad columns of the last vector hold exp(0 - lse): not a class
          o[e] = real ? (infeasible ? __builtin_nanf("") : gs * o[e]) : 0.f;
          acc[j][e] += o[e];
        }
        Vec<bf16_t>::store(g + (size_t)v * 8, o);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row buffer is rewritten by the next row
    __builtin_amdgcn_wave_barrier();
  }
  // column sums of the f32 gradient (decoder.bias): wave partial -> own LDS row -> fixed-order sum over the 4 waves
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KV; ++j) {
    const int v = lane + 64 * j;
    if (v < nvec) {
      *reinterpret_cast<float4*>(s_row + (size_t)v * 8) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
      *reinterpret_cast<float4*>(s_row + (size_t)v * 8 + 4) = make_float4(acc[j][4], acc[j][5], acc[j][6], acc[j][7]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    bias_partials[(size_t)blockIdx.x * C + c] = (smem[c] + smem[pitch + c]) + (smem[2 * pitch + c] + smem[3 * pitch + c]);
}

static inline int lean_ns(int64_t S_max) {
  const int64_t ss = 2 * S_max + 1;
  if (ss <= 64 * 4) return 4;
  if (ss <= 64 * 8) return 8;
  if (ss <= 64 * 16) return 16;
  return 0;
}
static constexpr int kLeanRowsPerWg = 64;

template <int NS, bool EM_LDS>
__global__ __launch_bounds__(128) void ctc_alpha_beta_compact_kernel(const float* __restrict__ E, const int64_t* __restrict__ targets,
                                                                     const int32_t* __restrict__ in_lens,
                                                                     const int32_t* __restrict__ tgt_lens, int64_t T, int64_t CE,
                                                                     int64_t S_max, int blank_col, float* __restrict__ alpha,
                                                                     float* __restrict__ beta, int32_t* __restrict__ next_same,
                                                                     float* __restrict__ nll) {
  __shared__ int32_t s_tg[kCtcMaxS];
  extern __shared__ __attribute__((aligned(16))) float s_lp[];
  ctc_alpha_beta_body<NS, EM_LDS, 128, true>(E, targets, in_lens, tgt_lens, T, CE, S_max, blank_col, alpha, beta, next_same, nll, blockIdx.x,
                                             s_tg, s_lp);
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_ctc_lean_workspace_bytes(int64_t B, int64_t T, int64_t C, int64_t S_max) {
  const int64_t sm = S_max > 0 ? S_max : 1;
  const int64_t CE = (sm + 1 + 3) & ~(int64_t)3;
  return align_up(lasr_ctc_workspace_bytes(B, T, S_max), 256) + align_up((size_t)B * T * sizeof(float), 256) +
         align_up((size_t)B * T * CE * sizeof(float), 256) + align_up((size_t)cdiv(B * T, kLeanRowsPerWg) * C * sizeof(float), 256);
}

// logits [B*T][ldc] bf16 with row_stat / row_arg from lasr_gemm_rowstat (n_col_tiles column tiles).  Outputs: nll (B),
// argmax (B*T, may be NULL), grad [B*T][ldc] bf16 = gscale_b * d nll_b / d logits (1/B when gscale is NULL), bias_grad (C) f32.
extern "C" int lasr_ctc_loss_lean(const void* logits, int64_t ldc, const float* row_stat, const int32_t* row_arg, int n_col_tiles,
                                  const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T, int64_t C,
                                  int64_t S_max, int blank, float* nll, int32_t* argmax, void* grad, float* bias_grad,
                                  const float* gscale, void* workspace, size_t workspace_bytes, void* stream) {
  return ctc_loss_lean_job(logits, ldc, row_stat, row_arg, n_col_tiles, targets, in_lens, tgt_lens, B, T, C, S_max, blank, nll, argmax, grad,
                           bias_grad, gscale, workspace, workspace_bytes, nullptr, stream);
}

// job != null: the log-mel features of ANOTHER batch (the prefetch of the next step) are computed in the grid of the lattice kernel
// when its emissions fit one workgroup's LDS next to the labels, behind the gradient kernel otherwise
int lasr::ctc_loss_lean_job(const void* logits, int64_t ldc, const float* row_stat, const int32_t* row_arg, int n_col_tiles,
                            const int64_t* targets, const int32_t* in_lens, const int32_t* tgt_lens, int64_t B, int64_t T, int64_t C,
                            int64_t S_max, int blank, float* nll, int32_t* argmax, void* grad, float* bias_grad, const float* gscale,
                            void* workspace, size_t workspace_bytes, const MelJob* job, void* stream) {
  LASR_CHECK_ARG(logits && row_stat && row_arg && targets && in_lens && tgt_lens && nll && grad && bias_grad && workspace,
                 "lasr_ctc_loss_lean: null pointer");
  LASR_CHECK_SHAPE(B > 0 && T > 0 && C > 1 && S_max >= 0 && blank >= 0 && blank < C && ldc == ((C + 7) & ~(int64_t)7) && ldc <= 9216 &&
                       n_col_tiles >= 1 && n_col_tiles <= 64 && B * T * ldc < ((int64_t)1 << 31),
                   "lasr_ctc_loss_lean: shape (C=%lld ldc=%lld tiles=%d)", (long long)C, (long long)ldc, n_col_tiles);
  const int ns = lean_ns(S_max);
  LASR_CHECK_SHAPE(ns != 0, "lasr_ctc_loss_lean: S_max=%lld exceeds the 511-label lattice the kernels are built for", (long long)S_max);
  if (workspace_bytes < lasr_ctc_lean_workspace_bytes(B, T, C, S_max)) return fail(LASR_E_WORKSPACE, "lasr_ctc_loss_lean: workspace");
  const int64_t sm = S_max > 0 ? S_max : 1, N = B * T;
  const int CE = (int)((sm + 1 + 3) & ~(int64_t)3);
  char* w = reinterpret_cast<char*>(workspace);
  void* lattice_ws = w;
  w += align_up(lasr_ctc_workspace_bytes(B, T, S_max), 256);
  float* lse = reinterpret_cast<float*>(w);
  w += align_up((size_t)N * sizeof(float), 256);
  float* E = reinterpret_cast<float*>(w);
  w += align_up((size_t)N * CE * sizeof(float), 256);
  float* bias_partials = reinterpret_cast<float*>(w);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(lse_gather_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, st, reinterpret_cast<const bf16_t*>(logits), ldc, row_stat,
                     row_arg, n_col_tiles, targets, tgt_lens, N, T, sm, CE, blank, lse, argmax, E);
  LASR_LAUNCH_CHECK("lse_gather_kernel");
  const size_t ab = (size_t)B * T * 64 * ns;
  float* alpha = reinterpret_cast<float*>(lattice_ws);
  float* beta = alpha + ab;
  int32_t* next_same = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(lattice_ws) + align_up(2 * ab * sizeof(float), 256));
  const size_t em_bytes = (size_t)(T + 2) * CE * sizeof(float);
  // (160 KB of LDS per workgroup less the 2 KB of static label storage: T' = 801 with 46 emission columns needs 147.8 KB)
  const bool em_lds = em_bytes <= 156 * 1024 && !getenv("LASR_CTC_NO_LDS");
#define LASR_CTC_AB(NS_)                                                                                                       \
  do {                                                                                                                         \
    if (em_lds) {                                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_alpha_beta_compact_kernel<NS_, true>),                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);                                       \
      hipLaunchKernelGGL((ctc_alpha_beta_compact_kernel<NS_, true>), dim3((unsigned)B), dim3(128), em_bytes, st, E, targets, in_lens, \
                         tgt_lens, T, (int64_t)CE, sm, (int)sm, alpha, beta, next_same, nll);                                  \
    } else {                                                                                                                   \
      hipLaunchKernelGGL((ctc_alpha_beta_compact_kernel<NS_, false>), dim3((unsigned)B), dim3(128), 0, st, E, targets, in_lens, \
                         tgt_lens, T, (int64_t)CE, sm, (int)sm, alpha, beta, next_same, nll);                                  \
    }                                                                                                                          \
  } while (0)
  bool job_done = false;
  if (job && em_lds && compact_lattice_mel_fits(T, CE)) {
    LASR_TRY(launch_compact_lattice_mel(E, targets, in_lens, tgt_lens, B, T, CE, sm, (int)sm, alpha, beta, next_same, nll, ns, *job, stream));
    job_done = true;
  } else {
    if (ns == 4) LASR_CTC_AB(4); else if (ns == 8) LASR_CTC_AB(8); else LASR_CTC_AB(16);
    LASR_LAUNCH_CHECK("ctc_alpha_beta_compact_kernel");
  }
#undef LASR_CTC_AB
  const int cpad = (int)((C + 7) & ~(int64_t)7);
  const size_t shmem = 4 * ((size_t)cpad + 4 * (size_t)sm) * sizeof(float);
  LASR_CHECK_SHAPE(shmem <= 160 * 1024 && ldc / 8 <= 64 * 18, "lasr_ctc_loss_lean: C=%lld too large for the LDS row buffers", (long long)C);
  const int nwg = (int)cdiv(N, kLeanRowsPerWg);
  const bool kv9 = ldc / 8 <= 64 * 9;
#define LASR_CTC_G2(NS_, KV_)                                                                                                  \
  do {                                                                                                                         \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_grad_lean_kernel<NS_, KV_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((ctc_grad_lean_kernel<NS_, KV_>), dim3((unsigned)nwg), dim3(256), shmem, st, reinterpret_cast<const bf16_t*>(logits), ldc, lse, \
                       E, CE, targets, in_lens, tgt_lens, B, T, C, sm, blank, alpha, beta, next_same, nll, gscale,                \
                       reinterpret_cast<bf16_t*>(grad), bias_partials, kLeanRowsPerWg);                                       \
  } while (0)
#define LASR_CTC_G(NS_) do { if (kv9) LASR_CTC_G2(NS_, 9); else LASR_CTC_G2(NS_, 18); } while (0)
  if (ns == 4) LASR_CTC_G(4); else if (ns == 8) LASR_CTC_G(8); else LASR_CTC_G(16);
#undef LASR_CTC_G
#undef LASR_CTC_G2
  LASR_LAUNCH_CHECK("ctc_grad_lean_kernel");
  LASR_TRY(launch_reduce_partials(bias_partials, nwg, C, bias_grad, C, nullptr, st));   // f64, fixed order
  if (job && !job_done)
    return mel_fwd_src(job->src, job->sample_lens, job->aug, job->B, job->L, job->normalize, nullptr, job->out_btf, job->dtype, job->frames_out,
                       job->pct_out, job->ws, job->ws_bytes, stream);
  return 0;
}
