// MFMA GEMM for the 1x1 ("pointwise") convolutions and every GEMM-shaped gradient:
//   C[M][N] = sum_k opA(A)[m][k] * opB(B)[n][k]  (+bias[n]) (+addend[m][n]), rows masked, BN sums.
// Replaces pointwise_conv / reside.0 / last_cnn2.0 / decoder (models/QuartNet.py:31,63,146,275)
// and their autograd backward GEMMs (dgrad: transB=1, wgrad: transA=1,transB=1 with split-K).
//
// This file holds the exact-f32 path: v_mfma_f32_32x32x2_f32 (bitwise a k-ordered fmaf chain,
// 64 FLOP/clk/SIMD).  128x128x16 block tile, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32.
// Both operands are staged K-major in LDS ([k][m], [k][n]) so a fragment read is 32 consecutive
// floats per half-wave (conflict-free ds_read_b32) whatever the global layout was.
#include "common.h"
#include "gemm.h"
#include "lstm_body.h"
#include <cmath>
#include <climits>
#include <stdlib.h>

namespace lasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BM = 128, BN = 128, BK = 16;
static constexpr int LDS_LD = BM + 4;  // row stride of the K-major tiles

// Load one BK x 128 operand tile (rows = k, cols = m) of op(X) into registers (8 floats / thread).
// TRANS=false: X is [R][K] (k contiguous): thread -> row = tid/4 + 64*p, k = 4*(tid%4)..+3
// TRANS=true : X is [K][R] (r contiguous): thread -> k = tid/32 + 8*p, r = 4*(tid%32)..+3
template <typename T, bool TRANS>
__device__ __forceinline__ void load_tile(const T* __restrict__ X, int64_t ld, int64_t R, int64_t K, int64_t r0, int64_t k0,
                                          int64_t kend, bool vec_ok, float (&reg)[2][4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    reg[p][0] = reg[p][1] = reg[p][2] = reg[p][3] = 0.f;
    if (!TRANS) {
      const int64_t r = r0 + (tid >> 2) + 64 * p;
      const int64_t k = k0 + ((tid & 3) << 2);
      if (r < R) {
        const T* src = X + r * ld + k;
        if (vec_ok && k + 3 < kend) {
          Elem<T>::ld4(src, reg[p]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k + e < kend) reg[p][e] = Elem<T>::ld(src + e);
        }
      }
    } else {
      const int64_t k = k0 + (tid >> 5) + 8 * p;
      const int64_t r = r0 + ((tid & 31) << 2);
      if (k < kend) {
        const T* src = X + k * ld + r;
        if (vec_ok && r + 3 < R) {
          Elem<T>::ld4(src, reg[p]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (r + e < R) reg[p][e] = Elem<T>::ld(src + e);
        }
      }
    }
  }
}

template <bool TRANS>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float (&reg)[2][4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (!TRANS) {
      const int r = (tid >> 2) + 64 * p;
      const int k = (tid & 3) << 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) s[(k + e) * LDS_LD + r] = reg[p][e];
    } else {
      const int k = (tid >> 5) + 8 * p;
      const int r = (tid & 31) << 2;
      *reinterpret_cast<float4*>(s + k * LDS_LD + r) = make_float4(reg[p][0], reg[p][1], reg[p][2], reg[p][3]);
    }
  }
}


// grid: (ceil(N/BN), ceil(M/BM), split_k)
template <typename TAB, typename TC, bool TRANS_A, bool TRANS_B>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[BK * LDS_LD];
  __shared__ __attribute__((aligned(16))) float sB[BK * LDS_LD];
  __shared__ float s_stat[2][2][BN];  // [wm][sum|sq][col]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  const TAB* A = reinterpret_cast<const TAB*>(g.A);
  const TAB* B = reinterpret_cast<const TAB*>(g.B);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[2][4], rb[2][4];
  load_tile<TAB, TRANS_A>(A, g.lda, g.M, g.K, m0, kbeg, kend, g.vecA, ra);
  load_tile<TAB, TRANS_B>(B, g.ldb, g.N, g.K, n0, kbeg, kend, g.vecB, rb);
  const int half = lane >> 5, l31 = lane & 31;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
    store_tile<TRANS_A>(sA, ra);
    store_tile<TRANS_B>(sB, rb);
    __syncthreads();
    if (k0 + BK < kend) {
      load_tile<TAB, TRANS_A>(A, g.lda, g.M, g.K, m0, k0 + BK, kend, g.vecA, ra);
      load_tile<TAB, TRANS_B>(B, g.ldb, g.N, g.K, n0, k0 + BK, kend, g.vecB, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float* pa = sA + (kk + half) * LDS_LD + wm * 64 + l31;
      const float* pb = sB + (kk + half) * LDS_LD + wn * 64 + l31;
      const float a0 = pa[0], a1 = pa[32], b0 = pb[0], b1 = pb[32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // ---- epilogue.  acc[mi][ni][r]: row = wm*64 + mi*32 + (r&3) + 8*(r>>2) + 4*half, col = wn*64 + ni*32 + l31
  if (g.split_ws) {
    float* W = g.split_ws + (int64_t)blockIdx.z * g.M * g.N;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int64_t n = n0 + wn * 64 + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (m < g.M && n < g.N) W[m * g.N + n] = acc[mi][ni][r];
        }
      }
    return;
  }
  TC* C = reinterpret_cast<TC*>(g.C);
  const TC* addend = reinterpret_cast<const TC*>(g.addend);
  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + l31;
    const bool n_ok = n < g.N;
    const float bv = (g.bias && n_ok) ? g.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M && n_ok) {
          float v = acc[mi][ni][r] + bv;
          if (addend) v += Elem<TC>::ld(addend + m * g.ldc + n);
          if (g.row_lens) {
            const int64_t b = m / g.rows_per_seq;
            if (m - b * g.rows_per_seq >= g.row_lens[b]) v = 0.f;
          }
          Elem<TC>::st(C + m * g.ldc + n, v);
          if (g.stat_partials) {
            // statistics of the value as stored (bf16-rounded when TC is bf16)
            const float sv = Elem<TC>::kDtype == LASR_BF16 ? bf16_to_f32(f32_to_bf16(v)) : v;
            csum[ni] += sv;
            csq[ni] = fmaf(sv, sv, csq[ni]);
          }
        }
      }
    }
  }
  if (g.stat_partials) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      csum[ni] += __shfl_xor(csum[ni], 32, 64);
      csq[ni] += __shfl_xor(csq[ni], 32, 64);
      if (half == 0) {
        s_stat[wm][0][wn * 64 + ni * 32 + l31] = csum[ni];
        s_stat[wm][1][wn * 64 + ni * 32 + l31] = csq[ni];
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int64_t n = n0 + tid;
      if (n < g.N) {
        float* P = g.stat_partials + (int64_t)blockIdx.y * 2 * g.N;
        P[n] = s_stat[0][0][tid] + s_stat[1][0][tid];
        P[g.N + n] = s_stat[0][1][tid] + s_stat[1][1][tid];
      }
    }
  }
}

// stats[i] = sum_blk partials[blk][i], i < 2N
__global__ __launch_bounds__(256) void gemm_stats_reduce_kernel(const float* __restrict__ partials, int nblk, int64_t n2,
                                                                float* __restrict__ stats) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n2) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)partials[(int64_t)b * n2 + i];
  stats[i] = (float)s;
}

// C = sum_s ws[s] (+bias)(+addend)
template <typename TC>
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ ws, int split, int64_t M, int64_t N,
                                                                int64_t ldc, const float* __restrict__ bias,
                                                                const TC* __restrict__ addend, TC* __restrict__ C) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  float s = 0.f;
  for (int p = 0; p < split; ++p) s += ws[(int64_t)p * M * N + i];
  if (bias) s += bias[n];
  if (addend) s += Elem<TC>::ld(addend + m * ldc + n);
  Elem<TC>::st(C + m * ldc + n, s);
}

// two split-K problems (a unit's main and residual weight gradient) reduced by one launch
struct SplitReduce2 { const float* ws[2]; float* C[2]; const float* bias[2]; int split[2]; int64_t M[2], N[2]; };
__global__ __launch_bounds__(256) void gemm_split_reduce2_kernel(SplitReduce2 a) {
  const int q = blockIdx.y;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t mn = a.M[q] * a.N[q];
  if (i >= mn) return;
  const float* ws = a.ws[q];
  float s = 0.f;
  for (int p = 0; p < a.split[q]; ++p) s += ws[(int64_t)p * mn + i];
  if (a.bias[q]) s += a.bias[q][i % a.N[q]];
  a.C[q][i] = s;
}

template <typename TAB, typename TC>
static int launch_f32(const GemmArgs& g, int transA, int transB, dim3 grid, hipStream_t st) {
#define LASR_GEMM_CASE(TA_, TB_)                                                                              \
  hipLaunchKernelGGL((gemm_f32_kernel<TAB, TC, TA_, TB_>), grid, dim3(256), 0, st, g)
  if (!transA && !transB) LASR_GEMM_CASE(false, false);
  else if (!transA && transB) LASR_GEMM_CASE(false, true);
  else if (transA && !transB) LASR_GEMM_CASE(true, false);
  else LASR_GEMM_CASE(true, true);
#undef LASR_GEMM_CASE
  LASR_LAUNCH_CHECK("gemm_f32_kernel");
  return 0;
}

}  // namespace lasr

using namespace lasr;

extern "C" size_t lasr_gemm_workspace_bytes(int64_t M, int64_t N, int split_k, int want_stats) {
  size_t b = 0;
  if (split_k > 1) b += align_up((size_t)split_k * M * N * sizeof(float), 256);
  if (want_stats) b += align_up((size_t)cdiv(M, BM) * 2 * N * sizeof(float), 256);
  return b;
}

// stat_out != null: the BN partial sums stay unreduced in the workspace; *stat_out = {pointer, row tiles}
struct StatOut { const float* partials; int tiles; };
static int gemm_impl(const void* A, const void* B, void* C, int dtype_ab, int dtype_c, int64_t M, int64_t N, int64_t K,
                     int transA, int transB, const float* bias, const void* addend, const int32_t* row_lens,
                     int64_t rows_per_seq, float* stats, int split_k, void* workspace, size_t workspace_bytes,
                     void* stream, StatOut* stat_out, int64_t lda = 0, int64_t ldb = 0, int64_t ldc = 0, int* split_left_out = nullptr) {
  // split_left_out != null (split_k > 1): the split-K slabs stay unreduced at the head of the workspace ([*split_left_out][M*N] f32,
  // no bias) for a consumer that sums them itself (gemm_split_partials_one)
  LASR_CHECK_ARG(A && B && C, "lasr_gemm: null pointer");
  LASR_CHECK_ARG((dtype_ab == LASR_F32 || dtype_ab == LASR_BF16) && (dtype_c == LASR_F32 || dtype_c == LASR_BF16), "lasr_gemm: bad dtype");
  LASR_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && split_k >= 1 && split_k <= 1024, "lasr_gemm: M=%lld N=%lld K=%lld split=%d",
                   (long long)M, (long long)N, (long long)K, split_k);
  LASR_CHECK_ARG(!(row_lens && rows_per_seq <= 0), "lasr_gemm: rows_per_seq");
  LASR_CHECK_ARG(!(split_k > 1 && (row_lens || stats)), "lasr_gemm: split_k excludes masking/statistics");
  const size_t need = lasr_gemm_workspace_bytes(M, N, split_k, stats != nullptr);
  if (need > 0 && (!workspace || workspace_bytes < need)) return fail(LASR_E_WORKSPACE, "lasr_gemm: workspace %zu < %zu", workspace_bytes, need);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
  g.lda = lda > 0 ? lda : (transA ? M : K); g.ldb = ldb > 0 ? ldb : (transB ? N : K); g.ldc = ldc > 0 ? ldc : N;
  LASR_CHECK_SHAPE(g.lda >= (transA ? M : K) && g.ldb >= (transB ? N : K) && g.ldc >= N, "lasr_gemm: leading dimension smaller than the row");
  g.bias = bias; g.addend = addend; g.row_lens = row_lens; g.rows_per_seq = rows_per_seq;
  g.stat_partials = nullptr; g.split_ws = nullptr;
  const size_t esz = dtype_size(dtype_ab);
  // bf16 operands take the bf16-MFMA kernel (gemm_bf16.hip) - with an addend only in the form its epilogue reads (bf16 C,
  // 4-element groups aligned); everything else the f32-MFMA one
  const bool addend_ok = addend == nullptr || (dtype_c == LASR_BF16 && N % 4 == 0 && g.ldc % 4 == 0 && reinterpret_cast<uintptr_t>(addend) % 8 == 0);
  const bool use_bf16 = dtype_ab == LASR_BF16 && addend_ok && !getenv("LASR_FORCE_F32_MFMA");
  // vector loads (4 elements, or 8 for the bf16 kernel) need the row pitch and base to keep every group aligned
  const int vw = use_bf16 ? 8 : 4;
  g.vecA = (g.lda % vw == 0) && (reinterpret_cast<uintptr_t>(A) % (vw * esz) == 0);
  g.vecB = (g.ldb % vw == 0) && (reinterpret_cast<uintptr_t>(B) % (vw * esz) == 0);
  char* wsp = reinterpret_cast<char*>(workspace);
  if (split_k > 1) {
    g.split_ws = reinterpret_cast<float*>(wsp);
    wsp += align_up((size_t)split_k * M * N * sizeof(float), 256);
    const int64_t per = cdiv(cdiv(K, split_k), 64) * 64;  // whole K tiles of either kernel
    g.k_per_split = per;
    split_k = (int)cdiv(K, per);
  } else {
    g.k_per_split = K;
  }
  int grid_m = (int)cdiv(M, BM);
  if (stats) g.stat_partials = reinterpret_cast<float*>(wsp);
  dim3 grid((unsigned)cdiv(N, BN), (unsigned)grid_m, (unsigned)split_k);
  hipStream_t st = as_stream(stream);
  const double gbytes = (double)(M * K + N * K) * esz + (double)M * N * dtype_size(dtype_c);
  const int tok = prof_begin(LASR_PROF_GEMM, st, 2.0 * (double)M * (double)N * (double)K, gbytes);
  int rc;
  if (use_bf16) rc = launch_gemm_bf16(g, split_k, dtype_c, transA, transB, st, &grid_m);
  else if (dtype_ab == LASR_F32) rc = dtype_c == LASR_F32 ? launch_f32<float, float>(g, transA, transB, grid, st) : launch_f32<float, bf16_t>(g, transA, transB, grid, st);
  else rc = dtype_c == LASR_F32 ? launch_f32<bf16_t, float>(g, transA, transB, grid, st) : launch_f32<bf16_t, bf16_t>(g, transA, transB, grid, st);
  prof_end(tok, st);
  if (rc) return rc;
  if (g.split_ws && split_left_out) {
    *split_left_out = split_k;
  } else if (g.split_ws) {
    const int64_t mn = M * N;
    if (dtype_c == LASR_F32)
      hipLaunchKernelGGL(gemm_split_reduce_kernel<float>, dim3((unsigned)cdiv(mn, 256)), dim3(256), 0, st, g.split_ws, split_k, M, N, g.ldc, bias, (const float*)addend, (float*)C);
    else
      hipLaunchKernelGGL(gemm_split_reduce_kernel<bf16_t>, dim3((unsigned)cdiv(mn, 256)), dim3(256), 0, st, g.split_ws, split_k, M, N, g.ldc, bias, (const bf16_t*)addend, (bf16_t*)C);
    LASR_LAUNCH_CHECK("gemm_split_reduce_kernel");
  }
  if (stats && stat_out) {
    stat_out->partials = g.stat_partials; stat_out->tiles = grid_m;
  } else if (stats) {
    LASR_TRY(launch_reduce_partials(g.stat_partials, grid_m, 2 * N, stats, 2 * N, nullptr, st));
  }
  return 0;
}

// one split-K problem, slabs left unreduced: partials = workspace, [*splits][M*N] f32 (the bias is NOT added)
int lasr::gemm_split_partials_one(const void* A, const void* B, int dtype_ab, int64_t M, int64_t N, int64_t K, int transA, int transB,
                                  int split_k, void* workspace, size_t workspace_bytes, int* splits, void* stream) {
  LASR_CHECK_ARG(splits && split_k > 1 && workspace, "gemm_split_partials_one: bad argument");
  return gemm_impl(A, B, workspace /* unused: nothing is reduced into C */, dtype_ab, LASR_F32, M, N, K, transA, transB, nullptr, nullptr, nullptr,
                   0, nullptr, split_k, workspace, workspace_bytes, stream, nullptr, 0, 0, 0, splits);
}

extern "C" int lasr_gemm_ld(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int dtype_ab, int dtype_c,
                            int64_t M, int64_t N, int64_t K, int transA, int transB, const float* bias, int split_k, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return gemm_impl(A, B, C, dtype_ab, dtype_c, M, N, K, transA, transB, bias, nullptr, nullptr, 0, nullptr, split_k, workspace,
                   workspace_bytes, stream, nullptr, lda, ldb, ldc);
}

extern "C" int lasr_gemm(const void* A, const void* B, void* C, int dtype_ab, int dtype_c, int64_t M, int64_t N, int64_t K,
                         int transA, int transB, const float* bias, const void* addend, const int32_t* row_lens,
                         int64_t rows_per_seq, float* stats, int split_k, void* workspace, size_t workspace_bytes,
                         void* stream) {
  return gemm_impl(A, B, C, dtype_ab, dtype_c, M, N, K, transA, transB, bias, addend, row_lens, rows_per_seq, stats, split_k,
                   workspace, workspace_bytes, stream, nullptr);
}

// C [M][ldc] bf16 = A [M][K] . B [N][K]^T + bias, with per (row, 256-column tile) softmax statistics of the stored values
// (the decoder of a large-vocabulary model feeding lasr_ctc_loss_lean).  row_stat [M][tiles][2] f32, row_arg [M][tiles] i32
// with tiles = ceil(N / 256) (returned through n_col_tiles).
extern "C" int lasr_gemm_rowstat(const void* A, const void* B, const float* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                 float* row_stat, int32_t* row_arg, int* n_col_tiles, void* stream) {
  LASR_CHECK_ARG(A && B && C && row_stat && row_arg, "lasr_gemm_rowstat: null pointer");
  LASR_CHECK_SHAPE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && ldc >= N && ldc % 8 == 0, "lasr_gemm_rowstat: M=%lld N=%lld K=%lld ldc=%lld",
                   (long long)M, (long long)N, (long long)K, (long long)ldc);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = ldc;
  g.bias = bias; g.addend = nullptr; g.row_lens = nullptr; g.rows_per_seq = 0; g.stat_partials = nullptr; g.split_ws = nullptr;
  g.k_per_split = K;
  g.vecA = reinterpret_cast<uintptr_t>(A) % 16 == 0; g.vecB = reinterpret_cast<uintptr_t>(B) % 16 == 0;
  hipStream_t st = as_stream(stream);
  const int tok = prof_begin(LASR_PROF_GEMM, st, 2.0 * (double)M * N * K, (double)(M * K + N * K) * 2 + (double)M * N * 2);
  const int rc = launch_gemm_bf16_rowstat(g, row_stat, row_arg, n_col_tiles, st);
  prof_end(tok, st);
  return rc;
}
extern "C" size_t lasr_gemm_rowstat_bytes(int64_t M, int64_t N) { return (size_t)M * cdiv(N, 256) * (2 * sizeof(float) + sizeof(int32_t)); }

// ---- two independent problems in one launch (bf16 operands); anything else runs them one by one ----
extern "C" size_t lasr_gemm_batch_workspace_bytes(const lasr_gemm_problem* probs, int n_probs, int split_k) {
  size_t b = 0;
  for (int i = 0; i < n_probs; ++i) b += lasr_gemm_workspace_bytes(probs[i].M, probs[i].N, split_k, probs[i].stats != nullptr);
  return b;
}

// split_out != null: the split-K slabs are left unreduced; split_out[i] = {slabs, count}
struct SplitOut { const float* partials; int splits; };
static int gemm_batch_impl(const lasr_gemm_problem* probs, int n_probs, int dtype_ab, int dtype_c, int transA, int transB,
                           int split_k, void* workspace, size_t workspace_bytes, void* stream, StatOut* stat_out,
                           SplitOut* split_out = nullptr) {
  LASR_CHECK_ARG(probs && n_probs >= 1 && n_probs <= 2, "lasr_gemm_batch: 1 or 2 problems");
  const size_t need = lasr_gemm_batch_workspace_bytes(probs, n_probs, split_k);
  if (need > 0 && (!workspace || workspace_bytes < need)) return fail(LASR_E_WORKSPACE, "lasr_gemm_batch: workspace %zu < %zu", workspace_bytes, need);
  char* wsp = reinterpret_cast<char*>(workspace);
  if (split_out && (dtype_ab != LASR_BF16 || n_probs == 1))
    return fail(LASR_E_ARG, "lasr_gemm_batch_split_partials: two bf16 problems per call");
  if (dtype_ab != LASR_BF16 || n_probs == 1 || getenv("LASR_NO_GEMM_BATCH")) {
    for (int i = 0; i < n_probs; ++i) {
      const lasr_gemm_problem& q = probs[i];
      const size_t nb = lasr_gemm_workspace_bytes(q.M, q.N, split_k, q.stats != nullptr);
      LASR_TRY(gemm_impl(q.A, q.B, q.C, dtype_ab, dtype_c, q.M, q.N, q.K, transA, transB, q.bias, nullptr, q.row_lens, q.rows_per_seq,
                         q.stats, split_k, wsp, nb, stream, stat_out ? stat_out + i : nullptr));
      wsp += nb;
    }
    return 0;
  }
  LASR_CHECK_ARG(dtype_c == LASR_F32 || dtype_c == LASR_BF16, "lasr_gemm_batch: bad dtype");
  GemmArgs g[2];
  int splits[2], stat_tiles[2];
  for (int i = 0; i < 2; ++i) {
    const lasr_gemm_problem& q = probs[i];
    LASR_CHECK_ARG(q.A && q.B && q.C, "lasr_gemm_batch: null pointer");
    LASR_CHECK_SHAPE(q.M > 0 && q.N > 0 && q.K > 0 && split_k >= 1 && split_k <= 1024, "lasr_gemm_batch: shape");
    LASR_CHECK_ARG(!(q.row_lens && q.rows_per_seq <= 0) && !(split_k > 1 && (q.row_lens || q.stats)), "lasr_gemm_batch: epilogue options");
    GemmArgs& a = g[i];
    a.A = q.A; a.B = q.B; a.C = q.C; a.M = q.M; a.N = q.N; a.K = q.K;
    a.lda = transA ? q.M : q.K; a.ldb = transB ? q.N : q.K; a.ldc = q.N;
    a.bias = q.bias; a.addend = nullptr; a.row_lens = q.row_lens; a.rows_per_seq = q.rows_per_seq;
    a.stat_partials = nullptr; a.split_ws = nullptr;
    a.vecA = (a.lda % 8 == 0) && (reinterpret_cast<uintptr_t>(q.A) % 16 == 0);
    a.vecB = (a.ldb % 8 == 0) && (reinterpret_cast<uintptr_t>(q.B) % 16 == 0);
    int sk = split_k;
    if (split_k > 1) {
      a.split_ws = reinterpret_cast<float*>(wsp);
      wsp += align_up((size_t)split_k * q.M * q.N * sizeof(float), 256);
      const int64_t per = cdiv(cdiv(q.K, split_k), 64) * 64;
      a.k_per_split = per;
      sk = (int)cdiv(q.K, per);
    } else {
      a.k_per_split = q.K;
    }
    if (q.stats) { a.stat_partials = reinterpret_cast<float*>(wsp); wsp += align_up((size_t)cdiv(q.M, BM) * 2 * q.N * sizeof(float), 256); }
    splits[i] = sk;
  }
  hipStream_t st = as_stream(stream);
  double fl = 0, by = 0;
  for (int i = 0; i < 2; ++i) {
    fl += 2.0 * (double)probs[i].M * probs[i].N * probs[i].K;
    by += (double)(probs[i].M * probs[i].K + probs[i].N * probs[i].K) * 2 + (double)probs[i].M * probs[i].N * dtype_size(dtype_c);
  }
  const int tok = prof_begin(LASR_PROF_GEMM, st, fl, by);
  const int rc = launch_gemm_bf16_batch(g, splits, 2, dtype_c, transA, transB, st, stat_tiles);
  prof_end(tok, st);
  if (rc) return rc;
  if (split_out) {
    for (int i = 0; i < 2; ++i) { split_out[i].partials = g[i].split_ws; split_out[i].splits = splits[i]; }
  } else if (g[0].split_ws && g[1].split_ws && dtype_c == LASR_F32) {
    SplitReduce2 a;
    int64_t mx = 0;
    for (int i = 0; i < 2; ++i) {
      a.ws[i] = g[i].split_ws; a.C[i] = reinterpret_cast<float*>(probs[i].C); a.bias[i] = probs[i].bias; a.split[i] = splits[i];
      a.M[i] = probs[i].M; a.N[i] = probs[i].N;
      mx = std::max(mx, probs[i].M * probs[i].N);
    }
    hipLaunchKernelGGL(gemm_split_reduce2_kernel, dim3((unsigned)cdiv(mx, 256), 2), dim3(256), 0, st, a);
    LASR_LAUNCH_CHECK("gemm_split_reduce2_kernel");
  } else {
    for (int i = 0; i < 2; ++i) {
      const lasr_gemm_problem& q = probs[i];
      if (g[i].split_ws) {
        const int64_t mn = q.M * q.N;
        if (dtype_c == LASR_F32)
          hipLaunchKernelGGL(gemm_split_reduce_kernel<float>, dim3((unsigned)cdiv(mn, 256)), dim3(256), 0, st, g[i].split_ws, splits[i], q.M, q.N, q.N, q.bias, (const float*)nullptr, (float*)q.C);
        else
          hipLaunchKernelGGL(gemm_split_reduce_kernel<bf16_t>, dim3((unsigned)cdiv(mn, 256)), dim3(256), 0, st, g[i].split_ws, splits[i], q.M, q.N, q.N, q.bias, (const bf16_t*)nullptr, (bf16_t*)q.C);
        LASR_LAUNCH_CHECK("gemm_split_reduce_kernel");
      }
    }
  }
  for (int i = 0; i < 2; ++i) {
    const lasr_gemm_problem& q = probs[i];
    if (q.stats && stat_out) {
      stat_out[i].partials = g[i].stat_partials; stat_out[i].tiles = stat_tiles[i];
    } else if (q.stats) {
      LASR_TRY(launch_reduce_partials(g[i].stat_partials, stat_tiles[i], 2 * q.N, q.stats, 2 * q.N, nullptr, st));
    }
  }
  return 0;
}

extern "C" int lasr_gemm_batch(const lasr_gemm_problem* probs, int n_probs, int dtype_ab, int dtype_c, int transA, int transB,
                               int split_k, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_batch_impl(probs, n_probs, dtype_ab, dtype_c, transA, transB, split_k, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int lasr_gemm_batch_partials(const lasr_gemm_problem* probs, int n_probs, int dtype_ab, int dtype_c, int transA, int transB,
                                        void* workspace, size_t workspace_bytes, const float** stat_partials, int* stat_tiles,
                                        void* stream) {
  LASR_CHECK_ARG(stat_partials && stat_tiles, "lasr_gemm_batch_partials: null output");
  StatOut so[2] = {{nullptr, 0}, {nullptr, 0}};
  LASR_TRY(gemm_batch_impl(probs, n_probs, dtype_ab, dtype_c, transA, transB, 1, workspace, workspace_bytes, stream, so));
  for (int i = 0; i < n_probs; ++i) { stat_partials[i] = so[i].partials; stat_tiles[i] = so[i].tiles; }
  return 0;
}

extern "C" int lasr_gemm_batch_split_partials(const lasr_gemm_problem* probs, int n_probs, int dtype_ab, int transA, int transB,
                                              int split_k, void* workspace, size_t workspace_bytes, const float** partials,
                                              int* splits, void* stream) {
  LASR_CHECK_ARG(partials && splits && split_k > 1, "lasr_gemm_batch_split_partials: bad argument");
  SplitOut so[2] = {{nullptr, 0}, {nullptr, 0}};
  LASR_TRY(gemm_batch_impl(probs, n_probs, dtype_ab, LASR_F32, transA, transB, split_k, workspace, workspace_bytes, stream, nullptr, so));
  for (int i = 0; i < n_probs; ++i) { partials[i] = so[i].partials; splits[i] = so[i].splits; }
  return 0;
}

static int multi_split_impl(const lasr_gemm_problem* probs, int n_probs, int split_k, float* const* slabs, int* splits, void* stream,
                            const lstm::BwdArgs* lstm_job, int lstm_wgs, const lasr_reduce_desc* riders = nullptr, int n_riders = 0,
                            int* riders_taken = nullptr);

extern "C" int lasr_gemm_multi_split_partials(const lasr_gemm_problem* probs, int n_probs, int split_k, float* const* slabs,
                                              int* splits, void* stream) {
  return multi_split_impl(probs, n_probs, split_k, slabs, splits, stream, nullptr, 0);
}

int lasr::gemm_multi_split_partials_riders(const lasr_gemm_problem* probs, int n_probs, int split_k, float* const* slabs, int* splits,
                                           const lasr_reduce_desc* riders, int n_riders, int* riders_taken, void* stream) {
  return multi_split_impl(probs, n_probs, split_k, slabs, splits, stream, nullptr, 0, riders, n_riders, riders_taken);
}

int lasr::gemm_multi_split_partials_with_bilstm_bwd(const lasr_gemm_problem* probs, int n_probs, int* n_taken, int split_k, float* const* slabs, int* splits,
                                                    const lstm::BwdArgs& rec, int dtype, int64_t B, void* stream) {
  const int64_t T_ = rec.Tt;
  static const bool off = getenv("LASR_LSTM_BESIDE_WGRAD") && atoi(getenv("LASR_LSTM_BESIDE_WGRAD")) == 0;
  static const bool small_only_ab = getenv("LASR_WGRAD_SMALL_TILE") != nullptr;   // the A/B switch takes the 128-row tiles: no room for the recurrences' slot
  if (off || small_only_ab || dtype != LASR_BF16 || B % 8 != 0 || B > 128 || n_probs < 1 || !n_taken) return 1;
  // How many slices the tiles should be cut into so that they are through when the recurrences are (measured on the MI355X, round 4:
  // 0.36 us per recurrence step; a 256 x 256 tile 1.7 us per 64 rows of K + ~15 us around them), and how many tiles of that length
  // fit in ONE round beside the B recurrence workgroups.  The problems past that budget stay with the caller: the stage's closing
  // launch has room for them (cfg4: 64 of the tiles here in 3 slices beside 2 B = 64 recurrence workgroups: the launch takes 216 us, the
  // closing launch 73 instead of 48).  LASR_LSTM_WGRAD_BUDGET=0: all of them, as before.
  static const bool budgeted = !(getenv("LASR_LSTM_WGRAD_BUDGET") && atoi(getenv("LASR_LSTM_WGRAD_BUDGET")) == 0);
  const double rec_us = 0.36 * (double)T_;
  int want = 1;
  while (want < std::min(split_k, 16) && (double)cdiv(cdiv(probs[0].K, want), 64) * 1.7 + 15.0 > rec_us) ++want;
  static const bool pair = getenv("LASR_LSTM_PAIR") && atoi(getenv("LASR_LSTM_PAIR")) == 1;   // (gemm_bf16.hip: the two directions of an utterance in one workgroup)
  const int64_t lstm_wgs = pair ? B : 2 * B;
  const int64_t budget = budgeted ? std::max<int64_t>((256 - lstm_wgs) / want, 1) : 256 - lstm_wgs;
  int64_t tiles_big = 0;
  int take = 0;
  for (int i = 0; i < n_probs; ++i) {
    const lasr_gemm_problem& q = probs[i];
    if (q.M % 8 || q.N % 8 || reinterpret_cast<uintptr_t>(q.A) % 16 || reinterpret_cast<uintptr_t>(q.B) % 16 || q.K < 1024) return 1;
    const int64_t t = cdiv(q.M, 256) * cdiv(q.N, 256);
    if (tiles_big + t > budget) break;
    tiles_big += t;
    take = i + 1;
  }
  if (take < 1 || tiles_big + lstm_wgs > 256) return 1;           // the recurrences and one slice of every tile must make ONE round
  n_probs = take;
  *n_taken = take;
  LASR_CHECK_ARG(rec.dout && rec.whh_f && rec.whh_r && rec.lens && rec.saved && rec.dg_f && rec.dg_r && rec.pwhh, "gemm + BiLSTM grid: null pointer");
  LASR_TRY(multi_split_impl(probs, n_probs, split_k, slabs, splits, stream, &rec, (int)lstm_wgs));
  // dW_hh (and the bias gradients' column sums) from the stored gate gradients (lstm.hip); the caller sums the partials
  return lstm::launch_dwhh_partials(rec, B, as_stream(stream));
}

static int multi_split_impl(const lasr_gemm_problem* probs, int n_probs, int split_k, float* const* slabs, int* splits, void* stream,
                            const lstm::BwdArgs* lstm_job, int lstm_wgs, const lasr_reduce_desc* riders, int n_riders, int* riders_taken) {
  if (riders_taken) *riders_taken = 0;
  LASR_CHECK_ARG(probs && slabs && splits && n_probs >= 1 && n_probs <= 32 && split_k >= 1 && split_k <= 1024,
                 "lasr_gemm_multi_split_partials: bad argument");
  GemmArgs g[32];
  int gz[32];
  // Tile form and slice count for the whole launch (split_k is the cap the slabs were sized for).
  //  * 256 x 256 tiles, one workgroup per CU: as many slices as keep the launch to one round of workgroups;
  //  * 128 x 128 tiles, two workgroups per CU: ~3 rounds of resident workgroups (1536 tile-slices).
  bool vec = true;
  int64_t tiles_big = 0, tiles_small = 0, kmin = INT64_MAX;
  for (int i = 0; i < n_probs; ++i) {
    const lasr_gemm_problem& q = probs[i];
    LASR_CHECK_ARG(q.A && q.B && slabs[i], "lasr_gemm_multi_split_partials: null pointer");
    LASR_CHECK_SHAPE(q.M > 0 && q.N > 0 && q.K > 0, "lasr_gemm_multi_split_partials: shape");
    vec = vec && q.M % 8 == 0 && q.N % 8 == 0 && reinterpret_cast<uintptr_t>(q.A) % 16 == 0 && reinterpret_cast<uintptr_t>(q.B) % 16 == 0;
    tiles_big += cdiv(q.M, 256) * cdiv(q.N, 256);
    tiles_small += cdiv(q.M, 128) * cdiv(q.N, 128);
    kmin = std::min<int64_t>(kmin, q.K);
  }
  static const bool small_only = getenv("LASR_WGRAD_SMALL_TILE") != nullptr;   // A/B switch
  const bool big_tile = !small_only && vec && kmin >= 1024;
  int split = 1;
  if (big_tile) {
    // one round of workgroups, as many CUs as the slice count reaches (measured on the 71-tile stage of the plain model:
    // 3 slices / 213 workgroups 175 us, 5 / 355 180 us, 7 / 497 194 us, 10 / 710 198 us: a second round costs its
    // prologue and another 256 KB slab per tile)
    split = (int)std::min<int64_t>(std::max<int64_t>((256 - lstm_wgs) / std::max<int64_t>(tiles_big, 1), 1), std::min(split_k, 16));   // (CUs the recurrences take)
  } else {
    split = (int)std::min<int64_t>(std::max<int64_t>(cdiv(1536, tiles_small), 1), split_k);
  }
  for (int i = 0; i < n_probs; ++i) {
    const lasr_gemm_problem& q = probs[i];
    GemmArgs& a = g[i];
    a.A = q.A; a.B = q.B; a.C = nullptr; a.M = q.M; a.N = q.N; a.K = q.K;
    a.lda = q.M; a.ldb = q.N; a.ldc = q.N;                       // transA = transB = 1: A is [K][M], B is [K][N]
    a.bias = nullptr; a.addend = nullptr; a.row_lens = nullptr; a.rows_per_seq = 0; a.stat_partials = nullptr;
    a.split_ws = slabs[i];
    a.vecA = (a.lda % 8 == 0) && (reinterpret_cast<uintptr_t>(q.A) % 16 == 0);
    a.vecB = (a.ldb % 8 == 0) && (reinterpret_cast<uintptr_t>(q.B) % 16 == 0);
    const int64_t per = cdiv(cdiv(q.K, split), 64) * 64;         // whole K tiles per slice
    a.k_per_split = per;
    gz[i] = (int)cdiv(q.K, per);
    splits[i] = gz[i];
  }
  hipStream_t st = as_stream(stream);
  double fl = 0, by = 0;
  for (int i = 0; i < n_probs; ++i) {
    fl += 2.0 * (double)probs[i].M * probs[i].N * probs[i].K;
    by += (double)(probs[i].M * probs[i].K + probs[i].N * probs[i].K) * 2 + (double)probs[i].M * probs[i].N * 4;
  }
  const int tok = prof_begin(LASR_PROF_GEMM, st, fl, by);
  if (lstm_job && !big_tile) { prof_end(tok, st); return fail(LASR_E_SHAPE, "gemm + BiLSTM grid needs the 256-row tile form"); }
  const int rc = launch_gemm_bf16_multi(g, gz, n_probs, big_tile, st, lstm_job, lstm_wgs, riders, n_riders, riders_taken);
  prof_end(tok, st);
  return rc;
}

// ---------------------------------------------------------------- folded eval form ------------------------------------
namespace lasr {
struct FoldMany { lasr_fold_desc d[32]; };
// Wcat[o][0:ci] = bf16(a[o] W[o][i]), Wcat[o][ci:2ci] = bf16(a2[o] Wr[o][i]) (second part only with a residual branch),
// bias[o] = b[o] (+ b2[o]);  a | b = coef[0:co] | coef[co:2co] of the eval-mode BN (lasr_bn_eval_coef_many)
__global__ __launch_bounds__(256) void fold_bn_weights_kernel(FoldMany f) {
  const lasr_fold_desc q = f.d[blockIdx.y];
  const int64_t ld = q.w_res ? 2 * q.ci : q.ci;
  const int64_t n = q.co * ld;
  bf16_t* out = reinterpret_cast<bf16_t*>(q.w_out);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t o = i / ld, c = i - o * ld;
    const float v = c < q.ci ? q.coef[o] * q.w[o * q.ci + c] : q.coef2[o] * q.w_res[o * q.ci + (c - q.ci)];
    out[i] = f32_to_bf16(v);
  }
  if (blockIdx.x == 0)
    for (int64_t o = threadIdx.x; o < q.co; o += 256) q.bias_out[o] = q.coef[q.co + o] + (q.w_res ? q.coef2[q.co + o] : 0.f);
}
}  // namespace lasr

extern "C" int lasr_fold_bn_weights_many(const lasr_fold_desc* descs, int n_descs, void* stream) {
  LASR_CHECK_ARG(descs && n_descs >= 1 && n_descs <= 32, "lasr_fold_bn_weights_many: 1..32 layers");
  lasr::FoldMany f;
  for (int i = 0; i < n_descs; ++i) {
    const lasr_fold_desc& q = descs[i];
    LASR_CHECK_ARG(q.w && q.coef && q.w_out && q.bias_out && q.co > 0 && q.ci > 0 && (!q.w_res || q.coef2), "lasr_fold_bn_weights_many: bad layer");
    f.d[i] = q;
  }
  hipLaunchKernelGGL(lasr::fold_bn_weights_kernel, dim3(256, (unsigned)n_descs), dim3(256), 0, as_stream(stream), f);
  LASR_LAUNCH_CHECK("fold_bn_weights_kernel");
  return 0;
}

extern "C" int lasr_gemm_dual(const void* A1, int64_t K1, const void* A2, int64_t K2, const void* W, const float* bias, void* C,
                              int64_t M, int64_t N, const int32_t* row_lens, int64_t rows_per_seq, int act, void* stream) {
  LASR_CHECK_ARG(A1 && A2 && W && bias && C, "lasr_gemm_dual: null pointer");
  LASR_CHECK_ARG(!(row_lens && rows_per_seq <= 0), "lasr_gemm_dual: rows_per_seq");
  LASR_CHECK_ARG(act == LASR_ACT_NONE || act == LASR_ACT_RELU || act == LASR_ACT_SWISH, "lasr_gemm_dual: act");
  LASR_CHECK_SHAPE(M > 0 && N > 0 && K1 > 0 && K2 > 0, "lasr_gemm_dual: shape");
  GemmArgs g;
  g.A = A1; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K1 + K2; g.lda = K1; g.ldb = K1 + K2; g.ldc = N;
  g.bias = bias; g.addend = nullptr; g.row_lens = row_lens; g.rows_per_seq = rows_per_seq; g.stat_partials = nullptr; g.split_ws = nullptr;
  g.k_per_split = K1 + K2; g.vecA = 1; g.vecB = 1;
  hipStream_t st = as_stream(stream);
  const int tok = prof_begin(LASR_PROF_GEMM, st, 2.0 * (double)M * N * (K1 + K2), (double)(M * (K1 + K2) + N * (K1 + K2) + M * N) * 2);
  const int rc = launch_gemm_bf16_dual(g, A2, K2, K1, bias, act, st);
  prof_end(tok, st);
  return rc;
}
