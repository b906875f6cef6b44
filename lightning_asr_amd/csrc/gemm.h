// Shared between the f32-MFMA (gemm.hip) and bf16-MFMA (gemm_bf16.hip) GEMM kernels.
#pragma once
#include "common.h"

namespace lasr {

struct GemmArgs {
  const void* A; const void* B; void* C;
  int64_t M, N, K, lda, ldb, ldc;
  const float* bias; const void* addend;
  const int32_t* row_lens; int64_t rows_per_seq;
  float* stat_partials;  // [gridM][2][N] or null
  float* split_ws;       // [split][M][N] or null
  int64_t k_per_split;
  int vecA, vecB;
};

// bf16-MFMA path (gemm_bf16.hip); returns LASR_E_SHAPE-free 0 on launch, or -100 when the
// request needs the f32 path (f32 operands or an addend).
// gz = split-K slices; the launcher picks the tile (128x128 or 256x256) and reports through stat_tiles
// how many row tiles wrote BN partial sums ([tile][2][N] in stat_partials).
int launch_gemm_bf16(const GemmArgs& g, int gz, int dtype_c, int transA, int transB, hipStream_t st, int* stat_tiles);
int launch_gemm_bf16_batch(const GemmArgs* g, const int* gz, int n, int dtype_c, int transA, int transB, hipStream_t st,
                           int* stat_tiles);

// up to 32 split-K weight-gradient problems (transA = transB = 1, f32 slabs in g[i].split_ws), one launch
int launch_gemm_bf16_dual(const GemmArgs& g, const void* A2, int64_t lda2, int64_t K1, const float* bias, int act, hipStream_t st);
int launch_gemm_bf16_multi(const GemmArgs* g, const int* gz, int n, bool big_tile, hipStream_t st);
int launch_gemm_bf16_rowstat(const GemmArgs& g, float* row_stat, int32_t* row_arg, int* n_col_tiles, hipStream_t st);

}  // namespace lasr
