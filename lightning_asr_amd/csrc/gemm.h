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
namespace lstm { struct BwdArgs; }
// lstm_job (optional): the BiLSTM backward recurrences run in the same grid as lstm_wgs workgroups, ahead of the tiles (lstm_body.h)
// riders (optional): reductions that are already complete when this launch is issued; the first *riders_taken of them are done by
// extra workgroups of this launch on the CUs its one round of tiles leaves idle (reduce_body.h) - the caller reduces the rest
int launch_gemm_bf16_multi(const GemmArgs* g, const int* gz, int n, bool big_tile, hipStream_t st, const lstm::BwdArgs* lstm_job = nullptr,
                           int lstm_wgs = 0, const lasr_reduce_desc* riders = nullptr, int n_riders = 0, int* riders_taken = nullptr);
// gemm.hip: lasr_gemm_multi_split_partials with riders (above)
int gemm_multi_split_partials_riders(const lasr_gemm_problem* probs, int n_probs, int split_k, float* const* slabs, int* splits,
                                     const lasr_reduce_desc* riders, int n_riders, int* riders_taken, void* stream);
// gemm.hip: lasr_gemm(split_k > 1, f32 result) without its final sum: the slabs stay at the head of `workspace`, [*splits][M*N] f32
int gemm_split_partials_one(const void* A, const void* B, int dtype_ab, int64_t M, int64_t N, int64_t K, int transA, int transB, int split_k,
                            void* workspace, size_t workspace_bytes, int* splits, void* stream);
// lasr_gemm_multi_split_partials with the context branch's BiLSTM backward recurrences (lstm_body.h) in its grid, and the dW_hh /
// bias-gradient partial sums (rec.pwhh [2][B * kDwZ][G*H], rec.pbias [2][B * kDwZ][G]: the caller reduces them) behind it; returns 1
// without launching anything when the shapes do not take the combined grid (the caller then makes the separate calls).
// Only the first *n_taken problems are launched (as many tiles as finish, in one round of workgroups, about when the recurrences do):
// the caller keeps the rest for its next batched launch.
int gemm_multi_split_partials_with_bilstm_bwd(const lasr_gemm_problem* probs, int n_probs, int* n_taken, int split_k, float* const* slabs, int* splits,
                                              const lstm::BwdArgs& rec, int dtype, int64_t B, void* stream);
int launch_gemm_bf16_rowstat(const GemmArgs& g, float* row_stat, int32_t* row_arg, int* n_col_tiles, hipStream_t st);

}  // namespace lasr
