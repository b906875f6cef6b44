// bf16 MFMA GEMM (v_mfma_f32_32x32x16_bf16, f32 accumulate) for the 1x1 convolutions and their
// gradients in the bf16 activation mode.  Same contract as gemm.hip:
//   C[M][N] = sum_k opA(A)[m][k] * opB(B)[n][k] (+bias) with row masking and BN column sums.
//
// 128x128x64 block tile, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).
// Operand staging in LDS depends on which index is contiguous in HBM:
//   K-contiguous  (activations [rows][K], weights [N][K]):  image [128 rows][64 k], 144-byte rows
//                  (9 sixteen-byte slots: odd, so a half-wave's ds_read_b128 is conflict-free);
//   row-contiguous ([K][rows]: the gradients' transposed operands): image [64 k][128 rows],
//                  320-byte rows, fragments fetched with ds_read_b64_tr_b16 (hardware 4x16
//                  transpose), so no operand is ever transposed through HBM or VALU.
// bf16 results leave through an LDS transpose so every global store is 16 B per lane.
#include "gemm.h"

namespace lasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

static constexpr int TM = 128, TN = 128, TK = 64;
static constexpr int LD_KC = 144;            // bytes per row of a K-contiguous image ([row][k])
static constexpr int LD_RC = 320;            // bytes per row of a row-contiguous image ([k][row])
static constexpr int OPER_BYTES = 20480;     // max(128*144, 64*320)
static constexpr int EPI_LD = 144;           // bytes per row of a wave's 64x64 bf16 output image

union Frag { uint4 u; s16x8 s; bf16x8 b; };

// ---- global -> registers: 4 x 16 B per thread per operand tile ------------------------------------
template <bool TRANS>
__device__ __forceinline__ void load_oper(const bf16_t* __restrict__ X, int64_t ld, int64_t R, int64_t r0, int64_t k0,
                                          int64_t kend, bool vec_ok, uint4 (&reg)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = tid + 256 * p;
    int64_t row, col, row_lim, col_lim;
    if (!TRANS) { row = r0 + (c >> 3); col = k0 + ((c & 7) << 3); row_lim = R; col_lim = kend; }   // [r][k]
    else        { row = k0 + (c >> 4); col = r0 + ((c & 15) << 3); row_lim = kend; col_lim = R; }   // [k][r]
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < row_lim) {
      const bf16_t* src = X + row * ld + col;
      if (vec_ok && col + 7 < col_lim) {
        v = *reinterpret_cast<const uint4*>(src);
      } else {
        bf16_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (col + i < col_lim) ? src[i] : (bf16_t)0;
        v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
        v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
      }
    }
    reg[p] = v;
  }
}

template <bool TRANS>
__device__ __forceinline__ void store_oper(char* __restrict__ s, const uint4 (&reg)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = tid + 256 * p;
    const int off = !TRANS ? (c >> 3) * LD_KC + ((c & 7) << 4) : (c >> 4) * LD_RC + ((c & 15) << 4);
    *reinterpret_cast<uint4*>(s + off) = reg[p];
  }
}

// ---- LDS -> MFMA fragment: 8 consecutive k of row (rb + lane&31), k = ks*16 + 8*(lane>>5) + j ----
template <bool TRANS>
__device__ __forceinline__ bf16x8 load_frag(const char* __restrict__ s, int rb, int ks, int lane) {
  Frag f;
  if (!TRANS) {
    f.u = *reinterpret_cast<const uint4*>(s + (rb + (lane & 31)) * LD_KC + ks * 32 + (lane >> 5) * 16);
  } else {
    // 16-lane group g: rows rb + 16*(g&1) .. +15, k-half (g>>1).  Lane 4q+p of the group addresses
    // k-row q, columns 4p..4p+3 of the 4x16 block; it receives column (lane&15), rows 0..3.
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int kbase = ks * 16 + 8 * (g >> 1) + q;
    const int col = rb + 16 * (g & 1) + 4 * p;
    typedef __attribute__((address_space(3))) s16x4 lds_s4;
    const char* a0 = s + kbase * LD_RC + col * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a0 + 4 * LD_RC));
    f.s = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  return f.b;
}

// grid: (ceil(N/TN), ceil(M/TM), split_k)
template <typename TC, bool TRANS_A, bool TRANS_B>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * OPER_BYTES];
  __shared__ float s_stat[2][2][TN];
  char* sA = smem;
  char* sB = smem + OPER_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int64_t m0 = (int64_t)blockIdx.y * TM, n0 = (int64_t)blockIdx.x * TN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  load_oper<TRANS_A>(A, g.lda, g.M, m0, kbeg, kend, g.vecA, ra);
  load_oper<TRANS_B>(B, g.ldb, g.N, n0, kbeg, kend, g.vecB, rb);
  for (int64_t k0 = kbeg; k0 < kend; k0 += TK) {
    __syncthreads();
    store_oper<TRANS_A>(sA, ra);
    store_oper<TRANS_B>(sB, rb);
    __syncthreads();
    if (k0 + TK < kend) {
      load_oper<TRANS_A>(A, g.lda, g.M, m0, k0 + TK, kend, g.vecA, ra);
      load_oper<TRANS_B>(B, g.ldb, g.N, n0, k0 + TK, kend, g.vecB, rb);
    }
#pragma unroll
    for (int ks = 0; ks < TK / 16; ++ks) {
      const bf16x8 a0 = load_frag<TRANS_A>(sA, wm * 64, ks, lane);
      const bf16x8 a1 = load_frag<TRANS_A>(sA, wm * 64 + 32, ks, lane);
      const bf16x8 b0 = load_frag<TRANS_B>(sB, wn * 64, ks, lane);
      const bf16x8 b1 = load_frag<TRANS_B>(sB, wn * 64 + 32, ks, lane);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  const int half = lane >> 5, l31 = lane & 31;
  // acc[mi][ni][r]: row = wm*64 + mi*32 + (r&3) + 8*(r>>2) + 4*half, col = wn*64 + ni*32 + l31
  if (g.split_ws || Elem<TC>::kDtype == LASR_F32) {
    // f32 destinations (split-K slabs, logits, weight gradients): straight from the accumulators,
    // 32 lanes x 4 B = one 128-byte segment per row
    float* W = g.split_ws ? g.split_ws + (int64_t)blockIdx.z * g.M * g.N : reinterpret_cast<float*>(g.C);
    const int64_t ldw = g.split_ws ? g.N : g.ldc;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int64_t n = n0 + wn * 64 + ni * 32 + l31;
      const float bv = (!g.split_ws && g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (m < g.M && n < g.N) {
            float v = acc[mi][ni][r] + bv;
            if (!g.split_ws && g.row_lens) {
              const int64_t b = m / g.rows_per_seq;
              if (m - b * g.rows_per_seq >= g.row_lens[b]) v = 0.f;
            }
            W[m * ldw + n] = v;
          }
        }
    }
    return;
  }

  // ---- bf16 destination: bias, row mask, round, column sums; then LDS transpose and 16-byte stores
  __syncthreads();  // every wave is done reading the operand images
  char* epi = smem + wid * (64 * EPI_LD);
  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int64_t n = n0 + wn * 64 + ni * 32 + l31;
    const bool n_ok = n < g.N;
    const float bv = (g.bias && n_ok) ? g.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int64_t m = m0 + wm * 64 + lr;
        float v = acc[mi][ni][r] + bv;
        if (g.row_lens && m < g.M) {
          const int64_t b = m / g.rows_per_seq;
          if (m - b * g.rows_per_seq >= g.row_lens[b]) v = 0.f;
        }
        const bf16_t q = f32_to_bf16(v);
        if (m < g.M && n_ok) {
          const float sv = bf16_to_f32(q);
          csum[ni] += sv;
          csq[ni] = fmaf(sv, sv, csq[ni]);
        }
        *reinterpret_cast<bf16_t*>(epi + lr * EPI_LD + (ni * 32 + l31) * 2) = q;
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
  }
  __syncthreads();
  {
    // each wave drains its own 64x64 image: lane -> (row = it*8 + lane/8, 8 columns at (lane%8)*8)
    bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
    const int cg = lane & 7, rr = lane >> 3;
    const int64_t n = n0 + wn * 64 + cg * 8;
    const bool vec_ok = (g.ldc % 8 == 0) && (reinterpret_cast<uintptr_t>(C) % 16 == 0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int lr = it * 8 + rr;
      const int64_t m = m0 + wm * 64 + lr;
      if (m < g.M && n < g.N) {
        const uint4 v = *reinterpret_cast<const uint4*>(epi + lr * EPI_LD + cg * 16);
        bf16_t* dst = C + m * g.ldc + n;
        if (vec_ok && n + 7 < g.N) {
          *reinterpret_cast<uint4*>(dst) = v;
        } else {
          const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
          for (int i = 0; i < 8 && n + i < g.N; ++i) dst[i] = e[i];
        }
      }
    }
  }
  if (g.stat_partials && tid < TN) {
    const int64_t n = n0 + tid;
    if (n < g.N) {
      float* P = g.stat_partials + (int64_t)blockIdx.y * 2 * g.N;
      P[n] = s_stat[0][0][tid] + s_stat[1][0][tid];
      P[g.N + n] = s_stat[0][1][tid] + s_stat[1][1][tid];
    }
  }
}

int launch_gemm_bf16(const GemmArgs& g, int dtype_c, int transA, int transB, dim3 grid, hipStream_t st) {
#define LASR_BF16_CASE(TC_, TA_, TB_) hipLaunchKernelGGL((gemm_bf16_kernel<TC_, TA_, TB_>), grid, dim3(256), 0, st, g)
#define LASR_BF16_TC(TC_)                                         \
  do {                                                            \
    if (!transA && !transB) LASR_BF16_CASE(TC_, false, false);    \
    else if (!transA && transB) LASR_BF16_CASE(TC_, false, true); \
    else if (transA && !transB) LASR_BF16_CASE(TC_, true, false); \
    else LASR_BF16_CASE(TC_, true, true);                         \
  } while (0)
  if (dtype_c == LASR_F32) LASR_BF16_TC(float); else LASR_BF16_TC(bf16_t);
#undef LASR_BF16_TC
#undef LASR_BF16_CASE
  LASR_LAUNCH_CHECK("gemm_bf16_kernel");
  return 0;
}

}  // namespace lasr
