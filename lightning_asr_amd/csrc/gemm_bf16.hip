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
#include "reduce_body.h"
#include "lstm_body.h"
#include <algorithm>
#include <stdlib.h>

namespace lasr {

// The lane's bias values for its 2 x 4 accumulator quads (columns nbase + ni*32 + 8*j + e), fetched in ONE batch and masked with bit
// operations.  Written as `col < N ? bias[col] : 0.f` inside the epilogue loops, every value became a branch around its load with a
// wait behind it: 32 memory round trips one after the other per tile (and per mi), read off the ISA in round 4.
__device__ __forceinline__ void load_bias_quads(const float* __restrict__ bias, int nbase, int N, float (&bv)[2][4][4]) {
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = nbase + ni * 32 + 8 * j + e;
        bv[ni][j][e] = __uint_as_float(__float_as_uint(bias[min(n, N - 1)]) & (n < N ? 0xffffffffu : 0u));
      }
}


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

// Compact kernel arguments: 32-bit element offsets inside a matrix (host checks M*ld < 2^31).
struct Bf16Args {
  const bf16_t* A; const bf16_t* B; void* C;
  const float* bias; const int32_t* row_lens; float* stat_partials; float* split_ws;
  const bf16_t* addend;   // 128x128 tile, bf16 C only: C = round(A B^T + bias + addend), addend [M][ldc] (N, ldc multiples of 4)
  int M, N, K, lda, ldb, ldc, rows_per_seq, k_per_split;
  int gm, gn, gz;      // tile grid
  int vecA, vecB, vecC;
  // folded eval form (DUAL instantiation only): A = [A | A2] along K, rows of the first part masked by row_lens, act in the epilogue
  const bf16_t* A2; int lda2, K1, act;
  // large-vocabulary decoder (256-row tile form, bf16 C): per (row, column tile) softmax statistics of the STORED values -
  // row_stat [M][gn][2] = (max, sum exp(x - max)), row_arg [M][gn] = first argmax column - so log_softmax never reads the logits
  float* row_stat; int32_t* row_arg;
};
// Up to two independent problems of the same kind in ONE launch (a unit's main + residual 1x1 conv, or
// their two weight gradients): twice the workgroups per launch keeps two per CU resident for the
// split-K weight gradients (256 tiles each) and lets one problem's store tail overlap the other's loads.
struct Bf16Batch { Bf16Args p[2]; int tiles0, total; };

// ---- global -> registers -> LDS: 4 x 16 B per thread per operand tile --------------------------------
// 16-byte chunk c = tid + NT*p of a [ROWS x 64] operand tile.  K-contiguous operand: row c>>3, k (c&7)*8;
// row-contiguous ([k][row]) operand: k = c / (ROWS/8), row (c % (ROWS/8))*8.
//
// Aligned operands (pitch % 8 == 0, 16-byte base: every tensor of the model) take the BRANCH-FREE pair
// below: four unconditional loads from clamped addresses, so nothing but their consumer waits on them
// and a K step's loads stay in flight under the previous step's MFMAs; chunks past the K range are zeroed
// when they are written to LDS (chunks past the M/N edge repeat the last row: those outputs are never
// stored).  Per-lane bounds branches around the loads (the generic path, kept for unaligned shapes)
// make the compiler drain vmcnt at every join.
// 16-byte chunk of 8 bf16 of which the first `nvalid` (any integer; <= 0 none, >= 8 all) are inside the K range
__device__ __forceinline__ uint4 mask_chunk(const uint4& v, int nvalid) {
  const int n = max(0, min(nvalid, 8));
  if (n == 8) return v;                                  // wave-uniform in every aligned-K problem
  auto dm = [n](int i) -> uint32_t { return n >= 2 * i + 2 ? 0xffffffffu : (n == 2 * i + 1 ? 0x0000ffffu : 0u); };
  return make_uint4(v.x & dm(0), v.y & dm(1), v.z & dm(2), v.w & dm(3));
}

template <bool TRANS, int ROWS, int NT, int NCH = 4, bool NTL = false>
__device__ __forceinline__ void load_vec(const bf16_t* __restrict__ X, int ld, int R, int r0, int k0, int kend, uint4 (&reg)[NCH]) {
  constexpr int RCH = ROWS / 8;
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = tid + NT * p;
    uint32_t off;
    if (!TRANS) off = (uint32_t)min(r0 + (c >> 3), R - 1) * (uint32_t)ld + (uint32_t)min(k0 + ((c & 7) << 3), ((kend + 7) & ~7) - 8);
    else off = (uint32_t)min(k0 + c / RCH, kend - 1) * (uint32_t)ld + (uint32_t)min(r0 + ((c % RCH) << 3), ((R + 7) & ~7) - 8);
    reg[p] = Vec<bf16_t>::raw_if_nt<NTL>(X + off);
  }
}

// FULL: the step lies wholly inside the K range - nothing to zero (see store_chunk)
template <bool TRANS, int ROWS, int NT, int LDK_, int LDR_, int NCH = 4, bool FULL = false>
__device__ __forceinline__ void store_vec(char* __restrict__ s, const uint4 (&reg)[NCH], int k0, int kend) {
  constexpr int RCH = ROWS / 8;
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = tid + NT * p;
    const int off = !TRANS ? (c >> 3) * LDK_ + ((c & 7) << 4) : (c / RCH) * LDR_ + ((c % RCH) << 4);
    if constexpr (FULL) {
      *reinterpret_cast<uint4*>(s + off) = reg[p];
    } else {
      // elements past the K range become zero: whole k-rows of a row-contiguous operand, the tail elements of a
      // K-contiguous chunk (K need not be a multiple of 8; the pitch is, so the chunk itself is in bounds)
      const int nvalid = !TRANS ? kend - (k0 + ((c & 7) << 3)) : (k0 + c / RCH < kend ? 8 : 0);
      *reinterpret_cast<uint4*>(s + off) = mask_chunk(reg[p], nvalid);
    }
  }
}

// single-chunk forms (chunk p of the thread), for K loops that spread the staging between MFMA groups
template <bool TRANS, int ROWS, int NT, bool NTL = false>
__device__ __forceinline__ uint4 load_chunk(const bf16_t* __restrict__ X, int ld, int R, int r0, int k0, int kend, int p) {
  constexpr int RCH = ROWS / 8;
  const int c = threadIdx.x + NT * p;
  uint32_t off;
  if (!TRANS) off = (uint32_t)min(r0 + (c >> 3), R - 1) * (uint32_t)ld + (uint32_t)min(k0 + ((c & 7) << 3), ((kend + 7) & ~7) - 8);
  else off = (uint32_t)min(k0 + c / RCH, kend - 1) * (uint32_t)ld + (uint32_t)min(r0 + ((c % RCH) << 3), ((R + 7) & ~7) - 8);
  return Vec<bf16_t>::raw_if_nt<NTL>(X + off);
}
// FULL: the K step that is staged lies wholly inside the K range (every step but a slice's last): no element to zero - the per-chunk
// masks (compare, selects, four ANDs, a divergent branch for the partial chunk) cost ~60 VALU instructions and three exec-mask branches
// per K step and wave in the steady-state loop, where they can never apply (round 5)
template <bool TRANS, int ROWS, int NT, int LDK_, int LDR_, bool FULL = false>
__device__ __forceinline__ void store_chunk(char* __restrict__ s, const uint4& v, int k0, int kend, int p) {
  constexpr int RCH = ROWS / 8;
  const int c = threadIdx.x + NT * p;
  const int off = !TRANS ? (c >> 3) * LDK_ + ((c & 7) << 4) : (c / RCH) * LDR_ + ((c % RCH) << 4);
  if constexpr (FULL) {
    *reinterpret_cast<uint4*>(s + off) = v;
  } else {
    const int nvalid = !TRANS ? kend - (k0 + ((c & 7) << 3)) : (k0 + c / RCH < kend ? 8 : 0);
    *reinterpret_cast<uint4*>(s + off) = mask_chunk(v, nvalid);
  }
}

// generic path: any pitch/alignment, element-granular edges
template <bool TRANS>
__device__ __forceinline__ void load_oper(const bf16_t* __restrict__ X, int ld, int R, int r0, int k0, int kend, bool vec_ok,
                                          uint4 (&reg)[4]) {
  const int tid = threadIdx.x;
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    const int c = tid + 256 * p;
    int row, col, row_lim, col_lim;
    if (!TRANS) { row = r0 + (c >> 3); col = k0 + ((c & 7) << 3); row_lim = R; col_lim = kend; }   // [r][k]
    else        { row = k0 + (c >> 4); col = r0 + ((c & 15) << 3); row_lim = kend; col_lim = R; }   // [k][r]
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < row_lim && col < col_lim) {
      const bf16_t* src = X + (uint32_t)row * (uint32_t)ld + (uint32_t)col;
      if (vec_ok && col + 7 < col_lim) {
        v = *reinterpret_cast<const uint4*>(src);
      } else {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        for (int i = 0; i < 8; ++i)
          if (col + i < col_lim) w[i >> 1] |= (uint32_t)src[i] << (16 * (i & 1));
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    if (p == 0) reg[0] = v; else if (p == 1) reg[1] = v; else if (p == 2) reg[2] = v; else reg[3] = v;
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
template <bool TRANS, int LD_RC_ = LD_RC>
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
    const char* a0 = s + kbase * LD_RC_ + col * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a0 + 4 * LD_RC_));
    f.s = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  return f.b;
}

// 1-D grid of gm*gn*gz tiles.  Workgroups are dealt round-robin to the 8 XCDs (b and b+8 share an
// XCD's L2), so the linear id is remapped (bijectively) to give every XCD a contiguous run of tiles
// in (z, m, n) order: the gn tiles that re-read one 128-row A panel, and the tiles of one split-K
// slice, then hit the same L2 instead of fetching the panel from HBM once per XCD.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// two f32 -> one dword of two bf16 (RNE): a single v_cvt_pk_bf16_f32
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const pk_f32x2 f = {lo, hi};
  const pk_bf16x2 h = __builtin_convertvector(f, pk_bf16x2);
  return __builtin_bit_cast(uint32_t, h);
}
// N edge / unaligned C: element stores of the first n (1..8) values of v.  Out of line: the aligned path is the product path.
__device__ __noinline__ void store_bf16_tail(bf16_t* dst, uint4 v, int n) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < n) dst[i] = (bf16_t)(w[i >> 1] >> (16 * (i & 1)));
}

// one 128x128 output tile (tile `lid` of problem g); shared by the two-problem and the many-problem kernels
template <typename TC, bool TRANS_A, bool TRANS_B, bool VEC>
__device__ __forceinline__ void gemm_bf16_tile(const Bf16Args& g, const int lid) {
  __shared__ __attribute__((aligned(16))) char smem[2 * OPER_BYTES];
  __shared__ float s_stat[2][2][TN];
  __shared__ float s_keep[TM];
  char* sA = smem;
  char* sB = smem + OPER_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tn = lid % g.gn, tm = (lid / g.gn) % g.gm, tz = lid / (g.gn * g.gm);
  const int m0 = tm * TM, n0 = tn * TN;
  const int kbeg = tz * g.k_per_split;
  const int kend = min(kbeg + g.k_per_split, g.K);

  // 1.0 for output rows that are stored as computed, 0.0 for rows past the matrix edge or past their
  // utterance's length (MaskCNN): one integer division per ROW here, none in the epilogue.
  if (tid < TM) {
    const int m = m0 + tid;
    bool keep = m < g.M;
    if (keep && g.row_lens) {
      const int b = m / g.rows_per_seq;
      keep = (m - b * g.rows_per_seq) < g.row_lens[b];
    }
    s_keep[tid] = keep ? 1.f : 0.f;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two register sets: the global loads of K-tile i+2 are issued while tile i is multiplied, so a
  // load has two full iterations to land (one CU holds 2 workgroups x 64 KB in flight).  With K of
  // only 256-1024 the kernel is a short dependent chain of load latencies, not a bandwidth stream.
  uint4 ra0[4], rb0[4], ra1[4], rb1[4];
#define LASR_LOAD(RA_, RB_, K0_)                                                          \
  if constexpr (VEC) {                                                                   \
    load_vec<TRANS_A, TM, 256>(g.A, g.lda, g.M, m0, (K0_), kend, RA_);                   \
    load_vec<TRANS_B, TN, 256>(g.B, g.ldb, g.N, n0, (K0_), kend, RB_);                   \
  } else {                                                                               \
    load_oper<TRANS_A>(g.A, g.lda, g.M, m0, (K0_), kend, g.vecA, RA_);                   \
    load_oper<TRANS_B>(g.B, g.ldb, g.N, n0, (K0_), kend, g.vecB, RB_);                   \
  }
#define LASR_STORE(RA_, RB_, K0_)                                                         \
  if constexpr (VEC) {                                                                   \
    if ((K0_) + TK <= kend) {             /* workgroup-uniform: an interior step, no K-tail masks */ \
      store_vec<TRANS_A, TM, 256, LD_KC, LD_RC, 4, true>(sA, RA_, (K0_), kend);          \
      store_vec<TRANS_B, TN, 256, LD_KC, LD_RC, 4, true>(sB, RB_, (K0_), kend);          \
    } else {                                                                             \
      store_vec<TRANS_A, TM, 256, LD_KC, LD_RC>(sA, RA_, (K0_), kend);                   \
      store_vec<TRANS_B, TN, 256, LD_KC, LD_RC>(sB, RB_, (K0_), kend);                   \
    }                                                                                    \
  } else {                                                                               \
    store_oper<TRANS_A>(sA, RA_);                                                        \
    store_oper<TRANS_B>(sB, RB_);                                                        \
  }
  LASR_LOAD(ra0, rb0, kbeg)
  if (kbeg + TK < kend) { LASR_LOAD(ra1, rb1, kbeg + TK) }
#define LASR_K_STEP(RA_, RB_, KCUR_, KNEXT_)                                              \
  {                                                                                      \
    __syncthreads();                                                                     \
    LASR_STORE(RA_, RB_, KCUR_)                                                          \
    __syncthreads();                                                                     \
    if ((KNEXT_) < kend) { LASR_LOAD(RA_, RB_, KNEXT_) }                                  \
    _Pragma("unroll") for (int ks = 0; ks < TK / 16; ++ks) {                             \
      const bf16x8 a0 = load_frag<TRANS_A>(sA, wm * 64, ks, lane);                       \
      const bf16x8 a1 = load_frag<TRANS_A>(sA, wm * 64 + 32, ks, lane);                  \
      const bf16x8 b0 = load_frag<TRANS_B>(sB, wn * 64, ks, lane);                       \
      const bf16x8 b1 = load_frag<TRANS_B>(sB, wn * 64 + 32, ks, lane);                  \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);   \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[0][1], 0, 0, 0);   \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[1][0], 0, 0, 0);   \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);   \
    }                                                                                    \
  }
  for (int k0 = kbeg; k0 < kend; k0 += 2 * TK) {
    LASR_K_STEP(ra0, rb0, k0, k0 + 2 * TK)
    if (k0 + TK < kend) LASR_K_STEP(ra1, rb1, k0 + TK, k0 + 3 * TK)
  }
#undef LASR_LOAD
#undef LASR_STORE
#undef LASR_K_STEP

  const int half = lane >> 5, l31 = lane & 31;
  // The MFMA operands are swapped (D = B_tile * A_tile^T), so the tile sits TRANSPOSED in the accumulators:
  // acc[mi][ni][r]: row = wm*64 + mi*32 + l31, col = wn*64 + ni*32 + 8*(r>>2) + 4*half + (r&3) - a lane owns one output
  // row per MFMA tile and 4 consecutive columns per register quad: 16-byte f32 stores, or packed bf16 pairs and 8-byte
  // LDS writes (the natural order left one 4-byte store / one 2-byte LDS write per element).
  if constexpr (Elem<TC>::kDtype == LASR_F32) {
    // f32 destinations (split-K slabs, logits, weight gradients): straight from the accumulators
    float* W = g.split_ws ? g.split_ws + (size_t)tz * (size_t)g.M * (size_t)g.N : reinterpret_cast<float*>(g.C);
    const int ldw = g.split_ws ? g.N : g.ldc;
    const bool vec4 = (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    const bool plain = g.split_ws != nullptr;                 // slabs: no bias, no mask
    const bool has_bias = !plain && g.bias != nullptr;      // workgroup-uniform
    float bv[2][4][4];
    if (has_bias) load_bias_quads(g.bias, n0 + wn * 64 + 4 * half, g.N, bv);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int lr = wm * 64 + mi * 32 + l31;
      const int m = m0 + lr;
      const float kf = plain ? 1.f : s_keep[lr];
      if (m < g.M) {
        float* wrow = W + (uint32_t)m * (uint32_t)ldw;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + ni * 32 + 8 * j + 4 * half;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[mi][ni][4 * j + e];
              if (!plain) v[e] = (v[e] + (has_bias ? bv[ni][j][e] : 0.f)) * kf;
            }
            if (vec4 && n + 3 < g.N) {
              *reinterpret_cast<float4*>(wrow + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < g.N) wrow[n + e] = v[e];
            }
          }
        }
      }
    }
  } else {
    // ---- bf16 destination: bias, row mask, round; per-wave 64 x 64 LDS image; read back as 16-byte vectors for the
    //      column sums (statistics of the values as stored) and the stores (8 rows x 128 B per instruction)
    // addend (the accumulating data-gradient GEMMs of the context branch): 4 consecutive bf16 per register quad, all 16 loads
    // of the lane issued before the barrier
    uint2 adv[2][2][4];
    const bool has_add = g.addend != nullptr;                 // workgroup-uniform
    if (has_add) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const uint32_t mrow = (uint32_t)min(m0 + wm * 64 + mi * 32 + l31, g.M - 1) * (uint32_t)g.ldc;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            adv[mi][ni][j] = *reinterpret_cast<const uint2*>(g.addend + mrow + (uint32_t)min(n0 + wn * 64 + ni * 32 + 8 * j + 4 * half, g.N - 4));
      }
    }
    __syncthreads();  // every wave is done reading the operand images
    char* epi = smem + wid * (64 * EPI_LD);
    const bool has_bias = g.bias != nullptr;
    float bv[2][4][4];
    if (has_bias) load_bias_quads(g.bias, n0 + wn * 64 + 4 * half, g.N, bv);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const uint32_t km = s_keep[wm * 64 + mi * 32 + l31] != 0.f ? 0xffffffffu : 0u;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * j + e];
          if (has_add) {
            const uint2 a = adv[mi][ni][j];
            v[0] += __uint_as_float(a.x << 16); v[1] += __uint_as_float(a.x & 0xffff0000u);
            v[2] += __uint_as_float(a.y << 16); v[3] += __uint_as_float(a.y & 0xffff0000u);
          }
          if (has_bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[ni][j][e];
          }
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]) & km;
          pk.y = pack_bf16x2(v[2], v[3]) & km;
          *reinterpret_cast<uint2*>(epi + (mi * 32 + l31) * EPI_LD + (ni * 32 + 8 * j + 4 * half) * 2) = pk;
        }
      }
    }
    __syncthreads();
    {
      // each wave drains its own 64x64 image: lane -> (row = it*8 + lane/8, 8 columns at (lane%8)*8)
      bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
      const int cg = lane & 7, rr = lane >> 3;
      const int n = n0 + wn * 64 + cg * 8;
      const bool full_n = g.vecC && n + 7 < g.N;
      const bool want_stats = g.stat_partials != nullptr;
      float cs[8], cq[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { cs[i] = 0.f; cq[i] = 0.f; }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int lr = it * 8 + rr;
        const int m = m0 + wm * 64 + lr;
        const uint4 v = *reinterpret_cast<const uint4*>(epi + lr * EPI_LD + cg * 16);
        if (want_stats) {                          // rows past M / past the utterance hold zeros
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x0 = __uint_as_float(w[i] << 16), x1 = __uint_as_float(w[i] & 0xffff0000u);
            cs[2 * i] += x0; cq[2 * i] = fmaf(x0, x0, cq[2 * i]);
            cs[2 * i + 1] += x1; cq[2 * i + 1] = fmaf(x1, x1, cq[2 * i + 1]);
          }
        }
        if (m < g.M && n < g.N) {
          bf16_t* dst = C + (uint32_t)m * (uint32_t)g.ldc + (uint32_t)n;
          if (full_n) *reinterpret_cast<uint4*>(dst) = v;
          else store_bf16_tail(dst, v, min(g.N - n, 8));
        }
      }
      if (want_stats) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int d = 8; d < 64; d <<= 1) {
            cs[i] += __shfl_xor(cs[i], d, 64);
            cq[i] += __shfl_xor(cq[i], d, 64);
          }
        }
        if (rr == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            s_stat[wm][0][wn * 64 + cg * 8 + i] = cs[i];
            s_stat[wm][1][wn * 64 + cg * 8 + i] = cq[i];
          }
        }
      }
    }
    if (g.stat_partials) {
      __syncthreads();
      if (tid < TN) {
        const int n = n0 + tid;
        if (n < g.N) {
          float* P = g.stat_partials + (size_t)tm * 2 * g.N;
          P[n] = s_stat[0][0][tid] + s_stat[1][0][tid];
          P[g.N + n] = s_stat[0][1][tid] + s_stat[1][1][tid];
        }
      }
    }
  }
}

template <typename TC, bool TRANS_A, bool TRANS_B, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(Bf16Batch gb) {
  const int lid_all = xcd_remap(blockIdx.x, gb.total);
  const bool second = lid_all >= gb.tiles0;              // workgroup-uniform
  const Bf16Args& g = second ? gb.p[1] : gb.p[0];
  gemm_bf16_tile<TC, TRANS_A, TRANS_B, VEC>(g, second ? lid_all - gb.tiles0 : lid_all);
}

// Up to 32 problems in one launch: the 1x1 weight gradients of a whole backward stage.  Per unit their split-K
// launch is 16 K steps per workgroup between a prologue and a 32 MB partial-slab epilogue; batched, a stage needs
// a third of the slices for the same number of workgroups, so slices are 3x longer and the slabs 3x smaller.
static constexpr int kMaxMulti = 32;
// riders (round 5): reductions of the stage that are already COMPLETE when the weight-gradient launch is issued (the depthwise
// weight gradients' per-utterance partials, 64 MB of lasr_reduce_many's 123 MB at cfg2) run as n_rwg extra workgroups behind the tiles -
// on the CUs the one round of tiles leaves idle (71 tiles x 3 slices = 213 of 256)
static constexpr int kMaxRiders = 24;
struct Bf16Multi { Bf16Args p[kMaxMulti]; int start[kMaxMulti + 1]; int n, total; lasr_reduce_desc rd[kMaxRiders]; int n_rd, n_rwg; };
template <bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_multi_kernel(Bf16Multi gm) {
  const int lid_all = xcd_remap(blockIdx.x, gm.total);
  int i = 0;
  while (i + 1 < gm.n && gm.start[i + 1] <= lid_all) ++i;   // workgroup-uniform scan of at most 32 entries
  gemm_bf16_tile<float, true, true, VEC>(gm.p[i], lid_all - gm.start[i]);
}

// =====================================================================================================
// 256x256x64 tile, 512 threads (8 waves as 2 x 4, each 128 x 64 = 4 x 2 MFMA tiles, 128 accumulator
// VGPRs), ONE workgroup per CU, operand images double-buffered in LDS (2 x 72 KB) with one barrier per
// K step.  Against the 128x128 tile it halves the operand bytes pulled from L2 and written to LDS per
// FLOP and reads 0.75 instead of 1 fragment per MFMA; a unit's main + residual problem (63 x 2 tiles
// each at N=512) fill 252 of the 256 CUs in a single round.  bf16 results only.
namespace big {
static constexpr int BTM = 256, NT = 512;
static constexpr int LDR = 576;               // [k][256 rows] image: 512 B + 64 B pad (bank residue 16 dwords, as LD_RC)
static constexpr int OPER = 36864;            // A image: 256*144 == 64*576
// Two tile widths.  WIDE: 256x256, 8 waves as 2 x 4, wave tile 128x64 (4 x 2 MFMA tiles).  NARROW: 256x128, 8 waves as
// 4 x 2, wave tile 64x64 (2 x 2): for problems with N <= 256 (the 256-channel blocks), where 256-wide tiles would leave
// half of the CUs without a workgroup (63 x 1 tiles per problem).
template <bool NARROW>
struct Cfg {
  static constexpr int BN = NARROW ? 128 : 256;
  static constexpr int WN = NARROW ? 2 : 4;             // waves along N (8 / WN along M)
  static constexpr int MI = NARROW ? 2 : 4;             // 32-row MFMA tiles per wave
  static constexpr int NCB = BN * 8 / NT;               // 16-byte B chunks per thread and K step: 2 / 4
  static constexpr int LDRB = BN * 2 + 64;              // pitch of a row-contiguous B image: 320 / 576
  static constexpr int OPER_B = (BN * LD_KC > 64 * LDRB) ? BN * LD_KC : 64 * LDRB;   // 20480 / 36864
  static constexpr int BUF = OPER + OPER_B;             // A + B image of one K step
  static constexpr int EPB = BN * 2 + 16;               // output image pitch: 272 / 528
};
}  // namespace big

// One K step of the 256x256 tile: 4 groups of (6 fragment reads, 8 MFMAs); after group ks the thread moves
// its chunk ks of the NEXT step from registers into the other LDS image and (re)issues the global load of
// the step after that into the same registers, so LDS writes and VMEM issue ride in the MFMA shadows
// instead of forming a separate all-waves staging phase in front of them.
// A operand of the folded eval form: K range [0, K1) from A (rows past their utterance's length read as zero: km),
// [K1, K) from A2; K1 and K - K1 are multiples of the K step, so a step lies in one operand
__device__ __forceinline__ uint4 load_chunk_dual(const Bf16Args& g, int m0, int k0, int p, const uint32_t (&kma)[4]) {
  const int c = threadIdx.x + big::NT * p;
  const bool first = k0 < g.K1;                        // workgroup-uniform
  const bf16_t* X = first ? g.A : g.A2;
  const int ld = first ? g.lda : g.lda2;
  const int kk = (first ? k0 : k0 - g.K1) + ((c & 7) << 3);
  (void)kma;   // RAW: the row mask is applied when the chunk is staged (dual_row_mask) - masked here, the load issued for the step
               // after next was waited for on the spot instead of under a K step of MFMAs (round 4)
  return *reinterpret_cast<const uint4*>(X + (uint32_t)min(m0 + (c >> 3), g.M - 1) * (uint32_t)ld + (uint32_t)kk);
}
// the folded eval form's row mask for chunk p of the K step that starts at k0 (first operand only)
__device__ __forceinline__ uint4 dual_row_mask(const Bf16Args& g, uint4 v, int k0, int p, const uint32_t (&kma)[4]) {
  const uint32_t mk = k0 < g.K1 ? kma[p] : 0xffffffffu;   // workgroup-uniform choice
  return make_uint4(v.x & mk, v.y & mk, v.z & mk, v.w & mk);
}

// FULLST: the step being staged (k_store) is not the slice's last one (store_chunk<FULL>)
template <bool TRANS_A, bool TRANS_B, bool NARROW, bool STORE, bool LOAD, bool DUAL = false, bool NTL = false, bool FULLST = false>
__device__ __forceinline__ void big_step(const char* __restrict__ sA, const char* __restrict__ sB, char* __restrict__ dA,
                                         char* __restrict__ dB, f32x16 (&acc)[big::Cfg<NARROW>::MI][2], uint4 (&ra)[4],
                                         uint4 (&rb)[big::Cfg<NARROW>::NCB], const Bf16Args& g, int m0, int n0, int k_store,
                                         int k_load, int kend, int wm, int wn, int lane, const uint32_t (&kma)[4]) {
  using namespace big;
  using CF = Cfg<NARROW>;
  constexpr int MI = CF::MI;
  // fragments of group ks+1 are requested before the MFMAs of group ks are issued (two register sets), so a
  // wave's MFMA stream does not stop for its own LDS latency; only the first group after the barrier waits
  bf16x8 a[2][MI], b[2][2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) b[0][ni] = load_frag<TRANS_B, CF::LDRB>(sB, wn * 64 + ni * 32, 0, lane);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) a[0][mi] = load_frag<TRANS_A, LDR>(sA, wm * (32 * MI) + mi * 32, 0, lane);
#pragma unroll
  for (int ks = 0; ks < TK / 16; ++ks) {
    const int cur = ks & 1, nxt = cur ^ 1;
    if (ks + 1 < TK / 16) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[nxt][ni] = load_frag<TRANS_B, CF::LDRB>(sB, wn * 64 + ni * 32, ks + 1, lane);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[nxt][mi] = load_frag<TRANS_A, LDR>(sA, wm * (32 * MI) + mi * 32, ks + 1, lane);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      // operands swapped: D = B_tile * A_tile^T, i.e. the tile TRANSPOSED in the accumulators - lane (l31, half) holds output
      // row l31 and, per register quad, 4 CONSECUTIVE output columns 8*(r>>2) + 4*half + (r&3): the epilogue packs them into
      // 8-byte row-major LDS writes (the natural order left 2-byte writes: 128 ds_write_b16 per lane, 9 us per launch)
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[cur][ni], a[cur][mi], acc[mi][ni], 0, 0, 0);
    if constexpr (STORE) {
      if constexpr (DUAL) store_chunk<TRANS_A, BTM, NT, LD_KC, LDR, FULLST>(dA, dual_row_mask(g, ra[ks], k_store, ks, kma), k_store, kend, ks);
      else store_chunk<TRANS_A, BTM, NT, LD_KC, LDR, FULLST>(dA, ra[ks], k_store, kend, ks);
      if constexpr (CF::NCB == 4) store_chunk<TRANS_B, CF::BN, NT, LD_KC, CF::LDRB, FULLST>(dB, rb[ks], k_store, kend, ks);
      else if (ks < CF::NCB) store_chunk<TRANS_B, CF::BN, NT, LD_KC, CF::LDRB, FULLST>(dB, rb[ks < CF::NCB ? ks : 0], k_store, kend, ks);
    }
    if constexpr (LOAD) {
      if constexpr (DUAL) ra[ks] = load_chunk_dual(g, m0, k_load, ks, kma);
      else ra[ks] = load_chunk<TRANS_A, BTM, NT, NTL>(g.A, g.lda, g.M, m0, k_load, kend, ks);
      if constexpr (CF::NCB == 4) rb[ks] = load_chunk<TRANS_B, CF::BN, NT, NTL>(g.B, g.ldb, g.N, n0, k_load, kend, ks);
      else if (ks < CF::NCB) rb[ks < CF::NCB ? ks : 0] = load_chunk<TRANS_B, CF::BN, NT, NTL>(g.B, g.ldb, g.N, n0, k_load, kend, ks);
    }
  }
}

#ifdef LASR_GEMM_STAMPS
// Debug builds only (never in liblasr.so): per-workgroup wall-clock stamps, 8 x u64 per workgroup, for the
// first 4096 workgroups of a launch.  The buffer is set by lasr_debug_set_gemm_stamps (tools/gemm_stamps.py).
__device__ unsigned long long* g_stamps = nullptr;
#define LASR_STAMP(i_) do { if (stamps && threadIdx.x == 0 && blockIdx.x < 4096) stamps[blockIdx.x * 8 + (i_)] = wall_clock64(); } while (0)
// slots 6, 7: shader cycles (s_memtime) of the K loop, and of them the cycles wave 0 stood at the loop's barriers
#else
#define LASR_STAMP(i_) do {} while (0)
#endif

// one 256 x 256 (256 x 128) output tile `lid` of problem g; SLAB: split-K slice into the f32 slab g.split_ws (no LDS image,
// no statistics) - shared by the two-problem kernel and the many-problem weight-gradient kernel
template <bool TRANS_A, bool TRANS_B, bool NARROW, bool SLAB, bool DUAL = false, bool NTL = false>
__device__ __forceinline__ void gemm_bf16_big_tile(const Bf16Args& g, const int lid) {
#ifdef LASR_GEMM_STAMPS
  unsigned long long* stamps = g_stamps;
#endif
  LASR_STAMP(0);
  using big::BTM; using big::NT; using big::LDR; using big::OPER;
  using CF = big::Cfg<NARROW>;
  constexpr int BTN = CF::BN, BUF = CF::BUF, MI = CF::MI;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  __shared__ float s_keep[BTM];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / CF::WN, wn = wid % CF::WN;
  const int tn = lid % g.gn, tm = (lid / g.gn) % g.gm, tz = lid / (g.gn * g.gm);
  const int m0 = tm * BTM, n0 = tn * BTN;
  const int kbeg = tz * g.k_per_split;
  const int kend = min(kbeg + g.k_per_split, g.K);
  const int nk = (kend - kbeg + TK - 1) / TK;

  // MaskCNN row mask of the epilogue: the utterance's length word is REQUESTED here, in front of the first tile loads, and consumed
  // behind them (below).  Written in one piece here, the load sat in a divergent branch with a wait at its join: one memory round trip
  // before the first operand load of every masked launch (round 4, read off the ISA).  Unconditional load from an address that is
  // valid either way; the folded eval form masks the first A operand's rows instead.
  const bool row_masked = !DUAL && g.row_lens != nullptr;   // workgroup-uniform
  int keep_pos, keep_len;
  {
    const int m = min(m0 + (tid & (BTM - 1)), g.M - 1);
    const int b = m / g.rows_per_seq;
    keep_pos = m - b * g.rows_per_seq;
    const int32_t* lp = row_masked ? g.row_lens + b : reinterpret_cast<const int32_t*>(g.B);
    keep_len = *lp;
  }

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[CF::NCB];
  uint32_t kma[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};   // DUAL: row masks of this thread's four A chunks
  if constexpr (DUAL) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(m0 + ((tid + NT * p) >> 3), g.M - 1);
      if (g.row_lens) {
        const int b = m / g.rows_per_seq;
        kma[p] = (m - b * g.rows_per_seq) < g.row_lens[b] ? 0xffffffffu : 0u;
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = load_chunk_dual(g, m0, kbeg, p, kma);
  } else {
    load_vec<TRANS_A, BTM, NT, 4, NTL>(g.A, g.lda, g.M, m0, kbeg, kend, ra);
  }
  load_vec<TRANS_B, BTN, NT, CF::NCB, NTL>(g.B, g.ldb, g.N, n0, kbeg, kend, rb);
  if constexpr (DUAL) {
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = dual_row_mask(g, ra[p], kbeg, p, kma);
  }
  store_vec<TRANS_A, BTM, NT, LD_KC, LDR>(smem, ra, kbeg, kend);
  store_vec<TRANS_B, BTN, NT, LD_KC, CF::LDRB, CF::NCB>(smem + OPER, rb, kbeg, kend);
  if (tid < BTM) s_keep[tid] = (m0 + tid < g.M && (!row_masked || keep_pos < keep_len)) ? 1.f : 0.f;   // (the length word arrived before the tile did)
  if (nk > 1) {
    if constexpr (DUAL) {
#pragma unroll
      for (int p = 0; p < 4; ++p) ra[p] = load_chunk_dual(g, m0, kbeg + TK, p, kma);
    } else {
      load_vec<TRANS_A, BTM, NT, 4, NTL>(g.A, g.lda, g.M, m0, kbeg + TK, kend, ra);
    }
    load_vec<TRANS_B, BTN, NT, CF::NCB, NTL>(g.B, g.ldb, g.N, n0, kbeg + TK, kend, rb);
  }
  __syncthreads();
  LASR_STAMP(1);
#ifdef LASR_GEMM_STAMPS
  unsigned long long kstamp_barrier = 0;                   // shader cycles this wave stood at the loop's barriers
  const unsigned long long kstamp_c0 = __builtin_readcyclecounter();
#endif
  {
    // steady state: compute step it, stage step it+1 (registers -> other image), fetch step it+2
    int it = 0;
    for (; it + 2 < nk; ++it) {
      const char* sA = smem + (it & 1) * BUF;
      char* dA = smem + ((it + 1) & 1) * BUF;
      // (it + 2 < nk: the step staged here, it + 1, is never the last one)
      big_step<TRANS_A, TRANS_B, NARROW, true, true, DUAL, NTL, true>(sA, sA + OPER, dA, dA + OPER, acc, ra, rb, g, m0, n0, kbeg + (it + 1) * TK,
                                             kbeg + (it + 2) * TK, kend, wm, wn, lane, kma);
#ifdef LASR_GEMM_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long tb0 = __builtin_readcyclecounter();
      __syncthreads();
      kstamp_barrier += __builtin_readcyclecounter() - tb0;
#else
      __syncthreads();
#endif
    }
    if (it + 1 < nk) {
      const char* sA = smem + (it & 1) * BUF;
      char* dA = smem + ((it + 1) & 1) * BUF;
      big_step<TRANS_A, TRANS_B, NARROW, true, false, DUAL>(sA, sA + OPER, dA, dA + OPER, acc, ra, rb, g, m0, n0, kbeg + (it + 1) * TK, 0, kend,
                                              wm, wn, lane, kma);
      __syncthreads();
      ++it;
    }
    const char* sA = smem + (it & 1) * BUF;
    big_step<TRANS_A, TRANS_B, NARROW, false, false, DUAL>(sA, sA + OPER, nullptr, nullptr, acc, ra, rb, g, m0, n0, 0, 0, kend, wm, wn, lane, kma);
    __syncthreads();
  }
  LASR_STAMP(2);
#ifdef LASR_GEMM_STAMPS
  if (stamps && threadIdx.x == 0 && blockIdx.x < 4096) {
    stamps[blockIdx.x * 8 + 6] = __builtin_readcyclecounter() - kstamp_c0;
    stamps[blockIdx.x * 8 + 7] = kstamp_barrier;
  }
#endif
  // ---- epilogue: bias, row mask, bf16 rounding, BN column sums.  The whole tile is laid out row-major in LDS as
  //      bf16 (528 / 272-byte rows over the operand images, free after the last barrier): a lane owns output row l31
  //      of each of its MFMA tiles and 4 consecutive columns per register quad -> one packed conversion per pair and
  //      one 8-byte LDS write per quad (conflict-free: pitch = 4 dwords mod 32).  Then every wave reads 32 complete
  //      tile rows back as 16-byte vectors, adds them into its column sums (the statistics are of the values as
  //      stored) and stores them: one instruction = 2 rows x 512 contiguous bytes.
  //      (Measured on the 16 032 x 512 x 512 launch: K loop done at 16.8 us; the epilogue cost 21 us with per-element
  //      conversion, statistics and 2-byte LDS writes in the natural accumulator order, of which the global stores
  //      themselves were 1 us.)
  constexpr int EPB = CF::EPB;
  const int half = lane >> 5, l31 = lane & 31;
  const int nb = n0 + wn * 64;
  if constexpr (SLAB) {
    // split-K slice: f32 straight from the accumulators, 16 bytes (4 consecutive columns) per lane and quad; a 128-byte
    // line of the slab is completed by four instructions of the same wave (L2 merges them; the slab is written once)
    float* W = g.split_ws + (size_t)tz * (size_t)g.M * (size_t)g.N;
    const bool vec4 = (g.N & 3) == 0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * (32 * MI) + mi * 32 + l31;
      if (m < g.M) {
        float* wrow = W + (size_t)m * (size_t)g.N;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = nb + ni * 32 + 8 * j + 4 * half;
            if (vec4 && n + 3 < g.N) {
              *reinterpret_cast<float4*>(wrow + n) = make_float4(acc[mi][ni][4 * j], acc[mi][ni][4 * j + 1], acc[mi][ni][4 * j + 2], acc[mi][ni][4 * j + 3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < g.N) wrow[n + e] = acc[mi][ni][4 * j + e];
            }
          }
        }
      }
    }
    return;
  }
  bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
  {
    char* img = smem + (wm * (32 * MI) + l31) * EPB + (wn * 64 + 4 * half) * 2;
    const bool has_bias = g.bias != nullptr;               // workgroup-uniform
    float bv[2][4][4];
    if (has_bias) load_bias_quads(g.bias, nb + 4 * half, g.N, bv);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const uint32_t km = s_keep[wm * (32 * MI) + mi * 32 + l31] != 0.f ? 0xffffffffu : 0u;   // MaskCNN / M edge: the row is stored as zeros
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * j + e];
          if (has_bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[ni][j][e];
          }
          if constexpr (DUAL) {                             // folded eval form: activation in the epilogue
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (g.act == LASR_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
              else if (g.act == LASR_ACT_SWISH) v[e] = v[e] / (1.f + __expf(-v[e]));
            }
          }
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]) & km;
          pk.y = pack_bf16x2(v[2], v[3]) & km;
          *reinterpret_cast<uint2*>(img + mi * 32 * EPB + (ni * 32 + 8 * j) * 2) = pk;
        }
      }
    }
  }
  __syncthreads();
  LASR_STAMP(5);
  // every wave stores 32 complete tile rows; a row is BTN*2 bytes = LPR lanes x 16 B, RPI rows per instruction
  constexpr int LPR = BTN / 8, RPI = 64 / LPR, NIT = 32 / RPI;
  const int lc = lane % LPR, lrow = lane / LPR;
  const int nst = n0 + lc * 8;
  const bool full_n = g.vecC && nst + 7 < g.N;
  const bool want_stats = g.stat_partials != nullptr;      // workgroup-uniform
  const bool want_rowstat = g.row_stat != nullptr;         // workgroup-uniform
  float cs[8], cq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { cs[i] = 0.f; cq[i] = 0.f; }
  // all of the lane's row chunks are read back first: the loop below has uniform branches (statistics or not, row statistics or not),
  // i.e. basic blocks, and with the read inside it every row paid its own LDS round trip before its store was issued
  uint4 vv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) vv[it] = *reinterpret_cast<const uint4*>(smem + (wid * 32 + it * RPI + lrow) * EPB + lc * 16);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int lr = wid * 32 + it * RPI + lrow;   // (orders that put a block's waves on adjacent rows at the same time measured 1.8x slower)
    const int m = m0 + lr;
    const uint4 v = vv[it];
    if (want_rowstat) {
      // this row's 8 stored values per lane -> (max, first argmax, sum exp(x - max)) over the tile's valid columns: the LPR lanes
      // of a row are contiguous, so xor-shuffles below LPR stay inside the row
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      float x[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[2 * i] = __uint_as_float(w[i] << 16); x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
      float mx = -INFINITY;
      int mi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (nst + i < g.N && x[i] > mx) { mx = x[i]; mi = nst + i; }
#pragma unroll
      for (int d = 1; d < LPR; d <<= 1) {
        const float om = __shfl_xor(mx, d, 64);
        const int oi = __shfl_xor(mi, d, 64);
        if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
      }
      float se = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (nst + i < g.N) se += __expf(x[i] - mx);
#pragma unroll
      for (int d = 1; d < LPR; d <<= 1) se += __shfl_xor(se, d, 64);
      if (lc == 0 && m < g.M) {
        const size_t o = (size_t)m * g.gn + tn;
        g.row_stat[2 * o] = mx; g.row_stat[2 * o + 1] = se; g.row_arg[o] = mi;
      }
    }
    if (want_stats) {                            // rows past M / past the utterance hold zeros
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = __uint_as_float(w[i] << 16), x1 = __uint_as_float(w[i] & 0xffff0000u);
        cs[2 * i] += x0; cq[2 * i] = fmaf(x0, x0, cq[2 * i]);
        cs[2 * i + 1] += x1; cq[2 * i + 1] = fmaf(x1, x1, cq[2 * i + 1]);
      }
    }
    if (m < g.M && nst < g.N) {
      bf16_t* dst = C + (uint32_t)m * (uint32_t)g.ldc + (uint32_t)nst;
      if (full_n) {
        *reinterpret_cast<uint4*>(dst) = v;
      } else {
        store_bf16_tail(dst, v, min(g.N - nst, 8));
      }
    }
  }
  LASR_STAMP(3);
  if (want_stats) {
    // lanes with the same lc hold the same 8 columns (rows lrow, lrow + RPI, ...): fold them, then one row of partial sums
    // per wave into LDS (the image is dead once every wave is through its rows) and a fixed-order sum over the 8 waves
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int d = LPR; d < 64; d <<= 1) {
        cs[i] += __shfl_xor(cs[i], d, 64);
        cq[i] += __shfl_xor(cq[i], d, 64);
      }
    }
    __syncthreads();
    float* s_part = reinterpret_cast<float*>(smem);        // [8 waves][2][BTN]
    if (lrow == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s_part[(wid * 2 + 0) * BTN + lc * 8 + i] = cs[i];
        s_part[(wid * 2 + 1) * BTN + lc * 8 + i] = cq[i];
      }
    }
    __syncthreads();
    if (tid < BTN) {
      const int n = n0 + tid;
      if (n < g.N) {
        float* P = g.stat_partials + (size_t)tm * 2 * g.N;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { s0 += s_part[(w * 2 + 0) * BTN + tid]; s1 += s_part[(w * 2 + 1) * BTN + tid]; }
        P[n] = s0;
        P[g.N + n] = s1;
      }
    }
  }
#ifdef LASR_GEMM_STAMPS
  __builtin_amdgcn_s_waitcnt(0);   // the stores of this wave have been accepted
  LASR_STAMP(4);
#endif
}

template <bool TRANS_A, bool TRANS_B, bool NARROW>
__global__ __launch_bounds__(512, 1) void gemm_bf16_big_kernel(Bf16Batch gb) {
  const int lid_all = xcd_remap(blockIdx.x, gb.total);
  const bool second = lid_all >= gb.tiles0;
  const Bf16Args& g = second ? gb.p[1] : gb.p[0];
  gemm_bf16_big_tile<TRANS_A, TRANS_B, NARROW, false>(g, second ? lid_all - gb.tiles0 : lid_all);
}

// Folded eval form of a residual unit: out = act([u_masked | x] . [a W | a2 Wr]^T + (b + b2)), one problem per launch,
// 256 x 128 tiles (a lone problem fills the chip only with the narrow tile).
__global__ __launch_bounds__(512, 1) void gemm_bf16_dual_kernel(Bf16Batch gb) {
  gemm_bf16_big_tile<false, false, true, false, true>(gb.p[0], xcd_remap(blockIdx.x, gb.total));
}

// The stage-batched 1x1 weight gradients on the 256 x 256 tile: C_i = A_i^T B_i with A_i [K][M_i], B_i [K][N_i]
// (K = B*T' = 16 032 rows of dy and of the layer input), split-K slices into f32 slabs.  One workgroup per CU and
// slice; the K loop of this tile runs at ~1.3 PFLOP/s against ~0.7 for the 128 x 128 form.
// NTL: operand loads with the non-temporal hint - every operand of a stage's weight gradients (dy, dy2, u, x of all its units) is read
// here for the last time, most of it cold; the slabs this launch WRITES are what the reduction right behind it reads
template <bool NTL>
__global__ __launch_bounds__(512, 1) void gemm_bf16_big_multi_kernel(Bf16Multi gm) {
  if ((int)blockIdx.x >= gm.total) {                      // workgroup-uniform: a rider workgroup (reduce_body.h)
    const int r = blockIdx.x - gm.total;
    for (int s = 0; s < gm.n_rd; ++s) {
      const lasr_reduce_desc q = gm.rd[s];
      for (int64_t i = (int64_t)r * big::NT + threadIdx.x; i < q.n; i += (int64_t)gm.n_rwg * big::NT) reduce_many_elem(q, i);
    }
    return;
  }
  const int lid_all = xcd_remap(blockIdx.x, gm.total);
  int i = 0;
  while (i + 1 < gm.n && gm.start[i + 1] <= lid_all) ++i;   // workgroup-uniform scan of at most 32 entries
  gemm_bf16_big_tile<true, true, false, true, false, NTL>(gm.p[i], lid_all - gm.start[i]);
}

// The stage's weight gradients with the BiLSTM backward recurrences of the context branch in the SAME grid (Context / ContextSE,
// round 4): the first n_wg workgroups run two recurrences each - the forward and the reverse direction of one utterance: same length,
// so their per-step barriers pair up without an idle phase - in 192-thread slots (three whole waves; lanes 160-191 compute on clamped
// indices and store nothing, waves 6 and 7 leave at once); the GEMM tiles follow.  The recurrence is ~870 ns of dependent instructions
// per frame whatever else runs, the weight gradients of the units above block3 do not depend on it: 437 us on 64 small workgroups
// with 190 CUs idle becomes ~500 us with the stage's biggest launch inside it (cfg4 step 3.751 -> 3.646 ms).
struct LstmJob { lstm::BwdArgs a; int n_wg; int n_utt; };
template <bool NTL, int KP, bool PAIR>
__global__ __launch_bounds__(512, 1) void gemm_bf16_big_multi_lstm_kernel(Bf16Multi gm, LstmJob job) {
  if ((int)blockIdx.x < job.n_wg) {                       // workgroup-uniform
    __shared__ lstm::BwdSmem sm[2];
    const int slot = threadIdx.x / 192;
    if (slot >= (PAIR ? 2 : 1)) return;                   // (a finished wave is not waited for by s_barrier)
    const int b = PAIR ? blockIdx.x : blockIdx.x >> 1, dir = PAIR ? slot : (blockIdx.x & 1);
    lstm::bilstm_bwd_body<bf16_t, true, KP>(job.a, b, dir, (int)threadIdx.x - slot * 192, 192, sm[slot], [] { lds_barrier(); });
    return;
  }
  const int lid_all = xcd_remap((int)blockIdx.x - job.n_wg, gm.total);
  int i = 0;
  while (i + 1 < gm.n && gm.start[i + 1] <= lid_all) ++i;   // workgroup-uniform scan of at most 32 entries
  gemm_bf16_big_tile<true, true, false, true, false, NTL>(gm.p[i], lid_all - gm.start[i]);
}

static int fill_args(Bf16Args& a, const GemmArgs& g, int tm, int tn, int gz) {
  const int64_t lim = (int64_t)1 << 31;
  const int64_t gn = cdiv(g.N, tn), gm = cdiv(g.M, tm);
  if (g.M * g.lda >= lim || g.K * g.lda >= lim || g.N * g.ldb >= lim || g.K * g.ldb >= lim || g.M * g.ldc >= lim ||
      gn * gm * gz >= lim / 2)
    return fail(LASR_E_SHAPE, "lasr_gemm(bf16): matrix exceeds the kernel's 32-bit element offsets");
  a.A = reinterpret_cast<const bf16_t*>(g.A); a.B = reinterpret_cast<const bf16_t*>(g.B); a.C = g.C;
  a.bias = g.bias; a.row_lens = g.row_lens; a.stat_partials = g.stat_partials; a.split_ws = g.split_ws;
  a.M = (int)g.M; a.N = (int)g.N; a.K = (int)g.K; a.lda = (int)g.lda; a.ldb = (int)g.ldb; a.ldc = (int)g.ldc;
  a.rows_per_seq = (int)(g.rows_per_seq > 0 ? g.rows_per_seq : 1); a.k_per_split = (int)g.k_per_split;
  a.gn = (int)gn; a.gm = (int)gm; a.gz = gz;
  a.vecA = g.vecA; a.vecB = g.vecB;
  a.vecC = (g.ldc % 8 == 0) && (reinterpret_cast<uintptr_t>(g.C) % 16 == 0);
  a.A2 = nullptr; a.lda2 = 0; a.K1 = 0; a.act = 0;
  a.row_stat = nullptr; a.row_arg = nullptr;
  a.addend = reinterpret_cast<const bf16_t*>(g.addend);
  return 0;
}

// g[0..n): problems sharing dtype_c / transposition; gz[i] = split-K slices of problem i.
// stat_tiles[i] receives the number of row tiles problem i writes BN partial sums for.
int launch_gemm_bf16_batch(const GemmArgs* g, const int* gz, int n, int dtype_c, int transA, int transB, hipStream_t st,
                           int* stat_tiles) {
  const bool f32_out = dtype_c == LASR_F32 || g[0].split_ws != nullptr;   // split-K slabs are f32
  // The 256x256 tile pays when its (one per CU) workgroups cover most of the chip.
  static const int big_min = getenv("LASR_GEMM_BIG_MIN_TILES") ? atoi(getenv("LASR_GEMM_BIG_MIN_TILES")) : 120;
  int64_t nmax = 0;
  for (int i = 0; i < n; ++i) nmax = std::max<int64_t>(nmax, g[i].N);
  // LASR_GEMM_FORCE_NARROW=1: 256x128 tiles for the wide layers too (504 workgroups, two rounds on an idle chip).  A/B switch for the
  // N > 1 step: beside RCCL's channel kernels a 252-tile launch runs two rounds as soon as 5 CUs are taken
  static const bool force_narrow = getenv("LASR_GEMM_FORCE_NARROW") && atoi(getenv("LASR_GEMM_FORCE_NARROW")) != 0;
  const bool narrow = nmax <= 256 || force_narrow;          // 256x128 tiles: two tile columns for the 256-channel layers
  const int btn = narrow ? 128 : 256;
  int64_t big_tiles = 0;
  for (int i = 0; i < n; ++i) big_tiles += cdiv(g[i].M, big::BTM) * cdiv(g[i].N, btn) * gz[i];
  bool vec = true;   // every operand aligned for 16-byte chunks (pitch % 8, base % 16): true for all model tensors
  for (int i = 0; i < n; ++i) vec = vec && g[i].vecA && g[i].vecB;
  bool has_add = false;
  for (int i = 0; i < n; ++i) has_add = has_add || g[i].addend != nullptr;   // (epilogue of the 128x128 tile only)
  const bool use_big = !f32_out && vec && big_tiles >= big_min && !has_add;
  const int tm = use_big ? big::BTM : TM, tn = use_big ? btn : TN;
  Bf16Batch b;
  LASR_TRY(fill_args(b.p[0], g[0], tm, tn, gz[0]));
  b.p[1] = b.p[0];
  if (n > 1) LASR_TRY(fill_args(b.p[1], g[1], tm, tn, gz[1]));
  b.tiles0 = b.p[0].gn * b.p[0].gm * b.p[0].gz;
  b.total = b.tiles0 + (n > 1 ? b.p[1].gn * b.p[1].gm * b.p[1].gz : 0);
  if (stat_tiles)
    for (int i = 0; i < n; ++i) stat_tiles[i] = b.p[i].gm;
  const dim3 grid1((unsigned)b.total);
  if (use_big) {
#define LASR_BIG_CASE(TA_, TB_)                                                                              \
  do {                                                                                                     \
    if (narrow) hipLaunchKernelGGL((gemm_bf16_big_kernel<TA_, TB_, true>), grid1, dim3(big::NT), 0, st, b);  \
    else hipLaunchKernelGGL((gemm_bf16_big_kernel<TA_, TB_, false>), grid1, dim3(big::NT), 0, st, b);        \
  } while (0)
    if (!transA && !transB) LASR_BIG_CASE(false, false);
    else if (!transA && transB) LASR_BIG_CASE(false, true);
    else if (transA && !transB) LASR_BIG_CASE(true, false);
    else LASR_BIG_CASE(true, true);
#undef LASR_BIG_CASE
    LASR_LAUNCH_CHECK("gemm_bf16_big_kernel");
    return 0;
  }
#define LASR_BF16_CASE(TC_, TA_, TB_)                                                                    \
  do {                                                                                                  \
    if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<TC_, TA_, TB_, true>), grid1, dim3(256), 0, st, b);    \
    else hipLaunchKernelGGL((gemm_bf16_kernel<TC_, TA_, TB_, false>), grid1, dim3(256), 0, st, b);       \
  } while (0)
#define LASR_BF16_TC(TC_)                                         \
  do {                                                            \
    if (!transA && !transB) LASR_BF16_CASE(TC_, false, false);    \
    else if (!transA && transB) LASR_BF16_CASE(TC_, false, true); \
    else if (transA && !transB) LASR_BF16_CASE(TC_, true, false); \
    else LASR_BF16_CASE(TC_, true, true);                         \
  } while (0)
  if (f32_out) LASR_BF16_TC(float); else LASR_BF16_TC(bf16_t);
#undef LASR_BF16_TC
#undef LASR_BF16_CASE
  LASR_LAUNCH_CHECK("gemm_bf16_kernel");
  return 0;
}

int launch_gemm_bf16_dual(const GemmArgs& g, const void* A2, int64_t lda2, int64_t K1, const float* bias, int act, hipStream_t st) {
  if (K1 <= 0 || K1 >= g.K || K1 % TK || (g.K - K1) % TK || lda2 % 8 || g.lda % 8 || g.ldb % 8 || g.ldc % 8 ||
      reinterpret_cast<uintptr_t>(g.A) % 16 || reinterpret_cast<uintptr_t>(A2) % 16 || reinterpret_cast<uintptr_t>(g.B) % 16 ||
      reinterpret_cast<uintptr_t>(g.C) % 16 || !bias)
    return fail(LASR_E_SHAPE, "lasr_gemm_dual: K parts must be multiples of %d, pitches of 8, pointers 16-byte aligned, bias given", TK);
  Bf16Batch b;
  LASR_TRY(fill_args(b.p[0], g, big::BTM, 128, 1));
  b.p[0].A2 = reinterpret_cast<const bf16_t*>(A2); b.p[0].lda2 = (int)lda2; b.p[0].K1 = (int)K1; b.p[0].act = act;
  b.p[0].bias = bias;
  b.p[1] = b.p[0];
  b.tiles0 = b.p[0].gn * b.p[0].gm;
  b.total = b.tiles0;
  hipLaunchKernelGGL(gemm_bf16_dual_kernel, dim3((unsigned)b.total), dim3(big::NT), 0, st, b);
  LASR_LAUNCH_CHECK("gemm_bf16_dual_kernel");
  return 0;
}

// One problem C = A B^T (+bias) with bf16 output and the softmax row statistics, on the 256 x 256 tile (the decoder of a
// large-vocabulary model: M = B*T' rows, N = classes, K = 1024).  Returns the number of column tiles through n_col_tiles.
int launch_gemm_bf16_rowstat(const GemmArgs& g, float* row_stat, int32_t* row_arg, int* n_col_tiles, hipStream_t st) {
  if (!g.vecA || !g.vecB || g.ldc % 8 || reinterpret_cast<uintptr_t>(g.C) % 16 || !row_stat || !row_arg)
    return fail(LASR_E_SHAPE, "lasr_gemm_rowstat: operands must be 16-byte aligned with pitches that are multiples of 8");
  Bf16Batch b;
  LASR_TRY(fill_args(b.p[0], g, big::BTM, 256, 1));
  b.p[0].row_stat = row_stat; b.p[0].row_arg = row_arg;
  b.p[1] = b.p[0];
  b.tiles0 = b.p[0].gn * b.p[0].gm;
  b.total = b.tiles0;
  if (n_col_tiles) *n_col_tiles = b.p[0].gn;
  hipLaunchKernelGGL((gemm_bf16_big_kernel<false, false, false>), dim3((unsigned)b.total), dim3(big::NT), 0, st, b);
  LASR_LAUNCH_CHECK("gemm_bf16_big_kernel(rowstat)");
  return 0;
}

// n <= 32 split-K problems with f32 slab output, both operands row-contiguous ([K][M], [K][N]: weight gradients)
int launch_gemm_bf16_multi(const GemmArgs* g, const int* gz, int n, bool big_tile, hipStream_t st, const lstm::BwdArgs* lstm_job, int lstm_wgs,
                           const lasr_reduce_desc* riders, int n_riders, int* riders_taken) {
  if (n < 1 || n > kMaxMulti) return fail(LASR_E_ARG, "lasr_gemm_multi_split_partials: 1..%d problems", kMaxMulti);
  if (riders_taken) *riders_taken = 0;
  Bf16Multi m;
  m.n_rd = 0; m.n_rwg = 0;
  bool vec = true;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    if (!g[i].split_ws) return fail(LASR_E_ARG, "lasr_gemm_multi_split_partials: every problem needs a slab buffer");
    LASR_TRY(fill_args(m.p[i], g[i], big_tile ? big::BTM : TM, big_tile ? 256 : TN, gz[i]));
    vec = vec && g[i].vecA && g[i].vecB;
    m.start[i] = total;
    total += m.p[i].gn * m.p[i].gm * m.p[i].gz;
  }
  for (int i = n; i < kMaxMulti; ++i) m.p[i] = m.p[0];
  for (int i = n; i <= kMaxMulti; ++i) m.start[i] = total;
  m.n = n; m.total = total;
  if (big_tile) {
    if (!vec) return fail(LASR_E_ARG, "lasr_gemm_multi_split_partials: the 256-row tile needs 16-byte aligned operand rows");
    if (lstm_job) {                                        // the context branch's recurrences ride in this grid (bf16 model)
      // One recurrence per workgroup (2 B workgroups) or the two directions of an utterance in one (B workgroups, LASR_LSTM_PAIR=1);
      // operand ring 16 steps ahead.  Measured in the cfg4 step after the recurrence's rewrite (profiles/r04_cfg4_lstm_pair_kp.txt):
      // paired / 8 steps 3.411 ms, paired / 16 3.395, single / 8 3.382, single / 16 3.378 - a recurrence alone on its CU keeps the
      // stand-alone kernel's step time (two of them share the CU's LDS pipe and issue slots), and under the GEMM tiles' traffic a load
      // takes longer than 8 steps to arrive.
      static const bool pair = getenv("LASR_LSTM_PAIR") && atoi(getenv("LASR_LSTM_PAIR")) == 1;
      LstmJob job; job.a = *lstm_job; job.n_wg = lstm_wgs; job.n_utt = pair ? lstm_wgs : lstm_wgs / 2;
      if (job.n_wg % 8) return fail(LASR_E_SHAPE, "gemm + BiLSTM grid: the batch must be a multiple of 8 utterances");
      if (pair) hipLaunchKernelGGL((gemm_bf16_big_multi_lstm_kernel<false, 16, true>), dim3((unsigned)(total + job.n_wg)), dim3(big::NT), 0, st, m, job);
      else hipLaunchKernelGGL((gemm_bf16_big_multi_lstm_kernel<false, 16, false>), dim3((unsigned)(total + job.n_wg)), dim3(big::NT), 0, st, m, job);
      LASR_LAUNCH_CHECK("gemm_bf16_big_multi_lstm_kernel");
      return 0;
    }
    // riders: as many of the offered (complete) reductions as the idle CUs can finish while the tiles run.  A rider workgroup streams
    // at ~20 GB/s (one CU's share of what the memory system gives a streaming read), a tile slice takes ~1.7 us per K step of 64 + ~20 us
    static const bool riders_off = getenv("LASR_WGRAD_RIDERS") && atoi(getenv("LASR_WGRAD_RIDERS")) == 0;
    const int idle = 256 - total;
    if (!riders_off && riders && riders_taken && n_riders > 0 && idle >= 8) {
      int ksteps = 0;
      for (int i = 0; i < n; ++i) ksteps = std::max(ksteps, (int)cdiv(m.p[i].k_per_split, 64));
      double budget = (double)idle * 20e9 * ((double)ksteps * 1.7e-6 + 20e-6) * 0.8;
      while (m.n_rd < n_riders && m.n_rd < kMaxRiders) {
        const lasr_reduce_desc& q = riders[m.n_rd];
        const double bytes = (double)q.n * (q.n_partials + 1) * sizeof(float);
        if (bytes > budget) break;
        budget -= bytes;
        m.rd[m.n_rd++] = q;
      }
      if (m.n_rd > 0) { m.n_rwg = idle; *riders_taken = m.n_rd; }
    }
    for (int i = m.n_rd; i < kMaxRiders; ++i) m.rd[i] = lasr_reduce_desc{nullptr, nullptr, 0, 0};
    if (nt_loads_mask() & 8) hipLaunchKernelGGL(gemm_bf16_big_multi_kernel<true>, dim3((unsigned)(total + m.n_rwg)), dim3(big::NT), 0, st, m);
    else hipLaunchKernelGGL(gemm_bf16_big_multi_kernel<false>, dim3((unsigned)(total + m.n_rwg)), dim3(big::NT), 0, st, m);
    LASR_LAUNCH_CHECK("gemm_bf16_big_multi_kernel");
    return 0;
  }
  if (lstm_job) return fail(LASR_E_SHAPE, "gemm + BiLSTM grid needs the 256-row tile form");
  if (vec) hipLaunchKernelGGL(gemm_bf16_multi_kernel<true>, dim3((unsigned)total), dim3(256), 0, st, m);
  else hipLaunchKernelGGL(gemm_bf16_multi_kernel<false>, dim3((unsigned)total), dim3(256), 0, st, m);
  LASR_LAUNCH_CHECK("gemm_bf16_multi_kernel");
  return 0;
}

int launch_gemm_bf16(const GemmArgs& g, int gz, int dtype_c, int transA, int transB, hipStream_t st, int* stat_tiles) {
  return launch_gemm_bf16_batch(&g, &gz, 1, dtype_c, transA, transB, st, stat_tiles);
}

}  // namespace lasr

#ifdef LASR_GEMM_STAMPS
extern "C" int lasr_debug_set_gemm_stamps(void* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(lasr::g_stamps), &p, sizeof(p));
}
#endif
