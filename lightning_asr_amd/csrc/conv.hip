// Depthwise 1-D convolution over channels-last [B][T][C] activations: forward (also used, with the
// taps reversed, as the data gradient) and the weight gradient.
// Replaces nn.Conv1d(groups=C, padding=k//2, bias=False) at models/QuartNet.py:19-21,30 and its
// autograd backward.  VALU work (0.13 of 2.5 GMAC/utterance); time tiles are staged through LDS
// with their k-1 halo, channels ride the lanes so every global access is a full coalesced row.
#include "common.h"
#include "fused.h"

namespace lasr {

// ---- the step's precomputed depthwise tap tables (fused.h) ---------------------------------------------------------------------
static thread_local const DwTapCtx* g_dw_tap_ctx = nullptr;
void dw_taps_set_ctx(const DwTapCtx* ctx) { g_dw_tap_ctx = ctx; }
bool dw_taps_enabled() {
  static const bool on = !(getenv("LASR_DW_TAPS") && atoi(getenv("LASR_DW_TAPS")) == 0);
  return on;
}
// the layer's table ([2][C][kDwTapRow]) when the model call in progress on this thread registered one for `w`; the kernels copy whole
// 64-channel row blocks, so the layer's width must be a multiple of 64
const uint32_t* dw_taps_for(const float* w, int flip, int64_t C) {
  (void)flip;
  const DwTapCtx* c = g_dw_tap_ctx;
  if (!c || C % 64 != 0) return nullptr;
  for (int i = 0; i < c->n; ++i)
    if (c->w[i] == w && c->C[i] == (int)C) return c->t[i];
  return nullptr;
}

static constexpr int kCB = 64;    // channels per workgroup (16 lanes x 4 channels)
static constexpr int kTT = 128;   // output frames per workgroup (16 lanes x 8 outputs)
static constexpr int kR = 8;      // outputs per thread
static constexpr int kMaxK = 128;

// grid: (ceil(Tout/(16 R)), ceil(C/kCB), B), block 256, dynamic LDS: x tile + taps.  R = outputs per thread: 8 (128-frame tiles) or
// 4 (64-frame tiles: twice the workgroups for the one stride-2 layer, whose 64 channels x 32 utterances x 4 tiles left half the chip idle)
template <typename T, int R = kR>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const T* __restrict__ addend, T* __restrict__ y, int64_t Tin,
                                                         int64_t Tout, int64_t C, int k, int stride, int flip) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = k / 2;
  constexpr int TT = 16 * R;
  const int in_rows = (TT - 1) * stride + k;
  float* s_x = smem;                         // [in_rows][kCB]
  float* s_w = smem + (size_t)in_rows * kCB;  // [k][kCB]
  const int b = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.y * kCB;
  const int64_t t0 = (int64_t)blockIdx.x * TT;
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int64_t c = c0 + cl * 4;
  const bool c_ok = c < C;  // C % 4 == 0
  const T* xb = x + (int64_t)b * Tin * C;

  // taps -> LDS [j][channel]
  // (rounds of eight unconditional loads - clamped channel, value masked by bit operations - then their LDS writes: a guarded load per
  //  element was a memory round trip per element, nine in a row for k = 33; round 4)
  for (int i0 = 0; i0 < k * kCB; i0 += 8 * 256) {
    float tv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u * 256 + (int)threadIdx.x, k * kCB - 1);
      const int ch = i / k, j = i - ch * k;
      tv[u] = w[min(c0 + ch, C - 1) * k + (flip ? (k - 1 - j) : j)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + (int)threadIdx.x;
      if (i < k * kCB) {
        const int ch = i / k, j = i - ch * k;
        s_w[j * kCB + ch] = __uint_as_float(__float_as_uint(tv[u]) & (c0 + ch < C ? 0xffffffffu : 0u));
      }
    }
  }
  // input rows t0*stride - pad ... (+in_rows), zero outside [0, Tin)
  const int64_t in0 = t0 * stride - pad;
  // eight rows per thread in flight, every load unconditional (clamped address, masked value): a guarded load per row
  // made the staging a chain of ~18 exposed memory round trips (15 us for the 64-channel stride-2 layer)
  const int64_t cld = c_ok ? c : C - 4;
  for (int rb = tl; rb < in_rows; rb += 16 * 8) {
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t ti = in0 + rb + 16 * u;
      Elem<T>::ld4(xb + min(max(ti, (int64_t)0), Tin - 1) * C + cld, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = rb + 16 * u;
      const int64_t ti = in0 + r;
      const float m = (c_ok && ti >= 0 && ti < Tin) ? 1.f : 0.f;
      if (r < in_rows) *reinterpret_cast<float4*>(s_x + (size_t)r * kCB + cl * 4) = make_float4(v[u][0] * m, v[u][1] * m, v[u][2] * m, v[u][3] * m);
    }
  }
  __syncthreads();

  float acc[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
  const float* xs = s_x + (size_t)(tl * R * stride) * kCB + cl * 4;
  const float* ws = s_w + cl * 4;
  for (int j = 0; j < k; ++j) {
    const float4 wv = *reinterpret_cast<const float4*>(ws + j * kCB);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + (size_t)(r * stride + j) * kCB);
      acc[r][0] = fmaf(wv.x, xv.x, acc[r][0]);
      acc[r][1] = fmaf(wv.y, xv.y, acc[r][1]);
      acc[r][2] = fmaf(wv.z, xv.z, acc[r][2]);
      acc[r][3] = fmaf(wv.w, xv.w, acc[r][3]);
    }
  }
  if (!c_ok) return;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t t = t0 + tl * R + r;
    if (t < Tout) {
      const int64_t off = ((int64_t)b * Tout + t) * C + c;
      if (addend) {
        float a[4];
        Elem<T>::ld4(addend + off, a);
        acc[r][0] += a[0]; acc[r][1] += a[1]; acc[r][2] += a[2]; acc[r][3] += a[3];
      }
      Elem<T>::st4(y + off, acc[r]);
    }
  }
}

// Weight gradient.  grid: (n_chunks, ceil(C/kCB), B); each workgroup walks `chunk` output frames in
// steps of kWT and writes a partial [kCB][k] to partials[(b*n_chunks+chunk)][C][k].
static constexpr int kWT = 64;       // frames per LDS step
static constexpr int kWChunk = 256;  // frames per workgroup (stride-1 kernel)
static constexpr int kWChunkG = 64;  // frames per workgroup of the generic kernel: the only stride-2 layer has 64 channels, so
                                     // (chunks x B) must supply the parallelism: 8 x 32 workgroups at T' = 501
static constexpr int kQ = kMaxK / 16;

template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* __restrict__ partials, int64_t Tin, int64_t Tout,
                                                           int64_t C, int k, int stride) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pad = k / 2;
  const int in_rows = (kWT - 1) * stride + k;
  float* s_x = smem;                           // [in_rows][kCB]
  float* s_d = smem + (size_t)in_rows * kCB;    // [kWT][kCB]
  const int b = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.y * kCB;
  const int cl = threadIdx.x & 15, jl = threadIdx.x >> 4;
  const int64_t c = c0 + cl * 4;
  const bool c_ok = c < C;
  const T* xb = x + (int64_t)b * Tin * C;
  const T* db = dy + (int64_t)b * Tout * C;
  float acc[kQ][4];
#pragma unroll
  for (int q = 0; q < kQ; ++q) acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
  const int nq = (k + 15) / 16;
  const int64_t tbeg = (int64_t)blockIdx.x * kWChunkG;
  const int64_t tend = tbeg + kWChunkG < Tout ? tbeg + kWChunkG : Tout;
  for (int64_t t0 = tbeg; t0 < tend; t0 += kWT) {
    __syncthreads();
    const int64_t in0 = t0 * stride - pad;
    // branch-free staging, eight x rows (then the four dy rows) per thread in flight
    const int64_t cld = c_ok ? c : C - 4;
    for (int rb = jl; rb < in_rows; rb += 16 * 8) {
      float v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t ti = in0 + rb + 16 * u;
        Elem<T>::ld4(xb + min(max(ti, (int64_t)0), Tin - 1) * C + cld, v[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 16 * u;
        const int64_t ti = in0 + r;
        const float m = (c_ok && ti >= 0 && ti < Tin) ? 1.f : 0.f;
        if (r < in_rows) *reinterpret_cast<float4*>(s_x + (size_t)r * kCB + cl * 4) = make_float4(v[u][0] * m, v[u][1] * m, v[u][2] * m, v[u][3] * m);
      }
    }
    {
      float v[kWT / 16][4];
#pragma unroll
      for (int u = 0; u < kWT / 16; ++u) Elem<T>::ld4(db + min(t0 + jl + 16 * u, Tout - 1) * C + cld, v[u]);
#pragma unroll
      for (int u = 0; u < kWT / 16; ++u) {
        const int r = jl + 16 * u;
        const float m = (c_ok && t0 + r < tend) ? 1.f : 0.f;
        *reinterpret_cast<float4*>(s_d + (size_t)r * kCB + cl * 4) = make_float4(v[u][0] * m, v[u][1] * m, v[u][2] * m, v[u][3] * m);
      }
    }
    __syncthreads();
    // The tap-group count as a compile-time constant and every LDS read of two frames issued before their FMAs (clamped row: rows
    // past the staged window belong to taps >= k, discarded below).  With `if (q < nq) if (row < in_rows)` around each read, every
    // one of the 64 x nq reads was its own LDS round trip: 20 us for the 34 M MACs of the one stride-2 layer (round 4).
    auto inner = [&](auto nq_c) {
      constexpr int NQ = decltype(nq_c)::value;
#pragma unroll 2
      for (int t = 0; t < kWT; t += 2) {
        float4 dv[2], xv[2][NQ];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          dv[u] = *reinterpret_cast<const float4*>(s_d + (size_t)(t + u) * kCB + cl * 4);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            xv[u][q] = *reinterpret_cast<const float4*>(s_x + (size_t)min((t + u) * stride + jl + 16 * q, in_rows - 1) * kCB + cl * 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            acc[q][0] = fmaf(dv[u].x, xv[u][q].x, acc[q][0]);
            acc[q][1] = fmaf(dv[u].y, xv[u][q].y, acc[q][1]);
            acc[q][2] = fmaf(dv[u].z, xv[u][q].z, acc[q][2]);
            acc[q][3] = fmaf(dv[u].w, xv[u][q].w, acc[q][3]);
          }
      }
    };
    static_assert(kWT % 2 == 0 && kQ == 8, "dwconv_wgrad_kernel: tap groups");
    switch (nq) {                                          // workgroup-uniform
      case 1: inner(std::integral_constant<int, 1>{}); break;
      case 2: inner(std::integral_constant<int, 2>{}); break;
      case 3: inner(std::integral_constant<int, 3>{}); break;
      case 4: inner(std::integral_constant<int, 4>{}); break;
      case 5: inner(std::integral_constant<int, 5>{}); break;
      case 6: inner(std::integral_constant<int, 6>{}); break;
      case 7: inner(std::integral_constant<int, 7>{}); break;
      default: inner(std::integral_constant<int, 8>{}); break;
    }
  }
  if (!c_ok) return;
  float* out = partials + ((int64_t)b * gridDim.x + blockIdx.x) * C * k;
#pragma unroll
  for (int q = 0; q < kQ; ++q) {
    const int j = jl + 16 * q;
    if (q < nq && j < k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) out[(c + e) * k + j] = acc[q][e];
    }
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partials, int n_part, int64_t n,
                                                           float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < n_part; ++p) s += partials[(int64_t)p * n + i];
  out[i] = s;
}


// ------------------------------------------------------------------ stride-1 fast paths --------
// LDS keeps the activation tile in its storage type (bf16 halves the footprint -> 3 workgroups/CU);
// a lane reads 4 channels of one frame with one ds_read (b128 for f32, b64 for bf16).
template <typename T>
__device__ __forceinline__ float4 lds_ld4(const T* p);
template <>
__device__ __forceinline__ float4 lds_ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 lds_ld4<bf16_t>(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
template <typename T>
__device__ __forceinline__ void lds_copy4(T* dst, const T* src, bool ok);
template <>
__device__ __forceinline__ void lds_copy4<float>(float* dst, const float* src, bool ok) {
  *reinterpret_cast<float4*>(dst) = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <>
__device__ __forceinline__ void lds_copy4<bf16_t>(bf16_t* dst, const bf16_t* src, bool ok) {
  *reinterpret_cast<uint2*>(dst) = ok ? *reinterpret_cast<const uint2*>(src) : make_uint2(0u, 0u);
}

// Forward / data-gradient, stride 1.  Each lane owns 4 channels x 8 consecutive output frames and
// slides an 8-frame register window over the taps: per tap one new frame and one tap vector are
// read from LDS for 32 FMAs (the generic kernel above reads 9 vectors for the same work).
// The taps are zero-padded to kpad = 8*ceil(k/8) so the window rotation unrolls with static indices.
// Tile staging: every thread first issues ALL its 16-byte global loads (7 in flight for bf16), then
// writes LDS; tile rows are padded (kTileLd) so the 16 time-lanes of a half-wave hit disjoint banks;
// in bf16 mode the taps are kept as bf16 in LDS too, which fits 4 workgroups per CU (one full round
// of the 1024-workgroup grid).
template <typename T>
struct S1 {
  static constexpr int kVec = 16 / sizeof(T);                    // elements per 16-byte global load
  static constexpr int kLanesPerRow = kCB / kVec;                // 8 (bf16) / 16 (f32)
  static constexpr int kRowsPerPass = 256 / kLanesPerRow;        // 32 / 16
  static constexpr int kTileLd = kCB + (sizeof(T) == 2 ? 8 : 0); // elements per LDS row: 144 B (bf16), 256 B (f32)
  static constexpr int kMaxRows = kTT + kMaxK + 8;
  static constexpr int kPasses = (kMaxRows + kRowsPerPass - 1) / kRowsPerPass;
};

template <typename T>
__global__ __launch_bounds__(256) void dwconv_s1_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const T* __restrict__ addend, T* __restrict__ y, int64_t Tlen, int64_t C,
                                                        int k, int flip) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  using P = S1<T>;
  const int pad = k / 2;
  const int kpad = (k + 7) & ~7;
  const int in_rows = kTT + kpad + 8;
  T* s_w = reinterpret_cast<T*>(smem_raw);                                          // [kpad][kCB] taps in T
  T* s_x = reinterpret_cast<T*>(smem_raw + (size_t)kpad * kCB * sizeof(T));          // [in_rows][kTileLd]
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * kCB;
  const int t0 = blockIdx.x * kTT;
  const T* xb = x + (int64_t)b * Tlen * C;
  // ---- stage the input tile: all loads first, then the LDS writes
  {
    const int lr = threadIdx.x / P::kLanesPerRow, lc = (threadIdx.x % P::kLanesPerRow) * P::kVec;
    const bool c_in = c0 + lc < C;   // C % kVec == 0 is checked on the host
    uint4 v[P::kPasses];
#pragma unroll
    for (int p = 0; p < P::kPasses; ++p) {
      const int r = lr + p * P::kRowsPerPass;
      const int64_t ti = (int64_t)t0 - pad + r;
      v[p] = make_uint4(0u, 0u, 0u, 0u);
      if (r < in_rows && c_in && ti >= 0 && ti < Tlen) v[p] = *reinterpret_cast<const uint4*>(xb + ti * C + c0 + lc);
    }
#pragma unroll
    for (int p = 0; p < P::kPasses; ++p) {
      const int r = lr + p * P::kRowsPerPass;
      if (r < in_rows) *reinterpret_cast<uint4*>(s_x + (size_t)r * P::kTileLd + lc) = v[p];
    }
  }
  // taps -> LDS [j][channel]: consecutive lanes take consecutive channels, so the LDS stores are
  // conflict-free (lanes along j would all hit one bank); the strided global reads stay in L1, each
  // 128-byte line of w serving 32 consecutive taps of its channel.
  for (int i = threadIdx.x; i < kpad * kCB; i += 256) {
    const int j = i >> 6, ch = i & (kCB - 1);
    float v = 0.f;
    if (j < k && c0 + ch < C) v = w[(int64_t)(c0 + ch) * k + (flip ? (k - 1 - j) : j)];
    Elem<T>::st(s_w + j * kCB + ch, v);
  }
  __syncthreads();

  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int c = c0 + cl * 4;
  float4 acc[kR], win[8];
#pragma unroll
  for (int r = 0; r < kR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const T* xs = s_x + (size_t)(tl * kR) * P::kTileLd + cl * 4;
  const T* ws = s_w + cl * 4;
#pragma unroll
  for (int i = 0; i < 7; ++i) win[i] = lds_ld4<T>(xs + (size_t)i * P::kTileLd);
  for (int j0 = 0; j0 < kpad; j0 += 8) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      win[(jj + 7) & 7] = lds_ld4<T>(xs + (size_t)(j0 + jj + 7) * P::kTileLd);
      const float4 wv = lds_ld4<T>(ws + (j0 + jj) * kCB);
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const float4 xv = win[(jj + r) & 7];
        acc[r].x = fmaf(wv.x, xv.x, acc[r].x);
        acc[r].y = fmaf(wv.y, xv.y, acc[r].y);
        acc[r].z = fmaf(wv.z, xv.z, acc[r].z);
        acc[r].w = fmaf(wv.w, xv.w, acc[r].w);
      }
    }
  }
  if (c >= C) return;
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const int64_t t = (int64_t)t0 + tl * kR + r;
    if (t < Tlen) {
      const int64_t off = ((int64_t)b * Tlen + t) * C + c;
      float o[4] = {acc[r].x, acc[r].y, acc[r].z, acc[r].w};
      if (addend) {
        float a[4];
        Elem<T>::ld4(addend + off, a);
        o[0] += a[0]; o[1] += a[1]; o[2] += a[2]; o[3] += a[3];
      }
      Elem<T>::st4(y + off, o);
    }
  }
}

// bf16 stride-1 depthwise conv on the packed dot unit: v_dot2c_f32_bf16 does two bf16 MACs into an f32
// accumulator per lane per issue, so pairing taps (j, j+1) halves the VALU work that bounds the f32-FMA
// form above (2*k FLOP per output element against 32 B of traffic: the kernel is VALU-, not HBM-bound).
// Same tiling (64 channels x 128 frames per workgroup, lane = 4 channels x 8 outputs).  The operands are
//   P[u][ch]  = (x[u][ch], x[u+1][ch])   built by one v_perm_b32 from two consecutive LDS rows,
//   W2[jp][ch] = (w[2jp][ch], w[2jp+1][ch])  packed when the taps are staged,
// and output r accumulates dot2(P[tb + r + 2jp], W2[jp]) over jp.  A ring of 8 P registers per channel
// slides two frames per tap pair: 2 row reads (8 B), 1 tap read (16 B), 8 v_perm and 32 dot2 per lane
// per 64 MACs.  Products of bf16 operands are exact in f32, as in the FMA form (whose taps are bf16 too).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
// (lo16 of a, lo16 of b) and (hi16 of a, hi16 of b)
__device__ __forceinline__ uint32_t pair_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pair_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__global__ __launch_bounds__(256) void dwconv_s1_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                             const bf16_t* __restrict__ addend, bf16_t* __restrict__ y, int64_t Tlen,
                                                             int64_t C, int k, int flip) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  using P = S1<bf16_t>;
  const int pad = k / 2;
  const int kpad = (k + 7) & ~7;
  const int in_rows = kTT + kpad + 8;
  uint32_t* s_w2 = reinterpret_cast<uint32_t*>(smem_raw);                                    // [kpad/2][kCB] tap pairs
  bf16_t* s_x = reinterpret_cast<bf16_t*>(smem_raw + (size_t)(kpad / 2) * kCB * sizeof(uint32_t));  // [in_rows][72]
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * kCB;
  const int t0 = blockIdx.x * kTT;
  const bf16_t* xb = x + (int64_t)b * Tlen * C;
  {
    const int lr = threadIdx.x / P::kLanesPerRow, lc = (threadIdx.x % P::kLanesPerRow) * P::kVec;
    const bool c_in = c0 + lc < C;
    uint4 v[P::kPasses];
#pragma unroll
    for (int p = 0; p < P::kPasses; ++p) {
      const int r = lr + p * P::kRowsPerPass;
      const int64_t ti = (int64_t)t0 - pad + r;
      v[p] = make_uint4(0u, 0u, 0u, 0u);
      if (r < in_rows && c_in && ti >= 0 && ti < Tlen) v[p] = *reinterpret_cast<const uint4*>(xb + ti * C + c0 + lc);
    }
#pragma unroll
    for (int p = 0; p < P::kPasses; ++p) {
      const int r = lr + p * P::kRowsPerPass;
      if (r < in_rows) *reinterpret_cast<uint4*>(s_x + (size_t)r * P::kTileLd + lc) = v[p];
    }
  }
  for (int i = threadIdx.x; i < (kpad / 2) * kCB; i += 256) {
    const int jp = i >> 6, ch = i & (kCB - 1);
    float v0 = 0.f, v1 = 0.f;
    if (c0 + ch < C) {
      const float* wc = w + (int64_t)(c0 + ch) * k;
      const int j0 = 2 * jp, j1 = 2 * jp + 1;
      if (j0 < k) v0 = wc[flip ? (k - 1 - j0) : j0];
      if (j1 < k) v1 = wc[flip ? (k - 1 - j1) : j1];
    }
    s_w2[i] = (uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16);
  }
  __syncthreads();

  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int c = c0 + cl * 4;
  float acc[kR][4];
#pragma unroll
  for (int r = 0; r < kR; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
  const char* xs = reinterpret_cast<const char*>(s_x + (size_t)(tl * kR) * P::kTileLd + cl * 4);
  constexpr int LD = P::kTileLd * 2;   // bytes per LDS row
  const uint32_t* ws = s_w2 + cl * 4;
  uint32_t pw[8][4];
  uint2 prev = *reinterpret_cast<const uint2*>(xs);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const uint2 nx = *reinterpret_cast<const uint2*>(xs + (u + 1) * LD);
    pw[u][0] = pair_lo(prev.x, nx.x); pw[u][1] = pair_hi(prev.x, nx.x);
    pw[u][2] = pair_lo(prev.y, nx.y); pw[u][3] = pair_hi(prev.y, nx.y);
    prev = nx;
  }
  // prev = row tb+8
  const int npair = kpad / 2;
  for (int jp0 = 0; jp0 < npair; jp0 += 4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int jp = jp0 + q;
      const uint4 wv = *reinterpret_cast<const uint4*>(ws + jp * kCB);
      const uint2 ra = *reinterpret_cast<const uint2*>(xs + (2 * jp + 9) * LD);
      const uint2 rb = *reinterpret_cast<const uint2*>(xs + (2 * jp + 10) * LD);
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int sl = (r + 2 * q) & 7;
        acc[r][0] = dot2_bf16(pw[sl][0], wv.x, acc[r][0]);
        acc[r][1] = dot2_bf16(pw[sl][1], wv.y, acc[r][1]);
        acc[r][2] = dot2_bf16(pw[sl][2], wv.z, acc[r][2]);
        acc[r][3] = dot2_bf16(pw[sl][3], wv.w, acc[r][3]);
      }
      const int s0 = (2 * q) & 7, s1 = (2 * q + 1) & 7;
      pw[s0][0] = pair_lo(prev.x, ra.x); pw[s0][1] = pair_hi(prev.x, ra.x);
      pw[s0][2] = pair_lo(prev.y, ra.y); pw[s0][3] = pair_hi(prev.y, ra.y);
      pw[s1][0] = pair_lo(ra.x, rb.x); pw[s1][1] = pair_hi(ra.x, rb.x);
      pw[s1][2] = pair_lo(ra.y, rb.y); pw[s1][3] = pair_hi(ra.y, rb.y);
      prev = rb;
    }
  }
  if (c >= C) return;
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    const int64_t t = (int64_t)t0 + tl * kR + r;
    if (t < Tlen) {
      const int64_t off = ((int64_t)b * Tlen + t) * C + c;
      float o[4] = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
      if (addend) {
        float a[4];
        Elem<bf16_t>::ld4(addend + off, a);
        o[0] += a[0]; o[1] += a[1]; o[2] += a[2]; o[3] += a[3];
      }
      Elem<bf16_t>::st4(y + off, o);
    }
  }
}

// ---- bf16 stride-1 depthwise conv on the matrix cores ------------------------------------------------
// The VALU forms above are bound by issue rate (2*k FLOP per element: 15 us per 512-channel layer against
// 7 us of HBM traffic).  Per channel the convolution is a banded Toeplitz product,
//     y[16*blk + m] = sum_kap  A[m][kap] * img[16*blk + kap],    A[m][kap] = w'[kap - m - sh]
// (img = the channel's time series shifted by P = round-up-8(pad), sh = P - pad, w' = taps, reversed for the
// data gradient), i.e. D(16 x 16) += A(16 x 32) * B(32 x 16) on v_mfma_f32_16x16x32_bf16 with M = 16
// consecutive output frames, N = 16 frame blocks (256 frames per MFMA set) and K = the 15 + k + sh wide input
// window in steps of 32: 2-4 MFMAs per 256 outputs of a channel.  A is the same for every block of a
// channel: it is built once per channel from two packed tap tables (even / odd start) in LDS.
// What is left is data movement:
//   * B wants 8 consecutive FRAMES of one channel per lane, HBM has channels contiguous.  The tile is staged
//     as it is ([frame][channel], 16-byte writes), read back through ds_read_b64_tr_b16 (lane <- 4 frames of
//     one channel) and written frame-contiguous ([channel][frame], 8-byte writes): one LDS round trip, all
//     accesses conflict-free by the pitches chosen below; B fragments are then plain ds_read_b128.
//   * D leaves the MFMA with lane = (block, 4 consecutive frames) for ONE channel; a wave runs 8 channels
//     (one 16-byte octet) into 8 accumulator sets, so a lane ends up holding 8 channels x 4 frames and
//     stores 16 bytes per frame straight to HBM: no output transpose.  Lines are completed by the 8 waves.
// Workgroup = 64 channels x one utterance (512-frame time tiles), 512 threads, wave = channel octet.
namespace dwm {
static constexpr int TT = 512;            // output frames per time tile (2 MFMA sets of 256)
static constexpr int KWMAX = 128;         // input window per 16-frame block, max (k <= 101)
static constexpr int TIN = TT - 16 + KWMAX;   // 624 staged frames at most
static constexpr int LDI = 1296;          // bytes per channel row of the frame-contiguous image: a multiple of 16 (the B fragments are
                                          // 16-byte reads; at 1288 = 2 mod 32 dwords the 8-byte transposition writes were conflict-free but
                                          // every odd channel row misaligned them: phase 2 took 5.8 us instead of 2), writes 2-way
static constexpr int RS = 128;            // frames per staging round
static constexpr int LDST = 160;          // bytes per frame row of the staging image (40 dwords: see the tr-read banking)
static constexpr int IMG_BYTES = kCB * LDI;            // 82 944
static constexpr int STAGE_BYTES = RS * LDST;          // 20 480
static constexpr int WROW = 160;                       // dwords per channel: TE[0..79] | TO[0..79], packed bf16 tap pairs of the
                                                       // zero-padded row W[i] = w'[i - 24]: TE[i] = (W[2i], W[2i+1]), TO[i] = (W[2i+1], W[2i+2])
static constexpr int WSM_OFF = IMG_BYTES + STAGE_BYTES;
static constexpr int SMEM = WSM_OFF + kCB * WROW * 4;  // 144 384
static_assert(LDI >= TIN * 2 && LDI % 16 == 0, "image pitch");
}
// FUSE (forward only): the layer's input is not read from HBM but made here - x = act(y a + b [+ y2 a2 + b2]), the BatchNorm +
// residual add + activation of the unit BELOW (bn_act_fwd_kernel's arithmetic, operation for operation) - from that unit's y / y2
// and BN coefficients while the tile is staged; the tile's own rows are written to `out` on the way (the residual 1x1 conv and the
// backward read them), halo rows are recomputed by the neighbouring tile.  One launch and one 16 MB read per unit less.
struct DwBnIn {
  const bf16_t* y; const bf16_t* y2; const float* coef; const float* coef2; bf16_t* out; int act;
};
typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));
typedef short dw_s16x4 __attribute__((ext_vector_type(4)));

#ifdef LASR_DW_STAMPS
__device__ unsigned long long* g_dw_stamps = nullptr;   // debug builds only (tools/dw_stamps.py)
#define DW_STAMP(i_) do { if (g_dw_stamps && threadIdx.x == 0) g_dw_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i_)] = wall_clock64(); } while (0)
#else
#define DW_STAMP(i_) do {} while (0)
#endif
// NSET: 256-frame MFMA sets per time tile (2: one workgroup per 64 channels x 512 frames; 1: half tiles, twice the
// workgroups - for layers whose C/64 x B grid would leave CUs idle); gridDim.z workgroups share an utterance's tiles.
// (bx, by, bz) of (gx, B, gz): channel group, utterance, time-tile lane - the kernel's own grid or a slice of a fused grid
template <int NKS, int NSET, bool FUSE = false>
__device__ __forceinline__ void dwconv_s1_mfma_body(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                    const bf16_t* __restrict__ addend, bf16_t* __restrict__ y,
                                                    int Tlen, int C, int k, int flip, int bx, int by, int bz, int gz,
                                                    char* smem_raw, const uint32_t* __restrict__ taps, const DwBnIn bn = DwBnIn{}) {
  using namespace dwm;
  char* img = smem_raw;                               // [64 channels][LDI]: frame tau at byte 2*tau
  char* stage = smem_raw + IMG_BYTES;                 // [RS frames][LDST]
  uint32_t* wsm = reinterpret_cast<uint32_t*>(smem_raw + WSM_OFF);   // [64][WROW]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = by, c0 = bx * kCB;
  // the block's taps -> LDS once, as packed bf16 pair tables (reversed for the data gradient); a Toeplitz
  // fragment W[s .. s+7] is 4 consecutive dwords of TE (s even) or TO (s odd): no per-channel work in phase 2.
  // Loads are unconditional (clamped index, masked value) and all issued before the first conversion.
  if (taps) {   // workgroup-uniform: the step's precomputed tables (fused.h) - the workgroup's 64 rows are one contiguous 40 KB block
    static_assert(WROW == kDwTapRow, "tap table row");
    const uint4* src = reinterpret_cast<const uint4*>(taps + ((size_t)flip * C + c0) * WROW);
    uint4 tv[kCB * WROW / 4 / 512];
#pragma unroll
    for (int it = 0; it < kCB * WROW / 4 / 512; ++it) tv[it] = src[tid + 512 * it];
#pragma unroll
    for (int it = 0; it < kCB * WROW / 4 / 512; ++it) reinterpret_cast<uint4*>(wsm)[tid + 512 * it] = tv[it];
  } else {   // (kept in front of the tile's loads: issued behind them it cost 4 us more)
    constexpr int kIt = kCB * WROW / 512;             // 20
    float f0[kIt], f1[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int i = tid + 512 * it;
      const int ch = i / WROW, idx = i - ch * WROW;
      const int i0 = idx < 80 ? 2 * idx : 2 * (idx - 80) + 1;
      const int j0 = i0 - 24, j1 = i0 - 23;
      const float* wc = w + (size_t)min(c0 + ch, C - 1) * k;
      const bool cok = c0 + ch < C;
      const int q0 = min(max(j0, 0), k - 1), q1 = min(max(j1, 0), k - 1);
      const float a0 = wc[flip ? k - 1 - q0 : q0], a1 = wc[flip ? k - 1 - q1 : q1];
      // masked with bit operations, not `cond ? a : 0.f`: the compiler turns a select whose operand is a load into a branch
      // around the load (and waits for it at the join) - in the 32-channel backward kernel that was 17 memory round trips one
      // after the other before the first tile load was issued (round 4, read off the ISA)
      f0[it] = __uint_as_float(__float_as_uint(a0) & ((cok && j0 >= 0 && j0 < k) ? 0xffffffffu : 0u));
      f1[it] = __uint_as_float(__float_as_uint(a1) & ((cok && j1 >= 0 && j1 < k) ? 0xffffffffu : 0u));
    }
    __builtin_amdgcn_sched_barrier(0);                // every load above leaves before the first conversion below waits for one (left to
                                                      // itself the scheduler interleaves them: five batches, five round trips)
#pragma unroll
    for (int it = 0; it < kIt; ++it) wsm[tid + 512 * it] = (uint32_t)f32_to_bf16(f0[it]) | ((uint32_t)f32_to_bf16(f1[it]) << 16);
  }
  const int pad = k / 2, P = (pad + 7) & ~7, sh = P - pad;
  constexpr int KW = 32 * NKS;                        // = round-up-32(15 + k + sh) (host dispatch)
  const bf16_t* xb = FUSE ? nullptr : x + (size_t)b * Tlen * C;
  const int n16 = lane & 15, g4 = lane >> 4;          // MFMA lane coordinates
  // FUSE: the BN coefficients of the thread's channel octet (tid & 7: the same in every chunk it stages)
  float ca[8], cb[8], ca2[8], cb2[8];
  if (FUSE) {
    const int ccl = min(c0 + ((tid & 7) << 3), C - 8);
    auto ld8 = [](const float* p, float (&o)[8]) {
      const float4 u = *reinterpret_cast<const float4*>(p), q = *reinterpret_cast<const float4*>(p + 4);
      o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = u.w; o[4] = q.x; o[5] = q.y; o[6] = q.z; o[7] = q.w;
    };
    ld8(bn.coef + ccl, ca); ld8(bn.coef + C + ccl, cb);
    if (bn.y2) { ld8(bn.coef2 + ccl, ca2); ld8(bn.coef2 + C + ccl, cb2); }
  }

  DW_STAMP(0);
  constexpr int TTS = 256 * NSET;                     // output frames per time tile
  for (int tA = bz * TTS; tA < Tlen; tA += TTS * gz) {
    const int tin = TTS - 16 + KW;                     // staged frames: t = tA - P + tau, tau in [0, tin)
    // ---- phase 1: HBM -> staging ([frame][channel]) -> transposed image ([channel][frame]) ------------------
    // all global loads of the tile first (two 16-byte chunks per thread and round), then round by round
    constexpr int kRounds = (TTS - 16 + KWMAX + RS - 1) / RS;      // 5 (3 for half tiles)
    uint4 v[kRounds][2];
    if (!FUSE) {
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = tid + 512 * h;                 // chunk of the round: frame ch>>3, channel octet ch&7
          const int tau = r * RS + (ch >> 3);
          const int t = tA - P + tau;
          const int cc = c0 + ((ch & 7) << 3);
          const bool ok = tau < tin && t >= 0 && t < Tlen && cc < C;
          const uint4 ld = *reinterpret_cast<const uint4*>(xb + (size_t)min(max(t, 0), Tlen - 1) * C + min(cc, C - 8));
          const uint32_t mk = ok ? 0xffffffffu : 0u;
          v[r][h] = make_uint4(ld.x & mk, ld.y & mk, ld.z & mk, ld.w & mk);
        }
      }
    } else {
      // the same chunks of y (and y2) of the unit below; x is made from them in issue order, the tile's own frames leave for `out`
      uint4 ly[kRounds][2], lr[kRounds][2];
      const size_t ub = (size_t)b * Tlen * C;
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = tid + 512 * h;
          const int t = tA - P + r * RS + (ch >> 3);
          const size_t off = ub + (size_t)min(max(t, 0), Tlen - 1) * C + min(c0 + ((ch & 7) << 3), C - 8);
          ly[r][h] = Vec<bf16_t>::raw(bn.y + off);
          if (bn.y2) lr[r][h] = Vec<bf16_t>::raw(bn.y2 + off);
        }
      }
      with_act(bn.act, [&](auto act_c) {                // (the activation code as a constant: no branches per element)
      constexpr int act = decltype(act_c)::value;
#pragma unroll
      for (int r = 0; r < kRounds; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = tid + 512 * h;
          const int tau = r * RS + (ch >> 3);
          const int t = tA - P + tau;
          const int cc = c0 + ((ch & 7) << 3);
          const bool ok = tau < tin && t >= 0 && t < Tlen && cc < C;
          float yv[8], rv[8], o[8];
          Vec<bf16_t>::unpack(ly[r][h], yv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = fmaf(yv[j], ca[j], cb[j]);
          if (bn.y2) {                                  // workgroup-uniform
            Vec<bf16_t>::unpack(lr[r][h], rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += fmaf(rv[j], ca2[j], cb2[j]);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = act_fwd(o[j], act);
          const uint4 pk = Vec<bf16_t>::pack(o);
          if (ok && t >= tA && t < tA + TTS) *reinterpret_cast<uint4*>(bn.out + ub + (size_t)t * C + cc) = pk;
          const uint32_t mk = ok ? 0xffffffffu : 0u;
          v[r][h] = make_uint4(pk.x & mk, pk.y & mk, pk.z & mk, pk.w & mk);
        }
      }
      });
    }
    // the addend of the data gradient (same [frame][channel] tile as the output) is requested now, behind the tile's own
    // loads, and used in phase 3: its latency rides under the staging and the MFMAs (it was exposed once per half: ~3 us)
    uint4 ra[NSET][4];
    if (addend) {                                       // workgroup-uniform
#pragma unroll
      for (int ns = 0; ns < NSET; ++ns)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int ci = tid + 512 * it;              // chunk: frame ci >> 3 of the set, octet ci & 7
          const int t = tA + ns * 256 + (ci >> 3), cc = c0 + (ci & 7) * 8;
          ra[ns][it] = Vec<bf16_t>::raw(addend + ((size_t)b * Tlen + min(t, Tlen - 1)) * C + min(cc, C - 8));
        }
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      if (r * RS < tin) {                             // workgroup-uniform
        __syncthreads();                              // the previous round's (or tile's) readers are done
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = tid + 512 * h;
          *reinterpret_cast<uint4*>(stage + (ch >> 3) * LDST + ((ch & 7) << 4)) = v[r][h];
        }
        __syncthreads();
        // 16 frames x 16 channels per wave instruction: lanes 0-15 / 16-31 / 32-47 / 48-63 take the frame
        // quads 0, 4, 8, 12 of the block (a half-wave's two quads sit 4 rows = 160 dwords = 32 banks apart)
        typedef __attribute__((address_space(3))) dw_s16x4 lds_s4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int blk = wid * 4 + it;               // 32 blocks per round: 8 frame groups x 4 channel groups
          const int fb = (blk >> 2) * 16 + g4 * 4, cg = blk & 3;
          const int q = n16 >> 2, pp = n16 & 3;
          const dw_s16x4 d = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(stage + (fb + q) * LDST + (cg * 16 + pp * 4) * 2));
          const int tau = r * RS + fb;
          *reinterpret_cast<dw_s16x4*>(img + (cg * 16 + n16) * LDI + tau * 2) = d;
        }
      }
    }
    __syncthreads();
    DW_STAMP(1);

    // ---- phase 2: one channel octet per wave, 8 channels x 2 MFMA sets x 4 frames per lane --------------------
    dw_f32x4 acc[8][NSET];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch)
#pragma unroll
      for (int ns = 0; ns < NSET; ++ns) acc[ch][ns] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    // Round 5 (the depthwise backward's lesson, same ISA pattern: `s_waitcnt lgkmcnt(0..3)` right behind the reads of the step that is
    // about to run): the phase's 8 x NKS steps as ONE software pipeline - tap (A) and window (B) fragments kDF steps ahead through a
    // register ring, consecutive steps on different channels (step i -> channel i % 8, K step i / 8).  The BN-fused instantiation
    // (211-219 registers) has no room for the ring and keeps the per-channel form.
    constexpr bool kPipe = !FUSE;
    if constexpr (kPipe) {
      constexpr int NSF = 8 * NKS, kDF = 4;
      const int s0 = 8 * g4 - n16 - sh + 24;
      const uint32_t* wbase = wsm + wid * 8 * WROW + ((s0 & 1) ? 80 + ((s0 - 1) >> 1) : (s0 >> 1));
      const char* ibase = img + wid * 8 * LDI + (16 * n16 + 8 * g4) * 2;
      uint32_t ta[kDF][4];
      uint4 tb0[kDF], tb1[kDF];
      auto issue_f = [&](int i, uint32_t (&da)[4], uint4& d0, uint4& d1) {
        const int ch = i & 7, ks = i >> 3;
        const uint32_t* wr = wbase + ch * WROW + 16 * ks;
#pragma unroll
        for (int d = 0; d < 4; ++d) da[d] = wr[d];
        d0 = *reinterpret_cast<const uint4*>(ibase + ch * LDI + 64 * ks);
        if (NSET > 1) d1 = *reinterpret_cast<const uint4*>(ibase + ch * LDI + 512 + 64 * ks);
      };
#pragma unroll
      for (int i = 0; i < kDF; ++i) issue_f(i, ta[i], tb0[i], tb1[i]);
#pragma unroll
      for (int i = 0; i < NSF; ++i) {
        const int ch = i & 7;
        union { uint32_t u[4]; dw_bf16x8 v; } af, bf0, bf1;
#pragma unroll
        for (int d = 0; d < 4; ++d) af.u[d] = ta[i % kDF][d];
        bf0.u[0] = tb0[i % kDF].x; bf0.u[1] = tb0[i % kDF].y; bf0.u[2] = tb0[i % kDF].z; bf0.u[3] = tb0[i % kDF].w;
        if (NSET > 1) { bf1.u[0] = tb1[i % kDF].x; bf1.u[1] = tb1[i % kDF].y; bf1.u[2] = tb1[i % kDF].z; bf1.u[3] = tb1[i % kDF].w; }
        if (i + kDF < NSF) issue_f(i + kDF, ta[i % kDF], tb0[i % kDF], tb1[i % kDF]);
        acc[ch][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf0.v, acc[ch][0], 0, 0, 0);
        if (NSET > 1) acc[ch][NSET - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf1.v, acc[ch][NSET - 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);             // (without it the scheduler sinks the ring's reads back to one step ahead)
      }
    }
#pragma unroll
    for (int ch = 0; ch < (kPipe ? 0 : 8); ++ch) {
      const int cl = wid * 8 + ch;
      const char* row = img + cl * LDI;
      // A[m][kap] = W[kap - m - sh + 24]: lane (m = n16, K group g4) needs W[s0 + 32*ks .. +7], s0 = 8*g4 - m - sh + 24:
      // 4 dwords of the channel's even- or odd-start table.  Every LDS read of the channel is issued up front.
      const int s0 = 8 * g4 - n16 - sh + 24;
      const uint32_t* wrow = wsm + cl * WROW + ((s0 & 1) ? 80 + ((s0 - 1) >> 1) : (s0 >> 1));
      uint32_t wa[NKS][4];
      uint4 b0[NKS], b1[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[ks][i] = wrow[16 * ks + i];
        b0[ks] = *reinterpret_cast<const uint4*>(row + (16 * n16 + 32 * ks + 8 * g4) * 2);
        if (NSET > 1) b1[ks] = *reinterpret_cast<const uint4*>(row + (256 + 16 * n16 + 32 * ks + 8 * g4) * 2);
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        union { uint32_t u[4]; dw_bf16x8 v; } af, bf0, bf1;
#pragma unroll
        for (int i = 0; i < 4; ++i) af.u[i] = wa[ks][i];
        bf0.u[0] = b0[ks].x; bf0.u[1] = b0[ks].y; bf0.u[2] = b0[ks].z; bf0.u[3] = b0[ks].w;
        acc[ch][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf0.v, acc[ch][0], 0, 0, 0);
        if (NSET > 1) {
          bf1.u[0] = b1[ks].x; bf1.u[1] = b1[ks].y; bf1.u[2] = b1[ks].z; bf1.u[3] = b1[ks].w;
          acc[ch][NSET - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf1.v, acc[ch][NSET - 1], 0, 0, 0);
        }
      }
    }
    DW_STAMP(2);
    // ---- phase 3: lane = (block n16, frames 4*g4 .. +3) holds 8 channels (its wave's octet) per frame.  Each
    //      frame's 16 bytes go to an LDS image [frame][64 channels] (rows permuted so that the 16 blocks of a
    //      wave instruction fall on 16 consecutive rows: conflict-free 16-byte writes), then the tile leaves as
    //      whole 128-byte channel rows (8 lanes x 16 bytes), with the addend read the same way.
    // One 256-frame MFMA set at a time (the f32 form of a half tile, kept for the addend sum, fills the image space).
    constexpr int LDO = 144, LDOF = 272;              // row pitch: bf16 (36 dwords) / f32 (68 dwords), both = 4 mod 32
#pragma unroll
    for (int ns = 0; ns < NSET; ++ns) {
      __syncthreads();                                // every wave is done with what the space held before
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fl = n16 * 16 + g4 * 4 + r;         // frame within the set; physical row (fl & 15) * 16 + (fl >> 4)
        const int prow = (fl & 15) * 16 + (fl >> 4);
        float o[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) o[ch] = acc[ch][ns][r];
        if (!addend) {
          Vec<bf16_t>::store(reinterpret_cast<bf16_t*>(img + prow * LDO) + wid * 8, o);
        } else {                                      // the sum with the addend is rounded once, at the end
          *reinterpret_cast<float4*>(img + prow * LDOF + wid * 32) = make_float4(o[0], o[1], o[2], o[3]);
          *reinterpret_cast<float4*>(img + prow * LDOF + wid * 32 + 16) = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
      __syncthreads();
      // the four chunks of the thread: the addend loads first (branch-free: clamped offsets, a dummy source when
      // there is no addend), then the arithmetic and the predicated stores
      size_t offs[4];
      bool okc[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int ci = tid + 512 * it;                // chunk: frame ci >> 3 of the set, octet ci & 7
        const int fl = ci >> 3, oc = ci & 7;
        const int t = tA + ns * 256 + fl, cc = c0 + oc * 8;
        okc[it] = t < Tlen && cc < C;
        offs[it] = ((size_t)b * Tlen + min(t, Tlen - 1)) * C + min(cc, C - 8);
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int ci = tid + 512 * it;
        const int fl = ci >> 3, oc = ci & 7;
        const int prow = (fl & 15) * 16 + (fl >> 4);
        if (!addend) {
          if (okc[it]) *reinterpret_cast<uint4*>(y + offs[it]) = *reinterpret_cast<const uint4*>(img + prow * LDO + oc * 16);
        } else {
          const float4 lo = *reinterpret_cast<const float4*>(img + prow * LDOF + oc * 32);
          const float4 hi = *reinterpret_cast<const float4*>(img + prow * LDOF + oc * 32 + 16);
          float a8[8];
          Vec<bf16_t>::unpack(ra[ns][it], a8);
          float o[8] = {lo.x + a8[0], lo.y + a8[1], lo.z + a8[2], lo.w + a8[3], hi.x + a8[4], hi.y + a8[5], hi.z + a8[6], hi.w + a8[7]};
          if (okc[it]) Vec<bf16_t>::store(y + offs[it], o);
        }
      }
    }
  }
#ifdef LASR_DW_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
#endif
  DW_STAMP(3);
}

template <int NKS, int NSET>
__global__ __launch_bounds__(512, 1) void dwconv_s1_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                const bf16_t* __restrict__ addend, bf16_t* __restrict__ y,
                                                                int Tlen, int C, int k, int flip, const uint32_t* __restrict__ taps) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  dwconv_s1_mfma_body<NKS, NSET>(x, w, addend, y, Tlen, C, k, flip, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, smem_raw, taps);
}

// forward with the BN + add + activation of the unit below made in the staging loop (DwBnIn)
template <int NKS, int NSET>
__global__ __launch_bounds__(512, 1) void dwconv_s1_mfma_bn_kernel(DwBnIn bn, const float* __restrict__ w, bf16_t* __restrict__ y, int Tlen,
                                                                   int C, int k, const uint32_t* __restrict__ taps) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  dwconv_s1_mfma_body<NKS, NSET, true>(nullptr, w, nullptr, y, Tlen, C, k, 0, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, smem_raw, taps, bn);
}

// Weight gradient, stride 1.  Lane = 4 channels x 8 consecutive taps (group jg) x one time split;
// the 8 frames x[t+j0 .. t+j0+7] slide through a register window as t advances, so a frame costs
// two LDS vector reads (new x row, dy row) for 32 FMAs.  16 lane groups per workgroup are dealt as
// ceil(k/8) tap groups x ts time splits; the splits are summed through LDS at the end.
static constexpr int kSub = 32;  // frames per time split per LDS step

template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_s1_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                              float* __restrict__ partials, int64_t Tlen, int64_t C, int k,
                                                              int ts) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int pad = k / 2;
  const int ng = (k + 7) >> 3;          // tap groups
  const int kpad = ng * 8;
  const int WT = kSub * ts;             // frames per LDS step
  const int in_rows = WT + kpad + 8;
  using P = S1<T>;
  constexpr int LD = P::kTileLd;                                    // padded row pitch (144 B in bf16): conflict-free 8-byte reads
  T* s_x = reinterpret_cast<T*>(smem_raw);                         // [in_rows][LD]
  T* s_d = s_x + (size_t)in_rows * LD;                              // [WT][LD]
  const int b = blockIdx.z;
  const int64_t c0 = (int64_t)blockIdx.y * kCB;
  const int cl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int jg = grp % ng, sp = grp / ng;
  const bool active = sp < ts;
  const T* xb = x + (int64_t)b * Tlen * C;
  const T* db = dy + (int64_t)b * Tlen * C;
  float4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t tbeg = (int64_t)blockIdx.x * kWChunk;
  const int64_t tend = tbeg + kWChunk < Tlen ? tbeg + kWChunk : Tlen;
  for (int64_t t0 = tbeg; t0 < tend; t0 += WT) {
    __syncthreads();
    const int64_t in0 = t0 - pad;
    {
      // all 16-byte loads of the step first (x rows, then dy rows), then the LDS writes
      constexpr int kXP = (3 * kSub + kMaxK + 8 + P::kRowsPerPass - 1) / P::kRowsPerPass, kDP = (3 * kSub + P::kRowsPerPass - 1) / P::kRowsPerPass;
      const int lr = threadIdx.x / P::kLanesPerRow, lc = (threadIdx.x % P::kLanesPerRow) * P::kVec;
      const bool c_in = c0 + lc < C;
      uint4 vx[kXP], vd[kDP];
#pragma unroll
      for (int p = 0; p < kXP; ++p) {
        const int r = lr + p * P::kRowsPerPass;
        const int64_t ti = in0 + r;
        // unconditional load from a clamped address, value masked: no exec-mask branch (and vmcnt drain) per pass
        const uint4 ld = *reinterpret_cast<const uint4*>(xb + (int64_t)min(max(ti, (int64_t)0), Tlen - 1) * C + min(c0 + lc, C - P::kVec));
        const uint32_t mk = (r < in_rows && c_in && ti >= 0 && ti < Tlen) ? 0xffffffffu : 0u;
        vx[p] = make_uint4(ld.x & mk, ld.y & mk, ld.z & mk, ld.w & mk);
      }
#pragma unroll
      for (int p = 0; p < kDP; ++p) {
        const int r = lr + p * P::kRowsPerPass;
        const int64_t t = t0 + r;
        const uint4 ld = *reinterpret_cast<const uint4*>(db + (int64_t)min(t, Tlen - 1) * C + min(c0 + lc, C - P::kVec));
        const uint32_t mk = (r < WT && c_in && t < tend) ? 0xffffffffu : 0u;
        vd[p] = make_uint4(ld.x & mk, ld.y & mk, ld.z & mk, ld.w & mk);
      }
#pragma unroll
      for (int p = 0; p < kXP; ++p) {
        const int r = lr + p * P::kRowsPerPass;
        if (r < in_rows) *reinterpret_cast<uint4*>(s_x + (size_t)r * LD + lc) = vx[p];
      }
#pragma unroll
      for (int p = 0; p < kDP; ++p) {
        const int r = lr + p * P::kRowsPerPass;
        if (r < WT) *reinterpret_cast<uint4*>(s_d + (size_t)r * LD + lc) = vd[p];
      }
    }
    __syncthreads();
    if (active) {
      const T* xs = s_x + (size_t)(sp * kSub + jg * 8) * LD + cl * 4;   // frame t, tap j0 -> row t + j0
      const T* ds = s_d + (size_t)(sp * kSub) * LD + cl * 4;
      float4 win[8];
#pragma unroll
      for (int i = 0; i < 7; ++i) win[i] = lds_ld4<T>(xs + (size_t)i * LD);
      for (int tq = 0; tq < kSub; tq += 8) {
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          win[(tt + 7) & 7] = lds_ld4<T>(xs + (size_t)(tq + tt + 7) * LD);
          const float4 dv = lds_ld4<T>(ds + (size_t)(tq + tt) * LD);
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const float4 xv = win[(tt + jj) & 7];
            acc[jj].x = fmaf(dv.x, xv.x, acc[jj].x);
            acc[jj].y = fmaf(dv.y, xv.y, acc[jj].y);
            acc[jj].z = fmaf(dv.z, xv.z, acc[jj].z);
            acc[jj].w = fmaf(dv.w, xv.w, acc[jj].w);
          }
        }
      }
    }
  }
  // sum the time splits through LDS: s_red[sp][j][channel]
  __syncthreads();
  float* s_red = reinterpret_cast<float*>(smem_raw);
  if (active) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      *reinterpret_cast<float4*>(s_red + ((size_t)(sp * kpad + jg * 8 + jj)) * kCB + cl * 4) = acc[jj];
  }
  __syncthreads();
  float* out = partials + ((int64_t)b * gridDim.x + blockIdx.x) * C * k;
  for (int i = threadIdx.x; i < k * kCB; i += 256) {
    const int ch = i / k, j = i - ch * k;
    if (c0 + ch < C) {
      float s = 0.f;
      for (int q = 0; q < ts; ++q) s += s_red[((size_t)(q * kpad + j)) * kCB + ch];
      out[(c0 + ch) * k + j] = s;
    }
  }
}

// ---- bf16 stride-1 depthwise WEIGHT gradient on the matrix cores ---------------------------------------
// dW_c[j] = sum_b sum_t dY_c[b,t] * X_c[b, t + j - pad] is a correlation: per (channel, utterance) a matrix-VECTOR
// product, and the obvious matrix forms need an operand that depends on both output indices.  Re-indexing does it:
// with img[tau] = X[tau - P] (P = round-up-8(pad), sh = P - pad), j + sh = 16 n + m and u = t + m,
//     dW'[16 n + m] = sum_u  dY[u - m] * img[u + 16 n]   =   (A B)[m][n],   A[m][u] = dY[u - m],  B[u][n] = img[u + 16 n]
// A is a Toeplitz matrix of dY (no n), B a strided-window matrix of X (no m): one v_mfma_f32_16x16x32_bf16 per 32
// values of u covers 16 x NC taps (NC = ceil((k+sh)/16) <= 7 of the 16 columns are used - still ~10x the MAC rate of
// the VALU form, which ran at 68 % of its own bound).  Data movement as in the forward kernel: both tensors go
// through the transposing LDS round trip into frame-contiguous images; B fragments are aligned 16-byte reads, A
// fragments start at any element: five dwords from the even position below and a 16-bit funnel shift for odd starts.
// Workgroup = 64 channels x one utterance (u tiles of 288), 512 threads, wave = channel octet, accumulators live
// across the tiles; the per-utterance partials are summed by the caller (lasr_reduce_many / reduce_partials).
namespace dwg {
static constexpr int TU = 288;                          // u values per tile: 9 MFMA K steps
static constexpr int XF = TU + 16 * 7 + 8;              // 408 staged x frames  (lambda = u_local + 16 n)
static constexpr int DF = TU + 16;                      // 304 staged dY frames (tau = u_local - m + 16)
static constexpr int LDX = XF * 2;                      // 816 B
static constexpr int LDD = DF * 2 + 16;                 // 624 B: 304 frames + one spare dword pair for the funnel shift
static constexpr int RSG = 256;                         // frames per staging round (two rounds per image)
static constexpr int X_OFF = 0, D0_OFF = kCB * LDX;     // 52 224
static constexpr int ST_OFF = D0_OFF + kCB * LDD;       // 92 160
static constexpr int SMEM = ST_OFF + RSG * dwm::LDST;   // 133 120
static constexpr int XR = (XF + RSG - 1) / RSG, DR = (DF + RSG - 1) / RSG;   // 2 + 2 staging rounds
}

__device__ __forceinline__ void dwconv_wgrad_s1_mfma_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          float* __restrict__ partials, int Tlen, int C, int k, int bx, int by,
                                                          int bz, int gz, char* smem_raw) {
  using namespace dwg;
  char* ximg = smem_raw + X_OFF;
  char* d0 = smem_raw + D0_OFF;
  char* stage = smem_raw + ST_OFF;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = by, c0 = bx * kCB;
  const int pad = k / 2, P = (pad + 7) & ~7, sh = P - pad;
  const int n16 = lane & 15, g4 = lane >> 4;
  const bf16_t* xb = x + (size_t)b * Tlen * C;
  const bf16_t* db = dy + (size_t)b * Tlen * C;
  typedef __attribute__((address_space(3))) dw_s16x4 lds_s4;

  dw_f32x4 acc[8];
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) acc[ch] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
  // the spare dwords behind frame 303 of every dY row read as zero
  for (int i = tid; i < kCB * 4; i += 512) *reinterpret_cast<uint32_t*>(d0 + (i >> 2) * LDD + DF * 2 + (i & 3) * 4) = 0u;

  const int n_tiles = (Tlen + 15 + TU) / TU;            // u runs over [0, T + 15]
  DW_STAMP(0);
  // a tile's global loads (branch-free: clamped address, masked value); tile q+1's are issued before tile q's MFMAs
  uint4 v[XR + DR][4];
  auto issue_loads = [&](int q) {
#pragma unroll
    for (int r = 0; r < XR + DR; ++r) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int chk = tid + 512 * h;
        const bool isx = r < XR;
        const int fr = (isx ? r : r - XR) * RSG + (chk >> 3);         // frame index inside the image
        const int t = isx ? TU * q + fr - P : TU * q - 16 + fr;
        const int cc = c0 + ((chk & 7) << 3);
        const bool ok = fr < (isx ? XF : DF) && t >= 0 && t < Tlen && cc < C;
        const bf16_t* src = isx ? xb : db;
        const uint4 ld = *reinterpret_cast<const uint4*>(src + (size_t)min(max(t, 0), Tlen - 1) * C + min(cc, C - 8));
        const uint32_t mk = ok ? 0xffffffffu : 0u;
        v[r][h] = make_uint4(ld.x & mk, ld.y & mk, ld.z & mk, ld.w & mk);
      }
    }
  };
  // gridDim.z workgroups share an utterance's tiles (narrow layers: C/64 x B alone would leave CUs idle)
  const int zq = bz, zn = gz;
  issue_loads(zq);
  for (int q = zq; q < n_tiles; q += zn) {
    // ---- phase 1: round by round through the transposing staging
#pragma unroll
    for (int r = 0; r < XR + DR; ++r) {
      const bool isx = r < XR;
      const int rr = isx ? r : r - XR;
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int chk = tid + 512 * h;
        *reinterpret_cast<uint4*>(stage + (chk >> 3) * dwm::LDST + ((chk & 7) << 4)) = v[r][h];
      }
      __syncthreads();
      char* img = isx ? ximg : d0;
      const int ldi = isx ? LDX : LDD, nfr = isx ? XF : DF;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int blk = wid * 8 + it;                   // 64 blocks of 16 frames x 16 channels per round
        const int fb = (blk >> 2) * 16 + g4 * 4, cg = blk & 3;
        const int qq = n16 >> 2, pp = n16 & 3;
        const dw_s16x4 d = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(stage + (fb + qq) * dwm::LDST + (cg * 16 + pp * 4) * 2));
        const int fr = rr * RSG + fb;
        if (fr < nfr) *reinterpret_cast<dw_s16x4*>(img + (cg * 16 + n16) * ldi + fr * 2) = d;   // nfr is a multiple of 4
      }
    }
    __syncthreads();
    if (q == zq) DW_STAMP(1);
    if (q + zn < n_tiles) issue_loads(q + zn);
    if (q == zq) DW_STAMP(2);

    // ---- phase 2: 9 K steps per channel, 8 channels per wave
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int cl = wid * 8 + ch;
      // A[m][u] = dY[u - m]: elements e .. e+7 of the dY image, e = 32 ks + 8 g4 - m + 16 (m = n16).  e has the lane's
      // parity: five dwords from floor(e/2) and a funnel shift by 16 bits for the odd lanes (v_alignbit, 0 for even)
      const int e0 = 8 * g4 - n16 + 16;
      const uint32_t shft = (e0 & 1) * 16;
      const uint32_t* arow = reinterpret_cast<const uint32_t*>(d0 + cl * LDD) + (e0 >> 1);
      const char* brow = ximg + cl * LDX + (16 * n16 + 8 * g4) * 2;
#pragma unroll
      for (int kb = 0; kb < TU / 32; kb += 3) {
        uint32_t wa[3][5];
        uint4 bb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
          for (int i = 0; i < 5; ++i) wa[j][i] = arow[16 * (kb + j) + i];
          bb[j] = *reinterpret_cast<const uint4*>(brow + 64 * (kb + j));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          union { uint32_t u[4]; dw_bf16x8 v; } af, bf;
#pragma unroll
          for (int i = 0; i < 4; ++i) af.u[i] = __builtin_amdgcn_alignbit(wa[j][i + 1], wa[j][i], shft);
          bf.u[0] = bb[j].x; bf.u[1] = bb[j].y; bf.u[2] = bb[j].z; bf.u[3] = bb[j].w;
          acc[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf.v, acc[ch], 0, 0, 0);
        }
      }
    }
  }
  DW_STAMP(3);
  // ---- D[m][n] = dW'[16 n + m]: lane (n = n16, rows 4 g4 + r) -> tap j = 16 n + 4 g4 + r - sh
  float* out = partials + ((size_t)b * zn + zq) * C * k;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    const int c = c0 + wid * 8 + ch;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * n16 + 4 * g4 + r - sh;
      if (c < C && j >= 0 && j < k) out[(size_t)c * k + j] = acc[ch][r];
    }
  }
}

__global__ __launch_bounds__(512, 1) void dwconv_wgrad_s1_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                      float* __restrict__ partials, int Tlen, int C, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  dwconv_wgrad_s1_mfma_body(x, dy, partials, Tlen, C, k, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, smem_raw);
}

// Both consumers of a unit's d(depthwise output) in ONE launch: the weight gradient (x, dy -> per-utterance partials) and
// the data gradient (dy, flipped taps, + residual addend -> dx) are independent, so their workgroups share a grid (the
// weight-gradient ones first): one launch ramp less per unit, and the second kernel's workgroups start as the first's
// drain instead of behind a launch boundary.
struct DwBwd {
  const bf16_t* x; const bf16_t* dy; const float* w; const bf16_t* addend; bf16_t* dx; float* partials;
  int Tlen, C, k;
  int gx, B, gz_w, gz_d, n_w;     // channel groups, utterances, time lanes of either part, workgroups of the weight-gradient part
};
template <int NKS, int NSET>
__global__ __launch_bounds__(512, 1) void dwconv_bwd_s1_mfma_kernel(DwBwd a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int id = blockIdx.x;
  if (id < a.n_w) {               // workgroup-uniform
    const int bx = id % a.gx, by = (id / a.gx) % a.B, bz = id / (a.gx * a.B);
    dwconv_wgrad_s1_mfma_body(a.x, a.dy, a.partials, a.Tlen, a.C, a.k, bx, by, bz, a.gz_w, smem_raw);
  } else {
    id -= a.n_w;
    const int bx = id % a.gx, by = (id / a.gx) % a.B, bz = id / (a.gx * a.B);
    dwconv_s1_mfma_body<NKS, NSET>(a.dy, a.w, a.addend, a.dx, a.Tlen, a.C, a.k, 1, bx, by, bz, a.gz_d, smem_raw, nullptr);
  }
}


// ---- the two consumers of d(depthwise output) from ONE staging of it ---------------------------------------------------------
// dwconv_bwd_s1_mfma_kernel above runs the weight gradient (x, dy) and the data gradient (dy, flipped taps, + addend) as two
// kinds of workgroup: dy is fetched and pushed through the transposing LDS round trip twice, and 2 x (C/64) x B workgroups of
// 144 KB take two rounds on the chip.  Here ONE workgroup does both for its channels over 256-frame time tiles:
//   images (frame-contiguous, t = tA - P + tau, tau in [0, 384)):  D = dy,  X = x
//   weight gradient   dW'[16 n + m] += sum_u dY[u - m] X[u + 16 n - P],  u in [tA, tA + 256)  (+32 on the last tile: u runs to T + 15)
//                     A = Toeplitz fragments of D (funnel shift for odd starts), B = strided windows of X, accumulators live across tiles
//   data gradient     dx[tA + 16 blk + m] = sum_kap W'[kap - m - sh] D[16 blk + kap]          (A from the tap tables, B from D)
// so dy crosses HBM and the staging once, the tap tables are built once per workgroup, and (C/CB) x B x gz workgroups make one round.
// NW = waves per workgroup = channel octets: 8 (64 channels, whole 128-byte rows, one workgroup per CU) or 4 (32 channels,
// two workgroups per CU).  The next tile's global loads are issued before the MFMA phases of the current one.
// v_mov_b32 dpp row_shl:n - lane i of every 16-lane row takes the value of lane i + n of that row, lanes past the row's end read zero
// (bound_ctrl).  n is a compile-time constant after unrolling (the DPP control is an immediate).
__device__ __forceinline__ uint32_t dpp_row_shl(uint32_t v, int n) {
  switch (n) {
    case 0: return v;
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true);
    case 6: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x106, 0xf, 0xf, true);
    case 8: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xf, 0xf, true);
    default: __builtin_unreachable();
  }
}

namespace dwu {
static constexpr int TT = 256;            // frames per tile
static constexpr int FR = 384;            // staged frames per image and tile
static constexpr int LDI = 784;           // bytes per channel row: 196 dwords = 4 mod 32 (conflict-free 8-byte transposition writes), multiple of 16
static constexpr int RS = 128;            // frames per staging round
template <int NKS, int NW>
struct Cfg {
  static constexpr int CB = 8 * NW, NT = 64 * NW;
  static constexpr int LDST = NW == 8 ? 160 : 96;          // staging row pitch: 40 / 24 dwords (conflict-free tr reads and 16-byte writes)
  static constexpr int WROW = 2 * (16 * NKS + 16);          // dwords per channel of the tap tables: TE | TO, each 16 NKS + 16
  static constexpr int D_OFF = 0, X_OFF = CB * LDI, ST_OFF = 2 * CB * LDI, W_OFF = ST_OFF + RS * LDST;
  static constexpr int SMEM = W_OFF + CB * WROW * 4;
  static constexpr int LDO = CB * 2 + 16, LDOF = CB * 4 + 16;   // output image row pitch: bf16 / f32
  static_assert(TT * LDOF <= ST_OFF, "output image fits the two input images");
};
}

struct DwUni {
  const bf16_t* x; const bf16_t* dy; const float* w; const bf16_t* addend; bf16_t* dx; float* partials;
  int Tlen, C, k, gx, B, gz, total;
  const uint32_t* taps;   // the step's precomputed reversed-tap table of the layer ([C][kDwTapRow], fused.h) or null
};

template <int NKS, int NW, bool NTL = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void dwconv_bwd_uni_kernel(DwUni a) {
  using K = dwu::Cfg<NKS, NW>;
  using namespace dwu;
  constexpr int CB = K::CB, NT = K::NT, LDST = K::LDST, WROW = K::WROW, HALF = WROW / 2, CG = CB / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // workgroup id -> (channel group, utterance, time lane).  Blocks b and b + 8 share an XCD (round-robin placement, speed only):
  // the two 32-channel halves of a 128-byte row are dealt to one XCD, next to each other in its queue.
  int L;
  {
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    L = NW == 8 ? id : (((slot >> 1) * 8 + xcd) * 2 + (slot & 1));
  }
  if (L >= a.total) return;                              // workgroup-uniform (padded grid)
  const int bx = L % a.gx, by = (L / a.gx) % a.B, bz = L / (a.gx * a.B), gz = a.gz;
  char* dimg = smem_raw + K::D_OFF;
  char* ximg = smem_raw + K::X_OFF;
  char* stage = smem_raw + K::ST_OFF;
  uint32_t* wsm = reinterpret_cast<uint32_t*>(smem_raw + K::W_OFF);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int Tlen = a.Tlen, C = a.C, k = a.k;
  const int b = by, c0 = bx * CB;
  const int pad = k / 2, P = (pad + 7) & ~7, sh = P - pad;
  const int n16 = lane & 15, g4 = lane >> 4;
  const bf16_t* xb = a.x + (size_t)b * Tlen * C;
  const bf16_t* db = a.dy + (size_t)b * Tlen * C;
  const bf16_t* addend = a.addend;
  typedef __attribute__((address_space(3))) dw_s16x4 lds_s4;

  dw_f32x4 accw[8];
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) accw[ch] = dw_f32x4{0.f, 0.f, 0.f, 0.f};

  const int n_tiles = (Tlen + TT - 1) / TT;
  constexpr int kRounds = FR / RS;                       // 3 per image
  uint4 vd[kRounds][2], vx[kRounds][2], ra[4];
  // a tile's global loads: D and X chunks (frame ch / NW, octet ch % NW of each round) and the addend of its 256 output frames
  auto issue_loads = [&](int q) {
    const int tA = q * TT;
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ch = tid + NT * h;
        const int t = tA - P + r * RS + ch / NW;
        const int cc = c0 + (ch % NW) * 8;
        const size_t off = (size_t)min(max(t, 0), Tlen - 1) * C + min(cc, C - 8);
        // kept RAW: the out-of-range mask is applied when the chunk is staged (phase 1).  Masked here, the next tile's loads - issued
        // before the MFMA phases to travel under them - were waited for on the spot (round 4, read off the ISA: vmcnt(11) ... (0)
        // right behind the sixteen loads)
        vd[r][h] = *reinterpret_cast<const uint4*>(db + off);
        vx[r][h] = Vec<bf16_t>::raw_if_nt<NTL>(xb + off);            // the unit's forward input: written ~1.5 ms and > 2 GB of traffic ago
      }
    }
    if (addend) {                                        // workgroup-uniform
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int ci = tid + NT * it;
        const int t = tA + ci / NW, cc = c0 + (ci % NW) * 8;
        ra[it] = Vec<bf16_t>::raw(addend + ((size_t)b * Tlen + min(t, Tlen - 1)) * C + min(cc, C - 8));
      }
    }
  };

  issue_loads(bz);                                       // the first tile's loads travel while the tap tables are built
  // the workgroup's taps, reversed (data gradient), as packed bf16 pair tables: W[i] = w'[i - 24], TE[i] = (W[2i], W[2i+1]),
  // TO[i] = (W[2i+1], W[2i+2]); loads unconditional (clamped index, masked value), all issued before the first conversion
  if (a.taps) {   // workgroup-uniform: rows of the step's precomputed table (TE | TO of 80 each there, HALF each here)
    constexpr int kV = CB * WROW / 4 / NT;               // 16-byte groups per thread
    static_assert(CB * WROW / 4 % NT == 0 && HALF % 4 == 0 && HALF <= kDwTapRow / 2, "tap table copy");
    uint4 tv[kV];
#pragma unroll
    for (int it = 0; it < kV; ++it) {
      const int i4 = tid + NT * it;
      const int ch = i4 / (WROW / 4), idx = (i4 - ch * (WROW / 4)) * 4;
      tv[it] = *reinterpret_cast<const uint4*>(a.taps + (size_t)(c0 + ch) * kDwTapRow + (idx < HALF ? idx : kDwTapRow / 2 + idx - HALF));
    }
#pragma unroll
    for (int it = 0; it < kV; ++it) reinterpret_cast<uint4*>(wsm)[tid + NT * it] = tv[it];
  } else {
    constexpr int kIt = CB * WROW / NT;
    float f0[kIt], f1[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int i = tid + NT * it;
      const int ch = i / WROW, idx = i - ch * WROW;
      const int i0 = idx < HALF ? 2 * idx : 2 * (idx - HALF) + 1;
      const int j0 = i0 - 24, j1 = i0 - 23;
      const float* wc = a.w + (size_t)min(c0 + ch, C - 1) * k;
      const bool cok = c0 + ch < C;
      const int q0 = min(max(j0, 0), k - 1), q1 = min(max(j1, 0), k - 1);
      const float a0 = wc[k - 1 - q0], a1 = wc[k - 1 - q1];
      // masked with bit operations, not `cond ? a : 0.f`: the compiler turns a select whose operand is a load into a branch
      // around the load (and waits for it at the join) - in the 32-channel backward kernel that was 17 memory round trips one
      // after the other before the first tile load was issued (round 4, read off the ISA)
      f0[it] = __uint_as_float(__float_as_uint(a0) & ((cok && j0 >= 0 && j0 < k) ? 0xffffffffu : 0u));
      f1[it] = __uint_as_float(__float_as_uint(a1) & ((cok && j1 >= 0 && j1 < k) ? 0xffffffffu : 0u));
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) wsm[tid + NT * it] = (uint32_t)f32_to_bf16(f0[it]) | ((uint32_t)f32_to_bf16(f1[it]) << 16);
  }

  DW_STAMP(0);
  for (int q = bz; q < n_tiles; q += gz) {
    const int tA = q * TT;
    const bool last = q == n_tiles - 1;
    // ---- phase 1: registers -> staging ([frame][channel]) -> transposed images ([channel][frame]), D then X -------------
#pragma unroll
    for (int r2 = 0; r2 < 2 * kRounds; ++r2) {
      const bool isx = r2 >= kRounds;
      const int r = isx ? r2 - kRounds : r2;
      __syncthreads();                                   // the previous round's readers (or the previous tile's output pass) are done
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ch = tid + NT * h;
        const int t = tA - P + r * RS + ch / NW;
        const uint32_t mk = (t >= 0 && t < Tlen && c0 + (ch % NW) * 8 < C) ? 0xffffffffu : 0u;   // frames / channels outside the tensor: zeros
        const uint4 v = isx ? vx[r][h] : vd[r][h];
        *reinterpret_cast<uint4*>(stage + (ch / NW) * LDST + ((ch % NW) << 4)) = make_uint4(v.x & mk, v.y & mk, v.z & mk, v.w & mk);
      }
      __syncthreads();
      char* img = isx ? ximg : dimg;
      // (round 5: the four transposing reads issued together ahead of the four image writes - as written, read / write / read / write,
      //  the compiler keeps that order (an LDS write may alias the next read) and every block pays its own LDS round trip - was built
      //  and measured: this phase 2.94 -> 2.64 us per tile, the MFMA phases behind it 3.62 -> 3.89 and 1.40 -> 1.61 (the register
      //  allocation moved), step 2.0118 -> 2.0160 ms.  Left as it was.)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int blk = wid * 4 + it;                    // 8 frame groups x CG channel groups of 16 x 16 per round
        const int fb = (blk / CG) * 16 + g4 * 4, cg = blk % CG;
        const int qq = n16 >> 2, pp = n16 & 3;
        const dw_s16x4 d = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(stage + (fb + qq) * LDST + (cg * 16 + pp * 4) * 2));
        *reinterpret_cast<dw_s16x4*>(img + (cg * 16 + n16) * LDI + (r * RS + fb) * 2) = d;
      }
    }
    __syncthreads();
    const uint4 ra0 = ra[0], ra1 = ra[1], ra2 = ra[2], ra3 = ra[3];   // this tile's addend; the registers go to the next tile's loads
    if (q == bz) DW_STAMP(1);
    if (q + gz < n_tiles) issue_loads(q + gz);
    if (q == bz) DW_STAMP(2);

    // ---- phase 2a: weight gradient, 8 (9 on the last tile) K steps of 32 u values per channel ---------------------------
    const int nku = last ? 9 : 8;
    (void)nku;
#ifndef LASR_DW_NO_BFRAG_DPP
    {
      // Round 5, read off the phase stamps and the ISA: the phase took 4.7 us per tile whatever its LDS byte count was (36 or 24 bytes
      // per lane and MFMA: same time) - 156 cycles per MFMA: every K step's fragments were requested right before the step that used
      // them, i.e. every MFMA waited for an LDS round trip under the load of eight waves.  Now the 64 steps of a tile (8 channels x 8 K
      // steps; the ninth K step of the utterance's last tile follows behind) form ONE software pipeline: the Toeplitz (A) fragments
      // travel kDA steps ahead through a register ring, a channel pair's two B reads one pair ahead, and consecutive MFMAs alternate
      // between the two channels of a pair (no MFMA waits for its predecessor's accumulator).
      //   A[m][u] = dY[u - m]: elements e .. e+7 of the D image, e = P + 32 ks + 8 g4 - m (m = n16): five dwords from floor(e/2) and
      //   a funnel shift by 16 bits for the odd lanes.
      //   B[u][n] = X[u + 16 n]: the fragment of K step ks + s is the fragment of step ks moved by 2 s COLUMNS (32 u values = two
      //   16-frame column strides), and a column is a lane of the 16-lane DPP row.  Only NC = ceil((k + sh) / 16) <= 7 of the 16
      //   columns carry taps, so ONE 16-byte LDS read serves four K steps (columns 2 s .. 2 s + 6 <= 12), the others as `v_mov_b32
      //   dpp row_shl:2s` (lane i <- lane i + 2 s; lanes shifted in from outside the row read as zero - columns without taps).
      constexpr int kDA = 4, NST = 64;
      const int e0 = P + 8 * g4 - n16;
      const uint32_t shft = (e0 & 1) * 16;
      const uint32_t* abase = reinterpret_cast<const uint32_t*>(dimg + wid * 8 * LDI) + (e0 >> 1);
      const char* bbase = ximg + wid * 8 * LDI + (16 * n16 + 8 * g4) * 2;
      // step i -> (channel, K step): pairs of channels, the two channels of a pair alternate
      auto ch_of = [](int i) { return 2 * (i >> 4) + (i & 1); };
      auto ks_of = [](int i) { return (i >> 1) & 7; };
      uint32_t wa[kDA][5];
      uint4 bq[2][2][2];                                   // [pair parity][channel of the pair][K steps 0-3 | 4-7]
      auto issue_a = [&](int i, uint32_t (&dst)[5]) {
        const uint32_t* ar = abase + ch_of(i) * (LDI / 4) + 16 * ks_of(i);
#pragma unroll
        for (int d = 0; d < 5; ++d) dst[d] = ar[d];
      };
      auto issue_b = [&](int pair, uint4 (&dst)[2][2]) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const char* br = bbase + (2 * pair + c2) * LDI;
          dst[c2][0] = *reinterpret_cast<const uint4*>(br);
          dst[c2][1] = *reinterpret_cast<const uint4*>(br + 64 * 4);
        }
      };
      // (Also built: the NEXT tile's sixteen global loads - a 1.4 us burst in front of this phase, the CU's address path takes a
      //  16-byte-per-lane load at 64 B per clock - dealt over this sequence, one every four steps.  The compiler would not have it: with
      //  the loads' register arrays written from inside the unrolled sequence it either ran out of registers (16 address registers
      //  kept alive across the sequence: 256 + spills) or demoted the arrays to scratch.  Not measured; the burst stays.)
      issue_b(0, bq[0]);
#pragma unroll
      for (int i = 0; i < kDA; ++i) issue_a(i, wa[i]);
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const int ch = ch_of(i), ks = ks_of(i), pair = i >> 4;
        if ((i & 15) == 0 && pair + 1 < 4) issue_b(pair + 1, bq[(pair + 1) & 1]);
        union { uint32_t u[4]; dw_bf16x8 v; } af, bf;
#pragma unroll
        for (int d = 0; d < 4; ++d) af.u[d] = __builtin_amdgcn_alignbit(wa[i % kDA][d + 1], wa[i % kDA][d], shft);
        if (i + kDA < NST) issue_a(i + kDA, wa[i % kDA]);
        const uint4 base = bq[pair & 1][i & 1][ks >> 2];
        const int sft = 2 * (ks & 3);
        bf.u[0] = dpp_row_shl(base.x, sft); bf.u[1] = dpp_row_shl(base.y, sft);
        bf.u[2] = dpp_row_shl(base.z, sft); bf.u[3] = dpp_row_shl(base.w, sft);
        accw[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf.v, accw[ch], 0, 0, 0);
      }
      if (last) {                                          // workgroup-uniform: the ninth K step of the utterance's last tile
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
          const uint32_t* ar = abase + ch * (LDI / 4) + 16 * 8;
          const uint4 bb = *reinterpret_cast<const uint4*>(bbase + ch * LDI + 64 * 8);
          uint32_t w9[5];
#pragma unroll
          for (int d = 0; d < 5; ++d) w9[d] = ar[d];
          union { uint32_t u[4]; dw_bf16x8 v; } af, bf;
#pragma unroll
          for (int d = 0; d < 4; ++d) af.u[d] = __builtin_amdgcn_alignbit(w9[d + 1], w9[d], shft);
          bf.u[0] = bb.x; bf.u[1] = bb.y; bf.u[2] = bb.z; bf.u[3] = bb.w;
          accw[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf.v, accw[ch], 0, 0, 0);
        }
      }
    }
#else
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int cl = wid * 8 + ch;
      const int e0 = P + 8 * g4 - n16;
      const uint32_t shft = (e0 & 1) * 16;
      const uint32_t* arow = reinterpret_cast<const uint32_t*>(dimg + cl * LDI) + (e0 >> 1);
      const char* brow = ximg + cl * LDI + (16 * n16 + 8 * g4) * 2;
#pragma unroll
      for (int kb = 0; kb < 9; kb += 3) {
        uint32_t wa[3][5];
        uint4 bb[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
          for (int i = 0; i < 5; ++i) wa[j][i] = arow[16 * (kb + j) + i];
          bb[j] = *reinterpret_cast<const uint4*>(brow + 64 * (kb + j));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (kb + j < nku) {                            // workgroup-uniform
            union { uint32_t u[4]; dw_bf16x8 v; } af, bf;
#pragma unroll
            for (int i = 0; i < 4; ++i) af.u[i] = __builtin_amdgcn_alignbit(wa[j][i + 1], wa[j][i], shft);
            bf.u[0] = bb[j].x; bf.u[1] = bb[j].y; bf.u[2] = bb[j].z; bf.u[3] = bb[j].w;
            accw[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf.v, accw[ch], 0, 0, 0);
          }
        }
      }
    }
#endif
    if (q == bz) DW_STAMP(3);
    // ---- phase 2b: data gradient of the tile's 256 frames, NKS K steps per channel ----------------------------------------
    dw_f32x4 accd[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) accd[ch] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
    // (round 5: the weight gradient's software pipeline applied here as well - fragments six steps ahead, MFMAs alternating over the
    //  channels - was built and measured: this phase 1.40 -> 1.17 us per tile, but at 246 registers the allocator gave the weight
    //  gradient's ring up: THAT phase went 3.62 -> 6.85 us and the kernel 28.4 -> 33.9 us in the step.  Left per channel.)
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int cl = wid * 8 + ch;
      const char* row = dimg + cl * LDI;
      const int s0 = 8 * g4 - n16 - sh + 24;
      const uint32_t* wrow = wsm + cl * WROW + ((s0 & 1) ? HALF + ((s0 - 1) >> 1) : (s0 >> 1));
      uint32_t wa[NKS][4];
      uint4 b0[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[ks][i] = wrow[16 * ks + i];
        b0[ks] = *reinterpret_cast<const uint4*>(row + (16 * n16 + 32 * ks + 8 * g4) * 2);
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        union { uint32_t u[4]; dw_bf16x8 v; } af, bf0;
#pragma unroll
        for (int i = 0; i < 4; ++i) af.u[i] = wa[ks][i];
        bf0.u[0] = b0[ks].x; bf0.u[1] = b0[ks].y; bf0.u[2] = b0[ks].z; bf0.u[3] = b0[ks].w;
        accd[ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf0.v, accd[ch], 0, 0, 0);
      }
    }
    // ---- phase 3: lane = (block n16, frames 4 g4 .. +3) holds its wave's 8 channels per frame: 16 bytes (bf16) or 32 (f32, when
    //      an addend is summed in and rounded once) per frame into an LDS image [frame][CB] with rows permuted for conflict-free
    //      writes, then the tile leaves as whole rows of 16-byte pieces
    if (q == bz) DW_STAMP(4);
    __syncthreads();                                     // every wave is done with the input images
    char* oimg = smem_raw;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int fl = n16 * 16 + g4 * 4 + r;
      const int prow = (fl & 15) * 16 + (fl >> 4);
      float o[8];
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) o[ch] = accd[ch][r];
      if (!addend) {
        Vec<bf16_t>::store(reinterpret_cast<bf16_t*>(oimg + prow * K::LDO) + wid * 8, o);
      } else {
        *reinterpret_cast<float4*>(oimg + prow * K::LDOF + wid * 32) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(oimg + prow * K::LDOF + wid * 32 + 16) = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
    __syncthreads();
    const uint4 rav[4] = {ra0, ra1, ra2, ra3};
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int ci = tid + NT * it;
      const int fl = ci / NW, oc = ci % NW;
      const int t = tA + fl, cc = c0 + oc * 8;
      const bool ok = t < Tlen && cc < C;
      const size_t off = ((size_t)b * Tlen + min(t, Tlen - 1)) * C + min(cc, C - 8);
      const int prow = (fl & 15) * 16 + (fl >> 4);
      if (!addend) {
        if (ok) *reinterpret_cast<uint4*>(a.dx + off) = *reinterpret_cast<const uint4*>(oimg + prow * K::LDO + oc * 16);
      } else {
        const float4 lo = *reinterpret_cast<const float4*>(oimg + prow * K::LDOF + oc * 32);
        const float4 hi = *reinterpret_cast<const float4*>(oimg + prow * K::LDOF + oc * 32 + 16);
        float a8[8];
        Vec<bf16_t>::unpack(rav[it], a8);
        float o[8] = {lo.x + a8[0], lo.y + a8[1], lo.z + a8[2], lo.w + a8[3], hi.x + a8[4], hi.y + a8[5], hi.z + a8[6], hi.w + a8[7]};
        if (ok) Vec<bf16_t>::store(a.dx + off, o);
      }
    }
    if (q == bz) DW_STAMP(5);
  }
#ifdef LASR_DW_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
#endif
  DW_STAMP(6);
  // ---- D[m][n] = dW'[16 n + m]: lane (n = n16, rows 4 g4 + r) -> tap j = 16 n + 4 g4 + r - sh
  float* out = a.partials + ((size_t)b * gz + bz) * C * k;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    const int c = c0 + wid * 8 + ch;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * n16 + 4 * g4 + r - sh;
      if (c < C && j >= 0 && j < k) out[(size_t)c * k + j] = accw[ch][r];
    }
  }
}

}  // namespace lasr

using namespace lasr;

static int64_t conv_out_len(int64_t Tin, int k, int stride) { return (Tin + 2 * (k / 2) - k) / stride + 1; }

extern "C" int lasr_dwconv_fwd(const void* x, const float* w, const void* addend, void* y, int dtype, int64_t B, int64_t Tin,
                               int64_t C, int k, int stride, int flip, void* stream) {
  LASR_CHECK_ARG(x && w && y, "lasr_dwconv_fwd: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_dwconv_fwd: bad dtype");
  LASR_CHECK_SHAPE(k >= 1 && k <= kMaxK && (k & 1) && (stride == 1 || stride == 2) && C % 4 == 0 && B > 0 && B < 65536 && Tin > 0,
                   "lasr_dwconv_fwd: k=%d stride=%d C=%lld", k, stride, (long long)C);
  LASR_CHECK_SHAPE(!(flip && stride != 1), "lasr_dwconv_fwd: flip needs stride 1");
  LASR_CHECK_SHAPE(stride != 1 || (C % (int64_t)(16 / dtype_size(dtype)) == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0),
                   "lasr_dwconv_fwd: the stride-1 kernel loads 16-byte channel vectors (C=%lld)", (long long)C);
  const int64_t Tout = conv_out_len(Tin, k, stride);
  const int in_rows = (kTT - 1) * stride + k;
  const size_t shmem = ((size_t)in_rows + k) * kCB * sizeof(float);
  dim3 grid((unsigned)cdiv(Tout, kTT), (unsigned)cdiv(C, kCB), (unsigned)B);
  const int tok = prof_begin(LASR_PROF_DWCONV, as_stream(stream), 2.0 * (double)B * Tout * C * k,
                             (double)B * (Tin + Tout * (addend ? 2 : 1)) * C * dtype_size(dtype));
  if (stride == 1) {
    const int kpad = (k + 7) & ~7;
    const size_t esz1 = dtype_size(dtype);
    const size_t sh1 = (size_t)kpad * kCB * esz1 + (size_t)(kTT + kpad + 8) * (kCB + (esz1 == 2 ? 8 : 0)) * esz1;
    if (dtype == LASR_F32) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_s1_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(dwconv_s1_kernel<float>, grid, dim3(256), sh1, as_stream(stream), (const float*)x, w, (const float*)addend,
                         (float*)y, Tin, C, k, flip);
    } else {
      static const bool fma_form = getenv("LASR_DWCONV_FMA") != nullptr;   // A/B switches: the f32-FMA and the dot2 VALU forms
      static const bool dot2_form = getenv("LASR_DWCONV_DOT2") != nullptr;
      const int padk = k / 2, shk = ((padk + 7) & ~7) - padk;
      if (!fma_form && !dot2_form && C % 8 == 0 && 15 + k + shk <= dwm::KWMAX && Tin < (1 << 30)) {
        const int nks = (15 + k + shk + 31) / 32;
        // one workgroup per (64 channels, utterance) and 512-frame tile fills the chip from C = 512 on (B = 32);
        // narrower layers take 256-frame half tiles spread over gridDim.z
        static const bool no_half = getenv("LASR_DWCONV_NO_HALF") != nullptr;
        const bool half = !no_half && Tin > 256 && cdiv(C, kCB) * B * cdiv(Tin, (int64_t)512) < 200;
        const dim3 gridm((unsigned)cdiv(C, kCB), (unsigned)B, half ? (unsigned)std::min<int64_t>(cdiv(Tin, (int64_t)256), 8) : 1u);
        const uint32_t* taps = dw_taps_for(w, flip, C);   // the step's precomputed tap tables, when the model call in progress made them
#define LASR_DWM2(N_, S_)                                                                                                    \
  do {                                                                                                                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_s1_mfma_kernel<N_, S_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((dwconv_s1_mfma_kernel<N_, S_>), gridm, dim3(512), dwm::SMEM, as_stream(stream), (const bf16_t*)x, w,  \
                       (const bf16_t*)addend, (bf16_t*)y, (int)Tin, (int)C, k, flip, taps);                                  \
  } while (0)
#define LASR_DWM(N_) do { if (half) LASR_DWM2(N_, 1); else LASR_DWM2(N_, 2); } while (0)
        if (nks == 1) LASR_DWM(1); else if (nks == 2) LASR_DWM(2); else if (nks == 3) LASR_DWM(3); else LASR_DWM(4);
#undef LASR_DWM
#undef LASR_DWM2
      } else if (fma_form) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_s1_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(dwconv_s1_kernel<bf16_t>, grid, dim3(256), sh1, as_stream(stream), (const bf16_t*)x, w,
                           (const bf16_t*)addend, (bf16_t*)y, Tin, C, k, flip);
      } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_s1_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(dwconv_s1_bf16_kernel, grid, dim3(256), sh1, as_stream(stream), (const bf16_t*)x, w, (const bf16_t*)addend,
                           (bf16_t*)y, Tin, C, k, flip);
      }
    }
  } else {
    // 64-frame tiles when 128-frame tiles would not fill the chip (first_cnn: 64 channels, stride 2)
    const bool small = (int64_t)grid.x * grid.y * grid.z < 200;
    const dim3 g2 = small ? dim3((unsigned)cdiv(Tout, 64), grid.y, grid.z) : grid;
#define LASR_DWF(T_, R_)                                                                                                          \
  do {                                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_fwd_kernel<T_, R_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((dwconv_fwd_kernel<T_, R_>), g2, dim3(256), shmem, as_stream(stream), (const T_*)x, w, (const T_*)addend, (T_*)y, Tin,  \
                       Tout, C, k, stride, flip);                                                                                 \
  } while (0)
    if (dtype == LASR_F32) { if (small) LASR_DWF(float, 4); else LASR_DWF(float, 8); }
    else { if (small) LASR_DWF(bf16_t, 4); else LASR_DWF(bf16_t, 8); }
#undef LASR_DWF
  }
  prof_end(tok, as_stream(stream));
  LASR_LAUNCH_CHECK("dwconv_fwd_kernel");
  return 0;
}

int lasr::dwconv_fwd_bn(const void* y, const float* coef, const void* y2, const float* coef2, int act, const float* w, void* out, void* u,
                        int64_t B, int64_t T, int64_t C, int k, void* stream) {
  LASR_CHECK_ARG(y && coef && w && out && u && (!y2 || coef2), "dwconv_fwd_bn: null pointer");
  static const bool off = getenv("LASR_DWCONV_FMA") || getenv("LASR_DWCONV_DOT2");
  const int padk = k / 2, shk = ((padk + 7) & ~7) - padk;
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (off || k < 1 || k > kMaxK || !(k & 1) || C % 8 != 0 || C < 8 || 15 + k + shk > dwm::KWMAX || T <= 0 || T >= (1 << 30) || B <= 0 ||
      B >= 65536 || !al16(y) || !al16(y2) || !al16(out) || !al16(u) || !al16(coef) || !al16(coef2))
    return 1;
  const int nks = (15 + k + shk + 31) / 32;
  static const bool no_half = getenv("LASR_DWCONV_NO_HALF") != nullptr;
  const bool half = !no_half && T > 256 && cdiv(C, kCB) * B * cdiv(T, (int64_t)512) < 200;     // (the grid of lasr_dwconv_fwd)
  // Full 512-frame tiles are one workgroup per CU in ONE round, every workgroup in the same phase at the same time: the second
  // input and the extra output are pure added HBM time there (measured at C = 512: 21.6 us against 13.7 + 7.6 for the two
  // launches, and the GEMM behind it starts into twice the write-back).  Half tiles run two staggered rounds: 13.2 against 9.5 + 7.6.
  static const bool full_too = getenv("LASR_BN_DW_FUSE") && atoi(getenv("LASR_BN_DW_FUSE")) == 2;
  if (!half && !full_too) return 1;
  const dim3 gridm((unsigned)cdiv(C, kCB), (unsigned)B, half ? (unsigned)std::min<int64_t>(cdiv(T, (int64_t)256), 8) : 1u);
  DwBnIn bn;
  bn.y = (const bf16_t*)y; bn.y2 = (const bf16_t*)y2; bn.coef = coef; bn.coef2 = coef2; bn.out = (bf16_t*)out; bn.act = act;
  const int tok = prof_begin(LASR_PROF_DWCONV, as_stream(stream), 2.0 * (double)B * T * C * k, (double)B * T * C * (y2 ? 4 : 3) * 2);
  const uint32_t* taps = dw_taps_for(w, 0, C);
#define LASR_DWMB2(N_, S_)                                                                                                  \
  do {                                                                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_s1_mfma_bn_kernel<N_, S_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((dwconv_s1_mfma_bn_kernel<N_, S_>), gridm, dim3(512), dwm::SMEM, as_stream(stream), bn, w, (bf16_t*)u, (int)T, (int)C, k, taps); \
  } while (0)
#define LASR_DWMB(N_) do { if (half) LASR_DWMB2(N_, 1); else LASR_DWMB2(N_, 2); } while (0)
  if (nks == 1) LASR_DWMB(1); else if (nks == 2) LASR_DWMB(2); else if (nks == 3) LASR_DWMB(3); else LASR_DWMB(4);
#undef LASR_DWMB
#undef LASR_DWMB2
  prof_end(tok, as_stream(stream));
  LASR_LAUNCH_CHECK("dwconv_s1_mfma_bn_kernel");
  return 0;
}

extern "C" size_t lasr_dwconv_wgrad_workspace_bytes(int64_t B, int64_t Tout, int64_t C, int k) {
  return (size_t)B * cdiv(Tout, kWChunkG) * C * k * sizeof(float);   // the finer of the two chunkings
}

static int dwconv_wgrad_impl(const void* x, const void* dy, float* dw, int dtype, int64_t B, int64_t Tin, int64_t C, int k,
                            int stride, void* workspace, size_t workspace_bytes, void* stream, int* n_partials_out);

extern "C" int lasr_dwconv_wgrad(const void* x, const void* dy, float* dw, int dtype, int64_t B, int64_t Tin, int64_t C, int k,
                                 int stride, void* workspace, size_t workspace_bytes, void* stream) {
  LASR_CHECK_ARG(dw, "lasr_dwconv_wgrad: null pointer");
  return dwconv_wgrad_impl(x, dy, dw, dtype, B, Tin, C, k, stride, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int lasr_dwconv_wgrad_partials(const void* x, const void* dy, int dtype, int64_t B, int64_t Tin, int64_t C, int k,
                                          int stride, void* workspace, size_t workspace_bytes, int* n_partials, void* stream) {
  LASR_CHECK_ARG(n_partials, "lasr_dwconv_wgrad_partials: null pointer");
  return dwconv_wgrad_impl(x, dy, nullptr, dtype, B, Tin, C, k, stride, workspace, workspace_bytes, stream, n_partials);
}

static int dwconv_wgrad_impl(const void* x, const void* dy, float* dw, int dtype, int64_t B, int64_t Tin, int64_t C, int k,
                             int stride, void* workspace, size_t workspace_bytes, void* stream, int* n_partials_out) {
  LASR_CHECK_ARG(x && dy && workspace, "lasr_dwconv_wgrad: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_dwconv_wgrad: bad dtype");
  LASR_CHECK_SHAPE(k >= 1 && k <= kMaxK && (k & 1) && (stride == 1 || stride == 2) && C % 4 == 0 && B > 0 && B < 65536 && Tin > 0,
                   "lasr_dwconv_wgrad: k=%d stride=%d C=%lld", k, stride, (long long)C);
  const int64_t Tout = conv_out_len(Tin, k, stride);
  if (workspace_bytes < lasr_dwconv_wgrad_workspace_bytes(B, Tout, C, k)) return fail(LASR_E_WORKSPACE, "lasr_dwconv_wgrad: workspace");
  {
    static const bool valu_form = getenv("LASR_DWWGRAD_VALU") != nullptr;   // A/B switch: the VALU form below
    const int padk = k / 2, shk = ((padk + 7) & ~7) - padk;
    if (!valu_form && stride == 1 && dtype == LASR_BF16 && C % 8 == 0 && k + shk <= 16 * 7 && Tin < (1 << 30)) {
      float* parts = reinterpret_cast<float*>(workspace);   // [B * zsplit][C*k]: zsplit partials per utterance
      // one workgroup per (64 channels, utterance) fills the chip from C = 512 on; narrower layers split the time tiles
      const int n_tiles = (int)((Tin + 15 + dwg::TU) / dwg::TU);
      static const int zmax = getenv("LASR_DWWGRAD_ZSPLIT") ? atoi(getenv("LASR_DWWGRAD_ZSPLIT")) : 2;
      int zsplit = 1;
      while (zsplit < zmax && zsplit * 2 <= n_tiles && cdiv(C, kCB) * B * zsplit < 200) zsplit *= 2;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_s1_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(dwconv_wgrad_s1_mfma_kernel, dim3((unsigned)cdiv(C, kCB), (unsigned)B, (unsigned)zsplit), dim3(512), dwg::SMEM,
                         as_stream(stream), (const bf16_t*)x, (const bf16_t*)dy, parts, (int)Tin, (int)C, k);
      LASR_LAUNCH_CHECK("dwconv_wgrad_s1_mfma_kernel");
      if (n_partials_out) { *n_partials_out = (int)B * zsplit; return 0; }
      return launch_reduce_partials(parts, (int)B * zsplit, C * k, dw, C * k, nullptr, as_stream(stream));
    }
  }
  const int n_chunks = (int)cdiv(Tout, stride == 1 ? kWChunk : kWChunkG);
  const int in_rows = (kWT - 1) * stride + k;
  const size_t shmem = ((size_t)in_rows + kWT) * kCB * sizeof(float);
  dim3 grid((unsigned)n_chunks, (unsigned)cdiv(C, kCB), (unsigned)B);
  float* partials = reinterpret_cast<float*>(workspace);
  if (stride == 1) {
    const int ng = (k + 7) >> 3;
    int ts = 16 / ng;
    if (ts > 3) ts = 3;
    const int WT = kSub * ts, kpad = ng * 8;
    size_t sh1 = (size_t)(WT + kpad + 8 + WT) * (kCB + (dtype_size(dtype) == 2 ? 8 : 0)) * dtype_size(dtype);   // padded rows
    const size_t red = (size_t)ts * kpad * kCB * sizeof(float);
    if (red > sh1) sh1 = red;
    if (dtype == LASR_F32) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_s1_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(dwconv_wgrad_s1_kernel<float>, grid, dim3(256), sh1, as_stream(stream), (const float*)x, (const float*)dy,
                         partials, Tin, C, k, ts);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_s1_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(dwconv_wgrad_s1_kernel<bf16_t>, grid, dim3(256), sh1, as_stream(stream), (const bf16_t*)x, (const bf16_t*)dy,
                         partials, Tin, C, k, ts);
    }
  } else if (dtype == LASR_F32) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), shmem, as_stream(stream), (const float*)x, (const float*)dy,
                       partials, Tin, Tout, C, k, stride);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, grid, dim3(256), shmem, as_stream(stream), (const bf16_t*)x,
                       (const bf16_t*)dy, partials, Tin, Tout, C, k, stride);
  }
  LASR_LAUNCH_CHECK("dwconv_wgrad_kernel");
  const int64_t n = C * k;
  if (n_partials_out) { *n_partials_out = (int)(B * n_chunks); return 0; }   // the caller sums the slabs later (lasr_reduce_many)
  return launch_reduce_partials(partials, (int)(B * n_chunks), n, dw, n, nullptr, as_stream(stream));
}

#ifdef LASR_DW_STAMPS
extern "C" int lasr_debug_set_dw_stamps(void* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(lasr::g_dw_stamps), &p, sizeof(p));
}
#endif

extern "C" int lasr_dwconv_bwd_fused(const void* x, const void* dy, const float* w, const void* addend, void* dx, int dtype,
                                     int64_t B, int64_t T, int64_t C, int k, void* workspace, size_t workspace_bytes,
                                     int* n_partials, void* stream) {
  LASR_CHECK_ARG(x && dy && w && dx && workspace && n_partials, "lasr_dwconv_bwd_fused: null pointer");
  LASR_CHECK_ARG(dtype == LASR_F32 || dtype == LASR_BF16, "lasr_dwconv_bwd_fused: bad dtype");
  LASR_CHECK_SHAPE(k >= 1 && k <= kMaxK && (k & 1) && C % 4 == 0 && B > 0 && B < 65536 && T > 0, "lasr_dwconv_bwd_fused: k=%d C=%lld", k,
                   (long long)C);
  const int padk = k / 2, shk = ((padk + 7) & ~7) - padk;
  static const bool no_fused = getenv("LASR_DW_NO_FUSED_BWD") != nullptr;     // A/B switch: the two launches
  static const bool valu = getenv("LASR_DWWGRAD_VALU") || getenv("LASR_DWCONV_FMA") || getenv("LASR_DWCONV_DOT2");
  const bool ok = !no_fused && !valu && dtype == LASR_BF16 && C % 8 == 0 && 15 + k + shk <= dwm::KWMAX && k + shk <= 16 * 7 &&
                  T < (1 << 30) && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0 &&
                  workspace_bytes >= lasr_dwconv_wgrad_workspace_bytes(B, T, C, k);
  if (!ok) {
    LASR_TRY(lasr_dwconv_wgrad_partials(x, dy, dtype, B, T, C, k, 1, workspace, workspace_bytes, n_partials, stream));
    return lasr_dwconv_fwd(dy, w, addend, dx, dtype, B, T, C, k, 1, 1, stream);
  }
  const int nks = (15 + k + shk + 31) / 32;
  {
    // one workgroup for both consumers of dy (dwconv_bwd_uni_kernel): LASR_DW_UNI = 32 (default: 32-channel workgroups, two per CU;
    // k > 75 takes the 64-channel form), 64 (64-channel workgroups, one per CU), 0 (the two-kind grid below, for A/B runs).
    // Measured in the cfg2 step (rocprofv3, profiles/r03c_*): 36.9 / 38.1 / 37.8 us per 512-channel unit - see DESIGN.md
    static const int uni = getenv("LASR_DW_UNI") ? atoi(getenv("LASR_DW_UNI")) : 32;
    const bool u32 = uni == 32 && nks <= 3;
    if (uni == 64 || uni == 32) {
      DwUni u;
      u.x = (const bf16_t*)x; u.dy = (const bf16_t*)dy; u.w = w; u.addend = (const bf16_t*)addend; u.dx = (bf16_t*)dx;
      u.partials = reinterpret_cast<float*>(workspace);
      u.Tlen = (int)T; u.C = (int)C; u.k = k; u.B = (int)B;
      {
        const uint32_t* tp = dw_taps_for(w, 1, C);        // reversed taps: the second half of the layer's table
        u.taps = tp ? tp + (size_t)C * kDwTapRow : nullptr;
      }
      const int cb = u32 ? 32 : 64;
      u.gx = (int)cdiv(C, cb);
      const int n_tiles = (int)cdiv(T, (int64_t)dwu::TT);
      const int want = u32 ? 400 : 200;                  // workgroups that fill the chip (two / one per CU)
      int gz = 1;
      while (gz * 2 <= n_tiles && (int64_t)u.gx * B * gz < want) gz *= 2;
      u.gz = gz;
      u.total = (int)(u.gx * B * gz);
      const int tok = prof_begin(LASR_PROF_DWCONV, as_stream(stream), 4.0 * (double)B * T * C * k, (double)B * T * C * (addend ? 4 : 3) * 2);
      const bool ntl = (nt_loads_mask() & 4) != 0;
#define LASR_DWU(N_, W_) do { if (ntl) LASR_DWU2(N_, W_, true); else LASR_DWU2(N_, W_, false); } while (0)
#define LASR_DWU2(N_, W_, NT_)                                                                                                    \
  do {                                                                                                                            \
    using Kc = dwu::Cfg<N_, W_>;                                                                                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_bwd_uni_kernel<N_, W_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    const unsigned grid = (unsigned)((u.total + 15) / 16 * 16);                                                                     \
    hipLaunchKernelGGL((dwconv_bwd_uni_kernel<N_, W_, NT_>), dim3(grid), dim3(64 * W_), Kc::SMEM, as_stream(stream), u);             \
  } while (0)
      if (u32) { if (nks == 1) LASR_DWU(1, 4); else if (nks == 2) LASR_DWU(2, 4); else LASR_DWU(3, 4); }
      else { if (nks == 1) LASR_DWU(1, 8); else if (nks == 2) LASR_DWU(2, 8); else if (nks == 3) LASR_DWU(3, 8); else LASR_DWU(4, 8); }
#undef LASR_DWU
#undef LASR_DWU2
      prof_end(tok, as_stream(stream));
      LASR_LAUNCH_CHECK("dwconv_bwd_uni_kernel");
      *n_partials = (int)B * gz;
      return 0;
    }
  }
  DwBwd a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.w = w; a.addend = (const bf16_t*)addend; a.dx = (bf16_t*)dx;
  a.partials = reinterpret_cast<float*>(workspace);
  a.Tlen = (int)T; a.C = (int)C; a.k = k; a.gx = (int)cdiv(C, kCB); a.B = (int)B;
  // the same grid choices as the separate launches
  const int n_tiles = (int)((T + 15 + dwg::TU) / dwg::TU);
  static const int zmax = getenv("LASR_DWWGRAD_ZSPLIT") ? atoi(getenv("LASR_DWWGRAD_ZSPLIT")) : 2;
  int zsplit = 1;
  while (zsplit < zmax && zsplit * 2 <= n_tiles && cdiv(C, kCB) * B * zsplit < 200) zsplit *= 2;
  a.gz_w = zsplit;
  a.n_w = a.gx * a.B * a.gz_w;
  static const bool no_half = getenv("LASR_DWCONV_NO_HALF") != nullptr;
  const bool half = !no_half && T > 256 && cdiv(C, kCB) * B * cdiv(T, (int64_t)512) < 200;
  a.gz_d = half ? (int)std::min<int64_t>(cdiv(T, (int64_t)256), 8) : 1;
  const unsigned total = (unsigned)(a.n_w + a.gx * a.B * a.gz_d);
  constexpr int kSmem = dwm::SMEM > dwg::SMEM ? dwm::SMEM : dwg::SMEM;
  const int tok = prof_begin(LASR_PROF_DWCONV, as_stream(stream), 4.0 * (double)B * T * C * k, (double)B * T * C * (addend ? 5 : 4) * 2);
#define LASR_DWB2(N_, S_)                                                                                                      \
  do {                                                                                                                         \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_bwd_s1_mfma_kernel<N_, S_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    hipLaunchKernelGGL((dwconv_bwd_s1_mfma_kernel<N_, S_>), dim3(total), dim3(512), kSmem, as_stream(stream), a);                \
  } while (0)
#define LASR_DWB(N_) do { if (half) LASR_DWB2(N_, 1); else LASR_DWB2(N_, 2); } while (0)
  if (nks == 1) LASR_DWB(1); else if (nks == 2) LASR_DWB(2); else if (nks == 3) LASR_DWB(3); else LASR_DWB(4);
#undef LASR_DWB
#undef LASR_DWB2
  prof_end(tok, as_stream(stream));
  LASR_LAUNCH_CHECK("dwconv_bwd_s1_mfma_kernel");
  *n_partials = (int)B * zsplit;
  return 0;
}
